import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order: the parity tests of the kernels first (the headline traversal kernel, the BASELINE configurations at full
# size, the scans, construction), then everything in-process, and the tests that start other processes (front-end binaries,
# torch.distributed launches) LAST -- under `pytest -x` a front-end fault must not hide a kernel-parity result.
_ORDER = ["test_traverse_gpu", "test_fullsize_gpu", "test_scan_gpu", "test_build_gpu", "test_concurrency_gpu",
          "test_oracle_golden", "test_scene", "test_abi", "test_obj_loader", "test_dist_cpu"]
_LAST = ["test_dist_gpu", "test_cpp_api"]


def pytest_collection_modifyitems(session, config, items):
    if os.environ.get("HAGRID_TEST_ORDER") == "alpha":          # tools/gpu_cli_exit_hunt.sh: the order of the round-2 driver run
        return
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _ORDER:
            return _ORDER.index(name)
        if name in _LAST:
            return 1000 + _LAST.index(name)
        return 500
    items.sort(key=key)                                          # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
