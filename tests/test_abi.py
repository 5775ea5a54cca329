"""The C ABI library loads and exports every symbol include/hagrid_amd.h declares (no GPU needed)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from hagrid_amd import lib
    return lib


def declared_symbols(header=os.path.join(ROOT, "include", "hagrid_amd.h")):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hagrid_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built):
    L = built.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/hagrid_amd.h but not exported"
    assert sorted(built.SIGNATURES) == names, "hagrid_amd/lib.py signature table out of sync with the header"
    assert L.hagrid_abi_version() == built.ABI_VERSION == 3
    # the product library carries no test hooks (they live in libhagrid_amd_kat.so)
    import subprocess
    exported = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "hagrid_kat_" not in exported and "hagrid_kat_" not in open(os.path.join(ROOT, "include", "hagrid_amd.h")).read()


def test_kat_library_exports_its_header(built):
    """libhagrid_amd_kat.so (known-answer hooks, tests and dev tools only) against hagrid_amd/csrc/kat/hagrid_amd_kat.h."""
    K = built.load_kat()
    names = [n for n in declared_symbols(os.path.join(ROOT, "hagrid_amd", "csrc", "kat", "hagrid_amd_kat.h")) if n.startswith("hagrid_kat_")]
    assert len(names) >= 10 and sorted(built.KAT_SIGNATURES) == names
    for n in names:
        assert hasattr(K, n), n


def test_struct_layout_matches_header(built):
    """ctypes mirrors == what a C compiler makes of include/hagrid_amd.h (sizes and the offsets of a few members)."""
    import ctypes as C
    import subprocess, tempfile
    assert C.sizeof(built.GridPOD) == 4 * 8 + 6 * 4 + 3 * 4 + 5 * 4 + 32 * 4
    prog = r'''#include <stdio.h>
#include <stddef.h>
#include "hagrid_amd.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(hagrid_grid), sizeof(hagrid_traversal_stats), sizeof(hagrid_build_counts),
           offsetof(hagrid_grid, offsets), offsetof(hagrid_build_counts, level_refs), offsetof(hagrid_build_counts, merge_cells),
           offsetof(hagrid_build_counts, compress_refs_out), sizeof(hagrid_blob_header), offsetof(hagrid_blob_header, bbox_min),
           offsetof(hagrid_blob_header, off_entries));
    return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c"); exe = os.path.join(d, "s")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    B = built.BuildCounts
    want = [C.sizeof(built.GridPOD), C.sizeof(built.TraversalStats), C.sizeof(B), built.GridPOD.offsets.offset,
            B.level_refs.offset, B.merge_cells.offset, B.compress_refs_out.offset, C.sizeof(built.BlobHeader),
            built.BlobHeader.bbox_min.offset, built.BlobHeader.off_entries.offset]
    assert got == want


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hagrid_amd import api
    with pytest.raises(api.HagridError):
        api.MemManager()


def test_header_is_plain_c():
    """include/hagrid_amd.h compiles as C99 (what a cgo / JNI / ctypes author needs) and a C user of the whole ABI links
    against the library's exported names."""
    import subprocess, tempfile
    src = os.path.join(ROOT, "tests", "cpp", "c_abi_user.c")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", src], check=True)
    with tempfile.TemporaryDirectory() as d:
        obj = os.path.join(d, "u.o")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", obj], check=True)
        undefined = subprocess.run(["nm", "-u", obj], capture_output=True, text=True, check=True).stdout
        used = sorted(set(re.findall(r"\b(hagrid_[a-z0-9_]+)", undefined)))
        assert len(used) >= 20 and set(used) <= set(declared_symbols())


def test_product_library_kernel_count(built):
    """Variant sprawl stays pruned: at most 120 kernels in the product library's code objects (113 since round 6's pruning; tools/count_kernels.py reads the .kd symbols of every
    gfx950 code object in the library's fat binary), none of them a three-kernel scan form (those live in the test library)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "count_kernels.py"), "-v"], capture_output=True, text=True, check=True).stdout
    m = re.search(r"(\d+) kernels in", out)
    assert m and 40 <= int(m.group(1)) <= 120, out[-300:]
    assert "scan_spine" not in out.replace("scan_spine<int>(int*, int, int const*, int*)", "", 1), "the three-kernel scan belongs to the test library (compress.hip uses its halves once)"
