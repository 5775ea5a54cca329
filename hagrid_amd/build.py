"""Builds hagrid_amd/libhagrid_amd.so in-tree: every csrc/*.hip is compiled for gfx950 with hipcc and
the objects are linked WITHOUT naming a HIP runtime, so the library binds to whichever libamdhip64 the
process already holds (PyTorch's when loaded from Python, /opt/rocm's when linked into a C++ program).
csrc/kat/*.hip -- known-answer hooks and diagnostic instantiations, used by tests/ and tools/dev_*.py only --
become a second library, hagrid_amd/libhagrid_amd_kat.so, which links against the first.

Floating-point policy: no contraction, no fast-math, correctly rounded divide/sqrt -- the
kernels must agree bit for bit with the CPU oracle.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "obj")
LIB = os.path.join(HERE, "libhagrid_amd.so")
KAT_LIB = os.path.join(HERE, "libhagrid_amd_kat.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-DHOST=__host__", "-DDEVICE=__device__",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
    "-Wall", "-Wno-unused-function",
] + os.environ.get("HAGRID_HIPCC_EXTRA", "").split()     # experiments only (tools/dev_flags.sh)
if os.environ.get("HAGRID_DEBUG_SYNC", "0") not in ("", "0"):
    FLAGS.append("-DHAGRID_DEBUG_SYNC")                    # per-kernel synchronisation + error check (ctx.h: HG_DBG); use with --force


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def source_hash() -> str:
    """sha256 over the sources the libraries are built from (csrc/, include/, this recipe): what a profile or a traffic
    figure has to name to say which kernels it measured (bench.py: roofline.traffic_source)."""
    import hashlib
    files = sorted(glob.glob(os.path.join(CSRC, "**", "*.hip"), recursive=True) + glob.glob(os.path.join(CSRC, "**", "*.h"), recursive=True) +
                   glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True) + [os.path.abspath(__file__)])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode()); h.update(b"\0")
        h.update(open(f, "rb").read()); h.update(b"\0")
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "**", "*.h"), recursive=True) + glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True)
    hdrs.append(os.path.abspath(__file__))
    jobs = []

    def objects(srcs, prefix=""):
        objs = []
        for s in srcs:
            o = os.path.join(OBJ, prefix + os.path.basename(s)[:-4] + ".o")
            objs.append(o)
            if force or not _newer(o, [s] + hdrs):
                jobs.append([HIPCC, *FLAGS, "-c", s, "-o", o])
        return objs

    objs = objects(sorted(glob.glob(os.path.join(CSRC, "*.hip"))))
    kat_objs = objects(sorted(glob.glob(os.path.join(CSRC, "kat", "*.hip"))), "kat_")

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not _newer(LIB, objs):
        run(["g++", "-shared", "-fPIC", "-o", LIB, *objs, "-lpthread", "-ldl"])
    if jobs or force or not _newer(KAT_LIB, kat_objs + [LIB]):
        run(["g++", "-shared", "-fPIC", "-o", KAT_LIB, *kat_objs, "-L" + HERE, "-lhagrid_amd", "-Wl,-rpath,$ORIGIN", "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
