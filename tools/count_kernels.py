"""DEV TOOL: kernels per code object of a built library -- the .hip_fatbin section is cut at its bundle headers, every gfx950 code object is
read with llvm-readelf and its kernel descriptors (symbols ending in .kd) are counted.  usage: python tools/count_kernels.py [library] [-v]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = next((a for a in sys.argv[1:] if not a.startswith("-")), os.path.join(ROOT, "hagrid_amd", "libhagrid_amd.so"))
verbose = "-v" in sys.argv
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
data = open(lib, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
total, names = 0, []
for m in re.finditer(re.escape(MAGIC), data):
    base = m.start()
    n = int.from_bytes(data[base + 24:base + 32], "little")
    p = base + 32
    for _ in range(n):
        off, size, tlen = (int.from_bytes(data[p + 8 * i:p + 8 * i + 8], "little") for i in range(3))
        triple = data[p + 24:p + 24 + tlen].decode(); p += 24 + tlen
        if "gfx" not in triple or size == 0: continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(data[base + off:base + off + size]); f.flush()
            out = subprocess.run([READELF, "-s", "--wide", f.name], capture_output=True, text=True).stdout
        ks = sorted({l.split()[-1][:-3] for l in out.splitlines() if l.rstrip().endswith(".kd")})
        total += len(ks); names += ks
if verbose:
    for k in subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines(): print(k[:200])
print(f"{total} kernels in {os.path.relpath(lib, ROOT)}")
