"""DEV TOOL: registers, spills, scratch, LDS and occupancy of every kernel of one translation unit, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage with the product's flags; nothing is written).  usage: python tools/kernel_resources.py traverse [filter]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hagrid_amd import build as B
unit = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(B.CSRC, unit + ".hip") if not unit.endswith(".hip") else unit
r = subprocess.run([B.HIPCC, *B.FLAGS, "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None: cur[m.group(1).split(" [")[0]] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(x["name"] for x in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{len(rows)} kernels in {os.path.basename(src)}")
for x, n in zip(rows, names):
    if flt in n:
        n = re.sub(r"^void ", "", n); n = re.sub(r"\(hagrid_trav::TraverseArgs\)$", "", n)
        print(f"vgpr {x.get('VGPRs', 0):3d} spill {x.get('VGPRs Spill', 0):2d} scratch {x.get('ScratchSize', 0):3d} sgpr {x.get('SGPRs', 0):3d} lds {x.get('LDS Size', 0):5d} occ {x.get('Occupancy', 0)}  {n[:170]}")
