"""ANALYSIS (CPU, oracle): what the general slim traversal image (trav_image.hip, "general layout") would hold for a grid -- one 16-byte record per
voxel-map entry, bounds as byte offsets from the entry's own region, cells whose bounds do not fit a byte as WIDE records -- on the clustered scene
(shift 6) and on soups at higher second-level densities.  Counts entries by kind and, along primary rays, the share of cell steps that would restart
from the top level, descend through links or fetch a wide record.

    python tests/analysis/general_image_model.py [clustered | soup:N:SD]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hagrid_amd import scene
from oracle import oracle as O

what = sys.argv[1] if len(sys.argv) > 1 else "clustered"
if what == "clustered": tris = scene.make_clustered(); sd = 2.4
else: _, n, sd = what.split(":"); tris = scene.make_soup(int(n)); sd = float(sd)
t0 = time.time()
G = O.Grid.full(tris, snd_density=sd)
print("grid", G.summary(), f"({time.time() - t0:.1f} s)")
entries = G.entries.copy(); cells = G.cells.copy(); shift = G.shift; dims = G.dims
num_top = dims[0] * dims[1] * dims[2]
E = entries.shape[0]
org = np.zeros((E, 3), np.int32); s_of = np.full(E, -1, np.int32)
t = np.arange(num_top)
org[:num_top, 0] = (t % dims[0]) << shift; org[:num_top, 1] = ((t // dims[0]) % dims[1]) << shift; org[:num_top, 2] = (t // (dims[0] * dims[1])) << shift
s_of[:num_top] = shift
front = np.flatnonzero(entries[:num_top] & 3)
levels = 0
while front.size:
    levels += 1
    nxt = []
    for k in (1, 2, 3):
        sel = front[(entries[front] & 3) == k]
        if not sel.size: continue
        n = 1 << (3 * k); c = np.arange(n)
        child = (entries[sel] >> 2)[:, None] + c[None, :]
        s = s_of[sel] - k
        off = np.stack([c & ((1 << k) - 1), (c >> k) & ((1 << k) - 1), c >> (2 * k)], axis=1)        # (n, 3)
        org[child] = org[sel][:, None, :] + (off[None, :, :] << s[:, None, None])
        s_of[child] = s[:, None]
        ch = child.reshape(-1)
        nxt.append(ch[(entries[ch] & 3) != 0])
    front = np.concatenate(nxt) if nxt else np.zeros(0, np.int64)
assert (s_of >= 0).all()
leaf = (entries & 3) == 0
c = (entries[leaf] >> 2).astype(np.int64)
lo = np.stack([cells["min"][c, i] for i in range(3)], axis=1) if cells.dtype.names and "min" in cells.dtype.names else None
if lo is None:
    names = cells.dtype.names; print(names)
    raise SystemExit
hi = np.stack([cells["max"][c, i] for i in range(3)], axis=1)
o = org[leaf]
dl = o - lo; dh = hi - o
assert (dl >= 0).all() and (dh >= 0).all()
fits = ((dl <= 255) & (dh <= 255)).all(axis=1)
n_refs = (cells["end"][c] - cells["begin"][c])
print(f"entries {E} (top {num_top}), flattened levels below the top {levels}, leaf {leaf.sum()}, links {E - leaf.sum()}")
print(f"leaf entries whose cell does not fit byte offsets: {(~fits).sum()} ({(~fits).mean():.2%}); distinct such cells {np.unique(c[~fits]).size} of {cells.shape[0]}")
print(f"image: {16 * E / 1e6:.1f} MB + wide {16 * np.unique(c[~fits]).size / 1e6:.2f} MB; leaf entries by list length <=4: {(n_refs <= 4).mean():.2%}")
