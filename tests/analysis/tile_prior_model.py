"""ANALYSIS (CPU, oracle traces): can the cost of an 8 x 8 pixel tile be PREDICTED from the grid alone -- a walk of a few sample rays per tile through the TOP level only, summing a
per-top-level-cell "cells per crossing" k(T) weighted by the transmittance exp(-sum tau(T)) -- well enough to order the tiles of a first launch?
Ground truth: the cells of the longest ray of every tile on the oracle's traces.   python tests/analysis/tile_prior_model.py [soup|clustered|config3] [W]"""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hagrid_amd import scene
from oracle import oracle as O


def trace(G, tris, rays, cap=4):
    n = rays.shape[0]
    L = O.lib(); L.orc_traverse_trace.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.orc_traverse_trace.restype = None
    lens = np.zeros((n, cap), np.uint8); nc = np.zeros(n, np.int32); nids = np.zeros(n, np.int32)
    r = np.ascontiguousarray(rays, np.float32)
    L.orc_traverse_trace(C.byref(G.g), tris.ctypes.data, r.ctypes.data, n, cap, lens.ctypes.data, nc.ctypes.data, 0, None, nids.ctypes.data)
    return nc, nids


def top_tables(G, gamma):
    """k[T, a], tau[T, a]: expected cells entered / expected terminations of an axis-a line through top-level cell T"""
    dims = np.array(G.dims); s = G.shift; S = 1 << s
    cells = G.cells
    lo = np.stack([cells["min"][:, i] for i in range(3)], 1).astype(np.int64); hi = np.stack([cells["max"][:, i] for i in range(3)], 1).astype(np.int64)
    n = (cells["end"] - cells["begin"]).astype(np.float64)
    p = 1.0 - np.exp(-gamma * n)
    nT = int(dims.prod())
    k = np.zeros((nT, 3)); tau = np.zeros((nT, 3)); refs = np.zeros(nT)
    tlo = lo >> s; thi = (hi - 1) >> s
    span = thi - tlo + 1
    small = (span <= 3).all(1) & (hi > lo).all(1)
    print("cells", lo.shape[0], "spanning more than 3 top-level cells on an axis:", int((~small).sum()))
    for dx in range(3):
        for dy in range(3):
            for dz in range(3):
                d = np.array([dx, dy, dz])
                m = small & (span > d).all(1)
                if not m.any(): continue
                t = tlo[m] + d
                clo = np.maximum(lo[m], t * S); chi = np.minimum(hi[m], (t + 1) * S)
                e = (chi - clo).astype(np.float64) / S
                T = t[:, 0] + dims[0] * (t[:, 1] + dims[1] * t[:, 2])
                for a in range(3):
                    area = e[:, (a + 1) % 3] * e[:, (a + 2) % 3]
                    np.add.at(k[:, a], T, area); np.add.at(tau[:, a], T, area * p[m])
                np.add.at(refs, T, n[m] * e.prod(1) ** (1 / 3))
    return k, tau


def walk(G, rays, k, tau, iso=True):
    dims = np.array(G.dims); lo = np.asarray(G.bbox_min, np.float64); hi = np.asarray(G.bbox_max, np.float64)
    cs = (hi - lo) / dims
    o = rays[:, 0:3].astype(np.float64); d = rays[:, 4:7].astype(np.float64); tmin = rays[:, 3].astype(np.float64); tmax = rays[:, 7].astype(np.float64)
    inv = 1.0 / np.where(d == 0, 1e-30, d)
    t0 = (lo - o) * inv; t1 = (hi - o) * inv
    tn = np.minimum(t0, t1).max(1); tf = np.maximum(t0, t1).min(1)
    tn = np.maximum(tn, tmin); tf = np.minimum(tf, tmax)
    alive = tn < tf
    p = o + d * (tn[:, None] + 1e-9)
    c = np.clip(np.floor((p - lo) / cs), 0, dims - 1).astype(np.int64)
    step = np.where(d >= 0, 1, -1)
    nextb = lo + (c + (d >= 0)) * cs
    tnext = (nextb - o) * inv
    tdelta = np.abs(cs * inv)
    t = tn.copy(); est = np.zeros(len(rays)); trans = np.ones(len(rays)); steps = 0
    rate = np.abs(d) / cs                     # top-level cells per unit t, per axis
    while alive.any() and steps < 400:
        steps += 1
        a = np.argmin(tnext, 1); te = np.minimum(tnext[np.arange(len(a)), a], tf)
        dt = np.where(alive, np.maximum(te - t, 0), 0)
        T = c[:, 0] + dims[0] * (c[:, 1] + dims[1] * c[:, 2]); T = np.where(alive, T, 0)
        delta = dt[:, None] * rate
        if iso: kk = delta.sum(1) * k[T].mean(1); tt = delta.sum(1) * tau[T].mean(1)
        else: kk = (delta * k[T]).sum(1); tt = (delta * tau[T]).sum(1)
        est += trans * kk * np.where(tt > 1e-9, (1 - np.exp(-tt)) / np.maximum(tt, 1e-9), 1.0)          # cells entered before termination inside this cell
        trans *= np.exp(-tt)
        t = te
        idx = np.arange(len(a))
        c[idx, a] += np.where(alive, step[idx, a], 0); tnext[idx, a] += tdelta[idx, a]
        alive &= (te < tf) & (c >= 0).all(1) & (c < dims).all(1)
        c = np.clip(c, 0, dims - 1)
    return est, steps


def makespan(cost, order, slots=8192):
    """greedy: tiles in `order` onto `slots` wavefront slots, duration = cost"""
    import heapq
    h = [0.0] * slots
    for i in order:
        t = heapq.heappop(h); heapq.heappush(h, t + cost[i])
    return max(h)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "clustered"; W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024; H = W
    tris, params = {"soup": (scene.make_soup(1_000_000), {}), "clustered": (scene.make_clustered(), {}), "config3": (scene.make_soup(1_000_000), dict(top_density=0.15, snd_density=3.0)),
                    "gradient": (scene.make_gradient(), {}), "shell": (scene.make_shell(), {})}[which]
    t0 = time.time(); G = O.Grid.full(tris, **params); print(which, "grid", G.dims, "shift", G.shift, "cells", G.num_cells, "%.1f s" % (time.time() - t0))
    rays = scene.make_rays_primary(G.bbox_min, G.bbox_max, W, H)
    t0 = time.time(); nc, nids = trace(G, tris, rays); print("trace %.1f s" % (time.time() - t0), "cells/ray %.2f ids/ray %.2f" % (nc.mean(), nids.mean()))
    true = nc.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)).ravel().astype(np.float64)
    true2 = (nc + 0.5 * nids).reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)).ravel()
    ntile = true.size
    ys, xs = np.meshgrid(np.arange(H // 8), np.arange(W // 8), indexing="ij")
    for gamma in (0.15, 0.3, 0.6):
        k, tau = top_tables(G, gamma)
        for samples in (((2, 2), (5, 2), (2, 5), (5, 5)), ((0, 0), (7, 0), (0, 7), (7, 7), (3, 3))):
            ests = []
            for (sx, sy) in samples:
                idx = ((ys * 8 + sy) * W + xs * 8 + sx).ravel()
                e, steps = walk(G, rays[idx], k, tau, iso=True)
                ests.append(e)
            E = np.stack(ests, 1)
            for name, est in (("max", E.max(1)), ("mean", E.mean(1))):
                from scipy.stats import spearmanr
                rho = spearmanr(est, true).correlation
                top = max(1, ntile // 8)
                tset = set(np.argsort(-true)[:top]); eset = set(np.argsort(-est)[:top])
                ms_true = makespan(true, np.argsort(-true, kind="stable")); ms_est = makespan(true, np.argsort(-est, kind="stable")); ms_def = makespan(true, np.arange(ntile))
                print(f"gamma {gamma} samples {len(samples)} {name}: spearman {rho:.3f}  top-eighth overlap {len(tset & eset) / top:.3f}  makespan default {ms_def:.0f} est-order {ms_est:.0f} true-order {ms_true:.0f} (sum/slots {true.sum() / 8192:.0f}, max {true.max():.0f})  walk steps {steps}")
    np.savez("/tmp/prior/%s_%d.npz" % (which, W), true=true, true2=true2)


if __name__ == "__main__":
    main()
