"""Two contexts, two HIP streams, two host threads at once: include/hagrid_amd.h promises that contexts are independent
(the reference keeps per-TU __constant__ state and allows one grid per process, traverse.cu:7-12).  Each thread builds its
own scene and traverses its own rays repeatedly; results must equal the ones of a quiet, single-context run."""
import threading

import numpy as np
import pytest

from hagrid_amd import scene

pytestmark = pytest.mark.gpu


def run_once(mem, tris, rays, rounds, out, key, barrier=None):
    from hagrid_amd import api
    try:
        d_tris = mem.upload(tris)
        d_rays = mem.upload(rays); d_hits = mem.alloc(16 * rays.shape[0])
        res = []
        for r in range(rounds):
            if barrier is not None:
                barrier.wait()
            grid = api.build_all(mem, d_tris, tris.shape[0], compress=bool(r & 1))
            api.setup_traversal(grid)
            api.traverse_grid(grid, d_tris, d_rays, d_hits, rays.shape[0])
            hits = mem.download(d_hits, api.HIT_DTYPE, rays.shape[0])
            d = grid.download()
            res.append((grid.summary(), int(d["entries"].astype(np.int64).sum()), int(d["ref_ids"].astype(np.int64).sum()), hits.copy()))
            grid.free()
        out[key] = res
    except Exception as e:          # surfaces in the main thread
        out[key] = e


def test_two_contexts_two_streams_concurrently():
    import torch
    from hagrid_amd import api
    scenes = [scene.make_soup(200_000), scene.make_soup(150_000, seed=99)]
    lo, hi = np.zeros(3, np.float32), np.ones(3, np.float32)
    rays = [scene.make_rays_primary(lo, hi, 512, 512), scene.make_rays_incoherent(lo, hi, 300_000, 5)]
    rounds = 4
    # quiet reference runs, one context at a time on the null stream
    quiet = {}
    for i in range(2):
        m = api.MemManager(keep=True)
        run_once(m, scenes[i], rays[i], rounds, quiet, i)
        m.close()
        assert not isinstance(quiet[i], Exception), quiet[i]
    # concurrent: each context on its own stream, driven by its own host thread
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    mems = [api.MemManager(keep=True), api.MemManager(keep=False)]
    for m, s in zip(mems, streams):
        m.use_stream(s.cuda_stream)
    busy = {}
    bar = threading.Barrier(2)
    threads = [threading.Thread(target=run_once, args=(mems[i], scenes[i], rays[i], rounds, busy, i, bar)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
        assert not t.is_alive()
    for i in range(2):
        assert not isinstance(busy[i], Exception), busy[i]
        for r in range(rounds):
            qs, qe, qr, qh = quiet[i][r]
            bs, be, br, bh = busy[i][r]
            assert qs == bs and qe == be and qr == br, (i, r)
            assert (qh["id"] == bh["id"]).all() and (qh["t"].view(np.uint32) == bh["t"].view(np.uint32)).all(), (i, r)
    for m in mems:
        m.close()


def test_set_stream_drains_the_old_stream():
    """Switching streams must not let work queued on the old stream race with re-used pool slots: traverse on
    stream A, switch to stream B, free + re-use buffers, then read the hits back -- identical to a synchronous run."""
    import torch
    from hagrid_amd import api
    tris = scene.make_soup(100_000)
    mem = api.MemManager(keep=True)
    d_tris = mem.upload(tris)
    grid = api.build_all(mem, d_tris, tris.shape[0])
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 1024, 1024)
    n = rays.shape[0]
    d_rays = mem.upload(rays); d_hits = mem.alloc(16 * n)
    api.setup_traversal(grid)
    api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    want = mem.download(d_hits, api.HIT_DTYPE, n)
    mem.zero(d_hits, 16 * n)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    mem.use_stream(a.cuda_stream)
    mem.set_ray_binning(1)                       # allocates and frees pool buffers around the launch
    for _ in range(5):
        api.traverse_grid(grid, d_tris, d_rays, d_hits, n)
    mem.use_stream(b.cuda_stream)                # drains stream a
    scratch = mem.alloc(64 << 20); mem.one(scratch, 64 << 20); mem.free(scratch)
    got = mem.download(d_hits, api.HIT_DTYPE, n)
    assert (got["id"] == want["id"]).all() and (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()
    mem.use_stream(None)
    mem.close()


def test_shared_traversal_image_two_batches_in_flight():
    """hagrid_share_traversal: a second context (own stream, own pool) traverses with the image the first one built.  Launches of
    the two contexts interleave on the GPU; every hit buffer must equal the quiet single-context result, and the borrower must
    neither free the image nor be allowed to release the grid."""
    import torch
    from hagrid_amd import api
    tris = scene.make_soup(200_000)
    a = api.MemManager(keep=True)
    b = api.MemManager(keep=True)
    with pytest.raises(api.HagridError):
        api.share_traversal(b, _grid_stub(a))                      # nothing to share yet
    d_tris = a.upload(tris)
    grid = api.build_all(a, d_tris, tris.shape[0])
    rays = [scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 768, 768),
            scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 768, 768, sample=1, num_samples=2)]
    n = rays[0].shape[0]
    api.setup_traversal(grid)
    # quiet results, context a alone
    want = []
    d_rays_a = a.upload(rays[0]); d_hits_a = a.alloc(16 * n)
    for r in rays:
        a.copy_h2d(d_rays_a, r)
        api.traverse_grid(grid, d_tris, d_rays_a, d_hits_a, n)
        want.append(a.download(d_hits_a, api.HIT_DTYPE, n).copy())
    a.copy_h2d(d_rays_a, rays[0])
    image_bytes = a.image_bytes(grid)
    used_b = b.usage()
    gb = api.share_traversal(b, grid)
    assert b.usage() == used_b and b.image_bytes(gb) == image_bytes            # no copy of the image in b's pool
    d_rays_b = b.upload(rays[1]); d_hits_b = b.alloc(16 * n)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    a.use_stream(sa.cuda_stream); b.use_stream(sb.cuda_stream)
    for _ in range(50):
        api.traverse_grid(grid, d_tris, d_rays_a, d_hits_a, n)
        api.traverse_grid(gb, d_tris, d_rays_b, d_hits_b, n)
    got_a = a.download(d_hits_a, api.HIT_DTYPE, n); got_b = b.download(d_hits_b, api.HIT_DTYPE, n)
    for got, w in ((got_a, want[0]), (got_b, want[1])):
        assert (got["id"] == w["id"]).all() and (got["t"].view(np.uint32) == w["t"].view(np.uint32)).all()
    with pytest.raises(api.HagridError):
        api.release_for_traversal(gb)                              # only the owner may give the construction format up
    # the borrower builds an image of its own: the share ends, the owner's image is untouched
    api.setup_traversal(gb)
    assert b.usage() >= used_b + image_bytes
    b.zero(d_hits_b, 16 * n)
    api.traverse_grid(gb, d_tris, d_rays_b, d_hits_b, n)
    got_b = b.download(d_hits_b, api.HIT_DTYPE, n)
    assert (got_b["id"] == want[1]["id"]).all()
    b.use_stream(None); b.close()
    a.zero(d_hits_a, 16 * n)
    api.traverse_grid(grid, d_tris, d_rays_a, d_hits_a, n)          # the owner still has its image after the borrower is gone
    got_a = a.download(d_hits_a, api.HIT_DTYPE, n)
    assert (got_a["id"] == want[0]["id"]).all() and a.image_bytes(grid) == image_bytes
    a.use_stream(None); a.close()


def _grid_stub(mem):
    from hagrid_amd import api
    g = api.Grid(); g.mem = mem
    return g


def test_borrowed_traversal_image_is_refused_once_its_owner_drops_it():
    """hagrid_share_traversal hands out the owner's image: once the owner sets traversal up again, runs a construction
    pass, frees a grid array or goes away, the borrower's next traverse_grid must fail with an error instead of reading pool memory
    that has been handed on; renewing the share (or an own setup_traversal) makes it work again."""
    from hagrid_amd import api
    tris = scene.make_soup(50_000)
    a = api.MemManager(keep=True)
    b = api.MemManager(keep=True)
    d_tris = a.upload(tris)
    grid = api.build_all(a, d_tris, tris.shape[0])
    rays = scene.make_rays_primary(grid.bbox_min, grid.bbox_max, 256, 256)
    n = rays.shape[0]
    d_rays_a = a.upload(rays); d_hits_a = a.alloc(16 * n)
    d_rays_b = b.upload(rays); d_hits_b = b.alloc(16 * n)
    api.setup_traversal(grid)
    api.traverse_grid(grid, d_tris, d_rays_a, d_hits_a, n)
    want = a.download(d_hits_a, api.HIT_DTYPE, n).copy()

    def borrower_ok(gb):
        b.zero(d_hits_b, 16 * n)
        api.traverse_grid(gb, d_tris, d_rays_b, d_hits_b, n)
        got = b.download(d_hits_b, api.HIT_DTYPE, n)
        return (got["id"] == want["id"]).all() and (got["t"].view(np.uint32) == want["t"].view(np.uint32)).all()

    gb = api.share_traversal(b, grid)
    assert borrower_ok(gb)
    # 1. the owner sets traversal up again: a new image, the old one is back in the owner's pool
    api.setup_traversal(grid)
    with pytest.raises(api.HagridError, match="dropped by its owner"):
        api.traverse_grid(gb, d_tris, d_rays_b, d_hits_b, n)
    gb = api.share_traversal(b, grid)                              # renewed
    assert borrower_ok(gb)
    # 2. a construction pass in the owner's context
    api.expand_grid(a, grid, d_tris, 1)
    with pytest.raises(api.HagridError, match="dropped by its owner"):
        api.traverse_grid(gb, d_tris, d_rays_b, d_hits_b, n)
    with pytest.raises(api.HagridError):
        api.share_traversal(b, grid)                               # the owner has no image now
    api.setup_traversal(grid)
    api.traverse_grid(grid, d_tris, d_rays_a, d_hits_a, n)
    want = a.download(d_hits_a, api.HIT_DTYPE, n).copy()           # (same hits: expansion never changes them)
    gb = api.share_traversal(b, grid)
    assert borrower_ok(gb)
    # 3. the borrower's own setup ends the share and is not affected by the owner any more
    api.setup_traversal(gb)
    api.setup_traversal(grid)
    assert borrower_ok(gb)
    # 4. the owner goes away altogether while a share is live
    gb = api.share_traversal(b, grid)
    assert borrower_ok(gb)
    a.close()
    with pytest.raises(api.HagridError, match="dropped by its owner"):
        api.traverse_grid(gb, d_tris, d_rays_b, d_hits_b, n)
    b.close()


def test_large_scans_of_two_contexts_while_traversal_keeps_the_cus_busy():
    """the single-pass scan must make progress when its launch is only partly resident -- two threads build
    large scenes (scans over millions of cells) while a third context keeps every CU busy with traversal launches."""
    import torch
    from hagrid_amd import api
    big = [scene.make_soup(1_000_000), scene.make_soup(800_000, seed=7)]
    quiet = []
    for t in big:
        m = api.MemManager(keep=True)
        g = api.build_all(m, m.upload(t), t.shape[0])
        d = g.download()
        quiet.append((g.summary(), int(d["entries"].astype(np.int64).sum()), int(d["ref_ids"].astype(np.int64).sum())))
        m.close()
    # the traffic: a context of its own traversing 2048^2 rays over and over on its own stream
    tm = api.MemManager(keep=True)
    small = scene.make_soup(200_000, seed=3)
    d_small = tm.upload(small)
    tg = api.build_all(tm, d_small, small.shape[0])
    api.setup_traversal(tg)
    rays = scene.make_rays_primary(tg.bbox_min, tg.bbox_max, 2048, 2048)
    d_rays = tm.upload(rays); d_hits = tm.alloc(16 * rays.shape[0])
    api.traverse_grid(tg, d_small, d_rays, d_hits, rays.shape[0])
    want = tm.download(d_hits, api.HIT_DTYPE, rays.shape[0]).copy()
    streams = [torch.cuda.Stream() for _ in range(3)]
    tm.use_stream(streams[2].cuda_stream)
    stop = threading.Event()
    out = {}

    def traffic():
        try:
            while not stop.is_set():
                for _ in range(8):
                    api.traverse_grid(tg, d_small, d_rays, d_hits, rays.shape[0])
                tm.synchronize()
            out["t"] = True
        except Exception as e:
            out["t"] = e

    def builder(i):
        try:
            m = api.MemManager(keep=True)
            m.use_stream(streams[i].cuda_stream)
            d_t = m.upload(big[i])
            res = []
            for _ in range(3):
                g = api.build_all(m, d_t, big[i].shape[0])
                d = g.download()
                res.append((g.summary(), int(d["entries"].astype(np.int64).sum()), int(d["ref_ids"].astype(np.int64).sum())))
                g.free()
            m.use_stream(None); m.close()
            out[i] = res
        except Exception as e:
            out[i] = e

    th = [threading.Thread(target=traffic)] + [threading.Thread(target=builder, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th[1:]:
        t.join(timeout=300)
        assert not t.is_alive(), "a construction did not finish while the device was shared: scan without forward progress?"
    stop.set(); th[0].join(timeout=60)
    assert not th[0].is_alive() and out["t"] is True, out.get("t")
    for i in range(2):
        assert not isinstance(out[i], Exception), out[i]
        for r in out[i]:
            assert r == quiet[i]
    got = tm.download(d_hits, api.HIT_DTYPE, rays.shape[0])
    assert (got["id"] == want["id"]).all()
    tm.use_stream(None); tm.close()
