"""The synthetic generators (hagrid_amd/scene.py) are the bench's and the tests' inputs: pin their bits (integer PRNG,
no libm), their layouts and the slicing property multi-GPU runs rely on.  No GPU."""
import zlib

import numpy as np

from hagrid_amd import scene


def test_prng_and_buffers_are_bit_pinned():
    u = scene.uniform01(7, np.arange(4, dtype=np.uint64))
    assert np.allclose(u, [0.3898297, 0.01678824, 0.90076065, 0.58293027], atol=1e-7) and u.dtype == np.float32
    assert ((u * 2 ** 24) % 1 == 0).all()                         # exactly 24 random bits each
    assert zlib.crc32(scene.make_soup(1000).tobytes()) == 1861140402
    assert zlib.crc32(scene.make_rays_incoherent([0, 0, 0], [1, 2, 3], 1000, 42).tobytes()) == 3936610879
    assert zlib.crc32(scene.make_rays_primary([0, 0, 0], [1, 1, 1], 64, 32).tobytes()) == 1449931021
    assert zlib.crc32(scene.make_rays_primary([0, 0, 0], [1, 1, 1], 64, 32, sample=1, num_samples=4).tobytes()) == 4122960228


def test_soup_layout_matches_main_cpp_packing():
    t = scene.make_soup(5000)
    assert t.dtype == np.float32 and t.shape == (5000, 12)
    e1, e2 = t[:, 4:7], t[:, 8:11]
    n = np.stack([t[:, 3], t[:, 7], t[:, 11]], axis=1)
    want = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    assert np.allclose(n, want, rtol=1e-4, atol=1e-7)            # n = cross(e1, e2) spread over the w slots
    s = 5000 ** (-1 / 3)
    assert (np.abs(e1) <= s * 1.0001).all() and (np.abs(e2) <= s * 1.0001).all()
    assert (t[:, 0:3] >= 0).all() and (t[:, 0:3] < 1).all()


def test_any_slice_can_be_generated_independently():
    full = scene.make_soup(3000)
    assert (scene.make_soup(3000, first=1000, count=500) == full[1000:1500]).all()
    lo, hi = [0, 0, 0], [1, 1, 1]
    rays = scene.make_rays_incoherent(lo, hi, 5000, 9)
    assert (scene.make_rays_incoherent(lo, hi, 1200, 9, first=3000) == rays[3000:4200]).all()
    prim = scene.make_rays_primary(lo, hi, 100, 50)
    assert (scene.make_rays_primary(lo, hi, 100, 50, first=2500, count=700) == prim[2500:3200]).all()


def test_ray_conventions():
    lo, hi = np.float32([0, 0, 0]), np.float32([1, 1, 1])
    r = scene.make_rays_incoherent(lo, hi, 20000, 3)
    l2 = (r[:, 4:7].astype(np.float64) ** 2).sum(axis=1)
    assert (l2 > 0.01 * 0.999).all() and (l2 <= 1.0001).all()     # rejection sampling in the unit ball, un-normalised
    assert (r[:, 3] == 0).all() and (r[:, 7] == scene.FLT_MAX).all()
    assert (r[:, 0:3] >= lo).all() and (r[:, 0:3] <= hi).all()
    p = scene.make_rays_primary(lo, hi, 64, 64)
    assert (p[:, 0:3] == p[0, 0:3]).all()                          # one eye
    assert np.isclose(p[0, 7], np.sqrt(3), rtol=1e-6)              # tmax = clip = |extents|
    # the centre column / row have an exactly zero direction component (kx = 0, ky = 0)
    assert (p[np.arange(64) * 64 + 32, 4] == 0).all() and (p[32 * 64:33 * 64, 5] == 0).all()


def test_clustered_scene_is_pinned_and_well_formed():
    """scene.make_clustered: same bits everywhere (the GPU box regenerates it), normals = cross(e1, e2) of the scaled edges,
    blobs where the docstring says."""
    import hashlib
    t = scene.make_clustered(2000, 3, 1000)
    assert t.shape == (5000, 12) and t.dtype == np.float32
    assert hashlib.sha256(t.tobytes()).hexdigest()[:16] == "9700c36972a7f1aa"
    e1, e2 = t[:, 4:7], t[:, 8:11]
    n = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                  e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1).astype(np.float32)
    assert (n[:, 0] == t[:, 3]).all() and (n[:, 1] == t[:, 7]).all() and (n[:, 2] == t[:, 11]).all()
    blob = t[2000:3000, 0:3]
    assert (blob >= np.float32([0.15, 0.30, 0.20])).all() and (blob <= np.float32([0.19, 0.34, 0.24]) + 1e-6).all()
