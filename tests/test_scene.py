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


def test_stadium_mesh_is_pinned_connected_and_spans_orders_of_magnitude():
    """scene.make_stadium_mesh: an indexed mesh (shared vertices: every edge of a torus or a sphere belongs to exactly two faces), no degenerate triangle, edges from
    the hall's 1.0 down to the grain of dust; the same bits everywhere (no libm in it)."""
    import hashlib
    V, F = scene.make_stadium_mesh(0.1)
    assert V.dtype == np.float32 and F.dtype == np.int32 and F.min() == 0 and F.max() == V.shape[0] - 1
    assert hashlib.sha256(V.tobytes() + F.tobytes()).hexdigest()[:16] == "7b9b521d0f9a1490"
    t = scene.tris_from_mesh(V, F)
    e1 = t[:, 4:7].astype(np.float64); e2 = t[:, 8:11].astype(np.float64)
    assert (np.linalg.norm(np.cross(e1, e2), axis=1) > 0).all()
    longest = np.maximum(np.linalg.norm(e1, axis=1), np.maximum(np.linalg.norm(e2, axis=1), np.linalg.norm(e1 + e2, axis=1)))
    assert longest.max() > 1.0 and longest.min() < 2e-3
    # closed surfaces: the faces of the LAST object (a coarse sphere: 16 x 8) form a closed 2-manifold, every undirected edge twice, once in each direction
    nf = 2 * 16 * 6 + 2 * 16
    f = F[-nf:].astype(np.int64)
    edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = edges[:, 0] * (1 << 32) + edges[:, 1]; rev = edges[:, 1] * (1 << 32) + edges[:, 0]
    assert len(np.unique(key)) == len(key) and set(key.tolist()) == set(rev.tolist())
    # the full-size scene: about a million triangles, four orders of magnitude
    V, F = scene.make_stadium_mesh()
    assert 900_000 < F.shape[0] < 1_000_000 and V.shape[0] < 0.51 * F.shape[0] + 1000
    a = V[F[:, 0]].astype(np.float64); b = V[F[:, 1]].astype(np.float64)
    assert np.linalg.norm(a - b, axis=1).min() < 2e-4


def test_stadium_obj_round_trip_through_the_loader():
    """write_obj -> include/hagrid/load_obj.h (the front door of hagrid_cli; fan of main.cpp:246-275) gives back the very Tri records of tris_from_mesh: nine
    significant digits carry a float32, negative and v/vt/vn index forms resolve to the same vertices."""
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    V, F = scene.make_stadium_mesh(0.06)
    want = scene.tris_from_mesh(V, F)
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "obj_dump")
        subprocess.run(["g++", "-std=c++11", "-O2", "-ffp-contract=off", "-DHOST=", "-DDEVICE=", "-I", os.path.join(root, "include"),
                        os.path.join(root, "tests", "cpp", "obj_dump.cpp"), "-o", exe], check=True)
        obj = os.path.join(d, "stadium.obj")
        scene.write_obj(obj, V, F)
        r = subprocess.run([exe, obj], capture_output=True, check=True)
        head, _, body = r.stdout.partition(b"\n")
        assert int(head) == want.shape[0]
        assert body == want.tobytes()
