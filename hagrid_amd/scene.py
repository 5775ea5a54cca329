"""Synthetic scenes and ray buffers.

Everything is generated from a counter-based 64-bit integer PRNG (splitmix64 finaliser keyed by
``seed + (index + 1) * golden``), so any machine produces identical bits and any slice of a buffer
can be generated independently (rank ``r`` of a multi-GPU run generates only its own rays).
No libm, no ``numpy.random``.

Layouts follow the reference PODs:
  Tri  = 12 x f32: v0.xyz, n.x, e1.xyz, n.y, e2.xyz, n.z   (prims.h:13-25, main.cpp:259-267)
  Ray  =  8 x f32: org.xyz, tmin, dir.xyz, tmax           (ray.h:9-20)
  Hit  = {i32 id, f32 t, f32 u, f32 v}                     (ray.h:22-33)
"""
from __future__ import annotations

import numpy as np

HIT_DTYPE = np.dtype([("id", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
CELL_DTYPE = np.dtype([("min", "<i4", 3), ("begin", "<i4"), ("max", "<i4", 3), ("end", "<i4")])
SMALL_CELL_DTYPE = np.dtype([("min", "<u2", 3), ("max", "<u2", 3), ("begin", "<i4")])

FLT_MAX = np.float32(3.4028234663852886e38)

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

SCENE_SEED_BASE = 0x48414752494400  # + N          
RAY_SEED_BASE = 0x52415953          # + config #


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def uniform01(seed: int, index: np.ndarray) -> np.ndarray:
    """float32 in [0,1): (splitmix64(seed, index) >> 40) * 2^-24."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (index.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = _mix(z)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def _cross(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    # vec.h:104-109 in float32, no fused ops
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                     a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                     a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1).astype(np.float32)


def tris_from_vertices(v0: np.ndarray, v1: np.ndarray, v2: np.ndarray) -> np.ndarray:
    """Pack triangles exactly like main.cpp:259-267: e1 = v0 - v1, e2 = v2 - v0, n = cross(e1, e2)."""
    v0 = v0.astype(np.float32); v1 = v1.astype(np.float32); v2 = v2.astype(np.float32)
    e1 = v0 - v1
    e2 = v2 - v0
    n = _cross(e1, e2)
    out = np.empty((v0.shape[0], 12), dtype=np.float32)
    out[:, 0:3] = v0; out[:, 3] = n[:, 0]
    out[:, 4:7] = e1; out[:, 7] = n[:, 1]
    out[:, 8:11] = e2; out[:, 11] = n[:, 2]
    return out


def make_soup(num_tris: int, seed: int | None = None, first: int = 0, count: int | None = None) -> np.ndarray:
    """Scene "soup-N": c ~ U[0,1)^3, a, b ~ U[-s, s]^3, s = N^(-1/3); v0 = c, v1 = c + a, v2 = c + b."""
    if seed is None:
        seed = SCENE_SEED_BASE + num_tris
    if count is None:
        count = num_tris - first
    s = np.float32(float(num_tris) ** (-1.0 / 3.0))
    idx = (np.arange(first, first + count, dtype=np.uint64)[:, None] * np.uint64(9)
           + np.arange(9, dtype=np.uint64)[None, :])
    u = uniform01(seed, idx)
    c = u[:, 0:3]
    a = (np.float32(2.0) * u[:, 3:6] - np.float32(1.0)) * s
    b = (np.float32(2.0) * u[:, 6:9] - np.float32(1.0)) * s
    return tris_from_vertices(c, c + a, c + b)


def make_clustered(num_sparse: int = 100000, clusters: int = 6, per_cluster: int = 150000) -> np.ndarray:
    """A very non-uniform scene (teapot in a stadium, the case irregular grids exist for): a sparse soup over the unit cube
    and `clusters` dense blobs, each a soup shrunk to 4 % of the cube.  With the defaults: 1M triangles, grid shift 6, cell
    lists of up to ~20 references inside the blobs.  Normals are recomputed from the scaled edges as make_soup does."""
    parts = [make_soup(num_sparse, seed=7)]
    for k in range(clusters):
        c = make_soup(per_cluster, seed=20 + k).copy()
        centre = np.float32([0.15 + 0.14 * k, 0.3 + 0.08 * k, 0.2 + 0.1 * k])
        c[:, 0:3] = c[:, 0:3] * np.float32(0.04) + centre
        c[:, 4:7] *= np.float32(0.03); c[:, 8:11] *= np.float32(0.03)
        n = _cross(c[:, 4:7], c[:, 8:11]).astype(np.float32)
        c[:, 3] = n[:, 0]; c[:, 7] = n[:, 1]; c[:, 11] = n[:, 2]
        parts.append(c)
    return np.ascontiguousarray(np.concatenate(parts), np.float32)


def make_gradient(num_tris: int = 1_000_000, seed: int | None = None) -> np.ndarray:
    """A soup whose density rises towards one corner: positions squared, edges scaled with the local stretch.  Long tiles of many cells with short lists -- the
    scene family on which counting iterations mispredicts what four lanes per ray buy (profiles/NOTES.md "Round 5")."""
    base = make_soup(num_tris, seed=seed)
    v0 = base[:, 0:3].astype(np.float64); e1 = -base[:, 4:7].astype(np.float64); e2 = base[:, 8:11].astype(np.float64)      # v1 = v0 - e1, v2 = v0 + e2 (prims.h)
    k = np.maximum(2.0 * v0, 0.05); v0 = v0 ** 2
    return tris_from_vertices(v0.astype(np.float32), (v0 + e1 * k).astype(np.float32), (v0 + e2 * k).astype(np.float32))


def make_shell(num_tris: int = 1_000_000, seed: int | None = None) -> np.ndarray:
    """A surface: the soup's triangles moved onto a sphere of radius 0.4, nothing inside or around it."""
    base = make_soup(num_tris, seed=seed)
    v0 = base[:, 0:3].astype(np.float64); e1 = -base[:, 4:7].astype(np.float64); e2 = base[:, 8:11].astype(np.float64)
    d = v0 - 0.5; d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6); v0 = 0.5 + 0.4 * d
    return tris_from_vertices(v0.astype(np.float32), (v0 + e1).astype(np.float32), (v0 + e2).astype(np.float32))


def _sincos_turns(turns: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """sin and cos of 2 pi * turns in float64 from + and x only (octant reduction, Taylor series): any machine produces identical bits (no libm)."""
    t = np.asarray(turns, np.float64); t = t - np.floor(t)
    q = np.floor(t * 8.0 + 0.5)                       # nearest multiple of an eighth turn
    x = (t - q / 8.0) * 6.283185307179586             # |x| <= pi / 8
    x2 = x * x
    s = x * (1.0 + x2 * (-1.0 / 6 + x2 * (1.0 / 120 + x2 * (-1.0 / 5040 + x2 * (1.0 / 362880 + x2 * (-1.0 / 39916800 + x2 * (1.0 / 6227020800)))))))
    c = 1.0 + x2 * (-0.5 + x2 * (1.0 / 24 + x2 * (-1.0 / 720 + x2 * (1.0 / 40320 + x2 * (-1.0 / 3628800 + x2 * (1.0 / 479001600 + x2 * (-1.0 / 87178291200)))))))
    r = 0.7071067811865476
    sq = np.array([0.0, r, 1.0, r, 0.0, -r, -1.0, -r])[q.astype(np.int64) % 8]; cq = np.array([1.0, r, 0.0, -r, -1.0, -r, 0.0, r])[q.astype(np.int64) % 8]
    return sq * c + cq * s, cq * c - sq * s


def _grid_faces(nu: int, nv: int, wrap_u: bool, wrap_v: bool, base: int) -> np.ndarray:
    """two triangles per quad of an nu x nv vertex lattice (vertex (i, j) = base + i * nv + j), wrapping where asked: shared vertices, no duplicates"""
    iu = np.arange(nu if wrap_u else nu - 1); jv = np.arange(nv if wrap_v else nv - 1)
    i, j = np.meshgrid(iu, jv, indexing="ij"); i = i.ravel(); j = j.ravel()
    i1 = (i + 1) % nu; j1 = (j + 1) % nv
    a = base + i * nv + j; b = base + i1 * nv + j; c = base + i1 * nv + j1; d = base + i * nv + j1
    return np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int32)


def make_stadium_mesh(detail: float = 1.0) -> tuple[np.ndarray, np.ndarray]:
    """"Teapot in a stadium" (the reference README's own motivation, README.md:5-12) as an INDEXED mesh: finely tessellated connected surfaces -- tori and
    spheres with shared vertices, a grain of dust among them -- inside a hall of ten huge triangles, with terraces of long thin ones and pillars of slivers as
    high as the hall.  With detail = 1: ~0.96M triangles whose edges span four orders of magnitude (1.0 ... 1e-4).  Returns (vertices float32 [nv, 3], faces
    int32 [nf, 3], zero-based); write_obj() / tris_from_mesh() take it from there.  `detail` scales the tessellation (tests use small ones)."""
    V = []; F = []; nvert = 0

    def add(verts, faces):
        nonlocal nvert
        V.append(np.asarray(verts, np.float64)); F.append(np.asarray(faces, np.int32)); nvert += len(verts)

    # the hall: floor, ceiling, two side walls, back wall of the unit cube (open towards the camera at -z): 8 shared corners, 10 triangles of edge 1
    corners = [[x, y, z] for z in (0.0, 1.0) for y in (0.0, 1.0) for x in (0.0, 1.0)]
    quads = [(0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5), (4, 5, 7, 6)]
    add(corners, [[nvert + q[0], nvert + q[1], nvert + q[2]] for q in quads] + [[nvert + q[0], nvert + q[2], nvert + q[3]] for q in quads])
    # terraces along the back wall: a staircase profile swept across x (treads and risers 0.9 long, 0.03 deep)
    steps = 12
    prof = [(0.0 + 0.03 * ((k + 1) // 2), 0.97 - 0.03 * (k // 2)) for k in range(2 * steps + 1)]          # (y, z) of the profile's corners
    tv = [[x, y, z] for (y, z) in prof for x in (0.05, 0.95)]
    add(tv, _grid_faces(len(prof), 2, False, False, nvert))
    # pillars: coarse cylinders from floor to ceiling, 16 segments: slivers 1.0 high and 0.004 wide
    for k in range(8):
        cx, cz = (0.12 if k % 2 == 0 else 0.88), 0.15 + 0.2 * (k // 2)
        sn, cs = _sincos_turns(np.arange(16) / 16.0)
        ring = [[cx + 0.01 * c, y, cz + 0.01 * s_] for c, s_ in zip(cs, sn) for y in (0.0, 1.0)]
        add(ring, _grid_faces(16, 2, True, False, nvert))

    def torus(centre, R, r, nu, nv, tilt):
        u = np.arange(nu) / nu; v = np.arange(nv) / nv
        su, cu = _sincos_turns(u); sv, cv = _sincos_turns(v); st, ct = _sincos_turns(np.array([tilt]))
        x = (R + r * cv[None, :]) * cu[:, None]; z = (R + r * cv[None, :]) * su[:, None]; y = np.broadcast_to(r * sv[None, :], x.shape)
        y2 = y * ct[0] - z * st[0]; z2 = y * st[0] + z * ct[0]                                   # tilted about the x axis
        add(np.stack([x + centre[0], y2 + centre[1], z2 + centre[2]], -1).reshape(-1, 3), _grid_faces(nu, nv, True, True, nvert))

    def sphere(centre, r, nu, nv):
        # nu meridians x (nv - 1) rings between two pole vertices (fans at the poles: no degenerate triangles)
        u = np.arange(nu) / nu; lat = np.arange(1, nv) / (2.0 * nv)                              # turns from the north pole, (0, 1/2)
        su, cu = _sincos_turns(u); sl, cl = _sincos_turns(lat)
        x = r * sl[None, :] * cu[:, None]; z = r * sl[None, :] * su[:, None]; y = np.broadcast_to(r * cl[None, :], x.shape)
        base = nvert
        body = np.stack([x + centre[0], y + centre[1], z + centre[2]], -1).reshape(-1, 3)
        faces = _grid_faces(nu, nv - 1, True, False, base)
        north, south = base + nu * (nv - 1), base + nu * (nv - 1) + 1
        i = np.arange(nu); i1 = (i + 1) % nu
        fans = np.concatenate([np.stack([np.full(nu, north), base + i1 * (nv - 1), base + i * (nv - 1)], 1),
                               np.stack([np.full(nu, south), base + i * (nv - 1) + nv - 2, base + i1 * (nv - 1) + nv - 2], 1)]).astype(np.int32)
        add(np.concatenate([body, [[centre[0], centre[1] + r, centre[2]], [centre[0], centre[1] - r, centre[2]]]]), np.concatenate([faces, fans]))

    d = lambda n: max(8, int(round(n * detail)))
    for k, (c, tilt) in enumerate([((0.30, 0.12, 0.35), 0.0), ((0.68, 0.15, 0.40), 0.07), ((0.45, 0.30, 0.62), 0.19), ((0.60, 0.10, 0.25), 0.31)]):
        torus(c, 0.08, 0.03, d(400), d(200), tilt)                                               # 4 x 160k triangles, edge ~1.3e-3
    for c in ((0.40, 0.06, 0.20), (0.75, 0.30, 0.65), (0.22, 0.25, 0.55)):
        sphere(c, 0.05, d(300), d(150))                                                          # 3 x 90k, edge ~1e-3
    sphere((0.50, 0.02, 0.30), 0.004, d(200), d(100))                                            # a grain of dust: 40k triangles, edge ~1e-4
    for c in ((0.2, 0.8, 0.5), (0.5, 0.85, 0.7), (0.8, 0.8, 0.45)):
        sphere(c, 0.1, 16, 8)                                                                    # lamps: coarse, edge ~0.04
    return np.concatenate(V).astype(np.float32), np.concatenate(F).astype(np.int32)


def tris_from_mesh(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """the triangles of an indexed mesh as main.cpp:259-267 packs them (v0, e1 = v0 - v1, e2 = v2 - v0, n)"""
    return np.ascontiguousarray(tris_from_vertices(verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]))


def make_stadium(detail: float = 1.0) -> np.ndarray:
    """the triangles of make_stadium_mesh(), in face order"""
    return tris_from_mesh(*make_stadium_mesh(detail))


def write_obj(path: str, verts: np.ndarray, faces: np.ndarray, mixed_forms: bool = True) -> None:
    """An indexed mesh as a Wavefront OBJ file the reference's loader reads (load_obj.cpp:78-239): vertices with nine significant digits (a float32 survives the
    round trip), one-based faces; with mixed_forms every third face is written as v/vt/vn and every third with negative indices."""
    nv = verts.shape[0]
    with open(path, "w") as f:
        f.write("# hagrid_amd.scene.write_obj\nvt 0 0\nvn 0 0 1\n")
        f.write("".join("v %.9g %.9g %.9g\n" % (float(x), float(y), float(z)) for x, y, z in verts.astype(np.float64)))
        a = faces.astype(np.int64) + 1
        lines = []
        for k in range(0, a.shape[0], 1 << 16):
            blk = a[k:k + (1 << 16)]
            for i, (p, q, r) in enumerate(blk, start=k):
                m = i % 3 if mixed_forms else 0
                if m == 0: lines.append("f %d %d %d\n" % (p, q, r))
                elif m == 1: lines.append("f %d/1/1 %d/1/1 %d/1/1\n" % (p, q, r))
                else: lines.append("f %d %d %d\n" % (p - nv - 1, q - nv - 1, r - nv - 1))
            f.write("".join(lines)); lines = []


def make_rays_aimed(bbox_min, bbox_max, num_rays: int, seed: int, first: int = 0) -> np.ndarray:
    """Incoherent origins (make_rays_incoherent) with directions towards the blobs of make_clustered, with some spread: ray i aims at blob i % 6.  The rays
    that end inside the dense parts of a very non-uniform scene (bench.py --config clustered --rays aimed; tests/test_fullsize_gpu.py)."""
    rays = make_rays_incoherent(bbox_min, bbox_max, num_rays, seed, first=first).copy()
    k = (np.arange(first, first + num_rays) % 6).astype(np.float32)
    centre = np.stack([np.float32(0.17) + np.float32(0.14) * k, np.float32(0.32) + np.float32(0.08) * k, np.float32(0.22) + np.float32(0.1) * k], axis=1).astype(np.float32)
    rays[:, 4:7] = centre - rays[:, 0:3] + np.float32(0.02) * rays[:, 4:7]
    return np.ascontiguousarray(rays, np.float32)


def tris_bbox(tris: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Scene bounding box over the three vertices (prims.h:27-31)."""
    v0 = tris[:, 0:3]; v1 = v0 - tris[:, 4:7]; v2 = v0 + tris[:, 8:11]
    lo = np.minimum(v0, np.minimum(v1, v2)).min(axis=0)
    hi = np.maximum(v0, np.maximum(v1, v2)).max(axis=0)
    return lo.astype(np.float32), hi.astype(np.float32)


def make_rays_incoherent(bbox_min, bbox_max, num_rays: int, seed: int, first: int = 0,
                         tmin: float = 0.0, tmax: float = float(FLT_MAX)) -> np.ndarray:
    """org ~ U(bbox); dir rejection-sampled from U[-1,1]^3 with 0.01 < |d|^2 <= 1, un-normalised."""
    lo = np.asarray(bbox_min, dtype=np.float32); hi = np.asarray(bbox_max, dtype=np.float32)
    ids = np.arange(first, first + num_rays, dtype=np.uint64)
    rays = np.empty((num_rays, 8), dtype=np.float32)
    uo = uniform01(seed, ids[:, None] * np.uint64(3) + np.arange(3, dtype=np.uint64)[None, :])
    rays[:, 0:3] = lo + uo * (hi - lo)
    rays[:, 3] = np.float32(tmin)
    rays[:, 7] = np.float32(tmax)
    todo = np.arange(num_rays)
    dseed = seed ^ 0x6469720000000000
    for attempt in range(64):
        if todo.size == 0:
            break
        base = (ids[todo] * np.uint64(64) + np.uint64(attempt)) * np.uint64(3)
        u = uniform01(dseed, base[:, None] + np.arange(3, dtype=np.uint64)[None, :])
        d = np.float32(2.0) * u - np.float32(1.0)
        l2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        ok = (l2 > np.float32(0.01)) & (l2 <= np.float32(1.0))
        rays[todo[ok], 4:7] = d[ok]
        todo = todo[~ok]
    if todo.size:
        rays[todo, 4:7] = np.float32([0.0, 0.0, 1.0])
    return rays


def camera(bbox_min, bbox_max, eye_dist: float = 0.8, fov: float = 60.0, ratio: float = 1.0, yaw: float = 0.0, strafe: float = 0.0):
    """gen_camera (main.cpp:42-50) looking down +z at the bbox centre from eye_dist * diagonal away.  yaw (radians) turns the view about the
    up axis and strafe (scene diagonals) moves the eye sideways: what the reference's viewer does per mouse pixel (0.005 rad) and per key
    event (0.005 x the scene size), main.cpp:579-586 -- a frame loop with a moving camera (tools/dev_moving_camera.py, bench.py)."""
    lo = np.asarray(bbox_min, dtype=np.float32); hi = np.asarray(bbox_max, dtype=np.float32)
    ext = hi - lo
    diag = np.float32(np.sqrt(np.float32(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2])))
    center = np.float32(0.5) * (hi + lo)
    eye = (center + np.float32([0.0, 0.0, -1.0]) * np.float32(eye_dist) * diag).astype(np.float32)
    up0 = np.float32([0.0, 1.0, 0.0])
    if yaw or strafe:
        eye = (eye + np.float32([1.0, 0.0, 0.0]) * np.float32(strafe) * diag).astype(np.float32)
        fwd = np.float32([np.sin(yaw), 0.0, np.cos(yaw)])
        center = (eye + fwd * np.float32(eye_dist) * diag).astype(np.float32)

    def norm(v):
        return (v * (np.float32(1.0) / np.float32(np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))))).astype(np.float32)

    def cross(a, b):
        return np.float32([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])

    f = np.float32(np.tan(np.pi * fov / 360.0))
    cdir = norm(center - eye)
    right = norm(cross(cdir, up0)) * np.float32(f * np.float32(ratio))
    up = norm(cross(right, cdir)) * f
    return eye, cdir, right.astype(np.float32), up.astype(np.float32), diag


def make_rays_primary(bbox_min, bbox_max, width: int, height: int, first: int = 0, count: int | None = None,
                      eye_dist: float = 0.8, fov: float = 60.0, sample: int = 0, num_samples: int = 1, yaw: float = 0.0, strafe: float = 0.0) -> np.ndarray:
    """gen_rays (main.cpp:52-66): pixel (x, y) -> dir = cam.dir + right*kx + up*ky, tmax = clip = |extents|.
    sample / num_samples shifts the pixel by a sub-pixel offset in x (sample 0 of 1 = the reference's rays):
    the weak-scaling batches of a multi-GPU run are the N sub-pixel samples of the same camera."""
    eye, cdir, right, up, diag = camera(bbox_min, bbox_max, eye_dist, fov, width / float(height), yaw, strafe)
    if count is None:
        count = width * height - first
    pid = np.arange(first, first + count, dtype=np.int64)
    x = (pid % width).astype(np.float32); y = (pid // width).astype(np.float32)
    if sample:
        x = x + np.float32(sample) / np.float32(num_samples)
    kx = np.float32(2.0) * x / np.float32(width) - np.float32(1.0)
    ky = np.float32(1.0) - np.float32(2.0) * y / np.float32(height)
    rays = np.empty((count, 8), dtype=np.float32)
    rays[:, 0:3] = eye
    rays[:, 3] = np.float32(0.0)
    rays[:, 4:7] = cdir[None, :] + right[None, :] * kx[:, None] + up[None, :] * ky[:, None]
    rays[:, 7] = diag
    return rays


def make_rays_bounce(tris: np.ndarray, rays: np.ndarray, hits: np.ndarray, bbox_min, bbox_max, seed: int,
                     first: int = 0) -> np.ndarray:
    """Diffuse-bounce rays (BASELINE config 5): from each hit, org = p + 1e-4 * n, cosine-weighted
    direction about the ray-facing normal from two PRNG floats keyed by the ray index; misses are
    re-drawn as incoherent rays."""
    n_rays = rays.shape[0]
    hid = hits["id"]
    out = make_rays_incoherent(bbox_min, bbox_max, n_rays, seed ^ 0x6D69737300000000, first)
    hit_mask = hid >= 0
    if not hit_mask.any():
        return out
    r = rays[hit_mask]; t = hits["t"][hit_mask]
    tri = tris[hid[hit_mask]]
    p = r[:, 0:3] + r[:, 4:7] * t[:, None]
    n = np.stack([tri[:, 3], tri[:, 7], tri[:, 11]], axis=1)
    ln = np.sqrt(np.maximum((n * n).sum(axis=1), np.float32(1e-30))).astype(np.float32)
    n = n / ln[:, None]
    facing = (n * r[:, 4:7]).sum(axis=1) > 0
    n[facing] = -n[facing]
    ids = np.arange(first, first + n_rays, dtype=np.uint64)[hit_mask]
    u = uniform01(seed, ids[:, None] * np.uint64(2) + np.arange(2, dtype=np.uint64)[None, :])
    # cosine-weighted hemisphere via a polynomial-free construction: disk point by rejection-free
    # concentric mapping would need trig; use (r, phi) with sqrt only and a rational unit circle
    # parametrisation phi -> ((1-s^2)/(1+s^2), 2s/(1+s^2)), s in [-1,1), mirrored by the second bit.
    s = np.float32(2.0) * u[:, 1] - np.float32(1.0)
    half = (ids & np.uint64(1)).astype(np.float32) * np.float32(2.0) - np.float32(1.0)
    cx = (np.float32(1.0) - s * s) / (np.float32(1.0) + s * s) * half
    cy = np.float32(2.0) * s / (np.float32(1.0) + s * s)
    rad = np.sqrt(u[:, 0]).astype(np.float32)
    dx = rad * cx; dy = rad * cy
    dz = np.sqrt(np.maximum(np.float32(1.0) - u[:, 0], np.float32(0.0))).astype(np.float32)
    # orthonormal basis (Frisvad-free, branch on the dominant axis)
    a = np.where((np.abs(n[:, 0]) > np.float32(0.5))[:, None], np.float32([0.0, 1.0, 0.0])[None, :], np.float32([1.0, 0.0, 0.0])[None, :]).astype(np.float32)
    tx = _cross(a, n)
    tx = tx / np.sqrt((tx * tx).sum(axis=1)).astype(np.float32)[:, None]
    ty = _cross(n, tx)
    d = tx * dx[:, None] + ty * dy[:, None] + n * dz[:, None]
    b = np.empty((r.shape[0], 8), dtype=np.float32)
    b[:, 0:3] = p + np.float32(1e-4) * n
    b[:, 3] = np.float32(0.0)
    b[:, 4:7] = d
    b[:, 7] = FLT_MAX
    out[hit_mask] = b
    return out.astype(np.float32)


def generate_parallel(gen, first: int, count: int, chunk: int = 1 << 20, threads: int | None = None) -> np.ndarray:
    """gen(first, count) -> (count, 8) float32 for any slice (the generators above are counter-based): builds
    [first, first + count) from chunks made on a thread pool (numpy releases the GIL inside its loops)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty((count, 8), dtype=np.float32)
    starts = list(range(0, count, chunk))

    def work(o):
        c = min(chunk, count - o)
        out[o:o + c] = gen(first + o, c)

    with ThreadPoolExecutor(max_workers=threads or min(32, os.cpu_count() or 1)) as ex:
        list(ex.map(work, starts))
    return out


def shard_range(num_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous ray range of rank ``rank``: [g*n/G, (g+1)*n/G)."""
    return (num_items * rank) // world, (num_items * (rank + 1)) // world
