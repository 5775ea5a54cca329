// C++ drop-in check: the call sequence of the reference's main.cpp (:471-506, :535, :410-432) against
// include/hagrid/*.h, compiled as plain C++ (-DHOST= -DDEVICE=, the way the reference compiles main.cpp) and
// linked with libhagrid_amd.so.  Verifies the hits against a host brute force using the same headers.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hagrid/build.h"
#include "hagrid/mem_manager.h"
#include "hagrid/traverse.h"

using namespace hagrid;

static uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
static float rnd(uint64_t seed, uint64_t i) { return float(mix(seed + (i + 1) * 0x9E3779B97F4A7C15ull) >> 40) * (1.0f / 16777216.0f); }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000, nrays = argc > 2 ? atoi(argv[2]) : 4096;
    std::vector<Tri> host_tris(n);
    const float s = 1.0f / cbrtf(float(n));
    for (int i = 0; i < n; i++) {
        vec3 c(rnd(1, 9 * i), rnd(1, 9 * i + 1), rnd(1, 9 * i + 2));
        vec3 a = (2.0f * vec3(rnd(1, 9 * i + 3), rnd(1, 9 * i + 4), rnd(1, 9 * i + 5)) - vec3(1.0f)) * s;
        vec3 b = (2.0f * vec3(rnd(1, 9 * i + 6), rnd(1, 9 * i + 7), rnd(1, 9 * i + 8)) - vec3(1.0f)) * s;
        vec3 v0 = c, v1 = c + a, v2 = c + b, e1 = v0 - v1, e2 = v2 - v0, nn = cross(e1, e2);
        host_tris[i] = Tri(v0, nn.x, e1, nn.y, e2, nn.z);          // main.cpp:259-267
    }
    MemManager mem(true);
    auto tris = mem.alloc<Tri>(host_tris.size());
    mem.copy<Copy::HST_TO_DEV>(tris, host_tris.data(), host_tris.size());

    Grid grid;
    grid.entries = nullptr; grid.cells = nullptr; grid.ref_ids = nullptr; grid.small_cells = nullptr;
    float build_ms = 0;
    for (int it = 0; it < 2; it++) {                               // main.cpp:481-508
        mem.free(grid.entries); mem.free(grid.cells); mem.free(grid.ref_ids);
        build_ms = profile([&] {
            build_grid(mem, tris, n, grid, 0.12f, 2.4f);
            merge_grid(mem, grid, 0.995f);
            flatten_grid(mem, grid);
            expand_grid(mem, grid, tris, 3);
        });
    }
    auto dims = grid.dims << grid.shift;
    printf("Grid built in %g ms (%dx%dx%d, %d cells, %d references)\n", build_ms, dims.x, dims.y, dims.z, grid.num_cells, grid.num_refs);

    setup_traversal(grid);
    std::vector<Ray> host_rays(nrays);
    const vec3 lo = grid.bbox.min, ext = grid.bbox.extents();
    for (int i = 0; i < nrays; i++) {
        vec3 o = lo + vec3(rnd(2, 6 * i), rnd(2, 6 * i + 1), rnd(2, 6 * i + 2)) * ext;
        vec3 d = 2.0f * vec3(rnd(2, 6 * i + 3), rnd(2, 6 * i + 4), rnd(2, 6 * i + 5)) - vec3(1.0f);
        host_rays[i] = Ray(o, 0.0f, d, FLT_MAX);
    }
    Ray* rays = mem.alloc<Ray>(nrays);
    Hit* hits = mem.alloc<Hit>(nrays);
    mem.copy<Copy::HST_TO_DEV>(rays, host_rays.data(), host_rays.size());
    float ms = profile([&] { traverse_grid(grid, tris, rays, hits, nrays); });
    std::vector<Hit> host_hits(nrays);
    mem.copy<Copy::DEV_TO_HST>(host_hits.data(), hits, host_hits.size());

    int bad = 0, intr = 0;
    for (int i = 0; i < nrays; i++) {
        Hit h(-1, host_rays[i].tmax, 0, 0);
        for (int t = 0; t < n; t++) intersect_prim_ray(host_tris[t], Ray(host_rays[i].org, 0.0f, host_rays[i].dir, h.t), t, h);
        intr += host_hits[i].id >= 0;
        if (h.id != host_hits[i].id || h.t != host_hits[i].t) bad++;
    }
    printf("%d intersection(s), %g ms, %d mismatches vs host brute force\n", intr, ms, bad);
    // compress, traverse again
    if (compress_grid(mem, grid)) {
        traverse_grid(grid, tris, rays, hits, nrays);
        std::vector<Hit> h2(nrays);
        mem.copy<Copy::DEV_TO_HST>(h2.data(), hits, h2.size());
        for (int i = 0; i < nrays; i++) if (h2[i].id != host_hits[i].id || h2[i].t != host_hits[i].t) bad++;
    }
    // extensions over the same walk: occlusion rays report a hit exactly where the nearest-hit walk does, and the
    // barycentrics variant leaves (id, t) untouched and puts the hit point at v0 - u e1 + v e2
    {
        std::vector<Hit> ha(nrays), hu(nrays);
        traverse_grid_any_hit(grid, tris, rays, hits, nrays);
        mem.copy<Copy::DEV_TO_HST>(ha.data(), hits, ha.size());
        traverse_grid_with_uvs(grid, tris, rays, hits, nrays);
        mem.copy<Copy::DEV_TO_HST>(hu.data(), hits, hu.size());
        int bad_ext = 0;
        for (int i = 0; i < nrays; i++) {
            if ((ha[i].id >= 0) != (host_hits[i].id >= 0)) bad_ext++;
            if (hu[i].id != host_hits[i].id || hu[i].t != host_hits[i].t) bad_ext++;
            if (hu[i].id >= 0) {
                const Tri& t = host_tris[hu[i].id];
                const vec3 p = t.v0 - t.e1 * hu[i].u + t.e2 * hu[i].v, q = host_rays[i].org + host_rays[i].dir * hu[i].t;
                if (length(p - q) > 1e-3f) bad_ext++;
            }
        }
        printf("%d mismatches in the any-hit / barycentric variants\n", bad_ext);
        bad += bad_ext;
    }
    // a second manager (context, stream) traversing with the first one's traversal image
    {
        setup_traversal(grid);
        MemManager mem2(false);
        mem.make_current();
        share_traversal(mem2, mem);
        Hit* hits2 = mem2.alloc<Hit>(nrays);
        traverse_grid(grid, tris, rays, hits, nrays);
        traverse_grid(mem2, grid, tris, rays, hits2, nrays);
        std::vector<Hit> h1(nrays), h2(nrays);
        mem.copy<Copy::DEV_TO_HST>(h1.data(), hits, h1.size());
        mem2.copy<Copy::DEV_TO_HST>(h2.data(), hits2, h2.size());
        int bad_share = 0;
        for (int i = 0; i < nrays; i++) if (h1[i].id != h2[i].id || h1[i].t != h2[i].t || h1[i].id != host_hits[i].id) bad_share++;
        printf("%d mismatches with a shared traversal image\n", bad_share);
        bad += bad_share;
        mem2.free(hits2);
    }
    mem.make_current();
    mem.free(rays); mem.free(hits);
    mem.free(grid.entries); mem.free(grid.cells); mem.free(grid.ref_ids); mem.free(grid.small_cells); mem.free(tris);
    printf("peak usage %.1f MB, usage after free %zu\n", mem.max_usage() / 1048576.0, mem.usage());
    fflush(stdout);
    return bad == 0 && intr > 0 ? 0 : 1;
}
