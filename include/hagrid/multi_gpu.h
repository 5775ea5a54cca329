// hagrid/multi_gpu.h -- extension of the reference's C++ interface (which knows one GPU): the grid as one
// buffer.  One process per GPU; the rank that built the grid broadcasts it once over RCCL / xGMI (north_star), ray batches
// are sharded without further communication (rays are independent, the grid is read-only: traverse.cu:35-38).
// Header-only shims over hagrid_grid_broadcast / hagrid_grid_save / hagrid_grid_load of include/hagrid_amd.h.
#ifndef HAGRID_MULTI_GPU_H
#define HAGRID_MULTI_GPU_H

#include <string>

#include "build.h"
#include "mem_manager.h"
#include "prims.h"

namespace hagrid {

/// Rank `root` passes its finished grid and triangles; every other rank receives them into buffers of `mem` (freed like the
/// arrays of a built grid: mem.free(grid.entries) ...).  `comm` is an ncclComm_t spanning the ranks.
inline void broadcast_grid(MemManager& mem, Grid& grid, Tri*& tris, int& num_tris, void* comm, int rank, int root = 0) {
    hagrid_grid p = detail::to_pod(grid);
    void* t = tris;
    detail::check(mem.context(), hagrid_grid_broadcast(mem.context(), comm, rank, root, &p, &t, &num_tris));
    if (rank != root) { detail::from_pod(grid, p); tris = static_cast<Tri*>(t); }
}

/// Contiguous share of rank `rank` of `n` items: [n * rank / world, n * (rank + 1) / world)  (SURVEY.md 8(e)).
inline void shard_range(size_t n, int rank, int world, size_t& begin, size_t& end) {
    begin = n * size_t(rank) / size_t(world);
    end = n * size_t(rank + 1) / size_t(world);
}

inline void save_grid(MemManager& mem, const Grid& grid, const Tri* tris, int num_tris, const std::string& path) {
    hagrid_grid p = detail::to_pod(grid);
    detail::check(mem.context(), hagrid_grid_save(mem.context(), &p, tris, num_tris, path.c_str()));
}

inline bool load_grid(MemManager& mem, const std::string& path, Grid& grid, Tri*& tris, int& num_tris) {
    hagrid_grid p;
    void* t = nullptr;
    if (hagrid_grid_load(mem.context(), path.c_str(), &p, &t, &num_tris) != HAGRID_OK) return false;
    detail::from_pod(grid, p);
    tris = static_cast<Tri*>(t);
    return true;
}

} // namespace hagrid

#endif // HAGRID_MULTI_GPU_H
