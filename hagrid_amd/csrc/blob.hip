// blob.hip -- the grid as ONE contiguous device buffer: header + entries + cells | small_cells + ref_ids + triangles.
//
// No reference counterpart (the reference never serialises a grid and knows one GPU: SURVEY.md section 5 "checkpoint / resume"
// and 8(e)).  The blob is what travels: one RCCL broadcast from the GPU that built the grid to the others (north_star: "the
// built grid broadcast once over RCCL/xGMI"), and the same bytes are the file form (hagrid_grid_save / hagrid_grid_load).
//   pack    four device-to-device copies into a pool buffer
//   unpack  in place: the blob's pool slot is split into the four arrays, which are then ordinary pool buffers of the grid
//           (released one by one with hagrid_mem_free, as main.cpp:496-498 does with a built grid) -- no copy
//   bcast   header first (256 bytes, the receivers size their buffer), then the payload, straight from / into pool memory.
//           RCCL is bound at run time (dlopen), so the library itself carries no dependency on it.
#include "ctx.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

using namespace hagrid_impl;

namespace {

constexpr uint32_t kBlobMagic = 0x42524748u;   // "HGRB"
constexpr size_t kAlign = 128;
size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

static_assert(sizeof(hagrid_blob_header) == 256, "blob header layout");

int fill_header(hagrid_ctx* ctx, const hagrid_grid* g, int num_tris, hagrid_blob_header& h) {
    if (!g || !g->entries || !g->ref_ids || (!g->cells && !g->small_cells)) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: incomplete grid");
    if (num_tris < 0 || g->num_cells < 0 || g->num_entries < 0 || g->num_refs < 0 || g->num_offsets < 0 || g->num_offsets > HAGRID_MAX_LEVELS)
        HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: bad counts");
    memset(&h, 0, sizeof(h));
    h.magic = kBlobMagic; h.version = 1;
    for (int i = 0; i < 3; i++) { h.dims[i] = g->dims[i]; h.bbox_min[i] = g->bbox_min[i]; h.bbox_max[i] = g->bbox_max[i]; }
    h.shift = g->shift; h.num_cells = g->num_cells; h.num_entries = g->num_entries; h.num_refs = g->num_refs; h.num_tris = num_tris;
    h.compressed = g->small_cells ? 1 : 0;
    h.num_offsets = g->num_offsets;
    for (int i = 0; i < g->num_offsets; i++) h.offsets[i] = g->offsets[i];
    // every section takes at least one 128-byte unit, so the four arrays have four different addresses
    size_t at = sizeof(hagrid_blob_header);
    h.off_entries = at; at += align_up(std::max<size_t>(size_t(g->num_entries) * 4, 1));
    h.off_cells = at;   at += align_up(std::max<size_t>(size_t(g->num_cells) * (h.compressed ? 16 : 32), 1));
    h.off_refs = at;    at += align_up(std::max<size_t>(size_t(g->num_refs) * 4, 1));
    h.off_tris = at;    at += align_up(std::max<size_t>(size_t(num_tris) * 48, 1));
    h.total_bytes = at;
    return HAGRID_OK;
}

int check_header(hagrid_ctx* ctx, const hagrid_blob_header& h, size_t bytes) {
    if (h.magic != kBlobMagic || h.version != 1) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: bad magic / version");
    hagrid_grid g;
    memset(&g, 0, sizeof(g));
    g.entries = g.ref_ids = g.cells = reinterpret_cast<void*>(1);
    if (h.compressed) { g.small_cells = g.cells; g.cells = nullptr; }
    g.num_cells = h.num_cells; g.num_entries = h.num_entries; g.num_refs = h.num_refs; g.num_offsets = h.num_offsets;
    hagrid_blob_header want;
    HG_TRY(fill_header(ctx, &g, h.num_tris, want));
    if (want.off_entries != h.off_entries || want.off_cells != h.off_cells || want.off_refs != h.off_refs || want.off_tris != h.off_tris ||
        want.total_bytes != h.total_bytes || h.total_bytes > bytes) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: inconsistent section table");
    if (h.shift < 0 || h.shift > 15 || h.dims[0] <= 0 || h.dims[1] <= 0 || h.dims[2] <= 0) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: bad dimensions");
    // the top level of the voxel map is one entry per top-level cell and comes first; offsets[] are the cumulative entry counts
    const long long top = (long long)h.dims[0] * h.dims[1] * h.dims[2];
    if (top > h.num_entries) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: fewer entries than top-level cells");
    if (h.num_offsets < 1 || h.offsets[0] != top || h.offsets[h.num_offsets - 1] != h.num_entries) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: level offsets do not describe the voxel map");
    for (int i = 1; i < h.num_offsets; i++)
        if (h.offsets[i] < h.offsets[i - 1]) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: level offsets do not describe the voxel map");
    if (h.num_cells < 1 || (h.compressed && h.num_refs < 1)) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: no cells");
    return HAGRID_OK;
}

// One pass over the arrays of an unpacked blob: every index a traversal will follow stays inside its array.  A file is foreign
// data; the walk itself checks nothing.  flag bits: 1 entry, 2 cell, 4 reference.
__global__ void validate_arrays(const uint32_t* entries, int num_entries, const int4* cells, const uint4* small_cells, int num_cells,
                                const int* refs, int num_refs, int num_tris, int* flag) {
    const int stride = gridDim.x * blockDim.x;
    int bad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < num_entries; i += stride) {
        const uint32_t e = entries[i], log_dim = e & 3u, begin = e >> 2;
        if (log_dim == 0) { if (begin >= uint32_t(num_cells)) bad |= 1; }
        else if ((unsigned long long)begin + (1ull << (3 * log_dim)) > (unsigned long long)num_entries || begin <= uint32_t(i)) bad |= 1;   // children lie behind their parent
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < num_cells; i += stride) {
        if (cells) {
            const int4 lo = cells[2 * size_t(i)], hi = cells[2 * size_t(i) + 1];
            if (lo.w < 0 || hi.w < lo.w || hi.w > num_refs || lo.x > hi.x || lo.y > hi.y || lo.z > hi.z) bad |= 2;
        } else {
            const int begin = int(small_cells[i].w);
            if (begin < -1 || begin >= num_refs) bad |= 2;
        }
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < num_refs; i += stride) {
        const int r = refs[i];
        if (r >= num_tris || r < (cells ? 0 : -1)) bad |= 4;
    }
    if (!cells && blockIdx.x == 0 && threadIdx.x == 0 && num_refs > 0 && refs[num_refs - 1] != -1) bad |= 4;      // the last list ends
    if (bad) atomicOr(flag, bad);
}

int validate_unpacked(hagrid_ctx* ctx, const hagrid_grid& g, int num_tris) {
    int* flag = ctx->dscratch + kScrBlobFlag;
    HG_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    const long long most = std::max<long long>(std::max(g.num_entries, g.num_cells), g.num_refs);
    const int blocks = int(std::min<long long>((most + 255) / 256, 4096));
    validate_arrays<<<std::max(blocks, 1), 256, 0, ctx->stream>>>(static_cast<const uint32_t*>(g.entries), g.num_entries, static_cast<const int4*>(g.cells),
                                                                  static_cast<const uint4*>(g.small_cells), g.num_cells, static_cast<const int*>(g.ref_ids), g.num_refs, num_tris, flag);
    HG_HIP(ctx, hipGetLastError());
    int bad = 0;
    HG_TRY(read_back(ctx, flag, &bad, sizeof(int)));
    if (bad & 1) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: a voxel-map entry points outside the entries / cells");
    if (bad & 2) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: a cell's reference range lies outside the reference array");
    if (bad & 4) HG_FAIL(ctx, HAGRID_EINVAL, "grid blob: a reference names no triangle (or the last list has no end)");
    return HAGRID_OK;
}

void grid_from_header(const hagrid_blob_header& h, char* base, hagrid_grid* g, void** tris, int* num_tris) {
    memset(g, 0, sizeof(*g));
    g->entries = base + h.off_entries;
    g->ref_ids = base + h.off_refs;
    if (h.compressed) g->small_cells = base + h.off_cells; else g->cells = base + h.off_cells;
    for (int i = 0; i < 3; i++) { g->dims[i] = h.dims[i]; g->bbox_min[i] = h.bbox_min[i]; g->bbox_max[i] = h.bbox_max[i]; }
    g->num_cells = h.num_cells; g->num_entries = h.num_entries; g->num_refs = h.num_refs; g->shift = h.shift;
    g->num_offsets = h.num_offsets;
    for (int i = 0; i < h.num_offsets; i++) g->offsets[i] = h.offsets[i];
    if (tris) *tris = base + h.off_tris;
    if (num_tris) *num_tris = h.num_tris;
}

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------------
typedef int (*nccl_broadcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_comm_count_fn)(void*, int*);
typedef const char* (*nccl_error_fn)(int);
struct Rccl {
    nccl_broadcast_fn broadcast = nullptr;
    nccl_all_reduce_fn all_reduce = nullptr;
    nccl_comm_count_fn comm_count = nullptr, comm_rank = nullptr;
    nccl_error_fn error_string = nullptr;
    bool tried = false;
};
Rccl& rccl() {
    static Rccl r;
    if (!r.tried) {
        r.tried = true;
        // the copy the process already holds (PyTorch's, or the one the C++ program linked), else ROCm's
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        void* sym = h ? dlsym(h, "ncclBroadcast") : dlsym(RTLD_DEFAULT, "ncclBroadcast");
        void* err = h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString");
        r.broadcast = reinterpret_cast<nccl_broadcast_fn>(sym);
        auto find = [&](const char* name) { return h ? dlsym(h, name) : dlsym(RTLD_DEFAULT, name); };
        r.all_reduce = reinterpret_cast<nccl_all_reduce_fn>(find("ncclAllReduce"));
        r.comm_count = reinterpret_cast<nccl_comm_count_fn>(find("ncclCommCount"));
        r.comm_rank = reinterpret_cast<nccl_comm_count_fn>(find("ncclCommUserRank"));
        r.error_string = reinterpret_cast<nccl_error_fn>(err);
    }
    return r;
}
constexpr int kNcclUint8 = 1;      // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)
constexpr int kNcclInt32 = 2, kNcclSum = 0;

} // namespace

extern "C" size_t hagrid_grid_blob_bytes(const hagrid_grid* grid, int num_tris) {
    hagrid_blob_header h;
    if (fill_header(nullptr, grid, num_tris, h) != HAGRID_OK) return 0;
    return size_t(h.total_bytes);
}

extern "C" int hagrid_grid_pack(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris, int num_tris, void** blob, size_t* bytes) {
    if (!ctx || !blob) return HAGRID_EINVAL;
    *blob = nullptr;
    hagrid_blob_header h;
    HG_TRY(fill_header(ctx, grid, num_tris, h));
    if (num_tris > 0 && !tris) HG_FAIL(ctx, HAGRID_EINVAL, "grid_pack: no triangles");
    HG_HIP(ctx, hipSetDevice(ctx->device));
    char* b = static_cast<char*>(hagrid_mem_alloc(ctx, size_t(h.total_bytes)));
    if (!b) return HAGRID_ENOMEM;
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(b, &h, sizeof(h), hipMemcpyHostToDevice, st);
    auto copy = [&](size_t off, const void* src, size_t n) { if (e == hipSuccess && n) e = hipMemcpyAsync(b + off, src, n, hipMemcpyDeviceToDevice, st); };
    copy(h.off_entries, grid->entries, size_t(h.num_entries) * 4);
    copy(h.off_cells, h.compressed ? grid->small_cells : grid->cells, size_t(h.num_cells) * (h.compressed ? 16 : 32));
    copy(h.off_refs, grid->ref_ids, size_t(h.num_refs) * 4);
    copy(h.off_tris, tris, size_t(num_tris) * 48);
    if (e == hipSuccess) e = hipStreamSynchronize(st);           // the header lives on this stack frame
    if (e != hipSuccess) { hagrid_mem_free(ctx, b); HG_FAIL(ctx, HAGRID_EHIP, hipGetErrorString(e)); }
    *blob = b;
    if (bytes) *bytes = size_t(h.total_bytes);
    return HAGRID_OK;
}

extern "C" int hagrid_grid_unpack(hagrid_ctx* ctx, void* blob, size_t bytes, hagrid_grid* grid, void** tris, int* num_tris) {
    if (!ctx || !blob || !grid || !tris) return HAGRID_EINVAL;
    if (bytes < sizeof(hagrid_blob_header)) HG_FAIL(ctx, HAGRID_EINVAL, "grid_unpack: too small");
    HG_HIP(ctx, hipSetDevice(ctx->device));
    hagrid_blob_header h;
    HG_TRY(read_back(ctx, blob, &h, sizeof(h)));
    HG_TRY(check_header(ctx, h, bytes));
    hagrid_grid g; void* t = nullptr; int n = 0;
    grid_from_header(h, static_cast<char*>(blob), &g, &t, &n);
    HG_TRY(validate_unpacked(ctx, g, n));
    void* parts[4] = { g.entries, g.cells ? g.cells : g.small_cells, g.ref_ids, t };
    HG_TRY(pool_split(ctx, blob, parts, 4));
    *grid = g; *tris = t;
    if (num_tris) *num_tris = n;
    return HAGRID_OK;
}

extern "C" int hagrid_grid_save(hagrid_ctx* ctx, const hagrid_grid* grid, const void* tris, int num_tris, const char* path) {
    if (!ctx || !path) return HAGRID_EINVAL;
    void* blob = nullptr; size_t bytes = 0;
    HG_TRY(hagrid_grid_pack(ctx, grid, tris, num_tris, &blob, &bytes));
    std::vector<char> host;
    try { host.resize(bytes); } catch (const std::bad_alloc&) { hagrid_mem_free(ctx, blob); HG_FAIL(ctx, HAGRID_ENOMEM, "grid_save: no host memory for the blob"); }
    int rc = hagrid_mem_copy_d2h(ctx, host.data(), blob, bytes);
    hagrid_mem_free(ctx, blob);
    if (rc != HAGRID_OK) return rc;
    FILE* f = fopen(path, "wb");
    if (!f) HG_FAIL(ctx, HAGRID_EINVAL, "grid_save: cannot open the file for writing");
    const size_t done = fwrite(host.data(), 1, bytes, f);
    const int closed = fclose(f);
    if (done != bytes || closed != 0) HG_FAIL(ctx, HAGRID_EINVAL, "grid_save: short write");
    return HAGRID_OK;
}

extern "C" int hagrid_grid_load(hagrid_ctx* ctx, const char* path, hagrid_grid* grid, void** tris, int* num_tris) {
    if (!ctx || !path || !grid || !tris) return HAGRID_EINVAL;
    FILE* f = fopen(path, "rb");
    if (!f) HG_FAIL(ctx, HAGRID_EINVAL, "grid_load: cannot open the file");
    hagrid_blob_header h;
    if (fread(&h, 1, sizeof(h), f) != sizeof(h)) { fclose(f); HG_FAIL(ctx, HAGRID_EINVAL, "grid_load: no header"); }
    int rc = check_header(ctx, h, size_t(h.total_bytes));
    if (rc != HAGRID_OK) { fclose(f); return rc; }
    std::vector<char> host;
    try { host.resize(size_t(h.total_bytes)); } catch (const std::bad_alloc&) { fclose(f); HG_FAIL(ctx, HAGRID_ENOMEM, "grid_load: no host memory for the blob (header asks for more than there is)"); }
    memcpy(host.data(), &h, sizeof(h));
    const size_t rest = size_t(h.total_bytes) - sizeof(h);
    const size_t got = fread(host.data() + sizeof(h), 1, rest, f);
    fclose(f);
    if (got != rest) HG_FAIL(ctx, HAGRID_EINVAL, "grid_load: truncated file");
    void* blob = hagrid_mem_alloc(ctx, host.size());
    if (!blob) return HAGRID_ENOMEM;
    rc = hagrid_mem_copy_h2d(ctx, blob, host.data(), host.size());
    if (rc == HAGRID_OK) rc = hagrid_grid_unpack(ctx, blob, host.size(), grid, tris, num_tris);
    if (rc != HAGRID_OK) hagrid_mem_free(ctx, blob);
    return rc;
}

extern "C" int hagrid_grid_broadcast(hagrid_ctx* ctx, void* comm, int rank, int root, hagrid_grid* grid, void** tris, int* num_tris) {
    if (!ctx || !comm || !grid || !tris || !num_tris) return HAGRID_EINVAL;
    Rccl& r = rccl();
    if (!r.broadcast || !r.all_reduce) HG_FAIL(ctx, HAGRID_EINVAL, "grid_broadcast: RCCL (librccl.so) is not available in this process");
    HG_HIP(ctx, hipSetDevice(ctx->device));
    auto nccl_ok = [&](int rc, const char* what) -> int {
        if (rc == 0) return HAGRID_OK;
        char msg[256];
        snprintf(msg, sizeof(msg), "grid_broadcast: %s failed: %s", what, r.error_string ? r.error_string(rc) : "RCCL error");
        return fail(ctx, HAGRID_EHIP, __FILE__, __LINE__, msg);
    };
    // the communicator is the one the caller says it is: a rank / root outside it would make the collectives below wait for ever
    if (r.comm_count && r.comm_rank) {
        int world = 0, me = -1;
        if (r.comm_count(comm, &world) != 0 || r.comm_rank(comm, &me) != 0) HG_FAIL(ctx, HAGRID_EINVAL, "grid_broadcast: not an RCCL communicator");
        if (me != rank || root < 0 || root >= world) HG_FAIL(ctx, HAGRID_EINVAL, "grid_broadcast: rank / root do not match the communicator");
    }
    // Every rank takes part in the same sequence of collectives whatever happens to it locally: header, agreement, [payload],
    // agreement.  A rank that fails (the root cannot pack, a receiver has no memory) says so in the agreement and all ranks return
    // an error together -- nobody is left waiting inside a collective.
    PoolTemps tmp(ctx);
    hagrid_blob_header* dh = tmp.get<hagrid_blob_header>(1);
    int* dflag = tmp.get<int>(1);
    if (!dh || !dflag) return HAGRID_ENOMEM;                   // (256 + 4 bytes: a rank that cannot get these cannot run at all)
    auto agree = [&](int local_rc) -> int {                    // > 0: that many ranks failed
        int bad = local_rc != HAGRID_OK ? 1 : 0;
        hipError_t e = hipMemcpyAsync(dflag, &bad, sizeof(int), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);            // `bad` lives on this frame
        if (e != hipSuccess) return fail(ctx, HAGRID_EHIP, __FILE__, __LINE__, hipGetErrorString(e));
        int rc = nccl_ok(r.all_reduce(dflag, dflag, 1, kNcclInt32, kNcclSum, comm, ctx->stream), "ncclAllReduce (agreement)");
        if (rc != HAGRID_OK) return rc;
        rc = read_back(ctx, dflag, &bad, sizeof(int));
        return rc != HAGRID_OK ? rc : bad;
    };
    void* blob = nullptr; size_t bytes = 0;
    hagrid_blob_header h;
    memset(&h, 0, sizeof(h));
    int rc = HAGRID_OK;
    // 1. the header: the receivers learn the size.  A root that cannot pack sends zeros (no magic).
    if (rank == root) {
        rc = hagrid_grid_pack(ctx, grid, *tris, *num_tris, &blob, &bytes);
        hipError_t e = rc == HAGRID_OK ? hipMemcpyAsync(dh, blob, sizeof(h), hipMemcpyDeviceToDevice, ctx->stream) : hipMemsetAsync(dh, 0, sizeof(h), ctx->stream);
        if (e != hipSuccess && rc == HAGRID_OK) rc = fail(ctx, HAGRID_EHIP, __FILE__, __LINE__, hipGetErrorString(e));
    }
    int sent = nccl_ok(r.broadcast(dh, dh, sizeof(h), kNcclUint8, root, comm, ctx->stream), "ncclBroadcast (header)");
    if (rc == HAGRID_OK) rc = sent;
    if (rc == HAGRID_OK) rc = read_back(ctx, dh, &h, sizeof(h));
    if (rc == HAGRID_OK && rank != root) {
        rc = check_header(ctx, h, size_t(h.total_bytes));
        if (rc == HAGRID_OK) { bytes = size_t(h.total_bytes); blob = hagrid_mem_alloc(ctx, bytes); if (!blob) rc = HAGRID_ENOMEM; }
    }
    int failed = agree(rc);
    if (failed != 0) {
        if (blob) hagrid_mem_free(ctx, blob);
        if (rc != HAGRID_OK) return rc;
        if (failed < 0) return failed;
        HG_FAIL(ctx, HAGRID_EINVAL, "grid_broadcast: another rank could not take part (its grid is incomplete or it has no memory for the blob)");
    }
    // 2. the payload, straight from / into pool memory (the header travels again: the blob stays self-describing)
    rc = nccl_ok(r.broadcast(blob, blob, bytes, kNcclUint8, root, comm, ctx->stream), "ncclBroadcast (payload)");
    if (rc == HAGRID_OK) { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) rc = fail(ctx, HAGRID_EHIP, __FILE__, __LINE__, hipGetErrorString(e)); }
    if (rank == root) hagrid_mem_free(ctx, blob);              // the root keeps its own arrays
    else if (rc == HAGRID_OK) {
        rc = hagrid_grid_unpack(ctx, blob, bytes, grid, tris, num_tris);
        if (rc != HAGRID_OK) hagrid_mem_free(ctx, blob);
    } else hagrid_mem_free(ctx, blob);
    failed = agree(rc);
    if (rc != HAGRID_OK) return rc;
    if (failed < 0) return failed;
    if (failed > 0) HG_FAIL(ctx, HAGRID_EINVAL, "grid_broadcast: another rank could not unpack the grid");
    return HAGRID_OK;
}
