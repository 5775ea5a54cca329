// ctx.hip -- context, buffer pool (MemManager backend) and the event timer behind the C ABI.
//
// Reference interfaces replaced: MemManager's out-of-line members (mem_manager.h:106-112,
// mem_manager.cu:6-75: alloc_slot, free_slot, copy_*, zero_dev, one_dev, debug_slots) and
// profile() (profile.cu:5-18).
#include "ctx.h"

#include <chrono>

#include <algorithm>
#include <cstring>
#include <iostream>

using namespace hagrid_impl;

extern "C" int hagrid_abi_version(void) { return HAGRID_ABI_VERSION; }

extern "C" int hagrid_ctx_create(hagrid_ctx** out, int device, int keep) {
    if (!out) return HAGRID_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return HAGRID_ENODEV;
    hagrid_ctx* ctx = new hagrid_ctx();
    ctx->device = device;
    ctx->keep = keep != 0;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return HAGRID_ENODEV; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete ctx; return HAGRID_ENODEV; }
    ctx->num_cus = prop.multiProcessorCount;
    if (hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess ||
        hipHostMalloc((void**)&ctx->mailbox, 320 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&ctx->dscratch, 256 * sizeof(int)) != hipSuccess) {
        hagrid_ctx_destroy(ctx);
        return HAGRID_EHIP;
    }
    memset(ctx->mailbox, 0, 320 * sizeof(int));      // (the host polls words of it that the device may never have written: row lengths, tile-order epochs)
    *out = ctx;
    return HAGRID_OK;
}

extern "C" void hagrid_ctx_destroy(hagrid_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& h : ctx->hints) h.rowlen_pending = false;               // (their copies and events are behind the synchronisation above)
    if (!ctx->image.borrowed && ctx->image.alive) ctx->image.alive->store(false);      // borrowers of this context's image: refused from now on
    for (auto& s : ctx->slots)
        if (s.ptr) (void)hipFree(s.ptr);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    for (auto& h : ctx->hints) { if (h.rowlen_evt) (void)hipEventDestroy(h.rowlen_evt); if (h.lpt_buf) (void)hipFree(h.lpt_buf); for (auto& e : h.trial_evt) if (e) (void)hipEventDestroy(e); for (auto& pr : h.share_evt) for (auto& e : pr) if (e) (void)hipEventDestroy(e); }
    if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
    if (ctx->dscratch) (void)hipFree(ctx->dscratch);
    if (ctx->bin_diff) (void)hipFree(ctx->bin_diff);
    if (ctx->row_scores) (void)hipFree(ctx->row_scores);
    if (ctx->lb_state) (void)hipFree(ctx->lb_state);
    delete ctx;
}

extern "C" int hagrid_ctx_set_stream(hagrid_ctx* ctx, void* stream) {
    if (!ctx) return HAGRID_EINVAL;
    hipStream_t next = static_cast<hipStream_t>(stream);
    if (next == ctx->stream) return HAGRID_OK;
    // Pool slots freed in keep mode, the device scratch words and the look-back status words are re-used under the
    // assumption that all work of a context is ordered by ONE stream: drain the old stream before switching.
    HG_HIP(ctx, hipSetDevice(ctx->device));
    HG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = next;
    return HAGRID_OK;
}

extern "C" int hagrid_ctx_synchronize(hagrid_ctx* ctx) {
    if (!ctx) return HAGRID_EINVAL;
    HG_HIP(ctx, hipSetDevice(ctx->device));
    HG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HAGRID_OK;
}

extern "C" int hagrid_get_build_counts(const hagrid_ctx* ctx, hagrid_build_counts* out) {
    if (!ctx || !out) return HAGRID_EINVAL;
    *out = ctx->counts;
    return HAGRID_OK;
}

extern "C" const char* hagrid_last_error(const hagrid_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

extern "C" int hagrid_device_info(const hagrid_ctx* ctx, char* name, int name_len, int* compute_units, int64_t* total_mem) {
    if (!ctx) return HAGRID_EINVAL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return HAGRID_EHIP;
    if (name && name_len > 0) { strncpy(name, prop.gcnArchName, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (total_mem) *total_mem = (int64_t)prop.totalGlobalMem;
    return HAGRID_OK;
}

// ---- buffer pool ---------------------------------------------------------------------------------
// Contract of the reference's MemManager: alloc re-uses a free slot when possible, free(nullptr) is a
// no-op, a pointer not obtained from alloc is an error, keep mode retains freed buffers, usage() /
// max_usage() count bytes held from the device allocator.  Policy here: the smallest free slot that
// fits; otherwise the largest free slot is re-grown; otherwise a new slot.

static void release_slot_memory(hagrid_ctx* ctx, Slot& s) {
    if (s.ptr) (void)hipFree(s.ptr);
    ctx->usage -= s.size;
    s.ptr = nullptr;
    s.size = 0;
}

extern "C" void* hagrid_mem_alloc(hagrid_ctx* ctx, size_t bytes) {
    if (!ctx) return nullptr;
    if (bytes == 0) bytes = 4;
    bytes = (bytes + 255) & ~size_t(255);
    int fit = -1, biggest = -1, empty = -1;
    for (int i = 0, n = (int)ctx->slots.size(); i < n; i++) {
        const Slot& s = ctx->slots[i];
        if (s.in_use) continue;
        if (!s.ptr) { if (empty < 0) empty = i; continue; }
        if (s.size >= bytes && (fit < 0 || s.size < ctx->slots[fit].size)) fit = i;
        if (biggest < 0 || s.size > ctx->slots[biggest].size) biggest = i;
    }
    int idx = fit;
    // a fitting slot more than 4x too large is left for a bigger request unless nothing else is free
    if (idx >= 0 && ctx->slots[idx].size > 4 * bytes && ctx->slots[idx].size > (size_t(64) << 20)) idx = -1;
    if (idx < 0) {
        idx = fit < 0 ? (biggest >= 0 && ctx->keep ? biggest : empty) : empty;
        if (idx < 0) { idx = (int)ctx->slots.size(); ctx->slots.emplace_back(); }
        Slot& s = ctx->slots[idx];
        if (s.ptr) release_slot_memory(ctx, s);
        void* p = nullptr;
        (void)hipSetDevice(ctx->device);
        if (hipMalloc(&p, bytes) != hipSuccess) {
            // give back every cached buffer and retry once
            for (auto& t : ctx->slots) if (!t.in_use && t.ptr) release_slot_memory(ctx, t);
            if (hipMalloc(&p, bytes) != hipSuccess) { fail(ctx, HAGRID_ENOMEM, __FILE__, __LINE__, "hipMalloc failed"); return nullptr; }
        }
        s.ptr = p; s.size = bytes;
        ctx->usage += bytes;
        ctx->max_usage = std::max(ctx->max_usage, ctx->usage);
    }
    Slot& s = ctx->slots[idx];
    s.in_use = true;
    s.refs = 1;
    ctx->tracker[s.ptr] = idx;
    return s.ptr;
}

int hagrid_impl::pool_split(hagrid_ctx* ctx, void* base, void* const* parts, int n) {
    auto it = ctx->tracker.find(base);
    if (it == ctx->tracker.end() || n <= 0) HG_FAIL(ctx, HAGRID_EINVAL, "pool_split: not a pool buffer");
    const int idx = it->second;
    Slot& s = ctx->slots[idx];
    if (s.refs != 1 || s.ptr != base) HG_FAIL(ctx, HAGRID_EINVAL, "pool_split: buffer is already split");
    for (int i = 0; i < n; i++) {
        const char* p = static_cast<const char*>(parts[i]);
        if (p < static_cast<const char*>(base) || p >= static_cast<const char*>(base) + s.size) HG_FAIL(ctx, HAGRID_EINVAL, "pool_split: part outside the buffer");
        for (int j = 0; j < i; j++) if (parts[j] == parts[i]) HG_FAIL(ctx, HAGRID_EINVAL, "pool_split: parts coincide");
    }
    ctx->tracker.erase(it);
    for (int i = 0; i < n; i++) ctx->tracker[parts[i]] = idx;
    s.refs = n;
    return HAGRID_OK;
}

extern "C" int hagrid_mem_free(hagrid_ctx* ctx, void* ptr) {
    if (!ctx) return HAGRID_EINVAL;
    if (!ptr) return HAGRID_OK;
    auto it = ctx->tracker.find(ptr);
    if (it == ctx->tracker.end()) HG_FAIL(ctx, HAGRID_EINVAL, "free of a pointer that does not come from this MemManager");
    trav_image_source_touched(ctx, ptr, 1);       // a traversal image derived from this buffer goes with it
    it = ctx->tracker.find(ptr);
    Slot& s = ctx->slots[it->second];
    ctx->tracker.erase(it);
    if (--s.refs > 0) return HAGRID_OK;          // other parts of a split buffer are still alive
    s.in_use = false;
    if (!ctx->keep) {
        // buffers may still be read by work queued on the stream
        HG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        release_slot_memory(ctx, s);
    }
    return HAGRID_OK;
}

extern "C" int hagrid_mem_copy_h2d(hagrid_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return HAGRID_EINVAL;
    if (!bytes) return HAGRID_OK;
    trav_image_source_touched(ctx, dst, bytes);
    HG_HIP(ctx, hipSetDevice(ctx->device));
    HG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HAGRID_OK;
}
extern "C" int hagrid_mem_copy_d2h(hagrid_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return HAGRID_EINVAL;
    if (!bytes) return HAGRID_OK;
    HG_HIP(ctx, hipSetDevice(ctx->device));
    HG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HAGRID_OK;
}
extern "C" int hagrid_mem_copy_d2d(hagrid_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return HAGRID_EINVAL;
    if (!bytes) return HAGRID_OK;
    trav_image_source_touched(ctx, dst, bytes);
    HG_HIP(ctx, hipSetDevice(ctx->device));
    HG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return HAGRID_OK;
}
extern "C" int hagrid_mem_zero(hagrid_ctx* ctx, void* ptr, size_t bytes) {
    if (!ctx) return HAGRID_EINVAL;
    if (!bytes) return HAGRID_OK;
    trav_image_source_touched(ctx, ptr, bytes);
    HG_HIP(ctx, hipSetDevice(ctx->device));
    HG_HIP(ctx, hipMemsetAsync(ptr, 0, bytes, ctx->stream));
    return HAGRID_OK;
}
extern "C" int hagrid_mem_one(hagrid_ctx* ctx, void* ptr, size_t bytes) {
    if (!ctx) return HAGRID_EINVAL;
    if (!bytes) return HAGRID_OK;
    trav_image_source_touched(ctx, ptr, bytes);
    HG_HIP(ctx, hipSetDevice(ctx->device));
    HG_HIP(ctx, hipMemsetAsync(ptr, 0xFF, bytes, ctx->stream));
    return HAGRID_OK;
}
extern "C" size_t hagrid_mem_usage(const hagrid_ctx* ctx) { return ctx ? ctx->usage : 0; }
extern "C" size_t hagrid_mem_max_usage(const hagrid_ctx* ctx) { return ctx ? ctx->max_usage : 0; }

extern "C" void hagrid_mem_debug_slots(const hagrid_ctx* ctx) {
    if (!ctx) return;
    size_t total = 0;
    std::cout << "SLOTS: " << std::endl;
    for (const auto& s : ctx->slots) {
        std::cout << "[" << (s.in_use ? 'X' : ' ') << "] " << (double)s.size / (1024.0 * 1024.0) << "MB" << std::endl;
        total += s.size;
    }
    std::cout << (double)total / (1024.0 * 1024.0) << "MB total" << std::endl;
}

// ---- profile ---------------------------------------------------------------------------------------

extern "C" int hagrid_profile_begin(hagrid_ctx* ctx) {
    if (!ctx) return HAGRID_EINVAL;
    HG_HIP(ctx, hipSetDevice(ctx->device));
    HG_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    return HAGRID_OK;
}
extern "C" float hagrid_profile_end(hagrid_ctx* ctx) {
    if (!ctx) return -1.0f;
    float ms = -1.0f;
    if (hipEventRecord(ctx->ev_end, ctx->stream) != hipSuccess) return -1.0f;
    if (hipEventSynchronize(ctx->ev_end) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end) != hipSuccess) return -1.0f;
    return ms;
}

// ---- measured bandwidth peak (SURVEY.md 8(d) "BW_peak": a device copy / triad figure from the same run) ---------------------
namespace {
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// four 16-byte accesses per array in flight per lane, non-temporal (read once / written once), one pass over the arrays
__global__ void __launch_bounds__(256) bw_copy_kernel(const f32x4_t* __restrict__ a, f32x4_t* __restrict__ c, size_t n) {
    const size_t base = size_t(blockIdx.x) * 1024 + threadIdx.x;
    f32x4_t v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j * 256 < n) v[j] = __builtin_nontemporal_load(a + base + j * 256);
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j * 256 < n) __builtin_nontemporal_store(v[j], c + base + j * 256);
}
__global__ void __launch_bounds__(256) bw_triad_kernel(const f32x4_t* __restrict__ a, const f32x4_t* __restrict__ b, f32x4_t* __restrict__ c, size_t n) {
    const size_t base = size_t(blockIdx.x) * 1024 + threadIdx.x;
    f32x4_t v[4], w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j * 256 < n) { v[j] = __builtin_nontemporal_load(a + base + j * 256); w[j] = __builtin_nontemporal_load(b + base + j * 256); }
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j * 256 < n) __builtin_nontemporal_store(v[j] + 3.0f * w[j], c + base + j * 256);
}
} // namespace

extern "C" int hagrid_bandwidth_probe(hagrid_ctx* ctx, size_t bytes, int iters, float* copy_gbps, float* triad_gbps) {
    if (!ctx || iters <= 0 || bytes < (size_t(1) << 20)) return HAGRID_EINVAL;
    HG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = bytes / 16;
    f32x4_t* a = pool_alloc<f32x4_t>(ctx, n);
    f32x4_t* b = pool_alloc<f32x4_t>(ctx, n);
    f32x4_t* c = pool_alloc<f32x4_t>(ctx, n);
    int rc = HAGRID_OK;
    if (!a || !b || !c) rc = HAGRID_ENOMEM;
    if (rc == HAGRID_OK) {
        (void)hipMemsetAsync(a, 0, n * 16, ctx->stream); (void)hipMemsetAsync(b, 0, n * 16, ctx->stream);
        const int blocks = int((n + 1023) / 1024);
        float best_copy = 0.0f, best_triad = 0.0f;
        for (int it = 0; it < iters + 1 && rc == HAGRID_OK; it++) {           // the first round is a warm-up
            float ms = -1.0f;
            if (hagrid_profile_begin(ctx) != HAGRID_OK) { rc = HAGRID_EHIP; break; }
            bw_copy_kernel<<<blocks, 256, 0, ctx->stream>>>(a, c, n); HG_DBG(ctx);
            ms = hagrid_profile_end(ctx);
            if (ms > 0.0f && it) best_copy = std::max(best_copy, float(2.0 * double(n) * 16.0 / (double(ms) * 1e6)));
            if (hagrid_profile_begin(ctx) != HAGRID_OK) { rc = HAGRID_EHIP; break; }
            bw_triad_kernel<<<blocks, 256, 0, ctx->stream>>>(a, b, c, n); HG_DBG(ctx);
            ms = hagrid_profile_end(ctx);
            if (ms > 0.0f && it) best_triad = std::max(best_triad, float(3.0 * double(n) * 16.0 / (double(ms) * 1e6)));
        }
        if (copy_gbps) *copy_gbps = best_copy;
        if (triad_gbps) *triad_gbps = best_triad;
        if (hipGetLastError() != hipSuccess) rc = HAGRID_EHIP;
    }
    hagrid_mem_free(ctx, a); hagrid_mem_free(ctx, b); hagrid_mem_free(ctx, c);
    return rc;
}

unsigned long long* hagrid_impl::lookback_state(hagrid_ctx* ctx, int tiles, int words_per_tile, unsigned* epoch) {
    const size_t need = size_t(tiles > 0 ? tiles : 1) * size_t(words_per_tile);
    if (need > ctx->lb_words || ctx->lb_epoch >= (1u << 30) - 2u) {
        // grow (or restart the epochs): the words must start out as "never published"
        (void)hipStreamSynchronize(ctx->stream);
        if (need > ctx->lb_words) {
            if (ctx->lb_state) (void)hipFree(ctx->lb_state);
            ctx->lb_state = nullptr; ctx->lb_words = 0;
            const size_t words = need + need / 2 + 1024;
            if (hipMalloc((void**)&ctx->lb_state, words * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            ctx->lb_words = words;
        }
        (void)hipMemsetAsync(ctx->lb_state, 0, ctx->lb_words * sizeof(unsigned long long), ctx->stream);
        ctx->lb_epoch = 0;
    }
    *epoch = ++ctx->lb_epoch;
    return ctx->lb_state;
}

void hagrid_impl::debug_sync(hagrid_ctx* ctx, const char* file, int line) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "%s(%d): %s\n", file, line, hipGetErrorString(e));
        abort();
    }
}

extern "C" int hagrid_debug_sync_enabled(void) {
#ifdef HAGRID_DEBUG_SYNC
    return 1;
#else
    return 0;
#endif
}

// A scalar read-back is a host round trip in the middle of a chain of dependent kernels (a dozen per construction).  As hipMemcpyAsync + hipStreamSynchronize it is
// a blit kernel (~4.5 us), the synchronisation's wake-up and the next launch's latency.  Here ONE wavefront copies the words into the pinned mailbox and stores an
// epoch behind them (system scope); the host spins on the epoch -- it sees the words about a microsecond after the wavefront ends.  A launch that never ends (a fault
// upstream) is caught by the fall-back: after 20 ms without the epoch the stream is synchronised the ordinary way, which reports the error.
namespace {
__global__ void __launch_bounds__(256) publish_words(const int* __restrict__ src, int n, int* dst, int* flag, int epoch) {
    if (int(threadIdx.x) < n) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
} // namespace

int hagrid_impl::read_back(hagrid_ctx* ctx, const void* dptr, void* hptr, size_t bytes) {
    if (ctx->opt_fast_readback && bytes > 0 && bytes <= 256 * sizeof(int) && (bytes & 3u) == 0 && (reinterpret_cast<uintptr_t>(dptr) & 3u) == 0) {
        int* flag = ctx->mailbox + kMbxReadBackEpoch;
        const int epoch = ++ctx->readback_epoch;
        publish_words<<<1, 256, 0, ctx->stream>>>(static_cast<const int*>(dptr), int(bytes / 4), ctx->mailbox, flag, epoch);
        HG_HIP(ctx, hipGetLastError());
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == epoch) break;
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
                HG_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (a healthy launch is done by now; a faulted one reports here)
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != epoch) HG_FAIL(ctx, HAGRID_EHIP, "read_back: the words never arrived");
                break;
            }
        }
        memcpy(hptr, ctx->mailbox, bytes);
        return HAGRID_OK;
    }
    if (bytes <= 256 * sizeof(int)) {
        HG_HIP(ctx, hipMemcpyAsync(ctx->mailbox, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(hptr, ctx->mailbox, bytes);
    } else {
        HG_HIP(ctx, hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return HAGRID_OK;
}
