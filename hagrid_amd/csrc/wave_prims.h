// wave_prims.h -- hand-written wave64 / workgroup primitives for the construction passes.
//
// These replace the reference's CUB wrappers (src/parallel.cuh:12-89: DeviceScan::ExclusiveSum,
// DeviceReduce::Reduce, DevicePartition::Flagged, DeviceRadixSort::SortPairs):
//
//   * device_scan<V>(...)        ordered exclusive scan with a fused input functor and a fused output
//                                functor (the TransformInputIterator + follow-up kernel pairs of
//                                build.cu:557-560, :597, flatten.cu:136), chained on the device through a
//                                carry word so consecutive scans need no host round trip;
//   * wave_append / block_sum    unordered compaction: one atomicAdd per wavefront (ballot / prefix over
//                                64 lanes), used wherever only the SET of items matters;
//   * block_reduce_*             shuffle reductions.
//
// Workgroups are 256 threads = 4 wavefronts of 64 lanes.
#pragma once
#include "ctx.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace hagrid_impl {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kScanItems = 8;                        // items per thread in device_scan
constexpr int kScanTile = kBlock * kScanItems;       // items per workgroup

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// The hardware deals consecutive workgroups to the eight XCDs in turn, and every XCD has its own L2.  A kernel that gathers from the neighbourhood of
// its cell (the cells next to it, the references and triangles it shares with them) and takes its cells in the order of blockIdx.x has every such line
// fetched into several L2s.  xcd_block gives workgroup b of nb a position such that the XCDs take CHUNKS of 2^kXcdChunkLog2 consecutive positions in
// turn (a bijection on [0, nb)): neighbours meet in one L2, and the eight XCDs still advance through the array side by side (an eighth of the array
// per XCD leaves them unevenly loaded: the all-cells passes of the expansion ran 2 - 12 % slower).  Only for kernels whose workgroups are
// independent of each other's order (no look-back).
#ifndef HG_XCD_CHUNK_LOG2
#define HG_XCD_CHUNK_LOG2 6
#endif
constexpr int kXcdChunkLog2 = HG_XCD_CHUNK_LOG2;
__device__ __forceinline__ int xcd_block(int b, int nb) {
    if (kXcdChunkLog2 < 0) return b;
    const int full = (nb >> (kXcdChunkLog2 + 3)) << (kXcdChunkLog2 + 3);       // positions in complete groups of 8 chunks; the rest keeps its order
    if (b >= full) return b;
    const int xcd = b & 7, j = b >> 3;
    return ((((j >> kXcdChunkLog2) << 3) + xcd) << kXcdChunkLog2) + (j & ((1 << kXcdChunkLog2) - 1));
}

// ---- value types for scans: int and a pair of ints ---------------------------------------------------
struct Int2 { int a, b; };   // trivial: lives in __shared__ arrays
__host__ __device__ inline Int2 operator+(Int2 x, Int2 y) { return Int2{x.a + y.a, x.b + y.b}; }
__host__ __device__ inline int zero_of(int) { return 0; }
__host__ __device__ inline Int2 zero_of(Int2) { return Int2{0, 0}; }

__device__ __forceinline__ int shfl_up_v(int v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ Int2 shfl_up_v(Int2 v, int d) { return Int2{__shfl_up(v.a, d, 64), __shfl_up(v.b, d, 64)}; }
__device__ __forceinline__ int shfl_xor_v(int v, int d) { return __shfl_xor(v, d, 64); }
__device__ __forceinline__ Int2 shfl_xor_v(Int2 v, int d) { return Int2{__shfl_xor(v.a, d, 64), __shfl_xor(v.b, d, 64)}; }
__device__ __forceinline__ int shfl_v(int v, int l) { return __shfl(v, l, 64); }
__device__ __forceinline__ Int2 shfl_v(Int2 v, int l) { return Int2{__shfl(v.a, l, 64), __shfl(v.b, l, 64)}; }

/// Inclusive prefix sum over the 64 lanes of a wavefront (Hillis-Steele on register shuffles).
template <typename V>
__device__ __forceinline__ V wave_inclusive_scan(V v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        V o = shfl_up_v(v, d);
        if (l >= d) v = v + o;
    }
    return v;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { float o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { float o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

/// Sum over the workgroup; result valid in every thread.  `lds` holds kWaves ints.
__device__ __forceinline__ int block_sum(int v, int* lds) {
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) lds[wave_id()] = v;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) r += lds[w];
    return r;
}

/// Unordered append: every lane asks for `n` consecutive slots of a global array whose fill count is
/// *counter; returns the lane's first slot.  One atomic per wavefront.
__device__ __forceinline__ int wave_append(int n, int* counter) {
    const int incl = wave_inclusive_scan(n);
    const int total = __shfl(incl, 63, 64);
    int base = 0;
    if (lane_id() == 63 && total > 0) base = atomicAdd(counter, total);
    base = __shfl(base, 63, 64);
    return base + incl - n;
}

// ---- ordered device-wide exclusive scan ------------------------------------------------------------
// out(i, exclusive_prefix(i) + carry) for i < n, where in(i) supplies the values.  Reduce-then-scan:
//   scan_partials : one partial sum per workgroup tile
//   scan_spine    : a single workgroup scans the partials (adds *carry_in), publishes the grand total
//   scan_apply    : each workgroup rescans its tile with its offset and calls out()
// The spine writes total_out[0] = carry + sum, which the next scan may take as its carry_in: chains of
// scans (one per grid level) run back to back without the host.

template <typename V, typename In>
__global__ void __launch_bounds__(kBlock) scan_partials(In in, int n, V* partials) {
    __shared__ V lds[kWaves];
    const int base = blockIdx.x * kScanTile;
    V s = zero_of(V());
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
        const int i = base + j * kBlock + threadIdx.x;
        if (i < n) s = s + in(i);
    }
    s = wave_inclusive_scan(s);
    if (lane_id() == 63) lds[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        V t = lds[0];
        for (int w = 1; w < kWaves; w++) t = t + lds[w];
        partials[blockIdx.x] = t;
    }
}

template <typename V>
__global__ void __launch_bounds__(kBlock) scan_spine(V* partials, int num_partials, const V* carry_in, V* total_out) {
    __shared__ V lds[kWaves];
    __shared__ V running;
    if (threadIdx.x == 0) running = carry_in ? *carry_in : zero_of(V());
    __syncthreads();
    for (int base = 0; base < num_partials; base += kBlock) {
        const int i = base + threadIdx.x;
        const V v = i < num_partials ? partials[i] : zero_of(V());
        const V incl = wave_inclusive_scan(v);
        if (lane_id() == 63) lds[wave_id()] = incl;
        __syncthreads();
        V excl = running;
        for (int w = 0; w < wave_id(); w++) excl = excl + lds[w];
        const V prev = shfl_up_v(incl, 1);
        if (lane_id() > 0) excl = excl + prev;
        if (i < num_partials) partials[i] = excl;
        __syncthreads();
        if (threadIdx.x == 0) {
            V t = running;
            for (int w = 0; w < kWaves; w++) t = t + lds[w];
            running = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = running;
}

template <typename V, typename In, typename Out>
__global__ void __launch_bounds__(kBlock) scan_apply(In in, Out out, int n, const V* partials) {
    __shared__ V lds[kWaves];
    __shared__ V running;
    if (threadIdx.x == 0) running = partials[blockIdx.x];
    __syncthreads();
    const int base = blockIdx.x * kScanTile;
#pragma unroll 1
    for (int j = 0; j < kScanItems; j++) {
        const int i = base + j * kBlock + threadIdx.x;
        if (base + j * kBlock >= n) break;
        const V v = i < n ? in(i) : zero_of(V());
        const V incl = wave_inclusive_scan(v);
        if (lane_id() == 63) lds[wave_id()] = incl;
        __syncthreads();
        V excl = running;
        for (int w = 0; w < wave_id(); w++) excl = excl + lds[w];
        const V prev = shfl_up_v(incl, 1);
        if (lane_id() > 0) excl = excl + prev;
        if (i < n) out(i, excl);
        __syncthreads();
        if (threadIdx.x == 0) {
            V t = running;
            for (int w = 0; w < kWaves; w++) t = t + lds[w];
            running = t;
        }
        __syncthreads();
    }
}

// ---- single-pass variant: decoupled look-back ------------------------------------------------------------------------
// One kernel, every input read once: a tile publishes its aggregate, looks back over its predecessors' published
// aggregates / inclusive prefixes (64 at a time, one per lane of wavefront 0), publishes its own inclusive prefix and scans
// its items from registers.  A status word carries (epoch, flag, 32-bit value) and is written / read with one agent-scope
// 64-bit atomic, so it is valid across the 8 L2s; the epoch makes last call's words invalid without clearing the array.
// Workgroup b owns the tiles b, b + G, b + 2G, ... in increasing order, and the launch has no more workgroups than an EMPTY device keeps
// resident of this kernel: then all of them run and the lowest unfinished tile always belongs to a running workgroup that has nothing
// left to wait for.  The device need not be empty, though (other contexts, streams, processes), and the contract promises nothing about
// dispatch order or residency -- so nobody waits for ever: a lane whose predecessor tile has not published anything after `help_after`
// polls computes that tile's aggregate ITSELF from the items (any workgroup may: an aggregate is a pure function of the input; scans in
// place: see the second look behind the sum) and goes on.  In the worst case a workgroup sums its way back to tile 0 alone: slow, never stuck.  (Round 3 first handed tiles out by a
// ticket counter instead: a same-address atomic per workgroup and tile, ~88 per us -- 3600-tile scans took 38 instead of 20 us.)
constexpr int kLbVec = 4;                            // consecutive items per lane and row
constexpr int kLbItems = 8;                          // items per thread of the look-back scan
constexpr int kLbWindows = 4;                        // predecessors examined per look-back round: 64 lanes x kLbWindows
constexpr int kLbTile = kBlock * kLbItems;
constexpr unsigned kLbAggregate = 1u, kLbPrefix = 2u;

__device__ __forceinline__ unsigned long long lb_pack(unsigned epoch, unsigned flag, int value) {
    return ((unsigned long long)epoch << 34) | ((unsigned long long)flag << 32) | (unsigned)value;
}
__device__ __forceinline__ void lb_store(unsigned long long* p, unsigned long long w) {
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lb_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_publish(unsigned long long* state, int tile, unsigned epoch, unsigned flag, int v) { lb_store(state + tile, lb_pack(epoch, flag, v)); }
__device__ __forceinline__ void lb_publish(unsigned long long* state, int tile, unsigned epoch, unsigned flag, Int2 v) {
    lb_store(state + 2 * size_t(tile), lb_pack(epoch, flag, v.a));
    lb_store(state + 2 * size_t(tile) + 1, lb_pack(epoch, flag, v.b));
}
// status of tile `t` once it is valid for this epoch: returns the flag, fills v
__device__ __forceinline__ unsigned lb_wait(const unsigned long long* state, int t, unsigned epoch, int& v) {
    unsigned long long w;
    do { w = lb_load(state + t); } while ((unsigned)(w >> 34) != epoch || ((w >> 32) & 3u) == 0u);
    v = int(unsigned(w));
    return unsigned(w >> 32) & 3u;
}
__device__ __forceinline__ unsigned lb_wait(const unsigned long long* state, int t, unsigned epoch, Int2& v) {
    // a tile publishes both words as `aggregate` and later both as `prefix`: read until the two agree
    int a, b;
    unsigned fa, fb;
    do {
        fa = lb_wait(state, 2 * t, epoch, a);
        fb = lb_wait(state, 2 * t + 1, epoch, b);
    } while (fa != fb);
    v = Int2{a, b};
    return fa;
}
// one look at tile `t`: false while its status is not valid for this epoch (Int2: while its two words disagree)
__device__ __forceinline__ bool lb_try(const unsigned long long* state, int t, unsigned epoch, int& v, unsigned& flag) {
    const unsigned long long w = lb_load(state + t);
    v = int(unsigned(w)); flag = unsigned(w >> 32) & 3u;
    return (unsigned)(w >> 34) == epoch && flag != 0u;
}
__device__ __forceinline__ bool lb_try(const unsigned long long* state, int t, unsigned epoch, Int2& v, unsigned& flag) {
    const unsigned long long a = lb_load(state + 2 * size_t(t)), b = lb_load(state + 2 * size_t(t) + 1);
    v = Int2{int(unsigned(a)), int(unsigned(b))}; flag = unsigned(a >> 32) & 3u;
    return (unsigned)(a >> 34) == epoch && (unsigned)(b >> 34) == epoch && flag != 0u && flag == (unsigned(b >> 32) & 3u);
}
template <typename V> constexpr int lb_words() { return int(sizeof(V) / sizeof(int)); }

// A functor may offer the kLbVec consecutive items of a lane in one go (In::load4(i, n, v): items i .. i + 3, zero past n;
// Out::store4(i, n, v)) so that it can use 16-byte accesses; the per-item call operator is the fallback.
__device__ __forceinline__ bool lb_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
template <typename V, typename In> __device__ __forceinline__ auto lb_load_row(const In& in, int i, int n, V* v, int) -> decltype(in.load4(i, n, v), void()) { in.load4(i, n, v); }
template <typename V, typename In> __device__ __forceinline__ void lb_load_row(const In& in, int i, int n, V* v, long) {
#pragma unroll
    for (int c = 0; c < kLbVec; c++) v[c] = i + c < n ? in(i + c) : zero_of(V());
}
template <typename V, typename Out> __device__ __forceinline__ auto lb_store_row(const Out& out, int i, int n, const V* v, int) -> decltype(out.store4(i, n, v), void()) { out.store4(i, n, v); }
template <typename V, typename Out> __device__ __forceinline__ void lb_store_row(const Out& out, int i, int n, const V* v, long) {
#pragma unroll
    for (int c = 0; c < kLbVec; c++) if (i + c < n) out(i + c, v[c]);
}

template <typename V, typename In, typename Out>
__global__ void __launch_bounds__(kBlock) scan_lookback(In in, Out out, int n, int tiles, unsigned long long* state, unsigned epoch, unsigned help_after, const V* carry_in, V* total_out) {
    __shared__ V lds[kWaves];
    __shared__ V tile_prefix;
    // A wavefront owns a contiguous run of the tile; a lane owns kLbVec CONSECUTIVE items per row (four 4-byte items = one 16-byte
    // access per array: the streaming rate of the part needs wide accesses), a row is 64 x kLbVec items.
    constexpr int kWaveItems = kLbItems * 64, kRows = kLbItems / kLbVec;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int base = tile * kLbTile + wave_id() * kWaveItems + lane_id() * kLbVec;
        // all loads of the tile first, then the scan of the wavefront's run entirely in registers (no workgroup barrier): after
        // this v[j] is the exclusive prefix of item j inside the run and `run` the run's total
        V v[kLbItems];
#pragma unroll
        for (int r = 0; r < kRows; r++) lb_load_row<V>(in, base + r * 64 * kLbVec, n, v + r * kLbVec, 0);
        V run = zero_of(V());
#pragma unroll
        for (int r = 0; r < kRows; r++) {
            V mine = zero_of(V());                                       // the lane's own items first (serial), one wavefront scan per row
#pragma unroll
            for (int c = 0; c < kLbVec; c++) { const V x = v[r * kLbVec + c]; v[r * kLbVec + c] = mine; mine = mine + x; }
            const V incl = wave_inclusive_scan(mine);
            const V before = shfl_up_v(incl, 1);
            const V lane_off = lane_id() > 0 ? run + before : run;
#pragma unroll
            for (int c = 0; c < kLbVec; c++) v[r * kLbVec + c] = v[r * kLbVec + c] + lane_off;
            run = run + V(shfl_v(incl, 63));
        }
        __syncthreads();                                                 // the previous tile's readers of lds / tile_prefix are done
        if (lane_id() == 0) lds[wave_id()] = run;
        __syncthreads();
        if (wave_id() == 0) {
            V agg = lds[0];
            for (int w = 1; w < kWaves; w++) agg = agg + lds[w];
            V excl = zero_of(V());
            if (tile == 0) {
                if (carry_in) excl = *carry_in;
            } else {
                if (lane_id() == 0) lb_publish(state, tile, epoch, kLbAggregate, agg);
                // look back over kLbWindows x 64 predecessors per round (the status loads of a round are independent: one
                // round trip for all of them), nearest first: add aggregates up to and including the nearest inclusive prefix
                int p = tile - 1;
                for (;;) {
                    V pv[kLbWindows];
                    unsigned flag[kLbWindows];
                    bool ok[kLbWindows];
#pragma unroll
                    for (int w = 0; w < kLbWindows; w++) {
                        pv[w] = zero_of(V()); flag[w] = kLbPrefix;      // lanes before tile 0 end the search with a zero
                        ok[w] = p - w * 64 - lane_id() < 0;
                    }
                    for (unsigned polls = 0;; polls++) {                 // all outstanding looks of a trip are in flight together
                        bool all = true;
#pragma unroll
                        for (int w = 0; w < kLbWindows; w++) {
                            if (!ok[w]) {
                                ok[w] = lb_try(state, p - w * 64 - lane_id(), epoch, pv[w], flag[w]);
                                if (!ok[w] && polls >= help_after) {     // its owner may not be running at all: sum the tile here
                                    const int t = p - w * 64 - lane_id();
                                    const int first = t * kLbTile, last = min(n, first + kLbTile);
                                    V sum = zero_of(V());
                                    for (int i = first; i < last; i++) sum = sum + in(i);
                                    if (t == 0 && carry_in) sum = sum + *carry_in;       // tile 0's inclusive prefix starts from the carry
                                    pv[w] = sum; flag[w] = t == 0 ? kLbPrefix : kLbAggregate; ok[w] = true;
                                    // Several callers scan IN PLACE (out(i) overwrites what in(i) reads).  An owner that was only late publishes and then
                                    // stores its outputs, so the serial sum above may have read a mix of items and outputs.  Look once more, ordered behind
                                    // the reads of the sum: a word published by now wins and the sum is dropped; a tile still unpublished has stored no output
                                    // yet (outputs follow the owner's publish), so everything the sum read was input.
                                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                                    V pub = zero_of(V()); unsigned pflag = 0u;
                                    if (lb_try(state, t, epoch, pub, pflag)) { pv[w] = pub; flag[w] = pflag; }
                                }
                            }
                            all = all && ok[w];
                        }
                        if (all) break;
                    }
                    unsigned long long is_prefix[kLbWindows];
#pragma unroll
                    for (int w = 0; w < kLbWindows; w++) is_prefix[w] = __ballot(flag[w] == kLbPrefix);
                    bool found = false;
#pragma unroll
                    for (int w = 0; w < kLbWindows; w++) {
                        if (found) break;
                        const int first = __ffsll((long long)is_prefix[w]) - 1;      // nearest predecessor of this window with a prefix
                        if (first < 0 || lane_id() <= first) excl = excl + pv[w];
                        found = first >= 0;
                    }
                    if (found) break;
                    p -= 64 * kLbWindows;
                }
                // sum over the lanes
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) excl = excl + V(shfl_xor_v(excl, d));
            }
            if (lane_id() == 0) {
                lb_publish(state, tile, epoch, kLbPrefix, excl + agg);
                // In-place callers (out(i) overwrites what in(i) reads): a helper that sums this tile from its items (above) relies on "status still unpublished =>
                // no output stored yet".  The publishes are agent-scope atomic stores of this lane; the outputs are plain stores of every thread behind the
                // barrier below.  Waiting here until the atomics are acknowledged puts them in front of every output at the memory side.  (An agent-scope
                // release fence in front of the outputs says the same formally and writes the L2 back per tile: +0.6 ms per construction, measured.)
                // This is a gfx9 argument (stores and atomics are both counted by vmcnt there; gfx10+ count stores in vscnt): the library is built for gfx950 only.
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "scan_lookback: the in-place ordering relies on gfx9's vmcnt counting stores; use a release fence on this target"
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tile_prefix = excl;
                if (total_out && tile == tiles - 1) *total_out = excl + agg;
            }
        }
        __syncthreads();
        // what is left after the wait: one add and the output per item
        V offset = tile_prefix;
        for (int w = 0; w < wave_id(); w++) offset = offset + lds[w];
#pragma unroll
        for (int r = 0; r < kRows; r++) {
#pragma unroll
            for (int c = 0; c < kLbVec; c++) v[r * kLbVec + c] = offset + v[r * kLbVec + c];
            lb_store_row<V>(out, base + r * 64 * kLbVec, n, v + r * kLbVec, 0);
        }
    }
    if (tiles == 0 && blockIdx.x == 0 && threadIdx.x == 0 && total_out) *total_out = carry_in ? *carry_in : zero_of(V());
}

inline int scan_num_tiles(int n) { return (n + kScanTile - 1) / kScanTile; }

/// Single-pass scan.  `state` holds lb_words<V>() 64-bit words per tile and must never have seen `epoch` before
/// (hagrid_impl::lookback_state hands out both).
template <typename V, typename In, typename Out>
inline void device_scan_lookback(hipStream_t stream, int num_cus, In in, Out out, int n, unsigned long long* state, unsigned epoch, unsigned help_after, const V* carry_in, V* total_out) {
    // workgroups that are resident together: what the runtime reports for this instantiation, less one per CU (the report can
    // be one too high, MI355X_MICROARCH.md "Residency and cooperative launch"), at most 8, at least 1
    static const int per_cu = [] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, scan_lookback<V, In, Out>, kBlock, 0) != hipSuccess) { (void)hipGetLastError(); n = 2; }
        return std::max(1, std::min(n - 1, 8));
    }();
    const int tiles = (n + kLbTile - 1) / kLbTile;
    const int blocks = std::max(1, std::min(tiles, std::max(num_cus, 1) * per_cu));
    scan_lookback<V, In, Out><<<blocks, kBlock, 0, stream>>>(in, out, n, tiles, state, epoch, help_after, carry_in, total_out);
}

/// Launches the three scan kernels.  `partials` must hold scan_num_tiles(n) values of V.
/// n == 0 still runs the spine so that total_out = carry.
template <typename V, typename In, typename Out>
inline void device_scan(hipStream_t stream, In in, Out out, int n, V* partials, const V* carry_in, V* total_out) {
    const int tiles = scan_num_tiles(n);
    if (tiles > 0) scan_partials<V, In><<<tiles, kBlock, 0, stream>>>(in, n, partials);
    scan_spine<V><<<1, kBlock, 0, stream>>>(partials, tiles, carry_in, total_out);
    if (tiles > 0) scan_apply<V, In, Out><<<tiles, kBlock, 0, stream>>>(in, out, n, partials);
}

/// The scan the construction passes call: the look-back form (the three-kernel form is selected by hagrid_kat_scan only).
/// Returns false if the status words could not be allocated.
template <typename V, typename In, typename Out>
inline bool ctx_scan(hagrid_ctx* ctx, In in, Out out, int n, V* partials, const V* carry_in, V* total_out) {
#ifdef HG_SCAN_THREE_KERNELS                         // the test library's cross-check (kat/scan_kat.hip): the product instantiates the look-back form only
    if (!ctx->opt_lookback) { device_scan<V>(ctx->stream, in, out, n, partials, carry_in, total_out); HG_DBG(ctx); return true; }
#endif
    unsigned epoch = 0;
    unsigned long long* state = lookback_state(ctx, scan_num_tiles(n), lb_words<V>(), &epoch);
    if (!state) return false;
    // (a status poll is an L2 round trip, ~1 us: 4096 polls are milliseconds -- far beyond any wait of a healthy launch; opt_lookback == 2,
    // a test setting, helps at the first miss so that the helping path is exercised)
    device_scan_lookback<V>(ctx->stream, ctx->num_cus, in, out, n, state, epoch, ctx->opt_lookback == 2 ? 0u : 4096u, carry_in, total_out);
    HG_DBG(ctx);
    return true;
}

} // namespace hagrid_impl
