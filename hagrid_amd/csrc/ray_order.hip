// ray_order.hip -- how a batch meets the lanes: the row length of an image-ordered batch, found on the device (it steers the tile
// packets of trav_common.h), and ray binning, the counting sort of an unordered batch on 512 Morton bins of the rays' entry points
// (north_star: "per-wavefront ray packets sorted in LDS to tame divergence").  Neither changes a hit: both only choose which lane
// traverses which ray.  No reference counterpart (the reference traverses rays in buffer order, traverse.cu:35-38).
#include "trav_common.h"
#include "wave_prims.h"

using namespace hagrid;
using namespace hagrid_impl;
using namespace hagrid_trav;

namespace {

// Row length of an image-ordered batch, or 0: the (origin, direction) of consecutive rays advances by a constant step
// s = ray[1] - ray[0] along a row (perspective: the direction; orthographic: the origin) and jumps at a row break.
// w = index of the first break; accepted if it is a multiple of 8, the second row starts with the same step and, when
// there is a third row, ray 2w is a break too.  One workgroup; the answer stays on the device (no host round trip).
// A wrong answer can only cost speed: any row length gives a valid lane <-> ray assignment.
constexpr int kDetectBlock = 1024;
constexpr int kDetectLimit = 1 << 16;

__device__ __forceinline__ float ray_step_dev2(const float4* __restrict__ rays, int i, const float (&s)[6]) {
    // squared distance between (ray[i+1] - ray[i]) and s over origin and direction
    const float4 a0 = rays[2 * size_t(i)], a1 = rays[2 * size_t(i) + 1], b0 = rays[2 * size_t(i) + 2], b1 = rays[2 * size_t(i) + 3];
    const float d[6] = {b0.x - a0.x - s[0], b0.y - a0.y - s[1], b0.z - a0.z - s[2], b1.x - a1.x - s[3], b1.y - a1.y - s[4], b1.z - a1.z - s[5]};
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
}

// Second criterion, for image-ordered batches whose directions are not a function of the pixel (bounce rays leaving the
// primary hit points): the ORIGINS of vertically neighbouring pixels are close.  Every candidate row length w = 64, 72, ...
// gets the clamped mean squared distance |org[i + w] - org[i]|^2 / tau^2 over 256 sampled i (tau = 1/64 of the grid
// diagonal); the true row length is the minimum (one pixel apart; w +- 8 is eight pixels apart, 2w two rows).  Accepted if
// it stands out from the mean over all candidates and horizontally neighbouring origins are as close (but not all identical).  Blocks 1.. of the same
// launch do the scoring, the block that finishes last picks -- no extra launch, nothing waits; used only when the first
// criterion found nothing.  Like the first one it can only cost speed if it is wrong.
constexpr int kRowCandidates = 2048;                 // w = 8 * (c + 8): 64 .. 16440
constexpr int kRowSamples = 256;

__global__ void __launch_bounds__(kDetectBlock) detect_ray_rows(const float4* __restrict__ rays, int n, int* __restrict__ out,
                                                                int* __restrict__ scores, float inv_tau2, int origins_only) {
    __shared__ int first_break;
    __shared__ int lds_score[kDetectBlock / 64];
    __shared__ int ticket;
    __shared__ unsigned long long best[kDetectBlock / 64];
    __shared__ long long sums[kDetectBlock / 64];
    __shared__ int counts[kDetectBlock / 64];
    // origins_only: a second launch after the first criterion; nothing to do if that one found the row length
    if (origins_only && __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) return;
    if (blockIdx.x == 0 && !origins_only) {
        if (threadIdx.x == 0) { first_break = 0x7fffffff; out[0] = 0; }
        int w1 = 0;
        if (n >= 128) {
            const float4 a0 = rays[0], a1 = rays[1], b0 = rays[2], b1 = rays[3];
            const float s[6] = {b0.x - a0.x, b0.y - a0.y, b0.z - a0.z, b1.x - a1.x, b1.y - a1.y, b1.z - a1.z};
            const float s2 = s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3] + s[4] * s[4] + s[5] * s[5];
            if ((s2 > 0.0f) && (s2 < 3.0e38f)) {
                const float tol = 0.25f * s2;
                const int limit = min(n - 1, kDetectLimit);              // pairs (i, i + 1) with i < limit
                __syncthreads();
                for (int base = 1; base < limit; base += kDetectBlock) {
                    const int i = base + int(threadIdx.x);
                    if (i < limit && !(ray_step_dev2(rays, i, s) <= tol)) atomicMin(&first_break, i + 1);
                    __syncthreads();
                    const int found = first_break;
                    __syncthreads();
                    if (found != 0x7fffffff) break;
                }
                if (threadIdx.x == 0) {
                    const int w = first_break;
                    bool ok = w != 0x7fffffff && w >= 8 && (w & 7) == 0 && n / w >= 8;
                    if (ok) ok = ray_step_dev2(rays, w, s) <= tol;                                   // second row advances like the first
                    if (ok && n > 2 * w) ok = !(ray_step_dev2(rays, 2 * w - 1, s) <= tol);            // and ends where the first did
                    w1 = ok ? w : 0;
                }
            }
        }
        if (threadIdx.x == 0) __hip_atomic_store(out, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gridDim.x == 1) return;
    } else if (blockIdx.x != 0) {
        // four candidates per block, one per group of 256 threads; candidate kRowCandidates is the horizontal neighbour (w = 1)
        const int c = (int(blockIdx.x) - 1) * 4 + int(threadIdx.x >> 8);
        const int w = c < kRowCandidates ? 8 * (c + 8) : 1;
        int v = 1024;
        if (c <= kRowCandidates && n - w > 0) {
            uint32_t h = uint32_t(c) * 2654435761u + (threadIdx.x & 255u) * 40503u + 12345u;
            h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
            const int i = int(h % uint32_t(n - w));
            const float4 p = rays[2 * size_t(i)], q = rays[2 * size_t(i + w)];
            const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
            const float d2 = (dx * dx + dy * dy + dz * dz) * inv_tau2;
            v = d2 < 1.0f ? int(d2 * 1024.0f) : 1024;            // NaN -> 1024
        }
        v = wave_sum(v);
        if (lane_id() == 0) lds_score[wave_id()] = v;
        __syncthreads();
        if ((threadIdx.x & 255) == 0 && c <= kRowCandidates) {
            const int g = int(threadIdx.x >> 8) * 4;
            __hip_atomic_store(scores + c, lds_score[g] + lds_score[g + 1] + lds_score[g + 2] + lds_score[g + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the block that finishes last picks.  Scores and out[0] travel as agent-scope atomics (a __threadfence per thread costs an
    // L2 write-back each on this part: 150 us for the launch); the ticket is the release / acquire point.
    __syncthreads();
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(scores + kRowCandidates + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != int(gridDim.x) - 1) return;
    unsigned long long m = ~0ull;                                    // (score << 32 | w), minimum
    long long total = 0; int counted = 0;                            // mean score of the candidates
    for (int c = int(threadIdx.x); c < kRowCandidates; c += kDetectBlock) {
        const int w = 8 * (c + 8);
        if (n / w >= 8) {
            const unsigned sc = (unsigned)__hip_atomic_load(scores + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long key = ((unsigned long long)sc << 32) | unsigned(w);
            m = key < m ? key : m;
            total += sc; counted++;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long o = __shfl_xor(m, d, 64); m = o < m ? o : m;
        total += __shfl_xor(total, d, 64); counted += __shfl_xor(counted, d, 64);
    }
    if (lane_id() == 0) { best[wave_id()] = m; sums[wave_id()] = total; counts[wave_id()] = counted; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kDetectBlock / 64; i++) { m = best[i] < m ? best[i] : m; total += sums[i]; counted += counts[i]; }
        // The row length stands out: its score lies clearly (8 % of the clamp) below the mean of all candidates, and so does the
        // score of horizontally neighbouring origins.  Unrelated origins score ~1.0 everywhere; rows of hit points with
        // silhouettes and rays that left the scene 0.2-0.9.  horizontal == 0: all origins coincide (a pinhole camera) -- no information.
        const int full = kRowSamples * 1024;
        const long long mean = counted ? total / counted : 0;
        const int horizontal = __hip_atomic_load(scores + kRowCandidates, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int w1 = __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long bar = mean - full * 8 / 100;
        if (w1 == 0 && m != ~0ull && (long long)(m >> 32) < bar && horizontal < bar && horizontal > 0) out[0] = int(unsigned(m)) | kRowsFromOrigins;       // (rays in image order WITHOUT coherent directions: traverse.hip's tile-order rule)
        scores[kRowCandidates + 1] = 0;                              // ready for the next batch
    }
}

// ---- ray binning (extension; north_star: "ray packets sorted ... to tame divergence") ----------------------------------
// A batch without spatial order (random origins and directions) makes every load of a wavefront touch 64 unrelated cache
// lines.  Measured on MI355X (tools/dev_sort_potential.py): ordering such a batch by a coarse Morton key of the ray's
// position -- 8 x 8 x 8 bins are enough, the direction octant does not matter -- lifts traversal from 1.0 to 2.3-2.6
// Grays/s.  So the device does a counting sort on 512 bins, not a general sort:
//   ray_bin_count   : key = Morton3(entry point of the ray into the grid box, 3 bits per axis); per-workgroup histogram in
//                     LDS, written to table[bin][workgroup]
//   device_scan     : exclusive scan of the table in (bin, workgroup) order = first slot of every (bin, workgroup) run
//   ray_bin_scatter : slot = run start + rank inside the run (LDS atomic), perm[slot] = ray index
// The order inside a bin is irrelevant.  No global atomics; one extra 4-byte word per ray.
// BITS bits per axis: 3 (512 bins).  (Rounds 3 - 5 took 4096 bins for working sets beyond 512 MB -- a bin's share of the image in an XCD's L2; same box, round 6,
// gpurun_out/r6geo: never faster -- the 8M-triangle soup with 16M / 64M rays +2 % with 4096 bins, scenes within the cache +4 ... +10 % -- and removed.)
constexpr int kBinItems = 16;                       // rays per thread
constexpr int kBinTile = kBlock * kBinItems;        // rays per workgroup

__device__ __forceinline__ uint32_t spread3(uint32_t x) {   // 4 bits -> every third bit
    return (x & 1u) | ((x & 2u) << 2) | ((x & 4u) << 4) | ((x & 8u) << 6);
}

template <int BITS>
__device__ __forceinline__ int ray_bin_key(const TraverseArgs& a, int id) {
    constexpr int kBinBits = BITS;
    const float4 r0 = a.rays[2 * size_t(id)], r1 = a.rays[2 * size_t(id) + 1];
    const vec3 org(r0.x, r0.y, r0.z), dir(r1.x, r1.y, r1.z);
    const vec3 inv_dir(safe_rcp(dir.x), safe_rcp(dir.y), safe_rcp(dir.z));
    const vec3 gmin(a.min_x, a.min_y, a.min_z), gmax(a.max_x, a.max_y, a.max_z);
    const vec3 ta = (gmin - org) * inv_dir, tb = (gmax - org) * inv_dir;
    const vec3 t0 = min(ta, tb);
    float ts = detail::fmax2(detail::fmax2(t0.x, detail::fmax2(t0.y, t0.z)), r0.w);
    if (!(ts == ts) || ts > 3.0e38f || ts < -3.0e38f) ts = 0.0f;
    const vec3 p = (ts * dir + org - gmin) / (gmax - gmin) * float(1 << kBinBits);
    const int m = (1 << kBinBits) - 1;
    const int x = min(max(int(detail::fmin2(detail::fmax2(p.x, 0.0f), float(m))), 0), m);
    const int y = min(max(int(detail::fmin2(detail::fmax2(p.y, 0.0f), float(m))), 0), m);
    const int z = min(max(int(detail::fmin2(detail::fmax2(p.z, 0.0f), float(m))), 0), m);
    return int(spread3(uint32_t(x)) | (spread3(uint32_t(y)) << 1) | (spread3(uint32_t(z)) << 2));
}

// auto mode: `skip_if` (the row length found by detect_ray_rows) > 0 means the batch is image-ordered and is left alone;
// `diff` (64 words) receives the number of neighbouring rays (i, i + 1) whose keys differ -- the coherence estimate
template <int BITS>
__global__ void __launch_bounds__(kBlock) ray_bin_count(const TraverseArgs a, unsigned short* __restrict__ keys, int* __restrict__ table,
                                                        const int* __restrict__ skip_if, int* __restrict__ diff) {
    constexpr int kBins = 1 << (3 * BITS);
    __shared__ int hist[kBins];
    __shared__ unsigned short tile_keys[kBinTile];
    __shared__ int lds[kWaves];
    if (skip_if && *skip_if > 0) return;
    for (int i = threadIdx.x; i < kBins; i += kBlock) hist[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * kBinTile;
    for (int j = 0; j < kBinItems; j++) {
        const int id = base + j * kBlock + threadIdx.x;
        if (id < a.num_rays) {
            const int k = ray_bin_key<BITS>(a, id);
            keys[id] = (unsigned short)k;
            if (diff) tile_keys[j * kBlock + threadIdx.x] = (unsigned short)k;
            atomicAdd(&hist[k], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += kBlock) table[size_t(i) * gridDim.x + blockIdx.x] = hist[i];
    if (diff) {
        int d = 0;
        for (int j = 0; j < kBinItems; j++) {
            const int i = j * kBlock + threadIdx.x;
            if (base + i + 1 < a.num_rays && i + 1 < kBinTile) d += tile_keys[i] != tile_keys[i + 1];
        }
        d = block_sum(d, lds);
        if (threadIdx.x == 0 && d) atomicAdd(diff + (blockIdx.x & 63), d);
    }
}

// auto mode: bin the batch iff it is not image-ordered and more than half of its neighbouring rays fall into different bins
__global__ void __launch_bounds__(64) ray_bin_decide(const int* __restrict__ row_len, int* __restrict__ diff, int num_rays, int* __restrict__ flag) {
    int d = diff[threadIdx.x];
    diff[threadIdx.x] = 0;                       // ready for the next batch
    d = wave_sum(d);
    if (threadIdx.x == 0) flag[0] = ((*row_len & kRowLenMask) == 0 && 2ll * d > num_rays) ? 1 : 0;
}

// The rays of a tile are first put in bin order inside LDS (local histogram -> local scan -> local rank), then written out: lanes
// that are neighbours in LDS write neighbouring words of `perm`, so a store instruction touches the runs of a few bins instead of 64
// unrelated lines (the lane-by-lane form moved 512 MB in 2.3 ms for 128M rays: bound by write transactions, not by bytes).
template <int BITS>
__global__ void __launch_bounds__(kBlock) ray_bin_scatter(const unsigned short* __restrict__ keys, const int* __restrict__ table_scan,
                                                          int num_rays, int* __restrict__ perm, const int* __restrict__ only_if) {
    constexpr int kBins = 1 << (3 * BITS), kPer = kBins / kBlock;       // bins per thread in the local scan
    static_assert(kBins % kBlock == 0, "whole bins per thread in the local scan");
    __shared__ int count[kBins];               // rays of the tile per bin (the cursor of the local ranks), then the first slot of the (bin, workgroup) run in perm
    __shared__ int lstart[kBins + 1];          // first LDS slot of every bin
    __shared__ int sorted_id[kBinTile];
    __shared__ unsigned short sorted_key[kBinTile];
    __shared__ int wsum[kWaves];
    if (only_if && *only_if == 0) return;
    for (int i = threadIdx.x; i < kBins; i += kBlock) count[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * kBinTile;
    int key[kBinItems], rank[kBinItems];
#pragma unroll
    for (int j = 0; j < kBinItems; j++) {
        const int id = base + j * kBlock + threadIdx.x;
        key[j] = id < num_rays ? int(keys[id]) : -1;
    }
#pragma unroll
    for (int j = 0; j < kBinItems; j++) rank[j] = key[j] >= 0 ? atomicAdd(&count[key[j]], 1) : 0;
    __syncthreads();
    {   // exclusive scan of the counts: kPer consecutive bins per thread, wavefront scan, wavefront sums through LDS
        int c[kPer], sum = 0;
#pragma unroll
        for (int q = 0; q < kPer; q++) { c[q] = count[kPer * threadIdx.x + q]; sum += c[q]; }
        const int incl = wave_inclusive_scan(sum);
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        int run = incl - sum;
        for (int w = 0; w < wave_id(); w++) run += wsum[w];
#pragma unroll
        for (int q = 0; q < kPer; q++) { lstart[kPer * threadIdx.x + q] = run; run += c[q]; }
        if (threadIdx.x == kBlock - 1) lstart[kBins] = run;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += kBlock) count[i] = table_scan[size_t(i) * gridDim.x + blockIdx.x];      // (the ranks are drawn: the array is free)
#pragma unroll
    for (int j = 0; j < kBinItems; j++)
        if (key[j] >= 0) {
            const int at = lstart[key[j]] + rank[j];
            sorted_id[at] = base + j * kBlock + threadIdx.x;
            sorted_key[at] = (unsigned short)key[j];
        }
    __syncthreads();
    const int total = lstart[kBins];
    for (int i = threadIdx.x; i < total; i += kBlock) {
        const int k = sorted_key[i];
        perm[count[k] + (i - lstart[k])] = sorted_id[i];
    }
}

// ---- tile order: longest first ---------------------------------------------------------------------------------------------
// order[0 .. n) = the tile indices by descending cost (clamped to 4095); cost[] is cleared for the next launches to fill.  One workgroup, a
// counting sort on the whole key: histogram in LDS (coalesced reads of the costs, LDS atomics), one scan over the 4096 bins, ranks from the
// bins' counters.  Tiles of equal cost keep their order at the granularity of a sweep (1024 tiles); inside a sweep the LDS atomics decide --
// the order only steers which wavefront takes which tile.  (A stable radix sort with a chunk of the input per thread took 70 us for
// 16 384 tiles, every gather of a cost a dependent load: 3 % of the launches it was meant to speed up.)
constexpr int kOrderBlock = 1024, kOrderBins = 4096;
// The sort also leaves behind WHAT the order was learned on: a copy of the sample ray order_still_fits (trav_common.h) compares the buffer with.
// rot: the order is stored rotated by this many positions -- its LAST rot positions hold the longest tiles, which the tail kernel dispatches first, four lanes per
// ray (trav_kernels.h, a.quad_head).
// suggest (pinned host word, polled): how many tiles cost at least head_tenths / 10 times the MEDIAN working tile, if they are more than a twelfth of the working
// tiles -- the share the next sort is rotated by (the host rounds and caps it): a launch whose tiles cost about the same has none, a scene with a few dense
// objects a tenth of its tiles.
__global__ void __launch_bounds__(kOrderBlock) tile_order_kernel(int* __restrict__ cost, int* __restrict__ order, int n, int rot, int head_tenths, int* suggest,
                                                                 const float4* __restrict__ rays, int num_rays, float4* __restrict__ samples) {
    __shared__ int median_cost;
    __shared__ int bins[kOrderBins];
    __shared__ int wave_total[kOrderBlock / 64];
    const int t = threadIdx.x;
    if (t < 2 && samples) samples[t] = rays[2 * order_sample_index(num_rays) + t];
    for (int k = t; k < kOrderBins; k += kOrderBlock) bins[k] = 0;
    __syncthreads();
    auto bin_of = [&](int c) { return kOrderBins - 1 - min(kOrderBins - 1, max(0, c)); };        // descending cost = ascending bin
    // (up to 32 768 tiles a thread keeps the bins of its tiles in registers: the second sweep over the costs would be sixteen dependent loads)
    constexpr int kKeep = 32;
    int mine[kKeep];
    const bool keep = n <= kKeep * kOrderBlock;
    if (keep) {
#pragma unroll
        for (int r = 0; r < kKeep; r++) { const int i = r * kOrderBlock + t; mine[r] = i < n ? bin_of(cost[i]) : -1; }
#pragma unroll
        for (int r = 0; r < kKeep; r++) if (mine[r] >= 0) atomicAdd(&bins[mine[r]], 1);
    } else {
        for (int i = t; i < n; i += kOrderBlock) atomicAdd(&bins[bin_of(cost[i])], 1);
    }
    __syncthreads();
    // exclusive scan over the bins: four consecutive bins per thread
    int c[4], sum = 0;
    for (int k = 0; k < 4; k++) { c[k] = bins[4 * t + k]; sum += c[k]; }
    const int incl = wave_inclusive_scan(sum);
    if ((t & 63) == 63) wave_total[t >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (t >> 6); w++) run += wave_total[w];
    if (t == 0) median_cost = 0;
    int start[4];
    for (int k = 0; k < 4; k++) { start[k] = run; bins[4 * t + k] = run; run += c[k]; }
    __syncthreads();
    // the median over the tiles that do any work (cost >= 3: the tiles of an image's empty margin would pull it to nothing); bins[b] = tiles of a cost above kOrderBins - 1 - b
    const int live = bins[kOrderBins - 3];
    for (int k = 0; k < 4; k++)
        if (c[k] > 0 && start[k] <= live / 2 && live / 2 < start[k] + c[k]) median_cost = kOrderBins - 1 - (4 * t + k);
    __syncthreads();
    if (t == 0 && suggest) {
        int sg = 0;
        if (head_tenths > 0 && live > 0) {
            const int thr = min(kOrderBins - 1, max(3, (median_cost * head_tenths + 9) / 10));        // cost >= thr  <=>  bin <= kOrderBins - 1 - thr
            sg = bins[kOrderBins - thr];
            if (sg * 12 < live) sg = 0;             // a thin tail (a twelfth of the working tiles or less): the launch is not as long as a few dense places
        }
        *suggest = sg;
    }
    __syncthreads();
    if (keep) {
#pragma unroll
        for (int r = 0; r < kKeep; r++) {
            if (r * kOrderBlock >= n) break;
            const int i = r * kOrderBlock + t;
            if (mine[r] >= 0) { const int p = atomicAdd(&bins[mine[r]], 1); order[p >= rot ? p - rot : p + (n - rot)] = i; cost[i] = 0; }
            __syncthreads();                                   // (sweep after sweep: equal costs stay in tile order across sweeps)
        }
    } else {
        for (int i0 = 0; i0 < n; i0 += kOrderBlock) {
            const int i = i0 + t;
            if (i < n) { const int p = atomicAdd(&bins[bin_of(cost[i])], 1); order[p >= rot ? p - rot : p + (n - rot)] = i; cost[i] = 0; }
            __syncthreads();
        }
    }
}

struct TableIn { const int* t; __device__ int operator()(int i) const { return t[i]; } };
struct TableOut { int* t; __device__ void operator()(int i, int s) const { t[i] = s; } };

} // namespace

// row length of an image-ordered batch -> row_len[0] on the device.  The origin criterion costs ~17 us (2049 candidates x 256
// sampled pairs) behind every 16th call over a buffer: it runs as a second launch for batches of at least kOriginMinRays rays (256K; rounds 5's 4M was stale -- with
// rows a launch gets tile packets AND its measured share of four-lane tiles: bounce rays at 1024^2 +11 % on the soup, +5 % on configuration 3's grid, +3 % clustered,
// +10 % on the stadium mesh, 1920 x 1080 +1 %: same box, round 6, gpurun_out/r6geo/origin_rows.txt) and returns at once when the first criterion has already answered.
void hagrid_trav::launch_detect(hagrid_ctx* ctx, const TraverseArgs& a, int num_rays, int* row_len, int origin_min_rays) {
    detect_ray_rows<<<1, kDetectBlock, 0, ctx->stream>>>(a.rays, num_rays, row_len, nullptr, 0.0f, 0); HG_DBG(ctx);
    const vec3 ext(a.max_x - a.min_x, a.max_y - a.min_y, a.max_z - a.min_z);
    const float tau = length(ext) / 64.0f;
    if (num_rays < origin_min_rays || !(tau > 0.0f) || !(tau < 3.0e18f)) return;
    if (!ctx->row_scores) {
        if (hipMalloc((void**)&ctx->row_scores, (kRowCandidates + 8) * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); ctx->row_scores = nullptr; return; }
        (void)hipMemsetAsync(ctx->row_scores, 0, (kRowCandidates + 8) * sizeof(int), ctx->stream);
    }
    detect_ray_rows<<<1 + (kRowCandidates + 1 + 3) / 4, kDetectBlock, 0, ctx->stream>>>(a.rays, num_rays, row_len, ctx->row_scores, 1.0f / (tau * tau), 1); HG_DBG(ctx);
}


// Tile order of the tail kernel (traverse.hip): the buffers of the context grown to `tiles` entries (cost, order; cost cleared)
bool hagrid_trav::tile_order_buffers(hagrid_ctx* ctx, hagrid_ctx::RayHints& h, int tiles) {
    if (h.lpt_cap >= tiles && h.lpt_buf) return true;
    // two sizes only -- up to 16 384 tiles (1024^2 and smaller) or the largest launch the order is used for -- so a slot grows at most once
    // (growing waits for the stream: the old buffers may be in use)
    if (h.lpt_buf) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(h.lpt_buf); h.lpt_buf = nullptr; h.lpt_cap = 0; }
    const int cap = tiles <= (1 << 14) ? (1 << 14) : kMaxOrderTiles;
    if (hipMalloc((void**)&h.lpt_buf, size_t(cap) * 2 * sizeof(int) + 8 * sizeof(float4)) != hipSuccess) { (void)hipGetLastError(); h.lpt_buf = nullptr; return false; }
    (void)hipMemsetAsync(h.lpt_buf, 0, size_t(cap) * 2 * sizeof(int) + 8 * sizeof(float4), ctx->stream);
    h.lpt_cap = cap; h.lpt_valid = false;
    return true;
}
void hagrid_trav::launch_tile_order(hagrid_ctx* ctx, hagrid_ctx::RayHints& h, int tiles, const TraverseArgs& a, int rot, int* suggest) {
    int* cost = h.lpt_buf, *order = cost + h.lpt_cap;
    rot = std::max(0, std::min(rot, tiles));
    tile_order_kernel<<<1, kOrderBlock, 0, ctx->stream>>>(cost, order, tiles, rot, ctx->opt_quad_head, suggest, a.rays, a.num_rays, tile_order_samples(h)); HG_DBG(ctx);
    h.lpt_rot = rot;
    h.lpt_epoch++;
}

namespace {
template <int BITS>
int bin_rays_bits(hagrid_ctx* ctx, TraverseArgs& a, int num_rays, PoolTemps& tmp) {
    constexpr int kBins = 1 << (3 * BITS);
    const int tiles = grid_blocks(num_rays, kBinTile);
    const int table_n = kBins * tiles;
    int* perm = tmp.get<int>(size_t(num_rays));
    unsigned short* bin_keys = tmp.get<unsigned short>(size_t(num_rays));
    int* bin_table = tmp.get<int>(size_t(table_n));
    int* bin_partials = tmp.get<int>(size_t(scan_num_tiles(table_n)) + 1);
    if (!perm || !bin_keys || !bin_table || !bin_partials) return HAGRID_ENOMEM;
    if (ctx->ray_binning == 2) {
        // automatic: everything is decided on the device, nobody waits.  row length (image-ordered batches are left to the
        // tile packets) -> keys + coherence estimate -> scan -> decision -> scatter; the traversal kernel reads the decision.
        int* row_len = ctx->dscratch + kScrRowLenBinned;
        int* flag = ctx->dscratch + kScrRowLenBinned + 1;
        if (!ctx->bin_diff) {
            HG_HIP(ctx, hipMalloc((void**)&ctx->bin_diff, 64 * sizeof(int)));
            HG_HIP(ctx, hipMemsetAsync(ctx->bin_diff, 0, 64 * sizeof(int), ctx->stream));
        }
        launch_detect(ctx, a, ctx->opt_image_width >= 0 ? num_rays : 0, row_len);
        ray_bin_count<BITS><<<tiles, kBlock, 0, ctx->stream>>>(a, bin_keys, bin_table, row_len, ctx->bin_diff); HG_DBG(ctx);
        if (!ctx_scan<int>(ctx, TableIn{bin_table}, TableOut{bin_table}, table_n, bin_partials, (const int*)nullptr, (int*)nullptr)) return HAGRID_ENOMEM;
        ray_bin_decide<<<1, 64, 0, ctx->stream>>>(row_len, ctx->bin_diff, num_rays, flag); HG_DBG(ctx);
        ray_bin_scatter<BITS><<<tiles, kBlock, 0, ctx->stream>>>(bin_keys, bin_table, num_rays, perm, flag); HG_DBG(ctx);
        a.perm_flag = flag;
        if (ctx->opt_image_width == 0) a.row_len = row_len;
        else if (ctx->opt_image_width > 0) a.row_len_hint = ctx->opt_image_width;
    } else {
        ray_bin_count<BITS><<<tiles, kBlock, 0, ctx->stream>>>(a, bin_keys, bin_table, nullptr, nullptr); HG_DBG(ctx);
        if (!ctx_scan<int>(ctx, TableIn{bin_table}, TableOut{bin_table}, table_n, bin_partials, (const int*)nullptr, (int*)nullptr)) return HAGRID_ENOMEM;
        ray_bin_scatter<BITS><<<tiles, kBlock, 0, ctx->stream>>>(bin_keys, bin_table, num_rays, perm, nullptr); HG_DBG(ctx);
    }
    a.perm = perm;
    return HAGRID_OK;
}
} // namespace

int hagrid_trav::bin_rays(hagrid_ctx* ctx, TraverseArgs& a, int num_rays, PoolTemps& tmp) {
    a.perm = nullptr;
    if (!ctx->ray_binning || num_rays <= kBinTile) return HAGRID_OK;
    // 512 bins: the wavefronts an XCD has resident at a time come from one or two bins, and what they gather -- the bin's share of the traversal image and of the
    // triangles, and a margin around it -- mostly stays in that XCD's 4 MB of L2.
    return bin_rays_bits<3>(ctx, a, num_rays, tmp);
}
