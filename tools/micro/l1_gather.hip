// DEV MICROBENCHMARK: what does a divergent gather cost in the vector L1 (TA/TCP) of gfx950, and does it help when
// neighbouring lanes fetch the pieces of one record cooperatively?  Working set sized to stay in L2.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l1_gather.hip -o /tmp/l1_gather && /tmp/l1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: one 16-B load per lane, random line            (records = lanes)
// MODE 1: 48-B record per lane, three 16-B loads         (triangle today)
// MODE 2: 32-B record per lane, two 16-B loads           (cell today)
// MODE 3: 48-B record per QUAD, lanes 0..2 load one piece each; 4 instructions fetch the 64 records of a wave
// MODE 4: 32-B record per PAIR of lanes; 2 instructions fetch the 64 records of a wave
// MODE 5: one 4-B load per lane, random line             (voxel-map entry, reference id)
// MODE 6: 64-B record per QUAD, 4 lanes one piece each; 4 instructions per 64 records
// every mode fetches 64 records per wave per iteration; bytes differ
template <int MODE>
__global__ void __launch_bounds__(64) gather(const uint4* __restrict__ data, uint32_t mask16, int iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint32_t seed = (wave * 1315423911u) ^ (uint32_t(it) * 2654435761u);
        if (MODE == 0) {
            const uint4 v = data[mix(seed + lane) & mask16];
            acc ^= v.x ^ v.w;
        } else if (MODE == 1) {
            const uint32_t r = (mix(seed + lane) & mask16) / 3 * 3;
            const uint4 a = data[r], b = data[r + 1], c = data[r + 2];
            acc ^= a.x ^ b.y ^ c.z;
        } else if (MODE == 2) {
            const uint32_t r = (mix(seed + lane) & mask16) & ~1u;
            const uint4 a = data[r], b = data[r + 1];
            acc ^= a.x ^ b.y;
        } else if (MODE == 3) {
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t rec = j * 16 + (lane >> 2);                  // record handled by this quad
                const uint32_t r = (mix(seed + rec) & mask16) / 3 * 3;
                const uint32_t piece = lane & 3;
                const uint4 v = data[r + (piece < 3 ? piece : 0)];
                acc ^= v.x ^ v.y;
            }
        } else if (MODE == 4) {
            #pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t rec = j * 32 + (lane >> 1);
                const uint32_t r = (mix(seed + rec) & mask16) & ~1u;
                const uint4 v = data[r + (lane & 1)];
                acc ^= v.x ^ v.y;
            }
        } else if (MODE == 5) {
            const uint32_t* d = reinterpret_cast<const uint32_t*>(data);
            acc ^= d[mix(seed + lane) & (mask16 * 4 + 3)];
        } else if (MODE == 6) {
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t rec = j * 16 + (lane >> 2);
                const uint32_t r = (mix(seed + rec) & mask16) & ~3u;
                const uint4 v = data[r + (lane & 3)];
                acc ^= v.x ^ v.y;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
void run(const char* name, const uint4* d, uint32_t mask16, uint32_t* out, int waves, int iters, double bytes_per_rec) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    gather<MODE><<<waves, 64>>>(d, mask16, iters, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        gather<MODE><<<waves, 64>>>(d, mask16, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double recs = double(waves) * 64 * iters;
    // cycles per record per CU at 2.4 GHz, 256 CUs
    const double cyc = best * 1e-3 * 2.4e9 * 256 / recs;
    printf("%-44s %8.3f ms  %7.2f Grec/s  %6.2f CU-cycles/record  %7.1f GB/s useful\n", name, best, recs / best / 1e6, cyc, recs * bytes_per_rec / best / 1e6);
}

int main(int argc, char** argv) {
    size_t ws_kb = argc > 1 ? atoi(argv[1]) : 16384;               // working set in KiB (power of two, >= 4)
    if (ws_kb < 4) ws_kb = 4;
    const int waves = 256 * 32 * 4, iters = 64;
    const size_t n16 = ws_kb * 1024 / 16;
    uint4* d; uint32_t* out;
    CK(hipMalloc(&d, n16 * 16 + 256)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 1, n16 * 16 + 256));
    const uint32_t mask16 = uint32_t(n16 - 1);
    printf("working set %zu KiB, %d waves x %d iterations x 64 records\n", ws_kb, waves, iters);
    run<5>("4 B per lane, random", d, mask16, out, waves, iters, 4);
    run<0>("16 B per lane, random", d, mask16, out, waves, iters, 16);
    run<2>("32 B record per lane (2 loads)", d, mask16, out, waves, iters, 32);
    run<4>("32 B record per lane PAIR (cooperative)", d, mask16, out, waves, iters, 32);
    run<1>("48 B record per lane (3 loads)", d, mask16, out, waves, iters, 48);
    run<3>("48 B record per QUAD (cooperative)", d, mask16, out, waves, iters, 48);
    run<6>("64 B record per QUAD (cooperative)", d, mask16, out, waves, iters, 64);
    return 0;
}
