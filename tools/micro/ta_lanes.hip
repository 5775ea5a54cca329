// DEV MICROBENCHMARK: what does a vector load cost in the texture-address / L1 path of gfx950 when only some lanes of the
// wavefront are active?  (The traversal kernel issues its record / triangle loads with ~14 of 64 lanes live.)
// Every wavefront issues `iters` independent loads per lane from a table that fits the vector L1 (or the L2), with the first
// `active` lanes enabled; reported: CU-cycles per wavefront INSTRUCTION (2.4 GHz, 256 CUs, 32 wavefronts per CU resident).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ta_lanes.hip -o /tmp/ta_lanes && /tmp/ta_lanes [working set KiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// WIDTH: bytes per lane (4 or 16); SPREAD: 1 = every lane its own random line, 0 = all active lanes read consecutive 16-byte pieces
template <int WIDTH, int SPREAD>
__global__ void __launch_bounds__(64) loads(const uint4* __restrict__ data, uint32_t mask16, int iters, int active, uint32_t* out) {
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    uint32_t acc = 0;
    if (int(lane) < active) {
        for (int it = 0; it < iters; it++) {
            const uint32_t seed = (wave * 1315423911u) ^ (uint32_t(it) * 2654435761u);
            const uint32_t r = SPREAD ? (mix(seed + lane) & mask16) : ((mix(seed) + lane) & mask16);
            if (WIDTH == 16) { const uint4 v = data[r]; acc ^= v.x ^ v.w; }
            else acc ^= reinterpret_cast<const uint32_t*>(data)[r * 4];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int WIDTH, int SPREAD>
void run(const char* name, const uint4* d, uint32_t mask16, uint32_t* out) {
    const int waves = 256 * 32 * 4, iters = 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-46s", name);
    for (int active : {64, 32, 16, 8, 4, 1}) {
        loads<WIDTH, SPREAD><<<waves, 64>>>(d, mask16, iters, active, out);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0));
            loads<WIDTH, SPREAD><<<waves, 64>>>(d, mask16, iters, active, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("  %2d lanes: %6.1f", active, best * 1e-3 * 2.4e9 * 256 / (double(waves) * iters));
    }
    printf("   CU-cycles per wavefront instruction\n");
}

int main(int argc, char** argv) {
    size_t ws_kb = argc > 1 ? atoi(argv[1]) : 16;
    const size_t n16 = ws_kb * 1024 / 16;
    uint4* d; uint32_t* out;
    CK(hipMalloc(&d, n16 * 16 + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 1, n16 * 16 + 4096));
    const uint32_t mask16 = uint32_t(n16 - 1);
    printf("working set %zu KiB\n", ws_kb);
    run<16, 1>("16 B per lane, every lane its own line", d, mask16, out);
    run<16, 0>("16 B per lane, consecutive pieces", d, mask16, out);
    run<4, 1>("4 B per lane, every lane its own line", d, mask16, out);
    run<4, 0>("4 B per lane, consecutive 16-byte pieces", d, mask16, out);
    return 0;
}
