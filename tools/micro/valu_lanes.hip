// DEV MICROBENCHMARK: does a VALU instruction of a wavefront with few live lanes cost the SIMD less?  (gfx950 executes a wave64
// instruction as four passes of 16 lanes; if passes whose 16 lanes are all disabled were skipped, packing the live rays of the
// traversal kernel into few quarters would pay.)  Every wavefront runs `iters` x 32 independent v_fma_f32 with the lanes of `mask`
// enabled; 8 wavefronts per SIMD resident; reported: SIMD-cycles per wavefront instruction at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_lanes.hip -o /tmp/valu_lanes && /tmp/valu_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mixed launch: even workgroups run `mask`, odd ones `mask_odd`; per-class wall-clock time of a wavefront (100 MHz counter) summed into times[0 / 1]
__global__ void __launch_bounds__(64) fmas_mixed(unsigned long long mask, unsigned long long mask_odd, int iters, float* out, unsigned long long* times) {
    float a0 = threadIdx.x, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, a4 = 4.0f, a5 = 5.0f, a6 = 6.0f, a7 = 7.0f;
    const float m = 1.0000001f, c = 1e-9f;
    const unsigned long long mk = (blockIdx.x & 1) ? mask_odd : mask;
    const unsigned long long t0 = wall_clock64();
    if ((mk >> threadIdx.x) & 1ull) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) atomicAdd(times + (blockIdx.x & 1), t1 - t0);
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 0.12345f) out[0] = s;
}

__global__ void __launch_bounds__(64) fmas(unsigned long long mask, int iters, float* out) {
    float a0 = threadIdx.x, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, a4 = 4.0f, a5 = 5.0f, a6 = 6.0f, a7 = 7.0f;
    const float m = 1.0000001f, c = 1e-9f;
    if ((mask >> threadIdx.x) & 1ull) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        }
    }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 0.12345f) out[0] = s;
}

int main() {
    float* out; CK(hipMalloc(&out, 64));
    const int waves = 256 * 32 * 2, iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* name; unsigned long long mask; } cases[] = {
        {"64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15 (one quarter)", 0xffffull}, {"lane 0", 1ull},
        {"lanes 0,16,32,48 (one per quarter)", 0x0001000100010001ull}, {"lanes 0-3", 0xfull}, {"lanes 32-47", 0xffffull << 32},
        {"lanes 0-7", 0xffull}, {"lanes 0-11", 0xfffull}, {"lanes 0-5", 0x3full}, {"even lanes 0-30 (16 lanes)", 0x55555555ull}, {"lanes 0-14", 0x7fffull}};
    for (auto& cs : cases) {
        fmas<<<waves, 64>>>(cs.mask, iters, out); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0)); fmas<<<waves, 64>>>(cs.mask, iters, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        // instructions per SIMD = waves * iters * 32 / (256 CUs * 4 SIMDs)
        printf("%-40s %7.3f ms   %5.2f SIMD-cycles per wavefront v_fma\n", cs.name, best, best * 1e-3 * 2.4e9 / (double(waves) * iters * 32 / 1024.0));
    }
    // the same comparison inside ONE launch (same clocks, same neighbours): even workgroups full, odd workgroups sparse
    unsigned long long* times; CK(hipMalloc(&times, 16));
    for (unsigned long long sparse : {1ull, 0xfull, 0xffffull}) {
        CK(hipMemset(times, 0, 16));
        fmas_mixed<<<waves, 64>>>(~0ull, sparse, iters, out, times); CK(hipDeviceSynchronize());
        unsigned long long h[2]; CK(hipMemcpy(h, times, 16, hipMemcpyDeviceToHost));
        printf("mixed launch, odd workgroups with mask %#llx: full wavefront %.1f us, sparse wavefront %.1f us on average\n", sparse,
               h[0] / (waves / 2.0) / 100.0, h[1] / (waves / 2.0) / 100.0);
    }
    return 0;
}
