// DEV MICROBENCHMARK: does a global_load_lds into an LDS sink work as a cache prefetch on gfx950?
// ONE wavefront walks a chain of random 48-byte records (each lane its own chain; the next index is computed, not loaded, so the
// address of the record after next is known early).  Per step: three 16-B loads of the record, a dependent use.
//   mode 0: plain            mode 1: touch the NEXT record with global_load_lds (4 B) before using the current one
//   mode 2: touch it with a normal 4-B load whose value is consumed a step later
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_prefetch.hip -o /tmp/lds_prefetch && /tmp/lds_prefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(64) chain(const uint4* __restrict__ data, uint32_t mask, int steps, uint32_t* out, int lanes) {
    __shared__ int sink[64];
    const uint32_t lane = threadIdx.x;
    if (int(lane) >= lanes) return;
    uint32_t acc = 0, cur = mix(lane * 7919u + blockIdx.x) & mask, late = 0;
    for (int s = 0; s < steps; s++) {
        const uint32_t nxt = mix(cur + uint32_t(s) * 2654435761u) & mask;          // known before the loads of `cur` return
        const uint4* p = data + size_t(cur) * 3;
        if (MODE == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(data + size_t(nxt) * 3),
                                                        (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
        uint32_t touched = 0;
        if (MODE == 2) touched = reinterpret_cast<const uint32_t*>(data + size_t(nxt) * 3)[0];
        const uint4 a = p[0], b = p[1], c = p[2];
        acc += a.x ^ b.y ^ c.z;
        acc = mix(acc);                                                             // a dependent use: the chain cannot run ahead
        acc ^= late; late = touched;
        cur = nxt ^ (acc == 0x12345678u ? 1u : 0u);                               // data dependent in form, the guess `nxt` in fact
    }
    out[blockIdx.x * 64 + lane] = acc + late;
}

int main(int argc, char** argv) {
    const size_t records = size_t(1) << 22;                  // 4M records x 48 B = 192 MB: misses the L2
    uint4* d; uint32_t* o;
    CK(hipMalloc(&d, records * 48)); CK(hipMalloc(&o, 64 * 4 * 64));
    CK(hipMemset(d, 1, records * 48));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 2000;
    for (int lanes : {1, 8, 64}) for (int waves : {1, 64}) {
        float ms[3];
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0));
                if (mode == 0) chain<0><<<waves, 64>>>(d, uint32_t(records - 1), steps, o, lanes);
                if (mode == 1) chain<1><<<waves, 64>>>(d, uint32_t(records - 1), steps, o, lanes);
                if (mode == 2) chain<2><<<waves, 64>>>(d, uint32_t(records - 1), steps, o, lanes);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); best = t < best ? t : best;
            }
            ms[mode] = best;
        }
        printf("{\"lanes\": %d, \"wavefronts\": %d, \"ns per step: plain\": %.0f, \"lds-sink touch of the next record\": %.0f, \"register touch\": %.0f}\n",
               lanes, waves, ms[0] * 1e6 / steps, ms[1] * 1e6 / steps, ms[2] * 1e6 / steps);
    }
    return 0;
}
