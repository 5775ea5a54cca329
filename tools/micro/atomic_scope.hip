// micro-benchmark: throughput of non-returning / returning atomic adds on random words of a 32 MB array, by scope
// hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_scope.hip -o /tmp/atomic_scope && /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int SCOPE, bool RET>
__global__ void k(int* a, uint32_t mask, int per_thread, int* sink) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    int acc = 0;
    for (int i = 0; i < per_thread; i++) {
        int* p = a + (hash(t * 131u + i) & mask);
        if (RET) acc += __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, SCOPE);
        else (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, SCOPE);
    }
    if (RET && acc == 0x7fffffff) *sink = acc;
}
template <int SCOPE, bool RET> void run(const char* name, int* a, uint32_t mask, int* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 16, threads = 256, per = 4;
    k<SCOPE, RET><<<blocks, threads>>>(a, mask, per, sink);
    hipEventRecord(e0);
    k<SCOPE, RET><<<blocks, threads>>>(a, mask, per, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s words %9u: %8.1f us  %7.1f atomics/ns\n", name, mask + 1, ms * 1e3, double(blocks) * threads * per / (ms * 1e6));
}
int main() {
    int* a; int* sink; hipMalloc(&a, size_t(1) << 25); hipMalloc(&sink, 4); hipMemset(a, 0, size_t(1) << 25);
    for (uint32_t mask : {(1u << 23) - 1, (1u << 17) - 1, (1u << 10) - 1}) {
        run<__HIP_MEMORY_SCOPE_AGENT, false>("agent scope, no return", a, mask, sink);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("workgroup scope, no return", a, mask, sink);
        run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, returning", a, mask, sink);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, returning", a, mask, sink);
    }
    return 0;
}
