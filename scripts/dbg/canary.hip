#include <hip/hip_runtime.h>
#include <cstdint>
extern __shared__ __attribute__((aligned(16))) char smem[];
// fills lds_bytes of LDS with a pattern, spins `iters` rounds re-checking it, counts mismatches
__global__ void __launch_bounds__(256) canary_kernel(int lds_bytes, int iters, unsigned long long* bad) {
    unsigned* w = reinterpret_cast<unsigned*>(smem);
    const int n = lds_bytes / 4;
    const unsigned tag = 0xA5000000u ^ (blockIdx.x * 2654435761u);
    for (int i = threadIdx.x; i < n; i += 256) w[i] = tag + i;
    __syncthreads();
    unsigned long long cnt = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < n; i += 256) cnt += (w[i] != tag + i);
        __syncthreads();
    }
    if (cnt) atomicAdd(bad, cnt);
}
extern "C" int canary_launch(void* stream, int blocks, int lds_bytes, int iters, unsigned long long* bad) {
    hipFuncSetAttribute((const void*)canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, lds_bytes, iters, bad);
    return (int)hipGetLastError();
}
