// Write-stream roofline of the box: how fast can 16-byte stores alone fill HBM, in the shapes the fused STFT epilogue uses?
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/dbg/fill.hip -o scripts/dbg/libfill.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (a) grid-stride linear fill, 256 threads
__global__ void __launch_bounds__(256) fill_linear(f32x4* p, int64_t n16, float v) {
    const f32x4 x = {v, v, v, v};
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = x;
}
// (b) one workgroup of THR threads per contiguous region of `region` bytes (the fused STFT: 128 threads, 294 912 bytes, 126 lanes active)
template <int THR, int ACTIVE>
__global__ void __launch_bounds__(THR) fill_region(char* p, int64_t region, float v) {
    const f32x4 x = {v, v, v, v};
    char* dst = p + (int64_t)blockIdx.x * region + 16 * threadIdx.x;
    if ((int)threadIdx.x < ACTIVE)
        for (int64_t o = 0; o + 16 * threadIdx.x < region; o += 16 * ACTIVE) *reinterpret_cast<f32x4*>(dst + o) = x;
}
// (c) read-only and copy, for the same box
__global__ void __launch_bounds__(256) read_linear(const f32x4* p, int64_t n16, float* sink) {
    f32x4 a = {0, 0, 0, 0};
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) *sink = 1.f;
}
__global__ void __launch_bounds__(256) copy_linear(const f32x4* s, f32x4* d, int64_t n16) {
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}

// (d) the level-0 convolution's epilogue shape (conv3x3_bf16_m0_kernel): 8 waves, tile = 8 rows x 48 pixels x 96 bytes; per 16 pixels a wave
// issues one 16-byte store (channels 0..31: four lanes = 64 bytes per pixel, pixel stride 96) and one 8-byte store (channels 32..47).
// LINEAR = the same bytes as whole-row linear 16-byte stores (what an LDS-transposed epilogue would issue).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool LINEAR>
__global__ void __launch_bounds__(512) fill_m0(char* p, int Fw, int tiles_f, int tiles_t, float v) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const f32x4 x = {v, v, v, v};
    const f32x2 x2 = {v, v};
    for (int tile = blockIdx.x; tile < tiles_f * tiles_t; tile += gridDim.x) {
        const int tf = tile % tiles_f, tt = tile / tiles_f;
        char* row = p + ((int64_t)(tt * 8 + wave) * Fw + tf * 48) * 96;
        if (LINEAR) {
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i < 4 || lane < 32) *reinterpret_cast<f32x4*>(row + 1024 * i + 16 * lane) = x;
        } else {
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) {
                char* px = row + (ni * 16 + l15) * 96;
                *reinterpret_cast<f32x4*>(px + 16 * lq) = x;
                *reinterpret_cast<f32x2*>(px + 64 + 8 * lq) = x2;
            }
        }
    }
}

extern "C" int fill_bench(void* buf, void* buf2, int64_t bytes, int mode, int grid, int iters, float* us_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int64_t n16 = bytes / 16;
    const int64_t region = 294912;
    auto launch = [&]() {
        switch (mode) {
            case 0: hipLaunchKernelGGL(fill_linear, dim3(grid), dim3(256), 0, 0, (f32x4*)buf, n16, 1.f); break;
            case 1: hipLaunchKernelGGL((fill_region<128, 126>), dim3((unsigned)(bytes / region)), dim3(128), 0, 0, (char*)buf, region, 1.f); break;
            case 2: hipLaunchKernelGGL((fill_region<128, 128>), dim3((unsigned)(bytes / region)), dim3(128), 0, 0, (char*)buf, region, 1.f); break;
            case 3: hipLaunchKernelGGL((fill_region<256, 256>), dim3((unsigned)(bytes / region)), dim3(256), 0, 0, (char*)buf, region, 1.f); break;
            case 4: hipLaunchKernelGGL(read_linear, dim3(grid), dim3(256), 0, 0, (const f32x4*)buf, n16, (float*)buf2); break;
            case 6: hipLaunchKernelGGL((fill_m0<false>), dim3(grid), dim3(512), 0, 0, (char*)buf, 3072, 64, (int)(bytes / (3072 * 96 * 8)), 1.f); break;
            case 7: hipLaunchKernelGGL((fill_m0<true>), dim3(grid), dim3(512), 0, 0, (char*)buf, 3072, 64, (int)(bytes / (3072 * 96 * 8)), 1.f); break;
            case 5: hipLaunchKernelGGL(copy_linear, dim3(grid), dim3(256), 0, 0, (const f32x4*)buf, (f32x4*)buf2, n16); break;
        }
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *us_out = ms * 1000.f / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (int)hipGetLastError();
}
