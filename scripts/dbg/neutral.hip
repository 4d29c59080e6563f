// Neutral victim / neutral aggressor for the cross-stream investigation (VERDICT r3 "what's weak" 3).
// Nothing here shares code with libalsep: if these two reproduce the effect between themselves it belongs to the stack, if only
// libalsep's kernels do it is a bug of theirs.
//   nv_kernel   (victim)   : 256 threads; every thread reads 16 entries of a constant global table into registers (the access pattern of
//                            FftRegs::load: tab[(i % P) * stride]), checks them against the closed form, then round-trips them `iters`
//                            times through `lds_bytes` of LDS with barriers (write at one index, read at a permuted one) and checks
//                            again.  Counters: bad[0] = wrong table values in registers, bad[1] = wrong values after LDS, bad[2] = workgroups
//                            with any mismatch.
//   na_kernel   (aggressor): f16 MFMA GEMM-like loop, 256 threads, 64 KiB of dynamic LDS, __launch_bounds__(256, 2) and ~256 VGPRs
//                            (64 accumulator tiles' worth of registers): two workgroups per CU, like nn_gemm_hh_kernel / nn_conv_hh_kernel;
//                            streams A/B tiles from global memory through registers into LDS and reads them back as MFMA fragments.
#include <hip/hip_runtime.h>
#include <cstdint>

extern __shared__ __attribute__((aligned(16))) char smem[];

__host__ __device__ inline unsigned nv_hash(unsigned i) {
    i ^= i >> 16; i *= 0x7feb352du; i ^= i >> 15; i *= 0x846ca68bu; i ^= i >> 16;
    return i;
}

__global__ void nv_fill_kernel(uint2* tab, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) tab[i] = make_uint2(nv_hash(i), nv_hash(i + 0x9e3779b9u));
}

__global__ void __launch_bounds__(256) nv_kernel(const uint2* __restrict__ tab, int ntab, int lds_bytes, int iters,
                                                 unsigned long long* __restrict__ bad) {
    uint2* buf = reinterpret_cast<uint2*>(smem);
    const int n = lds_bytes / 8;                              // uint2 slots
    const int tid = threadIdx.x;
    uint2 w[16];
    int idx[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int i = tid + b * 256;
        idx[b] = (i % 512) * (ntab / 512);                    // strided like tw[(i % P) * TWS]
        w[b] = tab[idx[b]];
    }
    unsigned long long c0 = 0, c1 = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) c0 += (w[b].x != nv_hash(idx[b])) | (w[b].y != nv_hash(idx[b] + 0x9e3779b9u));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const int s = (tid + b * 256 + it * 977) % n;
            buf[s] = make_uint2(w[b].x ^ (unsigned)s, w[b].y + (unsigned)it);
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            // read the slot that thread (tid ^ 85), butterfly 15 - b wrote
            const int ot = tid ^ 85, ob = 15 - b;
            const int s = (ot + ob * 256 + it * 977) % n;
            const int oi = ((ot + ob * 256) % 512) * (ntab / 512);
            const uint2 r = buf[s];
            // a slot may have been overwritten by a later (tid, b) pair when n < 4096: only check slots written once
            const bool unique = n >= 4096;
            if (unique) c1 += (r.x != (nv_hash(oi) ^ (unsigned)s)) | (r.y != nv_hash(oi + 0x9e3779b9u) + (unsigned)it);
        }
        __syncthreads();
    }
    if (c0) atomicAdd(bad + 0, c0);
    if (c1) atomicAdd(bad + 1, c1);
    const int any = __syncthreads_or((int)(c0 + c1 != 0));
    if (any && tid == 0) atomicAdd(bad + 2, 1ull);
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// C[128 x 128 per workgroup] += A[128 x K] B[128 x K]^T, K slices of 64 through 2 x (128 x 64) halves of LDS = 32 KiB, declared 64 KiB.
// V256: the kernel touches v255, so its allocation is the full 256 VGPRs per lane that nn_gemm_hh_kernel / nn_conv_hh_kernel have
template <bool V256>
__global__ void __launch_bounds__(256, 2) na_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B, float* __restrict__ C,
                                                    int K, int tiles_n) {
    if (V256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    _Float16* As = reinterpret_cast<_Float16*>(smem);
    _Float16* Bs = As + 128 * 72;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const _Float16* a = A + (size_t)tm * 128 * K;
    const _Float16* b = B + (size_t)tn * 128 * K;
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    for (int k0 = 0; k0 < K; k0 += 64) {
        __syncthreads();
        // 128 rows x 8 groups of 8 halves per operand = 1024 16-byte pieces: 4 per thread per operand
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = tid + j * 256, r = p >> 3, g = p & 7;
            *reinterpret_cast<h8*>(As + r * 72 + g * 8) = *reinterpret_cast<const h8*>(a + (size_t)r * K + k0 + g * 8);
            *reinterpret_cast<h8*>(Bs + r * 72 + g * 8) = *reinterpret_cast<const h8*>(b + (size_t)r * K + k0 + g * 8);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const h8*>(As + (wm + i * 16 + l15) * 72 + ks * 32 + lq * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const h8*>(Bs + (wn + j * 16 + l15) * 72 + ks * 32 + lq * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    float* c = C + ((size_t)tm * 128 + wm) * (tiles_n * 128) + tn * 128 + wn;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) c[(size_t)(i * 16 + 4 * lq + r) * (tiles_n * 128) + j * 16 + l15] = acc[i][j][r];
}

extern "C" int nv_fill(void* stream, void* tab, int n) {
    hipLaunchKernelGGL(nv_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (uint2*)tab, n);
    return (int)hipGetLastError();
}
extern "C" int nv_launch(void* stream, int blocks, const void* tab, int ntab, int lds_bytes, int iters, unsigned long long* bad) {
    hipFuncSetAttribute((const void*)nv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(nv_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (const uint2*)tab, ntab, lds_bytes, iters, bad);
    return (int)hipGetLastError();
}
// M, N multiples of 128, K a multiple of 64; lds_bytes >= 36864 (declare 65536 to occupy what the library's kernels occupy)
extern "C" int na_launch(void* stream, const void* A, const void* B, void* C, int M, int N, int K, int lds_bytes, int v256) {
    if (v256) {
        hipFuncSetAttribute((const void*)na_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(na_kernel<true>, dim3((M / 128) * (N / 128)), dim3(256), lds_bytes, (hipStream_t)stream, (const _Float16*)A,
                           (const _Float16*)B, (float*)C, K, N / 128);
    } else {
        hipFuncSetAttribute((const void*)na_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(na_kernel<false>, dim3((M / 128) * (N / 128)), dim3(256), lds_bytes, (hipStream_t)stream, (const _Float16*)A,
                           (const _Float16*)B, (float*)C, K, N / 128);
    }
    return (int)hipGetLastError();
}
