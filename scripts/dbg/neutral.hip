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
template <bool V256, int MFMA = 1>          // MFMA: 1 f16, 0 none (VALU fma), 3 f32 (v_mfma_f32_16x16x4_f32), 4 bf16
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
                for (int j = 0; j < 4; ++j) {
                    if (MFMA == 1) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    } else if (MFMA == 3) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)af[i][e], (float)bf[j][e], acc[i][j], 0, 0, 0);
                    } else if (MFMA == 4) {
                        typedef __bf16 b8 __attribute__((ext_vector_type(8)));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, af[i]), __builtin_bit_cast(b8, bf[j]), acc[i][j], 0, 0, 0);
                    } else {                                  // the same operands through the VALU: no matrix instruction in the kernel
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][e] = fmaf((float)af[i][e], (float)bf[j][e + 4], acc[i][j][e]);
                    }
                }
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


// ---- second neutral victim: the instruction classes the FFT kernels are made of, each checked against exact small-integer results ----
// bad[0]: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel / neg modifiers (packed complex arithmetic), bad[1]: scalar v_fma_f32 chains,
// bad[2]: LDS round trips through ds_write2st64_b64 / ds_read2st64_b64 + ds_write_b128 / ds_read_b128, bad[3]: workgroups with any mismatch.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) nv2_kernel(int iters, unsigned long long* __restrict__ bad) {
    v2f* buf = reinterpret_cast<v2f*>(smem);                  // 8192 float2 slots = 64 KiB
    const int tid = threadIdx.x;
    unsigned long long c0 = 0, c1 = 0, c2 = 0;
    for (int it = 0; it < iters; ++it) {
        // complex multiply-accumulate on exact integers: (a.x + i a.y)(b.x + i b.y) + c, all components < 2^10
        const int ax = (tid & 15) + (it & 7), ay = 3 + (tid >> 6), bx = 5 - (it & 3), by = (tid >> 4) + 1, cx = it & 31, cy = 7;
        const v2f a = {(float)ax, (float)ay}, b = {(float)bx, (float)by}, c = {(float)cx, (float)cy};
        v2f t, r, q;
        // t = (-a.y b.y, a.y b.x) + c ; r = (a.x b.x, a.x b.y) + t     (the cx_mul of the FFT kernels: op_sel broadcasts, neg on one half)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(t) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(q) : "v"(r), "v"(c));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(q) : "v"(q), "v"(b));
        const int ex = ax * bx - ay * by + cx, ey = ax * by + ay * bx + cy;
        c0 += (r.x != (float)ex) | (r.y != (float)ey) | (q.x != (float)((ex + cx) * bx)) | (q.y != (float)((ey + cy) * by));
        // scalar chain
        float s = fmaf((float)ax, (float)bx, (float)cx);
        s = fmaf(-(float)ay, (float)by, s);
        c1 += (s != (float)ex);
        // LDS: the FFT passes' access shapes (stride-64 pairs, 16-byte vectors), values carry (tid, b, it)
#pragma unroll
        for (int b8 = 0; b8 < 4; ++b8) {
            const int i = tid + b8 * 256;                     // slots i, i + 1024 ... as buf[i + r * M]
            buf[i] = v2f{(float)(i + it), (float)(tid)};
            buf[i + 1024 * 4] = v2f{(float)(i - it), (float)(b8)};
        }
        __syncthreads();
#pragma unroll
        for (int b8 = 0; b8 < 4; ++b8) {
            const int ot = tid ^ 37, i = ot + (3 - b8) * 256;
            const v2f u = buf[i], w = buf[i + 1024 * 4];
            c2 += (u.x != (float)(i + it)) | (u.y != (float)ot) | (w.x != (float)(i - it)) | (w.y != (float)(3 - b8));
        }
        __syncthreads();
        v4f* b4 = reinterpret_cast<v4f*>(smem);
        b4[tid + 2048] = v4f{(float)tid, (float)it, (float)(tid + it), 1.f};
        __syncthreads();
        const v4f z = b4[(tid ^ 129) + 2048];
        c2 += (z.x != (float)(tid ^ 129)) | (z.y != (float)it) | (z.z != (float)((tid ^ 129) + it)) | (z.w != 1.f);
        __syncthreads();
    }
    if (c0) atomicAdd(bad + 0, c0);
    if (c1) atomicAdd(bad + 1, c1);
    if (c2) atomicAdd(bad + 2, c2);
    const int any = __syncthreads_or((int)(c0 + c1 + c2 != 0));
    if (any && tid == 0) atomicAdd(bad + 3, 1ull);
}
extern "C" int nv2_launch(void* stream, int blocks, int iters, unsigned long long* bad) {
    hipFuncSetAttribute((const void*)nv2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(nv2_kernel, dim3(blocks), dim3(256), 65536, (hipStream_t)stream, iters, bad);
    return (int)hipGetLastError();
}


// ---- third neutral victim: a dense radix-2 butterfly network on float2 values written as plain C++ (the compiler's SLP vectoriser turns it
// into v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, exactly as it does in stft_kernel), on small integers so that an int32 model of the same
// network gives the exact expected bits.  bad[0] = wrong components, bad[1] = workgroups with any.
struct c2 { float x, y; };
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return c2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return c2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c2 cmulw(c2 a, c2 w) { return c2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }   // w in {1, -i, -1, i, 2, 1+i}: exact
struct i2 { int x, y; };
__device__ __forceinline__ i2 iadd(i2 a, i2 b) { return i2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ i2 isub(i2 a, i2 b) { return i2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ i2 imulw(i2 a, i2 w) { return i2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__global__ void __launch_bounds__(256) nv3_kernel(int iters, unsigned long long* __restrict__ bad) {
    const int tid = threadIdx.x;
    const i2 wt[8] = {{1, 0}, {0, -1}, {-1, 0}, {0, 1}, {2, 0}, {1, 1}, {1, -1}, {0, 2}};
    unsigned long long cnt = 0;
    for (int it = 0; it < iters; ++it) {
        c2 u[16];
        i2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            v[e] = i2{((tid * 7 + e * 13 + it * 5) & 63) - 32, ((tid * 3 + e * 11 + it) & 63) - 32};
            u[e] = c2{(float)v[e].x, (float)v[e].y};
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {                      // four butterfly stages over the 16 values, a twiddle per butterfly
            const int h = 1 << st;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (e & h) continue;
                const i2 w = wt[(e + st + (tid & 7)) & 7];
                const c2 wf = c2{(float)w.x, (float)w.y};
                const c2 tb = cmulw(u[e + h], wf);
                const i2 ib = imulw(v[e + h], w);
                const c2 ua = u[e];
                const i2 va = v[e];
                u[e] = cadd(ua, tb); u[e + h] = csub(ua, tb);
                v[e] = iadd(va, ib); v[e + h] = isub(va, ib);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) cnt += (u[e].x != (float)v[e].x) | (u[e].y != (float)v[e].y);   // |values| < 2^6 * 2^4 * 3^4 < 2^24: exact
    }
    if (cnt) atomicAdd(bad + 0, cnt);
    const int any = __syncthreads_or((int)(cnt != 0));
    if (any && tid == 0) atomicAdd(bad + 1, 1ull);
}
extern "C" int nv3_launch(void* stream, int blocks, int iters, unsigned long long* bad) {
    hipLaunchKernelGGL(nv3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, bad);
    return (int)hipGetLastError();
}


// ---- fourth neutral victim: nv3's butterfly network with an LDS exchange between the stages (float values and their int32 model travel
// through separate halves of 64 KiB, written as 8-byte pairs, read at a permuted slot) -- the read / packed arithmetic / write rhythm of an FFT pass
__global__ void __launch_bounds__(256) nv4_kernel(int iters, unsigned long long* __restrict__ bad) {
    c2* fb = reinterpret_cast<c2*>(smem);                     // 4096 float2
    i2* ib = reinterpret_cast<i2*>(smem + 32768);             // 4096 int2
    const int tid = threadIdx.x;
    const i2 wt[8] = {{1, 0}, {0, -1}, {-1, 0}, {0, 1}, {2, 0}, {1, 1}, {1, -1}, {0, 2}};
    unsigned long long cnt = 0;
    for (int it = 0; it < iters; ++it) {
        c2 u[16];
        i2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            v[e] = i2{((tid * 7 + e * 13 + it * 5) & 31) - 16, ((tid * 3 + e * 11 + it) & 31) - 16};
            u[e] = c2{(float)v[e].x, (float)v[e].y};
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int h = 1 << st;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (e & h) continue;
                const i2 w = wt[(e + st + (tid & 7)) & 7];
                const c2 wf = c2{(float)w.x, (float)w.y};
                const c2 tb = cmulw(u[e + h], wf);
                const i2 ibv = imulw(v[e + h], w);
                const c2 ua = u[e];
                const i2 va = v[e];
                u[e] = cadd(ua, tb); u[e + h] = csub(ua, tb);
                v[e] = iadd(va, ibv); v[e + h] = isub(va, ibv);
            }
            if (st == 1) {                                    // exchange after the second stage: slot e * 256 + tid, read back from thread tid ^ 21
#pragma unroll
                for (int e = 0; e < 16; ++e) { fb[e * 256 + tid] = u[e]; ib[e * 256 + tid] = v[e]; }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 16; ++e) { u[e] = fb[(15 - e) * 256 + (tid ^ 21)]; v[e] = ib[(15 - e) * 256 + (tid ^ 21)]; }
                __syncthreads();
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) cnt += (u[e].x != (float)v[e].x) | (u[e].y != (float)v[e].y);
    }
    if (cnt) atomicAdd(bad + 0, cnt);
    const int any = __syncthreads_or((int)(cnt != 0));
    if (any && tid == 0) atomicAdd(bad + 1, 1ull);
}
extern "C" int nv4_launch(void* stream, int blocks, int iters, unsigned long long* bad) {
    hipFuncSetAttribute((const void*)nv4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(nv4_kernel, dim3(blocks), dim3(256), 65536, (hipStream_t)stream, iters, bad);
    return (int)hipGetLastError();
}


// ---- fifth neutral victim: the same network on REAL-valued data (arbitrary mantissas, irrational twiddles), float2 arithmetic that the compiler packs,
// checked against the same network in double precision (never packed) with a tolerance far below the corruption seen in the FFT kernels (5e-3
// relative) and far above float rounding: |float - double| > 1e-4 * 2^4 counts.
struct d2 { double x, y; };
__global__ void __launch_bounds__(256) nv5_kernel(int iters, unsigned long long* __restrict__ bad) {
    const int tid = threadIdx.x;
    unsigned long long cnt = 0;
    for (int it = 0; it < iters; ++it) {
        c2 u[16];
        d2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned h0 = nv_hash(tid * 16 + e + it * 4096 + blockIdx.x * 7919u), h1 = nv_hash(h0 + 0x9e3779b9u);
            u[e] = c2{(float)(int)(h0 >> 8) * (1.f / 8388608.f) - 1.f, (float)(int)(h1 >> 8) * (1.f / 8388608.f) - 1.f};
            v[e] = d2{(double)u[e].x, (double)u[e].y};
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int h = 1 << st;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (e & h) continue;
                const float ang = -0.39269908169872414f * (float)((e * 3 + st * 5 + (tid & 15)) & 15);
                const c2 wf = c2{__cosf(ang), __sinf(ang)};
                const d2 wd = d2{(double)wf.x, (double)wf.y};
                const c2 tb = cmulw(u[e + h], wf);
                const d2 td = d2{v[e + h].x * wd.x - v[e + h].y * wd.y, v[e + h].x * wd.y + v[e + h].y * wd.x};
                const c2 ua = u[e];
                const d2 va = v[e];
                u[e] = cadd(ua, tb); u[e + h] = csub(ua, tb);
                v[e] = d2{va.x + td.x, va.y + td.y}; v[e + h] = d2{va.x - td.x, va.y - td.y};
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) cnt += (fabs((double)u[e].x - v[e].x) > 1.6e-3) | (fabs((double)u[e].y - v[e].y) > 1.6e-3);
    }
    if (cnt) atomicAdd(bad + 0, cnt);
    const int any = __syncthreads_or((int)(cnt != 0));
    if (any && tid == 0) atomicAdd(bad + 1, 1ull);
}
extern "C" int nv5_launch(void* stream, int blocks, int iters, unsigned long long* bad) {
    hipLaunchKernelGGL(nv5_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, bad);
    return (int)hipGetLastError();
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
#define NA_MODE(code, MF)                                                                                                              \
    if (v256 == code) {                                                                                                                \
        hipFuncSetAttribute((const void*)na_kernel<false, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);                 \
        hipLaunchKernelGGL((na_kernel<false, MF>), dim3((M / 128) * (N / 128)), dim3(256), lds_bytes, (hipStream_t)stream,             \
                           (const _Float16*)A, (const _Float16*)B, (float*)C, K, N / 128);                                             \
        return (int)hipGetLastError();                                                                                                 \
    }
    NA_MODE(2, 0)                                             // no MFMA
    NA_MODE(3, 3)                                             // f32 MFMA
    NA_MODE(4, 4)                                             // bf16 MFMA
#undef NA_MODE
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
