// Generic channels-last fp32 building blocks for the Demucs (HTDemucs) family: what `audio_separator`'s DemucsSeparator runs for
// `htdemucs_6s.yaml` (reference call site modules/separator/stem_separator.py:459-503; the network source lives in the un-vendored
// `demucs>=4.0.1`, requirements.txt:19 -- PARITY UNPINNED, restated in oracle/htdemucs_oracle.py).
//
// First HIP path for this family (correct and generic, fp32 storage, exact-f32 MFMA for the contractions; not yet tuned):
//   bgemm        : strided batched GEMM on v_mfma_f32_16x16x4_f32 (attention scores Q K^T and P V straight from the packed
//                  in-projection [B,T,3C], no head transposes)
//   softmax_rows : row softmax in place (one workgroup per row, row cached in registers)
//   norm         : GroupNorm(1 group) / LayerNorm / MyGroupNorm over groups of R rows x C channels, statistics in fp64 by a
//                  two-stage reduction, per-channel affine, fused GELU or GLU
//   act, scale_add, add_bcast, vec_fma, vec_div, reflect_pad : element-wise
//   tconv_fold   : the overlap-add half of ConvTranspose([K,1], [S,1]) (K = 2 S) after its 1x1-conv half
//   meanstd / affine_stats : Demucs' whole-sample normalisation (unbiased std) and its inverse
//   demucs_spec_in / spec_out / mix_out : layout glue between the STFT kernels' [B,4,F,T] and the network's [B,F,T,C]
#include "alsep_common.h"
#include "mma.h"

namespace {

constexpr int kNnThreads = 256;

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ double block_sum(double v, double* red) {
    // wave reduction, then across the 4 waves through LDS
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s = fmaxf(s, red[w]);
    return s;
}

// C[b1,b2][m][n] = alpha * sum_k A[b1,b2][m][k] * B[b1,b2][n][k], every operand addressed by element strides.
// One wave = 64 rows (m) x 32 columns (n); a workgroup = 4 waves side by side along n.
struct GemmStrides { int64_t b1, b2, r, k; };
__global__ void __launch_bounds__(kNnThreads)
nn_bgemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int nb2, int M, int N, int K,
                GemmStrides sa, GemmStrides sb, GemmStrides sc, float alpha, const float* __restrict__ bias, int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int b1 = blockIdx.z / nb2, b2 = blockIdx.z % nb2;
    const float* a = A + b1 * sa.b1 + b2 * sa.b2;
    const float* b = B + b1 * sb.b1 + b2 * sb.b2;
    float* c = C + b1 * sc.b1 + b2 * sc.b2;
    const int m0 = blockIdx.y * 64, n0 = (blockIdx.x * 4 + wave) * 32;
    f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t arow[4], brow[2];
    bool av_[4], bv_[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) { const int r = m0 + m * 16 + l15; av_[m] = r < M; arow[m] = (int64_t)(av_[m] ? r : 0) * sa.r; }
#pragma unroll
    for (int n = 0; n < 2; ++n) { const int r = n0 + n * 16 + l15; bv_[n] = r < N; brow[n] = (int64_t)(bv_[n] ? r : 0) * sb.r; }
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + lq;
        const bool kv = k < K;
        float af[4], bf[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = (kv && av_[m]) ? a[arow[m] + (int64_t)k * sa.k] : 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) bf[n] = (kv && bv_[n]) ? b[brow[n] + (int64_t)k * sb.k] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = n0 + n * 16 + l15;
        if (col >= N) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + m * 16 + 4 * lq + r;
                if (row < M) {
                    float v = alpha * acc[m][n][r];
                    if (bias) v += bias[col];
                    if (act == 3) v = gelu_erf(v);
                    else if (act == 5) v = tanhf(v);
                    c[(int64_t)row * sc.r + (int64_t)col * sc.k] = v;
                }
            }
    }
}

// The same product for the common layout -- A [M][K] and B [N][K] both K-contiguous (activations x Linear weights, Q K^T), K % 16 == 0,
// rows 16-byte aligned -- as an LDS-tiled kernel: a workgroup computes 128 x 128 of C, four waves as 2 x 2 (64 x 64 each: 16
// accumulator blocks), K in slices of 16 staged through LDS as the rows lie (global float4 -> ds_write_b128; rows padded to 24 floats:
// the ds_read_b128 of 16 rows x {lq, lq + 1} is conflict-free, scripts/lds_bank_sim.py), the next slice's global loads in flight during
// the 64 MFMAs of the current one.  A lane reads ONE float4 per operand row and slice and feeds element s of it to MFMA step s, i.e. the
// four k of a v_mfma_f32_16x16x4_f32 are {s, 4 + s, 8 + s, 12 + s} of the slice instead of four consecutive ones: the same products in
// another fp32 summation order than nn_bgemm_kernel (both are valid orders of the reference's sum; the oracles' tolerances hold for both).
// CT: C is row-major with N contiguous -- the operands swap MFMA roles so that a lane owns four consecutive columns (float4 stores).
constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 16, kGemmLD = 24;
// BNN: B is [K][N] with N contiguous (the `P V` product: B = V as it lies) -- staged [16 k][128 n + 4] and read one float per k.
// K need not be a multiple of 16: the last slice's loads are cut at K (A rows must be readable up to K rounded up to 4: the launcher
// checks the row stride), so a softmax matrix with 801 columns in rows of 804 floats takes this path.
template <bool CT, bool BNN>
__global__ void __launch_bounds__(kNnThreads)
nn_gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int nb2, int M, int N, int K,
                  GemmStrides sa, GemmStrides sb, GemmStrides sc, float alpha, const float* __restrict__ bias, int act) {
    constexpr int LDBN = kGemmBN + 4;
    float* As = reinterpret_cast<float*>(alsep_smem);                       // [2][128][24]
    float* Bs = As + 2 * kGemmBM * kGemmLD;                                  // [2][128][24], or BNN: [2][16][132]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int b1 = blockIdx.z / nb2, b2 = blockIdx.z % nb2;
    const float* a = A + b1 * sa.b1 + b2 * sa.b2;
    const float* b = B + b1 * sb.b1 + b2 * sb.b2;
    float* c = C + b1 * sc.b1 + b2 * sc.b2;
    const int m0 = blockIdx.y * kGemmBM, n0 = blockIdx.x * kGemmBN;
    // staging duty: two float4 per operand and slice.  K-contiguous operand: rows (tid >> 2) and + 64, k-quad tid & 3;
    // BNN: k rows (tid >> 5) and + 8, column quad tid & 31
    const int sr = tid >> 2, sq = tid & 3;
    const int bk = tid >> 5, bq = tid & 31;
    const float* ga[2];
    const float* gb[2];
    bool va[2], vb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ra = m0 + sr + 64 * h;
        va[h] = ra < M;
        ga[h] = a + (int64_t)(va[h] ? ra : 0) * sa.r + 4 * sq;
        if (BNN) {
            vb[h] = n0 + 4 * bq < N;                             // N % 4 == 0 on this path
            gb[h] = b + (int64_t)(bk + 8 * h) * sb.k + n0 + 4 * bq;
        } else {
            const int rb = n0 + sr + 64 * h;
            vb[h] = rb < N;
            gb[h] = b + (int64_t)(vb[h] ? rb : 0) * sb.r + 4 * sq;
        }
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra_[2], rb_[2];
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto cut = [&](f32x4 v, int kq) {                         // zero the elements at k >= K of the quad starting at kq
        if (kq + 4 <= K) return v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (kq + e >= K) v[e] = 0.f;
        return v;
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kq = k0 + 4 * sq;
            ra_[h] = (va[h] && kq < K) ? cut(*reinterpret_cast<const f32x4*>(ga[h] + k0), kq) : zero4;
            if (BNN) {
                const int kr = k0 + bk + 8 * h;
                rb_[h] = (vb[h] && kr < K) ? *reinterpret_cast<const f32x4*>(gb[h] + (int64_t)k0 * sb.k) : zero4;
            } else {
                rb_[h] = (vb[h] && kq < K) ? cut(*reinterpret_cast<const f32x4*>(gb[h] + k0), kq) : zero4;
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<f32x4*>(As + ((size_t)buf * kGemmBM + sr + 64 * h) * kGemmLD + 4 * sq) = ra_[h];
            if (BNN) *reinterpret_cast<f32x4*>(Bs + ((size_t)buf * kGemmBK + bk + 8 * h) * LDBN + 4 * bq) = rb_[h];
            else *reinterpret_cast<f32x4*>(Bs + ((size_t)buf * kGemmBN + sr + 64 * h) * kGemmLD + 4 * sq) = rb_[h];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = (K + kGemmBK - 1) / kGemmBK;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * kGemmBK);
        f32x4 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = *reinterpret_cast<const f32x4*>(As + ((size_t)buf * kGemmBM + wm * 64 + i * 16 + l15) * kGemmLD + 4 * lq);
            if (!BNN) bf[i] = *reinterpret_cast<const f32x4*>(Bs + ((size_t)buf * kGemmBN + wn * 64 + i * 16 + l15) * kGemmLD + 4 * lq);
        }
        const float* bl = Bs + ((size_t)buf * kGemmBK + 4 * lq) * LDBN + wn * 64 + l15;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float bs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bs[j] = BNN ? bl[s4 * LDBN + j * 16] : bf[j][s4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (CT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bs[j], af[i][s4], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s4], bs[j], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // epilogue.  CT: D rows = n (4 lq + r), D columns = m (l15): C[m][n .. n + 3] is one float4
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (CT) {
                const int row = m0 + wm * 64 + i * 16 + l15, col = n0 + wn * 64 + j * 16 + 4 * lq;
                if (row < M && col < N) {                        // N % 4 == 0 on this path: the four columns are all inside
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = alpha * acc[i][j][r];
                        if (bias) t += bias[col + r];
                        if (act == 3) t = gelu_erf(t);
                        else if (act == 5) t = tanhf(t);
                        v[r] = t;
                    }
                    *reinterpret_cast<f32x4*>(c + (int64_t)row * sc.r + col) = v;
                }
            } else {
                const int col = n0 + wn * 64 + j * 16 + l15;
                if (col < N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + wm * 64 + i * 16 + 4 * lq + r;
                        if (row < M) {
                            float t = alpha * acc[i][j][r];
                            if (bias) t += bias[col];
                            if (act == 3) t = gelu_erf(t);
                            else if (act == 5) t = tanhf(t);
                            c[(int64_t)row * sc.r + (int64_t)col * sc.k] = t;
                        }
                    }
                }
            }
        }
}

#define ALSEP_NN_F32S_GEMM
#include "nn_f32s.h"

// 0: not applicable; 1: B [N][K] (K contiguous); 2: B [K][N] (N contiguous).  A is K-contiguous with 16-byte aligned rows that can
// be read up to K rounded up to 4 (row stride >= that), likewise a K-contiguous B.
static int gemm_tiled_mode(const float* A, const float* B, int M, int N, int K, const GemmStrides& a, const GemmStrides& b) {
    static const int on = [] { const char* e = getenv("ALSEP_NN_GEMM_TN"); return e ? atoi(e) : 1; }();
    const int64_t k4 = (K + 3) / 4 * 4;
    if (!on || a.k != 1 || a.r % 4 || a.r < k4 || a.b1 % 4 || a.b2 % 4 || b.b1 % 4 || b.b2 % 4 || (((uintptr_t)A | (uintptr_t)B) & 15) ||
        (int64_t)M * N < 64 * 64 || K < 16)
        return 0;
    if (b.k == 1 && b.r % 4 == 0 && b.r >= k4) return 1;
    if (b.r == 1 && b.k % 4 == 0 && N % 4 == 0) return 2;
    return 0;
}
static bool gemm_ct_ok(const float* C, int N, const GemmStrides& c) {
    return c.k == 1 && c.r % 4 == 0 && c.b1 % 4 == 0 && c.b2 % 4 == 0 && N % 4 == 0 && ((uintptr_t)C & 15) == 0;
}
static int launch_gemm(alsep_ctx* ctx, const float* A, const float* B, float* C, int nb, int nb2, int M, int N, int K, GemmStrides a,
                       GemmStrides b, GemmStrides c, float alpha, const float* bias, int act) {
    const int mode = gemm_tiled_mode(A, B, M, N, K, a, b);
    ProfScope prof(ctx, ALSEP_PROF_NN_GEMM);
    prof.work(2.0 * nb * (double)M * N * K, 4.0 * nb * ((double)M * K + (double)N * K + (double)M * N));
    if (mode && ctx->nn_split && ctx->nn_range) {             // the same product as three f16 MFMAs per (hi, lo) pair (nn_f32s.h)
        const dim3 grid((unsigned)ceil_div64(N, GemmSCfg::BR), (unsigned)ceil_div64(M, GemmSCfg::BC), (unsigned)nb);
        const bool ct = gemm_ct_ok(C, N, c);
#define ALSEP_GEMM_GO(CT_, BNN_)                                                                                                     \
    do {                                                                                                                             \
        ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)nn_gemm_split_kernel<CT_, BNN_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)GemmSCfg::lds_bytes));                                                               \
        hipLaunchKernelGGL((nn_gemm_split_kernel<CT_, BNN_>), grid, dim3(kNsThreads), GemmSCfg::lds_bytes, ctx->stream, A, B, C, nb2, M, N, \
                           K, a, b, c, alpha, bias, act, ctx->nn_range);                                                             \
    } while (0)
        if (mode == 1) { if (ct) ALSEP_GEMM_GO(true, false); else ALSEP_GEMM_GO(false, false); }
        else { if (ct) ALSEP_GEMM_GO(true, true); else ALSEP_GEMM_GO(false, true); }
#undef ALSEP_GEMM_GO
        ALSEP_LAUNCH_CHECK(ctx, "nn_gemm_split_kernel");
        return ALSEP_OK;
    }
    if (mode) {
        const dim3 grid((unsigned)ceil_div64(N, kGemmBN), (unsigned)ceil_div64(M, kGemmBM), (unsigned)nb);
        const size_t lds = 2 * (size_t)(kGemmBM + kGemmBN) * kGemmLD * sizeof(float);       // (the [16][132] B image is smaller)
        const bool ct = gemm_ct_ok(C, N, c);
#define ALSEP_GEMM_GO(CT_, BNN_)                                                                                                     \
    hipLaunchKernelGGL((nn_gemm_tn_kernel<CT_, BNN_>), grid, dim3(kNnThreads), lds, ctx->stream, A, B, C, nb2, M, N, K, a, b, c, alpha, bias, act)
        if (mode == 1) { if (ct) ALSEP_GEMM_GO(true, false); else ALSEP_GEMM_GO(false, false); }
        else { if (ct) ALSEP_GEMM_GO(true, true); else ALSEP_GEMM_GO(false, true); }
#undef ALSEP_GEMM_GO
        ALSEP_LAUNCH_CHECK(ctx, "nn_gemm_tn_kernel");
        return ALSEP_OK;
    }
    hipLaunchKernelGGL(nn_bgemm_kernel, dim3((unsigned)ceil_div64(N, 128), (unsigned)ceil_div64(M, 64), (unsigned)nb), dim3(kNnThreads), 0,
                       ctx->stream, A, B, C, nb2, M, N, K, a, b, c, alpha, bias, act);
    ALSEP_LAUNCH_CHECK(ctx, "nn_bgemm_kernel");
    return ALSEP_OK;
}

// softmax over the last dimension, in place; one workgroup per row
__global__ void __launch_bounds__(kNnThreads)
nn_softmax_rows_kernel(float* __restrict__ x, int n, int ld) {
    float* red = reinterpret_cast<float*>(alsep_smem);
    double* redd = reinterpret_cast<double*>(alsep_smem + 64);
    float* row = x + (int64_t)blockIdx.x * ld;
    float mx = -3.4e38f;
    for (int i = threadIdx.x; i < n; i += kNnThreads) mx = fmaxf(mx, row[i]);
    mx = block_max(mx, red);
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kNnThreads) {
        const float e = expf(row[i] - mx);
        row[i] = e;
        s += (double)e;
    }
    s = block_sum(s, redd);
    const float inv = (float)(1.0 / s);
    for (int i = threadIdx.x; i < n; i += kNnThreads) row[i] *= inv;
}

// partial sums of a group: grid (NB, G); part[(g * NB + blk) * 2 + {0,1}] = sum, sum of squares (fp64)
__global__ void __launch_bounds__(kNnThreads)
nn_stats_partial_kernel(const float* __restrict__ x, int64_t per_group, int nb, double* __restrict__ part) {
    double* red = reinterpret_cast<double*>(alsep_smem);
    const int g = blockIdx.y, blk = blockIdx.x;
    const float* xg = x + (int64_t)g * per_group;
    const int64_t chunk = (per_group + nb - 1) / nb;
    const int64_t lo = (int64_t)blk * chunk, hi = lo + chunk < per_group ? lo + chunk : per_group;
    double s = 0.0, q = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kNnThreads) {
        const double v = (double)xg[i];
        s += v;
        q += v * v;
    }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        part[((int64_t)g * nb + blk) * 2] = s;
        part[((int64_t)g * nb + blk) * 2 + 1] = q;
    }
}
// stats[g] = (mean, rstd or std): mode 0 -> rstd = 1/sqrt(var_biased + eps) (normalisation layers); mode 1 -> unbiased std
__global__ void nn_stats_final_kernel(const double* __restrict__ part, int nb, int64_t per_group, int G, float eps, int mode,
                                      float* __restrict__ stats) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nb; ++b) {
        s += part[((int64_t)g * nb + b) * 2];
        q += part[((int64_t)g * nb + b) * 2 + 1];
    }
    const double n = (double)per_group, mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * g] = (float)mean;
    if (mode == 0) stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    else stats[2 * g + 1] = (float)sqrt(per_group > 1 ? var * n / (n - 1.0) : 0.0);
}
// y = act(((x - mean_g) * rstd_g) * gamma[c] + beta[c]); act 0 none, 3 GELU, 4 GLU (C -> C/2 output channels)
__global__ void __launch_bounds__(kNnThreads)
nn_norm_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma,
                     const float* __restrict__ beta, const float* __restrict__ stats, int64_t n_out, int64_t rows_per_group, int C,
                     int act) {
    const int Co = act == 4 ? C / 2 : C;
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * kNnThreads) {
        const int c = (int)(i % Co);
        const int64_t row = i / Co;
        const int64_t g = row / rows_per_group;
        const float mean = stats[2 * g], rstd = stats[2 * g + 1];
        const float* xr = x + row * C;
        float v = (xr[c] - mean) * rstd;
        if (gamma) v = fmaf(v, gamma[c], beta[c]);
        if (act == 3) v = gelu_erf(v);
        else if (act == 4) {
            float u = (xr[c + Co] - mean) * rstd;
            if (gamma) u = fmaf(u, gamma[c + Co], beta[c + Co]);
            v *= sigmoidf_(u);
        }
        y[i] = v;
    }
}

__global__ void __launch_bounds__(kNnThreads)
nn_act_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n_out, int C, int act) {
    const int Co = act == 4 ? C / 2 : C;
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * kNnThreads) {
        const int c = (int)(i % Co);
        const int64_t row = i / Co;
        const float v = x[row * C + c];
        float r = v;
        if (act == 1) r = fmaxf(v, 0.f);
        else if (act == 3) r = gelu_erf(v);
        else if (act == 4) r = v * sigmoidf_(x[row * C + c + Co]);
        else if (act == 5) r = tanhf(v);
        y[i] = r;
    }
}

// y = a + scale[c] * b   (scale == nullptr: 1)
__global__ void __launch_bounds__(kNnThreads)
nn_scale_add_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ scale, float* __restrict__ y,
                    int64_t n, int C) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads)
        y[i] = scale ? fmaf(scale[i % C], b[i], a[i]) : a[i] + b[i];
}
// The same kernels four channels per thread (C % 4 == 0, 16-byte aligned rows, fewer than 2^31 quads: 32-bit index arithmetic instead of
// three 64-bit divisions per element, 16-byte accesses): HTDemucs spends a quarter of its kernel time in these.  Element for element the
// arithmetic of the scalar kernels above.
__global__ void __launch_bounds__(kNnThreads)
nn_norm_apply4_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                      const float* __restrict__ stats, unsigned n4, unsigned rows_per_group, int C, int act) {
    const unsigned Co4 = (unsigned)(act == 4 ? C / 2 : C) / 4;
    for (unsigned i = blockIdx.x * kNnThreads + threadIdx.x; i < n4; i += gridDim.x * kNnThreads) {
        const unsigned row = i / Co4, c = (i - row * Co4) * 4, g = row / rows_per_group;
        const float mean = stats[2 * g], rstd = stats[2 * g + 1];
        const float* xr = x + (int64_t)row * C + c;
        f32x4 v = *reinterpret_cast<const f32x4*>(xr);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd;
        if (gamma) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], gm[e], bt[e]);
        }
        if (act == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if (act == 4) {
            const unsigned Co = Co4 * 4;
            f32x4 u = *reinterpret_cast<const f32x4*>(xr + Co);
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = (u[e] - mean) * rstd;
            if (gamma) {
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c + Co), bt = *reinterpret_cast<const f32x4*>(beta + c + Co);
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = fmaf(u[e], gm[e], bt[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= sigmoidf_(u[e]);
        }
        *reinterpret_cast<f32x4*>(y + (int64_t)i * 4) = v;
    }
}
__global__ void __launch_bounds__(kNnThreads)
nn_act4_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned n4, int C, int act) {
    const unsigned Co4 = (unsigned)(act == 4 ? C / 2 : C) / 4;
    for (unsigned i = blockIdx.x * kNnThreads + threadIdx.x; i < n4; i += gridDim.x * kNnThreads) {
        const unsigned row = i / Co4, c = (i - row * Co4) * 4;
        const float* xr = x + (int64_t)row * C + c;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr);
        f32x4 r = v;
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaxf(v[e], 0.f);
        } else if (act == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = gelu_erf(v[e]);
        } else if (act == 4) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(xr + Co4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = v[e] * sigmoidf_(u[e]);
        } else if (act == 5) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = tanhf(v[e]);
        }
        *reinterpret_cast<f32x4*>(y + (int64_t)i * 4) = r;
    }
}
__global__ void __launch_bounds__(kNnThreads)
nn_scale_add4_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ scale, float* __restrict__ y,
                     unsigned n4, unsigned C4) {
    for (unsigned i = blockIdx.x * kNnThreads + threadIdx.x; i < n4; i += gridDim.x * kNnThreads) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + (int64_t)i * 4), bv = *reinterpret_cast<const f32x4*>(b + (int64_t)i * 4);
        f32x4 r;
        if (scale) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(scale + (i % C4) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaf(sv[e], bv[e], av[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = av[e] + bv[e];
        }
        *reinterpret_cast<f32x4*>(y + (int64_t)i * 4) = r;
    }
}
// nn_stats_partial_kernel with 16-byte loads (per_group % 4 == 0, aligned x): a block's range is rounded to whole quads
__global__ void __launch_bounds__(kNnThreads)
nn_stats_partial4_kernel(const float* __restrict__ x, int64_t per_group, int nb, double* __restrict__ part) {
    double* red = reinterpret_cast<double*>(alsep_smem);
    const int g = blockIdx.y, blk = blockIdx.x;
    const float* xg = x + (int64_t)g * per_group;
    const int64_t q_all = per_group / 4, chunk = (q_all + nb - 1) / nb;
    const int64_t lo = (int64_t)blk * chunk, hi = lo + chunk < q_all ? lo + chunk : q_all;
    double s = 0.0, q = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kNnThreads) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xg + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double d = (double)v[e];
            s += d;
            q += d * d;
        }
    }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        part[((int64_t)g * nb + blk) * 2] = s;
        part[((int64_t)g * nb + blk) * 2 + 1] = q;
    }
}
// y[i] += s * e[((i / inner) % period) * C + i % C]
__global__ void __launch_bounds__(kNnThreads)
nn_add_bcast_kernel(float* __restrict__ y, const float* __restrict__ e, float s, int64_t n, int64_t inner, int period, int C) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads)
        y[i] = fmaf(s, e[((i / inner) % period) * C + i % C], y[i]);
}
// y[r * ys + i] += w[i] * x[r * xs + i]
__global__ void __launch_bounds__(kNnThreads)
nn_vec_fma_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ w, int64_t rows, int64_t n,
                  int64_t ys, int64_t xs) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < rows * n; i += (int64_t)gridDim.x * kNnThreads) {
        const int64_t r = i / n, j = i % n;
        y[r * ys + j] = fmaf(w[j], x[r * xs + j], y[r * ys + j]);
    }
}
__global__ void __launch_bounds__(kNnThreads)
nn_vec_div_kernel(float* __restrict__ y, const float* __restrict__ w, int64_t rows, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < rows * n; i += (int64_t)gridDim.x * kNnThreads)
        {   // a sample no window covers with a positive weight (num_overlap 1: both fades are 0 at a chunk boundary) is 0 / 0 upstream,
            // which the training project's demix_track turns into 0 with nan_to_num
            const float d = w[i % n];
            y[i] = d != 0.f ? y[i] / d : 0.f;
        }
}
// torch F.pad(mode="reflect") along the last axis: y[r][i] = x[r][reflect(i - left)], length n -> n + left + right
__global__ void __launch_bounds__(kNnThreads)
nn_reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int64_t n, int64_t left, int64_t right) {
    const int64_t m = n + left + right;
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < rows * m; i += (int64_t)gridDim.x * kNnThreads) {
        const int64_t r = i / m;
        int64_t j = i % m - left;
        if (j < 0) j = -j;
        if (j >= n) j = 2 * (n - 1) - j;
        y[i] = x[r * n + j];
    }
}

// ConvTranspose([K,1], stride [S,1]) with K = 2 S, second half: g [B, I, J, K * Cout] holds G[i][k][co] = sum_ci x[i][ci] W[ci][co][k];
// y[b, o, j, co] = act(G[i][r] + G[i - 1][r + S] + bias[co]), o_full = o + pad = i S + r; y is [B, Lout, J, Cout]
__global__ void __launch_bounds__(kNnThreads)
nn_tconv_fold_kernel(const float* __restrict__ g, const float* __restrict__ bias, float* __restrict__ y, int64_t n, int I, int J,
                     int Cout, int S, int pad, int Lout, int act) {
    const int K = 2 * S;
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int co = (int)(i % Cout);
        const int64_t p = i / Cout;
        const int j = (int)(p % J);
        const int o = (int)((p / J) % Lout);
        const int64_t b = p / ((int64_t)J * Lout);
        const int of = o + pad, ii = of / S, r = of % S;
        float v = bias ? bias[co] : 0.f;
        if (ii < I) v += g[(((b * I + ii) * J + j) * (int64_t)K + r) * Cout + co];
        if (ii >= 1 && ii - 1 < I) v += g[(((b * I + ii - 1) * J + j) * (int64_t)K + r + S) * Cout + co];
        y[i] = act == 3 ? gelu_erf(v) : v;
    }
}

// inverse == 0: y = (x - mean) / (eps + std); inverse == 1: y = x * std + mean; stats[2 s] = mean, [2 s + 1] = std of sample s
__global__ void __launch_bounds__(kNnThreads)
nn_affine_stats_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ stats, int64_t n,
                       int64_t per_sample, float eps, int inverse) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int64_t s = i / per_sample;
        const float mean = stats[2 * s], sd = stats[2 * s + 1];
        y[i] = inverse ? fmaf(x[i], sd, mean) : (x[i] - mean) / (eps + sd);
    }
}

// HTDemucs._spec / _magnitude: spec [B, 4 = (L_re, L_im, R_re, R_im), F, Tt] (alsep_stft, reference layout) ->
// y [B, F, T, 4] = scale * spec[..., t_off : t_off + T]
__global__ void __launch_bounds__(kNnThreads)
demucs_spec_in_kernel(const float* __restrict__ spec, float* __restrict__ y, int64_t n, int F, int Tt, int T, int t_off, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int c = (int)(i & 3);
        const int64_t p = i >> 2;
        const int t = (int)(p % T), f = (int)((p / T) % F);
        const int64_t b = p / ((int64_t)T * F);
        y[i] = scale * spec[((b * 4 + c) * F + f) * (int64_t)Tt + t_off + t];
    }
}
// HTDemucs._mask (cac) + the frame padding of _ispec: x [B, F, T, S * 4] (normalised network output), stats of the input
// spectrogram -> spec [B * S, 4, F, Tt]; frames [t_off, t_off + T) = scale * (x * std + mean), the others zero
__global__ void __launch_bounds__(kNnThreads)
demucs_spec_out_kernel(const float* __restrict__ x, const float* __restrict__ stats, float* __restrict__ spec, int64_t n, int S, int F,
                       int Tt, int T, int t_off, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int tt = (int)(i % Tt);
        const int f = (int)((i / Tt) % F);
        const int c = (int)((i / ((int64_t)Tt * F)) & 3);
        const int64_t bs = i / ((int64_t)Tt * F * 4);
        const int64_t b = bs / S;
        const int s = (int)(bs % S);
        const int t = tt - t_off;
        float v = 0.f;
        if (t >= 0 && t < T) v = scale * fmaf(x[((b * F + f) * (int64_t)T + t) * (S * 4) + s * 4 + c], stats[2 * b + 1], stats[2 * b]);
        spec[i] = v;
    }
}
// out [B, S, 2, L] = (xt [B, L, S * 2] * stdt + meant) + xs [B * S, 2, L]
__global__ void __launch_bounds__(kNnThreads)
demucs_mix_out_kernel(const float* __restrict__ xt, const float* __restrict__ statst, const float* __restrict__ xs,
                      float* __restrict__ out, int64_t n, int S, int64_t L) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int64_t l = i % L;
        const int ch = (int)((i / L) & 1);
        const int64_t bs = i / (2 * L);
        const int64_t b = bs / S;
        const int s = (int)(bs % S);
        out[i] = fmaf(xt[(b * L + l) * (S * 2) + s * 2 + ch], statst[2 * b + 1], statst[2 * b]) + xs[i];
    }
}

// y [B, L, C] = x [B, C, L]  (and back with the roles of L and C swapped)
__global__ void __launch_bounds__(kNnThreads)
nn_swap_last2_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int64_t C, int64_t L) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int64_t c = i % C, l = (i / C) % L, b = i / (C * L);
        y[i] = x[(b * C + c) * L + l];
    }
}

// lucidrains RMSNorm: y = x / max(||x||_2, 1e-12) * sqrt(C) * gamma over the last axis; row r of x at x + r * x_stride, of y at y + r * y_stride
// (in place on a column slice of a wider matrix when both strides are that matrix's width).  One wave per row.
__global__ void __launch_bounds__(kNnThreads)
nn_rmsnorm_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma, int64_t rows, int C, int64_t x_stride,
                  int64_t y_stride) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * x_stride;
    double ss = 0.0;
    for (int c = lane; c < C; c += 64) ss += (double)xr[c] * (double)xr[c];
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = sqrtf((float)C) / fmaxf((float)sqrt(ss), 1e-12f);
    float* yr = y + row * y_stride;
    for (int c = lane; c < C; c += 64) yr[c] = xr[c] * inv * gamma[c];
}
// rotary embedding (rotary_embedding_torch, interleaved pairs, theta 10000) in place on heads x d columns starting at col_off of every row;
// position of row r = (r / pos_div) % pos_mod
__global__ void __launch_bounds__(kNnThreads)
nn_rotary_kernel(float* __restrict__ x, int64_t n_pairs, int64_t row_stride, int col_off, int heads, int d, int64_t pos_div, int64_t pos_mod) {
    const int half = d / 2;
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * kNnThreads) {
        const int j = (int)(i % half);
        const int h = (int)((i / half) % heads);
        const int64_t r = i / ((int64_t)half * heads);
        const float pos = (float)((r / pos_div) % pos_mod);
        const float inv = 1.f / powf(10000.f, (float)(2 * j) / (float)d);
        const float ang = pos * inv, cs = cosf(ang), sn = sinf(ang);
        float* p = x + r * row_stride + col_off + h * d + 2 * j;
        const float a = p[0], b = p[1];
        p[0] = a * cs - b * sn;
        p[1] = b * cs + a * sn;
    }
}
// out[r, h * d + j] *= sigmoid(gates[r, h])
__global__ void __launch_bounds__(kNnThreads)
nn_gate_kernel(float* __restrict__ out, const float* __restrict__ gates, int64_t n, int heads, int d) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int64_t r = i / ((int64_t)heads * d);
        const int h = (int)((i / d) % heads);
        out[i] *= sigmoidf_(gates[r * heads + h]);
    }
}
// Roformer band split input: feat[t, 2 i + c] = spec[(s * 2 + c), f, t] for the i-th entry m = midx[i] of the concatenated band index
// lists (m = 2 f + s: frequency-major, channel-minor); spec [4, F, T] as alsep_stft writes it
__global__ void __launch_bounds__(kNnThreads)
roformer_gather_kernel(const float* __restrict__ spec, const int* __restrict__ midx, float* __restrict__ feat, int64_t n, int n_idx, int F,
                       int T) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int c = (int)(i & 1);
        const int k = (int)((i >> 1) % n_idx);
        const int64_t t = i / (2 * (int64_t)n_idx);
        const int m = midx[k], s = m & 1, f = m >> 1;
        feat[i] = spec[((int64_t)(s * 2 + c) * F + f) * T + t];
    }
}
// Roformer output: for merged bin m = 2 f + s and frame t, mask = (1 / max(n_occ, 1)) * sum over the occurrences o of m in the bands of
// GLU(h)[o] (complex: columns col_a[o], col_a[o] + 1 gated by col_g[o], col_g[o] + 1 of h [T, H]); out = spec * mask (complex), same layout
__global__ void __launch_bounds__(kNnThreads)
roformer_mask_kernel(const float* __restrict__ spec, const float* __restrict__ h, const int* __restrict__ occ_start,
                     const int* __restrict__ col_a, const int* __restrict__ col_g, float* __restrict__ out, int64_t n, int F, int T, int H) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int t = (int)(i % T);
        const int f = (int)((i / T) % F);
        const int s = (int)(i / ((int64_t)T * F));
        const int m = 2 * f + s;
        const int o0 = occ_start[m], o1 = occ_start[m + 1];
        const float* hr = h + (int64_t)t * H;
        float mr = 0.f, mi = 0.f;
        for (int o = o0; o < o1; ++o) {
            mr += hr[col_a[o]] * sigmoidf_(hr[col_g[o]]);
            mi += hr[col_a[o] + 1] * sigmoidf_(hr[col_g[o] + 1]);
        }
        const float inv = 1.f / (float)(o1 - o0 > 1 ? o1 - o0 : 1);
        mr *= inv;
        mi *= inv;
        const int64_t ire = ((int64_t)(s * 2) * F + f) * T + t, iim = ((int64_t)(s * 2 + 1) * F + f) * T + t;
        const float zr = spec[ire], zi = spec[iim];
        out[ire] = zr * mr - zi * mi;
        out[iim] = zr * mi + zi * mr;
    }
}

// InstanceNorm2d statistics on channels-last data x [P pixels, C]: partial sums per channel over a slab of pixels; thread = channel
// (consecutive threads read consecutive channels of one pixel: coalesced); part [NB][C][2] in fp64
__global__ void __launch_bounds__(kNnThreads)
nn_chan_stats_partial_kernel(const float* __restrict__ x, int64_t P, int C, int nb, double* __restrict__ part) {
    const int c = blockIdx.x * kNnThreads + threadIdx.x;
    if (c >= C) return;
    const int64_t chunk = (P + nb - 1) / nb;
    const int64_t lo = (int64_t)blockIdx.y * chunk, hi = lo + chunk < P ? lo + chunk : P;
    double s = 0.0, q = 0.0;
    for (int64_t p = lo; p < hi; ++p) {
        const double v = (double)x[p * C + c];
        s += v;
        q += v * v;
    }
    part[((int64_t)blockIdx.y * C + c) * 2] = s;
    part[((int64_t)blockIdx.y * C + c) * 2 + 1] = q;
}
// the same for C <= 256 with 256 % C == 0: the workgroup's 256 threads cover 256 / C pixels at a time (thread = (pixel group, channel):
// still one contiguous run of channels per pixel) instead of leaving 256 - C threads idle behind a C-thread serial loop; the pixel
// groups' fp64 partials meet in LDS in a fixed order
__global__ void __launch_bounds__(kNnThreads)
nn_chan_stats_partial_small_kernel(const float* __restrict__ x, int64_t P, int C, int nb, double* __restrict__ part) {
    double* red = reinterpret_cast<double*>(alsep_smem);      // [256][2]
    const int c = threadIdx.x % C, g = threadIdx.x / C, G = kNnThreads / C;
    const int64_t chunk = (P + nb - 1) / nb;
    const int64_t lo = (int64_t)blockIdx.y * chunk, hi = lo + chunk < P ? lo + chunk : P;
    double s = 0.0, q = 0.0;
    for (int64_t p = lo + g; p < hi; p += G) {
        const double v = (double)x[p * C + c];
        s += v;
        q += v * v;
    }
    red[2 * threadIdx.x] = s;
    red[2 * threadIdx.x + 1] = q;
    __syncthreads();
    if (g == 0) {
        for (int k = 1; k < G; ++k) {
            s += red[2 * (k * C + c)];
            q += red[2 * (k * C + c) + 1];
        }
        part[((int64_t)blockIdx.y * C + c) * 2] = s;
        part[((int64_t)blockIdx.y * C + c) * 2 + 1] = q;
    }
}
// The same statistics for C % 4 == 0, C <= 1024 (every layer of MDX23C): thread = (pixel lane g, channel quad c4), one float4 per
// pixel and thread, four pixels in flight; a workgroup takes a slab of pixels, the pixel lanes' fp64 partials meet in LDS in a fixed
// order.  Against nn_chan_stats_partial_small_kernel (scalar loads, one dependent chain per thread, <= 256 workgroups: 160 GB/s on
// the 65 536 x 128 level-0 tensors, a quarter of MDX23C's time) this is a streaming read.
__global__ void __launch_bounds__(kNnThreads)
nn_chan_stats4_kernel(const float* __restrict__ x, int64_t P, int C, int nb, double* __restrict__ part) {
    double* red = reinterpret_cast<double*>(alsep_smem);      // [threads][8]
    const int Q = C / 4, G = kNnThreads / Q;                   // G >= 1 pixel lanes; threads beyond G * Q idle
    const int c4 = threadIdx.x % Q, g = threadIdx.x / Q;
    const int64_t chunk = (P + nb - 1) / nb;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < P ? lo + chunk : P;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    if (g < G) {
        const float* xp = x + 4 * c4;
        int64_t p = lo + g;
        for (; p + 3 * (int64_t)G < hi; p += 4 * (int64_t)G) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(xp + (p + (int64_t)u * G) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double d = (double)v[u][e]; s[e] += d; q[e] = fma(d, d, q[e]); }
        }
        for (; p < hi; p += G) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + p * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; s[e] += d; q[e] = fma(d, d, q[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[8 * threadIdx.x + e] = s[e]; red[8 * threadIdx.x + 4 + e] = q[e]; }
    __syncthreads();
    if (g == 0) {
        for (int k = 1; k < G; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += red[8 * (k * Q + c4) + e]; q[e] += red[8 * (k * Q + c4) + 4 + e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            part[((int64_t)blockIdx.x * C + 4 * c4 + e) * 2] = s[e];
            part[((int64_t)blockIdx.x * C + 4 * c4 + e) * 2 + 1] = q[e];
        }
    }
}
// reduction of the slabs' partials: 16 channels per workgroup, sixteen strands per channel (slab b = strand, strand + 16, ...; four
// loads in flight per strand), combined in a fixed order.  (64 channels x 4 strands on <= 12 workgroups was a 256-deep chain of
// dependent-latency loads: 33 us per call, 11 % of MDX23C's half-precision chunk.)
__global__ void __launch_bounds__(kNnThreads)
nn_chan_stats_final4_kernel(const double* __restrict__ part, int nb, int64_t P, int C, float eps, float* __restrict__ stats) {
    __shared__ double red[kNnThreads][2];
    constexpr int ST = kNnThreads / 16;
    const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const double* pc = part + 2 * (int64_t)c;
        int b = j;
        for (; b + 3 * ST < nb; b += 4 * ST) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[2 * u] = pc[(int64_t)(b + u * ST) * C * 2];
                v[2 * u + 1] = pc[(int64_t)(b + u * ST) * C * 2 + 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += v[2 * u]; q += v[2 * u + 1]; }
        }
        for (; b < nb; b += ST) { s += pc[(int64_t)b * C * 2]; q += pc[(int64_t)b * C * 2 + 1]; }
    }
    red[threadIdx.x][0] = s;
    red[threadIdx.x][1] = q;
    __syncthreads();
    if (j == 0 && c < C) {
        for (int k = 1; k < ST; ++k) { s += red[k * 16 + cl][0]; q += red[k * 16 + cl][1]; }
        const double mean = s / (double)P;
        double var = q / (double)P - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * c] = (float)mean;
        stats[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
__global__ void nn_chan_stats_final_kernel(const double* __restrict__ part, int nb, int64_t P, int C, float eps, float* __restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nb; ++b) {
        s += part[((int64_t)b * C + c) * 2];
        q += part[((int64_t)b * C + c) * 2 + 1];
    }
    const double mean = s / (double)P;
    double var = q / (double)P - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * c] = (float)mean;
    stats[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
__global__ void __launch_bounds__(kNnThreads)
nn_chan_norm_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                          const float* __restrict__ stats, int64_t n, int C, int act) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int c = (int)(i % C);
        float v = (x[i] - stats[2 * c]) * stats[2 * c + 1];
        if (gamma) v = fmaf(v, gamma[c], beta[c]);
        y[i] = act == 3 ? gelu_erf(v) : v;
    }
}
// the same, four consecutive channels per thread (C % 4 == 0, 16-byte aligned x / y): vector loads of the data, the statistics and the affine
// parameters instead of one modulo and four scalar loads per element
__global__ void __launch_bounds__(kNnThreads)
nn_chan_norm_apply4_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                           const float* __restrict__ stats, int64_t n4, int C, int act) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kNnThreads) {
        const int c = (int)((4 * i) % C);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + 4 * i);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(stats + 2 * c), s1 = *reinterpret_cast<const f32x4*>(stats + 2 * c + 4);
        const float mean[4] = {s0[0], s0[2], s1[0], s1[2]}, rstd[4] = {s0[1], s0[3], s1[1], s1[3]};
        f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (gamma) { g = *reinterpret_cast<const f32x4*>(gamma + c); b = *reinterpret_cast<const f32x4*>(beta + c); }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = (xv[e] - mean[e]) * rstd[e];
            if (gamma) v = fmaf(v, g[e], b[e]);
            o[e] = act == 3 ? gelu_erf(v) : v;
        }
        *reinterpret_cast<f32x4*>(y + 4 * i) = o;
    }
}
// the same with the result stored as IEEE half (the A operand of the half-precision convolution that follows; C % 4 == 0)
__global__ void __launch_bounds__(kNnThreads)
nn_chan_norm_apply_h_kernel(const float* __restrict__ x, _Float16* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                            const float* __restrict__ stats, int64_t n4, int C, int act) {
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    // four consecutive channels per thread: their (mean, rstd) pairs, gammas and betas as four 16-byte loads (C % 4 == 0: the quad never
    // straddles a pixel), two quads in flight per thread
    const int64_t stride = (int64_t)gridDim.x * kNnThreads;
    for (int64_t i0 = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        f32x4 xv[2], s0[2], s1[2], g[2], b[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = i0 + u * stride;
            ok[u] = i < n4;
            const int64_t ic = ok[u] ? i : i0;
            const int c = (int)((4 * ic) % C);
            xv[u] = *reinterpret_cast<const f32x4*>(x + 4 * ic);
            s0[u] = *reinterpret_cast<const f32x4*>(stats + 2 * c);           // mean c, rstd c, mean c+1, rstd c+1
            s1[u] = *reinterpret_cast<const f32x4*>(stats + 2 * c + 4);
            if (gamma) { g[u] = *reinterpret_cast<const f32x4*>(gamma + c); b[u] = *reinterpret_cast<const f32x4*>(beta + c); }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
            const float mean[4] = {s0[u][0], s0[u][2], s1[u][0], s1[u][2]}, rstd[4] = {s0[u][1], s0[u][3], s1[u][1], s1[u][3]};
            h16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = (xv[u][e] - mean[e]) * rstd[e];
                if (gamma) v = fmaf(v, g[u][e], b[u][e]);
                o[e] = (_Float16)(act == 3 ? gelu_erf(v) : v);
            }
            *reinterpret_cast<h16x4*>(y + 4 * (i0 + u * stride)) = o;
        }
    }
}
// the same with the half result stored TRANSPOSED per frame: x [T][F][C] float32 -> y [T][C][F] IEEE half, i.e. frequency-contiguous rows
// per (frame, channel): the "weight-side" operand of the TDF Linear over the frequency axis as nn_gemm_hh_kernel wants it (both operands
// K-contiguous; the Linear is then one batched GEMM per frame with the shared weight matrix as A).  32 x 32 tiles through LDS: reads
// contiguous along c, writes contiguous along f.
__global__ void __launch_bounds__(kNnThreads)
nn_chan_norm_apply_ht_kernel(const float* __restrict__ x, _Float16* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                             const float* __restrict__ stats, int F, int C, int act) {
    float* tile = reinterpret_cast<float*>(alsep_smem);      // [32][33]
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int f0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int64_t t = blockIdx.z;
    const int c = c0 + tx;
    float mean = 0.f, rstd = 0.f, g = 1.f, b = 0.f;
    if (c < C) {
        mean = stats[2 * c];
        rstd = stats[2 * c + 1];
        if (gamma) { g = gamma[c]; b = beta[c]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = f0 + ty + 8 * j;
        float v = 0.f;
        if (f < F && c < C) {
            v = (x[(t * F + f) * C + c] - mean) * rstd;
            if (gamma) v = fmaf(v, g, b);
            if (act == 3) v = gelu_erf(v);
        }
        tile[(ty + 8 * j) * 33 + tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cc = c0 + ty + 8 * j, f = f0 + tx;
        if (cc < C && f < F) y[(t * C + cc) * F + f] = (_Float16)tile[tx * 33 + ty + 8 * j];
    }
}
__global__ void __launch_bounds__(kNnThreads)
nn_mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) y[i] = a[i] * b[i];
}
// ConvTranspose2d(kernel = stride = (2, 2)), second half: g [H, W, 4 * Cout] (1x1 conv, columns (dy*2+dx)*Cout + co) ->
// y [2H, 2W, y_ct] channel slice [y_c0, y_c0 + Cout)
__global__ void __launch_bounds__(kNnThreads)
nn_depth_to_space2_kernel(const float* __restrict__ g, float* __restrict__ y, int64_t n, int H, int W, int Cout, int y_ct, int y_c0) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int co = (int)(i % Cout);
        const int64_t p = i / Cout;
        const int ox = (int)(p % (2 * W)), oy = (int)(p / (2 * W));
        y[p * y_ct + y_c0 + co] = g[(((int64_t)(oy >> 1) * W + (ox >> 1)) * 4 + (oy & 1) * 2 + (ox & 1)) * Cout + co];
    }
}
// MDX23C input: spec [4, dim_f, T] (alsep_stft reference layout) -> x [T, f, 4 k] with f = dim_f / k sub-band bins:
// x[t, ff, c2 * k + kk] = spec[c2, kk * f + ff, t]   (STFT.__call__ + cac2cws, transposed to frames-major)
__global__ void __launch_bounds__(kNnThreads)
mdx23c_spec_in_kernel(const float* __restrict__ spec, float* __restrict__ x, int64_t n, int f, int k, int T) {
    const int Cn = 4 * k, dim_f = f * k;
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int ch = (int)(i % Cn), kk = ch % k, c2 = ch / k;
        const int ff = (int)((i / Cn) % f);
        const int64_t t = i / ((int64_t)Cn * f);
        x[i] = spec[((int64_t)c2 * dim_f + kk * f + ff) * T + t];
    }
}
// MDX23C output: y [T, f, S * 4 k] -> spec [S, 4, dim_f, T]: spec[s, c2, kk * f + ff, t] = y[t, ff, (s * 4 + c2) * k + kk]   (cws2cac)
__global__ void __launch_bounds__(kNnThreads)
mdx23c_spec_out_kernel(const float* __restrict__ y, float* __restrict__ spec, int64_t n, int S, int f, int k, int T) {
    const int dim_f = f * k, Cn = S * 4 * k;
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int t = (int)(i % T);
        const int fb = (int)((i / T) % dim_f), kk = fb / f, ff = fb % f;
        const int sc = (int)(i / ((int64_t)T * dim_f));       // s * 4 + c2
        spec[i] = y[((int64_t)t * f + ff) * Cn + sc * k + kk];
    }
}

// ---- VR multi-band front / back end (reference modules/rvc/infer/lib/uvr5_pack/lib_v5/spec_utils.py, modules/rvc/infer/modules/uvr5/vr.py) ----
// complex spectrograms are float2-interleaved [2 channels, bins, frames]; band spectrograms from / for the FFT kernels are [4 = (L_re, L_im,
// R_re, R_im), Fb, Tb]
// out[c, o0 + i, t] = gain[i] * band[(2c, 2c+1), f0 + i, t0 + t], i < h, t < l   (combine_spectrograms :95-125; the high-end crop of vr.py:88-96)
__global__ void __launch_bounds__(kNnThreads)
vr_band_crop_kernel(const float* __restrict__ band, float* __restrict__ out, const float* __restrict__ gain, int64_t n, int Fb, int Tb, int f0,
                    int t0, int h, int l, int out_bins, int o0) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int t = (int)(i % l), b = (int)((i / l) % h), c = (int)(i / ((int64_t)l * h));
        const float g = gain ? gain[b] : 1.f;
        const int64_t src = ((int64_t)(2 * c) * Fb + f0 + b) * Tb + t0 + t;
        const int64_t dst = (((int64_t)c * out_bins + o0 + b) * l + t) * 2;
        out[dst] = g * band[src];
        out[dst + 1] = g * band[src + (int64_t)Fb * Tb];
    }
}
// y = pred * exp(i angle(X)), v = X - y   (vr.py:110-111; angle(0) = 0)
__global__ void __launch_bounds__(kNnThreads)
vr_split_pred_kernel(const float* __restrict__ pred, const float* __restrict__ X, float* __restrict__ y, float* __restrict__ v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const float xr = X[2 * i], xi = X[2 * i + 1];
        const float m = hypotf(xr, xi);
        const float pr = m > 0.f ? xr / m : 1.f, pi = m > 0.f ? xi / m : 0.f;
        const float yr = pred[i] * pr, yi = pred[i] * pi;
        y[2 * i] = yr;
        y[2 * i + 1] = yi;
        v[2 * i] = xr - yr;
        v[2 * i + 1] = xi - yi;
    }
}
// spec_utils.mirroring("mirroring", :453-470): mirror = flip(|spec_m[:, lo : lo + hh]|) * exp(i angle(he)); out = |he| <= |mirror| ? he : mirror
__global__ void __launch_bounds__(kNnThreads)
vr_mirror_kernel(const float* __restrict__ spec_m, const float* __restrict__ he, float* __restrict__ out, int64_t n, int bins, int hh, int l,
                 int lo) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int t = (int)(i % l), b = (int)((i / l) % hh), c = (int)(i / ((int64_t)l * hh));
        const int64_t sm = (((int64_t)c * bins + lo + (hh - 1 - b)) * l + t) * 2;
        const float mag = hypotf(spec_m[sm], spec_m[sm + 1]);
        const float hr = he[2 * i], hi = he[2 * i + 1];
        const float hm = hypotf(hr, hi);
        if (hm <= mag) {
            out[2 * i] = hr;
            out[2 * i + 1] = hi;
        } else {                                              // hm > mag >= 0: the phase of he is defined
            out[2 * i] = mag * hr / hm;
            out[2 * i + 1] = mag * hi / hm;
        }
    }
}
// one band's spectrogram for the iSTFT (cmb_spectrogram_to_wave :353-429): band[(2c, 2c+1), f, t] = gain[f] * (bins of spec_m
// [o0, o0 + h) at f in [f0, f0 + h); extra[c, f - e0, t] for f in [e0, e0 + eh) when given -- the later assignment wins; 0 elsewhere)
__global__ void __launch_bounds__(kNnThreads)
vr_band_spec_kernel(const float* __restrict__ spec_m, const float* __restrict__ extra, const float* __restrict__ gain, float* __restrict__ band,
                    int64_t n, int bins, int l, int Fb, int f0, int h, int o0, int e0, int eh) {
    for (int64_t i = (int64_t)blockIdx.x * kNnThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kNnThreads) {
        const int t = (int)(i % l), f = (int)((i / l) % Fb), c = (int)(i / ((int64_t)l * Fb));
        float re = 0.f, im = 0.f;
        if (extra && f >= e0 && f < e0 + eh) {
            const int64_t s = (((int64_t)c * eh + f - e0) * l + t) * 2;
            re = extra[s];
            im = extra[s + 1];
        } else if (f >= f0 && f < f0 + h) {
            const int64_t s = (((int64_t)c * bins + o0 + f - f0) * l + t) * 2;
            re = spec_m[s];
            im = spec_m[s + 1];
        }
        const float g = gain[f];
        band[((int64_t)(2 * c) * Fb + f) * l + t] = g * re;
        band[((int64_t)(2 * c + 1) * Fb + f) * l + t] = g * im;
    }
}

unsigned ew_grid(int64_t n) {
    int64_t b = ceil_div64(n, kNnThreads);
    if (b < 1) b = 1;
    return (unsigned)(b > 65536 ? 65536 : b);
}
}  // namespace

#define NN_ARG(cond, what) \
    if (!(cond)) return alsep_fail(ctx, ALSEP_ERR_ARG, what ": bad argument")

// 1: the generic float32 GEMM / convolution entry points of this context run their contractions as split-half products on the f16 matrix
// pipe (float32 in and out, 2^-22 per product; operands limited to the half range); 0: exact f32 MFMA (the default).  Not inside a capture.
extern "C" int alsep_nn_set_contraction(alsep_ctx* ctx, int split) {
    ALSEP_ENTER(ctx);
    if (!ctx || (split != 0 && split != 1)) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_set_contraction: bad argument");
    if (split && !ctx->nn_range) {
        ALSEP_HIP(ctx, hipMalloc((void**)&ctx->nn_range, 16));
        ALSEP_HIP(ctx, hipMemset(ctx->nn_range, 0, 16));
    }
    ctx->nn_split = split;
    return ALSEP_OK;
}

// *out = 1 when a split-contraction launch of this context met an operand beyond the half range (|x| > 65504, or not a number) since the
// last call: results computed since then are invalid.  Reads and clears the word; synchronises the context's stream.
extern "C" int alsep_nn_range_flag(alsep_ctx* ctx, int32_t* out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !out) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_range_flag: null argument");
    *out = 0;
    if (!ctx->nn_range) return ALSEP_OK;
    unsigned v = 0;
    ALSEP_HIP(ctx, hipMemcpyAsync(&v, ctx->nn_range, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    ALSEP_HIP(ctx, hipMemsetAsync(ctx->nn_range, 0, sizeof(v), ctx->stream));
    ALSEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = v != 0;
    return ALSEP_OK;
}

extern "C" int alsep_nn_bgemm(alsep_ctx* ctx, const float* A, const float* B, float* C, int nb1, int nb2, int M, int N, int K,
                              const int64_t* sa, const int64_t* sb, const int64_t* sc, float alpha) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && A && B && C && sa && sb && sc && nb1 > 0 && nb2 > 0 && M > 0 && N > 0 && K > 0 && (int64_t)nb1 * nb2 <= 65535,
           "alsep_nn_bgemm");
    GemmStrides a{sa[0], sa[1], sa[2], sa[3]}, b{sb[0], sb[1], sb[2], sb[3]}, c{sc[0], sc[1], sc[2], sc[3]};
    return launch_gemm(ctx, A, B, C, nb1 * nb2, nb2, M, N, K, a, b, c, alpha, nullptr, 0);
}

extern "C" int alsep_nn_bgemm_bias(alsep_ctx* ctx, const float* A, const float* B, float* C, int nb1, int nb2, int M, int N, int K,
                                   const int64_t* sa, const int64_t* sb, const int64_t* sc, float alpha, const float* bias, int act) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && A && B && C && sa && sb && sc && nb1 > 0 && nb2 > 0 && M > 0 && N > 0 && K > 0 && (int64_t)nb1 * nb2 <= 65535 &&
               (act == 0 || act == 3 || act == 5),
           "alsep_nn_bgemm_bias");
    GemmStrides a{sa[0], sa[1], sa[2], sa[3]}, b{sb[0], sb[1], sb[2], sb[3]}, c{sc[0], sc[1], sc[2], sc[3]};
    return launch_gemm(ctx, A, B, C, nb1 * nb2, nb2, M, N, K, a, b, c, alpha, bias, act);
}

// rows of at most 64 * kSmxPer elements (the attention matrices: 801 frames, 60 bands, 336 tokens): one WAVE per row, the row in
// registers between the three steps (max, exp + sum, scale) -- one read and one write of the matrix instead of three reads and two writes
// and two LDS reductions per row; the sum in fp64 as in the workgroup-per-row kernel (another order of the same additions)
constexpr int kSmxPer = 16;
__global__ void __launch_bounds__(kNnThreads)
nn_softmax_rows_wave_kernel(float* __restrict__ x, int64_t rows, int n, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (kNnThreads / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;                                   // whole waves leave together
    float* row = x + r * ld;
    float v[kSmxPer];
    float mx = -3.4e38f;
#pragma unroll
    for (int k = 0; k < kSmxPer; ++k) {
        const int i = lane + 64 * k;
        v[k] = i < n ? row[i] : -3.4e38f;
        mx = fmaxf(mx, v[k]);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kSmxPer; ++k) {
        const int i = lane + 64 * k;
        v[k] = i < n ? expf(v[k] - mx) : 0.f;
        s += (double)v[k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float inv = (float)(1.0 / s);
#pragma unroll
    for (int k = 0; k < kSmxPer; ++k) {
        const int i = lane + 64 * k;
        if (i < n) row[i] = v[k] * inv;
    }
}
static void launch_softmax(alsep_ctx* ctx, float* x, int64_t rows, int n, int ld) {
    if (n <= 64 * kSmxPer)
        hipLaunchKernelGGL(nn_softmax_rows_wave_kernel, dim3((unsigned)ceil_div64(rows, kNnThreads / 64)), dim3(kNnThreads), 0, ctx->stream, x, rows,
                           n, ld);
    else
        hipLaunchKernelGGL(nn_softmax_rows_kernel, dim3((unsigned)rows), dim3(kNnThreads), 128, ctx->stream, x, n, ld);
}

extern "C" int alsep_nn_softmax_rows(alsep_ctx* ctx, float* x, int64_t rows, int n) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && rows > 0 && rows <= 0x7fffffff && n > 0, "alsep_nn_softmax_rows");
    launch_softmax(ctx, x, rows, n, n);
    ALSEP_LAUNCH_CHECK(ctx, "nn_softmax_rows_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_softmax_rows_ld(alsep_ctx* ctx, float* x, int64_t rows, int n, int ld) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && rows > 0 && rows <= 0x7fffffff && n > 0 && ld >= n, "alsep_nn_softmax_rows_ld");
    launch_softmax(ctx, x, rows, n, ld);
    ALSEP_LAUNCH_CHECK(ctx, "nn_softmax_rows_kernel");
    return ALSEP_OK;
}

static int stats_nb(int64_t per_group, int64_t G) {
    int64_t nb = ceil_div64(per_group, 32768);
    if (nb < 1) nb = 1;
    const int64_t cap = G >= 256 ? 1 : 512 / G;               // enough workgroups for the chip, few partials per group
    if (nb > cap) nb = cap;
    return (int)(nb < 1 ? 1 : nb);
}
extern "C" int64_t alsep_nn_stats_workspace_bytes(int64_t G, int64_t per_group) {
    if (G <= 0 || per_group <= 0) return -1;
    return (int64_t)sizeof(double) * 2 * G * stats_nb(per_group, G) + (int64_t)sizeof(float) * 2 * G + 64;
}

static int run_stats(alsep_ctx* ctx, const float* x, int64_t G, int64_t per_group, float eps, int mode, void* workspace,
                     float** stats_out) {
    const int nb = stats_nb(per_group, G);
    double* part = reinterpret_cast<double*>(workspace);
    float* stats = reinterpret_cast<float*>(part + 2 * G * nb);
    if (per_group % 4 == 0 && !((uintptr_t)x & 15))
        hipLaunchKernelGGL(nn_stats_partial4_kernel, dim3((unsigned)nb, (unsigned)G), dim3(kNnThreads), 64, ctx->stream, x, per_group, nb, part);
    else
        hipLaunchKernelGGL(nn_stats_partial_kernel, dim3((unsigned)nb, (unsigned)G), dim3(kNnThreads), 64, ctx->stream, x, per_group, nb, part);
    hipLaunchKernelGGL(nn_stats_final_kernel, dim3((unsigned)ceil_div64(G, 64)), dim3(64), 0, ctx->stream, (const double*)part, nb,
                       per_group, (int)G, eps, mode, stats);
    *stats_out = stats;
    ALSEP_LAUNCH_CHECK(ctx, "nn_stats kernels");
    return ALSEP_OK;
}

extern "C" int alsep_nn_norm(alsep_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta, int64_t G, int64_t R,
                             int C, float eps, int act, void* workspace) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && workspace && G > 0 && G <= 65535 && R > 0 && C > 0 && (act == 0 || act == 3 || (act == 4 && C % 2 == 0)) &&
               ((gamma == nullptr) == (beta == nullptr)) && ((uintptr_t)workspace & 7) == 0,
           "alsep_nn_norm");
    float* stats = nullptr;
    int rc = run_stats(ctx, x, G, R * C, eps, 0, workspace, &stats);
    if (rc) return rc;
    const int64_t n_out = G * R * (act == 4 ? C / 2 : C);
    const int Co = act == 4 ? C / 2 : C;
    if (C % 4 == 0 && Co % 4 == 0 && n_out / 4 < ((int64_t)1 << 31) && R < ((int64_t)1 << 31) &&
        !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15))
        hipLaunchKernelGGL(nn_norm_apply4_kernel, dim3(ew_grid(n_out / 4)), dim3(kNnThreads), 0, ctx->stream, x, y, gamma, beta, (const float*)stats,
                           (unsigned)(n_out / 4), (unsigned)R, C, act);
    else
        hipLaunchKernelGGL(nn_norm_apply_kernel, dim3(ew_grid(n_out)), dim3(kNnThreads), 0, ctx->stream, x, y, gamma, beta,
                           (const float*)stats, n_out, R, C, act);
    ALSEP_LAUNCH_CHECK(ctx, "nn_norm_apply_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_meanstd(alsep_ctx* ctx, const float* x, int64_t nsamples, int64_t per_sample, float* stats, void* workspace) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && stats && workspace && nsamples > 0 && nsamples <= 65535 && per_sample > 0 && ((uintptr_t)workspace & 7) == 0,
           "alsep_nn_meanstd");
    float* st = nullptr;
    int rc = run_stats(ctx, x, nsamples, per_sample, 0.f, 1, workspace, &st);
    if (rc) return rc;
    ALSEP_HIP(ctx, hipMemcpyAsync(stats, st, sizeof(float) * 2 * nsamples, hipMemcpyDeviceToDevice, ctx->stream));
    return ALSEP_OK;
}

extern "C" int alsep_nn_affine_stats(alsep_ctx* ctx, const float* x, float* y, const float* stats, int64_t nsamples,
                                     int64_t per_sample, float eps, int inverse) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && stats && nsamples > 0 && per_sample > 0, "alsep_nn_affine_stats");
    const int64_t n = nsamples * per_sample;
    hipLaunchKernelGGL(nn_affine_stats_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, x, y, stats, n, per_sample, eps,
                       inverse);
    ALSEP_LAUNCH_CHECK(ctx, "nn_affine_stats_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_act(alsep_ctx* ctx, const float* x, float* y, int64_t rows, int C, int act) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && rows > 0 && C > 0 && (act == 1 || act == 3 || act == 5 || (act == 4 && C % 2 == 0)), "alsep_nn_act");
    const int64_t n_out = rows * (act == 4 ? C / 2 : C);
    if (C % 4 == 0 && (act == 4 ? C / 2 : C) % 4 == 0 && n_out / 4 < ((int64_t)1 << 31) && !(((uintptr_t)x | (uintptr_t)y) & 15))
        hipLaunchKernelGGL(nn_act4_kernel, dim3(ew_grid(n_out / 4)), dim3(kNnThreads), 0, ctx->stream, x, y, (unsigned)(n_out / 4), C, act);
    else
        hipLaunchKernelGGL(nn_act_kernel, dim3(ew_grid(n_out)), dim3(kNnThreads), 0, ctx->stream, x, y, n_out, C, act);
    ALSEP_LAUNCH_CHECK(ctx, "nn_act_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_scale_add(alsep_ctx* ctx, const float* a, const float* b, const float* scale, float* y, int64_t rows, int C) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && a && b && y && rows > 0 && C > 0, "alsep_nn_scale_add");
    if (C % 4 == 0 && rows * C / 4 < ((int64_t)1 << 31) && !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)y | (uintptr_t)scale) & 15))
        hipLaunchKernelGGL(nn_scale_add4_kernel, dim3(ew_grid(rows * C / 4)), dim3(kNnThreads), 0, ctx->stream, a, b, scale, y, (unsigned)(rows * C / 4),
                           (unsigned)(C / 4));
    else
        hipLaunchKernelGGL(nn_scale_add_kernel, dim3(ew_grid(rows * C)), dim3(kNnThreads), 0, ctx->stream, a, b, scale, y, rows * C, C);
    ALSEP_LAUNCH_CHECK(ctx, "nn_scale_add_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_add_bcast(alsep_ctx* ctx, float* y, const float* e, float s, int64_t n, int64_t inner, int period, int C) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && y && e && n > 0 && inner > 0 && period > 0 && C > 0, "alsep_nn_add_bcast");
    hipLaunchKernelGGL(nn_add_bcast_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, y, e, s, n, inner, period, C);
    ALSEP_LAUNCH_CHECK(ctx, "nn_add_bcast_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_vec_fma(alsep_ctx* ctx, float* y, const float* x, const float* w, int64_t rows, int64_t n, int64_t y_stride,
                                int64_t x_stride) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && y && x && w && rows > 0 && n > 0 && y_stride >= n && x_stride >= n, "alsep_nn_vec_fma");
    hipLaunchKernelGGL(nn_vec_fma_kernel, dim3(ew_grid(rows * n)), dim3(kNnThreads), 0, ctx->stream, y, x, w, rows, n, y_stride,
                       x_stride);
    ALSEP_LAUNCH_CHECK(ctx, "nn_vec_fma_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_vec_div(alsep_ctx* ctx, float* y, const float* w, int64_t rows, int64_t n) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && y && w && rows > 0 && n > 0, "alsep_nn_vec_div");
    hipLaunchKernelGGL(nn_vec_div_kernel, dim3(ew_grid(rows * n)), dim3(kNnThreads), 0, ctx->stream, y, w, rows, n);
    ALSEP_LAUNCH_CHECK(ctx, "nn_vec_div_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_reflect_pad(alsep_ctx* ctx, const float* x, float* y, int64_t rows, int64_t n, int64_t left, int64_t right) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && rows > 0 && n > 1 && left >= 0 && right >= 0 && left < n && right < n, "alsep_nn_reflect_pad");
    hipLaunchKernelGGL(nn_reflect_pad_kernel, dim3(ew_grid(rows * (n + left + right))), dim3(kNnThreads), 0, ctx->stream, x, y, rows, n,
                       left, right);
    ALSEP_LAUNCH_CHECK(ctx, "nn_reflect_pad_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_tconv_fold(alsep_ctx* ctx, const float* g, const float* bias, float* y, int64_t B, int I, int J, int Cout,
                                   int S, int pad, int Lout, int act) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && g && y && B > 0 && I > 0 && J > 0 && Cout > 0 && S > 0 && pad >= 0 && Lout > 0 && Lout + pad <= (I + 1) * S &&
               (act == 0 || act == 3),
           "alsep_nn_tconv_fold");
    const int64_t n = B * Lout * J * Cout;
    hipLaunchKernelGGL(nn_tconv_fold_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, g, bias, y, n, I, J, Cout, S, pad,
                       Lout, act);
    ALSEP_LAUNCH_CHECK(ctx, "nn_tconv_fold_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_demucs_spec_in(alsep_ctx* ctx, const float* spec, float* y, int64_t B, int F, int Tt, int T, int t_off,
                                    float scale) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && spec && y && B > 0 && F > 0 && T > 0 && t_off >= 0 && t_off + T <= Tt, "alsep_demucs_spec_in");
    const int64_t n = B * F * T * 4;
    hipLaunchKernelGGL(demucs_spec_in_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, spec, y, n, F, Tt, T, t_off, scale);
    ALSEP_LAUNCH_CHECK(ctx, "demucs_spec_in_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_demucs_spec_out(alsep_ctx* ctx, const float* x, const float* stats, float* spec, int64_t B, int S, int F, int Tt,
                                     int T, int t_off, float scale) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && stats && spec && B > 0 && S > 0 && F > 0 && T > 0 && t_off >= 0 && t_off + T <= Tt, "alsep_demucs_spec_out");
    const int64_t n = B * S * 4 * F * Tt;
    hipLaunchKernelGGL(demucs_spec_out_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, x, stats, spec, n, S, F, Tt, T, t_off,
                       scale);
    ALSEP_LAUNCH_CHECK(ctx, "demucs_spec_out_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_demucs_mix_out(alsep_ctx* ctx, const float* xt, const float* statst, const float* xs, float* out, int64_t B, int S,
                                    int64_t L) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && xt && statst && xs && out && B > 0 && S > 0 && L > 0, "alsep_demucs_mix_out");
    const int64_t n = B * S * 2 * L;
    hipLaunchKernelGGL(demucs_mix_out_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, xt, statst, xs, out, n, S, L);
    ALSEP_LAUNCH_CHECK(ctx, "demucs_mix_out_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_swap_last2(alsep_ctx* ctx, const float* x, float* y, int64_t B, int64_t C, int64_t L) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && B > 0 && C > 0 && L > 0, "alsep_nn_swap_last2");
    hipLaunchKernelGGL(nn_swap_last2_kernel, dim3(ew_grid(B * C * L)), dim3(kNnThreads), 0, ctx->stream, x, y, B * C * L, C, L);
    ALSEP_LAUNCH_CHECK(ctx, "nn_swap_last2_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_rmsnorm(alsep_ctx* ctx, const float* x, float* y, const float* gamma, int64_t rows, int C, int64_t x_stride,
                                int64_t y_stride) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && gamma && rows > 0 && C > 0 && x_stride >= C && y_stride >= C, "alsep_nn_rmsnorm");
    hipLaunchKernelGGL(nn_rmsnorm_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(kNnThreads), 0, ctx->stream, x, y, gamma, rows, C, x_stride,
                       y_stride);
    ALSEP_LAUNCH_CHECK(ctx, "nn_rmsnorm_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_rotary(alsep_ctx* ctx, float* x, int64_t rows, int64_t row_stride, int col_off, int heads, int d, int64_t pos_div,
                               int64_t pos_mod) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && rows > 0 && heads > 0 && d > 0 && d % 2 == 0 && col_off >= 0 && col_off + heads * d <= row_stride && pos_div > 0 &&
               pos_mod > 0,
           "alsep_nn_rotary");
    const int64_t n = rows * heads * (d / 2);
    hipLaunchKernelGGL(nn_rotary_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, x, n, row_stride, col_off, heads, d, pos_div,
                       pos_mod);
    ALSEP_LAUNCH_CHECK(ctx, "nn_rotary_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_gate(alsep_ctx* ctx, float* out, const float* gates, int64_t rows, int heads, int d) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && out && gates && rows > 0 && heads > 0 && d > 0, "alsep_nn_gate");
    hipLaunchKernelGGL(nn_gate_kernel, dim3(ew_grid(rows * heads * d)), dim3(kNnThreads), 0, ctx->stream, out, gates, rows * heads * d, heads, d);
    ALSEP_LAUNCH_CHECK(ctx, "nn_gate_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_roformer_gather(alsep_ctx* ctx, const float* spec, const int* midx, float* feat, int n_idx, int F, int T) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && spec && midx && feat && n_idx > 0 && F > 0 && T > 0, "alsep_roformer_gather");
    const int64_t n = (int64_t)T * n_idx * 2;
    hipLaunchKernelGGL(roformer_gather_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, spec, midx, feat, n, n_idx, F, T);
    ALSEP_LAUNCH_CHECK(ctx, "roformer_gather_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_roformer_mask(alsep_ctx* ctx, const float* spec, const float* h, const int* occ_start, const int* col_a, const int* col_g,
                                   float* out, int F, int T, int H) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && spec && h && occ_start && col_a && col_g && out && F > 0 && T > 0 && H > 0, "alsep_roformer_mask");
    const int64_t n = 2 * (int64_t)F * T;
    hipLaunchKernelGGL(roformer_mask_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, spec, h, occ_start, col_a, col_g, out, n, F, T,
                       H);
    ALSEP_LAUNCH_CHECK(ctx, "roformer_mask_kernel");
    return ALSEP_OK;
}

static int64_t instnorm_slabs(int64_t P, int C) {
    if (C % 4 == 0 && C <= 4 * kNnThreads) {                 // nn_chan_stats4_kernel: slabs of >= 64 pixels, at most 1024 of them
        const int64_t nb = ceil_div64(P, 64);
        return nb > 1024 ? 1024 : nb;
    }
    const int64_t nb = ceil_div64(P, 512);
    return nb > 256 ? 256 : nb;
}
extern "C" int64_t alsep_nn_instnorm_workspace_bytes(int64_t P, int C) {
    if (P <= 0 || C <= 0) return -1;
    return (int64_t)sizeof(double) * 2 * C * instnorm_slabs(P, C) + (int64_t)sizeof(float) * 2 * C + 64;
}
// statistics of x [P, C] per channel -> stats [C][2] = (mean, 1 / sqrt(var + eps)) behind the partials in the workspace
static float* instnorm_stats(alsep_ctx* ctx, const float* x, int64_t P, int C, float eps, void* workspace) {
    const int64_t nb = instnorm_slabs(P, C);
    double* part = reinterpret_cast<double*>(workspace);
    float* stats = reinterpret_cast<float*>(part + 2 * (int64_t)C * nb);
    if (C % 4 == 0 && C <= 4 * kNnThreads && ((uintptr_t)x & 15) == 0) {
        hipLaunchKernelGGL(nn_chan_stats4_kernel, dim3((unsigned)nb), dim3(kNnThreads), 8 * kNnThreads * sizeof(double), ctx->stream, x, P, C, (int)nb,
                           part);
        hipLaunchKernelGGL(nn_chan_stats_final4_kernel, dim3((unsigned)ceil_div64(C, 16)), dim3(kNnThreads), 0, ctx->stream, (const double*)part,
                           (int)nb, P, C, eps, stats);
        return stats;
    }
    if (C < kNnThreads && kNnThreads % C == 0)
        hipLaunchKernelGGL(nn_chan_stats_partial_small_kernel, dim3(1, (unsigned)nb), dim3(kNnThreads), 2 * kNnThreads * sizeof(double), ctx->stream,
                           x, P, C, (int)nb, part);
    else
        hipLaunchKernelGGL(nn_chan_stats_partial_kernel, dim3((unsigned)ceil_div64(C, kNnThreads), (unsigned)nb), dim3(kNnThreads), 0, ctx->stream, x,
                           P, C, (int)nb, part);
    hipLaunchKernelGGL(nn_chan_stats_final_kernel, dim3((unsigned)ceil_div64(C, 64)), dim3(64), 0, ctx->stream, (const double*)part, (int)nb, P, C,
                       eps, stats);
    return stats;
}
extern "C" int alsep_nn_instnorm(alsep_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta, int64_t P, int C, float eps,
                                 int act, void* workspace) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && workspace && P > 0 && C > 0 && (act == 0 || act == 3) && ((gamma == nullptr) == (beta == nullptr)) &&
               ((uintptr_t)workspace & 7) == 0,
           "alsep_nn_instnorm");
    const float* stats = instnorm_stats(ctx, x, P, C, eps, workspace);
    if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0)
        hipLaunchKernelGGL(nn_chan_norm_apply4_kernel, dim3(ew_grid(P * C / 4)), dim3(kNnThreads), 0, ctx->stream, x, y, gamma, beta, stats, P * C / 4, C,
                           act);
    else
        hipLaunchKernelGGL(nn_chan_norm_apply_kernel, dim3(ew_grid(P * C)), dim3(kNnThreads), 0, ctx->stream, x, y, gamma, beta, stats, P * C, C, act);
    ALSEP_LAUNCH_CHECK(ctx, "nn_instnorm kernels");
    return ALSEP_OK;
}
// the same with y stored as IEEE half (C % 4 == 0, 16-byte aligned x, 8-byte aligned y): the input of alsep_nn_conv2d_f16
extern "C" int alsep_nn_instnorm_f16(alsep_ctx* ctx, const float* x, void* y, const float* gamma, const float* beta, int64_t P, int C, float eps,
                                     int act, void* workspace) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && workspace && P > 0 && C > 0 && C % 4 == 0 && (act == 0 || act == 3) && ((gamma == nullptr) == (beta == nullptr)) &&
               ((uintptr_t)workspace & 7) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0,
           "alsep_nn_instnorm_f16");
    const float* stats = instnorm_stats(ctx, x, P, C, eps, workspace);
    hipLaunchKernelGGL(nn_chan_norm_apply_h_kernel, dim3(ew_grid(P * C / 4)), dim3(kNnThreads), 0, ctx->stream, x, (_Float16*)y, gamma, beta, stats,
                       P * C / 4, C, act);
    ALSEP_LAUNCH_CHECK(ctx, "nn_instnorm_f16 kernels");
    return ALSEP_OK;
}

// InstanceNorm2d (+ GELU) of x [T][F][C] with the IEEE-half result stored as [T][C][F] (see nn_chan_norm_apply_ht_kernel)
extern "C" int alsep_nn_instnorm_f16_t(alsep_ctx* ctx, const float* x, void* y, const float* gamma, const float* beta, int64_t T, int F, int C,
                                       float eps, int act, void* workspace) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && x && y && workspace && T > 0 && T < 65536 && F > 0 && C > 0 && (act == 0 || act == 3) && ((gamma == nullptr) == (beta == nullptr)) &&
               ((uintptr_t)workspace & 7) == 0,
           "alsep_nn_instnorm_f16_t");
    const float* stats = instnorm_stats(ctx, x, T * F, C, eps, workspace);
    hipLaunchKernelGGL(nn_chan_norm_apply_ht_kernel, dim3((unsigned)ceil_div64(F, 32), (unsigned)ceil_div64(C, 32), (unsigned)T), dim3(kNnThreads),
                       32 * 33 * sizeof(float), ctx->stream, x, (_Float16*)y, gamma, beta, stats, F, C, act);
    ALSEP_LAUNCH_CHECK(ctx, "nn_instnorm_f16_t kernels");
    return ALSEP_OK;
}

extern "C" int alsep_nn_mul(alsep_ctx* ctx, const float* a, const float* b, float* y, int64_t n) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && a && b && y && n > 0, "alsep_nn_mul");
    hipLaunchKernelGGL(nn_mul_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, a, b, y, n);
    ALSEP_LAUNCH_CHECK(ctx, "nn_mul_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_depth_to_space2(alsep_ctx* ctx, const float* g, float* y, int H, int W, int Cout, int y_ctotal, int y_coff) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && g && y && H > 0 && W > 0 && Cout > 0 && y_coff >= 0 && y_coff + Cout <= y_ctotal, "alsep_nn_depth_to_space2");
    const int64_t n = 4 * (int64_t)H * W * Cout;
    hipLaunchKernelGGL(nn_depth_to_space2_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, g, y, n, H, W, Cout, y_ctotal, y_coff);
    ALSEP_LAUNCH_CHECK(ctx, "nn_depth_to_space2_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_mdx23c_spec_in(alsep_ctx* ctx, const float* spec, float* x, int f, int k, int T) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && spec && x && f > 0 && k > 0 && T > 0, "alsep_mdx23c_spec_in");
    const int64_t n = (int64_t)T * f * 4 * k;
    hipLaunchKernelGGL(mdx23c_spec_in_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, spec, x, n, f, k, T);
    ALSEP_LAUNCH_CHECK(ctx, "mdx23c_spec_in_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_mdx23c_spec_out(alsep_ctx* ctx, const float* y, float* spec, int S, int f, int k, int T) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && y && spec && S > 0 && f > 0 && k > 0 && T > 0, "alsep_mdx23c_spec_out");
    const int64_t n = (int64_t)S * 4 * f * k * T;
    hipLaunchKernelGGL(mdx23c_spec_out_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, y, spec, n, S, f, k, T);
    ALSEP_LAUNCH_CHECK(ctx, "mdx23c_spec_out_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_band_crop(alsep_ctx* ctx, const float* band, float* out, const float* gain, int Fb, int Tb, int f0, int t0, int h, int l,
                                  int out_bins, int o0) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && band && out && Fb > 0 && Tb > 0 && f0 >= 0 && t0 >= 0 && h > 0 && l > 0 && f0 + h <= Fb && t0 + l <= Tb && o0 >= 0 &&
               o0 + h <= out_bins,
           "alsep_vr_band_crop");
    const int64_t n = 2 * (int64_t)h * l;
    hipLaunchKernelGGL(vr_band_crop_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, band, out, gain, n, Fb, Tb, f0, t0, h, l, out_bins,
                       o0);
    ALSEP_LAUNCH_CHECK(ctx, "vr_band_crop_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_split_pred(alsep_ctx* ctx, const float* pred, const float* X, float* y, float* v, int64_t n) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && pred && X && y && v && n > 0, "alsep_vr_split_pred");
    hipLaunchKernelGGL(vr_split_pred_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, pred, X, y, v, n);
    ALSEP_LAUNCH_CHECK(ctx, "vr_split_pred_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_mirror(alsep_ctx* ctx, const float* spec_m, const float* he, float* out, int bins, int hh, int l, int lo) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && spec_m && he && out && bins > 0 && hh > 0 && l > 0 && lo >= 0 && lo + hh <= bins, "alsep_vr_mirror");
    const int64_t n = 2 * (int64_t)hh * l;
    hipLaunchKernelGGL(vr_mirror_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, spec_m, he, out, n, bins, hh, l, lo);
    ALSEP_LAUNCH_CHECK(ctx, "vr_mirror_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_band_spec(alsep_ctx* ctx, const float* spec_m, const float* extra, const float* gain, float* band, int bins, int l,
                                  int Fb, int f0, int h, int o0, int e0, int eh) {
    ALSEP_ENTER(ctx);
    NN_ARG(ctx && spec_m && gain && band && bins > 0 && l > 0 && Fb > 0 && f0 >= 0 && h > 0 && f0 + h <= Fb && o0 >= 0 && o0 + h <= bins &&
               (!extra || (e0 >= 0 && eh > 0 && e0 + eh <= Fb)),
           "alsep_vr_band_spec");
    const int64_t n = 2 * (int64_t)Fb * l;
    hipLaunchKernelGGL(vr_band_spec_kernel, dim3(ew_grid(n)), dim3(kNnThreads), 0, ctx->stream, spec_m, extra, gain, band, n, bins, l, Fb, f0, h,
                       o0, e0, eh);
    ALSEP_LAUNCH_CHECK(ctx, "vr_band_spec_kernel");
    return ALSEP_OK;
}
