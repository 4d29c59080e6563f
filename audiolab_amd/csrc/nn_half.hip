// Half-precision MFMA kernels of the transformer model families (Mel-Band / BS Roformer; the reference runs them under torch autocast,
// modules/separator/stem_separator.py:106 ``use_autocast=True``: Linear layers and attention in IEEE half, everything else in float32).
// gfx950 only.  Activations stay float32 in HBM; the GEMM operands are rounded to f16 on their way into LDS (weights once, at load),
// products on v_mfma_f32_16x16x32_f16 with float32 accumulation, results stored as float32 -- i.e. autocast's rounding points for the
// inputs of a Linear, none for its output.
//
//   nn_gemm_h_kernel   C[M][N] = act(alpha A[M][K] W[N][K]^T + bias) (+ residual), A float32, W f16.
//   nn_attn_h_kernel   softmax(Q K^T) V for the packed, rotary-embedded q | k | v projection of a Roformer layer, one pass (no score
//                      matrix in HBM): online softmax in registers, S^T = K Q^T so that P feeds the second MFMA as it lies.
#include "alsep_common.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

constexpr int kHThreads = 256;

__device__ __forceinline__ float gelu_erf_h(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

__device__ __forceinline__ h16x8 to_h8(const f32x4& a, const f32x4& b) {
    h16x8 r;
    r[0] = (_Float16)a[0]; r[1] = (_Float16)a[1]; r[2] = (_Float16)a[2]; r[3] = (_Float16)a[3];
    r[4] = (_Float16)b[0]; r[5] = (_Float16)b[1]; r[6] = (_Float16)b[2]; r[7] = (_Float16)b[3];
    return r;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// GEMM.  Workgroup tile 128 (M) x 128 (N), 4 waves as 2 x 2 (64 x 64 each: 16 accumulator blocks), K in slices of 32 = one MFMA step.
// LDS image of a slice: rows of 32 halves (64 bytes = four 16-byte k-groups), group g of row r stored at g ^ ((r >> 1) & 3): the
// ds_read_b128 of 16 consecutive rows x one group is conflict-free, as is the 16-byte staging store.  Two slices are resident (the next
// one's global loads are in flight during the 16 MFMAs of the current one, converted and stored after them; one barrier per slice).
// Operand roles are swapped (D rows = n, D columns = m): a lane ends up with four consecutive n of one m -- one float4 store.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int kHgBM = 128, kHgBN = 128, kHgBK = 32;
constexpr size_t kHgLds = 2 * (size_t)(kHgBM + kHgBN) * kHgBK * sizeof(_Float16);      // 32 KiB

struct GemmHArgs {
    const float* A; int64_t lda, sa_b;
    const _Float16* B; int64_t ldb, sb_b;
    float* C; int64_t ldc, sc_b;
    const float* bias; int64_t bias_b;
    const float* R; int64_t ldr, sr_b;            // optional residual added after the activation: C = act(...) + R
    int M, N, K;
    float alpha;
    int act;
    const int* nvec;                              // optional: columns of batch b (<= N): the ragged last layers of the mask estimators
};

__device__ __forceinline__ int hg_slot(int row, int g) { return row * kHgBK + 8 * (g ^ ((row >> 1) & 3)); }

__global__ void __launch_bounds__(kHThreads, 2)
nn_gemm_h_kernel(GemmHArgs p) {
    _Float16* As = reinterpret_cast<_Float16*>(alsep_smem);                  // [2][128][32]
    _Float16* Bs = As + 2 * kHgBM * kHgBK;                                    // [2][128][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int bz = blockIdx.z;
    const float* a = p.A + bz * p.sa_b;
    const _Float16* b = p.B + bz * p.sb_b;
    float* c = p.C + bz * p.sc_b;
    const int m0 = blockIdx.y * kHgBM, n0 = blockIdx.x * kHgBN;
    const int Nb = p.nvec ? p.nvec[bz] : p.N;
    if (n0 >= Nb) return;                                                    // whole workgroups leave together
    // staging duty per slice: A 512 granules of 8 floats (two per thread), B 512 granules of 8 halves (two per thread)
    const int sr = tid >> 2, sg = tid & 3;                                   // rows sr and sr + 64, k-group sg
    const float* ga[2];
    const _Float16* gb[2];
    bool va[2], vb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ra = m0 + sr + 64 * h, rb = n0 + sr + 64 * h;
        va[h] = ra < p.M;
        vb[h] = rb < Nb;
        ga[h] = a + (int64_t)(va[h] ? ra : 0) * p.lda + 8 * sg;
        gb[h] = b + (int64_t)(vb[h] ? rb : 0) * p.ldb + 8 * sg;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra0[2], ra1[2];
    h16x8 rbv[2];
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 zh;
#pragma unroll
    for (int e = 0; e < 8; ++e) zh[e] = (_Float16)0.f;
    auto gload = [&](int k0) {                                               // K % 8 == 0: a granule is inside K or outside it
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool in = k0 + 8 * sg < p.K;
            ra0[h] = (va[h] && in) ? *reinterpret_cast<const f32x4*>(ga[h] + k0) : z4;
            ra1[h] = (va[h] && in) ? *reinterpret_cast<const f32x4*>(ga[h] + k0 + 4) : z4;
            rbv[h] = (vb[h] && in) ? *reinterpret_cast<const h16x8*>(gb[h] + k0) : zh;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = sr + 64 * h;
            *reinterpret_cast<h16x8*>(As + (size_t)buf * kHgBM * kHgBK + hg_slot(row, sg)) = to_h8(ra0[h], ra1[h]);
            *reinterpret_cast<h16x8*>(Bs + (size_t)buf * kHgBN * kHgBK + hg_slot(row, sg)) = rbv[h];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = (p.K + kHgBK - 1) / kHgBK;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * kHgBK);
        h16x8 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = *reinterpret_cast<const h16x8*>(As + (size_t)buf * kHgBM * kHgBK + hg_slot(wm * 64 + i * 16 + l15, lq));
            bf[i] = *reinterpret_cast<const h16x8*>(Bs + (size_t)buf * kHgBN * kHgBK + hg_slot(wn * 64 + i * 16 + l15, lq));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // epilogue: D rows = n (4 lq + r), D columns = m (l15): C[m][n .. n + 3] is one float4 (N % 4 == 0: all four inside or none)
    const float* bias = p.bias ? p.bias + bz * p.bias_b : nullptr;
    const float* res = p.R ? p.R + bz * p.sr_b : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 64 + i * 16 + l15;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + 4 * lq;
            if (col >= Nb) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = p.alpha * acc[i][j][r];
                if (bias) t += bias[col + r];
                if (p.act == 3) t = gelu_erf_h(t);
                else if (p.act == 5) t = tanhf(t);
                v[r] = t;
            }
            if (res) {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(res + (int64_t)row * p.ldr + col);
                v += rv;
            }
            *reinterpret_cast<f32x4*>(c + (int64_t)row * p.ldc + col) = v;
        }
    }
}

__global__ void __launch_bounds__(kHThreads)
to_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kHThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kHThreads) y[i] = (_Float16)x[i];
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Attention.  qkv: float32 rows of 3 * heads * 64 values (q | k | v, head-major inside each), rotary embedding already applied; a
// sequence is L rows `row_stride` floats apart starting at seq * seq_stride.  out: float32 rows of heads * 64, same row order through
// (o_seq_stride, o_row_stride).  head dimension 64.
//
// One workgroup = 64 queries of one (sequence, head), four waves of 16 queries; keys / values in chunks of 32 staged through LDS for
// all four waves (K rows of 64 halves, 16-byte groups XOR-swizzled by row & 7: conflict-free ds_read_b128; V TRANSPOSED, rows of 32
// keys padded to 36 halves: conflict-free ds_read_b64).  Per chunk and wave:
//     S^T[key][query] = K Q^T            2 key blocks x 2 MFMA steps (d = 64); Q fragments live in registers (scaled by d^-1/2)
//     online softmax per query           a query's 32 scores sit in the 4 lanes {l15, l15 + 16, + 32, + 48}: two shuffles for the max
//     O^T[d][query] += V^T P^T           4 d blocks x 1 MFMA step; P^T is the B operand AS THE LANE HOLDS IT (contraction index e of lane
//                                        quarter lq = key 16 (e / 4) + 4 lq + e % 4, V^T read in the same order), O rescaled per lane
// The row sums are carried per lane and reduced once at the end.  float32 everywhere outside the two MFMA operands.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int kAtD = 64, kAtKc = 32, kAtVld = 36;

// rotary embedding of 8 consecutive head-dimension values (4 interleaved pairs) by the table entries (cos, sin) of their position:
// (a, b) -> (a cos - b sin, b cos + a sin), the arithmetic of nn_rotary_kernel
__device__ __forceinline__ void rotate8(f32x4& x0, f32x4& x1, const float* __restrict__ cs) {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs), c1 = *reinterpret_cast<const f32x4*>(cs + 4);     // cos0 sin0 cos1 sin1 | cos2 sin2 cos3 sin3
    f32x4 r0, r1;
    r0[0] = x0[0] * c0[0] - x0[1] * c0[1]; r0[1] = x0[1] * c0[0] + x0[0] * c0[1];
    r0[2] = x0[2] * c0[2] - x0[3] * c0[3]; r0[3] = x0[3] * c0[2] + x0[2] * c0[3];
    r1[0] = x1[0] * c1[0] - x1[1] * c1[1]; r1[1] = x1[1] * c1[0] + x1[0] * c1[1];
    r1[2] = x1[2] * c1[2] - x1[3] * c1[3]; r1[3] = x1[3] * c1[2] + x1[2] * c1[3];
    x0 = r0;
    x1 = r1;
}

// table[pos][j] = (cos, sin)(pos / 10000^(2 j / d)), j < d / 2: the angles of nn_rotary_kernel
__global__ void __launch_bounds__(256)
rotary_table_kernel(float* __restrict__ table, int L, int d) {
    const int half = d / 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L * half; i += gridDim.x * 256) {
        const int j = i % half;
        const float pos = (float)(i / half);
        const float inv = 1.f / powf(10000.f, (float)(2 * j) / (float)d);
        const float ang = pos * inv;
        table[2 * i] = cosf(ang);
        table[2 * i + 1] = sinf(ang);
    }
}
constexpr size_t kAtLds = (size_t)kAtKc * kAtD * sizeof(_Float16) + (size_t)kAtD * kAtVld * sizeof(_Float16);     // 4096 + 4608

__global__ void __launch_bounds__(kHThreads)
nn_attn_h_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int heads, int64_t seq_stride, int64_t row_stride,
                 int64_t o_seq_stride, int64_t o_row_stride, float scale, const float* __restrict__ rot, const float* __restrict__ gates,
                 int64_t g_seq_stride, int64_t g_row_stride) {
    _Float16* Ks = reinterpret_cast<_Float16*>(alsep_smem);                  // [32 keys][64 d], swizzled groups
    _Float16* Vt = Ks + kAtKc * kAtD;                                         // [64 d][36]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int head = blockIdx.y, seq = blockIdx.z;
    const int inner = heads * kAtD;
    const float* base = qkv + seq * seq_stride + head * kAtD;
    const float* kbase = base + inner;
    const float* vbase = base + 2 * inner;
    const int q = blockIdx.x * 64 + wave * 16 + l15;                          // this lane's query (the MFMA column)
    const bool qok = q < L;
    // Q fragments: B operand of S^T = K Q^T: lane (col = query l15, quarter lq) holds Q[q][32 s + 8 lq .. + 7]
    h16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
        if (qok) {
            const float* src = base + (int64_t)q * row_stride + 32 * s + 8 * lq;
            a = *reinterpret_cast<const f32x4*>(src);
            b = *reinterpret_cast<const f32x4*>(src + 4);
            if (rot) rotate8(a, b, rot + ((int64_t)q * (kAtD / 2) + 16 * s + 4 * lq) * 2);
        }
        qf[s] = to_h8(a * scale, b * scale);
    }
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun = -3.0e38f, lsum = 0.f;                                        // running max (shared by the query's 4 lanes), this lane's partial sum
    const int skey = tid >> 3, sgrp = tid & 7;                                // staging duty: key skey, 8 d values from 8 sgrp
    for (int k0 = 0; k0 < L; k0 += kAtKc) {
        f32x4 ka = f32x4{0.f, 0.f, 0.f, 0.f}, kb = ka, va = ka, vb = ka;
        if (k0 + skey < L) {
            const int64_t off = (int64_t)(k0 + skey) * row_stride + 8 * sgrp;
            ka = *reinterpret_cast<const f32x4*>(kbase + off);
            kb = *reinterpret_cast<const f32x4*>(kbase + off + 4);
            if (rot) rotate8(ka, kb, rot + ((int64_t)(k0 + skey) * (kAtD / 2) + 4 * sgrp) * 2);
            va = *reinterpret_cast<const f32x4*>(vbase + off);
            vb = *reinterpret_cast<const f32x4*>(vbase + off + 4);
        }
        __syncthreads();                                                      // every wave is done with the previous chunk
        *reinterpret_cast<h16x8*>(Ks + skey * kAtD + 8 * (sgrp ^ (skey & 7))) = to_h8(ka, kb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Vt[(8 * sgrp + e) * kAtVld + skey] = (_Float16)va[e];
            Vt[(8 * sgrp + 4 + e) * kAtVld + skey] = (_Float16)vb[e];
        }
        __syncthreads();
        // S^T blocks: keys 16 kb + (4 lq + r), query l15
        f32x4 s[2];
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
            s[kb2] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int krow = 16 * kb2 + l15;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const h16x8 kf = *reinterpret_cast<const h16x8*>(Ks + krow * kAtD + 8 * ((4 * st + lq) ^ (krow & 7)));
                s[kb2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[st], s[kb2], 0, 0, 0);
            }
        }
        float cmax = -3.0e38f;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k0 + 16 * kb2 + 4 * lq + r >= L) s[kb2][r] = -3.0e38f;   // keys beyond the sequence
                cmax = fmaxf(cmax, s[kb2][r]);
            }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float mnew = fmaxf(mrun, cmax);
        const float corr = __expf(mrun - mnew);
        mrun = mnew;
        h16x8 pf;
        float psum = 0.f;
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = s[kb2][r] > -1.0e38f ? __expf(s[kb2][r] - mnew) : 0.f;
                psum += pv;
                pf[4 * kb2 + r] = (_Float16)pv;
            }
        lsum = lsum * corr + psum;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            o[d] *= corr;
            // A operand: V^T rows d = 16 d + l15, contraction index e of quarter lq = key 16 (e / 4) + 4 lq + e % 4
            const _Float16* vr = Vt + (16 * d + l15) * kAtVld + 4 * lq;
            const h16x4 v0 = *reinterpret_cast<const h16x4*>(vr), v1 = *reinterpret_cast<const h16x4*>(vr + 16);
            h16x8 vf;
            vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
            vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[d], 0, 0, 0);
        }
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (qok) {
        float inv = 1.f / lsum;
        if (gates) inv *= 1.f / (1.f + expf(-gates[seq * g_seq_stride + (int64_t)q * g_row_stride + head]));    // out * sigmoid(gate[row][head])
        float* dst = out + seq * o_seq_stride + (int64_t)q * o_row_stride + head * kAtD;
#pragma unroll
        for (int d = 0; d < 4; ++d) *reinterpret_cast<f32x4*>(dst + 16 * d + 4 * lq) = o[d] * inv;      // O^T rows d = 16 d + 4 lq + r
    }
}

// Roformer band split, input side, for ALL bands in one launch: gather a band's bins of one frame from the spectrogram ([4][F][T] as
// alsep_stft writes it; merged index m = 2 f + s), RMS-normalise them over the band's true width (lucidrains RMSNorm: x / max(||x||,
// 1e-12) sqrt(width) gamma, sum of squares in double as nn_rmsnorm_kernel) and store the row zero-padded to kmax: feat[band][t][kmax] is
// then the A operand of ONE batched Linear over the bands (weights zero-padded alike).  pidx[band][kmax / 2]: merged index or -1;
// gamma[band][kmax].  One wave per (band, frame).
__global__ void __launch_bounds__(kHThreads)
roformer_bandsplit_in_kernel(const float* __restrict__ spec, const int* __restrict__ pidx, const float* __restrict__ gamma,
                             const int* __restrict__ width, float* __restrict__ feat, int nb, int F, int T, int kmax) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (kHThreads / 64) + (threadIdx.x >> 6);
    if (w >= (int64_t)nb * T) return;
    const int band = (int)(w / T), t = (int)(w % T);
    const int* pi = pidx + (int64_t)band * (kmax / 2);
    const float* g = gamma + (int64_t)band * kmax;
    float* out = feat + w * kmax;
    const int64_t plane = (int64_t)F * T;
    constexpr int PER = 5;                                                   // kmax / 2 <= 320 entries per band
    float re[PER], im[PER];
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = lane + 64 * k;
        re[k] = im[k] = 0.f;
        if (i < kmax / 2) {
            const int m = pi[i];
            if (m >= 0) {
                const int f = m >> 1, sch = m & 1;
                re[k] = spec[(int64_t)(2 * sch) * plane + (int64_t)f * T + t];
                im[k] = spec[(int64_t)(2 * sch + 1) * plane + (int64_t)f * T + t];
                ss += (double)re[k] * (double)re[k] + (double)im[k] * (double)im[k];
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = sqrtf((float)width[band]) / fmaxf((float)sqrt(ss), 1e-12f);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = lane + 64 * k;
        if (i < kmax / 2) {
            out[2 * i] = re[k] * inv * g[2 * i];
            out[2 * i + 1] = im[k] * inv * g[2 * i + 1];
        }
    }
}

}  // namespace

// feat[band][t][kmax] = RMSNorm_band(gathered bins of frame t), zero-padded (see roformer_bandsplit_in_kernel); kmax <= 640, even
extern "C" int alsep_roformer_bandsplit_in(alsep_ctx* ctx, const float* spec, const int* pidx, const float* gamma, const int* width, float* feat,
                                           int nb, int F, int T, int kmax) {
    ALSEP_ENTER(ctx);
    if (!ctx || !spec || !pidx || !gamma || !width || !feat || nb < 1 || F < 1 || T < 1 || kmax < 2 || kmax % 2 || kmax > 640)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_roformer_bandsplit_in: bad argument");
    const int64_t waves = (int64_t)nb * T;
    hipLaunchKernelGGL(roformer_bandsplit_in_kernel, dim3((unsigned)ceil_div64(waves, kHThreads / 64)), dim3(kHThreads), 0, ctx->stream, spec, pidx,
                       gamma, width, feat, nb, F, T, kmax);
    ALSEP_LAUNCH_CHECK(ctx, "roformer_bandsplit_in_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_to_f16(alsep_ctx* ctx, const float* x, void* y, int64_t n) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y || n <= 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_to_f16: bad argument");
    int64_t g = (n + kHThreads - 1) / kHThreads;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(to_f16_kernel, dim3((unsigned)g), dim3(kHThreads), 0, ctx->stream, x, (_Float16*)y, n);
    ALSEP_LAUNCH_CHECK(ctx, "to_f16_kernel");
    return ALSEP_OK;
}

// C[b][M][N] = act(alpha A[b][M][K] W[b][N][K]^T + bias[b][N]) (+ R[b][M][N]); A / C / R float32 (row strides lda / ldc / ldr, batch
// strides in elements, 0 = shared), W f16 (row stride ldw).  Needs K % 8 == 0, N % 4 == 0, lda % 4 == 0, ldw % 8 == 0, ldc % 4 == 0,
// ldr % 4 == 0 and 16-byte aligned bases (returns ALSEP_ERR_ARG otherwise: the caller keeps such products on alsep_nn_bgemm_bias).
extern "C" int alsep_nn_gemm_f16w(alsep_ctx* ctx, const float* A, int64_t lda, int64_t sa_b, const void* W, int64_t ldw, int64_t sw_b, float* C,
                                  int64_t ldc, int64_t sc_b, const float* bias, int64_t bias_b, const float* R, int64_t ldr, int64_t sr_b,
                                  int nb, int M, int N, int K, float alpha, int act, const int* n_per_batch) {
    ALSEP_ENTER(ctx);
    if (!ctx || !A || !W || !C || nb < 1 || nb > 65535 || M < 1 || N < 1 || K < 8 || !(act == 0 || act == 3 || act == 5))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_gemm_f16w: bad argument");
    if (K % 8 || N % 4 || lda % 4 || ldw % 8 || ldc % 4 || sa_b % 4 || sw_b % 8 || sc_b % 4 || lda < K || ldw < K || ldc < N ||
        (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) || (R && (ldr % 4 || sr_b % 4 || ldr < N || ((uintptr_t)R & 15))))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_gemm_f16w: operands do not meet the alignment this kernel needs");
    GemmHArgs p{A, lda, sa_b, (const _Float16*)W, ldw, sw_b, C, ldc, sc_b, bias, bias_b, R, ldr, sr_b, M, N, K, alpha, act, n_per_batch};
    ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)nn_gemm_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHgLds));
    const dim3 grid((unsigned)ceil_div64(N, kHgBN), (unsigned)ceil_div64(M, kHgBM), (unsigned)nb);
    ProfScope prof(ctx, ALSEP_PROF_NN_GEMM_H);
    prof.work(2.0 * nb * (double)M * N * K, (double)nb * (4.0 * M * K + 2.0 * N * K + (R ? 8.0 : 4.0) * M * N));
    hipLaunchKernelGGL(nn_gemm_h_kernel, grid, dim3(kHThreads), kHgLds, ctx->stream, p);
    ALSEP_LAUNCH_CHECK(ctx, "nn_gemm_h_kernel");
    return ALSEP_OK;
}

// out[seq][row][head][64] = softmax(scale q k^T) v per (sequence, head) of a packed q | k | v projection (see nn_attn_h_kernel).
// Strides in floats; every row start must be 16-byte aligned (strides % 4 == 0); head dimension 64.  rot_table (optional,
// alsep_nn_rotary_table): q and k are rotary-embedded by their position in the sequence as they are loaded (qkv then holds the raw
// projection); gates (optional): the result is scaled by sigmoid(gates[seq * g_seq_stride + row * g_row_stride + head]).
extern "C" int alsep_nn_rotary_table(alsep_ctx* ctx, float* table, int L, int dim_head) {
    ALSEP_ENTER(ctx);
    if (!ctx || !table || L < 1 || dim_head < 2 || dim_head % 2) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_rotary_table: bad argument");
    hipLaunchKernelGGL(rotary_table_kernel, dim3((unsigned)ceil_div64((int64_t)L * dim_head / 2, 256)), dim3(256), 0, ctx->stream, table, L, dim_head);
    ALSEP_LAUNCH_CHECK(ctx, "rotary_table_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_attention_f16(alsep_ctx* ctx, const float* qkv, float* out, int n_seq, int L, int heads, int dim_head, int64_t seq_stride,
                                      int64_t row_stride, int64_t o_seq_stride, int64_t o_row_stride, float scale, const float* rot_table,
                                      const float* gates, int64_t g_seq_stride, int64_t g_row_stride) {
    ALSEP_ENTER(ctx);
    if (!ctx || !qkv || !out || n_seq < 1 || n_seq > 65535 || L < 1 || heads < 1 || heads > 65535)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: bad argument");
    if (dim_head != kAtD) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: head dimension %d (64 is implemented)", dim_head);
    if (seq_stride % 4 || row_stride % 4 || o_seq_stride % 4 || o_row_stride % 4 || (((uintptr_t)qkv | (uintptr_t)out) & 15))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: strides / bases must be multiples of 16 bytes");
    const dim3 grid((unsigned)ceil_div64(L, 64), (unsigned)heads, (unsigned)n_seq);
    ProfScope prof(ctx, ALSEP_PROF_NN_ATTN_H);
    prof.work(4.0 * n_seq * heads * (double)L * L * kAtD, 4.0 * n_seq * heads * (double)L * kAtD * 4.0);
    if (rot_table && ((uintptr_t)rot_table & 15)) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: rotary table must be 16-byte aligned");
    hipLaunchKernelGGL(nn_attn_h_kernel, grid, dim3(kHThreads), kAtLds, ctx->stream, qkv, out, L, heads, seq_stride, row_stride, o_seq_stride,
                       o_row_stride, scale, rot_table, gates, g_seq_stride, g_row_stride);
    ALSEP_LAUNCH_CHECK(ctx, "nn_attn_h_kernel");
    return ALSEP_OK;
}
