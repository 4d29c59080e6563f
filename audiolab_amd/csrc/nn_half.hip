// Half-precision MFMA kernels of the transformer model families (Mel-Band / BS Roformer; the reference runs them under torch autocast,
// modules/separator/stem_separator.py:106 ``use_autocast=True``: Linear layers and attention in IEEE half, everything else in float32).
// gfx950 only.
//
// Storage follows autocast: the residual stream and every statistic stay float32; what a Linear or the attention READS is IEEE half in
// HBM -- the RMSNorm output, the q | k | v projection, the attention output, the GELU / tanh hidden activations -- written in that
// type by the kernel that produces it (one rounding, where autocast rounds), products on v_mfma_f32_16x16x32_f16 with float32
// accumulation.  Why storage and not just operand rounding: these GEMMs have K = 384 ... 1536 against M = 48 060 rows, so they live on
// the per-CU intake path (~70 GB/s per CU from L2, MI355X_MICROARCH.md): measured with float32 activations converted on the way into
// LDS, the 128 x 128 tile moved 40 KB per 1.05 MFLOP (every 128-byte line of A fetched twice, half used each time) and ran at 0.26
// PFLOP/s with the MFMA pipe 10 % busy; half-precision activations in whole 128-byte lines are 32 KB per 2.1 MFLOP.
//
//   nn_rmsnorm_h_kernel  float32 rows -> RMS-normalised f16 rows (the A operand of the next Linear)
//   nn_gemm_hh_kernel    C[M][N] = act(alpha A[M][K] W[N][K]^T + bias) (+ residual): A, W f16; C f16 or float32; batched, ragged N
//   nn_attn_h_kernel     softmax(Q K^T) V of a packed f16 q | k | v projection in one pass (no score matrix in HBM), rotary embedding on
//                        load, head gates in the epilogue, f16 out
//   roformer_bandsplit_in_kernel   gather + RMSNorm of every band of a frame -> zero-padded f16 rows of one batched Linear
#include "alsep_common.h"
#include <alsep_gfx950_asm.h>
#include "nn_gemm_h2.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

constexpr int kHThreads = 256;

// GELU(erf) with erf by Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7 -- three orders below the half rounding of the stored result): one
// exponential and a degree-5 polynomial instead of erff's branchy 40-instruction expansion (40 of the 162 us of the 48 060 x 1536 Linear)
__device__ __forceinline__ float gelu_erf_h(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));   // v_rcp_f32 (1 ulp): the IEEE division is ten instructions, and this epilogue
                                                                      // is a third of the 48 060 x 1536 Linear's time
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = 1.f - poly * __expf(-x * x);
    return 0.5f * v * (1.f + copysignf(erf_abs, v));
}

__device__ __forceinline__ h16x8 to_h8(const f32x4& a, const f32x4& b) {
    h16x8 r;
    r[0] = (_Float16)a[0]; r[1] = (_Float16)a[1]; r[2] = (_Float16)a[2]; r[3] = (_Float16)a[3];
    r[4] = (_Float16)b[0]; r[5] = (_Float16)b[1]; r[6] = (_Float16)b[2]; r[7] = (_Float16)b[3];
    return r;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// GEMM.  Workgroup tile 128 (M) x 128 (N), 4 waves as 2 x 2 (64 x 64 each: 16 accumulator blocks), K in slices of 64 = two MFMA steps =
// one whole 128-byte line of every operand row.  LDS image of a slice: rows of 64 halves (eight 16-byte k-groups), group g of row r at
// g ^ (r & 7): the ds_read_b128 of 16 consecutive rows x one group and the staging ds_write_b128 are conflict-free
// (SQ_LDS_BANK_CONFLICT = 0).  Operand roles are swapped (D rows = n, D columns = m): a lane ends up with four consecutive n of one m.
//
// Pipeline: the global loads of slice kt + 2 are issued at the top of iteration kt into the register set that iteration kt - 1 emptied,
// slice kt + 1 (requested one iteration earlier) is stored to the other LDS buffer after the MFMAs of slice kt -- by then it has had a
// whole iteration to arrive, and the wait for it is a counted vmcnt that leaves the newer slice in flight.  Every load is UNCONDITIONAL
// (rows / k-groups outside the matrices read a clamped, valid address and are zeroed when they are stored): a load under a branch makes
// hipcc wait vmcnt(0) at the join and the prefetch distance collapses to zero; scheduling fences keep hipcc from hoisting the next
// iteration's register reads (and with them the wait) above the barrier.
// Grid: 1-D over (batch, m tile, n tile), n fastest, dealt to the XCDs in contiguous runs (consecutive workgroup ids go round-robin
// over the 8 XCDs): the N / 128 workgroups that read the same 128 rows of A run on ONE XCD and share its L2 (PMC: A fetched once).
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int kHgBM = 128, kHgBN = 128, kHgBK = 64;
constexpr size_t kHgLds = 2 * (size_t)(kHgBM + kHgBN) * kHgBK * sizeof(_Float16);      // 64 KiB: two workgroups per CU

struct GemmHArgs {
    const _Float16* A; int64_t lda, sa_b;
    const _Float16* B; int64_t ldb, sb_b;
    void* C; int64_t ldc, sc_b;
    const float* bias; int64_t bias_b;
    const float* R; int64_t ldr, sr_b;            // optional float32 residual added after the activation: C = act(...) + R
    int M, N, K;
    float alpha;
    const int* nvec;                              // optional: columns of batch b (<= N): the ragged last layers of the mask estimators
};

__device__ __forceinline__ int hg_slot(int row, int g) { return row * kHgBK + 8 * (g ^ (row & 7)); }

// one slice's worth of a thread's staging registers: four A granules and four W granules of 8 halves
struct HgStage {
    h16x8 a[4], b[4];
};

template <int ACT>
__device__ __forceinline__ float hg_act(float t) {
    if (ACT == 3) return gelu_erf_h(t);
    if (ACT == 5) return tanhf(t);
    return t;
}

#ifndef ALSEP_NN_HALF_CONV_TU
template <int ACT, bool CF16>
__global__ void __launch_bounds__(kHThreads, 2)
nn_gemm_hh_kernel(GemmHArgs p) {
    _Float16* As = reinterpret_cast<_Float16*>(alsep_smem);                  // [2][128][64]
    _Float16* Bs = As + 2 * kHgBM * kHgBK;                                    // [2][128][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + kHgBN - 1) / kHgBN, tiles_m = (p.M + kHgBM - 1) / kHgBM;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int bz = wg / (tiles_n * tiles_m);
    const int tm = (wg / tiles_n) % tiles_m, tn = wg % tiles_n;
    const _Float16* a = p.A + bz * p.sa_b;
    const _Float16* b = p.B + bz * p.sb_b;
    const int m0 = tm * kHgBM, n0 = tn * kHgBN;
    const int Nb = p.nvec ? p.nvec[bz] : p.N;
    if (n0 >= Nb) return;                                                    // whole workgroups leave together
    // staging duty per slice: 1024 granules per operand, four per thread: rows sr + 32 h, k-group sg (8 lanes = one 128-byte line)
    const int sr = tid >> 3, sg = tid & 7;
    // row addresses as one base per operand + 32-bit element offsets (the launcher checks rows * ld < 2^31): four 64-bit pointers per
    // operand were the registers that pushed the kernel into scratch at 256 VGPRs -- and kernels that use scratch corrupt each
    // other's spilled registers when they run concurrently from several HIP streams on this stack (the runner's lanes; measured:
    // 3e-3 ... 6e-2 run-to-run differences in the stems, gone with one hardware queue or without scratch), so none of the kernels a
    // lane launches may spill
    unsigned oa[4], ob[4];
    bool va[4], vb[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int ra = m0 + sr + 32 * h, rb = n0 + sr + 32 * h;
        va[h] = ra < p.M;
        vb[h] = rb < Nb;
        oa[h] = (unsigned)((va[h] ? ra : 0) * (int)p.lda);
        ob[h] = (unsigned)((vb[h] ? rb : 0) * (int)p.ldb);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 zh;
#pragma unroll
    for (int e = 0; e < 8; ++e) zh[e] = (_Float16)0.f;
    auto gload = [&](HgStage& r, int k0) {                                   // K % 8 == 0: a granule is inside K or outside it
        const int kq = k0 + 8 * sg;
        const int ks = kq < p.K ? kq : 0;                                    // clamped: always a valid address (zeroed in lstore)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            r.a[h] = *reinterpret_cast<const h16x8*>(a + (oa[h] + (unsigned)ks));
            r.b[h] = *reinterpret_cast<const h16x8*>(b + (ob[h] + (unsigned)ks));
        }
    };
    // the registers are first TOUCHED here, an iteration after their loads were issued
    auto lstore = [&](const HgStage& r, int buf, int k0) {
        const bool in = k0 + 8 * sg < p.K;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int row = sr + 32 * h;
            *reinterpret_cast<h16x8*>(As + (size_t)buf * kHgBM * kHgBK + hg_slot(row, sg)) = (va[h] && in) ? r.a[h] : zh;
            *reinterpret_cast<h16x8*>(Bs + (size_t)buf * kHgBN * kHgBK + hg_slot(row, sg)) = (vb[h] && in) ? r.b[h] : zh;
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            h16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const h16x8*>(As + (size_t)buf * kHgBM * kHgBK + hg_slot(wm * 64 + i * 16 + l15, 4 * st + lq));
                bf[i] = *reinterpret_cast<const h16x8*>(Bs + (size_t)buf * kHgBN * kHgBK + hg_slot(wn * 64 + i * 16 + l15, 4 * st + lq));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    const int nk = (p.K + kHgBK - 1) / kHgBK;
    HgStage r0, r1;
    gload(r0, 0);
    gload(r1, kHgBK);                                                        // beyond K: clamped address, zeroed when stored
    lstore(r0, 0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // even iteration: r0 is free (slice kt sits in LDS buffer 0), r1 holds slice kt + 1
        gload(r0, (kt + 2) * kHgBK);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        lstore(r1, 1, (kt + 1) * kHgBK);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 >= nk) break;
        // odd iteration: r1 is free, r0 holds slice kt + 2
        gload(r1, (kt + 3) * kHgBK);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        lstore(r0, 0, (kt + 2) * kHgBK);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: D rows = n (4 lq + r), D columns = m (l15): C[m][n .. n + 3] is one 16- / 8-byte store (N % 4 == 0)
    const float* bias = p.bias ? p.bias + bz * p.bias_b : nullptr;
    const float* res = p.R ? p.R + bz * p.sr_b : nullptr;
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + 4 * lq;
        bv[j] = (bias && col < Nb) ? *reinterpret_cast<const f32x4*>(bias + col) : z4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 64 + i * 16 + l15;
        if (row >= p.M) continue;
        f32x4 rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + 4 * lq;
            rv[j] = (res && col < Nb) ? *reinterpret_cast<const f32x4*>(res + (int64_t)row * p.ldr + col) : z4;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + 4 * lq;
            if (col >= Nb) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = hg_act<ACT>(p.alpha * acc[i][j][r] + bv[j][r]) + rv[j][r];
            if (CF16) {
                h16x4 hv;
                hv[0] = (_Float16)v[0]; hv[1] = (_Float16)v[1]; hv[2] = (_Float16)v[2]; hv[3] = (_Float16)v[3];
                *reinterpret_cast<h16x4*>(reinterpret_cast<_Float16*>(p.C) + bz * p.sc_b + (int64_t)row * p.ldc + col) = hv;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + bz * p.sc_b + (int64_t)row * p.ldc + col) = v;
            }
        }
    }
}

// the large-M form (nn_gemm_h2.h): persistent, 256 x 128 tiles, LDS-DMA ring across tile ends
template <int BM, int ACT, bool CF16, bool RES, bool RAGK>
__global__ void __launch_bounds__(h2::Geo<BM>::kThreads, 2)
nn_gemm_h2_kernel(h2::Args p) {
    h2::gemm_body<BM, ACT, CF16, RES, RAGK>(p, [](float t) { return hg_act<ACT>(t); });
}

#else   // ALSEP_NN_HALF_CONV_TU: the convolution kernel and its entry points, compiled as nn_conv_half.hip (which says why)
// ------------------------------------------------------------------------------------------------------------------------------------
// Convolution as the same GEMM (MDX23C's TFC convolutions in half-precision mode): M = output pixels, N = output channels,
// K = (tap, ci) with Cin % 64 == 0, so a 64-wide K slice is 64 consecutive input channels of ONE tap -- one 128-byte line of the pixel the
// tap lands on (channels-last activations stored as IEEE half by the InstanceNorm kernel that produced them).  Only the A operand's
// addressing differs from nn_gemm_hh_kernel: a thread's four staging rows keep their output pixel's (image base, iy0, ix0), and per
// slice the tap's offset is added; a tap outside the image reads a clamped address and is zeroed when it is stored to LDS (the
// validity travels with the register set it belongs to).  Weights [Cout][KH KW Cin] in IEEE half.  Epilogue: float32 into a channel
// slice of y (the residual stream), optional float32 residual (the block's shortcut branch) -- four consecutive channels per lane.
// ------------------------------------------------------------------------------------------------------------------------------------
struct ConvHArgs {
    const _Float16* x; const _Float16* w;
    float* y; const float* R;
    int64_t npix, ldr;
    int H, W, Cin, Cout, Ho, Wo, KH, KW, sh, sw, ph, pw, y_ct, y_c0;
    int splits, nk_per;                            // split K: workgroup (tile, s) sums slices [s nk_per, (s + 1) nk_per) into part[s]
    float* part;                                   // [splits][npix][Cout] float32 (splits > 1)
};

struct HcStage {
    h16x8 a[4], b[4];
    unsigned ok;                                                              // bit h: row h's tap is inside the image
};

__global__ void __launch_bounds__(kHThreads, 2)
nn_conv_hh_kernel(ConvHArgs p) {
    _Float16* As = reinterpret_cast<_Float16*>(alsep_smem);
    _Float16* Bs = As + 2 * kHgBM * kHgBK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.Cout + kHgBN - 1) / kHgBN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int sp = wg % p.splits, tile = wg / p.splits;
    const int64_t tm = tile / tiles_n;
    const int tn = tile % tiles_n;
    const int64_t m0 = tm * kHgBM;
    const int n0 = tn * kHgBN;
    const int K = p.KH * p.KW * p.Cin;
    const int sr = tid >> 3, sg = tid & 7;
    // per staging row: image base as a 32-bit element offset, (iy0, ix0) packed into one register, weight row offset (the launcher
    // checks that x and w have fewer than 2^31 elements): no 64-bit pointers per row -- the kernel must not spill (see the GEMM)
    unsigned xo[4];
    int yx0[4];                                                               // iy0 in the high half, ix0 in the low (both in [-32768, 32767])
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int pix = (int)m0 + sr + 32 * h;                               // fewer than 2^31 pixels (launcher)
        const bool va = pix < (int)p.npix;                                   // a row beyond the last pixel: iy0 far above the image, every tap "outside"
        const int pc = va ? pix : 0;
        const int img = pc / (p.Ho * p.Wo);
        const int rem = pc - img * (p.Ho * p.Wo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        yx0[h] = ((va ? oy * p.sh - p.ph : -32000) << 16) | ((ox * p.sw - p.pw) & 0xffff);
        xo[h] = (unsigned)img * (unsigned)(p.H * p.W * p.Cin) + 8u * (unsigned)sg;
    }
    // weight rows beyond Cout read the last row: their products land in output columns that are never stored, so nothing needs zeroing
    // there (nor in pixel rows beyond npix, nor in a slice beyond this workgroup's range, which is staged but never multiplied); only a
    // tap outside the image must contribute zeros
    const int wrow = n0 + sr;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 zh;
#pragma unroll
    for (int e = 0; e < 8; ++e) zh[e] = (_Float16)0.f;
    const int kt_lo = sp * p.nk_per;
    const int nk = min(K / kHgBK - kt_lo, p.nk_per);                         // this workgroup's slices: kt_lo + [0, nk)
    auto gload = [&](HcStage& r, int kt) {
        const int kc = kt_lo + (kt < nk ? kt : 0);                           // beyond the range: any valid slice (zeroed in lstore)
        const int k0 = kc * kHgBK;
        const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
        const int dy = tap / p.KW, dx = tap - dy * p.KW;
        r.ok = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int iy = (yx0[h] >> 16) + dy, ix = (int)(short)(yx0[h] & 0xffff) + dx;
            const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            r.ok |= (in ? 1u : 0u) << h;
            const unsigned off = in ? (unsigned)((iy * p.W + ix) * p.Cin + ci0) : 0u;
            r.a[h] = *reinterpret_cast<const h16x8*>(p.x + (xo[h] + off));
            r.b[h] = *reinterpret_cast<const h16x8*>(p.w + ((unsigned)(min(wrow + 32 * h, p.Cout - 1) * K) + 8u * (unsigned)sg + (unsigned)k0));
        }
    };
    auto lstore = [&](const HcStage& r, int buf) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int row = sr + 32 * h;
            *reinterpret_cast<h16x8*>(As + (size_t)buf * kHgBM * kHgBK + hg_slot(row, sg)) = ((r.ok >> h) & 1u) ? r.a[h] : zh;
            *reinterpret_cast<h16x8*>(Bs + (size_t)buf * kHgBN * kHgBK + hg_slot(row, sg)) = r.b[h];
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            h16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const h16x8*>(As + (size_t)buf * kHgBM * kHgBK + hg_slot(wm * 64 + i * 16 + l15, 4 * st + lq));
                bf[i] = *reinterpret_cast<const h16x8*>(Bs + (size_t)buf * kHgBN * kHgBK + hg_slot(wn * 64 + i * 16 + l15, 4 * st + lq));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    HcStage r0, r1;
    gload(r0, 0);
    gload(r1, 1);
    lstore(r0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        gload(r0, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        lstore(r1, 1);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 >= nk) break;
        gload(r1, kt + 3);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        lstore(r0, 0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    }
    {
    // the epilogue's lane coordinates are recomputed from the thread index: kept alive across the main loop they were the one register
    // too many (a kernel that touches scratch must not run on concurrent streams here, see the GEMM)
    const int te = opaque_vgpr((int)threadIdx.x);
    const int l15 = te & 15, lq = (te >> 4) & 3, wm = te >> 7, wn = (te >> 6) & 1;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.splits > 1) {                                                      // partial sums; conv_splitk_reduce_kernel finishes
        float* part = p.part + (int64_t)sp * p.npix * p.Cout;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t pix = m0 + wm * 64 + i * 16 + l15;
            if (pix >= p.npix) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + 4 * lq;
                if (col < p.Cout) *reinterpret_cast<f32x4*>(part + pix * p.Cout + col) = acc[i][j];
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t pix = m0 + wm * 64 + i * 16 + l15;
        if (pix >= p.npix) continue;
        f32x4 rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + 4 * lq;
            rv[j] = (p.R && col < p.Cout) ? *reinterpret_cast<const f32x4*>(p.R + pix * p.ldr + col) : z4;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + 4 * lq;
            if (col >= p.Cout) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + rv[j][r];
            *reinterpret_cast<f32x4*>(p.y + pix * p.y_ct + p.y_c0 + col) = v;
        }
    }
    }
}

// y slice = part[0] + part[1] + ... (in that order) (+ R): four channels per thread
__global__ void __launch_bounds__(kHThreads)
conv_splitk_reduce_kernel(const float* __restrict__ part, int splits, int64_t npix, int Cout, const float* __restrict__ R, int64_t ldr,
                          float* __restrict__ y, int y_ct, int y_c0) {
    const int64_t n4 = npix * (Cout / 4);
    for (int64_t i = (int64_t)blockIdx.x * kHThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kHThreads) {
        const int64_t pix = i / (Cout / 4);
        const int col = (int)(i - pix * (Cout / 4)) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(part + pix * Cout + col);
        for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(part + ((int64_t)s * npix + pix) * Cout + col);
        if (R) v += *reinterpret_cast<const f32x4*>(R + pix * ldr + col);
        *reinterpret_cast<f32x4*>(y + pix * y_ct + y_c0 + col) = v;
    }
}

#endif  // ALSEP_NN_HALF_CONV_TU (kernels)
#ifndef ALSEP_NN_HALF_CONV_TU
// lucidrains RMSNorm (y = x / max(||x||_2, 1e-12) sqrt(C) gamma, sum of squares in double as nn_rmsnorm_kernel) with the result stored
// as IEEE half: the A operand of the Linear that follows.  One wave per row.
__global__ void __launch_bounds__(kHThreads)
nn_rmsnorm_h_kernel(const float* __restrict__ x, _Float16* __restrict__ y, const float* __restrict__ gamma, int64_t rows, int C, int64_t x_stride,
                    int64_t y_stride) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kHThreads / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * x_stride;
    double ss = 0.0;
    for (int c = lane; c < C; c += 64) ss += (double)xr[c] * (double)xr[c];
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float inv = sqrtf((float)C) / fmaxf((float)sqrt(ss), 1e-12f);
    _Float16* yr = y + row * y_stride;
    for (int c = lane; c < C; c += 64) yr[c] = (_Float16)(xr[c] * inv * gamma[c]);
}

// The same for rows of C = 64 NQ values (the Roformers' 384 / 512): four rows per wave, sixteen lanes per row, a lane's NQ column quads loaded
// once as 16-byte vectors and kept in registers, 8-byte half stores (a wave instruction moves 1 KB in and 512 B out instead of 256 / 128 B):
// 29 -> 2x us per 48 060 x 384 norm.  The sum of squares is taken in double in another order than nn_rmsnorm_h_kernel's: the float32
// result differs from it by at most an ulp, below the half rounding of the stored row.
template <int NQ>
__global__ void __launch_bounds__(kHThreads)
nn_rmsnorm_h4_kernel(const float* __restrict__ x, _Float16* __restrict__ y, const float* __restrict__ gamma, int64_t rows, int64_t x_stride,
                     int64_t y_stride) {
    constexpr int C = 64 * NQ;
    const int lane = threadIdx.x & 63, j = lane & 15;
    const int64_t row = ((int64_t)blockIdx.x * (kHThreads / 64) + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool ok = row < rows;                                               // every lane takes part in the shuffles: a row beyond the end reads the last one
    const float* xr = x + (ok ? row : rows - 1) * x_stride + 4 * j;
    f32x4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = *reinterpret_cast<const f32x4*>(xr + 64 * q);
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += (double)v[q][e] * (double)v[q][e];
    ss += __shfl_xor(ss, 8);
    ss += __shfl_xor(ss, 4);
    ss += __shfl_xor(ss, 2);
    ss += __shfl_xor(ss, 1);
    const float inv = sqrtf((float)C) / fmaxf((float)sqrt(ss), 1e-12f);
    if (!ok) return;
    _Float16* yr = y + row * y_stride + 4 * j;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * j + 64 * q);
        h16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)(v[q][e] * inv * g[e]);
        *reinterpret_cast<h16x4*>(yr + 64 * q) = o;
    }
}

__global__ void __launch_bounds__(kHThreads)
to_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kHThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kHThreads) y[i] = (_Float16)x[i];
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Attention.  qkv: IEEE-half rows of 3 * heads * 64 values (q | k | v, head-major inside each) as the projection wrote them; a
// sequence is L rows `row_stride` elements apart starting at seq * seq_stride.  q and k are rotary-embedded by their position in the
// sequence as they are loaded (float32 arithmetic on the stored halves, table of rotary_table_kernel), q scaled by d^-1/2; the result is
// scaled by sigmoid(gates[row][head]) and stored as IEEE half (the A operand of the output projection) through (o_seq_stride,
// o_row_stride).  head dimension 64.
//
// One workgroup = 64 QB queries of one (sequence, head), four waves of QB blocks of 16 queries; keys / values in chunks of 64 staged
// through LDS for all four waves (K rows of 64 halves, 16-byte groups XOR-swizzled by row & 7: conflict-free ds_read_b128; V TRANSPOSED,
// rows of 64 keys padded to 72 halves: conflict-free ds_read_b64).  Per chunk and wave:
//     S^T[key][query] = K Q^T            4 key blocks x 2 MFMA steps (d = 64) per query block; Q fragments live in registers (scaled by
//                                        d^-1/2 log2 e: the scores are base-2 exponents)
//     online softmax per query           a query's 64 scores sit in the 4 lanes {l15, l15 + 16, + 32, + 48}: two shuffles for the max
//     O^T[d][query] += V^T P^T           4 d blocks x 2 MFMA steps; P^T is the B operand AS THE LANE HOLDS IT (contraction index e of lane
//                                        quarter lq = key 32 ks + 16 (e / 4) + 4 lq + e % 4, V^T read in the same order), O rescaled per lane
// The row sums are carried per lane and reduced once at the end.  float32 everywhere outside the two MFMA operands.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int kAtD = 64, kAtKc = 64, kAtVld = 72;

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
constexpr size_t kAtLds = (size_t)kAtKc * kAtD * sizeof(_Float16) + (size_t)kAtD * kAtVld * sizeof(_Float16);     // 8192 + 9216

// 8 halves -> two float4
__device__ __forceinline__ void h8_to_f(const h16x8& v, f32x4& a, f32x4& b) {
    a[0] = (float)v[0]; a[1] = (float)v[1]; a[2] = (float)v[2]; a[3] = (float)v[3];
    b[0] = (float)v[4]; b[1] = (float)v[5]; b[2] = (float)v[6]; b[3] = (float)v[7];
}

// One pass over the keys in chunks of 64, online softmax, no score matrix in HBM.  A workgroup = 4 waves x QB blocks of 16 queries
// (QB = 2: 128 queries for the long sequences; QB = 1 for sequences of <= 64).  The kernel is bound by vector-instruction issue, not by
// the MFMAs (the 32-key, 64-query form ran ~0.39 vector instructions per (query, key) pair: 304 us for 60 x 8 sequences of 801), so the
// structure minimises those: the K / V staging and K's rotation are shared by twice the queries, a chunk is 64 keys (half the barriers
// and accumulator rescales per key), the scores are kept in the log2 domain (scale log2(e) folded into Q: one v_exp_f32 per score,
// no multiply), keys beyond the sequence are masked only in the one chunk that has them, and V^T is scattered into LDS as 4-byte
// pairs of keys.
template <int QB>
__global__ void __launch_bounds__(kHThreads)
nn_attn_h_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ out, int L, int heads, int64_t seq_stride, int64_t row_stride,
                 int64_t o_seq_stride, int64_t o_row_stride, float scale, const float* __restrict__ rot, const float* __restrict__ gates,
                 int64_t g_seq_stride, int64_t g_row_stride) {
    _Float16* Ks = reinterpret_cast<_Float16*>(alsep_smem);                  // [64 keys][64 d], swizzled 16-byte groups
    _Float16* Vt = Ks + kAtKc * kAtD;                                         // [64 d][72]: V transposed, rows padded by 8
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    // 1-D grid over (sequence, head, query block), query block fastest, dealt to the XCDs in contiguous runs: the workgroups that read the
    // same keys / values share one L2
    constexpr int QW = 64 * QB;                                               // queries per workgroup
    const int qblocks = (L + QW - 1) / QW;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = wg % qblocks, head = (wg / qblocks) % heads, seq = wg / (qblocks * heads);
    const int inner = heads * kAtD;
    const _Float16* base = qkv + seq * seq_stride + head * kAtD;
    const _Float16* kbase = base + inner;
    const _Float16* vbase = base + 2 * inner;
    // Q fragments: B operand of S^T = K Q^T: lane (col = query l15, quarter lq) holds Q[q][32 s + 8 lq .. + 7], rotated, times
    // scale log2(e)
    const float qs = scale * 1.44269504088896340736f;
    h16x8 qf[QB][2];
    int qrow[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        qrow[b] = qb * QW + (wave * QB + b) * 16 + l15;                       // this lane's query of block b (the MFMA column)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, c = a;
            if (qrow[b] < L) {
                h8_to_f(*reinterpret_cast<const h16x8*>(base + (int64_t)qrow[b] * row_stride + 32 * s2 + 8 * lq), a, c);
                if (rot) rotate8(a, c, rot + ((int64_t)qrow[b] * (kAtD / 2) + 16 * s2 + 4 * lq) * 2);
            }
            qf[b][s2] = to_h8(a * qs, c * qs);
        }
    }
    f32x4 o[QB][4];
    float mrun[QB], lsum[QB];                                                 // running max (shared by a query's 4 lanes), this lane's partial sum
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        mrun[b] = -3.0e38f;
        lsum[b] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[b][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // staging duty per chunk.  K: keys kkey and kkey + 32, 8 d values from 8 kgrp (8 lanes = one 128-byte row: coalesced).  V: the key
    // PAIR (2 vpair, 2 vpair + 1), 8 d values from 8 vgrp, one 32-lane half per d group: its eight 4-byte stores (one per d, two keys
    // each) then fall into 32 different banks
    const int kkey = tid >> 3, kgrp = tid & 7;
    const int vpair = tid & 31, vgrp = tid >> 5;
    // the next chunk's K / V rows are requested while this chunk is multiplied (one register set ahead: a chunk's loads used to be waited
    // for right where they were issued, every iteration, with only the other resident workgroups to cover the latency)
    h16x8 kv[2], vv[2];
    auto request = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kr = k0 + kkey + 32 * h, krc = kr < L ? kr : L - 1;     // clamped: the loads are unconditional; masked below
            kv[h] = *reinterpret_cast<const h16x8*>(kbase + (int64_t)krc * row_stride + 8 * kgrp);
            const int vr = k0 + 2 * vpair + h, vrc = vr < L ? vr : L - 1;
            vv[h] = *reinterpret_cast<const h16x8*>(vbase + (int64_t)vrc * row_stride + 8 * vgrp);
        }
    };
    request(0);
    for (int k0 = 0; k0 < L; k0 += kAtKc) {
        if (rot) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kr = k0 + kkey + 32 * h, krc = kr < L ? kr : L - 1;
                f32x4 ka, kb;
                h8_to_f(kv[h], ka, kb);
                rotate8(ka, kb, rot + ((int64_t)krc * (kAtD / 2) + 4 * kgrp) * 2);
                kv[h] = to_h8(ka, kb);
            }
        }
        __syncthreads();                                                      // every wave is done with the previous chunk
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = kkey + 32 * h;
            *reinterpret_cast<h16x8*>(Ks + row * kAtD + 8 * (kgrp ^ (row & 7))) = kv[h];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
            h16x2 pr;
            pr[0] = vv[0][e];
            pr[1] = vv[1][e];
            *reinterpret_cast<h16x2*>(Vt + (8 * vgrp + e) * kAtVld + 2 * vpair) = pr;
        }
        __syncthreads();
        request(k0 + kAtKc < L ? k0 + kAtKc : k0);                           // past the end: this chunk again (unconditional loads)
        // S^T blocks (log2 domain): keys 16 kb + (4 lq + r), query l15 of block b; the K fragments serve both query blocks
        f32x4 sc[QB][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int krow = 16 * kb + l15;
            h16x8 kf[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) kf[st] = *reinterpret_cast<const h16x8*>(Ks + krow * kAtD + 8 * ((4 * st + lq) ^ (krow & 7)));
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                sc[b][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[b][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                sc[b][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[b][1], sc[b][kb], 0, 0, 0);
            }
        }
        if (k0 + kAtKc > L) {                                                 // the last chunk only: keys beyond the sequence
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + 16 * kb + 4 * lq + r >= L) sc[b][kb][r] = -3.0e38f;
        }
        h16x8 pf[QB][2];
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            float cmax = fmaxf(fmaxf(sc[b][0][0], sc[b][0][1]), fmaxf(sc[b][0][2], sc[b][0][3]));
#pragma unroll
            for (int kb = 1; kb < 4; ++kb) cmax = fmaxf(cmax, fmaxf(fmaxf(sc[b][kb][0], sc[b][kb][1]), fmaxf(sc[b][kb][2], sc[b][kb][3])));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
            const float mnew = fmaxf(mrun[b], cmax);
            const float corr = __builtin_amdgcn_exp2f(mrun[b] - mnew);
            mrun[b] = mnew;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sc[b][kb][r] - mnew);   // a masked key: exp2(-3e38) = 0
                    psum += pv;
                    pf[b][kb >> 1][4 * (kb & 1) + r] = (_Float16)pv;
                }
            lsum[b] = lsum[b] * corr + psum;
#pragma unroll
            for (int d = 0; d < 4; ++d) o[b][d] *= corr;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // A operand: V^T rows d = 16 d + l15, contraction index e of quarter lq = key 32 ks + 16 (e / 4) + 4 lq + e % 4
                const _Float16* vr = Vt + (16 * d + l15) * kAtVld + 32 * ks + 4 * lq;
                const h16x4 v0 = *reinterpret_cast<const h16x4*>(vr), v1 = *reinterpret_cast<const h16x4*>(vr + 16);
                h16x8 vf;
                vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
#pragma unroll
                for (int b = 0; b < QB; ++b) o[b][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[b][ks], o[b][d], 0, 0, 0);
            }
    }
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        float ls = lsum[b];
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        const int q = qrow[b];
        if (q < L) {
            float inv = 1.f / ls;
            if (gates) inv *= 1.f / (1.f + expf(-gates[seq * g_seq_stride + (int64_t)q * g_row_stride + head]));    // out * sigmoid(gate[row][head])
            _Float16* dst = out + seq * o_seq_stride + (int64_t)q * o_row_stride + head * kAtD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {                                     // O^T rows d = 16 d + 4 lq + r
                h16x4 hv;
                hv[0] = (_Float16)(o[b][d][0] * inv); hv[1] = (_Float16)(o[b][d][1] * inv);
                hv[2] = (_Float16)(o[b][d][2] * inv); hv[3] = (_Float16)(o[b][d][3] * inv);
                *reinterpret_cast<h16x4*>(dst + 16 * d + 4 * lq) = hv;
            }
        }
    }
}

// Roformer band split, input side, for ALL bands in one launch: gather a band's bins of one frame from the spectrogram ([4][F][T] as
// alsep_stft writes it; merged index m = 2 f + s), RMS-normalise them over the band's true width (lucidrains RMSNorm: x / max(||x||,
// 1e-12) sqrt(width) gamma, sum of squares in double as nn_rmsnorm_kernel) and store the row zero-padded to kmax, as IEEE half: feat[band][t][kmax] is
// then the A operand of ONE batched Linear over the bands (weights zero-padded alike).  pidx[band][kmax / 2]: merged index or -1;
// gamma[band][kmax].  One wave per (band, frame).
__global__ void __launch_bounds__(kHThreads)
roformer_bandsplit_in_kernel(const float* __restrict__ spec, const int* __restrict__ pidx, const float* __restrict__ gamma,
                             const int* __restrict__ width, _Float16* __restrict__ feat, int nb, int F, int T, int kmax) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * (kHThreads / 64) + (threadIdx.x >> 6);
    if (w >= (int64_t)nb * T) return;
    const int band = (int)(w / T), t = (int)(w % T);
    const int* pi = pidx + (int64_t)band * (kmax / 2);
    const float* g = gamma + (int64_t)band * kmax;
    _Float16* out = feat + w * kmax;
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
            out[2 * i] = (_Float16)(re[k] * inv * g[2 * i]);
            out[2 * i + 1] = (_Float16)(im[k] * inv * g[2 * i + 1]);
        }
    }
}

#endif  // !ALSEP_NN_HALF_CONV_TU
}  // namespace

// feat[band][t][kmax] = RMSNorm_band(gathered bins of frame t), zero-padded (see roformer_bandsplit_in_kernel); kmax <= 640, even
#ifndef ALSEP_NN_HALF_CONV_TU
extern "C" int alsep_roformer_bandsplit_in(alsep_ctx* ctx, const float* spec, const int* pidx, const float* gamma, const int* width, void* feat,
                                           int nb, int F, int T, int kmax) {
    ALSEP_ENTER(ctx);
    if (!ctx || !spec || !pidx || !gamma || !width || !feat || nb < 1 || F < 1 || T < 1 || kmax < 2 || kmax % 2 || kmax > 640)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_roformer_bandsplit_in: bad argument");
    const int64_t waves = (int64_t)nb * T;
    hipLaunchKernelGGL(roformer_bandsplit_in_kernel, dim3((unsigned)ceil_div64(waves, kHThreads / 64)), dim3(kHThreads), 0, ctx->stream, spec, pidx,
                       gamma, width, (_Float16*)feat, nb, F, T, kmax);
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

// C[b][M][N] = act(alpha A[b][M][K] W[b][N][K]^T + bias[b][N]) (+ R[b][M][N]): A, W IEEE half; C half (c_f16) or float32; bias, R float32.
// Row strides ld*, batch strides s*_b in elements (0 = shared).  Needs K % 8 == 0, N % 4 == 0, lda % 8 == 0, ldw % 8 == 0, ldc % 4 == 0,
// ldr % 4 == 0 and 16-byte (C half: 8-byte) aligned bases (ALSEP_ERR_ARG otherwise).  n_per_batch (device, optional): columns of
// batch b.
extern "C" int alsep_nn_gemm_f16(alsep_ctx* ctx, const void* A, int64_t lda, int64_t sa_b, const void* W, int64_t ldw, int64_t sw_b, void* C,
                                 int c_f16, int64_t ldc, int64_t sc_b, const float* bias, int64_t bias_b, const float* R, int64_t ldr,
                                 int64_t sr_b, int nb, int M, int N, int K, float alpha, int act, const int* n_per_batch) {
    ALSEP_ENTER(ctx);
    if (!ctx || !A || !W || !C || nb < 1 || M < 1 || N < 1 || K < 8 || !(act == 0 || act == 3 || act == 5))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_gemm_f16: bad argument");
    if (K % 8 || N % 4 || lda % 8 || ldw % 8 || ldc % 4 || sa_b % 8 || sw_b % 8 || sc_b % 4 || bias_b % 4 || lda < K || ldw < K || ldc < N ||
        (((uintptr_t)A | (uintptr_t)W) & 15) || ((uintptr_t)C & (c_f16 ? 7 : 15)) || (bias && ((uintptr_t)bias & 15)) ||
        (R && (ldr % 4 || sr_b % 4 || ldr < N || ((uintptr_t)R & 15))))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_gemm_f16: operands do not meet the alignment this kernel needs");
    if ((int64_t)M * lda >= ((int64_t)1 << 31) || (int64_t)N * ldw >= ((int64_t)1 << 31))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_gemm_f16: an operand of 2^31 or more elements per batch (32-bit row offsets)");
    // Large M without per-batch ragged N: the persistent kernel (one workgroup per CU).  Its residual / bias requests carry 32-bit byte
    // offsets; an activation together with a residual stays with the tile-per-workgroup kernel (no caller has one).
    if (M >= 2048 && N >= 64 && !n_per_batch && !(R && act != 0) && (!R || (int64_t)M * ldr < ((int64_t)1 << 30)) ) {
        if (!ctx->zero_page) {
            ALSEP_HIP(ctx, hipMalloc(&ctx->zero_page, 256));
            ALSEP_HIP(ctx, hipMemsetAsync(ctx->zero_page, 0, 256, ctx->stream));
        }
        int g = device_cu_count(ctx) / 8 * 8;                        // a multiple of the 8 XCDs
        if (g < 8) g = 8;
        // tile height: 192 where its tiles fill the workgroups' rounds (rounds x height = rows a workgroup walks) at least 15 % better -- the
        // six-wave kernel is that much slower per flop (48 060 x 1536 x 384: 88-91 us against 73-75), so a 2 % better fill (BS Roformer's
        // 49 662 x 1536) must not select it; the 384-column Linears of Mel-Band (576 against 768) do
        const int64_t tn = ceil_div64(N, h2::BN);
        const int64_t t256 = ceil_div64(M, 256) * tn * nb, t192 = ceil_div64(M, 192) * tn * nb;
        const int bm = ceil_div64(t192, g) * 192 * 115 < ceil_div64(t256, g) * 256 * 100 ? 192 : 256;
        const int64_t nt = bm == 192 ? t192 : t256;
        if (nt > 0x7fffffff) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_gemm_f16: too many tiles");
        h2::Args q{(const _Float16*)A, lda, sa_b, (const _Float16*)W, ldw, sw_b, C, ldc, sc_b, bias, bias_b, R, ldr, sr_b, M, N, K, alpha,
                   (const _Float16*)ctx->zero_page, (int)ceil_div64(M, bm), (int)tn, (int)nt};
        if (nt < g) g = (int)((nt + 7) / 8 * 8);                     // fewer workgroups than tiles never idle a whole XCD
        ProfScope prof(ctx, ALSEP_PROF_NN_GEMM_H);
        prof.work(2.0 * nb * (double)M * N * K, (double)nb * (2.0 * M * K + 2.0 * N * K + ((c_f16 ? 2.0 : 4.0) + (R ? 4.0 : 0.0)) * M * N));
#define ALSEP_H2_LAUNCH(BM_, ACT_, CF_, RES_, RAG_)                                                                                        \
    do {                                                                                                                                    \
        typedef h2::Geo<BM_> Geo_;                                                                                                          \
        ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)nn_gemm_h2_kernel<BM_, ACT_, CF_, RES_, RAG_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)Geo_::kLds));                                                                              \
        hipLaunchKernelGGL((nn_gemm_h2_kernel<BM_, ACT_, CF_, RES_, RAG_>), dim3((unsigned)g), dim3(Geo_::kThreads), Geo_::kLds, ctx->stream, q); \
    } while (0)
#define ALSEP_H2_GO(ACT_, CF_, RES_)                                                                                                        \
    do {                                                                                                                                    \
        if (bm == 192) {                                                                                                                    \
            if (K % h2::BK) ALSEP_H2_LAUNCH(192, ACT_, CF_, RES_, true); else ALSEP_H2_LAUNCH(192, ACT_, CF_, RES_, false);                \
        } else {                                                                                                                            \
            if (K % h2::BK) ALSEP_H2_LAUNCH(256, ACT_, CF_, RES_, true); else ALSEP_H2_LAUNCH(256, ACT_, CF_, RES_, false);                \
        }                                                                                                                                   \
    } while (0)
        if (R) {
            if (c_f16) ALSEP_H2_GO(0, true, true); else ALSEP_H2_GO(0, false, true);
        } else if (c_f16) {
            if (act == 3) ALSEP_H2_GO(3, true, false); else if (act == 5) ALSEP_H2_GO(5, true, false); else ALSEP_H2_GO(0, true, false);
        } else {
            if (act == 3) ALSEP_H2_GO(3, false, false); else if (act == 5) ALSEP_H2_GO(5, false, false); else ALSEP_H2_GO(0, false, false);
        }
#undef ALSEP_H2_GO
#undef ALSEP_H2_LAUNCH
        ALSEP_LAUNCH_CHECK(ctx, "nn_gemm_h2_kernel");
        return ALSEP_OK;
    }
    GemmHArgs p{(const _Float16*)A, lda, sa_b, (const _Float16*)W, ldw, sw_b, C, ldc, sc_b, bias, bias_b, R, ldr, sr_b, M, N, K, alpha, n_per_batch};
    const int64_t n_wg = ceil_div64(N, kHgBN) * ceil_div64(M, kHgBM) * nb;
    if (n_wg > 0x7fffffff) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_gemm_f16: too many tiles");
    const dim3 grid((unsigned)n_wg);
    ProfScope prof(ctx, ALSEP_PROF_NN_GEMM_H);
    prof.work(2.0 * nb * (double)M * N * K, (double)nb * (2.0 * M * K + 2.0 * N * K + ((c_f16 ? 2.0 : 4.0) + (R ? 4.0 : 0.0)) * M * N));
#define ALSEP_HG_GO(ACT_, CF_)                                                                                                              \
    do {                                                                                                                                    \
        ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)nn_gemm_hh_kernel<ACT_, CF_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHgLds)); \
        hipLaunchKernelGGL((nn_gemm_hh_kernel<ACT_, CF_>), grid, dim3(kHThreads), kHgLds, ctx->stream, p);                                  \
    } while (0)
    if (c_f16) {
        if (act == 3) ALSEP_HG_GO(3, true); else if (act == 5) ALSEP_HG_GO(5, true); else ALSEP_HG_GO(0, true);
    } else {
        if (act == 3) ALSEP_HG_GO(3, false); else if (act == 5) ALSEP_HG_GO(5, false); else ALSEP_HG_GO(0, false);
    }
#undef ALSEP_HG_GO
    ALSEP_LAUNCH_CHECK(ctx, "nn_gemm_hh_kernel");
    return ALSEP_OK;
}

#else   // ALSEP_NN_HALF_CONV_TU
// Split K: a layer whose 128 x 128 tiles do not fill the chip (the deep levels of the U-Net: 8 x 32 pixels x 768 channels are 12 tiles
// with 108 K slices each -- 225 us at 12 TFLOP/s) is cut along K into `splits` ranges of >= 8 slices, each workgroup writes its partial
// tile, conv_splitk_reduce_kernel adds them in a fixed order (deterministic, unlike atomics).
static void conv_h_split(int64_t npix, int Cout, int K, int* splits, int* nk_per) {
    const int64_t tiles = ceil_div64(Cout, kHgBN) * ceil_div64(npix, kHgBM);
    const int nk = K / kHgBK;
    int s = 1;
    if (tiles < 256) {
        const int64_t want = ceil_div64(384, tiles);
        const int cap = nk / 8 > 1 ? nk / 8 : 1;
        s = (int)(want < cap ? want : cap);
    }
    *nk_per = (nk + s - 1) / s;
    *splits = (nk + *nk_per - 1) / *nk_per;
}
static int conv_h_geometry(int H, int W, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int* Ho, int* Wo) {
    *Ho = (H + 2 * pad_h - KH) / stride_h + 1;
    *Wo = (W + 2 * pad_w - KW) / stride_w + 1;
    return *Ho >= 1 && *Wo >= 1;
}
// bytes of workspace alsep_nn_conv2d_f16 needs for this layer (0: none)
extern "C" int64_t alsep_nn_conv2d_f16_workspace_bytes(int64_t B, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h, int stride_w,
                                                       int pad_h, int pad_w) {
    int Ho, Wo, splits, nk_per;
    if (B < 1 || H < 1 || W < 1 || Cin < 64 || Cin % 64 || Cout < 1 || KH < 1 || KW < 1 || stride_h < 1 || stride_w < 1 || pad_h < 0 || pad_w < 0 ||
        !conv_h_geometry(H, W, KH, KW, stride_h, stride_w, pad_h, pad_w, &Ho, &Wo))
        return -1;
    conv_h_split(B * Ho * Wo, Cout, KH * KW * Cin, &splits, &nk_per);
    return splits > 1 ? (int64_t)splits * B * Ho * Wo * Cout * (int64_t)sizeof(float) : 0;
}

// y[pixel][y_coff + co] = sum_{tap, ci} x[pixel's tap][ci] w[co][tap][ci] (+ R[pixel][co]): x, w IEEE half (x channels-last [B, H, W, Cin],
// w [Cout][KH][KW][Cin]), y / R float32.  Needs Cin % 64 == 0, Cout % 4 == 0, y_ctotal % 4 == 0, y_coff % 4 == 0, ldr % 4 == 0, 16-byte
// aligned bases, and alsep_nn_conv2d_f16_workspace_bytes(...) bytes of workspace (ALSEP_ERR_ARG otherwise).
extern "C" int alsep_nn_conv2d_f16(alsep_ctx* ctx, const void* x, const void* w, float* y, const float* R, int64_t ldr, int64_t B, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int y_ctotal, int y_coff,
                                   void* workspace, int64_t workspace_bytes) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !w || !y || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1 || stride_h < 1 || stride_w < 1 ||
        pad_h < 0 || pad_w < 0 || y_coff < 0 || y_coff + Cout > y_ctotal)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d_f16: bad argument");
    int Ho, Wo;
    if (!conv_h_geometry(H, W, KH, KW, stride_h, stride_w, pad_h, pad_w, &Ho, &Wo)) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d_f16: empty output");
    if (Cin % 64 || Cout % 4 || y_ctotal % 4 || y_coff % 4 || (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) ||
        (R && (ldr % 4 || ldr < Cout || ((uintptr_t)R & 15))))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d_f16: operands do not meet the alignment this kernel needs (Cin %% 64, Cout %% 4)");
    const int64_t npix = B * Ho * Wo;
    if (B * (int64_t)H * W * Cin >= ((int64_t)1 << 31) || (int64_t)Cout * KH * KW * Cin >= ((int64_t)1 << 31) || H > 32000 || W > 32000)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d_f16: an operand of 2^31 or more elements, or an image side above 32 000 (32-bit / packed offsets)");
    int splits, nk_per;
    conv_h_split(npix, Cout, KH * KW * Cin, &splits, &nk_per);
    if (splits > 1 && (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < (int64_t)splits * npix * Cout * (int64_t)sizeof(float)))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d_f16: workspace missing or too small (alsep_nn_conv2d_f16_workspace_bytes)");
    const int64_t n_wg = ceil_div64(Cout, kHgBN) * ceil_div64(npix, kHgBM) * splits;
    if (n_wg > 0x7fffffff) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d_f16: too many tiles");
    ConvHArgs p{(const _Float16*)x, (const _Float16*)w, y, R, npix, ldr, H, W, Cin, Cout, Ho, Wo, KH, KW, stride_h, stride_w, pad_h, pad_w, y_ctotal,
                y_coff, splits, nk_per, (float*)workspace};
    ProfScope prof(ctx, ALSEP_PROF_NN_CONV_H);
    const double K = (double)KH * KW * Cin;
    prof.work(2.0 * (double)npix * Cout * K, 2.0 * (double)B * H * W * Cin + 2.0 * Cout * K + (R ? 8.0 : 4.0) * (double)npix * Cout);
    ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)nn_conv_hh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHgLds));
    hipLaunchKernelGGL(nn_conv_hh_kernel, dim3((unsigned)n_wg), dim3(kHThreads), kHgLds, ctx->stream, p);
    ALSEP_LAUNCH_CHECK(ctx, "nn_conv_hh_kernel");
    if (splits > 1) {
        const int64_t n4 = npix * (Cout / 4);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>(ceil_div64(n4, kHThreads), 4096)), dim3(kHThreads), 0, ctx->stream,
                           (const float*)workspace, splits, npix, Cout, R, ldr, y, y_ctotal, y_coff);
        ALSEP_LAUNCH_CHECK(ctx, "conv_splitk_reduce_kernel");
    }
    return ALSEP_OK;
}

#endif  // ALSEP_NN_HALF_CONV_TU (entry points)
#ifndef ALSEP_NN_HALF_CONV_TU
// y[r][C] (IEEE half) = RMSNorm(x[r][C]) gamma (see nn_rmsnorm_h_kernel); strides in elements
extern "C" int alsep_nn_rmsnorm_f16(alsep_ctx* ctx, const float* x, void* y, const float* gamma, int64_t rows, int C, int64_t x_stride,
                                    int64_t y_stride) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y || !gamma || rows < 1 || C < 1 || x_stride < C || y_stride < C) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_rmsnorm_f16: bad argument");
    const bool vec = (C == 384 || C == 512) && x_stride % 4 == 0 && y_stride % 4 == 0 && !(((uintptr_t)x | (uintptr_t)gamma) & 15) && !((uintptr_t)y & 7);
    if (vec) {
        const dim3 grid((unsigned)ceil_div64(rows, 4 * (kHThreads / 64)));
        if (C == 384) hipLaunchKernelGGL(nn_rmsnorm_h4_kernel<6>, grid, dim3(kHThreads), 0, ctx->stream, x, (_Float16*)y, gamma, rows, x_stride, y_stride);
        else hipLaunchKernelGGL(nn_rmsnorm_h4_kernel<8>, grid, dim3(kHThreads), 0, ctx->stream, x, (_Float16*)y, gamma, rows, x_stride, y_stride);
        ALSEP_LAUNCH_CHECK(ctx, "nn_rmsnorm_h4_kernel");
        return ALSEP_OK;
    }
    hipLaunchKernelGGL(nn_rmsnorm_h_kernel, dim3((unsigned)ceil_div64(rows, kHThreads / 64)), dim3(kHThreads), 0, ctx->stream, x, (_Float16*)y, gamma,
                       rows, C, x_stride, y_stride);
    ALSEP_LAUNCH_CHECK(ctx, "nn_rmsnorm_h_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_rotary_table(alsep_ctx* ctx, float* table, int L, int dim_head) {
    ALSEP_ENTER(ctx);
    if (!ctx || !table || L < 1 || dim_head < 2 || dim_head % 2) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_rotary_table: bad argument");
    hipLaunchKernelGGL(rotary_table_kernel, dim3((unsigned)ceil_div64((int64_t)L * dim_head / 2, 256)), dim3(256), 0, ctx->stream, table, L, dim_head);
    ALSEP_LAUNCH_CHECK(ctx, "rotary_table_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_nn_attention_f16(alsep_ctx* ctx, const void* qkv, void* out, int n_seq, int L, int heads, int dim_head, int64_t seq_stride,
                                      int64_t row_stride, int64_t o_seq_stride, int64_t o_row_stride, float scale, const float* rot_table,
                                      const float* gates, int64_t g_seq_stride, int64_t g_row_stride) {
    ALSEP_ENTER(ctx);
    if (!ctx || !qkv || !out || n_seq < 1 || L < 1 || heads < 1)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: bad argument");
    if (dim_head != kAtD) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: head dimension %d (64 is implemented)", dim_head);
    if (seq_stride % 8 || row_stride % 8 || o_seq_stride % 4 || o_row_stride % 4 || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 7))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: strides / bases must be multiples of 16 (qkv) / 8 (out) bytes");
    const int QB = L > 64 ? 2 : 1;                               // query blocks of 16 per wave: 128 or 64 queries per workgroup
    const int64_t n_wg = ceil_div64(L, 64 * QB) * heads * n_seq;
    if (n_wg > 0x7fffffff) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: too many workgroups");
    const dim3 grid((unsigned)n_wg);
    ProfScope prof(ctx, ALSEP_PROF_NN_ATTN_H);
    prof.work(4.0 * n_seq * heads * (double)L * L * kAtD, 2.0 * n_seq * heads * (double)L * kAtD * 4.0);
    if (rot_table && ((uintptr_t)rot_table & 15)) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_attention_f16: rotary table must be 16-byte aligned");
    if (QB == 2)
        hipLaunchKernelGGL(nn_attn_h_kernel<2>, grid, dim3(kHThreads), kAtLds, ctx->stream, (const _Float16*)qkv, (_Float16*)out, L, heads, seq_stride,
                           row_stride, o_seq_stride, o_row_stride, scale, rot_table, gates, g_seq_stride, g_row_stride);
    else
        hipLaunchKernelGGL(nn_attn_h_kernel<1>, grid, dim3(kHThreads), kAtLds, ctx->stream, (const _Float16*)qkv, (_Float16*)out, L, heads, seq_stride,
                           row_stride, o_seq_stride, o_row_stride, scale, rot_table, gates, g_seq_stride, g_row_stride);
    ALSEP_LAUNCH_CHECK(ctx, "nn_attn_h_kernel");
    return ALSEP_OK;
}
#endif  // !ALSEP_NN_HALF_CONV_TU
