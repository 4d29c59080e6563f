// float32 storage with split-half contractions ("f32s"): the 1e-4 parity mode on the 16-bit matrix pipe.
//
// Included by tdfnet.hip inside its anonymous namespace (main translation unit only).  Activations, BatchNorm terms, residuals and
// epilogues are float32 exactly as in the float32 mode; only the contractions change: every float32 operand x is carried as two IEEE
// halves  hi = half(x),  lo = half((x - hi) * 2^11)  -- x - hi is exact in float32, so hi + lo / 2^11 reproduces x to 2^-23 relative (2^-36
// absolute below the half range) -- and a product a w is accumulated as
//     acc_h += a_hi w_hi                              (v_mfma_f32_16x16x32_f16, float32 accumulators)
//     acc_l += a_hi w_lo + a_lo w_hi                  (second accumulator set, worth 2^-11)
//     result = acc_h + acc_l / 2^11                   (the dropped a_lo w_lo term is 2^-22 of a product)
// Three 16-bit MFMAs (1024 flop / clk / SIMD each) replace eight v_mfma_f32_16x16x4_f32 (64 flop / clk / SIMD) per 16 x 16 x 32 block:
// 5.3 x the matrix throughput of the exact float32 kernels at the same storage and the same roundings everywhere else.  The lo parts
// are scaled so that they stay NORMAL halves wherever hi is (an unscaled residual of 2^-11 x would be subnormal for |x| < 0.25: gfx950's
// MFMA does keep subnormal inputs -- scripts/dbg/f16_denorm.hip -- but their spacing would cap the precision at 2^-25 absolute).
// Range: |activation| <= 65504 (half); every staging pass checks its operands and raises the network's range word; the host side reads it
// after each batch and runs an out-of-range batch again on the exact f32 MFMA kernels (tdfnet.py forward_nhwc).
// Weights are split once on the host (float64 -> hi / lo), activations where they are staged into LDS (VALU: cvt, sub, mul, cvt).
//
// Reference seam: the same network as the other modes (handlers/patch_separate.py:52; topology oracle/tdfnet_oracle.py).
#pragma once

#include "f32s_common.h"

// ------------------------------------------------------------------------------------------
// 3x3 convolution (pad 1) + scale / shift + ReLU, float32 in / out, split contraction.
// The tiling of conv3x3_kernel: 256 output pixels (TH x TW) x BN output channels per workgroup, wave w owns pixels [64 w, 64 w + 64); K
// loop over chunks of KC input channels; per chunk the halo patch and the weight block sit in LDS as a hi plane and a lo plane of halves.
// KC = 24: three 8-channel k-groups per pixel (pixel stride 48 B: ds_read_b128 of 16 consecutive pixels is conflict-free), 27 k-groups =
// 7 k-steps (one padded group), 76 KiB at 8 x 32 tiles -> two workgroups per CU, so one stages (global loads + the split's VALU) while the
// other runs its 36 MFMAs per k-step.
// ------------------------------------------------------------------------------------------
template <int KC, int BN, int TW>
struct ConvSCfg {
    static constexpr int G = 8;
    static constexpr int TH = 256 / TW, PW = TW + 2, PH = TH + 2;
    static constexpr int CG = KC / G;                  // k-groups per pixel per chunk
    static constexpr int KCP = (CG & 1) ? KC : KC + G; // pixel stride in halves: an odd number of 16-byte groups
    static constexpr int NG = 9 * CG, NS = (NG + 3) / 4;
    static constexpr int KP = NS * 4 * G + G;          // weight row stride in halves (odd in 16-byte groups)
    static constexpr int MR = BN / 16;
    static constexpr int PATCH = PH * PW * KCP;        // halves per patch plane
    static constexpr int WTS = BN * KP;                // halves per weight plane
    static constexpr int WPIECES = (2 * WTS * 2 + 1023) / 1024;   // 1-KiB LDS-DMA pieces of a chunk's weight image (hi + lo planes, padded)
    static constexpr int WIMG = WPIECES * 512;         // halves per chunk image in global memory and in LDS
    static constexpr size_t lds_bytes = sizeof(hs_t) * ((size_t)2 * PATCH + WIMG);
    static_assert(KC % G == 0 && BN % 16 == 0 && 256 % TW == 0 && TW % 16 == 0, "bad split conv tile");
};

template <int KC, int BN, int TW>
__global__ void __launch_bounds__(kThreads, 2)
conv3x3_f32s_kernel(const float* __restrict__ X, float* __restrict__ Y, const hs_t* __restrict__ Wp,
                    const float* __restrict__ scale, const float* __restrict__ shift, int Th, int Fw, int Cin,
                    int Cout, int tiles_t, int tiles_f, int ntiles, int ny_fastest, unsigned* __restrict__ range_flag) {
    typedef ConvSCfg<KC, BN, TW> Cf;
    bool bad = false;
    hs_t* ph = reinterpret_cast<hs_t*>(alsep_smem);
    hs_t* pl = ph + Cf::PATCH;
    hs_t* wh = pl + Cf::PATCH;                               // 2 PATCH halves = a multiple of 16 bytes
    hs_t* wl = wh + Cf::WTS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    // ny_fastest (1-D grid, ntiles % 8 == 0): the Cout / BN workgroups of one tile run back to back on one XCD (its patch comes from HBM once)
    int tile, ny;
    if (ny_fastest) {
        const int nyc = Cout / BN, x = blockIdx.x & 7, i = blockIdx.x >> 3;
        tile = x * (ntiles >> 3) + i / nyc;
        ny = i % nyc;
    } else {
        tile = xcd_remap(blockIdx.x, ntiles);
        ny = blockIdx.y;
    }
    const int tf = tile % tiles_f;  tile /= tiles_f;
    const int tt = tile % tiles_t;
    const int64_t b = tile / tiles_t;
    const int t0 = tt * Cf::TH, f0 = tf * TW;
    const int nq = Cin / KC;

    int pbase[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int pm = wave * 64 + ni * 16 + l15;
        pbase[ni] = ((pm / TW) * Cf::PW + (pm % TW)) * Cf::KCP;
    }
    f32x4 acch[Cf::MR][4], accl[Cf::MR][4];
#pragma unroll
    for (int mi = 0; mi < Cf::MR; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acch[mi][ni] = accl[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* xb = X + b * (int64_t)Th * Fw * Cin;
    constexpr int QG = KC / 4;                               // 16-byte float groups per pixel per chunk
    constexpr int NIT = (Cf::PH * Cf::PW * QG + kThreads - 1) / kThreads;     // patch items per thread per chunk
    // a chunk's PATCH loads are issued one chunk ahead, into registers, so that they fly during the MFMAs of the chunk before; its WEIGHT
    // image (already halves, in LDS order) is copied by LDS-DMA as soon as the previous chunk's fragment reads are over, beside the split
    // (VALU) and the LDS stores of the patch -- all of which run beside the other resident workgroup's MFMAs
    f32x4 xr[NIT];
    int poff[NIT];                                           // element offset of an item's pixel (channel group g4; < 2^31: one image)
    unsigned pvalid = 0;                                     // bit i: item i lies inside the image (others are staged as zeros)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int it = tid + i * kThreads;
        const int pix = it / QG, g4 = it % QG;
        const int t = t0 - 1 + pix / Cf::PW, f = f0 - 1 + pix % Cf::PW;
        const bool in = it < Cf::PH * Cf::PW * QG && t >= 0 && t < Th && f >= 0 && f < Fw;
        poff[i] = in ? (t * Fw + f) * Cin + g4 * 4 : 0;      // clamped: the load is unconditional (a load under a branch drains vmcnt at the join)
        pvalid |= (unsigned)in << i;
    }
    auto prefetch = [&](int q) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) xr[i] = *reinterpret_cast<const f32x4*>(xb + poff[i] + q * KC);
    };
    prefetch(0);
    for (int q = 0; q < nq; ++q) {
        __syncthreads();                                     // previous chunk's fragment reads are done
        const hs_t* wsrc = Wp + ((int64_t)ny * nq + q) * Cf::WIMG;            // hi plane, then lo plane, padded to whole 1-KiB pieces
        for (int i = wave; i < Cf::WPIECES; i += 4) glds16(wsrc + ((size_t)i * 64 + lane) * 8, wh + (size_t)i * 512);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int it = tid + i * kThreads;
            if (it < Cf::PH * Cf::PW * QG) {
                const int pix = it / QG, g4 = it % QG;
                const bool in = (pvalid >> i) & 1u;
                const float x[4] = {in ? xr[i][0] : 0.f, in ? xr[i][1] : 0.f, in ? xr[i][2] : 0.f, in ? xr[i][3] : 0.f};
                hsx4 hi, lo;
                split4(x, hi, lo, bad);
                *reinterpret_cast<hsx4*>(ph + pix * Cf::KCP + g4 * 4) = hi;
                *reinterpret_cast<hsx4*>(pl + pix * Cf::KCP + g4 * 4) = lo;
            }
        }
        __syncthreads();                                     // drains the LDS-DMA (vmcnt(0)) before the barrier
        if (q + 1 < nq) prefetch(q + 1);
#pragma unroll
        for (int s = 0; s < Cf::NS; ++s) {
            const int grp = 4 * s + lq;
            const int gc = grp < Cf::NG ? grp : Cf::NG - 1;   // padded groups: weights are zero there
            const int tap = gc / Cf::CG, cg = gc % Cf::CG;
            const int koff = ((tap / 3) * Cf::PW + (tap % 3)) * Cf::KCP + cg * 8;
            hsx8 xh[4], xl[4], fh[Cf::MR], fl[Cf::MR];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                xh[ni] = lds_hs(ph + pbase[ni] + koff);
                xl[ni] = lds_hs(pl + pbase[ni] + koff);
            }
#pragma unroll
            for (int mi = 0; mi < Cf::MR; ++mi) {
                fh[mi] = lds_hs(wh + (mi * 16 + l15) * Cf::KP + grp * 8);
                fl[mi] = lds_hs(wl + (mi * 16 + l15) * Cf::KP + grp * 8);
            }
#pragma unroll
            for (int mi = 0; mi < Cf::MR; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    mma_hs(acch[mi][ni], fh[mi], xh[ni]);
                    mma_hs(accl[mi][ni], fl[mi], xh[ni]);
                    mma_hs(accl[mi][ni], fh[mi], xl[ni]);
                }
        }
    }
    if (bad) atomicMax(range_flag, 1u);
    // epilogue: lane holds channels co..co+3 (rows 4*lq+r) of pixel column l15
    float* yb = Y + b * (int64_t)Th * Fw * Cout;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int pm = wave * 64 + ni * 16 + l15;
        const int t = t0 + pm / TW, f = f0 + pm % TW;
        if (t < Th && f < Fw) {
#pragma unroll
            for (int mi = 0; mi < Cf::MR; ++mi) {
                const int co = ny * BN + mi * 16 + 4 * lq;
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    y[r] = relu_nan(fmaf(fmaf(accl[mi][ni][r], kSplitInv, acch[mi][ni][r]), scale[co + r], shift[co + r]));
                store4(yb + ((int64_t)t * Fw + f) * Cout + co, y);
            }
        }
    }
}

// packed image: [ny][q][hi | lo][BN rows][KP halves] (+ padding to whole 1-KiB LDS-DMA pieces per chunk); k-group grp = tap * CG + cg holds input channels q KC + 8 cg .. of tap `tap`
template <int KC, int BN>
std::vector<hs_t> pack_conv3x3_split(const std::vector<float>& w, int cin, int cout) {
    typedef ConvSCfg<KC, BN, 64> Cf;                          // KP, CG, NG do not depend on TW
    const int nq = cin / KC, nn = cout / BN;
    std::vector<hs_t> out((size_t)nn * nq * Cf::WIMG, (hs_t)0.f);
    for (int j = 0; j < nn; ++j)
        for (int q = 0; q < nq; ++q)
            for (int r = 0; r < BN; ++r)
                for (int grp = 0; grp < Cf::NG; ++grp)
                    for (int e = 0; e < 8; ++e) {
                        const int tap = grp / Cf::CG, cg = grp % Cf::CG;
                        const int ci = q * KC + cg * 8 + e, co = j * BN + r;
                        const float v = w[(((size_t)co * cin + ci) * 3 + tap / 3) * 3 + tap % 3];
                        const size_t base = ((size_t)j * nq + q) * Cf::WIMG + (size_t)r * Cf::KP + grp * 8 + e;
                        split_host(v, &out[base], &out[base + Cf::WTS]);
                    }
    return out;
}

// ------------------------------------------------------------------------------------------
// tile GEMM of the ds / us convolutions and the TDF linears, float32 in / out, split contraction: 64 weight rows x 128 activation
// columns per workgroup, K tiles of 64 (8 k-groups of 8 halves), hi and lo planes for both operands (54 KiB: two workgroups per CU).
// ------------------------------------------------------------------------------------------
// weights: [hi | lo][Mp][Kp] halves; a thread stages rows tid / 8 + 32 j (j < 2), k-group tid % 8 of both planes
// ds: X [B,2T',2F',C] -> Y [B,T',F',M], K = 4C;  us: X [B,T',F',K] -> Y [B,2T',2F',C2] * skip, M = 4*C2   (pix_gemm_kernel's contract)
template <int MODE>
__global__ void __launch_bounds__(kThreads, 2)
pix_gemm_f32s_kernel(const float* __restrict__ X, float* __restrict__ Y, const hs_t* __restrict__ Wp,
                     const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ skip,
                     int M, int Mp, int K, int Kp, int64_t ncols, int Tp, int Fp, int C, int C2, unsigned* __restrict__ range_flag,
                     int nrb_fast) {
    typedef GemmSCfg Gc;
    bool bad = false;
    hs_t* Wh = reinterpret_cast<hs_t*>(alsep_smem);
    hs_t* Wl = Wh + Gc::WS;
    hs_t* Xh = Wl + Gc::WS;
    hs_t* Xl = Xh + Gc::XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    // nrb_fast (1-D grid, column tiles % 8 == 0): the Mp / 64 row blocks that read one column tile run back to back on one XCD, so the
    // activations come from HBM once and from that XCD's L2 afterwards (grid.x-fastest order re-read them once per row block: measured)
    int ctile = blockIdx.x, rb = blockIdx.y;
    if (nrb_fast) {
        const int nct = gridDim.x / nrb_fast, x = blockIdx.x & 7, i = blockIdx.x >> 3;
        ctile = x * (nct >> 3) + i / nrb_fast;
        rb = i % nrb_fast;
    }
    const int64_t col0 = (int64_t)ctile * Gc::BC;
    const int row0 = rb * Gc::BR;

    // this thread stages the 16-byte float group (tid % 16) of columns tid / 16 + 16 j, j < 8
    constexpr int FG = Gc::BK / 4;                           // float groups per column per K tile
    int64_t cbase[8];
    bool cvalid[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t col = col0 + tid / FG + 16 * j;
        cvalid[j] = col < ncols;
        if (MODE == PIX_DS) {
            const int64_t fp = col % Fp, tp = (col / Fp) % Tp, bb = col / ((int64_t)Fp * Tp);
            cbase[j] = ((bb * 2 * Tp + 2 * tp) * (2 * (int64_t)Fp) + 2 * fp) * C;
        } else {
            cbase[j] = col * K;
        }
    }
    f32x4 acch[4][2], accl[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acch[mi][ni] = accl[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the next K tile's global loads fly during this tile's MFMAs (registers); loads are unconditional on clamped addresses
    f32x4 xr[8];
    f32x4 wrh[2], wrl[2];
    const int wr_r = tid >> 3, wr_g = tid & 7;
    const int64_t wplane = (int64_t)Mp * Kp;
    auto prefetch = [&](int k0) {
        const int k = k0 + (tid % FG) * 4;
        int64_t off = k < K ? k : 0;
        if (MODE == PIX_DS) {                                // two contiguous runs of 2C (dy = 0, 1)
            const int seg = 2 * C, kc = k < K ? k : 0;
            off = (int64_t)(kc / seg) * (2 * (int64_t)Fp * C) + (kc % seg);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[j] = *reinterpret_cast<const f32x4*>(X + (cvalid[j] ? cbase[j] : 0) + off);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t src = (int64_t)(row0 + wr_r + 32 * j) * Kp + k0 + wr_g * 8;
            wrh[j] = *reinterpret_cast<const f32x4*>(Wp + src);
            wrl[j] = *reinterpret_cast<const f32x4*>(Wp + wplane + src);
        }
    };
    prefetch(0);
    for (int k0 = 0; k0 < Kp; k0 += Gc::BK) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<f32x4*>(Wh + (wr_r + 32 * j) * Gc::LD + wr_g * 8) = wrh[j];
            *reinterpret_cast<f32x4*>(Wl + (wr_r + 32 * j) * Gc::LD + wr_g * 8) = wrl[j];
        }
        const bool kin = k0 + (tid % FG) * 4 < K;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = cvalid[j] && kin;
            const float x[4] = {in ? xr[j][0] : 0.f, in ? xr[j][1] : 0.f, in ? xr[j][2] : 0.f, in ? xr[j][3] : 0.f};
            hsx4 hi, lo;
            split4(x, hi, lo, bad);
            const int o = (tid / FG + 16 * j) * Gc::LD + (tid % FG) * 4;
            *reinterpret_cast<hsx4*>(Xh + o) = hi;
            *reinterpret_cast<hsx4*>(Xl + o) = lo;
        }
        __syncthreads();
        if (k0 + Gc::BK < Kp) prefetch(k0 + Gc::BK);
        gemm_tile_compute_s<true>(Wh, Wl, Xh, Xl, acch, accl, wave, l15, lq);
    }
    if (bad) atomicMax(range_flag, 1u);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int64_t col = col0 + wave * 32 + ni * 16 + l15;
        if (col >= ncols) continue;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = row0 + mi * 16 + 4 * lq;
            if (row >= M) continue;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                y[r] = relu_nan(fmaf(fmaf(accl[mi][ni][r], kSplitInv, acch[mi][ni][r]), scale[row + r], shift[row + r]));
            if (MODE == PIX_DS) {
                store4(Y + col * M + row, y);
            } else {
                const int d = row / C2, co = row % C2;
                const int64_t fp = col % Fp, tp = (col / Fp) % Tp, bb = col / ((int64_t)Fp * Tp);
                const int64_t o = ((bb * 2 * Tp + 2 * tp + (d >> 1)) * (2 * (int64_t)Fp) + 2 * fp + (d & 1)) * C2 + co;
                float s[4];
                load4(skip + o, s);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] *= s[r];
                store4(Y + o, y);
            }
        }
    }
}

// TDF linear over the F axis (tdf_gemm_kernel's contract): columns are units of 16 channels of one (b, t); the k axis (f) is strided in
// memory, so the stage transposes through LDS: a thread loads the same 4 channels of TWO consecutive f rows and writes, per channel, the
// pair of halves as one 4-byte LDS store (k is the fast axis of the LDS image)
template <bool RESIDUAL>
__global__ void __launch_bounds__(kThreads, 2)
tdf_gemm_f32s_kernel(const float* __restrict__ X, float* __restrict__ Y, const hs_t* __restrict__ Wp,
                     const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
                     const float* __restrict__ R, int M, int Mp, int K, int Kp, int64_t nunits, int C, unsigned* __restrict__ range_flag,
                     int nrb_fast) {
    typedef GemmSCfg Gc;
    bool bad = false;
    hs_t* Wh = reinterpret_cast<hs_t*>(alsep_smem);
    hs_t* Wl = Wh + Gc::WS;
    hs_t* Xh = Wl + Gc::WS;
    hs_t* Xl = Xh + Gc::XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    int ctile = blockIdx.x, rb = blockIdx.y;                 // nrb_fast: as in pix_gemm_f32s_kernel
    if (nrb_fast) {
        const int nct = gridDim.x / nrb_fast, x = blockIdx.x & 7, i = blockIdx.x >> 3;
        ctile = x * (nct >> 3) + i / nrb_fast;
        rb = i % nrb_fast;
    }
    const int64_t u0 = (int64_t)ctile * 8;
    const int row0 = rb * Gc::BR;
    const int upc = C / 16;

    // staging: wave w stages units 2 w, 2 w + 1; inside a wave lane = (k pair kp = lane / 4, channel quad cgi = lane % 4): the four lanes
    // of a k row read one 64-byte run, and a wave's 4-byte LDS stores (two consecutive k of one channel) fall on 32 banks at most two deep
    // (bank = 4 column + kp mod 32 with the 36-dword column stride; a unit-fastest map put 16 lanes on one bank)
    const int cgi = lane & 3, kp = lane >> 2;               // kp in [0, 16): k rows 2 kp, 2 kp + 1 of a 32-row half tile
    int64_t xbase[2];
    bool uvalid[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t u = u0 + wave * 2 + h;
        uvalid[h] = u < nunits;
        xbase[h] = uvalid[h] ? ((u / upc) * (int64_t)K) * C + (u % upc) * 16 + cgi * 4 : 0;
    }
    f32x4 acch[4][2], accl[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acch[mi][ni] = accl[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // item (unit half h, k half kh): rows k0 + 32 kh + 2 kp (+ 1); loads are unconditional on clamped rows (zeroed when stored)
    f32x4 xa[4], xb2[4];
    f32x4 wrh[2], wrl[2];                                   // this thread's two weight pieces per plane (bit copies): rows tid / 8 + 32 j, group tid % 8
    const int wr_r = tid >> 3, wr_g = tid & 7;
    const int64_t wplane = (int64_t)Mp * Kp;
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const int k = k0 + 32 * kh + 2 * kp;
                const int ka = k < K ? k : 0, kb = k + 1 < K ? k + 1 : 0;
                xa[h * 2 + kh] = *reinterpret_cast<const f32x4*>(X + xbase[h] + (int64_t)ka * C);
                xb2[h * 2 + kh] = *reinterpret_cast<const f32x4*>(X + xbase[h] + (int64_t)kb * C);
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t src = (int64_t)(row0 + wr_r + 32 * j) * Kp + k0 + wr_g * 8;
            wrh[j] = *reinterpret_cast<const f32x4*>(Wp + src);
            wrl[j] = *reinterpret_cast<const f32x4*>(Wp + wplane + src);
        }
    };
    prefetch(0);
    for (int k0 = 0; k0 < Kp; k0 += Gc::BK) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<f32x4*>(Wh + (wr_r + 32 * j) * Gc::LD + wr_g * 8) = wrh[j];
            *reinterpret_cast<f32x4*>(Wl + (wr_r + 32 * j) * Gc::LD + wr_g * 8) = wrl[j];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const int kk = 32 * kh + 2 * kp, k = k0 + kk;
                const bool va = uvalid[h] && k < K, vb = uvalid[h] && k + 1 < K;
                const f32x4 ra = xa[h * 2 + kh], rb = xb2[h * 2 + kh];
                const float fa[4] = {va ? ra[0] : 0.f, va ? ra[1] : 0.f, va ? ra[2] : 0.f, va ? ra[3] : 0.f};
                const float fb[4] = {vb ? rb[0] : 0.f, vb ? rb[1] : 0.f, vb ? rb[2] : 0.f, vb ? rb[3] : 0.f};
                hsx4 ha, la, hb, lb;
                split4(fa, ha, la, bad);
                split4(fb, hb, lb, bad);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = ((wave * 2 + h) * 16 + cgi * 4 + i) * Gc::LD + kk;
                    *reinterpret_cast<hsx2*>(Xh + o) = hsx2{ha[i], hb[i]};
                    *reinterpret_cast<hsx2*>(Xl + o) = hsx2{la[i], lb[i]};
                }
            }
        __syncthreads();
        if (k0 + Gc::BK < Kp) prefetch(k0 + Gc::BK);
        gemm_tile_compute_s<false>(Wh, Wl, Xh, Xl, acch, accl, wave, l15, lq);
    }
    if (bad) atomicMax(range_flag, 1u);
    // D rows = channel within unit (4*lq + r), D cols = weight row f' (l15)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int64_t uu = u0 + wave * 2 + ni;
        if (uu >= nunits) continue;
        const int64_t bt = uu / upc;
        const int c = (int)(uu % upc) * 16 + 4 * lq;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int fo = row0 + mi * 16 + l15;
            if (fo >= M) continue;
            const float bv = bias ? bias[fo] : 0.f;
            const int64_t o = (bt * M + fo) * C + c;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                y[r] = relu_nan(fmaf(fmaf(accl[mi][ni][r], kSplitInv, acch[mi][ni][r]) + bv, scale[c + r], shift[c + r]));
            if (RESIDUAL) {
                float x[4];
                load4(R + o, x);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] += x[r];
            }
            store4(Y + o, y);
        }
    }
}

// [M][K] row-major float32 -> [hi | lo][Mp][Kp] halves, zero padded to the tile
inline std::vector<hs_t> pack_gemm_split(const std::vector<float>& wmk, int M, int K, int* Mp_out, int* Kp_out) {
    typedef GemmSCfg Gc;
    const int Mp = (int)ceil_div64(M, Gc::BR) * Gc::BR, Kp = (int)ceil_div64(K, Gc::BK) * Gc::BK;
    std::vector<hs_t> pk((size_t)2 * Mp * Kp, (hs_t)0.f);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) split_host(wmk[(size_t)m * K + k], &pk[(size_t)m * Kp + k], &pk[(size_t)Mp * Kp + (size_t)m * Kp + k]);
    *Mp_out = Mp;
    *Kp_out = Kp;
    return pk;
}
