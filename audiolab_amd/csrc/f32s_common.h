// Shared pieces of the split-half contraction ("f32s": float32 storage, contractions as three f16 MFMA products of (hi, lo) half pairs with
// float32 accumulation; the scheme and its error budget are described at the top of tdfnet_f32s.h).  Included by tdfnet_f32s.h (TFC-TDF
// network kernels) and nn_f32s.h (the generic float32 GEMM / convolution of the other model families).
#pragma once

typedef _Float16 hs_t;
typedef _Float16 hsx8 __attribute__((ext_vector_type(8)));
typedef _Float16 hsx4 __attribute__((ext_vector_type(4)));
typedef _Float16 hsx2 __attribute__((ext_vector_type(2)));
constexpr float kSplitScale = 2048.f, kSplitInv = 1.f / 2048.f;

__device__ __forceinline__ void mma_hs(f32x4& acc, const hsx8& a, const hsx8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ hsx8 lds_hs(const hs_t* p) { return *reinterpret_cast<const hsx8*>(p); }
// ReLU that keeps a NaN (fmaxf would turn the NaN of an out-of-range activation into a silent 0; the runners test the stems for it)
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

// x -> (hi, lo): four values at a time (one 16-byte float group = two 8-byte half groups).  `bad` collects "some |x| is beyond the half
// range, or not a number": gfx950 turns the out-of-range products into finite garbage further down (measured: no NaN reaches the output), so
// the range is checked where the operands are made and reported through the network's range flag (alsep_net_range_flag).
__device__ __forceinline__ void split4(const float (&x)[4], hsx4& hi, hsx4& lo, bool& bad) {
    const float m = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
    bad |= !(m <= 65504.f) | (x[0] != x[0]) | (x[1] != x[1]) | (x[2] != x[2]) | (x[3] != x[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const hs_t h = (hs_t)x[e];
        hi[e] = h;
        lo[e] = (hs_t)((x[e] - (float)h) * kSplitScale);
    }
}
inline void split_host(float x, hs_t* hi, hs_t* lo) {
    const hs_t h = (hs_t)x;
    *hi = h;
    *lo = (hs_t)(((double)x - (double)(float)h) * 2048.0);
}

// ------------------------------------------------------------------------------------------
// 64 x 128 tile of a split GEMM: "W" operand 64 rows, "X" operand 128 rows (wave w owns X rows [32 w, 32 w + 32)), K tiles of 64 (8 k-groups
// of 8 halves), a hi and a lo plane per operand in LDS with rows padded to 72 halves (an odd number of 16-byte groups: conflict-free
// ds_read_b128); 54 KiB: two workgroups per CU.
// ------------------------------------------------------------------------------------------
struct GemmSCfg {
    static constexpr int BR = 64, BC = 128, KG = 8, BK = 64;
    static constexpr int LD = BK + 8;                   // 9 groups: odd
    static constexpr int WS = BR * LD, XS = BC * LD;     // halves per plane
    static constexpr size_t lds_bytes = sizeof(hs_t) * 2 * (size_t)(WS + XS);
};

template <bool W_IS_A>
__device__ __forceinline__ void gemm_tile_compute_s(const hs_t* Wh, const hs_t* Wl, const hs_t* Xh, const hs_t* Xl, f32x4 (&acch)[4][2],
                                                    f32x4 (&accl)[4][2], int wave, int l15, int lq) {
    typedef GemmSCfg Gc;
#pragma unroll
    for (int ks = 0; ks < Gc::KG / 4; ++ks) {
        hsx8 fh[4], fl[4], xh[2], xl[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            fh[mi] = lds_hs(Wh + (mi * 16 + l15) * Gc::LD + (ks * 4 + lq) * 8);
            fl[mi] = lds_hs(Wl + (mi * 16 + l15) * Gc::LD + (ks * 4 + lq) * 8);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            xh[ni] = lds_hs(Xh + (wave * 32 + ni * 16 + l15) * Gc::LD + (ks * 4 + lq) * 8);
            xl[ni] = lds_hs(Xl + (wave * 32 + ni * 16 + l15) * Gc::LD + (ks * 4 + lq) * 8);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (W_IS_A) {
                    mma_hs(acch[mi][ni], fh[mi], xh[ni]);
                    mma_hs(accl[mi][ni], fl[mi], xh[ni]);
                    mma_hs(accl[mi][ni], fh[mi], xl[ni]);
                } else {
                    mma_hs(acch[mi][ni], xh[ni], fh[mi]);
                    mma_hs(accl[mi][ni], xh[ni], fl[mi]);
                    mma_hs(accl[mi][ni], xl[ni], fh[mi]);
                }
            }
    }
}

