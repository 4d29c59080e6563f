// Split-half contraction for the GENERIC float32 GEMM / convolution of the HTDemucs, Roformer, MDX23C and VR families (nn.hip, vrnet.hip):
// the same products as nn_gemm_tn_kernel / nn_conv2d_tiled_kernel, float32 in and out, with every operand carried as an IEEE-half pair
// (hi, scaled lo) and three f16 MFMAs per 16 x 16 x 32 block (f32s_common.h; scheme and error budget: tdfnet_f32s.h).  Unlike the TFC-TDF
// network, whose weights are split once on the host, these entry points see plain float32 pointers for BOTH operands, so both are split
// where they are staged into LDS.  Selected per context (alsep_nn_set_contraction); an operand beyond the half range raises the context's
// range word (alsep_nn_range_flag), which the runners read per track and answer by running the track again on the exact kernels.
//
// Tile: 128 rows of the activation-like operand ("X": pixels, or the rows of A) x 64 rows of the weight-like operand ("W": output channels,
// or the rows of B), K tiles of 64; the next K tile's global loads are in flight (registers) during the MFMAs of the current one.
// Reference call sites: the third-party networks behind stem_separator.py:380-383, :466, :541 (see nn.hip / vrnet.hip headers).
#pragma once
#include "f32s_common.h"

constexpr int kNsThreads = 256;

// K-contiguous rows [rows][K] -> LDS planes [rows][LD]: thread (row = tid / 16 + 16 j, quad = tid % 16) of a 64-wide K tile
template <int NR>                                             // rows of the tile: 128 (X) or 64 (W)
struct NsRowStage {
    static constexpr int NJ = NR / 16;
    f32x4 r[NJ];
};

// W operand given as [K][N] (N contiguous: convolution weights [tap, ci][co], the V of P V): 64 k x 64 n per tile, transposed on the way
// into LDS.  Lane map: k pair kp = tid % 16 (rows 2 kp, 2 kp + 1, and + 32 for the second half), n quad = tid / 16: a wave's 4-byte LDS
// stores (two consecutive k of one n) fall on 32 banks at most two deep.
struct NsKnStage {
    f32x4 a[2], b[2];                                        // [half]: rows k, k + 1
};

#ifdef ALSEP_NN_F32S_GEMM       // nn.hip (needs GemmStrides, gelu_erf)
// C[b][m][n] = act(alpha * sum_k A[b][m][k] B[b][n][k] + bias[n]); CT: C row-major with n contiguous (float4 stores); BNN: B is [K][N]
template <bool CT, bool BNN>
__global__ void __launch_bounds__(kNsThreads, 2)
nn_gemm_split_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int nb2, int M, int N, int K,
                     GemmStrides sa, GemmStrides sb, GemmStrides sc, float alpha, const float* __restrict__ bias, int act,
                     unsigned* __restrict__ range_flag) {
    typedef GemmSCfg Gc;
    hs_t* Wh = reinterpret_cast<hs_t*>(alsep_smem);
    hs_t* Wl = Wh + Gc::WS;
    hs_t* Xh = Wl + Gc::WS;
    hs_t* Xl = Xh + Gc::XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int b1 = blockIdx.z / nb2, b2 = blockIdx.z % nb2;
    const float* a = A + b1 * sa.b1 + b2 * sa.b2;
    const float* b = B + b1 * sb.b1 + b2 * sb.b2;
    float* c = C + b1 * sc.b1 + b2 * sc.b2;
    const int m0 = blockIdx.y * Gc::BC, n0 = blockIdx.x * Gc::BR;
    bool bad = false;

    const int sq = tid & 15, sr = tid >> 4;                  // K-contiguous staging: quad, row (+ 16 j)
    const int kp = tid & 15, nq = tid >> 4;                  // [K][N] staging: k pair, n quad
    NsRowStage<Gc::BC> xa;
    NsRowStage<Gc::BR> wb;
    NsKnStage wk;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto cut = [&](f32x4 v, int kq) {                         // zero the elements at k >= K of the quad starting at kq
        if (kq + 4 <= K) return v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (kq + e >= K) v[e] = 0.f;
        return v;
    };
    auto prefetch = [&](int k0) {                             // unconditional loads on clamped addresses; validity is applied when stored
        const int kq = k0 + 4 * sq;
        const int kc = kq < K ? kq : 0;
#pragma unroll
        for (int j = 0; j < Gc::BC / 16; ++j) {
            const int row = m0 + sr + 16 * j;
            xa.r[j] = *reinterpret_cast<const f32x4*>(a + (int64_t)(row < M ? row : 0) * sa.r + kc);
        }
        if (!BNN) {
#pragma unroll
            for (int j = 0; j < Gc::BR / 16; ++j) {
                const int row = n0 + sr + 16 * j;
                wb.r[j] = *reinterpret_cast<const f32x4*>(b + (int64_t)(row < N ? row : 0) * sb.r + kc);
            }
        } else {
            const int nn = n0 + 4 * nq;
            const int nc = nn < N ? nn : 0;                  // N % 4 == 0 on this path
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + 32 * h + 2 * kp;
                wk.a[h] = *reinterpret_cast<const f32x4*>(b + (int64_t)(k < K ? k : 0) * sb.k + nc);
                wk.b[h] = *reinterpret_cast<const f32x4*>(b + (int64_t)(k + 1 < K ? k + 1 : 0) * sb.k + nc);
            }
        }
    };
    auto stage = [&](int k0) {
        const int kq = k0 + 4 * sq;
#pragma unroll
        for (int j = 0; j < Gc::BC / 16; ++j) {
            const bool in = m0 + sr + 16 * j < M && kq < K;
            const f32x4 v = in ? cut(xa.r[j], kq) : z4;
            const float x[4] = {v[0], v[1], v[2], v[3]};
            hsx4 hi, lo;
            split4(x, hi, lo, bad);
            const int o = (sr + 16 * j) * Gc::LD + 4 * sq;
            *reinterpret_cast<hsx4*>(Xh + o) = hi;
            *reinterpret_cast<hsx4*>(Xl + o) = lo;
        }
        if (!BNN) {
#pragma unroll
            for (int j = 0; j < Gc::BR / 16; ++j) {
                const bool in = n0 + sr + 16 * j < N && kq < K;
                const f32x4 v = in ? cut(wb.r[j], kq) : z4;
                const float x[4] = {v[0], v[1], v[2], v[3]};
                hsx4 hi, lo;
                split4(x, hi, lo, bad);
                const int o = (sr + 16 * j) * Gc::LD + 4 * sq;
                *reinterpret_cast<hsx4*>(Wh + o) = hi;
                *reinterpret_cast<hsx4*>(Wl + o) = lo;
            }
        } else {
            const bool nin = n0 + 4 * nq < N;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kk = 32 * h + 2 * kp, k = k0 + kk;
                const bool va = nin && k < K, vb = nin && k + 1 < K;
                const float fa[4] = {va ? wk.a[h][0] : 0.f, va ? wk.a[h][1] : 0.f, va ? wk.a[h][2] : 0.f, va ? wk.a[h][3] : 0.f};
                const float fb[4] = {vb ? wk.b[h][0] : 0.f, vb ? wk.b[h][1] : 0.f, vb ? wk.b[h][2] : 0.f, vb ? wk.b[h][3] : 0.f};
                hsx4 ha, la, hb, lb;
                split4(fa, ha, la, bad);
                split4(fb, hb, lb, bad);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = (4 * nq + i) * Gc::LD + kk;
                    *reinterpret_cast<hsx2*>(Wh + o) = hsx2{ha[i], hb[i]};
                    *reinterpret_cast<hsx2*>(Wl + o) = hsx2{la[i], lb[i]};
                }
            }
        }
    };
    f32x4 acch[4][2], accl[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acch[mi][ni] = accl[mi][ni] = z4;
    prefetch(0);
    for (int k0 = 0; k0 < K; k0 += Gc::BK) {
        __syncthreads();
        stage(k0);
        __syncthreads();
        if (k0 + Gc::BK < K) prefetch(k0 + Gc::BK);
        gemm_tile_compute_s<true>(Wh, Wl, Xh, Xl, acch, accl, wave, l15, lq);
    }
    if (bad) atomicMax(range_flag, 1u);
    // D rows = W rows (n = 4 lq + r), D columns = X rows (m = l15)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int row = m0 + wave * 32 + ni * 16 + l15;
        if (row >= M) continue;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int col = n0 + mi * 16 + 4 * lq;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = alpha * fmaf(accl[mi][ni][r], kSplitInv, acch[mi][ni][r]);
                if (bias && col + r < N) t += bias[col + r];
                if (act == 3) t = gelu_erf(t);
                else if (act == 5) t = tanhf(t);
                v[r] = t;
            }
            if (CT) {
                if (col < N) *reinterpret_cast<f32x4*>(c + (int64_t)row * sc.r + col) = v;     // N % 4 == 0 on this path
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < N) c[(int64_t)row * sc.r + (int64_t)(col + r) * sc.k] = v[r];
            }
        }
    }
}

#endif  // ALSEP_NN_F32S_GEMM

#ifdef ALSEP_NN_F32S_CONV       // vrnet.hip (needs vr_act)
// nn_conv2d_tiled_kernel's contract (Cin % 16 == 0, Cout % 4 == 0): M = output pixels (128 per workgroup), N = output channels (64), K =
// (tap, ci) in slices of 16 consecutive input channels of one tap; a K tile of 64 is four slices, each with its own tap
template <int UNUSED = 0>                                    // (a template so that only the translation unit that launches it emits it)
__global__ void __launch_bounds__(kNsThreads, 2)
nn_conv2d_split_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                       const float* __restrict__ shift, float* __restrict__ y, int64_t npix, int H, int W, int Cin, int Cout, int Ho, int Wo,
                       int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int act, int y_ct, int y_c0,
                       int vec_store, unsigned* __restrict__ range_flag) {
    typedef GemmSCfg Gc;
    hs_t* Wh = reinterpret_cast<hs_t*>(alsep_smem);
    hs_t* Wl = Wh + Gc::WS;
    hs_t* Xh = Wl + Gc::WS;
    hs_t* Xl = Xh + Gc::XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * Gc::BC;
    const int n0 = blockIdx.x * Gc::BR;
    const int Ktot = KH * KW * Cin;
    bool bad = false;
    // pixel staging: thread = (channel quad sq = tid % 4, slice ss = (tid / 4) % 4, pixel sr = tid / 16 (+ 16 j))
    const int sq = tid & 3, ss = (tid >> 2) & 3, sr = tid >> 4;
    int oy[8], ox[8];
    int64_t xb[8];                                           // element offset of the pixel's image (+ channel quad)
    unsigned pv = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t p = m0 + sr + 16 * j;
        const bool in = p < npix;
        const int64_t pp = in ? p : 0;
        ox[j] = (int)(pp % Wo) * stride_w - pad_w;
        oy[j] = (int)((pp / Wo) % Ho) * stride_h - pad_h;
        xb[j] = (pp / ((int64_t)Wo * Ho)) * (int64_t)H * W * Cin + 4 * sq;
        pv |= (unsigned)in << j;
    }
    const int kp = tid & 15, nq = tid >> 4;                  // weight staging: k pair, channel quad
    const bool nin = n0 + 4 * nq < Cout;                     // Cout % 4 == 0 on this path
    const float* wb = w + (nin ? n0 + 4 * nq : 0);
    f32x4 xr[8];
    unsigned xv = 0;                                         // bit j: xr[j] is a real sample (inside the image, inside K)
    NsKnStage wk;
    auto prefetch = [&](int k0) {
        const int ks = k0 + 16 * ss;                          // this thread's slice
        const bool kin = ks < Ktot;
        const int ksc = kin ? ks : 0;
        const int tap = ksc / Cin, ci0 = ksc - tap * Cin;
        const int dy = (tap / KW) * dil_h, dx = (tap % KW) * dil_w;
        xv = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = oy[j] + dy, ix = ox[j] + dx;
            const bool in = kin && ((pv >> j) & 1u) && iy >= 0 && iy < H && ix >= 0 && ix < W;
            xr[j] = *reinterpret_cast<const f32x4*>(x + xb[j] + (in ? ((int64_t)iy * W + ix) * Cin + ci0 : 0));
            xv |= (unsigned)in << j;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + 32 * h + 2 * kp;
            wk.a[h] = *reinterpret_cast<const f32x4*>(wb + (int64_t)(k < Ktot ? k : 0) * Cout);
            wk.b[h] = *reinterpret_cast<const f32x4*>(wb + (int64_t)(k + 1 < Ktot ? k + 1 : 0) * Cout);
        }
    };
    auto stage = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = (xv >> j) & 1u;
            const float v[4] = {in ? xr[j][0] : 0.f, in ? xr[j][1] : 0.f, in ? xr[j][2] : 0.f, in ? xr[j][3] : 0.f};
            hsx4 hi, lo;
            split4(v, hi, lo, bad);
            const int o = (sr + 16 * j) * Gc::LD + 16 * ss + 4 * sq;
            *reinterpret_cast<hsx4*>(Xh + o) = hi;
            *reinterpret_cast<hsx4*>(Xl + o) = lo;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = 32 * h + 2 * kp, k = k0 + kk;
            const bool va = nin && k < Ktot, vb = nin && k + 1 < Ktot;
            const float fa[4] = {va ? wk.a[h][0] : 0.f, va ? wk.a[h][1] : 0.f, va ? wk.a[h][2] : 0.f, va ? wk.a[h][3] : 0.f};
            const float fb[4] = {vb ? wk.b[h][0] : 0.f, vb ? wk.b[h][1] : 0.f, vb ? wk.b[h][2] : 0.f, vb ? wk.b[h][3] : 0.f};
            hsx4 ha, la, hb, lb;
            split4(fa, ha, la, bad);
            split4(fb, hb, lb, bad);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = (4 * nq + i) * Gc::LD + kk;
                *reinterpret_cast<hsx2*>(Wh + o) = hsx2{ha[i], hb[i]};
                *reinterpret_cast<hsx2*>(Wl + o) = hsx2{la[i], lb[i]};
            }
        }
    };
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acch[4][2], accl[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acch[mi][ni] = accl[mi][ni] = z4;
    prefetch(0);
    for (int k0 = 0; k0 < Ktot; k0 += Gc::BK) {
        __syncthreads();
        stage(k0);
        __syncthreads();
        if (k0 + Gc::BK < Ktot) prefetch(k0 + Gc::BK);
        gemm_tile_compute_s<true>(Wh, Wl, Xh, Xl, acch, accl, wave, l15, lq);
    }
    if (bad) atomicMax(range_flag, 1u);
    // D rows = channel (4 lq + r), D columns = pixel (l15)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int64_t p = m0 + wave * 32 + ni * 16 + l15;
        if (p >= npix) continue;
        float* yp = y + p * y_ct + y_c0;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int co = n0 + mi * 16 + 4 * lq;
            if (co >= Cout) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = vr_act(fmaf(fmaf(accl[mi][ni][r], kSplitInv, acch[mi][ni][r]), scale[co + r], shift[co + r]), act);
            if (vec_store) *reinterpret_cast<f32x4*>(yp + co) = v;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) yp[co + r] = v[r];
            }
        }
    }
}
#endif  // ALSEP_NN_F32S_CONV
