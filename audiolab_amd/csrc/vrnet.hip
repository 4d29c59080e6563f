// Building blocks of the VR-architecture networks (CascadedASPPNet: reference
// modules/rvc/infer/lib/uvr5_pack/lib_v5/nets*.py, layers*.py) on channels-last fp32 tensors [B, H = bins, W = frames, C].
//
// First HIP path for this family: correct and generic (any channel count, kernel size, stride, dilation), fp32 storage
// and exact-f32 MFMA (v_mfma_f32_16x16x4_f32, a k-ordered fmaf chain) so that it can be pinned against the in-tree torch
// modules; not yet tuned (operands come straight from global memory / L1, no LDS staging).
//   conv2d      : implicit GEMM, one wave = 32 output pixels x 64 output channels, K = (tap, ci) in steps of 4;
//                 epilogue y = act(acc * scale + shift) (BatchNorm folded), written into a channel slice of the output
//                 tensor (so the concatenations of the decoders / ASPP need no copy)
//   depthwise   : 3x3 dilated, one thread per (pixel, channel)
//   resize      : bilinear, align_corners=True (F.interpolate in layers*.py:84, 111), into a channel slice
//   copy_slice  : crop_center along frames (spec_utils.py:12-27) + concat
//   mean_h      : AdaptiveAvgPool2d((1, None)) (layers*.py:93)
//   vr_mask     : sigmoid, replicate pad along bins, aggressiveness powers, * mix (nets*.py:80-111)
#include "alsep_common.h"
#include "mma.h"

namespace {

constexpr int kVrThreads = 256;

__device__ __forceinline__ float vr_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v > 0.f ? v : 0.01f * v;            // nn.LeakyReLU default slope
    if (act == 3) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));   // nn.GELU (exact, erf)
    return v;
}

// x [B,H,W,Cin], w [KH][KW][Cin][Cout] (output channel fastest: the A-fragment loads of 16 lanes are one 64-byte run),
// y [B,Ho,Wo,y_ct] at channel offset y_c0.  One wave = 32 output pixels x 64 output channels: 8 MFMAs per k-step for
// 2 activation + 4 weight loads per lane.
__global__ void __launch_bounds__(kVrThreads)
vr_conv2d_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                 const float* __restrict__ shift, float* __restrict__ y, int64_t npix, int H, int W, int Cin, int Cout,
                 int Ho, int Wo, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int act,
                 int y_ct, int y_c0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int co0 = blockIdx.y * 64;
    int64_t p[2];
    bool pv[2];
    int oy[2], ox[2];
    const float* xb[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        p[n] = ((int64_t)blockIdx.x * 4 + wave) * 32 + n * 16 + l15;     // this lane's output pixels (B columns)
        pv[n] = p[n] < npix;
        const int64_t pp = pv[n] ? p[n] : 0;
        ox[n] = (int)(pp % Wo);
        oy[n] = (int)((pp / Wo) % Ho);
        xb[n] = x + (pp / ((int64_t)Wo * Ho)) * (int64_t)H * W * Cin;
    }
    const int taps = KH * KW;
    f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < taps; ++tap) {
        const float* xp[2];
        bool inb[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int iy = oy[n] * stride_h - pad_h + (tap / KW) * dil_h, ix = ox[n] * stride_w - pad_w + (tap % KW) * dil_w;
            inb[n] = pv[n] && iy >= 0 && iy < H && ix >= 0 && ix < W;
            xp[n] = xb[n] + ((int64_t)(inb[n] ? iy : 0) * W + (inb[n] ? ix : 0)) * Cin;
        }
        const float* wt = w + (int64_t)tap * Cin * Cout;
        for (int c = 0; c < Cin; c += 4) {
            const int ci = c + lq;
            const bool cv = ci < Cin;
            float bv[2], av[4];
#pragma unroll
            for (int n = 0; n < 2; ++n) bv[n] = (inb[n] && cv) ? xp[n][ci] : 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int co = co0 + m * 16 + l15;
                av[m] = (cv && co < Cout) ? wt[(int64_t)ci * Cout + co] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        if (!pv[n]) continue;
        float* yp = y + p[n] * y_ct + y_c0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + m * 16 + 4 * lq + r;         // C/D layout: row = 4*(l>>4)+r, col = l&15
                if (co < Cout) yp[co] = vr_act(fmaf(acc[m][n][r], scale[co], shift[co]), act);
            }
    }
}

// The same convolution as an LDS-tiled implicit GEMM for Cin % 16 == 0 (every layer but the networks' first ones): M = output pixels,
// N = output channels, K = (tap, ci) -- a K-slice of 16 is sixteen consecutive input channels of one tap, i.e. 64 contiguous bytes per
// pixel.  A workgroup computes 128 pixels x 128 channels, four waves as 2 x 2 (64 x 64 each: 16 accumulator blocks); per slice the
// activation rows go global float4 -> LDS [pixel][16 + 8 pad] (conflict-free ds_read_b128) and the weights [16 k][co + 4 pad]
// (ds_read_b32 per k: 16 consecutive channels x two k rows, four apart, fall on disjoint banks); the next slice's global loads are in flight during
// the 64 MFMAs of the current one.  The weights are the MFMA's row operand, so a lane ends up with four consecutive output channels
// of one pixel: float4 epilogue stores into the channel slice.  As nn_gemm_tn_kernel, the four k of one v_mfma_f32_16x16x4_f32 are
// {s, 4 + s, 8 + s, 12 + s} of the slice: another fp32 summation order than vr_conv2d_kernel's (the oracles' tolerances hold for both).
constexpr int kCvBK = 16, kCvLDA = 24;
// WB: 16-row blocks per wave along pixels and along channels (4: the 128 x 128 tile; 2: a 64 x 64 tile for layers whose 128 x 128 grid
// would leave most CUs idle -- the deep levels of the U-Nets: a few hundred pixels x many channels, a long K loop on a dozen workgroups)
template <int WB>
__global__ void __launch_bounds__(kVrThreads)
nn_conv2d_tiled_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                       const float* __restrict__ shift, float* __restrict__ y, int64_t npix, int H, int W, int Cin, int Cout, int Ho, int Wo,
                       int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int act, int y_ct, int y_c0,
                       int vec_store) {
    constexpr int BM = 32 * WB, BN = 32 * WB, LDB = BN + 4;                // lanes lq, lq + 1 read k rows 4 apart: 4 LDB = 16 mod 32 banks
    constexpr int NA = BM * 4 / kVrThreads, NBQ = kCvBK * (BN / 4) / kVrThreads;      // float4 per thread and slice: 2 / 1
    float* As = reinterpret_cast<float*>(alsep_smem);                       // [2][BM][24]
    float* Bs = As + 2 * BM * kCvLDA;                                        // [2][16][LDB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    // staging duty.  A: pixels (tid >> 2) (+ 64), channel quad tid & 3 of the slice; B: k rows tid / (BN / 4) (+ 8), channel quad tid % (BN / 4)
    const int sr = tid >> 2, sq = tid & 3;
    const float* xb[NA];
    int oy[NA], ox[NA];
    bool pv[NA];
#pragma unroll
    for (int h = 0; h < NA; ++h) {
        const int64_t p = m0 + sr + 64 * h;
        pv[h] = p < npix;
        const int64_t pp = pv[h] ? p : 0;
        ox[h] = (int)(pp % Wo) * stride_w - pad_w;
        oy[h] = (int)((pp / Wo) % Ho) * stride_h - pad_h;
        xb[h] = x + (pp / ((int64_t)Wo * Ho)) * (int64_t)H * W * Cin + 4 * sq;
    }
    constexpr int QN = BN / 4;
    const int bk = tid / QN, bq = tid % QN;
    const bool bv = n0 + 4 * bq < Cout;                                      // Cout % 4 == 0 on this path
    const float* wb = w + n0 + 4 * bq;
    f32x4 acc[WB][WB];
#pragma unroll
    for (int i = 0; i < WB; ++i)
#pragma unroll
        for (int j = 0; j < WB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[NA], rb[NBQ];
    auto gload = [&](int k0) {
        const int tap = k0 / Cin, ci0 = k0 - tap * Cin;
        const int dy = (tap / KW) * dil_h, dx = (tap % KW) * dil_w;
#pragma unroll
        for (int h = 0; h < NA; ++h) {
            const int iy = oy[h] + dy, ix = ox[h] + dx;
            const bool in = pv[h] && iy >= 0 && iy < H && ix >= 0 && ix < W;
            ra[h] = in ? *reinterpret_cast<const f32x4*>(xb[h] + ((int64_t)iy * W + ix) * Cin + ci0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int h = 0; h < NBQ; ++h)
            rb[h] = bv ? *reinterpret_cast<const f32x4*>(wb + (int64_t)(k0 + bk + (kCvBK / NBQ) * h) * Cout) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < NA; ++h) *reinterpret_cast<f32x4*>(As + ((size_t)buf * BM + sr + 64 * h) * kCvLDA + 4 * sq) = ra[h];
#pragma unroll
        for (int h = 0; h < NBQ; ++h) *reinterpret_cast<f32x4*>(Bs + ((size_t)buf * kCvBK + bk + (kCvBK / NBQ) * h) * LDB + 4 * bq) = rb[h];
    };
    const int nk = KH * KW * Cin / kCvBK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * kCvBK);
        f32x4 af[WB];
#pragma unroll
        for (int i = 0; i < WB; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + ((size_t)buf * BM + wm * 16 * WB + i * 16 + l15) * kCvLDA + 4 * lq);
        const float* bl = Bs + ((size_t)buf * kCvBK + 4 * lq) * LDB + wn * 16 * WB + l15;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float bf[WB];
#pragma unroll
            for (int j = 0; j < WB; ++j) bf[j] = bl[s4 * LDB + j * 16];
#pragma unroll
            for (int i = 0; i < WB; ++i)
#pragma unroll
                for (int j = 0; j < WB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i][s4], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // D rows = channel (4 lq + r), D columns = pixel (l15)
#pragma unroll
    for (int i = 0; i < WB; ++i) {
        const int64_t p = m0 + wm * 16 * WB + i * 16 + l15;
        if (p >= npix) continue;
        float* yp = y + p * y_ct + y_c0;
#pragma unroll
        for (int j = 0; j < WB; ++j) {
            const int co = n0 + wn * 16 * WB + j * 16 + 4 * lq;
            if (co >= Cout) continue;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = vr_act(fmaf(acc[i][j][r], scale[co + r], shift[co + r]), act);
            if (vec_store) *reinterpret_cast<f32x4*>(yp + co) = v;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) yp[co + r] = v[r];
            }
        }
    }
}

#define ALSEP_NN_F32S_CONV
#include "nn_f32s.h"

// the tiled kernel applies when a K-slice stays inside one tap and the rows are 16-byte aligned
static bool conv_tiled_ok(const float* x, const float* w, int Cin, int Cout, int64_t npix) {
    static const int on = [] { const char* e = getenv("ALSEP_NN_CONV_TILED"); return e ? atoi(e) : 1; }();
    return on && Cin % kCvBK == 0 && Cout % 4 == 0 && npix >= 64 && (((uintptr_t)x | (uintptr_t)w) & 15) == 0;
}
static void launch_conv_tiled(alsep_ctx* ctx, const float* x, const float* w, const float* scale, const float* shift, float* y, int64_t npix,
                              int H, int W, int Cin, int Cout, int Ho, int Wo, int KH, int KW, int stride_h, int stride_w, int pad_h,
                              int pad_w, int dil_h, int dil_w, int act, int y_ct, int y_c0) {
    const int vec = (y_ct % 4 == 0 && y_c0 % 4 == 0 && ((uintptr_t)y & 15) == 0) ? 1 : 0;
    if (ctx->nn_split && ctx->nn_range) {                    // split-half contraction (nn_f32s.h): 128 pixels x 64 channels per workgroup
        (void)hipFuncSetAttribute((const void*)nn_conv2d_split_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GemmSCfg::lds_bytes);
        hipLaunchKernelGGL(nn_conv2d_split_kernel<0>, dim3((unsigned)((Cout + 63) / 64), (unsigned)ceil_div64(npix, 128)), dim3(kNsThreads),
                           GemmSCfg::lds_bytes, ctx->stream, x, w, scale, shift, y, npix, H, W, Cin, Cout, Ho, Wo, KH, KW, stride_h, stride_w, pad_h,
                           pad_w, dil_h, dil_w, act, y_ct, y_c0, vec, ctx->nn_range);
        note_launch(ctx, "nn_conv2d_split_kernel");
        return;
    }
    const int64_t big = ((Cout + 127) / 128) * ceil_div64(npix, 128);
    if (big >= 192) {                                        // enough 128 x 128 tiles for the chip
        const size_t lds = (2 * (size_t)128 * kCvLDA + 2 * (size_t)kCvBK * (128 + 4)) * sizeof(float);
        hipLaunchKernelGGL(nn_conv2d_tiled_kernel<4>, dim3((unsigned)((Cout + 127) / 128), (unsigned)ceil_div64(npix, 128)), dim3(kVrThreads), lds,
                           ctx->stream, x, w, scale, shift, y, npix, H, W, Cin, Cout, Ho, Wo, KH, KW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                           act, y_ct, y_c0, vec);
    } else {
        const size_t lds = (2 * (size_t)64 * kCvLDA + 2 * (size_t)kCvBK * (64 + 4)) * sizeof(float);
        hipLaunchKernelGGL(nn_conv2d_tiled_kernel<2>, dim3((unsigned)((Cout + 63) / 64), (unsigned)ceil_div64(npix, 64)), dim3(kVrThreads), lds,
                           ctx->stream, x, w, scale, shift, y, npix, H, W, Cin, Cout, Ho, Wo, KH, KW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                           act, y_ct, y_c0, vec);
    }
}

// depthwise KHxKW (groups = C), stride 1: x [B,H,W,C], w [C][KH][KW], y [B,H,W,C]
__global__ void __launch_bounds__(kVrThreads)
vr_depthwise_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int64_t n, int H, int W,
                    int C, int KH, int KW, int pad, int dil) {
    const int64_t i = (int64_t)blockIdx.x * kVrThreads + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int ox = (int)(p % W), oy = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    const float* xb = x + b * (int64_t)H * W * C;
    float s = 0.f;
    for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) {
            const int iy = oy - pad + ky * dil, ix = ox - pad + kx * dil;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) s = fmaf(xb[((int64_t)iy * W + ix) * C + c], w[(c * KH + ky) * KW + kx], s);
        }
    y[i] = s;
}

// bilinear resize, align_corners=True: x [B,H,W,C] -> y [B,Ho,Wo,y_ct] channel slice
__global__ void __launch_bounds__(kVrThreads)
vr_resize_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int H, int W, int C, int Ho, int Wo, int y_ct,
                 int y_c0) {
    const int64_t i = (int64_t)blockIdx.x * kVrThreads + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
    const int64_t b = p / ((int64_t)Wo * Ho);
    // torch area_pixel_compute_source_index(align_corners=True): src = dst * (in - 1) / (out - 1)
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* xb = x + b * (int64_t)H * W * C + c;
    const float v00 = xb[((int64_t)y0 * W + x0) * C], v01 = xb[((int64_t)y0 * W + x1) * C];
    const float v10 = xb[((int64_t)y1 * W + x0) * C], v11 = xb[((int64_t)y1 * W + x1) * C];
    // same association as torch's upsample_bilinear2d: h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
    const float hx = 1.f - lx, hy = 1.f - ly;
    y[p * y_ct + y_c0 + c] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

// y[b, h, wq, y_c0 + c] = x[b, h, w_off + wq, c]
__global__ void __launch_bounds__(kVrThreads)
vr_copy_slice_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int Wx, int C, int w_off, int Wy, int y_ct,
                     int y_c0) {
    const int64_t i = (int64_t)blockIdx.x * kVrThreads + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int wq = (int)(p % Wy);
    const int64_t row = p / Wy;                              // (b, h)
    y[p * y_ct + y_c0 + c] = x[(row * Wx + w_off + wq) * C + c];
}

// y[b, 0, w, c] = mean_h x[b, h, w, c]
__global__ void __launch_bounds__(kVrThreads)
vr_mean_h_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int H, int W, int C) {
    const int64_t i = (int64_t)blockIdx.x * kVrThreads + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / ((int64_t)W * C), wc = i % ((int64_t)W * C);
    const float* xb = x + b * (int64_t)H * W * C + wc;
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += xb[(int64_t)h * W * C];
    y[i] = s / (float)H;
}

// out[b,h,w,c] = mask^pow * mix, mask = sigmoid(logit[b, min(h, Hm-1), w, c]); pow = 1 + v/3 below split_bin, 1 + v above
__global__ void __launch_bounds__(kVrThreads)
vr_mask_kernel(const float* __restrict__ logit, const float* __restrict__ mix, float* __restrict__ out, int64_t n, int Hm,
               int Hout, int W, int C, int split_bin, float aggr) {
    const int64_t i = (int64_t)blockIdx.x * kVrThreads + threadIdx.x;
    if (i >= n) return;
    const int64_t wc = i % ((int64_t)W * C);
    const int h = (int)((i / ((int64_t)W * C)) % Hout);
    const int64_t b = i / ((int64_t)W * C * Hout);
    const int hm = h < Hm ? h : Hm - 1;                      // F.pad(mode="replicate") along bins
    float m = 1.f / (1.f + expf(-logit[(b * Hm + hm) * (int64_t)W * C + wc]));
    if (aggr >= 0.f) m = powf(m, h < split_bin ? 1.f + aggr / 3.f : 1.f + aggr);
    out[i] = m * mix[i];
}

// One direction of nn.LSTM (layers_new.py:117-121), recurrent part.  pre [T, N, 4 Hd] holds x_t W_ih^T + b_ih + b_hh (gate
// order i, f, g, o as torch), whh [4 Hd][Hd]; out [T, N, out_stride] receives h_t at column out_off.  One workgroup per
// (n, direction given by `reverse`): thread r < 4 Hd keeps row r of W_hh in registers, h_{t-1} lives in LDS.
template <int HD>
__global__ void __launch_bounds__(4 * HD)
vr_lstm_kernel(const float* __restrict__ pre, const float* __restrict__ whh, float* __restrict__ out, int T, int N, int out_stride,
               int out_off, int reverse) {
    float* hs = reinterpret_cast<float*>(alsep_smem);        // [HD] h_{t-1}
    float* gs = hs + HD;                                     // [4 HD] gate pre-activations of this step
    const int r = threadIdx.x, n = blockIdx.x;
    float wrow[HD];
#pragma unroll
    for (int k = 0; k < HD; ++k) wrow[k] = whh[r * HD + k];
    float c = 0.f;
    if (r < HD) hs[r] = 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        float g = pre[((int64_t)t * N + n) * (4 * HD) + r];
#pragma unroll
        for (int k = 0; k < HD; ++k) g = fmaf(wrow[k], hs[k], g);
        gs[r] = g;
        __syncthreads();
        if (r < HD) {
            const float gi = 1.f / (1.f + expf(-gs[r])), gf = 1.f / (1.f + expf(-gs[HD + r]));
            const float gg = tanhf(gs[2 * HD + r]), go = 1.f / (1.f + expf(-gs[3 * HD + r]));
            c = gf * c + gi * gg;
            const float h = go * tanhf(c);
            hs[r] = h;
            out[((int64_t)t * N + n) * out_stride + out_off + r] = h;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int alsep_vr_lstm(alsep_ctx* ctx, const float* pre, const float* whh, float* out, int T, int N, int hidden,
                             int out_stride, int out_off, int reverse) {
    ALSEP_ENTER(ctx);
    if (!ctx || !pre || !whh || !out) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_lstm: null argument");
    if (T <= 0 || N <= 0 || out_off < 0 || out_off + hidden > out_stride)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_lstm: bad shape");
    const size_t lds = 5 * (size_t)hidden * sizeof(float);
    switch (hidden) {
#define ALSEP_LSTM(HD_)                                                                                                  \
    case HD_:                                                                                                            \
        hipLaunchKernelGGL(vr_lstm_kernel<HD_>, dim3((unsigned)N), dim3(4 * HD_), lds, ctx->stream, pre, whh, out, T, N, \
                           out_stride, out_off, reverse);                                                                \
        break;
        ALSEP_LSTM(16) ALSEP_LSTM(32) ALSEP_LSTM(64)
#undef ALSEP_LSTM
        default: return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_lstm: hidden size %d (have 16, 32, 64)", hidden);
    }
    ALSEP_LAUNCH_CHECK(ctx, "vr_lstm_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_conv2d(alsep_ctx* ctx, const float* x, const float* w, const float* scale, const float* shift, float* y,
                               int64_t B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w,
                               int dil_h, int dil_w, int act, int y_ctotal, int y_coff) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !w || !scale || !shift || !y) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_conv2d: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad_h < 0 || pad_w < 0 ||
        dil_h <= 0 || dil_w <= 0 || act < 0 || act > 2 || y_coff < 0 || y_coff + Cout > y_ctotal)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_conv2d: bad shape");
    const int Ho = (H + 2 * pad_h - dil_h * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad_w - dil_w * (KW - 1) - 1) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_conv2d: empty output");
    const int64_t npix = B * Ho * Wo;
    const int64_t gx = ceil_div64(npix, 128);
    if (gx > 0x7fffffff) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_conv2d: too many pixels");
    if (conv_tiled_ok(x, w, Cin, Cout, npix) && gx <= 65535) {
        launch_conv_tiled(ctx, x, w, scale, shift, y, npix, H, W, Cin, Cout, Ho, Wo, KH, KW, stride, stride, pad_h, pad_w, dil_h, dil_w, act,
                          y_ctotal, y_coff);
        ALSEP_LAUNCH_CHECK(ctx, "nn_conv2d_tiled_kernel");
        return ALSEP_OK;
    }
    hipLaunchKernelGGL(vr_conv2d_kernel, dim3((unsigned)gx, (unsigned)((Cout + 63) / 64)), dim3(kVrThreads), 0, ctx->stream, x, w,
                       scale, shift, y, npix, H, W, Cin, Cout, Ho, Wo, KH, KW, stride, stride, pad_h, pad_w, dil_h, dil_w, act, y_ctotal, y_coff);
    ALSEP_LAUNCH_CHECK(ctx, "vr_conv2d_kernel");
    return ALSEP_OK;
}

// The same kernel with a stride per axis and GELU among the activations: Conv1d / Conv2d([K,1], [S,1]) / 1x1 / 3x3 layers of the
// Demucs family (a Conv1d over [B, L, C] is the W = 1 case; a Linear is the 1x1 case with the rows as pixels).
extern "C" int alsep_nn_conv2d(alsep_ctx* ctx, const float* x, const float* w, const float* scale, const float* shift, float* y,
                               int64_t B, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h, int stride_w, int pad_h,
                               int pad_w, int dil_h, int dil_w, int act, int y_ctotal, int y_coff) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !w || !scale || !shift || !y) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride_h <= 0 || stride_w <= 0 || pad_h < 0 ||
        pad_w < 0 || dil_h <= 0 || dil_w <= 0 || act < 0 || act > 3 || y_coff < 0 || y_coff + Cout > y_ctotal)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d: bad shape");
    const int Ho = (H + 2 * pad_h - dil_h * (KH - 1) - 1) / stride_h + 1, Wo = (W + 2 * pad_w - dil_w * (KW - 1) - 1) / stride_w + 1;
    if (Ho <= 0 || Wo <= 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d: empty output");
    const int64_t npix = B * Ho * Wo;
    const int64_t gx = ceil_div64(npix, 128);
    if (gx > 0x7fffffff) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_nn_conv2d: too many pixels");
    ProfScope prof(ctx, ALSEP_PROF_NN_CONV);
    prof.work(2.0 * (double)npix * Cout * Cin * KH * KW, 4.0 * ((double)B * H * W * Cin + (double)npix * Cout + (double)KH * KW * Cin * Cout));
    if (conv_tiled_ok(x, w, Cin, Cout, npix) && gx <= 65535) {
        launch_conv_tiled(ctx, x, w, scale, shift, y, npix, H, W, Cin, Cout, Ho, Wo, KH, KW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w,
                          act, y_ctotal, y_coff);
        ALSEP_LAUNCH_CHECK(ctx, "nn_conv2d_tiled_kernel");
        return ALSEP_OK;
    }
    hipLaunchKernelGGL(vr_conv2d_kernel, dim3((unsigned)gx, (unsigned)((Cout + 63) / 64)), dim3(kVrThreads), 0, ctx->stream, x, w,
                       scale, shift, y, npix, H, W, Cin, Cout, Ho, Wo, KH, KW, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, act,
                       y_ctotal, y_coff);
    ALSEP_LAUNCH_CHECK(ctx, "nn_conv2d_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_depthwise(alsep_ctx* ctx, const float* x, const float* w, float* y, int64_t B, int H, int W, int C,
                                  int KH, int KW, int pad, int dil) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !w || !y) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_depthwise: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || pad < 0 || dil <= 0 || 2 * pad != dil * (KH - 1) ||
        2 * pad != dil * (KW - 1))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_depthwise: bad shape (same-size output only)");
    const int64_t n = B * H * W * C;
    hipLaunchKernelGGL(vr_depthwise_kernel, dim3((unsigned)ceil_div64(n, kVrThreads)), dim3(kVrThreads), 0, ctx->stream, x, w, y, n,
                       H, W, C, KH, KW, pad, dil);
    ALSEP_LAUNCH_CHECK(ctx, "vr_depthwise_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_resize_bilinear(alsep_ctx* ctx, const float* x, float* y, int64_t B, int H, int W, int C, int Ho, int Wo,
                                        int y_ctotal, int y_coff) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_resize_bilinear: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || y_coff < 0 || y_coff + C > y_ctotal)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_resize_bilinear: bad shape");
    const int64_t n = B * Ho * Wo * C;
    hipLaunchKernelGGL(vr_resize_kernel, dim3((unsigned)ceil_div64(n, kVrThreads)), dim3(kVrThreads), 0, ctx->stream, x, y, n, H, W,
                       C, Ho, Wo, y_ctotal, y_coff);
    ALSEP_LAUNCH_CHECK(ctx, "vr_resize_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_copy_slice(alsep_ctx* ctx, const float* x, float* y, int64_t BH, int Wx, int C, int w_off, int Wy,
                                   int y_ctotal, int y_coff) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_copy_slice: null argument");
    if (BH <= 0 || Wx <= 0 || C <= 0 || Wy <= 0 || w_off < 0 || w_off + Wy > Wx || y_coff < 0 || y_coff + C > y_ctotal)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_copy_slice: bad shape");
    const int64_t n = BH * Wy * C;
    hipLaunchKernelGGL(vr_copy_slice_kernel, dim3((unsigned)ceil_div64(n, kVrThreads)), dim3(kVrThreads), 0, ctx->stream, x, y, n,
                       Wx, C, w_off, Wy, y_ctotal, y_coff);
    ALSEP_LAUNCH_CHECK(ctx, "vr_copy_slice_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_mean_h(alsep_ctx* ctx, const float* x, float* y, int64_t B, int H, int W, int C) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_mean_h: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_mean_h: bad shape");
    const int64_t n = B * W * C;
    hipLaunchKernelGGL(vr_mean_h_kernel, dim3((unsigned)ceil_div64(n, kVrThreads)), dim3(kVrThreads), 0, ctx->stream, x, y, n, H, W,
                       C);
    ALSEP_LAUNCH_CHECK(ctx, "vr_mean_h_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_vr_mask(alsep_ctx* ctx, const float* logit, const float* mix, float* out, int64_t B, int Hm, int Hout, int W,
                             int C, int split_bin, float aggressiveness) {
    ALSEP_ENTER(ctx);
    if (!ctx || !logit || !mix || !out) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_mask: null argument");
    if (B <= 0 || Hm <= 0 || Hout < Hm || W <= 0 || C <= 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_vr_mask: bad shape");
    const int64_t n = B * Hout * W * C;
    hipLaunchKernelGGL(vr_mask_kernel, dim3((unsigned)ceil_div64(n, kVrThreads)), dim3(kVrThreads), 0, ctx->stream, logit, mix, out,
                       n, Hm, Hout, W, C, split_bin, aggressiveness);
    ALSEP_LAUNCH_CHECK(ctx, "vr_mask_kernel");
    return ALSEP_OK;
}
