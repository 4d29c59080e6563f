// Three-pass STFT for n_fft = 256 * R2 (4096 = 16*16*16, 6144 = 16*16*24, 7680 = 16*16*30): the production front end.
//
// Semantics: ConvTDFNetTrim.stft, reference modules/rvc/infer/modules/uvr5/mdxnet.py:41-56 (same as
// stft_kernel in fft.hip, which stays as the generic-size kernel).
//
// Why three passes: the generic kernel is bound by instruction issue (~96 instructions per point) and by
// LDS stores (the slowest LDS operation on gfx950, MI355X_MICROARCH.md section LDS); with radices 16, 16, R2
//   * pass A takes its inputs straight from global memory (PCM x window) and pass C leaves its outputs in
//     registers, so the frame crosses LDS twice instead of six times;
//   * pass C gives thread j the butterflies k = j and k = 256 - j, i.e. both Z[k] and Z[N-k] of the
//     two-for-one split XL[k] = (Z[k] + conj Z[N-k]) / 2, XR[k] = (Z[k] - conj Z[N-k]) / 2i, so the split needs
//     no further exchange (thread 0 holds the two self-paired butterflies 0 and 128);
//   * all complex arithmetic is packed (v_pk_*_f32 on (re, im) pairs, modifiers spelled out in
//     alsep_gfx950_asm.h), the 1/2 of the split is folded into the window.
// One workgroup = one frame of both channels = 128 threads (2 waves), LDS = n_fft * 8 bytes.
#pragma once

#include <type_traits>

namespace r16 {

constexpr int kThreads = 128;
// LDS bytes of stft_r16_kernel<R2>: the plain frame (n_fft values) or, when the pass-A rows are padded (R2 = 30), 18 per row
template <int R2> constexpr int stft_lds_bytes() { return ((R2 % 8 == 0) ? 256 * R2 : 16 * R2 * 18) * 8; }

__device__ __forceinline__ v2f mk(float x, float y) { v2f r = {x, y}; return r; }

// u[r] *= w[r] for r = FIRST..R-1, first halves of all products before the second halves (see cx_mul_p1)
template <int R, int FIRST = 1> __device__ __forceinline__ void cx_mul_n(v2f (&u)[R], const v2f (&w)[R]) {
    v2f t[R];
#pragma unroll
    for (int r = FIRST; r < R; ++r) t[r] = cx_mul_p1(u[r], w[r]);
#pragma unroll
    for (int r = FIRST; r < R; ++r) u[r] = cx_mul_p2(u[r], w[r], t[r]);
}

// forward DFT-4 in place: (a, b, c, d) -> (X0, X1, X2, X3)
__device__ __forceinline__ void dft4(v2f& a, v2f& b, v2f& c, v2f& d) {
    const v2f t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    a = t0 + t2;
    c = t0 - t2;
    b = cx_add_mi(t1, t3);
    d = cx_add_pi(t1, t3);
}
// the same with input c already owed a factor -i (c' = -i c)
__device__ __forceinline__ void dft4_c_negi(v2f& a, v2f& b, v2f& c, v2f& d) {
    const v2f t0 = cx_add_mi(a, c), t1 = cx_add_pi(a, c), t2 = b + d, t3 = b - d;
    a = t0 + t2;
    c = t0 - t2;
    b = cx_add_mi(t1, t3);
    d = cx_add_pi(t1, t3);
}

// W_16^m = exp(-2 pi i m / 16)
#define R16_C1 0.92387953251128675613f
#define R16_S1 0.38268343236508977173f
#define R16_H 0.70710678118654752440f

// Forward DFT-16 in place, 4 x 4 Cooley-Tukey; output X[q] ends up in u[(q & 3) * 4 + (q >> 2)].
__device__ __forceinline__ void dft16(v2f (&u)[16]) {
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) dft4(u[r2], u[4 + r2], u[8 + r2], u[12 + r2]);   // -> A[r2][q1] at u[4 q1 + r2]
    const v2f w1 = mk(R16_C1, -R16_S1), w2 = mk(R16_H, -R16_H), w3 = mk(R16_S1, -R16_C1);
    const v2f w6 = mk(-R16_H, -R16_H), w9 = mk(-R16_C1, R16_S1);
    // q1 = 1: r2 = 1, 2, 3 -> W^1, W^2, W^3;  q1 = 2: W^2, W^4 (= -i, folded below), W^6;  q1 = 3: W^3, W^6, W^9
    const v2f t5 = cx_mul_p1(u[5], w1), t6 = cx_mul_p1(u[6], w2), t7 = cx_mul_p1(u[7], w3), t9 = cx_mul_p1(u[9], w2);
    const v2f t11 = cx_mul_p1(u[11], w6), t13 = cx_mul_p1(u[13], w3), t14 = cx_mul_p1(u[14], w6), t15 = cx_mul_p1(u[15], w9);
    u[5] = cx_mul_p2(u[5], w1, t5);
    u[6] = cx_mul_p2(u[6], w2, t6);
    u[7] = cx_mul_p2(u[7], w3, t7);
    u[9] = cx_mul_p2(u[9], w2, t9);
    u[11] = cx_mul_p2(u[11], w6, t11);
    u[13] = cx_mul_p2(u[13], w3, t13);
    u[14] = cx_mul_p2(u[14], w6, t14);
    u[15] = cx_mul_p2(u[15], w9, t15);
    dft4(u[0], u[1], u[2], u[3]);
    dft4(u[4], u[5], u[6], u[7]);
    dft4_c_negi(u[8], u[9], u[10], u[11]);
    dft4(u[12], u[13], u[14], u[15]);
}
__device__ __forceinline__ constexpr int dft16_slot(int q) { return (q & 3) * 4 + (q >> 2); }

// forward DFT-8 in place, natural order
__device__ __forceinline__ void dft8(v2f& u0, v2f& u1, v2f& u2, v2f& u3, v2f& u4, v2f& u5, v2f& u6, v2f& u7) {
    dft4(u0, u2, u4, u6);                  // e0..e3 in u0, u2, u4, u6
    dft4(u1, u3, u5, u7);                  // o0..o3 in u1, u3, u5, u7
    const v2f c1 = mk(R16_H, -R16_H), c3 = mk(-R16_H, -R16_H);
    const v2f p1 = cx_mul_p1(u3, c1), p3 = cx_mul_p1(u7, c3);
    const v2f o1 = cx_mul_p2(u3, c1, p1), o3 = cx_mul_p2(u7, c3, p3);
    const v2f e0 = u0, e1 = u2, e2 = u4, e3 = u6, o0 = u1, o2 = u5;
    u0 = e0 + o0;  u4 = e0 - o0;
    u1 = e1 + o1;  u5 = e1 - o1;
    u2 = cx_add_mi(e2, o2);  u6 = cx_add_pi(e2, o2);
    u3 = e3 + o3;  u7 = e3 - o3;
}
// forward DFT-3 in place
__device__ __forceinline__ void dft3(v2f& u0, v2f& u1, v2f& u2) {
    const v2f s = mk(0.86602540378443864676f, 0.86602540378443864676f);
    const v2f t1 = u1 + u2, d = u1 - u2;
    const v2f t2 = u0 - 0.5f * t1;
    u0 = u0 + t1;
    u1 = cx_fma_mi(d, s, t2);
    u2 = cx_fma_pi(d, s, t2);
}

// Last-pass DFT of size R2 in place; output X[q] ends up in u[LastDft<R2>::slot(q)].
template <int R2> struct LastDft;
template <> struct LastDft<16> {
    static __device__ __forceinline__ void run(v2f (&u)[16]) { dft16(u); }
    static __device__ __forceinline__ constexpr int slot(int q) { return dft16_slot(q); }
};
// 24 = 3 x 8 by the prime-factor map (no inner twiddles): input r = (8 n1 + 3 n2) mod 24, output
// q = (16 k1 + 9 k2) mod 24; DFT-8 over n2 then DFT-3 over n1, both in place.
template <> struct LastDft<24> {
    static __device__ __forceinline__ constexpr int in_idx(int n1, int n2) { return (8 * n1 + 3 * n2) % 24; }
    static __device__ __forceinline__ void run(v2f (&u)[24]) {
#pragma unroll
        for (int n1 = 0; n1 < 3; ++n1)
            dft8(u[in_idx(n1, 0)], u[in_idx(n1, 1)], u[in_idx(n1, 2)], u[in_idx(n1, 3)], u[in_idx(n1, 4)],
                 u[in_idx(n1, 5)], u[in_idx(n1, 6)], u[in_idx(n1, 7)]);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) dft3(u[in_idx(0, k2)], u[in_idx(1, k2)], u[in_idx(2, k2)]);
    }
    // X[(16 k1 + 9 k2) % 24] sits where input (n1 = k1, n2 = k2) sat; k1 = q mod 3, k2 = q mod 8 (CRT)
    static __device__ __forceinline__ constexpr int slot(int q) { return in_idx(q % 3, q % 8); }
};

// forward DFT-5 in place
__device__ __forceinline__ void dft5(v2f& u0, v2f& u1, v2f& u2, v2f& u3, v2f& u4) {
    const float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;
    const float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;
    const v2f t1 = u1 + u4, t2 = u2 + u3, t3 = u1 - u4, t4 = u2 - u3;
    const v2f m1 = u0 + C1 * t1 + C2 * t2, m2 = u0 + C2 * t1 + C1 * t2;
    const v2f n1 = S1 * t3 + S2 * t4, n2 = S2 * t3 - S1 * t4;
    u0 = u0 + t1 + t2;
    u1 = cx_add_mi(m1, n1);                // m1 - i n1
    u4 = cx_add_pi(m1, n1);                // m1 + i n1
    u2 = cx_add_mi(m2, n2);
    u3 = cx_add_pi(m2, n2);
}
// forward DFT-6 in place by the prime-factor map 2 x 3: X[m] ends up in v[dft6_pos(m)]
__device__ __forceinline__ constexpr int dft6_in(int a, int b) { return (3 * a + 2 * b) % 6; }
__device__ __forceinline__ constexpr int dft6_pos(int m) { return dft6_in(m % 2, m % 3); }
__device__ __forceinline__ void dft6(v2f& v0, v2f& v1, v2f& v2, v2f& v3, v2f& v4, v2f& v5) {
    v2f* v[6] = {&v0, &v1, &v2, &v3, &v4, &v5};
#pragma unroll
    for (int b = 0; b < 3; ++b) {                            // DFT-2 over a
        const v2f p = *v[dft6_in(0, b)], q = *v[dft6_in(1, b)];
        *v[dft6_in(0, b)] = p + q;
        *v[dft6_in(1, b)] = p - q;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) dft3(*v[dft6_in(a, 0)], *v[dft6_in(a, 1)], *v[dft6_in(a, 2)]);
}
// 30 = 5 x 6 by the prime-factor map: input r = (6 n1 + 5 n2) mod 30, output q = (6 k1 + 25 k2) mod 30 (k1 = q mod 5,
// k2 = q mod 6); DFT-5 over n1, then DFT-6 (itself 2 x 3) over n2, all in place.
template <> struct LastDft<30> {
    static __device__ __forceinline__ constexpr int in_idx(int n1, int n2) { return (6 * n1 + 5 * n2) % 30; }
    static __device__ __forceinline__ void run(v2f (&u)[30]) {
#pragma unroll
        for (int n2 = 0; n2 < 6; ++n2)
            dft5(u[in_idx(0, n2)], u[in_idx(1, n2)], u[in_idx(2, n2)], u[in_idx(3, n2)], u[in_idx(4, n2)]);
#pragma unroll
        for (int k1 = 0; k1 < 5; ++k1)
            dft6(u[in_idx(k1, 0)], u[in_idx(k1, 1)], u[in_idx(k1, 2)], u[in_idx(k1, 3)], u[in_idx(k1, 4)], u[in_idx(k1, 5)]);
    }
    static __device__ __forceinline__ constexpr int slot(int q) { return in_idx(q % 5, dft6_pos(q % 6)); }
};

// w[r] = w1^r for r = 1..R-1 by doubling: level by level, w[p + j] = w[p] w[j] (j < p) and w[2p] = w[p]^2 for
// p = 1, 2, 4, ...; the products of one level are independent (first halves batched before second halves) and every
// power is at most ceil(log2 R) products away from w1.
template <int R> __device__ __forceinline__ void twiddle_powers(v2f w1, v2f (&w)[R]) {
    w[1] = w1;
#pragma unroll
    for (int p = 1; p < R; p *= 2) {
        v2f t[32];
#pragma unroll
        for (int j = 1; j <= p; ++j)
            if (p + j < R) t[j] = cx_mul_p1(w[p], w[j]);
#pragma unroll
        for (int j = 1; j <= p; ++j)
            if (p + j < R) w[p + j] = cx_mul_p2(w[p], w[j], t[j]);
    }
}

template <typename OutT> __device__ __forceinline__ void store_bin(OutT* spec, int64_t frame_off, int64_t b, int k,
                                                                     int T, int t, int dim_f, int layout, v2f L, v2f R);
template <> __device__ __forceinline__ void store_bin<float>(float* spec, int64_t frame_off, int64_t b, int k, int T,
                                                             int t, int dim_f, int layout, v2f L, v2f R) {
    if (layout == ALSEP_LAYOUT_NHWC) {
        f32x4 v = {L.x, L.y, R.x, R.y};
        *reinterpret_cast<f32x4*>(spec + (frame_off + k) * 4) = v;
    } else {
        const int64_t plane = (int64_t)dim_f * T;
        float* o = spec + b * 4 * plane + (int64_t)k * T + t;
        o[0] = L.x; o[plane] = L.y; o[2 * plane] = R.x; o[3 * plane] = R.y;
    }
}
template <> __device__ __forceinline__ void store_bin<bf16_t>(bf16_t* spec, int64_t frame_off, int64_t b, int k, int T,
                                                              int t, int dim_f, int layout, v2f L, v2f R) {
    if (layout == ALSEP_LAYOUT_NHWC) {
        bf16x4 v;
        v[0] = (bf16_t)L.x; v[1] = (bf16_t)L.y; v[2] = (bf16_t)R.x; v[3] = (bf16_t)R.y;
        *reinterpret_cast<bf16x4*>(spec + (frame_off + k) * 4) = v;
    } else {
        const int64_t plane = (int64_t)dim_f * T;
        bf16_t* o = spec + b * 4 * plane + (int64_t)k * T + t;
        o[0] = (bf16_t)L.x; o[plane] = (bf16_t)L.y; o[2 * plane] = (bf16_t)R.x; o[3 * plane] = (bf16_t)R.y;
    }
}

// Fused first layer (FUSE): the first 1x1 convolution of the TFC-TDF U-Net (4 spectrogram channels -> g = 48, folded BatchNorm, ReLU;
// first_conv_kernel in tdfnet.hip) applied in the epilogue, so that the kernel writes the network's level-0 activation [B][T][F][48]
// instead of the spectrogram: the spectrogram's HBM round trip (8 bytes per bin written, read back by first_conv_kernel) and that
// kernel's launch disappear.
// Bit-identical to stft + first_conv: the bin values are rounded to the storage type exactly where the spectrogram store rounded them,
// each output is the same chain w.x x0, fma(w.y, x1), fma(w.z, x2), fma(w.w, x3), fma(., scale in_scale, shift), and the ReLU is taken on
// the rounded value's 16-bit pattern (signed max with 0: rounding keeps the sign, and -0 becomes +0 as fmaxf(-0, 0) does).
// Work split of the epilogue: once every thread holds its pass-C inputs the frame buffer is dead and receives the frame's rounded bins
// ([dim_f] x 8 bytes); the store phase then writes the frame's activation row in whole 128-byte lines, 2048 bytes per round (see (2) in
// the kernel).  (Earlier forms -- every thread computing all 48 channels of its own bins, weights re-read per bin pair from LDS or
// through scalar loads -- were bound by exactly those re-reads: 1.0-1.3 ms per launch of 52 chunks; one fixed channel group per thread
// with 126 active lanes wrote rounds of 2016 bytes that split lines between store instructions: 0.90 ms.)
struct FirstConvArgs {
    const float* w;        // [48][4]
    const float* scale;    // [48]
    const float* shift;    // [48]
    float in_scale;
    int zero_low;          // bins below this index enter the network as zeros (the overlap-add runner's zero_low_bins)
};
constexpr int kFirstConvG = 48;
constexpr int kFirstConvGroups = kFirstConvG / 8;            // 16-byte pieces per bin

// grid (T, n_chunks), 128 threads.
template <int R2, typename OutT, int LAYOUT, bool FUSE = false>
__global__ void __launch_bounds__(kThreads) ALSEP_WAVES_PER_EU(R2 > 24 ? 1 : 2)      // 7680: LDS allows two workgroups (one wave per SIMD) anyway
stft_r16_kernel(const float* __restrict__ pcm, int64_t ch_stride, int64_t chunk_stride, int chunk, int hop, int dim_f,
                int T, const float2* __restrict__ tw_, OutT* __restrict__ spec, FirstConvArgs fc, int n_frames = 0) {
    constexpr int N = 256 * R2, NT = kThreads;
    constexpr int M = N / 16;                 // butterflies of passes A and B
    constexpr int NB = (M + NT - 1) / NT;     // per thread (2, 3; 4 for R2 = 30, where the last threads own fewer)
    constexpr bool FULL = M % NT == 0;
    // pass-A image: rows of 16 values.  With (M/16) % 8 == 0 the XOR key of a row survives pass B's row stride and the
    // rows stay 16 wide (swizzled granules); otherwise (7680: M/16 = 30) rows are padded to 18 values instead
    constexpr bool SWZ = (M / 16) % 8 == 0;
    constexpr int RW = SWZ ? 16 : 18;
    static_assert(NB <= 4, "geometry");
    const v2f* __restrict__ tw = reinterpret_cast<const v2f*>(tw_);
    v2f* buf = reinterpret_cast<v2f*>(alsep_smem);
    // consecutive workgroup ids are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of frames so
    // that the 6-fold re-read of every PCM sample (hop = n_fft / 6) is served by ONE L2 instead of all eight.
    // PERSIST (the fused front end at 7680 points; n_frames = its frame count): a 1-D grid of as many workgroups as fit the chip at once (a multiple of 8), each walking
    // frames id, id + grid, ... of the same XCD's run -- a workgroup's 64-deep queue of epilogue stores then drains under its NEXT
    // frame's loads and passes instead of holding its LDS until the last store has left (7680: the frame buffer allows two workgroups
    // per CU, and a quarter of the launch was that wait).
    // Only where it pays (PERSIST): at 6144 points three workgroups per CU already hide a frame's FFT under the others' stores, and the
    // frame loop costs the kernel 12 % (740 -> 830 us); at 7680 (two per CU) it takes 1000 -> 844 us.
    constexpr bool PERSIST = FUSE && R2 > 24;
    const int n_wg = gridDim.x * gridDim.y, total = PERSIST ? n_frames : n_wg;
    for (int id = blockIdx.x + gridDim.x * blockIdx.y, it = 0; PERSIST ? id < total : it < 1; id += n_wg, ++it) {
    // everything derived from the thread index is computed afresh per frame (an opaque copy): hoisted out of the frame loop these values
    // stay live across it, and the kernel has no register to spare (it spilled 130-320 dwords with them hoisted)
    const int tid = PERSIST ? opaque_vgpr((int)threadIdx.x) : (int)threadIdx.x;
    const int wg = xcd_remap(id, total);
    const int t = wg % T;
    const int64_t b = wg / T;
    const int p0 = t * hop - N / 2;
    const float* xl = pcm + b * chunk_stride;
    const float* xr = xl + ch_stride;
    // first-order twiddles of passes B and C, fetched now: behind a barrier each would expose one memory latency
    const int ka = tid, kb = tid ? 256 - tid : 128;
    const v2f wB1 = tw[(tid & 15) * (N / 256)], wa1 = tw[ka], wb1 = tw[kb];
    // ---- pass A: radix 16, P = 1.  Butterfly i takes x[i + M r] * w[i + M r]; writes row i (16 values).
    // Thread tid owns the NB consecutive butterflies i = NB tid + bb: its inputs for one r are NB consecutive samples
    // (one 8- / 12-byte load per channel; 32 loads in flight per thread instead of 96, all under the 63-deep vmcnt).
    // The window is not loaded: 0.5 w[n] = 1/4 - 1/4 cos(2 pi n / N) and cos(2 pi (i + M r) / N) = Re(W_N^i W_16^r),
    // W_N^i from the twiddle table (NB loads), W_16^r literals: two FMAs per sample.
    // LDS image after pass A: row i at i*16, its 16-byte granule g (values 2g, 2g+1) at position g ^ (i & 7):
    // conflict-free ds_write_b128 here, conflict-free ds_read_b64 in pass B.
    {
        v2f u[NB][16];
        v2f wi[NB];
        const bool own = FULL || NB * tid < M;                   // 7680: threads 120..127 own no pass-A butterfly
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) wi[bb] = own ? tw[NB * tid + bb] : mk(1.f, 0.f);
        if (!own) {
#pragma unroll
            for (int bb = 0; bb < NB; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) u[bb][r] = mk(0.f, 0.f);
        } else if (p0 >= 0 && p0 + N <= chunk) {                 // interior frame (wave-uniform)
            // uniform base (SGPR pair) + 32-bit lane offset: no per-load 64-bit address arithmetic
            const unsigned voff = (unsigned)tid * (4u * NB);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float l[NB], rr[NB];
                load_floats<NB>(reinterpret_cast<const char*>(xl + p0 + M * r) + voff, l);
                load_floats<NB>(reinterpret_cast<const char*>(xr + p0 + M * r) + voff, rr);
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) u[bb][r] = mk(l[bb], rr[bb]);
            }
        } else {                                                 // reflect padding (center=True)
#pragma unroll
            for (int bb = 0; bb < NB; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int p = p0 + NB * tid + bb + M * r;
                    if (p < 0) p = -p;
                    if (p >= chunk) p = 2 * (chunk - 1) - p;
                    u[bb][r] = mk(xl[p], xr[p]);
                }
        }
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    constexpr double kA = 6.283185307179586476925286766559 / 16.0;
                    const float cr = (float)(-0.25 * __builtin_cos(kA * r)), sr = (float)(-0.25 * __builtin_sin(kA * r));
                    // cos(a + b) = cos a cos b - sin a sin b, wi = (cos a, -sin a)
                    const float w = fmaf(wi[bb].x, cr, fmaf(wi[bb].y, sr, 0.25f));
                    u[bb][r] *= w;
                }
            }
            dft16(u[bb]);
            const int i = NB * tid + bb;
            f32x4* row = reinterpret_cast<f32x4*>(buf + i * RW);
            if (own) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const v2f lo = u[bb][dft16_slot(2 * g)], hi = u[bb][dft16_slot(2 * g + 1)];
                    f32x4 v = {lo.x, lo.y, hi.x, hi.y};
                    row[SWZ ? (g ^ (i & 7)) : g] = v;
                }
            }
        }
    }
    __syncthreads();

    // ---- pass B: radix 16, P = 16.  Butterfly i (k = i & 15) takes logical buf[i + M r] * W_256^(k r) and
    // writes logical (i >> 4) * 256 + 16 q + k (plain layout from here on).
    {
        v2f u[NB][16];
        const int k = tid & 15;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int i = tid + NT * bb;
            const int row = i >> 4;                              // SWZ: (row + (M/16) r) & 7 == row & 7
            const v2f* src = buf + row * RW + (SWZ ? ((((k >> 1) ^ (row & 7)) << 1) | (k & 1)) : k);
            if (FULL || i < M) {
#pragma unroll
                for (int r = 0; r < 16; ++r) u[bb][r] = src[(M / 16) * RW * r];
            }
        }
        v2f w[16];
        twiddle_powers<16>(wB1, w);
        __syncthreads();                           // every read of the pass-A image is done
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int i = tid + NT * bb;
            if (FULL || i < M) {
                cx_mul_n<16>(u[bb], w);
                dft16(u[bb]);
                v2f* dst = buf + (i >> 4) * 256 + k;
#pragma unroll
                for (int q = 0; q < 16; ++q) dst[16 * q] = u[bb][dft16_slot(q)];
            }
        }
    }
    __syncthreads();

    // ---- pass C: radix R2, P = 256.  Thread j: butterflies ka = j and kb = 256 - j (thread 0: 0 and 128).
    {
        v2f za[R2], zb[R2];
#pragma unroll
        for (int r = 0; r < R2; ++r) za[r] = buf[ka + 256 * r];
#pragma unroll
        for (int r = 0; r < R2; ++r) zb[r] = buf[kb + 256 * r];
        if constexpr (FUSE) __syncthreads();                     // the frame buffer is reused for the rounded bins below
        {
            v2f w[R2];
            twiddle_powers<R2>(wa1, w);
            cx_mul_n<R2>(za, w);
            LastDft<R2>::run(za);
        }
        {
            v2f w[R2];
            twiddle_powers<R2>(wb1, w);
            cx_mul_n<R2>(zb, w);
            LastDft<R2>::run(zb);
        }
        // Two-for-one split.  Bin k1 = ka + 256 q pairs with N - k1 = kb + 256 (R2-1-q)  [thread 0: 256 (R2-q)],
        // bin k2 = kb + 256 q with N - k2 = ka + 256 (R2-1-q)                                [thread 0: kb + ...].
        const bool t0 = tid == 0;
        const int64_t frame_off = (b * T + t) * (int64_t)dim_f;
        auto emit = [&](auto full) {
#pragma unroll
            for (int q = 0; q < R2 / 2; ++q) {
                const v2f a_q = za[LastDft<R2>::slot(q)], b_q = zb[LastDft<R2>::slot(q)];
                const v2f a_m = za[LastDft<R2>::slot(R2 - 1 - q)], b_m = zb[LastDft<R2>::slot(R2 - 1 - q)];
                const v2f a_w = za[LastDft<R2>::slot((R2 - q) % R2)];
                const v2f n1 = t0 ? a_w : b_m;
                const v2f n2 = t0 ? b_m : a_m;
                const int k1 = ka + 256 * q, k2 = kb + 256 * q;
                if ((decltype(full)::value || k1 < dim_f))
                    store_bin<OutT>(spec, frame_off, b, k1, T, t, dim_f, LAYOUT, cx_add_conj(a_q, n1), cx_sub_conj_divi(a_q, n1));
                if ((decltype(full)::value || k2 < dim_f))
                    store_bin<OutT>(spec, frame_off, b, k2, T, t, dim_f, LAYOUT, cx_add_conj(b_q, n2), cx_sub_conj_divi(b_q, n2));
            }
        };
        if constexpr (FUSE) {
            typedef OutT x4_t __attribute__((ext_vector_type(4)));
            typedef OutT out8_t __attribute__((ext_vector_type(8)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            // (1) the frame's bins, rounded, into the dead frame buffer (the barrier after the za / zb loads made it free)
            x4_t* xs = reinterpret_cast<x4_t*>(alsep_smem);
#pragma unroll
            for (int q = 0; q < R2 / 2; ++q) {
                const v2f a_q = za[LastDft<R2>::slot(q)], b_q = zb[LastDft<R2>::slot(q)];
                const v2f a_m = za[LastDft<R2>::slot(R2 - 1 - q)], b_m = zb[LastDft<R2>::slot(R2 - 1 - q)];
                const v2f a_w = za[LastDft<R2>::slot((R2 - q) % R2)];
                const v2f n1 = t0 ? a_w : b_m;
                const v2f n2 = t0 ? b_m : a_m;
                const int k1 = ka + 256 * q, k2 = kb + 256 * q;
                const v2f l1 = cx_add_conj(a_q, n1), r1 = cx_sub_conj_divi(a_q, n1), l2 = cx_add_conj(b_q, n2), r2 = cx_sub_conj_divi(b_q, n2);
                const bool z1 = k1 < fc.zero_low, z2 = k2 < fc.zero_low;
                x4_t v1, v2;
                v1[0] = (OutT)(z1 ? 0.f : l1.x); v1[1] = (OutT)(z1 ? 0.f : l1.y); v1[2] = (OutT)(z1 ? 0.f : r1.x); v1[3] = (OutT)(z1 ? 0.f : r1.y);
                v2[0] = (OutT)(z2 ? 0.f : l2.x); v2[1] = (OutT)(z2 ? 0.f : l2.y); v2[2] = (OutT)(z2 ? 0.f : r2.x); v2[3] = (OutT)(z2 ? 0.f : r2.y);
                if (k1 < dim_f) xs[k1] = v1;
                if (k2 < dim_f) xs[k2] = v2;
            }
            // (2) the store phase walks the frame's activation row (dim_f x 96 bytes) LINEARLY, 128 pieces of 16 bytes = 2048 bytes = 16 whole
            // 128-byte lines per round: piece 128 r + tid = (bin, channel group) = (piece / 6, piece % 6).  128 = 2 mod 6, so a thread's
            // channel group cycles with period 3 in r: it keeps the weights of its three groups (tid % 6 + 2 s) % 6 in registers (144
            // values -- the FFT's registers are dead by now) and a super-round of three rounds covers 64 bins.  (The earlier form gave a
            // thread ONE group and 126 active lanes, i.e. rounds of 2016 bytes whose ends split a line between two store instructions:
            // a bare fill in that shape runs at 4.15 TB/s on this box against 5.55 TB/s for whole lines, profiles/r04_fill_bench.txt.)
            v2f wx[3][4], wy[3][4], wz[3][4], ww[3][4], sc[3][4], sh[3][4];
            int bl[3];
#pragma unroll
            for (int sr = 0; sr < 3; ++sr) {
                const int j = kThreads * sr + tid;
                const int cg = j % kFirstConvGroups;
                bl[sr] = j / kFirstConvGroups;
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const int c = 8 * cg + 2 * pr;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(fc.w + 4 * c), w1 = *reinterpret_cast<const f32x4*>(fc.w + 4 * c + 4);
                    wx[sr][pr] = mk(w0[0], w1[0]); wy[sr][pr] = mk(w0[1], w1[1]); wz[sr][pr] = mk(w0[2], w1[2]); ww[sr][pr] = mk(w0[3], w1[3]);
                    sc[sr][pr] = mk(fc.scale[c] * fc.in_scale, fc.scale[c + 1] * fc.in_scale);
                    sh[sr][pr] = mk(fc.shift[c], fc.shift[c + 1]);
                }
            }
            __syncthreads();
            char* dst = reinterpret_cast<char*>(spec + frame_off * kFirstConvG) + 16 * tid;
            typedef OutT o2_t __attribute__((ext_vector_type(2)));
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            auto one = [&](x4_t x, auto srt) {
                constexpr int sr = decltype(srt)::value;
                const float x0 = (float)x[0], x1 = (float)x[1], x2 = (float)x[2], x3 = (float)x[3];
                s16x8 o;
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    v2f a = wx[sr][pr] * x0;
                    a = __builtin_elementwise_fma(wy[sr][pr], mk(x1, x1), a);
                    a = __builtin_elementwise_fma(wz[sr][pr], mk(x2, x2), a);
                    a = __builtin_elementwise_fma(ww[sr][pr], mk(x3, x3), a);
                    a = __builtin_elementwise_fma(a, sc[sr][pr], sh[sr][pr]);
                    const s16x2 h = __builtin_bit_cast(s16x2, __builtin_convertvector(a, o2_t));    // one packed convert per pair
                    o[2 * pr] = h[0];
                    o[2 * pr + 1] = h[1];
                }
                const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                return __builtin_elementwise_max(o, zero);
            };
            constexpr int kRoundBytes = kThreads * 16;                                // 2048
            constexpr int kSuperBins = 3 * kThreads / kFirstConvGroups;               // 64 bins per super-round
            const int supers = dim_f / kSuperBins;                                    // whole super-rounds: every bin below dim_f
            int I = 0;
            for (; I + 2 <= supers; I += 2) {                                         // six rounds in flight
                x4_t x[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) x[u] = xs[kSuperBins * (I + u / 3) + bl[u % 3]];
                char* d = dst + (int64_t)(3 * kRoundBytes) * I;
                stream_store(reinterpret_cast<s16x8*>(d), one(x[0], std::integral_constant<int, 0>()));
                stream_store(reinterpret_cast<s16x8*>(d + kRoundBytes), one(x[1], std::integral_constant<int, 1>()));
                stream_store(reinterpret_cast<s16x8*>(d + 2 * kRoundBytes), one(x[2], std::integral_constant<int, 2>()));
                stream_store(reinterpret_cast<s16x8*>(d + 3 * kRoundBytes), one(x[3], std::integral_constant<int, 0>()));
                stream_store(reinterpret_cast<s16x8*>(d + 4 * kRoundBytes), one(x[4], std::integral_constant<int, 1>()));
                stream_store(reinterpret_cast<s16x8*>(d + 5 * kRoundBytes), one(x[5], std::integral_constant<int, 2>()));
            }
            for (; kSuperBins * I < dim_f; ++I) {                                     // the odd super-round and a ragged end (dim_f % 64)
                char* d = dst + (int64_t)(3 * kRoundBytes) * I;
                const int b0 = kSuperBins * I + bl[0], b1 = kSuperBins * I + bl[1], b2 = kSuperBins * I + bl[2];
                if (b0 < dim_f) stream_store(reinterpret_cast<s16x8*>(d), one(xs[b0], std::integral_constant<int, 0>()));
                if (b1 < dim_f) stream_store(reinterpret_cast<s16x8*>(d + kRoundBytes), one(xs[b1], std::integral_constant<int, 1>()));
                if (b2 < dim_f) stream_store(reinterpret_cast<s16x8*>(d + 2 * kRoundBytes), one(xs[b2], std::integral_constant<int, 2>()));
            }
        } else {
            if (dim_f >= N / 2) emit(std::true_type());          // every bin below N/2 is kept: no per-bin predicate
            else emit(std::false_type());
            if (t0 && dim_f > N / 2) {                           // Nyquist bin, self-paired
                const v2f z = za[LastDft<R2>::slot(R2 / 2)];
                store_bin<OutT>(spec, frame_off, b, N / 2, T, t, dim_f, LAYOUT, cx_add_conj(z, z), cx_sub_conj_divi(z, z));
            }
        }
    }
    if (PERSIST && id + n_wg < total) __syncthreads();       // the frame buffer (the epilogue's rounded bins) is free for the next frame's pass A
    }
}

// ------------------------------------------------------------------------------------------------
// iSTFT + overlap-add for n_fft = 256 * R2, hop = 128 * NBH (production: 6144 / 1024).
//
// Semantics: ConvTDFNetTrim.istft, mdxnet.py:58-75 (same contract as istft_regring_kernel in fft.hip).
// y = conj(FFT(conj Z)) / N with the passes in the order R2, 16, 16:
//   * pass A (radix R2, no twiddles) builds its inputs straight from the spectrogram: thread j owns the
//     butterflies j and 256 - j, whose inputs are exactly conj Z[k] and conj Z[N-k] of the bins k = j + 256 r and
//     (256 - j) + 256 r (r < R2/2) that it loads -- the Hermitian extension needs no exchange;
//   * pass C leaves sample n = tid + 128 (bb + NB q) in thread tid, so the overlap-add accumulator lives in
//     registers: 1/N * window, add, emit the NBH finished entries per frame, slide by NBH.
// LDS: rows of R2 values padded to R2 + 1 (conflict-free ds_write_b64 of the pass-A rows).
// ------------------------------------------------------------------------------------------------
template <typename InT> __device__ __forceinline__ void load_bin(const InT* p, v2f& L, v2f& R);
template <> __device__ __forceinline__ void load_bin<float>(const float* p, v2f& L, v2f& R) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(p);
    L = mk(q[0], q[1]); R = mk(q[2], q[3]);
}
template <> __device__ __forceinline__ void load_bin<bf16_t>(const bf16_t* p, v2f& L, v2f& R) {
    const bf16x4 q = *reinterpret_cast<const bf16x4*>(p);
    L = mk((float)q[0], (float)q[1]); R = mk((float)q[2], (float)q[3]);
}

template <typename InT> __device__ __forceinline__ void load_bin_g(const ALSEP_GLOBAL char* p, v2f& L, v2f& R);
template <> __device__ __forceinline__ void load_bin_g<float>(const ALSEP_GLOBAL char* p, v2f& L, v2f& R) {
    const f32x4 q = *reinterpret_cast<const ALSEP_GLOBAL f32x4*>(p);
    L = mk(q[0], q[1]); R = mk(q[2], q[3]);
}
template <> __device__ __forceinline__ void load_bin_g<bf16_t>(const ALSEP_GLOBAL char* p, v2f& L, v2f& R) {
    const bf16x4 q = *reinterpret_cast<const ALSEP_GLOBAL bf16x4*>(p);
    L = mk((float)q[0], (float)q[1]); R = mk((float)q[2], (float)q[3]);
}

template <int R2> constexpr int istft_lds_bytes() { return 256 * (R2 + 1) * 8; }

// grid (n_groups, n_chunks); each workgroup finishes `run` consecutive hop-blocks (as istft_regring_kernel).
// Occupancy: two waves per SIMD (256 VGPRs) where that needs no scratch -- the channels-last, full-band variant of the bench.  The
// variants with per-bin predicates or plane-strided loads (reference layout: HTDemucs, the model_run seam; dim_f < N/2) spill 47-195
// registers at 256 and get 512 instead: a kernel that uses scratch must not run from several HIP streams at once on this stack (the
// runners' lanes do exactly that; see nn_half.hip), and one wave per SIMD is the lesser cost for these off-bench geometries.
template <int R2, int NBH, typename InT, int LAYOUT, bool FULL>   // FULL: dim_f >= N/2 (no zero-filled bins below Nyquist)
__global__ void __launch_bounds__(kThreads)
ALSEP_WAVES_PER_EU_IF(LAYOUT == ALSEP_LAYOUT_NHWC && FULL, 2, 1)
istft_r16_kernel(const InT* __restrict__ spec, int dim_f, int T, const float2* __restrict__ tw_,
                 const float* __restrict__ env, int j_lo, int j_hi, int run,
                 float* __restrict__ out, int64_t out_ch_stride, int64_t out_chunk_stride, int64_t keep_lo,
                 int64_t keep_hi, int64_t out_limit) {
    constexpr int N = 256 * R2, NT = kThreads, HOP = 128 * NBH;
    constexpr int M = 16 * R2;                // butterflies of passes B and C
    constexpr int NB = M / NT;                // per thread
    constexpr int NA = N / NT;                // accumulator entries per thread
    constexpr int RS = R2 + 1;                // padded row stride of the pass-A image
    constexpr int H = R2 / 2;
    constexpr int Q = (N + HOP - 1) / HOP;
    static_assert(M % NT == 0 && NA % NBH == 0, "geometry");
    const v2f* __restrict__ tw = reinterpret_cast<const v2f*>(tw_);
    v2f* buf = reinterpret_cast<v2f*>(alsep_smem);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    const int j0 = j_lo + blockIdx.x * run;
    const int j1 = min(j0 + run, j_hi);
    if (j0 >= j1) return;
    const bool t0 = tid == 0;
    const int ka = tid, kb = tid ? 256 - tid : 128;

    // per-butterfly constants of passes B and C (kept across frames)
    int rowB[NB];                             // pass-A image offset of butterfly i's first input
    int dstB[NB];                             // pass-B output offset
    v2f wB1[NB], wC1[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        const int i = tid + NT * bb;
        const int a = i / R2, k = i - a * R2;
        rowB[bb] = a * RS + k;
        dstB[bb] = a * M + k;
        wB1[bb] = tw[k * 16];                 // W_{16 R2}^k
        wC1[bb] = tw[i];                      // W_N^i
    }

    v2f acc[NA];
#pragma unroll
    for (int m = 0; m < NA; ++m) acc[m] = mk(0.f, 0.f);
    // global addresses as uniform base (SGPR pair) + 32-bit lane offset: 64-bit per-load pointers would be hoisted
    // out of the frame loop by the compiler and spill
    const v2f wt = tw[tid];                   // (cos, -sin)(2 pi tid / N)
    const unsigned voffA = (unsigned)ka * (4u * (unsigned)sizeof(InT)), voffB = (unsigned)kb * (4u * (unsigned)sizeof(InT));

    const int t_start = max(0, j0 - Q + 1);
    for (int t = t_start; t < j1; ++t) {
        if (t < T) {
            // ---- pass A
            {
                const InT* frame = spec + (b * T + t) * (int64_t)dim_f * 4;   // NHWC row of this frame (uniform)
                v2f za[R2], zb[R2];
                v2f mA[H], mB[H];
                // One bin (L_re, L_im, R_re, R_im).  GUARDED = false (dim_f >= N/2, the production band: every bin of the
                // half spectrum is stored): plain loads, so the 2 H loads of a frame are in flight together.  Otherwise the
                // address is clamped into the row and the value zeroed by a select -- never a branch around the load: a
                // per-load `if (k < dim_f)` costs one exec-masked branch and one s_waitcnt vmcnt(0) per bin, i.e. 2 H
                // serialised memory round trips per frame (that was 3/4 of this kernel's time).
                auto fetch = [&](auto guarded, int r, int k, unsigned voff, v2f& L, v2f& R) {
                    if (LAYOUT == ALSEP_LAYOUT_NHWC) {
                        if (!decltype(guarded)::value) {
                            load_bin_g<InT>(opaque_uniform_gptr(reinterpret_cast<const char*>(frame + 256 * r * 4)) + voff, L, R);
                        } else {
                            const int kc = min(k, dim_f - 1);
                            load_bin_g<InT>(opaque_uniform_gptr(reinterpret_cast<const char*>(frame)) + (unsigned)kc * (4u * (unsigned)sizeof(InT)), L, R);
                            if (k >= dim_f) { L = mk(0.f, 0.f); R = mk(0.f, 0.f); }
                        }
                    } else {
                        L = mk(0.f, 0.f); R = mk(0.f, 0.f);
                        if (k < dim_f) {
                            const int64_t plane = (int64_t)dim_f * T;
                            const InT* s = spec + b * 4 * plane + (int64_t)k * T + t;
                            L = mk(to_f32(s[0]), to_f32(s[plane]));
                            R = mk(to_f32(s[2 * plane]), to_f32(s[3 * plane]));
                        }
                    }
                };
                auto pass_a_inputs = [&](auto guarded) {
#pragma unroll
                    for (int r = 0; r < H; ++r) {
                        v2f L, R;
                        fetch(guarded, r, ka + 256 * r, voffA, L, R);
                        if (r == 0 && t0) { L.y = 0.f; R.y = 0.f; }      // c2r ignores Im of DC
                        za[r] = cx_conj_add_pi(L, R);                      // conj Z[k]
                        mA[r] = cx_add_mi(L, R);                           // conj Z[N-k]
                    }
#pragma unroll
                    for (int r = 0; r < H; ++r) {
                        v2f L, R;
                        fetch(guarded, r, kb + 256 * r, voffB, L, R);
                        zb[r] = cx_conj_add_pi(L, R);
                        mB[r] = cx_add_mi(L, R);
                    }
                };
                pass_a_inputs(std::integral_constant<bool, !FULL>{});
                v2f nyq = mk(0.f, 0.f);
                if (t0 && dim_f > N / 2) {                               // Nyquist bin: Im ignored
                    v2f L, R;
                    if (LAYOUT == ALSEP_LAYOUT_NHWC) {
                        load_bin<InT>(spec + ((b * T + t) * (int64_t)dim_f + N / 2) * 4, L, R);
                    } else {
                        const int64_t plane = (int64_t)dim_f * T;
                        const InT* s = spec + b * 4 * plane + (int64_t)(N / 2) * T + t;
                        L = mk(to_f32(s[0]), 0.f);
                        R = mk(to_f32(s[2 * plane]), 0.f);
                    }
                    nyq = mk(L.x, -R.x);
                }
                // upper halves: index s = R2-1-r of butterfly a is N - (kb + 256 r), of b is N - (ka + 256 r);
                // thread 0 (a = 0, b = 128): a[R2 - r] = N - 256 r, a[R2/2] = Nyquist, b[R2-1-r] = N - (128 + 256 r)
#pragma unroll
                for (int s = H; s < R2; ++s) {
                    const v2f a0 = s == H ? nyq : mA[(R2 - s) % H];
                    za[s] = t0 ? a0 : mB[R2 - 1 - s];
                    zb[s] = t0 ? mB[R2 - 1 - s] : mA[R2 - 1 - s];
                }
                LastDft<R2>::run(za);
                LastDft<R2>::run(zb);
                v2f* ra = buf + ka * RS;
                v2f* rb = buf + kb * RS;
#pragma unroll
                for (int q = 0; q < R2; ++q) ra[q] = za[LastDft<R2>::slot(q)];
#pragma unroll
                for (int q = 0; q < R2; ++q) rb[q] = zb[LastDft<R2>::slot(q)];
            }
            __syncthreads();
            // ---- pass B: radix 16, P = R2
            {
                v2f u[NB][16];
#pragma unroll
                for (int bb = 0; bb < NB; ++bb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) u[bb][r] = buf[rowB[bb] + 16 * RS * r];
                __syncthreads();                                         // every read of the pass-A image is done
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) {
                    v2f w[16];
                    twiddle_powers<16>(wB1[bb], w);
                    cx_mul_n<16>(u[bb], w);
                    dft16(u[bb]);
#pragma unroll
                    for (int q = 0; q < 16; ++q) buf[dstB[bb] + R2 * q] = u[bb][dft16_slot(q)];
                }
            }
            __syncthreads();
            // ---- pass C: radix 16, P = 16 R2; output sample n = tid + 128 (bb + NB q).  Nothing is written to LDS
            // here, so the butterflies go one at a time (32 live registers instead of 32 NB beside the accumulator).
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                v2f u[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) u[r] = buf[tid + NT * bb + M * r];
                v2f w[16];
                twiddle_powers<16>(wC1[bb], w);
                cx_mul_n<16>(u, w);
                dft16(u);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = bb + NB * q;
                    // window / N at n = tid + 128 m, from W_N^tid (a register) and the constant W_N^(128 m):
                    // cos(2 pi n / N) = Re(W_N^tid W_N^(128 m)); two FMAs instead of a table load per sample
                    constexpr double kA = 6.283185307179586476925286766559 * (double)NT / (double)N;
                    const float cm = (float)(-0.5 / N * __builtin_cos(kA * m)), sm = (float)(-0.5 / N * __builtin_sin(kA * m));
                    const float wv = fmaf(wt.x, sgpr_literal(cm), fmaf(wt.y, sgpr_literal(sm), (float)(0.5 / N)));
                    acc[m] += u[dft16_slot(q)] * wv;                     // conj applied at the store
                }
            }
            __syncthreads();                                             // buf is rewritten by the next frame
        }
        // hop-block j = t is complete
        if (t >= j0) {
            // envelope values of the NBH finished samples first, as one batch of loads (the table has
            // N + HOP (T - 1) entries; the clamp keeps an unguarded load inside it), then the guarded stores
            float e[NBH];
            const int64_t p_last = (int64_t)(T - 1) * HOP + N - 1;
#pragma unroll
            for (int j = 0; j < NBH; ++j) e[j] = env[min((int64_t)t * HOP + j * NT + tid, p_last)];
#pragma unroll
            for (int j = 0; j < NBH; ++j) {
                const int64_t p = (int64_t)t * HOP + j * NT + tid;
                const int64_t s = p - N / 2;
                if (s >= keep_lo && s < keep_hi) {
                    const int64_t o = b * out_chunk_stride + (s - keep_lo);
                    if (o < out_limit) {
                        const float r = 1.0f / e[j];
                        out[o] = acc[j].x * r;
                        out[out_ch_stride + o] = -acc[j].y * r;
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NA - NBH; ++m) acc[m] = acc[m + NBH];
#pragma unroll
        for (int m = NA - NBH; m < NA; ++m) acc[m] = mk(0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// iSTFT + overlap-add for n_fft = 7680 (the UVR vocal models' geometry), hop 1024: passes in the order 16, 16, 30.
// The register overlap-add needs the LAST pass's outputs of a thread to be samples tid + 128 m: with radix 30 last (P = 256) butterfly
// i in {tid, tid + 128} leaves samples i + 256 q = tid + 128 (bbit + 2 q), m < 60 (the accumulator is padded to 64 = 8 hop blocks of 8).
// Pass A (radix 16, P = 1, inputs 480 apart) builds its inputs straight from the spectrogram with the Hermitian extension folded in,
// as the 6144 kernel does: butterflies j and 480 - j are one work item whose 16 loaded bins j + 480 r and (480 - j) + 480 r (r < 8)
// are everything both need (u_j[r >= 8] = conj Z[N - k] of the other's bins); j = 0 and j = 240 pair with themselves.  241 items on
// 128 threads.  Passes A and B use the STFT kernel's LDS images (rows of 16 padded to 18; then plain).  y = conj(FFT(conj Z)) / N.
// One wave per SIMD (the 64-entry accumulator + a radix-30 butterfly with its twiddles exceed 256 registers; no scratch).
// ------------------------------------------------------------------------------------------------
constexpr int istft_r30_lds_bytes() { return 480 * 18 * 8; }     // pass-A image 69 120 B >= the plain 7680 x 8 B of pass B

template <int NBH, typename InT, int LAYOUT>
__global__ void __launch_bounds__(kThreads) ALSEP_WAVES_PER_EU(1)
istft_r30_kernel(const InT* __restrict__ spec, int dim_f, int T, const float2* __restrict__ tw_,
                 const float* __restrict__ env, int j_lo, int j_hi, int run,
                 float* __restrict__ out, int64_t out_ch_stride, int64_t out_chunk_stride, int64_t keep_lo,
                 int64_t keep_hi, int64_t out_limit) {
    constexpr int N = 7680, NT = kThreads, HOP = 128 * NBH, M = 480, NB = 4, RW = 18, NA = 64, R3 = 30;
    constexpr int Q = (N + HOP - 1) / HOP;
    static_assert(NT == 128 && NBH == 8, "geometry");
    const v2f* __restrict__ tw = reinterpret_cast<const v2f*>(tw_);
    v2f* buf = reinterpret_cast<v2f*>(alsep_smem);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    const int j0 = j_lo + blockIdx.x * run;
    const int j1 = min(j0 + run, j_hi);
    if (j0 >= j1) return;
    v2f acc[NA];
#pragma unroll
    for (int m = 0; m < NA; ++m) acc[m] = mk(0.f, 0.f);
    const v2f wt = tw[tid];                                   // (cos, -sin)(2 pi tid / N): the window, as in the 6144 kernel
    const v2f wB1 = tw[(tid & 15) * (N / 256)];               // W_256^k of pass B
    const int64_t plane = (int64_t)dim_f * T;

    const int t_start = max(0, j0 - Q + 1);
    for (int t = t_start; t < j1; ++t) {
        if (t < T) {
            // one bin (L, R) = (L_re + i L_im, R_re + i R_im); zero at and above dim_f (clamped address + select: no branch around a load)
            auto fetch = [&](int k, v2f& L, v2f& R) {
                const int kc = min(k, dim_f - 1);
                if (LAYOUT == ALSEP_LAYOUT_NHWC) {
                    load_bin<InT>(spec + ((b * T + t) * (int64_t)dim_f + kc) * 4, L, R);
                } else {
                    const InT* sp = spec + b * 4 * plane + (int64_t)kc * T + t;
                    L = mk(to_f32(sp[0]), to_f32(sp[plane]));
                    R = mk(to_f32(sp[2 * plane]), to_f32(sp[3 * plane]));
                }
                if (k >= dim_f) { L = mk(0.f, 0.f); R = mk(0.f, 0.f); }
            };
            auto store_row = [&](int j, v2f (&u)[16]) {
                dft16(u);
                f32x4* row = reinterpret_cast<f32x4*>(buf + j * RW);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const v2f lo = u[dft16_slot(2 * g)], hi = u[dft16_slot(2 * g + 1)];
                    row[g] = f32x4{lo.x, lo.y, hi.x, hi.y};
                }
            };
            // ---- pass A
#pragma unroll 1
            for (int it = 0; it < 2; ++it) {
                const int j = tid + NT * it;                      // work item: butterflies j and 480 - j
                if (j > M / 2) break;                             // 241 items
                const bool self = (j == 0) || (j == M / 2);
                v2f za[8], ma[8], zb[8], mb[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    v2f L, R;
                    fetch(j + M * r, L, R);
                    if (j == 0 && r == 0) { L.y = 0.f; R.y = 0.f; }   // c2r ignores Im of DC
                    za[r] = cx_conj_add_pi(L, R);                 // conj Z[k]
                    ma[r] = cx_add_mi(L, R);                      // conj Z[N - k]
                }
                if (!self) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        v2f L, R;
                        fetch((M - j) + M * r, L, R);
                        zb[r] = cx_conj_add_pi(L, R);
                        mb[r] = cx_add_mi(L, R);
                    }
                }
                v2f u[16];
                if (j == 0) {
                    // inputs 480 r: r < 8 direct, r = 8 the Nyquist bin (Im ignored), r > 8 the mirror of bin 480 (16 - r)
                    v2f nyq = mk(0.f, 0.f);
                    if (dim_f > N / 2) {
                        v2f L, R;
                        fetch(N / 2, L, R);
                        nyq = mk(L.x, -R.x);
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) u[r] = za[r];
                    u[8] = nyq;
#pragma unroll
                    for (int r = 9; r < 16; ++r) u[r] = ma[16 - r];
                    store_row(0, u);
                } else if (j == M / 2) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { u[r] = za[r]; u[8 + r] = ma[7 - r]; }
                    store_row(M / 2, u);
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { u[r] = za[r]; u[8 + r] = mb[7 - r]; }      // index j + 480 (8 + r) = N - ((480 - j) + 480 (7 - r))
                    store_row(j, u);
#pragma unroll
                    for (int r = 0; r < 8; ++r) { u[r] = zb[r]; u[8 + r] = ma[7 - r]; }
                    store_row(M - j, u);
                }
            }
            __syncthreads();
            // ---- pass B: radix 16, P = 16 (the STFT kernel's pass B on the padded rows)
            {
                v2f u[NB][16];
                const int k = tid & 15;
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) {
                    const int i = tid + NT * bb;
                    const v2f* src = buf + (i >> 4) * RW + k;
                    if (i < M) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) u[bb][r] = src[(M / 16) * RW * r];
                    }
                }
                v2f w[16];
                twiddle_powers<16>(wB1, w);
                __syncthreads();                                  // every read of the pass-A image is done
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) {
                    const int i = tid + NT * bb;
                    if (i < M) {
                        cx_mul_n<16>(u[bb], w);
                        dft16(u[bb]);
                        v2f* dst = buf + (i >> 4) * 256 + k;
#pragma unroll
                        for (int q = 0; q < 16; ++q) dst[16 * q] = u[bb][dft16_slot(q)];
                    }
                }
            }
            __syncthreads();
            // ---- pass C: radix 30, P = 256; butterfly i = tid + 128 bbit leaves sample i + 256 q = tid + 128 (bbit + 2 q)
#pragma unroll
            for (int bbit = 0; bbit < 2; ++bbit) {
                const int i = tid + NT * bbit;
                v2f z[R3];
#pragma unroll
                for (int r = 0; r < R3; ++r) z[r] = buf[i + 256 * r];
                v2f w[R3];
                twiddle_powers<R3>(tw[i], w);
                cx_mul_n<R3>(z, w);
                LastDft<R3>::run(z);
#pragma unroll
                for (int q = 0; q < R3; ++q) {
                    const int m = bbit + 2 * q;
                    constexpr double kA = 6.283185307179586476925286766559 * (double)NT / (double)N;
                    const float cm = (float)(-0.5 / N * __builtin_cos(kA * m)), sm = (float)(-0.5 / N * __builtin_sin(kA * m));
                    const float wv = fmaf(wt.x, sgpr_literal(cm), fmaf(wt.y, sgpr_literal(sm), (float)(0.5 / N)));
                    acc[m] += z[LastDft<R3>::slot(q)] * wv;      // conj applied at the store
                }
            }
            __syncthreads();                                      // buf is rewritten by the next frame
        }
        if (t >= j0) {                                            // hop-block j = t is complete
            float e[NBH];
            const int64_t p_last = (int64_t)(T - 1) * HOP + N - 1;
#pragma unroll
            for (int j = 0; j < NBH; ++j) e[j] = env[min((int64_t)t * HOP + j * NT + tid, p_last)];
#pragma unroll
            for (int j = 0; j < NBH; ++j) {
                const int64_t p = (int64_t)t * HOP + j * NT + tid;
                const int64_t sidx = p - N / 2;
                if (sidx >= keep_lo && sidx < keep_hi) {
                    const int64_t o = b * out_chunk_stride + (sidx - keep_lo);
                    if (o < out_limit) {
                        const float r = 1.0f / e[j];
                        out[o] = acc[j].x * r;
                        out[out_ch_stride + o] = -acc[j].y * r;
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NA - NBH; ++m) acc[m] = acc[m + NBH];
#pragma unroll
        for (int m = NA - NBH; m < NA; ++m) acc[m] = mk(0.f, 0.f);
    }
}

}  // namespace r16
