// STFT / iSTFT of the MDX front and back end as LDS-resident mixed-radix FFTs (gfx950).
//
// Semantics: ConvTDFNetTrim.stft / .istft, reference modules/rvc/infer/modules/uvr5/mdxnet.py:41-75
// (torch.stft/istft, periodic Hann, center=True, no normalisation).
//
// Design
//  * One workgroup transforms one frame of BOTH stereo channels with a single complex FFT of
//    n_fft points ("two-for-one": z = xL + i*xR; XL[k] = (Z[k]+conj Z[N-k])/2,
//    XR[k] = (Z[k]-conj Z[N-k])/(2i)).  The whole transform lives in LDS (n_fft*8 B: 48 KiB at
//    6144), Stockham auto-sort passes of radix 8/5/4/3/2, one read + one write of the LDS
//    buffer per pass, radix sequence fixed at compile time per n_fft.
//  * STFT loads are coalesced float reads of the PCM (each sample is re-read n_fft/hop times,
//    served from L2); stores in NHWC layout are one 16-byte (f32) or 8-byte (bf16) vector per
//    bin: [B,T,dim_f,(L_re,L_im,R_re,R_im)].
//  * iSTFT walks a run of consecutive frames per workgroup, overlap-adds in an LDS ring of
//    ceil(n_fft/hop) hop-blocks, divides by the exact sum-of-w^2 envelope table (the envelope
//    ripples when hop does not divide n_fft: 7680/1024) and writes each finished hop-block once,
//    straight to its final place (chunk trim / stitch fused into the store).
#include "alsep_common.h"

#include <cmath>
#include <cstdlib>

#include <alsep_gfx950_asm.h>

struct alsep_plan {
    alsep_ctx* ctx = nullptr;
    int n_fft = 0, hop = 0, dim_f = 0, dim_t = 0;
    int chunk = 0;
    float2* tw = nullptr;     // W_N^j = exp(-2*pi*i*j/N), j in [0,N)
    float* win = nullptr;     // periodic Hann window, N floats
    float* env = nullptr;     // sum_t w^2 over the padded chunk timeline, N + hop*(T-1)
    int64_t env_len = 0;
};

// ------------------------------------------------------------------------------------------
// complex helpers + small DFTs (forward, sign -)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_negi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

template <int R> __device__ __forceinline__ void dft(float2 (&u)[R]);

template <> __device__ __forceinline__ void dft<2>(float2 (&u)[2]) {
    float2 a = u[0], b = u[1];
    u[0] = cadd(a, b);
    u[1] = csub(a, b);
}
template <> __device__ __forceinline__ void dft<3>(float2 (&u)[3]) {
    const float S = 0.86602540378443864676f;
    float2 t1 = cadd(u[1], u[2]);
    float2 t2 = make_float2(u[0].x - 0.5f * t1.x, u[0].y - 0.5f * t1.y);
    float2 t3 = cscale(csub(u[1], u[2]), S);
    u[0] = cadd(u[0], t1);
    u[1] = make_float2(t2.x + t3.y, t2.y - t3.x);
    u[2] = make_float2(t2.x - t3.y, t2.y + t3.x);
}
template <> __device__ __forceinline__ void dft<4>(float2 (&u)[4]) {
    float2 t0 = cadd(u[0], u[2]), t1 = csub(u[0], u[2]);
    float2 t2 = cadd(u[1], u[3]), t3 = cmul_negi(csub(u[1], u[3]));
    u[0] = cadd(t0, t2);
    u[2] = csub(t0, t2);
    u[1] = cadd(t1, t3);
    u[3] = csub(t1, t3);
}
template <> __device__ __forceinline__ void dft<5>(float2 (&u)[5]) {
    const float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;
    const float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;
    float2 t1 = cadd(u[1], u[4]), t2 = cadd(u[2], u[3]);
    float2 t3 = csub(u[1], u[4]), t4 = csub(u[2], u[3]);
    float2 m1 = make_float2(u[0].x + C1 * t1.x + C2 * t2.x, u[0].y + C1 * t1.y + C2 * t2.y);
    float2 m2 = make_float2(u[0].x + C2 * t1.x + C1 * t2.x, u[0].y + C2 * t1.y + C1 * t2.y);
    float2 n1 = make_float2(S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y);
    float2 n2 = make_float2(S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y);
    u[0] = cadd(u[0], cadd(t1, t2));
    u[1] = make_float2(m1.x + n1.y, m1.y - n1.x);     // m1 - i n1
    u[4] = make_float2(m1.x - n1.y, m1.y + n1.x);     // m1 + i n1
    u[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
    u[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
}
template <> __device__ __forceinline__ void dft<8>(float2 (&u)[8]) {
    const float H = 0.70710678118654752440f;
    float2 e[4] = {u[0], u[2], u[4], u[6]};
    float2 o[4] = {u[1], u[3], u[5], u[7]};
    dft<4>(e);
    dft<4>(o);
    float2 o1 = make_float2(H * (o[1].x + o[1].y), H * (o[1].y - o[1].x));      // * (1-i)/sqrt2
    float2 o2 = cmul_negi(o[2]);                                                 // * (-i)
    float2 o3 = make_float2(H * (o[3].y - o[3].x), -H * (o[3].x + o[3].y));     // * (-1-i)/sqrt2
    u[0] = cadd(e[0], o[0]);  u[4] = csub(e[0], o[0]);
    u[1] = cadd(e[1], o1);    u[5] = csub(e[1], o1);
    u[2] = cadd(e[2], o2);    u[6] = csub(e[2], o2);
    u[3] = cadd(e[3], o3);    u[7] = csub(e[3], o3);
}

// One Stockham pass of radix R over the LDS buffer; P = product of the radices already done.
// Butterfly i reads buf[i + r*N/R], twiddles by W_{P*R}^{k*r} (k = i mod P) and writes
// buf[(i-k)*R + k + r*P].  All reads precede all writes (barrier), so one buffer suffices.
// w1[b] = W_{P*R}^{k} of this thread's b-th butterfly, preloaded into registers (FftRegs): a table
// read inside the pass would put one dependent memory latency on every pass of the chain.
template <int N, int NT, int P, int R>
__device__ __forceinline__ void fft_pass(float2* buf, const float2 (&w1)[(N / R + NT - 1) / NT], int tid) {
    constexpr int M = N / R;
    constexpr int NB = (M + NT - 1) / NT;
    float2 u[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tid + b * NT;
        if (i < M) {
#pragma unroll
            for (int r = 0; r < R; ++r) u[b][r] = buf[i + r * M];
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int i = tid + b * NT;
        if (i < M) {
            const int k = i % P;
            if (P > 1) {
                // W^r by products of W^1 (depth <= 3 complex multiplies)
                float2 w[R];
                w[1] = w1[b];
#pragma unroll
                for (int r = 2; r < R; ++r) w[r] = (r & 1) ? cmul(w[r - 1], w[1]) : cmul(w[r >> 1], w[r >> 1]);
#pragma unroll
                for (int r = 1; r < R; ++r) u[b][r] = cmul(u[b][r], w[r]);
            }
            dft<R>(u[b]);
            const int j = (i - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) buf[j + r * P] = u[b][r];
        }
    }
    __syncthreads();
}

// The whole transform as a chain of passes, each with its first-order twiddles in registers:
// load() once per workgroup (all table reads issued together, before the first pass), run() per frame.
template <int N, int NT, int P, int... Rs> struct FftRegs;
template <int N, int NT, int P> struct FftRegs<N, NT, P> {
    static_assert(P == N, "radices must multiply to N");
    __device__ __forceinline__ void load(const float2*, int) {}
    __device__ __forceinline__ void run(float2*, int) const {}
};
template <int N, int NT, int P, int R0, int... Rs> struct FftRegs<N, NT, P, R0, Rs...> {
    static constexpr int M = N / R0, NB = (M + NT - 1) / NT, TWS = N / (P * R0);
    float2 w1[NB];
    FftRegs<N, NT, P * R0, Rs...> next;
    __device__ __forceinline__ void load(const float2* __restrict__ tw, int tid) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = tid + b * NT;
            w1[b] = (P > 1 && i < M) ? tw[(i % P) * TWS] : make_float2(1.f, 0.f);
        }
        next.load(tw, tid);
    }
    __device__ __forceinline__ void run(float2* buf, int tid) const {
        fft_pass<N, NT, P, R0>(buf, w1, tid);
        next.run(buf, tid);
    }
};

// radix sequences per supported n_fft
template <int N, int NT> struct Fft;
#define ALSEP_FFT(N_, ...)                                                                 \
    template <int NT> struct Fft<N_, NT> { typedef FftRegs<N_, NT, 1, __VA_ARGS__> Regs; };
ALSEP_FFT(256, 8, 8, 4)
ALSEP_FFT(320, 5, 8, 8)
ALSEP_FFT(384, 3, 2, 8, 8)
ALSEP_FFT(480, 5, 3, 8, 4)
ALSEP_FFT(512, 8, 8, 8)
ALSEP_FFT(640, 5, 8, 8, 2)
ALSEP_FFT(960, 5, 3, 8, 8)
ALSEP_FFT(1024, 8, 8, 8, 2)
ALSEP_FFT(2048, 8, 8, 8, 4)
ALSEP_FFT(4096, 8, 8, 8, 8)
ALSEP_FFT(5120, 5, 8, 8, 8, 2)
ALSEP_FFT(6144, 3, 4, 8, 8, 8)
ALSEP_FFT(7680, 5, 8, 3, 8, 8)
ALSEP_FFT(8192, 8, 8, 8, 8, 2)
ALSEP_FFT(16384, 8, 8, 8, 8, 4)
#undef ALSEP_FFT

// 5120: UVR-MDX-NET_Crowd_HQ_1; 6144 / 7680: the UVR vocal / instrumental models; 4096 / 8192 / 16384: the KUIELab drums / other / bass
// models (kuielab_a_bass.onnx is the alt-bass model of stem_separator.py:512) and HTDemucs (4096)
// 320 / 640 / 960 (+ 512): the four bands of the VR models (lib_v5/modelparams/4band_v2.json, 4band_v3.json)
#define ALSEP_FOR_EACH_NFFT(X) X(256) X(320) X(384) X(480) X(512) X(640) X(960) X(1024) X(2048) X(4096) X(5120) X(6144) X(7680) X(8192) X(16384)

constexpr int kFftThreads = 256;

template <typename OutT> __device__ __forceinline__ void store_spec4(OutT* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store_spec4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store_spec4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    bf16x4 v;
    v[0] = (bf16_t)a; v[1] = (bf16_t)b; v[2] = (bf16_t)c; v[3] = (bf16_t)d;
    *reinterpret_cast<bf16x4*>(p) = v;
}
template <typename T> __device__ __forceinline__ void load_spec4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load_spec4<float>(const float* p, float (&v)[4]) {
    float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
template <> __device__ __forceinline__ void load_spec4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    bf16x4 q = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
}

// ------------------------------------------------------------------------------------------
// STFT: grid (T, n_chunks), one frame of both channels per workgroup.
// ------------------------------------------------------------------------------------------
template <int N, typename OutT, int LAYOUT>
__global__ void __launch_bounds__(kFftThreads)
stft_kernel(const float* __restrict__ pcm, int64_t ch_stride, int64_t chunk_stride, int chunk, int hop,
            int dim_f, int T, const float2* __restrict__ tw, const float* __restrict__ win, OutT* __restrict__ spec) {
    constexpr int NT = kFftThreads;
    float2* buf = reinterpret_cast<float2*>(alsep_smem);
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    const int64_t b = blockIdx.y;
    const float* xl = pcm + b * chunk_stride;
    const float* xr = xl + ch_stride;
    typename Fft<N, NT>::Regs fft;
    fft.load(tw, tid);
    const int p0 = t * hop - N / 2;
    for (int n = tid; n < N; n += NT) {
        int p = p0 + n;
        if (p < 0) p = -p;                                   // reflect (center=True)
        if (p >= chunk) p = 2 * (chunk - 1) - p;
        const float w = win[n];
        buf[n] = make_float2(xl[p] * w, xr[p] * w);
    }
    __syncthreads();
    fft.run(buf, tid);
    for (int k = tid; k < dim_f; k += NT) {
        const float2 zk = buf[k];
        const float2 zn = buf[k == 0 ? 0 : N - k];
        const float lre = 0.5f * (zk.x + zn.x), lim = 0.5f * (zk.y - zn.y);
        const float rre = 0.5f * (zk.y + zn.y), rim = -0.5f * (zk.x - zn.x);
        if (LAYOUT == ALSEP_LAYOUT_NHWC) {
            store_spec4<OutT>(spec + ((b * T + t) * (int64_t)dim_f + k) * 4, lre, lim, rre, rim);
        } else {
            const int64_t plane = (int64_t)dim_f * T;
            OutT* o = spec + b * 4 * plane + (int64_t)k * T + t;
            o[0] = from_f32<OutT>(lre);
            o[plane] = from_f32<OutT>(lim);
            o[2 * plane] = from_f32<OutT>(rre);
            o[3 * plane] = from_f32<OutT>(rim);
        }
    }
}

#include "fft_r16.h"

// ------------------------------------------------------------------------------------------
// iSTFT + overlap-add + envelope divide + trim/stitch store.
// grid (n_groups, n_chunks); each workgroup finishes `run` consecutive hop-blocks.
// LDS: buf[N] (FFT) + ring[Q*hop] (OLA accumulator, float2 = both channels), Q = ceil(N/hop).
// ------------------------------------------------------------------------------------------
template <int N, typename InT, int LAYOUT>
__global__ void __launch_bounds__(kFftThreads)
istft_kernel(const InT* __restrict__ spec, int hop, int dim_f, int T, const float2* __restrict__ tw,
             const float* __restrict__ win, const float* __restrict__ env, int j_lo, int j_hi, int run, float* __restrict__ out,
             int64_t out_ch_stride, int64_t out_chunk_stride, int64_t keep_lo, int64_t keep_hi,
             int64_t out_limit) {
    constexpr int NT = kFftThreads;
    float2* buf = reinterpret_cast<float2*>(alsep_smem);
    float2* ring = buf + N;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    const int Q = (N + hop - 1) / hop;
    const int RN = Q * hop;
    const int j0 = j_lo + blockIdx.x * run;
    const int j1 = min(j0 + run, j_hi);
    if (j0 >= j1) return;
    for (int i = tid; i < RN; i += NT) ring[i] = make_float2(0.f, 0.f);
    const float inv_n = 1.0f / (float)N;
    const int t_start = max(0, j0 - Q + 1);
    for (int t = t_start; t < j1; ++t) {
        if (t < T) {
            typename Fft<N, NT>::Regs fft;                   // twiddle reads issued with the spectrogram loads,
            fft.load(tw, tid);                               // not kept live across frames (register pressure)
            // conj(Z) with Z[k] = XL[k] + i XR[k], Hermitian-extended; bins >= dim_f are zero.
            for (int k = tid; k <= N / 2; k += NT) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (k < dim_f) {
                    if (LAYOUT == ALSEP_LAYOUT_NHWC) {
                        load_spec4<InT>(spec + ((b * T + t) * (int64_t)dim_f + k) * 4, v);
                    } else {
                        const int64_t plane = (int64_t)dim_f * T;
                        const InT* s = spec + b * 4 * plane + (int64_t)k * T + t;
                        v[0] = to_f32(s[0]); v[1] = to_f32(s[plane]);
                        v[2] = to_f32(s[2 * plane]); v[3] = to_f32(s[3 * plane]);
                    }
                }
                if (k == 0 || k == N / 2) {                  // c2r ignores Im of DC / Nyquist
                    buf[k] = make_float2(v[0], -v[2]);
                } else {
                    buf[k] = make_float2(v[0] - v[3], -(v[1] + v[2]));
                    buf[N - k] = make_float2(v[0] + v[3], v[1] - v[2]);
                }
            }
            __syncthreads();
            fft.run(buf, tid);
            const int base = (t % Q) * hop;                  // (t*hop) mod RN
            for (int n = tid; n < N; n += NT) {
                const float w = win[n] * inv_n;
                int r = base + n;
                if (r >= RN) r -= RN;
                float2 a = ring[r];
                a.x += w * buf[n].x;                         // z = conj(buf)/N
                a.y -= w * buf[n].y;
                ring[r] = a;
            }
            __syncthreads();
        }
        // hop-block j = t is complete: every frame covering it (t-Q+1..t) has been added.
        const int slot = (t % Q) * hop;
        if (t >= j0) {
            const int64_t p_base = (int64_t)t * hop;
            for (int i = tid; i < hop; i += NT) {
                const int64_t p = p_base + i;
                const int64_t s = p - N / 2;
                if (s >= keep_lo && s < keep_hi) {
                    const int64_t o = b * out_chunk_stride + (s - keep_lo);
                    if (o < out_limit) {
                        const float e = 1.0f / env[p];
                        const float2 a = ring[slot + i];
                        out[o] = a.x * e;
                        out[out_ch_stride + o] = a.y * e;
                    }
                }
            }
        }
        for (int i = tid; i < hop; i += NT) ring[slot + i] = make_float2(0.f, 0.f);
        __syncthreads();
    }
}

// iSTFT with the overlap-add accumulator in REGISTERS (hop and n_fft multiples of the 256 threads):
// thread tid owns window positions j*256 + tid, j < N/256; after each frame the first hop/256 entries
// (the finished hop-block) are written out and the array slides down.  LDS then holds only the FFT
// buffer (48 KiB at 6144), so three workgroups fit a CU instead of one -- the kernel is a chain of
// dependent passes per frame and needs the other workgroups to hide its latency.
template <int N, int HOP, typename InT, int LAYOUT>
__global__ void __launch_bounds__(kFftThreads)
istft_regring_kernel(const InT* __restrict__ spec, int dim_f, int T, const float2* __restrict__ tw,
                     const float* __restrict__ win, const float* __restrict__ env, int j_lo, int j_hi, int run,
                     float* __restrict__ out, int64_t out_ch_stride, int64_t out_chunk_stride, int64_t keep_lo,
                     int64_t keep_hi, int64_t out_limit) {
    constexpr int NT = kFftThreads;
    constexpr int NA = N / NT, NB = HOP / NT;               // accumulator entries, entries per hop-block
    constexpr int Q = (N + HOP - 1) / HOP;
    static_assert(N % NT == 0 && HOP % NT == 0, "register ring needs hop and n_fft to be multiples of 256");
    float2* buf = reinterpret_cast<float2*>(alsep_smem);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    const int j0 = j_lo + blockIdx.x * run;
    const int j1 = min(j0 + run, j_hi);
    if (j0 >= j1) return;
    float2 acc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) acc[j] = make_float2(0.f, 0.f);
    float wv[NA];
    const float inv_n = 1.0f / (float)N;
#pragma unroll
    for (int j = 0; j < NA; ++j) wv[j] = win[j * NT + tid] * inv_n;
    const int t_start = max(0, j0 - Q + 1);
    for (int t = t_start; t < j1; ++t) {
        if (t < T) {
            typename Fft<N, NT>::Regs fft;
            fft.load(tw, tid);
            if (LAYOUT == ALSEP_LAYOUT_NHWC) {
                // All of a thread's bins as ONE batch of loads (address clamped into the frame's row, value zeroed by a
                // select for bins >= dim_f): a per-bin `if (k < dim_f)` around the load is an exec-masked branch plus
                // s_waitcnt vmcnt(0) per bin, i.e. N/512 serialised memory round trips per frame.
                constexpr int KI = N / 2 / NT;
                static_assert(N % (2 * NT) == 0, "register ring: n_fft multiple of 512");
                const InT* row = spec + (b * T + t) * (int64_t)dim_f * 4;
                float v[KI][4];
#pragma unroll
                for (int i = 0; i < KI; ++i) load_spec4<InT>(row + (int64_t)min(tid + i * NT, dim_f - 1) * 4, v[i]);
                float vn[4] = {0.f, 0.f, 0.f, 0.f};
                if (tid == 0 && N / 2 < dim_f) load_spec4<InT>(row + (int64_t)(N / 2) * 4, vn);
#pragma unroll
                for (int i = 0; i < KI; ++i) {
                    const int k = tid + i * NT;
                    if (k >= dim_f) { v[i][0] = 0.f; v[i][1] = 0.f; v[i][2] = 0.f; v[i][3] = 0.f; }
                    if (i == 0 && tid == 0) {                 // c2r ignores Im of DC
                        buf[0] = make_float2(v[i][0], -v[i][2]);
                    } else {
                        buf[k] = make_float2(v[i][0] - v[i][3], -(v[i][1] + v[i][2]));
                        buf[N - k] = make_float2(v[i][0] + v[i][3], v[i][1] - v[i][2]);
                    }
                }
                if (tid == 0) buf[N / 2] = make_float2(vn[0], -vn[2]);   // ... and of Nyquist
            } else
            for (int k = tid; k <= N / 2; k += NT) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (k < dim_f) {
                    if (LAYOUT == ALSEP_LAYOUT_NHWC) {
                        load_spec4<InT>(spec + ((b * T + t) * (int64_t)dim_f + k) * 4, v);
                    } else {
                        const int64_t plane = (int64_t)dim_f * T;
                        const InT* s = spec + b * 4 * plane + (int64_t)k * T + t;
                        v[0] = to_f32(s[0]); v[1] = to_f32(s[plane]);
                        v[2] = to_f32(s[2 * plane]); v[3] = to_f32(s[3 * plane]);
                    }
                }
                if (k == 0 || k == N / 2) {
                    buf[k] = make_float2(v[0], -v[2]);
                } else {
                    buf[k] = make_float2(v[0] - v[3], -(v[1] + v[2]));
                    buf[N - k] = make_float2(v[0] + v[3], v[1] - v[2]);
                }
            }
            __syncthreads();
            fft.run(buf, tid);
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const float2 z = buf[j * NT + tid];
                acc[j].x += wv[j] * z.x;
                acc[j].y -= wv[j] * z.y;
            }
            __syncthreads();                                 // buf is rewritten by the next frame
        }
        if (t >= j0) {
            float ev[NB];                                    // envelope values as one batch of loads, then the guarded stores
            const int64_t p_last = (int64_t)(T - 1) * HOP + N - 1;   // the table has N + HOP (T - 1) entries
#pragma unroll
            for (int j = 0; j < NB; ++j) ev[j] = env[min((int64_t)t * HOP + j * NT + tid, p_last)];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int64_t p = (int64_t)t * HOP + j * NT + tid;
                const int64_t s = p - N / 2;
                if (s >= keep_lo && s < keep_hi) {
                    const int64_t o = b * out_chunk_stride + (s - keep_lo);
                    if (o < out_limit) {
                        const float e = 1.0f / ev[j];
                        out[o] = acc[j].x * e;
                        out[out_ch_stride + o] = acc[j].y * e;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NA - NB; ++j) acc[j] = acc[j + NB];
#pragma unroll
        for (int j = NA - NB; j < NA; ++j) acc[j] = make_float2(0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// layout conversion REF [B,4,F,T] <-> NHWC [B,T,F,4] via 32x32 LDS tiles (coalesced both ways)
// ------------------------------------------------------------------------------------------
template <typename T, int TO_NHWC>
__global__ void __launch_bounds__(256)
spec_convert_kernel(const T* __restrict__ src, T* __restrict__ dst, int F, int Tn) {
    T* tile = reinterpret_cast<T*>(alsep_smem);             // [4][32][33]
    const int64_t b = blockIdx.z;
    const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t plane = (int64_t)F * Tn;
    if (TO_NHWC) {
        for (int c = 0; c < 4; ++c)
            for (int r = ty; r < 32; r += 8) {               // r: f within tile, tx: t within tile
                const int f = f0 + r, t = t0 + tx;
                if (f < F && t < Tn) tile[(c * 32 + r) * 33 + tx] = src[(b * 4 + c) * plane + (int64_t)f * Tn + t];
            }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {                   // r: t within tile; tx: f within tile
            const int f = f0 + tx, t = t0 + r;
            if (f < F && t < Tn) {
                T* o = dst + ((b * Tn + t) * (int64_t)F + f) * 4;
                for (int c = 0; c < 4; ++c) o[c] = tile[(c * 32 + tx) * 33 + r];
            }
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            const int f = f0 + tx, t = t0 + r;
            if (f < F && t < Tn) {
                const T* s = src + ((b * Tn + t) * (int64_t)F + f) * 4;
                for (int c = 0; c < 4; ++c) tile[(c * 32 + tx) * 33 + r] = s[c];
            }
        }
        __syncthreads();
        for (int c = 0; c < 4; ++c)
            for (int r = ty; r < 32; r += 8) {
                const int f = f0 + r, t = t0 + tx;
                if (f < F && t < Tn) dst[(b * 4 + c) * plane + (int64_t)f * Tn + t] = tile[(c * 32 + r) * 33 + tx];
            }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
#ifndef ALSEP_F16_TU
extern "C" int alsep_stft_f16tu(alsep_ctx*, const alsep_plan*, const float*, int64_t, int64_t, int64_t, void*, int, int);
extern "C" int alsep_istft_f16tu(alsep_ctx*, const alsep_plan*, const void*, int, int, int64_t, float*, int64_t, int64_t, int64_t, int64_t,
                                 int64_t);
extern "C" int alsep_plan_supported_nfft(int n_fft) {
    switch (n_fft) {
#define X(N_) case N_:
        ALSEP_FOR_EACH_NFFT(X)
#undef X
        return 1;
        default: return 0;
    }
}

extern "C" int alsep_plan_create(alsep_ctx* ctx, int n_fft, int hop, int dim_f, int dim_t, alsep_plan** out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !out) return ALSEP_ERR_ARG;
    if (!alsep_plan_supported_nfft(n_fft)) return alsep_fail(ctx, ALSEP_ERR_ARG, "n_fft=%d has no FFT kernel", n_fft);
    if (hop <= 0 || hop > n_fft || dim_t < 2 || dim_f < 1 || dim_f > n_fft / 2 + 1)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "bad STFT geometry n_fft=%d hop=%d dim_f=%d dim_t=%d", n_fft, hop, dim_f, dim_t);
    const int64_t chunk = (int64_t)hop * (dim_t - 1);
    if (chunk <= n_fft / 2)     // torch.stft's reflect padding needs pad < length
        return alsep_fail(ctx, ALSEP_ERR_ARG, "chunk %lld must exceed n_fft/2=%d", (long long)chunk, n_fft / 2);
    const int Q = (n_fft + hop - 1) / hop;
    const bool regring = hop == 1024 && (n_fft % 1024 == 0 || n_fft == 7680);   // register-ring iSTFT: LDS holds one frame only
    if ((size_t)(regring ? n_fft : n_fft + Q * hop) * sizeof(float2) > 160 * 1024)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "n_fft=%d hop=%d needs more than 160 KiB of LDS", n_fft, hop);
    alsep_plan* p = new alsep_plan();
    p->ctx = ctx; p->n_fft = n_fft; p->hop = hop; p->dim_f = dim_f; p->dim_t = dim_t; p->chunk = (int)chunk;
    std::vector<float2> tw(n_fft);
    std::vector<float> win(n_fft);
    std::vector<double> w2(n_fft);
    for (int j = 0; j < n_fft; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)n_fft;
        tw[j] = make_float2((float)cos(a), (float)sin(a));
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)j / (double)n_fft);
        win[j] = (float)w;
        w2[j] = (double)win[j] * (double)win[j];
    }
    p->env_len = (int64_t)n_fft + (int64_t)hop * (dim_t - 1);
    std::vector<double> envd(p->env_len, 0.0);
    for (int t = 0; t < dim_t; ++t)
        for (int n = 0; n < n_fft; ++n) envd[(int64_t)t * hop + n] += w2[n];
    std::vector<float> env(p->env_len);
    for (int64_t i = 0; i < p->env_len; ++i) env[i] = (float)envd[i];
    if (hipMalloc((void**)&p->tw, sizeof(float2) * n_fft) != hipSuccess ||
        hipMalloc((void**)&p->win, sizeof(float) * n_fft) != hipSuccess ||
        hipMalloc((void**)&p->env, sizeof(float) * p->env_len) != hipSuccess) {
        alsep_plan_destroy(p);
        return alsep_fail(ctx, ALSEP_ERR_NOMEM, "plan tables: hipMalloc failed");
    }
    // blocking copies on purpose: the host vectors die at return
    if (hipMemcpy(p->tw, tw.data(), sizeof(float2) * n_fft, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->win, win.data(), sizeof(float) * n_fft, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->env, env.data(), sizeof(float) * p->env_len, hipMemcpyHostToDevice) != hipSuccess) {
        alsep_plan_destroy(p);
        return alsep_fail(ctx, ALSEP_ERR_HIP, "plan tables: hipMemcpy failed");
    }
    *out = p;
    return ALSEP_OK;
}

extern "C" int alsep_plan_destroy(alsep_plan* plan) {
    if (!plan) return ALSEP_OK;
    if (plan->tw) (void)hipFree(plan->tw);
    if (plan->win) (void)hipFree(plan->win);
    if (plan->env) (void)hipFree(plan->env);
    delete plan;
    return ALSEP_OK;
}
#endif  // !ALSEP_F16_TU (plans are dtype-free: main translation unit only)

// Hop-blocks per workgroup of the persistent iSTFT kernels.  A workgroup spends run + Q - 1 frames on `run` blocks
// (Q - 1 warm-up frames re-done by every workgroup), and the grid runs in ceil(workgroups / slots) rounds: pick the
// run with the smallest rounds x frames (one nearly full round beats two half-empty ones).
static int istft_pick_run(int n_blocks, int Q, int64_t n_chunks, int slots) {
    int best = 16;
    double best_cost = 1e30;
    for (int run = 1; run <= 64; ++run) {                    // (from 1: a single chunk of 801 frames -- the Roformers -- is one round of 801 short workgroups)
        const int64_t wgs = n_chunks * ((n_blocks + run - 1) / run);
        const int64_t rounds = (wgs + slots - 1) / slots;
        const double cost = (double)rounds * (run + Q - 1);
        if (cost < best_cost) { best_cost = cost; best = run; }
    }
    return best;
}

// ALSEP_STFT_R16=0 falls back to the generic multi-pass kernel for 4096 / 6144 (A/B timing, cross-check)
static int stft_r16_enabled() {
    static const int v = [] { const char* e = getenv("ALSEP_STFT_R16"); return e ? atoi(e) : 1; }();
    return v;
}

template <int N, typename OutT, int LAYOUT>
static int launch_stft(alsep_ctx* ctx, const alsep_plan* p, const float* pcm, int64_t ch_stride,
                       int64_t chunk_stride, int64_t n_chunks, void* spec) {
    const size_t lds = sizeof(float2) * N;
    if constexpr (N == 4096 || N == 6144 || N == 7680) {
        if (stft_r16_enabled()) {
            const size_t lds = r16::stft_lds_bytes<N / 256>();                            // default: one frame per two-wave workgroup
            ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)r16::stft_r16_kernel<N / 256, OutT, LAYOUT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ProfScope prof(ctx, ALSEP_PROF_STFT);
            for (int64_t b0 = 0; b0 < n_chunks; b0 += 32768) {
                const int64_t nb = std::min<int64_t>(32768, n_chunks - b0);
                const int64_t spec_off = b0 * 4 * (int64_t)p->dim_f * p->dim_t;
                hipLaunchKernelGGL((r16::stft_r16_kernel<N / 256, OutT, LAYOUT>), dim3(p->dim_t, (unsigned)nb),
                                   dim3(r16::kThreads), lds, ctx->stream, pcm + b0 * chunk_stride, ch_stride, chunk_stride,
                                   p->chunk, p->hop, p->dim_f, p->dim_t, (const float2*)p->tw, (OutT*)spec + spec_off, r16::FirstConvArgs{});
            }
            ALSEP_LAUNCH_CHECK(ctx, "stft_r16_kernel");
            return ALSEP_OK;
        }
    }
    ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)stft_kernel<N, OutT, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope prof(ctx, ALSEP_PROF_STFT);
    // grid.y is limited to 65535: split long batches
    for (int64_t b0 = 0; b0 < n_chunks; b0 += 32768) {
        const int64_t nb = std::min<int64_t>(32768, n_chunks - b0);
        const int64_t spec_off = b0 * 4 * (int64_t)p->dim_f * p->dim_t;
        hipLaunchKernelGGL((stft_kernel<N, OutT, LAYOUT>), dim3(p->dim_t, (unsigned)nb), dim3(kFftThreads), lds,
                           ctx->stream, pcm + b0 * chunk_stride, ch_stride, chunk_stride, p->chunk, p->hop,
                           p->dim_f, p->dim_t, (const float2*)p->tw, (const float*)p->win, (OutT*)spec + spec_off);
    }
    ALSEP_LAUNCH_CHECK(ctx, "stft_kernel");
    return ALSEP_OK;
}

// STFT with the network's first 1x1 convolution in its epilogue (r16::stft_r16_kernel<..., FUSE>): act [n_chunks][T][dim_f][48] in the
// half-precision storage type of this translation unit.  Internal: called by alsep_net_forward_pcm (tdfnet.hip).  Returns
// ALSEP_ERR_STATE for a geometry without a three-pass kernel (the caller then runs stft + first conv separately).
template <int N>
static int launch_stft_first_conv(alsep_ctx* ctx, const alsep_plan* p, const float* pcm, int64_t ch_stride, int64_t chunk_stride, int64_t n_chunks,
                                  bf16_t* act, const r16::FirstConvArgs& fc) {
    typedef bf16_t OutT;
    const size_t lds = r16::stft_lds_bytes<N / 256>();                    // >= dim_f * 8 bytes of rounded bins (dim_f <= N / 2)
    static_assert(sizeof(OutT) == 2, "stage geometry");
    ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)r16::stft_r16_kernel<N / 256, OutT, ALSEP_LAYOUT_NHWC, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope prof(ctx, ALSEP_PROF_STFT);
    prof.work(0.0, (double)n_chunks * (2.0 * p->chunk * 4 + (double)p->dim_f * p->dim_t * r16::kFirstConvG * sizeof(OutT)));
    for (int64_t b0 = 0; b0 < n_chunks; b0 += 32768) {
        const int64_t nb = std::min<int64_t>(32768, n_chunks - b0);
        const int64_t off = b0 * (int64_t)r16::kFirstConvG * p->dim_f * p->dim_t;
        const int64_t frames = nb * p->dim_t;
        if (N / 256 > 24) {
            // persistent (fft_r16.h, PERSIST): as many workgroups as are resident at once (LDS: two per CU at 7680 points), a multiple of 8
            int64_t g = (int64_t)device_cu_count(ctx) * (int64_t)((160 * 1024) / lds);
            g = g / 8 * 8;
            if (g > frames) g = (frames + 7) / 8 * 8;
            if (g < 8) g = 8;
            hipLaunchKernelGGL((r16::stft_r16_kernel<N / 256, OutT, ALSEP_LAYOUT_NHWC, true>), dim3((unsigned)g), dim3(r16::kThreads), lds,
                               ctx->stream, pcm + b0 * chunk_stride, ch_stride, chunk_stride, p->chunk, p->hop, p->dim_f, p->dim_t,
                               (const float2*)p->tw, act + off, fc, (int)frames);
        } else {
            hipLaunchKernelGGL((r16::stft_r16_kernel<N / 256, OutT, ALSEP_LAYOUT_NHWC, true>), dim3(p->dim_t, (unsigned)nb), dim3(r16::kThreads), lds,
                               ctx->stream, pcm + b0 * chunk_stride, ch_stride, chunk_stride, p->chunk, p->hop, p->dim_f, p->dim_t,
                               (const float2*)p->tw, act + off, fc, 0);
        }
    }
    ALSEP_LAUNCH_CHECK(ctx, "stft_first_conv_kernel");
    return ALSEP_OK;
}

int ALSEP_TU_NAME(alsep_stft_first_conv)(alsep_ctx* ctx, const alsep_plan* plan, const float* pcm, int64_t ch_stride, int64_t chunk_stride,
                                         int64_t n_chunks, void* act, const float* w, const float* scale, const float* shift, int g, float in_scale,
                                         int zero_low) {
    if (!ctx || !plan || !pcm || !act || !w || !scale || !shift) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_stft_first_conv: null argument");
    if (g != r16::kFirstConvG || plan->dim_f > plan->n_fft / 2 || !stft_r16_enabled()) return ALSEP_ERR_STATE;
    if (plan->n_fft != 4096 && plan->n_fft != 6144 && plan->n_fft != 7680) return ALSEP_ERR_STATE;
    if (n_chunks <= 0) return ALSEP_OK;
    const r16::FirstConvArgs fc{w, scale, shift, in_scale, zero_low};
    if (plan->n_fft == 4096) return launch_stft_first_conv<4096>(ctx, plan, pcm, ch_stride, chunk_stride, n_chunks, (bf16_t*)act, fc);
    if (plan->n_fft == 6144) return launch_stft_first_conv<6144>(ctx, plan, pcm, ch_stride, chunk_stride, n_chunks, (bf16_t*)act, fc);
    if (plan->n_fft == 7680) return launch_stft_first_conv<7680>(ctx, plan, pcm, ch_stride, chunk_stride, n_chunks, (bf16_t*)act, fc);
    return ALSEP_ERR_STATE;
}

extern "C" int ALSEP_TU_NAME(alsep_stft)(alsep_ctx* ctx, const alsep_plan* plan, const float* pcm, int64_t ch_stride,
                          int64_t chunk_stride, int64_t n_chunks, void* spec, int dtype, int layout) {
    ALSEP_ENTER(ctx);
#ifndef ALSEP_F16_TU
    if (dtype == ALSEP_F16) return alsep_stft_f16tu(ctx, plan, pcm, ch_stride, chunk_stride, n_chunks, spec, dtype, layout);
#endif
    if (!ctx || !plan || !pcm || !spec) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_stft: null argument");
    if (n_chunks == 0) return ALSEP_OK;
    if (n_chunks < 0 || (dtype != ALSEP_F32 && dtype != ALSEP_HALF_DTYPE) ||
        (layout != ALSEP_LAYOUT_REF && layout != ALSEP_LAYOUT_NHWC))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_stft: bad n_chunks/dtype/layout");
#define X(N_)                                                                                          \
    if (plan->n_fft == N_) {                                                                           \
        if (dtype == ALSEP_F32)                                                                        \
            return layout == ALSEP_LAYOUT_NHWC                                                         \
                       ? launch_stft<N_, float, ALSEP_LAYOUT_NHWC>(ctx, plan, pcm, ch_stride, chunk_stride, n_chunks, spec) \
                       : launch_stft<N_, float, ALSEP_LAYOUT_REF>(ctx, plan, pcm, ch_stride, chunk_stride, n_chunks, spec); \
        return layout == ALSEP_LAYOUT_NHWC                                                             \
                   ? launch_stft<N_, bf16_t, ALSEP_LAYOUT_NHWC>(ctx, plan, pcm, ch_stride, chunk_stride, n_chunks, spec) \
                   : launch_stft<N_, bf16_t, ALSEP_LAYOUT_REF>(ctx, plan, pcm, ch_stride, chunk_stride, n_chunks, spec); \
    }
    ALSEP_FOR_EACH_NFFT(X)
#undef X
    return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_stft: unsupported n_fft %d", plan->n_fft);
}


template <int N, typename InT, int LAYOUT>
static int launch_istft(alsep_ctx* ctx, const alsep_plan* p, const void* spec, int64_t n_chunks, float* out,
                        int64_t out_ch_stride, int64_t out_chunk_stride, int64_t keep_lo, int64_t keep_hi,
                        int64_t out_limit) {
    const int Q = (N + p->hop - 1) / p->hop;
    const int j_lo = (int)((keep_lo + N / 2) / p->hop);
    const int j_hi = (int)((keep_hi - 1 + N / 2) / p->hop) + 1;
    ProfScope prof(ctx, ALSEP_PROF_ISTFT);
    if constexpr (N == 4096 || N == 6144) {
        static const int r16_on = [] { const char* e = getenv("ALSEP_ISTFT_R16"); return e ? atoi(e) : 1; }();
        if (p->hop == 1024 && r16_on) {                      // production geometry: three-pass kernel
            constexpr int R2 = N / 256;
            const size_t lds_r = r16::istft_lds_bytes<R2>();
            const bool full = p->dim_f >= N / 2;                // production band: every bin below Nyquist is stored
            auto kern = full ? r16::istft_r16_kernel<R2, 8, InT, LAYOUT, true> : r16::istft_r16_kernel<R2, 8, InT, LAYOUT, false>;
            ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
            static const int run_env = [] { const char* e = getenv("ALSEP_ISTFT_RUN"); return e ? atoi(e) : 0; }();
            const int run = run_env > 0 ? run_env : istft_pick_run(j_hi - j_lo, Q, n_chunks, 3 * device_cu_count(ctx));
            const int groups_r = (j_hi - j_lo + run - 1) / run;
            for (int64_t b0 = 0; b0 < n_chunks; b0 += 32768) {
                const int64_t nb = std::min<int64_t>(32768, n_chunks - b0);
                const int64_t spec_off = b0 * 4 * (int64_t)p->dim_f * p->dim_t;
                hipLaunchKernelGGL(kern, dim3(groups_r, (unsigned)nb),
                                   dim3(r16::kThreads), lds_r, ctx->stream, (const InT*)spec + spec_off, p->dim_f, p->dim_t,
                                   (const float2*)p->tw, (const float*)p->env, j_lo, j_hi, run,
                                   out + b0 * out_chunk_stride, out_ch_stride, out_chunk_stride, keep_lo, keep_hi,
                                   out_limit - b0 * out_chunk_stride);
            }
            ALSEP_LAUNCH_CHECK(ctx, "istft_r16_kernel");
            return ALSEP_OK;
        }
    }
    if constexpr (N == 7680) {
        static const int r30_on = [] { const char* e = getenv("ALSEP_ISTFT_R16"); return e ? atoi(e) : 1; }();
        if (p->hop == 1024 && r30_on) {                      // the vocal models' geometry: three passes in the order 16, 16, 30
            const size_t lds_r = r16::istft_r30_lds_bytes();
            auto kern = r16::istft_r30_kernel<8, InT, LAYOUT>;
            ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
            static const int run_env = [] { const char* e = getenv("ALSEP_ISTFT_RUN"); return e ? atoi(e) : 0; }();
            const int run = run_env > 0 ? run_env : istft_pick_run(j_hi - j_lo, Q, n_chunks, 2 * device_cu_count(ctx));
            const int groups_r = (j_hi - j_lo + run - 1) / run;
            for (int64_t b0 = 0; b0 < n_chunks; b0 += 32768) {
                const int64_t nb = std::min<int64_t>(32768, n_chunks - b0);
                const int64_t spec_off = b0 * 4 * (int64_t)p->dim_f * p->dim_t;
                hipLaunchKernelGGL(kern, dim3(groups_r, (unsigned)nb), dim3(r16::kThreads), lds_r, ctx->stream, (const InT*)spec + spec_off,
                                   p->dim_f, p->dim_t, (const float2*)p->tw, (const float*)p->env, j_lo, j_hi, run,
                                   out + b0 * out_chunk_stride, out_ch_stride, out_chunk_stride, keep_lo, keep_hi,
                                   out_limit - b0 * out_chunk_stride);
            }
            ALSEP_LAUNCH_CHECK(ctx, "istft_r30_kernel");
            return ALSEP_OK;
        }
    }
    if constexpr (N % 1024 == 0 || N == 7680) {
        if (p->hop == 1024) {                                // production geometry: register ring, 3 workgroups per CU
            const size_t lds_r = sizeof(float2) * (size_t)N;
            ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)istft_regring_kernel<N, 1024, InT, LAYOUT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
            static const int run_env = [] { const char* e = getenv("ALSEP_ISTFT_RUN"); return e ? atoi(e) : 0; }();
            const int run = run_env > 0 ? run_env : istft_pick_run(j_hi - j_lo, Q, n_chunks, 3 * device_cu_count(ctx));
            const int groups_r = (j_hi - j_lo + run - 1) / run;
            for (int64_t b0 = 0; b0 < n_chunks; b0 += 32768) {
                const int64_t nb = std::min<int64_t>(32768, n_chunks - b0);
                const int64_t spec_off = b0 * 4 * (int64_t)p->dim_f * p->dim_t;
                hipLaunchKernelGGL((istft_regring_kernel<N, 1024, InT, LAYOUT>), dim3(groups_r, (unsigned)nb), dim3(kFftThreads),
                                   lds_r, ctx->stream, (const InT*)spec + spec_off, p->dim_f, p->dim_t, (const float2*)p->tw,
                                   (const float*)p->win, (const float*)p->env, j_lo, j_hi, run, out + b0 * out_chunk_stride,
                                   out_ch_stride, out_chunk_stride, keep_lo, keep_hi, out_limit - b0 * out_chunk_stride);
            }
            ALSEP_LAUNCH_CHECK(ctx, "istft_regring_kernel");
            return ALSEP_OK;
        }
    }
    const size_t lds = sizeof(float2) * (size_t)(N + Q * p->hop);
    ALSEP_HIP(ctx, hipFuncSetAttribute((const void*)istft_kernel<N, InT, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // hop-blocks per workgroup: chosen like the three-pass kernels' (a single chunk of 801 frames -- a Roformer's -- as 801 short workgroups
    // instead of 51 long ones on 256 CUs: the Q - 1 warm-up frames cost less than the idle CUs)
    const int slots_g = device_cu_count(ctx) * (int)std::max<size_t>(1, (160 * 1024) / lds);
    const int run_g = istft_pick_run(j_hi - j_lo, Q, n_chunks, slots_g);
    const int groups = (j_hi - j_lo + run_g - 1) / run_g;
    for (int64_t b0 = 0; b0 < n_chunks; b0 += 32768) {
        const int64_t nb = std::min<int64_t>(32768, n_chunks - b0);
        const int64_t spec_off = b0 * 4 * (int64_t)p->dim_f * p->dim_t;
        hipLaunchKernelGGL((istft_kernel<N, InT, LAYOUT>), dim3(groups, (unsigned)nb), dim3(kFftThreads), lds,
                           ctx->stream, (const InT*)spec + spec_off, p->hop, p->dim_f, p->dim_t,
                           (const float2*)p->tw, (const float*)p->win, (const float*)p->env, j_lo, j_hi, run_g,
                           out + b0 * out_chunk_stride, out_ch_stride, out_chunk_stride, keep_lo, keep_hi,
                           out_limit - b0 * out_chunk_stride);
    }
    ALSEP_LAUNCH_CHECK(ctx, "istft_kernel");
    return ALSEP_OK;
}

extern "C" int ALSEP_TU_NAME(alsep_istft)(alsep_ctx* ctx, const alsep_plan* plan, const void* spec, int dtype, int layout,
                           int64_t n_chunks, float* out, int64_t out_ch_stride, int64_t out_chunk_stride,
                           int64_t keep_lo, int64_t keep_hi, int64_t out_limit) {
    ALSEP_ENTER(ctx);
#ifndef ALSEP_F16_TU
    if (dtype == ALSEP_F16)
        return alsep_istft_f16tu(ctx, plan, spec, dtype, layout, n_chunks, out, out_ch_stride, out_chunk_stride, keep_lo, keep_hi, out_limit);
#endif
    if (!ctx || !plan || !spec || !out) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_istft: null argument");
    if (n_chunks == 0) return ALSEP_OK;
    if (n_chunks < 0 || keep_lo < 0 || keep_hi > plan->chunk || keep_lo >= keep_hi || out_limit <= 0 ||
        (dtype != ALSEP_F32 && dtype != ALSEP_HALF_DTYPE) || (layout != ALSEP_LAYOUT_REF && layout != ALSEP_LAYOUT_NHWC))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_istft: bad argument");
#define X(N_)                                                                                          \
    if (plan->n_fft == N_) {                                                                           \
        if (dtype == ALSEP_F32)                                                                        \
            return layout == ALSEP_LAYOUT_NHWC                                                         \
                       ? launch_istft<N_, float, ALSEP_LAYOUT_NHWC>(ctx, plan, spec, n_chunks, out, out_ch_stride, out_chunk_stride, keep_lo, keep_hi, out_limit) \
                       : launch_istft<N_, float, ALSEP_LAYOUT_REF>(ctx, plan, spec, n_chunks, out, out_ch_stride, out_chunk_stride, keep_lo, keep_hi, out_limit); \
        return layout == ALSEP_LAYOUT_NHWC                                                             \
                   ? launch_istft<N_, bf16_t, ALSEP_LAYOUT_NHWC>(ctx, plan, spec, n_chunks, out, out_ch_stride, out_chunk_stride, keep_lo, keep_hi, out_limit) \
                   : launch_istft<N_, bf16_t, ALSEP_LAYOUT_REF>(ctx, plan, spec, n_chunks, out, out_ch_stride, out_chunk_stride, keep_lo, keep_hi, out_limit); \
    }
    ALSEP_FOR_EACH_NFFT(X)
#undef X
    return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_istft: unsupported n_fft %d", plan->n_fft);
}

#ifndef ALSEP_F16_TU
extern "C" int alsep_spec_convert(alsep_ctx* ctx, const void* src, void* dst, int dtype, int src_layout,
                                  int64_t B, int64_t dim_f, int64_t T) {
    ALSEP_ENTER(ctx);
    if (!ctx || !src || !dst) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_spec_convert: null argument");
    if (B == 0) return ALSEP_OK;
    if (B < 0 || B > 65535 || dim_f <= 0 || T <= 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_spec_convert: bad shape");
    dim3 grid((unsigned)ceil_div64(dim_f, 32), (unsigned)ceil_div64(T, 32), (unsigned)B);
    const bool to_nhwc = src_layout == ALSEP_LAYOUT_REF;
    if (dtype == ALSEP_F32) {
        const size_t lds = 4 * 32 * 33 * sizeof(float);
        if (to_nhwc) hipLaunchKernelGGL((spec_convert_kernel<float, 1>), grid, dim3(256), lds, ctx->stream, (const float*)src, (float*)dst, (int)dim_f, (int)T);
        else hipLaunchKernelGGL((spec_convert_kernel<float, 0>), grid, dim3(256), lds, ctx->stream, (const float*)src, (float*)dst, (int)dim_f, (int)T);
    } else if (dtype == ALSEP_BF16 || dtype == ALSEP_F16) {      // a 16-bit transposing copy: the element type does not matter
        const size_t lds = 4 * 32 * 33 * sizeof(bf16_t);
        if (to_nhwc) hipLaunchKernelGGL((spec_convert_kernel<bf16_t, 1>), grid, dim3(256), lds, ctx->stream, (const bf16_t*)src, (bf16_t*)dst, (int)dim_f, (int)T);
        else hipLaunchKernelGGL((spec_convert_kernel<bf16_t, 0>), grid, dim3(256), lds, ctx->stream, (const bf16_t*)src, (bf16_t*)dst, (int)dim_f, (int)T);
    } else {
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_spec_convert: bad dtype");
    }
    ALSEP_LAUNCH_CHECK(ctx, "spec_convert_kernel");
    return ALSEP_OK;
}
#endif  // !ALSEP_F16_TU
