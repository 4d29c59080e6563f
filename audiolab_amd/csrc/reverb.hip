// Reverb impulse-response extraction on the GPU (reference handlers/reverb.py:112-172, called from
// modules/separator/stem_separator.py:822-829): the whole-track FFT work -- circular cross-correlation (fft_xcorr, :55-66) and Wiener
// deconvolution (:94-106) -- plus the decay envelope (:74-81).  gfx950 only.
//
// Arithmetic: double precision.  A 5-minute track is a 13-million-point deconvolution whose quotient conj(H) Y / (|H|^2 + eps) divides by
// spectral nulls of the dry signal; single precision has no bits to spare there, fp64 costs nothing that matters (one-off work per
// track, ~10 streaming passes over <= 1 GB), and every transform below is then exact to ~1e-13 against numpy's.
//
// FFT: out-of-place Stockham autosort passes in global memory, radix 8 / 4 / 2 with the butterflies in registers (one thread = one
// radix-R butterfly: R coalesced 16-byte loads a stride n / R apart, R stores).  A pass streams the array once, so a transform is
// log8(n) HBM passes -- 9 for 2^25 points = 9 GB of traffic, a few milliseconds; nothing here is worth staging through LDS.
// Twiddles come from sincospi on an exactly reduced integer fraction (no table, no accumulated angle error).
// Lengths that are not powers of two (the Wiener step transforms len(wet) points, an arbitrary number) go through Bluestein's chirp
// convolution on a power-of-two transform of >= 2 n - 1 points; the chirp angle pi j^2 / n is reduced as the integer j^2 mod 2n.
#include "alsep_common.h"

namespace {

struct alignas(16) cplx {
    double x, y;
};

constexpr int kRvThreads = 256;

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cplx cconj(cplx a) { return {a.x, -a.y}; }
// a * (sign * i)
__device__ __forceinline__ cplx cmul_si(cplx a, double sign) { return {-sign * a.y, sign * a.x}; }

// exp(sign * 2 pi i * num / den), 0 <= num < den: the fraction is exact in double (den <= 2^28)
__device__ __forceinline__ cplx unit_root(int64_t num, int64_t den, double sign) {
    double s, c;
    sincospi(2.0 * (double)num / (double)den, &s, &c);
    return {c, sign * s};
}

template <int R> __device__ __forceinline__ void dft_small(cplx (&v)[R], double sign);

template <> __device__ __forceinline__ void dft_small<2>(cplx (&v)[2], double) {
    const cplx a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

template <> __device__ __forceinline__ void dft_small<4>(cplx (&v)[4], double sign) {
    const cplx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]), t3 = cmul_si(csub(v[1], v[3]), sign);
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
}

template <> __device__ __forceinline__ void dft_small<8>(cplx (&v)[8], double sign) {
    cplx e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft_small<4>(e, sign);
    dft_small<4>(o, sign);
    const double h = 0.70710678118654752440;
    o[1] = cmul(o[1], cplx{h, sign * h});
    o[2] = cmul_si(o[2], sign);
    o[3] = cmul(o[3], cplx{-h, sign * h});
    for (int k = 0; k < 4; ++k) {
        v[k] = cadd(e[k], o[k]);
        v[k + 4] = csub(e[k], o[k]);
    }
}

// One Stockham pass of radix R over n points; ns = product of the radices of the passes before it (the length of the sub-transforms
// already done).  Butterfly j takes in[j + t n/R], t = 0 .. R-1, twiddles input t by w^(t k), k = j mod ns, w = exp(sign 2 pi i / (ns R)),
// and stores output u at out[(j - k) R + k + u ns].
template <int R>
__global__ void __launch_bounds__(kRvThreads)
fft_pass_kernel(const cplx* __restrict__ in, cplx* __restrict__ out, int64_t n, int64_t ns, double sign) {
    const int64_t m = n / R;
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < m; j += (int64_t)gridDim.x * kRvThreads) {
        const int64_t k = j & (ns - 1);
        cplx v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = in[j + t * m];
        if (ns > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = cmul(v[t], unit_root(((int64_t)t * k) % (ns * R), ns * R, sign));
        }
        dft_small<R>(v, sign);
        const int64_t j0 = (j - k) * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) out[j0 + t * ns] = v[t];
    }
}

unsigned rv_grid(int64_t work) {
    int64_t g = (work + kRvThreads - 1) / kRvThreads;
    if (g < 1) g = 1;
    if (g > 65536) g = 65536;
    return (unsigned)g;
}

// n = 2^log2n points, result returned in *res (either a or b: the passes ping-pong); unnormalised, sign -1 forward / +1 inverse
int fft_pow2(alsep_ctx* ctx, cplx* a, cplx* b, int log2n, double sign, cplx** res) {
    const int64_t n = (int64_t)1 << log2n;
    int64_t ns = 1;
    int left = log2n;
    cplx *src = a, *dst = b;
    while (left > 0) {
        const int r = left >= 3 ? 3 : left;             // 8, 8, ..., then 4 or 2
        const int64_t m = n >> r;
        if (r == 3) hipLaunchKernelGGL(fft_pass_kernel<8>, dim3(rv_grid(m)), dim3(kRvThreads), 0, ctx->stream, src, dst, n, ns, sign);
        else if (r == 2) hipLaunchKernelGGL(fft_pass_kernel<4>, dim3(rv_grid(m)), dim3(kRvThreads), 0, ctx->stream, src, dst, n, ns, sign);
        else hipLaunchKernelGGL(fft_pass_kernel<2>, dim3(rv_grid(m)), dim3(kRvThreads), 0, ctx->stream, src, dst, n, ns, sign);
        ALSEP_LAUNCH_CHECK(ctx, "fft_pass_kernel");
        ns <<= r;
        left -= r;
        cplx* t = src; src = dst; dst = t;
    }
    *res = src;
    return ALSEP_OK;
}

// ---- Bluestein: X[k] = w[k] sum_j (x[j] w[j]) conj(w)[k - j],  w[j] = exp(sign i pi j^2 / L) --------------------------------------
__device__ __forceinline__ cplx chirp(int64_t j, int64_t L, double sign) {
    const uint64_t q = ((uint64_t)j * (uint64_t)j) % (uint64_t)(2 * L);      // j < 2^31: j^2 < 2^62
    double s, c;
    sincospi((double)q / (double)L, &s, &c);
    return {c, sign * s};
}

__global__ void __launch_bounds__(kRvThreads)
chirp_in_kernel(const cplx* __restrict__ x, cplx* __restrict__ a, cplx* __restrict__ b, int64_t L, int64_t M, double sign) {
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < M; j += (int64_t)gridDim.x * kRvThreads) {
        cplx av = {0.0, 0.0}, bv = {0.0, 0.0};
        if (j < L) {
            const cplx w = chirp(j, L, sign);
            av = cmul(x[j], w);
            bv = cconj(w);
        } else if (M - j < L) {
            bv = cconj(chirp(M - j, L, sign));
        }
        a[j] = av;
        b[j] = bv;
    }
}

__global__ void __launch_bounds__(kRvThreads)
cmul_inplace_kernel(cplx* __restrict__ a, const cplx* __restrict__ b, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < n; j += (int64_t)gridDim.x * kRvThreads) a[j] = cmul(a[j], b[j]);
}

__global__ void __launch_bounds__(kRvThreads)
chirp_out_kernel(const cplx* __restrict__ c, cplx* __restrict__ X, int64_t L, int64_t M, double sign) {
    const double inv = 1.0 / (double)M;
    for (int64_t k = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; k < L; k += (int64_t)gridDim.x * kRvThreads) {
        const cplx v = cmul(chirp(k, L, sign), c[k]);
        X[k] = {v.x * inv, v.y * inv};
    }
}

int log2_ceil(int64_t v) {
    int l = 0;
    while (((int64_t)1 << l) < v) ++l;
    return l;
}

// ---- reverb.py pieces ------------------------------------------------------------------------------------------------------------
// to_mono (:52-53) of two [C, N] float32 tensors packed as re = a, im = b, zero beyond each signal's length.  np.mean(axis=1) on a
// float32 [N, C] array: float32 sum over the channels, divided by C in float32.
__device__ __forceinline__ float mono_at(const float* x, int c, int64_t n, int64_t ld, int64_t j) {
    if (j >= n) return 0.f;
    float s = x[j];
    for (int ch = 1; ch < c; ++ch) s += x[(int64_t)ch * ld + j];
    return c > 1 ? s / (float)c : s;
}

__global__ void __launch_bounds__(kRvThreads)
mono_pair_kernel(const float* __restrict__ a, int ca, int64_t na, int64_t lda, const float* __restrict__ b, int cb, int64_t nb, int64_t ldb,
                 cplx* __restrict__ z, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < n; j += (int64_t)gridDim.x * kRvThreads)
        z[j] = {(double)mono_at(a, ca, na, lda, j), (double)mono_at(b, cb, nb, ldb, j)};
}

// Z = DFT_n(a + i b) of two real signals -> their spectra by Hermitian symmetry: A[k] = (Z[k] + conj Z[n-k]) / 2,
// B[k] = (Z[k] - conj Z[n-k]) / (2 i)
__device__ __forceinline__ void split_spectra(cplx zk, cplx znk, cplx* A, cplx* B) {
    const cplx c = cconj(znk);
    *A = {0.5 * (zk.x + c.x), 0.5 * (zk.y + c.y)};
    const cplx d = csub(zk, c);                                             // d / (2 i) = (d.y, -d.x) / 2
    *B = {0.5 * d.y, -0.5 * d.x};
}

// fft_xcorr (:55-66): P[k] = A[k] conj(B[k]) over the full (Hermitian) spectrum, in place; thread k handles the pair (k, n - k)
__global__ void __launch_bounds__(kRvThreads)
xcorr_product_kernel(cplx* __restrict__ z, int64_t n) {
    for (int64_t k = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; k <= n / 2; k += (int64_t)gridDim.x * kRvThreads) {
        const int64_t nk = (n - k) % n;
        cplx A, B;
        split_spectra(z[k], z[nk], &A, &B);
        const cplx p = cmul(A, cconj(B));
        z[k] = p;
        if (nk != k) z[nk] = cconj(p);
    }
}

// wiener_deconvolution (:94-106): Z = DFT_n(y + i h) -> S = the Hermitian spectrum numpy's irfft builds from
// G[k] = conj(H) Y / (|H|^2 + eps), k = 0 .. n/2, for its default output length m = 2 (n/2): bins 0 and m/2 contribute their real
// parts only
__global__ void __launch_bounds__(kRvThreads)
wiener_kernel(const cplx* __restrict__ z, int64_t n, double eps, cplx* __restrict__ s, int64_t m) {
    for (int64_t k = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; k <= m / 2; k += (int64_t)gridDim.x * kRvThreads) {
        cplx Y, H;
        split_spectra(z[k], z[(n - k) % n], &Y, &H);
        const cplx num = cmul(cconj(H), Y);
        const double den = H.x * H.x + H.y * H.y + eps;
        cplx g = {num.x / den, num.y / den};
        if (k == 0 || 2 * k == m) {
            g.y = 0.0;
            s[k] = g;
        } else {
            s[k] = g;
            s[m - k] = cconj(g);
        }
    }
}

// first index of the largest real part over [0, n): per-block candidates, the host picks among them
__global__ void __launch_bounds__(kRvThreads)
argmax_real_kernel(const cplx* __restrict__ z, int64_t n, double* __restrict__ best_val, int64_t* __restrict__ best_idx) {
    double* sv = (double*)alsep_smem;
    int64_t* si = (int64_t*)(alsep_smem + kRvThreads * sizeof(double));
    double bv = -1.0e300;
    int64_t bi = n;
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < n; j += (int64_t)gridDim.x * kRvThreads) {
        const double v = z[j].x;
        if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = kRvThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double ov = sv[threadIdx.x + s];
            const int64_t oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { best_val[blockIdx.x] = sv[0]; best_idx[blockIdx.x] = si[0]; }
}

__global__ void __launch_bounds__(kRvThreads)
real_scaled_kernel(const cplx* __restrict__ z, double scale, double* __restrict__ out, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < n; j += (int64_t)gridDim.x * kRvThreads) out[j] = z[j].x * scale;
}

__global__ void __launch_bounds__(kRvThreads)
magnitude_kernel(const cplx* __restrict__ s, double* __restrict__ out, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < n; j += (int64_t)gridDim.x * kRvThreads) out[j] = hypot(s[j].x, s[j].y);
}

__global__ void __launch_bounds__(kRvThreads)
real_to_cplx_kernel(const double* __restrict__ x, cplx* __restrict__ z, int64_t n) {
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < n; j += (int64_t)gridDim.x * kRvThreads) z[j] = {x[j], 0.0};
}

// estimate_rt60's curve (:74-81) in the reference's float32 arithmetic: 20 log10(sqrt(sum_c x^2) + 1e-10), or |x| + 1e-10 for a
// one-channel signal
__global__ void __launch_bounds__(kRvThreads)
envelope_db_kernel(const float* __restrict__ x, int c, int64_t n, int64_t ld, float* __restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; j < n; j += (int64_t)gridDim.x * kRvThreads) {
        float env;
        if (c == 1) {
            env = fabsf(x[j]);
        } else {
            float s = x[j] * x[j];
            for (int ch = 1; ch < c; ++ch) { const float v = x[(int64_t)ch * ld + j]; s += v * v; }
            env = sqrtf(s);
        }
        out[j] = 20.0f * log10f(env + 1e-10f);
    }
}

int dft_impl(alsep_ctx* ctx, const cplx* in, cplx* out, int64_t n, double sign, cplx* ws) {
    const int l2 = log2_ceil(n);
    if (((int64_t)1 << l2) == n) {                                          // power of two: straight Stockham, ws = n points
        ALSEP_HIP(ctx, hipMemcpyAsync(out, in, (size_t)n * sizeof(cplx), hipMemcpyDeviceToDevice, ctx->stream));
        cplx* res = nullptr;
        if (int rc = fft_pow2(ctx, out, ws, l2, sign, &res)) return rc;
        if (res != out) ALSEP_HIP(ctx, hipMemcpyAsync(out, res, (size_t)n * sizeof(cplx), hipMemcpyDeviceToDevice, ctx->stream));
        return ALSEP_OK;
    }
    const int lm = log2_ceil(2 * n - 1);
    const int64_t M = (int64_t)1 << lm;
    cplx *a = ws, *b = ws + M, *t = ws + 2 * M;
    hipLaunchKernelGGL(chirp_in_kernel, dim3(rv_grid(M)), dim3(kRvThreads), 0, ctx->stream, in, a, b, n, M, sign);
    ALSEP_LAUNCH_CHECK(ctx, "chirp_in_kernel");
    cplx *fa = nullptr, *fb = nullptr, *fc = nullptr;
    if (int rc = fft_pow2(ctx, a, t, lm, -1.0, &fa)) return rc;
    cplx* spare_a = fa == a ? t : a;                                        // the buffer of {a, t} that does not hold fa
    if (int rc = fft_pow2(ctx, b, spare_a, lm, -1.0, &fb)) return rc;
    cplx* spare_b = fb == b ? spare_a : b;
    hipLaunchKernelGGL(cmul_inplace_kernel, dim3(rv_grid(M)), dim3(kRvThreads), 0, ctx->stream, fa, fb, M);
    ALSEP_LAUNCH_CHECK(ctx, "cmul_inplace_kernel");
    if (int rc = fft_pow2(ctx, fa, spare_b, lm, +1.0, &fc)) return rc;
    (void)fb;
    hipLaunchKernelGGL(chirp_out_kernel, dim3(rv_grid(n)), dim3(kRvThreads), 0, ctx->stream, fc, out, n, M, sign);
    ALSEP_LAUNCH_CHECK(ctx, "chirp_out_kernel");
    return ALSEP_OK;
}

// ---- scipy.signal.resample (Fourier method), what librosa.resample(res_type="scipy") runs: the VR band chain going up (spec_utils.py
// :427).  Two real rows travel as one complex signal (the operator is real and linear, so the complex-input branch of the scipy routine
// applied to L + i R resamples both).
__global__ void __launch_bounds__(kRvThreads)
pack_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, cplx* __restrict__ z, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kRvThreads)
        z[i] = cplx{(double)a[i], b ? (double)b[i] : 0.0};
}

// Y (length num) from X (length nx), the spectrum copy of scipy.signal.resample's complex branch: N = min(num, nx), the N/2 + 1 lowest
// non-negative frequencies, the N - (N/2 + 1) negative ones at the end, and the Nyquist rule for even N (upsampling: halved and mirrored
// to -N/2; downsampling: X[-N/2] added to the bin at -N/2).
__global__ void __launch_bounds__(kRvThreads)
resample_spectrum_kernel(const cplx* __restrict__ X, cplx* __restrict__ Y, int64_t nx, int64_t num) {
    const int64_t N = num < nx ? num : nx, nyq = N / 2 + 1, neg = N - nyq;   // Y[num - neg ..] = X[nx - neg ..]
    for (int64_t k = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; k < num; k += (int64_t)gridDim.x * kRvThreads) {
        cplx v{0.0, 0.0};
        if (k < nyq) v = X[k];
        else if (N > 2 && k >= num - neg) v = X[nx - (num - k)];
        if (N % 2 == 0) {
            if (num < nx) {                                                  // index -N/2 of Y is num - N/2 = N/2 here (num == N)
                if (k == num - N / 2) { const cplx e = X[nx - N / 2]; v.x += e.x; v.y += e.y; }
            } else if (nx < num) {
                if (k == N / 2) { v.x *= 0.5; v.y *= 0.5; }
                if (k == num - N / 2) { v = X[N / 2]; v.x *= 0.5; v.y *= 0.5; }
            }
        }
        Y[k] = v;
    }
}

__global__ void __launch_bounds__(kRvThreads)
unpack_rows_kernel(const cplx* __restrict__ z, float* __restrict__ a, float* __restrict__ b, int64_t n, double scale) {
    for (int64_t i = (int64_t)blockIdx.x * kRvThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kRvThreads) {
        a[i] = (float)(z[i].x * scale);
        if (b) b[i] = (float)(z[i].y * scale);
    }
}

int64_t dft_ws_points(int64_t n) {
    const int l2 = log2_ceil(n);
    if (((int64_t)1 << l2) == n) return n;
    return 3 * ((int64_t)1 << log2_ceil(2 * n - 1));
}

constexpr int64_t kMaxPoints = (int64_t)1 << 27;                            // 2^27 complex doubles = 2 GiB per buffer

}  // namespace

extern "C" int64_t alsep_dft_f64_workspace_bytes(int64_t n) {
    if (n <= 0 || n > kMaxPoints) return -1;
    return dft_ws_points(n) * (int64_t)sizeof(cplx);
}

extern "C" int alsep_dft_f64(alsep_ctx* ctx, const double* in, double* out, int64_t n, int inverse, void* ws, int64_t ws_bytes) {
    ALSEP_ENTER(ctx);
    if (!ctx || !in || !out || !ws || n <= 0 || n > kMaxPoints || in == out) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_dft_f64: bad argument");
    if (ws_bytes < dft_ws_points(n) * (int64_t)sizeof(cplx)) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_dft_f64: workspace too small");
    return dft_impl(ctx, (const cplx*)in, (cplx*)out, n, inverse ? +1.0 : -1.0, (cplx*)ws);
}

extern "C" int64_t alsep_reverb_workspace_bytes(int64_t n_wet, int64_t n_dry) {
    if (n_wet <= 0 || n_dry <= 0) return -1;
    const int64_t nx = (int64_t)1 << log2_ceil(n_wet + n_dry - 1);          // fft_xcorr's transform length
    if (nx > kMaxPoints || 2 * n_wet > kMaxPoints) return -1;
    const int64_t pts_x = 2 * nx;                                           // z + ping-pong
    const int64_t pts_w = 2 * n_wet + dft_ws_points(n_wet);                 // z, spectrum / result, DFT workspace
    const int64_t pts = pts_x > pts_w ? pts_x : pts_w;
    return pts * (int64_t)sizeof(cplx) + 65536 * 16;                        // + the argmax candidates
}

// fft_xcorr + argmax (reverb.py:55-66, :130-131): index of the first maximum of the circular cross-correlation of the two mono signals
// over its first n_wet + n_dry - 1 entries.  wet / dry: float32 [C, n] device tensors with row strides ld.
extern "C" int alsep_reverb_xcorr_argmax(alsep_ctx* ctx, const float* wet, int c_wet, int64_t n_wet, int64_t ld_wet, const float* dry, int c_dry,
                                         int64_t n_dry, int64_t ld_dry, void* ws, int64_t ws_bytes, int64_t* argmax_out, double* probe_out,
                                         const int64_t* probe_idx, int n_probe) {
    ALSEP_ENTER(ctx);
    if (!ctx || !wet || !dry || !ws || !argmax_out || c_wet < 1 || c_dry < 1 || n_wet < 1 || n_dry < 1 || ld_wet < n_wet || ld_dry < n_dry)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_reverb_xcorr_argmax: bad argument");
    const int64_t need = alsep_reverb_workspace_bytes(n_wet, n_dry);
    if (need < 0 || ws_bytes < need) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_reverb_xcorr_argmax: workspace too small (or track too long)");
    const int64_t N = n_wet + n_dry - 1;
    // probe indices are checked before anything is enqueued
    const int n_pr = (n_probe > 0 && probe_idx && probe_out) ? n_probe : 0;
    for (int i = 0; i < n_pr; ++i)
        if (probe_idx[i] < 0 || probe_idx[i] >= N) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_reverb_xcorr_argmax: probe index out of range");
    std::vector<cplx> pv(n_pr);
    const int lx = log2_ceil(N);
    const int64_t nx = (int64_t)1 << lx;
    cplx* z = (cplx*)ws;
    cplx* t = z + nx;
    hipLaunchKernelGGL(mono_pair_kernel, dim3(rv_grid(nx)), dim3(kRvThreads), 0, ctx->stream, wet, c_wet, n_wet, ld_wet, dry, c_dry, n_dry, ld_dry, z, nx);
    ALSEP_LAUNCH_CHECK(ctx, "mono_pair_kernel");
    cplx* f = nullptr;
    if (int rc = fft_pow2(ctx, z, t, lx, -1.0, &f)) return rc;
    hipLaunchKernelGGL(xcorr_product_kernel, dim3(rv_grid(nx / 2 + 1)), dim3(kRvThreads), 0, ctx->stream, f, nx);
    ALSEP_LAUNCH_CHECK(ctx, "xcorr_product_kernel");
    cplx* corr = nullptr;
    if (int rc = fft_pow2(ctx, f, f == z ? t : z, lx, +1.0, &corr)) return rc;
    // candidates behind the two transform buffers
    const unsigned blocks = rv_grid(N) > 1024 ? 1024 : rv_grid(N);
    double* cand_v = (double*)((cplx*)ws + 2 * nx);
    int64_t* cand_i = (int64_t*)(cand_v + 1024);
    hipLaunchKernelGGL(argmax_real_kernel, dim3(blocks), dim3(kRvThreads), kRvThreads * 16, ctx->stream, corr, N, cand_v, cand_i);
    ALSEP_LAUNCH_CHECK(ctx, "argmax_real_kernel");
    // host landing buffers live on the heap and every path out of here passes the stream synchronisation below, so no pending copy
    // ever targets a dead frame
    std::vector<double> hv(blocks);
    std::vector<int64_t> hi(blocks);
    hipError_t e = hipMemcpyAsync(hv.data(), cand_v, blocks * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hi.data(), cand_i, blocks * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
    for (int i = 0; i < n_pr && e == hipSuccess; ++i)
        e = hipMemcpyAsync(&pv[i], corr + probe_idx[i], sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess || es != hipSuccess)
        return alsep_fail(ctx, ALSEP_ERR_HIP, "alsep_reverb_xcorr_argmax: copy back failed: %s", hipGetErrorString(e != hipSuccess ? e : es));
    double bv = hv[0];
    int64_t bi = hi[0];
    for (unsigned b = 1; b < blocks; ++b)
        if (hv[b] > bv || (hv[b] == bv && hi[b] < bi)) { bv = hv[b]; bi = hi[b]; }
    *argmax_out = bi;
    for (int i = 0; i < n_pr; ++i) probe_out[i] = pv[i].x / (double)nx;      // irfft's 1 / n
    return ALSEP_OK;
}

// wiener_deconvolution(wet_mono, dry_mono, eps)[:n_out] (reverb.py:94-106, :142-143) -> ir_out (device, double).  The result has
// 2 * (n_wet / 2) samples (numpy's irfft default length); n_out is clipped to it and returned through n_written.
extern "C" int alsep_reverb_wiener_ir(alsep_ctx* ctx, const float* wet, int c_wet, int64_t n_wet, int64_t ld_wet, const float* dry, int c_dry,
                                      int64_t n_dry, int64_t ld_dry, double eps, void* ws, int64_t ws_bytes, double* ir_out, int64_t n_out,
                                      int64_t* n_written) {
    ALSEP_ENTER(ctx);
    if (!ctx || !wet || !dry || !ws || !ir_out || !n_written || c_wet < 1 || c_dry < 1 || n_wet < 2 || n_dry < 1 || n_out < 1 ||
        ld_wet < n_wet || ld_dry < n_dry)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_reverb_wiener_ir: bad argument");
    const int64_t need = alsep_reverb_workspace_bytes(n_wet, n_dry);
    if (need < 0 || ws_bytes < need) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_reverb_wiener_ir: workspace too small (or track too long)");
    const int64_t n = n_wet, m = 2 * (n / 2);
    cplx* z = (cplx*)ws;
    cplx* s = z + n;
    cplx* dws = s + n;
    // rfft(kernel, len(signal)) truncates or zero-pads the dry signal to the wet length
    hipLaunchKernelGGL(mono_pair_kernel, dim3(rv_grid(n)), dim3(kRvThreads), 0, ctx->stream, wet, c_wet, n_wet, ld_wet, dry, c_dry,
                       n_dry < n ? n_dry : n, ld_dry, z, n);
    ALSEP_LAUNCH_CHECK(ctx, "mono_pair_kernel");
    if (int rc = dft_impl(ctx, z, s, n, -1.0, dws)) return rc;               // s = DFT_n(y + i h)
    hipLaunchKernelGGL(wiener_kernel, dim3(rv_grid(m / 2 + 1)), dim3(kRvThreads), 0, ctx->stream, s, n, eps, z, m);   // z = Hermitian quotient
    ALSEP_LAUNCH_CHECK(ctx, "wiener_kernel");
    if (int rc = dft_impl(ctx, z, s, m, +1.0, dws)) return rc;               // s = m * irfft
    const int64_t k = n_out < m ? n_out : m;
    hipLaunchKernelGGL(real_scaled_kernel, dim3(rv_grid(k)), dim3(kRvThreads), 0, ctx->stream, s, 1.0 / (double)m, ir_out, k);
    ALSEP_LAUNCH_CHECK(ctx, "real_scaled_kernel");
    *n_written = k;
    return ALSEP_OK;
}

// |rfft(x)| of a real double signal of any length (the spectral centroid of reverb.py:155-157 takes it of the <= 2 s impulse response):
// mag_out[k], k = 0 .. n/2
extern "C" int alsep_rfft_mag_f64(alsep_ctx* ctx, const double* x, int64_t n, void* ws, int64_t ws_bytes, double* mag_out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !ws || !mag_out || n < 1 || n > kMaxPoints) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_rfft_mag_f64: bad argument");
    if (ws_bytes < (2 * n + dft_ws_points(n)) * (int64_t)sizeof(cplx)) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_rfft_mag_f64: workspace too small");
    cplx* z = (cplx*)ws;
    cplx* s = z + n;
    hipLaunchKernelGGL(real_to_cplx_kernel, dim3(rv_grid(n)), dim3(kRvThreads), 0, ctx->stream, x, z, n);
    ALSEP_LAUNCH_CHECK(ctx, "real_to_cplx_kernel");
    if (int rc = dft_impl(ctx, z, s, n, -1.0, s + n)) return rc;
    hipLaunchKernelGGL(magnitude_kernel, dim3(rv_grid(n / 2 + 1)), dim3(kRvThreads), 0, ctx->stream, s, mag_out, n / 2 + 1);
    ALSEP_LAUNCH_CHECK(ctx, "rfft_mag_kernel");
    return ALSEP_OK;
}

static int64_t resample_fft_points(int64_t n_in, int64_t n_out) {
    const int64_t a = dft_ws_points(n_in), b = dft_ws_points(n_out);
    return 2 * n_in + 2 * n_out + (a > b ? a : b);
}

extern "C" int64_t alsep_resample_fft_workspace_bytes(int64_t n_in, int64_t n_out) {
    if (n_in <= 0 || n_out <= 0 || n_in > kMaxPoints || n_out > kMaxPoints) return -1;
    return resample_fft_points(n_in, n_out) * (int64_t)sizeof(cplx);
}

// x [rows, n_in] (row stride ldx) -> y [rows, n_out] (row stride ldy), float32; rows in pairs through one complex transform each
extern "C" int alsep_resample_fft(alsep_ctx* ctx, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t n_in, int64_t n_out,
                                  void* ws, int64_t ws_bytes) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y || !ws || rows <= 0 || n_in <= 0 || n_out <= 0 || n_in > kMaxPoints || n_out > kMaxPoints || ldx < n_in || ldy < n_out)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_resample_fft: bad argument");
    if (ws_bytes < resample_fft_points(n_in, n_out) * (int64_t)sizeof(cplx)) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_resample_fft: workspace too small");
    cplx* z = (cplx*)ws;                 // n_in: packed input
    cplx* X = z + n_in;                  // n_in: its spectrum
    cplx* Y = X + n_in;                  // n_out: resampled spectrum
    cplx* o = Y + n_out;                 // n_out: inverse transform
    cplx* dws = o + n_out;
    for (int64_t r = 0; r < rows; r += 2) {
        const float* a = x + r * ldx;
        const float* b = r + 1 < rows ? x + (r + 1) * ldx : nullptr;
        hipLaunchKernelGGL(pack_rows_kernel, dim3(rv_grid(n_in)), dim3(kRvThreads), 0, ctx->stream, a, b, z, n_in);
        ALSEP_LAUNCH_CHECK(ctx, "pack_rows_kernel");
        if (int rc = dft_impl(ctx, z, X, n_in, -1.0, dws)) return rc;
        hipLaunchKernelGGL(resample_spectrum_kernel, dim3(rv_grid(n_out)), dim3(kRvThreads), 0, ctx->stream, X, Y, n_in, n_out);
        ALSEP_LAUNCH_CHECK(ctx, "resample_spectrum_kernel");
        if (int rc = dft_impl(ctx, Y, o, n_out, +1.0, dws)) return rc;
        // ifft (1 / n_out) times n_out / n_in
        hipLaunchKernelGGL(unpack_rows_kernel, dim3(rv_grid(n_out)), dim3(kRvThreads), 0, ctx->stream, o, y + r * ldy,
                           r + 1 < rows ? y + (r + 1) * ldy : nullptr, n_out, 1.0 / (double)n_in);
        ALSEP_LAUNCH_CHECK(ctx, "unpack_rows_kernel");
    }
    return ALSEP_OK;
}

// the decay curve estimate_rt60 fits (reverb.py:74-81), float32 as the reference computes it: x [C, n] -> out [n]
extern "C" int alsep_reverb_envelope_db(alsep_ctx* ctx, const float* x, int c, int64_t n, int64_t ld, float* out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !out || c < 1 || n < 1 || ld < n) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_reverb_envelope_db: bad argument");
    hipLaunchKernelGGL(envelope_db_kernel, dim3(rv_grid(n)), dim3(kRvThreads), 0, ctx->stream, x, c, n, ld, out);
    ALSEP_LAUNCH_CHECK(ctx, "envelope_db_kernel");
    return ALSEP_OK;
}
