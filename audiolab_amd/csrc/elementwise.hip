// Element-wise and reduction ops of the ensemble stage and the MDX runner (HBM-bound).
// Reference semantics: modules/separator/stem_separator.py:241-262 (_blend_tracks),
// :173-239 (_residual_subtract), :415-456 (de-bleed); mdxnet.py:168-173 (denoise average).
#include "alsep_common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;      // grid-stride above this (cdna_hip_programming.md Guideline 11)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(kThreads)
axpby_kernel(float a, const float* __restrict__ x, float b, float* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float4 yv = reinterpret_cast<float4*>(y)[i];
        yv.x = a * xv.x + b * yv.x; yv.y = a * xv.y + b * yv.y;
        yv.z = a * xv.z + b * yv.z; yv.w = a * xv.w + b * yv.w;
        reinterpret_cast<float4*>(y)[i] = yv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}

// |x| >= 0 so the float bit pattern orders like an unsigned integer: one atomicMax per block.
__global__ void __launch_bounds__(kThreads)
peak_abs_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out_bits) {
    float* red = reinterpret_cast<float*>(alsep_smem);
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const float v = fabsf(x[i]);
        m = v > m ? v : m;                                  // NaN-ignoring like np.max? (np.max propagates; inputs are finite)
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(out_bits, __float_as_uint(r));
    }
}

__global__ void __launch_bounds__(kThreads)
scale_by_device_kernel(float* __restrict__ y, int64_t n, float num, const float* __restrict__ den, float floor_) {
    const float d = *den;
    const float s = num / (d > floor_ ? d : floor_);
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) y[i] *= s;
}

// per-block partial sums in double, finished by one block in a fixed order (deterministic)
__global__ void __launch_bounds__(kThreads)
dot3_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, double* __restrict__ part) {
    double* red = reinterpret_cast<double*>(alsep_smem);    // [3][4]
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    double sab = 0, saa = 0, sbb = 0;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const double av = a[i], bv = b[i];
        sab += av * bv; saa += av * av; sbb += bv * bv;
    }
    sab = wave_sum(sab); saa = wave_sum(saa); sbb = wave_sum(sbb);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = sab; red[4 + w] = saa; red[8 + w] = sbb; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double* r = red + 4 * threadIdx.x;
        part[(int64_t)blockIdx.x * 3 + threadIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
    }
}
__global__ void __launch_bounds__(64)
dot3_final_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
    if (threadIdx.x < 3) {
        double s = 0;
        for (int i = 0; i < nblocks; ++i) s += part[(int64_t)i * 3 + threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// one block per lag; double accumulation
__global__ void __launch_bounds__(kThreads)
xcorr_window_kernel(const float* __restrict__ ref, const float* __restrict__ sig, int64_t probe, int max_shift,
                    double* __restrict__ corr) {
    double* red = reinterpret_cast<double*>(alsep_smem);
    const int lag = (int)blockIdx.x - max_shift;
    const int64_t lo = lag >= 0 ? 0 : -lag;                 // n range so that 0 <= n+lag < probe
    const int64_t hi = lag >= 0 ? probe - lag : probe;
    double s = 0;
    for (int64_t nidx = lo + threadIdx.x; nidx < hi; nidx += kThreads) s += (double)ref[nidx + lag] * (double)sig[nidx];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) corr[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(kThreads)
shift_subtract_kernel(const float* __restrict__ ref, const float* __restrict__ sig, int64_t len, int lag, float alpha,
                      float* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < len; i += stride) {
        const int64_t j = i - lag;
        const float s = (j >= 0 && j < len) ? sig[j] : 0.f;
        float r = ref[i] - alpha * s;
        if (!(fabsf(r) <= 3.4028234e38f)) r = 0.f;          // nan_to_num(nan=0, +-inf=0), stem_separator.py:237-238
        out[i] = r;
    }
}

// Hann-window overlap-add of model chunks (the chunker of audio-separator's MDXSeparator.demix; see
// oracle/mdx_oracle.py demix_ola -- parity unpinned).  Gather form, no atomics: output sample p sums
// the chunks that cover it, each weighted by np.hanning(n_act_b)[p - start_b] (n_act_b = chunk, or less
// for chunks cut by the end of the padded mixture), and is divided by the sum of the weights.
__global__ void __launch_bounds__(kThreads)
ola_combine_kernel(const float* __restrict__ chunks, int64_t n_chunks, int64_t chunk, int64_t step, int64_t total,
                   int use_window, float gain, float* __restrict__ out, int64_t out_stride, int64_t p_lo, int64_t n_out) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_out; i += stride) {
        const int64_t p = p_lo + i;
        int64_t b_hi = p / step;
        if (b_hi > n_chunks - 1) b_hi = n_chunks - 1;
        int64_t b_lo = (p - chunk + step) / step;           // ceil((p - chunk + 1) / step)
        if (p - chunk + 1 <= 0) b_lo = 0;
        float accl = 0.f, accr = 0.f, div = 0.f;
        for (int64_t b = b_lo; b <= b_hi; ++b) {
            const int64_t start = b * step, j = p - start;
            int64_t n_act = total - start;
            if (n_act > chunk) n_act = chunk;
            if (j < 0 || j >= n_act) continue;
            float w = 1.f;
            if (use_window) w = n_act > 1 ? 0.5f - 0.5f * cosf(6.28318530717958647692f * (float)j / (float)(n_act - 1)) : 1.f;
            accl += w * chunks[(b * 2 + 0) * chunk + j];
            accr += w * chunks[(b * 2 + 1) * chunk + j];
            div += w;
        }
        out[i] = gain * (accl / div);                       // div == 0 only where np.hanning is 0 on every cover: NaN as the reference
        out[out_stride + i] = gain * (accr / div);
    }
}

// Sharded form of ola_combine: `chunks` holds only the chunks [b0, b1) of this rank; part [3][n_out] receives the RAW weighted sums
// of both channels and the summed weights over those chunks (zero where none of them covers).  The ranks' parts are summed by one
// collective and divided afterwards: at a shard seam the chunks of two ranks overlap.
__global__ void __launch_bounds__(kThreads)
ola_partial_kernel(const float* __restrict__ chunks, int64_t b0, int64_t b1, int64_t chunk, int64_t step, int64_t total,
                   int use_window, float* __restrict__ part, int64_t p_lo, int64_t n_out) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_out; i += stride) {
        const int64_t p = p_lo + i;
        int64_t b_hi = p / step;
        if (b_hi > b1 - 1) b_hi = b1 - 1;
        int64_t b_lo = (p - chunk + step) / step;
        if (p - chunk + 1 <= 0) b_lo = 0;
        if (b_lo < b0) b_lo = b0;
        float accl = 0.f, accr = 0.f, div = 0.f;
        for (int64_t b = b_lo; b <= b_hi; ++b) {
            const int64_t start = b * step, j = p - start;
            int64_t n_act = total - start;
            if (n_act > chunk) n_act = chunk;
            if (j < 0 || j >= n_act) continue;
            float w = 1.f;
            if (use_window) w = n_act > 1 ? 0.5f - 0.5f * cosf(6.28318530717958647692f * (float)j / (float)(n_act - 1)) : 1.f;
            accl += w * chunks[((b - b0) * 2 + 0) * chunk + j];
            accr += w * chunks[((b - b0) * 2 + 1) * chunk + j];
            div += w;
        }
        part[i] = accl;
        part[n_out + i] = accr;
        part[2 * n_out + i] = div;
    }
}
// out[c][i] = gain * part[c][i] / part[2][i]
__global__ void __launch_bounds__(kThreads)
ola_finish_kernel(const float* __restrict__ part, float gain, float* __restrict__ out, int64_t out_stride, int64_t n_out) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n_out; i += stride) {
        const float d = part[2 * n_out + i];
        out[i] = gain * (part[i] / d);
        out[out_stride + i] = gain * (part[n_out + i] / d);
    }
}

// Band-limited sample-rate conversion by a Kaiser-windowed sinc (what librosa.load(sr=44100) does to a 48 kHz input at
// stem_separator.py:865 -- there with libsoxr, a dependency that is not in /root/reference: the filter below is this build's own,
// PARITY UNPINNED).  y[r][m] = sum_n x[r][n] h(m / ratio - n), h(t) = fc sinc(fc t) kaiser(t / hw; beta), fc = min(1, ratio) * rolloff,
// support |t| <= hw = zeros / fc input samples.  Positions in fp64 (a 60-minute track has 1.7e8 samples).
__device__ __forceinline__ double bessel_i0(double x) {
    double s = 1.0, t = 1.0;
    const double q = x * x * 0.25;
    for (int k = 1; k < 40; ++k) {
        t *= q / ((double)k * (double)k);
        s += t;
        if (t < 1e-17 * s) break;
    }
    return s;
}
__global__ void __launch_bounds__(kThreads)
resample_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int64_t n_in, int64_t n_out, double ratio,
                double fc, double hw, double beta, double inv_i0_beta) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < rows * n_out; i += stride) {
        const int64_t r = i / n_out, m = i % n_out;
        const double t = (double)m / ratio;
        int64_t lo = (int64_t)ceil(t - hw), hi = (int64_t)floor(t + hw);
        if (lo < 0) lo = 0;
        if (hi > n_in - 1) hi = n_in - 1;
        const float* xr = x + r * n_in;
        double acc = 0.0;
        for (int64_t n = lo; n <= hi; ++n) {
            const double d = t - (double)n, u = d / hw;
            const double a = 3.14159265358979323846 * fc * d;
            const double sinc = fabs(a) < 1e-12 ? 1.0 : sin(a) / a;
            const double w = bessel_i0(beta * sqrt(fmax(0.0, 1.0 - u * u))) * inv_i0_beta;
            acc += (double)xr[n] * fc * sinc * w;
        }
        y[i] = (float)acc;
    }
}

// zero the lowest `nbins` frequency bins of a spectrogram (MDXSeparator.run_model: spek[:, :, :3, :] *= 0)
template <typename T>
__global__ void __launch_bounds__(kThreads)
zero_low_bins_kernel(T* __restrict__ spec, int layout, int64_t B, int dim_f, int Tn, int nbins) {
    const int64_t n = B * 4 * (int64_t)nbins * Tn;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        int64_t r = i;
        const int c = (int)(r % 4); r /= 4;
        const int f = (int)(r % nbins); r /= nbins;
        const int t = (int)(r % Tn);
        const int64_t b = r / Tn;
        const int64_t o = layout == ALSEP_LAYOUT_NHWC ? ((b * Tn + t) * (int64_t)dim_f + f) * 4 + c
                                                      : ((b * 4 + c) * (int64_t)dim_f + f) * Tn + t;
        spec[o] = (T)0.f;
    }
}

unsigned grid_for(int64_t n, int per_thread = 1) {
    int64_t b = ceil_div64(n, (int64_t)kThreads * per_thread);
    if (b < 1) b = 1;
    return (unsigned)(b > kMaxBlocks ? kMaxBlocks : b);
}
}  // namespace

extern "C" int alsep_abi_version(void) { return ALSEP_ABI_VERSION; }

extern "C" int alsep_experiments_enabled(void) {
#ifdef ALSEP_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int alsep_create(int device_id, void* hip_stream, alsep_ctx** out) {
    if (!out || device_id < 0) return ALSEP_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return ALSEP_ERR_HIP;
    alsep_ctx* c = new alsep_ctx();
    c->device = device_id;
    c->stream = (hipStream_t)hip_stream;
    *out = c;
    return ALSEP_OK;
}
extern "C" int alsep_destroy(alsep_ctx* ctx) {
    if (ctx)
        for (hipEvent_t ev : ctx->prof_events) (void)hipEventDestroy(ev);
    if (ctx && ctx->nn_range) (void)hipFree(ctx->nn_range);
    if (ctx && ctx->zero_page) (void)hipFree(ctx->zero_page);
    delete ctx;
    return ALSEP_OK;
}

extern "C" int alsep_profile_begin(alsep_ctx* ctx, int category) {
    if (!ctx || category < 0) return ALSEP_ERR_ARG;
    ctx->prof_category = category;
    ctx->prof_used = 0;
    ctx->prof_flops = ctx->prof_bytes = 0.0;
    return ALSEP_OK;
}

extern "C" int alsep_profile_work(alsep_ctx* ctx, double* flops, double* bytes) {
    if (!ctx || !flops || !bytes) return ALSEP_ERR_ARG;
    *flops = ctx->prof_flops;
    *bytes = ctx->prof_bytes;
    return ALSEP_OK;
}

extern "C" int alsep_profile_end(alsep_ctx* ctx, double* total_ms, int64_t* launches) {
    if (!ctx || !total_ms || !launches) return ALSEP_ERR_ARG;
    double sum = 0.0;
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        ALSEP_HIP(ctx, hipEventSynchronize(ctx->prof_events[i + 1]));
        float ms = 0.f;
        ALSEP_HIP(ctx, hipEventElapsedTime(&ms, ctx->prof_events[i], ctx->prof_events[i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int64_t)(ctx->prof_used / 2);
    ctx->prof_category = ALSEP_PROF_NONE;
    ctx->prof_used = 0;
    return ALSEP_OK;
}
extern "C" int64_t alsep_launch_count(const alsep_ctx* ctx, const char* kernel) {
    if (!ctx || !kernel) return -1;
    auto it = ctx->launches.find(kernel);
    return it == ctx->launches.end() ? 0 : it->second;
}
extern "C" int alsep_launch_counts_reset(alsep_ctx* ctx) {
    if (!ctx) return ALSEP_ERR_ARG;
    ctx->launches.clear();
    return ALSEP_OK;
}
extern "C" const char* alsep_last_error(const alsep_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int alsep_axpby(alsep_ctx* ctx, float a, const float* x, float b, float* y, int64_t n) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y || n < 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_axpby: bad argument");
    if (n == 0) return ALSEP_OK;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_axpby: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 4)), dim3(kThreads), 0, ctx->stream, a, x, b, y, n);
    ALSEP_LAUNCH_CHECK(ctx, "axpby_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_peak_abs(alsep_ctx* ctx, const float* x, int64_t n, float* out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !out || n < 0 || (n > 0 && !x)) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_peak_abs: bad argument");
    ALSEP_HIP(ctx, hipMemsetAsync(out, 0, sizeof(float), ctx->stream));
    if (n == 0) return ALSEP_OK;
    hipLaunchKernelGGL(peak_abs_kernel, dim3(grid_for(n, 8)), dim3(kThreads), 4 * sizeof(float), ctx->stream, x, n, (unsigned*)out);
    ALSEP_LAUNCH_CHECK(ctx, "peak_abs_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_scale_by_device(alsep_ctx* ctx, float* y, int64_t n, float num, const float* den, float floor_) {
    ALSEP_ENTER(ctx);
    if (!ctx || !y || !den || n < 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_scale_by_device: bad argument");
    if (n == 0) return ALSEP_OK;
    hipLaunchKernelGGL(scale_by_device_kernel, dim3(grid_for(n, 4)), dim3(kThreads), 0, ctx->stream, y, n, num, den, floor_);
    ALSEP_LAUNCH_CHECK(ctx, "scale_by_device_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_dot3(alsep_ctx* ctx, const float* a, const float* b, int64_t n, double* dots) {
    ALSEP_ENTER(ctx);
    if (!ctx || !a || !b || !dots || n < 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_dot3: bad argument");
    // partial sums live after the 3 results: caller provides 3 + 3*1024 doubles
    const unsigned nb = std::min<unsigned>(grid_for(n, 8), 1024u);
    double* part = dots + 4;
    hipLaunchKernelGGL(dot3_partial_kernel, dim3(nb), dim3(kThreads), 12 * sizeof(double), ctx->stream, a, b, n, part);
    hipLaunchKernelGGL(dot3_final_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)part, (int)nb, dots);
    ALSEP_LAUNCH_CHECK(ctx, "dot3 kernels");
    return ALSEP_OK;
}

extern "C" int alsep_xcorr_window(alsep_ctx* ctx, const float* ref, const float* sig, int64_t probe, int max_shift, double* corr) {
    ALSEP_ENTER(ctx);
    if (!ctx || !ref || !sig || !corr || probe <= 0 || max_shift < 0 || max_shift >= probe)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_xcorr_window: bad argument");
    hipLaunchKernelGGL(xcorr_window_kernel, dim3(2 * max_shift + 1), dim3(kThreads), 4 * sizeof(double), ctx->stream,
                       ref, sig, probe, max_shift, corr);
    ALSEP_LAUNCH_CHECK(ctx, "xcorr_window_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_shift_subtract(alsep_ctx* ctx, const float* ref, const float* sig, int64_t len, int lag, float alpha, float* out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !ref || !sig || !out || len < 0) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_shift_subtract: bad argument");
    if (len == 0) return ALSEP_OK;
    hipLaunchKernelGGL(shift_subtract_kernel, dim3(grid_for(len, 4)), dim3(kThreads), 0, ctx->stream, ref, sig, len, lag, alpha, out);
    ALSEP_LAUNCH_CHECK(ctx, "shift_subtract_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_ola_combine(alsep_ctx* ctx, const float* chunks, int64_t n_chunks, int64_t chunk, int64_t step,
                                 int64_t total, int use_window, float gain, float* out, int64_t out_stride, int64_t p_lo,
                                 int64_t n_out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !chunks || !out || n_chunks <= 0 || chunk <= 0 || step <= 0 || step > chunk || total <= 0 || p_lo < 0 ||
        n_out < 0 || p_lo + n_out > total || (n_chunks - 1) * step >= total)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_ola_combine: bad argument");
    if (n_out == 0) return ALSEP_OK;
    hipLaunchKernelGGL(ola_combine_kernel, dim3(grid_for(n_out, 2)), dim3(kThreads), 0, ctx->stream, chunks, n_chunks, chunk,
                       step, total, use_window, gain, out, out_stride, p_lo, n_out);
    ALSEP_LAUNCH_CHECK(ctx, "ola_combine_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_ola_partial(alsep_ctx* ctx, const float* chunks, int64_t b0, int64_t b1, int64_t chunk, int64_t step,
                                 int64_t total, int use_window, float* part, int64_t p_lo, int64_t n_out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !part || b0 < 0 || b1 < b0 || (b1 > b0 && !chunks) || chunk <= 0 || step <= 0 || step > chunk || total <= 0 ||
        p_lo < 0 || n_out < 0 || p_lo + n_out > total)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_ola_partial: bad argument");
    if (n_out == 0) return ALSEP_OK;
    hipLaunchKernelGGL(ola_partial_kernel, dim3(grid_for(n_out, 2)), dim3(kThreads), 0, ctx->stream, chunks, b0, b1, chunk, step,
                       total, use_window, part, p_lo, n_out);
    ALSEP_LAUNCH_CHECK(ctx, "ola_partial_kernel");
    return ALSEP_OK;
}
extern "C" int alsep_ola_finish(alsep_ctx* ctx, const float* part, float gain, float* out, int64_t out_stride, int64_t n_out) {
    ALSEP_ENTER(ctx);
    if (!ctx || !part || !out || n_out < 0 || out_stride < n_out) return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_ola_finish: bad argument");
    if (n_out == 0) return ALSEP_OK;
    hipLaunchKernelGGL(ola_finish_kernel, dim3(grid_for(n_out, 2)), dim3(kThreads), 0, ctx->stream, part, gain, out, out_stride, n_out);
    ALSEP_LAUNCH_CHECK(ctx, "ola_finish_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_resample(alsep_ctx* ctx, const float* x, float* y, int64_t rows, int64_t n_in, int64_t n_out, int sr_in,
                              int sr_out, int zeros, float rolloff, float beta) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y || rows <= 0 || n_in <= 0 || n_out <= 0 || sr_in <= 0 || sr_out <= 0 || zeros < 1 || zeros > 512 ||
        !(rolloff > 0.f && rolloff <= 1.f) || beta < 0.f)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_resample: bad argument");
    const double ratio = (double)sr_out / (double)sr_in;
    const double fc = (ratio < 1.0 ? ratio : 1.0) * (double)rolloff, hw = (double)zeros / fc;
    double i0 = 1.0, t = 1.0;                                 // I0(beta) on the host
    for (int k = 1; k < 60; ++k) { t *= ((double)beta * beta * 0.25) / ((double)k * k); i0 += t; }
    hipLaunchKernelGGL(resample_kernel, dim3(grid_for(rows * n_out, 1)), dim3(kThreads), 0, ctx->stream, x, y, rows, n_in, n_out, ratio,
                       fc, hw, (double)beta, 1.0 / i0);
    ALSEP_LAUNCH_CHECK(ctx, "resample_kernel");
    return ALSEP_OK;
}

// scipy.signal.resample_poly with its default window, i.e. what librosa.resample(res_type="polyphase") runs (the VR band chain going
// down, vr.py:74-79 with the parameter sets' "res_type": "polyphase"): upfirdn(h, x, up, down) with h padded in front by n_pre_pad
// zeros, entries [n_pre_remove, n_pre_remove + n_out) kept.  One thread per output sample:
//   y[j] = sum_n x[n] h[(j + n_pre_remove) down - n up - n_pre_pad]      over the n whose tap index lies inside h
// float32 accumulation as scipy's upfirdn on float32 data.
__global__ void __launch_bounds__(kThreads)
resample_poly_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int64_t n_in, int64_t n_out, int up, int down,
                     const float* __restrict__ h, int n_taps, int n_pre_pad, int n_pre_remove) {
    // grid-stride: grid_for() caps the grid at kMaxBlocks workgroups
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < rows * n_out; i += (int64_t)gridDim.x * kThreads) {
        const int64_t r = i / n_out, j = i - r * n_out;
        const int64_t top = (j + n_pre_remove) * (int64_t)down - n_pre_pad;     // tap index of x[0]
        // tap = top - n up in [0, n_taps)  <=>  n in [ceil((top - n_taps + 1) / up), floor(top / up)]
        int64_t n_hi = top >= 0 ? top / up : -1;
        const int64_t lo_num = top - n_taps + 1;
        const int64_t n_lo = lo_num > 0 ? (lo_num + up - 1) / up : 0;
        if (n_hi > n_in - 1) n_hi = n_in - 1;
        const float* xr = x + r * n_in;
        float acc = 0.f;
        for (int64_t n = n_lo; n <= n_hi; ++n) acc = fmaf(xr[n], h[top - n * up], acc);
        y[i] = acc;
    }
}

extern "C" int alsep_resample_poly(alsep_ctx* ctx, const float* x, float* y, int64_t rows, int64_t n_in, int64_t n_out, int up, int down,
                                   const float* taps, int n_taps, int n_pre_pad, int n_pre_remove) {
    ALSEP_ENTER(ctx);
    if (!ctx || !x || !y || !taps || rows <= 0 || n_in <= 0 || n_out <= 0 || up < 1 || down < 1 || n_taps < 1 || n_pre_pad < 0 ||
        n_pre_remove < 0)
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_resample_poly: bad argument");
    hipLaunchKernelGGL(resample_poly_kernel, dim3(grid_for(rows * n_out, 1)), dim3(kThreads), 0, ctx->stream, x, y, rows, n_in, n_out, up,
                       down, taps, n_taps, n_pre_pad, n_pre_remove);
    ALSEP_LAUNCH_CHECK(ctx, "resample_poly_kernel");
    return ALSEP_OK;
}

extern "C" int alsep_zero_low_bins(alsep_ctx* ctx, void* spec, int dtype, int layout, int64_t B, int64_t dim_f, int64_t T,
                                   int nbins) {
    ALSEP_ENTER(ctx);
    if (!ctx || !spec || B < 0 || nbins < 0 || nbins > dim_f || (dtype != ALSEP_F32 && dtype != ALSEP_BF16 && dtype != ALSEP_F16))
        return alsep_fail(ctx, ALSEP_ERR_ARG, "alsep_zero_low_bins: bad argument");
    if (B == 0 || nbins == 0) return ALSEP_OK;
    const int64_t n = B * 4 * nbins * T;
    if (dtype == ALSEP_F32)
        hipLaunchKernelGGL((zero_low_bins_kernel<float>), dim3(grid_for(n)), dim3(kThreads), 0, ctx->stream, (float*)spec, layout, B, (int)dim_f, (int)T, nbins);
    else
        hipLaunchKernelGGL((zero_low_bins_kernel<bf16_t>), dim3(grid_for(n)), dim3(kThreads), 0, ctx->stream, (bf16_t*)spec, layout, B, (int)dim_f, (int)T, nbins);
    ALSEP_LAUNCH_CHECK(ctx, "zero_low_bins_kernel");
    return ALSEP_OK;
}
