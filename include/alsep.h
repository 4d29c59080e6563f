/* alsep.h -- C ABI of libalsep.so, the MI355X (gfx950) kernels behind AudioLab's
 * Process->Separate path.
 *
 * The reference (d8ahazard/AudioLab) is pure Python; it has no FFI for this path.  Each
 * entry point below replaces the arithmetic the reference reaches through the calls cited
 * next to it (paths relative to the reference root).  Binding shown in INTEGRATION.md.
 *
 * Conventions
 *   - Every data pointer is a DEVICE pointer owned by the caller (PyTorch allocator,
 *     tensor.data_ptr()).  The library never frees caller memory.  Objects it creates
 *     (ctx, plan, net) own their private device tables and free them in *_destroy.
 *   - All work is enqueued on the hipStream_t given to alsep_create; calls return after
 *     enqueueing (the caller synchronises).  One ctx per (GPU, stream); not thread-safe.
 *   - Return value: 0 = ok, negative = error; alsep_last_error(ctx) describes the most
 *     recent failure.  No C++ exception crosses the ABI.
 *   - dtype: ALSEP_F32 = float32 storage + f32 MFMA (parity mode),
 *            ALSEP_BF16 = bfloat16 storage + bf16 MFMA with f32 accumulation.
 *   - layout of a spectrogram: ALSEP_LAYOUT_REF  = [B,4,dim_f,T]  (the reference's
 *     model I/O, mdxnet.py:56,170; patch_separate.py:52), channels (L_re,L_im,R_re,R_im);
 *     ALSEP_LAYOUT_NHWC = [B,T,dim_f,4] (the network's internal channels-last order).
 */
#ifndef ALSEP_H
#define ALSEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALSEP_ABI_VERSION 1

enum { ALSEP_F32 = 0, ALSEP_BF16 = 1 };
enum { ALSEP_LAYOUT_REF = 0, ALSEP_LAYOUT_NHWC = 1 };
enum {
    ALSEP_OK = 0,
    ALSEP_ERR_ARG = -1,        /* bad argument / unsupported geometry */
    ALSEP_ERR_HIP = -2,        /* a HIP runtime call failed */
    ALSEP_ERR_NOMEM = -3,
    ALSEP_ERR_STATE = -4
};

typedef struct alsep_ctx alsep_ctx;
typedef struct alsep_plan alsep_plan;     /* STFT geometry + twiddle/envelope tables */
typedef struct alsep_net alsep_net;       /* packed TFC-TDF U-Net weights */

int alsep_abi_version(void);

/* ctx: replaces device selection at modules/separator/stem_separator.py:99-100. */
int alsep_create(int device_id, void* hip_stream, alsep_ctx** out);
int alsep_destroy(alsep_ctx* ctx);
const char* alsep_last_error(const alsep_ctx* ctx);

/* Per-kernel-class timing with HIP events recorded on the ctx stream around every launch of
 * the selected class (used by bench.py for the live roofline figure).  begin selects a class,
 * end synchronises, returns the summed kernel time and launch count, and switches timing off. */
enum {
    ALSEP_PROF_NONE = 0,
    ALSEP_PROF_CONV3X3 = 1,      /* conv3x3_bf16_kernel<64> / conv3x3_kernel<...,64>: the plain main-tile kernel */
    ALSEP_PROF_CONV3X3_SMALL = 2,/* conv3x3_kernel, TW<64 tiles (deep levels) */
    ALSEP_PROF_TDF = 3,          /* tdf_gemm_kernel */
    ALSEP_PROF_PIX = 4,          /* pix_gemm_kernel (ds / us) */
    ALSEP_PROF_POINTWISE = 5,    /* first / final 1x1 conv */
    ALSEP_PROF_STFT = 6,
    ALSEP_PROF_ISTFT = 7,
    ALSEP_PROF_CONV3X3_REGW = 8, /* conv3x3_bf16_regw_kernel (persistent, level 0) */
    ALSEP_PROF_CONV3X3_PIPE = 9, /* conv3x3_bf16_pipe_kernel (opt-in) */
    ALSEP_PROF_CONV3X3_BIG = 10, /* conv3x3_bf16_big_kernel<2> (8-wave 8x64 tile, 96 channels: level 1) */
    ALSEP_PROF_CONV3X3_BIG3 = 11 /* conv3x3_bf16_big_kernel<3> (same kernel, 144 channels: level 2) */
};
int alsep_profile_begin(alsep_ctx* ctx, int category);
int alsep_profile_end(alsep_ctx* ctx, double* total_ms, int64_t* launches);
/* Launches of one kernel since alsep_create / alsep_launch_counts_reset, by the name the launch site reports
 * ("conv3x3_bf16_big_kernel<2>", "conv3x3_bf16_regw_kernel", "tdf_bf16_wide_kernel<res>", "us_stream_kernel",
 * "istft_r16_kernel" ...): lets a parity test prove WHICH kernel produced the result it checked.  -1 on a null argument. */
int64_t alsep_launch_count(const alsep_ctx* ctx, const char* kernel);
int alsep_launch_counts_reset(alsep_ctx* ctx);

/* STFT plan: geometry of ConvTDFNetTrim.__init__ (modules/rvc/infer/modules/uvr5/mdxnet.py:15-39).
 * dim_t is the frame count (2**dim_t_arg).  chunk = hop*(dim_t-1). */
int alsep_plan_create(alsep_ctx* ctx, int n_fft, int hop, int dim_f, int dim_t, alsep_plan** out);
int alsep_plan_destroy(alsep_plan* plan);
int alsep_plan_supported_nfft(int n_fft);   /* 1 if an FFT kernel is instantiated for n_fft */

/* STFT: ConvTDFNetTrim.stft (mdxnet.py:41-56): reflect-padded (center=True) periodic-Hann
 * frames, one-sided unnormalised DFT, bins [0,dim_f).
 * Chunk b reads pcm[ch*ch_stride + b*chunk_stride + s], s in [0,chunk): with
 * chunk_stride=gen and ch_stride=len this frames the zero-padded mix of demix_base
 * (mdxnet.py:152-163) in place; with chunk_stride=2*chunk, ch_stride=chunk it is [B,2,chunk]. */
int alsep_stft(alsep_ctx* ctx, const alsep_plan* plan, const float* pcm, int64_t ch_stride,
               int64_t chunk_stride, int64_t n_chunks, void* spec, int dtype, int layout);

/* iSTFT: ConvTDFNetTrim.istft (mdxnet.py:58-75): zero bins >= dim_f, inverse DFT, window,
 * overlap-add, divide by sum w^2, drop n_fft/2 each side.  Sample s of chunk b, with
 * keep_lo <= s < keep_hi, is written to out[ch*out_ch_stride + b*out_chunk_stride + (s-keep_lo)]
 * if that offset < out_limit.  keep=[0,chunk), strides (chunk, 2*chunk) give [B,2,chunk];
 * keep=[trim,chunk-trim), out_chunk_stride=gen stitches demix_base's output
 * (mdxnet.py:178-183) directly. */
int alsep_istft(alsep_ctx* ctx, const alsep_plan* plan, const void* spec, int dtype, int layout,
                int64_t n_chunks, float* out, int64_t out_ch_stride, int64_t out_chunk_stride,
                int64_t keep_lo, int64_t keep_hi, int64_t out_limit);

/* Hann-window overlap-add chunker of the third-party MDXSeparator.demix (what Separator.separate runs
 * today, stem_separator.py:281; package audio-separator>=0.32.0, setup.sh:96 -- PARITY UNPINNED):
 * chunks [n_chunks,2,chunk] are the model outputs of windows starting every `step` samples of the padded
 * mixture of length `total`; out[c*out_stride + i] = gain * sum_b w_b y_b / sum_b w_b at mixture position
 * p_lo + i, w_b = np.hanning(min(chunk, total - b*step)) (or 1 when use_window == 0). */
int alsep_ola_combine(alsep_ctx* ctx, const float* chunks, int64_t n_chunks, int64_t chunk, int64_t step,
                      int64_t total, int use_window, float gain, float* out, int64_t out_stride, int64_t p_lo,
                      int64_t n_out);
/* zero the lowest nbins bins of a spectrogram in place (MDXSeparator.run_model zeroes bins 0..2). */
int alsep_zero_low_bins(alsep_ctx* ctx, void* spec, int dtype, int layout, int64_t B, int64_t dim_f, int64_t T,
                        int nbins);

/* Spectrogram layout conversion REF <-> NHWC (same dtype). */
int alsep_spec_convert(alsep_ctx* ctx, const void* src, void* dst, int dtype, int src_layout,
                       int64_t B, int64_t dim_f, int64_t T);

/* TFC-TDF U-Net ("ConvTDFNet", the network inside the *.onnx models loaded at
 * stem_separator.py:394,512 and run through MDXSeparator.model_run,
 * handlers/patch_separate.py:52,58-62).
 * Weights arrive as a table of named fp32 device tensors in torch state_dict layout with
 * BatchNorm already folded to per-channel (scale, shift) by the caller; the library
 * repacks them into MFMA fragment order in its own memory. */
typedef struct alsep_net_config {
    int32_t dim_f;        /* frequency bins in/out */
    int32_t dim_t;        /* frames */
    int32_t num_blocks;   /* L (11): n = L/2 encoder and decoder levels + bottleneck */
    int32_t l;            /* convs per TFC block */
    int32_t g;            /* channel growth */
    int32_t bn;           /* TDF bottleneck factor (f -> f/bn -> f); 0 = single f->f linear */
    int32_t dtype;        /* ALSEP_F32 | ALSEP_BF16 */
    int32_t reserved;
} alsep_net_config;

typedef struct alsep_tensor {
    const char* name;     /* e.g. "encoding_blocks.0.tfc.H.1.0.weight", "...scale", "...shift" */
    const float* data;    /* device pointer, fp32, contiguous */
    int64_t numel;
} alsep_tensor;

int alsep_net_create(alsep_ctx* ctx, const alsep_net_config* cfg, const alsep_tensor* tensors,
                     int64_t n_tensors, alsep_net** out);
int alsep_net_destroy(alsep_net* net);
/* bytes of caller-provided scratch for a batch of B chunks */
int64_t alsep_net_workspace_bytes(const alsep_net* net, int64_t B);
/* spec_in/spec_out: [B,T,dim_f,4] (NHWC) in the net's dtype.
 *   spec_out = out_alpha * net(in_scale * spec_in) + out_beta * spec_out
 * (in_scale=1, alpha=1, beta=0: plain forward; the denoise average of mdxnet.py:168-173,
 * 0.5*f(x) - 0.5*f(-x), is two calls: (1, 0.5, 0) then (-1, -0.5, 1)). */
int alsep_net_forward(alsep_ctx* ctx, const alsep_net* net, const void* spec_in, void* spec_out,
                      int64_t B, void* workspace, int64_t workspace_bytes, float in_scale,
                      float out_alpha, float out_beta);

/* Element-wise / reduction ops of the ensemble stage (stem_separator.py:241-262,173-239,
 * 415-456) and of the MDX runner (mdxnet.py:168-173 denoise average, :211 secondary stem). */
/* y = a*x + b*y over n floats */
int alsep_axpby(alsep_ctx* ctx, float a, const float* x, float b, float* y, int64_t n);
/* out[0] = max |x| (out is a device float; written, not accumulated) */
int alsep_peak_abs(alsep_ctx* ctx, const float* x, int64_t n, float* out);
/* y *= s where s is read from a device float: y = y * (num / max(*den, floor)) */
int alsep_scale_by_device(alsep_ctx* ctx, float* y, int64_t n, float num, const float* den, float floor_);
/* dots[0..2] = <a,b>, <a,a>, <b,b> in double precision.  dots must hold 4 + 3*1024 device
 * doubles: [0..2] results, [4..] per-block partials (deterministic two-stage sum). */
int alsep_dot3(alsep_ctx* ctx, const float* a, const float* b, int64_t n, double* dots);
/* corr[j] = sum_n ref[n+lag]*sig[n], lag = j - max_shift, over the first `probe` samples
 * (np.correlate(ref[:probe], sig[:probe], "full") centre window, stem_separator.py:216-224) */
int alsep_xcorr_window(alsep_ctx* ctx, const float* ref, const float* sig, int64_t probe,
                       int max_shift, double* corr);
/* out[n] = ref[n] - alpha * sig[n-lag] (zero outside), n in [0,len) (stem_separator.py:196-232) */
int alsep_shift_subtract(alsep_ctx* ctx, const float* ref, const float* sig, int64_t len, int lag,
                         float alpha, float* out);

/* ---- VR-architecture building blocks (CascadedASPPNet: reference modules/rvc/infer/lib/uvr5_pack/lib_v5/nets*.py,
 * layers*.py).  Channels-last float32 tensors [B, H = bins, W = frames, C].  Outputs that feed a
 * torch.cat are written into the channel slice [y_coff, y_coff + C) of a tensor with y_ctotal channels. ---- */
/* y = act(conv2d(x, w) * scale + shift): nn.Conv2d(bias=False) + folded BatchNorm2d + activation (layers*.py:9-27);
 * w packed [KH][KW][Cin][Cout]; act 0 none, 1 ReLU, 2 LeakyReLU(0.01); padding / dilation per axis (layers_new.py:83-91 uses
 * (4,2), (8,4), (12,6)).  Output size as torch: (H + 2 pad_h - dil_h (KH-1) - 1) / stride + 1.  A Linear (+ bias, BatchNorm1d)
 * is the 1x1 case with the rows as pixels. */
int alsep_vr_conv2d(alsep_ctx* ctx, const float* x, const float* w, const float* scale, const float* shift, float* y,
                    int64_t B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int dil_h,
                    int dil_w, int act, int y_ctotal, int y_coff);
/* recurrent part of one direction of nn.LSTM (layers_new.py:117-121): pre [T,N,4*hidden] = x W_ih^T + b_ih + b_hh (gates
 * i,f,g,o), whh [4*hidden][hidden]; h_t is written to out[t, n, out_off .. out_off+hidden) of rows out_stride wide;
 * reverse != 0 walks t from T-1 down (the backward direction).  hidden in {16, 32, 64}. */
int alsep_vr_lstm(alsep_ctx* ctx, const float* pre, const float* whh, float* out, int T, int N, int hidden, int out_stride,
                  int out_off, int reverse);
/* depthwise (groups = C) KHxKW dilated convolution, same-size output (layers*.py:30-52); w packed [C][KH][KW] */
int alsep_vr_depthwise(alsep_ctx* ctx, const float* x, const float* w, float* y, int64_t B, int H, int W, int C, int KH, int KW,
                       int pad, int dil);
/* F.interpolate(mode="bilinear", align_corners=True) to (Ho, Wo) (layers*.py:84, 111) */
int alsep_vr_resize_bilinear(alsep_ctx* ctx, const float* x, float* y, int64_t B, int H, int W, int C, int Ho, int Wo,
                             int y_ctotal, int y_coff);
/* spec_utils.crop_center (spec_utils.py:12-27) + concat: y[bh, wq, y_coff + c] = x[bh, w_off + wq, c]; BH = B * H rows */
int alsep_vr_copy_slice(alsep_ctx* ctx, const float* x, float* y, int64_t BH, int Wx, int C, int w_off, int Wy, int y_ctotal,
                        int y_coff);
/* nn.AdaptiveAvgPool2d((1, None)) (layers*.py:93): mean over bins -> [B, 1, W, C] */
int alsep_vr_mean_h(alsep_ctx* ctx, const float* x, float* y, int64_t B, int H, int W, int C);
/* nets*.py:80-111: mask = sigmoid(logit) replicate-padded from Hm to Hout bins, mask ** (1 + a/3) below split_bin and
 * ** (1 + a) from it on when aggressiveness a >= 0, out = mask * mix */
int alsep_vr_mask(alsep_ctx* ctx, const float* logit, const float* mix, float* out, int64_t B, int Hm, int Hout, int W, int C,
                  int split_bin, float aggressiveness);

#ifdef __cplusplus
}
#endif
#endif /* ALSEP_H */
