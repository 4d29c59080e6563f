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
 *            ALSEP_BF16 = bfloat16 storage + bf16 MFMA with f32 accumulation,
 *            ALSEP_F16  = IEEE half storage + f16 MFMA with f32 accumulation (the type of the reference's
 *                         use_autocast=True, stem_separator.py:106; 8x finer rounding than bf16, range 65504).
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

enum { ALSEP_F32 = 0, ALSEP_BF16 = 1, ALSEP_F16 = 2 };
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
/* 1 when the library was compiled with -DALSEP_EXPERIMENTS (timing-experiment switches and superseded kernel variants present);
 * the product build returns 0: no environment variable can then change what is computed. */
int alsep_experiments_enabled(void);

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
    ALSEP_PROF_CONV3X3_BIG3 = 11,/* conv3x3_bf16_big_kernel<3> (same kernel, 144 channels: level 2) */
    ALSEP_PROF_NN_GEMM = 12,     /* nn_gemm_tn_kernel / nn_bgemm_kernel (float32 MFMA products of the transformer / Demucs / MDX23C families) */
    ALSEP_PROF_NN_CONV = 13,     /* nn_conv2d_tiled_kernel / vr_conv2d_kernel through alsep_nn_conv2d */
    ALSEP_PROF_NN_GEMM_H = 14,   /* nn_gemm_h_kernel (f16 MFMA Linear) */
    ALSEP_PROF_NN_ATTN_H = 15,   /* nn_attn_h_kernel (one-pass f16 attention) */
    ALSEP_PROF_NN_CONV_H = 16    /* nn_conv_hh_kernel (f16 MFMA convolution) + its split-K reduction */
};
int alsep_profile_begin(alsep_ctx* ctx, int category);
int alsep_profile_end(alsep_ctx* ctx, double* total_ms, int64_t* launches);
/* arithmetic / minimal HBM traffic of the launches bracketed since alsep_profile_begin, as their launch sites count them (the NN_*
 * categories: 2 M N K per product, operands + result once); call before alsep_profile_end */
int alsep_profile_work(alsep_ctx* ctx, double* flops, double* bytes);
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
/* The same overlap-add split over the ranks of a job: `chunks` holds only the chunks [b0, b1) of this rank; part [3][n_out] =
 * raw weighted sums of the two channels and the summed weights over those chunks.  The ranks' parts are SUMMED by one collective (at a
 * shard seam chunks of two ranks overlap), then alsep_ola_finish divides: out[c][i] = gain * part[c][i] / part[2][i]. */
int alsep_ola_partial(alsep_ctx* ctx, const float* chunks, int64_t b0, int64_t b1, int64_t chunk, int64_t step, int64_t total,
                      int use_window, float* part, int64_t p_lo, int64_t n_out);
int alsep_ola_finish(alsep_ctx* ctx, const float* part, float gain, float* out, int64_t out_stride, int64_t n_out);
/* Sample-rate conversion of rows of PCM, x [rows, n_in] at sr_in -> y [rows, n_out] at sr_out (n_out = ceil(n_in * sr_out / sr_in) as
 * librosa): what librosa.load(path, sr=44100) does to a 48 kHz input at modules/separator/stem_separator.py:865.  The reference uses
 * libsoxr there (not in /root/reference): this is the build's own Kaiser-windowed sinc (`zeros` zero crossings per side, cut-off
 * rolloff * min(sr_in, sr_out) / 2, Kaiser beta) -- PARITY UNPINNED; restated in oracle/mdx_oracle.py resample(). */
int alsep_resample(alsep_ctx* ctx, const float* x, float* y, int64_t rows, int64_t n_in, int64_t n_out, int sr_in, int sr_out,
                   int zeros, float rolloff, float beta);
/* scipy.signal.resample_poly(x, up, down) with a caller-made filter (its default: firwin(20 max(up, down) + 1, 1 / max(up, down),
 * window=("kaiser", 5.0)) * up in float32) -- what librosa.resample(res_type="polyphase") runs in the VR band chain going down
 * (modules/rvc/infer/modules/uvr5/vr.py:74-79, modelparams/4band_v*.json "res_type").  up / down already divided by their gcd;
 * n_pre_pad = down - half_len % down, n_pre_remove = (half_len + n_pre_pad) / down, n_out = ceil(n_in up / down) as scipy computes
 * them (audiolab_amd/vr_frontend.py).  x [rows, n_in] -> y [rows, n_out], taps on the device. */
int alsep_resample_poly(alsep_ctx* ctx, const float* x, float* y, int64_t rows, int64_t n_in, int64_t n_out, int up, int down,
                        const float* taps, int n_taps, int n_pre_pad, int n_pre_remove);
/* scipy.signal.resample(x, n_out) (Fourier method) -- librosa.resample(res_type="scipy"), the VR band chain going up
 * (modules/rvc/infer/lib/uvr5_pack/lib_v5/spec_utils.py:427): spectrum truncated / zero-padded with scipy's Nyquist rule, inverse
 * transform times n_out / n_in.  Any lengths <= 2^27 (float64 Bluestein transforms); rows go two per complex transform. */
int64_t alsep_resample_fft_workspace_bytes(int64_t n_in, int64_t n_out);
int alsep_resample_fft(alsep_ctx* ctx, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t n_in, int64_t n_out,
                       void* ws, int64_t ws_bytes);
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
    int32_t dtype;        /* ALSEP_F32 | ALSEP_BF16 | ALSEP_F16 */
    int32_t flags;        /* 0, or ALSEP_NET_SPLIT_F16 (float32 networks only) */
} alsep_net_config;
/* float32 storage with the contractions (3x3 convolutions, ds / us, TDF linears) on the 16-bit matrix pipe: every float32 operand is
 * carried as two IEEE halves (hi, scaled lo) and a product costs three f16 MFMAs accumulated in float32 -- 2^-22 relative per product,
 * the float32 mode's accuracy at 5 x its matrix throughput; activations must stay inside the half range (|x| <= 65504), beyond it the
 * outputs are Inf / NaN.  Without the flag a float32 network runs on v_mfma_f32_16x16x4_f32 (exact fmaf chains). */
#define ALSEP_NET_SPLIT_F16 1

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
/* alsep_stft (channels-last, the network's storage type) + alsep_net_forward in one call for the half-precision networks: the STFT and
 * the network's first 1x1 convolution run as ONE kernel (no spectrogram in HBM), bit-identical to the two calls.  pcm / ch_stride /
 * chunk_stride as alsep_stft; zero_low_bins: bins below it enter the network as zeros (alsep_zero_low_bins of the overlap-add runner).
 * Returns ALSEP_ERR_STATE when this (plan, network) pair has no fused kernel: call alsep_stft + alsep_net_forward then. */
int alsep_net_forward_pcm(alsep_ctx* ctx, const alsep_net* net, const alsep_plan* plan, const float* pcm, int64_t ch_stride,
                          int64_t chunk_stride, void* spec_out, int64_t B, void* workspace, int64_t workspace_bytes, float in_scale,
                          float out_alpha, float out_beta, int zero_low_bins);

/* The generic float32 GEMM / convolution entry points (alsep_nn_bgemm*, alsep_nn_conv2d, alsep_vr_conv2d on their tiled paths) of THIS
 * context: split = 1 runs their contractions as split-half products on the f16 matrix pipe (float32 in and out, 2^-22 per product,
 * operands limited to the half range), split = 0 (default) on exact f32 MFMA.  Call outside a stream capture. */
int alsep_nn_set_contraction(alsep_ctx* ctx, int split);
/* *out = 1 when a split-contraction launch of this context met an operand beyond the half range since the last call (results since then
 * are invalid: run them again with split = 0).  Reads and clears the word; synchronises ctx's stream. */
int alsep_nn_range_flag(alsep_ctx* ctx, int32_t* out);

/* ALSEP_NET_SPLIT_F16 networks: *out = 1 when any forward since the last call met an operand beyond the half range (|x| > 65504, or not
 * a number) -- the results of those forwards are invalid (rebuild the network without the flag for such input).  Reads and clears the
 * network's range word; synchronises ctx's stream.  *out = 0 for every other network. */
int alsep_net_range_flag(alsep_ctx* ctx, alsep_net* net, int32_t* out);

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

/* ---- Demucs family (HTDemucs: what audio_separator's DemucsSeparator runs for htdemucs_6s.yaml, loaded at
 * modules/separator/stem_separator.py:466 and run at :479; network source in the un-vendored demucs>=4.0.1, requirements.txt:19 --
 * PARITY UNPINNED).  Generic channels-last float32 building blocks; activations: 0 none, 1 ReLU, 2 LeakyReLU(0.01), 3 GELU (erf),
 * 4 GLU over the channel axis (C -> C/2; norm / act only). ---- */
/* alsep_vr_conv2d with a stride per axis and GELU: Conv1d over [B,L,C] is the W = 1 case, a Linear the 1x1 case */
int alsep_nn_conv2d(alsep_ctx* ctx, const float* x, const float* w, const float* scale, const float* shift, float* y, int64_t B, int H,
                    int W, int Cin, int Cout, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                    int act, int y_ctotal, int y_coff);
/* C[b1,b2][m][n] = alpha * sum_k A[b1,b2][m][k] B[b1,b2][n][k]; sa / sb / sc = element strides {b1, b2, row, k-or-column} (the
 * attention products of nn.MultiheadAttention straight from the packed in-projection, no head transposes) */
int alsep_nn_bgemm(alsep_ctx* ctx, const float* A, const float* B, float* C, int nb1, int nb2, int M, int N, int K, const int64_t* sa,
                   const int64_t* sb, const int64_t* sc, float alpha);
/* softmax over the last dimension of x [rows, n], in place */
int alsep_nn_softmax_rows(alsep_ctx* ctx, float* x, int64_t rows, int n);
/* the same on rows `ld` floats apart (a score matrix whose rows are padded to a multiple of 4 floats) */
int alsep_nn_softmax_rows_ld(alsep_ctx* ctx, float* x, int64_t rows, int n, int ld);
/* bytes of device scratch (8-byte aligned) for alsep_nn_norm / alsep_nn_meanstd over G groups of per_group elements */
int64_t alsep_nn_stats_workspace_bytes(int64_t G, int64_t per_group);
/* x [G, R, C]: normalise every group of R rows x C channels to zero mean / unit (biased) variance, y = act(n * gamma[c] + beta[c]):
 * nn.GroupNorm(1, C) (R = positions of one sample), nn.LayerNorm(C) (R = 1), demucs MyGroupNorm (R = tokens).  gamma / beta may
 * both be NULL.  act 0, 3 or 4. */
int alsep_nn_norm(alsep_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta, int64_t G, int64_t R, int C, float eps,
                  int act, void* workspace);
/* stats[2 s] = mean, stats[2 s + 1] = UNBIASED std of sample s (HTDemucs.forward: mean / std of the whole spectrogram / waveform) */
int alsep_nn_meanstd(alsep_ctx* ctx, const float* x, int64_t nsamples, int64_t per_sample, float* stats, void* workspace);
/* inverse 0: y = (x - mean) / (eps + std); inverse 1: y = x * std + mean */
int alsep_nn_affine_stats(alsep_ctx* ctx, const float* x, float* y, const float* stats, int64_t nsamples, int64_t per_sample, float eps,
                          int inverse);
/* y = act(x) over x [rows, C]; act 1, 3 or 4 */
int alsep_nn_act(alsep_ctx* ctx, const float* x, float* y, int64_t rows, int C, int act);
/* y = a + scale[c] * b (LayerScale residual; scale NULL: plain add) */
int alsep_nn_scale_add(alsep_ctx* ctx, const float* a, const float* b, const float* scale, float* y, int64_t rows, int C);
/* y[i] += s * e[((i / inner) % period) * C + i % C]: frequency embedding over [B,F,T,C] (inner = T*C, period = F), positional
 * embedding over [B,N,C] (inner = C, period = N) */
int alsep_nn_add_bcast(alsep_ctx* ctx, float* y, const float* e, float s, int64_t n, int64_t inner, int period, int C);
/* y[r * y_stride + i] += w[i] * x[r * x_stride + i], i < n (triangular overlap-add of demucs.apply.apply_model);  y[r][i] /= w[i] */
int alsep_nn_vec_fma(alsep_ctx* ctx, float* y, const float* x, const float* w, int64_t rows, int64_t n, int64_t y_stride, int64_t x_stride);
int alsep_nn_vec_div(alsep_ctx* ctx, float* y, const float* w, int64_t rows, int64_t n);
/* y [B, L, C] = x [B, C, L] (waveform [B,2,L] <-> channels-last [B,L,2]) */
int alsep_nn_swap_last2(alsep_ctx* ctx, const float* x, float* y, int64_t B, int64_t C, int64_t L);
/* F.pad(mode="reflect") along the last axis: [rows, n] -> [rows, left + n + right] */
int alsep_nn_reflect_pad(alsep_ctx* ctx, const float* x, float* y, int64_t rows, int64_t n, int64_t left, int64_t right);
/* second half of ConvTranspose([K,1], stride [S,1]), K = 2 S: g [B, I, J, K*Cout] = 1x1 conv of the input with the weights packed
 * [Cin][k*Cout + co]; y [B, Lout, J, Cout] = act(g[i][r] + g[i-1][r+S] + bias), o + pad = i S + r (the crop of HDecLayer.forward) */
int alsep_nn_tconv_fold(alsep_ctx* ctx, const float* g, const float* bias, float* y, int64_t B, int I, int J, int Cout, int S, int pad,
                        int Lout, int act);
/* spec [B,4,F,Tt] (alsep_stft, reference layout) -> y [B,F,T,4] = scale * spec[..., t_off : t_off + T]   (HTDemucs._spec/_magnitude) */
int alsep_demucs_spec_in(alsep_ctx* ctx, const float* spec, float* y, int64_t B, int F, int Tt, int T, int t_off, float scale);
/* x [B,F,T,S*4] (normalised output), stats of the input spectrogram -> spec [B*S,4,F,Tt]: frames [t_off, t_off+T) =
 * scale * (x * std + mean), the rest zero   (HTDemucs._mask with cac + the frame padding of _ispec) */
int alsep_demucs_spec_out(alsep_ctx* ctx, const float* x, const float* stats, float* spec, int64_t B, int S, int F, int Tt, int T,
                          int t_off, float scale);
/* out [B,S,2,L] = (xt [B,L,S*2] * stdt + meant) + xs [B*S,2,L]   (the last lines of HTDemucs.forward) */
int alsep_demucs_mix_out(alsep_ctx* ctx, const float* xt, const float* statst, const float* xs, float* out, int64_t B, int S, int64_t L);

/* ---- Roformer family (BS-RoFormer / Mel-Band RoFormer: the first members of the reference's default ensemble and its de-reverb /
 * de-echo models, stem_separator.py:379-382, 796-797; network source in the un-vendored audio-separator -- PARITY UNPINNED).  Further
 * building blocks on the same channels-last float32 tensors; activation 5 = tanh. ---- */
/* alsep_nn_bgemm with an epilogue: C = act(alpha * A B^T + bias[column]); bias may be NULL; act 0, 3 (GELU) or 5 (tanh) */
int alsep_nn_bgemm_bias(alsep_ctx* ctx, const float* A, const float* B, float* C, int nb1, int nb2, int M, int N, int K, const int64_t* sa,
                        const int64_t* sb, const int64_t* sc, float alpha, const float* bias, int act);
/* RMSNorm over the last axis: y = x / max(||x||, 1e-12) * sqrt(C) * gamma; rows at x + r * x_stride / y + r * y_stride (a column slice of a
 * wider matrix in place when both strides are its width) */
int alsep_nn_rmsnorm(alsep_ctx* ctx, const float* x, float* y, const float* gamma, int64_t rows, int C, int64_t x_stride, int64_t y_stride);
/* rotary position embedding (interleaved pairs, theta 10000) in place on the heads * d columns from col_off of every row of x
 * [rows, row_stride]; position of row r = (r / pos_div) % pos_mod */
int alsep_nn_rotary(alsep_ctx* ctx, float* x, int64_t rows, int64_t row_stride, int col_off, int heads, int d, int64_t pos_div, int64_t pos_mod);
/* out [rows, heads * d] *= sigmoid(gates [rows, heads]) per head */
int alsep_nn_gate(alsep_ctx* ctx, float* out, const float* gates, int64_t rows, int heads, int d);
/* band-split input: feat [T, 2 * n_idx], feat[t, 2 i + c] = spec[(s*2 + c), f, t] with midx[i] = 2 f + s (spec [4, F, T] from alsep_stft) */
int alsep_roformer_gather(alsep_ctx* ctx, const float* spec, const int* midx, float* feat, int n_idx, int F, int T);
/* mask estimator output h [T, H] -> complex masks (GLU, averaged over the bands that cover a bin: occurrences occ_start[m] .. occ_start[m+1]
 * with value columns col_a[o], col_a[o]+1 and gate columns col_g[o], col_g[o]+1), out = spec * mask, same [4, F, T] layout */
int alsep_roformer_mask(alsep_ctx* ctx, const float* spec, const float* h, const int* occ_start, const int* col_a, const int* col_g,
                        float* out, int F, int T, int H);

/* ---- MDX23C (TFC-TDF v3: ensemble slot 4 and the drum-kit splitter, stem_separator.py:383, 541; network source in the un-vendored
 * audio-separator -- PARITY UNPINNED).  Further building blocks. ---- */
/* nn.InstanceNorm2d(affine) on channels-last x [P, C] (one sample): per-channel statistics over the P pixels, y = act(n * gamma + beta);
 * act 0 or 3 (GELU); gamma / beta may both be NULL */
int64_t alsep_nn_instnorm_workspace_bytes(int64_t P, int C);
int alsep_nn_instnorm(alsep_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta, int64_t P, int C, float eps, int act,
                      void* workspace);
/* the same with y stored as IEEE half (C % 4 == 0): the activation a half-precision convolution reads (MDX23C's half-precision mode --
 * stem_separator.py:106 use_autocast=True: convolutions in half, normalisation in float32) */
int alsep_nn_instnorm_f16(alsep_ctx* ctx, const float* x, void* y, const float* gamma, const float* beta, int64_t P, int C, float eps, int act,
                          void* workspace);
/* the same for x [T][F][C] with the half result stored [T][C][F] (frequency-contiguous rows per frame and channel): the operand layout
 * of the TDF Linear over the frequency axis as one batched alsep_nn_gemm_f16 per frame (A = the shared weight matrix) */
int alsep_nn_instnorm_f16_t(alsep_ctx* ctx, const float* x, void* y, const float* gamma, const float* beta, int64_t T, int F, int C, float eps,
                            int act, void* workspace);
/* Conv2d on v_mfma_f32_16x16x32_f16: x IEEE half channels-last [B, H, W, Cin], w IEEE half [Cout][KH][KW][Cin], float32 result into the
 * channel slice [y_coff, y_coff + Cout) of y [B, Ho, Wo, y_ctotal], optionally + R (float32 [pixels][ldr]: a block's shortcut branch).
 * Cin % 64 == 0, Cout % 4 == 0; no bias, no activation (this network has neither after a convolution).  Layers with few output tiles are
 * split along K (partial tiles in `workspace`, summed in a fixed order): alsep_nn_conv2d_f16_workspace_bytes says how much they need. */
int64_t alsep_nn_conv2d_f16_workspace_bytes(int64_t B, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h, int stride_w, int pad_h,
                                            int pad_w);
int alsep_nn_conv2d_f16(alsep_ctx* ctx, const void* x, const void* w, float* y, const float* R, int64_t ldr, int64_t B, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int y_ctotal, int y_coff, void* workspace,
                        int64_t workspace_bytes);
/* y = a * b element-wise */
int alsep_nn_mul(alsep_ctx* ctx, const float* a, const float* b, float* y, int64_t n);
/* second half of ConvTranspose2d(kernel = stride = 2): g [H, W, 4*Cout] (1x1 conv, columns (dy*2+dx)*Cout + co) -> channel slice
 * [y_coff, y_coff + Cout) of y [2H, 2W, y_ctotal] */
int alsep_nn_depth_to_space2(alsep_ctx* ctx, const float* g, float* y, int H, int W, int Cout, int y_ctotal, int y_coff);
/* spec [4, f*k, T] (alsep_stft) -> x [T, f, 4k], x[t, ff, c2*k + kk] = spec[c2, kk*f + ff, t] (STFT + cac2cws of tfc_tdf_v3);
 * y [T, f, S*4k] -> spec [S, 4, f*k, T] (cws2cac) */
int alsep_mdx23c_spec_in(alsep_ctx* ctx, const float* spec, float* x, int f, int k, int T);
int alsep_mdx23c_spec_out(alsep_ctx* ctx, const float* y, float* spec, int S, int f, int k, int T);

/* ---- VR multi-band front / back end (in-tree reference: modules/rvc/infer/lib/uvr5_pack/lib_v5/spec_utils.py, modules/rvc/infer/modules/uvr5/
 * vr.py:43-196).  Complex spectrograms are float2-interleaved [2, bins, frames]; band spectrograms from / for alsep_stft / alsep_istft are
 * [4 = (L_re, L_im, R_re, R_im), Fb, Tb]. ---- */
/* out[c, o0 + i, t] = gain[i] * band[c, f0 + i, t0 + t] (i < h, t < l; gain may be NULL): combine_spectrograms (:95-125) and the high-end crop */
int alsep_vr_band_crop(alsep_ctx* ctx, const float* band, float* out, const float* gain, int Fb, int Tb, int f0, int t0, int h, int l,
                       int out_bins, int o0);
/* y = pred * exp(i angle X), v = X - y over n complex values (vr.py:110-111) */
int alsep_vr_split_pred(alsep_ctx* ctx, const float* pred, const float* X, float* y, float* v, int64_t n);
/* spec_utils.mirroring("mirroring") (:453-470): spec_m [2, bins, l], he / out [2, hh, l], lo = pre_filter_start - 10 - hh */
int alsep_vr_mirror(alsep_ctx* ctx, const float* spec_m, const float* he, float* out, int bins, int hh, int l, int lo);
/* one band's spectrogram for the iSTFT (cmb_spectrogram_to_wave :353-429): bins [o0, o0+h) of spec_m at [f0, f0+h), then `extra` [2, eh, l]
 * at [e0, e0+eh) (NULL: none), all times gain[f] (the low-pass / high-pass ramps and zeroed ranges), zero elsewhere; band [4, Fb, l] */
int alsep_vr_band_spec(alsep_ctx* ctx, const float* spec_m, const float* extra, const float* gain, float* band, int bins, int l, int Fb, int f0,
                       int h, int o0, int e0, int eh);

/* ---- half-precision MFMA path of the transformer model families (csrc/nn_half.hip): what torch autocast does to the Linear layers and
 * the attention of the Roformer models the reference loads with use_autocast=True (modules/separator/stem_separator.py:106, :379-382).
 * The residual stream and all statistics are float32; whatever a Linear or the attention reads is IEEE half in HBM, written in that
 * type by its producer; products on the f16 MFMA with float32 accumulation. */
int alsep_nn_to_f16(alsep_ctx* ctx, const float* x, void* y, int64_t n);
/* y[r][C] (half) = lucidrains RMSNorm of x[r][C] (float32) times gamma; strides in elements */
int alsep_nn_rmsnorm_f16(alsep_ctx* ctx, const float* x, void* y, const float* gamma, int64_t rows, int C, int64_t x_stride, int64_t y_stride);
/* C[b][M][N] = act(alpha A[b][M][K] W[b][N][K]^T + bias[b][N]) (+ R[b][M][N]); A, W half; C half (c_f16 != 0) or float32; bias, R float32;
 * row strides ld*, batch strides s*_b in elements (0: shared).  act 0 none, 3 GELU(erf), 5 tanh.  Needs K % 8 == 0, N % 4 == 0 and aligned
 * rows (ALSEP_ERR_ARG otherwise).  n_per_batch (device, optional): columns of batch b (<= N). */
int alsep_nn_gemm_f16(alsep_ctx* ctx, const void* A, int64_t lda, int64_t sa_b, const void* W, int64_t ldw, int64_t sw_b, void* C, int c_f16,
                      int64_t ldc, int64_t sc_b, const float* bias, int64_t bias_b, const float* R, int64_t ldr, int64_t sr_b, int nb, int M,
                      int N, int K, float alpha, int act, const int* n_per_batch);
/* out (half) = softmax(scale q k^T) v per (sequence, head), one pass (no score matrix in HBM).  qkv: half rows of 3 * heads * 64 values
 * (q | k | v); sequence s = L rows `row_stride` elements apart from s * seq_stride; out rows of heads * 64.  rot_table (optional,
 * alsep_nn_rotary_table): q and k are rotary-embedded by their position as they are loaded; gates (optional, float32): the result is
 * scaled by sigmoid(gates[seq * g_seq_stride + row * g_row_stride + head]). */
int alsep_nn_attention_f16(alsep_ctx* ctx, const void* qkv, void* out, int n_seq, int L, int heads, int dim_head, int64_t seq_stride,
                           int64_t row_stride, int64_t o_seq_stride, int64_t o_row_stride, float scale, const float* rot_table,
                           const float* gates, int64_t g_seq_stride, int64_t g_row_stride);
/* Roformer band split, input side, all bands in one launch: feat[band][t][kmax] (half) = RMSNorm over the band's `width[band]` gathered
 * spectrogram values of frame t (spec [4][F][T]; pidx[band][kmax / 2] merged bin index 2 f + s or -1; gamma[band][kmax]), zero-padded */
int alsep_roformer_bandsplit_in(alsep_ctx* ctx, const float* spec, const int* pidx, const float* gamma, const int* width, void* feat, int nb,
                                int F, int T, int kmax);
/* table[pos][j] = (cos, sin)(pos / 10000^(2 j / dim_head)), pos < L, j < dim_head / 2 (rotary_embedding_torch, interleaved pairs) */
int alsep_nn_rotary_table(alsep_ctx* ctx, float* table, int L, int dim_head);

/* ---- reverb impulse-response extraction: replaces handlers/reverb.py:112-172 (extract_reverb), called from
 * modules/separator/stem_separator.py:822-829 when a de-reverb transform ran on the vocals with store_reverb_ir.  Whole-track FFT work in
 * double precision (csrc/reverb.hip).  Audio arguments: float32 device tensors [channels][n] with row stride ld; the mono signals are
 * np.mean over the channels as reverb.py:52-53.  All functions need a workspace of alsep_reverb_workspace_bytes(n_wet, n_dry) bytes
 * (-1: a signal is too long, > 2^27 transform points). */
int64_t alsep_reverb_workspace_bytes(int64_t n_wet, int64_t n_dry);
/* fft_xcorr (reverb.py:55-66) + np.argmax over its n_wet + n_dry - 1 entries (:130-131) -> *argmax_out (host).  Optionally the
 * correlation values at probe_idx[0 .. n_probe) (host indices) -> probe_out (host), for tests.  Synchronises the stream. */
int alsep_reverb_xcorr_argmax(alsep_ctx* ctx, const float* wet, int c_wet, int64_t n_wet, int64_t ld_wet, const float* dry, int c_dry,
                              int64_t n_dry, int64_t ld_dry, void* ws, int64_t ws_bytes, int64_t* argmax_out, double* probe_out,
                              const int64_t* probe_idx, int n_probe);
/* wiener_deconvolution(wet_mono, dry_mono, eps)[:n_out] (reverb.py:94-106, :142-143) -> ir_out (device, float64); the deconvolution has
 * 2 * (n_wet / 2) samples (np.fft.irfft's default length), *n_written (host) = min(n_out, that). */
int alsep_reverb_wiener_ir(alsep_ctx* ctx, const float* wet, int c_wet, int64_t n_wet, int64_t ld_wet, const float* dry, int c_dry,
                           int64_t n_dry, int64_t ld_dry, double eps, void* ws, int64_t ws_bytes, double* ir_out, int64_t n_out,
                           int64_t* n_written);
/* the curve estimate_rt60 fits (reverb.py:74-81), in its float32 arithmetic: 20 log10(sqrt(sum_c x^2) + 1e-10) -> out [n] (device) */
int alsep_reverb_envelope_db(alsep_ctx* ctx, const float* x, int c, int64_t n, int64_t ld, float* out);
/* |np.fft.rfft(x)| of a real float64 device signal of any length n (spectral centroid, reverb.py:155-157) -> mag_out [n / 2 + 1];
 * workspace: (2 n) * 16 + alsep_dft_f64_workspace_bytes(n) bytes */
int alsep_rfft_mag_f64(alsep_ctx* ctx, const double* x, int64_t n, void* ws, int64_t ws_bytes, double* mag_out);
/* complex DFT of any length n <= 2^27 on interleaved float64 (re, im) device data, unnormalised (inverse != 0: the conjugate kernel);
 * in != out; workspace alsep_dft_f64_workspace_bytes(n) */
int64_t alsep_dft_f64_workspace_bytes(int64_t n);
int alsep_dft_f64(alsep_ctx* ctx, const double* in, double* out, int64_t n, int inverse, void* ws, int64_t ws_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ALSEP_H */
