// Shared declarations of libalsep's implementation files (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/alsep.h"

// All kernels use dynamic LDS only, through this one 16-byte aligned symbol
// (cdna_hip_programming.md Guideline 17: no static __shared__ in front of it).
extern __shared__ __attribute__((aligned(16))) char alsep_smem[];

// The 16-bit storage type of the half-precision kernels.  Every such kernel is written once against `bf16_t`; the translation units
// tdfnet_f16.hip / fft_f16.hip compile the same sources a second time with ALSEP_F16_TU defined, where the type is IEEE binary16
// (_Float16: 10 mantissa bits instead of 7, the reference's autocast type) and the MFMA is v_mfma_f32_16x16x32_f16.  Their entry
// points carry the suffix _f16tu and are reached through the dtype dispatch of the public functions (ALSEP_F16).
#ifdef ALSEP_F16_TU
typedef _Float16 bf16_t;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
#define ALSEP_HALF_DTYPE ALSEP_F16
#define ALSEP_TU_NAME(name) name##_f16tu
#else
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define ALSEP_HALF_DTYPE ALSEP_BF16
#define ALSEP_TU_NAME(name) name
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct alsep_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline)
    int prof_category = 0;
    std::vector<hipEvent_t> prof_events;     // start/stop pairs
    size_t prof_used = 0;
    double prof_flops = 0.0, prof_bytes = 0.0;   // work of the bracketed launches, added by their launch sites (ProfScope::work)
    // launches per kernel name since alsep_create / alsep_launch_counts_reset (tests assert WHICH kernel ran)
    std::map<std::string, int64_t> launches;
    int cu_count = 0;                        // multiprocessors of `device`, read once per ctx
    // generic float32 GEMM / convolution (alsep_nn_*): 1 = split-half contraction on the f16 matrix pipe (alsep_nn_set_contraction);
    // nn_range: one device word the split kernels raise when an operand leaves the half range (allocated with the first switch to 1)
    int nn_split = 0;
    unsigned* nn_range = nullptr;
    void* zero_page = nullptr;               // 256 zero bytes, allocated by the first kernel that pads a ragged K from it (nn_gemm_h2.h)
};

// multiprocessors of the context's device (256 on MI355X), read once
static inline int device_cu_count(alsep_ctx* ctx) {
#ifdef ALSEP_CPU_EMUL
    (void)ctx;
    return 256;
#else
    if (ctx->cu_count <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || v <= 0) v = 256;
        ctx->cu_count = v;
    }
    return ctx->cu_count;
#endif
}

static inline void note_launch(alsep_ctx* ctx, const char* kernel) {
    if (ctx) ++ctx->launches[kernel];
}

// RAII bracket: records a start/stop event pair around the launches of one kernel class when
// that class is being profiled (alsep_profile_begin); otherwise costs one integer compare.
struct ProfScope {
    alsep_ctx* ctx;
    bool on;
    ProfScope(alsep_ctx* c, int category) : ctx(c), on(c && c->prof_category == category) {
        if (!on) return;
        if (ctx->prof_used + 2 > ctx->prof_events.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            ctx->prof_events.push_back(a);
            ctx->prof_events.push_back(b);
        }
        (void)hipEventRecord(ctx->prof_events[ctx->prof_used], ctx->stream);
    }
    void work(double flops, double bytes) {
        if (on) { ctx->prof_flops += flops; ctx->prof_bytes += bytes; }
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(ctx->prof_events[ctx->prof_used + 1], ctx->stream);
        ctx->prof_used += 2;
    }
};

static inline int alsep_fail(alsep_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define ALSEP_HIP(ctx, call)                                                              \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return alsep_fail((ctx), ALSEP_ERR_HIP, "%s failed: %s (%s:%d)", #call,       \
                              hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)

// every entry point that allocates or launches makes the ctx's device current for its own duration and restores the caller's
// device on every return path (a Context('cuda:1') used from a thread whose current device is 0 must neither put its tables on
// device 0 nor leave the thread on device 1: torch's index-less allocations and current_stream() calls follow the current device)
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == device) return hipSuccess;
        hipError_t e = hipSetDevice(device);
        switched = (e == hipSuccess) && prev >= 0;
        return e;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

#define ALSEP_ENTER(ctx)                                                                  \
    DeviceGuard alsep_device_guard_;                                                      \
    do {                                                                                  \
        if ((ctx) && alsep_device_guard_.enter((ctx)->device) != hipSuccess)              \
            return alsep_fail((ctx), ALSEP_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->device); \
    } while (0)

#define ALSEP_LAUNCH_CHECK(ctx, what)                                                     \
    do {                                                                                  \
        note_launch((ctx), (what));                                                       \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess)                                                             \
            return alsep_fail((ctx), ALSEP_ERR_HIP, "launch of %s failed: %s", (what),    \
                              hipGetErrorString(e_));                                     \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = ALSEP_F32; };
template <> struct dtype_of<bf16_t> { static constexpr int value = ALSEP_HALF_DTYPE; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// A 16-byte store of a streamed activation (written once, read by a later kernel after gigabytes of other traffic).  ALSEP_NT_STORES=1
// makes these stores non-temporal (no L2 allocation); which one the product uses is decided by a same-box A/B (profiles/r04_nt_stores_ab.txt).
#ifndef ALSEP_NT_STORES
#define ALSEP_NT_STORES 0
#endif
template <typename P, typename V>
__device__ __forceinline__ void stream_store(P p, const V& v) {
#if ALSEP_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// XCD-aware remap of a linear workgroup id: the dispatcher deals consecutive ids round-robin
// over the 8 XCDs, so ids b and b+8 share an L2.  Give each XCD a contiguous run of tiles
// (bijective for any n; cdna_hip_programming.md T1).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// internal entry points shared between files
int alsep_plan_tables(alsep_ctx* ctx, alsep_plan* plan);
// fft.hip / fft_f16.hip: STFT with the first 1x1 convolution of the network in its epilogue (ALSEP_ERR_STATE: no such kernel for this plan)
int ALSEP_TU_NAME(alsep_stft_first_conv)(alsep_ctx* ctx, const alsep_plan* plan, const float* pcm, int64_t ch_stride, int64_t chunk_stride,
                                         int64_t n_chunks, void* act, const float* w, const float* scale, const float* shift, int g, float in_scale,
                                         int zero_low);
