// MFMA fragment helpers for gfx950 (wave64), shared by the conv / GEMM kernels.
//
// Both precisions use 16x16 output tiles so that epilogues are shared:
//   bf16: v_mfma_f32_16x16x32_bf16 -- lane l holds A[l&15][8*(l>>4)+e], B[8*(l>>4)+e][l&15], e<8
//   f32 : v_mfma_f32_16x16x4_f32   -- lane l holds A[l&15][l>>4],      B[l>>4][l&15]
//   C/D : col = l&15, row = 4*(l>>4)+r, r<4            (cdna_hip_programming.md section 3)
// A "k-group" is 16 bytes of consecutive k for one row/column (8 bf16 or 4 f32); one "k-step"
// is 4 groups (one per lane quarter).  In f32 a step is issued as four 16x16x4 MFMAs that take
// element e of every lane's group: the k order inside a step is permuted identically for both
// operands, which leaves the sum unchanged.
#pragma once
#include "alsep_common.h"

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
    typedef bf16x8 type;
    static constexpr int G = 8;       // elements per 16-byte k-group
};
template <> struct Frag<float> {
    typedef f32x4 type;
    static constexpr int G = 4;
};

template <typename T>
__device__ __forceinline__ typename Frag<T>::type lds_frag(const T* p) {
    return *reinterpret_cast<const typename Frag<T>::type*>(p);
}

__device__ __forceinline__ void mma_step(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#ifdef ALSEP_F16_TU
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
#else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
#endif
}
__device__ __forceinline__ void mma_step(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
}

// 16-byte global <-> LDS staging word
struct alignas(16) vec16 { uint32_t w[4]; };
__device__ __forceinline__ vec16 zero16() { vec16 v; v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0u; return v; }

// store 4 consecutive channels (the r = 0..3 accumulator rows of one lane)
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    bf16x4 q;
    q[0] = (bf16_t)v[0]; q[1] = (bf16_t)v[1]; q[2] = (bf16_t)v[2]; q[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = q;
}
// store 8 consecutive channels as one 16-byte access
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 q;
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = (bf16_t)v[e];
    stream_store(reinterpret_cast<bf16x8*>(p), q);
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    const bf16x4 q = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
}

// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt fields left at their maxima); gfx9+ encoding:
// vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt = [6:4], lgkmcnt = [11:8].
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
// workgroup barrier that does NOT drain outstanding LDS-DMA / global loads: LDS operations of this
// wave are completed (lgkmcnt(0)), vmcnt is left to the caller's counted waits.
__device__ __forceinline__ void barrier_nodrain() {
    __builtin_amdgcn_s_waitcnt((63 & 15) | ((63 >> 4) << 14) | (7 << 4) | (0 << 8));
    __builtin_amdgcn_s_barrier();
}
