// TEST-ONLY host emulation of the small HIP surface the alsep kernels use.
//
// Purpose: run the *unchanged* kernel sources of audiolab_amd/csrc on the CPU under
// AddressSanitizer (GPU ASan is not available on the MI355X pool), so that indexing /
// bounds errors are caught before a kernel is ever launched on a real GPU, and so that
// `pytest -m "not gpu"` exercises the kernels' logic against the oracle at small sizes.
//
// This directory shadows <hip/hip_runtime.h> for the emulation build ONLY
// (tests/cpu_emul/build_emul.sh, host clang++).  The product library
// (audiolab_amd/lib/libalsep.so, hipcc --offload-arch=gfx950) never sees this file, and
// no product code loads the emulated library.
//
// Model: the GPU threads of a workgroup are cooperative fibers on one OS thread (they yield
// only at barriers); workgroups are spread over a few OS threads; wave-collective builtins
// (MFMA, shuffles) exchange operands through a per-wave slab guarded by a 64-lane barrier.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define ALSEP_CPU_EMUL 1
#define __global__
#define __device__
#define __host__
#define __shared__ thread_local   /* LDS: one copy per worker OS thread (= per running workgroup) */
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
typedef void* hipStream_t;
typedef struct emul_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };

hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void* f, int attr, int value);
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace emul {
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void sync_block();
// wave-collective exchange: every lane of the calling wave deposits `bytes` at slot
// [lane] of the wave slab, waits for the others, then may read any lane's slot until
// wave_done() (a second wave barrier).
char* wave_slab();          // base of this wave's slab (64 slots x 256 B)
int lane_id();
void wave_sync();
}  // namespace emul

static inline void __syncthreads() { emul::sync_block(); }
// device math the host libm lacks
#define __expf(x) std::exp((float)(x))   /* glibc declares, but does not export, a symbol of this name */
static inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }   /* v_exp_f32 */
static inline float __builtin_amdgcn_rcpf(float x) { return 1.f / x; }   /* v_rcp_f32: 1 ulp on the device */
static inline void sincospi(double x, double* s, double* c) { *s = std::sin(M_PI * x); *c = std::cos(M_PI * x); }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                       \
    do {                                                                                   \
        (void)(stream);                                                                    \
        emul::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); });            \
    } while (0)

// ---- vector types / MFMA emulation (layouts per cdna_hip_programming.md section 3) -------
typedef __bf16 emul_bf16x8 __attribute__((ext_vector_type(8)));
typedef float emul_f32x4 __attribute__((ext_vector_type(4)));

// D[i][j] += sum_k A[i][k] B[k][j];  lane l: A[l&15][8*(l>>4)+e], B[8*(l>>4)+e][l&15], e=0..7;
// C/D: col = l&15, row = 4*(l>>4)+r.
static inline emul_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(emul_bf16x8 a, emul_bf16x8 b,
                                                                  emul_f32x4 c, int, int, int) {
    char* slab = emul::wave_slab();
    const int l = emul::lane_id();
    std::memcpy(slab + l * 256, &a, 16);
    std::memcpy(slab + l * 256 + 16, &b, 16);
    emul::wave_sync();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float s = c[r];
        for (int g = 0; g < 4; ++g) {
            emul_bf16x8 av, bv;
            std::memcpy(&av, slab + (16 * g + row) * 256, 16);
            std::memcpy(&bv, slab + (16 * g + col) * 256 + 16, 16);
            for (int e = 0; e < 8; ++e) s += (float)av[e] * (float)bv[e];
        }
        c[r] = s;
    }
    emul::wave_sync();
    return c;
}

typedef _Float16 emul_f16x8 __attribute__((ext_vector_type(8)));
static inline emul_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_f16(emul_f16x8 a, emul_f16x8 b, emul_f32x4 c, int, int, int) {
    char* slab = emul::wave_slab();
    const int l = emul::lane_id();
    std::memcpy(slab + l * 256, &a, 16);
    std::memcpy(slab + l * 256 + 16, &b, 16);
    emul::wave_sync();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float s = c[r];
        for (int g = 0; g < 4; ++g) {
            emul_f16x8 av, bv;
            std::memcpy(&av, slab + (16 * g + row) * 256, 16);
            std::memcpy(&bv, slab + (16 * g + col) * 256 + 16, 16);
            for (int e = 0; e < 8; ++e) s += (float)av[e] * (float)bv[e];
        }
        c[r] = s;
    }
    emul::wave_sync();
    return c;
}

// lane l: A[l&15][l>>4], B[l>>4][l&15]; k-ordered fmaf chain (matches the hardware's numerics).
static inline emul_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emul_f32x4 c,
                                                               int, int, int) {
    char* slab = emul::wave_slab();
    const int l = emul::lane_id();
    std::memcpy(slab + l * 256, &a, 4);
    std::memcpy(slab + l * 256 + 4, &b, 4);
    emul::wave_sync();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float s = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            std::memcpy(&av, slab + (16 * k + row) * 256, 4);
            std::memcpy(&bv, slab + (16 * k + col) * 256 + 4, 4);
            s = fmaf(av, bv, s);
        }
        c[r] = s;
    }
    emul::wave_sync();
    return c;
}

// The emulation executes memory operations synchronously: counted waits are no-ops, the raw
// barrier is the workgroup barrier.
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline void __builtin_amdgcn_s_barrier() { emul::sync_block(); }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // callers pass wave-uniform values only
// lanes are separate fibers here: a wave-level ordering point must actually rendezvous
static inline void __builtin_amdgcn_wave_barrier() { emul::wave_sync(); }

// LDS-DMA: lane l copies `size` bytes from its own global address to (wave-uniform LDS base) + size*l + off.
template <typename GP, typename LP>
static inline void __builtin_amdgcn_global_load_lds(GP g, LP l, unsigned size, unsigned off, unsigned) {
    std::memcpy((char*)(void*)l + off + (size_t)size * emul::lane_id(), (const void*)g, size);
}

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block of 16-bit elements; lane i receives column i of the 4 rows (row q in element q).
typedef __bf16 emul_bf16x4 __attribute__((ext_vector_type(4)));
template <typename LP>
static inline emul_bf16x4 __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LP p) {
    char* slab = emul::wave_slab();
    const int l = emul::lane_id();
    std::memcpy(slab + l * 256, (const void*)p, 8);
    emul::wave_sync();
    const int base = l & ~15, i = l & 15;
    emul_bf16x4 out;
    for (int q = 0; q < 4; ++q) {
        __bf16 row[4];
        std::memcpy(row, slab + (base + 4 * q + i / 4) * 256, 8);
        out[q] = row[i % 4];
    }
    emul::wave_sync();
    return out;
}

template <typename T>
static inline T emul_shfl_src(T v, int src_lane) {
    char* slab = emul::wave_slab();
    const int l = emul::lane_id();
    std::memcpy(slab + l * 256, &v, sizeof(T));
    emul::wave_sync();
    T out;
    std::memcpy(&out, slab + (src_lane & 63) * 256, sizeof(T));
    emul::wave_sync();
    return out;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int = 64) { return emul_shfl_src(v, emul::lane_id() ^ mask); }
template <typename T> static inline T __shfl_down(T v, int d, int = 64) {
    const int l = emul::lane_id();
    return emul_shfl_src(v, l + d < 64 ? l + d : l);
}
template <typename T> static inline T __shfl(T v, int src, int = 64) { return emul_shfl_src(v, src); }

float atomicAdd(float* p, float v);
int atomicAdd(int* p, int v);
unsigned atomicAdd(unsigned* p, unsigned v);
unsigned atomicMax(unsigned* p, unsigned v);
int atomicMax(int* p, int v);
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
using std::max;
using std::min;
