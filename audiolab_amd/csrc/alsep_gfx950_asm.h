// gfx950 instructions that have no usable builtin.  Included as <alsep_gfx950_asm.h>; the test-only
// CPU emulation build puts its own file of the same name earlier on the include path.
//
// ds_read_b64_tr_b16 (hardware transpose read): the clang builtin
// __builtin_amdgcn_ds_read_tr16_b64_* makes hipcc (ROCm 7.2) place "s_waitcnt vmcnt(0)" in front of it
// whenever an LDS-DMA is in flight, which would drain the prefetch ring every k-step; as inline asm
// the compiler does not track it, so the wait for its result is ours (lds_read_tr16_wait) and must be
// followed by a sched_barrier so that no MFMA is hoisted above it (cdna_hip_programming.md rule 18).
#pragma once

// Per 16-lane group: lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4 x 16 block of
// 16-bit elements; lane i receives column i of the 4 rows (row q in element q).  8-byte aligned.
__device__ __forceinline__ bf16x4 lds_read_tr16_b64(const bf16_t* lds_ptr) {
    bf16x4 v;
    const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds_ptr;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// Same read at lds_ptr + OFF bytes, OFF in the instruction's 16-bit immediate: one address VGPR serves a whole
// tile (as a separate "v" operand every constant offset becomes its own loop-invariant register).
template <int OFF>
__device__ __forceinline__ bf16x4 lds_read_tr16_b64_off(const bf16_t* lds_ptr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds offset field");
    bf16x4 v;
    const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds_ptr;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

// Wait for every outstanding LDS read of this wave (the asm reads above are invisible to hipcc's own
// lgkmcnt bookkeeping) and pin the instruction order around the wait.
__device__ __forceinline__ void lds_read_tr16_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Counted form: wait until at most N of this wave's LDS reads are outstanding (they complete in order).
template <int N>
__device__ __forceinline__ void lds_read_tr16_wait_n() {
    static_assert(N >= 0 && N < 16, "lgkmcnt is a 4-bit counter");
    __builtin_amdgcn_s_waitcnt((63 & 15) | ((63 >> 4) << 14) | (7 << 4) | (N << 8));
    __builtin_amdgcn_sched_barrier(0);
}

// A wave-uniform pointer the optimiser must treat as a fresh scalar (SGPR pair) here: keeps loop strength
// reduction from turning "scalar base + 32-bit lane offset" addresses of an unrolled loop into one 64-bit VGPR
// induction pointer per access (which then spill).
template <typename T>
__device__ __forceinline__ const T* opaque_uniform_ptr(const T* p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (const T*)v;
}

// The same, typed as a GLOBAL (address space 1) pointer: a pointer that went through the integer round trip above is
// generic, and loads through it are flat_load + "s_waitcnt vmcnt(0) lgkmcnt(0)" -- no counted waits, so at most two or
// three of an unrolled batch of loads are in flight.  Through this type they are global_load with counted vmcnt.
#define ALSEP_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const ALSEP_GLOBAL T* opaque_uniform_gptr(const T* p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (const ALSEP_GLOBAL T*)v;
}

// Asynchronous 16-byte global load into a register fragment, invisible to hipcc's waitcnt bookkeeping:
//   dst <- *(sbase + voff + OFF)      (sbase wave-uniform SGPR pair, voff a 32-bit lane offset, 0 <= OFF < 4096)
// A compiler-tracked load issued between LDS-DMAs inside a software-pipelined loop is waited for with
// s_waitcnt vmcnt(0) at its first use (ROCm 7.2's SIInsertWaitcnts loses the count across the loop's merges),
// which drains the whole prefetch ring every tile.  As asm the wait is ours: dst must not be read before a
// counted s_waitcnt vmcnt that covers it (wait_vmcnt<N>() in mma.h).  The compiler does not know dst is in
// flight, so it must have no reason to touch it in between (copy, spill): scripts/check_async_regs.py checks the
// generated code for exactly that.
template <int OFF>
__device__ __forceinline__ void global_load_async_bf16x8(bf16x8& dst, const void* sbase, unsigned voff) {
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed global offset field");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

// The same for a float32 quad and a byte offset without immediate (epilogue operands of nn_gemm_h2.h)
__device__ __forceinline__ void global_load_async_f32x4(f32x4& dst, const void* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// Asynchronous 16-byte LDS read into a register fragment, invisible to hipcc's waitcnt bookkeeping (as the global form above):
//   dst <- *(lds_ptr + OFF bytes), OFF in the instruction's 16-bit immediate.
// Why: in a software-pipelined k-loop (fragments of step s+1 requested before the MFMAs of step s) ROCm 7.2 waits
// "s_waitcnt lgkmcnt(0)" at the first use of step s's fragments although the counter is in order and lgkmcnt(7) would do --
// every second step then waits for the reads it has just issued.  As asm the wait is ours (lds_wait_n<N>): dst must not be read
// before it, and the compiler must have no reason to touch dst in between (scripts/check_async_regs.py).
template <int OFF>
__device__ __forceinline__ void lds_read_async_b128(bf16x8& dst, const bf16_t* lds_ptr) {
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds offset field");
    const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds_ptr;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// wait until at most N of this wave's LDS operations are outstanding (they complete in order); pins the instruction order
template <int N>
__device__ __forceinline__ void lds_wait_n() {
    static_assert(N >= 0 && N < 16, "lgkmcnt is a 4-bit counter");
    __builtin_amdgcn_s_waitcnt((63 & 15) | ((63 >> 4) << 14) | (7 << 4) | (N << 8));
    __builtin_amdgcn_sched_barrier(0);
}

// no instruction may be scheduled across this point
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// shader-clock counter (cycles) and the constant 100 MHz counter, for in-kernel phase stamps (timing experiments only)
__device__ __forceinline__ unsigned long long clock_cycles() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long clock_100mhz() { return __builtin_amdgcn_s_memrealtime(); }

// Extends the live range of a register value to this point (no code).
// A per-lane integer the optimiser must treat as a NEW value from here on: values derived from it are recomputed after this point
// instead of being kept alive (or spilled) across the code in between -- e.g. the lane / wave coordinates an epilogue needs again
// after a main loop that has no register to spare.
__device__ __forceinline__ int opaque_vgpr(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <typename T>
__device__ __forceinline__ void keep_vgprs_live(const T& v) {
    asm volatile("" ::"v"(v));
}

// ------------------------------------------------------------------------------------------------
// Packed-f32 complex arithmetic on (re, im) register pairs.  hipcc (ROCm 7.2) does not fold the half swap /
// negation of a multiply by +-i or of a complex product into the op_sel / neg modifiers of v_pk_*_f32 (it emits
// v_xor + v_mov per operand), which doubles the instruction count of an FFT butterfly; these helpers spell the
// modifiers out.  op_sel[i] / op_sel_hi[i] pick the half of source i used for the low / high result.
// ------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2f cx_add_mi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + (+i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f cx_add_pi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x)
__device__ __forceinline__ v2f cx_mul(v2f a, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// The two halves of cx_mul as separate steps.  gfx950 needs one wait state between an op_sel-writing VALU and a dependent
// VALU, and hipcc pads every asm-to-dependent-asm pair with s_nop; batching the first halves of several products
// before their second halves (cx_mul_n) leaves no adjacent dependent pair.
__device__ __forceinline__ v2f cx_mul_p1(v2f a, v2f w) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    return t;
}
__device__ __forceinline__ v2f cx_mul_p2(v2f a, v2f w, v2f t) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
__device__ __forceinline__ v2f cx_mul_conj(v2f a, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// c + (-i) s d = (c.x + s.x d.y, c.y - s.y d.x)   (s: a real scale in both halves)
__device__ __forceinline__ v2f cx_fma_mi(v2f d, v2f s, v2f c) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(d), "v"(s), "v"(c));
    return r;
}
// c + (+i) s d = (c.x - s.x d.y, c.y + s.y d.x)
__device__ __forceinline__ v2f cx_fma_pi(v2f d, v2f s, v2f c) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(d), "v"(s), "v"(c));
    return r;
}
// a + conj(b) = (a.x + b.x, a.y - b.y)
__device__ __forceinline__ v2f cx_add_conj(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a - conj(b)) / i = (a.y + b.y, b.x - a.x)
__device__ __forceinline__ v2f cx_sub_conj_divi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// conj(a + i b) = (a.x - b.y, -a.y - b.x)
__device__ __forceinline__ v2f cx_conj_add_pi(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// register budget of a kernel: exactly n waves per SIMD (512 / n VGPRs + AGPRs per lane)
#define ALSEP_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
// the same with a template-dependent choice (the attribute accepts value-dependent constant expressions)
#define ALSEP_WAVES_PER_EU_IF(cond, a, b) __attribute__((amdgpu_waves_per_eu((cond) ? (a) : (b), (cond) ? (a) : (b))))

// A literal the optimiser must materialise HERE, in an SGPR (s_mov, scalar unit): loop-invariant literals of an
// unrolled body are otherwise hoisted into one VGPR each and spill.
__device__ __forceinline__ float sgpr_literal(float c) {
    asm volatile("" : "+s"(c));
    return c;
}

// NB (2 or 3) consecutive floats from a 4-byte aligned address as ONE global_load_dwordx2 / x3.  The vector type of the
// 3-float case is 16 bytes wide in C++ (the instruction reads 12), so the host emulation has its own byte-exact version.
template <int NB>
__device__ __forceinline__ void load_floats(const void* p, float (&v)[NB]) {
    typedef float vec_t __attribute__((ext_vector_type(NB), aligned(4)));
    const vec_t q = *reinterpret_cast<const vec_t*>(p);
#pragma unroll
    for (int i = 0; i < NB; ++i) v[i] = q[i];
}
