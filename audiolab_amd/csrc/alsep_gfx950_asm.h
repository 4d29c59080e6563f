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

// Wait for every outstanding LDS read of this wave (the asm reads above are invisible to hipcc's own
// lgkmcnt bookkeeping) and pin the instruction order around the wait.
__device__ __forceinline__ void lds_read_tr16_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
