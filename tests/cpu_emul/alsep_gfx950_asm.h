// TEST-ONLY: host emulation of audiolab_amd/csrc/alsep_gfx950_asm.h (shadows it on the include path of
// the emulation build).  Same lane semantics, executed synchronously through the wave slab.
#pragma once

static inline bf16x4 lds_read_tr16_b64(const bf16_t* lds_ptr) {
    char* slab = emul::wave_slab();
    const int l = emul::lane_id();
    std::memcpy(slab + l * 256, (const void*)lds_ptr, 8);
    emul::wave_sync();
    const int base = l & ~15, i = l & 15;
    bf16x4 out;
    for (int q = 0; q < 4; ++q) {
        bf16_t row[4];
        std::memcpy(row, slab + (base + 4 * q + i / 4) * 256, 8);
        out[q] = row[i % 4];
    }
    emul::wave_sync();
    return out;
}

template <int OFF>
static inline bf16x4 lds_read_tr16_b64_off(const bf16_t* lds_ptr) {
    return lds_read_tr16_b64(reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(lds_ptr) + OFF));
}

static inline void lds_read_tr16_wait() {}

template <int N>
static inline void lds_read_tr16_wait_n() {}

template <typename T>
static inline const T* opaque_uniform_ptr(const T* p) { return p; }

#define ALSEP_GLOBAL
template <typename T>
static inline const T* opaque_uniform_gptr(const T* p) { return p; }

template <int OFF>
static inline void global_load_async_bf16x8(bf16x8& dst, const void* sbase, unsigned voff) {
    dst = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(sbase) + voff + OFF);
}

static inline void global_load_async_f32x4(f32x4& dst, const void* sbase, unsigned voff) {
    dst = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(sbase) + voff);
}

template <int OFF>
static inline void lds_read_async_b128(bf16x8& dst, const bf16_t* lds_ptr) {
    std::memcpy(&dst, reinterpret_cast<const char*>(lds_ptr) + OFF, 16);
}
template <int N>
static inline void lds_wait_n() {}
static inline void sched_fence() {}
static inline unsigned long long clock_cycles() { return 0; }
static inline unsigned long long clock_100mhz() { return 0; }

template <typename T>
static inline void keep_vgprs_live(const T&) {}
static inline int opaque_vgpr(int v) { return v; }

typedef float v2f __attribute__((ext_vector_type(2)));
static inline v2f cx_add_mi(v2f a, v2f b) { return v2f{a.x + b.y, a.y - b.x}; }
static inline v2f cx_add_pi(v2f a, v2f b) { return v2f{a.x - b.y, a.y + b.x}; }
static inline v2f cx_mul(v2f a, v2f w) { return v2f{fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.x, w.y, a.y * w.x)}; }
static inline v2f cx_mul_conj(v2f a, v2f w) { return v2f{fmaf(a.x, w.x, a.y * w.y), fmaf(-a.x, w.y, a.y * w.x)}; }
static inline v2f cx_fma_mi(v2f d, v2f s, v2f c) { return v2f{fmaf(s.x, d.y, c.x), fmaf(-s.y, d.x, c.y)}; }
static inline v2f cx_fma_pi(v2f d, v2f s, v2f c) { return v2f{fmaf(-s.x, d.y, c.x), fmaf(s.y, d.x, c.y)}; }
static inline v2f cx_add_conj(v2f a, v2f b) { return v2f{a.x + b.x, a.y - b.y}; }
static inline v2f cx_sub_conj_divi(v2f a, v2f b) { return v2f{a.y + b.y, b.x - a.x}; }
static inline v2f cx_conj_add_pi(v2f a, v2f b) { return v2f{a.x - b.y, -a.y - b.x}; }
#define ALSEP_WAVES_PER_EU(n)
#define ALSEP_WAVES_PER_EU_IF(cond, a, b)
static inline float sgpr_literal(float c) { return c; }
static inline v2f cx_mul_p1(v2f a, v2f w) { return v2f{-(a.y * w.y), a.y * w.x}; }
static inline v2f cx_mul_p2(v2f a, v2f w, v2f t) { return v2f{fmaf(a.x, w.x, t.x), fmaf(a.x, w.y, t.y)}; }
template <int NB>
static inline void load_floats(const void* p, float (&v)[NB]) { std::memcpy(v, p, NB * sizeof(float)); }
