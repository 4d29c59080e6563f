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

static inline void lds_read_tr16_wait() {}
