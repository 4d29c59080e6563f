// Half-precision GEMM of the transformer families at their large-M shapes (M = frames x bands = 48 060 rows against K = 384 ... 1536):
// C[M][N] = act(alpha A[M][K] W[N][K]^T + bias) (+ residual), the contract of nn_gemm_hh_kernel (nn_half.hip), as a PERSISTENT kernel whose
// operand stream never stops at a tile boundary.
//
// Why a second kernel: at K = 384 a 128 x 128 tile is six K slices long, and nn_gemm_hh_kernel pays a cold prologue (two slices requested,
// nothing to compute), and an epilogue with nothing in flight, per tile: measured 126 us for 48 060 x 1536 x 384 (0.45 PFLOP/s, MFMA pipe
// 18 % busy, 27 GB/s of operand intake per CU).  Here
//   * one workgroup per CU (8 waves, 256 x 128 tile: 2/3 of the operand bytes per flop of the 128 x 128 tile) walks its tiles as ONE
//     sequence of K slices -- slice s + 2 is requested (LDS-DMA, no staging registers) while slice s is multiplied, across tile ends, so the
//     epilogue of a tile runs with the next tile's first two slices already in flight;
//   * three 48 KB stages (A 256 rows x 128 B, W 128 rows x 128 B), one barrier per slice, counted vmcnt waits;
//   * the tiles of one XCD are a contiguous run (n fastest), taken 32 at a time by its 32 workgroups: the 12 workgroups that read the same
//     256 rows of A do so at the same time from the same L2;
//   * W rows are permuted on the way into LDS so that a lane's two accumulator blocks of a pair hold 8 CONSECUTIVE columns: the C store is
//     16 bytes (half) / 32 bytes (float32) per lane, 64 / 128 contiguous bytes per row per instruction.
// Rows beyond M and columns beyond N read a clamped (valid) address and are never stored -- rows / columns of a GEMM are independent, so
// they need no zeroing; K groups beyond a ragged K (RAGK) read a zero page.  (Per-batch ragged N -- GemmHArgs::nvec -- stays with
// nn_gemm_hh_kernel: its only users are the mask estimators' last layers, M = 801.)
#pragma once
#include "mma.h"
#include <alsep_gfx950_asm.h>

namespace h2 {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 128, BK = 64, STAGES = 3;
// Tile height BM = 256 (8 waves as 4 x 2) or 192 (6 waves as 3 x 2): every wave owns 64 x 64 of the tile either way.  The launcher takes
// the height whose tile count fills the 256 workgroups' rounds better: 48 060 rows x 384 columns are 564 tiles of 256 (2.2 per workgroup:
// three rounds, the third a fifth full) but 753 tiles of 192 (2.94).
template <int BM> struct Geo {
    static constexpr int kWaves = BM / 32, kThreads = 64 * kWaves;
    static constexpr int kStageHalves = (BM + BN) * BK;                      // 48 KB / 40 KB
    static constexpr int kInstr = (BM + BN) / 8;                             // 1 KB LDS-DMA instructions per stage: 48 / 40
    static constexpr int kDmaPerWave = (kInstr + kWaves - 1) / kWaves;       // 6 / 7 per wave (BM = 192: the two surplus ones land in a scratch KB)
    static constexpr size_t kLds = (size_t)STAGES * kStageHalves * sizeof(_Float16) + 1024;
    static_assert(BM / 8 == 4 * kWaves, "four A instructions per wave");
};

struct Args {
    const _Float16* A; int64_t lda, sa_b;
    const _Float16* B; int64_t ldb, sb_b;
    void* C; int64_t ldc, sc_b;
    const float* bias; int64_t bias_b;
    const float* R; int64_t ldr, sr_b;
    int M, N, K;
    float alpha;
    const _Float16* zero_page;                                       // >= 16 zero bytes: ragged K groups, and the bias of a product without one
    int tiles_m, tiles_n, ntiles;
};

__device__ __forceinline__ int slot(int row, int g) { return row * BK + 8 * (g ^ (row & 7)); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0,
                                     0);
}

// LDS row rho of the W tile holds column perm(rho) of the tile: inside every group of 32, block 2 j' + h, row 4 lq + r <-> column
// 32 j' + 8 lq + 4 h + r, so that a lane (lq) owns columns 8 lq .. 8 lq + 7 across the block pair
__device__ __forceinline__ int w_perm(int rho) {
    const int r32 = rho & 31;
    return (rho & ~31) + 8 * ((r32 >> 2) & 3) + 4 * (r32 >> 4) + (r32 & 3);
}

template <int BM, int ACT, bool CF16, bool RES, bool RAGK, bool STAMP = false, typename ActFn>
__device__ __forceinline__ void gemm_body(const Args& p, ActFn actf, unsigned long long* stamps = nullptr) {
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = 0, tk0 = 0;
    auto stamp = [&](int k) {
        if constexpr (STAMP) {
            const unsigned long long now = clock_cycles();
            tacc[k] += now - tlast;
            tlast = now;
        }
    };
    if constexpr (STAMP) tk0 = tlast = clock_cycles();
    typedef Geo<BM> G;
    constexpr int kWaves = G::kWaves, kStageHalves = G::kStageHalves, kDmaPerWave = G::kDmaPerWave;
    _Float16* lds = reinterpret_cast<_Float16*>(alsep_smem);
    _Float16* const scratch = lds + (size_t)STAGES * kStageHalves;          // 1 KB: where a surplus request lands
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = (p.K + BK - 1) / BK;
    // this workgroup's tiles: XCD x = id & 7 owns the contiguous run [lo, hi); its workgroups take them `per` at a time
    const int per = (int)gridDim.x >> 3, x = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
    const int lo = (int)(((int64_t)p.ntiles * x) >> 3), hi = (int)(((int64_t)p.ntiles * (x + 1)) >> 3);
    const int my_tiles = jx < hi - lo ? (hi - lo - jx + per - 1) / per : 0;
    if (my_tiles == 0) return;
    const int total = my_tiles * nk;
    const int per_b = p.tiles_m * p.tiles_n;

    // ---- issue side: the (tile, slice) the next request belongs to, and this lane's six source rows of it
    const int lrow = lane >> 3, lg = (lane & 7) ^ lrow;              // row inside an instruction's 8 rows; k-group this lane fetches
    int iss_n = 0, iss_k = 0, iss_stage = 0;                         // requests made so far, K offset of the next one, its stage
    const _Float16* iss_a = nullptr;
    const _Float16* iss_b = nullptr;
    unsigned off[kDmaPerWave];
    auto iss_tile = [&](int k) {
        const int t = lo + jx + (k < my_tiles ? k : my_tiles - 1) * per;
        const int bz = t / per_b, rem = t % per_b;
        const int m0 = (rem / p.tiles_n) * BM, n0 = (rem % p.tiles_n) * BN;
        const int Nb = p.N;
        iss_a = p.A + bz * p.sa_b;
        iss_b = p.B + bz * p.sb_b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = m0 + 8 * (wave + kWaves * j) + lrow;
            off[j] = (unsigned)((r < p.M ? r : p.M - 1) * (int)p.lda) + 8u * lg;
        }
#pragma unroll
        for (int j = 4; j < kDmaPerWave; ++j) {
            const int wi = wave + kWaves * (j - 4);                        // W instruction: LDS rows 8 wi .. 8 wi + 7 (BM = 192: wi >= 16 is surplus)
            const int c = n0 + w_perm(8 * (wi < 16 ? wi : 0) + lrow);
            off[j] = (unsigned)((c < Nb ? c : (Nb > 0 ? Nb - 1 : 0)) * (int)p.ldb) + 8u * lg;
        }
    };
    // one slice's six (BM = 192: seven) requests are issued as single instructions BETWEEN the MFMAs of the slice being multiplied (issue_one(j) from
    // compute()): all eight waves issuing their six right behind the barrier kept the CU's one address path busy for ~800 cycles with the
    // matrix pipes idle, and then the matrix pipes for ~850 with the address path idle (phase stamps, scripts/dbg/gemm_dev.hip)
    _Float16* iss_dst = lds;
    bool iss_zero = false;
    auto issue_prep = [&]() {
        iss_dst = lds + (size_t)iss_stage * kStageHalves + (size_t)wave * 512;
        iss_zero = RAGK && iss_k + 8 * lg >= p.K;
    };
    auto issue_one = [&](int j) {
        const _Float16* src = (j < 4 ? iss_a : iss_b) + (off[j] + (unsigned)iss_k);
        _Float16* dst = iss_dst + (size_t)j * kWaves * 512;                 // instruction wave + kWaves j of the stage (A first, then W)
        if (kWaves * kDmaPerWave > G::kInstr && j == kDmaPerWave - 1 && wave + kWaves * j >= G::kInstr) dst = scratch;
        glds16(iss_zero ? p.zero_page : src, dst);
    };
    auto issue_done = [&]() {
        iss_stage = iss_stage == STAGES - 1 ? 0 : iss_stage + 1;
        ++iss_n;
        if (iss_n < total) {                                         // past the end: the last slice again (same count of requests in flight)
            iss_k += BK;
            if (iss_k >= p.K) {
                iss_k = 0;
                iss_tile(iss_n / nk);
            }
        }
    };
    auto issue = [&]() {
        issue_prep();
#pragma unroll
        for (int j = 0; j < kDmaPerWave; ++j) issue_one(j);
        issue_done();
    };

    f32x4 acc[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto compute = [&](int stage) {
        const _Float16* As = lds + (size_t)stage * kStageHalves;
        const _Float16* Bs = As + BM * BK;
        issue_prep();
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            h16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const h16x8*>(As + slot(wm * 64 + i * 16 + l15, 4 * st + lq));
                bf[i] = *reinterpret_cast<const h16x8*>(Bs + slot(wn * 64 + i * 16 + l15, 4 * st + lq));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
                const int slot_ = i < 3 ? 3 * st + i : (st == 1 ? 6 : -1);   // three requests per k-step, each behind four MFMAs; a seventh last
                if (slot_ >= 0 && slot_ < kDmaPerWave) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_one(slot_);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        issue_done();
    };
    // ---- epilogue operands (bias: 4 x 16 bytes per lane; residual: 16 x 16 bytes), requested at the top of a tile's LAST slice, in front of
    // that iteration's LDS-DMA: by the time the MFMAs of the slice are through they have had a whole iteration to arrive, and the counted
    // wait that covers them ("at most the 6 requests of slice s + 2 outstanding") leaves the operand stream untouched.  As asm: a load the
    // compiler tracks inside this loop is waited for with vmcnt(0) at the loop head (alsep_gfx950_asm.h), which drains the ring every slice.
    f32x4 bv[2][2], rv[RES ? 4 : 1][2][2];
    auto tile_origin = [&](int k, int& bz, int& m0, int& n0) {
        const int t = lo + jx + k * per;
        bz = t / per_b;
        const int rem = t % per_b;
        m0 = (rem / p.tiles_n) * BM;
        n0 = (rem % p.tiles_n) * BN;
    };
    auto epilogue_request = [&](int k) {
        int bz, m0, n0;
        tile_origin(k, bz, m0, n0);
        const int cmax = p.N >= 4 ? p.N - 4 : 0;
        // no bias: the same four loads from the zero page (a conditional around the asm would make bv a merged value the compiler copies
        // before the data is in)
        const float* bias = p.bias ? p.bias + bz * p.bias_b : reinterpret_cast<const float*>(p.zero_page);
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = n0 + wn * 64 + 32 * jp + 8 * lq + 4 * h;
                global_load_async_f32x4(bv[jp][h], bias, p.bias ? 4u * (unsigned)(col < cmax ? col : cmax) : 0u);
            }
        if constexpr (RES) {
            const float* res = p.R + bz * p.sr_b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + wm * 64 + i * 16 + l15;
                const unsigned ro = (unsigned)((row < p.M ? row : p.M - 1) * (int)p.ldr);
#pragma unroll
                for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int col = n0 + wn * 64 + 32 * jp + 8 * lq + 4 * h;
                        global_load_async_f32x4(rv[i][jp][h], res, 4u * (ro + (unsigned)(col < cmax ? col : cmax)));
                    }
            }
        }
    };
    auto epilogue = [&](int k) {
        int bz, m0, n0;
        tile_origin(k, bz, m0, n0);
        const int Nb = p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + wm * 64 + i * 16 + l15;
            const bool row_ok = row < p.M;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const int col = n0 + wn * 64 + 32 * jp + 8 * lq;
                const bool first = row_ok && col < Nb, both = row_ok && col + 4 < Nb;
                f32x4 v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = actf(p.alpha * acc[i][2 * jp + h][r] + bv[jp][h][r]);
                        if constexpr (RES) t += rv[i][jp][h][r];
                        v[h][r] = t;
                    }
                if (CF16) {
                    _Float16* c = reinterpret_cast<_Float16*>(p.C) + bz * p.sc_b + (int64_t)row * p.ldc + col;
                    h16x8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[e >> 2][e & 3];
                    if (both) *reinterpret_cast<h16x8*>(c) = hv;
                    else if (first) *reinterpret_cast<h16x4*>(c) = __builtin_shufflevector(hv, hv, 0, 1, 2, 3);
                } else {
                    float* c = reinterpret_cast<float*>(p.C) + bz * p.sc_b + (int64_t)row * p.ldc + col;
                    if (first) *reinterpret_cast<f32x4*>(c) = v[0];
                    if (both) *reinterpret_cast<f32x4*>(c + 4) = v[1];
                }
            }
        }
    };

    iss_tile(0);
    issue();
    issue();
    zero_acc();
    int stage = 0;
    // Requests in flight at the top of a slice, oldest first: slice s, slice s + 1 (kDmaPerWave each): "at most kDmaPerWave outstanding" =
    // slice s is in (loads complete in order).  The first slice after a tile end needs no wait: the epilogue's wait has covered it.
    // A tile's LAST slice is written out on its own: the epilogue operands are requested, multiplied past and consumed in one region of
    // straight-line code.  (Requested in one conditional of a flat slice loop and consumed in another they were loop-carried values to
    // hipcc, which copied the freshly "defined" registers elsewhere right behind the asm -- before the data had landed -- and reused
    // the originals: wrong residuals, then wild addresses, on one launch in six.  scripts/check_async_regs.py, tests/test_codegen.py.)
    for (int tile_k = 0; tile_k < my_tiles; ++tile_k) {
        for (int ks = 0; ks < nk - 1; ++ks) {
            if (ks > 0 || tile_k == 0) wait_vmcnt<kDmaPerWave>();
            stamp(0);
            barrier_nodrain();                                  // slice s is in LDS for every wave; stage (s + 2) % 3 was read by s - 1: free
            stamp(1);
            __builtin_amdgcn_sched_barrier(0);
            compute(stage);
            stamp(3);
            stage = stage == STAGES - 1 ? 0 : stage + 1;
        }
        if (nk > 1 || tile_k == 0) wait_vmcnt<kDmaPerWave>();
        stamp(0);
        barrier_nodrain();
        stamp(1);
        __builtin_amdgcn_sched_barrier(0);
        epilogue_request(tile_k);
        __builtin_amdgcn_sched_barrier(0);
        stamp(2);
        compute(stage);
        stamp(3);
        stage = stage == STAGES - 1 ? 0 : stage + 1;
        // oldest first: slice s + 1, the epilogue operands, slice s + 2: everything but the newest requests (slice s + 2) is in
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<kDmaPerWave>();
        __builtin_amdgcn_sched_barrier(0);
        stamp(4);
        epilogue(tile_k);
        zero_acc();
        stamp(5);
    }
    wait_vmcnt<0>();                                                 // the surplus requests land before the wave ends
    if constexpr (STAMP) {
        if (lane == 0 && stamps) {
            unsigned long long* o = stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
            for (int k = 0; k < 6; ++k) o[k] = tacc[k];
            o[6] = clock_cycles() - tk0;
            o[7] = total;
        }
    }
}

}  // namespace h2
