// K3c: fused layer runs, second structure -- two 4-wave groups per workgroup that alternate roles.
//
// Replaces the same reference lines as chain.hip (MLP.forward, models/mlp.py:47-59, and its autograd data-gradient) for
// runs without skip layers; chain.hip keeps the skip-layer instantiation and stays selectable (PAPR_CHAIN=1) for A/B.
//
// What round 1's profile of chain.hip showed: two independent 4-wave workgroups per CU nearly serialise in training (one
// alone: 83k cycles per 4-layer tile, two: 78k per tile and CU) -- their row-per-lane stores (32 partial cache lines per
// instruction) and their weight-fragment loads queue in the same texture-addresser, and nothing keeps one workgroup in
// its k-loop while the other is in its row phases.  Here:
//
//   * ONE workgroup of 8 waves per CU = two groups of four (waves w and w + 4 share a SIMD).  Each group owns a 64-row
//     tile and its own 64 KB of A planes.  The groups run the same program half a period apart: while group A
//     multiplies (k-loop: MFMA + weight-fragment loads + A-fragment LDS reads), group B does everything else (row
//     phases), so the matrix pipe of every SIMD always has exactly one wave feeding it and that wave's partner fills the
//     vector / memory issue slots.  Two workgroup barriers per layer (after the k-loop, after the row phases) keep the
//     half-period offset; group 1 starts one barrier late, group 0 ends one barrier late.
//   * Row phases in a COALESCED layout.  The accumulators (C^T: lane = row, registers = 4-column runs) are dumped as raw
//     fp32 into the group's own A-plane space -- dead between the end of the k-loop and the next split -- and read back
//     with one wave per 16 rows, a lane holding 4 consecutive columns of a whole row.  In that layout: a row's 1 KB goes
//     out in ONE store instruction (8 full cache lines instead of 32 partial ones), the row maximum is a wave reduction
//     (no LDS atomics, no table, no barrier before the split), the LayerNorm cores are wave reductions in a fixed order,
//     the activation signs are four ballots per row, and the data-gradient run reads them back with one scalar load per
//     row and applies them as lane masks.  The fp32 image of a row occupies exactly the bytes of that row's two plane
//     rows (columns 0-127 where its hi row goes, 128-255 where its lo row goes), so a row is read, processed and
//     overwritten by one wave, four rows at a time: no barrier, no 64-register row buffer.
//     The only cross-wave step is dump -> read, a 4-wave counter barrier in LDS that the multiplying group never sees.
//   * Bank conflicts by XOR instead of padding (the planes must tile 64 KB exactly): 16-byte chunk c of plane row r lives
//     at chunk c ^ (r & 15); fp32 chunk c of a half row at c ^ (r & 7).
//
// Arithmetic is unchanged (same MFMA order per accumulator, same power-of-two row scales, same split): every stored
// value is bit-identical to chain.hip's except behind a LayerNorm core, whose sums now run in wave-reduction order.
#include "papr_common.h"
#include "h3_common.h"
#include "chain.h"
#include <stdlib.h>
#include <type_traits>

namespace {

#ifndef C2_NJ
#define C2_NJ 1                                 // 32-column tiles per multiplying wave: 1 = eight waves per group (64 x 32 each), 2 = four (64 x 64)
#endif
constexpr int NI = 2, NJ = C2_NJ;               // a multiplying wave: 64 rows x 32 NJ columns
constexpr int GW = 8 / NJ;                      // waves per group
constexpr int RB = 64 / GW;                     // rows per wave in the row phases (one block of the planes): 8 or 16
constexpr int C2_THREADS = 2 * GW * 64;
constexpr int C2_ROWS = 64;                     // rows per group tile
constexpr int C2_GROUP_BYTES = 65536;           // A planes of one group: GW blocks of RB rows
constexpr int C2_BLK_BYTES = RB * 1024;         // one block: hi rows (RB x 512 B) | lo rows (RB x 512 B); fp32 overlay of row u: columns 0-127 over hi row u, 128-255 over lo row u
constexpr int C2_LO = RB * 512;
constexpr int RQ = NJ == 1 ? 4 : 8;             // rows a wave carries through the row phase at a time (register budget: 128 or 256 VGPRs per wave)
constexpr bool C2_EARLY_STAGE = NJ == 2;        // the next tile's input rows are requested before the last row phase (32 registers) or after it
constexpr size_t C2_LDS_BYTES = 2 * C2_GROUP_BYTES + 64 + 512;      // planes, the two sync counters, 1 / scale of every A-plane row
constexpr int WD = 4;                           // weight-fragment ring depth (k-steps)

__device__ __forceinline__ float wave64_max(float v) { return wave_max(v); }

__device__ __forceinline__ float scale_from_max(unsigned bits, float& inv) {      // row max -> [2^13, 2^14)
    const int ea = bits ? (int)((bits >> 23) & 0xff) : 127 + 13;
    inv = pow2_from_biased(127 - 13 + (ea - 127));
    return pow2_from_biased(127 + 13 - (ea - 127));
}

// max over the 64 lanes of four registers at once (result valid in lane 63): the four chains interleave, so that the two
// wait states a DPP read needs behind the VALU write of its source are filled by the other rows' instructions
#define C2_DPP4(ctrl)                                                  \
    "v_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\t" \
    "v_max_f32_dpp %2, %2, %2 " ctrl "\n\tv_max_f32_dpp %3, %3, %3 " ctrl "\n\t"
__device__ __forceinline__ void wave_max4(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t" C2_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") C2_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        C2_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf") C2_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
        C2_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf") C2_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ float last_lane(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
// lanes l0 .. l0 + 3 of vec = four scalars (s_nop: the scalars come out of v_readlane, see select_by_masks)
#define put4(vec, l0, s0, s1, s2, s3)                                                                                             \
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %0, %3, %7\n\tv_writelane_b32 %0, %4, %8" \
        : "+v"(vec) : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "i"(l0), "i"((l0) + 1), "i"((l0) + 2), "i"((l0) + 3))

// uniform per-layer flags of the row phases: 0 / 1 = known at compile time (the hot instantiations), 2 = look at run time
template <int STORE, int BITS, int RMAX, int MORE, int NORM, int FULL>
struct RowCfg { static constexpr int store = STORE, bits = BITS, rmax = RMAX, more = MORE, norm = NORM, full = FULL; };

#ifdef PAPR_H3_TRACE
__device__ long long g_chain2_trace[2][512];
#ifdef PAPR_C2_TRACE_ALL       // every wave of the workgroup, 64 stamps each
#define C2_STAMP() do { if (blockIdx.x == 100 && lane == 0 && trace_slot < 64) g_chain2_trace[0][(grp * GW + wn) * (1024 / (2 * GW)) + trace_slot++] = __builtin_readcyclecounter(); } while (0)
#else
#define C2_STAMP() do { if (blockIdx.x == 100 && wn == 0 && lane == 0 && trace_slot < 512) g_chain2_trace[grp][trace_slot++] = __builtin_readcyclecounter(); } while (0)
#endif
#ifdef PAPR_C2_TRACE_FINE
#define C2_STAMP2() do { asm volatile("" ::: "memory"); C2_STAMP(); } while (0)
#else
#define C2_STAMP2() do {} while (0)
#endif
#else
#define C2_STAMP() do {} while (0)
#define C2_STAMP2() do {} while (0)
#endif

template <bool DGRAD>
__global__ __launch_bounds__(C2_THREADS, C2_THREADS / 256) void mlp_chain2_kernel(ChainArgs p, int iters, int generic_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / GW, wn = wave % GW;      // group; column slice while multiplying = row block in the row phases
    char* const planes = smem + grp * C2_GROUP_BYTES;
    unsigned* const sync_cnt = reinterpret_cast<unsigned*>(smem + 2 * C2_GROUP_BYTES) + grp * 8;
    if (tid == 0) { sync_cnt[0] = 0u; sync_cnt[8] = 0u; }
    unsigned sync_epoch = 0;
    const int hh = lane >> 5;
#ifdef PAPR_H3_TRACE
    int trace_slot = 0;
#endif

    // ---- addresses (bytes inside the group's planes)
    // multiplying layout: A fragment of 32-row tile i, k-step ks: row = 32 i + (lane & 31), 16-byte chunk 2 ks + hh
    // (row r of the tile lives in block r / RB at row r % RB; its 16-byte chunks are XOR-ed with r & 15 in the planes, r & 7 in the overlay)
    const int arow = lane & 31, ax = arow & 15;
    const unsigned a_base = (unsigned)((arow / RB) * C2_BLK_BYTES + (arow % RB) * 512 + ((hh ^ (ax & 1)) * 16));
    const unsigned a_xor = (unsigned)((ax & ~1) * 16);
    // dump: accumulator run (i, j, g) = row 32 i + arow, fp32 chunk 8 (NJ wn + j) + 2 g + hh of the row's overlay (64 chunks: the
    // first 32 over the hi row, the rest over the lo row)
    const unsigned d_base = (unsigned)((arow / RB) * C2_BLK_BYTES + (arow % RB) * 512);
    const unsigned d_x = (unsigned)(ax & 7);
    // row layout: lane holds columns 4 lane .. 4 lane + 3 = chunk lane & 31 of half lane >> 5
    const unsigned r_base = (unsigned)((lane >> 5) * C2_LO);
    const unsigned r_chunk = (unsigned)(lane & 31);
    // row layout: this wave owns block wn; lane holds columns 4 lane .. 4 lane + 3 of a row
    char* const blk = planes + wn * C2_BLK_BYTES;

    // 1 / scale of A-plane rows RB wn .. RB wn + RB - 1: written and read by this wave only (all lanes store the same value)
    float* const inv_tab = reinterpret_cast<float*>(smem + 2 * C2_GROUP_BYTES + 64) + grp * 64 + wn * RB;

    // ---- weight fragments: fragment (n-tile t, k-step s) of a layer's planes starts at ((t * ksteps + s) * 64 + lane) * 8 halfs
    half8 wfh[WD][NJ], wfl[WD][NJ];
    auto load_w = [&](int l, int ks, half8 (&qh)[NJ], half8 (&ql)[NJ]) {
        const ChainLayer& L = p.L[l];
        ks = ks < L.ksteps ? ks : L.ksteps - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int t = 32 * (wn * NJ + j) < L.N ? wn * NJ + j : 0;
            const long o = ((long)(t * L.ksteps + ks) * 64 + lane) * 8;
            qh[j] = *reinterpret_cast<const half8*>(L.w_hi + o);
            ql[j] = *reinterpret_cast<const half8*>(L.w_lo + o);
        }
    };

    // ---- barrier of this group's waves only (LDS counter): the other group is in its k-loop and must not be held up
    auto group_sync = [&]() {
        sync_epoch += GW;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(sync_cnt, 1u);
        while (*reinterpret_cast<volatile unsigned*>(sync_cnt) < sync_epoch) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
    };

    // ---- split a row held across the wave (lane: 4 columns) into the A planes of block wn, row u
    auto write_planes = [&](int u, const float4& v, float sc, int kpad) {
        if (4 * lane < kpad) {
            half4 hi, lo;
            split4(v, sc, hi, lo);
            char* dst = blk + u * 512 + (((lane >> 1) ^ ((wn * RB + u) & 15)) * 16) + (lane & 1) * 8;
            *reinterpret_cast<half4*>(dst) = hi;
            *reinterpret_cast<half4*>(dst + C2_LO) = lo;
        }
    };

    // ---- stage the input rows of a tile (coalesced: one row per load instruction), LayerNorm core in front of the run
    // eight rows (ub .. ub + 7 of block wn) of tile m0: issue the loads ...
    auto stage_load = [&](long m0, int ub, float4 (&v)[8]) {
        const int c = 4 * lane;
        asm volatile("" : "+s"(m0));                // (keeps the row addresses out of registers that would live across the k-loop)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            long m = m0 + wn * RB + ub + q;
            m = m < p.M ? m : p.M - 1;              // rows beyond M: the last row again
            const float* rowp = p.A0 + m * p.lda0;  // wave-uniform: scalar base + one lane offset
            v[q] = c < p.K0 ? *reinterpret_cast<const float4*>(rowp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // ... and turn them into A-plane rows: LayerNorm core in front of the run, row maxima, scales, split
    auto stage_rows = [&](long m0, const float4 (&va)[8], const float4 (&vb)[8]) {
        const int c = 4 * lane;
        const int kpad = p.L[0].k1steps * 16;
#pragma unroll 1
        for (int ub = 0; ub < RB; ub += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (RB == 16 && ub) ? vb[q] : va[q];
        if (!DGRAD && p.in_norm_stats != nullptr) {
            // LayerNorm core in front of the run (FeedForward.innorm): the wave holds the whole row
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const long mrow = m0 + wn * RB + ub + q;
                const int wdt = p.in_norm_width;
                const bool i0 = c < wdt, i1 = c + 1 < wdt, i2 = c + 2 < wdt, i3 = c + 3 < wdt;
                const float mean = wave_sum(((i0 ? v[q].x : 0.f) + (i1 ? v[q].y : 0.f)) + ((i2 ? v[q].z : 0.f) + (i3 ? v[q].w : 0.f))) / (float)wdt;
                float4 dl = make_float4(i0 ? v[q].x - mean : 0.f, i1 ? v[q].y - mean : 0.f, i2 ? v[q].z - mean : 0.f, i3 ? v[q].w - mean : 0.f);
                const float sigma = sqrtf(wave_sum((dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w)) / (float)(wdt - 1));
                const float rinv = 1.0f / (sigma + p.in_norm_eps);
                v[q] = make_float4(dl.x * rinv, dl.y * rinv, dl.z * rinv, dl.w * rinv);
                if (mrow < p.M) {
                    if (p.in_norm_writeback && c < p.K0) *reinterpret_cast<float4*>(p.A0 + mrow * p.lda0 + c) = v[q];
                    if (lane == 0) { p.in_norm_stats[mrow * 2] = rinv; p.in_norm_stats[mrow * 2 + 1] = sigma; }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            float mx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) mx[q] = fmaxf(fmaxf(fabsf(v[h + q].x), fabsf(v[h + q].y)), fmaxf(fabsf(v[h + q].z), fabsf(v[h + q].w)));
            wave_max4(mx[0], mx[1], mx[2], mx[3]);
            float smx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) smx[q] = last_lane(mx[q]);
            float mx4 = 0.f;
            put4(mx4, 0, smx[0], smx[1], smx[2], smx[3]);
            if (p.rowmax0 && lane < 4 && m0 + wn * RB + ub + h + lane < p.M) p.rowmax0[m0 + wn * RB + ub + h + lane] = mx4;
            float inv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sc = scale_from_max(__float_as_uint(smx[q]), inv[q]);
                write_planes(ub + h + q, v[h + q], sc, kpad);
            }
            *reinterpret_cast<float4*>(inv_tab + ub + h) = make_float4(inv[0], inv[1], inv[2], inv[3]);
        }
        }
    };

    long tile = 2L * blockIdx.x + grp;
    const long tstride = 2L * gridDim.x;
    {
        float4 sa[8], sb[8];
        stage_load(tile * C2_ROWS, 0, sa);
        if constexpr (RB == 16) stage_load(tile * C2_ROWS, 8, sb);
        stage_rows(tile * C2_ROWS, sa, sb);
    }
#pragma unroll
    for (int u = 0; u < WD - 1; ++u) load_w(0, u, wfh[u], wfl[u]);
    lds_barrier();                                  // planes ready (and the sync counters zeroed)
#ifdef C2_G1PRIO
    if (grp == 1) __builtin_amdgcn_s_setprio(C2_G1PRIO);
#endif
#ifndef C2_FREE
    if (grp == 1) lds_barrier();                    // group 1 runs half a period behind group 0
#else
    if (grp == 1) { for (int z = 0; z < 32; ++z) __builtin_amdgcn_s_sleep(127); }      // (experiment: free-running groups, group 1 starts ~4k cycles late)
#endif

#pragma unroll 1
    for (int it = 0; it < iters; ++it, tile += tstride) {
        const long m0 = tile * C2_ROWS;
#pragma unroll 1
        for (int l = 0; l < p.n_layers; ++l) {
            const ChainLayer& L = p.L[l];
            const int N = L.N, ksteps = L.ksteps;
            bool live[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) live[j] = 32 * (wn * NJ + j) < N;
            f32x16 acc[NI][NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

            // =========================== multiplying role: k-loop ===========================
            C2_STAMP();
            {
                const int kb = 0, ke = ksteps;
                auto a_addr = [&](int ks) { return a_base + (((unsigned)ks * 32u) ^ a_xor); };
                auto load_a = [&](int ks, half8 (&ah)[NI], half8 (&al)[NI]) {
                    ks = ks < ke ? ks : ke - 1;
                    const unsigned o = a_addr(ks);
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        ah[i] = *reinterpret_cast<const half8*>(planes + i * 32768 + o);
                        al[i] = *reinterpret_cast<const half8*>(planes + i * 32768 + o + C2_LO);
                    }
                };
                auto mma_live = [&](const half8 (&qh)[NJ], const half8 (&ql)[NJ], const half8 (&ah)[NI], const half8 (&al)[NI]) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!live[j]) continue;
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[j], al[i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[j], ah[i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[j], ah[i], acc[i][j], 0, 0, 0);
                        }
                    }
                };
                auto k_loop = [&](auto mma) {                   // waves with a dead column tile (narrow layers): simple form
                    half8 a0h[NI], a0l[NI], a1h[NI], a1l[NI];
                    load_a(kb, a0h, a0l);
#pragma unroll 1
                    for (int ks = kb; ks < ke; ks += 4) {
                        load_w(l, ks + 3, wfh[3], wfl[3]);
                        load_a(ks + 1, a1h, a1l);
                        __builtin_amdgcn_sched_barrier(0);
                        mma(wfh[0], wfl[0], a0h, a0l);
                        __builtin_amdgcn_sched_barrier(0);
                        load_w(l, ks + 4, wfh[0], wfl[0]);
                        load_a(ks + 2, a0h, a0l);
                        __builtin_amdgcn_sched_barrier(0);
                        mma(wfh[1], wfl[1], a1h, a1l);
                        __builtin_amdgcn_sched_barrier(0);
                        load_w(l, ks + 5, wfh[1], wfl[1]);
                        if (ks + 2 < ke) {
                            load_a(ks + 3, a1h, a1l);
                            __builtin_amdgcn_sched_barrier(0);
                            mma(wfh[2], wfl[2], a0h, a0l);
                            __builtin_amdgcn_sched_barrier(0);
                            load_w(l, ks + 6, wfh[2], wfl[2]);
                            load_a(ks + 4, a0h, a0l);
                            __builtin_amdgcn_sched_barrier(0);
                            mma(wfh[3], wfl[3], a1h, a1l);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                // The hot form: every request (weight fragment from L2, A fragment from LDS) sits alone behind an MFMA and
                // sched_barrier pins that order (chain.hip's k-loop, with the XOR-swizzled A addresses).
                const char* wb_h[NJ];
                const char* wb_l[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    wb_h[j] = reinterpret_cast<const char*>(L.w_hi) + (size_t)((wn * NJ + j) * ksteps) * 1024;
                    wb_l[j] = reinterpret_cast<const char*>(L.w_lo) + (size_t)((wn * NJ + j) * ksteps) * 1024;
                }
                const unsigned lane16 = (unsigned)lane * 16u;
#define C2_SB __builtin_amdgcn_sched_barrier(0)
#define C2_MFMA(i, j, w, x) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[j], x[i], acc[i][j], 0, 0, 0)
                auto step = [&](const half8 (&uh)[NJ], const half8 (&ul)[NJ], const half8 (&xh)[NI], const half8 (&xl)[NI],
                                int kw, half8 (&nh)[NJ], half8 (&nl)[NJ], int ka, half8 (&yh)[NI], half8 (&yl)[NI]) {
                    kw = kw < ksteps ? kw : ksteps - 1;
                    ka = ka < ke ? ka : ke - 1;
                    const unsigned ao = a_addr(ka);
                    auto request = [&](int q) {
                        if (q < 2 * NJ) {
                            const int j = q >> 1;
                            if (q & 1) nl[j] = *reinterpret_cast<const half8*>(wb_l[j] + (size_t)kw * 1024 + lane16);
                            else nh[j] = *reinterpret_cast<const half8*>(wb_h[j] + (size_t)kw * 1024 + lane16);
                        } else if (q < 2 * NJ + 2 * NI) {
                            const int a = q - 2 * NJ, i = a >> 1;
                            if (a & 1) yl[i] = *reinterpret_cast<const half8*>(planes + i * 32768 + ao + C2_LO);
                            else yh[i] = *reinterpret_cast<const half8*>(planes + i * 32768 + ao);
                        }
                    };
                    int q = 0;
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                if (term == 0) C2_MFMA(i, j, uh, xl);
                                else if (term == 1) C2_MFMA(i, j, ul, xh);
                                else C2_MFMA(i, j, uh, xh);
                                C2_SB;
                                request(q++);
                                C2_SB;
                            }
                };
                auto k_loop_full = [&]() {
                    half8 a0h[NI], a0l[NI], a1h[NI], a1l[NI];
                    load_a(kb, a0h, a0l);
#pragma unroll 1
                    for (int ks = kb; ks < ke; ks += 4) {
                        step(wfh[0], wfl[0], a0h, a0l, ks + 3, wfh[3], wfl[3], ks + 1, a1h, a1l);
                        step(wfh[1], wfl[1], a1h, a1l, ks + 4, wfh[0], wfl[0], ks + 2, a0h, a0l);
                        if (ks + 2 < ke) {
                            step(wfh[2], wfl[2], a0h, a0l, ks + 5, wfh[1], wfl[1], ks + 3, a1h, a1l);
                            step(wfh[3], wfl[3], a1h, a1l, ks + 6, wfh[2], wfl[2], ks + 4, a0h, a0l);
                        }
                    }
                };
#undef C2_MFMA
#undef C2_SB
#ifdef C2_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#ifndef C2_NOK           // (timing experiment: the row phases alone)
                if (live[NJ - 1]) k_loop_full();
                else if (live[0]) k_loop(mma_live);
#endif
#ifdef C2_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
            C2_STAMP();
#ifndef C2_FREE
            lds_barrier();                          // (a) this group's A reads are complete: its planes are dead until the split
#else
            group_sync();
#endif
            C2_STAMP();

            // =========================== the other role: row phases ===========================
#ifdef C2_PPRIO
            __builtin_amdgcn_s_setprio(C2_PPRIO);
#endif
            const bool more = l + 1 < p.n_layers;
            const bool next_tile = !more && it + 1 < iters;
            float4 sa[8], sb[8];                    // input rows of the next tile on their way (requested before the row phase)
            {   // first weight fragments of what this wave multiplies next (they arrive while it is busy here)
                const int ln = more ? l + 1 : 0;
#pragma unroll
                for (int u = 0; u < WD - 1; ++u) load_w(ln, u, wfh[u], wfl[u]);
            }
            // ---- dump the accumulators (raw, still scaled) into the fp32 overlay
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (!live[j]) continue;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c64 = 8 * (NJ * wn + j) + 2 * g;          // (+ hh: the chunk's lowest bit)
                        const unsigned c32 = (unsigned)(c64 & 31) + (unsigned)hh;
                        *reinterpret_cast<float4*>(planes + i * 32768 + d_base + (c64 >> 5) * C2_LO + ((c32 ^ d_x) * 16)) =
                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    }
                }
            C2_STAMP();
            group_sync();
            C2_STAMP();
            // ---- rows 16 wn .. 16 wn + 15, a whole row across the wave, four rows at a time.  Straight-line code matters: with the
            // per-layer flags tested row by row the compiler cut this phase into ~25 basic blocks per row and re-read kernel
            // arguments through the scalar cache inside them (1,250 cycles per row); the hot flag combinations are
            // therefore instantiated with the flags as constants, everything else takes the generic instantiation.
            {
                const int c = 4 * lane;
                const float slope = L.act == PAPR_ACT_RELU ? 0.f : (L.act == PAPR_ACT_LEAKY_RELU ? 0.2f : 1.f);
                const bool rt_store = L.C != nullptr, rt_bits = L.sign_bits != nullptr, rt_rmax = L.rowmax != nullptr;
                const bool rt_norm = !DGRAD && !more && p.norm_stats != nullptr;
                const bool rt_full = N == 256 && m0 + C2_ROWS <= p.M;
                const bool mask_rows = DGRAD && !rt_bits && L.mask != nullptr;
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!DGRAD && L.bias && c < N) b4 = *reinterpret_cast<const float4*>(L.bias + c);
                const int kpad_next = more ? p.L[l + 1].k1steps * 16 : 0;
                float* const crow = L.C + (m0 + wn * RB) * L.ldc;                       // row 16 wn (wave-uniform: scalar base + lane offset)
                const long ldc = L.ldc;
                // sign words: lane l keeps ITS 4 x 16 bits of the wave's 16 rows (two words: rows 0-7, rows 8-15; first value in
                // the top bit), 512 contiguous bytes per wave and layer -- written by the forward run, read back by the
                // data-gradient run, whose lanes hold the same columns of the same rows
                unsigned* const sgn = L.sign_bits + ((m0 + wn * RB) / RB) * (RB * 8) + (RB / 8) * lane;
                unsigned sw[2] = {0u, 0u};
                const bool blk_in = m0 + wn * RB < p.M;      // (blocks beyond M have no words: the area is chain_sign_rows(M) rows long)
                if (DGRAD && rt_bits && blk_in) {
                    if constexpr (RB == 16) { const uint2 w2 = *reinterpret_cast<const uint2*>(sgn); sw[0] = w2.x; sw[1] = w2.y; }
                    else sw[0] = *sgn;
                }
                // the loads above are waited for HERE: a wait inside the batch loop would also wait for the row stores of the
                // batch before it (loads and stores share vmcnt); the next tile's first input rows are requested behind them
                if (C2_EARLY_STAGE && next_tile) stage_load((tile + tstride) * C2_ROWS, 0, sa);
                asm volatile("" : "+v"(b4.x), "+v"(b4.y), "+v"(b4.z), "+v"(b4.w), "+v"(sw[0]), "+v"(sw[1]));
                auto rows = [&](auto cfg) {
                    using Cfg = decltype(cfg);
                    const bool f_store = Cfg::store == 2 ? rt_store : Cfg::store == 1;
                    const bool f_bits = Cfg::bits == 2 ? rt_bits : Cfg::bits == 1;
                    const bool f_rmax = Cfg::rmax == 2 ? rt_rmax : Cfg::rmax == 1;
                    const bool f_more = Cfg::more == 2 ? more : Cfg::more == 1;
                    const bool f_norm = Cfg::norm == 2 ? rt_norm : Cfg::norm == 1;
                    const bool f_full = Cfg::full == 2 ? rt_full : Cfg::full == 1;
                    const bool col_ok = f_full || c < N;
                    unsigned sword = DGRAD ? sw[0] : 0u;            // (one word per eight rows)
#pragma unroll 1
                    for (int ub = 0; ub < RB; ub += RQ) {           // RQ rows in flight: independent chains of four for the scheduler
                        float4 r[RQ];
                        if (RB == 16 && ub == 8) { if (DGRAD) sword = sw[1]; else { sw[0] = sword; sword = 0u; } }
#pragma unroll
                        for (int q = 0; q < RQ; ++q)
                            r[q] = *reinterpret_cast<const float4*>(blk + r_base + (ub + q) * 512 + ((r_chunk ^ (unsigned)((ub + q) & 7)) * 16));
                        const float4 inv4a = *reinterpret_cast<const float4*>(inv_tab + ub);
                        const float4 inv4b = RQ == 8 ? *reinterpret_cast<const float4*>(inv_tab + ub + 4) : inv4a;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the rows are in registers: their bytes may be overwritten
                        C2_STAMP2();
#pragma unroll
                        for (int q = 0; q < RQ; ++q) {
                            const int u = ub + q;
                            const float4 iv = q < 4 ? inv4a : inv4b;
                            const float inv = (q & 3) == 0 ? iv.x : (q & 3) == 1 ? iv.y : (q & 3) == 2 ? iv.z : iv.w;
                            if (DGRAD) {
                                r[q] = make_float4(r[q].x * inv, r[q].y * inv, r[q].z * inv, r[q].w * inv);
                                if (f_bits) {
                                    r[q].x = (int)sword < 0 ? r[q].x : r[q].x * slope; r[q].y = (int)(sword << 1) < 0 ? r[q].y : r[q].y * slope;
                                    r[q].z = (int)(sword << 2) < 0 ? r[q].z : r[q].z * slope; r[q].w = (int)(sword << 3) < 0 ? r[q].w : r[q].w * slope;
                                    sword <<= 4;
                                } else if (mask_rows) {
                                    long row = m0 + wn * RB + u;
                                    row = row < p.M ? row : p.M - 1;
                                    const float4 a4 = col_ok ? *reinterpret_cast<const float4*>(L.mask + row * L.ld_mask + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                                    r[q].x *= a4.x > 0.f ? 1.f : slope; r[q].y *= a4.y > 0.f ? 1.f : slope;
                                    r[q].z *= a4.z > 0.f ? 1.f : slope; r[q].w *= a4.w > 0.f ? 1.f : slope;
                                }
                            } else {
                                // acc * inv is exact (a power of two): fma(acc, inv, bias) = the separate multiply and add, bit for bit;
                                // activation as max(y, slope y + 0): slope 0 -> ReLU (+0 for negative y), 0.2 -> LeakyReLU, 1 -> none
                                r[q] = make_float4(__builtin_fmaf(r[q].x, inv, b4.x), __builtin_fmaf(r[q].y, inv, b4.y),
                                                   __builtin_fmaf(r[q].z, inv, b4.z), __builtin_fmaf(r[q].w, inv, b4.w));
                                r[q] = make_float4(fmaxf(r[q].x, __builtin_fmaf(r[q].x, slope, 0.f)), fmaxf(r[q].y, __builtin_fmaf(r[q].y, slope, 0.f)),
                                                   fmaxf(r[q].z, __builtin_fmaf(r[q].z, slope, 0.f)), fmaxf(r[q].w, __builtin_fmaf(r[q].w, slope, 0.f)));
                            }
                            if (!f_full && !col_ok) r[q] = make_float4(0.f, 0.f, 0.f, 0.f);       // columns beyond N: nothing was dumped there
                            if (f_norm) {
                                // LayerNorm core behind the run (FeedForward.outnorm, act = none): two-pass mean / unbiased std over the
                                // row's N columns, wave reductions in a fixed order
                                const long row = m0 + wn * RB + u;
                                const float mean = wave_sum((r[q].x + r[q].y) + (r[q].z + r[q].w)) / (float)N;
                                float4 dl = col_ok ? make_float4(r[q].x - mean, r[q].y - mean, r[q].z - mean, r[q].w - mean) : make_float4(0.f, 0.f, 0.f, 0.f);
                                const float sigma = sqrtf(wave_sum((dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w)) / (float)(N - 1));
                                const float rinv = 1.0f / (sigma + p.norm_eps);
                                r[q] = make_float4(dl.x * rinv, dl.y * rinv, dl.z * rinv, dl.w * rinv);
                                if (lane == 0 && row < p.M) { p.norm_stats[row * 2] = rinv; p.norm_stats[row * 2 + 1] = sigma; }
                            }
                            if (f_store && col_ok && (f_full || m0 + wn * RB + u < p.M)) *reinterpret_cast<float4*>(crow + u * ldc + c) = r[q];
                            if (!DGRAD && f_bits) {
                                sword = (sword << 1) | (r[q].x > 0.f ? 1u : 0u); sword = (sword << 1) | (r[q].y > 0.f ? 1u : 0u);
                                sword = (sword << 1) | (r[q].z > 0.f ? 1u : 0u); sword = (sword << 1) | (r[q].w > 0.f ? 1u : 0u);
                            }
                        }
                        C2_STAMP2();
                        if (f_more || f_rmax) {
#pragma unroll
                            for (int h = 0; h < RQ; h += 4) {
                                float mx[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) mx[q] = fmaxf(fmaxf(fabsf(r[h + q].x), fabsf(r[h + q].y)), fmaxf(fabsf(r[h + q].z), fabsf(r[h + q].w)));
                                wave_max4(mx[0], mx[1], mx[2], mx[3]);
                                float smx[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) smx[q] = last_lane(mx[q]);
                                if (f_rmax) {
                                    float mx4 = 0.f;
                                    put4(mx4, 0, smx[0], smx[1], smx[2], smx[3]);
                                    if (lane < 4 && (f_full || m0 + wn * RB + ub + h + lane < p.M)) L.rowmax[m0 + wn * RB + ub + h + lane] = mx4;
                                }
                                if (f_more) {
                                    float inv_n[4];
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const float sc = scale_from_max(__float_as_uint(smx[q]), inv_n[q]);
                                        write_planes(ub + h + q, r[h + q], sc, kpad_next);
                                    }
                                    *reinterpret_cast<float4*>(inv_tab + ub + h) = make_float4(inv_n[0], inv_n[1], inv_n[2], inv_n[3]);
                                }
                            }
                        }
                        C2_STAMP2();
                    }
                    if (!DGRAD && f_bits && blk_in) {
                        if constexpr (RB == 16) *reinterpret_cast<uint2*>(sgn) = make_uint2(sw[0], sword);
                        else *sgn = sword;
                    }
                };
                // hot combinations (everything 256 wide, tile inside M): training middle layer / inference middle layer / data-gradient
#ifdef C2_NOROWS         // (timing experiment: the k-loops alone)
                if (p.M < 0)
#endif
                if (!generic_only && rt_full && more && !rt_norm && !mask_rows && rt_store && rt_bits && rt_rmax) rows(RowCfg<1, 1, 1, 1, 0, 1>());
                else if (!generic_only && !DGRAD && rt_full && more && !rt_norm && !rt_store && !rt_bits && !rt_rmax) rows(RowCfg<0, 0, 0, 1, 0, 1>());
                else rows(RowCfg<2, 2, 2, 2, 2, 2>());
            }
            C2_STAMP();
            if (next_tile) {
                if constexpr (!C2_EARLY_STAGE) stage_load((tile + tstride) * C2_ROWS, 0, sa);
                if constexpr (RB == 16) stage_load((tile + tstride) * C2_ROWS, 8, sb);
                stage_rows((tile + tstride) * C2_ROWS, sa, sb);
            }
            C2_STAMP();
#ifdef C2_PPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#ifndef C2_FREE
            lds_barrier();                          // (c) planes of the next layer (or tile) ready
#else
            group_sync();
#endif
        }
    }
#ifndef C2_FREE
    if (grp == 0) lds_barrier();                    // group 0 ends half a period early
#endif
}

}  // namespace

#ifdef PAPR_H3_TRACE
extern "C" int papr_chain2_trace_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain2_trace), sizeof(long long) * 1024) == hipSuccess ? 0 : 1; }
#endif


size_t papr_chain2_lds_bytes() { return C2_LDS_BYTES; }

int papr_launch_chain2(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s) {
    PAPR_REQUIRE(a.n_layers >= 1 && a.n_layers <= CHAIN_MAX_LAYERS, "mlp_chain2: %d layers", a.n_layers);
    PAPR_REQUIRE(a.K0 % 4 == 0 && a.lda0 % 4 == 0 && a.K0 <= 256, "mlp_chain2: input width %d", a.K0);
    for (int l = 0; l < a.n_layers; ++l) PAPR_REQUIRE(a.L[l].k1steps == a.L[l].ksteps, "mlp_chain2: skip layers run on mlp_chain_kernel");
    if (a.M <= 0) return 0;
    const long tiles = (a.M + C2_ROWS - 1) / C2_ROWS;
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C2_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C2_LDS_BYTES);
        attr_set = true;
    }
    const long pairs = (tiles + 1) / 2;             // a workgroup carries two tiles at a time
    const unsigned grid = (unsigned)(pairs < n_cu ? pairs : n_cu);
    const int iters = (int)((tiles + 2L * grid - 1) / (2L * grid));
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin2(dgrad ? 10 : 9, a.M, a.n_layers, a.K0, bytes, flops, s);
    static const int generic_only = getenv("PAPR_C2_GENERIC") ? atoi(getenv("PAPR_C2_GENERIC")) : 0;      // (test switch: the hot instantiations off)
    if (dgrad) mlp_chain2_kernel<true><<<dim3(grid), dim3(C2_THREADS), C2_LDS_BYTES, s>>>(a, iters, generic_only == 1 || generic_only == 3);
    else mlp_chain2_kernel<false><<<dim3(grid), dim3(C2_THREADS), C2_LDS_BYTES, s>>>(a, iters, generic_only == 1 || generic_only == 2);
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("mlp_chain2");
    return 0;
}
