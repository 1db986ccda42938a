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

constexpr int C2_THREADS = 512;
constexpr int C2_ROWS = 64;                     // rows per group tile
constexpr int C2_GROUP_BYTES = 65536;           // A planes of one group: 4 blocks of 16 rows
constexpr int C2_BLK_BYTES = 16384;             // one block: hi rows (16 x 512 B) | lo rows (16 x 512 B); fp32 overlay of row u: columns 0-127 over hi row u, 128-255 over lo row u
constexpr int C2_LO = 8192;
constexpr size_t C2_LDS_BYTES = 2 * C2_GROUP_BYTES + 64 + 512;      // planes, the two sync counters, 1 / scale of every A-plane row
constexpr int NI = 2, NJ = 2;                   // a multiplying wave: 64 rows x 64 columns = 2 x 2 MFMA tiles
constexpr int WD = 4;                           // weight-fragment ring depth (k-steps)

__device__ __forceinline__ float wave64_max(float v) { return wave_max(v); }

__device__ __forceinline__ float scale_from_max(unsigned bits, float& inv) {      // row max -> [2^13, 2^14)
    const int ea = bits ? (int)((bits >> 23) & 0xff) : 127 + 13;
    inv = pow2_from_biased(127 - 13 + (ea - 127));
    return pow2_from_biased(127 + 13 - (ea - 127));
}

// r.c = mask_c bit of this lane ? r.c : r.c * slope, with the four 64-bit lane masks in scalar registers (one instruction
// per value).  The masks come out of v_readlane: on gfx950 a VALU instruction that reads an SGPR written by a VALU
// instruction needs two wait states in between (LLVM's VALUWriteSGPRVALURead), which hipcc does not insert in front of
// an asm statement -- without the s_nop the selects read the PREVIOUS contents of the pair's high register.
__device__ __forceinline__ void select_by_masks(float4& r, float slope, unsigned long long m0, unsigned long long m1, unsigned long long m2,
                                                unsigned long long m3) {
    const float4 f = make_float4(r.x * slope, r.y * slope, r.z * slope, r.w * slope);
    float4 o;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %4, %8, %12\n\tv_cndmask_b32_e64 %1, %5, %9, %13\n\t"
        "v_cndmask_b32_e64 %2, %6, %10, %14\n\tv_cndmask_b32_e64 %3, %7, %11, %15"
        : "=&v"(o.x), "=&v"(o.y), "=&v"(o.z), "=&v"(o.w)
        : "v"(f.x), "v"(f.y), "v"(f.z), "v"(f.w), "v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
    r = o;
}

// lanes 4 q .. 4 q + 3 of (lo, hi) = the four 64-bit masks (scalars into single lanes of two vector registers); s_nop: the
// masks come out of v_cmp (VALU writes SGPR -> VALU reads SGPR: two wait states, see select_by_masks)
#define put_masks(lo, hi, q, m0, m1, m2, m3)                                                                                  \
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %10\n\tv_writelane_b32 %1, %3, %10\n\tv_writelane_b32 %0, %4, %11\n\tv_writelane_b32 %1, %5, %11\n\t" \
        "v_writelane_b32 %0, %6, %12\n\tv_writelane_b32 %1, %7, %12\n\tv_writelane_b32 %0, %8, %13\n\tv_writelane_b32 %1, %9, %13"                  \
        : "+v"(lo), "+v"(hi)                                                                                                   \
        : "s"((unsigned)(m0)), "s"((unsigned)((m0) >> 32)), "s"((unsigned)(m1)), "s"((unsigned)((m1) >> 32)), "s"((unsigned)(m2)),        \
          "s"((unsigned)((m2) >> 32)), "s"((unsigned)(m3)), "s"((unsigned)((m3) >> 32)), "i"(4 * (q)), "i"(4 * (q) + 1), "i"(4 * (q) + 2), "i"(4 * (q) + 3))

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

#if defined(C2_DBG) && (C2_DBG == 8 || C2_DBG == 13)
__device__ unsigned g_c2_dbg[16];
#endif
#ifdef PAPR_H3_TRACE
__device__ long long g_chain2_trace[2][512];
#define C2_STAMP() do { if (blockIdx.x == 100 && wn == 0 && lane == 0 && trace_slot < 512) g_chain2_trace[grp][trace_slot++] = __builtin_readcyclecounter(); } while (0)
#else
#define C2_STAMP() do {} while (0)
#endif

template <bool DGRAD>
__global__ __launch_bounds__(C2_THREADS, 2) void mlp_chain2_kernel(ChainArgs p, int iters, int generic_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;       // group; column quarter while multiplying = 16-row block in the row phases
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
    const int arow = lane & 31, ax = arow & 15;
    const unsigned a_base = (unsigned)((arow >> 4) * C2_BLK_BYTES + ax * 512 + ((hh ^ (ax & 1)) * 16));
    const unsigned a_xor = (unsigned)((ax & ~1) * 16);
    // dump: accumulator run (i, j, g) = row 32 i + arow, fp32 chunk 16 wn + 8 j + 2 g + hh of the row's overlay: half wn >> 1,
    // chunk 16 (wn & 1) + 8 j + 2 g + hh inside the half
    const unsigned d_base = (unsigned)((arow >> 4) * C2_BLK_BYTES + (wn >> 1) * C2_LO + ax * 512);
    const unsigned d_x = (unsigned)(ax & 7);
    // row layout: lane holds columns 4 lane .. 4 lane + 3 = chunk lane & 31 of half lane >> 5
    const unsigned r_base = (unsigned)((lane >> 5) * C2_LO);
    const unsigned r_chunk = (unsigned)(lane & 31);
    // row layout: this wave owns block wn; lane holds columns 4 lane .. 4 lane + 3 of a row
    char* const blk = planes + wn * C2_BLK_BYTES;

    // 1 / scale of A-plane rows 16 wn .. 16 wn + 15: written and read by this wave only (all lanes store the same value)
    float* const inv_tab = reinterpret_cast<float*>(smem + 2 * C2_GROUP_BYTES + 64) + grp * 64 + wn * 16;

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

    // ---- 4-wave barrier of this group only (LDS counter): the other group is in its k-loop and must not be held up
    auto group_sync = [&]() {
        sync_epoch += 4;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if defined(C2_DBG) && C2_DBG == 5
        typedef __attribute__((address_space(3))) unsigned lds_u32;
        lds_u32* cnt3 = (lds_u32*)sync_cnt;
        if (lane == 0) __hip_atomic_fetch_add(cnt3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(cnt3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < sync_epoch) __builtin_amdgcn_s_sleep(2);
#else
        if (lane == 0) atomicAdd(sync_cnt, 1u);
        while (*reinterpret_cast<volatile unsigned*>(sync_cnt) < sync_epoch) __builtin_amdgcn_s_sleep(2);
#endif
        asm volatile("" ::: "memory");
    };

    // ---- split a row held across the wave (lane: 4 columns) into the A planes of block wn, row u
    auto write_planes = [&](int u, const float4& v, float sc, int kpad) {
        if (4 * lane < kpad) {
            half4 hi, lo;
            split4(v, sc, hi, lo);
            char* dst = blk + u * 512 + (((lane >> 1) ^ u) * 16) + (lane & 1) * 8;
            *reinterpret_cast<half4*>(dst) = hi;
#if defined(C2_DBG) && C2_DBG == 9
            asm volatile("" ::: "memory");          // (debug: two ds_write_b64 instead of one ds_write2st64_b64)
#endif
            *reinterpret_cast<half4*>(dst + C2_LO) = lo;
        }
    };

    // ---- stage the input rows of a tile (coalesced: one row per load instruction), LayerNorm core in front of the run
    auto stage = [&](long m0) {
        const int c = 4 * lane;
        const int kpad = p.L[0].k1steps * 16;
        const bool in_norm = !DGRAD && p.in_norm_stats != nullptr;
        const float* src = p.A0 + (m0 + wn * 16) * p.lda0 + c;
#pragma unroll 1
        for (int ub = 0; ub < 16; ub += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                long dm = ub + q;
                dm = m0 + wn * 16 + dm < p.M ? dm : p.M - 1 - (m0 + wn * 16);      // rows beyond M: the last row again
                v[q] = c < p.K0 ? *reinterpret_cast<const float4*>(src + dm * p.lda0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (in_norm) {
                // LayerNorm core in front of the run (FeedForward.innorm): the wave holds the whole row
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const long mrow = m0 + wn * 16 + ub + q;
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
                if (p.rowmax0 && lane < 4 && m0 + wn * 16 + ub + h + lane < p.M) p.rowmax0[m0 + wn * 16 + ub + h + lane] = mx4;
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
    stage(tile * C2_ROWS);
#pragma unroll
    for (int u = 0; u < WD - 1; ++u) load_w(0, u, wfh[u], wfl[u]);
    lds_barrier();                                  // planes ready (and the sync counters zeroed)
    if (grp == 1) lds_barrier();                    // group 1 runs half a period behind group 0

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
#if defined(C2_DBG) && C2_DBG == 13
            unsigned chk[16];                   // debug: fold of every plane row of block wn before the k-loop ...
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint4 t = *reinterpret_cast<const uint4*>(blk + (lane >> 5) * C2_LO + u * 512 + (lane & 31) * 16);
                chk[u] = t.x ^ t.y ^ t.z ^ t.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            {
                const int kb = 0, ke = ksteps;
                auto a_addr = [&](int ks) { return a_base + (((unsigned)ks * 32u) ^ a_xor); };
                auto load_a = [&](int ks, half8 (&ah)[NI], half8 (&al)[NI]) {
                    ks = ks < ke ? ks : ke - 1;
                    const unsigned o = a_addr(ks);
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        ah[i] = *reinterpret_cast<const half8*>(planes + i * 2 * C2_BLK_BYTES + o);
                        al[i] = *reinterpret_cast<const half8*>(planes + i * 2 * C2_BLK_BYTES + o + C2_LO);
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
                            if (a & 1) yl[i] = *reinterpret_cast<const half8*>(planes + i * 2 * C2_BLK_BYTES + ao + C2_LO);
                            else yh[i] = *reinterpret_cast<const half8*>(planes + i * 2 * C2_BLK_BYTES + ao);
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
                if (live[NJ - 1]) k_loop_full();
                else if (live[0]) k_loop(mma_live);
#ifdef C2_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
            C2_STAMP();
#if defined(C2_DBG) && C2_DBG == 13
#pragma unroll
            for (int u = 0; u < 16; ++u) {      // ... and after it: nobody may have written this group's planes in between
                const uint4 t = *reinterpret_cast<const uint4*>(blk + (lane >> 5) * C2_LO + u * 512 + (lane & 31) * 16);
                if ((t.x ^ t.y ^ t.z ^ t.w) != chk[u]) atomicAdd(&g_c2_dbg[u], 1u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            lds_barrier();                          // (a) this group's A reads are complete: its planes are dead until the split
#if defined(C2_DBG) && C2_DBG == 3
            __builtin_amdgcn_s_sleep(127);
#endif
#if defined(C2_DBG) && C2_DBG == 4
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
            C2_STAMP();

            // =========================== the other role: row phases ===========================
            const bool more = l + 1 < p.n_layers;
            const bool next_tile = !more && it + 1 < iters;
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
                        const unsigned c32 = (unsigned)((wn & 1) * 16 + j * 8 + 2 * g) + (unsigned)hh;
                        *reinterpret_cast<float4*>(planes + i * 2 * C2_BLK_BYTES + d_base + ((c32 ^ d_x) * 16)) =
                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    }
                }
            C2_STAMP();
            group_sync();
#if defined(C2_DBG) && C2_DBG == 1
            __builtin_amdgcn_s_sleep(127);
#endif
#if defined(C2_DBG) && C2_DBG == 6
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_sleep(8);
#endif
#if defined(C2_DBG) && C2_DBG == 7
            if (wn == 0) __builtin_amdgcn_s_sleep(127);
#endif
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
                float* const crow = L.C + (m0 + wn * 16) * L.ldc + c;                   // this lane's 16 bytes of row 16 wn
                const long ldc = L.ldc;
                unsigned* const sgn = L.sign_bits + (m0 + wn * 16) * 8;              // the wave's 512 bytes of sign words
                // data-gradient: lane 4 u + cc holds the 64-bit mask of row u, column phase cc
                unsigned sg_lo = 0u, sg_hi = 0u;
                if (DGRAD && rt_bits) {
                    long srow = m0 + wn * 16 + (lane >> 2);
                    srow = srow < p.M ? srow : p.M - 1;
                    const uint2 w2 = *reinterpret_cast<const uint2*>(L.sign_bits + (srow * 8 + 2 * (lane & 3)));
                    sg_lo = w2.x; sg_hi = w2.y;
                }
                // the loads above are waited for HERE: a wait inside the batch loop would also wait for the row stores of the
                // batch before it (loads and stores share vmcnt)
                asm volatile("" : "+v"(b4.x), "+v"(b4.y), "+v"(b4.z), "+v"(b4.w), "+v"(sg_lo), "+v"(sg_hi));
                auto rows = [&](auto cfg) {
                    using Cfg = decltype(cfg);
                    const bool f_store = Cfg::store == 2 ? rt_store : Cfg::store == 1;
                    const bool f_bits = Cfg::bits == 2 ? rt_bits : Cfg::bits == 1;
                    const bool f_rmax = Cfg::rmax == 2 ? rt_rmax : Cfg::rmax == 1;
                    const bool f_more = Cfg::more == 2 ? more : Cfg::more == 1;
                    const bool f_norm = Cfg::norm == 2 ? rt_norm : Cfg::norm == 1;
                    const bool f_full = Cfg::full == 2 ? rt_full : Cfg::full == 1;
                    const bool col_ok = f_full || c < N;
#pragma unroll 1
                    for (int ub = 0; ub < 16; ub += 4) {
                        float4 r[4];
                        unsigned bt_lo = 0u, bt_hi = 0u;        // forward: lane 4 q + cc collects the sign mask of row ub + q, column phase cc
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            r[q] = *reinterpret_cast<const float4*>(blk + r_base + (ub + q) * 512 + ((r_chunk ^ (unsigned)((ub + q) & 7)) * 16));
                        const float4 inv4 = *reinterpret_cast<const float4*>(inv_tab + ub);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the four rows are in registers: their bytes may be overwritten
#if defined(C2_DBG) && C2_DBG == 8
                        {   // debug: read the four rows and the scales a second time; do they still hold the same values?
                            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const volatile float* tp = reinterpret_cast<const volatile float*>(blk + r_base + (ub + q) * 512 + ((r_chunk ^ (unsigned)((ub + q) & 7)) * 16));
                                const float4 t = make_float4(tp[0], tp[1], tp[2], tp[3]);
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                if (t.x != r[q].x || t.y != r[q].y || t.z != r[q].z || t.w != r[q].w) atomicAdd(&g_c2_dbg[q], 1u);
                            }
                            const volatile float* tp = inv_tab + ub;
                            const float4 t = make_float4(tp[0], tp[1], tp[2], tp[3]);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            if (t.x != inv4.x || t.y != inv4.y || t.z != inv4.z || t.w != inv4.w) atomicAdd(&g_c2_dbg[4], 1u);
                            if (lane == 0) atomicAdd(&g_c2_dbg[5], 1u);
                        }
#endif
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int u = ub + q;
                            const float inv = q == 0 ? inv4.x : q == 1 ? inv4.y : q == 2 ? inv4.z : inv4.w;
                            if (DGRAD) {
                                r[q] = make_float4(r[q].x * inv, r[q].y * inv, r[q].z * inv, r[q].w * inv);
                                if (f_bits) {
                                    auto word2 = [&](int cc) {
                                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)sg_lo, 4 * u + cc), hi = (unsigned)__builtin_amdgcn_readlane((int)sg_hi, 4 * u + cc);
                                        return ((unsigned long long)hi << 32) | lo;
                                    };
                                    select_by_masks(r[q], slope, word2(0), word2(1), word2(2), word2(3));
                                } else if (mask_rows) {
                                    long row = m0 + wn * 16 + u;
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
                                const long row = m0 + wn * 16 + u;
                                const float mean = wave_sum((r[q].x + r[q].y) + (r[q].z + r[q].w)) / (float)N;
                                float4 dl = col_ok ? make_float4(r[q].x - mean, r[q].y - mean, r[q].z - mean, r[q].w - mean) : make_float4(0.f, 0.f, 0.f, 0.f);
                                const float sigma = sqrtf(wave_sum((dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w)) / (float)(N - 1));
                                const float rinv = 1.0f / (sigma + p.norm_eps);
                                r[q] = make_float4(dl.x * rinv, dl.y * rinv, dl.z * rinv, dl.w * rinv);
                                if (lane == 0 && row < p.M) { p.norm_stats[row * 2] = rinv; p.norm_stats[row * 2 + 1] = sigma; }
                            }
                            if (f_store && col_ok && (f_full || m0 + wn * 16 + u < p.M)) *reinterpret_cast<float4*>(crow + u * ldc) = r[q];
                            if (!DGRAD && f_bits) {
                                const unsigned long long k0 = __ballot(r[q].x > 0.f), k1 = __ballot(r[q].y > 0.f), k2 = __ballot(r[q].z > 0.f), k3 = __ballot(r[q].w > 0.f);
                                put_masks(bt_lo, bt_hi, q, k0, k1, k2, k3);
                            }
                        }
                        if (f_more || f_rmax) {
                            float mx[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) mx[q] = fmaxf(fmaxf(fabsf(r[q].x), fabsf(r[q].y)), fmaxf(fabsf(r[q].z), fabsf(r[q].w)));
                            float smx[4];
#if defined(C2_DBG) && (C2_DBG == 10 || C2_DBG == 11)
#pragma unroll
                            for (int q = 0; q < 4; ++q) smx[q] = wave64_max(mx[q]);
#else
                            wave_max4(mx[0], mx[1], mx[2], mx[3]);
#pragma unroll
                            for (int q = 0; q < 4; ++q) smx[q] = last_lane(mx[q]);
#endif
                            if (f_rmax) {
                                float mx4 = 0.f;
#if defined(C2_DBG) && (C2_DBG == 10 || C2_DBG == 12)
                                mx4 = lane == 0 ? smx[0] : lane == 1 ? smx[1] : lane == 2 ? smx[2] : smx[3];
#else
                                put4(mx4, 0, smx[0], smx[1], smx[2], smx[3]);
#endif
                                if (lane < 4 && (f_full || m0 + wn * 16 + ub + lane < p.M)) L.rowmax[m0 + wn * 16 + ub + lane] = mx4;
                            }
                            if (f_more) {
                                float inv_n[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const float sc = scale_from_max(__float_as_uint(smx[q]), inv_n[q]);
                                    write_planes(ub + q, r[q], sc, kpad_next);
                                }
                                *reinterpret_cast<float4*>(inv_tab + ub) = make_float4(inv_n[0], inv_n[1], inv_n[2], inv_n[3]);
                            }
                        }
                        // 128 contiguous bytes per batch: word pair 4 q + cc of rows ub .. ub + 3
                        if (!DGRAD && f_bits && lane < 16 && (f_full || m0 + wn * 16 + ub + (lane >> 2) < p.M))
                            *reinterpret_cast<uint2*>(sgn + (ub * 8 + 2 * lane)) = make_uint2(bt_lo, bt_hi);
                    }
                };
                // hot combinations (everything 256 wide, tile inside M): training middle layer / inference middle layer / data-gradient
                if (!generic_only && rt_full && more && !rt_norm && !mask_rows && rt_store && rt_bits && rt_rmax) rows(RowCfg<1, 1, 1, 1, 0, 1>());
                else if (!generic_only && !DGRAD && rt_full && more && !rt_norm && !rt_store && !rt_bits && !rt_rmax) rows(RowCfg<0, 0, 0, 1, 0, 1>());
                else rows(RowCfg<2, 2, 2, 2, 2, 2>());
            }
            C2_STAMP();
            if (next_tile) stage((tile + tstride) * C2_ROWS);
            C2_STAMP();
#if defined(C2_DBG) && C2_DBG == 2
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_sleep(127);
#endif
            lds_barrier();                          // (c) planes of the next layer (or tile) ready
        }
    }
    if (grp == 0) lds_barrier();                    // group 0 ends half a period early
}

}  // namespace

#ifdef PAPR_H3_TRACE
extern "C" int papr_chain2_trace_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain2_trace), sizeof(long long) * 1024) == hipSuccess ? 0 : 1; }
#endif

#if defined(C2_DBG) && (C2_DBG == 8 || C2_DBG == 13)
extern "C" int papr_chain2_dbg_read(unsigned* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c2_dbg), sizeof(unsigned) * 16) == hipSuccess ? 0 : 1; }
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
