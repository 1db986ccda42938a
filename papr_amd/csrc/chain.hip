// K3b: a whole run of embedding-MLP layers in ONE kernel (split-f16 MFMA, see gemm.hip for the arithmetic).
//
// Replaces the same reference lines as gemm_nt (MLP.forward, models/mlp.py:47-59, and its autograd
// data-gradient) for runs of consecutive layers at most 256 wide (a forward run may contain skip layers: their second
// K segment is the run's own input; data-gradient runs stop at them).
//
// Why: layer by layer, the split-f16 GEMM is HBM-bound -- it reads the (M x 256) input and writes the
// (M x 256) output of every layer.  But a workgroup that owns COMPLETE rows of a layer's output owns the
// A operand of the next layer for the same rows.  So a workgroup carries 64 rows through all layers of
// the run: the activations stay in LDS as split f16 planes, HBM sees the input once and each saved
// activation once (training), or only the last one (inference).  The data-gradient run is the same
// kernel walking the layers downwards with the transposed weights and the activation derivative as mask.
//
// Shape: 4 waves (one per SIMD), each a 64-row x 64-column corner of the 64 x 256 tile; 76.5 KB of LDS,
// so TWO workgroups share a CU and one multiplies while the other is in its row phases.
//
// Per layer and tile:   k-loop   16 k-steps x 12 MFMA per wave.  The MFMA takes the W fragment as its row
//                                operand, so the accumulators hold C^T: a lane owns ONE ROW of the tile
//                                and its registers are runs of 4 consecutive columns.  A fragments from
//                                LDS (resident: no staging, no barriers), W fragments straight from L2 in
//                                fragment order through a register ring (three k-steps ahead), every
//                                request alone behind an MFMA (sched_barrier pins that; see the k-loop)
//                       phase 1  in registers, no LDS bounce: un-scale + bias + activation (one fma, one
//                                max) or derivative mask (sign words the forward run left: one bit per
//                                activation), 16-byte row stores, row maximum -> LDS atomic
//                       phase 2  per-row power-of-two scale from the row maxima, hi/lo split, 8-byte LDS
//                                writes: the rows become the A planes of the next layer
// with one LDS-only barrier between the stages.  The last layer of a forward run can standardise its rows
// (LayerNorm core of the key / query embeddings) before they are stored.
//
// Measured (512,000 rows, 256-wide layers, MI355X): 215 us per layer in inference and 300 / 285 us per
// forward / data-gradient layer in training, against 369 / 483 us for the per-layer kernel.  Two thirds of
// the time is not matrix work.  In-box ablations (scripts/probes/build_variant.sh + the CH_EXP_* switches below):
// the weight-fragment stream costs 22 %, the wait for a tile's input rows 13 % (training), storing the layers
// 29 % of a forward run (10 points of it the row-per-lane pattern); the instruction count of the row phases
// hardly matters (an MFMA stream and a vector stream of two waves of a SIMD run side by side,
// scripts/probes/mfma_vs_valu_waves.hip) -- the waves wait.
#include "papr_common.h"
#include "h3_common.h"
#include "chain.h"
#include <stdlib.h>
#include <type_traits>

namespace {

#ifndef CH_NI
#define CH_NI 2                         // 32-row tiles per wave = rows per workgroup tile / 32 (2: two workgroups fit a CU; 4: one)
#endif
constexpr int NI = CH_NI;
constexpr int CH_BM = 32 * NI;              // rows per tile
#ifndef CH_START_DELAY
#define CH_START_DELAY 0              // x 8128 cycles (s_sleep 127); 0, 1 and 2 measured the same
#endif
// Tile shape (compile-time, A/B-tested on 512,000 x 256 x 256 x 4 layers; forward / data-gradient / inference run in us):
//   NJ=2 NI=2  4 waves of 64 x 64, two workgroups per CU      1165 / 1177 / 884   <- default
//   NJ=1 NI=2  8 waves of 64 x 32, two workgroups per CU      1221 / 1111 / 904   (twice the waves: no gain)
//   NJ=1 NI=4  8 waves of 128 x 32, one workgroup per CU      1181 / 1135 / 942   (half the W traffic through L1: no gain)
//   NJ=2 NI=4  4 waves of 128 x 64, one workgroup per CU      1354 / 1240 / 1107  (nothing overlaps the row phases)
//   NJ=2 NI=1  4 waves of 32 x 64, three workgroups per CU    1336 / 1281 / 986   (twice the W traffic per MFMA)
#ifndef CH_STAGE_ROWS
#define CH_STAGE_ROWS 8                 // input rows a wave requests at once when it stages a tile (of its 64 / waves rows); 16 (one round trip instead of two) measured 2-5 % slower: 16 more spilled registers
#endif
#ifndef CH_NJ
#define CH_NJ 2                         // 32-column tiles per wave: 2 = four waves of 64 x 64 (one per SIMD), 1 = eight waves of 64 x 32 (two per SIMD)
#endif
constexpr int NJ = CH_NJ;
constexpr int CH_WAVES = 8 / NJ;
constexpr int CH_THREADS = 64 * CH_WAVES;
constexpr int CH_AP = 264;                  // A-plane row pitch in halfs (528 B: conflict-free ds_read_b128)
constexpr int CH_PLANE = CH_BM * CH_AP;     // halfs per plane
constexpr size_t CH_LDS_BYTES = (size_t)2 * CH_PLANE * sizeof(_Float16) + 3 * CH_BM * sizeof(float) + 2 * CH_BM * sizeof(unsigned) + CHAIN_MAX_LAYERS * 256 * sizeof(float);

// max over the 8 lanes that share a row segment (lanes 8q .. 8q+7), on the DPP network
__device__ __forceinline__ float seg8_max(float v) {
    int x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false)));    // quad_perm [1,0,3,2]
    x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false)));    // quad_perm [2,3,0,1]
    x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false)));   // row_half_mirror
    return v;
}
__device__ __forceinline__ float wave64_max(float v) {
    v = seg8_max(v);
    int x = __float_as_int(v);
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false)));   // row_mirror: 16 lanes
    // the four 16-lane rows meet on the scalar unit (a ds_bpermute butterfly costs two LDS round trips per row)
    const int r = __float_as_int(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(r, 0)), b = __int_as_float(__builtin_amdgcn_readlane(r, 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(r, 32)), d = __int_as_float(__builtin_amdgcn_readlane(r, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

__device__ __forceinline__ float scale_from_max(unsigned bits, float& inv) {      // row max -> [2^13, 2^14)
    const int ea = bits ? (int)((bits >> 23) & 0xff) : 127 + 13;
    inv = pow2_from_biased(127 - 13 + (ea - 127));
    return pow2_from_biased(127 + 13 - (ea - 127));
}

#ifdef PAPR_H3_TRACE
__device__ long long g_chain_trace[256];
#define CH_STAMP() do { if (blockIdx.x == 100 && threadIdx.x == 0 && trace_slot < 256) g_chain_trace[trace_slot++] = __builtin_readcyclecounter(); } while (0)
#else
#define CH_STAMP() do {} while (0)
#endif

// BITS (data-gradient only): the activation derivative comes from the sign words the forward run left behind
// (L.sign_bits) instead of the fp32 activation rows (L.mask): 16 MB instead of 524 MB per layer, and 2 registers
// instead of 64.
constexpr int CH_WGS_PER_CU = NI == 1 ? 3 : (NI == 2 ? 2 : 1);     // what fits the LDS (42 / 76.5 / 144 KB per workgroup)
// SKIP (forward only): the run contains a skip layer (second K segment = the run's input); a separate instantiation, so
// that the extra control flow around the k-loop does not cost the common kernels registers.
template <bool DGRAD, bool BITS, bool SKIP = false>
__global__ __launch_bounds__(CH_THREADS, CH_WGS_PER_CU * CH_WAVES / 4) void mlp_chain_kernel(ChainArgs p, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Ah = reinterpret_cast<_Float16*>(smem);
    _Float16* Al = Ah + CH_PLANE;
    float* inv_tab = reinterpret_cast<float*>(Al + CH_PLANE);            // [64] 1/scale of the rows of the A planes
    unsigned* rmax_tab = reinterpret_cast<unsigned*>(inv_tab + CH_BM);   // [2][64] max|C| bit patterns, by layer parity
    float* scl_tab = reinterpret_cast<float*>(rmax_tab + 2 * CH_BM);      // [rows] scale of the rows of the A planes (= 1 / inv_tab)
    unsigned* xmax_tab = reinterpret_cast<unsigned*>(scl_tab + CH_BM);    // [rows] max |x| bits of the tile's input rows (skip layers multiply x again)
    float* bias_tab = reinterpret_cast<float*>(xmax_tab + CH_BM);    // [layers][256]: a global load in phase 1 would wait for the stores before it (loads and stores share vmcnt)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int wm = 0;                        // (one wave row: the tile is 64 rows)
    const int wn = wave;                         // wave tile: all 64 rows, columns 32 NJ wn .. + 32 NJ - 1
    const int frag = (lane & 31) * CH_AP + 8 * (lane >> 5);

#ifdef PAPR_H3_TRACE
    int trace_slot = 0;
#endif
    for (int l = 0; l < p.n_layers; ++l)
        if (tid < 256) bias_tab[l * 256 + tid] = (!DGRAD && p.L[l].bias && tid < p.L[l].N) ? p.L[l].bias[tid] : 0.f;

    // W fragments: fragment (n-tile t, k-step s) of a layer's planes starts at ((t * ksteps + s) * 64 + lane) * 8 halfs.
    // They come straight from L2 (~1k cycles under load, a k-step is 384): a ring of WD k-steps, WD-1 in
    // flight, and the first ones of the NEXT layer are requested before this layer's row phases.
    constexpr int WD = 4;                        // ring depth (the k-loop below is written out for 4)
    half8 wfh[WD][NJ], wfl[WD][NJ];
    auto load_w = [&](int l, int ks, half8 (&qh)[NJ], half8 (&ql)[NJ]) {
        const ChainLayer& L = p.L[l];
        ks = ks < L.ksteps ? ks : L.ksteps - 1;
#ifdef CH_EXP_NO_W
        if (ks > 2) return;                  // experiment: only the first fragments are ever loaded
#endif
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int t = 32 * (wn * NJ + j) < L.N ? wn * NJ + j : 0;
            const long o = ((long)(t * L.ksteps + ks) * 64 + lane) * 8;
            qh[j] = *reinterpret_cast<const half8*>(L.w_hi + o);
#ifdef CH_EXP_HALF_W                    // experiment (wrong numbers): only the hi plane crosses L1 -- half the weight bytes, same matrix work
            ql[j] = qh[j];
#else
            ql[j] = *reinterpret_cast<const half8*>(L.w_lo + o);
#endif
        }
    };
#pragma unroll
    for (int u = 0; u < WD - 1; ++u) load_w(0, u, wfh[u], wfl[u]);
    // Two workgroups share a CU so that one multiplies while the other is in its row phases.  (Experiment: start the
    // second half of the grid late, in case equal timing keeps the pairs in lockstep.  No effect: off.)
    if (blockIdx.x >= gridDim.x / 2) {
#pragma unroll 1
        for (int z = 0; z < CH_START_DELAY; ++z) __builtin_amdgcn_s_sleep(127);
    }
    for (int tile = blockIdx.x; tile < tiles_m; tile += gridDim.x) {
        const long m0 = (long)tile * CH_BM;
        CH_STAMP();
        // ---- stage the tile's input rows: wave w carries rows w, w + CH_WAVES, ...; row maximum on the way.  (Requesting
        // them during the previous tile's last layer bought nothing -- the other workgroup of the CU fills the wait -- and
        // cost 64 registers.)  first = false: a skip layer multiplies x again, with the scale its rows already carry.
        auto stage_rows = [&](int kpad, bool first, auto batch) {
            constexpr int SR = decltype(batch)::value;            // rows in flight per wave: 8, or 4 where the accumulators are live
            const int c = 4 * lane;
#pragma unroll 1
            for (int u0 = 0; u0 < CH_BM / CH_WAVES; u0 += SR) {
                float4 v[SR];
#pragma unroll
                for (int u = 0; u < SR; ++u) {
                    long m = m0 + wave + CH_WAVES * (u0 + u);
                    m = m < p.M ? m : p.M - 1;
#ifdef CH_EXP_NO_STAGING
                    v[u] = make_float4((float)m, 1.f, 2.f, 3.f);
#else
                    v[u] = c < p.K0 ? *reinterpret_cast<const float4*>(p.A0 + m * p.lda0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
                }
#pragma unroll
                for (int u = 0; u < SR; ++u) {
                    const int r = wave + CH_WAVES * (u0 + u);
                    float sc;
                    if (first) {
                        if (!DGRAD && p.in_norm_stats) {
                            // LayerNorm core in front of the run (FeedForward.innorm): the wave holds the whole row
                            const int wdt = p.in_norm_width;
                            const bool i0 = c < wdt, i1 = c + 1 < wdt, i2 = c + 2 < wdt, i3 = c + 3 < wdt;
                            const float mean = wave_sum(((i0 ? v[u].x : 0.f) + (i1 ? v[u].y : 0.f)) + ((i2 ? v[u].z : 0.f) + (i3 ? v[u].w : 0.f))) / (float)wdt;
                            float4 dl = make_float4(i0 ? v[u].x - mean : 0.f, i1 ? v[u].y - mean : 0.f, i2 ? v[u].z - mean : 0.f, i3 ? v[u].w - mean : 0.f);
                            const float sigma = sqrtf(wave_sum((dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w)) / (float)(wdt - 1));
                            const float rinv = 1.0f / (sigma + p.in_norm_eps);
                            v[u] = make_float4(dl.x * rinv, dl.y * rinv, dl.z * rinv, dl.w * rinv);
                            const long mrow = m0 + r;
                            if (mrow < p.M) {
                                if (p.in_norm_writeback && c < p.K0) *reinterpret_cast<float4*>(p.A0 + mrow * p.lda0 + c) = v[u];
                                if (lane == 0) { p.in_norm_stats[mrow * 2] = rinv; p.in_norm_stats[mrow * 2 + 1] = sigma; }
                            }
                        }
                        float mx = fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
                        mx = wave64_max(mx);
                        // (when layer 0 feeds a skip layer later, that layer rescales; layer 0 itself only sees x)
                        float inv;
                        sc = scale_from_max(__float_as_uint(mx), inv);
                        if (lane == 0) {
                            inv_tab[r] = inv;
                            scl_tab[r] = sc;
                            xmax_tab[r] = __float_as_uint(mx);
                            if (p.rowmax0 && m0 + r < p.M) p.rowmax0[m0 + r] = mx;
                        }
                    } else {
                        sc = scl_tab[r];
                    }
                    if (c < kpad) {
                        half4 hi, lo;
                        split4(v[u], sc, hi, lo);
                        *reinterpret_cast<half4*>(Ah + r * CH_AP + c) = hi;
                        *reinterpret_cast<half4*>(Al + r * CH_AP + c) = lo;
                    }
                }
            }
        };
        stage_rows(p.L[0].k1steps * 16, true, std::integral_constant<int, CH_STAGE_ROWS>());
        for (int t = tid; t < CH_BM; t += CH_THREADS) rmax_tab[t] = 0u;
        lds_barrier();
        CH_STAMP();

#pragma unroll 1
        for (int l = 0; l < p.n_layers; ++l) {
            const ChainLayer& L = p.L[l];
            const int par = l & 1;
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

            // data-gradient: the activation rows whose derivative masks the result are requested before the k-loop
            // (a load issued among the stores of phase 1 would wait for them: loads and stores share vmcnt)
            const int hh = lane >> 5;
            float4 aux[NI][NJ][4];
            unsigned mbits[NI] = {};
            auto load_aux = [&](int i) {
                long row = m0 + i * 32 + (lane & 31);
                row = row < p.M ? row : p.M - 1;
                if (BITS) {
                    mbits[i] = L.sign_bits[(long)(wn * 2 + hh) * p.M + row];
                    return;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        int col = (wn * NJ + j) * 32 + 8 * g + 4 * hh;
                        col = col < N ? col : 0;
                        aux[i][j][g] = *reinterpret_cast<const float4*>(L.mask + row * L.ld_mask + col);
                    }
            };
            const bool masked = DGRAD && (BITS ? L.sign_bits != nullptr : L.mask != nullptr);
            if (masked) {
#pragma unroll
                for (int i = 0; i < NI; ++i) load_aux(i);
            }

            // ---- k-loop.  The A fragments of k-step s+1 are read from LDS while k-step s multiplies (hipcc does not
            // pipeline the reads by itself: it places them right in front of their MFMAs and waits).
            // (kb, ke): the k-steps of one K segment -- all of them, or [0, k1steps) and [k1steps, ksteps) of a skip layer.  W
            // fragments are indexed by the layer-wide k-step, A fragments by the step inside the segment (the A planes hold
            // one segment at a time).
            int kb = 0, ke = SKIP ? L.k1steps : ksteps;
            auto load_a = [&](int ks, half8 (&ah)[NI], half8 (&al)[NI]) {
                ks = (ks < ke ? ks : ke - 1) - kb;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int o = (wm * 2 + i) * 32 * CH_AP + frag + ks * 16;
                    ah[i] = *reinterpret_cast<const half8*>(Ah + o);
                    al[i] = *reinterpret_cast<const half8*>(Al + o);
                }
            };
            // W fragment as the row operand: the accumulators hold C^T (lane = row m, registers = columns n)
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
            auto k_loop = [&](auto mma) {                       // waves with a dead column tile (narrow layers): simple form
                half8 a0h[NI], a0l[NI], a1h[NI], a1l[NI];
                load_a(kb, a0h, a0l);
#pragma unroll 1
                for (int ks = kb; ks < ke; ks += 4) {          // segment lengths are even (planes are padded to 32 columns)
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
            // The hot form (both column tiles live).  One wave per SIMD and workgroup means nothing else covers this
            // wave's issue slots: a block of ~25 load / address instructions between two k-steps leaves the matrix pipe
            // idle for ~130 cycles per 384-cycle step (measured).  So every request sits alone behind an MFMA -- a
            // 32-cycle MFMA hides about five issue slots -- and sched_barrier pins that order (left alone, hipcc sinks the
            // loads to just before their use and waits with vmcnt(0)).
            // (scalar base + 32-bit lane offset: a 64-bit VALU address computation per load costs far more than its issue
            // slot next to MFMAs)
            const char* wb_h[NJ];
            const char* wb_l[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                wb_h[j] = reinterpret_cast<const char*>(L.w_hi) + (size_t)((wn * NJ + j) * ksteps) * 1024;
                wb_l[j] = reinterpret_cast<const char*>(L.w_lo) + (size_t)((wn * NJ + j) * ksteps) * 1024;
            }
            const unsigned lane16 = (unsigned)lane * 16u;
#define CH_SB __builtin_amdgcn_sched_barrier(0)
#define CH_MFMA(i, j, w, x) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[j], x[i], acc[i][j], 0, 0, 0)
            auto step = [&](const half8 (&uh)[NJ], const half8 (&ul)[NJ], const half8 (&xh)[NI], const half8 (&xl)[NI],
                            int kw, half8 (&nh)[NJ], half8 (&nl)[NJ], int ka, half8 (&yh)[NI], half8 (&yl)[NI]) {
                kw = kw < ksteps ? kw : ksteps - 1;
                ka = (ka < ke ? ka : ke - 1) - kb;
#ifdef CH_EXP_NO_W
                const bool ldw = kw <= 2;
#else
                constexpr bool ldw = true;
#endif
#ifdef CH_EXP_NO_A
                const bool lda = ka <= 1;
#else
                constexpr bool lda = true;
#endif
                // request q of this step: the 2 NJ W fragments of k-step kw, then the 2 NI A fragments of k-step ka
                auto request = [&](int q) {
                    if (q < 2 * NJ) {
                        if (!ldw) return;
                        const int j = q >> 1;
                        if (q & 1) nl[j] = *reinterpret_cast<const half8*>(wb_l[j] + (size_t)kw * 1024 + lane16);
                        else nh[j] = *reinterpret_cast<const half8*>(wb_h[j] + (size_t)kw * 1024 + lane16);
                    } else if (q < 2 * NJ + 2 * NI) {
                        if (!lda) return;
                        const int a = q - 2 * NJ, i = a >> 1;
                        if (a & 1) yl[i] = *reinterpret_cast<const half8*>(Al + i * 32 * CH_AP + frag + ka * 16);
                        else yh[i] = *reinterpret_cast<const half8*>(Ah + i * 32 * CH_AP + frag + ka * 16);
                    }
                };
                int q = 0;                                      // (everything below unrolls: q is a compile-time constant at each use)
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            if (term == 0) CH_MFMA(i, j, uh, xl);
                            else if (term == 1) CH_MFMA(i, j, ul, xh);
                            else CH_MFMA(i, j, uh, xh);
                            CH_SB;
                            request(q++);
                            CH_SB;
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
#undef CH_MFMA
#undef CH_SB
#ifndef CH_EXP_NO_KLOOP
            // A skip layer has a second K segment: the run's own input x (reference MLP skip_layers, models/mlp.py:54-55).  Its
            // rows come into the A planes again -- split with the scale the rows already carry, which the previous layer's
            // phase 2 chose from max(row max of its output, row max of x) -- and the k-loop goes on.  k1steps is a multiple
            // of 4: the W ring is back at set 0 and already holds k-steps k1steps, +1, +2.  (One loop, so that the k-loop code
            // exists once: a second copy costs registers.)
#pragma unroll 1
            for (int seg = 0; seg < (SKIP && L.k1steps < ksteps ? 2 : 1); ++seg) {
                if (SKIP && seg) {
                    lds_barrier();                              // everyone is done with the first segment's A planes
                    stage_rows((ksteps - L.k1steps) * 16, false, std::integral_constant<int, 4>());
                    lds_barrier();
                    kb = L.k1steps; ke = ksteps;
                }
                if (live[NJ - 1]) k_loop_full();                // every column tile of this wave holds real columns
                else if (live[0]) k_loop(mma_live);
            }
#endif
            {   // first fragments of what comes next: the next layer, or layer 0 of the next tile
                const int ln = l + 1 < p.n_layers ? l + 1 : 0;
#pragma unroll
                for (int u = 0; u < WD - 1; ++u) load_w(ln, u, wfh[u], wfl[u]);
            }
            CH_STAMP();
            // Phase 1 touches no LDS the k-loop reads (row-maximum table, bias table, 1/scale table): it may start while
            // other waves still multiply.  Only the standardising last layer puts its partial sums where the A planes are.
            if (!DGRAD && l + 1 == p.n_layers && p.norm_stats) lds_barrier();
            CH_STAMP();

            // ---- phase 1 (accumulators hold C^T: lane = row, registers = columns): un-scale, bias / activation or
            // derivative mask, 16-byte stores, row maxima
            // Instruction count matters here (one wave per SIMD and workgroup: nothing hides it).  acc * inv is exact (inv is
            // a power of two), so fma(acc, inv, bias) equals the separate multiply and add bit for bit; ReLU is one
            // v_max.  Columns beyond N need no masking: the weight planes are zero there, so acc is.
            const bool more = l + 1 < p.n_layers;
            const float slope = L.act == PAPR_ACT_RELU ? 0.f : (L.act == PAPR_ACT_LEAKY_RELU ? 0.2f : 1.f);     // data-gradient: derivative on the negative side
            const bool want_bits = !DGRAD && L.sign_bits != nullptr;      // forward: leave the signs for the data-gradient run
            // Straight-line code matters here: with a test around every 16-byte store the compiler cut the phase into
            // ~50 basic blocks, each with its own LDS wait, exec-mask juggling and a reload of the row stride (~450 scalar
            // instructions per layer and wave).  So: compute everything first, then the sign words under ONE uniform test,
            // then all stores of a 32-row block under ONE lane mask (row < M), with the row's base pointer computed once.
            float* const Cbase = L.C;
            const long ldc = L.ldc;
            const bool cols_whole = (N & 31) == 0;              // every live 32-column tile lies wholly inside the matrix
            auto rows_phase = [&](auto act_fn) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int rl = i * 32 + (lane & 31);
                    const long row = m0 + rl;
                    const float inv = inv_tab[rl];
                    float mx = 0.f;
                    unsigned mb = mbits[i];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!live[j]) { mb <<= 16; continue; }
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = (wn * NJ + j) * 32 + 8 * g + 4 * hh;
                            float4 r;
                            if (DGRAD) {
                                r = make_float4(acc[i][j][4 * g] * inv, acc[i][j][4 * g + 1] * inv, acc[i][j][4 * g + 2] * inv, acc[i][j][4 * g + 3] * inv);
                                if (masked) {
                                    if (BITS) {
                                        r.x *= (int)mb < 0 ? 1.f : slope; r.y *= (int)(mb << 1) < 0 ? 1.f : slope;
                                        r.z *= (int)(mb << 2) < 0 ? 1.f : slope; r.w *= (int)(mb << 3) < 0 ? 1.f : slope;
                                        mb <<= 4;
                                    } else {
                                        const float4 a4 = aux[i][j][g];
                                        r.x *= a4.x > 0.f ? 1.f : slope; r.y *= a4.y > 0.f ? 1.f : slope;
                                        r.z *= a4.z > 0.f ? 1.f : slope; r.w *= a4.w > 0.f ? 1.f : slope;
                                    }
                                }
                            } else {
                                const float4 b4 = *reinterpret_cast<const float4*>(bias_tab + l * 256 + col);
                                r.x = act_fn(__builtin_fmaf(acc[i][j][4 * g], inv, b4.x), 0.f);
                                r.y = act_fn(__builtin_fmaf(acc[i][j][4 * g + 1], inv, b4.y), 0.f);
                                r.z = act_fn(__builtin_fmaf(acc[i][j][4 * g + 2], inv, b4.z), 0.f);
                                r.w = act_fn(__builtin_fmaf(acc[i][j][4 * g + 3], inv, b4.w), 0.f);
                            }
                            acc[i][j][4 * g] = r.x; acc[i][j][4 * g + 1] = r.y; acc[i][j][4 * g + 2] = r.z; acc[i][j][4 * g + 3] = r.w;
                            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
                        }
                    }
                    if (live[0]) atomicMax(rmax_tab + par * CH_BM + rl, __float_as_uint(mx));
                    if (want_bits) {                            // forward: sign word of this lane's 32 values, first value in the top bit
                        unsigned sb = 0u;
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            if (!live[j]) { sb <<= 16; continue; }
#pragma unroll
                            for (int e = 0; e < 16; ++e) sb = (sb << 1) | (acc[i][j][e] > 0.f ? 1u : 0u);
                        }
                        if (row < p.M) L.sign_bits[(long)(wn * 2 + hh) * p.M + row] = sb << (32 - 16 * NJ);
                    }
                    if (Cbase) {
                        // (plain stores: the 32-byte pieces of a line meet in L2; non-temporal stores double the run time)
                        float* const crow = Cbase + row * ldc + 4 * hh;
                        if (cols_whole) {
                            if (row < p.M) {
#pragma unroll
                                for (int j = 0; j < NJ; ++j) {
                                    if (!live[j]) continue;
#pragma unroll
                                    for (int g = 0; g < 4; ++g)
#ifdef CH_EXP_COALESCED_STORES      // experiment (wrong layout, same bytes): what fully coalesced stores of the tile would cost
                                        *reinterpret_cast<float4*>(Cbase + m0 * ldc + ((long)((wn * NI + i) * NJ * 4 + j * 4 + g) * 64 + lane) * 4) =
#else
                                        *reinterpret_cast<float4*>(crow + (wn * NJ + j) * 32 + 8 * g) =
#endif
                                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    if (live[j] && (wn * NJ + j) * 32 + 8 * g + 4 * hh < N && row < p.M)
                                        *reinterpret_cast<float4*>(crow + (wn * NJ + j) * 32 + 8 * g) =
                                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                        }
                    }
                }
            };
#ifndef CH_EXP_NO_PHASES
            if (!DGRAD && !more && p.norm_stats) {
                // ---- last layer with LayerNorm core behind it (act = none): the rows are complete in this workgroup, so
                // they are standardised before they ever reach memory.  Two-pass mean / unbiased std like papr_rownorm_fwd;
                // the partial sums of a row's 2 x CH_WAVES lanes meet in LDS in a FIXED order (atomics would make the last
                // bits, and with them the chunk-invariance of evaluate, depend on timing).  The tables alias the A planes,
                // which nobody reads after the barrier above.
                constexpr int PW = 2 * CH_WAVES;
                float* part0 = smem;
                float* part1 = smem + CH_BM * PW;
                float mean[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int rl = i * 32 + (lane & 31);
                    const float inv = inv_tab[rl];
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!live[j]) continue;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = (wn * NJ + j) * 32 + 8 * g + 4 * hh;
                            const float4 b4 = *reinterpret_cast<const float4*>(bias_tab + l * 256 + col);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float r = __builtin_fmaf(acc[i][j][4 * g + c], inv, c == 0 ? b4.x : c == 1 ? b4.y : c == 2 ? b4.z : b4.w);
                                acc[i][j][4 * g + c] = r;
                                sum += col + c < N ? r : 0.f;
                            }
                        }
                    }
                    part0[rl * PW + wn * 2 + hh] = sum;
                }
                lds_barrier();
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int rl = i * 32 + (lane & 31);
                    float tot = 0.f;
#pragma unroll
                    for (int q = 0; q < PW; ++q) tot += part0[rl * PW + q];
                    mean[i] = tot / (float)N;
                    float ss = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!live[j]) continue;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = (wn * NJ + j) * 32 + 8 * g + 4 * hh;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float dlt = acc[i][j][4 * g + c] - mean[i];
                                acc[i][j][4 * g + c] = dlt;
                                ss += col + c < N ? dlt * dlt : 0.f;
                            }
                        }
                    }
                    part1[rl * PW + wn * 2 + hh] = ss;
                }
                lds_barrier();
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int rl = i * 32 + (lane & 31);
                    const long row = m0 + rl;
                    float tot = 0.f;
#pragma unroll
                    for (int q = 0; q < PW; ++q) tot += part1[rl * PW + q];
                    const float sigma = sqrtf(tot / (float)(N - 1));
                    const float rinv = 1.0f / (sigma + p.norm_eps);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!live[j]) continue;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = (wn * NJ + j) * 32 + 8 * g + 4 * hh;
                            const float4 y = make_float4(acc[i][j][4 * g] * rinv, acc[i][j][4 * g + 1] * rinv, acc[i][j][4 * g + 2] * rinv, acc[i][j][4 * g + 3] * rinv);
                            if (L.C && col < N && row < p.M) *reinterpret_cast<float4*>(L.C + row * L.ldc + col) = y;
                        }
                    }
                    if (wn == 0 && hh == 0 && row < p.M) { p.norm_stats[row * 2] = rinv; p.norm_stats[row * 2 + 1] = sigma; }
                }
            }
            else if (DGRAD || L.act == PAPR_ACT_NONE) rows_phase([](float v, float) { return v; });
            else if (L.act == PAPR_ACT_RELU) rows_phase([](float v, float) { return fmaxf(v, 0.f); });
            else rows_phase([](float v, float) { return fmaxf(v, 0.2f * v); });
#endif
            CH_STAMP();
            lds_barrier();                                      // row maxima complete
            CH_STAMP();

            // ---- phase 2: the rows become the A planes of the next layer
            for (int t = tid; t < CH_BM; t += CH_THREADS) {
                if (L.rowmax && m0 + t < p.M) L.rowmax[m0 + t] = __uint_as_float(rmax_tab[par * CH_BM + t]);
                rmax_tab[(par ^ 1) * CH_BM + t] = 0u;
            }
#ifndef CH_EXP_NO_PHASES
            if (more) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int rl = i * 32 + (lane & 31);
                    float inv;
                    unsigned mxb = rmax_tab[par * CH_BM + rl];
                    if (SKIP && p.L[l + 1].k1steps < p.L[l + 1].ksteps) mxb = max(mxb, xmax_tab[rl]);      // next layer also multiplies x
                    const float sc = scale_from_max(mxb, inv);
                    if (wn == 0 && hh == 0) { inv_tab[rl] = inv; scl_tab[rl] = sc; }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!live[j]) continue;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = (wn * NJ + j) * 32 + 8 * g + 4 * hh;
                            half4 hi, lo;
                            split4(make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]), sc, hi, lo);
                            *reinterpret_cast<half4*>(Ah + rl * CH_AP + col) = hi;
                            *reinterpret_cast<half4*>(Al + rl * CH_AP + col) = lo;
                        }
                    }
                }
            }
#endif
            lds_barrier();
            CH_STAMP();
        }
    }
}

}  // namespace

#ifdef PAPR_H3_TRACE
extern "C" int papr_chain_trace_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_trace), sizeof(long long) * 256) == hipSuccess ? 0 : 1; }
#endif

size_t papr_chain_lds_bytes() { return CH_LDS_BYTES; }

int papr_chain_version(const ChainArgs& a) {
    // PAPR_CHAIN=1: every run on this file's kernel, 2: chain2.hip, 3: chain3.hip (A/B); default: chain4.hip.  An MLP with a skip layer
    // (legacy: the sign-word layout of this file in all its runs) stays here unless the version is 4
    static const int version = getenv("PAPR_CHAIN") ? atoi(getenv("PAPR_CHAIN")) : 4;
    if (version == 4) return 4;
    return a.legacy ? 1 : version;
}

int papr_launch_chain(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s) {
    const int version = papr_chain_version(a);
    if (version == 4) return papr_launch_chain4(a, dgrad, bytes, flops, s);
    if (version == 3) return papr_launch_chain3(a, dgrad, bytes, flops, s);
    if (version == 2) return papr_launch_chain2(a, dgrad, bytes, flops, s);
    PAPR_REQUIRE(a.n_layers >= 1 && a.n_layers <= CHAIN_MAX_LAYERS, "mlp_chain: %d layers", a.n_layers);
    PAPR_REQUIRE(a.K0 % 4 == 0 && a.lda0 % 4 == 0 && a.K0 <= 256, "mlp_chain: input width %d", a.K0);
    if (a.M <= 0) return 0;
    const int tiles_m = (int)((a.M + CH_BM - 1) / CH_BM);
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_LDS_BYTES);
        attr_set = true;
    }
    static const int wgs_per_cu = getenv("PAPR_CHAIN_WGS_PER_CU") ? atoi(getenv("PAPR_CHAIN_WGS_PER_CU")) : CH_WGS_PER_CU;   // (A/B switch)
    dim3 grid((unsigned)(tiles_m < wgs_per_cu * n_cu ? tiles_m : wgs_per_cu * n_cu));     // two workgroups per CU: one multiplies while the other is in its row phases
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin2(dgrad ? 10 : 9, a.M, a.n_layers, a.K0, bytes, flops, s);
    // a data-gradient run reads sign words when every masked layer has them, fp32 activation rows otherwise
    bool bits = dgrad;
    for (int l = 0; l < a.n_layers; ++l) bits = bits && (a.L[l].mask == nullptr || a.L[l].sign_bits != nullptr);
    if (dgrad && bits) mlp_chain_kernel<true, true><<<grid, dim3(CH_THREADS), CH_LDS_BYTES, s>>>(a, tiles_m);
    else if (dgrad) mlp_chain_kernel<true, false><<<grid, dim3(CH_THREADS), CH_LDS_BYTES, s>>>(a, tiles_m);
    else {
        bool skip = false;
        for (int l = 0; l < a.n_layers; ++l) skip = skip || a.L[l].k1steps < a.L[l].ksteps;
        if (skip) mlp_chain_kernel<false, false, true><<<grid, dim3(CH_THREADS), CH_LDS_BYTES, s>>>(a, tiles_m);
        else mlp_chain_kernel<false, false><<<grid, dim3(CH_THREADS), CH_LDS_BYTES, s>>>(a, tiles_m);
    }
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("mlp_chain");
    return 0;
}
