// K3e: fused layer runs, fourth structure -- the row phases run on the accumulators' own layout.
//
// Replaces the same reference lines as chain3.hip (MLP.forward, models/mlp.py:47-59, with skip_layers :30-31,54-55, and its autograd
// data-gradient).  chain3.hip left the kernel instruction-issue bound in its row phases (profiles/r02_pmc_sq_insts: ~14 vector / scalar /
// LDS instructions per matrix instruction): every slot dumped the accumulators to LDS, read them back row-per-wave, found the row
// maximum with 24 DPP steps per four rows, moved scales through the scalar unit and kept a 1 / scale table.  Here:
//
//   * NO DUMP.  The matrix instruction takes the weight fragment as its row operand, so the accumulators hold C^T: a lane owns ONE
//     row of its 32-row tile and 16 of the wave's 32 columns.  The weight fragments are laid out (split_weight_batch_kernel, perm = 1)
//     so that those 16 are CONSECUTIVE columns: lane (row, h) of wave w holds columns 32 w + 16 h .. + 15, register 4 g + c = column
//     32 w + 16 h + 4 g + c.  Bias, activation, sign bits, row stores (four 16-byte stores = 64 contiguous bytes per lane, a full
//     128-byte line per lane pair), the split into f16 planes (two 16-byte LDS writes per plane and 32-row tile) all happen on the
//     accumulator registers.
//   * ROW MAXIMUM BY ONE EXCHANGE.  A lane reduces its 16 values with v_max3, meets its partner lane (the row's other 16 columns of
//     this wave) with one v_permlane32_swap, and leaves ONE float per wave and row in LDS; after the slot's barrier every lane reads
//     the eight partial maxima of its row.  No DPP chains, no scalar-unit round trip, no 1 / scale table walk.
//   * TWO ACCUMULATOR SETS, ONE BARRIER PER SLOT.  Slot (T, l): multiply tile T by layer l [K], first half of T's row phase [P1: bias,
//     activation, stores, sign word, partial maxima] -- and the second half of the OTHER tile's row phase for the layer it finished a
//     slot ago [P2: row maximum -> scale -> split -> planes], which needs that tile's post-activation values: they simply stay in its
//     accumulator registers.  Waves 0-3 run K, P1, P2, waves 4-7 (their SIMD partners) P2, K, P1: each SIMD has one wave on the
//     matrix pipe and one on the vector pipe most of the time.  chain3.hip: barrier, dump, barrier per slot.
//   * Sign words: one 32-bit word per lane and 64-row tile (bit 31 - (16 i + e) = value e of 32-row tile i), 256 contiguous bytes per
//     wave; the data-gradient run's lanes hold the same values of the same rows.
//   * LayerNorm core behind the run: wave-local (mean, M2) of the row's 32 columns through the same exchange, combined with the
//     pairwise (Chan) formula -- one exchange instead of the two a two-pass mean / variance would need.
//   * Skip layers ([previous output | x] as input, models/mlp.py:54-55) ride as a second K segment: once the first segment has been
//     multiplied the tile's planes are dead, the run's input rows are staged into them again (split with the scale the accumulators
//     already carry: the previous layer's P2 chose it from max(row max of its output, row max of x)) and the k-loop goes on into
//     the same accumulators.
//
// Arithmetic per output element is chain3.hip's (same products in the same order per accumulator, same power-of-two row scales, same
// split), so every stored value is bit-identical to chain.hip / chain3.hip except behind a LayerNorm core.
// Built like chain3.hip: no packed-fp32 VALU, weights in a[0:127] by name (papr_amd/build.py).
#include "papr_common.h"
#include "h3_common.h"
#include "chain.h"
#include "chain4_kloop.inc"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int NI = 2;                           // a multiplying wave: 64 rows x 32 columns
constexpr int GW = 8;                           // waves
constexpr int RB = 8;                           // rows per block of the planes (the staging wave's rows)
constexpr int C4_THREADS = GW * 64;
constexpr int C4_ROWS = 64;                     // rows per tile
constexpr int C4_TILE_BYTES = 65536;            // A planes of one tile: 8 blocks of RB rows
constexpr int C4_BLK_BYTES = RB * 1024;         // one block: hi rows (RB x 512 B) | lo rows (RB x 512 B)
constexpr int C4_LO = RB * 512;
constexpr int KS = 16;                          // k-steps of a 256-wide layer
// LDS: planes of X and Y | 1 / scale of every plane row [2][64] | partial row maxima [2][8 waves][64] | (mean, M2) partials [2][8][64] | bias [8 layers][256]
constexpr int C4_OFF_INV = 2 * C4_TILE_BYTES;
constexpr int C4_OFF_PMAX = C4_OFF_INV + 2 * C4_ROWS * 4;
constexpr int C4_OFF_NRM = C4_OFF_PMAX + 2 * GW * C4_ROWS * 4;
constexpr int C4_OFF_BIAS = C4_OFF_NRM + 2 * GW * C4_ROWS * 8;
constexpr int C4_OFF_XMAX = C4_OFF_BIAS + CHAIN_MAX_LAYERS * 256 * 4;        // max |.| of the staged input rows [2][64] (skip layers)
constexpr size_t C4_LDS_BYTES = C4_OFF_XMAX + 2 * C4_ROWS * 4;

__device__ __forceinline__ float scale_from_max(unsigned bits, float& inv) {      // row max -> [2^13, 2^14)
    const int ea = bits ? (int)((bits >> 23) & 0xff) : 127 + 13;
    inv = pow2_from_biased(127 - 13 + (ea - 127));
    return pow2_from_biased(127 + 13 - (ea - 127));
}

#define C4_DPP4(ctrl)                                                  \
    "v_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\t" \
    "v_max_f32_dpp %2, %2, %2 " ctrl "\n\tv_max_f32_dpp %3, %3, %3 " ctrl "\n\t"
__device__ __forceinline__ void wave_max4(float& a, float& b, float& c, float& d) {      // (staging only: a row across the wave; result in lane 63)
    asm("s_nop 1\n\t" C4_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") C4_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        C4_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf") C4_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
        C4_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf") C4_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ float last_lane(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
#define put4(vec, l0, s0, s1, s2, s3)                                                                                             \
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %0, %3, %7\n\tv_writelane_b32 %0, %4, %8" \
        : "+v"(vec) : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "i"(l0), "i"((l0) + 1), "i"((l0) + 2), "i"((l0) + 3))

// both lanes of a pair (l, l ^ 32) receive op(own, partner's) -- lanes 0-31 hold the row's columns 0-15 of the wave, lanes 32-63 columns 16-31.
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second.  Through the builtin, NOT inline
// asm: the instruction needs wait states behind the VALU writes of its operands, which the compiler only counts for instructions it
// knows (the first version, as asm right behind the two copies, read stale registers for some rows)
__device__ __forceinline__ void pair_swap(float& lo_copy, float& hi_copy) {      // in: both = v; out: lo_copy = v of the pair's lower lane, hi_copy = of its upper lane, in both lanes
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_copy), __float_as_uint(hi_copy), false, false);
    lo_copy = __uint_as_float(r[0]);
    hi_copy = __uint_as_float(r[1]);
}
__device__ __forceinline__ float pair_max(float v) { float a = v, b = v; pair_swap(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float pair_sum(float v) { float a = v, b = v; pair_swap(a, b); return a + b; }

// uniform per-layer flags of the row phases: 0 / 1 = known at compile time (the hot instantiations), 2 = look at run time
template <int RELU, int STORE, int BITS, int FULL>
struct P1Cfg { static constexpr int relu = RELU, store = STORE, bits = BITS, full = FULL; };
template <int RMAX, int MORE, int FULL, int ROWS>
struct P2Cfg { static constexpr int rmax = RMAX, more = MORE, full = FULL, rows = ROWS; };

#define C4_WH0 "a[0:3]"
#define C4_WL0 "a[4:7]"
#define C4_CH0 "a0", "a1", "a2", "a3"
#define C4_WCLOB0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7"
#define C4_WH1 "a[8:11]"
#define C4_WL1 "a[12:15]"
#define C4_CH1 "a8", "a9", "a10", "a11"
#define C4_WCLOB1 "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define C4_WH2 "a[16:19]"
#define C4_WL2 "a[20:23]"
#define C4_CH2 "a16", "a17", "a18", "a19"
#define C4_WCLOB2 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23"
#define C4_WH3 "a[24:27]"
#define C4_WL3 "a[28:31]"
#define C4_CH3 "a24", "a25", "a26", "a27"
#define C4_WCLOB3 "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
#define C4_WH4 "a[32:35]"
#define C4_WL4 "a[36:39]"
#define C4_CH4 "a32", "a33", "a34", "a35"
#define C4_WCLOB4 "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39"
#define C4_WH5 "a[40:43]"
#define C4_WL5 "a[44:47]"
#define C4_CH5 "a40", "a41", "a42", "a43"
#define C4_WCLOB5 "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47"
#define C4_WH6 "a[48:51]"
#define C4_WL6 "a[52:55]"
#define C4_CH6 "a48", "a49", "a50", "a51"
#define C4_WCLOB6 "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55"
#define C4_WH7 "a[56:59]"
#define C4_WL7 "a[60:63]"
#define C4_CH7 "a56", "a57", "a58", "a59"
#define C4_WCLOB7 "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define C4_WH8 "a[64:67]"
#define C4_WL8 "a[68:71]"
#define C4_CH8 "a64", "a65", "a66", "a67"
#define C4_WCLOB8 "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71"
#define C4_WH9 "a[72:75]"
#define C4_WL9 "a[76:79]"
#define C4_CH9 "a72", "a73", "a74", "a75"
#define C4_WCLOB9 "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79"
#define C4_WH10 "a[80:83]"
#define C4_WL10 "a[84:87]"
#define C4_CH10 "a80", "a81", "a82", "a83"
#define C4_WCLOB10 "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87"
#define C4_WH11 "a[88:91]"
#define C4_WL11 "a[92:95]"
#define C4_CH11 "a88", "a89", "a90", "a91"
#define C4_WCLOB11 "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#define C4_WH12 "a[96:99]"
#define C4_WL12 "a[100:103]"
#define C4_CH12 "a96", "a97", "a98", "a99"
#define C4_WCLOB12 "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103"
#define C4_WH13 "a[104:107]"
#define C4_WL13 "a[108:111]"
#define C4_CH13 "a104", "a105", "a106", "a107"
#define C4_WCLOB13 "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111"
#define C4_WH14 "a[112:115]"
#define C4_WL14 "a[116:119]"
#define C4_CH14 "a112", "a113", "a114", "a115"
#define C4_WCLOB14 "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119"
#define C4_WH15 "a[120:123]"
#define C4_WL15 "a[124:127]"
#define C4_CH15 "a120", "a121", "a122", "a123"
#define C4_WCLOB15 "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define C4_CAT_(a, b) a##b
#define C4_CAT(a, b) C4_CAT_(a, b)
#define C4_OFF0 "0"
#define C4_OFF1 "1024"
#define C4_OFF2 "2048"
#define C4_OFF3 "3072"
#define C4_WLOAD(ks, q, bh, bl)                                                                                                 \
    do {                                                                                                                        \
        if constexpr (ONE)                                                                                                      \
            asm volatile("global_load_dwordx4 " C4_CAT(C4_WH, ks) ", %0, %1 offset:" C4_CAT(C4_OFF, q)                               \
                         : : "v"(w_lane), "s"((bh) + ((ks) >> 2) * 4096) : C4_CAT(C4_CH, ks), "memory");                        \
        else                                                                                                                    \
            asm volatile("global_load_dwordx4 " C4_CAT(C4_WH, ks) ", %0, %1 offset:" C4_CAT(C4_OFF, q) "\n\t"                      \
                         "global_load_dwordx4 " C4_CAT(C4_WL, ks) ", %0, %2 offset:" C4_CAT(C4_OFF, q)                               \
                         : : "v"(w_lane), "s"((bh) + ((ks) >> 2) * 4096), "s"((bl) + ((ks) >> 2) * 4096) : C4_CAT(C4_WCLOB, ks), "memory"); \
    } while (0)
#define C4_WLOAD_IF(ks, q, n, bh, bl) if ((ks) < (n)) C4_WLOAD(ks, q, bh, bl)
#define C4_WLOAD_ALL(n, bh, bl)                                                                                                 \
    C4_WLOAD_IF(0, 0, n, bh, bl); C4_WLOAD_IF(1, 1, n, bh, bl); C4_WLOAD_IF(2, 2, n, bh, bl); C4_WLOAD_IF(3, 3, n, bh, bl);     \
    C4_WLOAD_IF(4, 0, n, bh, bl); C4_WLOAD_IF(5, 1, n, bh, bl); C4_WLOAD_IF(6, 2, n, bh, bl); C4_WLOAD_IF(7, 3, n, bh, bl);     \
    C4_WLOAD_IF(8, 0, n, bh, bl); C4_WLOAD_IF(9, 1, n, bh, bl); C4_WLOAD_IF(10, 2, n, bh, bl); C4_WLOAD_IF(11, 3, n, bh, bl);   \
    C4_WLOAD_IF(12, 0, n, bh, bl); C4_WLOAD_IF(13, 1, n, bh, bl); C4_WLOAD_IF(14, 2, n, bh, bl); C4_WLOAD_IF(15, 3, n, bh, bl)

// The run as the kernel walks it: a layer is one step, a skip layer two (its second K segment multiplies the run's input rows again).
constexpr int C4_MAX_STEPS = 2 * CHAIN_MAX_LAYERS;
constexpr int C4_FIRST = 1;                     // the step starts a layer: accumulators from zero
constexpr int C4_LAST = 2;                      // the step ends a layer: row phases follow
constexpr int C4_RESTAGE = 4;                   // before the step the run's input rows are staged into the tile's planes again (second K segment of a skip layer)
struct C4Step { int layer, kbeg, kcnt, flags; };
struct C4Plan {                                 // (one packed word per step: the kernel reads them with scalar loads, which have no byte form)
    int n_steps; int w[C4_MAX_STEPS];
    void push(int layer, int kbeg, int kcnt, int flags) { w[n_steps++] = layer | (kbeg << 8) | (kcnt << 16) | (flags << 24); }
};

#ifdef PAPR_C4_TRACE                                // cycle stamps of one workgroup: 8 waves x 256 stamps (scripts/probes/chain4_trace.py)
__device__ long long g_chain4_trace[2048];
#define C4_STAMP() do { asm volatile("" ::: "memory"); if (blockIdx.x == 100 && lane0 == 0 && trace_slot < 256) g_chain4_trace[wn * 256 + trace_slot++] = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); } while (0)
#else
#define C4_STAMP() do {} while (0)
#endif

__device__ __forceinline__ long uniform64(long v) {       // a wave-uniform value the compiler keeps in scalar registers and does not move out of loops
    int lo = __builtin_amdgcn_readfirstlane((int)v), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

// ONE: the reduced-precision mode (one f16 product per fp32 product, hi planes only; the counterpart of the reference's fp16
// autocast, models/attn.py:248)
template <bool DGRAD, bool ONE>
__global__ __launch_bounds__(C4_THREADS, 2) void mlp_chain4_kernel(ChainArgs p, C4Plan plan, int iters, int generic_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool k_first = wn < 4;                    // waves w and w + 4 share a SIMD
#ifdef PAPR_C4_TRACE
    int trace_slot = 0;
#endif
    // Every piece below derives its lane addresses from its own opaque copy of the lane number: values the compiler can compute
    // once ahead of the slot loop live across everything, and with two accumulator sets there is nothing to spare.
#define C4_LANE() int lane = lane0; asm volatile("" : "+v"(lane))
    // LDS layouts (bytes inside a tile's planes): row r of the tile lives in block r / RB at row r % RB (hi rows | lo rows of the
    // block); its 16-byte chunks (8 halfs = 8 columns) are XOR-ed with r & 15.  A fragment of 32-row tile i, k-step ks: row
    // 32 i + (lane & 31), chunk 2 ks + (lane >> 5).
    float* const inv_all = reinterpret_cast<float*>(smem + C4_OFF_INV);         // [2][64]
    float* const pmax_all = reinterpret_cast<float*>(smem + C4_OFF_PMAX);       // [2][8][64]
    float2* const nrm_all = reinterpret_cast<float2*>(smem + C4_OFF_NRM);       // [2][8][64]
    float* const bias_all = reinterpret_cast<float*>(smem + C4_OFF_BIAS);       // [layers][256]
    float* const xmax_all = reinterpret_cast<float*>(smem + C4_OFF_XMAX);       // [2][64]
    const int n_layers = p.n_layers;
    auto plan_step = [&](int i) __attribute__((always_inline)) {
        const int v = __builtin_amdgcn_readfirstlane(plan.w[i]);
        return C4Step{v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff, (v >> 24) & 0xff};
    };

    // ---- this wave's slice of a layer: fragment (n-tile t, k-step s) starts at ((t * ksteps + s) * 64 + lane) * 8 halfs.
    // The fragments live in a[0:127] BY NAME (chain3.hip: why): k-step j of the current step: hi = a[8 j : 8 j + 3], lo = a[8 j + 4 : 8 j + 7].
    auto frag_base = [&](const ChainLayer& L, const _Float16* w, int kbeg) __attribute__((always_inline)) {
        const int t = 32 * wn < L.N ? wn : 0;       // (a wave without columns in this layer: any valid address)
        return reinterpret_cast<const char*>(w) + (size_t)(t * L.ksteps + kbeg) * 1024;
    };

    // ---- multiply the tile in `planes` by step st (accumulators from zero or continued); with a next step sn each k-step's
    // registers are refilled with that step's fragment as soon as the k-step is done
    auto k_run = [&](const char* planes, C4Step st, bool refill, C4Step sn, f32x16 (&acc)[NI]) __attribute__((always_inline)) {
        const ChainLayer& L = p.L[st.layer];
        const int ksteps = st.kcnt;
        const bool live = 32 * wn < L.N;
        const char *nh = nullptr, *nl = nullptr;
        int nks = 0;
        if (refill) { const ChainLayer& Ln = p.L[sn.layer]; nh = frag_base(Ln, Ln.w_hi, sn.kbeg); nl = frag_base(Ln, Ln.w_lo, sn.kbeg); nks = sn.kcnt; }
        const bool first = (st.flags & C4_FIRST) != 0;
        C4_LANE();
        const int arow = lane & 31, ax = arow & 15;
        const unsigned ab = (unsigned)((arow / RB) * C4_BLK_BYTES + (arow % RB) * 512 + (((lane >> 5) ^ (ax & 1)) * 16));
        const unsigned axr = (unsigned)((ax & ~1) * 16);
        const unsigned w_lane = (unsigned)lane * 16u;   // (weight loads: a wave-uniform base + this lane offset)
        // (this step's fragments were requested a slot ago and waited for in `pending` -- ahead of the row stores: a wait here would
        // also wait for those, loads and stores share the counter)
        if (live && ksteps == KS && first) {
            // ---- the hot form: ONE asm statement for the whole k-loop (chain4_kloop.inc, scripts/gen_chain4_kloop.py)
            const unsigned pb = (unsigned)(size_t)planes + ab;
            unsigned ad[8];                         // LDS address of k-step j's fragments (k-step j + 8: + 256)
#pragma unroll
            for (int j = 0; j < 8; ++j) ad[j] = pb + (((unsigned)j * 32u) ^ axr);
            half8 f00, f01, f02, f03, f10, f11, f12, f13;       // fragment buffers (asm temporaries)
#define C4_KLOOP_OPERANDS                                                                                                       \
            [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [f00] "=&v"(f00), [f01] "=&v"(f01), [f02] "=&v"(f02), [f03] "=&v"(f03),         \
            [f10] "=&v"(f10), [f11] "=&v"(f11), [f12] "=&v"(f12), [f13] "=&v"(f13)                                              \
            : [ad0] "v"(ad[0]), [ad1] "v"(ad[1]), [ad2] "v"(ad[2]), [ad3] "v"(ad[3]), [ad4] "v"(ad[4]), [ad5] "v"(ad[5]),       \
              [ad6] "v"(ad[6]), [ad7] "v"(ad[7]), [wv] "v"(w_lane), [bh0] "s"(nh), [bh1] "s"(nh + 4096), [bh2] "s"(nh + 8192),  \
              [bh3] "s"(nh + 12288), [bl0] "s"(nl), [bl1] "s"(nl + 4096), [bl2] "s"(nl + 8192), [bl3] "s"(nl + 12288)
            if (nks == KS) {
                if constexpr (ONE) asm volatile(C4_KLOOP1_LD : C4_KLOOP_OPERANDS : C4_KLOOP_AGPRS, "memory");
                else asm volatile(C4_KLOOP3_LD : C4_KLOOP_OPERANDS : C4_KLOOP_AGPRS, "memory");
            } else {
                if constexpr (ONE) asm volatile(C4_KLOOP1_NL : C4_KLOOP_OPERANDS : "memory");
                else asm volatile(C4_KLOOP3_NL : C4_KLOOP_OPERANDS : "memory");
                C4_WLOAD_ALL(nks, nh, nl);          // (a narrower next step: its fragments in a bunch)
            }
#undef C4_KLOOP_OPERANDS
        } else {
            // ---- any other width / a continued accumulation: the plain form (the compiler places the LDS reads and their waits)
            if (first) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            }
            half8 ah[2][NI], al[2][NI];
            auto load_a = [&](int ks, half8 (&qh)[NI], half8 (&ql)[NI]) {
                ks = ks < ksteps ? ks : ksteps - 1;
                const unsigned o = ab + (((unsigned)ks * 32u) ^ axr);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    qh[i] = *reinterpret_cast<const half8*>(planes + i * 32768 + o);
                    if constexpr (!ONE) ql[i] = *reinterpret_cast<const half8*>(planes + i * 32768 + o + C4_LO);
                }
            };
            if (live) load_a(0, ah[0], al[0]);
            // six matrix instructions of a k-step: hi.lo, lo.hi, hi.hi for both row tiles (the order of chain.hip per accumulator);
            // then the registers of the k-step take the next step's fragment
#define C4_KSTEP(ks, q)                                                                                                         \
            if (live && (ks) < ksteps) {                                                                                        \
                load_a((ks) + 1, ah[((ks) + 1) & 1], al[((ks) + 1) & 1]);                                                       \
                if constexpr (ONE)                                                                                              \
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, " C4_CAT(C4_WH, ks) ", %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, " C4_CAT(C4_WH, ks) ", %3, %1" \
                                 : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[(ks) & 1][0]), "v"(ah[(ks) & 1][1]));                    \
                else                                                                                                            \
                asm volatile("v_mfma_f32_32x32x16_f16 %0, " C4_CAT(C4_WH, ks) ", %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, " C4_CAT(C4_WH, ks) ", %5, %1\n\t" \
                             "v_mfma_f32_32x32x16_f16 %0, " C4_CAT(C4_WL, ks) ", %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, " C4_CAT(C4_WL, ks) ", %3, %1\n\t" \
                             "v_mfma_f32_32x32x16_f16 %0, " C4_CAT(C4_WH, ks) ", %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, " C4_CAT(C4_WH, ks) ", %3, %1"       \
                             : "+v"(acc[0]), "+v"(acc[1])                                                                       \
                             : "v"(ah[(ks) & 1][0]), "v"(ah[(ks) & 1][1]), "v"(al[(ks) & 1][0]), "v"(al[(ks) & 1][1]));        \
            }                                                                                                                   \
            C4_WLOAD_IF(ks, q, nks, nh, nl)
            asm volatile("s_nop 1" ::: "memory");   // (the zeroed accumulators: VALU write -> matrix read)
            C4_KSTEP(0, 0); C4_KSTEP(1, 1); C4_KSTEP(2, 2); C4_KSTEP(3, 3); C4_KSTEP(4, 0); C4_KSTEP(5, 1); C4_KSTEP(6, 2); C4_KSTEP(7, 3);
            C4_KSTEP(8, 0); C4_KSTEP(9, 1); C4_KSTEP(10, 2); C4_KSTEP(11, 3); C4_KSTEP(12, 0); C4_KSTEP(13, 1); C4_KSTEP(14, 2); C4_KSTEP(15, 3);
#undef C4_KSTEP
            // the last results leave the matrix pipe 16 passes after issue; hipcc does not count wait states behind inline asm
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
        }
    };

    // ---- split a row held across the wave (lane: 4 columns) into the A planes: block wn, row u (staging)
    auto write_planes = [&](char* planes, unsigned wp, unsigned wq, int u, const float4& v, float sc, bool in_k, _Float16* gdst = nullptr) __attribute__((always_inline)) {
        if (in_k) {
            char* dst = planes + wn * C4_BLK_BYTES + u * 512 + (wp ^ (unsigned)(((wn * RB + u) & 15) * 16)) + wq;
            if constexpr (ONE) {                    // hi = f16(v * s) only
                unsigned h01, h23;
                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(v.x), "v"(sc));
                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(v.z), "v"(sc));
                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v.y), "v"(sc));
                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v.w), "v"(sc));
                *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
                if (gdst) *reinterpret_cast<uint2*>(gdst) = make_uint2(h01, h23);
            } else {
                half4 hi, lo;
                split4(v, sc, hi, lo);
                *reinterpret_cast<half4*>(dst) = hi;
                *reinterpret_cast<half4*>(dst + C4_LO) = lo;
            }
        }
    };

    // ---- stage the run's input rows of a tile (coalesced: one row per load instruction; this wave: rows RB wn .. RB wn + 7):
    // LayerNorm core in front of the run, row maxima, scales, split.  again: the second K segment of a skip layer -- the rows are
    // split with the scale the tile's accumulators already carry (inv_tab), nothing else is written.
    auto stage = [&](char* planes, float* inv_tab, float* xmax_tab, long m0, bool again, int kpad) __attribute__((always_inline)) {
        const int M32 = (int)p.M;
        int r0 = __builtin_amdgcn_readfirstlane((int)m0 + wn * RB);
        C4_LANE();
        const int c = 4 * lane;
        const unsigned wp = (unsigned)(lane >> 1) * 16u, wq = (unsigned)(lane & 1) * 8u;
        asm volatile("" : "+s"(r0));
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int m = r0 + q;
            m = m < M32 ? m : M32 - 1;              // rows beyond M: the last row again
            const float* rowp = p.A0 + (long)m * p.lda0;  // wave-uniform: scalar base + one lane offset
            v[q] = c < p.K0 ? *reinterpret_cast<const float4*>(rowp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (again) {
            // (with in_norm_stats the first staging wrote the standardised rows back: these ARE the rows the first layer multiplied)
            const float4 i0 = *reinterpret_cast<const float4*>(inv_tab + wn * RB), i1 = *reinterpret_cast<const float4*>(inv_tab + wn * RB + 4);
            const float iv[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float sc = __uint_as_float((254u << 23) - __float_as_uint(iv[q]));       // 1 / (a power of two)
                write_planes(planes, wp, wq, q, v[q], sc, c < kpad);
            }
            return;
        }
        if (!DGRAD && p.in_norm_stats != nullptr) {
            // LayerNorm core in front of the run (FeedForward.innorm): the wave holds the whole row
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int mrow = r0 + q;
                const int wdt = p.in_norm_width;
                const bool i0 = c < wdt, i1 = c + 1 < wdt, i2 = c + 2 < wdt, i3 = c + 3 < wdt;
                const float mean = wave_sum(((i0 ? v[q].x : 0.f) + (i1 ? v[q].y : 0.f)) + ((i2 ? v[q].z : 0.f) + (i3 ? v[q].w : 0.f))) / (float)wdt;
                float4 dl = make_float4(i0 ? v[q].x - mean : 0.f, i1 ? v[q].y - mean : 0.f, i2 ? v[q].z - mean : 0.f, i3 ? v[q].w - mean : 0.f);
                const float sigma = sqrtf(wave_sum((dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w)) / (float)(wdt - 1));
                const float rinv = 1.0f / (sigma + p.in_norm_eps);
                v[q] = make_float4(dl.x * rinv, dl.y * rinv, dl.z * rinv, dl.w * rinv);
                if (mrow < M32) {
                    if (p.in_norm_writeback && c < p.K0) *reinterpret_cast<float4*>(p.A0 + (long)mrow * p.lda0 + c) = v[q];
                    if (lane == 0) { p.in_norm_stats[(long)mrow * 2] = rinv; p.in_norm_stats[(long)mrow * 2 + 1] = sigma; }
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
            if (p.rowmax0 && lane < 4 && r0 + h + lane < M32) p.rowmax0[r0 + h + lane] = mx4;
            float inv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sc = scale_from_max(__float_as_uint(smx[q]), inv[q]);
                _Float16* gdst = nullptr;
                if (ONE && p.a0_half != nullptr && r0 + h + q < M32) gdst = p.a0_half + (long)(r0 + h + q) * p.lda0_half + c;
                write_planes(planes, wp, wq, h + q, v[h + q], sc, c < kpad, gdst);
            }
            *reinterpret_cast<float4*>(inv_tab + wn * RB + h) = make_float4(inv[0], inv[1], inv[2], inv[3]);
            *reinterpret_cast<float4*>(xmax_tab + wn * RB + h) = make_float4(smx[0], smx[1], smx[2], smx[3]);
        }
    };

    // ---- P1: first half of the row phase of layer l for tile h at rows m0, on the accumulators: bias / activation (forward) or
    // 1 / scale and the activation derivative (data-gradient), row stores, sign word, partial row maxima (or LayerNorm partials).
    // The post-activation values stay in `a` for P2.
    auto p1_run = [&](f32x16 (&a)[NI], int h, long m0, int l, unsigned sw_in) __attribute__((always_inline)) {
        l = __builtin_amdgcn_readfirstlane(l);
        const int M32 = (int)p.M;
        int t0 = __builtin_amdgcn_readfirstlane((int)m0);
        C4_LANE();
        asm volatile("" : "+s"(l), "+s"(t0));
        const ChainLayer& L = p.L[l];
        const int N = L.N;
        const int arow = lane & 31, hh = lane >> 5;
        float* const pm = pmax_all + (h * GW + wn) * C4_ROWS;
        const bool live = 32 * wn < N;
        const bool rt_norm = !DGRAD && l + 1 == n_layers && p.norm_stats != nullptr;
        if (!live) {                                // a wave without columns in this layer: its partial maxima are zero
            if (hh == 0) { pm[arow] = 0.f; pm[32 + arow] = 0.f; }
            return;
        }
        const int cb = 32 * wn + 16 * hh;           // this lane's 16 columns
        const float slope = L.act == PAPR_ACT_RELU ? 0.f : (L.act == PAPR_ACT_LEAKY_RELU ? 0.2f : 1.f);
        const bool rt_half = ONE && L.c_half != 0 && L.C != nullptr;          // f16 rows out (written by P2 with the split), no fp32 rows
        const bool mask_rows = DGRAD && L.sign_bits == nullptr && L.mask != nullptr;
        const bool rt_store = L.C != nullptr && !rt_half && !rt_norm && N % 32 != 0, rt_bits = L.sign_bits != nullptr || mask_rows;       // (fp32 rows of a width that is a multiple of 32 leave in P2, whole cache lines at a time)
        const bool rt_full = N == 256 && t0 + C4_ROWS <= M32;
        const float* const inv_tab = inv_all + h * C4_ROWS;
        const float* const bias = bias_all + l * 256 + cb;
        const long ldc = L.ldc;
        auto rows = [&](auto cfg) {
            using Cfg = decltype(cfg);
            const bool f_store = Cfg::store == 2 ? rt_store : Cfg::store == 1;
            const bool f_bits = Cfg::bits == 2 ? rt_bits : Cfg::bits == 1;
            const bool f_full = Cfg::full == 2 ? rt_full : Cfg::full == 1;
            const bool f_relu = Cfg::relu == 2 ? false : Cfg::relu == 1;       // (2: the slope is looked at at run time)
            unsigned word = DGRAD ? sw_in : 0u;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rt = 32 * i + arow, m = t0 + rt;
                const bool in_m = f_full || m < M32;
                const float inv = inv_tab[rt];
                float lmax = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool col_ok = f_full || cb + 4 * g < N;
                    float y[4];
                    if (!DGRAD) {
                        // acc * inv is exact (a power of two): fma(acc, inv, bias) = the separate multiply and add, bit for bit;
                        // activation as max(y, slope y + 0): slope 0 -> ReLU (+0 for negative y), 0.2 -> LeakyReLU, 1 -> none
                        const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * g);
                        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float pre = __builtin_fmaf(a[i][4 * g + c], inv, bb[c]);
                            y[c] = f_relu ? fmaxf(pre, 0.f) : fmaxf(pre, __builtin_fmaf(pre, slope, 0.f));
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float gv = a[i][4 * g + c] * inv;
                            if (f_bits) {
                                const int bit = (int)(word << (16 * i + 4 * g + c));     // (bit 31 - n of the word = value n, first value in the top bit)
                                if (f_relu) y[c] = __uint_as_float(__float_as_uint(gv) & (unsigned)(bit >> 31));
                                else y[c] = bit < 0 ? gv : gv * slope;
                            } else y[c] = gv;
                        }
                    }
                    if (!f_full && !col_ok) { y[0] = 0.f; y[1] = 0.f; y[2] = 0.f; y[3] = 0.f; }       // columns beyond N: zero weights, but a bias-free zero all the same
#ifndef C4_X_NOSTORE                                // (timing experiments: pieces left out, results wrong)
                    if (f_store && col_ok && in_m) *reinterpret_cast<float4*>(L.C + (long)m * ldc + cb + 4 * g) = make_float4(y[0], y[1], y[2], y[3]);
#endif
#ifndef C4_X_NOBITS
                    if (!DGRAD && f_bits) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) word = (word << 1) | (y[c] > 0.f ? 1u : 0u);
                    }
#endif
                    lmax = fmaxf(fmaxf(lmax, fabsf(y[0])), fmaxf(fabsf(y[1]), fmaxf(fabsf(y[2]), fabsf(y[3]))));
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[i][4 * g + c] = y[c];
                }
                if (!rt_norm) {
                    lmax = pair_max(lmax);
                    if (hh == 0) pm[rt] = lmax;
                } else {
                    // LayerNorm core behind the run (FeedForward.outnorm, act = none): this wave's 32 columns of the row -> (mean, M2),
                    // sums in a fixed order; P2 combines the waves
                    float s = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) s += (a[i][4 * g] + a[i][4 * g + 1]) + (a[i][4 * g + 2] + a[i][4 * g + 3]);
                    const float mean_w = pair_sum(s) * (1.f / 32.f);
                    float q = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float d0 = a[i][4 * g] - mean_w, d1 = a[i][4 * g + 1] - mean_w, d2 = a[i][4 * g + 2] - mean_w, d3 = a[i][4 * g + 3] - mean_w;
                        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    }
                    q = pair_sum(q);
                    if (hh == 0) nrm_all[(h * GW + wn) * C4_ROWS + rt] = make_float2(mean_w, q);
                }
            }
            if (!DGRAD && f_bits && (f_full || t0 < M32)) L.sign_bits[(long)(t0 / C4_ROWS) * (GW * 64) + wn * 64 + lane] = word;
        };
        if (mask_rows) {
            // no sign words from a fused forward run: form this lane's word from the fp32 activation rows (one 16-byte load in
            // flight at a time -- the rare path must not cost the hot ones registers)
            sw_in = 0u;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int m = t0 + 32 * i + arow, row = m < M32 ? m : M32 - 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cb + 4 * g < N) a4 = *reinterpret_cast<const float4*>(L.mask + (long)row * L.ld_mask + cb + 4 * g);
                    asm volatile("" : "+v"(a4.x), "+v"(a4.y), "+v"(a4.z), "+v"(a4.w) :: "memory");
                    sw_in = (sw_in << 4) | (a4.x > 0.f ? 8u : 0u) | (a4.y > 0.f ? 4u : 0u) | (a4.z > 0.f ? 2u : 0u) | (a4.w > 0.f ? 1u : 0u);
                }
            }
        }
        const bool relu = L.act == PAPR_ACT_RELU;
        // hot combinations (everything 256 wide, tile inside M, no norm): training / inference / data-gradient, ReLU or any slope
        if (generic_only || !rt_full || rt_norm) rows(P1Cfg<2, 2, 2, 2>());
        else if (!rt_store && rt_bits) { if (relu) rows(P1Cfg<1, 0, 1, 1>()); else rows(P1Cfg<2, 0, 1, 1>()); }
        else if (!DGRAD && !rt_store && !rt_bits) { if (relu) rows(P1Cfg<1, 0, 0, 1>()); else rows(P1Cfg<2, 0, 0, 1>()); }
        else rows(P1Cfg<2, 2, 2, 2>());
    };

    // ---- P2: second half of the row phase (a slot later, behind the barrier): row maximum from the eight partial maxima, scale,
    // split of the values still in `a` into the tile's planes = the next layer's input; or the LayerNorm core's second half.
    // xmax: the next layer is a skip layer -- the scale must cover the run's input rows as well (their maxima: xmax_tab)
    auto p2_run = [&](f32x16 (&a)[NI], int h, long m0, int l) __attribute__((always_inline)) {
        l = __builtin_amdgcn_readfirstlane(l);
        const int M32 = (int)p.M;
        int t0 = __builtin_amdgcn_readfirstlane((int)m0);
        C4_LANE();
        asm volatile("" : "+s"(l), "+s"(t0));
        const ChainLayer& L = p.L[l];
        const int N = L.N;
        const int arow = lane & 31, hh = lane >> 5;
        const bool more = l + 1 < n_layers;
        const bool rt_norm = !DGRAD && !more && p.norm_stats != nullptr;
        const bool rt_rmax = L.rowmax != nullptr;
        const bool rt_half = ONE && L.c_half != 0 && L.C != nullptr;
        const bool rt_rows = L.C != nullptr && !rt_half && N % 32 == 0;         // this layer's fp32 rows go to memory from here
        if (!more && !rt_rmax && !rt_norm && !rt_rows) return;
        const bool live = 32 * wn < N;
        const int cb = 32 * wn + 16 * hh;
        const bool rt_full = N == 256 && t0 + C4_ROWS <= M32;
        const float* const pm = pmax_all + h * GW * C4_ROWS;
        float* const inv_tab = inv_all + h * C4_ROWS;
        char* const planes = smem + h * C4_TILE_BYTES;
        const bool skip_next = more && p.L[l + 1].k1steps < p.L[l + 1].ksteps;
        // ---- row stores through LDS.  A lane owns one row and 64 of its bytes: stored directly (chain.hip did), an instruction touches 32
        // cache lines with two 16-byte pieces each -- the store path, not the matrix pipe, bounded the training runs (NOSTORE experiment:
        // 1099 -> 793 us per 4-layer run).  Here the wave's 64 x 32 block (8 KB) takes a round trip through LDS so that eight consecutive
        // lanes write one row's 128 bytes.  The space is the wave's OWN piece of the tile's dead planes, no other wave touches it meanwhile:
        //   with a following layer: the 128 bytes per row that this wave's split overwrites right afterwards (hi and lo chunks 4 wn .. 4 wn + 3,
        //     XOR-ed with the row like the planes themselves);
        //   last layer of the run: block wn (rows 8 wn .. 8 wn + 7 of the planes, 8 KB), which this wave's staging of the next tile
        //     overwrites afterwards -- the other waves' splits do not exist then.
        auto store_rows = [&](bool full) __attribute__((always_inline)) {
            if (!live) return;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rt = 32 * i + arow;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    char* dst = more ? planes + (rt >> 3) * C4_BLK_BYTES + (rt & 7) * 512 + hh * C4_LO + ((((unsigned)(4 * wn + g)) ^ (unsigned)(rt & 15)) * 16)
                                     : planes + wn * C4_BLK_BYTES + rt * 128 + ((((unsigned)(4 * hh + g)) ^ (unsigned)(rt & 7)) * 16);
                    *reinterpret_cast<float4*>(dst) = make_float4(a[i][4 * g], a[i][4 * g + 1], a[i][4 * g + 2], a[i][4 * g + 3]);
                }
            }
            const int q = lane >> 3, pc = lane & 7;     // store instruction s: row slot q, 16-byte piece pc of the row's 128 bytes
#pragma unroll
            for (int sx = 0; sx < 8; ++sx) {
                if ((sx & 1) == 0 && sx) asm volatile("" ::: "memory");     // (two rows' pieces in flight: 8 registers -- with two accumulator sets there is little room)
                // (chunk-private layout: the four rows that meet in one service group of a 16-byte LDS read differ in bits 2-3, i.e. in their XOR pattern)
                const int r = more ? (sx & 3) + 4 * (q & 3) + 16 * (q >> 2) + 32 * (sx >> 2) : 8 * sx + q;
                const char* src = more ? planes + (r >> 3) * C4_BLK_BYTES + (r & 7) * 512 + (pc >> 2) * C4_LO + ((((unsigned)(4 * wn + (pc & 3))) ^ (unsigned)(r & 15)) * 16)
                                       : planes + wn * C4_BLK_BYTES + r * 128 + ((((unsigned)pc) ^ (unsigned)(r & 7)) * 16);
                const float4 v = *reinterpret_cast<const float4*>(src);
                const int m = t0 + r;
                if (full || m < M32) *reinterpret_cast<float4*>(L.C + (long)m * L.ldc + 32 * wn + 4 * pc) = v;
            }
        };
        if (rt_norm) {
            const int nw = N / 32;                  // (the launcher keeps the norm in the run only for N a multiple of 32)
            const float2* const nr = nrm_all + h * GW * C4_ROWS;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rt = 32 * i + arow, m = t0 + rt;
                float mean = 0.f;
                for (int w = 0; w < nw; ++w) mean += nr[w * C4_ROWS + rt].x;
                mean /= (float)nw;
                float m2 = 0.f, dm = 0.f;
                for (int w = 0; w < nw; ++w) { const float2 t = nr[w * C4_ROWS + rt]; m2 += t.y; const float d = t.x - mean; dm += d * d; }
                m2 += 32.f * dm;
                const float sigma = sqrtf(m2 / (float)(N - 1));
                const float rinv = 1.0f / (sigma + p.norm_eps);
                if (live && m < M32 && L.C != nullptr) {     // (direct stores: one layer per run, and a third copy of the staged path costs the kernel its registers)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(L.C + (long)m * L.ldc + cb + 4 * g) =
                            make_float4((a[i][4 * g] - mean) * rinv, (a[i][4 * g + 1] - mean) * rinv, (a[i][4 * g + 2] - mean) * rinv, (a[i][4 * g + 3] - mean) * rinv);
                }
                if (wn == 0 && hh == 0 && m < M32) { p.norm_stats[(long)m * 2] = rinv; p.norm_stats[(long)m * 2 + 1] = sigma; }
            }
            return;
        }
        auto rows = [&](auto cfg) {
            using Cfg = decltype(cfg);
            const bool f_rmax = Cfg::rmax == 2 ? rt_rmax : Cfg::rmax == 1;
            const bool f_more = Cfg::more == 2 ? more : Cfg::more == 1;
            const bool f_full = Cfg::full == 2 ? rt_full : Cfg::full == 1;
            const bool f_rows = Cfg::rows == 2 ? rt_rows : Cfg::rows == 1;
#ifndef C4_X_NOSTORE
            if (f_rows) store_rows(f_full);
#endif
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rt = 32 * i + arow, m = t0 + rt;
                const bool in_m = f_full || m < M32;
                float mx = fmaxf(fmaxf(fmaxf(pm[rt], pm[C4_ROWS + rt]), fmaxf(pm[2 * C4_ROWS + rt], pm[3 * C4_ROWS + rt])),
                                 fmaxf(fmaxf(pm[4 * C4_ROWS + rt], pm[5 * C4_ROWS + rt]), fmaxf(pm[6 * C4_ROWS + rt], pm[7 * C4_ROWS + rt])));
                if (f_rmax && wn == 0 && hh == 0 && in_m) L.rowmax[m] = mx;
                if (f_more) {
                    if (Cfg::full != 1 && skip_next) mx = fmaxf(mx, xmax_all[h * C4_ROWS + rt]);
                    float inv;
                    const float sc = scale_from_max(__float_as_uint(mx), inv);
                    if (wn == 0 && hh == 0) inv_tab[rt] = inv;
                    if (live) {
                        char* const row = planes + (rt >> 3) * C4_BLK_BYTES + (rt & 7) * 512;
                        const unsigned x = (unsigned)(rt & 15);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {           // the lane's two 16-byte chunks of the row
                            const float4 v0 = make_float4(a[i][8 * q], a[i][8 * q + 1], a[i][8 * q + 2], a[i][8 * q + 3]);
                            const float4 v1 = make_float4(a[i][8 * q + 4], a[i][8 * q + 5], a[i][8 * q + 6], a[i][8 * q + 7]);
                            char* const dst = row + ((((unsigned)(cb >> 3) + q) ^ x) * 16);
                            if constexpr (ONE) {
                                unsigned h0, h1, h2, h3;
                                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(v0.x), "v"(sc));
                                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(v0.z), "v"(sc));
                                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h2) : "v"(v1.x), "v"(sc));
                                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h3) : "v"(v1.z), "v"(sc));
                                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(v0.y), "v"(sc));
                                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(v0.w), "v"(sc));
                                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h2) : "v"(v1.y), "v"(sc));
                                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h3) : "v"(v1.w), "v"(sc));
                                const uint4 hv = make_uint4(h0, h1, h2, h3);
                                *reinterpret_cast<uint4*>(dst) = hv;
                                // the f16 rows the weight-gradient reads (c_half): the same halfs, ldc counts halfs
                                if (rt_half && in_m) *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(L.C) + (long)m * L.ldc + cb + 8 * q) = hv;
                            } else {
                                half4 hi0, lo0, hi1, lo1;
                                split4(v0, sc, hi0, lo0);
                                split4(v1, sc, hi1, lo1);
                                const uint2 a0 = *reinterpret_cast<const uint2*>(&hi0), a1 = *reinterpret_cast<const uint2*>(&hi1);
                                const uint2 b0 = *reinterpret_cast<const uint2*>(&lo0), b1 = *reinterpret_cast<const uint2*>(&lo1);
                                *reinterpret_cast<uint4*>(dst) = make_uint4(a0.x, a0.y, a1.x, a1.y);
                                *reinterpret_cast<uint4*>(dst + C4_LO) = make_uint4(b0.x, b0.y, b1.x, b1.y);
                            }
                        }
                    }
                }
            }
        };
        if (!generic_only && rt_full && more && !skip_next && rt_rmax && rt_rows) rows(P2Cfg<1, 1, 1, 1>());       // training forward / data-gradient
        else if (!generic_only && rt_full && more && !skip_next && !rt_rmax && !rt_rows) rows(P2Cfg<0, 1, 1, 0>());       // inference
        else rows(P2Cfg<2, 2, 2, 2>());
    };

    // ---- schedule
    if (!DGRAD) {                                   // biases of all layers -> LDS (zero beyond a layer's width)
        for (int idx = tid; idx < n_layers * 256; idx += C4_THREADS) {
            const ChainLayer& L = p.L[idx >> 8];
            const int c = idx & 255;
            bias_all[idx] = (L.bias != nullptr && c < L.N) ? L.bias[c] : 0.f;
        }
    }
    const int n_steps = plan.n_steps;
    long pair = blockIdx.x;
    const long pstride = gridDim.x;
    const int kpad0 = p.L[0].k1steps * 16;
    stage(smem, inv_all, xmax_all, 2 * pair * C4_ROWS, false, kpad0);
    {
        const C4Step s0 = plan_step(0);
        const ChainLayer& L0 = p.L[s0.layer];
        const char* bh = frag_base(L0, L0.w_hi, s0.kbeg);
        const char* bl = frag_base(L0, L0.w_lo, s0.kbeg);
        const int n0 = s0.kcnt;
        const unsigned w_lane = (unsigned)lane0 * 16u;
        C4_WLOAD_ALL(n0, bh, bl);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first step's fragments
    lds_barrier();                                  // planes of the first X, the bias table

    f32x16 accX[NI], accY[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { accX[i][e] = 0.f; accY[i][e] = 0.f; }

    // slot (h, si) of pair `it`: multiply tile h by step si [+ P1 of its layer]; what the OTHER tile still owes from the slot before
    // (P2 of the layer it finished, the next tile's staging, or the re-staging of a skip layer's second segment)
    auto slot = [&](f32x16 (&aT)[NI], f32x16 (&aU)[NI], const int h, int si, int it, long pr) __attribute__((always_inline)) {
        const long mX = 2 * pr * C4_ROWS, mY = mX + C4_ROWS;
        const long mT = h == 0 ? mX : mY;
        char* const planesT = smem + h * C4_TILE_BYTES;
        char* const planesU = smem + (1 - h) * C4_TILE_BYTES;
        float* const invU = inv_all + (1 - h) * C4_ROWS;
        float* const xmaxU = xmax_all + (1 - h) * C4_ROWS;
        const C4Step st = plan_step(si);
        // the other tile's previous slot
        int psi; long pm0;
        if (h == 1) { psi = si; pm0 = mX; }
        else if (si > 0) { psi = si - 1; pm0 = mY; }
        else { psi = it > 0 ? n_steps - 1 : -1; pm0 = mY - 2 * pstride * C4_ROWS; }
        long sm0 = -1;                              // tile whose input rows are staged into the other tile's planes afterwards
        if (h == 0 && si == 0) sm0 = mY;
        if (h == 1 && si + 1 == n_steps && it + 1 < iters) sm0 = mX + 2 * pstride * C4_ROWS;
        const C4Step sn = plan_step(si + 1 < n_steps ? si + 1 : 0);
        unsigned sw = 0u;
        if (DGRAD && (st.flags & C4_LAST) && p.L[st.layer].sign_bits != nullptr && mT < p.M)
            sw = p.L[st.layer].sign_bits[(long)(mT / C4_ROWS) * (GW * 64) + wn * 64 + lane0];
        auto pending = [&]() __attribute__((always_inline)) {
            // the next k-loop's weight fragments (requested behind the last k-loop of tile Y, at least one P1 ago) have landed: waited for
            // HERE, in front of this wave's row stores -- at the top of the k-loop the same wait sat behind them (1044 -> us per 4-layer run)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (psi >= 0) {
                const C4Step sp = plan_step(psi);
                if (sp.flags & C4_LAST) p2_run(aU, 1 - h, pm0, sp.layer);
                else { const C4Step sq = plan_step(psi + 1); stage(planesU, invU, xmaxU, pm0, true, sq.kcnt * 16); }
            }
            if (sm0 >= 0) stage(planesU, invU, xmaxU, sm0, false, kpad0);
        };
        C4_STAMP();                                 // five stamps per slot: start | rows-first P2 done | K done | P1 done | k-first P2 done | (next start = barrier passed)
#ifndef C4_X_NOP2
        if (!k_first) pending();
#endif
        C4_STAMP();
#ifndef C4_X_NOK
        k_run(planesT, st, h == 1, sn, aT);
#endif
        C4_STAMP();
#ifndef C4_X_NOP1
        if (st.flags & C4_LAST) p1_run(aT, h, mT, st.layer, sw);
#endif
        C4_STAMP();
#ifndef C4_X_NOP2
        if (k_first) pending();
#endif
        C4_STAMP();
        lds_barrier();
    };
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int si = 0; si < n_steps; ++si) {
            slot(accX, accY, 0, si, it, pair);
            slot(accY, accX, 1, si, it, pair);
        }
        pair += pstride;
    }
    // the last Y of this workgroup still owes the second half of its last layer's row phase
    p2_run(accY, 1, 2 * (pair - pstride) * C4_ROWS + C4_ROWS, plan_step(n_steps - 1).layer);
}

}  // namespace

#ifdef PAPR_C4_TRACE
extern "C" int papr_chain4_trace_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain4_trace), sizeof(long long) * 2048) == hipSuccess ? 0 : 1; }
#endif

size_t papr_chain4_lds_bytes() { return C4_LDS_BYTES; }

int papr_launch_chain4(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s) {
    PAPR_REQUIRE(a.n_layers >= 1 && a.n_layers <= CHAIN_MAX_LAYERS, "mlp_chain4: %d layers", a.n_layers);
    PAPR_REQUIRE(a.K0 % 4 == 0 && a.lda0 % 4 == 0 && a.K0 <= 256, "mlp_chain4: input width %d", a.K0);
    C4Plan plan = {};
    for (int l = 0; l < a.n_layers; ++l) {
        const ChainLayer& L = a.L[l];
        PAPR_REQUIRE(L.k1steps >= 1 && L.k1steps <= KS && L.ksteps >= L.k1steps && L.ksteps - L.k1steps <= KS, "mlp_chain4: layer %d: %d + %d k-steps", l, L.k1steps, L.ksteps - L.k1steps);
        PAPR_REQUIRE(!L.c_half || (a.one_product && l + 1 < a.n_layers && L.rowmax && L.ldc % 8 == 0),
                     "mlp_chain4: layer %d: f16 rows need the one-product mode, a following layer, the row maxima and 16-byte rows", l);
        PAPR_REQUIRE(L.N % 4 == 0 && L.N <= 256 && (L.C == nullptr || L.ldc % 4 == 0), "mlp_chain4: layer %d: width %d, row stride %ld", l, L.N, L.ldc);
        if (L.k1steps == L.ksteps) plan.push(l, 0, L.ksteps, C4_FIRST | C4_LAST);
        else {
            PAPR_REQUIRE(!dgrad && l > 0, "mlp_chain4: layer %d: a second K segment needs a forward run and a layer in front", l);
            plan.push(l, 0, L.k1steps, C4_FIRST);
            plan.push(l, L.k1steps, L.ksteps - L.k1steps, C4_LAST | C4_RESTAGE);
        }
    }
    PAPR_REQUIRE(!a.norm_stats || a.L[a.n_layers - 1].N % 32 == 0, "mlp_chain4: a LayerNorm core behind the run needs a width that is a multiple of 32");
    if (a.M <= 0) return 0;
    const long tiles = (a.M + C4_ROWS - 1) / C4_ROWS;
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain4_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C4_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain4_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C4_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain4_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C4_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain4_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C4_LDS_BYTES);
        attr_set = true;
    }
    const long pairs = (tiles + 1) / 2;             // a workgroup carries two tiles at a time
    const unsigned grid = (unsigned)(pairs < n_cu ? pairs : n_cu);
    const int iters = (int)((pairs + grid - 1) / grid);
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin2(dgrad ? 10 : 9, a.M, a.n_layers, a.K0, bytes, flops, s);
    static const int generic_only = getenv("PAPR_C4_GENERIC") ? atoi(getenv("PAPR_C4_GENERIC")) : 0;      // (test switch: the hot instantiations off)
    if (a.one_product) {
        if (dgrad) mlp_chain4_kernel<true, true><<<dim3(grid), dim3(C4_THREADS), C4_LDS_BYTES, s>>>(a, plan, iters, generic_only);
        else mlp_chain4_kernel<false, true><<<dim3(grid), dim3(C4_THREADS), C4_LDS_BYTES, s>>>(a, plan, iters, generic_only);
    } else {
        if (dgrad) mlp_chain4_kernel<true, false><<<dim3(grid), dim3(C4_THREADS), C4_LDS_BYTES, s>>>(a, plan, iters, generic_only);
        else mlp_chain4_kernel<false, false><<<dim3(grid), dim3(C4_THREADS), C4_LDS_BYTES, s>>>(a, plan, iters, generic_only);
    }
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("mlp_chain4");
    return 0;
}
