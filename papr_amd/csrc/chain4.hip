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
#include "chain4_fused.inc"
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
#define C4_HIDDEN_VGPRS "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define C4_GET_X0 "v_mov_b32 %0, v64\n\tv_mov_b32 %1, v65\n\tv_mov_b32 %2, v66\n\tv_mov_b32 %3, v67\n\tv_mov_b32 %4, v68\n\tv_mov_b32 %5, v69\n\tv_mov_b32 %6, v70\n\tv_mov_b32 %7, v71\n\tv_mov_b32 %8, v72\n\tv_mov_b32 %9, v73\n\tv_mov_b32 %10, v74\n\tv_mov_b32 %11, v75\n\tv_mov_b32 %12, v76\n\tv_mov_b32 %13, v77\n\tv_mov_b32 %14, v78\n\tv_mov_b32 %15, v79"
#define C4_PUT_X0 "v_mov_b32 v64, %0\n\tv_mov_b32 v65, %1\n\tv_mov_b32 v66, %2\n\tv_mov_b32 v67, %3\n\tv_mov_b32 v68, %4\n\tv_mov_b32 v69, %5\n\tv_mov_b32 v70, %6\n\tv_mov_b32 v71, %7\n\tv_mov_b32 v72, %8\n\tv_mov_b32 v73, %9\n\tv_mov_b32 v74, %10\n\tv_mov_b32 v75, %11\n\tv_mov_b32 v76, %12\n\tv_mov_b32 v77, %13\n\tv_mov_b32 v78, %14\n\tv_mov_b32 v79, %15"
#define C4_GET_X1 "v_mov_b32 %0, v80\n\tv_mov_b32 %1, v81\n\tv_mov_b32 %2, v82\n\tv_mov_b32 %3, v83\n\tv_mov_b32 %4, v84\n\tv_mov_b32 %5, v85\n\tv_mov_b32 %6, v86\n\tv_mov_b32 %7, v87\n\tv_mov_b32 %8, v88\n\tv_mov_b32 %9, v89\n\tv_mov_b32 %10, v90\n\tv_mov_b32 %11, v91\n\tv_mov_b32 %12, v92\n\tv_mov_b32 %13, v93\n\tv_mov_b32 %14, v94\n\tv_mov_b32 %15, v95"
#define C4_PUT_X1 "v_mov_b32 v80, %0\n\tv_mov_b32 v81, %1\n\tv_mov_b32 v82, %2\n\tv_mov_b32 v83, %3\n\tv_mov_b32 v84, %4\n\tv_mov_b32 v85, %5\n\tv_mov_b32 v86, %6\n\tv_mov_b32 v87, %7\n\tv_mov_b32 v88, %8\n\tv_mov_b32 v89, %9\n\tv_mov_b32 v90, %10\n\tv_mov_b32 v91, %11\n\tv_mov_b32 v92, %12\n\tv_mov_b32 v93, %13\n\tv_mov_b32 v94, %14\n\tv_mov_b32 v95, %15"
#define C4_GET_Y0 "v_mov_b32 %0, v96\n\tv_mov_b32 %1, v97\n\tv_mov_b32 %2, v98\n\tv_mov_b32 %3, v99\n\tv_mov_b32 %4, v100\n\tv_mov_b32 %5, v101\n\tv_mov_b32 %6, v102\n\tv_mov_b32 %7, v103\n\tv_mov_b32 %8, v104\n\tv_mov_b32 %9, v105\n\tv_mov_b32 %10, v106\n\tv_mov_b32 %11, v107\n\tv_mov_b32 %12, v108\n\tv_mov_b32 %13, v109\n\tv_mov_b32 %14, v110\n\tv_mov_b32 %15, v111"
#define C4_PUT_Y0 "v_mov_b32 v96, %0\n\tv_mov_b32 v97, %1\n\tv_mov_b32 v98, %2\n\tv_mov_b32 v99, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v102, %6\n\tv_mov_b32 v103, %7\n\tv_mov_b32 v104, %8\n\tv_mov_b32 v105, %9\n\tv_mov_b32 v106, %10\n\tv_mov_b32 v107, %11\n\tv_mov_b32 v108, %12\n\tv_mov_b32 v109, %13\n\tv_mov_b32 v110, %14\n\tv_mov_b32 v111, %15"
#define C4_GET_Y1 "v_mov_b32 %0, v112\n\tv_mov_b32 %1, v113\n\tv_mov_b32 %2, v114\n\tv_mov_b32 %3, v115\n\tv_mov_b32 %4, v116\n\tv_mov_b32 %5, v117\n\tv_mov_b32 %6, v118\n\tv_mov_b32 %7, v119\n\tv_mov_b32 %8, v120\n\tv_mov_b32 %9, v121\n\tv_mov_b32 %10, v122\n\tv_mov_b32 %11, v123\n\tv_mov_b32 %12, v124\n\tv_mov_b32 %13, v125\n\tv_mov_b32 %14, v126\n\tv_mov_b32 %15, v127"
#define C4_PUT_Y1 "v_mov_b32 v112, %0\n\tv_mov_b32 v113, %1\n\tv_mov_b32 v114, %2\n\tv_mov_b32 v115, %3\n\tv_mov_b32 v116, %4\n\tv_mov_b32 v117, %5\n\tv_mov_b32 v118, %6\n\tv_mov_b32 v119, %7\n\tv_mov_b32 v120, %8\n\tv_mov_b32 v121, %9\n\tv_mov_b32 v122, %10\n\tv_mov_b32 v123, %11\n\tv_mov_b32 v124, %12\n\tv_mov_b32 v125, %13\n\tv_mov_b32 v126, %14\n\tv_mov_b32 v127, %15"
#define C4_ZERO_X "v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\tv_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\tv_mov_b32 v72, 0\n\tv_mov_b32 v73, 0\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\tv_mov_b32 v76, 0\n\tv_mov_b32 v77, 0\n\tv_mov_b32 v78, 0\n\tv_mov_b32 v79, 0\n\tv_mov_b32 v80, 0\n\tv_mov_b32 v81, 0\n\tv_mov_b32 v82, 0\n\tv_mov_b32 v83, 0\n\tv_mov_b32 v84, 0\n\tv_mov_b32 v85, 0\n\tv_mov_b32 v86, 0\n\tv_mov_b32 v87, 0\n\tv_mov_b32 v88, 0\n\tv_mov_b32 v89, 0\n\tv_mov_b32 v90, 0\n\tv_mov_b32 v91, 0\n\tv_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\tv_mov_b32 v95, 0"
#define C4_ZERO_Y "v_mov_b32 v96, 0\n\tv_mov_b32 v97, 0\n\tv_mov_b32 v98, 0\n\tv_mov_b32 v99, 0\n\tv_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\tv_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\tv_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\tv_mov_b32 v126, 0\n\tv_mov_b32 v127, 0"
#define C4_OUT16(t, o) "=v"(t[(o) + 0]), "=v"(t[(o) + 1]), "=v"(t[(o) + 2]), "=v"(t[(o) + 3]), "=v"(t[(o) + 4]), "=v"(t[(o) + 5]), "=v"(t[(o) + 6]), "=v"(t[(o) + 7]), "=v"(t[(o) + 8]), "=v"(t[(o) + 9]), "=v"(t[(o) + 10]), "=v"(t[(o) + 11]), "=v"(t[(o) + 12]), "=v"(t[(o) + 13]), "=v"(t[(o) + 14]), "=v"(t[(o) + 15])
#define C4_IN16(t, o) "v"(t[(o) + 0]), "v"(t[(o) + 1]), "v"(t[(o) + 2]), "v"(t[(o) + 3]), "v"(t[(o) + 4]), "v"(t[(o) + 5]), "v"(t[(o) + 6]), "v"(t[(o) + 7]), "v"(t[(o) + 8]), "v"(t[(o) + 9]), "v"(t[(o) + 10]), "v"(t[(o) + 11]), "v"(t[(o) + 12]), "v"(t[(o) + 13]), "v"(t[(o) + 14]), "v"(t[(o) + 15])
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
#ifdef PAPR_C4_TRACE_FINE                           // two more stamps per slot inside `pending`: P2 done | staging loads landed
#define C4_STAMP2() C4_STAMP()
#else
#define C4_STAMP2() do {} while (0)
#endif
#else
#define C4_STAMP() do {} while (0)
#define C4_STAMP2() do {} while (0)
#endif

__device__ __forceinline__ long uniform64(long v) {       // a wave-uniform value the compiler keeps in scalar registers and does not move out of loops
    int lo = __builtin_amdgcn_readfirstlane((int)v), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

// ONE: the reduced-precision mode (one f16 product per fp32 product, hi planes only; the counterpart of the reference's fp16
// autocast, models/attn.py:248)
// REGISTERS.  256 per lane: a[0:127] = the layer's weight fragments (by name, chain3.hip: why); v64-v127 = the two accumulator sets
// (tile X: v[64:95], tile Y: v[96:127]), ALSO by name: as C++ variables they are 64 registers that live across everything, and every
// statement with other needs made the register allocator move or spill whole 16-register tuples (770 spilled registers with the fused
// slots in).  The compiler gets v0-v63 (amdgpu_num_vgpr) and never sees the rest: inline asm names them, the C++ row phases copy a
// tile's 32 values in and out (acc_get / acc_put).
template <bool DGRAD, bool ONE>
__global__ __launch_bounds__(C4_THREADS, 2) __attribute__((amdgpu_num_vgpr(64))) void mlp_chain4_kernel(ChainArgs p, C4Plan plan, int iters, int generic_only, int fused_on, int prefetch_on) {
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

    // ---- the accumulators: v64-v127, outside the compiler's view (see REGISTERS above)
    asm volatile("" ::: C4_HIDDEN_VGPRS);            // (so that the kernel's register count covers them)
    // a tile's 32 values into C++ variables and back (generic row phases; two statements of 16 operands each: an asm takes at most 30)
    auto acc_get16 = [&](const int h, const int i, f32x16& a) __attribute__((always_inline)) {      // (h, i: constants after inlining)
        float t[16];
        if (h == 0 && i == 0) asm volatile(C4_GET_X0 : C4_OUT16(t, 0));
        else if (h == 0) asm volatile(C4_GET_X1 : C4_OUT16(t, 0));
        else if (i == 0) asm volatile(C4_GET_Y0 : C4_OUT16(t, 0));
        else asm volatile(C4_GET_Y1 : C4_OUT16(t, 0));
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = t[e];
    };
    auto acc_put16 = [&](const int h, const int i, const f32x16& a) __attribute__((always_inline)) {
        float t[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = a[e];
        if (h == 0 && i == 0) asm volatile(C4_PUT_X0 : : C4_IN16(t, 0) : "memory");
        else if (h == 0) asm volatile(C4_PUT_X1 : : C4_IN16(t, 0) : "memory");
        else if (i == 0) asm volatile(C4_PUT_Y0 : : C4_IN16(t, 0) : "memory");
        else asm volatile(C4_PUT_Y1 : : C4_IN16(t, 0) : "memory");
    };

    // ---- multiply the tile in `planes` by a step (chain4_krun.inc): k_run_X for tile X, k_run_Y for tile Y
#define C4_KRUN_NAME k_run_X
#define C4_KL(x) x##_X
#define C4_A0 "v[64:79]"
#define C4_A1 "v[80:95]"
#define C4_ZERO C4_ZERO_X
#include "chain4_krun.inc"
#undef C4_ZERO
#undef C4_KRUN_NAME
#undef C4_KL
#undef C4_A0
#undef C4_A1
#define C4_KRUN_NAME k_run_Y
#define C4_KL(x) x##_Y
#define C4_A0 "v[96:111]"
#define C4_A1 "v[112:127]"
#define C4_ZERO C4_ZERO_Y
#include "chain4_krun.inc"
#undef C4_ZERO
#undef C4_KRUN_NAME
#undef C4_KL
#undef C4_A0
#undef C4_A1

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
        if (!again) { asm volatile("s_waitcnt vmcnt(0)" :: "v"(v[7].x) : "memory"); }
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
    auto p1_run = [&](const int h, long m0, int l, unsigned sw_in) __attribute__((always_inline)) {
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
                f32x16 ai;                            // (one 32-row tile's values at a time: the compiler has 64 registers)
                acc_get16(h, i, ai);
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
                            const float pre = __builtin_fmaf(ai[4 * g + c], inv, bb[c]);
                            y[c] = f_relu ? fmaxf(pre, 0.f) : fmaxf(pre, __builtin_fmaf(pre, slope, 0.f));
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float gv = ai[4 * g + c] * inv;
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
                    for (int c = 0; c < 4; ++c) ai[4 * g + c] = y[c];
                }
                if (!rt_norm) {
                    lmax = pair_max(lmax);
                    if (hh == 0) pm[rt] = lmax;
                } else {
                    // LayerNorm core behind the run (FeedForward.outnorm, act = none): this wave's 32 columns of the row -> (mean, M2),
                    // sums in a fixed order; P2 combines the waves
                    float s = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) s += (ai[4 * g] + ai[4 * g + 1]) + (ai[4 * g + 2] + ai[4 * g + 3]);
                    const float mean_w = pair_sum(s) * (1.f / 32.f);
                    float q = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float d0 = ai[4 * g] - mean_w, d1 = ai[4 * g + 1] - mean_w, d2 = ai[4 * g + 2] - mean_w, d3 = ai[4 * g + 3] - mean_w;
                        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    }
                    q = pair_sum(q);
                    if (hh == 0) nrm_all[(h * GW + wn) * C4_ROWS + rt] = make_float2(mean_w, q);
                }
                acc_put16(h, i, ai);
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
    auto p2_run = [&](const int h, long m0, int l) __attribute__((always_inline)) {
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
                f32x16 ai;
                acc_get16(h, i, ai);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    char* dst = more ? planes + (rt >> 3) * C4_BLK_BYTES + (rt & 7) * 512 + hh * C4_LO + ((((unsigned)(4 * wn + g)) ^ (unsigned)(rt & 15)) * 16)
                                     : planes + wn * C4_BLK_BYTES + rt * 128 + ((((unsigned)(4 * hh + g)) ^ (unsigned)(rt & 7)) * 16);
                    *reinterpret_cast<float4*>(dst) = make_float4(ai[4 * g], ai[4 * g + 1], ai[4 * g + 2], ai[4 * g + 3]);
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
                if (live) {                           // the standardised values replace the tile's accumulators; store_rows takes them from there
                    f32x16 ai;
                    acc_get16(h, i, ai);
#pragma unroll
                    for (int e = 0; e < 16; ++e) ai[e] = (ai[e] - mean) * rinv;
                    acc_put16(h, i, ai);
                }
                if (wn == 0 && hh == 0 && m < M32) { p.norm_stats[(long)m * 2] = rinv; p.norm_stats[(long)m * 2 + 1] = sigma; }
            }
            if (L.C != nullptr) store_rows(false);      // (whole cache lines through the wave's block of the dead planes: last layer of the run)
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
                    f32x16 ai;
                    acc_get16(h, i, ai);
                    if (Cfg::full != 1 && skip_next) mx = fmaxf(mx, xmax_all[h * C4_ROWS + rt]);
                    float inv;
                    const float sc = scale_from_max(__float_as_uint(mx), inv);
                    if (wn == 0 && hh == 0) inv_tab[rt] = inv;
                    if (live) {
                        char* const row = planes + (rt >> 3) * C4_BLK_BYTES + (rt & 7) * 512;
                        const unsigned x = (unsigned)(rt & 15);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {           // the lane's two 16-byte chunks of the row
                            const float4 v0 = make_float4(ai[8 * q], ai[8 * q + 1], ai[8 * q + 2], ai[8 * q + 3]);
                            const float4 v1 = make_float4(ai[8 * q + 4], ai[8 * q + 5], ai[8 * q + 6], ai[8 * q + 7]);
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



    // ---- the hot slot as ONE statement (chain4_fused.inc, generated by scripts/gen_chain4_fused.py -- the why and the how are there):
    // the k-loop of tile T (step st: 16 k-steps from zero), P1 | s_barrier | P2 of tile U's layer pl, on the accumulators where they live
    // (tile X = v[64:95], tile Y = v[96:127]): one flavour per tile.
    // mode: 0 training forward, 1 inference, 2 data-gradient.
#define C4F_INPUTS                                                                                                              \
        [pbx] "v"(pbx), [wv] "v"(wv), [invad] "v"(invad), [biasad] "v"(biasad), [pmw] "v"(pmw), [pmr] "v"(pmr), [stw] "v"(stw),    \
        [rdb] "v"(rdb), [gso] "v"(gso), [plw] "v"(plw), [sgn] "s"(sgn), [rmp] "s"(rmp), [crow0] "s"(crow0), [crow1] "s"(crow1),  \
        [slope] "s"(slope), [c254] "s"(254), [pfb] "s"(pfb), [pfs] "s"(pfs)
#define C4F_LD_INPUTS , [nhlo] "s"(nhlo), [nhhi] "s"(nhhi), [nllo] "s"(nllo), [nlhi] "s"(nlhi)
#define C4F_RUN(NAME)                                                                                                           \
        do {                                                                                                                    \
            if (kc == 8) asm volatile(NAME##_Y_LD_K8 : [word] "+v"(word) : C4F_INPUTS C4F_LD_INPUTS : C4F_CLOBBERS, C4F_LD_CLOBBERS, C4F_AGPRS);   \
            else if (kc == 10) asm volatile(NAME##_Y_LD_K10 : [word] "+v"(word) : C4F_INPUTS C4F_LD_INPUTS : C4F_CLOBBERS, C4F_LD_CLOBBERS, C4F_AGPRS);   \
            else if (h == 0) asm volatile(NAME##_X_NL : [word] "+v"(word) : C4F_INPUTS : C4F_CLOBBERS);                         \
            else if (ld) asm volatile(NAME##_Y_LD : [word] "+v"(word) : C4F_INPUTS C4F_LD_INPUTS : C4F_CLOBBERS, C4F_LD_CLOBBERS, C4F_AGPRS);  \
            else asm volatile(NAME##_Y_NL : [word] "+v"(word) : C4F_INPUTS : C4F_CLOBBERS);                                     \
        } while (0)
    auto fused_slot = [&](const int h, const int kc, C4Step sn, bool refill, long pm0, int pl, unsigned& swU, int mode, long pf_m0) __attribute__((always_inline)) {
        C4_LANE();
        pl = __builtin_amdgcn_readfirstlane(pl);
        const ChainLayer& LP = p.L[pl];
        const int t0 = __builtin_amdgcn_readfirstlane((int)pm0);
        const unsigned planesT = (unsigned)(size_t)(smem + h * C4_TILE_BYTES), planesU = (unsigned)(size_t)(smem + (1 - h) * C4_TILE_BYTES);
        const int arow = lane & 31, hh = lane >> 5, ax = arow & 15, q = lane >> 3, pc = lane & 7;
        const unsigned rowoff = (unsigned)((arow >> 3) * C4_BLK_BYTES + (arow & 7) * 512);
        const unsigned pbx = (planesT + rowoff + (unsigned)((hh ^ (ax & 1)) * 16)) ^ (unsigned)((ax & ~1) * 16);
        const unsigned wv = (unsigned)lane * 16u;
        const unsigned invbase = (unsigned)(size_t)(inv_all + (1 - h) * C4_ROWS);
        const unsigned invad = invbase + (unsigned)arow * 4u;
        const unsigned biasad = (unsigned)(size_t)(bias_all + pl * 256 + 32 * wn + 16 * hh);
        const unsigned pmr = (unsigned)(size_t)(pmax_all + (1 - h) * GW * C4_ROWS) + (unsigned)arow * 4u, pmw = pmr + (unsigned)wn * 256u;
        const unsigned stw = planesU + rowoff + (unsigned)hh * C4_LO + (((unsigned)(4 * wn) ^ (unsigned)ax) << 4);
        const unsigned rdb = planesU + (unsigned)((((q >> 1) & 1) + 2 * (q >> 2)) * C4_BLK_BYTES + (q & 1) * 2048 + (pc >> 2) * C4_LO) +
                             ((unsigned)(4 * (wn ^ (q & 3)) + (pc & 3)) << 4);
        const unsigned gso = (unsigned)((4 * (q & 3) + 16 * (q >> 2)) * 1024 + wn * 128 + pc * 16);
        const unsigned plw = planesU + rowoff + (((unsigned)(4 * wn + 2 * hh) ^ (unsigned)ax) << 4);
        const char* sgn = reinterpret_cast<const char*>(LP.sign_bits + (long)(t0 / C4_ROWS) * (GW * 64) + wn * 64);
        const char* rmp = reinterpret_cast<const char*>(LP.rowmax + t0) - invbase;      // (the store's lane offset is the 1 / scale table's LDS address)
        const char* crow0 = reinterpret_cast<const char*>(LP.C + (long)t0 * 256);
        const char* crow1 = crow0 + 32768;
        const float slope = LP.act == PAPR_ACT_LEAKY_RELU ? 0.2f : 0.f;
        const bool relu = LP.act == PAPR_ACT_RELU;
        const ChainLayer& Ln = p.L[sn.layer];
        const char *nh = frag_base(Ln, Ln.w_hi, sn.kbeg), *nl = frag_base(Ln, Ln.w_lo, sn.kbeg);
        const unsigned nhlo = (unsigned)(size_t)nh, nhhi = (unsigned)((size_t)nh >> 32), nllo = (unsigned)(size_t)nl, nlhi = (unsigned)((size_t)nl >> 32);
        const bool ld = refill && sn.kcnt == KS;
        // prefetch (see the generator): this wave's eight input rows of the tile at pf_m0, or nothing (pf_m0 < 0)
        const char* pfb = reinterpret_cast<const char*>(p.A0 + (pf_m0 >= 0 ? pf_m0 + wn * RB : 0) * p.lda0);
        const unsigned pfs = pf_m0 >= 0 ? (unsigned)p.lda0 * 4u : 0u;
        unsigned word = swU;
        // this step's fragments have landed: requested behind tile Y's k-loop of the step before, so tile X waits here -- and tile Y,
        // a slot later, finds them waited for (a wait there would only wait for tile X's row stores: loads and stores share the counter)
        if (h == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (DGRAD) {
            if (relu) C4F_RUN(C4F_DGRAD_RELU); else C4F_RUN(C4F_DGRAD_LEAKY);
        } else if (mode == 0) {
            if (relu) C4F_RUN(C4F_FWD_RELU); else C4F_RUN(C4F_FWD_LEAKY);
        } else {
            if (relu) C4F_RUN(C4F_INF_RELU); else C4F_RUN(C4F_INF_LEAKY);
        }
        if (refill && !ld) { const int nks = sn.kcnt; const unsigned w_lane = wv; C4_WLOAD_ALL(nks, nh, nl); }     // (a narrower next step: its fragments in a bunch)
    };

    // slot (h, si) of pair `it`, tile T = h:   P1 of the OTHER tile U (the layer it finished multiplying a slot ago)  |  barrier  |
    // multiply T by step si  ||  P2 of U (or the next tile's staging, or the re-staging of a skip layer's second segment)  |  barrier.
    // Both halves of U's row phase sit a slot behind its k-loop, so that they can share the slot with T's matrix instructions: in the hot
    // slots as ONE interleaved statement (fused_slot), elsewhere with the waves in two roles (waves 0-3 multiply first, waves 4-7 last).
    unsigned swX = 0u, swY = 0u;                    // data-gradient: the sign word of the tile's pending P1, requested a slot ahead
    auto slot = [&](unsigned& swT, unsigned& swU, const int h, int si, int it, long pr) __attribute__((always_inline)) {
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
        const C4Step sp = plan_step(psi >= 0 ? psi : 0);
        const bool p_rows = psi >= 0 && (sp.flags & C4_LAST) != 0;     // U owes a row phase
        if (DGRAD) {                                // T's sign word for the next slot
            swT = 0u;
            if ((st.flags & C4_LAST) && p.L[st.layer].sign_bits != nullptr && mT < p.M)
                swT = p.L[st.layer].sign_bits[(long)(mT / C4_ROWS) * (GW * 64) + wn * 64 + lane0];
        }
        // is this a hot slot?  T: a whole 256 x 256 step from zero; U: a middle layer (256 wide, a following layer that is no skip layer),
        // its tile inside M, one of the three hot flag sets, rows of 256 floats; nothing to stage
        int fmode = -1;
        // (a short first layer -- 8 or 10 k-steps -- has fused forms for tile Y with the next step's 16 k-steps of fragments to request)
        const bool k_hot = st.kcnt == KS || (h == 1 && (st.kcnt == 8 || st.kcnt == 10) && sn.kcnt == KS);
        if (!ONE && fused_on && !generic_only && p_rows && sm0 < 0 && k_hot && (st.flags & C4_FIRST) && (st.flags & C4_LAST) && p.L[st.layer].N == 256) {
            const ChainLayer& LP = p.L[sp.layer];
            const bool mid = LP.N == 256 && sp.layer + 1 < n_layers && p.L[sp.layer + 1].k1steps == p.L[sp.layer + 1].ksteps && pm0 + C4_ROWS <= p.M;
            const bool st_ = LP.C != nullptr, bi = LP.sign_bits != nullptr, rm = LP.rowmax != nullptr;
            const bool act_ok = LP.act == PAPR_ACT_RELU || LP.act == PAPR_ACT_LEAKY_RELU;
            if (mid && act_ok && st_ && bi && rm && LP.ldc == 256 && !LP.c_half) fmode = DGRAD ? 2 : 0;
            else if (mid && act_ok && !DGRAD && !st_ && !bi && !rm) fmode = 1;
        }
        if (fmode >= 0) {
            C4_STAMP();
            // (the next pair's tile: staged two to four slots from now; a lane reads 16 bytes at 16 lane of each row, so the whole
            // kilobyte behind a row's start must lie inside the array: one more tile of margin)
            // (measured: the block costs its slot 3-4k cycles and the staging slots two to four slots later were not shorter for it -- they
            // are bound by their own 600 instructions without matrix work beside them, not by the rows' arrival.  PAPR_C4_PREFETCH=1 to try.)
            long pf_m0 = -1;
            if (prefetch_on && si + 2 == n_steps && it + 1 < iters && mT + 2 * pstride * C4_ROWS + 3 * C4_ROWS <= p.M) pf_m0 = mT + 2 * pstride * C4_ROWS;
            fused_slot(h, st.kcnt, sn, h == 1, pm0, sp.layer, swU, fmode, pf_m0);
            C4_STAMP(); C4_STAMP(); C4_STAMP(); C4_STAMP(); C4_STAMP2(); C4_STAMP2();
            lds_barrier();
            return;
        }
        C4_STAMP();                                 // five stamps per slot: start | P1 done | barrier passed | first piece done | second piece done | (next start = barrier passed)
#ifndef C4_X_NOP1
        if (p_rows) p1_run(1 - h, pm0, sp.layer, swU);
#endif
        C4_STAMP();
        if (p_rows) lds_barrier();                  // every wave's partial maxima of U are in LDS
        C4_STAMP();
        auto pending = [&]() __attribute__((always_inline)) {
            // the k-loop's weight fragments (requested behind the last k-loop of tile Y) have landed: waited for HERE, in front of this
            // wave's row stores -- loads and stores share the counter
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (p_rows) p2_run(1 - h, pm0, sp.layer);
            C4_STAMP2();
            if (!p_rows && psi >= 0) { const C4Step sq = plan_step(psi + 1); stage(planesU, invU, xmaxU, pm0, true, sq.kcnt * 16); }
            if (sm0 >= 0) stage(planesU, invU, xmaxU, sm0, false, kpad0);
            C4_STAMP2();
        };
#ifndef C4_X_NOP2
        if (!k_first) pending();
#endif
#ifndef C4_X_NOK
        if (k_first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (h == 0) k_run_X(planesT, st, false, sn); else k_run_Y(planesT, st, true, sn);
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
            slot(swX, swY, 0, si, it, pair);
            slot(swY, swX, 1, si, it, pair);
        }
        pair += pstride;
    }
    // the last Y of this workgroup still owes its last layer's row phase
    {
        const long mYl = 2 * (pair - pstride) * C4_ROWS + C4_ROWS;
        const int ll = plan_step(n_steps - 1).layer;
        p1_run(1, mYl, ll, swY);
        lds_barrier();
        p2_run(1, mYl, ll);
    }
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
    static const int fused_on = getenv("PAPR_C4_FUSED") ? atoi(getenv("PAPR_C4_FUSED")) : 1;            // (A/B switch: 0 = the two-role slots everywhere)
    static const int prefetch_on = getenv("PAPR_C4_PREFETCH") ? atoi(getenv("PAPR_C4_PREFETCH")) : 0;   // (A/B switch: the next pair's input rows pulled towards the caches from inside a fused slot)
    if (a.one_product) {
        if (dgrad) mlp_chain4_kernel<true, true><<<dim3(grid), dim3(C4_THREADS), C4_LDS_BYTES, s>>>(a, plan, iters, generic_only, fused_on, prefetch_on);
        else mlp_chain4_kernel<false, true><<<dim3(grid), dim3(C4_THREADS), C4_LDS_BYTES, s>>>(a, plan, iters, generic_only, fused_on, prefetch_on);
    } else {
        if (dgrad) mlp_chain4_kernel<true, false><<<dim3(grid), dim3(C4_THREADS), C4_LDS_BYTES, s>>>(a, plan, iters, generic_only, fused_on, prefetch_on);
        else mlp_chain4_kernel<false, false><<<dim3(grid), dim3(C4_THREADS), C4_LDS_BYTES, s>>>(a, plan, iters, generic_only, fused_on, prefetch_on);
    }
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("mlp_chain4");
    return 0;
}
