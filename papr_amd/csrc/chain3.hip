// K3d: fused layer runs, third structure -- the layer's weights stay in registers.
//
// Replaces the same reference lines as chain.hip (MLP.forward, models/mlp.py:47-59, and its autograd data-gradient) for
// runs without skip layers; chain.hip keeps the skip-layer instantiation, chain2.hip stays selectable (PAPR_CHAIN=2).
//
// What the per-wave cycle stamps of chain2.hip showed (profiles/chain2_trace_r02.txt): with the row phases switched off
// its k-loops alone take 10.2k cycles per 64-row tile and layer against 6.1k of matrix-pipe time -- every 64-row tile
// re-reads the layer's 256 KB of weight fragments from L2, 256 CUs at once: 29 B/clk/CU, half of what the L2 delivers at
// best, and the same for any 64-row tiling whatever the wave arrangement.  With the k-loops switched off its row phases
// alone take 8.6k.  Two wave groups in opposite roles add up to 11.5k-16k per tile and layer.  So:
//
//   * WEIGHTS IN REGISTERS.  Eight waves per workgroup (two per SIMD, 256 registers each); wave w multiplies ALL rows
//     by columns 32 w .. 32 w + 31 and keeps that slice of the layer -- 16 k-steps x (hi, lo) fragments = 128 registers
//     -- for two 64-row tiles X and Y, both resident in LDS as split-f16 planes (64 KB each).  The k-loop has no
//     global loads at all; the weight traffic per row halves.  The next layer's fragments are requested into each
//     k-step's registers right after that k-step's last use (tile Y), one slot (~8k cycles) before they are needed.
//   * ROLES BY TIME, NOT BY WAVE.  A slot = [multiply tile A with layer l] + [row phases of tile B: bias, activation,
//     row maximum, split into planes, stores].  The two pieces touch different tiles and are independent inside a
//     slot, so waves 0-3 multiply first and do their rows second, waves 4-7 (their SIMD partners) the other way round:
//     each SIMD always has one wave in the matrix pipe and one in the vector / memory pipes.  Between slots: barrier,
//     every wave dumps its 64 x 32 accumulator block as raw fp32 over A's planes (dead now), barrier.
//     Slot order per tile pair: (X, l) | rows of (Y, l - 1);  (Y, l) | rows of (X, l);  ...
//   * Row phases, LDS layouts (XOR swizzles, fp32 overlay over the row's own hi / lo plane rows), lane-local sign
//     words: chain2.hip's at 8 rows per wave.
//
// Arithmetic is unchanged (same MFMA order per accumulator, same power-of-two row scales, same split): every stored
// value is bit-identical to chain2.hip's, and to chain.hip's except behind a LayerNorm core (wave-reduction order).
// Built without packed-fp32 VALU like chain2.hip (papr_amd/build.py).
#include "papr_common.h"
#include "h3_common.h"
#include "chain.h"
#include "chain3_kloop.inc"
#include "chain3_fused.inc"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int NI = 2;                           // a multiplying wave: 64 rows x 32 columns
constexpr int GW = 8;                           // waves
constexpr int RB = 8;                           // rows per wave in the row phases (one block of the planes)
constexpr int C3_THREADS = GW * 64;
constexpr int C3_ROWS = 64;                     // rows per tile
constexpr int C3_TILE_BYTES = 65536;            // A planes of one tile: GW blocks of RB rows
constexpr int C3_BLK_BYTES = RB * 1024;         // one block: hi rows (RB x 512 B) | lo rows (RB x 512 B); fp32 overlay of row u: columns 0-127 over hi row u, 128-255 over lo row u
constexpr int C3_LO = RB * 512;
constexpr int RQ = 4;                           // rows a wave carries through the row phase at a time
constexpr int KS = 16;                          // k-steps of a 256-wide layer
constexpr int C3_PARK_F4 = 7;                   // float4s of its 8 accumulator float4s that a multiply-first wave parks in LDS during its row phases
constexpr int C3_PARK_BYTES = C3_PARK_F4 * 1024;
constexpr size_t C3_LDS_BYTES = 2 * C3_TILE_BYTES + 2 * C3_ROWS * sizeof(float) + 4 * C3_PARK_BYTES;      // planes of X and Y, 1 / scale of every plane row, parking

__device__ __forceinline__ float scale_from_max(unsigned bits, float& inv) {      // row max -> [2^13, 2^14)
    const int ea = bits ? (int)((bits >> 23) & 0xff) : 127 + 13;
    inv = pow2_from_biased(127 - 13 + (ea - 127));
    return pow2_from_biased(127 + 13 - (ea - 127));
}

// max over the 64 lanes of four registers at once (result valid in lane 63): the four chains interleave, so that the two
// wait states a DPP read needs behind the VALU write of its source are filled by the other rows' instructions
#define C3_DPP4(ctrl)                                                  \
    "v_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\t" \
    "v_max_f32_dpp %2, %2, %2 " ctrl "\n\tv_max_f32_dpp %3, %3, %3 " ctrl "\n\t"
__device__ __forceinline__ void wave_max4(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t" C3_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") C3_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        C3_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf") C3_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
        C3_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf") C3_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ float last_lane(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
// lanes l0 .. l0 + 3 of vec = four scalars (s_nop: the scalars come out of v_readlane; VALU-written SGPRs need two wait
// states before a VALU reads them and hipcc does not count inline asm)
#define put4(vec, l0, s0, s1, s2, s3)                                                                                             \
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %0, %3, %7\n\tv_writelane_b32 %0, %4, %8" \
        : "+v"(vec) : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "i"(l0), "i"((l0) + 1), "i"((l0) + 2), "i"((l0) + 3))

// uniform per-layer flags of the row phases: 0 / 1 = known at compile time (the hot instantiations), 2 = look at run time
template <int STORE, int BITS, int RMAX, int MORE, int NORM, int FULL>
struct RowCfg { static constexpr int store = STORE, bits = BITS, rmax = RMAX, more = MORE, norm = NORM, full = FULL; };

#ifdef PAPR_H3_TRACE
__device__ long long g_chain3_trace[1024];      // 8 waves x 128 stamps
#define C3_STAMP() do { if (blockIdx.x == 100 && lane0 == 0 && trace_slot < 128) g_chain3_trace[wn * 128 + trace_slot++] = __builtin_readcyclecounter(); } while (0)
#ifdef PAPR_C3_TRACE_FINE
#define C3_STAMP2() do { asm volatile("" ::: "memory"); C3_STAMP(); } while (0)
#else
#define C3_STAMP2() do {} while (0)
#endif
#ifdef PAPR_C3_TRACE_K                              // four more stamps inside k_run: entry, fragments landed, blocks issued, results out
#define C3_STAMPK() do { asm volatile("" ::: "memory"); C3_STAMP(); } while (0)
#else
#define C3_STAMPK() do {} while (0)
#endif
#else
#define C3_STAMP() do {} while (0)
#define C3_STAMP2() do {} while (0)
#define C3_STAMPK() do {} while (0)
#endif

__device__ __forceinline__ long uniform64(long v) {       // a wave-uniform value the compiler keeps in scalar registers and does not move out of loops
    int lo = __builtin_amdgcn_readfirstlane((int)v), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

// ONE: the reduced-precision mode (PAPR_GEMM_MODE=h1, the counterpart of the reference's fp16 autocast, models/attn.py:248): one
// f16 product per fp32 product -- only the hi planes of weights and activations are loaded, multiplied and written.
template <bool DGRAD, bool ONE>
__global__ __launch_bounds__(C3_THREADS, 2) void mlp_chain3_kernel(ChainArgs p, int iters, int generic_only, int fused_on) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool k_first = wn < 4;                    // waves w and w + 4 share a SIMD
#ifdef PAPR_H3_TRACE
    int trace_slot = 0;
#endif
    // Every piece below derives its lane addresses from its own opaque copy of the lane number: values the compiler can compute
    // once ahead of the slot loop live across everything (twenty-odd registers of offsets and masks), and the k-loop has
    // none to spare.
#define C3_LANE() int lane = lane0; asm volatile("" : "+v"(lane))
    // LDS layouts (bytes inside a tile's planes):
    //   multiplying: A fragment of 32-row tile i, k-step ks: row = 32 i + (lane & 31), 16-byte chunk 2 ks + (lane >> 5); row r of the
    //     tile lives in block r / RB at row r % RB; its 16-byte chunks are XOR-ed with r & 15 in the planes, r & 7 in the overlay
    //   dump: accumulator run (i, g) = row 32 i + (lane & 31), fp32 chunk 8 wn + 2 g + (lane >> 5) of the row's overlay (64 chunks:
    //     the first 32 over the hi row, the rest over the lo row)
    //   rows: this wave owns block wn; lane holds columns 4 lane .. 4 lane + 3 of a row = chunk lane & 31 of half lane >> 5
    float* const inv_all = reinterpret_cast<float*>(smem + 2 * C3_TILE_BYTES);      // [2][64]

    // ---- this wave's slice of the current layer: fragment (n-tile t, k-step s) starts at ((t * ksteps + s) * 64 + lane) * 8 halfs
    // The fragments live in a[0:127] BY NAME, outside the compiler's allocation: k-step ks: hi = a[8 ks : 8 ks + 3], lo =
    // a[8 ks + 4 : 8 ks + 7]; the loads and matrix instructions that touch them are inline asm.  The compiler keeps 128 VGPRs
    // (SIRegisterInfo halves the unified budget of a kernel whose inline asm names AGPRs).  It does not know these registers
    // are taken between the asm statements: it moves values of its own into AGPRs only when it runs out of VGPRs, so the
    // build fails if the object code holds a v_accvgpr instruction (papr_amd/build.py) -- keep the pressure below 128.
    // (As ordinary variables -- VGPR or AGPR class -- the 160 long-lived registers, fragments + accumulators, were shuffled
    // through scratch at every loop boundary: 200-350 spilled registers per kernel.)  The compiler does not see these loads
    // in flight either: k_run waits for them itself.
    auto frag_base = [&](const ChainLayer& L, const _Float16* w) {
        const int t = 32 * wn < L.N ? wn : 0;       // (a wave without columns in this layer: any valid address)
        return reinterpret_cast<const char*>(w) + (size_t)(t * L.ksteps) * 1024;
    };
#define C3_WH0 "a[0:3]"
#define C3_WL0 "a[4:7]"
#define C3_CH0 "a0", "a1", "a2", "a3"
#define C3_CL0 "a4", "a5", "a6", "a7"
#define C3_WCLOB0 C3_CH0, C3_CL0
#define C3_WH1 "a[8:11]"
#define C3_WL1 "a[12:15]"
#define C3_CH1 "a8", "a9", "a10", "a11"
#define C3_CL1 "a12", "a13", "a14", "a15"
#define C3_WCLOB1 C3_CH1, C3_CL1
#define C3_WH2 "a[16:19]"
#define C3_WL2 "a[20:23]"
#define C3_CH2 "a16", "a17", "a18", "a19"
#define C3_CL2 "a20", "a21", "a22", "a23"
#define C3_WCLOB2 C3_CH2, C3_CL2
#define C3_WH3 "a[24:27]"
#define C3_WL3 "a[28:31]"
#define C3_CH3 "a24", "a25", "a26", "a27"
#define C3_CL3 "a28", "a29", "a30", "a31"
#define C3_WCLOB3 C3_CH3, C3_CL3
#define C3_WH4 "a[32:35]"
#define C3_WL4 "a[36:39]"
#define C3_CH4 "a32", "a33", "a34", "a35"
#define C3_CL4 "a36", "a37", "a38", "a39"
#define C3_WCLOB4 C3_CH4, C3_CL4
#define C3_WH5 "a[40:43]"
#define C3_WL5 "a[44:47]"
#define C3_CH5 "a40", "a41", "a42", "a43"
#define C3_CL5 "a44", "a45", "a46", "a47"
#define C3_WCLOB5 C3_CH5, C3_CL5
#define C3_WH6 "a[48:51]"
#define C3_WL6 "a[52:55]"
#define C3_CH6 "a48", "a49", "a50", "a51"
#define C3_CL6 "a52", "a53", "a54", "a55"
#define C3_WCLOB6 C3_CH6, C3_CL6
#define C3_WH7 "a[56:59]"
#define C3_WL7 "a[60:63]"
#define C3_CH7 "a56", "a57", "a58", "a59"
#define C3_CL7 "a60", "a61", "a62", "a63"
#define C3_WCLOB7 C3_CH7, C3_CL7
#define C3_WH8 "a[64:67]"
#define C3_WL8 "a[68:71]"
#define C3_CH8 "a64", "a65", "a66", "a67"
#define C3_CL8 "a68", "a69", "a70", "a71"
#define C3_WCLOB8 C3_CH8, C3_CL8
#define C3_WH9 "a[72:75]"
#define C3_WL9 "a[76:79]"
#define C3_CH9 "a72", "a73", "a74", "a75"
#define C3_CL9 "a76", "a77", "a78", "a79"
#define C3_WCLOB9 C3_CH9, C3_CL9
#define C3_WH10 "a[80:83]"
#define C3_WL10 "a[84:87]"
#define C3_CH10 "a80", "a81", "a82", "a83"
#define C3_CL10 "a84", "a85", "a86", "a87"
#define C3_WCLOB10 C3_CH10, C3_CL10
#define C3_WH11 "a[88:91]"
#define C3_WL11 "a[92:95]"
#define C3_CH11 "a88", "a89", "a90", "a91"
#define C3_CL11 "a92", "a93", "a94", "a95"
#define C3_WCLOB11 C3_CH11, C3_CL11
#define C3_WH12 "a[96:99]"
#define C3_WL12 "a[100:103]"
#define C3_CH12 "a96", "a97", "a98", "a99"
#define C3_CL12 "a100", "a101", "a102", "a103"
#define C3_WCLOB12 C3_CH12, C3_CL12
#define C3_WH13 "a[104:107]"
#define C3_WL13 "a[108:111]"
#define C3_CH13 "a104", "a105", "a106", "a107"
#define C3_CL13 "a108", "a109", "a110", "a111"
#define C3_WCLOB13 C3_CH13, C3_CL13
#define C3_WH14 "a[112:115]"
#define C3_WL14 "a[116:119]"
#define C3_CH14 "a112", "a113", "a114", "a115"
#define C3_CL14 "a116", "a117", "a118", "a119"
#define C3_WCLOB14 C3_CH14, C3_CL14
#define C3_WH15 "a[120:123]"
#define C3_WL15 "a[124:127]"
#define C3_CH15 "a120", "a121", "a122", "a123"
#define C3_CL15 "a124", "a125", "a126", "a127"
#define C3_WCLOB15 C3_CH15, C3_CL15
#define C3_CAT_(a, b) a##b
#define C3_CAT(a, b) C3_CAT_(a, b)
#define C3_OFF0 "0"
#define C3_OFF1 "1024"
#define C3_OFF2 "2048"
#define C3_OFF3 "3072"
#define C3_WLOAD(ks, q, bh, bl)                                                                                                 \
    do {                                                                                                                        \
        if constexpr (ONE)                                                                                                      \
            asm volatile("global_load_dwordx4 " C3_CAT(C3_WH, ks) ", %0, %1 offset:" C3_CAT(C3_OFF, q)                               \
                         : : "v"(w_lane), "s"((bh) + ((ks) >> 2) * 4096) : C3_CAT(C3_CH, ks), "memory");                        \
        else                                                                                                                    \
            asm volatile("global_load_dwordx4 " C3_CAT(C3_WH, ks) ", %0, %1 offset:" C3_CAT(C3_OFF, q) "\n\t"                      \
                         "global_load_dwordx4 " C3_CAT(C3_WL, ks) ", %0, %2 offset:" C3_CAT(C3_OFF, q)                               \
                         : : "v"(w_lane), "s"((bh) + ((ks) >> 2) * 4096), "s"((bl) + ((ks) >> 2) * 4096) : C3_CAT(C3_WCLOB, ks), "memory"); \
    } while (0)
#define C3_WLOAD_IF(ks, q, n, bh, bl) if ((ks) < (n)) C3_WLOAD(ks, q, bh, bl)
#define C3_WLOAD_ALL(n, bh, bl)                                                                                                 \
    C3_WLOAD_IF(0, 0, n, bh, bl); C3_WLOAD_IF(1, 1, n, bh, bl); C3_WLOAD_IF(2, 2, n, bh, bl); C3_WLOAD_IF(3, 3, n, bh, bl);     \
    C3_WLOAD_IF(4, 0, n, bh, bl); C3_WLOAD_IF(5, 1, n, bh, bl); C3_WLOAD_IF(6, 2, n, bh, bl); C3_WLOAD_IF(7, 3, n, bh, bl);     \
    C3_WLOAD_IF(8, 0, n, bh, bl); C3_WLOAD_IF(9, 1, n, bh, bl); C3_WLOAD_IF(10, 2, n, bh, bl); C3_WLOAD_IF(11, 3, n, bh, bl);   \
    C3_WLOAD_IF(12, 0, n, bh, bl); C3_WLOAD_IF(13, 1, n, bh, bl); C3_WLOAD_IF(14, 2, n, bh, bl); C3_WLOAD_IF(15, 3, n, bh, bl)

    // ---- multiply the tile in `planes` by layer l; with ln >= 0 each k-step's registers are refilled with layer ln's
    // fragment as soon as the k-step is done
    f32x16 acc[NI];
    auto k_run = [&](const char* planes, int l, int ln) {
        const ChainLayer& L = p.L[l];
        const int ksteps = L.ksteps;
        const bool live = 32 * wn < L.N;
        const char *nh = nullptr, *nl = nullptr;
        int nks = 0;
        if (ln >= 0) { nh = frag_base(p.L[ln], p.L[ln].w_hi); nl = frag_base(p.L[ln], p.L[ln].w_lo); nks = p.L[ln].ksteps; }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        C3_LANE();
        const int arow = lane & 31, ax = arow & 15;
        const unsigned ab = (unsigned)((arow / RB) * C3_BLK_BYTES + (arow % RB) * 512 + (((lane >> 5) ^ (ax & 1)) * 16));
        const unsigned axr = (unsigned)((ax & ~1) * 16);
        const unsigned w_lane = (unsigned)lane * 16u;   // (weight loads: a wave-uniform base + this lane offset)
        C3_STAMPK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this layer's fragments (requested a slot ago) have landed
        C3_STAMPK();
        if (live && ksteps == KS) {
            // ---- the hot form (256-wide input): ONE asm statement for the whole k-loop (chain3_kloop.inc, generated by
            // scripts/gen_chain3_kloop.py; the why is in that script).  One wave keeps the matrix pipe busy only if its matrix
            // instructions issue back to back (scripts/probes/mfma_chain_rate.hip: 33.5 cycles per instruction; with the four
            // A-fragment reads of the next k-step in a bunch between the blocks: 52; one per gap with hand-counted waits: 35.2,
            // mfma_block_pattern.hip): every gap between two of them holds exactly one memory instruction -- the four LDS reads
            // of k-step ks + 2's fragments, IN PLACE behind the last use of each register (10-12 matrix instructions of lead with
            // two buffers: an LDS round trip beside seven other busy waves is longer than one block), and the two weight loads that
            // refill the k-step's registers with the next layer's fragment behind their last use.
            const unsigned pb = (unsigned)(size_t)planes + ab;
            unsigned ad[8];                         // LDS address of k-step j's fragments (k-step j + 8: + 256)
#pragma unroll
            for (int j = 0; j < 8; ++j) ad[j] = pb + (((unsigned)j * 32u) ^ axr);
            half8 f00, f01, f02, f03, f10, f11, f12, f13;       // fragment buffers (asm temporaries)
#define C3_KLOOP_OPERANDS                                                                                                       \
            [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [f00] "=&v"(f00), [f01] "=&v"(f01), [f02] "=&v"(f02), [f03] "=&v"(f03),         \
            [f10] "=&v"(f10), [f11] "=&v"(f11), [f12] "=&v"(f12), [f13] "=&v"(f13)                                              \
            : [ad0] "v"(ad[0]), [ad1] "v"(ad[1]), [ad2] "v"(ad[2]), [ad3] "v"(ad[3]), [ad4] "v"(ad[4]), [ad5] "v"(ad[5]),       \
              [ad6] "v"(ad[6]), [ad7] "v"(ad[7]), [wv] "v"(w_lane), [bh0] "s"(nh), [bh1] "s"(nh + 4096), [bh2] "s"(nh + 8192),  \
              [bh3] "s"(nh + 12288), [bl0] "s"(nl), [bl1] "s"(nl + 4096), [bl2] "s"(nl + 8192), [bl3] "s"(nl + 12288)
            if (nks == KS) {
                if constexpr (ONE) asm volatile(C3_KLOOP1_LD : C3_KLOOP_OPERANDS : C3_KLOOP_AGPRS, "memory");
                else asm volatile(C3_KLOOP3_LD : C3_KLOOP_OPERANDS : C3_KLOOP_AGPRS, "memory");
            } else {
                if constexpr (ONE) asm volatile(C3_KLOOP1_NL : C3_KLOOP_OPERANDS : "memory");
                else asm volatile(C3_KLOOP3_NL : C3_KLOOP_OPERANDS : "memory");
                C3_WLOAD_ALL(nks, nh, nl);          // (a narrower next layer: its fragments in a bunch)
            }
#undef C3_KLOOP_OPERANDS
        } else {
            // ---- any other width: the plain form (the compiler places the LDS reads and their waits)
            half8 ah[2][NI], al[2][NI];
            auto load_a = [&](int ks, half8 (&qh)[NI], half8 (&ql)[NI]) {
                ks = ks < ksteps ? ks : ksteps - 1;
                const unsigned o = ab + (((unsigned)ks * 32u) ^ axr);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    qh[i] = *reinterpret_cast<const half8*>(planes + i * 32768 + o);
                    if constexpr (!ONE) ql[i] = *reinterpret_cast<const half8*>(planes + i * 32768 + o + C3_LO);
                }
            };
            if (live) load_a(0, ah[0], al[0]);
            // six matrix instructions of a k-step: hi.lo, lo.hi, hi.hi for both row tiles (the order of chain.hip / chain2.hip per
            // accumulator); then the registers of the k-step take the next layer's fragment
#define C3_KSTEP(ks, q)                                                                                                         \
            if (live && (ks) < ksteps) {                                                                                        \
                load_a((ks) + 1, ah[((ks) + 1) & 1], al[((ks) + 1) & 1]);                                                       \
                if constexpr (ONE)                                                                                              \
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, " C3_CAT(C3_WH, ks) ", %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, " C3_CAT(C3_WH, ks) ", %3, %1" \
                                 : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[(ks) & 1][0]), "v"(ah[(ks) & 1][1]));                    \
                else                                                                                                            \
                asm volatile("v_mfma_f32_32x32x16_f16 %0, " C3_CAT(C3_WH, ks) ", %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, " C3_CAT(C3_WH, ks) ", %5, %1\n\t" \
                             "v_mfma_f32_32x32x16_f16 %0, " C3_CAT(C3_WL, ks) ", %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, " C3_CAT(C3_WL, ks) ", %3, %1\n\t" \
                             "v_mfma_f32_32x32x16_f16 %0, " C3_CAT(C3_WH, ks) ", %2, %0\n\tv_mfma_f32_32x32x16_f16 %1, " C3_CAT(C3_WH, ks) ", %3, %1"       \
                             : "+v"(acc[0]), "+v"(acc[1])                                                                       \
                             : "v"(ah[(ks) & 1][0]), "v"(ah[(ks) & 1][1]), "v"(al[(ks) & 1][0]), "v"(al[(ks) & 1][1]));        \
            }                                                                                                                   \
            C3_WLOAD_IF(ks, q, nks, nh, nl)
            asm volatile("s_nop 1" ::: "memory");   // (the zeroed accumulators: VALU write -> matrix read)
            C3_KSTEP(0, 0); C3_KSTEP(1, 1); C3_KSTEP(2, 2); C3_KSTEP(3, 3); C3_KSTEP(4, 0); C3_KSTEP(5, 1); C3_KSTEP(6, 2); C3_KSTEP(7, 3);
            C3_KSTEP(8, 0); C3_KSTEP(9, 1); C3_KSTEP(10, 2); C3_KSTEP(11, 3); C3_KSTEP(12, 0); C3_KSTEP(13, 1); C3_KSTEP(14, 2); C3_KSTEP(15, 3);
#undef C3_KSTEP
        }
        C3_STAMPK();
        // the last results leave the matrix pipe 16 passes after issue; hipcc does not count wait states behind inline asm
        // (the hot form ends with the same wait states inside its statement)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
        C3_STAMPK();
    };

    // ---- a multiply-first wave carries its accumulators through its row phases: 32 registers next to the 128 of the weights
    // leave 96, the row phases want 115 -- and a compiler short of VGPRs helps itself to AGPRs (see above).  So 28 of the 32
    // wait in LDS meanwhile (the last 28 KB of the CU's 160), four stay.
    char* const park0 = smem + 2 * C3_TILE_BYTES + 2 * C3_ROWS * sizeof(float) + (wn & 3) * C3_PARK_BYTES;
    float kept[4];                                  // (the four that stay: plain registers, so that both 16-register tuples are free meanwhile)
    auto park_acc = [&]() {
        C3_LANE();
        char* const park = park0 + lane * 16;
#pragma unroll
        for (int f = 0; f < C3_PARK_F4; ++f) {
            const f32x16& a = acc[f >> 2];
            const int e = 4 * (f & 3);
            *reinterpret_cast<float4*>(park + f * 1024) = make_float4(a[e], a[e + 1], a[e + 2], a[e + 3]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) kept[e] = acc[1][12 + e];
    };
    auto unpark_acc = [&]() {                       // (defines all 32: the compiler sees the accumulators dead during the row phases)
        C3_LANE();
        const char* const park = park0 + lane * 16;
#pragma unroll
        for (int f = 0; f < C3_PARK_F4; ++f) {
            const float4 v = *reinterpret_cast<const float4*>(park + f * 1024);
            f32x16& a = acc[f >> 2];
            const int e = 4 * (f & 3);
            a[e] = v.x; a[e + 1] = v.y; a[e + 2] = v.z; a[e + 3] = v.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[1][12 + e] = kept[e];
    };

    // ---- the accumulators (raw, still scaled) go into the fp32 overlay of the tile they were computed from
    auto dump = [&](char* planes, int l) {
        if (32 * wn >= p.L[l].N) return;
        C3_LANE();
        const int arow = lane & 31, hh = lane >> 5;
        const unsigned db = (unsigned)((arow / RB) * C3_BLK_BYTES + (arow % RB) * 512), dx = (unsigned)(arow & 7);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c64 = 8 * wn + 2 * g;                 // (+ hh: the chunk's lowest bit)
                const unsigned c32 = (unsigned)(c64 & 31) + (unsigned)hh;
                *reinterpret_cast<float4*>(planes + i * 32768 + db + (c64 >> 5) * C3_LO + ((c32 ^ dx) * 16)) =
                    make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
            }
    };

    // ---- split a row held across the wave (lane: 4 columns) into the A planes of block wn, row u
    // (wp = (lane >> 1) * 16, wq = (lane & 1) * 8 of the caller: see there)
    // (in_k: this lane's columns are inside the next layer's input; a constant true in the hot forms, so that the splits of four
    // rows interleave)
    // (ONE: gdst = where this lane's four halfs go in memory as well -- the f16 rows the weight-gradient reads -- or null)
    auto write_planes = [&](char* planes, unsigned wp, unsigned wq, int u, const float4& v, float sc, bool in_k, _Float16* gdst = nullptr) {
        if (in_k) {
            char* dst = planes + wn * C3_BLK_BYTES + u * 512 + (wp ^ (unsigned)(((wn * RB + u) & 15) * 16)) + wq;
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
                *reinterpret_cast<half4*>(dst + C3_LO) = lo;
            }
        }
    };

    // ---- stage the input rows of a tile (coalesced: one row per load instruction; this wave: rows RB wn .. RB wn + 7):
    // LayerNorm core in front of the run, row maxima, scales, split
    auto stage = [&](char* planes, float* inv_tab, long m0) {
        // (row numbers: 32-bit scalars, and nothing here is a loop invariant -- addresses and masks computed ahead would live
        // across the k-loops, next to 160 registers of weights and accumulators)
        const int M32 = (int)p.M;
        int r0 = __builtin_amdgcn_readfirstlane((int)m0) + wn * RB;
        C3_LANE();
        const int c = 4 * lane;
        const unsigned wp = (unsigned)(lane >> 1) * 16u, wq = (unsigned)(lane & 1) * 8u;
        asm volatile("" : "+s"(r0));
        const int kpad = p.L[0].k1steps * 16;
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int m = r0 + q;
            m = m < M32 ? m : M32 - 1;              // rows beyond M: the last row again
            const float* rowp = p.A0 + (long)m * p.lda0;  // wave-uniform: scalar base + one lane offset
            v[q] = c < p.K0 ? *reinterpret_cast<const float4*>(rowp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        }
    };

    // ---- what the next row phases read from memory before they can start (this lane's bias values / sign word), requested a
    // slot ahead: an L2 round trip at the top of every row phase was 5 % of the slot
    float4 b4n = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned swn = 0u;
    auto rows_prefetch = [&](long m0, int l) {
        C3_LANE();
        l = __builtin_amdgcn_readfirstlane(l);
        const ChainLayer& L = p.L[l];
        if (!DGRAD) {
            b4n = make_float4(0.f, 0.f, 0.f, 0.f);
            if (L.bias && 4 * lane < L.N) b4n = *reinterpret_cast<const float4*>(L.bias + 4 * lane);
        } else {
            const int r0 = __builtin_amdgcn_readfirstlane((int)m0) + wn * RB;
            swn = 0u;
            if (L.sign_bits != nullptr && r0 < (int)p.M) swn = L.sign_bits[(long)(r0 / RB) * (RB * 8) + lane];
        }
    };

    // ---- row phases of layer l for the tile at rows m0 (dumped into `planes`): rows RB wn .. RB wn + 7, a whole row across the
    // wave, RQ rows at a time.  Straight-line code matters (chain2.hip): the hot flag combinations are instantiated with the
    // flags as constants, everything else takes the generic instantiation.
    auto p_run = [&](char* planes, float* inv_tab, long m0, int l) {
        l = __builtin_amdgcn_readfirstlane(l);
        const int M32 = (int)p.M;
        int t0 = __builtin_amdgcn_readfirstlane((int)m0);
        C3_LANE();
        const int c = 4 * lane;
        const unsigned wp = (unsigned)(lane >> 1) * 16u, wq = (unsigned)(lane & 1) * 8u, rc = (unsigned)(lane & 31) * 16u;
        asm volatile("" : "+s"(l), "+s"(t0));       // (as in stage)
        const int r0 = t0 + wn * RB;
        const ChainLayer& L = p.L[l];
        const int N = L.N;
        const bool more = l + 1 < p.n_layers;
        const float slope = L.act == PAPR_ACT_RELU ? 0.f : (L.act == PAPR_ACT_LEAKY_RELU ? 0.2f : 1.f);
        const bool rt_half = ONE && L.c_half != 0 && L.C != nullptr;          // f16 rows out (with the split for the next layer), no fp32 rows
        const bool rt_store = L.C != nullptr && !rt_half, rt_bits = L.sign_bits != nullptr, rt_rmax = L.rowmax != nullptr;
        const bool rt_norm = !DGRAD && !more && p.norm_stats != nullptr;
        const bool rt_full = N == 256 && t0 + C3_ROWS <= M32;
        const bool mask_rows = DGRAD && !rt_bits && L.mask != nullptr;
        const bool blk_in = r0 < M32;                                           // this wave's block has rows inside M
        float4 b4 = b4n;                            // (requested by rows_prefetch one slot ago)
        const int kpad_next = more ? p.L[l + 1].k1steps * 16 : 0;
        float* const crow = L.C + (long)r0 * L.ldc;                             // row RB wn (wave-uniform: scalar base + lane offset)
        const long ldc = L.ldc;
        // sign words: lane l keeps ITS 4 x 8 bits of the wave's 8 rows (first value in the top bit), 256 contiguous bytes per
        // wave and layer -- written by the forward run, read back by the data-gradient run, whose lanes hold the same columns
        // of the same rows
        unsigned* const sgn = L.sign_bits + (long)(r0 / RB) * (RB * 8) + lane;
        unsigned sw = swn;
        // they are waited for HERE: a wait inside the batch loop would also wait for the row stores of the batch before it (loads
        // and stores share vmcnt)
        asm volatile("" : "+v"(b4.x), "+v"(b4.y), "+v"(b4.z), "+v"(b4.w), "+v"(sw));
        float* rmax = L.rowmax;                     // (opaque copy: the compiler reloaded the pointer from the kernel arguments in
        asm volatile("" : "+s"(rmax));              // every batch -- an s_load and a wait for everything the LDS still owes)
        const char* const blk = planes + wn * C3_BLK_BYTES + (lane >> 5) * C3_LO;
        float* const inv_w = inv_tab + wn * RB;
        auto rows = [&](auto cfg) {
            using Cfg = decltype(cfg);
            const bool f_store = Cfg::store == 2 ? rt_store : Cfg::store == 1;
            const bool f_half = ONE && (Cfg::store == 2 ? rt_half : Cfg::store == 3);
            const bool f_bits = Cfg::bits == 2 ? rt_bits : Cfg::bits == 1;
            const bool f_rmax = Cfg::rmax == 2 ? rt_rmax : Cfg::rmax == 1;
            const bool f_more = Cfg::more == 2 ? more : Cfg::more == 1;
            const bool f_norm = Cfg::norm == 2 ? rt_norm : Cfg::norm == 1;
            const bool f_full = Cfg::full == 2 ? rt_full : Cfg::full == 1;
            const bool col_ok = f_full || c < N;
            unsigned sword = DGRAD ? sw : 0u;
            // rows in flight: all eight of the wave in the hot forms (twice the independent chains for the scheduler, one LDS round
            // trip instead of two: inference -4 %, data-gradient -1 %, h1 -3 % on the 4-layer run) except the training forward
            // one (eight row stores in a bunch: +2 %); four where the flags are looked at at run time
            constexpr int RQL = Cfg::full == 1 && (DGRAD || ONE || Cfg::store == 0) ? 8 : RQ;
#pragma unroll 1
            for (int ub = 0; ub < RB; ub += RQL) {
                float4 r[RQL];
#pragma unroll
                for (int q = 0; q < RQL; ++q)
                    r[q] = *reinterpret_cast<const float4*>(blk + (ub + q) * 512 + (rc ^ (unsigned)(((ub + q) & 7) * 16)));
                const float4 inv4 = *reinterpret_cast<const float4*>(inv_w + ub);
                const float4 inv4b = RQL == 8 ? *reinterpret_cast<const float4*>(inv_w + ub + 4) : inv4;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the rows are in registers: their bytes may be overwritten
                C3_STAMP2();
#pragma unroll
                for (int q = 0; q < RQL; ++q) {
                    const int u = ub + q;
                    const float4 iv = q < 4 ? inv4 : inv4b;
                    const float inv = (q & 3) == 0 ? iv.x : (q & 3) == 1 ? iv.y : (q & 3) == 2 ? iv.z : iv.w;
                    if (DGRAD) {
                        r[q] = make_float4(r[q].x * inv, r[q].y * inv, r[q].z * inv, r[q].w * inv);
                        if (f_bits) {
                            r[q].x = (int)sword < 0 ? r[q].x : r[q].x * slope; r[q].y = (int)(sword << 1) < 0 ? r[q].y : r[q].y * slope;
                            r[q].z = (int)(sword << 2) < 0 ? r[q].z : r[q].z * slope; r[q].w = (int)(sword << 3) < 0 ? r[q].w : r[q].w * slope;
                            sword <<= 4;
                        } else if (mask_rows) {
                            int row = r0 + u;
                            row = row < M32 ? row : M32 - 1;
                            const float4 a4 = col_ok ? *reinterpret_cast<const float4*>(L.mask + (long)row * L.ld_mask + c) : make_float4(0.f, 0.f, 0.f, 0.f);
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
                        const int row = r0 + u;
                        const float mean = wave_sum((r[q].x + r[q].y) + (r[q].z + r[q].w)) / (float)N;
                        float4 dl = col_ok ? make_float4(r[q].x - mean, r[q].y - mean, r[q].z - mean, r[q].w - mean) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float sigma = sqrtf(wave_sum((dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w)) / (float)(N - 1));
                        const float rinv = 1.0f / (sigma + p.norm_eps);
                        r[q] = make_float4(dl.x * rinv, dl.y * rinv, dl.z * rinv, dl.w * rinv);
                        if (lane == 0 && row < M32) { p.norm_stats[(long)row * 2] = rinv; p.norm_stats[(long)row * 2 + 1] = sigma; }
                    }
                    if (f_store && col_ok && (f_full || r0 + u < M32)) *reinterpret_cast<float4*>(crow + u * ldc + c) = r[q];
                    if (!DGRAD && f_bits) {
                        sword = (sword << 1) | (r[q].x > 0.f ? 1u : 0u); sword = (sword << 1) | (r[q].y > 0.f ? 1u : 0u);
                        sword = (sword << 1) | (r[q].z > 0.f ? 1u : 0u); sword = (sword << 1) | (r[q].w > 0.f ? 1u : 0u);
                    }
                }
                C3_STAMP2();
                if (f_more || f_rmax) {
#pragma unroll
                  for (int h = 0; h < RQL; h += 4) {
                    float mx[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) mx[q] = fmaxf(fmaxf(fabsf(r[h + q].x), fabsf(r[h + q].y)), fmaxf(fabsf(r[h + q].z), fabsf(r[h + q].w)));
#ifndef C3_X_NOMAX                                  // (timing experiments: pieces of the row phases left out, results wrong)
                    wave_max4(mx[0], mx[1], mx[2], mx[3]);
#endif
                    float smx[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) smx[q] = last_lane(mx[q]);
                    C3_STAMP2();
                    if (f_rmax) {
                        float mx4 = 0.f;
                        put4(mx4, 0, smx[0], smx[1], smx[2], smx[3]);
                        if (lane < 4 && (f_full || r0 + ub + h + lane < M32)) rmax[r0 + ub + h + lane] = mx4;
                    }
                    if (f_more) {
                        float inv_n[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float sc = scale_from_max(__float_as_uint(smx[q]), inv_n[q]);
#ifndef C3_X_NOSPLIT
                            _Float16* gdst = nullptr;
                            if (f_half && (f_full || r0 + ub + h + q < M32))
                                gdst = reinterpret_cast<_Float16*>(L.C) + (long)(r0 + ub + h + q) * ldc + c;
                            write_planes(planes, wp, wq, ub + h + q, r[h + q], sc, Cfg::full == 1 || c < kpad_next, gdst);  // (full: N = 256 = the next layer's input width)
#else
                            asm volatile("" :: "s"(sc));
#endif
                        }
                        *reinterpret_cast<float4*>(inv_w + ub + h) = make_float4(inv_n[0], inv_n[1], inv_n[2], inv_n[3]);
                    }
                  }
                }
                C3_STAMP2();
            }
            if (!DGRAD && f_bits && (f_full || blk_in)) *sgn = sword;
        };
        // hot combinations (everything 256 wide, tile inside M): training middle layer / inference middle layer / data-gradient
        if (!generic_only && rt_full && more && !rt_norm && !mask_rows && rt_store && rt_bits && rt_rmax) rows(RowCfg<1, 1, 1, 1, 0, 1>());
        else if (ONE && !generic_only && rt_full && more && !rt_norm && !mask_rows && rt_half && rt_bits && rt_rmax) rows(RowCfg<3, 1, 1, 1, 0, 1>());
        else if (!generic_only && !DGRAD && rt_full && more && !rt_norm && !rt_store && !rt_bits && !rt_rmax) rows(RowCfg<0, 0, 0, 1, 0, 1>());
        else rows(RowCfg<2, 2, 2, 2, 2, 2>());
    };

    // ---- the hot slot as ONE statement: the k-loop of tile A (layer l) and the row phases of tile B (layer pl) interleaved
    // instruction by instruction (chain3_fused.inc, generated by scripts/gen_chain3_fused.py -- the why and the how are there).
    // Every wave does both at once; no multiply-first / rows-first split, no parking.  mode: 0 training forward middle layer,
    // 1 inference middle layer, 2 data-gradient middle layer (what rows() calls its hot forms).
    auto fused_slot = [&](char* kpl, int l, int ln, char* ppl, float* pinv, long pm0, int pl, int mode) {
        C3_LANE();
        l = __builtin_amdgcn_readfirstlane(l);
        pl = __builtin_amdgcn_readfirstlane(pl);
        const ChainLayer& LP = p.L[pl];
        const int r0 = __builtin_amdgcn_readfirstlane((int)pm0) + wn * RB;
        const char *nh = nullptr, *nl = nullptr;
        if (ln >= 0) { nh = frag_base(p.L[ln], p.L[ln].w_hi); nl = frag_base(p.L[ln], p.L[ln].w_lo); }
        const bool ld = ln >= 0 && p.L[ln].ksteps == KS;
        const int arow = lane & 31, ax = arow & 15;
        const unsigned pb = (unsigned)(size_t)kpl + (unsigned)((arow / RB) * C3_BLK_BYTES + (arow % RB) * 512 + (((lane >> 5) ^ (ax & 1)) * 16));
        const unsigned axr = (unsigned)((ax & ~1) * 16);
        const unsigned wv = (unsigned)lane * 16u;
        const unsigned rdb = (unsigned)(size_t)ppl + wn * C3_BLK_BYTES + (lane >> 5) * C3_LO, rc = (unsigned)(lane & 31) * 16u;
        const unsigned wrb = (unsigned)(size_t)ppl + wn * C3_BLK_BYTES + (unsigned)(lane & 1) * 8u, wp = (unsigned)(lane >> 1) * 16u;
        const float4 b4 = b4n;
        unsigned sw = DGRAD ? swn : 0u;
        const bool halfrows = ONE && LP.c_half != 0;            // f16 rows out: 2-byte elements
        const char* crow = reinterpret_cast<const char*>(LP.C) + (long)r0 * LP.ldc * (halfrows ? 2 : 4);
        const unsigned ldcb = (unsigned)LP.ldc * (halfrows ? 2u : 4u);
        const char* rmp = reinterpret_cast<const char*>(LP.rowmax + r0);
        const float slope = LP.act == PAPR_ACT_RELU ? 0.f : (LP.act == PAPR_ACT_LEAKY_RELU ? 0.2f : 1.f);
        const unsigned invb = (unsigned)(size_t)(pinv + wn * RB), wnb = (unsigned)(wn & 1) * 128u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this layer's fragments (requested a slot ago) have landed
#define C3_FUSED_OPERANDS                                                                                                       \
        [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [sw] "+v"(sw)                                                                   \
        : [pb] "v"(pb), [axr] "v"(axr), [wv] "v"(wv), [rdb] "v"(rdb), [rc] "v"(rc), [wrb] "v"(wrb), [wp] "v"(wp),                \
          [b0] "v"(b4.x), [b1] "v"(b4.y), [b2] "v"(b4.z), [b3] "v"(b4.w), [nh] "s"(nh), [nl] "s"(nl), [crow] "s"(crow),           \
          [ldcb] "s"(ldcb), [rmp] "s"(rmp), [slope] "s"(slope), [invb] "s"(invb), [wnb] "s"(wnb)
        if constexpr (ONE && DGRAD) {
            if (halfrows) {
                if (ld) asm volatile(C3_FUSED1H_DGRAD_LD : C3_FUSED_OPERANDS : C3_FUSED_AGPRS, C3_FUSED_CLOBBERS);
                else asm volatile(C3_FUSED1H_DGRAD_NL : C3_FUSED_OPERANDS : C3_FUSED_CLOBBERS);
            } else {
                if (ld) asm volatile(C3_FUSED1_DGRAD_LD : C3_FUSED_OPERANDS : C3_FUSED_AGPRS, C3_FUSED_CLOBBERS);
                else asm volatile(C3_FUSED1_DGRAD_NL : C3_FUSED_OPERANDS : C3_FUSED_CLOBBERS);
            }
        } else if constexpr (ONE) {
            if (halfrows) {
                if (ld) asm volatile(C3_FUSED1H_FWD_LD : C3_FUSED_OPERANDS : C3_FUSED_AGPRS, C3_FUSED_CLOBBERS);
                else asm volatile(C3_FUSED1H_FWD_NL : C3_FUSED_OPERANDS : C3_FUSED_CLOBBERS);
            } else {
                if (ld) asm volatile(C3_FUSED1_FWD_LD : C3_FUSED_OPERANDS : C3_FUSED_AGPRS, C3_FUSED_CLOBBERS);
                else asm volatile(C3_FUSED1_FWD_NL : C3_FUSED_OPERANDS : C3_FUSED_CLOBBERS);
            }
        } else if constexpr (DGRAD) {
            if (ld) asm volatile(C3_FUSED_DGRAD_LD : C3_FUSED_OPERANDS : C3_FUSED_AGPRS, C3_FUSED_CLOBBERS);
            else asm volatile(C3_FUSED_DGRAD_NL : C3_FUSED_OPERANDS : C3_FUSED_CLOBBERS);
        } else if (mode == 0) {
            if (ld) asm volatile(C3_FUSED_FWD_LD : C3_FUSED_OPERANDS : C3_FUSED_AGPRS, C3_FUSED_CLOBBERS);
            else asm volatile(C3_FUSED_FWD_NL : C3_FUSED_OPERANDS : C3_FUSED_CLOBBERS);
        } else {
            if (ld) asm volatile(C3_FUSED_INF_LD : C3_FUSED_OPERANDS : C3_FUSED_AGPRS, C3_FUSED_CLOBBERS);
            else asm volatile(C3_FUSED_INF_NL : C3_FUSED_OPERANDS : C3_FUSED_CLOBBERS);
        }
#undef C3_FUSED_OPERANDS
        if (!ld && ln >= 0) { const int nks = p.L[ln].ksteps; const unsigned w_lane = wv; C3_WLOAD_ALL(nks, nh, nl); }     // (a narrower next layer)
        if (!DGRAD && mode == 0) LP.sign_bits[(long)(r0 / RB) * (RB * 8) + lane] = sw;
    };

    // ---- schedule
    const int n_layers = p.n_layers;
    long pair = blockIdx.x;
    const long pstride = gridDim.x;
    stage(smem, inv_all, 2 * pair * C3_ROWS);
    {
        const ChainLayer& L0 = p.L[0];
        const char* bh = frag_base(L0, L0.w_hi);
        const char* bl = frag_base(L0, L0.w_lo);
        const int n0 = L0.ksteps;
        const unsigned w_lane = (unsigned)lane0 * 16u;
        C3_WLOAD_ALL(n0, bh, bl);
    }
    lds_barrier();                                  // planes of the first X ready
#ifdef C3_YPRIO
    if (!k_first) __builtin_amdgcn_s_setprio(C3_YPRIO);      // (the younger wave of every SIMD loses the issue arbitration to its partner at equal priority)
#endif

    int it = 0, l = 0, h = 0;                       // slot: multiply tile h (X, Y) of pair `it` by layer l | row phases of the other tile
    const int n_slots = iters * n_layers * 2;
#pragma unroll 1
    for (int slot = 0; slot < n_slots; ++slot) {
        const long mX = 2 * pair * C3_ROWS, mY = mX + C3_ROWS;
        char* const kpl = smem + h * C3_TILE_BYTES;
        char* const ppl = smem + (1 - h) * C3_TILE_BYTES;
        float* const pinv = inv_all + (1 - h) * C3_ROWS;
        // what the row-phase half of this slot is
        int pl;                                     // layer whose rows are finished (-1: none)
        long pm0, sm0 = -1;                         // its tile; the tile staged into those planes afterwards (-1: none)
        if (h == 0) {
            if (l > 0) { pl = l - 1; pm0 = mY; }
            else { pl = it > 0 ? n_layers - 1 : -1; pm0 = mY - 2 * pstride * C3_ROWS; sm0 = mY; }
        } else {
            pl = l; pm0 = mX;
            if (l + 1 == n_layers && it + 1 < iters) sm0 = mX + 2 * pstride * C3_ROWS;
        }
        const int ln = h == 1 ? (l + 1 < n_layers ? l + 1 : 0) : -1;
        C3_STAMP();
        // is this a hot slot (both layers 256 wide, the rows' tile inside M, a middle layer, one of the three hot flag sets)?
        int fmode = -1;
        if (!generic_only && fused_on && pl >= 0 && sm0 < 0 && p.L[l].ksteps == KS && p.L[l].N == 256 && p.L[pl].N == 256 && pl + 1 < n_layers &&
            pm0 + C3_ROWS <= p.M) {
            const ChainLayer& LP = p.L[pl];
            const bool st = LP.C != nullptr, bi = LP.sign_bits != nullptr, rm = LP.rowmax != nullptr;
            if (st && bi && rm) fmode = DGRAD ? 2 : 0;
            else if (!ONE && fused_on > 1 && !DGRAD && !st && !bi && !rm) fmode = 1;        // (inference: the two-role slot with eight rows in flight is 5 % faster: PAPR_C3_FUSED=2 to compare)
        }
        if (fmode >= 0) {
            fused_slot(kpl, l, ln, ppl, pinv, pm0, pl, fmode);
            C3_STAMP(); C3_STAMP(); C3_STAMP();
        } else {
        // (two copies of the multiplying code around one copy of the row phases: on every path from the row phases to the dump
        // the accumulators are redefined, so the compiler has their 32 registers for the row phases)
#ifndef C3_NOK
        if (k_first) {
            k_run(kpl, l, ln);
            park_acc();
        }
#endif
        C3_STAMP();
#ifndef C3_NOROWS
#ifdef C3_SPLIT                                     // (timing experiment: waves 0-3 only multiply, waves 4-7 only do row phases)
        if (!k_first)
#endif
        if (pl >= 0) p_run(ppl, pinv, pm0, pl);
#endif
        if (sm0 >= 0) stage(ppl, pinv, sm0);
        C3_STAMP();
#ifndef C3_NOK
        if (k_first) unpark_acc();
#ifndef C3_SPLIT
        else k_run(kpl, l, ln);
#endif
#endif
        }
        {   // the row phases of the NEXT slot (the slot state advanced by one), or the ones behind the loop
            int h2 = h + 1, l2 = l;
            long pair2 = pair;
            if (h2 == 2) { h2 = 0; if (++l2 == n_layers) { l2 = 0; pair2 += pstride; } }
            const long nX = 2 * pair2 * C3_ROWS, nY = nX + C3_ROWS;
            int npl; long npm0;
            if (slot + 1 == n_slots) { npl = n_layers - 1; npm0 = mY; }
            else if (h2 == 0) { if (l2 > 0) { npl = l2 - 1; npm0 = nY; } else { npl = n_layers - 1; npm0 = nY - 2 * pstride * C3_ROWS; } }
            else { npl = l2; npm0 = nX; }
            rows_prefetch(npm0, npl);
        }
        C3_STAMP();
        lds_barrier();                              // the multiplied tile's planes are dead, the other tile's are written
        C3_STAMP();
        dump(kpl, l);
        C3_STAMP();
        lds_barrier();
        if (++h == 2) { h = 0; if (++l == n_layers) { l = 0; ++it; pair += pstride; } }
    }
    // the last Y of this workgroup still has its last layer's rows to finish
    p_run(smem + C3_TILE_BYTES, inv_all + C3_ROWS, 2 * (pair - pstride) * C3_ROWS + C3_ROWS, n_layers - 1);
}

}  // namespace

#ifdef PAPR_H3_TRACE
extern "C" int papr_chain3_trace_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain3_trace), sizeof(long long) * 1024) == hipSuccess ? 0 : 1; }
#endif

size_t papr_chain3_lds_bytes() { return C3_LDS_BYTES; }

int papr_launch_chain3(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s) {
    PAPR_REQUIRE(a.n_layers >= 1 && a.n_layers <= CHAIN_MAX_LAYERS, "mlp_chain3: %d layers", a.n_layers);
    PAPR_REQUIRE(a.K0 % 4 == 0 && a.lda0 % 4 == 0 && a.K0 <= 256, "mlp_chain3: input width %d", a.K0);
    for (int l = 0; l < a.n_layers; ++l) {
        PAPR_REQUIRE(a.L[l].k1steps == a.L[l].ksteps, "mlp_chain3: skip layers run on mlp_chain_kernel");
        PAPR_REQUIRE(a.L[l].ksteps >= 1 && a.L[l].ksteps <= KS, "mlp_chain3: %d k-steps", a.L[l].ksteps);
        PAPR_REQUIRE(!a.L[l].c_half || (a.one_product && l + 1 < a.n_layers && a.L[l].rowmax && a.L[l].ldc % 4 == 0),
                     "mlp_chain3: layer %d: f16 rows need the one-product mode, a following layer and the row maxima", l);
    }
    if (a.M <= 0) return 0;
    const long tiles = (a.M + C3_ROWS - 1) / C3_ROWS;
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain3_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C3_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain3_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C3_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain3_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C3_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain3_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C3_LDS_BYTES);
        attr_set = true;
    }
    const long pairs = (tiles + 1) / 2;             // a workgroup carries two tiles at a time
    const unsigned grid = (unsigned)(pairs < n_cu ? pairs : n_cu);
    const int iters = (int)((pairs + grid - 1) / grid);
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin2(dgrad ? 10 : 9, a.M, a.n_layers, a.K0, bytes, flops, s);
    static const int generic_only = getenv("PAPR_C2_GENERIC") ? atoi(getenv("PAPR_C2_GENERIC")) : 0;      // (test switch: the hot instantiations off)
    const int g_d = generic_only == 1 || generic_only == 3, g_f = generic_only == 1 || generic_only == 2;
    static const int fused_on = getenv("PAPR_C3_FUSED") ? atoi(getenv("PAPR_C3_FUSED")) : 1;        // (A/B switch: 0 = the two-role slots everywhere)
    if (a.one_product) {
        if (dgrad) mlp_chain3_kernel<true, true><<<dim3(grid), dim3(C3_THREADS), C3_LDS_BYTES, s>>>(a, iters, g_d, fused_on);
        else mlp_chain3_kernel<false, true><<<dim3(grid), dim3(C3_THREADS), C3_LDS_BYTES, s>>>(a, iters, g_f, fused_on);
    } else {
        if (dgrad) mlp_chain3_kernel<true, false><<<dim3(grid), dim3(C3_THREADS), C3_LDS_BYTES, s>>>(a, iters, g_d, fused_on);
        else mlp_chain3_kernel<false, false><<<dim3(grid), dim3(C3_THREADS), C3_LDS_BYTES, s>>>(a, iters, g_f, fused_on);
    }
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("mlp_chain3");
    return 0;
}
