// K3: the embedding MLPs as MFMA-tiled GEMMs over the (ray, neighbour) rows.
//
// Replaces MLP.forward (reference models/mlp.py:47-59: nn.Linear + ReLU/LeakyReLU chains of
// FeedForward, models/attn.py:113-117, and w_k / w_q, models/attn.py:217-218) and the autograd
// backward of those layers.  Two kernels carry all of it:
//
//   gemm_nt : C[m][n] = epi( sum_k A[m][k] * W[n][k] )      forward layers and data-gradients
//             (the data-gradient uses the transposed weight so that both run the same kernel)
//   gemm_tn : dW[n][k] = sum_m G[m][n] * X[m][k]             weight gradients, reduction over the
//             512,000 rows split across the chip (one slab per workgroup, then a slab reduction)
//
// fp32 parity mode: v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation, 256 FLOP/clk/CU,
// 157 TFLOP/s peak).  An MFMA of this type occupies the matrix pipe for 64 cycles and needs only one
// VGPR per operand, so LDS bandwidth is irrelevant; what matters is that every SIMD always has an
// MFMA to issue.  gemm_nt therefore runs two 8-wave workgroups per CU (128 VGPRs, 55 KB LDS each):
// while one is in its global->LDS hand-over or its epilogue, the other keeps the matrix pipe busy.
//
// LDS layout for gemm_nt: row-major [rows][32 + 4] fp32 (144-byte rows).  A lane reads 4 consecutive
// k with one ds_read_b128 and feeds them to 4 successive MFMAs: lanes 0-31 take k..k+3, lanes 32-63
// take k+4..k+7, so one 16-byte read per operand feeds 256 cycles of matrix work, and the 36-float
// pitch maps the 16 lanes of a read group onto 16 distinct 4-bank slots (conflict-free).
#include "papr_common.h"
#include "h3_common.h"
#include "chain.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <string.h>

namespace {

constexpr int BK = 32;        // default k-slab per stage (single-buffered variant)

struct NTArgs {
    const float* A;  long lda;  int K1;
    const float* A2; long lda2; int K2;
    const float* W;  int ldw;   int wcol2;
    const float* bias;
    const float* mask_src; long ld_mask;
    int act;          // forward: activation; dgrad: activation whose derivative masks the result
    int dgrad;        // 0: C = act(acc + bias)   1: C = acc * act'(mask_src)
    int accumulate;   // C += ...
    float* C; long ldc;
    long M; int N;
    const unsigned* amax_in;   // split-f16 mode: per row m, the bit pattern of max_k |A[m][k]| (M words); 0 -> scale 1
    const unsigned* amax_in2;  // split-f16 mode with a second K segment: the same for A2 (the row's scale covers both), or NULL
    _Float16* planes;          // split-f16 mode: scratch for the pre-split weight (H3_PLANE_HALFS halfs)
    unsigned* amax_out;        // split-f16 mode: per row, atomicMax'ed with the bit pattern of max_n |C[m][n]| (M words, zeroed) or NULL
};

// BKT: k-slab per stage (32 or 16).  DB: two LDS buffers and ONE barrier per stage (compute slab t,
// write slab t+1 into the other buffer, barrier) instead of barrier / write / barrier.
template <int BM, int BN, int TM, int TN, int BKT = BK, bool DB = false>
__global__ __launch_bounds__(64 * (BM / (32 * TM)) * (BN / (32 * TN)), (BM / (32 * TM)) * (BN / (32 * TN)) / 2)
void gemm_nt_kernel(NTArgs p) {
    constexpr int PITCH = BKT + 4;              // LDS row pitch (floats): 16-lane read groups hit 16 distinct 4-bank slots
    constexpr int KQ = BKT / 4;                 // float4 per slab row
    constexpr int WN = BN / (32 * TN);          // waves along n
    constexpr int WM = BM / (32 * TM);          // waves along m
    constexpr int NTHR = 64 * WM * WN;          // 4 or 8 waves; two workgroups per CU either way
    static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves per workgroup");
    constexpr int A_LD = BM * KQ / NTHR;        // float4 loads per thread for the A slab
    constexpr int W_LD = BN * KQ / NTHR;
    constexpr int BUF = (BM + BN) * PITCH;      // floats per LDS buffer

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [BM][PITCH]
    float* Ws = smem + BM * PITCH;     // [BN][PITCH]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const int nt1 = (p.K1 + BKT - 1) / BKT;
    const int nt2 = p.A2 ? (p.K2 + BKT - 1) / BKT : 0;
    const int nt = nt1 + nt2;

    float4 ra[A_LD], rw[W_LD];

    auto load_slab = [&](int kt) {
        const float* src; long ld; int klim, k0, wcol;
        if (kt < nt1) { src = p.A; ld = p.lda; klim = p.K1; k0 = kt * BKT; wcol = k0; }
        else { src = p.A2; ld = p.lda2; klim = p.K2; k0 = (kt - nt1) * BKT; wcol = p.wcol2 + k0; }
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            int f = tid + NTHR * i;
            int row = f / KQ, kq = (f % KQ) * 4;
            long m = m0 + row;
            ra[i] = (m < p.M && k0 + kq < klim) ? *reinterpret_cast<const float4*>(src + m * ld + k0 + kq)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < W_LD; ++i) {
            int f = tid + NTHR * i;
            int row = f / KQ, kq = (f % KQ) * 4;
            int n = n0 + row;
            rw[i] = (n < p.N && k0 + kq < klim) ? *reinterpret_cast<const float4*>(p.W + (long)n * p.ldw + wcol + kq)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            int f = tid + NTHR * i;
            *reinterpret_cast<float4*>(As + buf * BUF + (f / KQ) * PITCH + (f % KQ) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < W_LD; ++i) {
            int f = tid + NTHR * i;
            *reinterpret_cast<float4*>(Ws + buf * BUF + (f / KQ) * PITCH + (f % KQ) * 4) = rw[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const float* a_base = As + (wm * TM * 32 + (lane & 31)) * PITCH + 4 * (lane >> 5);
    const float* w_base = Ws + (wn * TN * 32 + (lane & 31)) * PITCH + 4 * (lane >> 5);

    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) load_slab(kt + 1);
        const int cur = DB ? (kt & 1) * BUF : 0;
#pragma unroll
        for (int kb = 0; kb < BKT; kb += 8) {
            float4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(a_base + cur + i * 32 * PITCH + kb);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const float4*>(w_base + cur + j * 32 * PITCH + kb);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, wf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, wf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, wf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, wf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (DB) {
            if (kt + 1 < nt) store_slab((kt + 1) & 1);      // the other buffer: its last readers passed the previous barrier
            __syncthreads();
        } else {
            __syncthreads();
            if (kt + 1 < nt) {
                store_slab(0);
                __syncthreads();
            }
        }
    }

    // Epilogue.  The accumulator layout is D[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31]: a
    // lane owns one column, so direct stores would be 4-byte.  Each 32x32 tile is instead bounced
    // through a per-wave LDS patch (free after the last barrier of the k loop) and leaves as 16-byte
    // row segments: 4 store instructions per tile instead of 16, and the activation-derivative mask
    // of the data-gradient / the old C of an accumulating call come in as 16-byte loads too.
    constexpr int EP = 36;
    float* patch = smem + wave * (32 * EP);
    const float slope = p.act == PAPR_ACT_RELU ? 0.f : (p.act == PAPR_ACT_LEAKY_RELU ? 0.2f : 1.f);
    const int pr = lane >> 3, pc = (lane & 7) * 4;     // this lane's (row, first column) inside a tile, +8 rows per step
    const int wr_off = (4 * (lane >> 5)) * EP + (lane & 31);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + pc;
        const bool col_ok = col < p.N;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!p.dgrad && p.bias && col_ok) b4 = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) patch[wr_off + ((e & 3) + 8 * (e >> 2)) * EP] = acc[i][j][e];
            const long row0 = m0 + (wm * TM + i) * 32 + pr;
            float4 v[4], aux[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const float4*>(patch + (pr + 8 * t) * EP + pc);
            const bool need_aux = p.dgrad ? (p.mask_src != nullptr) : false;
            if (need_aux || p.accumulate) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long row = row0 + 8 * t;
                    const bool ok = col_ok && row < p.M;
                    if (need_aux) {
                        aux[t] = ok ? *reinterpret_cast<const float4*>(p.mask_src + row * p.ld_mask + col) : make_float4(1.f, 1.f, 1.f, 1.f);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const long row = row0 + 8 * t;
                if (!(col_ok && row < p.M)) continue;
                float4 r = v[t];
                if (p.dgrad) {
                    if (need_aux) {
                        r.x *= aux[t].x > 0.f ? 1.f : slope; r.y *= aux[t].y > 0.f ? 1.f : slope;
                        r.z *= aux[t].z > 0.f ? 1.f : slope; r.w *= aux[t].w > 0.f ? 1.f : slope;
                    }
                } else {
                    r.x += b4.x; r.y += b4.y; r.z += b4.z; r.w += b4.w;
                    // (+0.f turns the -0 of a negative input times slope 0 into the +0 torch's relu returns)
                    r.x = r.x > 0.f ? r.x : r.x * slope + 0.f; r.y = r.y > 0.f ? r.y : r.y * slope + 0.f;
                    r.z = r.z > 0.f ? r.z : r.z * slope + 0.f; r.w = r.w > 0.f ? r.w : r.w * slope + 0.f;
                }
                float4* dst = reinterpret_cast<float4*>(p.C + row * p.ldc + col);
                if (p.accumulate) { float4 o = *dst; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
                *dst = r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt_h3: the same product at fp32-grade accuracy on the f16 matrix pipe (16x the fp32 MFMA rate).
// Every fp32 operand x is split on the fly into two halves, hi = f16(x s), lo = f16(x s - hi), with a
// power-of-two scale s that puts the tensor's max|x| just under 2^14 (exact, undone in the epilogue).
// hi + lo carries 22 mantissa bits; a.b ~ hi_a hi_b + hi_a lo_b + lo_a hi_b is three
// v_mfma_f32_32x32x16_f16 with fp32 accumulation (each product is exact in fp32), the dropped lo.lo
// term is 2^-22 relative.  Three matrix instructions instead of sixteen: the layer turns HBM-bound.
// Persistent streaming structure.  A slab's matrix work (~1.2k cycles) is far shorter than a loaded HBM
// round trip (8-11k cycles measured), and the first-touch latency plus the store tail of a one-tile
// workgroup cost more than its eight slabs.  So: one workgroup per CU walks many row tiles; the slabs of
// all its tiles form ONE stream; slab g+D is requested while slab g is multiplied (register ring, D = 4,
// unconditional loads so that hipcc counts vmcnt instead of draining it); LDS is double-buffered, so a
// slab costs one LDS-only barrier and the hi/lo split of the next slab overlaps the other waves' MFMAs;
// a tile's epilogue runs while the next tile's slabs are already in flight.
#ifdef PAPR_H3_TRACE
__device__ long long g_h3_trace[256];
#define H3_STAMP(slot) do { if (blockIdx.x == 100 && threadIdx.x == 0 && (slot) < 256) g_h3_trace[slot] = __builtin_readcyclecounter(); } while (0)
extern "C" int papr_h3_trace_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h3_trace), sizeof(long long) * 256) == hipSuccess ? 0 : 1; }
#else
#define H3_STAMP(slot) do {} while (0)
#endif

template <int BM, int BN, int TM, int TN, int D>
__global__ __launch_bounds__(512, 2) void gemm_nt_h3_kernel(NTArgs p, int tiles_m, const _Float16* __restrict__ w_hi, const _Float16* __restrict__ w_lo, int w_ksteps) {
    H3_STAMP(0);
    constexpr int WN = BN / (32 * TN), WM = BM / (32 * TM);
    static_assert(WM * WN == 8, "eight waves per workgroup");
    constexpr int NTHR = 512, BKT = 32, KQ = BKT / 4;
    constexpr int HP = BKT + 8;                 // LDS row pitch in halfs (80 B): conflict-free ds_read_b128
    constexpr int A_LD = BM * KQ / NTHR;
    constexpr int PLANE_A = BM * HP, BUF = 2 * PLANE_A;   // halfs per LDS buffer: only the A rows go through LDS
    constexpr int EP = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);         // two slab buffers: [Ah | Al] each
    float* patch_base = smem + BUF;                            // (2 * BUF halfs = BUF floats) epilogue patches behind them
    float* inv_tab = patch_base + 8 * 32 * EP;                 // [2][BM] 1/scale of the rows of the tile in flight (by tile parity)
    unsigned* rmax_tab = reinterpret_cast<unsigned*>(inv_tab + 2 * BM);   // [2][BM] running max|C| bits of those rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN;
    const int nt1 = (p.K1 + BKT - 1) / BKT;
    const int nt2 = p.A2 ? (p.K2 + BKT - 1) / BKT : 0;
    const int nt = nt1 + nt2;
    const int my_tiles = (tiles_m - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * nt;

    // Every ROW of A gets its own power-of-two scale (row max -> [2^13, 2^14)), undone per row in the epilogue:
    // rows are independent dot products, so results do not depend on which other rows share the launch
    // (chunked evaluation stays bitwise equal to whole-image evaluation) and a row of tiny gradients keeps
    // its 22 bits next to a row of large ones.
    float4 ra[D][A_LD];
    unsigned rs[D][A_LD];                        // row-max bit patterns, fetched with the slab
    // The weight arrives pre-split AND in MFMA fragment order (split_weight_kernel): a wave's B fragment of
    // (n-tile, k-step) is one coalesced 1 KB load from L2 straight into registers; it never touches LDS,
    // which cuts the LDS write traffic per slab from 61 KB to 20 KB.
    half8 wfh[2][TN][BKT / 16], wfl[2][TN][BKT / 16];
    // (tile, k-slab) of stream position g; nt is small, so the division is a handful of scalar ops
    auto slab_coords = [&](int g, long& m0, int& kt) -> int {
        int ti = __builtin_amdgcn_readfirstlane(g / nt);
        kt = g - ti * nt;
        m0 = ((long)blockIdx.x + (long)ti * gridDim.x) * BM;
        return ti;
    };
    // per-thread constants of the slab copy: row inside the tile and k offset inside the slab
    int a_row[A_LD], a_kq[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) { int f = tid + NTHR * i; a_row[i] = f / KQ; a_kq[i] = (f % KQ) * 4; }
    // fragment (n-tile t, k-step s) of the planes starts at ((t * w_ksteps + s) * 64 + lane) * 8 halfs
    auto load_wfrag = [&](int g, half8 (&qh)[TN][BKT / 16], half8 (&ql)[TN][BKT / 16]) {
        int ti = __builtin_amdgcn_readfirstlane(g / nt);
        int kt = g - ti * nt;
        int kcol = kt < nt1 ? kt * BKT : p.wcol2 + (kt - nt1) * BKT;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int sidx = 0; sidx < BKT / 16; ++sidx) {
                long o = ((long)((n0 / 32 + wn * TN + j) * w_ksteps + kcol / 16 + sidx) * 64 + lane) * 8;
                qh[j][sidx] = *reinterpret_cast<const half8*>(w_hi + o);
                ql[j][sidx] = *reinterpret_cast<const half8*>(w_lo + o);
            }
    };
    auto load_slab = [&](int g, float4 (&qa)[A_LD], unsigned (&qs)[A_LD]) {
        long m0; int kt;
        slab_coords(g, m0, kt);
        const float* src; long ld; int klim, k0, wcol;
        if (kt < nt1) { src = p.A; ld = p.lda; klim = p.K1; k0 = kt * BKT; wcol = k0; }
        else { src = p.A2; ld = p.lda2; klim = p.K2; k0 = (kt - nt1) * BKT; wcol = p.wcol2 + k0; }
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            long m = m0 + a_row[i];
            m = m < p.M ? m : p.M - 1;
            int kk = k0 + a_kq[i] < klim ? k0 + a_kq[i] : 0;
            qa[i] = *reinterpret_cast<const float4*>(src + m * ld + kk);
            qs[i] = p.amax_in2 ? max(p.amax_in[m], p.amax_in2[m]) : p.amax_in[m];
        }
        (void)wcol;
    };
    auto store_slab = [&](int g, const float4 (&qa)[A_LD], const unsigned (&qs)[A_LD]) {
        long m0; int kt;
        const int par = slab_coords(g, m0, kt) & 1;
        const int klim = kt < nt1 ? p.K1 : p.K2, k0 = (kt < nt1 ? kt : kt - nt1) * BKT;
        _Float16* Ah = lds + (g & 1) * BUF;
        _Float16* Al = Ah + PLANE_A;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            int off = a_row[i] * HP + a_kq[i];
            const bool ok = m0 + a_row[i] < p.M && k0 + a_kq[i] < klim;
            const int ea = qs[i] ? (int)((qs[i] >> 23) & 0xff) : 127 + 13;
            const float a_scale = pow2_from_biased(127 + 13 - (ea - 127));
            half4 hi, lo;
            split4(qa[i], ok ? a_scale : 0.f, hi, lo);
            *reinterpret_cast<half4*>(Ah + off) = hi;
            *reinterpret_cast<half4*>(Al + off) = lo;
            if (kt == 0 && a_kq[i] == 0) {          // first slab of a tile: publish the row's 1/scale, reset its max
                inv_tab[par * BM + a_row[i]] = pow2_from_biased(127 - 13 + (ea - 127));
                rmax_tab[par * BM + a_row[i]] = 0u;
            }
        }
    };

    f32x16 acc[TM][TN];
    auto clear_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    clear_acc();

    // epilogue of one finished tile: un-scale, bias / activation or derivative mask, 16-byte row stores
    // (per-wave LDS patch turns the column-per-lane accumulators into rows), max|C| for the next layer
    float* patch = patch_base + wave * (32 * EP);
    const float slope = p.act == PAPR_ACT_RELU ? 0.f : (p.act == PAPR_ACT_LEAKY_RELU ? 0.2f : 1.f);
    const int pr = lane >> 3, pc = (lane & 7) * 4;
    const int wr_off = (4 * (lane >> 5)) * EP + (lane & 31);
    const bool need_aux = p.dgrad && p.mask_src != nullptr;
    float4 bias4[TN];                           // loaded once: inside the tile loop it would be a dependent L2 trip per tile
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + pc;
        bias4[j] = (!p.dgrad && p.bias && col < p.N) ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto epilogue = [&](long m0, int par) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + pc;
            const bool col_ok = col < p.N;
            const float4 b4 = bias4[j];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[wr_off + ((e & 3) + 8 * (e >> 2)) * EP] = acc[i][j][e];
                const long row0 = m0 + (wm * TM + i) * 32 + pr;
                float4 v[4], aux[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const float4*>(patch + (pr + 8 * t) * EP + pc);
                if (need_aux) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const long row = row0 + 8 * t;
                        aux[t] = (col_ok && row < p.M) ? *reinterpret_cast<const float4*>(p.mask_src + row * p.ld_mask + col) : make_float4(1.f, 1.f, 1.f, 1.f);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long row = row0 + 8 * t;
                    if (!(col_ok && row < p.M)) continue;
                    const int rl = (wm * TM + i) * 32 + pr + 8 * t;          // row inside the tile
                    const float inv = inv_tab[par * BM + rl];
                    float4 r = v[t];
                    r.x *= inv; r.y *= inv; r.z *= inv; r.w *= inv;
                    if (p.dgrad) {
                        if (need_aux) {
                            r.x *= aux[t].x > 0.f ? 1.f : slope; r.y *= aux[t].y > 0.f ? 1.f : slope;
                            r.z *= aux[t].z > 0.f ? 1.f : slope; r.w *= aux[t].w > 0.f ? 1.f : slope;
                        }
                    } else {
                        r.x += b4.x; r.y += b4.y; r.z += b4.z; r.w += b4.w;
                        r.x = r.x > 0.f ? r.x : r.x * slope + 0.f; r.y = r.y > 0.f ? r.y : r.y * slope + 0.f;
                        r.z = r.z > 0.f ? r.z : r.z * slope + 0.f; r.w = r.w > 0.f ? r.w : r.w * slope + 0.f;
                    }
                    float4* dst = reinterpret_cast<float4*>(p.C + row * p.ldc + col);
                    if (p.accumulate) { float4 o = *dst; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
                    *dst = r;
                    if (p.amax_out)                                            // |x| bit patterns order like unsigned ints
                        atomicMax(rmax_tab + par * BM + rl, __float_as_uint(fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w)))));
                }
            }
        }
    };

    // lane (i = lane&31, h = lane>>5) feeds row i, k = 8h..8h+7 of each 16-wide k step (A and W alike)
    const int frag = (lane & 31) * HP + 8 * (lane >> 5);
#pragma unroll
    for (int u = 0; u < D; ++u)
        if (u < total) load_slab(u, ra[u], rs[u]);
    if (total > 0) { store_slab(0, ra[0], rs[0]); load_wfrag(0, wfh[0], wfl[0]); }
    lds_barrier();
    long pend_m0 = -1;                           // tile whose row maxima wait in LDS for their write-out
    int pend_par = 0;
    auto flush_rowmax = [&]() {                  // runs after the barrier that ended the tile's epilogue
        if (pend_m0 >= 0 && p.amax_out && tid < BM && pend_m0 + tid < p.M)
            atomicMax(p.amax_out + pend_m0 + tid, rmax_tab[pend_par * BM + tid]);
        pend_m0 = -1;
    };
    for (int g0 = 0; g0 < total; g0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int g = g0 + u;
            if (g >= total) break;
            flush_rowmax();
            const _Float16* Ah = lds + (g & 1) * BUF;
            const _Float16* Al = Ah + PLANE_A;
            if (g + 1 < total) load_wfrag(g + 1, wfh[(u + 1) & 1], wfl[(u + 1) & 1]);     // next slab's B fragments (L2)
#pragma unroll
            for (int ks = 0; ks < BKT; ks += 16) {
                half8 ah[TM], al[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    int o = (wm * TM + i) * 32 * HP + frag + ks;
                    ah[i] = *reinterpret_cast<const half8*>(Ah + o);
                    al[i] = *reinterpret_cast<const half8*>(Al + o);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const half8 wh = wfh[u & 1][j][ks / 16], wl = wfl[u & 1][j][ks / 16];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], wh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], wh, acc[i][j], 0, 0, 0);
                    }
                }
            }
            // slab g+1 (requested D-1 iterations ago) goes into the other LDS buffer, whose last readers
            // passed the previous barrier; ring set u is then free for slab g+D
            H3_STAMP(1 + g * 5);
            if (g + 1 < total) store_slab(g + 1, ra[(u + 1) % D], rs[(u + 1) % D]);
            H3_STAMP(2 + g * 5);
            if (g + D < total) load_slab(g + D, ra[u], rs[u]);
            H3_STAMP(3 + g * 5);
            long m0; int kt;
            const int par = slab_coords(g, m0, kt) & 1;
            if (kt == nt - 1) {
                epilogue(m0, par);
                clear_acc();
                pend_m0 = m0; pend_par = par;
            }
            H3_STAMP(4 + g * 5);
            lds_barrier();
            H3_STAMP(5 + g * 5);
        }
    }
    flush_rowmax();
}

// out[m] = bit pattern of max_k |x[m][k]|, one wave per row (width <= 1024, a multiple of 4)
__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ x, long M, int width, long ld, unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float mx = 0.f;
    for (int c = lane * 4; c < width; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(x + m * ld + c);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    mx = wave_max(mx);
    if (lane == 0) out[m] = __float_as_uint(mx);
}

template <int BM, int BN, int TM, int TN, int BKT = BK, bool DB = false>
int launch_nt(const NTArgs& a, hipStream_t s) {
    dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.N + BN - 1) / BN));
    size_t lds = (size_t)(DB ? 2 : 1) * (BM + BN) * (BKT + 4) * sizeof(float);
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin(BN == 256 ? 0 : BN == 128 ? 1 : BN == 64 ? 2 : 3, a.M, a.N, a.K1 + (a.A2 ? a.K2 : 0), s);
    gemm_nt_kernel<BM, BN, TM, TN, BKT, DB><<<grid, dim3(64 * (BM / (32 * TM)) * (BN / (32 * TN))), lds, s>>>(a);
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("gemm_nt");
    return 0;
}

// 8 waves x (64x64) beats 4 waves x (128x64) on the 128x256 tile: 586 vs 628 us per 512000x256x256
// layer (4 waves per SIMD hide the slab hand-over and the epilogue better).  PAPR_NT_WAVES4 keeps the
// old shape reachable for A/B runs.
#define NT_VARIANT papr_switch(PAPR_SW_NT_VARIANT)          // A/B switch: 1 / 2 = double-buffered tilings, 3 = four waves per SIMD

// Which wide GEMMs run on the split-f16 kernels.  PAPR_GEMM_MODE = h3 (default) | layers | dgrad | fwd | f32.
//   h3   : everything below, and runs of consecutive layers fused into one kernel (chain.hip).
//   layers: forward layers, data-gradients (586 -> 369 us and 634 -> 483 us per 512000x256x256 layer) and
//          weight-gradients (gemm_tn_h3), one launch per layer.
//   dgrad: forward layers and data-gradients.
//   fwd  : forward layers only.
//   f32  : every GEMM on v_mfma_f32_32x32x2_f32.
//   h1   : see GEMM_ONE_PRODUCT below.
// All three pass the same parity suite (RGB / fused / attention within 1e-4 of the reference, gradients
// within 2e-3, bitwise chunk invariance, reference loss trajectory within 5e-6).  The one measurable
// difference: after the reference's three Adam steps the point positions agree to 5e-5 in f32 mode and to
// 1e-4 in the split modes (22-bit operands; Adam divides by |g| for near-zero gradients).
// The mode is an ARGUMENT of papr_mlp_fwd / papr_mlp_bwd / papr_mlp_bwd_needs_weight_t (include/papr_hip.h: PAPR_MLP_*): the library reads no
// environment variable for it and keeps no setting between calls.  Inside a call it sits in a thread-local context so that the helpers
// below need not all carry it; the entry points set it from their argument first thing.
//   h1 (PAPR_MLP_H1): h3, and the fused runs (chain4.hip) multiply one f16 product per fp32 product: the throughput mode that stands for
//   the reference's fp16 autocast of the attention block (models/attn.py:248, `use_amp: true`); own tolerance in the tests.
static thread_local int t_mode = 4;            // 0 f32, 1 fwd, 2 dgrad, 3 layers, 4 h3, 5 h1
static thread_local bool t_h3_f16_rows = false;   // PAPR_MLP_H3_F16ROWS: h3 whose fused runs keep f16 rows for their weight gradients (round 6's gated experiment)
static inline bool mode_from_arg(int32_t m) {  // false: not a mode of papr_hip.h
    t_h3_f16_rows = m == PAPR_MLP_H3_F16ROWS;
    switch (m) { case PAPR_MLP_F32: t_mode = 0; return true; case PAPR_MLP_FWD: t_mode = 1; return true; case PAPR_MLP_DGRAD: t_mode = 2; return true;
                 case PAPR_MLP_LAYERS: t_mode = 3; return true; case PAPR_MLP_H3: case PAPR_MLP_H3_F16ROWS: t_mode = 4; return true;
                 case PAPR_MLP_H1: t_mode = 5; return true;
                 // (round 6: the one-product runs carry one scale per row and run and keep f16 rows only; a call that wants fp32 rows between a run and
                 //  its weight gradients runs in the parity arithmetic)
                 case PAPR_MLP_H1_F32ROWS: t_mode = 4; return true; default: return false; }
}
#define GEMM_MODE t_mode
#define GEMM_ONE_PRODUCT (t_mode == 5)
#define GEMM_H3_FWD (t_mode >= 1)
#define GEMM_H3 (t_mode >= 2)
#define GEMM_H3_WGRAD (t_mode >= 3)
#define GEMM_CHAIN (t_mode >= 4)
static inline bool one_product_now() { return t_mode == 5; }

// Caller-provided scratch of the split-f16 mode, carved from the workspace argument of papr_mlp_fwd / _bwd:
// two per-row max|.| arrays (rows of the layer input / of its output, swapped after every layer) and the
// pre-split weight planes of the launch in flight.
constexpr size_t H3_PLANE_HALFS = (size_t)CHAIN_MAX_LAYERS * 2 * 256 * 256;      // hi + lo planes of a weight up to 512 x 704, or of a fused run of up to 8 layers of 256 x 256
struct H3Scratch {
    unsigned* amax[2];
    unsigned* amax_x;            // row maxima of the chain's input x (skip layers read x again)
    _Float16* planes;
    int cur = 0;
    H3Scratch(void* ws, long M) {
        amax[0] = static_cast<unsigned*>(ws);
        amax[1] = amax[0] + M;
        amax_x = amax[1] + M;
        planes = reinterpret_cast<_Float16*>(amax_x + M);
    }
    static size_t bytes(long M) { return ((size_t)3 * M * sizeof(unsigned) + H3_PLANE_HALFS * sizeof(_Float16) + 255) / 256 * 256; }
    unsigned* in() { return amax[cur]; }
    unsigned* out() { return amax[cur ^ 1]; }
    void swap() { cur ^= 1; }
};


// The input rows of a fused run, split ahead of it (chain.h: ChainArgs::sr_*): row m -> hi = f16(x s), lo = f16(x s - hi) with s the power of two that
// brings the row's maximum into [2^13, 2^14), columns [K, Kq) zero; 1 / s; the maximum.  Optionally the LayerNorm core in front of the run
// (FeedForward.innorm, models/attn.py:39-42) first: rows standardised over their first nw columns, (1 / (std + eps), std) to `stats`, the standardised
// rows written back over x when `writeback`.  THE SAME ARITHMETIC, instruction for instruction, as the staging of mlp_chain4_kernel (chain4.hip:
// stage_finish) -- a lane holds four columns of a row, a wave the row: sums of four in the order ((a + b) + (c + d)), then wave_sum -- so a run gives
// the same bits whichever of the two split its rows (tests/test_hip_chain_variants.py, switch PAPR_SW_C4_DMA).
struct SplitRowsArgs {
    float* x; long ld; int K, Kq; long M;
    _Float16* hi; _Float16* lo; float* inv; float* mx;
    int nw; float eps; float* stats; int writeback;
};
__global__ __launch_bounds__(256) void split_rows_kernel(SplitRowsArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = 4 * lane;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
    for (long m0 = wave * 4; m0 < a.M; m0 += n_waves * 4) {
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long m = m0 + q < a.M ? m0 + q : a.M - 1;
            v[q] = c < a.K ? *reinterpret_cast<const float4*>(a.x + m * a.ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long m = m0 + q;
            if (a.stats != nullptr) {
                const int wdt = a.nw;
                const bool i0 = c < wdt, i1 = c + 1 < wdt, i2 = c + 2 < wdt, i3 = c + 3 < wdt;
                const float mean = wave_sum(((i0 ? v[q].x : 0.f) + (i1 ? v[q].y : 0.f)) + ((i2 ? v[q].z : 0.f) + (i3 ? v[q].w : 0.f))) / (float)wdt;
                float4 dl = make_float4(i0 ? v[q].x - mean : 0.f, i1 ? v[q].y - mean : 0.f, i2 ? v[q].z - mean : 0.f, i3 ? v[q].w - mean : 0.f);
                const float sigma = sqrtf(wave_sum((dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w)) / (float)(wdt - 1));
                const float rinv = 1.0f / (sigma + a.eps);
                v[q] = make_float4(dl.x * rinv, dl.y * rinv, dl.z * rinv, dl.w * rinv);
                if (m < a.M) {
                    if (a.writeback && c < a.K) *reinterpret_cast<float4*>(a.x + m * a.ld + c) = v[q];
                    if (lane == 0) { a.stats[m * 2] = rinv; a.stats[m * 2 + 1] = sigma; }
                }
            }
            const float smx = wave_max(fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w))));
            float inv;
            const float sc = h3_scale_from_max(__float_as_uint(smx), inv);
            if (m < a.M) {
                if (c < a.Kq) {
                    half4 hi, lo;
                    split4(v[q], sc, hi, lo);
                    *reinterpret_cast<half4*>(a.hi + m * a.Kq + c) = hi;
                    *reinterpret_cast<half4*>(a.lo + m * a.Kq + c) = lo;
                }
                if (lane == 0) { a.inv[m] = inv; a.mx[m] = smx; }
            }
        }
    }
}

int launch_split_rows(const SplitRowsArgs& a, hipStream_t s) {
    PAPR_REQUIRE(a.K % 4 == 0 && a.K <= 256 && a.Kq % 16 == 0 && a.Kq >= a.K && a.Kq <= 256 && a.ld % 4 == 0, "split_rows: width %d (planes %d), row stride %ld", a.K, a.Kq, a.ld);
    if (a.M <= 0) return 0;
    const long blocks = (a.M + 15) / 16;
    const int cap = papr_cu_count() * 8;
    split_rows_kernel<<<dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, s>>>(a);
    PAPR_CHECK_LAUNCH("split_rows");
    return 0;
}

// scratch of the pre-split input rows of a fused run, behind everything else of a workspace: planes of up to 256 columns, 1 / scale, maximum
struct SRScratch {
    _Float16* hi; _Float16* lo; float* inv; float* mx;
    SRScratch(void* base, long M) {
        hi = static_cast<_Float16*>(base); lo = hi + (size_t)M * 256;
        inv = reinterpret_cast<float*>(lo + (size_t)M * 256); mx = inv + M;
    }
    static size_t bytes(long M) { return ((size_t)M * (2 * 256 * sizeof(_Float16) + 2 * sizeof(float)) + 255) / 256 * 256; }
    // what a workspace reserves for it: the planes only while the split-ahead switch is on (1,032 bytes per row: 3.3 GB of dead scratch at the
    // 3.2 M rows of a 160,000-ray evaluate chunk otherwise); always the M floats at its head that papr_mlp_fwd uses for the row means of a
    // LayerNorm core whose statistics nobody gave.  Callers size their workspace per call (papr_mlp_*_workspace_bytes), so a switch flipped
    // later is seen by the next call.
    static size_t reserve(long M) { return papr_switch(PAPR_SW_C4_DMA) ? bytes(M) : ((size_t)M * sizeof(float) + 255) / 256 * 256; }
};

int launch_row_absmax(const float* x, long M, int width, long ld, unsigned* out, hipStream_t s) {
    row_absmax_kernel<<<dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s>>>(x, M, width, ld, out);
    PAPR_CHECK_LAUNCH("row_absmax");
    return 0;
}

// W (N, ldw) fp32 -> hi/lo f16 planes in MFMA fragment order: element e of lane l of fragment (n-tile t,
// k-step ks) is W[32 t + (l & 31)][16 ks + 8 (l >> 5) + e]; rows / columns beyond the matrix are zero.
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ W, int N, int ncols, int ldw, int n_tiles, int ksteps,
                                                           _Float16* __restrict__ hi, _Float16* __restrict__ lo) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_tiles * ksteps * 512) return;
    int e = idx & 7, l = (idx >> 3) & 63, f = idx >> 9;
    int t = f / ksteps, ks = f - t * ksteps;
    int n = 32 * t + (l & 31), c = 16 * ks + 8 * (l >> 5) + e;
    float x = (n < N && c < ncols) ? W[(long)n * ldw + c] : 0.f;
    _Float16 h = (_Float16)x;
    hi[idx] = h;
    lo[idx] = (_Float16)(x - (float)h);
}

int launch_nt_h3(const NTArgs& a, _Float16* planes, hipStream_t s) {
    constexpr int BM = 128, BN = 256;
    const int tiles_m = (int)((a.M + BM - 1) / BM);
    const int n_cu = papr_cu_count();
    dim3 grid((unsigned)(tiles_m < n_cu ? tiles_m : n_cu), (unsigned)((a.N + BN - 1) / BN));   // one persistent workgroup per CU
    size_t lds = (size_t)2 * 2 * BM * 40 * sizeof(_Float16) + (size_t)8 * 32 * 36 * sizeof(float) + (size_t)4 * BM * sizeof(float);
    if (papr_first_on_device(PAPR_ONCE_NT_H3))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_h3_kernel<BM, BN, 2, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // pre-split the weight once per launch (64k-180k elements): the hot loop then copies f16 planes verbatim
    const int ncols = a.A2 ? a.wcol2 + a.K2 : a.K1;
    const int pitch = (ncols + 31) / 32 * 32, rows_pad = (a.N + BN - 1) / BN * BN;
    const int ksteps = pitch / 16, n_tiles = rows_pad / 32;
    PAPR_REQUIRE(!a.A2 || a.wcol2 % 32 == 0, "gemm_nt_h3: second K segment must start at a multiple of 32");
    PAPR_REQUIRE((size_t)2 * rows_pad * pitch <= H3_PLANE_HALFS, "gemm_nt_h3: weight %d x %d too large for the split scratch", a.N, ncols);
    PAPR_REQUIRE(!a.A2 || a.K1 % 32 == 0, "gemm_nt_h3: first K segment must be a multiple of 32 when a second one follows");
    PAPR_REQUIRE(planes, "gemm_nt_h3: no workspace");
    _Float16* w_hi = planes;
    _Float16* w_lo = planes + (size_t)rows_pad * pitch;
    split_weight_kernel<<<dim3((rows_pad * pitch + 255) / 256), dim3(256), 0, s>>>(a.W, a.N, ncols, a.ldw, n_tiles, ksteps, w_hi, w_lo);
    PAPR_CHECK_LAUNCH("split_weight");
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin(a.dgrad ? 7 : 6, a.M, a.N, a.K1 + (a.A2 ? a.K2 : 0), s);
    gemm_nt_h3_kernel<BM, BN, 2, 2, 4><<<grid, dim3(512), lds, s>>>(a, tiles_m, w_hi, w_lo, ksteps);
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("gemm_nt_h3");
    return 0;
}

int gemm_nt(const NTArgs& a, hipStream_t s) {
    PAPR_REQUIRE(a.K1 % 4 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0, "gemm_nt: K1/lda/ldw must be multiples of 4 (%d,%ld,%d)", a.K1, a.lda, a.ldw);
    PAPR_REQUIRE(!a.A2 || (a.K2 % 4 == 0 && a.lda2 % 4 == 0 && a.wcol2 % 4 == 0), "gemm_nt: segment-2 sizes must be multiples of 4");
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.N > 128) {
        if (a.amax_in) return launch_nt_h3(a, a.planes, s);
        if (NT_VARIANT == 3) return launch_nt<128, 256, 4, 2>(a, s);
        if (NT_VARIANT == 1) return launch_nt<128, 256, 2, 2, 16, true>(a, s);
        if (NT_VARIANT == 2) return launch_nt<128, 256, 2, 2, 32, true>(a, s);
        return launch_nt<128, 256, 2, 2>(a, s);
    }
    if (a.N > 64) return launch_nt<128, 128, 2, 2>(a, s);
    if (a.N > 32) return launch_nt<256, 64, 2, 2>(a, s);
    return launch_nt<256, 32, 2, 1>(a, s);
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: slab[s][n][k] = sum_{m in slice s} G[m][n] * X[m][k]   (N, K <= 256), bias slab = col sums of G
// 512 threads = 8 waves as 2 (n) x 4 (k); each wave owns a 128 x 64 corner of the 256 x 256 tile.
constexpr int TN_ROWS = 32;   // m rows per stage
constexpr int SLAB = 256;     // output tile edge

struct TNArgs {
    const float* G; long ldg; int N;
    const float* X; long ldx; int K;
    long M; long rows_per_slice;
    float* slab;       // [S][256][256]
    float* bias_slab;  // [S][256]
};

__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(TNArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Gs = smem;                     // [32][256]
    float* Xs = smem + TN_ROWS * SLAB;    // [32][256]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave >> 2, wk = wave & 3;     // wave's corner: n in [128 wn, +128), k in [64 wk, +64)
    const long mbeg = (long)blockIdx.x * p.rows_per_slice;
    long mend = mbeg + p.rows_per_slice;
    if (mend > p.M) mend = p.M;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float colsum = 0.f;

    // staging map: 32 rows x 64 float4 per operand = 2048 float4 -> 4 per thread per operand
    float4 rg[4], rx[4];
    auto load_stage = [&](long mb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int f = tid + 512 * i;
            int row = f >> 6, c = (f & 63) * 4;
            long m = mb + row;
            rg[i] = (m < mend && c < p.N) ? *reinterpret_cast<const float4*>(p.G + m * p.ldg + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            rx[i] = (m < mend && c < p.K) ? *reinterpret_cast<const float4*>(p.X + m * p.ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int f = tid + 512 * i;
            *reinterpret_cast<float4*>(Gs + (f >> 6) * SLAB + (f & 63) * 4) = rg[i];
            *reinterpret_cast<float4*>(Xs + (f >> 6) * SLAB + (f & 63) * 4) = rx[i];
        }
    };

    const float* g_base = Gs + (lane >> 5) * SLAB + wn * 128 + (lane & 31);
    const float* x_base = Xs + (lane >> 5) * SLAB + wk * 64 + (lane & 31);
    const bool n_live = wn * 128 < p.N, k_live = wk * 64 < p.K;

    if (mbeg < mend) {
        load_stage(mbeg);
        store_stage();
    }
    __syncthreads();
    for (long mb = mbeg; mb < mend; mb += TN_ROWS) {
        const bool more = mb + TN_ROWS < mend;
        if (more) load_stage(mb + TN_ROWS);
        if (n_live && k_live) {
#pragma unroll 4
            for (int mm = 0; mm < TN_ROWS; mm += 2) {
                float gf[4], xf[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) gf[i] = g_base[mm * SLAB + i * 32];
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[j] = x_base[mm * SLAB + j * 32];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (tid < SLAB) {
#pragma unroll 8
            for (int mm = 0; mm < TN_ROWS; ++mm) colsum += Gs[mm * SLAB + tid];
        }
        __syncthreads();
        if (more) {
            store_stage();
            __syncthreads();
        }
    }

    float* out = p.slab + (long)blockIdx.x * SLAB * SLAB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = wk * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int n = wn * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                out[n * SLAB + k] = acc[i][j][e];
            }
        }
    if (tid < SLAB) p.bias_slab[(long)blockIdx.x * SLAB + tid] = colsum;
}

// dW[n][k] (ld) = sum_s slab[s][n][k] ;  db[n] = sum_s bias_slab[s][n]
// One workgroup = 16 float4 elements x 16 slab groups: every thread sums S/16 slabs with independent
// loads, then the 16 partial sums of an element meet in LDS.  (A thread-per-element loop over all S
// slabs is a 256-deep dependent load chain on 64 workgroups: 113 us instead of ~10.)
// PERM: the slabs come from gemm_tn_h3 with both axes in LDS-row order (row r holds column 4 (r & 63) + (r >> 6)).
struct ReduceJob { const float* slab; const float* bias_slab; int S, N, K; float* dW; int ldw; float* db; };
struct ReduceBatch { ReduceJob job[8]; };

template <bool PERM>
__global__ __launch_bounds__(256) void slab_reduce_kernel(ReduceBatch rb) {
    const ReduceJob& rj = rb.job[blockIdx.y];
    const float* __restrict__ slab = rj.slab;
    const float* __restrict__ bias_slab = rj.bias_slab;
    const int S = rj.S, N = rj.N, K = rj.K, ldw = rj.ldw;
    float* __restrict__ dW = rj.dW;
    float* __restrict__ db = rj.db;
    __shared__ float4 part[16][17];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;              // float4 index inside the 256x256 tile
    const int rn = e >> 6, rk = (e & 63) * 4;        // position inside the slab
    const int n = PERM ? 4 * (rn & 63) + (rn >> 6) : rn;
    const int k = PERM ? 4 * (rk & 63) + (rk >> 6) : rk;          // column of the first element
    constexpr int KS = PERM ? 4 : 1;                              // column step between the four elements
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N && k < K) {
        for (int s = grp; s < S; s += 16) {
            float4 v = *reinterpret_cast<const float4*>(slab + (long)s * SLAB * SLAB + rn * SLAB + rk);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    part[grp][el] = acc;
    __syncthreads();
    if (grp == 0 && n < N && k < K) {
#pragma unroll
        for (int g = 1; g < 16; ++g) { float4 v = part[g][el]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        float* dst = dW + (long)n * ldw + k;
        dst[0] = acc.x;
        if (k + KS < K) dst[KS] = acc.y;
        if (k + 2 * KS < K) dst[2 * KS] = acc.z;
        if (k + 3 * KS < K) dst[3 * KS] = acc.w;
    }
    if (db && blockIdx.x < 16) {                      // bias: 256 columns, 16 per workgroup
        const int c = blockIdx.x * 16 + el;
        float b = 0.f;
        if (c < N) for (int s = grp; s < S; s += 16) b += bias_slab[(long)s * SLAB + c];
        __syncthreads();
        part[grp][el].x = b;
        __syncthreads();
        if (grp == 0 && c < N) {
            for (int g = 1; g < 16; ++g) b += part[g][el].x;
            db[c] = b;
        }
    }
}

constexpr int MAX_SLICES = 256;

// dW (N x K, leading dim ldw) = G^T X ; db = column sums of G (optional)
int gemm_tn(const float* G, long ldg, int N, const float* X, long ldx, int K, long M, float* dW, int ldw, float* db,
            void* workspace, hipStream_t s) {
    PAPR_REQUIRE(N <= SLAB && K <= SLAB, "gemm_tn: N=%d, K=%d exceed %d", N, K, SLAB);
    PAPR_REQUIRE(N % 4 == 0 && K % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0, "gemm_tn: sizes must be multiples of 4");
    if (M <= 0) return 0;
    long stages = (M + TN_ROWS - 1) / TN_ROWS;
    int S = (int)(stages < MAX_SLICES ? stages : MAX_SLICES);
    long rows_per_slice = ((stages + S - 1) / S) * TN_ROWS;
    S = (int)((M + rows_per_slice - 1) / rows_per_slice);
    TNArgs a;
    a.G = G; a.ldg = ldg; a.N = N; a.X = X; a.ldx = ldx; a.K = K; a.M = M; a.rows_per_slice = rows_per_slice;
    a.slab = static_cast<float*>(workspace);
    a.bias_slab = a.slab + (size_t)MAX_SLICES * SLAB * SLAB;
    const bool prof = papr_prof_on();
    if (prof) papr_prof_begin(4, M, N, K, s);
    gemm_tn_kernel<<<dim3(S), dim3(512), 2 * TN_ROWS * SLAB * sizeof(float), s>>>(a);
    if (prof) papr_prof_end(s);
    PAPR_CHECK_LAUNCH("gemm_tn");
    ReduceBatch rb = {};
    rb.job[0] = ReduceJob{a.slab, a.bias_slab, S, N, K, dW, ldw, db};
    slab_reduce_kernel<false><<<dim3(SLAB * SLAB / 4 / 16, 1), dim3(256), 0, s>>>(rb);
    PAPR_CHECK_LAUNCH("slab_reduce");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// gemm_tn_h3: the weight-gradient on the f16 matrix pipe with split operands (see gemm_nt_h3 for the
// arithmetic).  The reduction runs over the rows m, so an MFMA operand is 8 CONSECUTIVE ROWS of one
// column: the LDS image is the transpose of the global one.  A thread loads a 4 x 4 block (4 rows, one
// float4 each); read column-wise its registers are already the transposed 4-row runs, so each column
// goes out as one 8-byte LDS write per plane -- no shuffles.  Column c = 4 l + j of lane l is stored at
// LDS row rho(c) = l + 64 j: the 16 lanes of a ds_write_b64 group then hit consecutive LDS rows, which
// the 72-byte row pitch (36 halfs) spreads over all 32 banks, and the 32 lanes of a fragment read
// (consecutive LDS rows again) over all 64.  The output tile therefore comes out with both axes
// permuted by rho; slab_reduce_kernel<true> undoes that when it writes dW.
//
// Scales: a per-row factor cannot leave a sum over rows, so G and X get ONE power-of-two scale per
// workgroup slice (max over the slice's per-row maxima, which the forward pass / the previous
// data-gradient already produced).  Rows far below the slice maximum lose relative precision, but the
// error they contribute is 2^-25 of the LARGEST term of the sum, below the fp32 rounding of the sum itself.
struct TNH3Args {
    const float* G; long ldg; int N;
    const float* X; long ldx; int K;
    long M; long rows_per_slice;
    const float* gmax; const float* xmax;      // per-row max |.| of G and of X
    float* slab; float* bias_slab;
    int g_half, x_half;                        // (ONE instantiation) the operand is f16 rows, each row scaled by the power of two that the fused-run
                                               // kernel made of the maximum in gmax / xmax; ldg / ldx then count halfs
    int g_rs, x_rs;                            // which power: 0 = the parity arithmetic's per-layer scale (PAPR_MLP_H3_F16ROWS; h3_scale_from_max), 1 / 2 = the
                                               // one-product mode's scale per row and RUN, forward / data-gradient clamp (one_scale_from_max): gmax / xmax
                                               // are then the maxima of the run's top gradient rows / input rows for EVERY layer of the run
};
__device__ __forceinline__ float inv_scale_from_row_max(float m, int rs) {       // 1 / (the fused-run kernels' row scale)
    const unsigned bits = __float_as_uint(m);
    float inv;
    if (rs) { (void)one_scale_from_max(bits, rs == 2 ? ONE_EMIN_DGRAD : ONE_EMIN_FWD, inv); return inv; }
    const int ea = bits ? (int)((bits >> 23) & 0xff) : 127 + 13;
    return pow2_from_biased(127 - 13 + (ea - 127));
}
// Several weight-gradients in one launch: a workgroup streams its slice of job 0, then of job 1, ...  The partial
// tile of a job (256 KB per workgroup) drains to memory while the next job's first rows are already on their way,
// and the launch ramp and tail are paid once per batch instead of once per layer.
constexpr int TN_BATCH = 8;
struct TNH3Batch { TNH3Args job[TN_BATCH]; int n; int par; };      // par: job-parallel launch (workgroup b: slice b / n of job b % n), else every workgroup walks all jobs

constexpr int T3_HP = 36;                         // LDS row pitch in halfs
constexpr int T3_PLANE = SLAB * T3_HP;            // halfs per plane (256 LDS rows x 32 m)
constexpr size_t T3_LDS_BYTES = (size_t)2 * 4 * T3_PLANE * sizeof(_Float16);    // two buffers of [Gh Gl Xh Xl]

__device__ __forceinline__ half8 lds_read8(const _Float16* q) {
    half4 a = *reinterpret_cast<const half4*>(q), b = *reinterpret_cast<const half4*>(q + 4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// FULL: every 32-column tile holds real columns (N, K > 131): the hot loop has no tile tests
__device__ __forceinline__ float comp4(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// FMT 0: fp32 operands, three f16 products per fp32 product.  PAPR_GEMM_MODE=h1: FMT 1 = fp32 operands, one product (hi planes
// only); FMT 2 = both operands are f16 rows (g_half / x_half), one product, two stages of rows in flight; FMT 3 (round 6, mode PAPR_MLP_H3_F16ROWS:
// the first layer of a run of the parity arithmetic) = G f16 rows, X fp32 rows, one product.
template <bool FULL, int FMT>
__global__ __launch_bounds__(512, 2) void gemm_tn_h3_kernel(TNH3Batch batch) {
    constexpr bool ONE = FMT != 0, HALF = FMT == 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);
    __shared__ float red[2][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: row tests and tile skips become SALU
    const int wn = wave >> 2, wk = wave & 3;     // wave's corner in LDS-row space: n-tiles 4 wn .. +3, k-tiles 2 wk, 2 wk + 1
  // job-parallel (batch.par): the launch's workgroups are dealt to the jobs -- a workgroup streams ONE
  // slice of ONE job, n times as long as a slice of the job-serial form, and leaves one partial tile instead of n: an n-th of the
  // partial-tile traffic and of the reduction behind it
  const int jb0 = batch.par ? (int)(blockIdx.x % (unsigned)batch.n) : 0, jb1 = batch.par ? jb0 + 1 : batch.n;
  const long slice = batch.par ? (long)(blockIdx.x / (unsigned)batch.n) : (long)blockIdx.x;
  for (int jb = jb0; jb < jb1; ++jb) {
    const TNH3Args& p = batch.job[jb];
    const long mbeg = slice * p.rows_per_slice;
    long mend = mbeg + p.rows_per_slice;
    if (mend > p.M) mend = p.M;
    if (mbeg >= mend) continue;                  // (workgroup-uniform: a job with fewer slices than the launch has workgroups)
    constexpr bool gh = FMT >= 2, xh = FMT == 2;

    // slice scales
    // (four rows per load: a slice is 2,000-14,000 rows, and one row per thread and trip was 4-28 dependent trips to memory in front of the first stage)
    float gm = 0.f, xm = 0.f;
    {
        const long n4 = (mend - mbeg) >> 2;          // (mbeg is a multiple of the 32-row stage: the tables' float4 are aligned)
        const float4* g4 = reinterpret_cast<const float4*>(p.gmax + mbeg);
        const float4* x4 = reinterpret_cast<const float4*>(p.xmax + mbeg);
        for (long q = tid; q < n4; q += 512) {
            const float4 a = g4[q], b = x4[q];
            gm = fmaxf(fmaxf(gm, fmaxf(a.x, a.y)), fmaxf(a.z, a.w)); xm = fmaxf(fmaxf(xm, fmaxf(b.x, b.y)), fmaxf(b.z, b.w));
        }
        for (long m = mbeg + 4 * n4 + tid; m < mend; m += 512) { gm = fmaxf(gm, p.gmax[m]); xm = fmaxf(xm, p.xmax[m]); }
    }
    gm = wave_max(gm); xm = wave_max(xm);
    if (lane == 0) { red[0][wave] = gm; red[1][wave] = xm; }
    __syncthreads();
    gm = red[0][0]; xm = red[1][0];
#pragma unroll
    for (int w = 1; w < 8; ++w) { gm = fmaxf(gm, red[0][w]); xm = fmaxf(xm, red[1][w]); }
    const int eg = gm > 0.f ? (int)((__float_as_uint(gm) >> 23) & 0xff) : 127 + 13;
    const int ex = xm > 0.f ? (int)((__float_as_uint(xm) >> 23) & 0xff) : 127 + 13;
    float g_scale = 4 * lane < p.N ? pow2_from_biased(127 + 13 - (eg - 127)) : 0.f;     // columns beyond the matrix become zeros
    float x_scale = 4 * lane < p.K ? pow2_from_biased(127 + 13 - (ex - 127)) : 0.f;
    float g_inv = pow2_from_biased(127 - 13 + (eg - 127)), x_inv = pow2_from_biased(127 - 13 + (ex - 127));
    // f16 rows that carry ONE scale per row and RUN (one-product mode, g_rs / x_rs): gmax / xmax are the maxima of the run's TOP gradient rows / INPUT rows,
    // not of this layer's rows -- which may be up to 2^12 larger in the same scaled domain (2^9 when this was found) (that is the run scale's headroom).  A slice scale made from those
    // maxima overflowed f16 as soon as a trained network's inner gradients outgrew its top gradients eightfold: inf in dW on every step, the GradScaler
    // halving its scale down to 2^-50, training at a standstill from step ~5,000 of a chair.yml run (found by the round's last 21,500-step run).  The rows
    // themselves are bounded by f16's own range, so the slice factor is taken from the rows' SCALES: row factor = (1 / scale_row) / (2 max over the slice
    // of 1 / scale) <= 1/2 -- a power of two times an f16 number that cannot overflow; rows far below the slice's largest lose low bits as before.
    if (gh && p.g_rs) { const float igm = inv_scale_from_row_max(gm, p.g_rs); g_scale = 4 * lane < p.N ? 0.5f / igm : 0.f; g_inv = 2.f * igm; }
    if (xh && p.x_rs) { const float ixm = inv_scale_from_row_max(xm, p.x_rs); x_scale = 4 * lane < p.K ? 0.5f / ixm : 0.f; x_inv = 2.f * ixm; }

    // which 32-row tiles of the LDS image hold real columns: tile t covers columns 4 (32 (t & 1) + i) + (t >> 1)
    bool live_n[4], live_k[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) live_n[i] = FULL || 128 * ((wn * 4 + i) & 1) + ((wn * 4 + i) >> 1) < p.N;
#pragma unroll
    for (int j = 0; j < 2; ++j) live_k[j] = FULL || 128 * ((wk * 2 + j) & 1) + ((wk * 2 + j) >> 1) < p.K;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};

    // stage = 32 rows; wave w carries rows 4 w .. 4 w + 3 of it, lane l the columns 4 l .. 4 l + 3 (one full row per load)
    const int cg = 4 * lane < p.N ? 4 * lane : 0, cx = 4 * lane < p.K ? 4 * lane : 0;
    float4 rg[4], rx[4];
    // (f16 rows: the four halfs of a lane ride in .x and .y of the float4)
    // Row addresses: the slice's first row and the lane's column once per job (64-bit), a row inside the slice as a 32-bit index clamped with one
    // s_min and multiplied by a 32-bit byte stride -- the scalar unit spent ~12 instructions per row load on 64-bit products and compares before
    // (sixteen row loads per stage and wave)
    const int nrows = (int)(mend - mbeg);
    const int gstride = (int)p.ldg * (gh ? 2 : 4), xstride = (int)p.ldx * (xh ? 2 : 4);        // bytes per row
    const char* const gbase = reinterpret_cast<const char*>(p.G) + (mbeg * p.ldg + cg) * (gh ? 2 : 4);
    const char* const xbase = reinterpret_cast<const char*>(p.X) + (mbeg * p.ldx + cx) * (xh ? 2 : 4);
    auto row_in_slice = [&](long st, int r) {
        if (!HALF) { const long m = mbeg + st * TN_ROWS + 4 * wave + r; return (int)((m < mend ? m : mend - 1) - mbeg); }
        const int i = (int)st * TN_ROWS + 4 * wave + r;
        return i < nrows ? i : nrows - 1;
    };
    // (the f16-row form only: it was bound by its scalar instructions, 601 -> 572 us on a five-layer run; the fp32-row form, bound elsewhere,
    //  measured 1,015-1,019 us this way against 998-1,014 and keeps its 64-bit row arithmetic)
    auto load_g = [&](int rel) {
        if (gh) { const float2 t = *reinterpret_cast<const float2*>(gbase + (long)rel * gstride); return make_float4(t.x, t.y, 0.f, 0.f); }
        return *reinterpret_cast<const float4*>(p.G + (mbeg + rel) * p.ldg + cg);
    };
    auto load_x = [&](int rel) {
        if (xh) { const float2 t = *reinterpret_cast<const float2*>(xbase + (long)rel * xstride); return make_float4(t.x, t.y, 0.f, 0.f); }
        return *reinterpret_cast<const float4*>(p.X + (mbeg + rel) * p.ldx + cx);
    };
    auto load_stage = [&](long st, float4 (&qg)[4], float4 (&qx)[4]) {      // unconditional, from clamped rows
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rel = row_in_slice(st, r);
            qg[r] = load_g(rel);
            qx[r] = load_x(rel);
        }
    };
    // element j of a lane's four columns of one row, as fp32 (an f16 row: still times the row's scale)
    auto elem = [&](const float4& q, int j, bool half) {
        if (!half) return comp4(q, j);
        const unsigned w = __float_as_uint(j < 2 ? q.x : q.y);
        const unsigned short hb = (unsigned short)((j & 1) ? (w >> 16) : (w & 0xffffu));
        return (float)*reinterpret_cast<const _Float16*>(&hb);
    };
    // per-row factors of stage st: scale of the slice (times 1 / the row's own scale for f16 rows), 0 for rows past the slice;
    // fg: what turns an element of G into its true value (bias gradient)
    // (f16 rows: the wave's four row maxima of a stage are one 16-byte scalar load, asked for a stage ahead -- at the top of a stage
    //  it would wait for a round trip to L2 before the first fragment read.  The index is clamped to the slice; up to three floats
    //  behind the array may be read for rows past M, whose factors are 0 anyway: both arrays have memory behind them.)
    // (The address is the same in every lane, and the compiler knows: it loaded the sixteen bytes with a vector load, moved them to scalar registers
    //  with eight v_readfirstlane -- each behind an s_waitcnt vmcnt that, loads returning in order, also waited for every row load of the stages
    //  ahead -- and did the factors' bit arithmetic on the scalar unit, ~100 scalar instructions per stage: 130 us of a 630-us launch.  `lane0`, a
    //  zero the compiler cannot see through, keeps value and arithmetic in vector registers, where the wait sits at the first use a stage later.)
    int lane0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane0));
    auto row_maxima = [&](long st, float4& mg, float4& mx) {
        if (!gh) return;
        long base = mbeg + st * TN_ROWS + 4 * wave;
        const long top = (mend - 1) & ~3L;
        base = base < top ? base : top;
        mg = *reinterpret_cast<const float4*>(p.gmax + base + lane0);
        mx = *reinterpret_cast<const float4*>(p.xmax + base + lane0);
    };
    auto row_factors = [&](long st, const float4& mg, const float4& mx, float (&sg)[4], float (&sx)[4], float (&fg)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = mbeg + st * TN_ROWS + 4 * wave + r;
            const bool ok = m < mend;
            const float ig = gh ? inv_scale_from_row_max(comp4(mg, r), p.g_rs) : 1.f, ix = xh ? inv_scale_from_row_max(comp4(mx, r), p.x_rs) : 1.f;
            sg[r] = ok ? g_scale * ig : 0.f;
            sx[r] = ok ? x_scale * ix : 0.f;
            fg[r] = ok ? ig : 0.f;
        }
    };
    auto put_block = [&](const float4 (&q)[4], const float (&sc)[4], _Float16* hi_plane, _Float16* lo_plane, bool half) {
        float y[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) y[r][j] = elem(q[r], j, half) * sc[r];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            half4 hi = half4{(_Float16)y[0][j], (_Float16)y[1][j], (_Float16)y[2][j], (_Float16)y[3][j]};
            half4 lo = half4{(_Float16)(y[0][j] - (float)hi[0]), (_Float16)(y[1][j] - (float)hi[1]),
                             (_Float16)(y[2][j] - (float)hi[2]), (_Float16)(y[3][j] - (float)hi[3])};
            const int off = (lane + 64 * j) * T3_HP + 4 * wave;
            *reinterpret_cast<half4*>(hi_plane + off) = hi;
            if (!ONE) *reinterpret_cast<half4*>(lo_plane + off) = lo;
        }
    };
    auto store_stage = [&](long st, const float4 (&qg)[4], const float4 (&qx)[4]) {
        _Float16* base = lds + (st & 1) * (4 * T3_PLANE);
        float sg[4], sx[4], fg[4];
        float4 mg0 = make_float4(0.f, 0.f, 0.f, 0.f), mx0 = mg0;
        row_maxima(st, mg0, mx0);
        row_factors(st, mg0, mx0, sg, sx, fg);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) colsum[j] = __builtin_fmaf(elem(qg[r], j, gh), fg[r], colsum[j]);
        put_block(qg, sg, base, base + T3_PLANE, gh);
        put_block(qx, sx, base + 2 * T3_PLANE, base + 3 * T3_PLANE, xh);
    };
    const int frag = (lane & 31) * T3_HP + 8 * (lane >> 5);
    // One register set of 32 rows in flight (a second set does not fit beside the 128 accumulator registers).
    const long nst = (mend - mbeg + TN_ROWS - 1) / TN_ROWS;
    load_stage(0, rg, rx);
    store_stage(0, rg, rx);
    // deep (both operands f16 rows): the rows are half as many bytes, so TWO stages ride in the same registers -- stage s in
    // (.x, .y) of the float4 when s is even, in (.z, .w) when it is odd -- and a load has two stage times to arrive instead of one
    constexpr bool deep = HALF;
    auto load_half_stage = [&](long st, auto oddc, bool do_g, bool do_x) {          // oddc: the parity of st, as a type
        constexpr bool odd_slot = decltype(oddc)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rel = row_in_slice(st, r);
            if (do_g) {
                const float2 t = *reinterpret_cast<const float2*>(gbase + (long)rel * gstride);
                if (odd_slot) { rg[r].z = t.x; rg[r].w = t.y; } else { rg[r].x = t.x; rg[r].y = t.y; }
            }
            if (do_x) {
                const float2 t = *reinterpret_cast<const float2*>(xbase + (long)rel * xstride);
                if (odd_slot) { rx[r].z = t.x; rx[r].w = t.y; } else { rx[r].x = t.x; rx[r].y = t.y; }
            }
        }
    };
    if (deep) { load_half_stage(1, std::true_type(), true, true); load_half_stage(2, std::false_type(), true, true); }
    else load_stage(1, rg, rx);
    float4 mgn = make_float4(0.f, 0.f, 0.f, 0.f), mxn = mgn;      // row maxima of the stage the next iteration splits
    row_maxima(1, mgn, mxn);
    lds_barrier();
    {
        // The split of stage st+1 rides between the matrix instructions of stage st.  A wave issues in order and a
        // matrix instruction waits for its pipe (32 cycles each), so vector work placed behind a run of matrix
        // instructions only starts when the last of them has been issued: the two kinds of work overlap only when
        // they alternate instruction by instruction.  A stage is cut into eight pieces -- (operand, column j of the
        // thread's 4 x 4 block) -- one per group of six matrix instructions, each piece in two steps behind one
        // instruction each (sched_barrier pins the order).  The two accumulators of a group
        // alternate, so that no instruction waits for the result of the one just before it.  An operand's four rows are
        // requested again (stage st+2) when its fourth column has left the registers.
        const int tile_x0 = (wk * 2) * 32 * T3_HP + frag, tile_x1 = (wk * 2 + 1) * 32 * T3_HP + frag;
        auto stage_body = [&](long st, auto oddc) {      // oddc: parity of st + 1 (only looked at when two stages are in flight)
            const _Float16* Gh = lds + (st & 1) * (4 * T3_PLANE);
            const _Float16* Gl = Gh + T3_PLANE;
            const _Float16* Xh = Gh + 2 * T3_PLANE;
            const _Float16* Xl = Gh + 3 * T3_PLANE;
            _Float16* nb = lds + ((st + 1) & 1) * (4 * T3_PLANE);        // stage st+1 goes here (rows past the slice: zeros)
            float sg[4], sx[4], okf[4];
            row_factors(st + 1, mgn, mxn, sg, sx, okf);
            row_maxima(st + 2, mgn, mxn);
            constexpr bool odd = deep && decltype(oddc)::value;      // where the rows being split (stage st + 1) sit in their registers
            auto hsrc = [&](const float4& q, int j) { return odd ? (j < 2 ? q.z : q.w) : (j < 2 ? q.x : q.y); };
            half8 xh0 = lds_read8(Xh + tile_x0), xh1 = lds_read8(Xh + tile_x1), xl0 = xh0, xl1 = xh1;
            half8 gfh = lds_read8(Gh + (wn * 4) * 32 * T3_HP + frag), gfl = gfh;
            if (!ONE) { xl0 = lds_read8(Xl + tile_x0); xl1 = lds_read8(Xl + tile_x1); gfl = lds_read8(Gl + (wn * 4) * 32 * T3_HP + frag); }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int i = g & 3;
                // piece of this group: column j of the G block (g < 4) or of the X block.  (Two pieces per group in the first
                // half of the stage, which leaves a reloaded operand six groups instead of four to arrive, measured slower:
                // 219 against 212 us.)
                const bool is_g = g < 4;
                const int j = g & 3;
                const float4 (&q)[4] = is_g ? rg : rx;
                const float (&sc)[4] = is_g ? sg : sx;
                const bool qhalf = is_g ? gh : xh;
                _Float16* hi_plane = nb + (is_g ? 0 : 2 * T3_PLANE);
                const bool on = FULL || live_n[i], k0 = FULL || live_k[0], k1 = FULL || live_k[1];
                half8 ghn = gfh, gln = gfl;
                unsigned h01, h23, l01, l23;             // packed f16 pairs (rows 0,1 and 2,3 of a column)
                // hi = f16(q * scale), two rows per register; lo = f16(q * scale - hi) in one fused instruction each (the
                // scale is a power of two, so the product is exact and this rounds like the two-step form)
                auto split_hi = [&](const int j) {
                    if (ONE && qhalf) {                  // f16 source: half j & 1 of register j >> 1 of the row, times its factor
                        if (j & 1) {
                            asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(h01) : "v"(hsrc(q[0], j)), "v"(sc[0]));
                            asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(h23) : "v"(hsrc(q[2], j)), "v"(sc[2]));
                            asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h01) : "v"(hsrc(q[1], j)), "v"(sc[1]));
                            asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h23) : "v"(hsrc(q[3], j)), "v"(sc[3]));
                        } else {
                            asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(h01) : "v"(hsrc(q[0], j)), "v"(sc[0]));
                            asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(h23) : "v"(hsrc(q[2], j)), "v"(sc[2]));
                            asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(h01) : "v"(hsrc(q[1], j)), "v"(sc[1]));
                            asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(h23) : "v"(hsrc(q[3], j)), "v"(sc[3]));
                        }
                    } else {
                        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(comp4(q[0], j)), "v"(sc[0]));
                        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(comp4(q[2], j)), "v"(sc[2]));
                        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(comp4(q[1], j)), "v"(sc[1]));
                        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(comp4(q[3], j)), "v"(sc[3]));
                    }
                    *reinterpret_cast<uint2*>(hi_plane + (lane + 64 * j) * T3_HP + 4 * wave) = make_uint2(h01, h23);
                };
                auto add_colsum = [&](const int j) {     // bias gradient: the true values of G's column j (okf: 1 / row scale, or 1, or 0)
                    if (ONE && qhalf) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (j & 1) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(colsum[j]) : "v"(hsrc(q[r], j)), "v"(okf[r]));
                            else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(colsum[j]) : "v"(hsrc(q[r], j)), "v"(okf[r]));
                        }
                    } else {
                        colsum[j] = __builtin_fmaf(comp4(q[3], j), okf[3], __builtin_fmaf(comp4(q[2], j), okf[2],
                                    __builtin_fmaf(comp4(q[1], j), okf[1], __builtin_fmaf(comp4(q[0], j), okf[0], colsum[j]))));
                    }
                };
                auto split_lo = [&](const int j) {
                    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l01) : "v"(comp4(q[0], j)), "v"(sc[0]), "v"(h01));
                    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l23) : "v"(comp4(q[2], j)), "v"(sc[2]), "v"(h23));
                    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(comp4(q[1], j)), "v"(sc[1]), "v"(h01));
                    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(comp4(q[3], j)), "v"(sc[3]), "v"(h23));
                    *reinterpret_cast<uint2*>(hi_plane + T3_PLANE + (lane + 64 * j) * T3_HP + 4 * wave) = make_uint2(l01, l23);
                };
                if (!ONE && on && k0) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfl, xh0, acc[i][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (g < 7) {                             // next group's G fragments
                    const int o = (wn * 4 + ((g + 1) & 3)) * 32 * T3_HP + frag + 16 * ((g + 1) >> 2);
                    ghn = lds_read8(Gh + o);
                    if (!ONE) gln = lds_read8(Gl + o);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!ONE && on && k1) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfl, xh1, acc[i][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#ifndef TN_ABL_NO_MFMA
                if (ONE && on && k0) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfh, xh0, acc[i][0], 0, 0, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifndef TN_ABL_NO_SPLIT
                split_hi(j);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (!ONE && on && k0) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfh, xl0, acc[i][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!ONE && on && k1) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfh, xl1, acc[i][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#ifndef TN_ABL_NO_MFMA
                if (ONE && on && k1) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfh, xh1, acc[i][1], 0, 0, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (!ONE) split_lo(j);
#ifndef TN_ABL_NO_COLSUM
                if (is_g) add_colsum(j);
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (!ONE && on && k0) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfh, xh0, acc[i][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!ONE && on && k1) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gfh, xh1, acc[i][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#ifndef TN_ABL_NO_LOAD
                if (g == 3) {                            // the G block has left the registers: its rows of stage st+2 (deep: st+3)
                    if (deep) load_half_stage(st + 3, oddc, true, false);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            rg[r] = load_g(row_in_slice(st + 2, r));
                        }
                    }
                }
#endif
                if (g == 3) {                            // second k step of the stage: its X fragments
                    xh0 = lds_read8(Xh + tile_x0 + 16); xh1 = lds_read8(Xh + tile_x1 + 16);
                    if (!ONE) { xl0 = lds_read8(Xl + tile_x0 + 16); xl1 = lds_read8(Xl + tile_x1 + 16); }
                }
                if (g == 7) {                            // the X rows of stage st+2
#ifndef TN_ABL_NO_LOAD
                    if (deep) load_half_stage(st + 3, oddc, false, true);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            rx[r] = load_x(row_in_slice(st + 2, r));
                        }
                    }
#endif
                }
                gfh = ghn; gfl = gln;
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier();
        };
        if (deep) {
            for (long st = 0; st < nst; st += 2) {
                stage_body(st, std::true_type());
                if (st + 1 < nst) stage_body(st + 1, std::false_type());
            }
        } else {
            for (long st = 0; st < nst; ++st) stage_body(st, std::false_type());
        }
    }

    // slab in LDS-row order on both axes (un-permuted by the reduction), un-scaled
    float* out = p.slab + slice * SLAB * SLAB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!(live_n[i] && live_k[j])) continue;
            const int kk = (wk * 2 + j) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int nn = (wn * 4 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                out[nn * SLAB + kk] = acc[i][j][e] * g_inv * x_inv;
            }
        }
    // column sums of G: the 8 waves hold partial sums of the same columns
    float* cs = smem;                               // the stage buffers are idle now (last barrier passed)
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[wave * SLAB + 4 * lane + j] = colsum[j];
    __syncthreads();
    if (tid < SLAB) {
        float b = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) b += cs[w * SLAB + tid];
        p.bias_slab[slice * SLAB + tid] = b;
    }
    // (the next job's first barrier comes before its first LDS write: the sums above are read by then)
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn_tr (round 6): the weight gradient of a full 256 x 256 layer from f16 x f16 rows WITHOUT a pass through registers.
//   * rows arrive by LDS-DMA (global_load_lds_dwordx4), four 32-row stages in flight (128 KB of the CU's LDS; the register-staged form keeps two,
//     in registers) -- the kernel is HBM-bound and bytes in flight are what buys bandwidth;
//   * the matrix operand is 8 CONSECUTIVE ROWS of one column: the transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a [4 rows][16 columns]
//     block, lane c receives column c's four rows) takes it straight out of a ROW-major image.  The image is cut into [32 rows][16 columns] subtiles of
//     1 KB (a DMA instruction fills one: lane i fetches row i / 2, 16-byte piece i & 1 -- the gather is on the SOURCE side, the destination of an LDS-DMA
//     instruction is linear), 1,152 bytes apart so that the two subtiles a 32-lane half reads sit on disjoint banks (MI355X guide: conflict-free layout);
//   * the rows' power-of-two scales differ from row to row and the sum runs over rows: the G fragment is multiplied (packed f16, exact) by
//     c[m] = (1 / scale_g[m]) (1 / scale_x[m]) / P, P = the largest such product of the slice -- rows far below it underflow, contributing less than the
//     fp32 rounding of the sum (the same argument as gemm_tn_h3's slice scale); the tile leaves times P.  c and 1 / scale_g (for the bias gradient, taken
//     from the UN-scaled fragments in fp32) come from a 32-entry table per stage that wave 0 makes a stage ahead from the row maxima, which arrive by DMA too.
// Same slices, same stage order, same 16-row matrix instructions as gemm_tn_h3_kernel<., 2>: partial tiles in natural order (slab_reduce_kernel<false>).
constexpr int TR_D = 4;                          // stage buffers: one being multiplied, three on their way
constexpr int TR_PAIR = 1088;                    // bytes from one row pair to the next (2 x 512 + 64: four consecutive pairs sit on the four quarters of the banks)
constexpr int TR_OP = 16 * TR_PAIR;              // one operand of one stage: pair q = rows q and q + 16 of the stage, 512 bytes each
constexpr int TR_STAGE = 2 * TR_OP;              // G | X
constexpr int TR_OFF_RAW = TR_D * TR_STAGE;      // per stage: max |g| [32] | max |x| [32] floats
constexpr int TR_OFF_C = TR_OFF_RAW + TR_D * 256;        // per stage: c[32] halfs
constexpr int TR_OFF_IG = TR_OFF_C + TR_D * 64;          // per stage: 1 / scale_g [32] floats
constexpr size_t TR_LDS_BYTES = TR_OFF_IG + TR_D * 128;
typedef __fp16 tr_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"      // (an LDS address is a 32-bit offset: the host pass of hipcc sees a 64-bit pointer type)

__device__ __forceinline__ void tr_dma16(const void* src, unsigned lds_dst) {       // 64 lanes x 16 bytes -> LDS [lds_dst, +1 KB), lane order
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void tr_dma4(const void* src, unsigned lds_dst) {        // 64 lanes x 4 bytes
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
template <int FOUR_ROWS = 4 * TR_PAIR>
__device__ __forceinline__ half8 tr_read8(unsigned a) {     // 8 consecutive rows of this lane's column: two transposing reads of four rows each (four pairs apart)
    const tr_h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tr_h4*)a);
    const tr_h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tr_h4*)(a + FOUR_ROWS));
    return half8{(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3], (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
}

struct TrFrags { half8 x[2]; half8 g[4]; half8 cv; float4 ig0, ig1; };

// The factor that carries both rows' scales, in EXPONENTS: (1 / scale_g)(1 / scale_x) is a power of two whose exponent is the sum of two exponents each inside
// fp32's normal range -- their product is not: the gradient rows of a query MLP late in a run with few points are ~1e-35, the product underflowed, 1 / P
// became inf and the weight gradients NaN (found by the round's last long lego run: NaN parameters at step 13,904, a memory fault once the neighbour
// search met NaN coordinates).  A row that is all zeros on either side contributes nothing and must not set the slice's exponent (the parity arithmetic
// gives a zero row the scale 1, far above the 2^-30 ... 2^-10 of real gradient rows).
constexpr int TR_EXP_NONE = -100000;
__device__ __forceinline__ int tr_pow2_exp(float p) { return (int)((__float_as_uint(p) >> 23) & 0xff) - 127; }       // p: a normal power of two
__device__ __forceinline__ int tr_pair_exp(float gmx, float xmx, int g_rs, int x_rs) {
    if (gmx == 0.f || xmx == 0.f) return TR_EXP_NONE;
    return tr_pow2_exp(inv_scale_from_row_max(gmx, g_rs)) + tr_pow2_exp(inv_scale_from_row_max(xmx, x_rs));
}
__device__ __forceinline__ float tr_factor(int e, int eP) {          // 2^(e - eP) <= 1 (what f16 cannot hold becomes 0 at the conversion)
    if (e <= TR_EXP_NONE / 2) return 0.f;
    const int d = e - eP;
    return d < -126 ? 0.f : pow2_from_biased(127 + d);
}

__global__ __launch_bounds__(512, 2) void gemm_tn_tr_kernel(TNH3Batch batch) {
    extern __shared__ __attribute__((aligned(16))) char tr_smem[];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;
    const unsigned lds0 = (unsigned)(size_t)tr_smem;
    const int jb = batch.par ? (int)(blockIdx.x % (unsigned)batch.n) : 0;
    const long slice = batch.par ? (long)(blockIdx.x / (unsigned)batch.n) : (long)blockIdx.x;
    const TNH3Args& p = batch.job[jb];
    const long mbeg = slice * p.rows_per_slice;
    long mend = mbeg + p.rows_per_slice;
    if (mend > p.M) mend = p.M;
    if (mbeg >= mend) return;
    const int nrows = (int)(mend - mbeg);
    const int g_rs = p.g_rs, x_rs = p.x_rs;

    // eP: the largest exponent of (1 / scale_g)(1 / scale_x) over the slice's rows (tr_pair_exp: kept as exponents; rows that are all zeros on either side do not
    // count -- with such a row setting it every real row's factor underflowed: the two-rank test's 16 x 16 images have them, 3 % of a gradient's norm was lost)
    auto pair_exp = [&](float gmx, float xmx) { return (float)tr_pair_exp(gmx, xmx, g_rs, x_rs); };      // (an exponent; exact as a float)
    float pm = (float)TR_EXP_NONE;
    {
        const int n4 = nrows >> 2;                   // (four rows per load; mbeg is a multiple of the stage: aligned)
        const float4* g4 = reinterpret_cast<const float4*>(p.gmax + mbeg);
        const float4* x4 = reinterpret_cast<const float4*>(p.xmax + mbeg);
        for (int q = tid; q < n4; q += 512) {
            const float4 a = g4[q], b = x4[q];
            pm = fmaxf(fmaxf(pm, fmaxf(pair_exp(a.x, b.x), pair_exp(a.y, b.y))), fmaxf(pair_exp(a.z, b.z), pair_exp(a.w, b.w)));
        }
        for (int m = 4 * n4 + tid; m < nrows; m += 512) pm = fmaxf(pm, pair_exp(p.gmax[mbeg + m], p.xmax[mbeg + m]));
    }
    pm = wave_max(pm);
    if (lane == 0) red[wave] = pm;
    __syncthreads();
    float Pf = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) Pf = fmaxf(Pf, red[w]);
    const int eP = Pf <= (float)(TR_EXP_NONE / 2) ? 0 : (int)Pf;       // the slice's largest exponent (a slice of zero rows: 0)

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float colsum = 0.f;                          // bias gradient: this wave sums G's tile wn * 4 + wk (the four waves of a wn read the same G fragments)

    const _Float16* const Gh = reinterpret_cast<const _Float16*>(p.G);
    const _Float16* const Xh = reinterpret_cast<const _Float16*>(p.X);
    const long nst = (nrows + TN_ROWS - 1) / TN_ROWS;
    // the requests of stage st (rows past the slice: the slice's last row again -- their c is 0).  One instruction = one row pair = two whole rows of
    // 512 bytes: lanes 0 .. 31 the 16-byte pieces of row q, lanes 32 .. 63 those of row q + 16 (whole cache lines: the first version of this kernel
    // gathered 32-byte pieces of 32 rows per instruction, four requests per line, and streamed 5.0 TB/s with the matrix work taken out)
    auto issue = [&](long st) {
        const unsigned buf = lds0 + (unsigned)(st % TR_D) * TR_STAGE;
        if (wave == 0) {
            int m = (int)st * TN_ROWS + (lane & 31);
            m = m < nrows ? m : nrows - 1;
            const float* src = (lane >> 5 ? p.xmax : p.gmax) + mbeg + m;
            tr_dma4(src, lds0 + TR_OFF_RAW + (unsigned)(st % TR_D) * 256);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = 4 * wave + q, op = idx >> 4, pair = idx & 15;
            int m = (int)st * TN_ROWS + pair + 16 * (lane >> 5);
            m = m < nrows ? m : nrows - 1;
            const _Float16* src = (op ? Xh + (mbeg + m) * p.ldx : Gh + (mbeg + m) * p.ldg) + 8 * (lane & 31);
            tr_dma16(src, buf + (unsigned)(op * TR_OP + pair * TR_PAIR));
        }
    };
    // the factor table of stage st (wave 0, lanes 0 .. 31), from the maxima in LDS
    auto make_table = [&](long st) {
        if (wave != 0 || lane >= 32) return;
        const int b = (int)(st % TR_D);
        const float* raw = reinterpret_cast<const float*>(tr_smem + TR_OFF_RAW + b * 256);
        const bool ok = (int)st * TN_ROWS + lane < nrows;
        const float ig = inv_scale_from_row_max(raw[lane], g_rs), ix = inv_scale_from_row_max(raw[32 + lane], x_rs);
        reinterpret_cast<_Float16*>(tr_smem + TR_OFF_C + b * 64)[lane] = ok ? (_Float16)tr_factor(tr_pair_exp(raw[lane], raw[32 + lane], g_rs, x_rs), eP) : (_Float16)0.f;
        reinterpret_cast<float*>(tr_smem + TR_OFF_IG + b * 128)[lane] = ok ? ig : 0.f;
    };
    // this lane's corner of a fragment: the 16-lane group (lane >> 4) & 1 takes the tile's columns 16 .. 31, the k-group lane >> 5 the rows 8 .. 15 of a
    // 16-row k-step; a lane of a group asks for row (lane & 15) >> 2 of four, 8-byte piece lane & 3 of the group's 32 bytes.  Row r of the stage: pair r & 15,
    // upper half for r >= 16 -- the second k-step of the stage is +512 bytes, tile t +64 t bytes, rows +4 are +4 pairs
    const unsigned frag = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * TR_PAIR + 32 * ((lane >> 4) & 1) + 8 * (lane & 3));
    const unsigned xfrag = frag + TR_OP + (unsigned)(wk * 2) * 64, gfrag = frag + (unsigned)(wn * 4) * 64;
    auto read_frags = [&](TrFrags& f, long st, int ks) {
        const unsigned sb = lds0 + (unsigned)(st % TR_D) * TR_STAGE + (unsigned)ks * 512;
        f.x[0] = tr_read8(sb + xfrag); f.x[1] = tr_read8(sb + xfrag + 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) f.g[i] = tr_read8(sb + gfrag + 64 * i);
        f.cv = *reinterpret_cast<const half8*>(tr_smem + TR_OFF_C + (st % TR_D) * 64 + (ks * 16 + 8 * (lane >> 5)) * 2);
        const float* ig = reinterpret_cast<const float*>(tr_smem + TR_OFF_IG + (st % TR_D) * 128) + ks * 16 + 8 * (lane >> 5);
        f.ig0 = *reinterpret_cast<const float4*>(ig); f.ig1 = *reinterpret_cast<const float4*>(ig + 4);
    };
    auto multiply = [&](const TrFrags& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const half8 gr = f.g[i];
            if (wk == i)                             // bias gradient: the true values of G, this lane's column, its eight rows of the k-step
                colsum = __builtin_fmaf((float)gr[7], f.ig1.w, __builtin_fmaf((float)gr[6], f.ig1.z, __builtin_fmaf((float)gr[5], f.ig1.y, __builtin_fmaf((float)gr[4], f.ig1.x,
                         __builtin_fmaf((float)gr[3], f.ig0.w, __builtin_fmaf((float)gr[2], f.ig0.z, __builtin_fmaf((float)gr[1], f.ig0.y, __builtin_fmaf((float)gr[0], f.ig0.x, colsum))))))));
            const half8 ga = gr * f.cv;              // (packed f16 multiplies by powers of two)
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga, f.x[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga, f.x[1], acc[i][1], 0, 0, 0);
        }
    };
    // Schedule: the fragments of a k-step are read while the matrix instructions of the k-step before run (two register sets), and the stage's ONE barrier
    // stands between its two k-steps: in front of it a wave has read everything of stage st (its buffer goes to the requests of stage st + 4 right
    // behind the barrier), behind it stage st + 1 is everybody's to read -- requested three stages earlier.
    //   wave 0's requests, in order: [maxima, 4 x rows] per stage.  In front of the barrier of stage st: the rows of st + 1 (and, wave 0, the maxima
    //   of st + 2: their table is made behind the barrier and first read behind the NEXT one) have landed when 8 (9) requests are outstanding.
    issue(0); issue(1); issue(2);
    if (wave == 0) {
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");          // maxima of stages 0 and 1 (behind them: 4 + 5 requests)
        make_table(0); make_table(1);
    } else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // rows of stage 0
    lds_barrier();
    issue(3);
    TrFrags fa, fb;
    read_frags(fa, 0, 0);
    for (long st = 0; st < nst; ++st) {
        read_frags(fb, st, 1);
#ifndef TR_ABL_NO_COMPUTE
        multiply(fa);
#endif
        if (wave == 0) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#ifndef TR_ABL_NO_BARRIER
        lds_barrier();
#endif
#ifndef TR_ABL_NO_DMA
        issue(st + TR_D);
#endif
        make_table(st + 2);
        read_frags(fa, st + 1, 0);
#ifndef TR_ABL_NO_COMPUTE
        multiply(fb);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the requests past the slice's end)
    const float P1 = pow2_from_biased(127 + (eP >> 1)), P2 = pow2_from_biased(127 + eP - (eP >> 1));       // 2^eP in two factors, each a normal number
    float* out = p.slab + slice * SLAB * SLAB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = (wk * 2 + j) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = (wn * 4 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                out[n * SLAB + k] = acc[i][j][e] * P1 * P2;
            }
        }
    {                                                // column sums: lane l and lane l + 32 hold the two row groups of column (wn * 4 + wk) * 32 + (l & 31)
        const float other = __shfl_xor(colsum, 32);
        if (lane < 32) p.bias_slab[slice * SLAB + (wn * 4 + wk) * 32 + lane] = colsum + other;
    }
}
// The same for a layer with 32 outputs (the value MLP's last layer: G rows of 64 bytes, X rows of 512): one n-tile, wave w the k-tile w -- one matrix
// instruction per wave and k-step, the kernel is the request stream.  A stage (32 rows) is 18 KB here: EIGHT buffers, seven stages on their way.
// G's rows arrive 16 to an instruction (lane: row lane >> 2, 16-byte piece lane & 3; waves 0 and 1), row-major at 64 bytes -- four consecutive rows
// on the four quarters of the banks as they are --, the maxima with wave 2.  One barrier per stage, in front of the stage's reads.
constexpr int TQ_D = 8;
constexpr int TQ_XOP = 16 * TR_PAIR, TQ_GOP = 32 * 64;
constexpr int TQ_STAGE = TQ_XOP + TQ_GOP;
constexpr int TQ_OFF_RAW = TQ_D * TQ_STAGE;
constexpr int TQ_OFF_C = TQ_OFF_RAW + TQ_D * 256;
constexpr int TQ_OFF_IG = TQ_OFF_C + TQ_D * 64;
constexpr size_t TQ_LDS_BYTES = TQ_OFF_IG + TQ_D * 128;

__global__ __launch_bounds__(512, 2) void gemm_tn_tr_n32_kernel(TNH3Batch batch) {
    extern __shared__ __attribute__((aligned(16))) char tr_smem[];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)tr_smem;
    const int jb = batch.par ? (int)(blockIdx.x % (unsigned)batch.n) : 0;
    const long slice = batch.par ? (long)(blockIdx.x / (unsigned)batch.n) : (long)blockIdx.x;
    const TNH3Args& p = batch.job[jb];
    const long mbeg = slice * p.rows_per_slice;
    long mend = mbeg + p.rows_per_slice;
    if (mend > p.M) mend = p.M;
    if (mbeg >= mend) return;
    const int nrows = (int)(mend - mbeg);
    const int g_rs = p.g_rs, x_rs = p.x_rs;
    auto pair_exp = [&](float gmx, float xmx) { return (float)tr_pair_exp(gmx, xmx, g_rs, x_rs); };      // (an exponent; exact as a float)
    float pm = (float)TR_EXP_NONE;
    {
        const int n4 = nrows >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(p.gmax + mbeg);
        const float4* x4 = reinterpret_cast<const float4*>(p.xmax + mbeg);
        for (int q = tid; q < n4; q += 512) {
            const float4 a = g4[q], b = x4[q];
            pm = fmaxf(fmaxf(pm, fmaxf(pair_exp(a.x, b.x), pair_exp(a.y, b.y))), fmaxf(pair_exp(a.z, b.z), pair_exp(a.w, b.w)));
        }
        for (int m = 4 * n4 + tid; m < nrows; m += 512) pm = fmaxf(pm, pair_exp(p.gmax[mbeg + m], p.xmax[mbeg + m]));
    }
    pm = wave_max(pm);
    if (lane == 0) red[wave] = pm;
    __syncthreads();
    float Pf = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) Pf = fmaxf(Pf, red[w]);
    const int eP = Pf <= (float)(TR_EXP_NONE / 2) ? 0 : (int)Pf;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float colsum = 0.f;                          // (wave 0)
    const _Float16* const Gh = reinterpret_cast<const _Float16*>(p.G);
    const _Float16* const Xh = reinterpret_cast<const _Float16*>(p.X);
    const long nst = (nrows + TN_ROWS - 1) / TN_ROWS;
    // a wave's requests of a stage, in order: [G rows (waves 0, 1) or the maxima (wave 2)], two row pairs of X
    auto issue = [&](long st) {
        const unsigned buf = lds0 + (unsigned)(st % TQ_D) * TQ_STAGE;
        if (wave < 2) {
#ifndef TQ_ABL_NO_G
            int m = (int)st * TN_ROWS + 16 * wave + (lane >> 2);
            m = m < nrows ? m : nrows - 1;
            tr_dma16(Gh + (mbeg + m) * p.ldg + 8 * (lane & 3), buf + (unsigned)(TQ_XOP + wave * 1024));
#endif
        } else if (wave == 2) {
            int m = (int)st * TN_ROWS + (lane & 31);
            m = m < nrows ? m : nrows - 1;
#ifndef TQ_ABL_NO_MAXIMA
            tr_dma4((lane >> 5 ? p.xmax : p.gmax) + mbeg + m, lds0 + TQ_OFF_RAW + (unsigned)(st % TQ_D) * 256);
#endif
        }
#ifdef TQ_ABL_NO_X
        return;
#endif
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pair = 2 * wave + q;
            int m = (int)st * TN_ROWS + pair + 16 * (lane >> 5);
            m = m < nrows ? m : nrows - 1;
            tr_dma16(Xh + (mbeg + m) * p.ldx + 8 * (lane & 31), buf + (unsigned)(pair * TR_PAIR));
        }
    };
    auto make_table = [&](long st) {             // (wave 2, lanes 0 .. 31)
        if (wave != 2 || lane >= 32) return;
        const int b = (int)(st % TQ_D);
        const float* raw = reinterpret_cast<const float*>(tr_smem + TQ_OFF_RAW + b * 256);
        const bool ok = (int)st * TN_ROWS + lane < nrows;
        reinterpret_cast<_Float16*>(tr_smem + TQ_OFF_C + b * 64)[lane] = ok ? (_Float16)tr_factor(tr_pair_exp(raw[lane], raw[32 + lane], g_rs, x_rs), eP) : (_Float16)0.f;
        reinterpret_cast<float*>(tr_smem + TQ_OFF_IG + b * 128)[lane] = ok ? inv_scale_from_row_max(raw[lane], g_rs) : 0.f;
    };
    const unsigned xfrag = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * TR_PAIR + 32 * ((lane >> 4) & 1) + 8 * (lane & 3)) + (unsigned)wave * 64;
    const unsigned gfrag = (unsigned)(TQ_XOP + (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + 32 * ((lane >> 4) & 1) + 8 * (lane & 3));
    for (long st = 0; st < TQ_D - 1; ++st) issue(st);
    for (long st = 0; st < nst; ++st) {
        // the stage's own requests have landed when those of the six stages behind it are outstanding (wave 2: and the maxima of the next stage)
        if (wave < 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (wave == 2) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if (st == 0) make_table(0);
        lds_barrier();
#ifndef TQ_ABL_NO_DMA
        issue(st + TQ_D - 1);
#endif
#ifndef TQ_ABL_NO_TABLE
        make_table(st + 1);
#endif
#ifdef TQ_ABL_NO_COMPUTE
        continue;
#endif
        const unsigned sb = lds0 + (unsigned)(st % TQ_D) * TQ_STAGE;
        const char* const tb = tr_smem + (st % TQ_D) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8 xb = tr_read8(sb + xfrag + ks * 512);
            const half8 gr = tr_read8<4 * 64>(sb + gfrag + ks * 16 * 64);
            const half8 cv = *reinterpret_cast<const half8*>(tb + TQ_OFF_C + (ks * 16 + 8 * (lane >> 5)) * 2);
            if (wave == 0) {
                const float* ig = reinterpret_cast<const float*>(tr_smem + TQ_OFF_IG + (st % TQ_D) * 128) + ks * 16 + 8 * (lane >> 5);
                const float4 i0 = *reinterpret_cast<const float4*>(ig), i1 = *reinterpret_cast<const float4*>(ig + 4);
                colsum = __builtin_fmaf((float)gr[7], i1.w, __builtin_fmaf((float)gr[6], i1.z, __builtin_fmaf((float)gr[5], i1.y, __builtin_fmaf((float)gr[4], i1.x,
                         __builtin_fmaf((float)gr[3], i0.w, __builtin_fmaf((float)gr[2], i0.z, __builtin_fmaf((float)gr[1], i0.y, __builtin_fmaf((float)gr[0], i0.x, colsum))))))));
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(gr * cv, xb, acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float P1 = pow2_from_biased(127 + (eP >> 1)), P2 = pow2_from_biased(127 + eP - (eP >> 1));
    float* out = p.slab + slice * SLAB * SLAB;
    const int k = wave * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * SLAB + k] = acc[e] * P1 * P2;
    if (wave == 0) {
        const float other = __shfl_xor(colsum, 32);
        if (lane < 32) p.bias_slab[slice * SLAB + lane] = colsum + other;
    }
}
#pragma clang diagnostic pop

// dW = G^T X on the split-f16 kernel; gmax / xmax: per-row max |.| of G and X (M floats each).  Jobs queue up and go out
// TN_BATCH to a launch (flush() at the latest before anything reads a result or reuses the partial-tile workspace).
constexpr size_t TN_JOB_FLOATS = (size_t)MAX_SLICES * (SLAB * SLAB + SLAB);
struct TNH3Queue {
    TNH3Batch batch;
    ReduceBatch red;
    bool full = false;
    int half = 0;
    int grid = 0;
    long long bytes = 0, flops = 0;
    void* workspace;
    hipStream_t s;
    TNH3Queue(void* ws, hipStream_t st) : workspace(ws), s(st) { batch.n = 0; }
    int push(const float* G, long ldg, int N, const float* X, long ldx, int K, long M, const float* gmax, const float* xmax,
             float* dW, int ldw, float* db, int g_half = 0, int x_half = 0, int g_rs = 0, int x_rs = 0) {
        PAPR_REQUIRE(N <= SLAB && K <= SLAB, "gemm_tn_h3: N=%d, K=%d exceed %d", N, K, SLAB);
        PAPR_REQUIRE(N % 4 == 0 && K % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0, "gemm_tn_h3: sizes must be multiples of 4");
        if (M <= 0) return 0;
        const bool job_full = N > 131 && K > 131;                       // 128 + 3 < N: all eight tiles live
        PAPR_REQUIRE(g_half || !x_half, "gemm_tn_h3: X f16 rows with G fp32 rows");
        // format 4 (round 6): both operands f16 rows of a full 256 x 256 layer with 512-byte rows -- rows by LDS-DMA, operands by the transposing LDS read
        const bool tr = g_half && x_half && N == SLAB && K == SLAB && ldg == SLAB && ldx == SLAB && papr_switch(PAPR_SW_TN_TR) != 0 &&
                        (reinterpret_cast<size_t>(G) & 15) == 0 && (reinterpret_cast<size_t>(X) & 15) == 0;
        // format 5: the same with 32 outputs (G rows of 32 halfs at any stride that keeps them 16-byte aligned)
        const bool trn = g_half && x_half && N == 32 && K == SLAB && ldx == SLAB && ldg % 8 == 0 && papr_switch(PAPR_SW_TN_TR) != 0 &&
                         (reinterpret_cast<size_t>(G) & 15) == 0 && (reinterpret_cast<size_t>(X) & 15) == 0;
        const int fmt = tr ? 4 : (trn ? 5 : (g_half ? (x_half ? 2 : 3) : 0));
        // (jobs with and without dead tiles share a launch -- on the instantiation with the tile tests: one launch and one reduction per run
        // instead of two, 0.05 ms per step; a batch of full jobs only keeps the test-free one)
        if (batch.n == TN_BATCH || (batch.n > 0 && fmt != half))
            if (int e = flush()) return e;
        full = batch.n == 0 ? job_full : (full && job_full); half = fmt;
        const int n_cu = papr_cu_count() > MAX_SLICES ? MAX_SLICES : papr_cu_count();
        long stages = (M + TN_ROWS - 1) / TN_ROWS;
        int S = (int)(stages < n_cu ? stages : n_cu);                   // one workgroup per CU streams one slice
        long rows_per_slice = ((stages + S - 1) / S) * TN_ROWS;
        S = (int)((M + rows_per_slice - 1) / rows_per_slice);
        TNH3Args& a = batch.job[batch.n];
        a.G = G; a.ldg = ldg; a.N = N; a.X = X; a.ldx = ldx; a.K = K; a.M = M; a.rows_per_slice = rows_per_slice;
        a.gmax = gmax; a.xmax = xmax; a.g_half = g_half; a.x_half = x_half; a.g_rs = g_rs; a.x_rs = x_rs;
        a.slab = static_cast<float*>(workspace) + (size_t)batch.n * TN_JOB_FLOATS;
        a.bias_slab = a.slab + (size_t)MAX_SLICES * SLAB * SLAB;
        red.job[batch.n] = ReduceJob{a.slab, a.bias_slab, S, N, K, dW, ldw, db};
        grid = S > grid ? S : grid;
        bytes += (g_half ? 2LL : 4LL) * M * N + (x_half ? 2LL : 4LL) * M * K; flops += 2LL * M * N * K;
        ++batch.n;
        return 0;
    }
    int flush() {
        if (batch.n == 0) return 0;
        // job-parallel layout: with several jobs of one length the workgroups are dealt to the jobs (see the kernel), evenly: a slice's time
        // follows its ROWS (a 32-row stage costs the same barriers and the same wait whatever the widths: dealt by bytes, the narrow
        // 256 -> 32 job of the value run got 18 of 256 workgroups and held the launch for 2.26 ms instead of 1.44)
        batch.par = 0;
        if (batch.n > 1 && papr_switch(PAPR_SW_TN_JOBPAR)) {
            bool same = true;
            for (int j = 0; j < batch.n; ++j) same &= batch.job[j].M == batch.job[0].M;
            const int n_cu = papr_cu_count() > MAX_SLICES ? MAX_SLICES : papr_cu_count();
            if (same && n_cu / batch.n >= 8) {
                const long M = batch.job[0].M, stages = (M + TN_ROWS - 1) / TN_ROWS;
                int S = n_cu / batch.n;
                S = (int)(stages < S ? stages : S);
                const long rows_per_slice = ((stages + S - 1) / S) * TN_ROWS;
                S = (int)((M + rows_per_slice - 1) / rows_per_slice);
                for (int j = 0; j < batch.n; ++j) { batch.job[j].rows_per_slice = rows_per_slice; red.job[j].S = S; }
                batch.par = 1;
                grid = S * batch.n;                 // (workgroup b: slice b / n of job b % n -- neighbouring workgroups, which share an XCD's L2 in turn, stream different jobs)
            }
        }
        if (papr_first_on_device(PAPR_ONCE_TN_H3)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_h3_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_h3_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_h3_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_h3_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_h3_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_h3_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_h3_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_h3_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS_BYTES);
        }
        const bool prof = papr_prof_on();
        if (prof) papr_prof_begin2(8, batch.job[0].M, batch.n, 0, bytes, flops, s);
        if (half == 4 || half == 5) {
            if (papr_first_on_device(PAPR_ONCE_TN_TR)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TR_LDS_BYTES);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tr_n32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TQ_LDS_BYTES);
            }
            if (half == 4) gemm_tn_tr_kernel<<<dim3(grid), dim3(512), TR_LDS_BYTES, s>>>(batch);
            else gemm_tn_tr_n32_kernel<<<dim3(grid), dim3(512), TQ_LDS_BYTES, s>>>(batch);
        } else if (half == 2) {                     // (f16 rows have one plane: one product -- also behind the parity arithmetic's runs, PAPR_MLP_H3_F16ROWS)
            if (full) gemm_tn_h3_kernel<true, 2><<<dim3(grid), dim3(512), T3_LDS_BYTES, s>>>(batch);
            else gemm_tn_h3_kernel<false, 2><<<dim3(grid), dim3(512), T3_LDS_BYTES, s>>>(batch);
        } else if (half == 3) {
            if (full) gemm_tn_h3_kernel<true, 3><<<dim3(grid), dim3(512), T3_LDS_BYTES, s>>>(batch);
            else gemm_tn_h3_kernel<false, 3><<<dim3(grid), dim3(512), T3_LDS_BYTES, s>>>(batch);
        } else if (GEMM_ONE_PRODUCT) {
            if (full) gemm_tn_h3_kernel<true, 1><<<dim3(grid), dim3(512), T3_LDS_BYTES, s>>>(batch);
            else gemm_tn_h3_kernel<false, 1><<<dim3(grid), dim3(512), T3_LDS_BYTES, s>>>(batch);
        } else if (full) gemm_tn_h3_kernel<true, 0><<<dim3(grid), dim3(512), T3_LDS_BYTES, s>>>(batch);
        else gemm_tn_h3_kernel<false, 0><<<dim3(grid), dim3(512), T3_LDS_BYTES, s>>>(batch);
        if (prof) papr_prof_end(s);
        PAPR_CHECK_LAUNCH("gemm_tn_h3");
        if (half == 4 || half == 5) slab_reduce_kernel<false><<<dim3(SLAB * SLAB / 4 / 16, batch.n), dim3(256), 0, s>>>(red);      // (natural order: no permutation to undo)
        else slab_reduce_kernel<true><<<dim3(SLAB * SLAB / 4 / 16, batch.n), dim3(256), 0, s>>>(red);
        PAPR_CHECK_LAUNCH("slab_reduce");
        batch.n = 0; grid = 0; bytes = 0; flops = 0;
        return 0;
    }
};

// in-place g *= act'(y)
__global__ __launch_bounds__(256) void act_grad_kernel(float* g, long ldg, const float* __restrict__ y, long ldy, long M,
                                                       int N, int act) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * N) return;
    long m = e / N;
    int n = (int)(e - m * N);
    g[m * ldg + n] *= papr_act_grad(y[m * ldy + n], act);
}

}  // namespace

constexpr size_t TN_SLAB_BYTES = (size_t)TN_BATCH * TN_JOB_FLOATS * sizeof(float);      // partial tiles of a batch of weight-gradients

extern "C" size_t papr_mlp_fwd_workspace_bytes(int64_t M) { return H3Scratch::bytes(M) + SRScratch::reserve(M); }
extern "C" size_t papr_mlp_saved_floats(int32_t n_layers, int64_t M) { return (size_t)n_layers * (M + CHAIN_SIGN_WORDS * chain_sign_rows(M)); }

// Layout of the `row_absmax` buffer of papr_mlp_fwd / papr_mlp_bwd (papr_mlp_saved_floats(n_layers, M) floats):
// [n_layers][M] row maxima, then [n_layers][CHAIN_SIGN_WORDS][chain_sign_rows(M)] sign words of the layer outputs (fused runs, chain.h).
static unsigned* saved_sign_words(const float* saved, int n_layers, long M, int layer) {
    return reinterpret_cast<unsigned*>(const_cast<float*>(saved)) + (size_t)n_layers * M + (size_t)layer * CHAIN_SIGN_WORDS * chain_sign_rows(M);
}

// layer i runs on the split-f16 forward kernel (and so leaves the row maxima of its input behind)
static bool layer_on_h3(const papr_layer& L) { return GEMM_H3_FWD && L.n_out > 128 && L.n_skip == 0; }

// Layers [b, e) that one fused launch (chain.hip) can carry: no skip inputs, at most 256 wide, the widths between
// two fused layers multiples of 32.  Returns e (e - b < 2: no fusion).
static int chain_run_end(const papr_layer* layers, int n_layers, int b, bool allow_skip = false) {
    if (!GEMM_CHAIN) return b;
    int e = b;
    while (e < n_layers && e - b < CHAIN_MAX_LAYERS) {
        const papr_layer& L = layers[e];
        if (L.n_skip > 0) {
            // a skip layer rides in a FORWARD run that starts at layer 0 (its second K segment is the run's own input)
            if (!(allow_skip && b == 0 && e > 0 && L.n_skip == layers[0].n_in && L.skip_col == L.n_in && L.n_in % 64 == 0)) break;
        }
        if (L.n_in > 256 || L.n_out > 256 || L.n_in % 4 || L.n_out % 4 || L.n_skip > 256) break;
        if (e > b && (layers[e - 1].n_out % 32 || layers[e - 1].n_out != L.n_in)) break;
        ++e;
    }
    return e;
}

// the forward pass left max |input row| of layer i in row_absmax: split-f16 layers and members of fused runs
static bool layer_rowmax_saved(const papr_layer* layers, int n_layers, int i) {
    for (int b = 0; b < n_layers;) {
        const int e = chain_run_end(layers, n_layers, b, true);
        if (e - b >= 2) {
            if (i >= b && i < e) return true;
            b = e;
        } else ++b;
    }
    return layer_on_h3(layers[i]);
}

// h1 mode: the rows a fused run leaves for its weight-gradients are f16 (scaled per row, chain.h: c_half) instead
// of fp32 -- the weight-gradient kernel is HBM-bound, so half the bytes is half its time.  The buffers stay the caller's
// fp32-sized ones: the f16 rows of a layer's output occupy the first half of outs[l] (row stride ld_out[l] halfs), and the f16
// copy of the run's input rows the second half of the run's first output buffer; likewise the gradient-row slots of the
// backward scratch.  Only the run's last output (forward) / the gradient that leaves the run (backward) remain fp32 rows.
// mode PAPR_MLP_H1_F32ROWS keeps fp32 rows (a different computation: other bits in the weight gradients; A/B).
#define H1_HALF_ROWS (one_product_now() || t_h3_f16_rows)
// forward run [b, e) of a training pass stores f16 rows (the backward pass asks the same question)
static bool run_half_rows(const papr_layer* layers, int n_layers, int b, int e, const int32_t* ld_out, bool training) {
    if (!H1_HALF_ROWS || !training || e - b < 2) return false;
    for (int l = 0; l < n_layers; ++l) if (layers[l].n_skip > 0) return false;          // (the backward pass cuts its runs at skip layers: the forward run's f16 rows would not line up)
    const int k0pad = (layers[b].n_in + 31) / 32 * 32;
    if (k0pad > ld_out[b]) return false;                                                // the input copy must fit behind outs[b]'s rows
    for (int l = b; l < e - 1; ++l) if (ld_out[l] % 8 || layers[l].n_out % 32) return false;
    return true;
}

// One-product mode (round 6): its fused runs keep f16 rows only (one scale per row and run: chain4.hip).  A TRAINING call with a run that cannot keep them
// (skip layers -- the backward pass cuts its runs there --, a middle layer whose width is no multiple of 32, an input copy that does not fit behind the run's
// first output), and a backward call without the forward pass's saved state, run in the parity arithmetic as a whole -- forward and backward take the same
// decision from the same arguments.
static void one_product_or_parity(const papr_layer* layers, int n_layers, const int32_t* ld_out, bool training, bool backward = false) {
    if (GEMM_ONE_PRODUCT && backward && !training) { t_mode = 4; return; }
    if (!GEMM_ONE_PRODUCT || !training) return;
    for (int b = 0; b < n_layers;) {
        const int e = chain_run_end(layers, n_layers, b, true);
        if (e - b >= 2) { if (!run_half_rows(layers, n_layers, b, e, ld_out, true)) { t_mode = 4; return; } b = e; }
        else ++b;
    }
}

// pre-split W (N x K, leading dimension ldw) into fragment-order planes at `planes`; returns the halfs used
// All weights of a fused run are split by ONE launch (grid.y = layer).  transposed: the planes hold W^T (the
// data-gradient run multiplies with the transposed weights; reading W column-wise here saves the caller a transpose).
struct SplitJob { const float* W; int N, K, ldw, transposed, n_tiles, ksteps; _Float16* hi; _Float16* lo; };
// perm: chain4.hip's fragment rows -- matrix-instruction row j of a 32-column tile carries column 16 ((j >> 2) & 1) + 4 (j >> 3) + (j & 3), so that
// an accumulator lane's 16 registers are 16 consecutive columns
struct SplitBatch { SplitJob job[CHAIN_MAX_LAYERS]; int perm; };

__global__ __launch_bounds__(256) void split_weight_batch_kernel(SplitBatch b) {
    const SplitJob& j = b.job[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= j.n_tiles * j.ksteps * 512) return;
    const int e = idx & 7, l = (idx >> 3) & 63, f = idx >> 9;
    const int t = f / j.ksteps, ks = f - t * j.ksteps;
    const int jr = l & 31;
    const int n = 32 * t + (b.perm ? 16 * ((jr >> 2) & 1) + 4 * (jr >> 3) + (jr & 3) : jr), c = 16 * ks + 8 * (l >> 5) + e;
    float x = 0.f;
    if (n < j.N && c < j.K) x = j.transposed ? j.W[(long)c * j.ldw + n] : j.W[(long)n * j.ldw + c];
    const _Float16 h = (_Float16)x;
    j.hi[idx] = h;
    j.lo[idx] = (_Float16)(x - (float)h);
}

// queue W (N x K result-by-input, or its transpose stored K x N with leading dimension ldw) for the run's split launch
static size_t chain_queue_weight(SplitBatch& b, int slot, const float* W, int N, int K, int ldw, int transposed, _Float16* planes, ChainLayer& L) {
    const int pitch = (K + 31) / 32 * 32, n_tiles = (N + 31) / 32;
    const size_t plane = (size_t)n_tiles * 32 * pitch;
    b.job[slot] = SplitJob{W, N, K, ldw, transposed, n_tiles, pitch / 16, planes, planes + plane};
    L.w_hi = planes; L.w_lo = planes + plane; L.ksteps = pitch / 16; L.k1steps = pitch / 16; L.N = N;
    return 2 * plane;
}
static int chain_split_launch(const SplitBatch& b, int n, hipStream_t s) {
    int most = 0;
    for (int i = 0; i < n; ++i) most = std::max(most, b.job[i].n_tiles * b.job[i].ksteps * 512);
    split_weight_batch_kernel<<<dim3((unsigned)((most + 255) / 256), (unsigned)n), dim3(256), 0, s>>>(b);
    PAPR_CHECK_LAUNCH("split_weight");
    return 0;
}

extern "C" int papr_mlp_fwd(const papr_layer* layers, int n_layers, float* x, int ldx, int64_t M,
                            float* const* outs, const int32_t* ld_out, float* row_absmax, const papr_row_norm* in_norm,
                            const papr_row_norm* out_norm, void* workspace, int32_t mode, papr_stream_t stream) {
    PAPR_REQUIRE(mode_from_arg(mode), "papr_mlp_fwd: unknown mode %d", mode);
    PAPR_REQUIRE(layers && x && outs && ld_out && n_layers >= 1, "papr_mlp_fwd: bad arguments");
    one_product_or_parity(layers, n_layers, ld_out, row_absmax != nullptr);
    hipStream_t s = as_stream(stream);
    bool have_amax = false;                 // split-f16 mode: row maxima of the current layer's input are in h3.in()
    bool norm_done = false;
    PAPR_REQUIRE(!out_norm || (out_norm->stats && out_norm->width >= 2), "papr_mlp_fwd: out_norm needs stats and a width");
    const bool want_dots = out_norm && out_norm->dots;
    PAPR_REQUIRE(!want_dots || (out_norm->dot_rows && out_norm->rows_per_dot >= 1 && out_norm->ld_dot % 4 == 0 && out_norm->ld_dot >= out_norm->width),
                 "papr_mlp_fwd: out_norm->dots needs dot_rows with 16-byte rows of at least `width` floats and rows_per_dot >= 1");
    bool dots_done = false;
    PAPR_REQUIRE(!in_norm || (in_norm->stats && in_norm->width >= 2 && in_norm->width <= ldx), "papr_mlp_fwd: in_norm needs stats and a width <= ldx");
    if (in_norm && !(chain_run_end(layers, n_layers, 0, true) >= 2)) {     // not staged by a fused run: one pass over x first
        if (in_norm->given_mean) { if (int e = papr_rownorm_apply(x, M, in_norm->width, ldx, in_norm->stats, in_norm->given_mean, stream)) return e; }
        else if (int e = papr_rownorm_fwd(x, M, in_norm->width, ldx, in_norm->eps, x, in_norm->stats, stream)) return e;
    }
    PAPR_REQUIRE(!GEMM_H3_FWD || workspace, "papr_mlp_fwd: workspace required (papr_mlp_fwd_workspace_bytes)");
    H3Scratch h3(workspace, M);
    for (int i = 0; i < n_layers; ++i) {
        const papr_layer& L = layers[i];
        PAPR_REQUIRE(L.weight && outs[i], "papr_mlp_fwd: layer %d has null weight/output", i);
        PAPR_REQUIRE(ld_out[i] >= L.n_out, "papr_mlp_fwd: layer %d output stride %d < %d", i, ld_out[i], L.n_out);
        if (const int e = chain_run_end(layers, n_layers, i, true); e - i >= 2) {
            // layers [i, e) in one launch; without row_absmax (inference) only the run's last result reaches memory
            float* saved = row_absmax;
            ChainArgs c = {};
            c.A0 = i == 0 ? x : outs[i - 1]; c.lda0 = i == 0 ? ldx : ld_out[i - 1]; c.K0 = L.n_in;
            if (i == 0 && in_norm) {                // standardised while the rows are staged
                bool run_has_skip = false;
                for (int l = i; l < e; ++l) run_has_skip |= layers[l].n_skip > 0;
                c.in_norm_width = in_norm->width; c.in_norm_eps = in_norm->eps; c.in_norm_stats = in_norm->stats;
                c.in_norm_mean = in_norm->given_mean;
                c.in_norm_writeback = (row_absmax != nullptr || run_has_skip || e < n_layers) ? 1 : 0;
                if (c.in_norm_mean == nullptr && !(!GEMM_ONE_PRODUCT && !t_h3_f16_rows && papr_switch(PAPR_SW_C4_DMA))) {
                    // nobody gave the statistics: one pass over the rows takes them (the run applies them while it stages the rows); the means go
                    // to the head of the scratch the split-ahead experiment would use
                    float* means = reinterpret_cast<float*>(static_cast<char*>(workspace) + H3Scratch::bytes(M));
                    if (int err = papr_rownorm_stats(c.A0, M, in_norm->width, (int)c.lda0, in_norm->eps, in_norm->stats, means, stream)) return err;
                    c.in_norm_mean = means;
                }
            }
            c.rowmax0 = saved ? saved + (size_t)i * M : nullptr;
            c.M = M; c.n_layers = e - i;
            c.one_product = GEMM_ONE_PRODUCT ? 1 : 0;
            if (!GEMM_ONE_PRODUCT && !t_h3_f16_rows && papr_switch(PAPR_SW_C4_DMA) && c.in_norm_mean == nullptr) {
                // the run's input rows split ahead of it: the run stages its tiles by LDS-DMA (chain.h: sr_*)
                SRScratch sr(static_cast<char*>(workspace) + H3Scratch::bytes(M), M);
                SplitRowsArgs q = {};
                q.x = c.A0; q.ld = c.lda0; q.K = c.K0; q.Kq = (c.K0 + 31) / 32 * 32; q.M = M;      // (the first k-loop reads whole pairs of k-steps: the planes are zero up to there)
                q.hi = sr.hi; q.lo = sr.lo; q.inv = sr.inv; q.mx = c.rowmax0 ? c.rowmax0 : sr.mx;
                q.nw = c.in_norm_width; q.eps = c.in_norm_eps; q.stats = c.in_norm_stats; q.writeback = c.in_norm_writeback;
                if (int err = launch_split_rows(q, s)) return err;
                c.sr_hi = q.hi; c.sr_lo = q.lo; c.sr_inv = q.inv; c.sr_max = q.mx; c.sr_ld = q.Kq;
            }
            const bool half_rows = run_half_rows(layers, n_layers, i, e, ld_out, saved != nullptr);
            // (papr_row_norm.leave_input: nobody reads the standardised rows back -- the caller said so, and this mode's weight gradient reads the f16 copy)
            if (i == 0 && in_norm && in_norm->leave_input && half_rows && GEMM_ONE_PRODUCT && e == n_layers) c.in_norm_writeback = 0;
            if (half_rows && GEMM_ONE_PRODUCT) { c.a0_half = reinterpret_cast<_Float16*>(outs[i]) + (size_t)M * ld_out[i]; c.lda0_half = ld_out[i]; }      // (PAPR_MLP_H3_F16ROWS: no copy -- the first weight gradient reads the fp32 rows)
            size_t used = 0;
            SplitBatch split = {};
            long long bytes = 4LL * M * L.n_in, flops = 0;
            for (int l = i; l < e; ++l) {
                ChainLayer& cl = c.L[l - i];
                PAPR_REQUIRE(layers[l].weight && outs[l] && ld_out[l] >= layers[l].n_out, "papr_mlp_fwd: layer %d has null weight/output", l);
                used += chain_queue_weight(split, l - i, layers[l].weight, layers[l].n_out, layers[l].n_in + layers[l].n_skip, layers[l].ldw, 0, h3.planes + used, cl);
                if (layers[l].n_skip > 0) cl.k1steps = layers[l].n_in / 16;      // [previous output | x]: the second segment multiplies the run's input again
                cl.bias = layers[l].bias; cl.act = layers[l].act;
                cl.C = (saved || l == e - 1) ? outs[l] : nullptr; cl.ldc = ld_out[l];
                cl.c_half = half_rows && l < e - 1 ? 1 : 0;
                cl.sign_bits = saved && layers[l].act != PAPR_ACT_NONE ? saved_sign_words(saved, n_layers, M, l) : nullptr;
                if (cl.C) bytes += (cl.c_half ? 2LL : 4LL) * M * layers[l].n_out;
                bytes += 4LL * layers[l].n_out * (layers[l].n_in + layers[l].n_skip) + 4LL * M * layers[l].n_skip;
                flops += 2LL * M * layers[l].n_out * (layers[l].n_in + layers[l].n_skip);
                cl.rowmax = l + 1 < n_layers ? (saved ? saved + (size_t)(l + 1) * M : (l == e - 1 ? reinterpret_cast<float*>(h3.out()) : nullptr)) : nullptr;
            }
            PAPR_REQUIRE(used <= H3_PLANE_HALFS, "papr_mlp_fwd: fused run needs %zu plane halfs", used);
            if (out_norm && e == n_layers && layers[e - 1].act == PAPR_ACT_NONE && out_norm->width == layers[e - 1].n_out) {
                c.norm_eps = out_norm->eps; c.norm_stats = out_norm->stats;      // standardised in the run's last row phase
                norm_done = true;
                if (want_dots) {                    // ... and multiplied with its ray's row there; inference: the rows themselves stay on the chip
                    c.dot_rows = out_norm->dot_rows; c.ld_dot = out_norm->ld_dot; c.rows_per_dot = out_norm->rows_per_dot; c.dots = out_norm->dots;
                    dots_done = true;
                    if (saved) c.norm_mean = out_norm->raw_mean;      // (training: the rows leave raw, papr_row_norm.raw_mean)
                    if (!saved) { bytes -= 4LL * M * layers[e - 1].n_out; c.L[e - 1 - i].C = nullptr; }
                    bytes += 4LL * M + 4LL * ((M + c.rows_per_dot - 1) / c.rows_per_dot) * layers[e - 1].n_out;
                }
            }
            split.perm = 1;
            if (int err = chain_split_launch(split, e - i, s)) return err;
            if (int err = papr_launch_chain(c, false, bytes, flops, s)) return err;
            if (!saved) h3.swap();
            have_amax = true;
            i = e - 1;
            continue;
        }
        NTArgs a = {};
        a.A = i == 0 ? x : outs[i - 1];
        a.lda = i == 0 ? ldx : ld_out[i - 1];
        a.K1 = L.n_in;
        if (L.n_skip > 0) { a.A2 = x; a.lda2 = ldx; a.K2 = L.n_skip; a.wcol2 = L.skip_col; }
        a.W = L.weight; a.ldw = L.ldw; a.bias = L.bias; a.act = L.act;
        a.C = outs[i]; a.ldc = ld_out[i]; a.M = M; a.N = L.n_out;
        const bool skip_h3 = GEMM_H3_FWD && L.n_skip > 0 && L.n_out > 128 && L.n_in % 32 == 0 && L.skip_col % 32 == 0;
        if (layer_on_h3(L) || skip_h3) {
            // with row_absmax the maxima of layer i's input rows stay in row_absmax[i*M ..) for papr_mlp_bwd
            if (skip_h3) {                  // the row's scale must cover both K segments: maxima of x as well
                if (int e = launch_row_absmax(x, M, L.n_skip, ldx, h3.amax_x, s)) return e;
                a.amax_in2 = h3.amax_x;
            }
            unsigned* saved = reinterpret_cast<unsigned*>(row_absmax);
            unsigned* in = saved ? saved + (size_t)i * M : h3.in();
            if (!have_amax)
                if (int e = launch_row_absmax(a.A, M, L.n_in, a.lda, in, s)) return e;
            a.amax_in = in;
            a.amax_out = saved && i + 1 < n_layers ? saved + (size_t)(i + 1) * M : h3.out();
            a.planes = h3.planes;
            PAPR_REQUIRE(hipMemsetAsync(a.amax_out, 0, (size_t)M * sizeof(unsigned), s) == hipSuccess, "papr_mlp_fwd: memset failed");
            if (int e = gemm_nt(a, s)) return e;
            h3.swap();
            have_amax = true;
            continue;
        }
        have_amax = false;
        if (int e = gemm_nt(a, s)) return e;
    }
    PAPR_REQUIRE(!(out_norm && out_norm->raw_mean) || (dots_done && row_absmax),
                 "papr_mlp_fwd: out_norm->raw_mean needs dots, row_absmax and a last layer that rides in a fused run (mode %d)", mode);
    if (out_norm && !norm_done)             // not inside a fused run: one more pass over the rows
        if (int e = papr_rownorm_fwd(outs[n_layers - 1], M, out_norm->width, ld_out[n_layers - 1], out_norm->eps, outs[n_layers - 1],
                                     out_norm->stats, stream)) return e;
    if (want_dots && !dots_done)
        return papr_row_dots(outs[n_layers - 1], M, out_norm->width, ld_out[n_layers - 1], out_norm->dot_rows, out_norm->ld_dot,
                             out_norm->rows_per_dot, out_norm->dots, stream);
    return 0;
}

// Scratch of the backward pass behind the slabs and the split-f16 scratch: the gradient rows of every layer of
// a fused run (the weight-gradients run after the run's data-gradient launch and need all of them) and their
// per-row maxima.
constexpr int G_LD = 256;
struct BwdRunScratch {
    float* g[CHAIN_MAX_LAYERS];
    float* gmax[CHAIN_MAX_LAYERS + 1];
    BwdRunScratch(void* base, long M) {
        float* q = static_cast<float*>(base);
        for (int i = 0; i < CHAIN_MAX_LAYERS; ++i) { g[i] = q; q += (size_t)M * G_LD; }
        for (int i = 0; i <= CHAIN_MAX_LAYERS; ++i) { gmax[i] = q; q += M; }
    }
    static size_t bytes(long M) { return ((size_t)CHAIN_MAX_LAYERS * M * G_LD + (size_t)(CHAIN_MAX_LAYERS + 1) * M + 4) * sizeof(float); }
};

extern "C" int papr_mlp_bwd_needs_weight_t(const papr_layer* layers, int n_layers, int need_dx, int32_t mode) {
    if (!mode_from_arg(mode) || !layers || n_layers < 1) return 1;
    bool any_skip = false;
    for (int i = 0; i < n_layers; ++i) any_skip |= layers[i].n_skip > 0;
    if (any_skip) return 1;                         // (skip layers and the accumulating layer-0 data-gradient read weight_t)
    for (int b = 0; b < n_layers;) {
        const int e = chain_run_end(layers, n_layers, b);
        if (e - b >= 2) { b = e; continue; }        // fused run: reads W^T out of `weight`
        if (b > 0 || need_dx) return 1;
        ++b;
    }
    return 0;
}

// papr_mlp_bwd's d_out_f16: the top layers are a fused run with f16 rows -- the one-product mode's (1) or PAPR_MLP_H3_F16ROWS' (2: the split form; the
// call as a whole runs in that arithmetic: t_mode is the caller's, one_product_or_parity() applied) --, no activation behind the last layer, and the
// launch has a layer that is not its last.  0: the call does not take them.
static int bwd_takes_f16_top(const papr_layer* layers, int n_layers, const int32_t* ld_out, bool need_dx) {
    if (!(GEMM_ONE_PRODUCT || t_h3_f16_rows) || layers[n_layers - 1].act != PAPR_ACT_NONE) return 0;
    int bt = -1;
    for (int b = 0; b < n_layers;) { const int e = chain_run_end(layers, n_layers, b); if (e - b >= 2) { if (e == n_layers) bt = b; b = e; } else ++b; }
    if (bt < 0 || !run_half_rows(layers, n_layers, bt, n_layers, ld_out, true)) return 0;
    const int last = bt == 0 ? (need_dx ? 0 : 1) : bt;
    if ((n_layers - 1) - last < 1) return 0;
    return GEMM_ONE_PRODUCT ? 1 : 2;
}

extern "C" int papr_mlp_bwd_takes_f16_rows(const papr_layer* layers, int n_layers, const int32_t* ld_out, int need_dx, int32_t mode) {
    if (!layers || !ld_out || n_layers < 1 || !mode_from_arg(mode)) return 0;
    one_product_or_parity(layers, n_layers, ld_out, true, true);
    return bwd_takes_f16_top(layers, n_layers, ld_out, need_dx != 0);
}

extern "C" size_t papr_mlp_bwd_workspace_bytes(int64_t M) { return TN_SLAB_BYTES + H3Scratch::bytes(M) + BwdRunScratch::bytes(M) + SRScratch::reserve(M); }

extern "C" int papr_mlp_bwd(const papr_layer* layers, int n_layers, const float* x, int ldx, int64_t M,
                            float* const* outs, const int32_t* ld_out, const float* row_absmax, float* d_out, const papr_f16_rows* d_out_f16,
                            float* scratch0, float* scratch1, int ld_scratch, float* const* d_weight,
                            float* const* d_bias, float* d_x, void* workspace, int32_t mode, papr_stream_t stream) {
    PAPR_REQUIRE(mode_from_arg(mode), "papr_mlp_bwd: unknown mode %d", mode);
    PAPR_REQUIRE(layers && x && outs && ld_out && (d_out || d_out_f16) && d_weight && d_bias && workspace && n_layers >= 1,
                 "papr_mlp_bwd: bad arguments");
    PAPR_REQUIRE(n_layers == 1 || (scratch0 && scratch1), "papr_mlp_bwd: scratch buffers required");
    one_product_or_parity(layers, n_layers, ld_out, row_absmax != nullptr, true);
    if (d_out_f16) {
        // the top gradient rows as the producer wrote them (papr_f16_rows): what the one-product data-gradient run's staging would have made of fp32 rows
        const int top_w = layers[n_layers - 1].n_out;
        const int kind = bwd_takes_f16_top(layers, n_layers, ld_out, d_x != nullptr);
        PAPR_REQUIRE(kind != 0 && (kind == 2) == (d_out_f16->lo != nullptr) && d_out_f16->hi && d_out_f16->inv && d_out_f16->scale && d_out_f16->max &&
                     d_out_f16->ld % 32 == 0 && d_out_f16->ld >= (top_w + 31) / 32 * 32,
                     "papr_mlp_bwd: d_out_f16 needs the form papr_mlp_bwd_takes_f16_rows() reports (this call: %d; lo %s) and rows of a multiple of 32 halfs",
                     kind, d_out_f16->lo ? "given" : "NULL");
    }
    hipStream_t s = as_stream(stream);
    bool any_skip = false;
    for (int i = 0; i < n_layers; ++i) any_skip |= layers[i].n_skip > 0;
    if (d_x && any_skip) {
        hipError_t e = hipMemsetAsync(d_x, 0, (size_t)M * ldx * sizeof(float), s);
        PAPR_REQUIRE(e == hipSuccess, "papr_mlp_bwd: memset failed");
    }
    H3Scratch h3(static_cast<char*>(workspace) + TN_SLAB_BYTES, M);
    BwdRunScratch runs(static_cast<char*>(workspace) + TN_SLAB_BYTES + H3Scratch::bytes(M), M);
    const unsigned* gmax = nullptr;          // per-row max |g| of the current gradient rows, when known
    auto need_gmax = [&](const float* g, long ldg, int width) -> int {
        if (gmax) return 0;
        if (int e = launch_row_absmax(g, M, width, ldg, h3.in(), s)) return e;
        gmax = h3.in();
        return 0;
    };
    // run boundaries as the forward pass saw them
    int run_begin[64];
    PAPR_REQUIRE(n_layers <= 64, "papr_mlp_bwd: %d layers", n_layers);
    for (int b = 0; b < n_layers;) {
        const int e = chain_run_end(layers, n_layers, b);
        if (e - b >= 2) { for (int l = b; l < e; ++l) run_begin[l] = b; b = e; }
        else { run_begin[b] = -1; ++b; }
    }
    // gradient w.r.t. the last layer's pre-activation
    float* g = d_out;
    long ldg = ld_out[n_layers - 1];
    {
        const papr_layer& L = layers[n_layers - 1];
        if (L.act != PAPR_ACT_NONE) {
            long n = (long)M * L.n_out;
            act_grad_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(g, ldg, outs[n_layers - 1], ldg, M, L.n_out, L.act);
            PAPR_CHECK_LAUNCH("act_grad");
        }
    }
    TNH3Queue tnq(workspace, s);             // split-f16 weight-gradients collect here; the fp32 kernel shares the workspace, so flush before it
    auto wgrad = [&](int i, const float* gi, long ldgi, const float* gmax_i, int g_half = 0, int x_half = 0, const float* x_rows = nullptr, long ld_x_rows = 0,
                     const float* xmax_run = nullptr) -> int {
        // (one-product mode, f16 rows: ONE scale per row and run -- the row maxima that stand for it are the run's top gradient rows' (gmax_i: the caller
        //  passes that table for every layer) and the run's input rows' (xmax_run))
        const int g_rs = g_half && GEMM_ONE_PRODUCT ? 2 : 0, x_rs = x_half && GEMM_ONE_PRODUCT ? 1 : 0;
        const float* const xmax_i = xmax_run ? xmax_run : row_absmax + (size_t)i * M;
        const papr_layer& L = layers[i];
        const float* in = x_half ? x_rows : (i == 0 ? x : outs[i - 1]);
        const long ld_in = x_half ? ld_x_rows : (i == 0 ? ldx : ld_out[i - 1]);
        PAPR_REQUIRE(d_weight[i], "papr_mlp_bwd: layer %d has null d_weight", i);
        // split-f16 when the forward pass left the row maxima of this layer's input
        const bool h3w = GEMM_H3_WGRAD && row_absmax && gmax_i && layer_rowmax_saved(layers, n_layers, i);
        // a layer wider than the 256 x 256 tile of the weight-gradient kernels (d_ff = 512 ...; round 5): block by block -- 256 output rows x 256
        // input columns of dW each, the operands' column windows through their row strides, the bias gradient with the first column block.  (The
        // row maxima are the whole rows': a window's slice scale is then at most the row's -- never too small.)
        for (int n0 = 0; n0 < L.n_out; n0 += SLAB)
            for (int k0 = 0; k0 < L.n_in; k0 += SLAB) {
                const int nb = L.n_out - n0 < SLAB ? L.n_out - n0 : SLAB, kb = L.n_in - k0 < SLAB ? L.n_in - k0 : SLAB;
                float* dw = d_weight[i] + (size_t)n0 * L.ldw + k0;
                float* db = (k0 == 0 && d_bias[i]) ? d_bias[i] + n0 : nullptr;
                if (h3w) {
                    PAPR_REQUIRE((!g_half && !x_half) || (n0 == 0 && k0 == 0 && nb == L.n_out && kb == L.n_in), "papr_mlp_bwd: layer %d: f16 rows of a blocked weight-gradient", i);
                    if (int e = tnq.push(gi + n0, ldgi, nb, in + k0, ld_in, kb, M, gmax_i, xmax_i, dw, L.ldw, db, g_half, x_half, g_rs, x_rs)) return e;
                } else {
                    PAPR_REQUIRE(!g_half && !x_half, "papr_mlp_bwd: layer %d: f16 rows without the split-f16 weight-gradient", i);
                    if (int e = tnq.flush()) return e;
                    if (int e = gemm_tn(gi + n0, ldgi, nb, in + k0, ld_in, kb, M, dw, L.ldw, db, workspace, s)) return e;
                }
            }
        if (L.n_skip > 0) {
            // the skip segment [previous output | x]: dW[:, skip_col ..) = G^T x -- split-f16 like the first segment when the forward
            // run left the maxima of the x rows (row_absmax[0 .. M): a fused run that starts at layer 0), fp32 MFMA otherwise
            if (h3w && L.n_skip <= SLAB && layer_rowmax_saved(layers, n_layers, 0)) {
                if (int e = tnq.push(gi, ldgi, L.n_out, x, ldx, L.n_skip, M, gmax_i, row_absmax, d_weight[i] + L.skip_col, L.ldw, nullptr, g_half, 0, g_rs, 0)) return e;
            } else {
                PAPR_REQUIRE(!g_half, "papr_mlp_bwd: layer %d: f16 gradient rows and a skip segment without the split-f16 weight gradient", i);
                if (int e = tnq.flush()) return e;
                if (int e = gemm_tn(gi, ldgi, L.n_out, x, ldx, L.n_skip, M, d_weight[i] + L.skip_col, L.ldw, nullptr, workspace, s)) return e;
            }
        }
        return 0;
    };
    for (int i = n_layers - 1; i >= 0; --i) {
        const papr_layer& L = layers[i];
        if (const int b = run_begin[i]; b >= 0) {
            // ---- fused run [b, i]: one data-gradient launch down to the gradient of the run's input, then the
            // weight-gradients of its layers
            const bool to_dx = b == 0 && d_x && !any_skip;       // layer 0's data-gradient joins the run unless d_x accumulates
            const int last = b == 0 ? (to_dx ? 0 : 1) : b;       // lowest layer whose data-gradient the launch computes
            ChainArgs c = {};
            c.A0 = g; c.lda0 = ldg; c.K0 = L.n_out; c.rowmax0 = runs.gmax[CHAIN_MAX_LAYERS];
            c.M = M; c.one_product = GEMM_ONE_PRODUCT ? 1 : 0;
            const bool top_f16 = d_out_f16 && i == n_layers - 1;     // (validated above: this run takes them)
            const bool split_ahead = !GEMM_ONE_PRODUCT && !t_h3_f16_rows && papr_switch(PAPR_SW_C4_DMA) != 0;       // (the kernel is launched below, once the run's width is known to be its own;
                                                                                                                  //  with f16 rows the staging also makes the top rows' f16 copy: not by DMA)
            // h1 mode: f16 rows (see run_half_rows).  x_half: what the forward run stored; g_half: this launch, if it has a layer
            // that is not its last (the copy of the top gradient rows goes behind that layer's rows)
            const bool x_half = run_half_rows(layers, n_layers, b, i + 1, ld_out, row_absmax != nullptr);
            const bool g_half = x_half && i - last >= 1;
            auto slot0 = [&](int l) { return l - (b > 0 ? b - 1 : 0); };
            if (g_half) { c.a0_half = reinterpret_cast<_Float16*>(runs.g[slot0(i - 1)]) + (size_t)M * G_LD; c.lda0_half = G_LD; }
            if (top_f16) {                          // staged by LDS-DMA; the rows are their own f16 copy, their maxima the producer's table
                PAPR_REQUIRE(g_half, "papr_mlp_bwd: d_out_f16 and a top run without f16 gradient rows");
                c.A0 = nullptr; c.rowmax0 = nullptr; c.a0_half = nullptr;
                c.sr_hi = reinterpret_cast<const _Float16*>(d_out_f16->hi); c.sr_lo = reinterpret_cast<const _Float16*>(d_out_f16->lo); c.sr_inv = d_out_f16->inv;
                c.sr_max = GEMM_ONE_PRODUCT ? d_out_f16->scale : d_out_f16->max;       // (the tile's second table: the scale itself in the one-product mode, the maximum in the parity arithmetic)
                c.sr_ld = d_out_f16->ld;
            }
            size_t used = 0;
            SplitBatch split = {};
            long long bytes = 4LL * M * L.n_out, flops = 0;
            for (int l = i; l >= last; --l) {
                const papr_layer& Ll = layers[l];
                ChainLayer& cl = c.L[c.n_layers];
                used += chain_queue_weight(split, c.n_layers, Ll.weight, Ll.n_in, Ll.n_out, Ll.ldw, 1, h3.planes + used, cl);   // W^T read in place
                ++c.n_layers;
                if (l > 0) {
                    cl.act = layers[l - 1].act;
                    if (cl.act != PAPR_ACT_NONE) {
                        cl.mask = outs[l - 1]; cl.ld_mask = ld_out[l - 1];
                        // the forward run of layer l-1 left one bit per activation: read those instead of the fp32 rows
                        if (row_absmax && run_begin[l - 1] >= 0) cl.sign_bits = saved_sign_words(row_absmax, n_layers, M, l - 1);
                    }
                    cl.C = runs.g[l - 1 - (b > 0 ? b - 1 : 0)]; cl.ldc = G_LD;
                    cl.rowmax = runs.gmax[l - 1 - (b > 0 ? b - 1 : 0)];
                    cl.c_half = g_half && l > last ? 1 : 0;
                    bytes += (cl.sign_bits ? (cl.c_half ? 2LL : 4LL) * M * Ll.n_in + 32LL * M : cl.mask ? 8LL * M * Ll.n_in : 4LL * M * Ll.n_in);
                } else {
                    cl.C = d_x; cl.ldc = ldx; cl.act = PAPR_ACT_NONE;
                    bytes += 4LL * M * Ll.n_in;
                }
                bytes += 4LL * Ll.n_out * Ll.n_in;
                flops += 2LL * M * Ll.n_out * Ll.n_in;
            }
            PAPR_REQUIRE(used <= H3_PLANE_HALFS, "papr_mlp_bwd: fused run needs %zu plane halfs", used);
            if (c.n_layers > 0) {
                split.perm = 1;
                if (int err = chain_split_launch(split, c.n_layers, s)) return err;
                if (split_ahead) {                  // the top gradient rows split ahead of the run (chain.h: sr_*); their maxima go where the run would leave them
                    SRScratch sr(static_cast<char*>(workspace) + TN_SLAB_BYTES + H3Scratch::bytes(M) + BwdRunScratch::bytes(M), M);
                    SplitRowsArgs q = {};
                    q.x = c.A0; q.ld = c.lda0; q.K = c.K0; q.Kq = (c.K0 + 31) / 32 * 32; q.M = M;      // (the first k-loop reads whole pairs of k-steps: the planes are zero up to there)
                    q.hi = sr.hi; q.lo = sr.lo; q.inv = sr.inv; q.mx = c.rowmax0;
                    if (int err = launch_split_rows(q, s)) return err;
                    c.sr_hi = q.hi; c.sr_lo = q.lo; c.sr_inv = q.inv; c.sr_max = q.mx; c.sr_ld = q.Kq;
                }
                if (int err = papr_launch_chain(c, true, bytes, flops, s)) return err;
            }
            auto slot = [&](int l) { return l - (b > 0 ? b - 1 : 0); };
            // (the launch above leaves max |g| of the run's top rows in the last gmax slot, computed while staging them)
            if (c.n_layers == 0)
                if (int err = launch_row_absmax(g, M, L.n_out, ldg, reinterpret_cast<unsigned*>(runs.gmax[CHAIN_MAX_LAYERS]), s)) return err;
            for (int l = i; l >= b; --l) {
                const float* gl = l == i ? g : runs.g[slot(l)];
                long ldl = l == i ? ldg : G_LD;
                const float* const gm_top = top_f16 ? d_out_f16->max : runs.gmax[CHAIN_MAX_LAYERS];
                const float* gm = l == i ? gm_top : runs.gmax[slot(l)];
                // f16 operands: G_l -- the top rows' copy, or what chain layer l + 1 stored (it is not the launch's last: l >= last);
                // X_l -- the input copy of the forward run, or what forward layer l - 1 stored
                int gh = 0, xh = 0;
                const float* xin = nullptr; long ldxin = 0;
                if (top_f16 && l == i) { gl = reinterpret_cast<const float*>(d_out_f16->hi); ldl = d_out_f16->ld; gh = 1; }
                else if (g_half && l == i) { gl = reinterpret_cast<const float*>(c.a0_half); ldl = G_LD; gh = 1; }
                else if (g_half && l >= last) gh = 1;
                // (one-product mode: the f16 gradient rows of a run all carry the scale of the run's TOP rows -- their maxima stand for it, chain.h: c_half)
                if (gh && GEMM_ONE_PRODUCT) gm = gm_top;
                if (x_half && gh) {                  // (a layer whose gradient rows are fp32 -- layer 0 when the launch stops above it -- reads its fp32 input)
                    xh = 1;
                    if (l == b && !GEMM_ONE_PRODUCT) xh = 0;      // (PAPR_MLP_H3_F16ROWS: the run's input rows have no f16 copy -- G f16, X fp32, gemm_tn_h3_kernel<., 3>)
                    else if (l == b) { xin = reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(outs[b]) + (size_t)M * ld_out[b]); ldxin = ld_out[b]; }
                    else { xin = outs[l - 1]; ldxin = ld_out[l - 1]; }
                }
                if (int err = wgrad(l, gl, ldl, gm, gh, xh, xin, ldxin, (xh && GEMM_ONE_PRODUCT) ? row_absmax + (size_t)b * M : nullptr)) return err;
            }
            if (int err = tnq.flush()) return err;               // (the next run reuses the gradient-row slots)
            if (b == 0 && d_x && !to_dx) {                       // d_x accumulates (a skip layer wrote into it): separate launch
                PAPR_REQUIRE(layers[0].weight_t, "papr_mlp_bwd: layer 0 needs weight_t");
                NTArgs a = {};
                a.A = runs.g[slot(0)]; a.lda = G_LD; a.K1 = layers[0].n_out;
                a.W = layers[0].weight_t; a.ldw = layers[0].ldwt;
                a.dgrad = 1; a.accumulate = 1;
                a.C = d_x; a.ldc = ldx; a.M = M; a.N = layers[0].n_in;
                if (int err = gemm_nt(a, s)) return err;
            }
            if (b > 0) { g = runs.g[slot(b - 1)]; ldg = G_LD; gmax = reinterpret_cast<const unsigned*>(runs.gmax[slot(b - 1)]); }
            i = b;                                               // (the loop's --i moves below the run)
            continue;
        }
        // ---- single layer
        const bool h3w = GEMM_H3_WGRAD && row_absmax && layer_rowmax_saved(layers, n_layers, i) && L.n_out <= SLAB && L.n_in <= SLAB;
        if (h3w)
            if (int e = need_gmax(g, ldg, L.n_out)) return e;
        if (int e = wgrad(i, g, ldg, h3w ? reinterpret_cast<const float*>(gmax) : nullptr)) return e;
        if (int e = tnq.flush()) return e;                       // (the data-gradient below overwrites the other scratch buffer)
        // data gradients
        if (L.n_skip > 0 && d_x) {
            PAPR_REQUIRE(L.weight_t, "papr_mlp_bwd: layer %d needs weight_t", i);
            NTArgs a = {};
            a.A = g; a.lda = ldg; a.K1 = L.n_out;
            a.W = L.weight_t + (long)L.skip_col * L.ldwt; a.ldw = L.ldwt;
            a.dgrad = 1; a.accumulate = 1;
            a.C = d_x; a.ldc = ldx; a.M = M; a.N = L.n_skip;
            if (int e = gemm_nt(a, s)) return e;
        }
        if (i > 0) {
            PAPR_REQUIRE(L.weight_t, "papr_mlp_bwd: layer %d needs weight_t", i);
            float* gnext = (g == scratch0) ? scratch1 : scratch0;
            NTArgs a = {};
            a.A = g; a.lda = ldg; a.K1 = L.n_out;
            a.W = L.weight_t; a.ldw = L.ldwt;
            a.dgrad = 1; a.act = layers[i - 1].act; a.mask_src = outs[i - 1]; a.ld_mask = ld_out[i - 1];
            a.C = gnext; a.ldc = ld_scratch; a.M = M; a.N = L.n_in;
            PAPR_REQUIRE(ld_scratch >= L.n_in, "papr_mlp_bwd: scratch stride %d < %d", ld_scratch, L.n_in);
            if (GEMM_H3 && a.N > 128) {                          // split-f16 data-gradient: row maxima in, row maxima out
                if (int e = need_gmax(g, ldg, L.n_out)) return e;
                a.amax_in = gmax;
                a.amax_out = gmax == h3.in() ? h3.out() : h3.in();
                a.planes = h3.planes;
                PAPR_REQUIRE(hipMemsetAsync(a.amax_out, 0, (size_t)M * sizeof(unsigned), s) == hipSuccess, "papr_mlp_bwd: memset failed");
                if (int e = gemm_nt(a, s)) return e;
                if (a.amax_out == h3.out()) h3.swap();
                gmax = a.amax_out;
            } else {
                if (int e = gemm_nt(a, s)) return e;
                gmax = nullptr;
            }
            g = gnext;
            ldg = ld_scratch;
        } else if (d_x) {
            PAPR_REQUIRE(L.weight_t, "papr_mlp_bwd: layer 0 needs weight_t");
            NTArgs a = {};
            a.A = g; a.lda = ldg; a.K1 = L.n_out;
            a.W = L.weight_t; a.ldw = L.ldwt;
            a.dgrad = 1; a.accumulate = any_skip ? 1 : 0;
            a.C = d_x; a.ldc = ldx; a.M = M; a.N = L.n_in;
            if (int e = gemm_nt(a, s)) return e;
        }
    }
    return 0;
}
