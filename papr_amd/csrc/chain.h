// Host-side interface of the fused layer-run kernel (chain4.hip), used by the drivers in gemm.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

constexpr int CHAIN_MAX_LAYERS = 8;
constexpr int CHAIN_SIGN_WORDS = 8;        // sign words per row and layer (one bit per value of a 256-wide row)

struct ChainLayer {
    const _Float16* w_hi;      // weight planes in MFMA fragment order (split_weight_kernel): n_tiles x ksteps fragments
    const _Float16* w_lo;
    const float* bias;         // forward: (N) or null
    const float* mask;         // data-gradient: activation outputs whose derivative multiplies this layer's result, or null
    long ld_mask;
    unsigned* sign_bits;       // chain_sign_rows(M) x CHAIN_SIGN_WORDS words: forward = written (bit set where the result is > 0), data-gradient = read in place of
                               // mask; or null.  One word per lane and 64-row tile: tile t owns words [512 t, 512 (t + 1)), wave w of the tile words 64 w .. 64 w + 63;
                               // bit 31 - (16 i + e) of lane (row & 31, h)'s word = column 32 w + 16 h + e of row 32 i + (row & 31) of the tile.
                               // ONE-PRODUCT mode (round 6: the bits come off packed f16 pairs, one instruction per pair): value e = 8 q + 2 j + b of the
                               // lane's sixteen (q: 16-byte chunk, j: pair, b: half) of row tile i sits at bit (15 - (4 (2 q + i) + j)) + 16 b
    float* C;                  // (M, ldc) result rows, or null when nobody needs them in memory
    long ldc;
    float* rowmax;             // (M) max |.| of every result row, or null
    int N;                     // result width (<= 256; a multiple of 32 unless this is the last layer of the run)
    int ksteps;                // input width / 16, planes padded to a multiple of 32 columns (even)
    int k1steps;               // = ksteps, or for a skip layer ([previous output | x] as input): the k-steps of the first
                               // segment (a multiple of 4); the remaining ones multiply the run's input rows A0 again
    int act;                   // forward: activation; data-gradient: activation whose derivative is applied
    int c_half;                // 1 (not the run's last layer): C receives f16 rows instead of fp32 ones (ldc counts halfs) -- the hi plane the next
                               // layer multiplies.  Parity arithmetic (PAPR_MLP_H3_F16ROWS): each row times the power of two that brings ITS maximum
                               // (rowmax, required) into [2^13, 2^14).  One-product mode: each row times the RUN's scale of that row (h3_common.h:
                               // one_scale_from_max of the run's input row / top gradient row; rowmax is not written).  The weight-gradient kernel
                               // reads them back with that scale (gemm.hip: TNH3Args::g_rs / x_rs)
};

struct ChainArgs {
    float* A0; long lda0; int K0;         // input rows (M, lda0), K0 <= 256 real columns (written only with in_norm_writeback)
    float* rowmax0;                       // (M) receives max |.| of every input row, or null
    _Float16* a0_half; long lda0_half;    // (one_product) copy of the staged input rows as scaled f16 rows (see c_half), or null
    // The input rows ALREADY split (papr_split_rows_launch, gemm.hip): hi / lo planes (M, sr_ld) halfs each -- row m times the power of two
    // that brings its maximum into [2^13, 2^14), columns from K0 on zero --, 1 / that power (sr_inv) and the maximum itself (sr_max), (M) floats
    // each.  With them a tile is staged by LDS-DMA (no registers, no vector instructions, the rows' way from memory hidden behind the slot's
    // matrix instructions) and rowmax0 / in_norm_* are the splitting kernel's business, not the run's.  Null: the run splits A0 itself.
    const _Float16* sr_hi; const _Float16* sr_lo; const float* sr_inv; const float* sr_max; int sr_ld;
    long M;
    int n_layers;
    int one_product;                      // 1 (PAPR_GEMM_MODE=h1): one f16 product per fp32 product -- hi planes only
    int in_norm_width;                    // forward: with in_norm_stats (+ in_norm_mean: given), the input rows are standardised over their first
    float in_norm_eps;                    // in_norm_width columns while they are staged (LayerNorm core in front of the run);
    float* in_norm_stats;                 // (M, 2) = 1/(std+eps), std; in_norm_writeback: the standardised rows replace A0 in
    int in_norm_writeback;                // memory as well (training: the backward pass reads them; runs with a skip layer)
    const float* in_norm_mean;            // with in_norm_stats (required then): the rows' means; in_norm_stats holds 1/(std+eps), std ALREADY -- the staging applies them
    float norm_eps;                       // forward: with norm_stats, the LAST layer's rows are standardised (LayerNorm core,
    float* norm_stats;                    // papr_row_norm in papr_hip.h) before they are stored; (M, 2) = 1/(std+eps), std
    const float* dot_rows; long ld_dot;   // with norm_stats and dots: dots[m] = (standardised row m) . dot_rows[m / rows_per_dot] -- the
    int rows_per_dot;                     // attention scores' dot products straight from the last row phase (the last layer's C may then
    float* dots;                          // be null: inference never writes the key embedding); (M) floats
    float* norm_mean;                     // with dots: the last layer's rows are stored RAW and their means go here (M) -- the consumer standardises
                                          // on the fly (papr_row_norm.raw_mean); the last row phase then takes no statistics at all
    ChainLayer L[CHAIN_MAX_LAYERS];
};

size_t papr_chain4_lds_bytes();
// rows of a layer's sign-word area: M rounded up to whole 64-row tiles (the area is CHAIN_SIGN_WORDS words per row)
inline long chain_sign_rows(long M) { return (M + 63) / 64 * 64; }
// bytes / flops: algorithmic totals of the launch for the profiling record.  The weight fragments must be laid out by
// split_weight_batch_kernel with perm = 1 (gemm.hip).
int papr_launch_chain(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s);
