// Host-side interface of the fused layer-run kernel (chain.hip), used by the drivers in gemm.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

constexpr int CHAIN_MAX_LAYERS = 8;
constexpr int CHAIN_SIGN_WORDS = 16;       // sign words per row and layer (one per wave and half-wave: 8 or 16 are used)

struct ChainLayer {
    const _Float16* w_hi;      // weight planes in MFMA fragment order (split_weight_kernel): n_tiles x ksteps fragments
    const _Float16* w_lo;
    const float* bias;         // forward: (N) or null
    const float* mask;         // data-gradient: activation outputs whose derivative multiplies this layer's result, or null
    long ld_mask;
    unsigned* sign_bits;       // (CHAIN_SIGN_WORDS, M) words: forward = written (bit set where the result is > 0), data-gradient = read in place of mask; or null.
                               // word [(2 wn + h) * M + m] holds the 32 columns 64 wn + 32 j + 8 g + 4 h + c of row m, (j, g, c) = 0 first, in the top bit
    float* C;                  // (M, ldc) result rows, or null when nobody needs them in memory
    long ldc;
    float* rowmax;             // (M) max |.| of every result row, or null
    int N;                     // result width (<= 256; a multiple of 32 unless this is the last layer of the run)
    int ksteps;                // input width / 16, planes padded to a multiple of 32 columns (even)
    int k1steps;               // = ksteps, or for a skip layer ([previous output | x] as input): the k-steps of the first
                               // segment (a multiple of 4); the remaining ones multiply the run's input rows A0 again
    int act;                   // forward: activation; data-gradient: activation whose derivative is applied
    int c_half;                // 1 (chain3.hip, one_product, not the run's last layer): C receives f16 rows instead of fp32 ones (ldc counts
                               // halfs) -- the hi plane the next layer multiplies, i.e. each row times the power of two that brings its
                               // maximum (rowmax, required) into [2^13, 2^14); the weight-gradient kernel reads them back with that scale
};

struct ChainArgs {
    float* A0; long lda0; int K0;         // input rows (M, lda0), K0 <= 256 real columns (written only with in_norm_writeback)
    float* rowmax0;                       // (M) receives max |.| of every input row, or null
    _Float16* a0_half; long lda0_half;    // (chain3.hip, one_product) copy of the staged input rows as scaled f16 rows (see c_half), or null
    long M;
    int n_layers;
    int one_product;                      // 1 (PAPR_GEMM_MODE=h1, chain3.hip only): one f16 product per fp32 product -- hi planes only
    int legacy;                           // 1: this MLP has a skip layer somewhere -- all its runs (forward and data-gradient) use chain.hip,
                                          // whose sign-word layout differs from chain2.hip's
    int in_norm_width;                    // forward: with in_norm_stats, the input rows are standardised over their first
    float in_norm_eps;                    // in_norm_width columns while they are staged (LayerNorm core in front of the run);
    float* in_norm_stats;                 // (M, 2) = 1/(std+eps), std; in_norm_writeback: the standardised rows replace A0 in
    int in_norm_writeback;                // memory as well (training: the backward pass reads them; runs with a skip layer)
    float norm_eps;                       // forward: with norm_stats, the LAST layer's rows are standardised (LayerNorm core,
    float* norm_stats;                    // papr_row_norm in papr_hip.h) before they are stored; (M, 2) = 1/(std+eps), std
    ChainLayer L[CHAIN_MAX_LAYERS];
};

// Sign words: chain.hip: word [(2 wn + h) * M + m] as described at ChainLayer::sign_bits; chain2.hip: the RB-row block b
// (RB = 8 or 16 rows per wave, chain2.hip) owns words [8 RB b, 8 RB (b + 1)): lane l of the wave that wrote it keeps word 8 RB b +
// (RB / 8) l (rows 0-7; + 1: rows 8-15), 4 bits per row (columns 4 l .. 4 l + 3), first value in the top bit.
size_t papr_chain_lds_bytes();
size_t papr_chain2_lds_bytes();
int papr_launch_chain2(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s);
// chain3.hip: chain2.hip's layout at RB = 8; only blocks with a row inside M are written or read.
size_t papr_chain3_lds_bytes();
int papr_launch_chain3(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s);
// chain4.hip: one word per lane and 64-row tile: tile t owns words [512 t, 512 (t + 1)), wave w of the tile words 64 w .. 64 w + 63; bit 31 - (16 i + e)
// of lane (row & 31, h)'s word = column 32 w + 16 h + e of row 32 i + (row & 31).  Only tiles with a row inside M are written or read.
size_t papr_chain4_lds_bytes();
int papr_launch_chain4(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s);
// which kernel carries a run (PAPR_CHAIN: 1 chain.hip, 2 chain2.hip, 3 chain3.hip, 4 = default chain4.hip); chain4.hip wants its weight
// fragments with the column permutation of split_weight_batch_kernel(perm = 1)
int papr_chain_version(const ChainArgs& a);
// rows of a layer's sign-word area: M rounded up to whole 64-row tiles (the area is CHAIN_SIGN_WORDS words per row)
inline long chain_sign_rows(long M) { return (M + 63) / 64 * 64; }
// bytes / flops: algorithmic totals of the launch for the profiling record
int papr_launch_chain(const ChainArgs& a, bool dgrad, long long bytes, long long flops, hipStream_t s);
