// Launchers of the U-Net head's kernels with everything a whole-network caller can hand them (small_unet.hip): weights already split into their
// f16 planes, tensor maxima left by the producing kernel instead of taken by an extra launch, outputs written straight into a channel slice of a
// wider map (the skip concatenation), ReLU masks and gradient sums folded into the kernels at the seams.  The single-layer entry points of
// include/papr_hip.h (papr_conv3x3_fwd, ...) run the same kernels through these.
#pragma once
#include "papr_common.h"

struct PaprConvLaunch {                      // one 3x3 layer (or its data-gradient: planes split with flip)
    const float* x; int B, H, W, c_in;       // input map (B H W, c_in), contiguous
    const _Float16* w_hi; const _Float16* w_lo;      // planes [c_out padded to 128][9 c_in]
    const float* bias; int c_out, relu;
    float* out; int ldo;                     // output rows: out + m * ldo (ldo >= c_out: a slice of a wider map)
    const unsigned* xmax; int n_xmax;        // max |x|: one word (n_xmax 1: the single-layer entry points) or a slot (PAPR_SLOT_W, papr_common.h) ...
    const unsigned* xmax2;                   // ... and a second slot (or null): a map concatenated from two producers
    unsigned* out_max;                       // or null: the slot that receives max |out| (zeroed by the caller)
    float* partial;                          // papr_i_conv_splits() > 1: splits x (B H W) x c_out floats
    bool one_product;                        // one f16 product per fp32 product (the arithmetic of the reference's fp16 autocast) instead of three
};
int papr_i_conv_splits(long M, int c_in, int c_out);
int papr_i_conv3x3(const PaprConvLaunch& c, hipStream_t s);

struct PaprSplitJob { const float* w; int N, C; long sn, sc, sky, skx; int flip; _Float16* hi; _Float16* lo; };
constexpr int PAPR_UNET_MAX_JOBS = 10, PAPR_UNET_IN_PARTS = 64;
// one launch: in_partial[0 .. 64) = partial maxima of |x| (n4 float4; the first words of x's slot), the n_slots slots at `slots` zeroed, every job's
// weight -> its planes
int papr_i_unet_prep(const float* x, long n4, unsigned* in_partial, unsigned* slots, int n_slots, const PaprSplitJob* jobs, int n_jobs, hipStream_t s);

size_t papr_i_conv3x3_wgrad_partial_bytes(long M, int c_in, int c_out);
// d_w (c_out, 3, 3, c_in), d_b (c_out) or null; maxima: the producers' slots (xmax2: see above, or null)
int papr_i_conv3x3_wgrad(const float* d_y, const float* x, int B, int H, int W, int c_in, int c_out, float* d_w, float* d_b, const unsigned* dymax,
                         const unsigned* xmax, const unsigned* xmax2, float* partial, bool one_product, hipStream_t s);

// ---- unet.hip
int papr_i_maxpool2_fwd(const float* x, int ld_in, int B, int H, int W, int C, float* out, unsigned* which, hipStream_t s);
// d_in (B H W, C) = (pool-backward(d_out, which) + skip[m * ld_skip + c]) * (y[m * ld_y + c] > 0);  out_max: atomicMax of max |d_in|
int papr_i_maxpool2_bwd_fused(const float* d_out, const unsigned* which, int B, int H, int W, int C, const float* skip, int ld_skip, const float* y, int ld_y,
                              float* d_in, unsigned* out_max, hipStream_t s);
int papr_i_upconv_fwd(const float* x, int B, int H, int W, int c_in, const float* wm, const float* bias, int c_out, float* out, int ldo, unsigned* out_max,
                      bool one_product, hipStream_t s);
// d_x (B H W, c_in) = dgrad(g rows at g + pixel * ldg) * (mask_y > 0)   (mask_y (B H W, c_in) or null)
int papr_i_upconv_dgrad(const float* g, int ldg, int B, int H, int W, int c_in, const float* wm, int c_out, const float* mask_y, float* d_x, unsigned* out_max,
                        bool one_product, hipStream_t s);
size_t papr_i_upconv_wgrad_bytes(long M, int c_in, int c_out);
int papr_i_upconv_wgrad(const float* g, int ldg, const float* x, int B, int H, int W, int c_in, int c_out, const unsigned* xmax, const unsigned* gmax, float* d_wm,
                        float* d_bias, void* ws, bool one_product, hipStream_t s);
// the 1x1 head backwards: d_x = (d_out w) * (mask_y > 0) with its maximum; d_w, d_b as papr_conv1x1_bwd
int papr_i_conv1x1_bwd(const float* d_out, const float* x, long M, int c_in, const float* w, int c_out, const float* mask_y, float* d_x, unsigned* out_max,
                       float* d_w, float* d_b, void* ws, hipStream_t s);
