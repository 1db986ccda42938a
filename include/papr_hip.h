/*
 * papr_hip.h -- C ABI of libpapr_hip.so: the MI355X (gfx950) PAPR per-ray render path.
 *
 * The reference (zvict/papr) has no FFI: its hot path is eager PyTorch inside
 * models/model.py, models/attn.py and models/mlp.py.  Each entry point below replaces the
 * group of reference lines it cites; INTEGRATION.md shows the ctypes binding that a
 * maintainer of the reference would add to call them.
 *
 * Conventions
 *   - plain device pointers + sizes, fp32 row-major, int32 indices; no torch / C++ types.
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     immediately; 0 = ok, non-zero = error (text via papr_last_error()).
 *   - no internal allocation, no settings kept between calls: scratch is passed in by the caller (the
 *     *_workspace_bytes helpers say how much), modes are arguments, and the library reads NO environment variable.
 *     Process-wide by design, and only through explicit calls: the optional profiler switch (papr_profile_enable),
 *     the last-error text, and the A/B switches of papr_set_switch (defaults = the product; none changes a result).
 *     Device facts (compute-unit count, kernel attributes) are cached per device id.
 *   - "ld" arguments are row strides in floats and must be multiples of 4 (16-byte rows).
 */
#ifndef PAPR_HIP_H
#define PAPR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* papr_stream_t;

enum { PAPR_ACT_NONE = 0, PAPR_ACT_RELU = 1, PAPR_ACT_LEAKY_RELU = 2 /* slope 0.2 */ };

int papr_abi_version(void);
const char* papr_last_error(void);

/* ------------------------------------------------------------------------------------
 * K1  ray -> k nearest points           replaces PAPR._calculate_global_distances
 *                                        (models/model.py:258-283): R x P distance tensors + topk.
 * points (P,3); rays_o (N,3); rays_d (R,3) with R = N * rays_per_image; ray r belongs to image
 * r / rays_per_image.  Distance = | v - d (v.d)/(d.d+eps) |, v = p - o, d used as given.
 * out_idx (R,k) int32, ascending in (distance, point index); out_dist (R,k) or NULL.  When several
 * points tie exactly at the k-th distance (the reference's topk(sorted=False) is unordered there) the ones with
 * the smallest indices are kept: the set is the k smallest by (distance, index), whichever form runs -- clouds of
 * 2,048 points and more are binned per call (bounding spheres of 64-point blocks, k < 64), smaller ones are
 * searched point by point.
 * Requires 1 <= k <= 256 and k <= P (k <= 63: the tuned forms -- seeded every-point search, binned cloud; 64 <= k <= 256, ABI 25: one ray per wave,
 * the set across several registers per lane, exact in the same total order).  workspace: papr_ray_knn_workspace_bytes(R, P) bytes, contents irrelevant
 * on entry (ray records, the binned copy of the cloud, block bounds, cell counters: all rebuilt by every call).
 * Points whose coordinates are NaN never enter a set; every out_idx entry is an index in [0, P) all the same (a cloud with fewer than k
 * finite points fills the rest with point 0 -- torch.topk of NaN distances returns valid indices too; ABI 27, end of round 6).
 */
size_t papr_ray_knn_workspace_bytes(int64_t R, int64_t P);
int papr_ray_knn(const float* points, int64_t P, const float* rays_o, const float* rays_d, int64_t R,
                 int64_t rays_per_image, int k, float eps, int32_t* out_idx, float* out_dist,
                 void* workspace, papr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K2  gather + ray geometry + positional encoding
 *     replaces points[idx] / pc_feats[idx] (models/model.py:330,431-435), _calculate_distances
 *     (models/model.py:285-310), _get_kqv (:396-437), posenc (models/utils.py:232-242) and the
 *     torch.cat in Embeddings.forward (models/attn.py:173-191).
 * Row layouts (raw, before any LayerNorm), M = R*k rows for key/value, R rows for the query:
 *   key  = [pe(p, L_key[0]), pe(s, L_key[1]), pe(u, L_key[2]), feats if key_has_feats]
 *   qry  = [pe(d, L_qry)]
 *   val  = [pe(s, L_val[0]), pe(u, L_val[1]), feats if val_has_feats]
 *   pe(x)[c*(with_self+2L) + ...] = x_c, sin(f^0 m x_c), cos(f^0 m x_c), sin(f^1 m x_c), ...
 * Padding columns up to ld_* are written as zero.
 */
typedef struct {
    int32_t k;              /* neighbours per ray */
    int32_t feat_dim;       /* width of pc_feats */
    int32_t L_key[3];
    int32_t L_qry;
    int32_t L_val[2];
    int32_t with_self;      /* embed_type 1 -> 1, embed_type 2 -> 0 */
    int32_t key_has_feats;  /* geoms.point_feats.use_ink */
    int32_t val_has_feats;  /* geoms.point_feats.use_inv */
    float pe_factor, pe_mult, eps;
    int32_t ld_key, ld_qry, ld_val;
} papr_feature_desc;

int papr_feature_widths(const papr_feature_desc* d, int32_t* key_w, int32_t* qry_w, int32_t* val_w);

int papr_build_features_fwd(const papr_feature_desc* d, const float* points, const float* pc_feats,
                            const float* rays_o, const float* rays_d, int64_t R, int64_t rays_per_image,
                            const int32_t* idx, float* key, float* qry, float* val,
                            float* sel_points /* (R,k,3) or NULL */,
                            float* key_stats /* (R*k, 2) or NULL */, float* key_mean /* (R*k) or NULL */, float key_norm_eps,
                            papr_stream_t stream);
/* key_stats / key_mean (ABI 24; both or neither; not with key_has_feats): the statistics of the LayerNorm core in front of the key MLP
 * (FeedForward.innorm, models/attn.py:39-42) for every key row -- key_stats[2m], [2m+1] = 1 / (std_unbiased + key_norm_eps), std; key_mean[m] --
 * taken while the row is written (the key rows themselves stay raw).  papr_mlp_fwd applies them through papr_row_norm.given_mean. */

/* Backward of K2: d_key / d_val are gradients w.r.t. the raw rows above.  Accumulates (atomic
 * adds) into d_points (P,3) and d_pc_feats (P,feat_dim); the caller zeroes them.  The key's pe(p)
 * block receives no gradient (points.detach(), models/model.py:405). */
int papr_build_features_bwd(const papr_feature_desc* d, const float* points, const float* rays_o,
                            const float* rays_d, int64_t R, int64_t rays_per_image, const int32_t* idx,
                            const float* d_key, const float* d_val, float* d_points, float* d_pc_feats,
                            papr_stream_t stream);

/* Same backward, but instead of atomics it writes one gradient row {dx,dy,dz,0} per pair into
 * d_pair_points (R*k,4); the per-point feature columns stay in d_val / d_key.  Feed both to
 * papr_segment_reduce for a balanced, atomic-free scatter (replaces the
 * index_put_(accumulate=True) backward of the three gathers, models/model.py:330,435,509). */
int papr_build_features_bwd_pairs(const papr_feature_desc* d, const float* points, const float* rays_o,
                                  const float* rays_d, int64_t R, int64_t rays_per_image, const int32_t* idx,
                                  const float* d_key, const float* d_val, float* d_pair_points,
                                  const float* key_mean, const float* key_stats, papr_stream_t stream);
/* key_mean / key_stats (ABI 25; both or neither; what papr_build_features_fwd wrote; not with key_has_feats): d_key is the gradient w.r.t. the
 * STANDARDISED key rows (the LayerNorm core in front of the key MLP) and its backward pass rides in this kernel -- the standardised values are
 * recomputed from the encodings, no papr_rownorm_bwd pass over the (R*k, 117) rows. */

/* order (M = R*k) int64: pair ids grouped by selected point (a stable sort of idx); sorted_pts (M) int32:
 * the point of each entry; seg (P+1) int64: group bounds.  d_points[p] += sum of pair_points rows,
 * d_influ[p] += sum of pair_influ, d_feats[p][c] += sum of rows[pair][col0 + c] (c < ncols <= 128) over
 * point p's group.  Any of the three inputs may be NULL.  accumulate = 0: the outputs must be zeroed by the
 * caller (points nobody selected are not written); accumulate = 1: every sum is ADDED to what the outputs hold
 * (a second pass: point features that feed both the key and the value branch, use_ink + use_inv).
 * No atomics (ABI 18): a group inside one 128-entry chunk is summed there, a group that spans several chunks from
 * the chunks' shares in chunk order -- the same bits on every run.  workspace: papr_segment_reduce_workspace_bytes(M). */
/* ------------------------------------------------------------------------------------
 * K6  point -> k nearest points of the cloud      replaces the two scipy KDTree queries of add_points_knn
 *                                                  (models/utils.py:27-29 `tree.query(points, k=sample_k)`, :59 `tree.query(query, k=k+1)`).
 * points (P,3); query_idx (Q) int32 indices into points, or NULL for Q = P "every point".  nn_idx (Q,k) int32 and nn_dist
 * (Q,k) double (or NULL): Euclidean distances computed in double from the float32 coordinates like the KDTree does, ascending
 * in (distance, point index) -- entry 0 is the query point itself (distance 0) unless the cloud holds a duplicate with a lower
 * index.  Requires 1 <= k <= 16 and k <= P.
 */
int papr_points_knn(const float* points, int64_t P, const int32_t* query_idx, int64_t Q, int32_t k, int32_t* nn_idx,
                    double* nn_dist, papr_stream_t stream);

/* The grouping papr_segment_reduce consumes, from idx (M = R*k selected point per pair, 0 <= idx < P): order = the stable sort
 * permutation of idx (pair ids ascending inside a group: torch.sort(idx, stable=True).indices), sorted_pts = idx[order],
 * seg[p] = first entry of point p's group, seg[P] = M.  A stable counting sort (ABI 20; before: rocPRIM's radix sort).
 * run >= 1: the caller's promise that every aligned run of `run` consecutive entries of idx holds DISTINCT points -- a ray's k
 * neighbours (run = k), or run = 1 for no promise at all (one entry per step: slow).  A broken promise still yields a valid grouping,
 * but the order inside a group is then no longer defined.  workspace: papr_group_pairs_workspace_bytes(M, P) bytes of device memory. */
size_t papr_group_pairs_workspace_bytes(int64_t M, int64_t P);
int papr_group_pairs(const int32_t* idx, int64_t M, int64_t P, int32_t run, int64_t* order, int32_t* sorted_pts, int64_t* seg,
                     void* workspace, size_t workspace_bytes, papr_stream_t stream);

size_t papr_segment_reduce_workspace_bytes(int64_t M);
int papr_segment_reduce(const int64_t* order, const int32_t* sorted_pts, const int64_t* seg, int64_t M, int64_t P,
                        const float* pair_points, const float* pair_influ, const float* rows, int ld, int col0,
                        int ncols, float* d_points, float* d_influ, float* d_feats, int accumulate, void* workspace,
                        papr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Row standardisation  y = (x - mean) / (std_unbiased + eps)      (the non-affine core of the
 * reference LayerNorm, models/attn.py:39-42; the a_2/b_2 affine is folded into the next Linear by
 * the host).  stats (rows,2) receives {1/(std+eps), std}.  x and y may alias.  width <= 1024.
 */
int papr_rownorm_fwd(const float* x, int64_t rows, int width, int ld, float eps, float* y, float* stats,
                     papr_stream_t stream);
int papr_rownorm_bwd(const float* dy, const float* y, const float* stats, int64_t rows, int width, int ld,
                     float eps, float* dx, papr_stream_t stream);
/* dots[m] = rows[m] . dot_rows[m / rows_per_dot] over `width` columns, m < M: the stand-alone form of papr_row_norm.dots (what
 * papr_mlp_fwd runs when the last layer is not inside a fused run). */
int papr_row_dots(const float* rows, int64_t M, int width, int ld, const float* dot_rows, int ld_dot, int rows_per_dot,
                  float* dots, papr_stream_t stream);

/* Backward of the score bias c0 = q'.b_k, q' = W_q Q + b_q (the bias row of w_k inside the reference's R k x d_model score product,
 * models/attn.py:217-225; here one number per ray, papr_row_dots) -- linear in Q: c0 = Q.(W_q^T b_k) + b_q.b_k.  Given d_c0 (R), with
 * u = Q^T d_c0 and s = sum(d_c0):
 *   d_Q (R, ldq)[:, :dq] += d_c0 (x) (W_q^T b_k);   d_wq (dm, ldwq)[:, :dq] += b_k (x) u;   d_bq (dm) = d_bq_in + s b_k (may alias d_bq_in);
 *   d_bk (dm) = W_q u + s b_q.
 * Q (R, ldq), wq = W_q (dm, ldwq), bk, bq (dm); ldq and ldwq multiples of 4.  Sums in a fixed order; two launches (ABI 26). */
size_t papr_qk_bias_bwd_workspace_bytes(int dq);
int papr_qk_bias_bwd(const float* Q, int ldq, int dq, int dm, const float* d_c0, int64_t R, const float* wq, int ldwq, const float* bk, const float* bq,
                     float* d_Q, float* d_wq, const float* d_bq_in, float* d_bq, float* d_bk, void* workspace, papr_stream_t stream);

/* The `mse` loss term (torch.nn.MSELoss() in the reference's BasicLoss, models/__init__.py:8-52): *loss = mean((pred - target)^2) over n numbers and,
 * with grad != NULL, grad[i] = 2 (pred[i] - target[i]) / n -- one kernel launch (ABI 26).  workspace: papr_mse_workspace_bytes() bytes; its ticket counter is
 * zeroed in-stream in front of every launch (ABI 27: a 4-byte memset -- an aborted launch can no longer poison later calls); calls sharing one workspace
 * must run on one stream. */
size_t papr_mse_workspace_bytes(void);
int papr_mse_fwd(const float* pred, const float* target, int64_t n, float* loss, float* grad, void* workspace, papr_stream_t stream);

/* The affine part of that LayerNorm, y = a_2 * xh + b_2 (models/attn.py:42), folded into the Linear layer behind it:
 *   W (a_2 * xh + b_2) + c  =  (W * a_2) xh + (W b_2 + c).
 * fwd: eff_w (n_out, ld_eff) = W[:, :n_in] * a_2 with columns n_in .. ld_eff-1 zeroed, eff_b = c + W b_2.
 * bwd: d_w = d_eff_w * a_2 + d_eff_b (x) b_2,  d_a2 = column sums of d_eff_w * W,  d_b2 = W^T d_eff_b
 * (d_c = d_eff_b: the caller's).  Sums run in a fixed order.  n_in, ld_eff <= 1024. */
int papr_ln_fold_fwd(const float* w, int32_t n_out, int32_t n_in, int32_t ldw, const float* c, const float* a2,
                     const float* b2, float* eff_w, int32_t ld_eff, float* eff_b, papr_stream_t stream);
int papr_ln_fold_bwd(const float* w, int32_t n_out, int32_t n_in, int32_t ldw, const float* a2, const float* b2,
                     const float* d_eff_w, int32_t ld_eff, const float* d_eff_b, float* d_w, float* d_a2, float* d_b2,
                     papr_stream_t stream);
/* The same for several layers in one launch each way (ABI 26): a PAPR step folds four affines (in front of the key and the query MLP, behind them
 * into w_k and w_q).  Forward reads w, c (or NULL), a2, b2 and writes eff_w, eff_b; backward reads w, a2, b2, d_eff_w, d_eff_b and writes d_w, d_a2,
 * d_b2 (the gradient of c is d_eff_b itself); the fields of the other direction are ignored. */
#define PAPR_LN_FOLD_MAX_JOBS 8
typedef struct {
    const float* w; int32_t n_out, n_in, ldw;
    const float* c; const float* a2; const float* b2;
    float* eff_w; int32_t ld_eff; float* eff_b;
    const float* d_eff_w; const float* d_eff_b; float* d_w; float* d_a2; float* d_b2;
} papr_ln_fold_job;
int papr_ln_fold_fwd_batch(const papr_ln_fold_job* jobs, int32_t n, papr_stream_t stream);
int papr_ln_fold_bwd_batch(const papr_ln_fold_job* jobs, int32_t n, papr_stream_t stream);


/* ------------------------------------------------------------------------------------
 * K3  embedding MLP chain on MFMA      replaces MLP.forward (models/mlp.py:47-59) and its autograd.
 * Layer i computes out_i = act_i(in_i W_i^T + b_i), in_0 = x, in_i = out_{i-1}; a layer with
 * n_skip > 0 consumes [in_i, x[:, :n_skip]] (skip_layers re-concatenation, models/mlp.py:54-55) and
 * its weight holds the n_in columns first and the skip columns from column `skip_col` on.
 * Weights are (n_out, ldw) row-major with zero padding; n_in and n_skip are padded to multiples of 4.
 */
typedef struct {
    const float* weight;    /* (n_out, ldw) */
    const float* weight_t;  /* (n_in_total, ldwt) transpose, needed by papr_mlp_bwd only */
    const float* bias;      /* (n_out) or NULL */
    int32_t n_in, n_out, ldw, ldwt;
    int32_t n_skip, skip_col;
    int32_t act;
} papr_layer;

/* outs[i] : (M, ld_out[i]) buffer of layer i's output.  For training pass distinct buffers (they are
 * the saved activations); for inference two ping-pong buffers may be reused.
 * workspace: papr_mlp_fwd_workspace_bytes(M) bytes (per-row operand scales and the pre-split weight of
 * the split-f16 GEMM; see gemm.hip).  Arithmetic: fp32 in, fp32 out; wide layers multiply on the f16
 * matrix pipe with every fp32 operand split into two halves (22 mantissa bits, fp32 accumulation) unless
 * `mode` says otherwise (PAPR_MLP_* below).
 * row_absmax: NULL (inference: nothing is kept for a backward pass, and only outs[n_layers-1] is defined
 * afterwards), or papr_mlp_saved_floats(n_layers, M) floats of state for papr_mlp_bwd: max_k |input row m
 * of layer i| at [i*M + m] for every layer that ran on a split-f16 kernel (other entries undefined),
 * followed by one sign bit per activation of the layers a fused run produced.  Pass the same buffer to
 * papr_mlp_bwd: weight-gradients then run with split-f16 operands scaled by the maxima (fp32 MFMA
 * without it), and the data-gradient run reads the sign bits instead of the fp32 activations. */
size_t papr_mlp_saved_floats(int32_t n_layers, int64_t M);
/* Optional LayerNorm core behind the last layer (the reference's FeedForward.outnorm, models/attn.py:113-117, without
 * its affine part, which the host folds into the next Linear): outs[n_layers-1] receives the standardised rows
 * y = (x - mean) / (std_unbiased + eps) and stats[2m], stats[2m+1] = 1 / (std + eps), std -- the layout papr_rownorm_fwd
 * writes and papr_rownorm_bwd reads.  A fused run applies it in its last row phase (no extra pass over the rows);
 * otherwise the library runs papr_rownorm_fwd in place after the last layer. */
typedef struct {
    float eps;
    int32_t width;      /* logical row width (= n_out of the last layer) */
    float* stats;       /* (M, 2) */
    /* out_norm only, optional (dots != NULL; ABI 17): dots[m] = y_m . dot_rows[m / rows_per_dot] over `width` columns -- the attention
     * scores' dot products (models/attn.py:219-221, with dot_rows = W_k^T (W_q Q + b_q) of the ray that owns row m) taken where the
     * standardised key rows are produced.  In inference (row_absmax == NULL) outs[n_layers-1] is then UNDEFINED afterwards: a fused run
     * never writes the (R*k, d_model) key embedding (papr_attn_tail_fwd: precomputed_dots).  Ignored for in_norm.  A fused run takes them as
     * (x_m . g - mean_m sum(g)) / (std_m + eps) on the un-standardised row x_m (since round 4: the same value up to rounding, the same bits in
     * training and inference); papr_row_dots multiplies the standardised rows. */
    const float* dot_rows;  /* (ceil(M / rows_per_dot), ld_dot) */
    int32_t ld_dot;         /* floats, a multiple of 4 */
    int32_t rows_per_dot;
    float* dots;            /* (M) */
    /* in_norm only, optional (ABI 24): the rows' means, with stats ALREADY holding 1 / (std + eps), std (papr_build_features_fwd: key_mean /
     * key_stats): the rows are standardised with them -- x <- (x - mean) / (std + eps) over `width` columns -- and no statistic is computed or
     * written.  NULL: the call computes the statistics itself. */
    const float* given_mean;
    /* out_norm only, optional (ABI 25; with dots and row_absmax, i.e. a training call whose last layer rides in a fused run -- anything else is an
     * error, not a fallback): outs[n_layers-1] receives the UN-standardised rows x and raw_mean[m] their means; stats and dots as above.  The
     * consumer standardises on the fly, y = (x - raw_mean) * stats[2m] -- the two instructions the run would have spent per element, bit for bit
     * (papr_attn_tail_bwd: kp_mean).  Why: the run's last row phase then needs no row statistics at all (they were 10-16k cycles per wave of a
     * 40k-cycle slot: every lane took them from eight partial tables), only the dot products' finish does. */
    float* raw_mean;
    /* in_norm only (ABI 27): 1 = the caller never reads x again (its own backward pass of the norm needs neither x nor the standardised rows -- e.g.
     * papr_build_features_bwd_pairs with key_mean / key_stats): the call MAY leave x as it is instead of overwriting it with the standardised rows.
     * It does where its own backward pass does not read them either (PAPR_MLP_H1 training runs: the weight gradient reads the run's f16 copy) --
     * 246 MB of stores per 512,000 key rows. */
    int32_t leave_input;
} papr_row_norm;

/* Rows in the fused runs' input format (ABI 27): what a run's staging makes of fp32 rows, written by the producer instead.
 *   lo == NULL, the one-product runs': hi (M, ld) halfs = row x scale, scale (M) = the power of two that brings the row's maximum into [2^3, 2^4) --
 *     data-gradient rows: exponent clamped at 2^-40 --, inv (M) = 1 / scale, max (M) = max |row|;
 *   lo != NULL, the parity runs' (split-f16): scale = the power of two that brings the maximum into [2^13, 2^14) (a zero row: 1), hi = f16(row x scale),
 *     lo (M, ld) = f16(row x scale - hi).
 * ld: halfs per row, a multiple of 32, columns beyond the row's width zero or absent. */
typedef struct { uint16_t* hi; float* inv; float* scale; float* max; int32_t ld; uint16_t* lo; } papr_f16_rows;

/* `mode` of papr_mlp_fwd / papr_mlp_bwd / papr_mlp_bwd_needs_weight_t: which arithmetic and which kernels carry the call.  An argument of
 * every call (ABI 16; before: a thread-local precision setting plus an environment variable read when the library loaded) --
 * the library keeps nothing between calls.  A backward call must name the mode of its forward call.
 *   PAPR_MLP_H3      fp32-parity default: wide layers as split-f16 products (hi.hi + hi.lo + lo.hi, fp32 accumulate), runs of layers fused
 *   PAPR_MLP_H1      one f16 product per fp32 product in the fused runs, f16 rows between a run and its weight gradients: the counterpart
 *                    of the reference running its attention block under fp16 autocast (`use_amp: true`, models/attn.py:248).  Since ABI 27 a run
 *                    carries ONE power-of-two scale per row through all of its layers (chosen from the maximum of the run's input row / top gradient
 *                    row: [2^3, 2^4), 2^12 of headroom before f16 overflows -- csrc/h3_common.h: one_scale_from_max) instead of one per row and layer;
 *                    a training call one of whose runs cannot keep f16 rows (skip layers, a middle width that is no multiple of 32) runs in the
 *                    parity arithmetic as a whole
 *   PAPR_MLP_F32     exact fp32 MFMA everywhere;  PAPR_MLP_FWD / _DGRAD / _LAYERS: A/B steps between F32 and H3 (split-f16 forward only /
 *                    + data-gradient / + weight gradient, one launch per layer)
 *   PAPR_MLP_H1_F32ROWS  (ABI 19-26: H1 with fp32 rows between a run and its weight gradients.)  Since ABI 27 the same as PAPR_MLP_H3: what a caller of
 *                    H1 asks for when it reads a run's inner rows itself (the one-product runs keep f16 rows only)
 *   PAPR_MLP_H3_F16ROWS  (ABI 27; round 6's gated experiment) H3 -- forward and data-gradient bit for bit -- whose fused runs keep the rows their weight
 *                    gradients read as f16 rows (each row's hi plane: the row times a power of two, rounded to f16) instead of fp32 rows: half the bytes
 *                    the runs store and the weight-gradient kernel reads, ONE f16 product per weight-gradient term (fp32 accumulation).  Results of the
 *                    forward pass and every data gradient are those of H3; weight and bias gradients carry ~2^-11 relative rounding per term.  The
 *                    inner rows of a run are then part of the saved state: papr_mlp_bwd must be given the forward call's row_absmax (as in H1)
 * An unknown value is an error. */
enum { PAPR_MLP_H3 = 0, PAPR_MLP_H1 = 1, PAPR_MLP_F32 = 2, PAPR_MLP_FWD = 3, PAPR_MLP_DGRAD = 4, PAPR_MLP_LAYERS = 5, PAPR_MLP_H1_F32ROWS = 6, PAPR_MLP_H3_F16ROWS = 7 };
size_t papr_mlp_fwd_workspace_bytes(int64_t M);
/* in_norm (optional): the same LayerNorm core in FRONT of layer 0 (FeedForward.innorm) over the first in_norm->width
 * columns of x.  x is then overwritten with its standardised rows (papr_mlp_bwd and papr_rownorm_bwd read them),
 * except in inference (row_absmax == NULL) inside a fused run without skip layers, where nobody reads x again. */
int papr_mlp_fwd(const papr_layer* layers, int n_layers, float* x, int ldx, int64_t M,
                 float* const* outs, const int32_t* ld_out, float* row_absmax, const papr_row_norm* in_norm,
                 const papr_row_norm* out_norm, void* workspace, int32_t mode, papr_stream_t stream);

/* Backward.  d_out: gradient w.r.t. the last layer's output (M, ld_out[n-1]); it is consumed
 * (overwritten).  scratch0/scratch1: two (M, max width) buffers.  d_weight[i] (n_out, ldw) and
 * d_bias[i] (n_out) are overwritten.  d_x (M, ldx) or NULL when the input needs no gradient.
 * workspace: papr_mlp_bwd_workspace_bytes(M) bytes (split-M slabs of the weight-gradient GEMMs + the
 * split-f16 scratch). */
size_t papr_mlp_bwd_workspace_bytes(int64_t M);
/* 1 if papr_mlp_bwd will read layers[i].weight_t for some layer (weight_t itself is not inspected), 0 if every
 * data-gradient runs inside a fused launch, which reads the transpose out of `weight` in place. */
int papr_mlp_bwd_needs_weight_t(const papr_layer* layers, int n_layers, int need_dx, int32_t mode);
int papr_mlp_bwd(const papr_layer* layers, int n_layers, const float* x, int ldx, int64_t M,
                 float* const* outs, const int32_t* ld_out, const float* row_absmax, float* d_out, const papr_f16_rows* d_out_f16,
                 float* scratch0, float* scratch1, int ld_scratch, float* const* d_weight,
                 float* const* d_bias, float* d_x, void* workspace, int32_t mode, papr_stream_t stream);
/* d_out_f16 (ABI 27; optional, then d_out may be NULL): the gradient rows in the one-product runs' input format (papr_attn_tail_bwd writes them so):
 * modes PAPR_MLP_H1 (lo == NULL) and PAPR_MLP_H3_F16ROWS (lo != NULL) only, the last layers a fused run of at least two layers whose last has no activation, the forward call's row_absmax given, f16 rows
 * kept by the forward call -- anything else is an error, not a fallback.  The run then stages its tiles by LDS-DMA (no pass over fp32 rows: they do not
 * exist) and the top layer's weight gradient reads the same rows.  papr_mlp_bwd_takes_f16_rows: 1 if a call with these arguments accepts them in the
 * one-product form, 2 in the parity form (the forward call's training state assumed), 0 if not at all -- ask before asking the producer for them. */
int papr_mlp_bwd_takes_f16_rows(const papr_layer* layers, int n_layers, const int32_t* ld_out, int need_dx, int32_t mode);

/* ------------------------------------------------------------------------------------
 * K4  attention tail        replaces attention("scaled-dot") + score_act (models/attn.py:217-225)
 *                           and the softmax / renormalise / weighted sum of models/model.py:519-534.
 * kp (R*k, ld_kp), qp (R, ld_qp): the two operands of the per-ray dot products, d_model wide;
 * score_bias (R) or NULL: added to every dot product of ray r before scaling.
 *   scores (R,k) = act((qp . kp_j + score_bias) / sqrt(scale_dim));  z = [scores * influ[idx], bkg_score];
 *   attn (R,k+1) = softmax(z);  fused (R,C) = sum_j attn_j / (sum_{j<k} attn_j if normalize) * v_j.
 * The literal reference form is kp = W_k K + b_k, qp = W_q Q + b_q, score_bias = NULL.  The host
 * (papr_amd/ops.py) instead passes kp = K (the key embedding itself), qp = W_k^T (W_q Q + b_q) and
 * score_bias = b_k . (W_q Q + b_q): the same bilinear form with the R*k-row W_k product replaced by
 * an R-row one.
 */
typedef struct {
    int32_t k, d_model, C;
    int32_t ld_kp, ld_qp, ld_v;
    int32_t score_act;
    int32_t normalize;
    float bkg_score;
    int32_t scale_dim;      /* scores are divided by sqrt(scale_dim) (the reference's d_model); 0 -> d_model */
    int32_t precomputed_dots;   /* papr_attn_tail_fwd only (ABI 17): 1 = kp holds the R*k dot products qp . kp_j already
                                 * (papr_row_norm.dots of the key MLP's call); qp is not read */
} papr_tail_desc;

int papr_attn_tail_fwd(const papr_tail_desc* d, const float* kp, const float* qp, const float* score_bias,
                       const float* v, const float* influ, const int32_t* idx, int64_t R, float* scores,
                       float* attn, float* fused, papr_stream_t stream);

/* d_v rows are overwritten; d_kp rows are overwritten; d_qp (R, ld_qp) overwritten; d_score_bias (R) or
 * NULL overwritten.  Influence gradient: either d_pair_influ (R*k) receives one term per pair (to be
 * summed per point by papr_segment_reduce; d_influ may be NULL), or, when d_pair_influ is NULL, the terms
 * are accumulated into d_influ (P) with atomic adds (caller zeroes).
 * kp_norm_stats (R*k, 2) or NULL: when the kp rows are the output of a row standardisation (the LayerNorm core
 * behind the key MLP, models/attn.py:39-42; stats as papr_rownorm_fwd / papr_row_norm leave them), d_kp receives the
 * gradient w.r.t. the rows BEFORE the standardisation, i.e. papr_rownorm_bwd is applied on the way out: a row of
 * d_kp is a multiple of the ray's qp row, so its two row sums come from the ray's sum of qp and from the score
 * itself and no extra pass over the (R*k, d_model) rows is needed.  score_bias (R) as in papr_attn_tail_fwd is
 * then required if one was used (NULL = none). */
int papr_attn_tail_bwd(const papr_tail_desc* d, const float* kp, const float* qp, const float* v,
                       const float* influ, const int32_t* idx, int64_t R, const float* scores,
                       const float* attn, const float* d_fused, const float* d_attn, float* d_kp,
                       float* d_qp, float* d_v, float* d_influ, float* d_score_bias, float* d_pair_influ,
                       const float* kp_norm_stats, const float* score_bias, const float* kp_mean,
                       const papr_f16_rows* d_kp_f16, const papr_f16_rows* d_v_f16, papr_stream_t stream);
/* d_kp_f16 / d_v_f16 (ABI 27; each optional): the gradient rows leave AS THE ONE-PRODUCT DATA-GRADIENT RUN TAKES THEM (papr_f16_rows: 2 bytes per element
 * plus three floats per row) instead of as fp32 rows -- d_kp / d_v may then be NULL.  What papr_mlp_bwd(mode PAPR_MLP_H1, d_out_f16) stages by LDS-DMA
 * without a pass over them (PAPR_MLP_H3_F16ROWS: the split form, lo != NULL); the values are bit for bit what that run would have made of the fp32 rows.  Needs k <= 63, d_model = 256 (d_kp_f16),
 * C = ld_v a multiple of 4 that divides 256 (d_v_f16). */
/* kp_mean (ABI 25; R*k floats or NULL; needs kp_norm_stats): the kp rows are RAW (papr_row_norm.raw_mean) -- every element read is standardised
 * first, (x - kp_mean[row]) * kp_norm_stats[2 row]. */

/* Background compositing, the last line of the attention tail (reference models/model.py:536-545): rgb (R, C) =
 * fg * (1 - a) + bkg * a with normalize (normalize_topk_attn: true), fg + bkg * a without; a = attn[:, col] (the
 * background token, attn rows of ld_attn floats), fg (R, C) the render head's output, bkg (C), C <= 8.  Backward: d_fg (R, C)
 * or NULL; d_attn (R, ld_attn), every element written (zero outside column col), or NULL; d_bkg (C) or NULL (needs
 * papr_composite_bwd_workspace_bytes(R); pixels summed block by block, blocks in a fixed order). */
int papr_composite_fwd(const float* fg, const float* attn, int32_t ld_attn, int32_t col, const float* bkg, int64_t R, int32_t C,
                       int32_t normalize, float* rgb, papr_stream_t stream);
size_t papr_composite_bwd_workspace_bytes(int64_t R);
int papr_composite_bwd(const float* d_rgb, const float* fg, const float* attn, int32_t ld_attn, int32_t col, const float* bkg, int64_t R,
                       int32_t C, int32_t normalize, float* d_fg, float* d_attn, float* d_bkg, void* workspace, papr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K5  3x3 convolution (stride 1, zero padding 1) over an NHWC map, split-f16 MFMA implicit GEMM: the 3x3 layers of the
 * U-Net render head (reference models/unet.py:16-33 inside SmallUNet, :182-258; torch.nn.Conv2d(k=3, padding=1) + ReLU).
 *   out (B*H*W, c_out) = act(bias + conv(x (B*H*W, c_in), w))        c_in % 32 == 0, c_out % 4 == 0
 * The weight is read where it lies: element (n, ky, kx, c) at w[n*w_stride_n + c*w_stride_c + ky*w_stride_ky +
 * kx*w_stride_kx] (the strides of the reference's (c_out, c_in, 3, 3) parameter in any memory format).  The data-gradient
 * of a layer is the same call on d_out with c_in / c_out and the n / c strides exchanged and flip_taps = 1.
 * workspace: papr_conv3x3_workspace_bytes() (the weight's f16 planes, the partial sums of small maps, and in its first
 * 256 bytes 64 slots for the input's maximum).  The caller zeroes those 256 bytes once, when it allocates the buffer, and
 * passes slot = (number of calls made with this buffer) mod 64: a call leaves its maximum in its slot and clears the one 32
 * calls behind, which spares a memset launch per call.
 */
size_t papr_conv3x3_weight_halfs(int32_t c_out, int32_t c_in);
size_t papr_conv3x3_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out);
int papr_conv3x3_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t c_in, const float* w, int64_t w_stride_n,
                     int64_t w_stride_c, int64_t w_stride_ky, int64_t w_stride_kx, int32_t flip_taps, const float* bias,
                     int32_t c_out, int32_t relu, float* out, void* workspace, int32_t slot, papr_stream_t stream);
/* Weight gradient of the layer: d_w (c_out, 3, 3, c_in) contiguous (= the reference's (c_out, c_in, 3, 3) gradient in
 * channels-last memory format) from d_out (B*H*W, c_out) -- already multiplied by the activation's derivative -- and the
 * layer's input x (B*H*W, c_in).  Channels multiples of 4.  workspace: papr_conv3x3_wgrad_workspace_bytes(), first 256
 * bytes zeroed by the caller at allocation; slot = (calls made with this buffer) mod 32. */
size_t papr_conv3x3_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out);
int papr_conv3x3_wgrad(const float* d_out, const float* x, int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out,
                       float* d_w, float* d_bias /* (c_out) column sums of d_out, or NULL */,
                       const uint32_t* d_out_max_bits, const uint32_t* x_max_bits /* bit pattern of max |.| of the tensor if the
                       caller has it (the slot a papr_conv3x3_fwd call on that tensor used: workspace + 4 * slot), else NULL */,
                       void* workspace, int32_t slot, papr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K5b  the other layers of the U-Net render head (reference models/unet.py), all over NHWC maps (rows = pixels):
 *
 * MaxPool2d(2) of the Down stages (models/unet.py:36-49 `nn.MaxPool2d(2)`): out (B, H/2, W/2, C), C % 4 == 0; an odd last
 * row / column is dropped.  `which` (one byte per output element, packed four to a word like the float4 they belong to, so
 * B*(H/2)*(W/2)*C/4 words; may be NULL at inference) keeps the window position 0 .. 3 of the maximum -- the FIRST one in
 * scan order, torch's rule -- for papr_maxpool2_bwd, which writes every element of d_in (B, H, W, C). */
int papr_maxpool2_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float* out, uint32_t* which, papr_stream_t stream);
int papr_maxpool2_bwd(const float* d_out, const uint32_t* which, int32_t B, int32_t H, int32_t W, int32_t C, float* d_in, papr_stream_t stream);
/* ConvTranspose2d(c_in, c_out, kernel_size=2, stride=2) of the Up stages (models/unet.py:62 `nn.ConvTranspose2d(in_channels,
 * in_channels // 2, kernel_size=2, stride=2)`), split-f16 MFMA, one launch per call, no workspace.  x (B, H, W, c_in),
 * out / d_out (B, 2H, 2W, c_out); wm = the weight as a contiguous (c_in, 2, 2, c_out) array, i.e. the reference's (c_in,
 * c_out, 2, 2) parameter in channels-last memory format (d_wm likewise); channels multiples of 64.
 * d_bias (c_out) = column sums of d_out, or NULL.  The weight-gradient reduces the pixels chunk by chunk (workspace
 * papr_upconv2x2_wgrad_workspace_bytes(): the chunks' partial tiles, added in a fixed order by a second launch). */
int papr_upconv2x2_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t c_in, const float* wm, const float* bias, int32_t c_out,
                       float* out, papr_stream_t stream);
int papr_upconv2x2_dgrad(const float* d_out, int32_t B, int32_t H, int32_t W, int32_t c_in, const float* wm, int32_t c_out, float* d_x,
                         papr_stream_t stream);
size_t papr_upconv2x2_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out);
int papr_upconv2x2_wgrad(const float* d_out, const float* x, int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t c_out, float* d_wm,
                         float* d_bias, void* workspace, papr_stream_t stream);
/* The 1x1 output convolution (models/unet.py:86-93 OutConv, `nn.Conv2d(in_channels, out_channels, kernel_size=1)`) for up to
 * four output channels: out (M, c_out) = bias + x (M, c_in) w^T, w (c_out, c_in) contiguous; plain fp32 (the layer is
 * HBM-bound).  Backward: d_x (M, c_in) or NULL; d_w (c_out, c_in) and d_bias (c_out) or NULL (pixels reduced chunk by chunk,
 * chunks added in a fixed order; workspace papr_conv1x1_bwd_workspace_bytes()); c_in a power of two 32 .. 256. */
int papr_conv1x1_fwd(const float* x, int64_t M, int32_t c_in, const float* w, const float* bias, int32_t c_out, float* out, papr_stream_t stream);
size_t papr_conv1x1_bwd_workspace_bytes(int64_t M, int32_t c_in, int32_t c_out);
int papr_conv1x1_bwd(const float* d_out, const float* x, int64_t M, int32_t c_in, const float* w, int32_t c_out, float* d_x, float* d_w,
                     float* d_bias, void* workspace, papr_stream_t stream);

/* K5c  The whole U-Net render head in one call each way: SmallUNet.forward (models/unet.py:206-258) in the shipped variant -- inc (c_in -> 128),
 * Down 128 -> 256, Down 256 -> 512, Up 512 -> 256, Up 256 -> 128 (transposed-convolution upsampling, single 3x3 convolution + ReLU per stage,
 * models/unet.py:194-198), 1x1 head -- over an NHWC map x (B, H, W, c_in) -> out (B, H, W, n_classes).  H and W multiples of 4, c_in a multiple
 * of 32, n_classes <= 4.  Same kernels as the single-layer entry points above, arranged so that no launch is spent on a tensor maximum, a weight
 * split per layer, a concatenation copy, a ReLU mask or a gradient sum (csrc/small_unet.hip; ABI 26).
 *   conv_w[i], conv_w_stride[i] = {stride of c_out, c_in, ky, kx} in floats, conv_b[i]: the five Conv2d(3x3) parameters (C_out, C_in, 3, 3) in network
 *     order inc, down1, down2, up1.conv, up2.conv -- any memory format;
 *   up_w[j] (j = 0: up1.up 512 -> 256, 1: up2.up 256 -> 128): the ConvTranspose2d parameter (C_in, C_out, 2, 2) as (C_in, 2, 2, C_out) contiguous
 *     (its channels-last memory), up_b[j];  out_w (n_classes, 128) contiguous, out_b.
 * state: papr_small_unet_state_bytes(keep) bytes, written by fwd; with keep != 0 it holds what papr_small_unet_bwd reads again (activations,
 *   pooling positions, the mirrored weight planes, the tensor maxima) and must reach that call unchanged together with x.
 * bwd: d_out (B, H, W, n_classes) -> d_x (B, H, W, c_in) or NULL, and every parameter gradient, laid out like the parameter arguments above
 *   (conv_w[i]: (C_out, 3, 3, C_in) contiguous = channels-last memory of (C_out, C_in, 3, 3)); workspace: papr_small_unet_bwd_workspace_bytes(). */
typedef struct {
    int32_t B, H, W, c_in, n_classes;
    int32_t one_product;      /* 0: three f16 products per fp32 product (fp32-parity); 1: one -- the arithmetic of the reference's fp16 autocast (`use_amp: true`,
                                 models/unet.py:212), fp32 accumulation and fp32 maps either way.  The 1x1 head is plain fp32 in both */
    const float* conv_w[5];
    int64_t conv_w_stride[5][4];
    const float* conv_b[5];
    const float* up_w[2];
    const float* up_b[2];
    const float* out_w;
    const float* out_b;
} papr_unet_desc;
typedef struct {
    float* conv_w[5];
    float* conv_b[5];
    float* up_w[2];
    float* up_b[2];
    float* out_w;
    float* out_b;
} papr_unet_grads;
size_t papr_small_unet_state_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t keep);
size_t papr_small_unet_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t c_in, int32_t n_classes);
int papr_small_unet_fwd(const papr_unet_desc* net, const float* x, float* out, void* state, int32_t keep, papr_stream_t stream);
int papr_small_unet_bwd(const papr_unet_desc* net, const float* x, const float* d_out, void* state, float* d_x, const papr_unet_grads* grads, void* workspace,
                        papr_stream_t stream);


/* ------------------------------------------------------------------------------------
 * K8  the optimizer step: every torch.optim.Adam instance of PAPR.step (reference models/model.py:439-460, one
 * `scaler.step(opt)` per parameter group; amsgrad off, maximize off, L2 weight decay) in one launch per 64 tensors.
 * The arrays are HOST arrays (they travel in the kernel arguments); pointers inside them are device pointers to the
 * optimizer's own tensors: parameter, gradient, exp_avg, exp_avg_sq (n floats each) and the step counter (one float on the
 * device, like torch's fused Adam keeps it: the call adds one and forms the bias corrections 1 - beta^step from it). */
#define PAPR_ADAM_MAX_TENSORS 64
#define PAPR_ADAM_MAX_GROUPS 8
typedef struct { float* p; const float* g; float* m; float* v; float* step; int64_t n; int32_t group; int32_t pad_; } papr_adam_tensor;
typedef struct { double lr, beta1, beta2, eps, weight_decay; int32_t found_slot; int32_t pad_; } papr_adam_group;      /* found_slot (ABI 25): papr_adam_step_scaled */
int papr_adam_step(const papr_adam_tensor* tensors, int32_t n_tensors, const papr_adam_group* groups, int32_t n_groups, papr_stream_t stream);
/* The same under a GradScaler (ABI 25; `use_amp: true`: reference train.py:172-177 scales the loss, models/model.py:442 steps through the scaler):
 * grad_scale (one float on the device, a power of two) -- every gradient is divided by it as it is read; found_inf (PAPR_ADAM_MAX_GROUPS + 1 floats
 * on the device, written: element s = 0, or 1 once any gradient element of a tensor whose group names found_slot s is NaN / +-inf; the last element
 * (ABI 26) = 1 once any slot is) -- the tensors of an overflowed slot then do not move at all: no parameter, no moment, no step counter.  One slot per OPTIMIZER gives torch.amp.GradScaler.step's contract
 * (the reference steps its optimizers one by one: an overflow in one of them skips that one only); the caller hands the slots to the scaler's
 * update().  The gradients themselves are left scaled. */
int papr_adam_step_scaled(const papr_adam_tensor* tensors, int32_t n_tensors, const papr_adam_group* groups, int32_t n_groups,
                          const float* grad_scale, float* found_inf, papr_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Optional timing of the GEMM / kNN launches with HIP events recorded on the launch stream
 * (used by bench.py for the live roofline figure; off by default, process-wide switch).
 * kernel ids: 0 gemm_nt<128x256> (fp32 MFMA)  1 gemm_nt<128x128>  2 gemm_nt<256x64>  3 gemm_nt<256x32>
 *             4 gemm_tn (split-M weight gradient)      5 ray_knn
 *             6 gemm_nt_h3 forward layer   7 gemm_nt_h3 data-gradient   (split-f16, 128x256 tile)
 *             8 gemm_tn_h3 (split-f16 weight gradient)
 *             9 mlp_chain forward run   10 mlp_chain data-gradient run   (several layers in one launch; N = number of
 *               layers, K = input width; bytes / flops below carry the algorithmic totals of the launch)
 */
typedef struct {
    int32_t kernel;
    int32_t N, K;
    int64_t M;      /* rows (rays for kernel 5, with N = points, K = k) */
    float ms;       /* hipEventElapsedTime between the events bracketing the launch */
    int64_t bytes;  /* kernels 9, 10: algorithmic HBM bytes of the launch (input rows + stored rows + masks read + weights); else 0 */
    int64_t flops;  /* kernels 9, 10: fp32-equivalent 2 M N K summed over the layers (each product costs three f16 MFMA products); else 0 */
} papr_profile_record;

int papr_profile_enable(int on);

/* Process-wide A/B and test switches (ABI 19; before: environment variables read by the library at first use).  Every value of every
 * switch but PAPR_SW_TN_JOBPAR gives BIT-IDENTICAL results (tests/test_hip_chain_variants.py, scripts/probes/knn_ab.py); they choose between forms of a kernel.
 * The defaults are what ships; papr_amd/hip.py forwards the historical PAPR_C4_* / PAPR_KNN_* environment names to these calls.
 *   PAPR_SW_C4_GENERIC  (0)    1: the fused-run kernel's specialised row-phase instantiations off
 *   PAPR_SW_C4_FUSED    (1)    0: its hot slots as two-role C++ slots instead of one generated statement
 *   PAPR_SW_C4_EARLY    (3)    bit 0 / bit 1: early request of the next tile's rows in forward / data-gradient staging slots
 *   PAPR_SW_KNN_BLOCKS  (1)    0: every point against every ray instead of the binned cloud
 *   PAPR_SW_KNN_T       (0)    rays per wave of the binned form, 0 = chosen from R
 *   PAPR_SW_WGRAD_WGS   (600)  workgroups the 3x3 weight-gradient aims for
 *   PAPR_SW_NT_VARIANT  (0)    tiling of the fp32-MFMA layer GEMM (1, 2: double-buffered; 3: four waves per SIMD)
 *   PAPR_SW_C4_DMA      (0)    1: a fused run takes its input rows split ahead of it (split_rows_kernel) and stages them by LDS-DMA instead of splitting
 *                              them itself while it stages them (an experiment of round 4 that measured slower end to end: DESIGN.md section 3)
 *   PAPR_SW_TN_JOBPAR   (1)    0: every workgroup of a batched weight-gradient launch walks all jobs (one partial tile per job and workgroup) instead of the
 *                              workgroups being dealt to the jobs (one partial tile per workgroup).  NOT bit-identical: the rows meet in another order
 *   PAPR_SW_C4_PAIRS    (1)    0: every hot slot of a fused forward run its own statement instead of two slots (tile Y's step, tile X's next step) per
 *                              statement (ABI 23)
 *   PAPR_SW_C4_PHASE    (6130) the workgroups of a fused run that carry one pair of tiles fewer than the others start late instead of finishing early
 *                              (out of phase with the rest at no cost).  value = 1000 x (the fewest steps a run must have) + (delay per step of the run in
 *                              hundreds of cycles -- shader cycles at 2.1 GHz, waited for on the constant 100 MHz clock since ABI 27: a time, whatever
 *                              the box's power state); 0: off
 *   PAPR_SW_C4_WCOPIES  (1)    timing experiment of probe builds (-DC4_X_WCOPIES; the regular library ignores it since ABI 27): 2-4 = the workgroups of
 *                              an XCD read a fused run's weight fragments from that many copies of the planes
 *   PAPR_SW_TN_TR       (1)    1: the weight gradients of full 256 x 256 layers from f16 rows on the LDS-DMA + transposing-read kernel (gemm_tn_tr_kernel, ABI 27:
 *                              whole rows by global_load_lds_dwordx4 four 32-row stages deep, operands by ds_read_b64_tr_b16, a k-step's fragments read
 *                              under the matrix instructions of the one before); 0: the register-staged kernel (gemm_tn_h3_kernel<., 2>).  A seven-job
 *                              launch of the value run 0.86 -> 0.56 + 0.13 ms, the step 8.8-8.9 -> 8.6 ms (same-box A/B); results equal to 1e-5 of a
 *                              tensor's largest element (one operand carries both rows' scales), not bit for bit
 *   PAPR_SW_C4_SUBPHASE (0)    timing experiment: workgroup b of a fused run starts ((b / 8) % 4) x value cycles late (sub-slot phases inside an XCD) */
enum { PAPR_SW_C4_GENERIC = 0, PAPR_SW_C4_FUSED = 1, PAPR_SW_C4_EARLY = 2, PAPR_SW_KNN_BLOCKS = 3, PAPR_SW_KNN_T = 4, PAPR_SW_WGRAD_WGS = 5,
       PAPR_SW_NT_VARIANT = 6, PAPR_SW_C4_DMA = 7, PAPR_SW_TN_JOBPAR = 8, PAPR_SW_C4_PAIRS = 9, PAPR_SW_C4_PHASE = 10, PAPR_SW_C4_SUBPHASE = 11, PAPR_SW_C4_WCOPIES = 12, PAPR_SW_TN_TR = 13, PAPR_SW_COUNT = 14 };
int papr_set_switch(int32_t which, int32_t value);
int32_t papr_get_switch(int32_t which);
/* Waits for the recorded events, writes up to `cap` records (oldest first), clears the log and
 * returns the number of records that were pending. */
int papr_profile_collect(papr_profile_record* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* PAPR_HIP_H */
