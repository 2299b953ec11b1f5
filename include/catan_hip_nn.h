/*
 * catan_hip_nn.h - the policy / value net's hand-written gfx950 kernels in libcatan_hip.so (same conventions as catan_hip.h:
 * plain C, caller-owned device pointers, asynchronous on the given stream, 0 / negative CATAN_E*).
 *
 * The reference keeps its net in PyTorch (RL/models) and so does this build (settlers_of_catan_rl_amd/
 * policy.py); these entry points are the fused kernels that policy.py / nn_kernels.py call where the library GEMMs and
 * elementwise kernels were launch- or bandwidth-bound at this net's small widths.  None of them is part of the env's drop-in
 * boundary - a reference-side binding needs catan_hip.h only.
 */
#ifndef CATAN_HIP_NN_H
#define CATAN_HIP_NN_H

#include "catan_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Backward of the tile encoder's pointwise sub-layer x_out = x + linear2(relu(linear1(LayerNorm(x)))) (width 64, hidden 128) for
 * everything but the weight gradients, one pass over the token rows (csrc/catan_te_bwd.hip).  bf16 row-major: dx [rows][64] = the
 * gradient of x_out; h [rows][128] = relu(linear1(.)); x [rows][64] = the LayerNorm's input; w2t [128][64] = linear2.weight^T;
 * w1t [64][128] = linear1.weight^T; ln_w float [64].  Out: dh [rows][128] = the gradient of linear1's output (catan_linear_wgrad
 * takes it for both weight gradients), dx_out [rows][64] = the gradient of x (LayerNorm backward + the residual dx); dln_w / dln_b
 * float [64] are ACCUMULATED into (zero first). */
int catan_ffn_bwd_dx(const void* dx, const void* h, const void* x, const void* w2t, const void* w1t, const float* ln_w, float eps, void* dh, void* dx_out,
                     float* dln_w, float* dln_b, int64_t rows, catan_stream_t stream);

/* catan_ffn_bwd_dx AND the sub-layer's two weight gradients in one pass (k_ffn_bwd_w): additionally n [rows][64] = LayerNorm(x) as the
 * forward stored it, OR n = NULL and ln_b float [64] = the LayerNorm's bias: the pass recomputes n from x (the forward need not store
 * it; ln_b may be NULL when n is given); dw2 float [64][128], db2 [64], dw1 [128][64], db1 [128] are ACCUMULATED into (zero first);
 * dh is not written. */
int catan_ffn_bwd(const void* dx, const void* h, const void* x, const void* n, const void* w2t, const void* w1t, const float* ln_w, const float* ln_b, float eps,
                  void* dx_out, float* dw2, float* db2, float* dw1, float* db1, float* dln_w, float* dln_b, int64_t rows, catan_stream_t stream);
/* catan_ffn_bwd plus the backward of the out-projection that produced x (x = x_in + o Wo^T + bo): d_o [rows][64] = dx_out Wo,
 * dwo float [64][64] and dbo [64] accumulated, from the dx_out rows while they are on chip.  o [rows][64] = the attention output,
 * wot = Wo^T bf16 [64][64] (k_ffn_bwd_w<true>). */
int catan_ffn_outproj_bwd(const void* dx, const void* h, const void* x, const void* n, const void* w2t, const void* w1t, const float* ln_w, const float* ln_b, float eps,
                          void* dx_out, float* dw2, float* db2, float* dw1, float* db1, float* dln_w, float* dln_b,
                          const void* o, const void* wot, void* d_o, float* dwo, float* dbo, int64_t rows, catan_stream_t stream);
/* catan_ffn_outproj_bwd with the FFN's hidden activation RECOMPUTED instead of read: h = relu(LayerNorm(x) w1^T + b1) from the rows of x
 * the pass stages anyway (w1 bf16 [128][64] row-major and b1 float [128], as the forward kernel reads them: the W1 block / b1 slice of
 * catan_tile_encoder_fwd's packed parameters), so the training forward need not store h (catan_te_saves_t.h[l] = NULL).  The LayerNorm
 * output is always recomputed (ln_b required). */
int catan_ffn_outproj_bwd_rh(const void* dx, const void* x, const void* w2t, const void* w1t, const void* w1, const float* b1, const float* ln_w, const float* ln_b, float eps,
                             void* dx_out, float* dw2, float* db2, float* dw1, float* db1, float* dln_w, float* dln_b,
                             const void* o, const void* wot, void* d_o, float* dwo, float* dbo, int64_t rows, catan_stream_t stream);
/* The attention sub-layer's input side x_mid = x + out_proj(attention(qkv(LayerNorm(x)))) (width 64): the gradient of x from dqkv
 * [rows][192] (catan_attention_bwd's output) - (dqkv . Wqkv) through the LayerNorm backward, plus the residual gradient dres = d(x_mid)
 * [rows][64] - in one pass.  wt = Wqkv^T bf16 [64][192]; x [rows][64] = the LayerNorm's input; dln_w / dln_b float [64] ACCUMULATED into. */
int catan_qkv_bwd_dx(const void* dqkv, const void* x, const void* dres, const void* wt, const float* ln_w, float eps, void* dx_out, float* dln_w, float* dln_b,
                     int64_t rows, catan_stream_t stream);

/* Every image of the net's fp32 parameters that a training step reads (bf16 copies, transposed bf16 copies, the fused tile
 * encoder's packed blocks), refreshed by one launch after the optimiser step (the reference relies on autocast's per-use casts,
 * RL/ppo/ppo.py has none: fp32 throughout).  table: n rows of catan_weight_image_t ON THE DEVICE; image i is the strided 2-D copy
 * dst[r * d_r + c * d_c] = conv(src[r * s_r + c * s_c]), strides in elements; mode 0: fp32 -> bf16, 1: fp32 -> fp32, 2: fp32 ->
 * bf16 -> fp32 (a bias as bf16 autocast hands it to a GEMM, kept in a float vector). */
typedef struct catan_weight_image {
    const float* src; void* dst; int32_t rows, cols; int64_t s_r, s_c, d_r, d_c; int32_t mode; int32_t reserved_;
} catan_weight_image_t;
int32_t catan_weight_image_bytes(void);
int catan_weight_images(const void* table, int32_t n, catan_stream_t stream);

/* The weight gradient of a WIDE linear layer over many rows (the observation trunk's 992 -> 512 product, RL/models/observation_module.py:58-63, in the
 * backward of RL/ppo/ppo.py:66): dw [out][dw_ld] (fp32; columns < in) = dy^T x, db [out] (may be NULL) = the column sums of dy; x bf16
 * [rows][in], dy bf16 [rows][out], in a multiple of 8, out a multiple of 128, both 16-byte aligned.  accumulate != 0: added to what dw / db
 * hold, else they are overwritten.  workspace: catan_wgrad_big_workspace_floats(rows, in, out) floats of scratch (the row groups' partial tiles,
 * added in index order: the result does not depend on the order in which workgroups finish).  128 x 128 output tiles, the workgroups that share
 * a row slice placed in one XCD so that the slice is read from HBM once (csrc/catan_wgrad_big.hip). */
int64_t catan_wgrad_big_workspace_floats(int64_t rows, int in_features, int out_features);
int catan_linear_wgrad_big(const void* x, const void* dy, float* dw, int64_t dw_ld, float* db, float* workspace, int64_t rows, int in_features,
                           int out_features, int accumulate, catan_stream_t stream);

/* The optimiser step of PPO.update - `nn.utils.clip_grad_norm_(parameters, max_grad_norm)` then `optim.Adam.step()` (RL/ppo/ppo.py:23,
 * 67-68) - over all parameters in two launches (sum of squared gradients per chunk; norm, clip coefficient and Adam update per chunk).
 * tensors: n_tensors rows {p, m, v} ON THE DEVICE (fp32, 16-byte aligned); chunks: n_chunks rows ON THE DEVICE, chunk i = elements
 * [offset, offset + count) of tensor `tensor`, count <= catan_adam_chunk_elements(), offset a multiple of 4; grads: n_tensors device
 * pointers ON THE DEVICE (this step's gradient of every tensor; null = the tensor takes no step, as torch skips a parameter without
 * gradient); partial: n_chunks doubles of scratch.  max_norm <= 0: no clipping.  bias_correction1 = 1 - beta1^t, bias_correction2_sqrt
 * = sqrt(1 - beta2^t) for the step number t (computed by the caller in double, as torch does on the host).  norm_out (may be null):
 * the total gradient norm before clipping.  Deterministic: the partial sums are added in index order. */
typedef struct catan_adam_tensor { float* p; float* m; float* v; } catan_adam_tensor_t;
typedef struct catan_adam_chunk { int32_t tensor; int32_t count; int64_t offset; } catan_adam_chunk_t;
int32_t catan_adam_chunk_elements(void);
int catan_adam_step(const void* tensors, const void* chunks, int32_t n_chunks, const void* grads, void* partial, float max_norm, float lr, float beta1,
                    float beta2, float eps, float bias_correction1, float bias_correction2_sqrt, float* norm_out, catan_stream_t stream);

/* catan_qkv_bwd_dx AND the QKV product's weight gradient in one pass (k_qkv_bwd_w): additionally n [rows][64] = LayerNorm 1's output,
 * or n = NULL and ln_b = the LayerNorm's bias (n recomputed from x, as catan_ffn_bwd); dw float [192][64] and db [192] are
 * ACCUMULATED into (zero first). */
int catan_qkv_bwd(const void* dqkv, const void* x, const void* dres, const void* n, const void* wt, const float* ln_w, const float* ln_b, float eps, void* dx_out,
                  float* dw, float* db, float* dln_w, float* dln_b, int64_t rows, catan_stream_t stream);

/* Row gathers of the learner (RL/ppo/ppo.py:44-50 builds a minibatch with `[obs[i] for i in indices]`; here the rollout is one
 * (T + 1, N, 1 787) bf16 tensor and a minibatch 204 800 of its 3 574-byte rows).
 * catan_gather_rows: dst row j = src row idx[j]; rows of `row_bytes` (even) at any even address and pitch.
 * catan_expand_rows: out row j = src row inv[j], rows of whole 16-byte pieces (a per-board result spread to the rows that show the board).
 * catan_segment_sum_rows: its backward - out row u = the sum (fp32, rounded to bf16) of the bf16 rows dy[order[j]], start[u] <= j <
 * start[u + 1]; start has segments + 1 entries; dy's rows lie dy_pitch_bytes apart (a column window of a wider gradient). */
int catan_gather_rows(const void* src, int64_t src_pitch_bytes, const int64_t* idx, int64_t n, void* dst, int64_t dst_pitch_bytes, int64_t row_bytes,
                      catan_stream_t stream);
int catan_expand_rows(const void* src, const int64_t* inv, int64_t n, void* out, int64_t row_bytes, catan_stream_t stream);
int catan_segment_sum_rows(const void* dy, int64_t dy_pitch_bytes, const int64_t* order, const int64_t* start, int64_t segments, void* out, int64_t row_bytes,
                           catan_stream_t stream);
/* out row r = srcs[0] row r | srcs[1] row r | ... (observation_module.py:58-60: the trunk input is the concatenation of the tile encoding and
 * the player modules' outputs).  srcs, row_bytes: HOST arrays of n <= 4 device pointers / row sizes (whole 16-byte pieces, contiguous
 * rows); out's rows lie out_pitch_bytes apart. */
int catan_concat_rows(const void* const* srcs, const int64_t* row_bytes, int n, void* out, int64_t out_pitch_bytes, int64_t rows, catan_stream_t stream);
/* The inputs of a recurrent resource head (RL/models/action_heads_module.py:258-329: give / receive lists of a proposed trade) evaluated
 * for GIVEN picks: with the four picks known every step's conditioning columns, mask and weight are functions of the earlier picks, so the
 * head runs ONCE over 4 * rows step-major rows (row i * rows + b = step i of row b).  acts int64, rows acts_ld elements apart, columns 0..3 =
 * the picks (0 = stop .. 5); cur_res float [rows][6]; fixed float [rows][kf] or NULL (kf <= 32): columns in front of the running counts.
 * -> cond [4 rows][kf + 6] (bfloat16 if cond_bf16 else float), mask float [4 rows][6], given int64 [4 rows], keep float [rows][4] (a step
 * counts only behind a non-stop pick, :306-312), out_final float [rows][6] (the counts of all four picks, column 0 cleared). */
int catan_recurrent_given(const int64_t* acts, int64_t acts_ld, const float* cur_res, const float* fixed, int32_t kf, int32_t from_hand, int64_t rows, int32_t cond_bf16,
                          void* cond, float* mask, int64_t* given, float* keep, float* out_final, catan_stream_t stream);
/* The backward of a gather whose index list is a concatenation of ranges of a permutation (the action heads' rows of a PPO minibatch:
 * action_heads_module.py:61-160 evaluates a head on the rows of its action types; here the rows are sorted by type once and every
 * head takes one or two runs of that order): out row perm[p] = the sum (fp32, rounded to bf16) of the bf16 rows dy[off_k + p - a_k]
 * over the ranges a_k <= p < b_k, plus the rows perm[p] of add0 / add1 (either may be NULL: the gradients that the source tensor's other
 * consumers produced, [n_perm][row] contiguous - autograd would add them in two more passes), zeros for a row with no term.
 * ranges: HOST array of n_ranges <= 16 triples (a, b, off), their rows of dy CONSECUTIVE from row 0 (off_0 = 0, off_k = off_k-1 + b_k-1 - a_k-1:
 * dy holds sum(b_k - a_k) rows; anything else is rejected); perm (device): MUST be a permutation of 0 .. n_perm - 1 - a row that perm misses
 * is not written (the caller's `out` is uninitialised memory); rows of whole 16-byte pieces, dy's rows dy_pitch_bytes apart. */
int catan_scatter_rows_ranges(const void* dy, int64_t dy_pitch_bytes, const int64_t* perm, int64_t n_perm, const int64_t* ranges, int n_ranges,
                              const void* add0, const void* add1, void* out, int64_t row_bytes, catan_stream_t stream);

/* Fused small-sequence multi-head attention of the policy net (RL/models/multi_headed_attention.py:25-54 as used by
 * tile_encoder.py:41-60 with L=19, 4 heads x 16 and by player_modules.py:55-69 with L<=25, 4 heads x 4).
 * qkv [B][L][3][H][HD] (fused Q/K/V projection), out [B][L][H*HD]; float32 or bfloat16 storage (is_bf16), fp32 math;
 * lens (int32 [B], may be NULL): keys >= lens[b] are masked.  bwd recomputes the probabilities. */
int catan_attention_fwd(const void* qkv, const int32_t* lens, void* out, int64_t B, int L, int H, int HD, int is_bf16, catan_stream_t stream);
int catan_attention_bwd(const void* qkv, const int32_t* lens, const void* dout, void* dqkv, int64_t B, int L, int H, int HD, int is_bf16, catan_stream_t stream);

/* LayerNorm over the last dimension D in {16, 25, 32, 64, 128, 256, 512} (x, y, dy, dx 16-byte aligned for D >= 128) with optional fused ReLU (the nn.LayerNorm + ReLU pairs of
 * RL/models/tile_encoder.py:83-91, player_modules.py:26-30,114-117).  x, y, dy, dx: [rows][D] float32 or bfloat16;
 * w, b, dw, db float32 [D]; dw/db are ACCUMULATED into (zero them first). */
int catan_layer_norm_fwd(const void* x, const float* w, const float* b, void* y, int64_t rows, int D, float eps, int relu, int is_bf16, catan_stream_t stream);
int catan_layer_norm_bwd(const void* x, const float* w, const float* b, const void* dy, void* dx, float* dw, float* db, int64_t rows, int D,
                         float eps, int relu, int is_bf16, catan_stream_t stream);
/* The same backward with the gradient of a SECOND use of x added in: dx = LayerNorm'(dy) + dres - the pre-norm sub-layers of
 * RL/models/tile_encoder.py compute x + sublayer(norm(x)), so x's gradient is the sum of the LayerNorm's and the residual
 * stream's; autograd would form it with a separate add over the whole tensor.  Widths 64, 128, 256, 512; x, dy, dres, dx
 * 16-byte aligned.  bf16: the LayerNorm term is rounded to bf16 before the add, exactly as the separate add sees it. */
int catan_layer_norm_bwd_res(const void* x, const float* w, const float* b, const void* dy, const void* dres, void* dx, float* dw, float* db,
                             int64_t rows, int D, float eps, int relu, int is_bf16, catan_stream_t stream);

/* Weight / bias gradient of a Linear layer with a huge row count and small widths (the tile / card / player modules of
 * RL/models: rows = 19 B .. 75 B, widths 6..256): dw[out][in] += sum_r dy[r][out] * x[r][in], db[out] += sum_r dy[r][out].
 * x [rows][in], dy [rows][out] bfloat16 row-major, 16-byte aligned; dw, db float32, ACCUMULATED into (zero them first);
 * db may be NULL.  MFMA (v_mfma_f32_16x16x32_bf16) with the rows split over the grid; in + 1 <= 160, out <= 256 - or a wider input
 * (in a multiple of 8 up to 1024, out a multiple of 8: the 512-wide trunk into the heads / the value head), done in column slices of 128. */
int catan_linear_wgrad_supported(int64_t rows, int in_features, int out_features);
int catan_linear_wgrad(const void* x, const void* dy, float* dw, float* db, int64_t rows, int in_features, int out_features,
                       catan_stream_t stream);
/* The same for a list of layers (a HOST array) in as few launches as their tile shapes allow: the action heads' first layers of a PPO
 * minibatch see only the rows whose action type uses the head (10^3..10^4 each), and a launch per (head, 128-column slice of the
 * 512-wide trunk) is all ramp and tail.  Every problem as catan_linear_wgrad takes it (dw / db accumulated into). */
typedef struct {
    const void* x; const void* dy; float* dw; float* db; int64_t rows; int32_t in_features, out_features;
    int32_t dw_ld, dw_col0;      /* dw_ld != 0: dw points at a wider gradient [out][dw_ld] and the layer's columns are dw_col0 .. dw_col0 + in - 1 of it
                                  * (a layer on a column slice of a parameter accumulates straight into the parameter's gradient); 0, 0: dw is [out][in] */
} catan_wgrad_problem_t;
int catan_linear_wgrad_grouped(const catan_wgrad_problem_t* problems, int32_t n, catan_stream_t stream);

/* y[r][n] = sum_k x[r][k] * w[n][k] (+ bias[n]) for huge row counts and small widths (forward and input-gradient GEMMs of
 * the same layers as catan_linear_wgrad): x [rows][in], w [out][in], bias [out] or NULL, y [rows][out], all bfloat16
 * row-major, x / w / y 16-byte aligned; in a multiple of 8 and <= 192, out <= 192, and W small enough for the register file (catan_linear_rows_supported).  MFMA with the rows split over the
 * grid and W resident in registers; HBM-bound. */
int catan_linear_rows_supported(int64_t rows, int in_features, int out_features);
int catan_linear_rows(const void* x, const void* w, const void* bias, void* y, int64_t rows, int in_features, int out_features,
                      catan_stream_t stream);
/* The same product with an elementwise epilogue fused into its row stores (out a multiple of 8; aux [rows][out] bfloat16,
 * 16-byte aligned), applied to the bf16-rounded product exactly as the separate op of the unfused net would:
 *   mode 0: none;  1: ReLU (the FFN's hidden layer, RL/models/ pointwise net);  2: + aux (the residual stream of an encoder
 *   sub-layer: x + sublayer(norm(x)));  3: zero where aux <= 0 (the backward of a ReLU whose OUTPUT is aux). */
int catan_linear_rows_fused(const void* x, const void* w, const void* bias, void* y, int64_t rows, int in_features, int out_features,
                            const void* aux, int mode, catan_stream_t stream);

/* One step of the optional LSTM policy (`include_lstm`, RL/models/policy.py:36-45,113-166: torch.nn.LSTM(512, 256), gate
 * order i, f, g, o): the arithmetic between the step's two GEMMs.  gx = x W_ih^T + b_ih + b_hh and gh = (h_prev*mask) W_hh^T,
 * both [rows][4*hidden] (fp32 or bf16); c_prev, h_out, c_out, dh, dc, dc_prev fp32 [rows][hidden]; mask fp32 [rows] or NULL
 * (the terminal mask that multiplies the incoming cell state, policy.py:119,149).
 *   c = sigmoid(f) * (c_prev * mask) + sigmoid(i) * tanh(g);  h = sigmoid(o) * tanh(c)
 * The backward recomputes the activations and writes the gradient of the pre-activation gates (= gradient of gx and of gh)
 * in the dtype of gx, and dc_prev.  hidden must be a multiple of 4, all buffers 16-byte aligned. */
int catan_lstm_cell_fwd(const void* gx, const void* gh, const float* c_prev, const float* mask, float* h_out, float* c_out, int64_t rows,
                        int hidden, int is_bf16, catan_stream_t stream);
int catan_lstm_cell_bwd(const void* gx, const void* gh, const float* c_prev, const float* mask, const float* dh, const float* dc, void* dgates,
                        float* dc_prev, int64_t rows, int hidden, int is_bf16, catan_stream_t stream);

/* Masked categorical action head (RL/distributions.py:10-40; sampled / evaluated in RL/models/action_heads_module.py:202-256):
 * logp = log_softmax(logits + log(mask)) per row.  logits fp32 [rows][K] contiguous; mask fp32 rows of pitch mask_ld (> 0 =
 * legal; may be a column window of the [rows][325] mask matrix); given int64 [rows] or NULL (evaluate those actions);
 * u fp32 [rows] uniform in [0,1) or NULL: inverse-CDF sample; both NULL: arg-max.  Outputs: action int64, its log-prob,
 * the row entropy -sum p logp over p > 0, and the row's log-sum-exp (kept for the backward).
 * backward: dlogits = dlogp * (onehot(action) - p) - dent * p * (logp + entropy), 0 on masked entries. */
int catan_categorical_fwd(const float* logits, const float* mask, int64_t mask_ld, const int64_t* given, const float* u, int64_t* action,
                          float* logp, float* entropy, float* lse, int64_t rows, int K, catan_stream_t stream);
int catan_categorical_bwd(const float* logits, const float* mask, int64_t mask_ld, const int64_t* action, const float* lse, const float* entropy,
                          const float* dlogp, const float* dent, float* dlogits, int64_t rows, int K, catan_stream_t stream);
/* catan_categorical_fwd / _bwd for GIVEN actions with the mask read as bits of the env's packed mask rows (catan_masks_packed's format: uint32
 * [n][pitch_words], bit i of the flat mask = word i >> 5, bit i & 31) instead of a float window: row j of the launch uses packed row rows_idx[j]
 * (NULL: j).  segs (HOST, 8 ints): n0, n1, then (bit offset, AND offset or -1) for the rows j < n0, n0 <= j < n1 and j >= n1 - the heads whose
 * mask row depends on the action type run on rows sorted by type (build_agent_model.py:113-124); an AND offset multiplies a second mask row in.
 * given: int64, given_ld elements between consecutive rows (a column of an action matrix), or NULL (arg-max). */
int catan_categorical_bits_fwd(const float* logits, const uint32_t* packed, int64_t pitch_words, const int64_t* rows_idx, const int32_t* segs, const int64_t* given,
                               int64_t given_ld, int64_t* action, float* logp, float* entropy, float* lse, int64_t rows, int K, catan_stream_t stream);
int catan_categorical_bits_bwd(const float* logits, const uint32_t* packed, int64_t pitch_words, const int64_t* rows_idx, const int32_t* segs, const int64_t* action,
                               const float* lse, const float* entropy, const float* dlogp, const float* dent, float* dlogits, int64_t rows, int K, catan_stream_t stream);


/* The tile encoder of the policy net (RL/models/tile_encoder.py:41-91: Linear(60, 64) + LayerNorm + ReLU, two pre-norm
 * transformer layers with 4 heads x 16 and a x2 feed-forward net, Linear(64, 25) + LayerNorm + ReLU per tile) as ONE forward
 * kernel for inference: a workgroup takes 8 boards through the whole encoder in LDS (csrc/catan_tile_encoder.hip).
 * tiles: bfloat16 [boards][19][60] contiguous, 8-byte aligned; out: bfloat16 [boards][475]; weights: bfloat16
 * [catan_tile_encoder_weight_elems()] = the matrices row-major [out][in] with `in` zero-padded to a multiple of 32 and `out`
 * to a multiple of 16, in the order first_layer [64][64], per layer qkv [192][64] (q, k, v nets stacked), out_proj [64][64],
 * linear1 [128][64], linear2 [64][128], then out_proj [32][64]; vecs: float [catan_tile_encoder_vec_elems()] = first-layer
 * bias, norm_2 weight, bias [64 each]; per layer: sublayer-0 norm weight, bias [64], qkv bias [192], out_proj bias [64],
 * sublayer-1 norm weight, bias [64], linear1 bias [128], linear2 bias [64]; then out_proj bias, norm weight, norm bias [32
 * each, 25 used].  (settlers_of_catan_rl_amd/nn_kernels.py packs them from the module.) */
int32_t catan_tile_encoder_weight_elems(void);
int32_t catan_tile_encoder_vec_elems(void);
int catan_tile_encoder_fwd(const void* tiles, const void* weights, const float* vecs, void* out, int64_t boards, catan_stream_t stream);
/* Training forward: the same kernel also stores, per tile token (boards x 19 rows, bf16, row-major [rows][width], 16-byte aligned
 * buffers), every activation the backward kernels of the encoder's sub-layers read - what the reference's autograd keeps for
 * tile_encoder.py:41-91: catan_layer_norm_bwd(_res), catan_attention_bwd, catan_linear_rows_fused (dX) and catan_linear_wgrad are
 * then run over them by the caller (settlers_of_catan_rl_amd/nn_kernels.py: _TileEncoderTrain). */
typedef struct catan_te_saves {
    void* tiles64;      /* [64]  tile features zero-padded 60 -> 64 (first_layer's input) */
    void* a0;           /* [64]  first_layer output, before LayerNorm + ReLU */
    void* xin[2];       /* [64]  encoder layer input (residual stream) */
    void* n1[2];        /* [64]  LayerNorm 1 output; may be NULL (not stored): catan_qkv_bwd recomputes it from xin */
    void* qkv[2];       /* [192] Q | K | V */
    void* o[2];         /* [64]  attention output */
    void* xmid[2];      /* [64]  residual stream after the attention sub-layer */
    void* n2[2];        /* [64]  LayerNorm 2 output; may be NULL: catan_ffn_bwd / catan_ffn_outproj_bwd recompute it from xmid */
    void* h[2];         /* [128] relu(linear1); may be NULL: catan_ffn_outproj_bwd_rh recomputes it */
    void* xfin;         /* [64]  last layer's output */
    void* p;            /* [25]  out_proj output, before the final LayerNorm + ReLU */
} catan_te_saves_t;
/* out_pitch: elements between two boards' rows of `out` (>= 475; the columns beyond 475 are not written) */
int catan_tile_encoder_fwd_train(const void* tiles, const void* weights, const float* vecs, void* out, int64_t out_pitch, const catan_te_saves_t* saves,
                                 int64_t boards, catan_stream_t stream);

/* One action head of the policy net for inference (RL/models/action_heads_module.py:202-228 + RL/distributions.py:10-40):
 * x = pre (+ cond . W1e^T) -> LayerNorm -> ReLU -> 128 x 128 -> 128 x K -> masked categorical, ONE kernel per head evaluation.
 * pre: bfloat16 [B][..], the head's 128 columns of the trunk product all heads share (row pitch pre_ld elements); cond: float
 * [B][ncond] conditioning columns that follow the trunk in mlp_1's input (ncond <= 32; NULL when 0); wts: bfloat16
 * [catan_head_weight_elems()] = W2 [128][128] | W3 [80][128] (rows >= K zero) | W1e^T [32][128] (column j of the conditioning
 * block of mlp_1.weight as row j); vec: float [catan_head_vec_elems()] = LayerNorm weight, bias, b2 (128 each), b3 [80];
 * mask: float [B][K] with row pitch mask_ld; u: uniform per row for the inverse-CDF sample, NULL = arg-max.
 * -> action int64 [B], logp float [B] (log-probability of the action under the masked distribution). */
int32_t catan_head_weight_elems(void);
int32_t catan_head_vec_elems(void);
int catan_head_fwd(const void* pre, int64_t pre_ld, const float* cond, int64_t cond_ld, int32_t ncond, const void* wts, const float* vec, float eps,
                   int32_t K, const float* mask, int64_t mask_ld, const float* u, int64_t* action, float* logp, int64_t B, catan_stream_t stream);
/* The same kernel with the autoregressive glue of the twelve heads inside ("chained"): which mask row of the env's [B][325] mask
 * matrix applies (type-conditional rows of heads 1, 6, 9: build_agent_model.py:113-124), the conditioning columns (action type,
 * played card, first resource, the trade heads' running lists; head 5: custom_mlp + LayerNorm + ReLU of proposed_trade), whether
 * the head enters the joint log-prob (log_prob_masks, build_agent_model.py:132-147), the recurrent give / receive lists with
 * their hand bookkeeping (action_heads_module.py:258-329).  A policy pass is twenty calls in the order head 0; 1, 2, 3; 5, 6, 11;
 * 4, 9, 10; 7 (steps 0..3); 8 (steps 0..3), handing `state` (float [B][catan_head_state_floats()], zeroed by the caller) on;
 * every call fills its columns of `actions` (int64 [B][18]); the last one writes the joint log-prob to logp_out.
 * custom (head 5): float [480] = custom_mlp W [32][12], b [32], custom_norm weight [32], bias [32]; forced (head 0): int64 [B],
 * entries >= 0 replace the sampled type (condition_on_action_type) or NULL; u: this evaluation's uniforms or NULL (arg-max). */
int32_t catan_head_state_floats(void);
int catan_head_chain(const void* pre, int64_t pre_ld, const void* wts, const float* vec, float eps, int32_t head_id, int32_t step, float* state,
                     const float* maskmat, const float* cur_res, const float* trade, const float* custom, const int64_t* forced, const float* u,
                     int64_t* actions, float* logp_out, int64_t B, catan_stream_t stream);

/* The dev-card list modules of the policy net (RL/models/player_modules.py:55-69: embedding(6 x 16) -> 4-head attention with
 * key mask -> out projection -> LayerNorm(16) -> zero the padding -> sum over the list), one fused kernel, evaluated per card
 * CLASS (a list has <= 6 distinct ids; see csrc/catan_nn.hip).  ids: [rows][pitch] integers of id_bytes (1, 4 or 8) bytes, the
 * first min(lens[r], 25) entries of a row are valid; params: float[catan_card_summary_params()] = S[4][6][6] (scaled q.k of the
 * six ids per head), V[6][16], out-projection W[16][16], its bias[16], LayerNorm weight[16], bias[16]; out / dout: float
 * [rows][16]; dparams: gradients of `params`, ACCUMULATED into (zero first).
 * keys (int32 [rows], may be NULL): the forward also writes each list's PATTERN number - its six counts, bounded by the deck
 * (2 x 15 x 6 x 3 x 3 x 3 = catan_card_summary_patterns() patterns; pattern k has counts c0 = k % 2, c1 = k / 2 % 15, c2 = k / 30 % 6,
 * c3 = k / 180 % 3, c4 = k / 540 % 3, c5 = k / 1620) - or -1 for counts outside the deck.  The gradient is linear in dout, so a
 * caller may sum dout per pattern and differentiate catan_card_summary_patterns() synthetic lists instead of all rows;
 * only_unkeyed (int32 [rows], may be NULL) makes the backward skip the rows whose entry is >= 0 (those went through the patterns);
 * n_unkeyed (int32 [1] on the device, may be NULL; needs only_unkeyed): the launch returns at once when it holds 0. */
int32_t catan_card_summary_params(void);
int32_t catan_card_summary_patterns(void);
int catan_card_summary_fwd(const void* ids, int id_bytes, int64_t pitch, const int32_t* lens, const float* params, float eps, float* out,
                           int32_t* keys, int64_t rows, catan_stream_t stream);
/* Inference: out[r] = table[pattern of list r] with table = float [catan_card_summary_patterns()][16], the outputs of
 * catan_card_summary_fwd on the synthetic list of every pattern (rebuilt by the caller when the weights change); lists whose
 * counts fall outside the deck are evaluated directly from `params`. */
int catan_card_summary_lookup(const void* ids, int id_bytes, int64_t pitch, const int32_t* lens, const float* table, const float* params, float eps,
                              float* out, int64_t rows, catan_stream_t stream);
int catan_card_summary_bwd(const void* ids, int id_bytes, int64_t pitch, const int32_t* lens, const float* params, float eps, const float* dout,
                           float* dparams, const int32_t* only_unkeyed, const int32_t* n_unkeyed, int64_t rows, catan_stream_t stream);
/* dpat[c][keys[r]][0..15] += dout[r][0..15] for the rows with keys[r] >= 0; dpat: float [replicas][catan_card_summary_patterns()][16],
 * zero before the call: copy c takes the rows of the workgroups b with b % replicas == c (the caller sums the copies) - a
 * handful of patterns covers most lists, and one copy would serialise the atomics of the whole grid on a few cache lines.
 * n_unkeyed (int32 [1] on the device, may be NULL): += the number of rows with keys[r] < 0. */
int catan_card_pattern_sum(const int32_t* keys, const float* dout, float* dpat, int replicas, int32_t* n_unkeyed, int64_t rows, catan_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* CATAN_HIP_NN_H */
