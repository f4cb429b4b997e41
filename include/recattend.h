/*
 * recattend.h — C ABI of librecattend.so: the MI355X (gfx950) implementation of the
 * recurrent-attention decode loop of renmengye/rec-attend-public.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes (no torch / TF
 * types) and returns an int:
 *     0   success
 *    <0   invalid argument (RA_E_*), nothing was launched / written
 *    >0   a hipError_t raised by the runtime (see ra_last_error_string()), or for the
 *         Hungarian entry points the documented positive status 1
 * The library never allocates memory that crosses the boundary and keeps no global
 * state besides a thread-local last-error string; every buffer (inputs, outputs,
 * workspaces) is owned by the caller.  Device pointers are HIP device pointers on the
 * current device; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All tensors are float32, NHWC, densely packed unless a stride is stated.
 *
 * Reference interfaces each entry point replaces are cited as file:line of the
 * reference checkout (renmengye/rec-attend-public).
 */
#ifndef RECATTEND_H_
#define RECATTEND_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RA_E_INVALID (-1)   /* null pointer / non-positive dim */
#define RA_E_SHAPE (-2)     /* unsupported shape (alignment / divisibility) */
#define RA_E_WORKSPACE (-3) /* workspace too small */
#define RA_E_HUNG_BFS (-12)      /* hungarian.cc:124-127 LOG(FATAL) */
#define RA_E_HUNG_PATH (-13)     /* hungarian.cc:146-150,156-160 */
#define RA_E_HUNG_FLOW (-14)     /* hungarian.cc:184-188 */
#define RA_E_HUNG_EQUALIZE (-15) /* hungarian.cc:446-450 */

/* ABI version: bumped whenever an entry point of this header is removed or changes its signature (additions do not
 * bump it).  101 = round 3's removal of the table-form attention entry points and the phase-fused patch net; 102-113 =
 * rounds 3-5 (signature changes of the controller / conv entry points as the kernels were rebuilt; `git log -p
 * include/recattend.h` has each).  ra_native.py refuses a library whose version differs from the header it was written against. */
#define RA_ABI_VERSION 113
int ra_version(void);
/* Human-readable description of the last non-zero return on this thread. */
const char *ra_last_error_string(void);

/* Test aid (no reference counterpart): fills the LDS of every CU with NaN, so a kernel that
 * reads shared memory it never wrote fails the parity tests deterministically. */
int ra_debug_poison_lds(void *stream);
/* Test aid (no reference counterpart): parks `n_wg` workgroups of `lds_bytes` dynamic LDS each (<= 160 KB; above 80 KB that is
 * one per CU) on XCD `xcd` (HW_REG_XCC_ID) for `millis` ms of wall clock (s_memrealtime, capped at 20 s): a launch of 8 * n_wg
 * workgroups whose members on any other XCD leave at once.  resident[0] (device int, zeroed by the caller, may be NULL) counts
 * the parked workgroups as they start.  For tests of what a launch that relies on co-residency (ra_controller_split_f32) does
 * when part of an XCD is taken: tests/test_full_model_gpu.py. */
int ra_debug_park_xcd(int xcd, int n_wg, int lds_bytes, int millis, int *resident, void *stream);

/* ------------------------------------------------------------------------------------
 * Hungarian matching — replaces the TF custom op
 *   REGISTER_OP("Hungarian").Input("weights: float").Output("matching: float")
 *     .Output("cover_x: float").Output("cover_y: float")          hungarian.cc:26-30
 *   HungarianOp::Compute                                           hungarian.cc:36-85
 * weights [B,N,M] -> matching [B,N,M], cover_x [B,N] (op shape [B,N,1]), cover_y [B,M]
 * (op shape [B,1,M]).  The op's 2-D form (hungarian.cc:58-60,490-504) is B == 1.
 * Bit-exact with the reference's float32 control flow.  Returns 0, or 1 when any example
 * hit the outer 1000-iteration cap (the reference logs an error and returns the partial
 * matching, hungarian.cc:363-377 — so does this), or RA_E_HUNG_* where the reference
 * would LOG(FATAL).  Host pointers; runs on the calling thread; re-entrant.
 * ---------------------------------------------------------------------------------- */
int ra_hungarian_f32(const float *weights, int B, int N, int M, float *matching,
                     float *cover_x, float *cover_y);

/* Same contract on DEVICE pointers (one workgroup per example), so the training step
 * needs no device->host sync.  status_dev (nullable): int[B] per-example codes as above
 * (the return value only reports launch errors).  ws: device scratch of at least
 * ra_hungarian_dev_workspace_bytes(B, N, M) bytes. */
size_t ra_hungarian_dev_workspace_bytes(int B, int N, int M);
int ra_hungarian_f32_dev(const float *weights, int B, int N, int M, float *matching,
                         float *cover_x, float *cover_y, int *status_dev, void *ws,
                         size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * K1  conv3x3 (+bias+BatchNorm(eval)+ReLU+maxpool) as an f32-MFMA implicit GEMM.
 * Replaces one layer of nnlib.cnn's run_cnn (nnlib.py:229-253: conv2d :6-12 + b,
 * batch_norm :65-128 in eval mode, act, max_pool :15-25) and — with RA_CONV_TRANSPOSED
 * packing + `upsample` — one layer of nnlib.dcnn's run_dcnn (nnlib.py:362-400:
 * concat(prev, skip), conv2d_transpose SAME stride 1|2, + b, BN, act).
 *
 * Input = channel-concat of src0 [B,Hs,Ws,C0] and (optional) src1 [B,Hs,Ws,C1];
 * C0 % 4 == 0, C1 % 4 == 0.  With upsample == 1 the conv runs over the zero-stuffed
 * image U[2i+1,2j+1] = src[i,j] of size [2Hs,2Ws] (the stride-2 transposed conv).
 * Output y [B,Ho,Wo,Cout], Ho = H/pool, Wo = W/pool, pool in {1,2} (H, W even if 2).
 * scale/shift [CoutP] fold the conv bias and BN:  y = max?(relu?(conv*scale + shift)).
 * wpacked comes from ra_conv_pack_weights (device copy of it).
 * ---------------------------------------------------------------------------------- */
#define RA_CONV_TRANSPOSED 1 /* w is a conv2d_transpose filter [3,3,Cout,Cin] (nnlib.py:320-325) */

/* Padded channel counts the kernel works in: CinP = roundup(Cin,4), CoutP in {16,32,64,128,...}. */
int ra_conv_cout_padded(int Cout);
/* Number of floats in the packed weight buffer for a [3,3,Cin,Cout] filter (Cin % 4 == 0). */
size_t ra_conv_packed_floats(int Cin, int Cout);
/* Host-side repack of a TF-layout filter into the kernel's B-operand order.
 * w: [3,3,Cin_w,Cout] (or [3,3,Cout,Cin_w] with RA_CONV_TRANSPOSED), host pointer.
 * chan_map (nullable): int[Cin] giving, for every kernel input channel, the filter's
 * input-channel index it multiplies, or -1 for a zero (padding) channel; NULL = identity
 * with Cin_w == Cin.  out: host buffer of ra_conv_packed_floats(Cin, Cout) floats. */
int ra_conv_pack_weights(const float *w, int Cin_w, int Cout, int Cin, const int *chan_map,
                         int flags, float *out);
/* Host-side fold of bias + BN(eval) into scale/shift [CoutP] (nnlib.py:119: eps = 1e-3).
 * beta/gamma/mean/var nullable together (no BN: scale = 1, shift = bias). */
int ra_conv_fold_bn(const float *bias, const float *beta, const float *gamma, const float *mean,
                    const float *var, int Cout, float eps, float *scale, float *shift);

/* plane (nullable): a [B,Hs,Ws] tensor that REPLACES input channel plane_chan of src0 — the
 * canvas kept outside the packed image (full_model.py:648: canvas is one channel of ccnn_inp). */
int ra_conv3x3_f32(const float *src0, int C0, const float *src1, int C1, int B, int Hs, int Ws,
                   int upsample, const float *wpacked, const float *scale, const float *shift,
                   int Cout, int relu, int pool, const float *plane, int plane_chan, float *y,
                   void *stream);

/* Mixed precision for the training step (model_opt['compute_dtype'] = 'bf16'; the reference trains in float32 —
 * this is an extension behind its option dictionary): the same layer with bf16 OPERANDS — pixels and weights are
 * rounded to bf16 (round-to-nearest-even) on their way from LDS / registers into v_mfma_f32_16x16x16_bf16 — and
 * float32 accumulation, epilogue and tensors. */
int ra_conv3x3_bf16ops_f32(const float *src0, int C0, const float *src1, int C1, int B, int Hs, int Ws,
                           int upsample, const float *wpacked, const float *scale, const float *shift,
                           int Cout, int relu, int pool, const float *plane, int plane_chan, float *y,
                           void *stream);

/* ---- bf16 STORAGE of the tensors between the conv layers' passes (model_opt['compute_dtype'] = 'bf16'; an extension: the
 * reference trains in float32, full_model.py:1039-1057).  On top of the bf16-operand kernels above, the pre-activation
 * outputs U, the activations Y and the gradients dY / dU are kept as bf16 in HBM (2 bytes per element, round to nearest
 * even on the way out, exact on the way in); BatchNorm statistics, sums, parameter gradients, master weights and Adam
 * state stay float32.  Tensors in bf16 are `void *`. ---- */
/* conv3x3 with bf16 operands; store_flags bit 0: src0 / src1 hold bf16, bit 1: y is written as bf16.  part != NULL: the
 * batch moments of the float32 accumulators as ra_conv3x3_moments_f32 (pool 1, Cout % 4 == 0); part == NULL: `pool` as given. */
int ra_conv3x3_bf16_f32(const void *src0, int C0, const void *src1, int C1, int B, int Hs, int Ws, int upsample,
                        const float *wpacked, const float *scale, const float *shift, int Cout, int relu, int pool, void *y,
                        float *part, size_t part_floats, int *nparts, int store_flags, void *stream);
/* ra_bn_act_pool_f32 with flags bit 0: u stored as bf16, bit 1: y written as bf16 (0, 1, 3; C % 4 == 0, C / 4 a power of two). */
int ra_bn_act_pool_bf16_f32(const void *u, const float *mean, const float *var, const float *gamma, const float *beta, float eps,
                            int relu, int pool, int B, int H, int W, int C, void *y, int flags, void *stream);
/* ra_bn_act_pool_bwd_acc_f32 / ra_bn_act_pool_bwd_grouped_f32 with flags bit 0: u read and du written as bf16, bit 1: dy read
 * as bf16 (0, 1, 3). */
int ra_bn_act_pool_bwd_acc_bf16_f32(const void *u, const void *dy, const float *mean, const float *var, const float *gamma,
                                    const float *beta, float eps, int relu, int pool, int B, int H, int W, int C, float *ws,
                                    size_t ws_floats, float *dgamma, float *dbeta, void *du, float *acc_gamma, float *acc_beta,
                                    int flags, void *stream);
int ra_bn_act_pool_bwd_grouped_bf16_f32(const void *u, const void *dy, const void *const *tabs, int G, float eps, int relu, int pool,
                                        int B, int H, int W, int C, float *ws, size_t ws_floats, float *dgamma, float *dbeta,
                                        void *du, int flags, void *stream);
/* ra_conv3x3_wgrad_acc_bf16ops_f32 with fmt bit 0: x stored as bf16, bit 1: du stored as bf16.  (The multi-call form
 * ra_conv3x3_wgrad_multi_acc_f32 takes the same two bits shifted up by one in its bf16_operands argument.) */
int ra_conv3x3_wgrad_acc_bf16_f32(const void *x, int Cin, int B, int Hs, int Ws, int upsample, const void *du, int Cout, float *ws,
                                  size_t ws_floats, const int *chan_map, int cin_w, int transposed, float *gw, float *gb, int fmt,
                                  void *stream);

/* The training forward of a BatchNorm layer (nnlib.py:98: tf.nn.moments of the conv output over batch, height, width):
 * ra_conv3x3_moments_f32 is ra_conv3x3_f32 / _bf16ops_f32 (pool 1, Cout % 4 == 0, no canvas plane) whose epilogue also
 * leaves per-wave channel sums of its output y BEFORE the ReLU in `part` (ra_conv3x3_moments_part_floats(Cout) floats;
 * *nparts = records written, a host int); ra_bn_moments_from_partials_f32 combines them (Chan's update, float64) into
 * mean[C] / var[C] (biased, as tf.nn.moments).  One small launch instead of two more passes over y. */
size_t ra_conv3x3_moments_part_floats(int Cout);
int ra_conv3x3_moments_f32(const float *src0, int C0, const float *src1, int C1, int B, int Hs, int Ws,
                           int upsample, const float *wpacked, const float *scale, const float *shift,
                           int Cout, int relu, int bf16_operands, float *y, float *part, size_t part_floats,
                           int *nparts, void *stream);
int ra_bn_moments_from_partials_f32(const float *part, int nparts, int C, float *mean, float *var, void *stream);

/* Two consecutive layers fused in one launch (the intermediate activation stays in LDS):
 *   A: conv3x3 + scale/shift [+ReLU], no pool, optional zero-stuffed (stride-2 transposed) input
 *   B: conv3x3 + scale/shift [+ReLU] + max-pool poolB
 * src [B,Hs,Ws,Cin] single source, Cin in {4,8,16,32}; CoutA in {8,16,32}; CoutB <= 32.
 * wpA / wpB from ra_conv_pack_weights with Cin resp. CoutA input channels. */
int ra_conv_pair_supported(int Cin, int CoutA, int CoutB);
int ra_conv_pair_f32(const float *src, int Cin, int B, int Hs, int Ws, int upsampleA,
                     const float *wpA, const float *scaleA, const float *shiftA, int CoutA, int reluA,
                     const float *wpB, const float *scaleB, const float *shiftB, int CoutB, int reluB,
                     int poolB, const float *plane, int plane_chan, float *y, void *stream);

/* K1s (round 5) — the same layer (conv3x3 SAME + folded BatchNorm + ReLU + max-pool, nnlib.py:229-253) as a DIRECT convolution
 * on the BF16 matrix pipe at float32 accuracy: every float32 operand is the exact sum of three bf16 pieces, and six of the
 * nine piece products (everything above 2^-24 of a product) run as v_mfma_f32_16x16x32_bf16 with float32 accumulation
 * (csrc/ra_conv_split.hip).  Cin in {16, 24, 32, 64}, Cout % 16 == 0, pool 1 | 2, H and W multiples of 4 (round 6: the 16 x 16
 * tiles — 8 rows at Cin = 64 — may be ragged at the right and bottom edges: KITTI's 56- and 28-pixel maps, the 24- and 12-pixel
 * patch maps) (ra_conv_split_supported).  wpacked: ra_conv_split_packed_halfs() 16-bit words, the filter's three bf16 pieces in B-operand
 * order, from the reference's [3,3,Cin,Cout] filter by ra_conv_split_pack_weights (host) or ra_conv_split_pack_weights_dev
 * (device pointers; transposed != 0: w is a conv2d_transpose filter [3,3,Cout,Cin] — taps flipped, in / out swapped, as
 * RA_CONV_TRANSPOSED: the packing of a cnn layer's DATA GRADIENT, which the training step runs as this conv with scale 1,
 * shift 0).  scale / shift as ra_conv3x3_f32. */
int ra_conv_split_supported(int Cin, int Cout, int pool, int H, int W);
size_t ra_conv_split_packed_halfs(int Cin, int Cout);
int ra_conv_split_pack_weights(const float *w, int Cin, int Cout, unsigned short *out);
int ra_conv_split_pack_weights_dev(const float *w, int Cin, int Cout, int transposed, unsigned short *out, void *stream);
int ra_conv_split_f32(const float *x, int B, int H, int W, int Cin, const unsigned short *wpacked, const float *scale,
                      const float *shift, int Cout, int relu, int pool, float *y, void *stream);
/* ... with channel plane_chan of x taken from the plane [B,H,W] instead (round 6; Cin 16 | 24 | 32): the FIRST controller-CNN
 * layer of the KITTI / Cityscapes architectures, whose packed input concat(x, canvas, d_in, y_in) (full_model.py:640-661; 13 / 21
 * channels packed to 16 / 24) keeps the canvas in its own plane (as ra_conv3x3_f32's plane / plane_chan). */
int ra_conv_split_plane_f32(const float *x, int B, int H, int W, int Cin, const float *plane, int plane_chan,
                            const unsigned short *wpacked, const float *scale, const float *shift, int Cout, int relu, int pool,
                            float *y, void *stream);

/* K1w: conv3x3 SAME + folded BN + ReLU + optional 2x2 max-pool (nnlib.py:229-253) as Winograd F(2x2, 3x3)
 * on the f32 MFMA: 2.25x fewer matrix multiplies than ra_conv3x3_f32 for the same layer, results equal
 * to ~1e-6 relative (summation order and the 0.5 factors of the filter transform).  Cin 16 | 32,
 * Cout % 16 == 0, H and W multiples of 16 (ra_conv_wino_supported).  wpacked: the transformed filters
 * G g G^T in MFMA B-operand order, ra_conv_wino_packed_floats() floats, from the reference's
 * [3,3,Cin,Cout] filter by ra_conv_wino_pack_weights (host).  scale / shift: the folded BatchNorm of
 * this timestep, as for ra_conv3x3_f32.  x [B,H,W,Cin] -> y [B,H/pool,W/pool,Cout]. */
int ra_conv_wino_supported(int Cin, int Cout, int pool, int H, int W);
size_t ra_conv_wino_packed_floats(int Cin, int Cout);
int ra_conv_wino_pack_weights(const float *w, int Cin, int Cout, float *out);
int ra_conv_wino_f32(const float *x, int B, int H, int W, int Cin, const float *wpacked,
                     const float *scale, const float *shift, int Cout, int relu, int pool, float *y,
                     void *stream);

/* K1pw: ra_conv_pair_f32 for the pair 8 -> 16 -> 16 with poolB = 2 (the controller CNN's L2+L3 at the CVPPP
 * arch, full_model.py:640-661 / nnlib.py:229-253) with layer B computed as Winograd F(2x2,3x3) straight
 * from the LDS tile layer A was written to.  wpA: ra_conv_pack_weights (Cin = 8); wpB_wino:
 * ra_conv_wino_pack_weights of layer B's [3,3,16,16] filter.  H, W multiples of 16. */
int ra_conv_pair_wino_supported(int Cin, int CoutA, int CoutB, int poolB, int H, int W);
int ra_conv_pair_wino_f32(const float *x, int B, int H, int W, const float *wpA, const float *scaleA,
                          const float *shiftA, int reluA, const float *wpB_wino, const float *scaleB,
                          const float *shiftB, int reluB, float *y, void *stream);

/* The first controller-CNN pair with the image part of layer A cached.  Of layer A's input
 * concat(x, canvas) (full_model.py:640-661) only the canvas changes between timesteps
 * (:843-848), and a convolution is linear in its input channels, so
 *   S = sum_{taps, ci != plane_chan} x * W          (no bias, no BN)
 * is computed ONCE per forward (ra_conv_first_cache_f32; `cache` holds it in the accumulator
 * layout of the pair kernel, ra_conv_first_cache_floats() floats) and every timestep computes
 *   A = relu?((S + conv(canvas, W[:, :, plane_chan, :])) * scaleA(tt) + shiftA(tt)),  B as in
 * ra_conv_pair_f32 with poolB = 2 — 3 instead of 12 MFMA k-steps for layer A and 4 instead of 20
 * staged bytes per pixel.  Cin = 4, CoutA = 8, CoutB <= 8 only (ra_conv_first_cache_supported);
 * src [B,H,W,4] packed image, plane [B,H,W] the canvas, wpA packed for Cin = 4. */
int ra_conv_first_cache_supported(int Cin, int CoutA, int CoutB, int poolB, int H, int W);
size_t ra_conv_first_cache_floats(int B, int H, int W);
int ra_conv_first_cache_f32(const float *src, int B, int H, int W, const float *wpA, int CoutA,
                            int plane_chan, float *cache, void *stream);
/* The first timestep of a forward, where the canvas is all zero (full_model.py:239): the plain pair
 * (layer A over all 4 channels) that also WRITES the cache — with a zero canvas layer A's raw sums are
 * exactly the image part — so that no separate cache launch is needed.  The caller must only use it
 * while `plane` is zero. */
int ra_conv_pair_fill_cache_f32(const float *src, const float *plane, int plane_chan, int B, int H,
                                int W, const float *wpA, const float *scaleA, const float *shiftA,
                                int reluA, const float *wpB, const float *scaleB,
                                const float *shiftB, int CoutB, int reluB, float *cache, float *y,
                                void *stream);
/* The same launch with a constant fill riding on it: fill_dst[0 .. fill_floats) = fill_value, written a few
 * stores per thread and tile by the workgroups of this (MFMA-bound) kernel while HBM is otherwise idle.  The
 * decode loop's once-per-forward prefill of y_out (sigmoid(beta) outside every attention window,
 * full_model.py:813-818) travels this way instead of as a 28 us launch of its own.  fill_dst 16-byte
 * aligned, fill_floats % 4 == 0, < 2 GiB; fill_dst == NULL: no fill. */
int ra_conv_pair_fill_cache_rider_f32(const float *src, const float *plane, int plane_chan, int B, int H, int W,
                                      const float *wpA, const float *scaleA, const float *shiftA, int reluA,
                                      const float *wpB, const float *scaleB, const float *shiftB, int CoutB,
                                      int reluB, float *cache, float *y, float *fill_dst, size_t fill_floats,
                                      float fill_value, void *stream);
int ra_conv_pair_cached_f32(const float *cache, const float *plane, int plane_chan, int B, int H,
                            int W, const float *wpA, const float *scaleA, const float *shiftA,
                            int reluA, const float *wpB, const float *scaleB, const float *shiftB,
                            int CoutB, int reluB, float *y, void *stream);

/* ------------------------------------------------------------------------------------
 * K2  controller: glimpse read-out + LSTM + glimpse MLP (x iters) + controller MLP +
 * attention-parameter decode.  Replaces full_model.py:668-722 (= box_model.py:416-468):
 * nnlib.lstm unroll (nnlib.py:637-649, state = [c|h], zeroed every call
 * full_model.py:674), nnlib.mlp run_mlp (nnlib.py:476-493) for glimpse_mlp (relu..,
 * softmax) and ctrl_mlp, modellib.get_unnormalized_attn (modellib.py:843-847),
 * get_normalized_var (:782-793).
 * One workgroup per example.
 * ---------------------------------------------------------------------------------- */
typedef struct ra_ctrl_desc {
  int G;        /* glimpse map size gh*gw (full_model.py:311) */
  int Cf;       /* feature depth (full_model.py:312), % 4 == 0 */
  int hid;      /* ctrl_rnn_hid_dim, % 4 == 0, <= 256 */
  int iters;    /* num_ctrl_rnn_iter */
  int n_gmlp;   /* num_glimpse_mlp_layers (>= 1); dims hid,..,hid,G */
  int n_cmlp;   /* num_ctrl_mlp_layers (>= 1); dims hid, mlp_dim.., 9 */
  int mlp_dim;  /* ctrl_mlp_dim, % 4 == 0 */
  int H, W;     /* image size, for the un-normalisation */
  int Fh, Fw;   /* filter (patch) size */
  int squash;       /* squash_ctrl_params (full_model.py:695-697) */
  int fixed_var;    /* full_model.py:702-706 */
  int dynamic_var;  /* full_model.py:708-709 */
  int fixed_gamma;  /* full_model.py:711-716 */
} ra_ctrl_desc;

/* Packed controller weights (device), produced on the host by ra_ctrl_pack_weights:
 *   lstm  [(Cf+hid)][4*hid]   rows = [x ; h], columns gate-major i,f,o,u ; + bias [4*hid]
 *   gmlp  layer l: [hid][NoutP_l] + bias[NoutP_l]   (last layer NoutP = roundup(G,4))
 *   cmlp  layer l: [..][NoutP_l] + bias            (last layer 9 -> 12)
 * Layout offsets are internal; both sides use ra_ctrl_packed_floats. */
size_t ra_ctrl_packed_floats(const ra_ctrl_desc *d);
/* lstm_w: 12 host pointers in the order w_xi,w_hi,b_i, w_xf,w_hf,b_f, w_xu,w_hu,b_u,
 * w_xo,w_ho,b_o (nnlib.py:532-609); gmlp_w / cmlp_w: {w_0,b_0,w_1,b_1,...}. */
int ra_ctrl_pack_weights(const ra_ctrl_desc *d, const float *const *lstm_w,
                         const float *const *gmlp_w, const float *const *cmlp_w, float *out);

#define RA_ATTN_STRIDE 16
/* attn record [B][RA_ATTN_STRIDE]: 0 ctr_y, 1 ctr_x, 2 size_y, 3 size_x, 4 lg_var_y,
 * 5 lg_var_x, 6 attn_gamma = exp(lg_gamma), 7 box_gamma = exp(box_lg_gamma),
 * 8 y_out_lg_gamma, 9 ctr_norm_y, 10 ctr_norm_x, 11 lg_size_y, 12 lg_size_x. */
int ra_controller_f32(const ra_ctrl_desc *d, const float *feat /*[B,G,Cf]*/, const float *wpacked,
                      int B, float *h_last /*[B,hid]*/, float *ctrl_out /*[B,9]*/,
                      float *glimpse_maps /*[B,iters,G] nullable*/,
                      float *attn /*[B,RA_ATTN_STRIDE]*/, void *stream);

/* Split form of the same controller: 16 workgroups per example, each keeping its slice of
 * the LSTM / glimpse-MLP weights in LDS for the whole launch and exchanging only the small
 * activation vectors through 8-byte {tag, value} granules in `ws` (device, must be zero-filled
 * ONCE when allocated; generation-tagged, so graph replays need no memset; one workspace per
 * concurrently running launch).  Same results up to float32 summation order.  B <= 14.
 * status_dev (nullable): set to 1 if a peer workgroup timed out.  * Round 5: where the device's workgroups report the XCC ids 0..7, the launch is a 1-D grid whose workgroups draw their (image,
 * slice) role from a per-XCD ticket (behind the images' granules in `ws`: ra_ctrl_split_workspace_bytes includes it), so that
 * the 16 workgroups of an image share ONE XCD's L2 and exchange through plain stores and L1-bypassing loads instead of
 * agent-scope atomics: 63 -> 57 us per launch at cfg2.  RA_CTRL_XCD=0 selects the grid (16, B) form.  Same results bit for bit.
*/
int ra_ctrl_split_supported(const ra_ctrl_desc *d);
size_t ra_ctrl_split_packed_floats(const ra_ctrl_desc *d);
size_t ra_ctrl_split_workspace_bytes(const ra_ctrl_desc *d, int B);
int ra_ctrl_split_pack_weights(const ra_ctrl_desc *d, const float *const *lstm_w,
                               const float *const *gmlp_w, const float *const *cmlp_w, float *out);
int ra_controller_split_f32(const ra_ctrl_desc *d, const float *feat, const float *wpacked, int B,
                            float *h_last, float *ctrl_out, float *glimpse_maps, float *attn,
                            void *ws, size_t ws_bytes, int *status_dev, void *stream);
/* K2b: the split controller with its 16 weight slices shared by groups of g = ra_ctrl_batch_group_images(d, B)
 * images — 4 for launches of up to 8 images, 8 above where that fits the LDS — (16 workgroups per GROUP
 * instead of per image; weights packed exactly as for ra_controller_split_f32).  Results equal ra_controller_split_f32's to float32 round-off; a launch is
 * ceil(B / g) * 16 workgroups (at most 224: RA_E_SHAPE beyond), so several
 * launches from different streams are resident together (ra_controller_split_f32 beside its own kind
 * can starve: its spinning workgroups hold the CUs their peers wait for).  ws: zero-filled,
 * ra_ctrl_batch_workspace_bytes(), owned by one stream of launches; status_dev as above. */
int ra_ctrl_batch_group_images(const ra_ctrl_desc *d, int B);
int ra_ctrl_batch_supported(const ra_ctrl_desc *d);
size_t ra_ctrl_batch_workspace_bytes(const ra_ctrl_desc *d, int B);
int ra_controller_batch_f32(const ra_ctrl_desc *d, const float *feat, const float *wpacked, int B,
                            float *h_last, float *ctrl_out, float *glimpse_maps, float *attn, void *ws,
                            size_t ws_bytes, int *status_dev, void *stream);
/* ... on the XCD-local exchange (round 6): group g's 16 workgroups run on XCD (g + xcd_offset) mod 8 and exchange through that XCD's
 * L2 (as ra_controller_split_f32's XCD-local form) instead of agent-scope atomics.  At most 8 groups per launch; the CALLER vouches that
 * launches which can run at the same time use offsets that keep their groups on different XCDs (DecodePipeline: slot k of 4 streams with 2
 * groups each -> offset 2 k).  xcd_offset < 0, more than 8 groups, RA_CTRL_XCD=0 or a device whose workgroups do not report XCC ids 0..7:
 * the agent-scope form (= ra_controller_batch_f32).  Same workspace (ra_ctrl_batch_workspace_bytes holds the role tickets). */
int ra_controller_batch_xcd_f32(const ra_ctrl_desc *d, const float *feat, const float *wpacked, int B,
                                float *h_last, float *ctrl_out, float *glimpse_maps, float *attn, void *ws,
                                size_t ws_bytes, int *status_dev, int xcd_offset, void *stream);

/* ------------------------------------------------------------------------------------
 * K3/K5  Gaussian attention.  Replaces modellib.get_gaussian_filter (modellib.py:581-612),
 * modellib.extract_patch (modellib.py:615-641) and their uses full_model.py:778-789
 * (read) and :810-818,:843-845 (write + canvas), :738-741 (attention box).
 * ---------------------------------------------------------------------------------- */
/* Dense filter bank, the literal operator: out[b,l,f] (modellib.py:610-611). */
int ra_gaussian_filter_f32(const float *center, const float *size, const float *lg_var, int B,
                           int L, int F, float *out, void *stream);

/* The attention resample of the decode loop (K3 extract, K5 paste, the attention box).  The filter weights
 * (modellib.get_gaussian_filter, modellib.py:581-612) are evaluated on the fly from the attention records: no
 * [L,F] tables; terms below e^-30 of a tap's peak are dropped.  The canvas may live in its own plane `canvas`
 * [B,H,W] (then channel canvas_chan of img is ignored / untouched); with canvas == NULL it is channel canvas_chan
 * of img.
 *   extract (modellib.extract_patch :615-641, full_model.py:788-789):
 *     patch[b,j,i,c] = attn_gamma_b * sum_{l,w} fy[b,l,j] * img[b,l,w,chan0+c] * fx[b,w,i]
 *     img [B,H,W,Ci] (Ci % 4 == 0), patch [B,Fh,Fw,Cp] (Cp % 4 == 0, Cp <= Ci - chan0, chan0 % 4 == 0);
 *     use_gamma = 0 skips the attn_gamma factor.
 *   paste (full_model.py:810-818,843-845):
 *     y = sigmoid(exp(y_out_lg_gamma) * (fy P fx^T) + beta);  if disable_overwrite: y *= (1 - canvas);
 *     y_out[b] = y;  canvas = max(canvas, y).  P = patch[b,:,:,pc] with channel stride Cp; y_out points at the
 *     [H,W] plane of example 0 and y_stride_b floats separate examples (a [B,T,H,W] tensor is written in place,
 *     full_model.py:855).
 *   box (full_model.py:738-741, box_model.py:479-482): box = sigmoid(box_gamma * (fy 1 fx^T) + beta). */
int ra_extract_direct_f32(const float *img, int Ci, int chan0, const float *canvas, int canvas_chan,
                          const float *attn, int B, int H, int W, int Fh, int Fw, int Cp,
                          int use_gamma, float *patch, void *stream);
/* extract + layer 0 of the attention CNN in ONE launch (full_model.py:788-795; nnlib.py:229-253 for a 3x3 layer without
 * pooling): x_patch as ra_extract_direct_f32 writes it AND y0[b,j,i,co] = relu?((sum_{ky,kx,c} w0[ky][kx][c][co] *
 * x_patch[b, j+ky-1, i+kx-1, c]) * scale[co] + shift[co]) (SAME padding; scale / shift = the folded bias + BatchNorm(eval)
 * of ra_conv_fold_bn).  w0: the layer's filter in the PACKED input's channel order, [3][3][4][Cout] floats (rows of input
 * channels the model does not feed are zero).  Workgroup (tap j, image) reduces the rows of taps j-1, j, j+1 once — their
 * bands overlap almost completely — so the conv costs no second launch and no cross-workgroup wait.  Shapes:
 * ra_extract_conv0_supported (one packed channel group Cp = 4, Fw <= 64, Cout <= 16, pool 1). */
int ra_extract_conv0_supported(int Cp, int Fh, int Fw, int Cout, int pool);
int ra_extract_conv0_f32(const float *img, int Ci, int chan0, const float *canvas, int canvas_chan,
                         const float *attn_rec, int B, int H, int W, int Fh, int Fw, int use_gamma, float *patch,
                         const float *w0, const float *scale, const float *shift, int Cout, int relu, float *y0,
                         void *stream);
/* flags (promises by the caller that let the paste touch only the attention window):
 *   RA_PASTE_Y_PREFILLED     y_out already holds sigmoid(beta) everywhere (ignored with
 *                            disable_overwrite, where untouched pixels are sigmoid(beta)*(1-canvas))
 *   RA_PASTE_CANVAS_FLOORED  canvas >= sigmoid(beta) everywhere already (true after the first
 *                            paste of a sequence that started from canvas = 0) */
#define RA_PASTE_Y_PREFILLED 1
#define RA_PASTE_CANVAS_FLOORED 2
int ra_paste_direct_f32(const float *patch, int Cp, int pc, const float *attn, int B, int H, int W,
                        int Fh, int Fw, float beta, int disable_overwrite, float *canvas, float *img,
                        int Ci, int canvas_chan, float *y_out, size_t y_stride_b, int flags,
                        void *stream);
/* ra_paste_direct_f32 on a canvas plane, with the score MLP of the same timestep
 * (full_model.py:794,821-822: s = sigmoid([h | h_core] . w + bias), h [B,K0], core [B,K1],
 * w [K0+K1], bias [1]) riding along as one extra workgroup per image: s_out[b * s_stride_b]. */
int ra_paste_score_direct_f32(const float *patch, int Cp, int pc, const float *attn, int B, int H,
                              int W, int Fh, int Fw, float beta, int disable_overwrite,
                              float *canvas, float *y_out, size_t y_stride_b, int flags,
                              const float *h, int K0, const float *core, int K1, const float *w,
                              const float *bias, float *s_out, size_t s_stride_b, void *stream);

/* Adjoints of the attention resample for the training step (ra_train.AttnExtract / AttnPaste), without the dense
 * [L,F] filter banks the reference differentiates through (modellib.py:581-641): with E(X) = fy^T X fx,
 *   RA_RESAMPLE_READ   x_patch = gamma E(img) (full_model.py:788-789): X = img [B,H,W,Cx] channels chan0 .. chan0+C,
 *                      Q = d x_patch [B,Fh,Fw,Cq]; scale = gamma [B]: out[:, 0:6] are multiplied by it;
 *                      out[:, 6] = sum E.Q = d gamma.
 *   RA_RESAMPLE_WRITE  y = sigmoid(exp(rec[8]) fy P fx^T + beta) (full_model.py:810-818): dY, Y [B,H,W] (the upstream
 *                      gradient and the forward's result), Q = P [B,Fh,Fw,Cq] channel 0; E [B,Fh,Fw,Ce] channel 0
 *                      receives dP; out[:, 6] = d y_lg_gamma.
 *   RA_RESAMPLE_BOX    box = sigmoid(rec[7] fy 1 fx^T + beta) (full_model.py:738-741): Q = NULL; div = box_gamma [B]:
 *                      out[:, 6] = d box_gamma.
 * out [B,8] = (d ctr_y, d ctr_x, d size_y, d size_x, d lg_var_y, d lg_var_x, the gamma gradient, 0) for the window
 * parameters of attn_rec [B,RA_ATTN_STRIDE].  ws: ra_resample_bwd_workspace_floats(B, Fh, C) floats (C = 1 for WRITE /
 * BOX).  Deterministic (fixed-order sums); banded like the forward kernels (weights below e^-30 of the peak dropped). */
#define RA_RESAMPLE_READ 0
#define RA_RESAMPLE_WRITE 1
#define RA_RESAMPLE_BOX 2
/* The controller of the training graph (full_model.py:668-689) as one forward and one backward launch per timestep
 * (ra_train.ControllerFn): feat [B,G,Cf]; Wg [Cf+hid, 4 hid] = the LSTM's [w_x ; w_h] with the gates side by side in the
 * order i f o u (nnlib.py:641-646), bg [4 hid]; W0 [hid,hid], W1 [hid,G] the two glimpse-MLP layers (ReLU, softmax over
 * the feature map); Wc [hid,nout] the one-layer controller MLP.  The forward writes h_last [B,hid], co [B,nout] and
 * `save` (B * ra_ctrl_train_save_floats floats: per iteration xh | gates | c | z1 | map).  The backward takes d h_last /
 * d co (either may be NULL), writes d feat [B,G,Cf] and the pre-activation gradients dpre [B,iters,4 hid], dz1
 * [B,iters,hid], dlog [B,iters,G]; the parameter gradients are the caller's GEMMs of the saved layer inputs against
 * them (one per weight per optimisation step).  ra_ctrl_train_supported: LDS / thread-count limits of the kernels. */
int ra_ctrl_train_supported(int G, int Cf, int hid, int iters, int nout);
size_t ra_ctrl_train_save_floats(int G, int Cf, int hid, int iters);
int ra_ctrl_train_fwd_f32(int B, int G, int Cf, int hid, int iters, int nout, const float *feat, const float *Wg,
                          const float *bg, const float *W0, const float *b0, const float *W1, const float *b1,
                          const float *Wc, const float *bc, float *h_last, float *co, float *save, void *stream);
int ra_ctrl_train_bwd_f32(int B, int G, int Cf, int hid, int iters, int nout, const float *feat, const float *Wg,
                          const float *W0, const float *W1, const float *Wc, const float *save, const float *dh_last,
                          const float *dco, float *dfeat, float *dpre, float *dz1, float *dlog, void *stream);
/* The same for any depth of the two MLPs (full_model.py:350-352,382-384; up to 4 layers each): the glimpse MLP is n_glimpse
 * layers [hid] * n_glimpse + [G] (ReLU on all but the last), the controller MLP n_ctrl layers [hid] + [mlp_dim] * (n_ctrl - 1) +
 * [nout].  gW / gb / cW / cb: HOST arrays of device pointers, one per layer.  save: B * ra_ctrl_train_save_floats_n floats
 * (per iteration xh | gates | c | the glimpse MLP's (n_glimpse - 1) hidden layers | map); save_c [B, (n_ctrl - 1) mlp_dim]: the
 * controller MLP's hidden layers (NULL for one layer).  The backward writes the pre-activation gradients of every dense
 * layer: dpre, dlog as above, dzg = (n_glimpse - 1) layers of [B, iters, hid] dzg_stride floats apart, dzc = (n_ctrl - 1)
 * layers of [B, mlp_dim] dzc_stride apart. */
int ra_ctrl_train_supported_n(int G, int Cf, int hid, int iters, int nout, int n_glimpse, int n_ctrl, int mlp_dim);
size_t ra_ctrl_train_save_floats_n(int G, int Cf, int hid, int iters, int n_glimpse);
int ra_ctrl_train_fwd_n_f32(int B, int G, int Cf, int hid, int iters, int nout, int n_glimpse, int n_ctrl, int mlp_dim,
                            const float *feat, const float *Wg, const float *bg, const float *const *gW,
                            const float *const *gb, const float *const *cW, const float *const *cb, float *h_last,
                            float *co, float *save, float *save_c, void *stream);
int ra_ctrl_train_bwd_n_f32(int B, int G, int Cf, int hid, int iters, int nout, int n_glimpse, int n_ctrl, int mlp_dim,
                            const float *feat, const float *Wg, const float *const *gW, const float *const *cW,
                            const float *save, const float *save_c, const float *dh_last, const float *dco, float *dfeat,
                            float *dpre, float *dlog, float *dzg, size_t dzg_stride, float *dzc, size_t dzc_stride,
                            void *stream);

size_t ra_resample_bwd_workspace_floats(int B, int Fh, int C);
int ra_resample_bwd_f32(int mode, const float *X, int Cx, int chan0, int C, const float *dY, const float *Y,
                        const float *attn_rec, const float *Q, int Cq, float *E, int Ce, int B, int H, int W, int Fh,
                        int Fw, const float *scale, int scale_stride, const float *div, int div_stride, float *ws,
                        size_t ws_floats, float *out, void *stream);
int ra_attn_box_direct_f32(const float *attn, int B, int H, int W, int Fh, int Fw, float beta,
                           float *box_out, size_t stride_b, void *stream);

/* Generic dense extract_patch with caller-supplied filters (the operator surface of
 * modellib.extract_patch for arbitrary f_y [B,H,FH], f_x [B,W,FW]; x [B,H,W,D]). */
int ra_extract_patch_dense_f32(const float *x, const float *f_y, const float *f_x, int B, int H,
                               int W, int D, int FH, int FW, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * K6  small dense layers (nnlib.mlp run_mlp, nnlib.py:476-493): out = act(x W + b).
 * x = concat(x0 [B,K0], x1 [B,K1]) (x1 nullable) — the score MLP's
 * concat([h_crnn, h_core]) of full_model.py:821-822.  W [K0+K1, N] row-major.
 * act: 0 none, 1 relu, 2 sigmoid, 3 softmax (over N), 4 tanh.
 * ---------------------------------------------------------------------------------- */
int ra_dense_f32(const float *x0, int K0, const float *x1, int K1, const float *W, const float *b,
                 int B, int N, int act, float *out, size_t out_stride_b, void *stream);

/* Elementwise helpers on the path. */
/* packed[b,h,w,:] = [x (D) | canvas=0 | d_in (Dd) | y_in (Dy) | zero pad] -> Cp channels
 * (full_model.py:239,640-661; the canvas lives as channel D of the packed image). */
int ra_pack_input_f32(const float *x, int D, const float *d_in, int Dd, const float *y_in, int Dy,
                      int B, int H, int W, int Cp, float *packed, void *stream);
/* ... and, in the same pass, zeroes the decode loop's separate canvas plane [B,H,W] (canvas_plane may be
 * NULL): one launch less per forward. */
int ra_pack_input_plane_f32(const float *x, int D, const float *d_in, int Dd, const float *y_in, int Dy,
                            int B, int H, int W, int Cp, float *packed, float *canvas_plane, void *stream);
/* canvas channel update used by box_model (box_model.py:500-504):
 * canvas = max(canvas, ysel - ysel*noise). ysel,noise [B,H,W]. */
int ra_canvas_max_f32(float *img, int Ci, int canvas_chan, const float *ysel, const float *noise,
                      int B, int H, int W, void *stream);

/* Stand-alone eval BatchNorm / affine (nnlib.batch_norm with phase_train=False,
 * nnlib.py:113-119): y[p,c] = relu?(x[p,c]*scale[c] + shift[c]), p < npix. */
int ra_affine_act_f32(const float *x, const float *scale, const float *shift, size_t npix, int C,
                      int relu, float *y, void *stream);
/* Stand-alone nnlib.max_pool (nnlib.py:15-25): ksize = stride = ratio, 'SAME' (-inf pad).
 * x [B,H,W,C] -> y [B,ceil(H/ratio),ceil(W/ratio),C]. */
int ra_max_pool_f32(const float *x, int B, int H, int W, int C, int ratio, float *y, void *stream);

/* ------------------------------------------------------------------------------------
 * Loss / statistics head of the training graph, forward only (full_model.py:913-1097).
 *
 * ra_pair_stats_f32 — modellib.f_iou(a, b, pairwise=True) (modellib.py:124-155),
 *   f_iou(a > 0.5, b, pairwise=True) (full_model.py:1064-1065) and f_dice(a > 0.5, b,
 *   pairwise=True) (modellib.py:71-104) in ONE pass over a [B,N,H*W] and b [B,M,H*W]
 *   (N, M <= 32, H*W % 4 == 0, 16-byte aligned).  Outputs (each nullable): iou_soft,
 *   iou_hard, dice_hard [B,N,M]; sum_a [B,N], sum_b [B,M] = per-instance pixel sums; inter
 *   [B,N,M] = the raw intersections sum(a*b) (exact integers for binary masks up to 2^24
 *   pixels); sum_a_hard [B,N] = pixel counts of (a > 0.5).
 *   ws: device scratch of ra_pair_stats_workspace_floats(B, H*W) floats.
 * ra_gt_box_f32 — modellib.get_gt_box (modellib.py:663-701) with center_shift_ratio 0:
 *   params [B,T,8] = top_left (y,x), bot_right (y,x) as returned by the reference, then the
 *   padded rectangle (tl_y, tl_x, br_y, br_x) the mask is filled with; box (nullable)
 *   [B,T,H,W] = the filled rectangle mask.  ws: ra_gt_box_workspace_floats(B, T) device floats.
 * ra_segm_match_f32 — modellib.f_segm_match (modellib.py:382-415): mask with s_gt, quantise
 *   to 1e-6, + 1e-5, Hungarian (device), mask again.  status: int[B] as ra_hungarian_f32_dev.
 * ra_loss_stats_f32 — every scalar of full_model.py:941-1081 for box_loss_fn = 'iou' and
 *   segm_loss_fn 0 = 'iou' / 1 = 'wt_cov': out[RA_STAT_COUNT] indexed by RA_STAT_*.
 *   sum_gt [B,T] = per-instance sums of y_gt (f_coverage_weight, modellib.py:277-289);
 *   ws: ra_loss_stats_workspace_floats(B) device floats.
 * ---------------------------------------------------------------------------------- */
enum {
  RA_STAT_LOSS = 0, RA_STAT_BOX_LOSS, RA_STAT_SEGM_LOSS, RA_STAT_CONF_LOSS, RA_STAT_IOU_SOFT,
  RA_STAT_IOU_SOFT_BOX, RA_STAT_WT_COV_SOFT, RA_STAT_UNWT_COV_SOFT, RA_STAT_IOU_HARD,
  RA_STAT_WT_COV_HARD, RA_STAT_UNWT_COV_HARD, RA_STAT_DICE, RA_STAT_COUNT_ACC, RA_STAT_DIC,
  RA_STAT_DIC_ABS, RA_STAT_COUNT
};
size_t ra_pair_stats_workspace_floats(int B, int HW);
int ra_pair_stats_f32(const float *a, const float *b, int B, int N, int M, int HW, float *ws,
                      size_t ws_floats, float *iou_soft, float *iou_hard, float *dice_hard,
                      float *sum_a, float *sum_b, float *inter, float *sum_a_hard, void *stream);
/* ra_pair_stats_f32 with a's planes addressed through strides (in floats, multiples of 4): a[img][row] starts at
 * img * a_img + row * a_row.  The training step keeps the masks of its T timesteps timestep-major ([T,B,H,W]: a_img = H*W,
 * a_row = B*H*W) — the pairwise IoU reads them in place instead of through a transposed copy. */
int ra_pair_stats_strided_f32(const float *a, size_t a_img, size_t a_row, const float *b, int B, int N, int M, int HW, float *ws,
                              size_t ws_floats, float *iou_soft, float *iou_hard, float *dice_hard, float *sum_a, float *sum_b,
                              float *inter, float *sum_a_hard, void *stream);
/* Soft IoU (modellib.f_iou, modellib.py:124-155) of one map per image, box [B,H,W], against the T rectangles
 * ra_gt_box_f32 fills (params [B,T,8], fields 4..7): iou [B,T] — the row the training graph takes per timestep
 * (full_model.py:744-758) without reading the T rectangle planes. */
size_t ra_box_iou_rects_workspace_floats(int B);
int ra_box_iou_rects_f32(const float *box, const float *params, int B, int T, int H, int W, float *ws,
                         size_t ws_floats, float *iou, void *stream);
size_t ra_gt_box_workspace_floats(int B, int T);
int ra_gt_box_f32(const float *y_gt, int B, int T, int H, int W, float padding_ratio,
                  float min_padding, float *ws, size_t ws_floats, float *params, float *box,
                  void *stream);

/*
 * ra_knob_setup_f32 — the training step's noisy ground-truth attention and knob masks from the partials ra_gt_box_f32
 * left in ITS workspace (same B, T; the call must come after it on the stream): modellib.get_gt_attn with per-instance
 * padding ratio pad [B,T] and centre shift [B,T,2] (full_model.py:567-577; modellib.py:688-701) -> ctr, size [B,T,2]
 * (an empty instance: top-left 0, bottom-right 2 * min_padding), and knob_box / knob_segm [B,T] = (u <= min(sched[k] *
 * scale_t, 1)), scale_t = 1 + log(1 + 3 t) with timescale (full_model.py:596-625); sched = 2 device floats (the
 * probabilities of this step).  One launch on B T numbers; float32 operations in the element-wise form's order.
 */
int ra_knob_setup_f32(const float *gt_box_ws, int B, int T, const float *pad, const float *shift, const float *u_box,
                      const float *u_segm, const float *sched, float min_padding, int timescale, float *ctr, float *size,
                      float *knob_box, float *knob_segm, void *stream);
size_t ra_segm_match_workspace_bytes(int B, int N);
int ra_segm_match_f32(const float *iou, const float *s_gt, int B, int N, void *ws, size_t ws_bytes,
                      float *match, int *status, void *stream);
/* ... with the B problems solved on HOST cores (ra_hungarian_f32 on up to `threads` threads) as a host function of the stream — a host
 * node of the graph when the stream is capturing (round 6): precondition kernel -> D2H copy into `pinned` -> host solve -> H2D ->
 * re-mask.  The device solver's launch lasts as long as its slowest problem (1.7 ms for a cfg4 step's 16 x 16 problems, on the
 * step's critical path); a host core needs ~0.5 ms per problem and the problems run side by side.  w_dev: B N N floats of device
 * scratch; pinned: page-locked host memory of ra_segm_match_host_block_bytes() (16-byte aligned) that carries the host function's
 * arguments and staging — it must stay allocated, and unused by anyone else, as long as a graph that captured the call lives.
 * match / status: device outputs as ra_segm_match_f32's (status[b]: ra_hungarian_f32's return code for problem b). */
size_t ra_segm_match_host_block_bytes(int B, int N);
int ra_segm_match_host_f32(const float *iou, const float *s_gt, int B, int N, float *w_dev, void *pinned, size_t pinned_bytes,
                           int threads, float *match, int *status, void *stream);
int ra_loss_stats_f32(const float *iou_soft, const float *iou_hard, const float *dice,
                      const float *match_real, const float *iou_box, const float *match_box,
                      const float *s_out, const float *s_gt, const float *sum_gt, int B, int T,
                      int fixed_order, int segm_loss_fn, float loss_mix_ratio, float *ws,
                      size_t ws_floats, float *out, void *stream);
size_t ra_loss_stats_workspace_floats(int B);

/* ------------------------------------------------------------------------------------
 * Evaluation post-processing and metrics (utils/postprocess.py, analysis.py:314-760).
 *
 * ra_postprocess_f32 — apply_confidence (postprocess.py:15-29), apply_one_label (:32-52),
 *   apply_threshold (:5-12) and, with fg [B,H,W] != NULL, mask_foreground (:139-147) in one
 *   pass: y_bin[b,t,p] = (t == argmax_t' y*s) && (max_t' y*s > thresh) [* fg]; s_hard
 *   (nullable) [B,T] = s_out > 0.5; union_out (nullable) [B,H,W] = max_t y_bin.
 * ra_union_f32 — y.max(axis=0) per image (analysis.py:547,570).
 * ra_remove_tiny_f32 — remove_tiny (postprocess.py:109-136): instance planes whose size
 *   (sizes [B,T], e.g. sum_a of ra_pair_stats_f32) is <= threshold are zeroed, conf too.
 * ra_eval_metrics_f32 — from inter [B,T,T] / sum_a / sum_b of the BINARY masks (outputs = a,
 *   ground truth = b) and s_gt: iou_pairwise (nullable) [B,T,T] (analysis.py:314-334); stats
 *   [B,RA_EVAL_COUNT]; inst (nullable) [B,RA_EVALI_COUNT,T].  fg_inter/fg_a/fg_b [B]
 *   (nullable together) = |union_a & union_b|, |union_a|, |union_b|; a_in_fgb [B,T] =
 *   |a_i & union_b|, b_in_fga [B,T] = |b_j & union_a| (both nullable).
 * ---------------------------------------------------------------------------------- */
enum {
  RA_EVAL_SBD = 0, RA_EVAL_WT_COV, RA_EVAL_UNWT_COV, RA_EVAL_FG_IOU, RA_EVAL_FG_DICE, RA_EVAL_FP,
  RA_EVAL_FN, RA_EVAL_COUNT_ACC, RA_EVAL_COUNT_MSE, RA_EVAL_DIC, RA_EVAL_DIC_ABS, RA_EVAL_NUM_OBJ,
  RA_EVAL_COUNT_OUT, RA_EVAL_COUNT
};
enum {
  RA_EVALI_OBJ_PR = 0, RA_EVALI_OBJ_RE, RA_EVALI_PIX_PR, RA_EVALI_PIX_RE, RA_EVALI_HAS_OUT,
  RA_EVALI_IS_GT, RA_EVALI_COUNT
};
int ra_postprocess_f32(const float *y_out, const float *s_out, int B, int T, int H, int W,
                       float thresh, const float *fg, float *y_bin, float *s_hard,
                       float *union_out, void *stream);
int ra_union_f32(const float *y, int B, int T, int HW, float *union_out, void *stream);
/* The evaluator's cv2 steps as plain kernels on N = B*T planes (full_model_eval.py:112-118):
 * ra_dilate_f32 — morph_single (postprocess.py:63-72): cv2.dilate(plane, ones(2 radius + 1)^2), out-of-image pixels ignored.
 * ra_resize_linear_f32 + ra_bilateral5_f32 — upsample_single (postprocess.py:93-106): cv2.resize(..., INTER_LINEAR) (pixel
 *   centres aligned, float32 two-tap weights) and cv2.bilateralFilter(b, 5, sigma_color, sigma_space) (circular radius-2
 *   neighbourhood, BORDER_REFLECT_101, the exact exponential where cv2 interpolates a table).  out != y. */
int ra_dilate_f32(const float *y, int N, int H, int W, int radius, float *out, void *stream);
int ra_resize_linear_f32(const float *y, int N, int Hs, int Ws, int H, int W, float *out, void *stream);
int ra_bilateral5_f32(const float *y, int N, int H, int W, float sigma_color, float sigma_space, float *out, void *stream);
int ra_remove_tiny_f32(float *y_bin, const float *sizes, float *conf, int B, int T, int HW,
                       float threshold, void *stream);
int ra_eval_metrics_f32(const float *inter, const float *sum_a, const float *sum_b,
                        const float *s_gt, const float *fg_inter, const float *fg_a,
                        const float *fg_b, const float *a_in_fgb, const float *b_in_fga, int B,
                        int T, float *iou_pairwise, float *stats, float *inst, void *stream);

/* In-graph augmentation, image_ops.random_transformation (image_ops.py:9-113) for given random
 * draws: zero-pad by `padding`, crop H x W at (off_y, off_x) in [0, 2*padding] (one offset per
 * batch), reverse along H (flip_v) / W (flip_h), then transpose H <-> W (needs H == W).
 * x, out: [N,H,W,C]; instance masks [B,T,H,W] go in as N = B*T, C = 1.  out != x.
 * The colour jitter (:99-103) is ra_colour_jitter_f32 below. */
int ra_random_transform_f32(const float *x, int N, int H, int W, int C, int padding, int off_y,
                            int off_x, int flip_v, int flip_h, int transpose, float *out,
                            void *stream);
/* The colour jitter of the same augmentation (image_ops.py:99-103,116-180: random_hue(0.1), random_saturation(0.9, 1.1),
 * tf.image.random_brightness(0.1), tf.image.random_contrast(0.9, 1.1)) for given draws, one of each per batch: RGB -> HSV,
 * hue = (hue + hue_delta + 1) mod 1, saturation = clip(saturation * factor, 0, 1), HSV -> RGB, + brightness_delta (no
 * clipping: float images), then (x - mean) * contrast_factor + mean with the mean of each image and channel.  TensorFlow's
 * kernels restated from their published algorithm; they cannot be run here.  x, out: [B,HW,3] (out may
 * be x); ws: ra_colour_jitter_workspace_floats(B) floats. */
size_t ra_colour_jitter_workspace_floats(int B);
int ra_colour_jitter_f32(const float *x, int B, int HW, float hue_delta, float saturation_factor, float brightness_delta,
                         float contrast_factor, float *ws, size_t ws_floats, float *out, void *stream);

/* out[b,p] = sum_t w[b,t] * y[b,t,p]: the ground-truth instance box_model's greedy match picks
 * (box_model.py:487-499).  Terms with w == 0 are skipped (0 * x is not evaluated). */
int ra_weighted_sum_f32(const float *w, const float *y, int B, int T, int HW, float *out,
                        void *stream);

/* ------------------------------------------------------------------------------------
 * Training step (full_model.py:1039-1057).
 * ra_adam_step_f32 — gradient clip + Adam on one flat float32 bucket of n parameters:
 *   g = clip(grads * grad_scale + wd_coef * params, -clip, clip)   (wd_coef nullable; the
 *       wd * l2_loss(w) terms of nnlib.py:59-61 belong to total_loss; grad_scale = 1 / world
 *       after the data-parallel RCCL sum),
 *   m, v, params updated as tf.train.AdamOptimizer does (epsilon outside the bias correction);
 *   lr_t = learn_rate * sqrt(1 - beta2^t) / (1 - beta1^t) is computed by the caller.
 * ---------------------------------------------------------------------------------- */
int ra_adam_step_f32(float *params, const float *grads, float *m, float *v, const float *wd_coef,
                     size_t n, float lr_t, float beta1, float beta2, float eps, float clip,
                     float grad_scale, void *stream);
/* The same update, applied ONLY IF none of the step's device status words says "failed": the first n_solver words are
 * Hungarian-solver return codes (negative = the reference's LOG(FATAL) cases, hungarian.cc:126,148,158,187,448; 1 = the
 * outer cap, where the reference carries on with the partial matching, :363-377 — not a failure), the n_other words behind
 * them fail when non-zero (the split controller's time-out word, another rank's failure flag).  A training loop that reads
 * those words one step late (no host sync per step) can then never have applied a failed step's gradients.  status
 * nullable with both counts 0 (= ra_adam_step_f32). */
int ra_adam_step_guarded_f32(float *params, const float *grads, float *m, float *v, const float *wd_coef,
                             size_t n, float lr_t, float beta1, float beta2, float eps, float clip,
                             float grad_scale, const int *status, int n_solver, int n_other, void *stream);

/* Train-mode layer pieces of nnlib.cnn / nnlib.dcnn (nnlib.py:229-253,362-400 with
 * phase_train = True).  The convolution itself is ra_conv3x3_f32 with scale = 1, shift = bias,
 * no ReLU, no pool (u = conv(x, w) + b); then
 *   ra_bn_moments_f32       mean[c], var[c] = tf.nn.moments(u, [0,1,2]) (nnlib.py:98; biased,
 *                           two passes, fixed summation order); u [npix, C], C <= 256.
 *   ra_bn_act_pool_f32      y = max_pool(relu?(gamma (u - mean) rsqrt(var + eps) + beta))
 *                           (nnlib.py:111-119,250-253); mean/var/gamma/beta nullable together
 *                           (use_bn False: y = pool(relu(u))).
 *   ra_bn_act_pool_bwd_f32  given dy [B,H/pool,W/pool,C]: dbeta, dgamma [C] and
 *                           du = gamma rstd (dv - mean(dv) - xhat mean(dv xhat)), dv = dy routed to
 *                           the first maximum of each pool window and masked by the ReLU — the
 *                           batch statistics are differentiated through, as tf.gradients does.
 *   ws: ra_bn_workspace_floats(C) device floats.
 * ra_conv_pack_weights_dev  ra_conv_pack_weights on device pointers (weights change every step).
 * ra_conv3x3_wgrad_f32      dWf[ky][kx][ci][co] = sum X[b,y+ky-1,x+kx-1,ci] dU[b,y,x,co] for the
 *                           SAME conv that ran (X zero-stuffed when upsample = 1), db[co] = sum dU
 *                           (db nullable); f32 MFMA with pixels as the K dimension; Cout <= 64;
 *                           ws: ra_conv3x3_wgrad_workspace_floats() floats.
 *   Backward-data is ra_conv3x3_f32 itself on the flipped / in-out-swapped packing
 *   (RA_CONV_TRANSPOSED for a cnn layer, plain packing of the [3,3,out,in] filter for a dcnn
 *   layer), followed for stride-2 layers by
 * ra_subsample_odd_f32      y[b,i,j,:] = x[b,2i+1,2j+1,:], x [B,2H,2W,C] (adjoint of zero-stuffing).
 * ra_weighted_sum_multi_f32 out[b,n,:] = sum_t w[b,n,t] y[b,t,:] + bias[b,n]: with the coefficients
 *                           of modellib.f_iou's quotient rule this is d loss / d y_out. */
size_t ra_bn_workspace_floats(int C);
int ra_bn_moments_f32(const float *u, size_t npix, int C, float *ws, size_t ws_floats, float *mean,
                      float *var, void *stream);
int ra_bn_act_pool_f32(const float *u, const float *mean, const float *var, const float *gamma,
                       const float *beta, float eps, int relu, int pool, int B, int H, int W, int C,
                       float *y, void *stream);
int ra_bn_act_pool_bwd_f32(const float *u, const float *dy, const float *mean, const float *var,
                           const float *gamma, const float *beta, float eps, int relu, int pool,
                           int B, int H, int W, int C, float *ws, size_t ws_floats, float *dgamma,
                           float *dbeta, float *du, void *stream);
int ra_conv_pack_weights_dev(const float *w, int Cin_w, int Cout, int Cin, const int *chan_map,
                             int flags, float *out, void *stream);
size_t ra_conv3x3_wgrad_workspace_floats(int Cin, int Cout, int B, int H, int W);
int ra_conv3x3_wgrad_f32(const float *x, int Cin, int B, int Hs, int Ws, int upsample,
                         const float *du, int Cout, float *ws, size_t ws_floats, float *dw,
                         float *db, void *stream);
/* The accumulating forms add the layer's parameter gradients straight into the trainer's flat
 * gradient bucket (one writer per element), replacing ~2 500 small `grad += g` launches of a step:
 *   ra_conv3x3_wgrad_acc_f32    gw += dW in the REFERENCE layout of the filter that ran — [3,3,cin_w,Cout],
 *                               or [3,3,Cout,cin_w] with flipped taps when transposed (nnlib.dcnn's
 *                               filters, nnlib.py:362-400) — chan_map (device, nullable) sends packed kernel
 *                               channel c to filter row chan_map[c] (-1: padding); gb += db (nullable).
 *   ra_bn_act_pool_bwd_acc_f32  ra_bn_act_pool_bwd_f32 + acc_gamma += dgamma, acc_beta += dbeta (nullable). */
int ra_conv3x3_wgrad_acc_f32(const float *x, int Cin, int B, int Hs, int Ws, int upsample,
                             const float *du, int Cout, float *ws, size_t ws_floats, const int *chan_map,
                             int cin_w, int transposed, float *gw, float *gb, void *stream);
/* The filter gradient with bf16 operands (x and du rounded to bf16 from LDS, float32 accumulation over the pixels
 * and float32 partial sums): the mixed-precision counterparts of the two entries above. */
int ra_conv3x3_wgrad_bf16ops_f32(const float *x, int Cin, int B, int Hs, int Ws, int upsample,
                                 const float *du, int Cout, float *ws, size_t ws_floats, float *dw,
                                 float *db, void *stream);
int ra_conv3x3_wgrad_acc_bf16ops_f32(const float *x, int Cin, int B, int Hs, int Ws, int upsample,
                                     const float *du, int Cout, float *ws, size_t ws_floats,
                                     const int *chan_map, int cin_w, int transposed, float *gw, float *gb,
                                     void *stream);
/* A layer whose filter is shared by the T timesteps of a training step (every nnlib.cnn / dcnn layer of full_model:
 * one set of weights, full_model.py:455-535) sums its T filter gradients — in ONE pass over the images of all T calls: ra_conv3x3_wgrad_multi_acc_f32 is ra_conv3x3_wgrad_acc_f32 on
 * nseg * Bseg images read through two device tables of nseg pointers (x [Bseg,Hs,Ws,Cin] and du [Bseg,H,W,Cout] of
 * each call; workspace for B = nseg * Bseg).  ra_ptr_table writes up to 64 device pointers given as a HOST array into
 * a device table by a kernel launch, so that the tables can be built inside a captured HIP graph. */
int ra_ptr_table(const void *const *host_ptrs, int n, void **dev_table, void *stream);
int ra_conv3x3_wgrad_multi_acc_f32(const void *const *xtab, const void *const *dutab, int nseg, int Cin, int Bseg,
                                   int Hs, int Ws, int upsample, int Cout, float *ws, size_t ws_floats,
                                   const int *chan_map, int cin_w, int transposed, float *gw, float *gb,
                                   int bf16_operands, void *stream);
int ra_bn_act_pool_bwd_acc_f32(const float *u, const float *dy, const float *mean, const float *var,
                               const float *gamma, const float *beta, float eps, int relu, int pool,
                               int B, int H, int W, int C, float *ws, size_t ws_floats, float *dgamma,
                               float *dbeta, float *du, float *acc_gamma, float *acc_beta, void *stream);
/* G calls of one layer (its G timesteps: per-timestep statistics and parameters, nnlib.py:121-127) stacked along the
 * batch and differentiated in one reduce / final / dx triple: u [G*B,H,W,C], dy [G*B,H/pool,W/pool,C], du like u;
 * tabs = a DEVICE table of 6 G pointers {mean, var, gamma, beta, gradient-bucket gamma, gradient-bucket beta}[G]
 * (the last two are added to); dgamma / dbeta [G,C]; ws of G * ra_bn_workspace_floats(C) floats.  C % 4 == 0 and
 * C / 4 a power of two <= 64 (RA_E_SHAPE otherwise: call the per-group entry). */
int ra_bn_act_pool_bwd_grouped_f32(const float *u, const float *dy, const void *const *tabs, int G, float eps,
                                   int relu, int pool, int B, int H, int W, int C, float *ws, size_t ws_floats,
                                   float *dgamma, float *dbeta, float *du, void *stream);
/* The same backward in two launches groups, for data-parallel training on WHOLE-batch statistics (nnlib.py:98
 * takes the moments over the whole batch): _reduce writes this rank's sums dbeta / dgamma (and adds them to
 * acc_*, may be NULL); the caller all-reduces the 2C sums; _dx finishes with the summed vectors and the
 * global pixel count n_total (mean / var are then the whole-batch moments too). */
int ra_bn_act_pool_bwd_reduce_f32(const float *u, const float *dy, const float *mean, const float *var,
                                  const float *gamma, const float *beta, float eps, int relu, int pool, int B, int H,
                                  int W, int C, float *ws, size_t ws_floats, float *dgamma, float *dbeta,
                                  float *acc_gamma, float *acc_beta, void *stream);
int ra_bn_act_pool_bwd_dx_f32(const float *u, const float *dy, const float *mean, const float *var,
                              const float *gamma, const float *beta, const float *dgamma_sum, const float *dbeta_sum,
                              double n_total, float eps, int relu, int pool, int B, int H, int W, int C, float *du,
                              void *stream);
/* The pointwise half of the controller's LSTM cell (nnlib.py:641-646; the GEMM half is a library call):
 * pre [B][4*hid] = gate pre-activations in the order (i, f, o, u), c_prev [B][hid]:
 *   c = sigm(f) c_prev + sigm(i) tanh(u),  h = sigm(o) tanh(c);  act [B][4*hid] keeps the gate values.
 * Backward: dh / dc nullable (a gradient that does not exist is zero) -> dpre [B][4*hid], dc_prev. */
int ra_lstm_cell_f32(const float *pre, const float *c_prev, int B, int hid, float *h, float *c,
                     float *act, void *stream);
int ra_lstm_cell_bwd_f32(const float *act, const float *c_prev, const float *c, const float *dh,
                         const float *dc, int B, int hid, float *dpre, float *dc_prev, void *stream);
/* modellib.get_gaussian_filter (modellib.py:581-612) and its adjoint for the training graph:
 *   out[b][l][j] = N(l; mu_j, exp(lg_var[b])),  mu_j = ctr[b] + (size[b] + 1) / NF * (j - (NF - 1) / 2);
 *   backward: g [B][L][NF] -> dctr, dsize, dlg_var [B] (the bank is recomputed, not stored). */
int ra_gauss_filter_f32(const float *ctr, const float *size, const float *lg_var, int B, int L, int NF,
                        float *out, void *stream);
int ra_gauss_filter_bwd_f32(const float *ctr, const float *size, const float *lg_var, const float *g,
                            int B, int L, int NF, float *dctr, float *dsize, float *dlg_var,
                            void *stream);
/* The same two with element strides between consecutive images (one column of a [B,2] tensor is handed in
 * as it lies; stride_lg_var may be 0: one variance for all images); stride_grad: of dctr / dsize / dlg_var. */
int ra_gauss_filter_strided_f32(const float *ctr, const float *size, const float *lg_var, int stride_ctr,
                                int stride_size, int stride_lg_var, int B, int L, int NF, float *out,
                                void *stream);
int ra_gauss_filter_strided_bwd_f32(const float *ctr, const float *size, const float *lg_var,
                                    int stride_ctr, int stride_size, int stride_lg_var, const float *g,
                                    int B, int L, int NF, float *dctr, float *dsize, float *dlg_var,
                                    int stride_grad, void *stream);
/* The scalar head of the training loss with box_loss_fn = segm_loss_fn = 'iou' (full_model.py:913-1035) in one launch each
 * way.  iou_s / iou_b: the pairwise soft IoU of the masks / attention boxes against the ground truth [B,T,T] (pred, gt);
 * m_s / m_b: their matchings (f_segm_match); s_out [B,T]: the scores.
 *   ra_loss_head_f32      pieces[0..5] = loss, box_loss, segm_loss, conf_loss, iou_soft, iou_soft_box, where
 *                         iou = mean_b sum(iou * m) / max(sum m, 1), conf = sum(-ms log(cummin s + 1e-5) - (1 - ms)
 *                         log(1 - reverse-cummax s + 1e-5)) / B / T with ms = sum_gt m_s, loss = -iou_box - iou_soft +
 *                         mix * conf.
 *   ra_loss_head_bwd_f32  g (device scalar, NULL = 1): d loss upstream.  Writes the coefficients of the two pairwise-IoU
 *                         adjoints, c1 [B,T,T] / c0 [B,T] for ra_weighted_sum_multi_f32 (from inter / sum_a / sum_b of
 *                         ra_pair_stats_f32, HW = pixels per plane), and d s_out [B,T] (the cumulative extrema route their
 *                         gradient to the positions torch.cummin / cummax report). */
int ra_loss_head_f32(const float *iou_s, const float *iou_b, const float *m_s, const float *m_b, const float *s_out, int B, int T,
                     float mix, float *pieces, void *stream);
int ra_loss_head_bwd_f32(const float *g, const float *m_s, const float *m_b, const float *s_out, const float *inter_s,
                         const float *sum_a_s, const float *sum_b_s, const float *inter_b, const float *sum_a_b,
                         const float *sum_b_b, int B, int T, int HW, float mix, float *c1_s, float *c0_s, float *c1_b,
                         float *c0_b, float *d_s_out, void *stream);
/* The attention head of the training graph in one launch each way (it is scalar math on nine numbers per image):
 *   ra_attn_head_f32   ctrl_out [B][stride >= 9] -> out [B][16] = cn[2] ls[2] ctr[2] size[2] lg_var[2] attn_gamma
 *                      box_gamma y_lg_gamma (full_model.py:702-722, modellib.py:752-764,812-825); flags: 1 squash
 *                      (tanh / -softplus), 2 fixed_var, 4 dynamic_var, 8 fixed_gamma.
 *   ra_attn_head_bwd_f32  gradients of those fields (dense [B,2] / [B], each nullable = zero) -> d ctrl_out [B][9].
 *   ra_knob_mix_f32    the ground-truth knob on the window (full_model.py:744-773): p2 = knob m + (1 - knob) p for
 *                      centre and size, m = sum_t match[b][t] gt[b][t][:]; knob [B] with element stride knob_stride,
 *                      ctr / size rows row_stride floats apart (fields of the head's record); outputs dense [B,2].
 *   ra_knob_mix_bwd_f32   d p = (1 - knob) g (g nullable).
 *   ra_attn_head_rec_f32 / ra_knob_mix_rec_f32   the same, also writing the window as the resample kernels' attention
 *                      record [B][RA_ATTN_STRIDE] (ctr, size, lg_var, attn_gamma, box_gamma, y_lg_gamma, zeros; the knob
 *                      form copies attn_rec with the mixed centre and size): every resample kernel reads only its own
 *                      gamma column, so ONE record serves the box, the extract and the paste of a timestep. */
int ra_attn_head_f32(const float *ctrl_out, int stride, int B, int H, int W, int Fh, int Fw, int flags,
                     float *out, void *stream);
int ra_attn_head_bwd_f32(const float *ctrl_out, int stride, const float *out, const float *g_cn,
                         const float *g_ls, const float *g_ctr, const float *g_size, const float *g_lg_var,
                         const float *g_attn_gamma, const float *g_box_gamma, const float *g_y_lg_gamma, int B,
                         int H, int W, int flags, float *d_ctrl_out, void *stream);
int ra_knob_mix_f32(const float *ctr, const float *size, const float *match, const float *ctr_gt,
                    const float *size_gt, const float *knob, int knob_stride, int row_stride, int B, int T,
                    float *ctr2, float *size2, void *stream);
int ra_knob_mix_bwd_f32(const float *g_ctr2, const float *g_size2, const float *knob, int knob_stride, int B,
                        float *d_ctr, float *d_size, void *stream);
int ra_attn_head_rec_f32(const float *ctrl_out, int stride, int B, int H, int W, int Fh, int Fw, int flags,
                         float *out, float *attn_rec, void *stream);
int ra_knob_mix_rec_f32(const float *ctr, const float *size, const float *match, const float *ctr_gt,
                        const float *size_gt, const float *knob, int knob_stride, int row_stride, int B, int T,
                        float *ctr2, float *size2, const float *attn_rec, float *attn_rec2, void *stream);
int ra_subsample_odd_f32(const float *x, int B, int H, int W, int C, float *y, void *stream);
/* The canvas update of a training timestep (full_model.py:826-848, stop_canvas_grad) and the next timestep's packed
 * controller-CNN input in one launch: inp_next = inp_prev [B,HW,C] with channel canvas_chan replaced by
 * max(y_c, canvas), y_c = y [B,HW], or with the ground-truth knob (match [B,T] != NULL)
 * knob[b] * (g - g * noise) + (1 - knob[b]) * y,  g = sum_t match[b,t] y_gt[b,t,:]  (noise [B,HW] nullable;
 * knob read at stride knob_stride).  No gradient flows through it. */
int ra_canvas_step_f32(const float *inp_prev, int C, int canvas_chan, int B, int HW, const float *y,
                       const float *match, const float *y_gt, int T, const float *noise, const float *knob,
                       int knob_stride, float *inp_next, void *stream);
int ra_weighted_sum_multi_f32(const float *w, const float *bias, const float *y, int B, int N, int T,
                              int HW, float *out, void *stream);
/* ... with out[b][n] written at b * o_img + n * o_row (floats, multiples of 4): the gradient of timestep-major masks. */
int ra_weighted_sum_multi_strided_f32(const float *w, const float *bias, const float *y, int B, int N, int T, int HW,
                                      float *out, size_t o_img, size_t o_row, void *stream);

/* Dynamic tile tickets for the persistent controller-CNN launches (no reference counterpart: nnlib.cnn, nnlib.py:229-253, is
 * one TF op per layer and knows no tiles).  ra_conv_pair_cached_f32, ra_conv_pair_wino_f32, ra_conv_wino_f32 and
 * ra_conv_split_f32 run persistent grids; by default a workgroup walks a fixed list of tiles, which is fastest when the
 * launch owns the GPU and up to 2x slower when other work (another batch's controller in the decode pipeline, another
 * process) keeps some of its workgroups from starting on time.  Between ra_tile_tickets_bind(scratch, slots) and
 * ra_tile_tickets_bind(NULL, 0) every such launch issued by the CALLING THREAD takes the next slot(s) of `scratch` (one per
 * output-channel slice) and its workgroups DRAW their tiles from per-XCD pools in it, so that a late workgroup costs only
 * its share.  scratch: device memory, 128-byte aligned, slots * ra_tile_tickets_slot_bytes() bytes, ZERO when the first of
 * those launches executes (e.g. ra_fill_f32 on the same stream), used by launches of ONE stream only, and re-zeroed before the
 * slots are handed out again (the decode engine binds at the start of a forward: one fill per forward, also inside its
 * captured graph).  Launches that find no slot left, or whose grid is too small to draw from all eight pools, keep the
 * static walk.  Returns 1 if bound, 0 if unbound (NULL, or a device whose workgroups do not report exactly the XCC ids
 * 0..7: the static walk stays), negative on error.  RA_TILE_TICKETS=0 disables it. */
int ra_tile_tickets_bind(void *scratch, int slots);
int ra_tile_tickets_slot_bytes(void);

/* p[0..n) = value (p 16-byte aligned): the canvas reset `canvas = zeros` (full_model.py:239) and
 * the sigmoid(beta) prefill of y_out behind RA_PASTE_Y_PREFILLED, as a library launch so that the
 * captured forward holds no framework kernel. */
int ra_fill_f32(float *p, size_t n, float value, void *stream);
/* out[i] = idx[i] >= 0 ? src[idx[i]] : 0 for i < n (device pointers).  The training step re-packs the controller's
 * weights for the decode loop's kernels once per optimisation step without leaving the device: the host packers
 * (ra_ctrl_split_pack_weights ...) only move values, so packing arrays of flat-bucket positions once gives the index map. */
int ra_gather_f32(const float *src, const int *idx, size_t n, float *out, void *stream);
/* C [M, N] += A^T B (and bias [N] += the column sums of B) over the rows k < K of the row-major A [K, M] (row stride lda)
 * and B [K, N] (ldb): the parameter gradients tf.gradients forms for the controller's dense layers (nnlib.mlp /
 * nnlib.lstm, full_model.py:668-689) from a step's layer inputs (A) and pre-activation gradients (B) of every
 * (timestep, image, glimpse iteration) — K is a few hundred rows, the output wide.  k_period > 0 skips the rows with
 * k % k_period == k_period - 1 (the last glimpse iteration of an image feeds no next one).  seg == NULL: plain C (row stride
 * ldc), bias or NULL.  seg != NULL: a DEVICE table of 3 * ceil(N / col_block) pointers, [rows < row_split | rows >=
 * row_split | bias] x column block (NULL entries skipped): every block is its own dense [rows, col_block] tensor — the
 * LSTM's eight weight and four bias parameters of the flat gradient bucket; row_split % 16 == col_block % 16 == 0. */
int ra_gemm_tn_acc_f32(const float *A, int lda, const float *B, int ldb, int K, int M, int N, int k_period, float *C, int ldc,
                       float *bias, const void *const *seg, int row_split, int col_block, void *stream);
/* modellib.f_greedy_match with matched == 0 (modellib.py:365-379; box_model.py:487-498):
 * match[b,t] = (score[b,t] == max_t score[b,:]) / #maxima.  score, match [B,T]. */
int ra_greedy_match_f32(const float *score, int B, int T, float *match, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RECATTEND_H_ */
