/*
 * sqair_hip.h — C-ABI of the MI355X-native SQAIR Discover/Propagate hot path (libsqair_hip.so).
 *
 * The reference (akosiorek/sqair) has no native/FFI layer: its operator API for this path is the
 * Python factory  load(img, coords, num, mean_img, debug) -> Model
 *   (reference: sqair/configs/mlp_mnist_model.py:74-150, called through
 *    sqair/experiment_tools.py:147-157 from sqair/scripts/experiment.py:118 and scripts/eval.py:138)
 * and the Model object it returns (reference: sqair/model.py:33-214).  This header is the boundary a
 * maintainer would bind from that Python side (ctypes stub in INTEGRATION.md); sqair_amd/ mirrors
 * load()/Model on top of it.
 *
 * Conventions
 *   - plain C symbols, no C++/torch types; every tensor is caller-owned DEVICE memory, contiguous
 *     row-major float32 (HBM-resident torch tensors' data_ptr() in practice);
 *   - the library allocates no device memory: parameters are re-laid-out into a caller-provided
 *     "packed" buffer, all intermediates live in a caller-provided workspace;
 *   - every launch function takes a hipStream_t (passed as void*) and is asynchronous on it;
 *   - return value 0 = ok, negative = error (message via sqair_last_error); one handle per
 *     (device, stream), not thread-safe across threads sharing a handle.
 *
 * Row convention: B' = B*K rows, particle-contiguous (b' = b*K + k), the layout
 * index.tile_input_for_iwae produces (reference: sqair/index.py:106-129) — the tiled observation is
 * never materialised, kernels index obs[t, b'/K].
 */
#ifndef SQAIR_HIP_H
#define SQAIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQAIR_ABI_VERSION 2   /* 2: sqair_build_id / sqair_build_flags added, sqair_profile_* removed */

/* Hyper-parameters of the path.  Field names follow the reference flags
 * (reference: sqair/common_model_flags.py:32-56, sqair/configs/mlp_mnist_model.py:42-52). */
typedef struct SqairConfig {
  int32_t img_h, img_w;          /* H, W of a (grayscale) frame                                  */
  int32_t glimpse_size;          /* G, flag glimpse_size                                         */
  int32_t n_steps_per_image;     /* N object slots, flag n_steps_per_image                       */
  int32_t n_what;                /* flag n_what                                                  */
  int32_t n_hidden;              /* 32 * n_units (get_params, common_model_flags.py:59-71)       */
  int32_t k_particles;           /* K, flag k_particles                                          */
  int32_t prop_prior_type;       /* 0 = rnn, 1 = rw, 2 = guided (propagate.py:36-40)             */
  int32_t disc_prior_type;       /* 0 = cat, 1 = geom (sqair_modules.py:204-224)                 */
  int32_t masked_glimpse;        /* flag masked_glimpse                                          */
  int32_t rec_where_prior;       /* flag rec_where_prior                                         */
  float prop_prior_step_bias;    /* flag prop_prior_step_bias                                    */
  float step_success_prob;       /* flag step_success_prob (geom prior only)                     */
  float output_std;              /* effective p(x|z) std: fl32(fl32(sqrt(flag))^2), modules.py:419-422 */
  float background_std;          /* same for background pixels (bg_std=None -> output_std)       */
  float where_prior_mean[4];     /* scale_prior x2, 0, 0 (non-recurrent where prior only)        */
  int32_t sample_from_prior;     /* flag sample_from_prior (mlp_mnist_model.py:51): needs sqair_set_generation_noise */
  int32_t generate_after;        /* SequentialAIR(generate_after=..) (seq.py:46, :198-200); <= 0: never generate     */
  int32_t time_cell;             /* flag time_transition: 0 = GRU (shipped), 1 = LSTM, 2 = VanillaRNN (common_model_flags.py:49) */
  int32_t prior_cell;            /* flag prior_transition: 0 = GRU (shipped), 1 = LSTM, 2 = VanillaRNN (mlp_mnist_model.py:125)  */
  int32_t rnn_cell;              /* flag transition (slot RNN of both cores): 0 = VanillaRNN (shipped), 1 = LSTM, 2 = GRU */
} SqairConfig;

typedef struct SqairHandle SqairHandle;

/* ---- lifetime / introspection (host only: usable without a GPU) -------------------------------- */
int sqair_abi_version(void);
/* Identity of the loaded binary: 16 hex digits = sha256 over the sources it was compiled from (sqair_amd/csrc/ + this header +
 * compiler flags; sqair_amd/csrc/build.py computes the same hash over the files on disk), and the build variant
 * ("product" | "timeline" | "knobs").  Measurements are quoted next to this id, and the Python binding refuses a binary whose
 * id is not the id of the sources beside it. */
const char* sqair_build_id(void);
const char* sqair_build_flags(void);
/* Returns -1 for a configuration outside this BUILD's limits.  libsqair_hip.so (the product) is laid out for the shipped model family:
 * n_what <= 50, n_steps_per_image <= 8, n_hidden <= 256; libsqair_hip_wide.so -- the same sources compiled with -DSQAIR_WIDE, the
 * same C-ABI, slower per-row kernels -- takes n_what <= 128, n_steps_per_image <= 16 (15, 16 only while the log-probability
 * adjoint's LDS staging fits: not with its 416-float slot record), n_hidden <= 512.  Both: n_hidden = 32 * n_units any multiple
 * of 16 (the kernels run on the next of 128 / 256 / 512 with inert padding units; parameters, gradients and final states cross
 * this ABI in the reference's shapes), k_particles <= 256, any H x W for inference; TRAINING (sqair_forward_train /
 * sqair_backward) takes frames up to 38 400 pixels (150 KB: the crop adjoint stages the frame in LDS) and says so at entry.
 * The reference's flags take any value
 * (sqair/common_model_flags.py:32-56, sqair/configs/mlp_mnist_model.py:42-52); a host picks the library by the flags
 * (sqair_amd/_capi.py: lib_path_for). */
int sqair_create(const SqairConfig* cfg, SqairHandle** out);
int sqair_destroy(SqairHandle* h);
const char* sqair_last_error(const SqairHandle* h);

/* Flat parameter buffer = the reference's trainable variables in the order of SURVEY.md Appendix C
 * (reference listing: notebooks/play.ipynb:239-362), each row-major fp32, no padding.
 * sqair_param_entry enumerates (name, offset, numel) so the Python side can cross-check its spec. */
int64_t sqair_param_count(const SqairHandle* h);
int sqair_param_entries(const SqairHandle* h);
int sqair_param_entry(const SqairHandle* h, int i, const char** name, int64_t* offset, int64_t* numel);

/* Sizes (bytes) of the caller-provided buffers. */
int64_t sqair_packed_bytes(const SqairHandle* h);
int64_t sqair_workspace_bytes(const SqairHandle* h, int T, int B);
int sqair_noise_width(const SqairHandle* h); /* 4 + n_what + 1 */
int sqair_get_config(const SqairHandle* h, SqairConfig* out);

/* ---- parameters ------------------------------------------------------------------------------- */
/* Re-lays the flat parameters out for the MFMA kernels (16x16x4-fp32 fragment order, K padded per
 * input segment, loop-invariant sub-matrices grouped) into `packed`.  Call after every parameter
 * update.  Replaces variable creation inside load() (mlp_mnist_model.py:99-148). */
int sqair_pack_params(SqairHandle* h, const float* flat_params, void* packed, void* stream);

/* ---- the forward pass --------------------------------------------------------------------------
 * Per-frame outputs, every pointer optional (NULL = not written); shapes are [T, B', ...] exactly
 * as the TensorArrays of SequentialAIR (reference: sqair/seq.py:121-177).  */
typedef struct SqairOutputs {
  float* what;                       /* [T,B',N,n_what] */
  float* what_loc;                   /* [T,B',N,n_what] */
  float* what_scale;                 /* [T,B',N,n_what] */
  float* where;                      /* [T,B',N,4] */
  float* where_loc;                  /* [T,B',N,4] */
  float* where_scale;                /* [T,B',N,4] */
  float* presence_prob;              /* [T,B',N] */
  float* presence;                   /* [T,B',N] */
  float* presence_logit;             /* [T,B',N] */
  float* obj_id;                     /* [T,B',N] */
  float* step_log_prob;              /* [T,B'] */
  float* canvas;                     /* [T,B',H,W] */
  float* glimpse;                    /* [T,B',N,G,G] */
  float* disc_what_log_prob;         /* [T,B',N] */
  float* disc_where_log_prob;        /* [T,B',N] */
  float* disc_what_prior_log_prob;   /* [T,B',N] */
  float* disc_where_prior_log_prob;  /* [T,B',N] */
  float* disc_log_prob;              /* [T,B'] */
  float* disc_prior_log_prob;        /* [T,B'] */
  float* disc_prob;                  /* [T,B',N+1] */
  float* prop_what_log_prob;         /* [T,B',N] */
  float* prop_where_log_prob;        /* [T,B',N] */
  float* prop_what_prior_log_prob;   /* [T,B',N] */
  float* prop_where_prior_log_prob;  /* [T,B',N] */
  float* prop_log_prob;              /* [T,B'] */
  float* prop_prior_log_prob;        /* [T,B'] */
  float* prop_prob;                  /* [T,B',N] */
  float* discrete_log_prob;          /* [T,B'] */
  float* num_prop_steps_per_sample;  /* [T,B'] */
  float* num_disc_steps_per_sample;  /* [T,B'] */
  float* num_steps_per_sample;       /* [T,B'] */
  float* prop_pres;                  /* [T,B',N] */
  float* disc_pres;                  /* [T,B',N] */
  float* data_ll_per_sample;         /* [T,B'] */
  float* kl_per_sample;              /* [T,B'] */
  float* log_q_z_given_x_per_sample; /* [T,B'] */
  float* log_p_z_per_sample;         /* [T,B'] */
  float* log_weights_per_timestep;   /* [T,B']  (required by sqair_elbo) */
  /* not reference outputs: final recurrent state, for state-level parity checks */
  float* final_temporal_state;       /* [B',N,n_hidden] */
  float* final_prior_state;          /* [B',N,n_hidden] */
  float* final_last_used_id;         /* [B'] */
} SqairOutputs;

/* Unrolls the model over T frames: replaces SequentialAIR.__call__ on the tiled observation
 * (reference: sqair/seq.py:69-84 -> _loop_body :179-269 -> SQAIRTimestep sqair_modules.py:446-582 ->
 * Propagate/Discover -> PropagationCore/DiscoveryCore sqair/core.py:164-359 -> AIRDecoder
 * sqair/modules.py:435-467 -> _compute_log_weights seq.py:271-276).
 *   obs    [T,B,H,W] in [0,1]
 *   noise  [T,B',2,N,noise_width]: s=0 propagation slot k, s=1 discovery step k; entries 0:4 eps of
 *          `where`, 4:4+n_what eps of `what`, last = uniform u of the presence Bernoulli
 *   t_offset: index of obs[0] in the full sequence (the categorical step prior is time dependent,
 *          sqair_modules.py:212-215); 0 for a whole sequence. */
int sqair_forward(SqairHandle* h, const float* flat_params, const void* packed, const float* obs,
                  const float* noise, int T, int B, int t_offset, const SqairOutputs* out,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* Training-mode forward pass: same launch sequence and results, but every intermediate the backward pass needs
 * (per-frame and per-slot activations, GRU gates, compaction permutation) is kept in the larger workspace of
 * sqair_train_workspace_bytes; sqair_backward consumes it. */
int64_t sqair_train_workspace_bytes(const SqairHandle* h, int T, int B);
int sqair_forward_train(SqairHandle* h, const float* flat_params, const void* packed, const float* obs,
                        const float* noise, int T, int B, int t_offset, const SqairOutputs* out,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* Same launch sequence recorded once into a HIP graph and replayed (the T x 2N sequential steps
 * are launch-latency bound).  Pointers are frozen at capture time: the caller keeps the same
 * buffers and refreshes their contents before each sqair_graph_launch. */
int sqair_graph_capture(SqairHandle* h, const float* flat_params, const void* packed, const float* obs,
                        const float* noise, int T, int B, int t_offset, const SqairOutputs* out,
                        void* workspace, int64_t workspace_bytes, void* stream);
int sqair_graph_launch(SqairHandle* h, void* stream);
int sqair_graph_nodes(const SqairHandle* h); /* kernel nodes in the captured graph */

/* ---- objective ---------------------------------------------------------------------------------
 * Fused IWAE / VIMCO reductions over [T,B,K] (reference: Model._build sqair/model.py:88-103,
 * targets.iwae / vimco_control_variate / vimco sqair/targets.py:38-75, make_target model.py:150-158,
 * ops.ess ops.py:52-59, _imp_weighted_mean model.py:202-205).
 *   log_w_t, disc_lp_t: [T,B*K]; scalars_out[16]:
 *     0 elbo_vae  1 elbo_iwae  2 vimco_target (already / T)  3 ess
 *   iw_means_in/out: optional n_means x [T,B*K] tensors -> n_means importance-weighted per-frame means */
int sqair_elbo(SqairHandle* h, const float* log_w_t, const float* disc_lp_t, int T, int B,
               float* log_weights /*[B,K]*/, float* elbo_iwae_per_example /*[B]*/,
               float* importance_weights /*[B,K]*/, float* vimco_signal /*[B,K]*/, float* scalars_out /*[16]*/,
               const float* const* iw_means_in, int n_means, float* iw_means_out, void* stream);

/* ---- per-kernel entry points (unit parity tests; same kernels sqair_forward launches) ----------- */
/* SpatialTransformer forward crop (reference: sqair/modules.py:170-218): img [B,H,W] shared by K
 * consecutive rows, where_logits [R,4] (R = B*K), optional mask [R,G*G]; out [R,G*G]. */
int sqair_st_crop(SqairHandle* h, const float* img, const float* where_logits, const float* mask,
                  float* out, int B, void* stream);
/* AIRDecoder canvas + Gaussian log-likelihood (reference: sqair/modules.py:435-467, seq.py:271-274):
 * glimpse [R,N,G*G], where [R,N,4], presence [R,N], img [B,H,W], mean_img [H,W];
 * canvas [R,H,W] (optional), data_ll [R]. */
int sqair_st_insert_loglik(SqairHandle* h, const float* glimpse, const float* where_logits,
                           const float* presence, const float* img, const float* mean_img,
                           float* canvas, float* data_ll, int B, void* stream);
/* y = act(x W + b) on the packed MFMA path with an ad-hoc pack of W [K,N] (test helper):
 * act 0 none, 1 elu, 2 tanh, 3 sigmoid, 4 softplus+0.01. */
int sqair_linear_test(SqairHandle* h, const float* x, const float* w, const float* b, float* y, int M,
                      int Kdim, int Ndim, int act, void* scratch, int64_t scratch_bytes, void* stream);
/* snt.GRU step through the two fused MFMA launches the forward pass uses:
 * x [M,Kx], hstate [M,n_hidden]; gru_flat = for g in (z,r,h): w_g [Kx,nh], u_g [nh,nh], b_g [nh]
 * (the order of the reference's GRU variables inside the flat parameter buffer). */
int sqair_gru_test(SqairHandle* h, const float* x, const float* hstate, const float* gru_flat, float* h_out,
                   int M, int Kx, void* scratch, int64_t scratch_bytes, void* stream);

/* snt.LSTM step (time_transition / prior_transition = LSTM; dm_sonnet 1.14 restated: gates (i, j, f, o) =
 * [x, h] w_gates + b_gates; c' = sigmoid(f + 1) c + sigmoid(i) tanh(j); h' = tanh(c') sigmoid(o)):
 * lstm_flat = w_gates [(Kx + nh), 4 nh] then b_gates [4 nh]; state_out [M, 2 nh] = [h' | c']. */
int sqair_lstm_test(SqairHandle* h, const float* x, const float* hstate, const float* cstate, const float* lstm_flat,
                    float* state_out, int M, int Kx, void* scratch, int64_t scratch_bytes, void* stream);
/* adjoint of the element-wise cell: gate pre-activations [M, 4 nh], c_prev, d h', d c' -> d gates, d c_prev */
int sqair_lstm_cell_bwd_test(SqairHandle* h, const float* gates, const float* c_prev, const float* d_h, const float* d_c,
                             float* d_gates, float* d_cprev, int M, void* stream);

/* ---- adjoint (backward) building blocks of the training step (sqair_st_crop_bwd,
 * sqair_st_insert_ll_bwd, ...; the reference gets them from TF autodiff, sqair/model.py:160) ---------- */
/* d/d(where logits) [R,4] and optionally d/d(mask) [R,G*G] of the (masked) crop, given d/d(out) [R,G*G]. */
int sqair_st_crop_bwd(SqairHandle* h, const float* img, const float* where_logits, const float* mask,
                      const float* g_out, float* d_where_logits, float* d_mask, int B, void* stream);
/* Adjoint of sqair_st_insert_loglik for an upstream gradient g_data_ll [R]: d_glimpse [R,N,G*G],
 * d_where_logits [R,N,4], d_mean_img [H,W] (summed over rows; NULL: not reduced, the per-row contributions stay in
 * scratch as [R, H*W]); scratch >= R*H*W*4 bytes. */
int sqair_st_insert_loglik_bwd(SqairHandle* h, const float* glimpse, const float* where_logits,
                               const float* presence, const float* img, const float* mean_img,
                               const float* g_data_ll, float* d_glimpse, float* d_where_logits,
                               float* d_mean_img, void* scratch, int64_t scratch_bytes, int B, void* stream);
/* Gradient of the VIMCO target (already / T) w.r.t. the per-frame log weights and discrete log-probs [T,B*K],
 * from the importance weights and the learning signal sqair_elbo returned. */
int sqair_elbo_bwd(SqairHandle* h, const float* importance_weights, const float* vimco_signal, int T, int B,
                   float* g_log_w_t, float* g_disc_lp_t, void* stream);
/* Decoder branch of sqair_backward (first slice of the training step).  Call after sqair_forward (with
 * SqairOutputs.glimpse == NULL so the decoded glimpses stay in the workspace) and sqair_elbo on the same
 * workspace.  Writes the gradients of the VIMCO target w.r.t. dec.mean_img, dec.l{0,1,2}.{w,b}, dec.output_scale
 * into flat_grad (flat-parameter layout; other entries untouched) and, optionally, the seed gradients on the
 * merged latents d_rec [T, B'*N, 64] (record order: where 0:4, what 4:54). */
int64_t sqair_backward_scratch_bytes(const SqairHandle* h, int T, int B);
int sqair_backward_decoder(SqairHandle* h, const float* flat_params, const void* packed, const float* obs,
                           const float* importance_weights, const float* vimco_signal, int T, int B,
                           void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                           float* flat_grad, float* d_rec_out, void* stream);
/* Full backward pass (replaces TF autodiff of Model.make_target, sqair/model.py:150-168,
 * through SequentialAIR / SQAIRTimestep, sqair/seq.py:60-150, sqair/sqair_modules.py:388-582): gradient of the VIMCO
 * target / T w.r.t. EVERY trainable parameter, written to flat_grad (flat-parameter layout, overwritten).
 * Call order on one stream:  sqair_forward_train(train_workspace)  ->  sqair_elbo  ->  sqair_backward with the same
 * obs / noise / T / B / t_offset, the importance weights and learning signal sqair_elbo returned.  `scratch` holds the
 * gradient records and pre-activation gradient tapes (sqair_backward_bytes). */
int64_t sqair_backward_bytes(const SqairHandle* h, int T, int B);
int sqair_backward(SqairHandle* h, const float* flat_params, const void* packed, const float* obs, const float* noise,
                   const float* importance_weights, const float* vimco_signal, int T, int B, int t_offset,
                   void* train_workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                   float* flat_grad, void* stream);

/* ---- workspace clearing.  By default every pass starts by zero-filling the caller's workspace (60 MB for inference,
 * 308 MB for the training tape at BASELINE configs[1]: ~1-2 % of a step), so that a workspace may hold garbage and may be
 * shared between shapes.  A caller that keeps ONE workspace per (T, B, inference | training) can clear it once with
 * sqair_clear_workspace and switch the per-pass fill off with sqair_set_workspace_clearing(h, 0): every buffer is then
 * either rewritten by the pass or keeps the zeros / finite padding it never overwrites.  Re-clear after changing T or B. */
int sqair_set_workspace_clearing(SqairHandle* h, int each_pass);
int sqair_clear_workspace(SqairHandle* h, void* workspace, int64_t workspace_bytes, int T, int B, int train, void* stream);

/* Run-time options of a handle (API calls of the caller; the library reads NO environment variables):
 *   "tail_fusion" (default 1): compute the tail of slot k inside slot k + 1's VanillaRNN launch; 0 = one launch per
 *                 operation.  Results are bit-identical either way (tests/test_hip_forward.py); affects the following passes
 *                 and captures.
 *   "what_fusion" (default 1): inference passes on the launch path compute a slot's what sample (what, what_loc, what_scale:
 *                 sqair/core.py:226-229, :336-359) in the epilogue of the dense layer that produces its operands -- the glimpse
 *                 encoder's Gaussian head of a discovery slot, the temporal cell's heads of a propagation slot -- instead of in
 *                 the slot tail (csrc/sqair_glue.h: WhatArgs).  Results are bit-identical either way; re-capture graphs after
 *                 changing it.  (Training passes and the slot chain always derive the sample in the tail.)
 *   "vi_target" (default 0): the learning signal sqair_elbo writes (and sqair_backward consumes) and the proxy loss in
 *                 scalars_out[2]: 0 = VIMCO, log w - control variate (sqair/targets.py:62-75: what the reference's make_target
 *                 uses); 1 = plain REINFORCE, log w (sqair/targets.py:78-89, advertised in Model.VI_TARGETS).
 *   "slot_chain" (default 0): run the strictly sequential slot launches of a frame's propagation loop, and of its discovery
 *                 loop, as ONE persistent launch each (csrc/sqair_chain.h: hand-offs through the XCD's L2, polled word by word).
 *                 Bit-identical results.  Serves the shipped cell configuration (VanillaRNN slot cell, GRU temporal cell,
 *                 n_hidden 256) up to 320 particle rows; other shapes keep one launch per op.  Set it BEFORE sizing / clearing
 *                 workspaces (the per-slot buffers are then kept apart like the training tape), and use it on a device this
 *                 process has to itself: the launch needs its 256 workgroups co-resident (another kernel occupying the CUs
 *                 makes it give up after ~20 ms with a status instead of results: sqair_chain_status).  Faster than the
 *                 launches up to ~128 particle rows (one 16-row tile per XCD), about equal at 160.
 *                 A launch reads a table of its ops that holds the operand ADDRESSES (frames, noise, parameters, workspace);
 *                 tables are cached by content in device arenas of the handle.  Passes on the same buffers reuse them; passes
 *                 on fresh buffers make new ones, and a full arena is recycled (after a stream synchronise) unless a captured
 *                 graph refers to it -- memory grows with the number of captured graphs, not with the number of passes.
 *   "slot_chain_arena_kb" (default 32768): size of one such arena; set before the first pass with the chain on.
 * Returns -2 for an unknown name. */
int sqair_set_option(SqairHandle* h, const char* name, int value);
/* Status of the slot chain's launches of the last pass on `workspace` (synchronises `stream`): 0 = all completed (or the chain is
 * off for this shape), else the first non-zero status (1 = an operand never arrived, 3 = the workgroups were not co-resident,
 * 6 = an earlier launch of the pass had failed); the text is in sqair_last_error. */
int sqair_chain_status(SqairHandle* h, void* workspace, int T, int B, int train, void* stream);
/* Debug mode of the reference (`debug=True` -> validate_args / allow_nan_stats=False on its distributions,
 * sqair/core.py:226, :261, sqair/modules.py:318-320; a TF runtime error inside sess.run): checks x[0:n] for NaN / Inf on
 * `stream`, SYNCHRONISES it, and returns -5 with "non-finite values in <what>: count, first index" in sqair_last_error.
 * flag_dev: caller-owned device int32[2] scratch. */
int sqair_check_finite(SqairHandle* h, const float* x, int64_t n, const char* what, int32_t* flag_dev, void* stream);
/* The same mode's argument validation of the Normal distributions: the posterior scales of `what` and `where` of EVERY propagation
 * and discovery slot of the last pass on `workspace` must be positive and finite -- also of slots the presence mask later removes
 * from the log-weights (which sqair_check_finite on the log-weights cannot see).  Synchronises; -5 + text on failure. */
int sqair_check_scales(SqairHandle* h, void* workspace, int T, int B, int train, int32_t* flag_dev, void* stream);

/* ---- per-dispatch timeline (measurement; bench.py's roofline, tools/timeline.py) -----------------------------------------
 * libsqair_hip_timeline.so is THIS library compiled with -DSQAIR_TIMELINE: every kernel takes one more argument and every
 * wave stores {start, end} of its life on the 100 MHz device wall clock (s_memrealtime) into its own 16-byte slot of `buf`.
 * Between _begin and _end every launch issued through the library (eagerly or into a capture) is assigned the next slot
 * range (an eager launch a fresh one every time); replaying a graph captured in between re-stamps the same slots.  _count /
 * _end return the number of records; _record(i)
 * gives kernel name, offset of the range in 8-byte words, waves (= pairs) and workgroups.  Reduce min(start) / max(end) over
 * the pairs of a record for first-wave-start / last-wave-end of that dispatch (zero pairs = padding).  A stamped step runs
 * ~3 % slower than the product library's (measured).  The production library returns -3 from
 * _begin (sqair_timeline_available() == 0) and carries none of this in its kernels. */
int sqair_timeline_available(void);
int sqair_timeline_begin(SqairHandle* h, void* buf, int64_t bytes);
int sqair_timeline_count(const SqairHandle* h);  /* records so far (recording continues); -4 once the stamp buffer has overflowed */
int sqair_timeline_end(SqairHandle* h);
int sqair_timeline_record(const SqairHandle* h, int i, const char** kernel, int64_t* offset_u64, int* waves, int* workgroups);

/* Generation modes (sqair/sqair_modules.py:157-170, :294-302, sqair/seq.py:198-200).  With
 * cfg.sample_from_prior the propagation posterior log-probabilities are evaluated at samples of the propagation PRIOR, and
 * in frames t > cfg.generate_after those samples replace what / where / presence of the propagated objects, discovery's
 * what ~ N(0, I), where ~ its (recurrent) prior and presence = 0.  The extra draws come from `gen_noise`, same layout
 * and meaning as `noise` ([T, B*K, 2, N, 4 + n_what + 1]); the pointer is remembered by the handle and used by the
 * following forward calls.  Forward / inference only. */
int sqair_set_generation_noise(SqairHandle* h, const float* gen_noise);
/* Device-side noise for one pass: fills noise[T, B*K, 2, N, 4 + n_what + 1] with eps ~ N(0,1) / u ~ U[0,1) (last entry
 * of every slot) from Philox4x32-10 keyed by (seed, step, position in the GLOBAL batch): a rank that owns sequences
 * [b0, b0 + B) of a global batch of global_B draws exactly the rows one GPU would have drawn for them.  Replaces the
 * tfd `.sample()` calls the reference makes inside its graph (sqair/core.py:226, sqair/modules.py:60, :485). */
int sqair_fill_noise(SqairHandle* h, float* noise, int T, int B, int global_B, int b0, uint64_t seed, uint64_t step,
                     void* stream);
/* Graph capture of any sequence of the calls above on one stream (the training step up to the gradient all-reduce
 * is ~3000 short dependent launches): _begin, issue the calls, _end(slot 0..3) -> node count (>= 0) or error (< 0);
 * _launch replays the slot.  Captured calls keep the pointers they were given. */
int sqair_capture_begin(SqairHandle* h, void* stream);
int sqair_capture_end(SqairHandle* h, void* stream, int slot);
int sqair_capture_launch(SqairHandle* h, int slot, void* stream);
/* flat_grad += l2 * flat_params: the l2_reg term of Model.make_target (sqair/targets.py:31-35, sqair/model.py:160). */
int sqair_add_l2_grad(SqairHandle* h, const float* flat_params, float* flat_grad, int64_t n, float l2, void* stream);
/* Fused optimiser step on the flat buffers: tf.train.RMSPropOptimizer(lr, momentum=0.9) as used by the
 * reference driver (sqair/scripts/experiment.py:140; TF defaults decay 0.9, epsilon 1e-10, ms initialised to 1):
 * ms <- decay ms + (1-decay) g^2; mom <- momentum mom + lr g / sqrt(ms + eps); theta <- theta - mom, with
 * g = grad_scale * flat_grad (grad_scale = 1/world after the data-parallel all-reduce(sum)). */
int sqair_rmsprop_step(SqairHandle* h, float* flat_params, const float* flat_grad, float* ms, float* mom, int64_t n,
                       float lr, float decay, float momentum, float epsilon, float grad_scale, void* stream);
/* Dense layer backward on the MFMA path (test helper): y = act(x W + b) forward; given dy returns dx [M,K],
 * dw [K,N] (reference [in,out] layout) and db [N]. */
int sqair_linear_bwd_test(SqairHandle* h, const float* x, const float* w, const float* y, const float* dy, float* dx,
                          float* dw, float* db, int M, int Kdim, int Ndim, int act, void* scratch,
                          int64_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SQAIR_HIP_H */
