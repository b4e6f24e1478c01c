// Launchers of the non-GEMM kernels (sqair_glue.hip).
#pragma once
#include "sqair_common.h"

// Limits of a build.  The product library is laid out for the shipped model family (slot record with 50 `what` entries, 8 slots,
// hidden layers up to 256 wide); libsqair_hip_wide.so is the same source compiled with -DSQAIR_WIDE for the rest of the range the
// reference's flags span: larger records, more slots, wider layers -- slower kernels (more registers / LDS per workgroup), same
// results, same C-ABI.  sqair_amd picks the library from the flags.
#ifdef SQAIR_WIDE
constexpr int SQ_MAXN = 16;          // n_steps_per_image
constexpr int SQ_MAX_NWHAT = 128;    // n_what
constexpr int SQ_MAX_NHIDDEN = 512;  // 32 * n_units (after padding to a multiple of 128)
#else
constexpr int SQ_MAXN = 8;           // max object slots supported by the small per-row kernels
constexpr int SQ_MAX_NWHAT = 50;
constexpr int SQ_MAX_NHIDDEN = 256;
#endif
constexpr int SQ_MAX_K = 256;        // k_particles

// Offsets (in floats) into the flat parameter buffer of the small layers the per-row kernels read
// straight from the unpacked parameters.
struct POff {
  int dec_mean_img, dec_output_scale;
  int rn_init_state, rn_init_sample, rn_readout_w, rn_readout_b, rn_cond_w, rn_cond_b, rn_h2h_w, rn_h2h_b,
      rn_i2h_w, rn_i2h_b;
  int sp_l0_w, sp_l0_b, sp_l1_w, sp_l1_b, step_prior_bias, step_prior_tbias;
  int disc_steps_l1_w, disc_steps_l1_b, prop_steps_l1_w, prop_steps_l1_b;
  int disc_scale_offset, prop_scale_offset, cholesky;
  int disc_rnn_init, prop_rnn_init, prior_init, temporal_init;
};

// floor(e / x) == __umulhi(e, mul) + (e & one) for 0 <= e < 2^32 / x with mul = floor(2^32 / x) + 1, one = 0; x = 1 (whose
// multiplier does not fit 32 bits): mul = 0, one = all ones.  No select, no branch.
struct SqMagic { unsigned mul, one; };
inline SqMagic sq_magic(int x) { return x <= 1 ? SqMagic{0u, 0xffffffffu} : SqMagic{(unsigned)((1ull << 32) / (unsigned)x + 1), 0u}; }
struct Dims {
  int H, W, G, N, nw, nh, K, R, B;  // R = B*K rows
  int nzw;                          // noise width = 4 + nw + 1
  int snh;                          // width of the temporal state of a slot: nh (GRU) or 2 nh = [hidden | cell] (LSTM)
  int toff;                         // offset of the features the model reads from it: 0 (GRU state) / nh (LSTM cell, core.py:284)
  int psnh;                         // width of the propagation prior's recurrent state: nh (GRU) or 2 nh (LSTM)
  int rsnh;                         // width of the slot RNN's trainable initial state: nh (VanillaRNN) or 2 nh (LSTM)
  SqMagic nw_mul, g_mul, k_mul;     // sq_magic(nw), sq_magic(G), sq_magic(K): e / x == sq_div(e, x_mul).  A runtime integer division is
                                    // ~25 instructions; in the slot tail and the crops it stood ahead of the operand loads of every
                                    // launch of the slot loop (forward step 3.44 -> 3.39 ms, training 7.90 -> 7.84)
  int P4;                           // floats between consecutive frames in the buffer the kernels read frames from: H * W, or --
                                    // inside a pass over frames whose H * W is not a multiple of 4 -- H * W rounded up to 4 (the
                                    // pass stages such frames through a zero-padded copy so that every frame starts 16-byte aligned)
};
#ifdef __HIPCC__
__device__ __forceinline__ int sq_div(int e, SqMagic m) { return (int)(__umulhi((unsigned)e, m.mul) + ((unsigned)e & m.one)); }
#endif
enum { RNN_VANILLA = 0, RNN_LSTM = 1, RNN_GRU = 2 };    // SqairConfig.rnn_cell (flag transition)
enum { CELL_GRU = 0, CELL_LSTM = 1, CELL_VANILLA = 2 };  // SqairConfig.time_cell / .prior_cell (flags time_transition / prior_transition)
// pre-activation columns of the slot RNN: nh (VanillaRNN), the four LSTM gates, or the GRU's [z | r | candidate]
inline int sq_rnn_width(const SqairConfig& c) { return c.n_hidden * (c.rnn_cell == RNN_LSTM ? 4 : (c.rnn_cell == RNN_GRU ? 3 : 1)); }
// gate pre-activation columns of the temporal / prior cell: 3 nh (GRU: z, r, candidate), 4 nh (LSTM), nh (VanillaRNN)
inline int sq_gate_width(const SqairConfig& c, int cell) { return c.n_hidden * (cell == CELL_LSTM ? 4 : (cell == CELL_VANILLA ? 1 : 3)); }
inline Dims make_dims(const SqairConfig& c, int B) {
  const bool lstm = c.time_cell == CELL_LSTM;
  return Dims{c.img_h, c.img_w, c.glimpse_size, c.n_steps_per_image, c.n_what, c.n_hidden, c.k_particles, B * c.k_particles, B,
              4 + c.n_what + 1, lstm ? 2 * c.n_hidden : c.n_hidden, lstm ? c.n_hidden : 0,
              (c.prior_cell == CELL_LSTM) ? 2 * c.n_hidden : c.n_hidden, c.rnn_cell == RNN_LSTM ? 2 * c.n_hidden : c.n_hidden,
              sq_magic(c.n_what), sq_magic(c.glimpse_size), sq_magic(c.k_particles), c.img_h * c.img_w};
}

#ifdef __HIPCC__
// Particle row handled by workgroup i of a launch with one workgroup per row (grid.x = R, R a multiple of 8).  Workgroups go to
// the eight XCDs round-robin (XCD = i % 8), each XCD has its own L2, and the K particles of a sequence read the SAME frame: with
// rows taken in launch order a frame was fetched into up to K of the eight L2s (PMC: 2.5x the algorithmic bytes of k_crop_row).
// With the number of sequences a multiple of 8, sequence b's K rows go to XCD b % 8, so a frame enters one L2.
__device__ __forceinline__ int sq_row_of_wg(int i, const Dims& d) {
  if ((d.B & 7) != 0) return i;
  const int xcd = i & 7, j = i >> 3;
  const int bb = sq_div(j, d.k_mul), k = j - bb * d.K;
  // (wave-uniform by construction; said so explicitly: the multiply-high of sq_div is a vector instruction, and a row index in a
  // VGPR turns every address of the kernel into vector arithmetic)
  return __builtin_amdgcn_readfirstlane((bb * 8 + xcd) * d.K + k);
}
#endif
enum CropMode { CROP_PLAIN = 0, CROP_PROP1 = 1, CROP_PROP2 = 2, CROP_DISC = 3 };
// Frames up to this many pixels are staged in LDS by the crop kernels while the where computation runs (a 50 x 50 frame is
// 10 KB: the copy hides behind the where sample); larger ones (BASELINE configs[4]: 128 x 128 = 64 KB per workgroup, of which a
// 20 x 20 glimpse touches at most 1600 pixels) are sampled where they lie -- four taps per glimpse pixel straight from L2, all
// in flight at once, instead of 54 dependent copy trips per thread
constexpr int SQ_CROP_STAGE_MAX_PIXELS = 4096;

struct CropArgs {
  int mode;
  const float* img;        // [B,H,W] frame t
  const float* logits;     // PLAIN: [R,4]
  const float* mask;       // optional [rows, G*G]
  int mask_row_mul, mask_row_add;  // mask row = r*mul + add
  float* out;              // [rows, G*G]
  int out_row_mul, out_row_add;
  const float* rec_prev;   // merged records of t-1 [R,N,168]
  float* rec_new;          // prop / disc records of this frame [R,N,168]
  const float* wb;         // PROP1: raw where-bias MLP output [(R*N), wb_ld]
  int wb_ld;
  const float* tp;         // PROP2 / DISC: transform MLP output [R, tp_ld] (loc 0:4, raw scale 4:8), or
  int tp_ld;
  const float* t2;         // ... its input [R, t2_ld]: the 256 -> 8 output layer is then evaluated in this launch
  int t2_ld;
  const float* w3;         // 16-byte aligned copy of transform.l2 {w [nh,8], b [8]} (workspace, see k_init_state)
  const float* noise;      // noise of frame t, [R,2,N,nzw]
  const float* flat;       // flat parameters
  int slot;                // PROP2 / DISC slot; PROP1 uses blockIdx.y
  float* tp_out;           // optional (training): the transform output [R, tp_out_ld] (loc 0:4, raw scale 4:8)
  int tp_out_ld;
};

int sq_launch_lstm_cell2(const float* gates, int g_ld, const float* c_prev, int c_ld, float* h_out, int h_ld, float* c_out, int co_ld,
                         int rows, int nh, hipStream_t s);
int sq_launch_lstm_cell(const float* gates, int g_ld, const float* c_prev, int c_ld, float* state_out, int o_ld, int rows, int nh,
                        hipStream_t s);
int sq_launch_init_state(float* rec_m, float* temporal_m, float* prior_m, float* last_id, float* disc_init_rec,
                         float* prop_rnn_init, float* disc_rnn_init, float* rn_init_state, float* w3_prop, float* w3_disc,
                         int w3p_off, int w3d_off, const float* flat,
                         POff po, Dims d, hipStream_t s);
int sq_launch_crop(const CropArgs& a, POff po, Dims d, int nslots, hipStream_t s);
// Tail of a propagation / discovery slot in one launch: what-sample, the what-dependent part of the steps
// predictor's hidden layer (small MFMA, K = 56), its output layer (dot with w2) and the presence Bernoulli.
struct TailArgs {
  int is_disc, slot;
  const float* hraw; int h_ld;      // prop: raw what-head (2 nw) + gate (3 nw) pre-activations
  const float* enc; int enc_ld;     // glimpse-encoder Gaussian (loc nw | scale nw)
  const float* rec_prev;            // merged records of t-1
  float* rec_new;                   // rec_p / rec_d
  const float* noise;               // noise of frame t
  const float* s1p; int s1p_ld;     // [R, nh/2] hidden pre-activation without the `what` term
  const float* wp;                  // packed [nh/32 ... ] weights of the `what` rows of steps.l0 (layer *_S1)
  const float* flat; int w2_off, b2_off;
  float* s1h_out; int s1h_ld;       // optional (training): the hidden activations elu(.) [R, nh/2]
  int what_done;                    // 1: the slot's what sample (what, what_loc, what_scale of rec_new) has been written by the layer that
                                    // produces its operands (k_linear_what below): the tail reads `what` instead of deriving it
};
// "What fusion" (round 6; INFERENCE passes only -- in a training pass the layer would also have to write its own output to the tape for
// the adjoint, in the reference column order: measured, 7.34 -> 7.38 ms per training step, against 7.35 without the fusion there): the what sample of a slot computed in the epilogue of the dense layer that
// produces its last operands -- a discovery slot's in the glimpse encoder's Gaussian head (what = loc + scale eps,
// sqair/core.py:226-229), a propagation slot's in the temporal cell's heads (the gated mixture of sqair/core.py:336-359) --
// through packs whose output columns put the two / five pre-activations of one `what` element into adjacent lanes
// (L_WHAT_HEAD_I, L_PROP_HEADS_I).  The element-wise work is then done ONCE per element by the lanes of one launch instead of
// by every one of the 16 column-tile workgroups of the next slot's fused RNN + tail launch, whose per-workgroup instruction
// stream is what that launch waits for (timing ablation: -2.8 % of the cfg-2 forward step).  Same arithmetic on the same
// operands: results are bit-identical to the tail deriving the sample (tests/test_hip_forward.py).
struct WhatArgs {
  int mode;                          // 0: discovery (pairs loc, scale); 1: propagation (t_loc, t_scale, forget, input, temporal gate)
  const float* x; int x_ld;          // A operand [M][K = 16 kc] (one segment, 16-byte aligned rows)
  const float* wp; const float* wzero; const float* bias;   // interleaved pack of the layer, the packed buffer's zero block, packed bias
  int M, kc, n_tiles, nw, N, slot, nzw;
  const float* noise;                // noise of frame t [R][2][N][nzw]
  const float* enc; int enc_ld;      // mode 1: glimpse-encoder Gaussian of the slot [R][loc nw | scale nw]
  const float* rec_prev;             // mode 1: merged records of t - 1 [R][N][rec::W]
  float* rec_new;                    // rec_p / rec_d of this frame: what, what_loc, what_scale of (row, slot) are written
  int layer_id;                      // (for the host-side launch log)
};
int sq_launch_linear_what(const WhatArgs& a, hipStream_t s);
int sq_launch_slot_tail(const TailArgs& a, Dims d, hipStream_t s);
int sq_launch_rnn_tail(const TailArgs& ta, Dims d, const float* hid, int hid_ld, const float* wp, const float* bias, const float* add,
                       int add_ld, float* out, int out_ld, int n_out, hipStream_t s, unsigned long long* prof_ts);
int sq_launch_latent_sum(const float* f, const float* rec_p, float* c, Dims d, hipStream_t s);

struct LogprobArgs {
  const float* rec_p; const float* rec_d; const float* rec_prev;
  const float* pstats; int ps_ld;     // raw prior linear output [(R*N), ps_ld]
  const float* spre;                  // [R,128] pre-activation of the where-prior conditioning state (without e)
  const float* flat;
  int t_global;                       // absolute frame index (categorical prior is time dependent)
  int t;                              // index of the first frame inside the output tensors
  int n_frames;                       // frames covered by the launch (inputs are [n_frames][...] contiguous)
  float* qz; float* pz; float* disc_lp;  // frame scalars [R]
  const float* gen;                   // sample_from_prior: generation records [n_frames][R*N][64] (see GenArgs), else NULL
  SqairOutputs out;
  SqairConfig cfg;
};
int sq_launch_logprob(const LogprobArgs& a, POff po, Dims d, hipStream_t s);

// Generation modes (sqair_modules.py:157-170, :294-302).  Generation record of slot (r, k), 64 floats:
//   [0:4] where ~ prior, [4:54] what ~ prior, [54] presence ~ Bernoulli(prior logit), [55] the posterior path's own
//   propagation presence, [56] the posterior path's own discovery presence of step k
namespace gen {
#ifdef SQAIR_WIDE
constexpr int WHERE = 0, WHAT = 4, PRES = 132, ORIG_PRES = 133, ORIG_DPRES = 134, W = 144;
#else
constexpr int WHERE = 0, WHAT = 4, PRES = 54, ORIG_PRES = 55, ORIG_DPRES = 56, W = 64;
#endif
}
struct GenArgs {
  float* rec_p; float* rec_d; const float* rec_prev;   // records of this frame (overwritten when generating)
  const float* pstats; int ps_ld; const float* spre;   // prior statistics / where-prior conditioning of this frame
  const float* gen_noise;                              // frame slice of the second noise tensor [R][2][N][nzw]
  float* gen;                                          // generation records of this frame [R*N][64]
  const float* flat;
  int do_generate;
  SqairConfig cfg;
};
int sq_launch_generate_prop(const GenArgs& a, POff po, Dims d, hipStream_t s);
int sq_launch_generate_disc(const GenArgs& a, POff po, Dims d, hipStream_t s);

struct CompactArgs {
  const float* rec_p; const float* rec_d; const float* rec_prev;
  const float* temporal_p; const float* prior_p;
  const float* last_id_prev; float* last_id_next;
  float* rec_next; float* temporal_next; float* prior_next;
  const float* flat;
  int t;
  SqairOutputs out;
  int* src_out;            // optional (training): source slot (0..2N-1) of every surviving slot [R,N]
};
int sq_launch_compact(const CompactArgs& a, POff po, Dims d, hipStream_t s);

struct InsertArgs {
  const float* glimpse;    // [R,N,G*G]
  const float* rec;        // merged records of this frame [R,N,rec_ld] (where at +0, presence at +54) or
  int rec_ld;              // plain mode: where [R,N,4] / presence [R,N] given separately
  const float* where_plain; const float* pres_plain;
  const float* img;        // [B,H,W]
  const float* mean_img;   // [H,W]
  float* canvas;           // optional [R,H,W]
  float* data_ll;          // [R]
  const float* qz; const float* pz;  // optional frame scalars -> log weight outputs
  int t;                   // first frame in the output tensors
  int n_frames;            // frames covered by the launch (0/1 = single)
  SqairOutputs out;        // only the scalar log-weight outputs are used (may be all NULL)
  float std_fg, std_bg;
};
int sq_launch_insert_loglik(const InsertArgs& a, Dims d, hipStream_t s);

int sq_launch_elbo(const float* log_w_t, const float* disc_lp_t, int T, int B, int K, float* log_weights,
                   float* elbo_per_ex, float* iw, float* signal, float* scalars, const float* const* means_in,
                   int n_means, float* means_out, hipStream_t s, int reinforce = 0);
