// Argument structs and launchers of the adjoint kernels (sqair_bwd.hip), shared with the backward driver (sqair_train.hip).
#pragma once
#include "sqair_glue.h"

struct LogprobBwdArgs {
  const float* rec_p; const float* rec_d; const float* rec_m;   // forward records [T][M][168] / merged [T+1][M][168]
  const float* pstats; int ps_ld; const float* spre;
  const float* g_lw; const float* g_dl;                          // [T][R]
  float* d_rec_p; float* d_rec_d; float* d_rec_m;                // gradient records (accumulated)
  float* d_pstats;                                               // [T][M][ps_ld] (written)
  float* d_spre;                                                 // [T][R][128] (written)
  const float* flat; float* flat_grad;
  int t_global0;
  SqairConfig cfg;
};

struct CompactBwdArgs {
  const int* src;                 // [R][N]
  const float* d_rec_next;        // gradient records of the merged slots [M][168]
  const float* d_temporal_next; const float* d_prior_next;   // [M][nh]
  float* d_rec_p; float* d_rec_d; // gradient records of this frame (accumulated)
  float* d_temporal_p; float* d_prior_p;                     // [M][nh] (written: every propagation slot)
  float* d_new_temporal; float* d_new_prior;                 // [R][snh], [R][psnh]: per row, the gradient reaching the trainable
                                                             // initial states through this frame's newly discovered objects
};

struct TailBwdArgs {
  int is_disc, slot;
  const float* rec_prev; const float* rec_new; float* d_rec_new; float* d_rec_prev;  // records / gradient records
  const float* s1h; int s1h_ld;         // saved hidden activations [R][nh/2]
  const float* hraw; int h_ld; const float* enc; int enc_ld; const float* noise;
  float* d_s1pre; int ds_ld;            // out: gradient of the hidden pre-activation [R][nh/2] (T1's extra columns)
  float* d_s1pre2; int ds2_ld;          // optional second copy (the S1 columns of the PRE gradient, propagation)
  int enc_pre;                          // disc: write d_enc as the PRE-activation gradient of the what head (softplus')
  float* d_raw_out; int dr_ld;          // out: d(raw steps-predictor output) per row; the steps.l1 {w, b} gradients are ONE
                                        // batched product s1h^T d_raw at the end of the sweep (160-way contended atomics before)
  float* d_enc; int de_ld;              // out (=): gradient of (loc, scale) of the glimpse encoder
  float* d_hraw; int dh_ld;             // out (=, prop): gradient of the raw head / gate pre-activations
  const float* flat; float* flat_grad;
  int w2_off, b2_off, wwhat_off;        // steps.l1 {w,b}; first `what` row of steps.l0.w ([in, nh/2] layout)
};

struct CropChainBwdArgs {
  int mode, slot;
  const float* img;                      // frame [B,H,W]
  const float* rec_prev; const float* rec_new;   // forward records (where lives in rec_new for PROP2 / DISC)
  float* d_rec_prev; float* d_rec_new;   // gradient records
  const float* wb; int wb_ld;            // PROP1: raw where-bias output
  float* d_wb;                           // PROP1 out: [M][wb_ld]
  const float* mask; int mask_row_mul, mask_row_add; float* d_mask;  // optional; d_mask accumulated (+=)
  int mask_dact;                         // 1: this launch is the last to add to d_mask -- it writes the gradient of the mask
                                         // layer's PRE-activation (x mask (1 - mask)) instead of the sum (was an element-wise launch per frame)
  const float* g_out; int g_row_mul, g_row_add;                       // d glimpse [rows][G2]
  const float* tp; int tp_ld;            // saved transform output (loc 0:4, raw 4:8)
  float* d_tp; int dtp_ld;               // out
  const float* noise; const float* flat; float* flat_grad;
  // optional: the adjoint of the transform's output layer (nh -> 8, fused into the crop on the way forward) in the same launch:
  // d_t2[r][j] = (sum_o d_tp[r][o] w3[j][o]) * elu'(t2[r][j]); w3 = [nh][8] as the forward crop reads it
  const float* w3; const float* t2; int t2_ld; float* d_t2; int dt2_ld;
};

int sq_launch_logprob_bwd(const LogprobBwdArgs& a, POff po, Dims d, int T, hipStream_t s);
int sq_launch_compact_bwd(const CompactBwdArgs& a, POff po, Dims d, hipStream_t s);
int sq_launch_slot_tail_bwd(const TailBwdArgs& a, Dims d, hipStream_t s);
int sq_launch_crop_chain_bwd(const CropChainBwdArgs& a, POff po, Dims d, int nslots, hipStream_t s);
int sq_launch_gru_bwd_a(const float* d_hn, int dhn_ld, const float* z, int z_ld, const float* hc, int hc_ld,
                        const float* hprev, int h_ld, float* dpre1, int dp_ld, float* d_h, int dh_ld, int rows, int nh,
                        int accumulate_dh, hipStream_t s, float* dup_z = nullptr, int dup_ld = 0, int dup_h_off = -1);
int sq_launch_gru_bwd_b(const float* d_rh, int drh_ld, const float* rg, int r_ld, const float* hprev, int h_ld,
                        float* dpre1, int dp_ld, float* d_h, int dh_ld, int rows, int nh, hipStream_t s, float* dup_r = nullptr,
                        int dup_ld = 0);
int sq_launch_lstm_cell_bwd(const float* gates, int g_ld, const float* c_prev, int c_ld, const float* d_h, int dh_ld, const float* d_c,
                            int dc_ld, float* d_gates, int dg_ld, float* d_cprev, int dcp_ld, int rows, int nh, hipStream_t s,
                            float* d_gates2 = nullptr, int dg2_ld = 0);   // d_c may be null (= 0); c_ld 0 broadcasts
int sq_launch_dact2(const float* din, int in_ld, const float* saved, int s_ld, float* dout, int out_ld, int rows, int cols,
                    int act_a, int act_b, int split, int acc, hipStream_t s);
int sq_launch_colsum(const float* dy, int ld, int rows, int cols, float* out, int acc, hipStream_t s);
int sq_launch_latent_sum_bwd(const float* d_c, const float* rec_p, const float* f_out, float* d_f, Dims d, hipStream_t s);
int sq_launch_sum_slots(const float* d_rnn, float* d_pre_d, float* d_pre_disc, int B, int K, int N, int nh, hipStream_t s);
int sq_launch_particle_sum(const float* in, float* out, int B, int K, int nh, hipStream_t s);
int sq_launch_axpy2d(const float* x, int x_ld, float* y, int y_ld, int rows, int cols, int acc, hipStream_t s);
int sq_launch_wgrad(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, float* db, int M, int Kdim,
                    int Ndim, int accumulate, hipStream_t s, const int* rowmap = nullptr, const float* alpha_ptr = nullptr);
int sq_launch_insert_bwd_frames(const float* glimpse, const float* rec, int rec_ld, const float* img, const float* mean_img,
                                const float* g_ll, float* d_glimpse, float* d_rec, int d_rec_ld, float* d_mean_rows,
                                float std_fg, float std_bg, int T, Dims d, hipStream_t s, const float* scale = nullptr,
                                float* d_scale = nullptr);   // scale / d_scale: the output scale's gradient in the same launch
int sq_launch_reduce_rows(const float* rows, float* out, int R, int P, int accumulate, hipStream_t s);
int sq_launch_elbo_bwd(const float* iw, const float* sig, int T, int B, int K, float* g_lw, float* g_dl, hipStream_t s);
int sq_launch_dot_scale(const float* a, const float* b, int64_t n, const float* scale, float* out, hipStream_t s);
int sq_launch_reduce_rows_atomic(const float* rows, float* out, int R, int P, hipStream_t s);
int sq_launch_wgrad_acc(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, int M, int Kdim, int Ndim,
                        hipStream_t s, const int* rowmap, const float* alpha_ptr, float* db_a, float* db_b);
// Weight-gradient blocks collected over the backward pass and issued together (k_wgrad_group, sqair_bwd.hip): same
// arguments as sq_launch_wgrad_acc; add() returns false for a block the grouped kernel does not take (small or unaligned:
// the caller launches it on its own).  Operands must stay untouched until flush().
struct WgDesc {
  const float* A; const float* dY; float* dW; const int* rowmap; const float* alpha_ptr; float* db_a; float* db_b;
  int lda, ldy, ldw, M, Kdim, Ndim, kt, n_tiles;
  // filled by flush(): rows and number of the block's M-chunks; a chunk's tiles in `tg` groups of `gs` tiles (a group is what
  // goes to ONE XCD together); qb[x] = the first of the block's zc * tg groups in XCD x's queue
  int m_per_wg, zc, tg, gs;
  unsigned short qb[8];
};
constexpr int SQ_WG_MAXD = 96;
// begin[x][i]: position in XCD x's queue where block i starts (i = nd: the queue's length; beyond: INT_MAX).  Workgroup b of the
// launch is entry b / 8 of the queue of XCD b % 8 (workgroups are handed to the XCDs round-robin).
struct WgGroup { WgDesc d[SQ_WG_MAXD]; int begin[8][SQ_WG_MAXD + 8]; int nd; };
struct WgradBatch {
  std::vector<WgDesc> blocks;
  bool add(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, int M, int Kdim, int Ndim, const int* rowmap,
           const float* alpha_ptr, float* db_a, float* db_b);
  int flush(hipStream_t s);
};

// Column sums collected at the end of the backward pass and issued as ONE launch (k_colsum_group): out[n] += sum_m dy[m][n] wt[m]
// (wt null: plain bias / initial-state gradients; with wt a one-column weight gradient, d w2 = A^T d_raw of the steps predictors).
// Seven to eleven such sums close the chain: as launches of their own each paid the dependent-launch gap and a cold round trip.
struct ColsumEntry { const float* dy; const float* wt; float* out; int ld, wld, rows, cols, first, nbx; };
constexpr int SQ_CS_MAXE = 12;
struct ColsumGroup { ColsumEntry e[SQ_CS_MAXE]; int n; };
struct ColsumBatch {
  ColsumGroup g{};
  int total = 0;
  void add(const float* dy, int ld, int rows, int cols, float* out, hipStream_t s, const float* wt = nullptr, int wld = 0);
  int flush(hipStream_t s);
};

int sq_launch_where_param_grads(const float* d_tp, int tp_ld, const float* d_rec_p, const float* rec_p, const float* noise, int T,
                                Dims d, float* flat_grad, POff po, hipStream_t s);
