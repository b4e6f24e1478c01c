// sqair_backward — reverse sweep of the whole forward pass: gradients of the VIMCO
// target / T (reference: Model.make_target sqair/model.py:150-168, targets.vimco sqair/targets.py:62-75; the
// reference obtains them from TF autodiff through tf.while_loop, both static_rnns, the resampler, the
// dynamic_partition) w.r.t. every trainable parameter, written into the flat gradient buffer.
//
// Structure mirrors sqair_forward (sections A..J of sq_forward_impl) in reverse:
//   ELBO adjoint -> decoder branch + log-probability adjoint for all T frames at once (they are off the recurrence)
//   -> for t = T-1..0: compaction^T, discovery slots N-1..0, latent summary, propagation slots N-1..0, the
//      loop-invariant pre-activation GEMM, crop #1 / mask / where-bias MLPs, prior GRU
//   -> initial states, input encoder -> ONE weight-gradient GEMM per (layer, segment) over all its uses (the tape
//      keeps activations and pre-activation gradients of every use in slot-inner order, M up to 2*T*B'*N rows).
// Every dense adjoint is the forward MFMA kernel on the transposed pack; gradients of z-record segments come out in
// record order and accumulate straight into "gradient records".
#include "sqair_internal.h"
#include "sqair_dx.h"
#include "sqair_bwd.h"

// a few elementwise helpers local to the driver
// zero fill as a kernel node: memset nodes in the middle of a long captured chain proved unreliable on replay
// (ROCm 7.2: the second replay of the training graph read stale scratch), a plain kernel keeps the chain uniform
__global__ void k_zero(float* __restrict__ p, int64_t n SQ_TLP) {
  SQ_TL_SCOPE;
  const int64_t n4 = n / 4;
  float4* p4 = reinterpret_cast<float4*>(p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    p4[i] = float4{0.0f, 0.0f, 0.0f, 0.0f};
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] = 0.0f;
}
#ifdef SQAIR_KNOBS
__global__ void k_fill_value(float* __restrict__ p, int64_t n, float v SQ_TLP) {
  SQ_TL_SCOPE;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
#endif
void sq_zero_fill(float* p, int64_t n, hipStream_t s) {  // p 16-byte aligned
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (n > 0) SQ_LAUNCH(k_zero, dim3((unsigned)blocks), dim3(256), 0, s, p, n);
}
__global__ void k_copy(const float* __restrict__ src, float* __restrict__ dst, int64_t n SQ_TLP) {
  SQ_TL_SCOPE;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void sq_copy(float* dst, const float* src, int64_t n, hipStream_t s) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (n > 0) SQ_LAUNCH(k_copy, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n);
}
// shifted z-record / RNN-state inputs of the slot RNNs as dense matrices for the batched weight gradient:
//   zs[(t,r,k)] = z-record of slot k-1 (k > 0) or `init_rec` (k = 0);  rs[(t,r,k)] = r tape of slot k-1 or rnn_init;
//   with a GRU slot RNN also the candidate's recurrent input rh = reset gate * rs.  Both phases (blockIdx.y: propagation,
//   discovery) in one launch -- they were four nodes at the end of the chain.
struct ShiftPhase { const float *rec_all, *r_tape, *init_rec, *rnn_init, *gate; float *zs, *rs, *rh; };
struct MulJob { const float *a, *b; float* out; };   // out[row][i] = a[row][i] b[row][i], all three [rows][nh]
struct ShiftArgs { ShiftPhase p[2]; MulJob m[2]; int rows, N, nh, gate_ld; };
// blockIdx.y: 0 / 1 = the phases, 2 / 3 = the element-wise products of the prior / temporal GRU candidates' recurrent inputs
// (reset gate * previous state), which need the same launch shape
__global__ void k_shift_inputs(const ShiftArgs a SQ_TLP) {
  SQ_TL_SCOPE;
  const int row = blockIdx.x;  // (t, r, k) flattened
  if (row >= a.rows) return;
  const int nh = a.nh;
  if (blockIdx.y >= 2) {
    const MulJob m = a.m[blockIdx.y - 2];
    for (int i = threadIdx.x; i < nh; i += blockDim.x) m.out[(size_t)row * nh + i] = m.a[(size_t)row * nh + i] * m.b[(size_t)row * nh + i];
    return;
  }
  const ShiftPhase p = a.p[blockIdx.y];
  const int k = row % a.N;
  for (int i = threadIdx.x; i < rec::ZWP; i += blockDim.x)
    p.zs[(size_t)row * rec::ZWP + i] = i < rec::ZW ? (k > 0 ? p.rec_all[(size_t)(row - 1) * rec::W + i] : p.init_rec[i]) : 0.0f;
  for (int i = threadIdx.x; i < nh; i += blockDim.x) {
    const float v = k > 0 ? p.r_tape[(size_t)(row - 1) * nh + i] : p.rnn_init[i];
    p.rs[(size_t)row * nh + i] = v;
    if (p.gate != nullptr) p.rh[(size_t)row * nh + i] = p.gate[(size_t)row * a.gate_ld + i] * v;
  }
}

// builder for the routed dX GEMM
struct Dx {
  DxArgs a;
  Dx(const float* dpre, int ld) {
    memset(&a, 0, sizeof(a));
    a.dpre = dpre; a.ld = ld;
  }
  DxRange& last() { return a.r[a.nranges - 1]; }
  Dx& to(int n0, int n1, float* dst, int dst_ld) {
    DxRange& r = a.r[a.nranges++];
    r.n0 = n0; r.n1 = n1; r.dst = dst; r.dst_ld = dst_ld; r.act_split = 1 << 30;
    return *this;
  }
  Dx& acc() { last().add = last().dst; last().add_ld = last().dst_ld; return *this; }
  Dx& add(const float* p, int ld) { last().add = p; last().add_ld = ld; return *this; }
  Dx& dact(const float* saved, int ld, int act_a, int act_b = ACT_NONE, int split = 1 << 30) {
    last().saved = saved; last().saved_ld = ld; last().act_a = act_a; last().act_b = act_b; last().act_split = split;
    return *this;
  }
  Dx& dup(float* p, int ld) { last().dst2 = p; last().dst2_ld = ld; return *this; }
  // GRU gate adjoints in this launch's epilogue (DxGru): mode 1 = after the d h' GEMM, mode 2 = after the d(r h) GEMM
  Dx& gru_a(const float* z, int z_ld, const float* hc, int hc_ld, const float* hprev, int h_ld, float* dpre1, int dp_ld, float* d_h,
            int dh_ld, int acc_dh, int nh, float* dupp = nullptr, int dup_ld = 0, int dup_h_off = -1) {
    a.gru = DxGru{1, z, z_ld, hc, hc_ld, hprev, h_ld, dpre1, dp_ld, d_h, dh_ld, acc_dh, dupp, dup_ld, dup_h_off, nh};
    return *this;
  }
  Dx& gru_b(const float* r, int r_ld, const float* hprev, int h_ld, float* dpre1, int dp_ld, float* d_h, int dh_ld, int nh,
            float* dupp = nullptr, int dup_ld = 0) {
    a.gru = DxGru{2, r, r_ld, nullptr, 0, hprev, h_ld, dpre1, dp_ld, d_h, dh_ld, 1, dupp, dup_ld, -1, nh};
    return *this;
  }
};

struct BwdSpace {
  float *g_lw, *g_dl;
  float *d_rec_m, *d_rec_p, *d_rec_d;          // gradient records
  // d temporal / prior merged state, one buffer per frame boundary 0..T inside the part of the scratch that is cleared once
  // per pass (by frame parity they needed a clear per frame: 10 extra launches on the critical path)
  float *d_tm0, *d_pm0;
  int64_t tm_stride, pm_stride;
  float* d_tm(int t) const { return d_tm0 + (size_t)t * tm_stride; }
  float* d_pm(int t) const { return d_pm0 + (size_t)t * pm_stride; }
  float *d_temporal_p, *d_prior_p;
  float *d_pstats, *d_spre, *d_raw;
  // per-frame pre-activation gradients (kept for the batched weight gradients)
  float *d_pgru1, *d_hid1, *d_wb, *d_maskpre, *d_pea, *d_peb, *d_m1, *d_pre, *d_lea, *d_leb, *d_pre_d, *d_pre_disc;
  // per-slot pre-activation gradients [2][T][R][N][W]
  float *d_rnn, *d_t1, *d_t2, *d_tp, *d_e1, *d_e2, *d_enc3, *d_gru1, *d_hraw;
  // scratch
  float *d_g, *d_g1, *d_c, *tmp, *d_r[2], *dhn, *d_rh, *d_enc;
  float *d_new_t, *d_new_p;                      // [T][R][snh | psnh]: compaction adjoint rows for the initial recurrent states
  float *d_init_p, *d_init_d, *d_rn0;            // per-frame dX rows of the trainable initial states: summed once after the sweep
  float *d_cs[2], *d_hk;                        // LSTM slot RNN: d cell state of the neighbouring slot, d hidden of this one
  float *d_gl, *d_mean_rows, *bufa, *bufb, *d_ia, *d_ib;
  float* bufc;  // [MT][nh] second pre-activation gradient of the decoder (bufb keeps the first for the grouped weight gradients)
  // A operands built for the batched weight gradients (fully written before they are read: outside the zero-filled part).
  // One copy per use -- the grouped launch at the end of the pass reads them all
  float *zs[2], *rs[2], *rh[4];
  float* grad_padded;  // padded configurations only: flat gradient in the padded shapes
  int64_t zero_total;  // floats from the base that the pass expects zeroed
  int64_t total;
};

static BwdSpace carve_bwd(const SqairHandle* h, int T, int B, float* base) {
  const SqairConfig& c = h->cfg;
  const int64_t nh = c.n_hidden, N = c.n_steps_per_image, R = (int64_t)B * c.k_particles, M = R * N;
  const int64_t G2 = c.glimpse_size * c.glimpse_size, P_ = c.img_h * c.img_w, MT = M * T;
  const int64_t pre_ld = h->layers[L_PRE].nt * 16;
  BwdSpace b;
  memset(&b, 0, sizeof(b));
  // Two groups: buffers the pass accumulates into or reads before it has written all of them come first and are cleared at the
  // start of every pass (zero_total floats); buffers that are WRITTEN IN FULL before anything reads them -- the decoder's
  // temporaries, the per-row mean-image terms, the A operands built for the batched weight gradients -- follow and are not.
  // The knob build can fill the second group with NaNs ahead of a pass (SQAIR_SCRATCH_POISON): the gradient tests pass on that
  // library, which is what "written in full" rests on.  (The per-layer pre-activation gradients do NOT qualify: with them
  // poisoned, NaNs reach enc.what_head / enc.mask / disc.* -- padded columns behind zero weights, and the paths a flag
  // disables, whose weight gradients are still taken from the zeroed buffers.)
  int64_t o = 0;
  for (int pass = 0; pass < 2; ++pass) {
  auto T_ = [&](float*& dst, int64_t n, bool written_in_full = false) {
    if ((int)written_in_full != pass) return;
    dst = base ? base + o : nullptr;
    o += align64(n);
  };
  T_(b.g_lw, T * R); T_(b.g_dl, T * R);
  T_(b.d_rec_m, (T + 1) * M * rec::W); T_(b.d_rec_p, MT * rec::W); T_(b.d_rec_d, MT * rec::W);
  const int64_t snh = c.time_cell == CELL_LSTM ? 2 * nh : nh, gw = sq_gate_width(c, c.time_cell);  // temporal state / gate widths
  const int64_t psnh = c.prior_cell == CELL_LSTM ? 2 * nh : nh, pgw = sq_gate_width(c, c.prior_cell);
  const int64_t rw = sq_rnn_width(c);  // slot-RNN pre-activation width
  b.tm_stride = align64(M * snh); b.pm_stride = align64(M * psnh);
  T_(b.d_tm0, (T + 1) * b.tm_stride); T_(b.d_pm0, (T + 1) * b.pm_stride);
  T_(b.d_temporal_p, M * snh); T_(b.d_prior_p, M * psnh);
  T_(b.d_pstats, MT * PS_LD); T_(b.d_spre, T * R * 128); T_(b.d_raw, 2 * MT);
  T_(b.d_pgru1, MT * pgw); T_(b.d_hid1, MT * 256); T_(b.d_wb, MT * WB_LD); T_(b.d_maskpre, MT * G2);
  T_(b.d_pea, MT * nh); T_(b.d_peb, MT * nh); T_(b.d_m1, MT * M1_LD); T_(b.d_pre, MT * pre_ld);
  T_(b.d_lea, MT * nh); T_(b.d_leb, MT * nh); T_(b.d_pre_d, T * R * rw); T_(b.d_pre_disc, (int64_t)T * B * rw);
  const int64_t S = 2 * MT;
  T_(b.d_rnn, S * rw); T_(b.d_t1, S * T1_LD); T_(b.d_t2, S * nh); T_(b.d_tp, S * TP_LD);
  T_(b.d_e1, S * nh); T_(b.d_e2, S * nh); T_(b.d_enc3, S * ENC_LD); T_(b.d_gru1, MT * gw);
  T_(b.d_hraw, MT * HRAW_LD);
  T_(b.d_g, R * G2); T_(b.d_g1, M * G2); T_(b.d_c, R * nh);
  T_(b.tmp, M * 512);
  T_(b.d_new_t, T * R * snh); T_(b.d_new_p, T * R * psnh);
  T_(b.d_init_p, T * R * nh); T_(b.d_init_d, T * R * nh); T_(b.d_rn0, T * R * 4);
  T_(b.d_r[0], R * nh); T_(b.d_r[1], R * nh); T_(b.dhn, M * nh); T_(b.d_rh, M * nh);
  T_(b.d_enc, R * ENC_LD);
  T_(b.d_cs[0], R * nh); T_(b.d_cs[1], R * nh); T_(b.d_hk, R * nh);
  const int64_t big = MT * (nh > G2 ? nh : G2);
  T_(b.d_gl, MT * G2, true); T_(b.d_mean_rows, T * R * P_, true); T_(b.bufa, big, true); T_(b.bufb, big, true);
  T_(b.d_ia, (int64_t)T * B * nh); T_(b.d_ib, (int64_t)T * B * nh);
  T_(b.bufc, MT * nh, true);
  for (int i = 0; i < 2; ++i) { T_(b.zs[i], MT * rec::ZWP, true); T_(b.rs[i], MT * nh, true); }
  for (int i = 0; i < 4; ++i) T_(b.rh[i], MT * nh, true);
  if (pass == 0) b.zero_total = o;
  }
  // padded configurations: the gradient in the padded shapes (gathered into the caller's buffer at the end of the pass)
  b.grad_padded = base ? base + o : nullptr;
  if (h->padded) o += align64(h->n_params);
  b.total = o;
  return b;
}

extern "C" int64_t sqair_backward_bytes(const SqairHandle* h, int T, int B) {
  if (!h || T < 1 || B < 1) return -1;
  return carve_bwd(h, T, B, nullptr).total * 4;
}

#define CK(x) do { int _r = (x); if (_r != 0) { sq_set_error(h, std::string("sqair_backward: ") + #x); return _r; } } while (0)

extern "C" int sqair_backward(SqairHandle* h, const float* flat, const void* packedv, const float* obs, const float* noise,
                              const float* importance_weights, const float* vimco_signal, int T, int B, int t_offset,
                              void* train_workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                              float* flat_grad, void* stream) {
  if (!h || !flat || !packedv || !obs || !noise || !importance_weights || !vimco_signal || !train_workspace || !scratch || !flat_grad)
    return -1;
  if (workspace_bytes < sqair_train_workspace_bytes(h, T, B) || scratch_bytes < sqair_backward_bytes(h, T, B)) {
    sq_set_error(h, "sqair_backward: workspace / scratch too small");
    return -1;
  }
  if (!sq_trainable_frame(h)) return -1;
  hipStream_t s = (hipStream_t)stream;
  const float* packed = (const float*)packedv;
  const SqairConfig& c = h->cfg;
  const int nh = c.n_hidden, nw = c.n_what, N = c.n_steps_per_image, K = c.k_particles;
  const int R = B * K, M = R * N, MT = M * T, G2 = c.glimpse_size * c.glimpse_size, P_ = c.img_h * c.img_w;
  const int nzw = 4 + nw + 1, RW = rec::W, nsp = nh / 2;
  Dims d = make_dims(c, B);
  const int snh = d.snh, gw = sq_gate_width(c, c.time_cell), psnh = d.psnh, pgw = sq_gate_width(c, c.prior_cell);
  const int rw = sq_rnn_width(c);  // slot-RNN pre-activation width; d_pre columns [rnn rw | T1 nh | S1 nsp | GRU z, r]
  const POff po = h->po;
  const Workspace w = sq_carve(h, T, B, (float*)train_workspace, true);
  const BwdSpace b = carve_bwd(h, T, B, (float*)scratch);
  const PackedLayout pl = packed_layout(h);
  // (frames whose H * W is not a multiple of 4: the zero-padded copy the forward pass left in the training workspace)
  if ((P_ & 3) != 0) { d.P4 = (P_ + 3) / 4 * 4; obs = w.obs_p; }
  const int PL = d.P4;
  const int pre_ld = h->layers[L_PRE].nt * 16;
  const int* rm_dev = (const int*)packed + pl.rm;

  if ((reinterpret_cast<uintptr_t>(flat_grad) & 15) != 0) {
    sq_set_error(h, "sqair_backward: flat_grad must be 16-byte aligned");
    return -1;
  }
  // padded configurations (SqairHandle): parameters are read from the padded copy inside the packed buffer, the gradient is
  // accumulated in the padded shapes and gathered into the caller's buffer at the end
  float* const user_grad = flat_grad;
  flat = sq_flat(h, flat, packedv);
  if (h->padded) flat_grad = b.grad_padded;
  sq_zero_fill(flat_grad, h->n_params, s);
  sq_zero_fill((float*)scratch, b.zero_total, s);
#ifdef SQAIR_KNOBS
  {  // measurement / test knob: NaNs into everything the pass claims to write in full before reading (carve_bwd)
    static const bool poison = SQ_KNOB_SET("SQAIR_SCRATCH_POISON");
    if (poison) SQ_LAUNCH(k_fill_value, dim3(2048), dim3(256), 0, s, (float*)scratch + b.zero_total, b.total - b.zero_total, __builtin_nanf(""));
  }
#endif

  // dX through the transposed pack: out[M][K of the forward layer] (+)= dpre[M][N] W^T.  Single-segment layers write
  // exactly their true input width; multi-segment layers write all 16 * kc padded columns (the caller splits them).
  auto dx = [&](LayerId id, const float* dpre, int ld, int Mrows, float* outp, int out_ld, bool acc,
                const float* scale_ptr = nullptr) -> int {
    const PackedLayer& LT = h->layersT[id];
    const PackedLayer& LF = h->layers[id];
    Lin l;
    l.seg(dpre, ld, LF.N).out(outp, out_ld).act(ACT_NONE);
    l.a.scale_ptr = scale_ptr;
    l.a.wp = packed + pl.w + LT.w_off; l.a.wzero = packed + pl.w; l.a.bias = packed + pl.b + LT.b_off;
    l.a.M = Mrows; l.a.N = LF.seg_width.size() == 1 ? LF.seg_width[0] : LT.N;
    if (acc) { l.a.add = outp; l.a.add_ld = out_ld; l.a.add_n = l.a.N; }
    const int rc = sq_launch_linear(l.a, LT, s);
    if (rc != 0) sq_set_error(h, "sqair_backward: A-operand contract violated in dX of layer " + std::to_string((int)id));
    return rc;
  };
  auto rundx = [&](LayerId id, Dx& dxa, int Mrows, const float* scale_ptr = nullptr) -> int {
    const PackedLayer& LT = h->layersT[id];
    dxa.a.width = h->layers[id].N; dxa.a.wp = packed + pl.w + LT.w_off; dxa.a.wzero = packed + pl.w;
    dxa.a.scale_ptr = scale_ptr; dxa.a.M = Mrows;
    const int rc = sq_launch_linear_dx(dxa.a, LT.kc, LT.nt, s);
    if (rc != 0) sq_set_error(h, "sqair_backward: A-operand contract violated in dX of layer " + std::to_string((int)id));
    return rc;
  };
  // batched weight + bias gradients of one layer over `rows` uses
  // `defer`: the block joins the grouped launch at the end of the pass (its operands must then stay as they are until there)
  WgradBatch wbatch;
  auto wgrad = [&](LayerId id, std::vector<std::pair<const float*, int>> segs, const float* dY, int ldy, int rows, bool defer = true) {
    // the bias gradient of a column block rides on the first weight-gradient launch of that block
    std::vector<char> bias_done(h->bg[id].size(), 0);
    for (const auto& e : h->wg[id]) {
      float *dba = nullptr, *dbb = nullptr;
      for (size_t bi = 0; bi < h->bg[id].size(); ++bi) {
        const auto& be = h->bg[id][bi];
        if (!bias_done[bi] && be.n0 == e.n0 && be.ncols == e.ncols) {
          if (!be.a.empty()) dba = flat_grad + P(h, be.a) + be.col0;
          if (!be.b.empty()) dbb = flat_grad + P(h, be.b) + be.col0;
          bias_done[bi] = 1;
        }
      }
      if (defer && wbatch.add(segs[e.seg].first, segs[e.seg].second, dY + e.n0, ldy, flat_grad + P(h, e.w) + e.col0, PC(h, e.w), rows,
                              h->layers[id].seg_width[e.seg], e.ncols, rm_dev + e.rm_off, nullptr, dba, dbb))
        continue;
      sq_launch_wgrad_acc(segs[e.seg].first, segs[e.seg].second, dY + e.n0, ldy, flat_grad + P(h, e.w) + e.col0, PC(h, e.w), rows,
                          h->layers[id].seg_width[e.seg], e.ncols, s, rm_dev + e.rm_off, nullptr, dba, dbb);
    }
    for (size_t bi = 0; bi < h->bg[id].size(); ++bi)
      if (!bias_done[bi]) {  // (no weight block with the same column range: plain column sums)
        const auto& e = h->bg[id][bi];
        if (!e.a.empty()) sq_launch_colsum(dY + e.n0, ldy, rows, e.ncols, flat_grad + P(h, e.a) + e.col0, 1, s);
        if (!e.b.empty()) sq_launch_colsum(dY + e.n0, ldy, rows, e.ncols, flat_grad + P(h, e.b) + e.col0, 1, s);
      }
  };
  auto slotp = [&](float* base, int W, int t, int ph, int k) { return base + (((size_t)(ph * T + t) * R * N) + k) * W; };
  auto cslotp = [&](const float* base, int W, int t, int ph, int k) { return base + (((size_t)(ph * T + t) * R * N) + k) * W; };

  // ================= 0. objective =================
  sq_launch_elbo_bwd(importance_weights, vimco_signal, T, B, K, b.g_lw, b.g_dl, s);

  // ================= J^T. decoder branch, all frames =================
  {
    const float* rec_all = w.rec_m_all + (size_t)M * RW;
    float* d_rec_all = b.d_rec_m + (size_t)M * RW;
    const float* gl = w.glimpse;
    const float* scale = flat + po.dec_output_scale;
    CK(sq_launch_insert_bwd_frames(gl, rec_all, RW, obs, flat + po.dec_mean_img, b.g_lw, b.d_gl, d_rec_all + rec::WHERE, RW,
                                   b.d_mean_rows, c.output_std, c.background_std, T, d, s, scale, flat_grad + po.dec_output_scale));
    sq_launch_reduce_rows_atomic(b.d_mean_rows, flat_grad + po.dec_mean_img, T * R, P_, s);
    if (!wbatch.add(w.dec_b, nh, b.d_gl, G2, flat_grad + P(h, "dec.l2.w"), G2, MT, nh, G2, nullptr, scale, flat_grad + P(h, "dec.l2.b"), nullptr))
      sq_launch_wgrad_acc(w.dec_b, nh, b.d_gl, G2, flat_grad + P(h, "dec.l2.w"), G2, MT, nh, G2, s, nullptr, scale,
                          flat_grad + P(h, "dec.l2.b"), nullptr);
    CK(dx(L_DEC2, b.d_gl, G2, MT, b.bufa, nh, false, scale));
    sq_launch_dact2(b.bufa, nh, w.dec_b, nh, b.bufb, nh, MT, nh, ACT_ELU, ACT_ELU, 1 << 30, 0, s);
    wgrad(L_DEC1, {{w.dec_a, nh}}, b.bufb, nh, MT);
    CK(dx(L_DEC1, b.bufb, nh, MT, b.bufa, nh, false));
    sq_launch_dact2(b.bufa, nh, w.dec_a, nh, b.bufc, nh, MT, nh, ACT_ELU, ACT_ELU, 1 << 30, 0, s);
    wgrad(L_DEC0, {{rec_all, RW}}, b.bufc, nh, MT);
    CK(dx(L_DEC0, b.bufc, nh, MT, d_rec_all, RW, true));
  }
  // ================= H^T. log-probabilities, all frames =================
  {
    LogprobBwdArgs la; memset(&la, 0, sizeof(la));
    la.rec_p = w.rec_p_all; la.rec_d = w.rec_d_all; la.rec_m = w.rec_m_all; la.pstats = w.pstats; la.ps_ld = PS_LD;
    la.spre = w.spre; la.g_lw = b.g_lw; la.g_dl = b.g_dl; la.d_rec_p = b.d_rec_p; la.d_rec_d = b.d_rec_d;
    la.d_rec_m = b.d_rec_m; la.d_pstats = b.d_pstats; la.d_spre = b.d_spre; la.flat = flat; la.flat_grad = flat_grad;
    la.t_global0 = t_offset; la.cfg = c;
    CK(sq_launch_logprob_bwd(la, po, d, T, s));
  }

  // the adjoint of the transform's 8-wide output layer rides in the crop adjoint that produces its input (80 launches fewer)
  const bool fuse_t3 = !SQ_KNOB_SET("SQAIR_NO_T3_FUSION");
  // ================= reverse sweep over the frames =================
  for (int t = T - 1; t >= 0; --t) {
    const float* img = obs + (size_t)t * B * PL;
    const float* nz = noise + (size_t)t * R * 2 * N * nzw;
    const float* rec_prev = w.rec_m_all + (size_t)t * M * RW;
    const float* rec_p_t = w.rec_p_all + (size_t)t * M * RW;
    const float* rec_d_t = w.rec_d_all + (size_t)t * M * RW;
    float* d_rec_prev = b.d_rec_m + (size_t)t * M * RW;
    float* d_rec_next = b.d_rec_m + (size_t)(t + 1) * M * RW;
    float* d_rec_p_t = b.d_rec_p + (size_t)t * M * RW;
    float* d_rec_d_t = b.d_rec_d + (size_t)t * M * RW;
    const float* temporal_prev = w.state(w.temporal_m, t, w.snh);
    const float* prior_prev = w.state(w.prior_m, t, w.psnh);
    float* d_tau = b.d_tm(t);       // d temporal_m[t]
    float* d_pprev = b.d_pm(t);     // d prior_m[t]
    const int rl = N * nh, t1l = N * T1_LD, gl2 = N * G2, el = N * ENC_LD, hl = N * HRAW_LD, tpl = N * TP_LD, s1l = N * S1_LD;

    // ---- I^T. compaction
    {
      CompactBwdArgs ka; memset(&ka, 0, sizeof(ka));
      ka.src = w.src + (size_t)t * M; ka.d_rec_next = d_rec_next; ka.d_temporal_next = b.d_tm(t + 1);
      ka.d_prior_next = b.d_pm(t + 1); ka.d_rec_p = d_rec_p_t; ka.d_rec_d = d_rec_d_t;
      ka.d_temporal_p = b.d_temporal_p; ka.d_prior_p = b.d_prior_p;
      ka.d_new_temporal = b.d_new_t + (size_t)t * R * snh; ka.d_new_prior = b.d_new_p + (size_t)t * R * psnh;
      sq_launch_compact_bwd(ka, po, d, s);
    }
    // ---- G^T. discovery steps
    float* d_pre_d = b.d_pre_d + (size_t)t * R * rw;
    for (int j = N - 1; j >= 0; --j) {
      float* d_t1 = slotp(b.d_t1, T1_LD, t, 1, j);
      float* d_t2 = slotp(b.d_t2, nh, t, 1, j);
      float* d_tp = slotp(b.d_tp, TP_LD, t, 1, j);
      float* d_e1 = slotp(b.d_e1, nh, t, 1, j);
      float* d_e2 = slotp(b.d_e2, nh, t, 1, j);
      float* d_enc3 = slotp(b.d_enc3, ENC_LD, t, 1, j);
      float* d_rnn = slotp(b.d_rnn, rw, t, 1, j);
      const int drl = N * rw;
      const float* r_j = cslotp(w.r, nh, t, 1, j);
      const float* t1 = cslotp(w.t1, T1_LD, t, 1, j);
      const float* t2 = cslotp(w.t2, nh, t, 1, j);
      const float* e1 = cslotp(w.e1, nh, t, 1, j);
      const float* e2 = cslotp(w.e2, nh, t, 1, j);
      const float* enc = cslotp(w.enc, ENC_LD, t, 1, j);
      {
        TailBwdArgs ta; memset(&ta, 0, sizeof(ta));
        ta.is_disc = 1; ta.slot = j; ta.rec_prev = rec_prev; ta.rec_new = rec_d_t; ta.d_rec_new = d_rec_d_t;
        ta.d_rec_prev = d_rec_prev; ta.s1h = cslotp(w.s1h, S1_LD, t, 1, j); ta.s1h_ld = s1l; ta.enc = enc; ta.enc_ld = el;
        ta.noise = nz; ta.d_s1pre = d_t1 + nh; ta.ds_ld = t1l; ta.d_enc = d_enc3; ta.de_ld = el; ta.enc_pre = 1; ta.d_raw_out = slotp(b.d_raw, 1, t, 1, j); ta.dr_ld = N; ta.flat = flat;
        ta.flat_grad = flat_grad; ta.w2_off = po.disc_steps_l1_w; ta.b2_off = po.disc_steps_l1_b;
        ta.wwhat_off = (int)P(h, "disc.steps.l0.w") + nh * nsp;
        CK(sq_launch_slot_tail_bwd(ta, d, s));
      }
      { Dx x(d_enc3, el); x.to(0, nh, d_e2, rl).dact(e2, rl, ACT_ELU); CK(rundx(L_WHAT_HEAD, x, R)); }
      { Dx x(d_e2, rl); x.to(0, nh, d_e1, rl).dact(e1, rl, ACT_ELU); CK(rundx(L_GENC1, x, R)); }
      { Dx x(d_e1, rl); x.to(0, G2, b.d_g, G2); CK(rundx(L_GENC0, x, R)); }
      {
        CropChainBwdArgs ca; memset(&ca, 0, sizeof(ca));
        ca.mode = CROP_DISC; ca.slot = j; ca.img = img; ca.rec_prev = rec_prev; ca.rec_new = rec_d_t; ca.d_rec_prev = d_rec_prev;
        ca.d_rec_new = d_rec_d_t; ca.g_out = b.d_g; ca.g_row_mul = 1; ca.tp = cslotp(w.tp, TP_LD, t, 1, j); ca.tp_ld = tpl;
        ca.d_tp = d_tp; ca.dtp_ld = tpl; ca.noise = nz; ca.flat = flat; ca.flat_grad = flat_grad;
        if (fuse_t3) { ca.w3 = w.w3_disc; ca.t2 = t2; ca.t2_ld = rl; ca.d_t2 = d_t2; ca.dt2_ld = rl; }
        CK(sq_launch_crop_chain_bwd(ca, po, d, 1, s));
      }
      if (!fuse_t3) { Dx x(d_tp, tpl); x.to(0, nh, d_t2, rl).dact(t2, rl, ACT_ELU); CK(rundx(L_DISC_T3, x, R)); }
      { Dx x(d_t2, rl); x.to(0, nh, d_t1, t1l).dact(t1, t1l, ACT_ELU); CK(rundx(L_DISC_T2, x, R)); }
      if (c.rnn_cell == RNN_LSTM) {  // d h_j = T1^T + the next slot's RNN; cell adjoint -> gate pre-activation gradients, d c_{j-1}
        Dx x(d_t1, t1l); x.to(0, nh, b.d_hk, nh);
        if (j < N - 1) x.add(b.d_r[j & 1], nh);
        CK(rundx(L_DISC_T1, x, R));
        sq_launch_lstm_cell_bwd(cslotp(w.rgates, 4 * nh, t, 1, j), N * 4 * nh, j == 0 ? w.disc_rnn_init + nh : cslotp(w.rc, nh, t, 1, j - 1),
                                j == 0 ? 0 : rl, b.d_hk, nh, j < N - 1 ? b.d_cs[j & 1] : nullptr, nh, d_rnn, drl, b.d_cs[(j + 1) & 1], nh, R, nh, s);
      } else if (c.rnn_cell == RNN_GRU) {  // GRU adjoint in the two stages of the temporal cell; d h_{j-1} starts in d_r[(j-1)&1] / tmp
        Dx x(d_t1, t1l); x.to(0, nh, b.d_hk, nh);
        if (j < N - 1) x.add(b.d_r[j & 1], nh);
        CK(rundx(L_DISC_T1, x, R));
        const float* g3 = cslotp(w.rgates, 3 * nh, t, 1, j);
        const float* hp = j == 0 ? w.disc_rnn_init : cslotp(w.r, nh, t, 1, j - 1);
        const int g3l = N * 3 * nh, hpl = j == 0 ? 0 : rl;
        float* dhp = j > 0 ? b.d_r[(j - 1) & 1] : b.d_init_d + (size_t)t * R * nh;
        sq_launch_gru_bwd_a(b.d_hk, nh, g3, g3l, g3 + 2 * nh, g3l, hp, hpl, d_rnn, drl, dhp, nh, R, nh, 0, s);
        { Dx y(d_rnn + 2 * nh, drl); y.to(0, nh, b.d_rh, nh); CK(rundx(L_DISC_RNN2, y, R)); }
        sq_launch_gru_bwd_b(b.d_rh, nh, g3 + nh, g3l, hp, hpl, d_rnn, drl, dhp, nh, R, nh, s);
      } else {  // d r_j = T1^T (incl. the steps-predictor columns) + what the next slot's RNN sent back; tanh' -> d pre-activation
        Dx x(d_t1, t1l); x.to(0, nh, d_rnn, rl);
        if (j < N - 1) x.add(b.d_r[j & 1], nh);
        x.dact(r_j, rl, ACT_TANH);
        CK(rundx(L_DISC_T1, x, R));
      }
      if (j > 0) {
        Dx x(d_rnn, drl);
        x.to(0, rec::ZW, d_rec_d_t + (size_t)(j - 1) * RW, N * RW).acc();
        x.to(rec::ZWP, rec::ZWP + nh, b.d_r[(j - 1) & 1], nh);
        if (c.rnn_cell == RNN_GRU) x.acc();   // the gate adjoints above already put their direct part there
        CK(rundx(L_DISC_RNN, x, R));
      } else {
        Dx x(d_rnn, drl); x.to(rec::ZWP, rec::ZWP + nh, b.d_init_d + (size_t)t * R * nh, nh);   // d (initial hidden state): column sum after the sweep
        if (c.rnn_cell == RNN_GRU) x.acc();
        CK(rundx(L_DISC_RNN, x, R));
        if (c.rnn_cell == RNN_LSTM) sq_launch_colsum(b.d_cs[1], nh, R, nh, flat_grad + po.disc_rnn_init + nh, 1, s);
      }
    }
    // ---- F^T. conditioning of discovery on the propagated latents
    sq_launch_sum_slots(b.d_rnn + (size_t)(T + t) * M * rw, d_pre_d, b.d_pre_disc + (size_t)t * B * rw, B, K, N, rw, s);
    { Dx x(d_pre_d, rw); x.to(0, nh, b.d_c, nh); CK(rundx(L_PRED, x, R)); }
    if (c.rec_where_prior) {
      Dx x(b.d_spre + (size_t)t * R * 128, 128);
      x.to(0, 4, b.d_rn0 + (size_t)t * R * 4, 4);
      x.to(16, 16 + nh, b.d_c, nh).acc();
      CK(rundx(L_RNCOND, x, R));
    }
    {
      float* d_leb = b.d_leb + (size_t)t * M * nh;
      float* d_lea = b.d_lea + (size_t)t * M * nh;
      const float* leb = w.frame(w.leb, (int64_t)M * nh, t);
      const float* lea = w.frame(w.lea, (int64_t)M * nh, t);
      sq_launch_latent_sum_bwd(b.d_c, rec_p_t, leb, d_leb, d, s);
      { Dx x(d_leb, nh); x.to(0, nh, d_lea, nh).dact(lea, nh, ACT_ELU); CK(rundx(L_LAT1, x, M)); }
      { Dx x(d_lea, nh); x.to(0, rec::ZW, d_rec_p_t, RW).acc(); CK(rundx(L_LAT0, x, M)); }
    }
    // ---- E^T. propagation slots
    float* d_pre = b.d_pre + (size_t)t * M * pre_ld;
    // the mask gradient of the frame accumulates in the frame's slice of d_maskpre (cleared with the scratch) and gets the
    // sigmoid's derivative in place below
    float* const d_mask_t = b.d_maskpre + (size_t)t * M * G2;
    const float* mask = w.frame(w.mask, (int64_t)M * G2, t);
    for (int k = N - 1; k >= 0; --k) {
      float* d_t1 = slotp(b.d_t1, T1_LD, t, 0, k);
      float* d_t2 = slotp(b.d_t2, nh, t, 0, k);
      float* d_tp = slotp(b.d_tp, TP_LD, t, 0, k);
      float* d_e1 = slotp(b.d_e1, nh, t, 0, k);
      float* d_e2 = slotp(b.d_e2, nh, t, 0, k);
      float* d_enc3 = slotp(b.d_enc3, ENC_LD, t, 0, k);
      float* d_rnn = slotp(b.d_rnn, rw, t, 0, k);
      const int drl = N * rw;
      float* d_gru1 = b.d_gru1 + ((size_t)t * M + k) * gw;   // [T][R][N][3nh (GRU) | 4nh (LSTM)], row stride N*gw
      float* d_hraw = b.d_hraw + ((size_t)t * M + k) * HRAW_LD;
      const int g1l = N * gw;
      const float* r_k = cslotp(w.r, nh, t, 0, k);
      const float* t1 = cslotp(w.t1, T1_LD, t, 0, k);
      const float* t2 = cslotp(w.t2, nh, t, 0, k);
      const float* e1 = cslotp(w.e1, nh, t, 0, k);
      const float* e2 = cslotp(w.e2, nh, t, 0, k);
      const float* enc = cslotp(w.enc, ENC_LD, t, 0, k);
      const float* tau_k = temporal_prev + (size_t)k * snh;   // GRU: the state; LSTM: [hidden | cell], features = cell
      float* d_tau_k = d_tau + (size_t)k * snh;
      float* d_pre_k = d_pre + (size_t)k * pre_ld;   // columns: rnn 0:nh | T1 nh:2nh | S1 2nh:2nh+nsp | z, r
      const int pre_rld = N * pre_ld;
      {
        TailBwdArgs ta; memset(&ta, 0, sizeof(ta));
        ta.is_disc = 0; ta.slot = k; ta.rec_prev = rec_prev; ta.rec_new = rec_p_t; ta.d_rec_new = d_rec_p_t;
        ta.d_rec_prev = d_rec_prev; ta.s1h = cslotp(w.s1h, S1_LD, t, 0, k); ta.s1h_ld = s1l;
        ta.hraw = cslotp(w.hraw, HRAW_LD, t, 0, k); ta.h_ld = hl; ta.enc = enc; ta.enc_ld = el; ta.noise = nz;
        ta.d_s1pre = d_t1 + nh; ta.ds_ld = t1l; ta.d_s1pre2 = d_pre_k + rw + nh; ta.ds2_ld = pre_rld;
        ta.d_enc = b.d_enc; ta.de_ld = ENC_LD; ta.d_hraw = d_hraw; ta.dh_ld = hl; ta.d_raw_out = slotp(b.d_raw, 1, t, 0, k); ta.dr_ld = N;
        ta.flat = flat; ta.flat_grad = flat_grad; ta.w2_off = po.prop_steps_l1_w; ta.b2_off = po.prop_steps_l1_b;
        ta.wwhat_off = (int)P(h, "prop.steps.l0.w") + 2 * nh * nsp;
        CK(sq_launch_slot_tail_bwd(ta, d, s));
      }
      // heads -> d tau'_k (the new HIDDEN state), + what the compaction sent back for this slot's new temporal state
      if (c.time_cell == CELL_VANILLA) {  // tanh' folded into the same launch; second copy: the hoisted recurrent block of d_pre
        Dx x(d_hraw, hl);
        x.to(0, nh, d_gru1, g1l).add(b.d_temporal_p + (size_t)k * nh, N * nh)
            .dact(w.frame(w.temporal_p, (int64_t)M * nh, t) + (size_t)k * nh, N * nh, ACT_TANH).dup(d_pre_k + rw + nh + nsp, pre_rld);
        CK(rundx(L_PROP_HEADS, x, R));
      } else if (c.time_cell == CELL_GRU) {  // d tau'_k and, in the same launch's epilogue, the gate adjoints that consume it
        Dx x(d_hraw, hl);
        x.to(0, nh, b.dhn, nh).add(b.d_temporal_p + (size_t)k * snh, N * snh)
            .gru_a(cslotp(w.gz, nh, t, 0, k), rl, cslotp(w.ghc, nh, t, 0, k), rl, tau_k, N * nh, d_gru1, g1l, d_tau_k, N * nh, 1, nh,
                   d_pre_k + rw + nh + nsp, pre_rld);
        CK(rundx(L_PROP_HEADS, x, R));
      } else {
        Dx x(d_hraw, hl); x.to(0, nh, b.dhn, nh).add(b.d_temporal_p + (size_t)k * snh, N * snh); CK(rundx(L_PROP_HEADS, x, R));
      }
      if (c.time_cell == CELL_VANILLA) {
      } else if (c.time_cell == CELL_LSTM) {
        // cell adjoint: gate pre-activation gradients (kept for the batched weight gradients and the recurrent-rows dX
        // after the slot loop) and d c_{t-1} -- the first writer of the cell half of d_tau (sections D^T / B^T accumulate)
        sq_launch_lstm_cell_bwd(w.lgates + ((size_t)t * M + k) * 4 * nh, N * 4 * nh, tau_k + nh, N * snh, b.dhn, nh,
                                b.d_temporal_p + (size_t)k * snh + nh, N * snh, d_gru1, g1l, d_tau_k + nh, N * snh, R, nh, s);
      } else {
        Dx x(d_gru1 + 2 * nh, g1l);
        x.to(0, nh, b.d_rh, nh).gru_b(cslotp(w.gr, nh, t, 0, k), rl, tau_k, N * nh, d_gru1, g1l, d_tau_k, N * nh, nh,
                                     d_pre_k + rw + 2 * nh + nsp, pre_rld);
        CK(rundx(L_PROP_GRU2, x, R));
      }
      {  // gate GEMM inputs [r_k nh | where 4 (pad 16) | glimpse-encoder (loc, scale) 2 nw]
        Dx x(d_gru1, g1l);
        x.to(0, nh, b.d_r[k & 1], nh);
        if (k < N - 1) x.acc();                                      // slot k+1's RNN already wrote d r_k there
        x.to(nh, nh + 4, d_rec_p_t + (size_t)k * RW + rec::WHERE, N * RW).acc();
        x.to(nh + 16, nh + 16 + 2 * nw, d_enc3, el).add(b.d_enc, ENC_LD).dact(enc, el, ACT_NONE, ACT_SOFTPLUS_MIN, nw);
        CK(rundx(L_PROP_GRU1, x, R));
      }
      { Dx x(d_enc3, el); x.to(0, nh, d_e2, rl).dact(e2, rl, ACT_ELU); CK(rundx(L_WHAT_HEAD, x, R)); }
      { Dx x(d_e2, rl); x.to(0, nh, d_e1, rl).dact(e1, rl, ACT_ELU); CK(rundx(L_GENC1, x, R)); }
      { Dx x(d_e1, rl); x.to(0, G2, b.d_g, G2); CK(rundx(L_GENC0, x, R)); }
      {
        CropChainBwdArgs ca; memset(&ca, 0, sizeof(ca));
        ca.mode = CROP_PROP2; ca.slot = k; ca.img = img; ca.rec_prev = rec_prev; ca.rec_new = rec_p_t; ca.d_rec_prev = d_rec_prev;
        ca.d_rec_new = d_rec_p_t; ca.mask = c.masked_glimpse ? mask : nullptr; ca.mask_row_mul = N; ca.mask_row_add = k;
        ca.d_mask = d_mask_t; ca.g_out = b.d_g; ca.g_row_mul = 1; ca.tp = cslotp(w.tp, TP_LD, t, 0, k); ca.tp_ld = tpl;
        ca.d_tp = d_tp; ca.dtp_ld = tpl; ca.noise = nz; ca.flat = flat; ca.flat_grad = flat_grad;
        if (fuse_t3) { ca.w3 = w.w3_prop; ca.t2 = t2; ca.t2_ld = rl; ca.d_t2 = d_t2; ca.dt2_ld = rl; }
        CK(sq_launch_crop_chain_bwd(ca, po, d, 1, s));
      }
      if (!fuse_t3) { Dx x(d_tp, tpl); x.to(0, nh, d_t2, rl).dact(t2, rl, ACT_ELU); CK(rundx(L_PROP_T3, x, R)); }
      { Dx x(d_t2, rl); x.to(0, nh, d_t1, t1l).dact(t1, t1l, ACT_ELU).dup(d_pre_k + rw, pre_rld); CK(rundx(L_PROP_T2, x, R)); }
      if (c.rnn_cell == RNN_LSTM) {  // d h_k total, then the cell adjoint (second copy: this slot's block of d_pre)
        Dx x(d_t1, t1l);
        x.to(0, nh, b.d_hk, nh).add(b.d_r[k & 1], nh);
        CK(rundx(L_PROP_T1, x, R));
        sq_launch_lstm_cell_bwd(cslotp(w.rgates, 4 * nh, t, 0, k), N * 4 * nh, k == 0 ? w.prop_rnn_init + nh : cslotp(w.rc, nh, t, 0, k - 1),
                                k == 0 ? 0 : rl, b.d_hk, nh, k < N - 1 ? b.d_cs[k & 1] : nullptr, nh, d_rnn, drl, b.d_cs[(k + 1) & 1], nh, R, nh, s,
                                d_pre_k, pre_rld);
      } else if (c.rnn_cell == RNN_GRU) {
        Dx x(d_t1, t1l);
        x.to(0, nh, b.d_hk, nh).add(b.d_r[k & 1], nh);
        CK(rundx(L_PROP_T1, x, R));
        const float* g3 = cslotp(w.rgates, 3 * nh, t, 0, k);
        const float* hp = k == 0 ? w.prop_rnn_init : cslotp(w.r, nh, t, 0, k - 1);
        const int g3l = N * 3 * nh, hpl = k == 0 ? 0 : rl;
        float* dhp = k > 0 ? b.d_r[(k - 1) & 1] : b.d_init_p + (size_t)t * R * nh;
        sq_launch_gru_bwd_a(b.d_hk, nh, g3, g3l, g3 + 2 * nh, g3l, hp, hpl, d_rnn, drl, dhp, nh, R, nh, 0, s, d_pre_k, pre_rld, 2 * nh);
        { Dx y(d_rnn + 2 * nh, drl); y.to(0, nh, b.d_rh, nh); CK(rundx(L_PROP_RNN2, y, R)); }
        sq_launch_gru_bwd_b(b.d_rh, nh, g3 + nh, g3l, hp, hpl, d_rnn, drl, dhp, nh, R, nh, s, d_pre_k + nh, pre_rld);
      } else {  // d r_k total = T1^T + (gate GEMM + next slot's RNN, accumulated in d_r[k & 1]); tanh' -> RNN pre-activation
        Dx x(d_t1, t1l);
        x.to(0, nh, d_rnn, rl).add(b.d_r[k & 1], nh).dact(r_k, rl, ACT_TANH).dup(d_pre_k, pre_rld);
        CK(rundx(L_PROP_T1, x, R));
      }
      if (k > 0) {
        Dx x(d_rnn, drl);
        x.to(0, rec::ZW, d_rec_p_t + (size_t)(k - 1) * RW, N * RW).acc();
        x.to(rec::ZWP, rec::ZWP + nh, b.d_r[(k - 1) & 1], nh);
        if (c.rnn_cell == RNN_GRU) x.acc();
        CK(rundx(L_PROP_RNN, x, R));
      } else {
        Dx x(d_rnn, drl); x.to(rec::ZWP, rec::ZWP + nh, b.d_init_p + (size_t)t * R * nh, nh);
        if (c.rnn_cell == RNN_GRU) x.acc();
        CK(rundx(L_PROP_RNN, x, R));
        if (c.rnn_cell == RNN_LSTM) sq_launch_colsum(b.d_cs[1], nh, R, nh, flat_grad + po.prop_rnn_init + nh, 1, s);
      }
    }
    // ---- D^T. the loop-invariant pre-activation GEMM: segments [m1 nw (pad 64) | z_{t-1} record 56 (pad 64) | temporal nh]
    float* d_m1 = b.d_m1 + (size_t)t * M * M1_LD;
    if (c.time_cell == CELL_LSTM) {  // recurrent rows of the LSTM gates, all slots at once: d h_{t-1} = d gates W_h^T (first writer)
      Dx x(b.d_gru1 + (size_t)t * M * gw, gw); x.to(0, nh, d_tau, snh); CK(rundx(L_PROP_GRU2, x, M));
    }
    {
      Dx x(d_pre, pre_ld);
      x.to(0, nw, d_m1, M1_LD);
      const int o1 = (nw + 15) / 16 * 16, o2 = o1 + rec::ZWP;   // padded segment starts: [m1 nw | record 56 | temporal nh]
      x.to(o1, o1 + rec::ZW, d_rec_prev, RW).acc();
      x.to(o2, o2 + nh, d_tau + d.toff, snh).acc();
      CK(rundx(L_PRE, x, M));
    }
    // ---- C^T. crop #1 and its encoder
    {
      float* d_peb = b.d_peb + (size_t)t * M * nh;
      float* d_pea = b.d_pea + (size_t)t * M * nh;
      const float* peb = w.frame(w.peb, (int64_t)M * nh, t);
      const float* pea = w.frame(w.pea, (int64_t)M * nh, t);
      { Dx x(d_m1, M1_LD); x.to(0, nh, d_peb, nh).dact(peb, nh, ACT_ELU); CK(rundx(L_WHAT_LOC, x, M)); }
      { Dx x(d_peb, nh); x.to(0, nh, d_pea, nh).dact(pea, nh, ACT_ELU); CK(rundx(L_GENC1, x, M)); }
      { Dx x(d_pea, nh); x.to(0, G2, b.d_g1, G2); CK(rundx(L_GENC0, x, M)); }
      CropChainBwdArgs ca; memset(&ca, 0, sizeof(ca));
      ca.mode = CROP_PROP1; ca.img = img; ca.rec_prev = rec_prev; ca.d_rec_prev = d_rec_prev; ca.wb = w.frame(w.wb, (int64_t)M * WB_LD, t);
      ca.wb_ld = WB_LD; ca.d_wb = b.d_wb + (size_t)t * M * WB_LD; ca.mask = c.masked_glimpse ? mask : nullptr; ca.mask_row_mul = N;
      ca.d_mask = d_mask_t; ca.g_out = b.d_g1; ca.g_row_mul = N; ca.flat = flat; ca.flat_grad = flat_grad;
      ca.mask_dact = 1;  // the frame's last contribution to d mask: the sigmoid's adjoint rides on its write (d_mask_t IS d_maskpre)
      CK(sq_launch_crop_chain_bwd(ca, po, d, N, s));
    }
    // ---- B^T. mask MLP and where-bias MLP
    {
      float* d_maskpre = b.d_maskpre + (size_t)t * M * G2;
      float* d_hid1 = b.d_hid1 + (size_t)t * M * 256;
      const float* hid1 = w.frame(w.hid1, (int64_t)M * 256, t);
      if (c.masked_glimpse) {
        Dx x(d_maskpre, G2); x.to(0, 128, d_hid1 + 128, 256).dact(hid1 + 128, 256, ACT_ELU); CK(rundx(L_MASK2, x, M));
      }
      { Dx x(b.d_wb + (size_t)t * M * WB_LD, WB_LD); x.to(0, 128, d_hid1, 256).dact(hid1, 256, ACT_ELU); CK(rundx(L_WB2, x, M)); }
      { Dx x(d_hid1, 256); x.to(0, nh, d_tau + d.toff, snh).acc(); CK(rundx(L_TAU1, x, M)); }
    }
    // ---- A^T. prior cell
    float* d_pgru1 = b.d_pgru1 + (size_t)t * M * pgw;
    if (c.prior_cell == CELL_VANILLA) {
      { Dx x(b.d_pstats + (size_t)t * M * PS_LD, PS_LD);
        x.to(0, nh, d_pgru1, pgw).add(b.d_prior_p, nh).dact(w.frame(w.prior_p, (int64_t)M * nh, t), nh, ACT_TANH);
        CK(rundx(L_PRIOR_LIN, x, M)); }
      Dx x(d_pgru1, pgw);   // [z_{t-1} record 56 (pad 64) | previous state nh]
      x.to(0, rec::ZW, d_rec_prev, RW).acc();
      x.to(rec::ZWP, rec::ZWP + nh, d_pprev, nh);
      CK(rundx(L_PRIOR_GRU1, x, M));
    } else {
    {
      Dx x(b.d_pstats + (size_t)t * M * PS_LD, PS_LD);
      x.to(0, nh, b.dhn, nh).add(b.d_prior_p, psnh);
      if (c.prior_cell == CELL_GRU)
        x.gru_a(w.frame(w.pgz, (int64_t)M * nh, t), nh, w.frame(w.pghc, (int64_t)M * nh, t), nh, prior_prev, nh, d_pgru1, 3 * nh, d_pprev,
                nh, 1, nh);
      CK(rundx(L_PRIOR_LIN, x, M));
    }
    if (c.prior_cell == CELL_LSTM) {
      sq_launch_lstm_cell_bwd(w.frame(w.pgz, (int64_t)M * 4 * nh, t), 4 * nh, prior_prev + nh, psnh, b.dhn, nh, b.d_prior_p + nh, psnh,
                              d_pgru1, pgw, d_pprev + nh, psnh, M, nh, s);
      Dx x(d_pgru1, pgw);   // [z_{t-1} record 56 (pad 64) | previous hidden state nh]
      x.to(0, rec::ZW, d_rec_prev, RW).acc();
      x.to(rec::ZWP, rec::ZWP + nh, d_pprev, psnh);
      CK(rundx(L_PRIOR_GRU1, x, M));
    } else {
      {
        Dx x(d_pgru1 + 2 * nh, 3 * nh);
        x.to(0, nh, b.d_rh, nh).gru_b(w.frame(w.pgr, (int64_t)M * nh, t), nh, prior_prev, nh, d_pgru1, 3 * nh, d_pprev, nh, nh);
        CK(rundx(L_PRIOR_GRU2, x, M));
      }
      {  // [z_{t-1} record 56 (pad 64) | prior state nh]
        Dx x(d_pgru1, 3 * nh);
        x.to(0, rec::ZW, d_rec_prev, RW).acc();
        x.to(rec::ZWP, rec::ZWP + nh, d_pprev, nh).acc();
        CK(rundx(L_PRIOR_GRU1, x, M));
      }
    }
    }
  }
  // ================= initial states, input encoder =================
  {  // one launch (they were seven to nine, each a dependent node of the chain)
    ColsumBatch cs;
    cs.add(b.d_init_p, nh, T * R, nh, flat_grad + po.prop_rnn_init, s);
    cs.add(b.d_init_d, nh, T * R, nh, flat_grad + po.disc_rnn_init, s);
    if (c.rec_where_prior) cs.add(b.d_rn0, 4, T * R, 4, flat_grad + po.rn_init_state, s);
    cs.add(b.d_new_t, snh, T * R, snh, flat_grad + po.temporal_init, s);
    cs.add(b.d_new_p, psnh, T * R, psnh, flat_grad + po.prior_init, s);
    cs.add(b.d_tm(0), snh, M, snh, flat_grad + po.temporal_init, s);
    cs.add(b.d_pm(0), psnh, M, psnh, flat_grad + po.prior_init, s);
    // output layer of the steps predictors (nh/2 -> 1): d w2 = s1h^T d_raw, d b2 = sum d_raw, all uses at once
    const size_t ph1 = (size_t)T * R * N;
    cs.add(w.s1h, S1_LD, (int)ph1, nsp, flat_grad + po.prop_steps_l1_w, s, b.d_raw, 1);
    cs.add(b.d_raw, 1, (int)ph1, 1, flat_grad + po.prop_steps_l1_b, s);
    cs.add(w.s1h + ph1 * S1_LD, S1_LD, (int)ph1, nsp, flat_grad + po.disc_steps_l1_w, s, b.d_raw + ph1, 1);
    cs.add(b.d_raw + ph1, 1, (int)ph1, 1, flat_grad + po.disc_steps_l1_b, s);
    CK(cs.flush(s));
  }
  {
    const int TB = T * B;
    // (the activation adjoints in the dX launches' epilogues, as in the slot chain)
    { Dx x(b.d_pre_disc, rw); x.to(0, nh, b.d_ib, nh).dact(w.ienc_b, nh, ACT_ELU); CK(rundx(L_PREDISC, x, TB)); }
    { Dx x(b.d_ib, nh); x.to(0, nh, b.d_ia, nh).dact(w.ienc_a, nh, ACT_ELU); CK(rundx(L_IENC1, x, TB)); }
    wgrad(L_PREDISC, {{w.ienc_b, nh}}, b.d_pre_disc, rw, TB);
    wgrad(L_IENC1, {{w.ienc_a, nh}}, b.d_ib, nh, TB);
    wgrad(L_IENC0, {{obs, PL}}, b.d_ia, nh, TB);
  }
  // ================= batched weight gradients =================
  {
    const float* tm_all = w.temporal_m;  // [T+1][M][snh]; frames 0..T-1 are the inputs
    const float* tau_all = tm_all + d.toff;
    const float* pm_all = w.prior_m;
    // shifted / gated operand matrices of the recurrent layers' weight gradients, one launch
    const size_t ph1 = (size_t)MT;
    {
      const bool gru = c.rnn_cell == RNN_GRU;  // candidate's recurrent matrix: A = r * h_{k-1}
      ShiftArgs sa;
      sa.p[0] = ShiftPhase{w.rec_p_all, w.r, w.zero_rec, w.prop_rnn_init, gru ? w.rgates + nh : nullptr, b.zs[0], b.rs[0], b.rh[1]};
      sa.p[1] = ShiftPhase{w.rec_d_all, w.r + ph1 * nh, w.disc_init_rec, w.disc_rnn_init, gru ? w.rgates + ph1 * 3 * nh + nh : nullptr,
                           b.zs[1], b.rs[1], b.rh[3]};
      sa.rows = MT; sa.N = N; sa.nh = nh; sa.gate_ld = 3 * nh;
      int ny = 2;
      sa.m[0] = sa.m[1] = MulJob{nullptr, nullptr, nullptr};
      if (c.prior_cell == CELL_GRU) sa.m[ny++ - 2] = MulJob{w.pgr, pm_all, b.rh[0]};
      if (c.time_cell == CELL_GRU) sa.m[ny++ - 2] = MulJob{w.gr, tm_all, b.rh[2]};
      SQ_LAUNCH(k_shift_inputs, dim3(MT, ny), dim3(64), 0, s, sa);
    }
    // prior GRU
    wgrad(L_PRIOR_GRU1, {{w.rec_m_all, RW}, {pm_all, psnh}}, b.d_pgru1, pgw, MT);
    if (c.prior_cell == CELL_GRU) wgrad(L_PRIOR_GRU2, {{b.rh[0], nh}}, b.d_pgru1 + 2 * nh, 3 * nh, MT);   // (b.rh: k_shift_inputs below)
    wgrad(L_PRIOR_LIN, {{w.prior_p, psnh}}, b.d_pstats, PS_LD, MT);
    // where-bias / mask MLPs
    wgrad(L_TAU1, {{tau_all, snh}}, b.d_hid1, 256, MT);
    wgrad(L_WB2, {{w.hid1, 256}}, b.d_wb, WB_LD, MT);
    if (c.masked_glimpse) wgrad(L_MASK2, {{w.hid1 + 128, 256}}, b.d_maskpre, G2, MT);
    // glimpse encoder: crop #1 (per frame, M rows) + both slot phases (2*T*R*N rows)
    wgrad(L_GENC0, {{w.g1, G2}}, b.d_pea, nh, MT);
    wgrad(L_GENC0, {{w.g2, G2}}, b.d_e1, nh, 2 * MT);
    wgrad(L_GENC1, {{w.pea, nh}}, b.d_peb, nh, MT);
    wgrad(L_GENC1, {{w.e1, nh}}, b.d_e2, nh, 2 * MT);
    wgrad(L_WHAT_LOC, {{w.peb, nh}}, b.d_m1, M1_LD, MT);
    wgrad(L_WHAT_HEAD, {{w.e2, nh}}, b.d_enc3, ENC_LD, 2 * MT);
    // loop-invariant pre-activations
    wgrad(L_PRE, {{w.m1, M1_LD}, {w.rec_m_all, RW}, {tau_all, snh}}, b.d_pre, pre_ld, MT);
    // propagation slot chain (phase 0 of the tapes)
    wgrad(L_PROP_RNN, {{b.zs[0], rec::ZWP}, {b.rs[0], nh}}, b.d_rnn, rw, MT);
    if (c.rnn_cell == RNN_GRU) wgrad(L_PROP_RNN2, {{b.rh[1], nh}}, b.d_rnn + 2 * nh, rw, MT);
    wgrad(L_PROP_T1, {{w.r, nh}}, b.d_t1, T1_LD, MT);
    wgrad(L_PROP_T2, {{w.t1, T1_LD}}, b.d_t2, nh, MT);
    wgrad(L_PROP_T3, {{w.t2, nh}}, b.d_tp, TP_LD, MT);
    wgrad(L_PROP_GRU1, {{w.r, nh}, {w.rec_p_all + rec::WHERE, RW}, {w.enc, ENC_LD}}, b.d_gru1, gw, MT);
    if (c.time_cell == CELL_LSTM) {
      wgrad(L_PROP_GRU2, {{tm_all, snh}}, b.d_gru1, gw, MT);   // recurrent rows + b_gates: A = h_{t-1}
    } else if (c.time_cell == CELL_GRU) {
      wgrad(L_PROP_GRU2, {{b.rh[2], nh}}, b.d_gru1 + 2 * nh, 3 * nh, MT);
    }
    wgrad(L_PROP_HEADS, {{w.temporal_p, snh}}, b.d_hraw, HRAW_LD, MT);
    wgrad(L_PROP_S1, {{w.rec_p_all, RW}}, b.d_t1 + nh, T1_LD, MT);
    // latent summary, discovery conditioning
    wgrad(L_LAT0, {{w.rec_p_all, RW}}, b.d_lea, nh, MT);
    wgrad(L_LAT1, {{w.lea, nh}}, b.d_leb, nh, MT);
    wgrad(L_PRED, {{w.c, nh}}, b.d_pre_d, rw, T * R);
    if (c.rec_where_prior) wgrad(L_RNCOND, {{w.rn_init_state, 0}, {w.c, nh}}, b.d_spre, 128, T * R);
    // discovery slot chain (phase 1 of the tapes)
    wgrad(L_DISC_RNN, {{b.zs[1], rec::ZWP}, {b.rs[1], nh}}, b.d_rnn + ph1 * rw, rw, MT);
    if (c.rnn_cell == RNN_GRU) wgrad(L_DISC_RNN2, {{b.rh[3], nh}}, b.d_rnn + ph1 * rw + 2 * nh, rw, MT);
    wgrad(L_DISC_T1, {{w.r + ph1 * nh, nh}}, b.d_t1 + ph1 * T1_LD, T1_LD, MT);
    wgrad(L_DISC_T2, {{w.t1 + ph1 * T1_LD, T1_LD}}, b.d_t2 + ph1 * nh, nh, MT);
    wgrad(L_DISC_T3, {{w.t2 + ph1 * nh, nh}}, b.d_tp + ph1 * TP_LD, TP_LD, MT);
    wgrad(L_DISC_S1, {{w.rec_d_all, RW}}, b.d_t1 + ph1 * T1_LD + nh, T1_LD, MT);
    sq_launch_where_param_grads(b.d_tp, TP_LD, b.d_rec_p, w.rec_p_all, noise, T, d, flat_grad, po, s);
  }
  CK(wbatch.flush(s));
  if (h->padded) sq_flat_gather(h, flat_grad, user_grad, packedv, s);
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Generic graph capture: the host brackets any sequence of C-ABI calls on one stream (e.g. sqair_forward_train,
// sqair_elbo, sqair_backward = one training step up to the gradient all-reduce) and replays it as ONE hipGraphLaunch —
// the step is ~3000 short dependent launches, so replay removes the host launch cost from the critical path.
// ------------------------------------------------------------------------------------------------
extern "C" int sqair_capture_begin(SqairHandle* h, void* stream) {
  if (!h) return -1;
  SQ_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return 0;
}
extern "C" int sqair_capture_end(SqairHandle* h, void* stream, int slot) {
  if (!h || slot < 0 || slot >= 4) return -1;
  hipGraph_t g = nullptr;
  SQ_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  if (h->cap_exec[slot]) { (void)hipGraphExecDestroy(h->cap_exec[slot]); h->cap_exec[slot] = nullptr; }
  if (h->cap_graph[slot]) { (void)hipGraphDestroy(h->cap_graph[slot]); h->cap_graph[slot] = nullptr; }
  h->cap_graph[slot] = g;
  size_t nn = 0;
  SQ_CHECK_HIP(hipGraphGetNodes(g, nullptr, &nn));
  SQ_CHECK_HIP(hipGraphInstantiate(&h->cap_exec[slot], g, nullptr, nullptr, 0));
  return (int)nn;
}
extern "C" int sqair_capture_launch(SqairHandle* h, int slot, void* stream) {
  if (!h || slot < 0 || slot >= 4 || !h->cap_exec[slot]) {
    sq_set_error(h, "sqair_capture_launch: empty slot");
    return -1;
  }
  SQ_CHECK_HIP(hipGraphLaunch(h->cap_exec[slot], (hipStream_t)stream));
  return 0;
}
// grad += l2 * theta  (targets.l2_reg, sqair/targets.py:31-35: weight * sum_v l2_loss(v) over ALL trainable variables)
__global__ void k_add_l2(const float* __restrict__ theta, float* __restrict__ grad, int64_t n, float l2 SQ_TLP) {
  SQ_TL_SCOPE;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) grad[i] += l2 * theta[i];
}
extern "C" int sqair_add_l2_grad(SqairHandle* h, const float* flat_params, float* flat_grad, int64_t n, float l2, void* stream) {
  if (!h || !flat_params || !flat_grad || n < 0) return -1;
  if (n == 0 || l2 == 0.0f) return 0;
  SQ_LAUNCH(k_add_l2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat_params, flat_grad, n, l2);
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Noise of a pass generated on the device ("noise blobs or Philox (seed, offset)"): eps ~ N(0, 1) for the
// Normal samples, u ~ U[0, 1) for the presence Bernoullis, layout [T, B', 2, N, 4 + n_what + 1].  Philox4x32-10 (Salmon et
// al., SC'11), counter = (global element index, step), key = seed: every element is a pure function of (seed, step,
// its position in the GLOBAL batch), so a rank that owns sequences [b0, b0 + B) of a global batch draws exactly the rows it
// would have seen on one GPU (the reference draws inside the TF graph: tfd .sample() at core.py:226, modules.py:60, :485).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ void k_fill_noise(float* __restrict__ noise, int64_t n_local, int64_t per_frame_local, int64_t per_frame_global,
                             int64_t row0_elems, int nzw, unsigned long long seed, unsigned long long step SQ_TLP) {
  SQ_TL_SCOPE;
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // quad of 4 consecutive LOCAL elements
  if (q * 4 >= n_local) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t e = q * 4 + j;
    if (e >= n_local) break;
    const int64_t t = e / per_frame_local, within = e - t * per_frame_local;
    const unsigned long long g = (unsigned long long)(t * per_frame_global + row0_elems + within);  // global element index
    unsigned r[4];  // one Philox block per element (counter = element index): elements are independent by construction
    philox4x32_10((unsigned)g, (unsigned)(g >> 32), (unsigned)step, (unsigned)(step >> 32), (unsigned)seed, (unsigned)(seed >> 32), r);
    float v;
    if ((int)(g % (unsigned long long)nzw) == nzw - 1) {
      v = (float)(r[2] >> 8) * (1.0f / 16777216.0f);  // u in [0, 1), 24 bits
    } else {  // Box-Muller, u1 in (0, 1]
      const float u1 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
      v = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    noise[e] = v;
  }
}
extern "C" int sqair_fill_noise(SqairHandle* h, float* noise, int T, int B, int global_B, int b0, uint64_t seed, uint64_t step,
                                void* stream) {
  if (!h || !noise || T < 1 || B < 1 || global_B < B || b0 < 0 || b0 + B > global_B) return -1;
  const SqairConfig& c = h->cfg;
  const int nzw = 4 + c.n_what + 1;
  const int64_t per_row = (int64_t)2 * c.n_steps_per_image * nzw;
  const int64_t per_frame_local = (int64_t)B * c.k_particles * per_row, per_frame_global = (int64_t)global_B * c.k_particles * per_row;
  const int64_t n_local = (int64_t)T * per_frame_local;
  const int64_t quads = (n_local + 3) / 4;
  SQ_LAUNCH(k_fill_noise, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, noise, n_local,
                     per_frame_local, per_frame_global, (int64_t)b0 * c.k_particles * per_row, nzw, (unsigned long long)seed,
                     (unsigned long long)step);
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}
