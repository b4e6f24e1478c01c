// Non-GEMM kernels of the SQAIR hot path: spatial-transformer crop / insert, sampling, presence,
// log-probabilities with the presence mask applied in-kernel, slot compaction and the IWAE / VIMCO
// reductions.  All HBM-bound gather / scatter / reduce work (class 2).
#include "sqair_glue.h"
#include "sqair_rowops.h"
#include "sqair_canvas.h"

// ------------------------------------------------------------------------------------------------
// initial recurrent state (reference: sqair/seq.py:86-100, sqair_modules.py:352-366, core.py:156-162)
// ------------------------------------------------------------------------------------------------
__global__ void k_init_state(float* __restrict__ rec_m, float* __restrict__ temporal_m, float* __restrict__ prior_m,
                             float* __restrict__ last_id, float* __restrict__ disc_init_rec,
                             float* __restrict__ prop_rnn_init, float* __restrict__ disc_rnn_init,
                             float* __restrict__ rn_init_state, float* __restrict__ w3_prop,
                             float* __restrict__ w3_disc, int w3p_off, int w3d_off,
                             const float* __restrict__ flat, POff po, Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  const int rs = blockIdx.x;  // row*N + slot
  const int tid = threadIdx.x;
  for (int i = tid; i < rec::W; i += blockDim.x) rec_m[(size_t)rs * rec::W + i] = (i == rec::ID) ? -1.0f : 0.0f;
  // LSTM: [hidden | cell] initial states are adjacent in the flat buffer (seq.temporal_init, seq.temporal_init_c)
  for (int i = tid; i < d.snh; i += blockDim.x) temporal_m[(size_t)rs * d.snh + i] = flat[po.temporal_init + i];
  for (int i = tid; i < d.psnh; i += blockDim.x) prior_m[(size_t)rs * d.psnh + i] = flat[po.prior_init + i];
  if (tid == 0 && (rs % d.N) == 0) last_id[rs / d.N] = -1.0f;
  if (rs == 0) {  // aligned copies of the small trainable initial states (GEMM A-operand contract)
    if (tid == 0) disc_init_rec[rec::PRES] = 1.0f;
    for (int i = tid; i < d.rsnh; i += blockDim.x) {  // LSTM: [hidden | cell] initial rows are adjacent in the flat buffer
      prop_rnn_init[i] = flat[po.prop_rnn_init + i];
      disc_rnn_init[i] = flat[po.disc_rnn_init + i];
    }
    if (tid < 4) rn_init_state[tid] = flat[po.rn_init_state + tid];
  }
  if (rs == 1 || (rs == 0 && d.R * d.N == 1)) {  // transform.l2 {w [nh,8], b [8]} are adjacent in the flat buffer
    for (int i = tid; i < d.nh * 8 + 8; i += blockDim.x) {
      w3_prop[i] = flat[w3p_off + i];
      w3_disc[i] = flat[w3d_off + i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LSTM cell epilogue of the propagation temporal cell (time_transition=LSTM; snt.LSTM, dm_sonnet 1.14: gates =
// [x, h] w_gates + b_gates split (i, j, f, o); c' = sigmoid(f + 1) c + sigmoid(i) tanh(j); h' = tanh(c') sigmoid(o)).
// `gates` holds the finished pre-activations of one slot, [rows][4 nh]; the new state goes to [h' | c'].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lstm_cell(const float* __restrict__ gates, int g_ld, const float* __restrict__ c_prev,
                                                   int c_ld, float* __restrict__ h_out, int h_ld, float* __restrict__ c_out,
                                                   int co_ld, int rows, int nh SQ_TLP) {
  SQ_TL_SCOPE;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * nh) return;
  const int r = e / nh, q = e - r * nh;
  const float* g = gates + (size_t)r * g_ld;
  const float gi = g[q], gj = g[nh + q], gf = g[2 * nh + q], go = g[3 * nh + q];
  const float c = sq_sigmoid(gf + 1.0f) * c_prev[(size_t)r * c_ld + q] + sq_sigmoid(gi) * sq_tanh(gj);
  h_out[(size_t)r * h_ld + q] = sq_tanh(c) * sq_sigmoid(go);
  c_out[(size_t)r * co_ld + q] = c;
}
// c_ld = 0 broadcasts one initial cell row; hidden and cell may live in different buffers (slot RNN) or side by side
int sq_launch_lstm_cell2(const float* gates, int g_ld, const float* c_prev, int c_ld, float* h_out, int h_ld, float* c_out, int co_ld,
                         int rows, int nh, hipStream_t s) {
  SQ_LAUNCH(k_lstm_cell, dim3((rows * nh + 255) / 256), dim3(256), 0, s, gates, g_ld, c_prev, c_ld, h_out, h_ld, c_out, co_ld,
                     rows, nh);
  return 0;
}
int sq_launch_lstm_cell(const float* gates, int g_ld, const float* c_prev, int c_ld, float* state_out, int o_ld, int rows, int nh,
                        hipStream_t s) {
  return sq_launch_lstm_cell2(gates, g_ld, c_prev, c_ld, state_out, o_ld, state_out + nh, o_ld, rows, nh, s);
}

int sq_launch_init_state(float* rec_m, float* temporal_m, float* prior_m, float* last_id, float* disc_init_rec,
                         float* prop_rnn_init, float* disc_rnn_init, float* rn_init_state, float* w3_prop, float* w3_disc,
                         int w3p_off, int w3d_off, const float* flat, POff po, Dims d, hipStream_t s) {
  SQ_LAUNCH(k_init_state, dim3(d.R * d.N), dim3(256), 0, s, rec_m, temporal_m, prior_m, last_id, disc_init_rec,
                     prop_rnn_init, disc_rnn_init, rn_init_state, w3_prop, w3_disc, w3p_off, w3d_off, flat, po, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Spatial-transformer crop (reference: sqair/modules.py:170-227; Sonnet AffineGridWarper +
// tf.contrib.resampler).  One workgroup per sequence b stages the frame in LDS
// once and cuts the glimpses of all K particles of that sequence from it.  The `where` sample of
// the propagation / discovery core is drawn in the same launch (core.py:323-334, :217-227).
// ------------------------------------------------------------------------------------------------

// one workgroup per particle row (a per-sequence kernel staging the frame in LDS for its K particles had only B = 32
// workgroups at the headline config and measured 10.3 us; this one 5-6 us with the frame served by L1 / L2)
template <bool STAGED>
__global__ __launch_bounds__(256) void k_crop_row(const CropArgs a, const POff po, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  x_crop_row<LdPlain, STAGED>(a, po, d, sq_row_of_wg(blockIdx.x, d), a.mode == CROP_PROP1 ? (int)blockIdx.y : a.slot, smem);
}

int sq_launch_crop(const CropArgs& a, POff po, Dims d, int nslots, hipStream_t s) {
  const bool staged = d.H * d.W <= SQ_CROP_STAGE_MAX_PIXELS;
  const size_t shm = (4 + (size_t)4 * d.G + (staged ? (size_t)d.H * d.W : 0)) * sizeof(float);
  if (staged) SQ_LAUNCH(k_crop_row<true>, dim3(d.R, nslots), dim3(256), shm, s, a, po, d);
  else SQ_LAUNCH(k_crop_row<false>, dim3(d.R, nslots), dim3(256), shm, s, a, po, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Tail of one propagation / discovery slot, fused (was: what-sample kernel, a dense launch, presence kernel):
//   what ~ N(.)           reference: sqair/core.py:336-359 (propagation, gated), :213-215 (discovery)
//   hidden = elu(s1p + what W_what)   StepsPredictor hidden layer; the r_k / temporal-state terms of its
//                         pre-activation (s1p) were produced by earlier launches (sqair/modules.py:506-524)
//   logit, presence       sqair/core.py:141-144
// One workgroup = 16 rows; 4 waves x 2 column tiles x 4 K-chunks of v_mfma_f32_16x16x4_f32 with the A operand
// read from an LDS tile of the freshly sampled `what`.
// ------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#ifdef SQAIR_WIDE
// The wide build's tail (n_what up to 128, n_hidden up to 512): the same arithmetic in the same order -- what sample, hidden
// layer = s1p + what W_what on the matrix cores with the A operand from an LDS tile of the fresh sample, output dot, presence --
// written as plain loops (operands fetched where they are used, no prefetch choreography): a slow path by design.
constexpr int ZLD = rec::ZWP + 4;
template <bool FULL_Z>
__device__ __forceinline__ void tail_body(const TailArgs& a, const Dims& d, const int row0, const bool STORE, float* zt, float (*rs)[16]) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int nw = d.nw, nsp = d.nh / 2;
  const int n_tiles = nsp / 16;          // wave w owns tiles w, w + 4, ...
  constexpr int ZCH = rec::ZWP / 16;     // K chunks of the z-record tile
  const f32x4_t* wp4 = reinterpret_cast<const f32x4_t*>(a.wp);
  const int pr = min(row0 + (tid & 15), d.R - 1);
  const float b2 = a.flat[a.b2_off];
  const float u = a.noise[(((size_t)pr * 2 + (a.is_disc ? 1 : 0)) * d.N + a.slot) * d.nzw + 4 + nw];
  float prev = 1.0f;
  if (!(a.is_disc && a.slot == 0))
    prev = a.is_disc ? a.rec_new[((size_t)pr * d.N + a.slot - 1) * rec::W + rec::PRES] : a.rec_prev[((size_t)pr * d.N + a.slot) * rec::W + rec::PRES];
  for (int i = tid; i < 16 * ZLD; i += 256) zt[i] = 0.0f;
  __syncthreads();
  for (int e = tid; e < 16 * nw; e += 256) {
    const int rr = e / nw, c = e - rr * nw;
    const int r = min(row0 + rr, d.R - 1);
    float loc = a.enc[(size_t)r * a.enc_ld + c], sc = a.enc[(size_t)r * a.enc_ld + nw + c];
    const float eps = a.noise[(((size_t)r * 2 + (a.is_disc ? 1 : 0)) * d.N + a.slot) * d.nzw + 4 + c];
    if (!a.is_disc) {
      const float* hr = a.hraw + (size_t)r * a.h_ld;
      const float tm1 = a.rec_prev[((size_t)r * d.N + a.slot) * rec::W + rec::WHAT + c];
      const float t_loc = hr[c];
      const float t_scale = sq_softplus(hr[nw + c]) + 1e-2f;
      const float fg = sq_gate(hr[2 * nw + c]);
      const float om_ig = sq_gate_compl(hr[3 * nw + c]), om_tg = sq_gate_compl(hr[4 * nw + c]);
      const float l2 = sq_mix3(fg, tm1, om_ig, loc, om_tg, t_loc);
      sc = sq_mix2(om_ig, sc, om_tg, t_scale);
      loc = l2;
    }
    const float what = loc + sc * eps;
    zt[rr * ZLD + rec::WHAT + c] = what;
    if (STORE && row0 + rr < d.R) {
      float* rn = a.rec_new + ((size_t)(row0 + rr) * d.N + a.slot) * rec::W;
      rn[rec::WHAT + c] = what;
      rn[rec::WHAT_LOC + c] = loc;
      rn[rec::WHAT_SCALE + c] = sc;
    }
  }
  __syncthreads();
  float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int tile = wave; tile < n_tiles; tile += 4) {
    const int col = tile * 16 + (lane & 15);
    f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c = 0; c < ZCH; ++c) {
      const f32x4_t av = *reinterpret_cast<const f32x4_t*>(&zt[(lane & 15) * ZLD + 16 * c + 4 * kq]);
      const f32x4_t bv = wp4[(size_t)(tile * ZCH + c) * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
    }
    const float w2v = a.flat[a.w2_off + col];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float hv = sq_elu(acc[i] + a.s1p[(size_t)min(row0 + 4 * kq + i, d.R - 1) * a.s1p_ld + col]);
      part[i] += hv * w2v;
      if (STORE && a.s1h_out != nullptr && row0 + 4 * kq + i < d.R) a.s1h_out[(size_t)(row0 + 4 * kq + i) * a.s1h_ld + col] = hv;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) part[i] = sq_row_sum(part[i]);
  if ((lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rs[wave][4 * kq + i] = part[i];
  }
  __syncthreads();
  if (tid < 16) {
    const float raw = rs[0][tid] + rs[1][tid] + rs[2][tid] + rs[3][tid] + b2;
    const float logit = prev * raw + (prev - 1.0f) * 88.0f;
    const float prob = sq_sigmoid(logit);
    const float pres = (u < prob ? 1.0f : 0.0f) * prev;
    if (STORE && row0 + tid < d.R) {
      float* rn = a.rec_new + ((size_t)(row0 + tid) * d.N + a.slot) * rec::W;
      rn[rec::PRES] = pres;
      rn[rec::LOGIT] = logit;
      rn[rec::PROB] = prob;
    }
  }
  (void)FULL_Z;
}
#else
constexpr int ZLD = 68;
// Body shared by k_slot_tail (one workgroup per 16 rows, results to memory) and k_rnn_tail (every workgroup of the NEXT slot's RNN
// layer re-derives the tail of its 16 rows and keeps the z-record in LDS as the layer's first A segment).  FULL_Z: also place
// where / presence / logit into the LDS tile `zt` (the what columns are always there); STORE: write the results to memory.
// WD ("what done"): the slot's what sample was written by the layer that produced its operands (k_linear_what below) -- the tail
// reads it instead of deriving it.  A TEMPLATE parameter: with both variants in one kernel the training pass, which never takes
// the short one, ran 1 % slower (the slot loop's kernels are sensitive to their code size).
template <bool FULL_Z, bool WD>
__device__ __forceinline__ void tail_body(const TailArgs& a, const Dims& d, const int row0, const bool STORE, float* zt, float (*rs)[16]) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int nw = d.nw, nsp = d.nh / 2;
  const int n_tiles = nsp / 16;  // 8 for nh = 256; wave w owns tiles w, w + 4, ...
  // ---- requests that do not depend on the sample: weights, partial pre-activations, w2, Bernoulli operands
  const f32x4_t* wp4 = reinterpret_cast<const f32x4_t*>(a.wp);
  f32x4_t bv[2][4];
  float sp[2][4], w2v[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tile = min(wave + 4 * t, n_tiles - 1);
    const int col = tile * 16 + (lane & 15);
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[t][c] = wp4[(size_t)(tile * 4 + c) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) sp[t][i] = a.s1p[(size_t)min(row0 + 4 * kq + i, d.R - 1) * a.s1p_ld + col];
    w2v[t] = a.flat[a.w2_off + col];
  }
  __builtin_amdgcn_sched_barrier(0);   // (see k_rnn_tail: issue this group's loads before the next group's address arithmetic)
  const int pr = min(row0 + (tid & 15), d.R - 1);
  const float b2 = a.flat[a.b2_off];
  const float u = a.noise[(((size_t)pr * 2 + (a.is_disc ? 1 : 0)) * d.N + a.slot) * d.nzw + 4 + nw];
  const float* prevp = a.is_disc ? (a.slot == 0 ? &a.flat[a.b2_off] : a.rec_new + ((size_t)pr * d.N + a.slot - 1) * rec::W + rec::PRES)
                                 : a.rec_prev + ((size_t)pr * d.N + a.slot) * rec::W + rec::PRES;
  float prev = *prevp;
  if (a.is_disc && a.slot == 0) prev = 1.0f;
  float whv[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // where sample of the slot (written by its crop launch): requested with the rest,
  if (FULL_Z) {                              // not behind the last barrier where it cost a memory round trip of its own; by EVERY
    // thread (only 16 use it): as a guarded load (`tid < 16`) hipcc put a wait for all earlier requests behind it, and the
    // what-sample operands below went out a memory round trip late
    const float* wh = a.rec_new + ((size_t)pr * d.N + a.slot) * rec::W + rec::WHERE;
#pragma unroll
    for (int q = 0; q < 4; ++q) whv[q] = wh[q];
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- what sample for the 16 rows (nw elements each), operands requested in one burst
  constexpr int EPT = 4;
  const int nel = 16 * nw;
  float v_loc[EPT], v_sc[EPT], v_eps[EPT], v_h[EPT][5], v_tm1[EPT];
  constexpr bool what_done = WD;
  if (what_done) {
    // the sample was written by the layer that produced its operands (k_linear_what): one load per element
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = min(tid + 256 * q, nel - 1);
      const int rr = sq_div(e, d.nw_mul), c = e - rr * nw;
      const int r = min(row0 + rr, d.R - 1);
      v_loc[q] = a.rec_new[((size_t)r * d.N + a.slot) * rec::W + rec::WHAT + c];
      v_sc[q] = 0.0f; v_eps[q] = 0.0f; v_tm1[q] = 0.0f;
#pragma unroll
      for (int g = 0; g < 5; ++g) v_h[q][g] = 0.0f;
    }
  }
  if (!what_done) {
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = min(tid + 256 * q, nel - 1);
    const int rr = sq_div(e, d.nw_mul), c = e - rr * nw;
    const int r = min(row0 + rr, d.R - 1);
    v_loc[q] = a.enc[(size_t)r * a.enc_ld + c];
    v_sc[q] = a.enc[(size_t)r * a.enc_ld + nw + c];
    v_eps[q] = a.noise[(((size_t)r * 2 + (a.is_disc ? 1 : 0)) * d.N + a.slot) * d.nzw + 4 + c];
    if (!a.is_disc) {
      const float* hr = a.hraw + (size_t)r * a.h_ld;
#pragma unroll
      for (int g = 0; g < 5; ++g) v_h[q][g] = hr[g * nw + c];
      v_tm1[q] = a.rec_prev[((size_t)r * d.N + a.slot) * rec::W + rec::WHAT + c];
    } else {
#pragma unroll
      for (int g = 0; g < 5; ++g) v_h[q][g] = 0.0f;
      v_tm1[q] = 0.0f;
    }
  }
  }
  for (int i = tid; i < 16 * ZLD; i += 256) {  // everything but the `what` columns of the tile is zero
    const int c = i % ZLD;
    if (c < rec::WHAT || c >= rec::WHAT + nw) zt[i] = 0.0f;
  }
  if (what_done) {
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int e = tid + 256 * q;
      if (e < nel) {
        const int rr = sq_div(e, d.nw_mul), c = e - rr * nw;
        zt[rr * ZLD + rec::WHAT + c] = v_loc[q];
      }
    }
  }
  if (!what_done) {
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = tid + 256 * q;
    if (e < nel) {
      const int rr = sq_div(e, d.nw_mul), c = e - rr * nw;
      float loc, sc;
      if (a.is_disc) {
        loc = v_loc[q];
        sc = v_sc[q];
      } else {
        const float t_loc = v_h[q][0];
        const float t_scale = sq_softplus(v_h[q][1]) + 1e-2f;
        const float fg = sq_gate(v_h[q][2]);
        const float om_ig = sq_gate_compl(v_h[q][3]), om_tg = sq_gate_compl(v_h[q][4]);   // 1 - input gate, 1 - temporal gate
        loc = sq_mix3(fg, v_tm1[q], om_ig, v_loc[q], om_tg, t_loc);
        sc = sq_mix2(om_ig, v_sc[q], om_tg, t_scale);
      }
      const float what = loc + sc * v_eps[q];
      zt[rr * ZLD + rec::WHAT + c] = what;
      if (STORE && row0 + rr < d.R) {
        float* rn = a.rec_new + ((size_t)(row0 + rr) * d.N + a.slot) * rec::W;
        rn[rec::WHAT + c] = what;
        rn[rec::WHAT_LOC + c] = loc;
        rn[rec::WHAT_SCALE + c] = sc;
      }
    }
  }
  }
  __syncthreads();
  // ---- hidden layer: acc = what W_what  (K = 56 padded to 64)
  f32x4_t acc[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4_t av = *reinterpret_cast<const f32x4_t*>(&zt[(lane & 15) * ZLD + 16 * c + 4 * kq]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[t][c].x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[t][c].y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[t][c].z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[t][c].w, acc[t], 0, 0, 0);
    }
  }
  // ---- output layer: sum over the nh/2 hidden units of elu(.) * w2; lane holds rows 4 kq + i, one column per tile
  float part[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (wave + 4 * t < n_tiles) {
        const float hv = sq_elu(acc[t][i] + sp[t][i]);
        v += hv * w2v[t];
        if (STORE && a.s1h_out != nullptr && row0 + 4 * kq + i < d.R)
          a.s1h_out[(size_t)(row0 + 4 * kq + i) * a.s1h_ld + (wave + 4 * t) * 16 + (lane & 15)] = hv;
      }
    part[i] = sq_row_sum(v);
  }
  if ((lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rs[wave][4 * kq + i] = part[i];
  }
  __syncthreads();
  if (tid < 16) {
    const float raw = rs[0][tid] + rs[1][tid] + rs[2][tid] + rs[3][tid] + b2;
    const float logit = prev * raw + (prev - 1.0f) * 88.0f;
    const float prob = sq_sigmoid(logit);
    const float pres = (u < prob ? 1.0f : 0.0f) * prev;
    if (STORE && row0 + tid < d.R) {
      float* rn = a.rec_new + ((size_t)(row0 + tid) * d.N + a.slot) * rec::W;
      rn[rec::PRES] = pres;
      rn[rec::LOGIT] = logit;
      rn[rec::PROB] = prob;
    }
    if (FULL_Z) {  // (after the hidden-layer MFMAs have read the tile: they want zeros outside the what columns)
      zt[tid * ZLD + rec::PRES] = pres;
      zt[tid * ZLD + rec::LOGIT] = logit;
#pragma unroll
      for (int q = 0; q < 4; ++q) zt[tid * ZLD + rec::WHERE + q] = whv[q];
    }
  }
}

#endif

template <bool WD>
__global__ __launch_bounds__(256) void k_slot_tail(const TailArgs a, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ __attribute__((aligned(16))) float zt[16 * ZLD];
  __shared__ float rs[4][16];
#ifdef SQAIR_WIDE
  tail_body<false>(a, d, blockIdx.x * 16, true, zt, rs);   // (the wide build has no what fusion: WD is always false there)
#else
  tail_body<false, WD>(a, d, blockIdx.x * 16, true, zt, rs);
#endif
}

#ifndef SQAIR_WIDE
// ------------------------------------------------------------------------------------------------
// Slot tail fused INTO the next slot's RNN layer (VanillaRNN slot cell, sqair/core.py:187-189, :304-305): the layer's first A
// segment is the z-record of the slot that has just been finished, and the tail that completes it (what sample, steps-predictor
// hidden layer, presence) is 16 rows of element-wise work plus a 16 x 64 x 128 MFMA product.  Every workgroup of the layer (one
// 16 x 16 output tile; 16 of them share a row tile) re-derives the tail of its rows and keeps the z-record in LDS, the column-tile-0
// workgroups write it to memory for the later consumers.  One dependent launch (~5.6 us as a graph node) less per slot, paid with
// ~1 us of redundant work inside the layer.  The RNN part is k_linear's arithmetic (4 waves split the K chunks g = wave + 4 i in
// order, two accumulators, LDS reduce), so the layer's result does not depend on whether the tail was fused.
// ------------------------------------------------------------------------------------------------
template <int NH, int TN, bool WD>  // hidden-state chunks per wave = nh / 64; TN column tiles per workgroup (the tail is derived once for them); WD: tail_body
__global__ __launch_bounds__(256) void k_rnn_tail(const TailArgs ta, const Dims d, const float* __restrict__ hid, const int hid_ld,
                                                  const float* __restrict__ wp0, const float* __restrict__ bias,
                                                  const float* __restrict__ add, const int add_ld, float* __restrict__ out,
                                                  const int out_ld, const int n_out, unsigned long long* __restrict__ prof_ts SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ __attribute__((aligned(16))) float zt[16 * ZLD];
  __shared__ float rs[4][16];
  __shared__ float red[TN * 4 * 256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, l15 = lane & 15;
  const int tile_n0 = blockIdx.x * TN, row0 = blockIdx.y * 16;
  const int n_tiles = (n_out + 15) >> 4;
  constexpr int KC = 4 + 4 * NH;
  unsigned long long t_start = 0;
  if (prof_ts != nullptr && tid == 0) t_start = wall_clock64();
  // operands of the layer that do not depend on the tail: weights of my chunks, the hidden-state segment, the epilogue operands
  const float* hrow = hid + (size_t)min(row0 + l15, d.R - 1) * hid_ld;
  f32x4_t bz[TN], bh[TN][NH], ah[NH];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const f32x4_t* __restrict__ wp = reinterpret_cast<const f32x4_t*>(wp0) + ((size_t)min(tile_n0 + t, n_tiles - 1) * KC) * 64 + lane;
    bz[t] = wp[(size_t)wave * 64];
#pragma unroll
    for (int i = 0; i < NH; ++i) bh[t][i] = wp[(size_t)(wave + 4 * (i + 1)) * 64];
  }
#pragma unroll
  for (int i = 0; i < NH; ++i) ah[i] = *reinterpret_cast<const f32x4_t*>(hrow + (wave + 4 * i) * 16 + kq * 4);
  // (fences: hipcc otherwise computes the addresses of ALL ~70 loads of the kernel -- 180 instructions of 64-bit address
  // arithmetic -- before it issues the first one; each group of requests goes out as soon as its own addresses are known)
  __builtin_amdgcn_sched_barrier(0);
  const int m = row0 + (tid >> 4);
  const int mc = min(m, d.R - 1);
  float p_bias[TN], p_add[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int nc = min((tile_n0 + t) * 16 + (tid & 15), n_out - 1);
    p_bias[t] = bias[nc];
    p_add[t] = add[(size_t)mc * add_ld + nc];
  }
  __builtin_amdgcn_sched_barrier(0);
  tail_body<true, WD>(ta, d, row0, blockIdx.x == 0, zt, rs);
  __syncthreads();
  const f32x4_t az = *reinterpret_cast<const f32x4_t*>(&zt[l15 * ZLD + 16 * wave + 4 * kq]);
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    f32x4_t acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(az.x, bz[t].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(az.y, bz[t].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(az.z, bz[t].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(az.w, bz[t].w, acc1, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i].x, bh[t][i].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i].y, bh[t][i].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i].z, bh[t][i].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i].w, bh[t][i].w, acc1, 0, 0, 0);
    }
    float* r = red + t * 1024 + wave * 256;
    r[(4 * kq + 0) * 16 + l15] = acc0.x + acc1.x;
    r[(4 * kq + 1) * 16 + l15] = acc0.y + acc1.y;
    r[(4 * kq + 2) * 16 + l15] = acc0.z + acc1.z;
    r[(4 * kq + 3) * 16 + l15] = acc0.w + acc1.w;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = (tile_n0 + t) * 16 + (tid & 15);
    const float* rt = red + t * 1024;
    if (m < d.R && n < n_out && tile_n0 + t < n_tiles)
      out[(size_t)m * out_ld + n] = sq_tanh(rt[tid] + rt[256 + tid] + rt[512 + tid] + rt[768 + tid] + p_bias[t] + p_add[t]);
  }
  if (prof_ts != nullptr) {
    __syncthreads();
    // (stamped by the last column-tile workgroup of every row tile and by workgroup (0, 0) only: one atomic pair per workgroup
    // serialises thousands of them on one address and made the many-workgroup launches look 2-3x longer than they are)
    if (tid == 0 && (blockIdx.x == gridDim.x - 1 || (blockIdx.x == 0 && blockIdx.y == 0))) {
      atomicMin(prof_ts, t_start);
      atomicMax(prof_ts + 4096, wall_clock64());
    }
  }
}

#endif

#ifndef SQAIR_WIDE
// ------------------------------------------------------------------------------------------------
// A slot's what sample in the epilogue of the dense layer that produces its operands (WhatArgs, sqair_glue.h).  The GEMM part is
// k_linear's (sqair_linear_kernel.inc: one 16 x 16 tile per workgroup, four waves on the K chunks g = wave + 4 j, two accumulators,
// LDS reduce in wave order), so a column's sum is the same number the plain layer writes; the packs put the 2 (discovery) / 5
// (propagation) pre-activations of one `what` element into adjacent lanes of a row, which exchange them with wave shuffles.
//   MODE 0  glimpse-encoder head of a DISCOVERY slot (sqair/core.py:226-229): columns (loc_c, raw scale_c), 8 elements per tile;
//           scale = softplus(.) + 1e-2, what = loc + scale eps -> rec_d.{what, what_loc, what_scale}
//   MODE 1  heads of a PROPAGATION slot (sqair/core.py:336-359): columns (t_loc, t_scale, forget, input, temporal gate) of element
//           c, 3 elements per tile; the gated mixture with the glimpse encoder's Gaussian and what_{t-1} -> rec_p.{...}
// Arithmetic = tail_body's, expression by expression.
// ------------------------------------------------------------------------------------------------
template <int NCH, int MODE>
__global__ __launch_bounds__(256) void k_linear_what(const WhatArgs a SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ float red[4 * 256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, l15 = lane & 15;
  const int tile_n = blockIdx.x, tile_m = blockIdx.y;
  const int arow = min(tile_m * 16 + l15, a.M - 1);
  const f32x4_t* __restrict__ wp = reinterpret_cast<const f32x4_t*>(a.wp) + ((size_t)tile_n * a.kc) * 64 + lane;
  const f32x4_t* __restrict__ wz = reinterpret_cast<const f32x4_t*>(a.wzero) + lane;
  const int nmine = (a.kc - wave + 3) >> 2;
  const float* __restrict__ rp = a.x + (size_t)arow * a.x_ld + kq * 4;
  f32x4_t av[NCH], bv[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const bool valid = j < nmine;
    const int g = valid ? wave + 4 * j : wave;
    av[j] = *reinterpret_cast<const f32x4_t*>(rp + g * 16);
    bv[j] = *(valid ? wp + (size_t)g * 64 : wz);
  }
  __builtin_amdgcn_sched_barrier(0);
  // epilogue operands, requested behind the operand loads and ahead of the MFMAs (clamped addresses, every lane)
  const int m = tile_m * 16 + (tid >> 4), col = tid & 15, mc = min(m, a.M - 1);
  const float p_bias = a.bias[tile_n * 16 + col];
  const int c = MODE == 0 ? tile_n * 8 + (col >> 1) : tile_n * 3 + col / 5;
  const int g5 = MODE == 0 ? (col & 1) : col % 5;
  const bool valid = c < a.nw && (MODE == 0 || col < 15) && m < a.M;
  const int cc = min(c, a.nw - 1);
  const float eps = a.noise[(((size_t)mc * 2 + (MODE == 0 ? 1 : 0)) * a.N + a.slot) * a.nzw + 4 + cc];
  float e_loc = 0.0f, e_sc = 0.0f, tm1 = 0.0f;
  if (MODE == 1) {
    e_loc = a.enc[(size_t)mc * a.enc_ld + cc];
    e_sc = a.enc[(size_t)mc * a.enc_ld + a.nw + cc];
    tm1 = a.rec_prev[((size_t)mc * a.N + a.slot) * rec::W + rec::WHAT + cc];
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4_t acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc1, 0, 0, 0);
  }
  float* r = red + wave * 256;
  r[(4 * kq + 0) * 16 + l15] = acc0.x + acc1.x;
  r[(4 * kq + 1) * 16 + l15] = acc0.y + acc1.y;
  r[(4 * kq + 2) * 16 + l15] = acc0.z + acc1.z;
  r[(4 * kq + 3) * 16 + l15] = acc0.w + acc1.w;
  __syncthreads();
  const float v = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid] + p_bias;
  float* rn = a.rec_new + ((size_t)mc * a.N + a.slot) * rec::W;
  if (MODE == 0) {
    const float t = g5 ? sq_softplus(v) + 1e-2f : v;
    const float sc = sq_dpp<0x101>(t);   // row_shl:1 -- lane i takes lane i + 1 of its row of 16 (an exact copy; no LDS round trip)
    if (g5 == 0 && valid) {
      const float loc = t;
      const float what = loc + sc * eps;
      rn[rec::WHAT + c] = what;
      rn[rec::WHAT_LOC + c] = loc;
      rn[rec::WHAT_SCALE + c] = sc;
    }
  } else {
    // lane g of an element: temporal loc | temporal scale | forget gate | 1 - input gate | 1 - temporal gate
    const float t = g5 == 0 ? v : (g5 == 1 ? sq_softplus(v) + 1e-2f : (g5 == 2 ? sq_gate(v) : sq_gate_compl(v)));
    const float t_scale = sq_dpp<0x101>(t), fg = sq_dpp<0x102>(t), om_ig = sq_dpp<0x103>(t), om_tg = sq_dpp<0x104>(t);   // row_shl:1..4
    if (g5 == 0 && valid) {
      const float t_loc = t;
      const float loc = sq_mix3(fg, tm1, om_ig, e_loc, om_tg, t_loc);
      const float sc = sq_mix2(om_ig, e_sc, om_tg, t_scale);
      const float what = loc + sc * eps;
      rn[rec::WHAT + c] = what;
      rn[rec::WHAT_LOC + c] = loc;
      rn[rec::WHAT_SCALE + c] = sc;
    }
  }
}
#endif
int sq_launch_linear_what(const WhatArgs& a, hipStream_t s) {
#ifdef SQAIR_WIDE
  (void)a; (void)s;
  return -1;
#else
  if ((reinterpret_cast<uintptr_t>(a.x) & 15) != 0 || (a.x_ld & 3) != 0 || a.M < 1 || (a.kc != 16 && a.kc != 8)) return -5;
  const dim3 g(a.n_tiles, (a.M + 15) / 16);
  if (a.kc == 16) {
    if (a.mode == 0) SQ_LAUNCH((k_linear_what<4, 0>), g, dim3(256), 0, s, a);
    else SQ_LAUNCH((k_linear_what<4, 1>), g, dim3(256), 0, s, a);
  } else {
    if (a.mode == 0) SQ_LAUNCH((k_linear_what<2, 0>), g, dim3(256), 0, s, a);
    else SQ_LAUNCH((k_linear_what<2, 1>), g, dim3(256), 0, s, a);
  }
  return 0;
#endif
}

int sq_launch_slot_tail(const TailArgs& a, Dims d, hipStream_t s) {
  if (a.what_done) SQ_LAUNCH(k_slot_tail<true>, dim3((d.R + 15) / 16), dim3(256), 0, s, a, d);
  else SQ_LAUNCH(k_slot_tail<false>, dim3((d.R + 15) / 16), dim3(256), 0, s, a, d);
  return 0;
}
// tail of slot `ta.slot` + the VanillaRNN layer of the next slot: out = tanh([z-record | hid] W + bias + add); wp / bias point at
// the layer's packed weights (K chunks: 4 of the z-record, nh / 16 of the hidden state) and packed bias
int sq_launch_rnn_tail(const TailArgs& ta, Dims d, const float* hid, int hid_ld, const float* wp, const float* bias, const float* add,
                       int add_ld, float* out, int out_ld, int n_out, hipStream_t s, unsigned long long* prof_ts) {
#ifdef SQAIR_WIDE
  (void)ta; (void)d; (void)hid; (void)hid_ld; (void)wp; (void)bias; (void)add; (void)add_ld; (void)out; (void)out_ld; (void)n_out; (void)s; (void)prof_ts;
  return -1;   // (the wide build never fuses the tail: can_fuse_tail, sqair_api.hip)
#else
  const int nt = (n_out + 15) / 16, mt = (d.R + 15) / 16;
  // More 16 x 16 tiles than CUs (from 272 particle rows on at n_hidden 256: cfg-4's 320): the launch would run in two rounds of
  // workgroups, each re-deriving the tail of its rows.  Two column tiles per workgroup then halve both the workgroups and the
  // tail work (same sums per tile: bit-identical).
#ifndef SQAIR_RNN_TAIL_TN2_TILES_DEFAULT
#define SQAIR_RNN_TAIL_TN2_TILES_DEFAULT 257
#endif
  static const int two_from = SQ_KNOB_INT("SQAIR_RNN_TAIL_TN2_TILES", SQAIR_RNN_TAIL_TN2_TILES_DEFAULT);
  if (nt * mt >= two_from && nt >= 2) {
    const dim3 g((nt + 1) / 2, mt);
#define SQ_RT(NH, TN)                                                                                                              \
    do {                                                                                                                             \
      if (ta.what_done) SQ_LAUNCH((k_rnn_tail<NH, TN, true>), g, dim3(256), 0, s, ta, d, hid, hid_ld, wp, bias, add, add_ld, out, out_ld, n_out, prof_ts); \
      else SQ_LAUNCH((k_rnn_tail<NH, TN, false>), g, dim3(256), 0, s, ta, d, hid, hid_ld, wp, bias, add, add_ld, out, out_ld, n_out, prof_ts); \
    } while (0)
    if (d.nh == 256) SQ_RT(4, 2);
    else if (d.nh == 128) SQ_RT(2, 2);
    else return -1;
    return 0;
  }
  const dim3 g(nt, mt);
  if (d.nh == 256) SQ_RT(4, 1);
  else if (d.nh == 128) SQ_RT(2, 1);
  else return -1;
#undef SQ_RT
  return 0;
#endif
}

// ------------------------------------------------------------------------------------------------
// DeepSets summary of the propagated latents (reference: sqair/sqair_modules.py:368-385)
// ------------------------------------------------------------------------------------------------
__global__ void k_latent_sum(const float* __restrict__ f, const float* __restrict__ rec_p, float* __restrict__ c, Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  const int r = blockIdx.x;
  for (int n = threadIdx.x; n < d.nh; n += blockDim.x) {
    float acc = 0.0f;
    for (int k = 0; k < d.N; ++k)
      acc += f[((size_t)r * d.N + k) * d.nh + n] * rec_p[((size_t)r * d.N + k) * rec::W + rec::PRES];
    c[(size_t)r * d.nh + n] = acc;
  }
}
int sq_launch_latent_sum(const float* f, const float* rec_p, float* c, Dims d, hipStream_t s) {
  SQ_LAUNCH(k_latent_sum, dim3(d.R), dim3(256), 0, s, f, rec_p, c, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// All log-probabilities of one frame, presence masks applied in-kernel.
// reference: Propagate._compute_log_probs sqair/sqair_modules.py:281-329 (+ PropagatePrior
// propagate.py:68-158, AffineDiagNormal modules.py:535-545), Discover._compute_log_probs /
// _make_priors sqair_modules.py:149-229, NumStepsDistribution prior.py:61-105 (float64),
// RecurrentNormalImpl modules.py:548-611, SQAIRTimestep sums :483-485, :505-507.
// One wavefront per row b'.
// ------------------------------------------------------------------------------------------------
constexpr int SQ_LOGPROB_WAVES = 2;   // (four waves per workgroup did not all fit the chip at once: two rounds of workgroups)
__global__ __launch_bounds__(64 * SQ_LOGPROB_WAVES) void k_logprob(const LogprobArgs a, const POff po, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  // Everything a row needs (3N slot records, N prior-stat rows, the conditioning state and ~1k small-layer
  // parameters) is pulled into LDS by all 256 threads in one burst of independent loads (a direct global-memory
  // walk was ~20 dependent round trips = 27 us); the wavefronts then share the slots -- wave w evaluates
  // propagation slots w, w + NWV, ... and, once e_sum is known, discovery slots w, w + NWV, ... -- and leave the per-slot
  // terms in LDS, which wave 0 sums in slot order (the order the single wave of rounds 1-2 summed them in, whose
  // ~3500 dependent instructions were 26 of the kernel's 29 us).
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int r = blockIdx.x;
  const int fr = blockIdx.y;               // frame (the launch covers all T frames of the pass)
  const int tid = threadIdx.x;
  constexpr int NWV = SQ_LOGPROB_WAVES, NT = 64 * NWV;
  const int N = d.N, nw = d.nw;
  const int RW = rec::W;
  const size_t fs = (size_t)fr * d.R * N;  // slot-rows per frame
  float* recp_s = smem;                    // N * RW
  float* recd_s = recp_s + N * RW;         // N * RW
  float* recm_s = recd_s + N * RW;         // N * RW
  float* ps_s = recm_s + N * RW;           // N * ps_ld
  float* spre_s = ps_s + N * a.ps_ld;      // 128
  float* ce_s = spre_s + 128;              // 128  rn.cond.w row of e
  float* h2h_s = ce_s + 128;               // 512 + 4
  float* i2h_s = h2h_s + 516;              // 16 + 4
  float* ro_s = i2h_s + 20;                // 32 + 8
  float* isamp_s = ro_s + 40;              // 4
  float* sp0_s = isamp_s + 4;              // 10 + 10
  float* sp1_s = sp0_s + 20;               // 10*(N+1) + (N+1)
  float* spb_s = sp1_s + 11 * (N + 1);     // 2*(N+1)
  float* ch_s = spb_s + 2 * (N + 1);       // 10
  const float* __restrict__ flat = a.flat;
  sq_wave_stage(recp_s, a.rec_p + (fs + (size_t)r * N) * RW, N * RW, tid & 63, tid >> 6, NWV);
  sq_wave_stage(recd_s, a.rec_d + (fs + (size_t)r * N) * RW, N * RW, tid & 63, tid >> 6, NWV);
  sq_wave_stage(recm_s, a.rec_prev + (fs + (size_t)r * N) * RW, N * RW, tid & 63, tid >> 6, NWV);
  sq_wave_stage(ps_s, a.pstats + (fs + (size_t)r * N) * a.ps_ld, N * a.ps_ld, tid & 63, tid >> 6, NWV);
  if (a.cfg.rec_where_prior) {
    if (tid < 128) {
      spre_s[tid] = a.spre[((size_t)fr * d.R + r) * 128 + tid];
      ce_s[tid] = flat[po.rn_cond_w + (4 + d.nh) * 128 + tid];
    }
    for (int i = tid; i < 512; i += NT) h2h_s[i] = flat[po.rn_h2h_w + i];
    if (tid < 4) { h2h_s[512 + tid] = flat[po.rn_h2h_b + tid]; i2h_s[16 + tid] = flat[po.rn_i2h_b + tid]; isamp_s[tid] = flat[po.rn_init_sample + tid]; }
    if (tid < 16) i2h_s[tid] = flat[po.rn_i2h_w + tid];
    if (tid < 32) ro_s[tid] = flat[po.rn_readout_w + tid];
    if (tid < 8) ro_s[32 + tid] = flat[po.rn_readout_b + tid];
  }
  if (tid < 10) { sp0_s[tid] = flat[po.sp_l0_w + tid]; sp0_s[10 + tid] = flat[po.sp_l0_b + tid]; ch_s[tid] = flat[po.cholesky + tid]; }
#ifdef SQAIR_WIDE
  for (int i = tid; i < 10 * (N + 1); i += NT) sp1_s[i] = flat[po.sp_l1_w + i];   // (up to 170 entries: more than the workgroup has threads)
#else
  if (tid < 10 * (N + 1)) sp1_s[tid] = flat[po.sp_l1_w + tid];
#endif
  if (tid <= N) {
    sp1_s[10 * (N + 1) + tid] = flat[po.sp_l1_b + tid];
    spb_s[tid] = flat[po.step_prior_bias + tid];
    spb_s[N + 1 + tid] = flat[po.step_prior_tbias + tid];
  }
  __shared__ float prop_s[SQ_MAXN][6], disc_s[SQ_MAXN][3];
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const size_t tr = (size_t)(a.t + fr) * d.R + r;
  const int t_global = a.t_global + fr;
  const float LOG2PI = 1.83787706640934548356f;

  for (int k = wave; k < N; k += NWV) {
    const float* rp = recp_s + k * RW;
    const float* rm = recm_s + k * RW;
    const float* ps = ps_s + k * a.ps_ld;
    // sample_from_prior (sqair_modules.py:294-318): the posteriors are evaluated at samples of the PRIOR (generation
    // record), the priors at the hidden outputs (replaced in generated frames), every mask and the prior Bernoulli at the
    // posterior path's own presence
    const float* gr = a.gen != nullptr ? a.gen + (fs + (size_t)r * N + k) * gen::W : nullptr;
    const float pres_hidden = rp[rec::PRES];
    const float pres = gr != nullptr ? gr[gen::ORIG_PRES] : pres_hidden, pres_tm1 = rm[rec::PRES];
    const float pres_q = gr != nullptr ? gr[gen::PRES] : pres;
    const float logit = rp[rec::LOGIT], logit_tm1 = rm[rec::LOGIT];
    float pl = ps[0] + a.cfg.prop_prior_step_bias;
    pl = pres_tm1 * pl + (pres_tm1 - 1.0f) * 88.0f;
    if (a.cfg.prop_prior_type != 0) pl = logit_tm1 + 0.1f * pl;
    float qw = 0.0f, pw = 0.0f;
    SQ_WHAT_LANES(wc, lane, nw) {
      const float x = rp[rec::WHAT + wc];
      qw += sq_normal_lp(gr != nullptr ? gr[gen::WHAT + wc] : x, rp[rec::WHAT_LOC + wc], rp[rec::WHAT_SCALE + wc]);
      float ploc = ps[5 + wc];
      if (a.cfg.prop_prior_type == 1) ploc = rm[rec::WHAT + wc];
      else if (a.cfg.prop_prior_type == 2) ploc = rm[rec::WHAT + wc] + 0.1f * ploc;
      pw += sq_normal_lp(x, ploc, sq_softplus(ps[9 + nw + wc]) + 1e-2f);
    }
    const float q_what = sq_wave_sum(qw), p_what = sq_wave_sum(pw);
    float pwh = 0.0f;
    if (lane < 4) {
      float ploc = ps[1 + lane];
      if (a.cfg.prop_prior_type == 1) ploc = rm[rec::WHERE + lane];
      else if (a.cfg.prop_prior_type == 2) ploc = rm[rec::WHERE + lane] + 0.1f * ploc;
      pwh = sq_normal_lp(rp[rec::WHERE + lane], ploc, sq_softplus(ps[5 + nw + lane]) + 1e-2f);
    }
    const float p_where = sq_wave_sum(pwh);
    // MultivariateNormalTriL log-prob: forward substitution with L = T * sc[:,None] + diag(sc)
    float q_where;
    {
      float y[4];
      float sq = 0.0f, logdet = 0.0f;
      for (int i = 0; i < 4; ++i) {
        const float sci = rp[rec::WHERE_SCALE + i];
        float acc = (gr != nullptr ? gr[gen::WHERE + i] : rp[rec::WHERE + i]) - rp[rec::WHERE_LOC + i];
        for (int jj = 0; jj < i; ++jj) acc -= tril4(ch_s, i, jj) * sci * y[jj];
        const float lii = tril4(ch_s, i, i) * sci + sci;
        y[i] = acc / lii;
        sq += y[i] * y[i];
        logdet += logf(fabsf(lii));
      }
      q_where = -0.5f * sq - logdet - 2.0f * LOG2PI;
    }
    const float q_pres = sq_bernoulli_lp(pres_q, logit);
    const float p_pres = sq_bernoulli_lp(pres, pl);
    const float m = pres_tm1 * pres;
    if (lane == 0) {
      const size_t o = tr * N + k;
      if (a.out.prop_what_log_prob) a.out.prop_what_log_prob[o] = q_what * m;
      if (a.out.prop_where_log_prob) a.out.prop_where_log_prob[o] = q_where * m;
      if (a.out.prop_what_prior_log_prob) a.out.prop_what_prior_log_prob[o] = p_what * m;
      if (a.out.prop_where_prior_log_prob) a.out.prop_where_prior_log_prob[o] = p_where * m;
      if (a.out.prop_prob) a.out.prop_prob[o] = expf(q_pres) * pres_tm1;
      if (a.out.prop_pres) a.out.prop_pres[o] = pres_hidden;
      prop_s[k][0] = (q_what + q_where) * m;
      prop_s[k][1] = (p_what + p_where) * m;
      prop_s[k][2] = q_pres * pres_tm1;
      prop_s[k][3] = p_pres * pres_tm1;
      prop_s[k][4] = (sq_sigmoid(pl) - 0.5f) / (float)N;
      prop_s[k][5] = pres;
    }
  }
  __syncthreads();
  float e_sum = 0.0f;
  for (int k = 0; k < N; ++k) e_sum += prop_s[k][4];

  // ---- discovery
  float hs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (a.cfg.rec_where_prior) {
    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = lane; i < 128; i += 64) {
      const float sv = sq_elu(spre_s[i] + e_sum * ce_s[i]);
      for (int jj = 0; jj < 4; ++jj) part[jj] += sv * h2h_s[i * 4 + jj];
    }
    for (int jj = 0; jj < 4; ++jj) hs[jj] = sq_wave_sum(part[jj]) + h2h_s[512 + jj] + i2h_s[16 + jj];
  }
  for (int j = wave; j < N; j += NWV) {
    const float* rd = recd_s + j * RW;
    const float pres = rd[rec::PRES];
    float qw = 0.0f, pw = 0.0f;
    SQ_WHAT_LANES(wc, lane, nw) {
      const float x = rd[rec::WHAT + wc];
      qw += sq_normal_lp(x, rd[rec::WHAT_LOC + wc], rd[rec::WHAT_SCALE + wc]);
      pw += sq_normal_lp(x, 0.0f, 1.0f);
    }
    const float q_what = sq_wave_sum(qw), p_what = sq_wave_sum(pw);
    float qwh = 0.0f, pwh = 0.0f;
    if (lane < 4) {
      const float x = rd[rec::WHERE + lane];
      qwh = sq_normal_lp(x, rd[rec::WHERE_LOC + lane], rd[rec::WHERE_SCALE + lane]);
      if (a.cfg.rec_where_prior) {
        const float* xp = j == 0 ? isamp_s : recd_s + (j - 1) * RW + rec::WHERE;
        float o[4];
        for (int mm = 0; mm < 4; ++mm) {
          float acc = hs[mm];
          for (int i = 0; i < 4; ++i) acc += xp[i] * i2h_s[i * 4 + mm];
          o[mm] = sq_tanh(acc);
        }
        float loc = ro_s[32 + lane], raw = ro_s[32 + 4 + lane];
        for (int mm = 0; mm < 4; ++mm) {
          loc += o[mm] * ro_s[mm * 8 + lane];
          raw += o[mm] * ro_s[mm * 8 + 4 + lane];
        }
        pwh = sq_normal_lp(x, loc, sq_softplus(raw) + 1e-2f);
      } else {
        pwh = sq_normal_lp(x, a.cfg.where_prior_mean[lane], 1.0f);
      }
    }
    const float q_where = sq_wave_sum(qwh), p_where = sq_wave_sum(pwh);
    if (lane == 0) {
      const size_t o = tr * N + j;
      if (a.out.disc_what_log_prob) a.out.disc_what_log_prob[o] = q_what * pres;
      if (a.out.disc_where_log_prob) a.out.disc_where_log_prob[o] = q_where * pres;
      if (a.out.disc_what_prior_log_prob) a.out.disc_what_prior_log_prob[o] = p_what * pres;
      if (a.out.disc_where_prior_log_prob) a.out.disc_where_prior_log_prob[o] = p_where * pres;
      if (a.out.disc_pres) a.out.disc_pres[o] = pres;
      disc_s[j][0] = (q_what + q_where) * pres;
      disc_s[j][1] = (p_what + p_where) * pres;
      // num_steps is what discovery itself inferred (sqair_modules.py:146), also when the frame is generated and the
      // hidden presence has been zeroed
      const bool generated = a.gen != nullptr && a.cfg.generate_after > 0 && t_global > a.cfg.generate_after;
      disc_s[j][2] = generated ? a.gen[(fs + (size_t)r * N + j) * gen::W + gen::ORIG_DPRES] : pres;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  // wave 0: the sums in slot order (every lane the same), then class c of the number-of-steps distributions on lane c <= N
  float q_prop = 0.0f, p_prop = 0.0f, q_pres_sum = 0.0f, p_pres_sum = 0.0f, n_prop = 0.0f;
  float q_disc = 0.0f, p_disc = 0.0f, n_disc = 0.0f;
  for (int k = 0; k < N; ++k) {
    q_prop += prop_s[k][0];
    p_prop += prop_s[k][1];
    q_pres_sum += prop_s[k][2];
    p_pres_sum += prop_s[k][3];
    n_prop += prop_s[k][5];
    q_disc += disc_s[k][0];
    p_disc += disc_s[k][1];
    n_disc += disc_s[k][2];
  }
  // NumStepsDistribution (float64 inside, prior.py:61-67): joint_c = (1 - p_c) prod_{i < c} p_i, joint_N = prod p_i, normalised
  const int n = (int)(n_disc + 0.5f);
  double mine = 0.0, at_n = 0.0, tot = 0.0;
  {
    double cum = 1.0;
    for (int j = 0; j <= N; ++j) {
      const double pj = j < N ? (double)recd_s[j * RW + rec::PROB] : 0.0;
      const double jc = (1.0 - pj) * cum;
      cum *= pj;
      tot += jc;
      mine = j == lane ? jc : mine;
      at_n = j == n ? jc : at_n;
    }
  }
  const float jn = (float)(at_n / tot);
  const float q_num = logf(fminf(fmaxf(jn, 1e-16f), 1.0f));
  float p_num;
  if (a.cfg.disc_prior_type == 1) {
    const float pr = 1.0f - a.cfg.step_success_prob;
    p_num = (float)n * log1pf(-pr) + logf(pr);
  } else {
    const int c = min(lane, N);
    float v = spb_s[c] + (t_global > 0 ? spb_s[N + 1 + c] : 0.0f) + sp1_s[10 * (N + 1) + c];
    for (int i = 0; i < 10; ++i) v += sq_elu(e_sum * sp0_s[i] + sp0_s[10 + i]) * sp1_s[i * (N + 1) + c];
    const float lg = lane <= N ? sq_elu(v) : -1e30f;
    float mx = lg;
    constexpr int CLS = SQ_MAXN + 1 <= 16 ? 16 : 32;   // N + 1 classes on lanes 0 .. CLS - 1
#pragma unroll
    for (int o = 1; o < CLS; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = lane <= N ? expf(lg - mx) : 0.0f;
#pragma unroll
    for (int o = 1; o < CLS; o <<= 1) se += __shfl_xor(se, o, 64);
    p_num = __shfl(lg, n, 64) - (mx + logf(se));
  }
  if (a.out.disc_prob && lane <= N) a.out.disc_prob[tr * (N + 1) + lane] = (float)(mine / tot);
  if (lane != 0) return;
  const float q_prop_tot = q_prop + q_pres_sum, p_prop_tot = p_prop + p_pres_sum;
  const float q_disc_tot = q_disc + q_num, p_disc_tot = p_disc + p_num;
  a.qz[(size_t)fr * d.R + r] = q_disc_tot + q_prop_tot;
  a.pz[(size_t)fr * d.R + r] = p_disc_tot + p_prop_tot;
  a.disc_lp[(size_t)fr * d.R + r] = q_pres_sum + q_num;
  if (a.out.prop_log_prob) a.out.prop_log_prob[tr] = q_pres_sum;
  if (a.out.prop_prior_log_prob) a.out.prop_prior_log_prob[tr] = p_pres_sum;
  if (a.out.disc_log_prob) a.out.disc_log_prob[tr] = q_num;
  if (a.out.disc_prior_log_prob) a.out.disc_prior_log_prob[tr] = p_num;
  if (a.out.step_log_prob) a.out.step_log_prob[tr] = q_pres_sum + q_num;
  if (a.out.discrete_log_prob) a.out.discrete_log_prob[tr] = q_pres_sum + q_num;
  if (a.out.num_prop_steps_per_sample) a.out.num_prop_steps_per_sample[tr] = n_prop;
  if (a.out.num_disc_steps_per_sample) a.out.num_disc_steps_per_sample[tr] = n_disc;
}
int sq_launch_logprob(const LogprobArgs& a, POff po, Dims d, hipStream_t s) {
  const size_t shm = ((size_t)3 * d.N * rec::W + (size_t)d.N * a.ps_ld + 128 + 128 + 516 + 20 + 40 + 4 + 20 +
                      11 * (d.N + 1) + 2 * (d.N + 1) + 16) * sizeof(float);
  if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_logprob, 150 * 1024) != 0) return -2;
  SQ_LAUNCH(k_logprob, dim3(d.R, a.n_frames), dim3(64 * SQ_LOGPROB_WAVES), shm, s, a, po, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Generation modes (reference: Propagate._compute_log_probs sqair/sqair_modules.py:294-302, Discover._compute_log_probs
// :157-170, do_generate sqair/seq.py:198-200, RecurrentNormal.sample sqair/modules.py:619-629).  One wavefront per row.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gen_prior_logit(const GenArgs& a, const float* rm, const float* ps) {
  const float pres_tm1 = rm[rec::PRES];
  float pl = ps[0] + a.cfg.prop_prior_step_bias;
  pl = pres_tm1 * pl + (pres_tm1 - 1.0f) * 88.0f;
  if (a.cfg.prop_prior_type != 0) pl = rm[rec::LOGIT] + 0.1f * pl;
  return pl;
}
// propagation: samples of the prior for every slot -> generation record; replaces the hidden outputs when generating
__global__ __launch_bounds__(64) void k_generate_prop(const GenArgs a, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  const int r = blockIdx.x, lane = threadIdx.x, N = d.N, nw = d.nw;
  for (int k = 0; k < N; ++k) {
    const size_t rk = (size_t)r * N + k;
    float* rp = a.rec_p + rk * rec::W;
    const float* rm = a.rec_prev + rk * rec::W;
    const float* ps = a.pstats + rk * a.ps_ld;
    const float* gn = a.gen_noise + (((size_t)r * 2 + 0) * N + k) * d.nzw;
    float* g = a.gen + rk * gen::W;
    SQ_WHAT_LANES(wc, lane, nw) {
      float ploc = ps[5 + wc];
      if (a.cfg.prop_prior_type == 1) ploc = rm[rec::WHAT + wc];
      else if (a.cfg.prop_prior_type == 2) ploc = rm[rec::WHAT + wc] + 0.1f * ploc;
      const float v = ploc + (sq_softplus(ps[9 + nw + wc]) + 1e-2f) * gn[4 + wc];
      g[gen::WHAT + wc] = v;
      if (a.do_generate) rp[rec::WHAT + wc] = v;
    }
    if (lane < 4) {
      float ploc = ps[1 + lane];
      if (a.cfg.prop_prior_type == 1) ploc = rm[rec::WHERE + lane];
      else if (a.cfg.prop_prior_type == 2) ploc = rm[rec::WHERE + lane] + 0.1f * ploc;
      const float v = ploc + (sq_softplus(ps[5 + nw + lane]) + 1e-2f) * gn[lane];
      g[gen::WHERE + lane] = v;
      if (a.do_generate) rp[rec::WHERE + lane] = v;
    }
    if (lane == 0) {
      const float sp = gn[4 + nw] < sq_sigmoid(gen_prior_logit(a, rm, ps)) ? 1.0f : 0.0f;
      g[gen::PRES] = sp;
      g[gen::ORIG_PRES] = rp[rec::PRES];
      if (a.do_generate) rp[rec::PRES] = sp;
    }
  }
}
// discovery (generated frames only): what ~ N(0, I), where ~ the where prior, presence = 0; keeps the original presence
__global__ __launch_bounds__(64) void k_generate_disc(const GenArgs a, const POff po, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  const int r = blockIdx.x, lane = threadIdx.x, N = d.N, nw = d.nw;
  const float* flat = a.flat;
  float hs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (a.cfg.rec_where_prior) {
    float e_sum = 0.0f;
    for (int k = 0; k < N; ++k)
      e_sum += (sq_sigmoid(gen_prior_logit(a, a.rec_prev + ((size_t)r * N + k) * rec::W, a.pstats + ((size_t)r * N + k) * a.ps_ld)) - 0.5f) / (float)N;
    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = lane; i < 128; i += 64) {
      const float sv = sq_elu(a.spre[(size_t)r * 128 + i] + e_sum * flat[po.rn_cond_w + (4 + d.nh) * 128 + i]);
      for (int jj = 0; jj < 4; ++jj) part[jj] += sv * flat[po.rn_h2h_w + i * 4 + jj];
    }
    for (int jj = 0; jj < 4; ++jj) hs[jj] = sq_wave_sum(part[jj]) + flat[po.rn_h2h_b + jj] + flat[po.rn_i2h_b + jj];
  }
  float xprev[4];
  for (int i = 0; i < 4; ++i) xprev[i] = flat[po.rn_init_sample + i];
  for (int j = 0; j < N; ++j) {
    const size_t rj = (size_t)r * N + j;
    float* rd = a.rec_d + rj * rec::W;
    const float* gn = a.gen_noise + (((size_t)r * 2 + 1) * N + j) * d.nzw;
    if (lane == 0) a.gen[rj * gen::W + gen::ORIG_DPRES] = rd[rec::PRES];
    SQ_WHAT_LANES(wc, lane, nw) rd[rec::WHAT + wc] = gn[4 + wc];
    float x[4];
    for (int c = 0; c < 4; ++c) {  // every lane computes the 4 components (tiny), lane 0 stores
      float loc, sc;
      if (a.cfg.rec_where_prior) {
        float o[4];
        for (int mm = 0; mm < 4; ++mm) {
          float acc = hs[mm];
          for (int i = 0; i < 4; ++i) acc += xprev[i] * flat[po.rn_i2h_w + i * 4 + mm];
          o[mm] = sq_tanh(acc);
        }
        loc = flat[po.rn_readout_b + c];
        float raw = flat[po.rn_readout_b + 4 + c];
        for (int mm = 0; mm < 4; ++mm) {
          loc += o[mm] * flat[po.rn_readout_w + mm * 8 + c];
          raw += o[mm] * flat[po.rn_readout_w + mm * 8 + 4 + c];
        }
        sc = sq_softplus(raw) + 1e-2f;
      } else {
        loc = a.cfg.where_prior_mean[c];
        sc = 1.0f;
      }
      x[c] = loc + sc * gn[c];
    }
    __syncthreads();
    if (lane < 4) rd[rec::WHERE + lane] = x[lane];
    if (lane == 0) rd[rec::PRES] = 0.0f;
    for (int c = 0; c < 4; ++c) xprev[c] = x[c];
  }
}
int sq_launch_generate_prop(const GenArgs& a, POff po, Dims d, hipStream_t s) {
  (void)po;
  SQ_LAUNCH(k_generate_prop, dim3(d.R), dim3(64), 0, s, a, d);
  return 0;
}
int sq_launch_generate_disc(const GenArgs& a, POff po, Dims d, hipStream_t s) {
  SQ_LAUNCH(k_generate_disc, dim3(d.R), dim3(64), 0, s, a, po, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Slot compaction (reference: SQAIRTimestep._choose_latents sqair/sqair_modules.py:514-582,
// index.compute_object_ids index.py:198-221, index.select_present index.py:132-165).
// One workgroup per row: 2N presence bits -> stable present-first permutation -> copy the N
// survivors (record + prior state + temporal state) into the next frame's state.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(const CompactArgs a, const POff po, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ int src_s[SQ_MAXN];
  __shared__ float id_s[SQ_MAXN];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int N = d.N, nh = d.nh;
  if (tid < 64) {
    // one lane per candidate slot (N propagated ++ N discovered): ballot + popcount give the stable
    // present-first permutation and the running discovery count that numbers new objects
    const int sl = tid;
    const bool in = sl < 2 * N;
    const float last = a.last_id_prev[r];
    float p = 0.0f, pid = -1.0f;
    if (in) {
      p = sl < N ? a.rec_p[((size_t)r * N + sl) * rec::W + rec::PRES] : a.rec_d[((size_t)r * N + (sl - N)) * rec::W + rec::PRES];
      if (sl < N) pid = a.rec_prev[((size_t)r * N + sl) * rec::W + rec::ID];
    }
    const unsigned long long all = (2 * N >= 64) ? ~0ull : ((1ull << (2 * N)) - 1ull);
    const unsigned long long present = __ballot(in && p != 0.0f) & all;
    const unsigned long long below = (1ull << sl) - 1ull;
    const int n_present = __popcll(present);
    const unsigned long long disc_bits = present >> N;  // bit j = discovery step j present
    float id;
    if (sl < N) id = pid * p - (1.0f - p);
    else {
      const int cum = __popcll(disc_bits & ((2ull << (sl - N)) - 1ull));  // inclusive cumsum of disc presence
      id = ((float)cum + last) * p - (1.0f - p);
    }
    if (in) {
      const int dst = (p != 0.0f) ? __popcll(present & below) : n_present + __popcll(~present & all & below);
      if (dst < N) {
        src_s[dst] = sl;
        id_s[dst] = id;
        if (a.src_out != nullptr) a.src_out[(size_t)r * N + dst] = sl;
      }
    }
    if (sl == 0) a.last_id_next[r] = last + (float)__popcll(disc_bits);
  }
  __syncthreads();
  const size_t tr = (size_t)a.t * d.R + r;
  // copy the N survivors: record (168) + temporal state + prior state, as 16-byte units (all rows are 16-byte aligned;
  // only the trainable initial states of newly discovered objects come from the unaligned flat parameter buffer)
  typedef float cf4 __attribute__((ext_vector_type(4)));
  typedef float cf4u __attribute__((ext_vector_type(4), aligned(4)));   // (flat parameter offsets are only 4-byte aligned)
  const int snh = d.snh, psnh = d.psnh;
  const int r4 = rec::W / 4, t4 = snh / 4, per4 = r4 + t4 + psnh / 4;
  // four units per thread and trip, every load of a trip ahead of its stores.  The kernel stays at two cold memory round trips
  // (presences -> permutation, then the survivors' rows) plus the write, ~4.8 us
  constexpr int CU = 4;
  for (int e0 = tid; e0 < N * per4; e0 += 256 * CU) {
    cf4 v[CU];
    cf4* dp[CU];
#pragma unroll
    for (int u = 0; u < CU; ++u) {
      const int e = e0 + 256 * u;
      dp[u] = nullptr;
      if (e < N * per4) {
        const int dst = e / per4, i = e - dst * per4;
        const int sidx = src_s[dst];
        const bool prop = sidx < N;
        const int ss = prop ? sidx : sidx - N;
        if (i < r4) {
          const float* rs = (prop ? a.rec_p : a.rec_d) + ((size_t)r * N + ss) * rec::W;
          v[u] = reinterpret_cast<const cf4*>(rs)[i];
          if (i == rec::ID / 4) v[u][rec::ID % 4] = id_s[dst];
          dp[u] = reinterpret_cast<cf4*>(a.rec_next + ((size_t)r * N + dst) * rec::W) + i;
        } else {
          const bool tmp = i < r4 + t4;
          const int q = tmp ? i - r4 : i - r4 - t4;
          const int w = tmp ? snh : psnh;
          const float* sp = prop ? (tmp ? a.temporal_p : a.prior_p) + ((size_t)r * N + ss) * w + 4 * q
                                 : a.flat + (tmp ? po.temporal_init : po.prior_init) + 4 * q;
          v[u] = *reinterpret_cast<const cf4u*>(sp);
          dp[u] = reinterpret_cast<cf4*>((tmp ? a.temporal_next : a.prior_next) + ((size_t)r * N + dst) * w) + q;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < CU; ++u)
      if (dp[u] != nullptr) *dp[u] = v[u];
  }
  // the 9 hidden outputs + object id (seq.py:121-134)
  constexpr int CW = rec::NWMAX <= 64 ? 64 : 128;   // lanes per slot: enough for the `what` entries
  for (int e = tid; e < N * CW; e += 256) {
    const int dst = e / CW, c = e & (CW - 1);
    const int sidx = src_s[dst];
    const float* rs = sidx < N ? a.rec_p + ((size_t)r * N + sidx) * rec::W : a.rec_d + ((size_t)r * N + (sidx - N)) * rec::W;
    const size_t o = tr * N + dst;
    if (c < d.nw) {
      if (a.out.what) a.out.what[o * d.nw + c] = rs[rec::WHAT + c];
      if (a.out.what_loc) a.out.what_loc[o * d.nw + c] = rs[rec::WHAT_LOC + c];
      if (a.out.what_scale) a.out.what_scale[o * d.nw + c] = rs[rec::WHAT_SCALE + c];
    }
    if (c < 4) {
      if (a.out.where) a.out.where[o * 4 + c] = rs[rec::WHERE + c];
      if (a.out.where_loc) a.out.where_loc[o * 4 + c] = rs[rec::WHERE_LOC + c];
      if (a.out.where_scale) a.out.where_scale[o * 4 + c] = rs[rec::WHERE_SCALE + c];
    }
    if (c == 0) {
      if (a.out.presence_prob) a.out.presence_prob[o] = rs[rec::PROB];
      if (a.out.presence) a.out.presence[o] = rs[rec::PRES];
      if (a.out.presence_logit) a.out.presence_logit[o] = rs[rec::LOGIT];
      if (a.out.obj_id) a.out.obj_id[o] = id_s[dst];
    }
  }
  if (tid == 0 && a.out.num_steps_per_sample) {
    float ns = 0.0f;
    for (int dst = 0; dst < N; ++dst) {
      const int sidx = src_s[dst];
      ns += sidx < N ? a.rec_p[((size_t)r * N + sidx) * rec::W + rec::PRES] : a.rec_d[((size_t)r * N + (sidx - N)) * rec::W + rec::PRES];
    }
    a.out.num_steps_per_sample[tr] = ns;
  }
}
int sq_launch_compact(const CompactArgs& a, POff po, Dims d, hipStream_t s) {
  SQ_LAUNCH(k_compact, dim3(d.R), dim3(256), 0, s, a, po, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Decoder back end: inverse spatial transformer of the N decoded glimpses onto the canvas, the
// written-to mask, mean image, Gaussian log-likelihood and the frame log-weight, fused.
// reference: AIRDecoder._decode/_add_mean_image/_build sqair/modules.py:435-467,
// SequentialAIR._compute_log_weights sqair/seq.py:271-276.  One workgroup per (row b', frame); the N
// glimpses sit in LDS, the canvas is built band by band over the slots' boxes (sqair_canvas.h), then
// every thread finishes its pixels of the band: the canvas is written at most once, the frame read once.
// ------------------------------------------------------------------------------------------------
template <int PF, int ROWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_insert_loglik(const InsertArgs a, const Dims d, const int band_rows SQ_TLP) {
  SQ_TL_SCOPE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.N, G = d.G, G2 = d.G * d.G, H = d.H, W = d.W, P = d.H * d.W;
  const CanvasLds c = sq_canvas_carve(smem, N, G, H, W, band_rows);
  __shared__ float red_s[4];
  const int r = sq_row_of_wg(blockIdx.x, d), tid = threadIdx.x;   // (the K particles of a sequence on one XCD: one L2 holds the frame)
  const int fr = blockIdx.y;  // frame
  const int b = sq_div(r, d.k_mul);
  const size_t fs = (size_t)fr * d.R * N + (size_t)r * N;  // first slot of this (frame, row)
  const float* __restrict__ img = a.img + ((size_t)fr * d.B + b) * d.P4;
  const size_t frr = (size_t)fr * d.R + r;
  const float qv = a.qz != nullptr ? a.qz[frr] : 0.0f, pv = a.qz != nullptr ? a.pz[frr] : 0.0f;  // requested early
  if (a.rec) sq_canvas_prologue(c, a.glimpse + fs * G2, a.rec + fs * a.rec_ld + rec::WHERE, a.rec_ld, a.rec + fs * a.rec_ld + rec::PRES, a.rec_ld, N, G, H, W);
  else sq_canvas_prologue(c, a.glimpse + fs * G2, a.where_plain + (size_t)r * N * 4, 4, a.pres_plain + (size_t)r * N, 1, N, G, H, W);
  float ll = 0.0f;
  // Two things every pixel paid for and few need.  (1) The likelihood's scale m std_fg + (1 - m) std_bg is ONE number when the
  // two flags agree (bg_std=None -> output_std, the shipped configuration: modules.py:419-422): its logarithm and reciprocal are
  // taken once, not per pixel.  (2) A pixel no glimpse's box touches has a written-to mask sum of exactly 0, i.e. m = sigmoid(-10):
  // a wavefront (64 consecutive pixels of a row) whose pixels are all outside every box skips the exponential -- most of a
  // 128 x 128 frame.  The value is the same sq_sigmoid(-10) either way.  (cfg-5: 111 -> 98 us.)
  const bool one_sd = a.std_fg == a.std_bg;
  const float inv_sd = 1.0f / a.std_fg, lp0 = -logf(a.std_fg) - 0.91893853320467274178f;
  const float m_bg = sq_sigmoid(-10.0f);
  for (int yb0 = 0; yb0 < H; yb0 += band_rows) {
    const int yb1 = min(H, yb0 + band_rows) - 1, n = (yb1 - yb0 + 1) * W, pix0 = yb0 * W;
    float xv[PF], mv[PF];  // the band's frame / mean-image values: in flight while the canvas is built
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int p = tid + q * 256;
      xv[q] = p < n ? img[pix0 + p] : 0.0f;
      mv[q] = p < n ? a.mean_img[pix0 + p] : 0.0f;
    }
    sq_canvas_band<ROWS>(c, yb0, yb1, N, G, H, W);
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int p = tid + q * 256;
      const float msv = p < n ? c.ms[p] : 0.0f;
      const bool any_on = __builtin_amdgcn_ballot_w64(msv != 0.0f) != 0ull;   // (wave-uniform)
      if (p < n) {
        const float m = any_on ? sq_sigmoid(-10.0f + msv * 20.0f) : m_bg;
        const float cv = c.cv[p] + mv[q] * m;
        if (one_sd) {
          const float dd = (xv[q] - cv) * inv_sd;
          ll += fmaf(-0.5f * dd, dd, lp0);
        } else {
          const float sd = m * a.std_fg + (1.0f - m) * a.std_bg;
          ll += sq_normal_lp(xv[q], cv, sd);
        }
        if (a.canvas) a.canvas[frr * P + pix0 + p] = cv;
      }
    }
    if (yb1 + 1 < H) __syncthreads();  // the next band clears c.cv / c.ms
  }
  ll = sq_wave_sum(ll);
  if ((tid & 63) == 0) red_s[tid >> 6] = ll;
  __syncthreads();
  if (tid == 0) {
    const float dll = red_s[0] + red_s[1] + red_s[2] + red_s[3];
    a.data_ll[frr] = dll;
    if (a.qz != nullptr) {
      const size_t tr = (size_t)a.t * d.R + frr;
      const float q = qv, p = pv;
      const float kl = q - p;
      if (a.out.data_ll_per_sample) a.out.data_ll_per_sample[tr] = dll;
      if (a.out.kl_per_sample) a.out.kl_per_sample[tr] = kl;
      if (a.out.log_q_z_given_x_per_sample) a.out.log_q_z_given_x_per_sample[tr] = q;
      if (a.out.log_p_z_per_sample) a.out.log_p_z_per_sample[tr] = p;
      if (a.out.log_weights_per_timestep) a.out.log_weights_per_timestep[tr] = dll - kl;
    }
  }
}
// The same for frames wider than a wavefront, ROW-WAVE formulation (sqair_canvas.h): wave w owns rows w, w + 4, ...; lane l the
// CPL adjacent columns from CPL l.  NMAX bounds the slots whose column taps a thread keeps in registers (3 NMAX CPL VGPRs); FULLW:
// W == 64 CPL (vector loads of the frame, no column guards).
template <int NMAX, int CPL, bool FULLW, bool ONE_SD, int WAVES, int PD>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(SQ_ROWS_WPE(NMAX, CPL), 8))) void k_insert_loglik_rows(const InsertArgs a, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  static_assert(CPL == 2 || CPL == 4, "column pairs");
  constexpr int NT = 64 * WAVES;
  constexpr int CP = CPL / 2;   // column pairs per lane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.N, G = d.G, G2 = d.G * d.G, H = d.H, W = d.W, P = d.H * d.W;
  const CanvasRowsLds c = sq_canvas_rows_carve(smem, N, G, H);
  __shared__ float red_s[WAVES];
  const int r = sq_row_of_wg(blockIdx.x, d), tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = blockIdx.y;  // frame
  const int b = sq_div(r, d.k_mul);
  const size_t fs = (size_t)fr * d.R * N + (size_t)r * N;  // first slot of this (frame, row)
  const char* __restrict__ img = reinterpret_cast<const char*>(a.img + ((size_t)fr * d.B + b) * d.P4);
  const char* __restrict__ mean = reinterpret_cast<const char*>(a.mean_img);
  const size_t frr = (size_t)fr * d.R + r;
  const float qv = a.qz != nullptr ? a.qz[frr] : 0.0f, pv = a.qz != nullptr ? a.pz[frr] : 0.0f;  // requested early
  // frame / mean-image values of a row: one 32-bit byte offset per lane on two scalar bases; PD rows in flight -- the ring slot of a
  // row is re-requested for the row PD trips ahead as soon as it has been read (a trip over a row no box meets is ~25 instructions:
  // one row ahead does not cover a memory round trip)
  sq_f2 xn[PD][CP], mn[PD][CP];
  auto request = [&](int u, int yy) {
    const int y = min(yy, H - 1);   // (scalar)
    if (FULLW) {
      const unsigned off = (unsigned)(y * W * 4) + (unsigned)lane * (CPL * 4);
      if (CPL == 2) {
        xn[u][0] = *reinterpret_cast<const sq_f2*>(img + off);
        mn[u][0] = *reinterpret_cast<const sq_f2*>(mean + off);
      } else {
        const sq_f4 xx = *reinterpret_cast<const sq_f4*>(img + off), mm = *reinterpret_cast<const sq_f4*>(mean + off);
        xn[u][0] = xx.xy; xn[u][CP - 1] = xx.zw; mn[u][0] = mm.xy; mn[u][CP - 1] = mm.zw;
      }
    } else {
#pragma unroll
      for (int q = 0; q < CP; ++q) {   // (clamped addresses, unconditional loads)
        const unsigned o0 = (unsigned)(y * W + min(lane * CPL + 2 * q, W - 1)) * 4, o1 = (unsigned)(y * W + min(lane * CPL + 2 * q + 1, W - 1)) * 4;
        xn[u][q] = sq_f2{*reinterpret_cast<const float*>(img + o0), *reinterpret_cast<const float*>(img + o1)};
        mn[u][q] = sq_f2{*reinterpret_cast<const float*>(mean + o0), *reinterpret_cast<const float*>(mean + o1)};
      }
    }
  };
#pragma unroll
  for (int u = 0; u < PD; ++u) request(u, wave + WAVES * u);
  if (a.rec) sq_canvas_rows_prologue<NT, (NMAX * 20 * 20 + NT - 1) / NT>(c, a.glimpse + fs * G2, a.rec + fs * a.rec_ld + rec::WHERE, a.rec_ld, a.rec + fs * a.rec_ld + rec::PRES, a.rec_ld, N, G, H, d.g_mul);
  else sq_canvas_rows_prologue<NT, (NMAX * 20 * 20 + NT - 1) / NT>(c, a.glimpse + fs * G2, a.where_plain + (size_t)r * N * 4, 4, a.pres_plain + (size_t)r * N, 1, N, G, H, d.g_mul);
  // column taps of this thread's CPL columns for every slot (registers)
  int xo[NMAX][CPL];
  sq_f2 wab[NMAX][CPL];        // {wa, wb}
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    const int kk = min(k, N - 1);
    const float sx = c.co[kk * 4 + 0], tx = c.co[kk * 4 + 2];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      CanvasAxisTap t{0, 0.0f, 0.0f};
      const int x = lane * CPL + q;
      if (k < N && (FULLW || x < W)) t = sq_canvas_axis_tap(sq_canvas_coord(x, W, sx, tx, G), G);
      xo[k][q] = t.i * 8;
      wab[k][q] = sq_f2{t.wa, t.wb};
    }
  }
  sq_f2 vw[CP];   // 1 for a column inside the frame
#pragma unroll
  for (int q = 0; q < CP; ++q) vw[q] = sq_f2{(FULLW || lane * CPL + 2 * q < W) ? 1.0f : 0.0f, (FULLW || lane * CPL + 2 * q + 1 < W) ? 1.0f : 0.0f};
  const float m_bg = sq_sigmoid(-10.0f);
  const char* __restrict__ prb = reinterpret_cast<const char*>(c.pr);
  float ll = 0.0f;          // generic scales: the sum of the pixels' log-densities
  sq_f2 ss = {0.0f, 0.0f};  // one scale: the sum of squared residuals
  unsigned rm_n = c.rmask[wave];
  for (int y0 = wave; y0 < H; y0 += WAVES * PD)
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    const int y = y0 + WAVES * u;
    if (y >= H) break;
    sq_f2 xv[CP], mv[CP], cvv[CP], mk[CP];
#pragma unroll
    for (int q = 0; q < CP; ++q) {
      xv[q] = xn[u][q];
      mv[q] = mn[u][q];
    }
    const unsigned rm = __builtin_amdgcn_readfirstlane(rm_n);   // slots whose box meets this row (one LDS address for the wave)
    request(u, y + WAVES * PD);
    rm_n = c.rmask[min(y + WAVES, H - 1)];
    if (rm == 0u) {   // (scalar branch) a row no box meets: canvas = mean image x sigmoid(-10)
#pragma unroll
      for (int q = 0; q < CP; ++q) {
        mk[q] = sq_f2{m_bg, m_bg};
        cvv[q] = mv[q] * mk[q];
      }
    } else {
      sq_f2 ms[CP];
      float cvs[CPL];
      sq_f2 msw[CPL];
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        cvs[q] = 0.0f;
        msw[q] = sq_f2{0.0f, 0.0f};
      }
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if (!(rm & (1u << k))) continue;   // (scalar bit test)
        const float4 yr = c.yrec[k * H + y];   // (one address for the wave: a broadcast)
        const int ro = __builtin_bit_cast(int, yr.x);
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const sq_f4 tp = *reinterpret_cast<const sq_f4a8*>(prb + (xo[k][q] + ro));   // {a0, b0, a1, b1}: upper / lower texel at i, i + 1
          const sq_f2 t = sq_tap_rows(wab[k][q], tp.xy, tp.zw);   // {row a, row b} x-interpolated
          cvs[q] = fmaf(yr.y, t.x, cvs[q]);
          cvs[q] = fmaf(yr.z, t.y, cvs[q]);
        }
        // mask sum: pk (wa_y + wb_y) (wa_x + wb_x), the x factor kept as its two terms {wa, wb} until the slots are through (their sum
        // per slot is loop-invariant: the compiler would hoist -- and spill -- it)
        const sq_f2 yzw = {yr.z, yr.w};
#pragma unroll
        for (int q = 0; q < CPL; ++q)
          asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\ts_nop 0" : "+v"(msw[q]) : "v"(yzw), "v"(wab[k][q]));
      }
#pragma unroll
      for (int q = 0; q < CP; ++q) ms[q] = sq_f2{msw[2 * q].x + msw[2 * q].y, msw[2 * q + 1].x + msw[2 * q + 1].y};
#pragma unroll
      for (int q = 0; q < CP; ++q) {
        mk[q] = sq_mask_sigmoid2(ms[q]);
        cvv[q] = sq_fma2(mv[q], mk[q], sq_f2{cvs[2 * q], cvs[2 * q + 1]});
      }
    }
#pragma unroll
    for (int q = 0; q < CP; ++q) {
      if (ONE_SD) {
        sq_f2 df = xv[q] - cvv[q];
        if (!FULLW) df *= vw[q];
        ss = sq_fma2(df, df, ss);
      } else {
        const sq_f2 sd = mk[q] * a.std_fg + (sq_f2{1.0f, 1.0f} - mk[q]) * a.std_bg;
        ll += vw[q].x * sq_normal_lp(xv[q].x, cvv[q].x, sd.x) + vw[q].y * sq_normal_lp(xv[q].y, cvv[q].y, sd.y);
      }
      if (a.canvas) {
        float* o = a.canvas + frr * P + y * W + lane * CPL + 2 * q;
        if (FULLW) *reinterpret_cast<sq_f2*>(o) = cvv[q];
        else {
          if (lane * CPL + 2 * q < W) o[0] = cvv[q].x;
          if (lane * CPL + 2 * q + 1 < W) o[1] = cvv[q].y;
        }
      }
    }
  }
  if (ONE_SD) {   // sum over pixels of -0.5 ((x - c) / sd)^2 - log sd - log sqrt(2 pi)
    const float inv_sd = 1.0f / a.std_fg, lp0 = -logf(a.std_fg) - 0.91893853320467274178f;
    const int nrows = (H - wave + WAVES - 1) / WAVES;
    float cnt = 0.0f;
#pragma unroll
    for (int q = 0; q < CP; ++q) cnt += vw[q].x + vw[q].y;
    ll = fmaf(-0.5f * inv_sd * inv_sd, ss.x + ss.y, lp0 * cnt * (float)nrows);
  }
  ll = sq_wave_sum(ll);
  if (lane == 0) red_s[wave] = ll;
  __syncthreads();
  if (tid == 0) {
    float dll = red_s[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) dll += red_s[w];
    a.data_ll[frr] = dll;
    if (a.qz != nullptr) {
      const size_t tr = (size_t)a.t * d.R + frr;
      const float kl = qv - pv;
      if (a.out.data_ll_per_sample) a.out.data_ll_per_sample[tr] = dll;
      if (a.out.kl_per_sample) a.out.kl_per_sample[tr] = kl;
      if (a.out.log_q_z_given_x_per_sample) a.out.log_q_z_given_x_per_sample[tr] = qv;
      if (a.out.log_p_z_per_sample) a.out.log_p_z_per_sample[tr] = pv;
      if (a.out.log_weights_per_timestep) a.out.log_weights_per_timestep[tr] = dll - kl;
    }
  }
}
#ifndef SQ_ROWS_WAVES
#define SQ_ROWS_WAVES 4   // wavefronts per workgroup
#endif
#ifndef SQ_ROWS_PD
#define SQ_ROWS_PD 1      // rows of frame / mean-image values in flight per wave
#endif
template <int NMAX, int CPL, bool FULLW, bool ONE_SD>
static int launch_insert_rows2(const InsertArgs& a, const Dims& d, dim3 grid, hipStream_t s) {
  constexpr int WAVES = SQ_ROWS_WAVES, PD = SQ_ROWS_PD;
  const size_t shm = sq_canvas_rows_lds_floats(d.N, d.G, d.H) * sizeof(float);
  if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_insert_loglik_rows<NMAX, CPL, FULLW, ONE_SD, WAVES, PD>, 150 * 1024) != 0) return -2;
  SQ_LAUNCH((k_insert_loglik_rows<NMAX, CPL, FULLW, ONE_SD, WAVES, PD>), grid, dim3(64 * WAVES), shm, s, a, d);
  return 0;
}
template <int NMAX, int CPL>
static int launch_insert_rows(const InsertArgs& a, const Dims& d, dim3 grid, hipStream_t s) {
  // vector loads of a row: W == 64 CPL, and rows / frames that start on the vector's alignment (the parameter buffer's mean image
  // is only float-aligned in general)
  const size_t al = (size_t)CPL * 4 - 1;
  const bool fullw = d.W == 64 * CPL && (((size_t)a.img | (size_t)a.mean_img | (size_t)a.canvas) & al) == 0 && (d.P4 * 4 & al) == 0;
  const bool one_sd = a.std_fg == a.std_bg;
  if (fullw) return one_sd ? launch_insert_rows2<NMAX, CPL, true, true>(a, d, grid, s) : launch_insert_rows2<NMAX, CPL, true, false>(a, d, grid, s);
  return one_sd ? launch_insert_rows2<NMAX, CPL, false, true>(a, d, grid, s) : launch_insert_rows2<NMAX, CPL, false, false>(a, d, grid, s);
}
int sq_launch_insert_loglik(const InsertArgs& a, Dims d, hipStream_t s) {
  const bool wide = d.W > SQ_CANVAS_WIDE;
  const int band_rows = sq_canvas_band_rows(d.H, d.W, wide ? SQ_CANVAS_PF_FWD_W : SQ_CANVAS_PF_FWD);
  const size_t shm = sq_canvas_lds_floats(d.N, d.G, d.H, d.W, band_rows) * sizeof(float);
  const dim3 grid(d.R, a.n_frames > 0 ? a.n_frames : 1);
  // frames of 65 .. 256 columns with up to 8 slots: the row-wave kernel (cfg-5: 89 -> see DESIGN); everything else in bands
  if (wide && d.W <= 256 && d.N <= 8 && d.G >= 2 && d.G <= 20 && sq_canvas_rows_lds_floats(d.N, d.G, d.H) * sizeof(float) <= 150 * 1024) {
    if (d.W <= 128) return d.N <= 4 ? launch_insert_rows<4, 2>(a, d, grid, s) : launch_insert_rows<8, 2>(a, d, grid, s);
    return d.N <= 4 ? launch_insert_rows<4, 4>(a, d, grid, s) : launch_insert_rows<8, 4>(a, d, grid, s);
  }
  if (wide) {
    if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_insert_loglik<SQ_CANVAS_PF_FWD_W, SQ_CANVAS_ROWS_FWD_W>, 150 * 1024) != 0) return -2;
    SQ_LAUNCH((k_insert_loglik<SQ_CANVAS_PF_FWD_W, SQ_CANVAS_ROWS_FWD_W>), grid, dim3(256), shm, s, a, d, band_rows);
  } else {
    if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_insert_loglik<SQ_CANVAS_PF_FWD, SQ_CANVAS_ROWS_FWD>, 150 * 1024) != 0) return -2;
    SQ_LAUNCH((k_insert_loglik<SQ_CANVAS_PF_FWD, SQ_CANVAS_ROWS_FWD>), grid, dim3(256), shm, s, a, d, band_rows);
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// IWAE / VIMCO reductions over [T, B, K] in one launch (reference: sqair/model.py:88-103,:150-158,
// :202-205, sqair/targets.py:38-75, sqair/ops.py:52-59).  K particles of a sequence sit on the lanes
// of a wavefront; sums over T are serial, logsumexp / leave-one-out terms are wave reductions.
// ------------------------------------------------------------------------------------------------
struct ElboMeans { const float* p[8]; };

__global__ __launch_bounds__(1024) void k_elbo(const float* __restrict__ log_w_t, const float* __restrict__ disc_lp_t,
                                              int T, int B, int K, float* log_weights, float* elbo_per_ex, float* iw_out,
                                              float* signal_out, float* scalars, ElboMeans means, int n_means, const int reinforce,
                                              float* means_out SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ float acc_s[16][4 + 8];
  __shared__ float stage_s[16][10][64];   // per wave: one strip of 64 (frame, particle) values per array
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int R = B * K;
  const int fpi = 64 / K, tl = min(lane / K, fpi - 1), kl = lane - (lane / K) * K;   // frames per load instruction (K <= 64)
  float a_vae = 0.0f, a_iwae = 0.0f, a_vimco = 0.0f, a_ess = 0.0f;
  float a_means[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // Two sequences per trip (b0 and b0 + 16), and EVERYTHING both need from memory -- T log-weights, T discrete log-probs and
  // T values of each importance-weighted mean per particle -- requested before the first reduction, without branches between
  // the loads: the values were just written by workgroups all over the chip, a round trip to them is ~2 us, and the kernel used
  // to make ~20 of them one after the other (17 us for 64 KB).
  for (int b0 = wave; b0 < B; b0 += 32) {
    float lw2[2], dl2[2], xm2[2][8];
    {
      const float* src[10];
      src[0] = log_w_t;
      src[1] = disc_lp_t != nullptr ? disc_lp_t : log_w_t;
#pragma unroll
      for (int q = 0; q < 8; ++q) src[2 + q] = q < n_means ? means.p[q] : log_w_t;
      float acc[2][10];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 10; ++i) acc[s2][i] = 0.0f;
      // lane = (frame in the chunk, particle): ONE load instruction per array fetches 64 / K frames of a sequence (all T = 10
      // of the benchmark), so a trip's 2 x 10 loads cover both sequences; the frames then meet through a wave-private LDS
      // strip, where lane k adds its particle's values in frame order
      float* strip = &stage_s[wave][0][0];
      for (int t0 = 0; t0 < T; t0 += fpi) {
        const int tt = t0 + tl;
        const bool ld = lane < fpi * K && tt < T;
        float v[2][10];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const size_t o = (size_t)min(tt, T - 1) * R + (size_t)min(b0 + 16 * s2, B - 1) * K + kl;
#pragma unroll
          for (int i = 0; i < 10; ++i) v[s2][i] = src[i][o];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int i = 0; i < 10; ++i) strip[i * 64 + lane] = ld ? v[s2][i] : 0.0f;
          __builtin_amdgcn_wave_barrier();   // (one wave: its LDS operations complete in order)
          if (lane < K)
            for (int f = 0; f < fpi && t0 + f < T; ++f)
#pragma unroll
              for (int i = 0; i < 10; ++i) acc[s2][i] += strip[i * 64 + f * K + lane];
          __builtin_amdgcn_wave_barrier();
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        lw2[s2] = acc[s2][0];
        dl2[s2] = disc_lp_t != nullptr ? acc[s2][1] : 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) xm2[s2][q] = acc[s2][2 + q] / (float)T;
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int b = b0 + 16 * s2;
      const bool act = lane < K && b < B;   // (b >= B, wave-uniform: contributes zeros, stores nothing)
      const float lw = act ? lw2[s2] : 0.0f, dl = act ? dl2[s2] : 0.0f;
      const float mx = sq_wave_max(act ? lw : -3.0e38f);
      const float ex = act ? expf(lw - mx) : 0.0f;
      const float se = sq_wave_sum(ex);
      const float lse = mx + logf(se);
      const float elbo = lse - logf((float)K);
      const float w = act ? ex / se : 0.0f;
      // VIMCO control variate (targets.py:46-59): replace w_k by the mean of the others, logmeanexp
      const float sum_lw = sq_wave_sum(act ? lw : 0.0f);
      float cv = 0.0f;
      {  // K == 1 gives 0/0 = NaN exactly like the reference's (k_particles - 1.) division (targets.py:55)
        const float abo = (sum_lw - lw) / ((float)K - 1.0f);
        float m2 = abo;  // max over the K terms of the leave-one-out sum (own weight replaced by abo)
        for (int j = 0; j < K; ++j) {
          const float lj = sq_read_lane(lw, j);
          if (j != lane) m2 = fmaxf(m2, lj);
        }
        float rest = 0.0f;  // sum_{j != k} exp(lw_j - m2), exact (no cancellation)
        for (int j = 0; j < K; ++j) {
          const float lj = sq_read_lane(lw, j);
          if (j != lane) rest += expf(lj - m2);
        }
        cv = m2 + logf(rest + expf(abo - m2)) - logf((float)K);
      }
      // learning signal: VIMCO's log w - control variate (targets.py:62-75), or plain REINFORCE's log w (targets.py:78-89)
      const float sig = act ? (reinforce ? lw : lw - cv) : 0.0f;
      const float loss = act ? (-elbo - sig * dl) : 0.0f;
      a_vae += sq_wave_sum(act ? lw : 0.0f);
      a_vimco += sq_wave_sum(loss);
      const float sw = sq_wave_sum(w), sw2 = sq_wave_sum(w * w);
      a_iwae += b < B ? elbo : 0.0f;
      a_ess += b < B ? sw * sw / sw2 : 0.0f;
      if (act) {
        if (log_weights) log_weights[b * K + lane] = lw;
        if (iw_out) iw_out[b * K + lane] = w;
        if (signal_out) signal_out[b * K + lane] = sig;
      }
      if (lane == 0 && b < B && elbo_per_ex) elbo_per_ex[b] = elbo;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < n_means) a_means[q] += sq_wave_sum(w * (act ? xm2[s2][q] : 0.0f));  // mean over (B,K) of iw * x * K == mean_b sum_k iw x
    }
  }
  if (lane == 0) {
    acc_s[wave][0] = a_vae; acc_s[wave][1] = a_iwae; acc_s[wave][2] = a_vimco; acc_s[wave][3] = a_ess;
    for (int q = 0; q < 8; ++q) acc_s[wave][4 + q] = a_means[q];
  }
  __syncthreads();
  if (threadIdx.x < 12) {   // column i of the per-wave partial sums on thread i, waves in order
    const int i = threadIdx.x;
    float t = 0.0f;
    for (int wv = 0; wv < 16; ++wv) t += acc_s[wv][i];
    if (i < 4) {
      if (scalars) scalars[i] = i == 0 ? t / (float)(B * K) : (i == 2 ? t / (float)(B * K) / (float)T : t / (float)B);
    } else if (i - 4 < n_means) {
      means_out[i - 4] = t / (float)B;
    }
  }
}

// The same reductions for K > 64 particles (the lane-per-particle kernel above covers the shipped K = 5 and anything up to a
// wavefront): one workgroup, particle k of the current sequence on thread k (K <= 1024), sequences one after the other, the
// sums of a sequence through LDS in a fixed order.  A slow path by design -- a few hundred KB of input -- with the formulas of
// k_elbo term by term (model.py:88-103, :150-158, :202-205; targets.py:38-75; ops.py:52-59).
__global__ __launch_bounds__(1024) void k_elbo_wide(const float* __restrict__ log_w_t, const float* __restrict__ disc_lp_t,
                                                   int T, int B, int K, float* log_weights, float* elbo_per_ex, float* iw_out,
                                                   float* signal_out, float* scalars, ElboMeans means, int n_means, const int reinforce,
                                                   float* means_out SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ float lw_s[1024], red_s[1024];
  const int k = threadIdx.x, R = B * K;
  const bool act = k < K;
  // block-wide max / sum of one value per thread, result in every thread (tree over LDS: the same order on every run)
  auto bmax = [&](float v) { red_s[k] = v; __syncthreads(); for (int o = 512; o > 0; o >>= 1) { if (k < o) red_s[k] = fmaxf(red_s[k], red_s[k + o]); __syncthreads(); } const float r = red_s[0]; __syncthreads(); return r; };
  auto bsum = [&](float v) { red_s[k] = v; __syncthreads(); for (int o = 512; o > 0; o >>= 1) { if (k < o) red_s[k] += red_s[k + o]; __syncthreads(); } const float r = red_s[0]; __syncthreads(); return r; };
  float a_vae = 0.0f, a_iwae = 0.0f, a_vimco = 0.0f, a_ess = 0.0f, a_means[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < B; ++b) {
    float lw = 0.0f, dl = 0.0f, xm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (act)
      for (int t = 0; t < T; ++t) {
        const size_t o = (size_t)t * R + (size_t)b * K + k;
        lw += log_w_t[o];
        if (disc_lp_t != nullptr) dl += disc_lp_t[o];
        for (int q = 0; q < n_means; ++q) xm[q] += means.p[q][o];
      }
    for (int q = 0; q < n_means; ++q) xm[q] /= (float)T;
    lw_s[k] = act ? lw : -3.0e38f;
    const float mx = bmax(act ? lw : -3.0e38f);
    const float ex = act ? expf(lw - mx) : 0.0f;
    const float se = bsum(ex);
    const float elbo = mx + logf(se) - logf((float)K);
    const float w = act ? ex / se : 0.0f;
    const float sum_lw = bsum(act ? lw : 0.0f);
    float cv = 0.0f;
    if (act) {   // VIMCO control variate (targets.py:46-59): own weight replaced by the mean of the others, logmeanexp
      const float abo = (sum_lw - lw) / ((float)K - 1.0f);
      float m2 = abo;
      for (int j = 0; j < K; ++j)
        if (j != k) m2 = fmaxf(m2, lw_s[j]);
      float rest = 0.0f;
      for (int j = 0; j < K; ++j)
        if (j != k) rest += expf(lw_s[j] - m2);
      cv = m2 + logf(rest + expf(abo - m2)) - logf((float)K);
    }
    const float sig = act ? (reinforce ? lw : lw - cv) : 0.0f;
    const float loss = act ? (-elbo - sig * dl) : 0.0f;
    a_vae += sum_lw;
    a_vimco += bsum(loss);
    const float sw = bsum(w), sw2 = bsum(w * w);
    a_iwae += elbo;
    a_ess += sw * sw / sw2;
    if (act) {
      if (log_weights) log_weights[b * K + k] = lw;
      if (iw_out) iw_out[b * K + k] = w;
      if (signal_out) signal_out[b * K + k] = sig;
    }
    if (k == 0 && elbo_per_ex) elbo_per_ex[b] = elbo;
    for (int q = 0; q < n_means; ++q) a_means[q] += bsum(w * (act ? xm[q] : 0.0f));
  }
  if (k == 0) {
    if (scalars) {
      scalars[0] = a_vae / (float)(B * K);
      scalars[1] = a_iwae / (float)B;
      scalars[2] = a_vimco / (float)(B * K) / (float)T;
      scalars[3] = a_ess / (float)B;
    }
    for (int q = 0; q < n_means; ++q) means_out[q] = a_means[q] / (float)B;
  }
}

int sq_launch_elbo(const float* log_w_t, const float* disc_lp_t, int T, int B, int K, float* log_weights,
                   float* elbo_per_ex, float* iw, float* signal, float* scalars, const float* const* means_in,
                   int n_means, float* means_out, hipStream_t s, int reinforce) {
  ElboMeans m;
  for (int i = 0; i < 8; ++i) m.p[i] = (means_in != nullptr && i < n_means) ? means_in[i] : nullptr;
  if (K > 64) {
    SQ_LAUNCH(k_elbo_wide, dim3(1), dim3(1024), 0, s, log_w_t, disc_lp_t, T, B, K, log_weights, elbo_per_ex, iw, signal, scalars, m,
              n_means, reinforce, means_out);
    return 0;
  }
  SQ_LAUNCH(k_elbo, dim3(1), dim3(1024), 0, s, log_w_t, disc_lp_t, T, B, K, log_weights, elbo_per_ex, iw,
                     signal, scalars, m, n_means, reinforce, means_out);
  return 0;
}
