// libsqair_hip.so — handle, parameter inventory, weight packing plan, and the launch sequence of the
// SQAIR forward pass.  Host code only; kernels live in sqair_linear.hip / sqair_glue.hip.
//
// The launch sequence of one frame restates SQAIRTimestep + AIRDecoder (reference:
// sqair/sqair_modules.py:446-582, sqair/core.py:164-359, sqair/propagate.py:68-184,
// sqair/modules.py:326-467, sqair/seq.py:179-276) with the loop-invariant work hoisted out of the
// N-slot recurrences:
//   * the input-encoder MLP (re-evaluated N times per frame by the reference, core.py:165) runs once
//     per sequence for all T*B frames as one [T*B, H*W] GEMM, already multiplied by its slice of the
//     discovery RNN's in_to_hidden matrix;
//   * everything in a propagation slot that depends only on t-1 quantities (where-bias MLP, mask MLP —
//     evaluated twice per slot on identical input by the reference, core.py:292,336 —, crop #1 and
//     its encoder, the temporal-state slices of the RNN / transform / steps-predictor / GRU-gate
//     pre-activations, the prior GRU) is batched over the N slots up front (M = B'*N rows);
//   * only the truly sequential part (explaining away: slot k needs slot k-1's sample) remains in
//     the per-slot chain.
#include <algorithm>
#include <mutex>
#include <set>
#include <utility>

#include "sqair_internal.h"
#include "sqair_chain.h"

void sq_set_error(SqairHandle* h, const std::string& msg) {
  if (h) h->err = msg;
}

// ------------------------------------------------------------------------------------------------
// parameter inventory — order of SURVEY.md Appendix C (reference: notebooks/play.ipynb:239-362)
// ------------------------------------------------------------------------------------------------
// A dimension of a parameter as a concatenation of segments, each with its size in the caller's (reference) shapes and in the
// padded shapes the kernels run on.  Integer expressions over it (`4 + nh + 1`, `2 * nh + nw`, `4 * nh`) keep the segment
// structure, so the inventory below reads like the reference's shape arithmetic and still knows where the padding sits.
struct PDim {
  std::vector<std::pair<int, int>> seg;  // (true size, padded size)
  PDim(int n) { seg.push_back({n, n}); }   // NOLINT: implicit on purpose
  PDim(int t, int p) { seg.push_back({t, p}); }
  int t() const { int n = 0; for (auto& s : seg) n += s.first; return n; }
  int p() const { int n = 0; for (auto& s : seg) n += s.second; return n; }
  // position in the padded dimension of every true index
  std::vector<int> map() const {
    std::vector<int> m;
    int base = 0;
    for (auto& s : seg) { for (int i = 0; i < s.first; ++i) m.push_back(base + i); base += s.second; }
    return m;
  }
};
static PDim operator+(PDim a, const PDim& b) { a.seg.insert(a.seg.end(), b.seg.begin(), b.seg.end()); return a; }
static PDim operator+(PDim a, int b) { return a + PDim(b); }
static PDim operator+(int a, const PDim& b) { return PDim(a) + b; }
static PDim operator*(int k, const PDim& a) { PDim r = a; for (int i = 1; i < k; ++i) r = r + a; return r; }

static void add_param(SqairHandle* h, const std::string& name, const PDim& rows, const PDim& cols) {
  ParamEntry e;
  e.name = name;
  e.off = h->n_params;
  e.rows = rows.p();
  e.cols = cols.p();
  e.numel = (int64_t)e.rows * e.cols;
  h->pidx[name] = (int)h->params.size();
  h->params.push_back(e);
  h->n_params += e.numel;
  ParamEntry u;   // the caller's view of the same variable
  u.name = name;
  u.off = h->n_uparams;
  u.rows = rows.t();
  u.cols = cols.t();
  u.numel = (int64_t)u.rows * u.cols;
  h->uparams.push_back(u);
  h->n_uparams += u.numel;
  const std::vector<int> rm = rows.map(), cm = cols.map();
  for (int r : rm)
    for (int c : cm) h->u2i.push_back((int)(e.off + (int64_t)r * e.cols + c));
}
static void add_lin(SqairHandle* h, const std::string& name, const PDim& fin, const PDim& fout) {
  add_param(h, name + ".w", fin, fout);
  add_param(h, name + ".b", 1, fout);
}
static void add_gru(SqairHandle* h, const std::string& name, const PDim& fin, const PDim& nh) {
  const char* g[3] = {"z", "r", "h"};
  for (int i = 0; i < 3; ++i) {
    add_param(h, name + ".w" + g[i], fin, nh);
    add_param(h, name + ".u" + g[i], nh, nh);
    add_param(h, name + ".b" + g[i], 1, nh);
  }
}
int64_t P(const SqairHandle* h, const std::string& name) {
  auto it = h->pidx.find(name);
  if (it == h->pidx.end()) {
    fprintf(stderr, "sqair: unknown parameter %s\n", name.c_str());
    abort();
  }
  return h->params[it->second].off;
}
int PC(const SqairHandle* h, const std::string& name) { return h->params[h->pidx.at(name)].cols; }

static void build_inventory(SqairHandle* h) {
  const SqairConfig& c = h->cfg;
  const int P_ = c.img_h * c.img_w, nw = c.n_what, N = c.n_steps_per_image;
  const int G2 = c.glimpse_size * c.glimpse_size;
  // (true, padded): n_hidden and the steps predictor's hidden width n_hidden // 2 (common_model_flags.py:59-71)
  const PDim nh(h->ucfg.n_hidden, c.n_hidden), nsp(h->ucfg.n_hidden / 2, c.n_hidden / 2);
  add_param(h, "dec.mean_img", c.img_h, c.img_w);
  add_lin(h, "dec.l0", nw, nh);
  add_lin(h, "dec.l1", nh, nh);
  add_lin(h, "dec.l2", nh, G2);
  add_param(h, "dec.output_scale", 1, 1);
  add_param(h, "disc.rnn_init", 1, nh);
  if (c.rnn_cell == RNN_LSTM) add_param(h, "disc.rnn_init_c", 1, nh);  // LSTMState(hidden, cell): adjacent rows
  add_lin(h, "disc.steps_prior.l0", 1, 10);
  add_lin(h, "disc.steps_prior.l1", 10, N + 1);
  add_param(h, "disc.rn.init_state", 1, 4);
  add_param(h, "disc.rn.init_sample", 1, 4);
  add_lin(h, "disc.rn.readout", 4, 8);
  add_lin(h, "disc.rn.cond", 4 + nh + 1, 128);
  add_lin(h, "disc.rn.h2h", 128, 4);
  add_lin(h, "disc.rn.i2h", 4, 4);
  add_lin(h, "enc.what_head", nh, 2 * nw);
  add_lin(h, "enc.mask.l0", nh, 128);
  add_lin(h, "enc.mask.l1", 128, G2);
  add_lin(h, "enc.input.l0", P_, nh);
  add_lin(h, "enc.input.l1", nh, nh);
  add_lin(h, "enc.glimpse.l0", G2, nh);
  add_lin(h, "enc.glimpse.l1", nh, nh);
  add_lin(h, "disc.steps.l0", nh + nw, nsp);
  add_lin(h, "disc.steps.l1", nsp, 1);
  add_lin(h, "disc.transform.l0", nh, nh);
  add_lin(h, "disc.transform.l1", nh, nh);
  add_lin(h, "disc.transform.l2", nh, 8);
  add_param(h, "disc.transform.scale_offset", 1, 1);
  if (c.rnn_cell == RNN_LSTM) add_lin(h, "disc.rnn_lstm", (nh + nh + nw + 4 + 1) + nh, 4 * nh);  // snt.LSTM w_gates [x | h], b_gates
  else if (c.rnn_cell == RNN_GRU) add_gru(h, "disc.rnn_gru", nh + nh + nw + 4 + 1, nh);
  else {
    add_lin(h, "disc.rnn.h2h", nh, nh);
    add_lin(h, "disc.rnn.i2h", nh + nh + nw + 4 + 1, nh);
  }
  add_param(h, "disc.step_prior_bias", 1, N + 1);
  add_param(h, "disc.step_prior_timestep_bias", 1, N + 1);
  if (c.time_cell == CELL_LSTM) add_lin(h, "prop.temporal_lstm", (nh + 4 + 2 * nw) + nh, 4 * nh);  // snt.LSTM: w_gates [x | h] rows, b_gates
  else if (c.time_cell == CELL_VANILLA) {  // snt.VanillaRNN: tanh(in_to_hidden(x) + hidden_to_hidden(h))
    add_lin(h, "prop.temporal_rnn.h2h", nh, nh);
    add_lin(h, "prop.temporal_rnn.i2h", nh + 4 + 2 * nw, nh);
  } else add_gru(h, "prop.temporal_gru", nh + 4 + 2 * nw, nh);
  if (c.prior_cell == CELL_LSTM) add_lin(h, "prop.prior_lstm", (nw + 4) + nh, 4 * nh);
  else if (c.prior_cell == CELL_VANILLA) {
    add_lin(h, "prop.prior_rnn.h2h", nh, nh);
    add_lin(h, "prop.prior_rnn.i2h", nw + 4, nh);
  } else add_gru(h, "prop.prior_gru", nw + 4, nh);
  add_lin(h, "prop.prior_linear", nh, 2 * (4 + nw) + 1);
  add_param(h, "prop.cholesky_scale", 1, 10);
  add_lin(h, "prop.where_bias.l0", nh, 128);
  add_lin(h, "prop.where_bias.l1", 128, 4);
  add_lin(h, "prop.steps.l0", 2 * nh + nw, nsp);
  add_lin(h, "prop.steps.l1", nsp, 1);
  add_lin(h, "prop.transform.l0", nh + 4 + nh, nh);   // [hidden | where_{t-1} | temporal] (core.py:325-326): the ORDER matters to the padding
  add_lin(h, "prop.transform.l1", nh, nh);
  add_lin(h, "prop.transform.l2", nh, 8);
  add_param(h, "prop.transform.scale_offset", 1, 1);
  add_lin(h, "prop.what_head", nh, 2 * nw);
  add_lin(h, "prop.gates", nh, 3 * nw);
  add_param(h, "prop.rnn_init", 1, nh);
  if (c.rnn_cell == RNN_LSTM) {
    add_param(h, "prop.rnn_init_c", 1, nh);
    add_lin(h, "prop.rnn_lstm", (nw + (nw + 5) + (nw + 5) + nh) + nh, 4 * nh);
  } else if (c.rnn_cell == RNN_GRU) {
    add_gru(h, "prop.rnn_gru", nw + (nw + 5) + (nw + 5) + nh, nh);
  } else {
    add_lin(h, "prop.rnn.h2h", nh, nh);
    add_lin(h, "prop.rnn.i2h", nw + (nw + 5) + (nw + 5) + nh, nh);
  }
  add_param(h, "seq.prior_init", 1, nh);
  if (c.prior_cell == CELL_LSTM) add_param(h, "seq.prior_init_c", 1, nh);
  add_param(h, "seq.temporal_init", 1, nh);
  if (c.time_cell == CELL_LSTM) add_param(h, "seq.temporal_init_c", 1, nh);  // LSTMState(hidden, cell): adjacent, read as one [2 nh] row
  add_lin(h, "seq.latent_enc.l0", nw + 4, nh);
  add_lin(h, "seq.latent_enc.l1", nh, nh);

  POff& o = h->po;
  o.dec_mean_img = (int)P(h, "dec.mean_img");
  o.dec_output_scale = (int)P(h, "dec.output_scale");
  o.rn_init_state = (int)P(h, "disc.rn.init_state");
  o.rn_init_sample = (int)P(h, "disc.rn.init_sample");
  o.rn_readout_w = (int)P(h, "disc.rn.readout.w");
  o.rn_readout_b = (int)P(h, "disc.rn.readout.b");
  o.rn_cond_w = (int)P(h, "disc.rn.cond.w");
  o.rn_cond_b = (int)P(h, "disc.rn.cond.b");
  o.rn_h2h_w = (int)P(h, "disc.rn.h2h.w");
  o.rn_h2h_b = (int)P(h, "disc.rn.h2h.b");
  o.rn_i2h_w = (int)P(h, "disc.rn.i2h.w");
  o.rn_i2h_b = (int)P(h, "disc.rn.i2h.b");
  o.sp_l0_w = (int)P(h, "disc.steps_prior.l0.w");
  o.sp_l0_b = (int)P(h, "disc.steps_prior.l0.b");
  o.sp_l1_w = (int)P(h, "disc.steps_prior.l1.w");
  o.sp_l1_b = (int)P(h, "disc.steps_prior.l1.b");
  o.step_prior_bias = (int)P(h, "disc.step_prior_bias");
  o.step_prior_tbias = (int)P(h, "disc.step_prior_timestep_bias");
  o.disc_steps_l1_w = (int)P(h, "disc.steps.l1.w");
  o.disc_steps_l1_b = (int)P(h, "disc.steps.l1.b");
  o.prop_steps_l1_w = (int)P(h, "prop.steps.l1.w");
  o.prop_steps_l1_b = (int)P(h, "prop.steps.l1.b");
  o.disc_scale_offset = (int)P(h, "disc.transform.scale_offset");
  o.prop_scale_offset = (int)P(h, "prop.transform.scale_offset");
  o.cholesky = (int)P(h, "prop.cholesky_scale");
  o.disc_rnn_init = (int)P(h, "disc.rnn_init");
  o.prop_rnn_init = (int)P(h, "prop.rnn_init");
  o.prior_init = (int)P(h, "seq.prior_init");
  o.temporal_init = (int)P(h, "seq.temporal_init");
  h->padded = h->n_params != h->n_uparams;
}

// ------------------------------------------------------------------------------------------------
// packing plan: for every packed weight element the index of its source in the flat buffer
// ------------------------------------------------------------------------------------------------
typedef std::vector<int> RowMap;  // per segment position: source row or -1
static RowMap rm_range(int start, int n) {
  RowMap r(n);
  for (int i = 0; i < n; ++i) r[i] = start + i;
  return r;
}
static RowMap rm_none(int n) { return RowMap(n, -1); }
// z-record segment (rec::ZW wide): where rows, what rows, presence row (-1 = unused)
static RowMap rm_zrec(int nw, int where0, int what0, int pres_row) {
  RowMap r(rec::ZW, -1);
  for (int i = 0; i < 4; ++i) r[rec::WHERE + i] = where0 >= 0 ? where0 + i : -1;
  for (int i = 0; i < nw; ++i) r[rec::WHAT + i] = what0 >= 0 ? what0 + i : -1;
  r[rec::PRES] = pres_row;
  return r;
}

struct SegSrc {
  std::string w;  // source weight matrix name ("" = none)
  RowMap rows;
};
struct ColBlock {
  int ncols;
  int col0;  // first source column
  std::vector<SegSrc> seg;
  std::string bias_a, bias_b;  // bias vector names ("" = none); element col0 + j
};

static void build_layer(SqairHandle* h, LayerId id, const std::vector<int>& seg_width, const std::vector<ColBlock>& blocks) {
  PackedLayer& L = h->layers[id];
  L.seg_width = seg_width;
  L.kc = 0;
  for (int w : seg_width) L.kc += (w + 15) / 16;
  L.N = 0;
  for (const ColBlock& b : blocks) L.N += b.ncols;
  L.nt = (L.N + 15) / 16;
  L.w_off = h->packed_w;
  L.b_off = h->packed_b;
  const int64_t nel = (int64_t)L.nt * L.kc * 256;
  h->widx.resize(h->packed_w + nel, -1);
  h->bidx_a.resize(h->packed_b + L.nt * 16, -1);
  h->bidx_b.resize(h->packed_b + L.nt * 16, -1);
  int n0 = 0;
  for (const ColBlock& b : blocks) {
    if ((int)b.seg.size() != (int)seg_width.size()) {
      fprintf(stderr, "sqair: layer %d: segment count mismatch\n", (int)id);
      abort();
    }
    for (size_t sgi = 0; sgi < seg_width.size(); ++sgi)
      if (!b.seg[sgi].w.empty()) {
        h->wg[id].push_back({n0, b.ncols, b.col0, (int)sgi, b.seg[sgi].w, (int64_t)h->rm_pool.size()});
        h->rm_pool.insert(h->rm_pool.end(), b.seg[sgi].rows.begin(), b.seg[sgi].rows.end());
      }
    if (!b.bias_a.empty() || !b.bias_b.empty()) h->bg[id].push_back({n0, b.ncols, b.col0, b.bias_a, b.bias_b});
    for (int j = 0; j < b.ncols; ++j) {
      const int n = n0 + j;
      const int tile = n / 16, ln = n % 16;
      if (!b.bias_a.empty()) h->bidx_a[L.b_off + n] = (int)(P(h, b.bias_a) + b.col0 + j);
      if (!b.bias_b.empty()) h->bidx_b[L.b_off + n] = (int)(P(h, b.bias_b) + b.col0 + j);
      int cbase = 0;
      for (size_t s = 0; s < seg_width.size(); ++s) {
        const SegSrc& src = b.seg[s];
        const int nch = (seg_width[s] + 15) / 16;
        if (!src.w.empty()) {
          if ((int)src.rows.size() != seg_width[s]) {
            fprintf(stderr, "sqair: layer %d seg %zu: rowmap %zu != width %d\n", (int)id, s, src.rows.size(), seg_width[s]);
            abort();
          }
          const int64_t woff = P(h, src.w);
          const int ldw = PC(h, src.w);
          for (int k = 0; k < seg_width[s]; ++k) {
            if (src.rows[k] < 0) continue;
            const int c = cbase + k / 16, kin = k % 16;
            const int lane = (kin / 4) * 16 + ln, comp = kin % 4;
            h->widx[L.w_off + (((int64_t)tile * L.kc + c) * 64 + lane) * 4 + comp] =
                (int)(woff + (int64_t)src.rows[k] * ldw + b.col0 + j);
          }
        }
        cbase += nch;
      }
    }
    n0 += b.ncols;
  }
  h->packed_w += nel;
  h->packed_b += L.nt * 16;
}

static ColBlock cb1(int ncols, int col0, const std::string& w, const RowMap& rows, const std::string& ba = "",
                    const std::string& bb = "") {
  ColBlock b;
  b.ncols = ncols;
  b.col0 = col0;
  b.seg.push_back({w, rows});
  b.bias_a = ba;
  b.bias_b = bb;
  return b;
}

static void build_plan(SqairHandle* h) {
  const SqairConfig& c = h->cfg;
  h->packed_w = 256;  // leading zero block: one K-chunk of zero weights for the ragged tail of the chunk loop
  h->widx.assign(256, -1);
  const int P_ = c.img_h * c.img_w, nh = c.n_hidden, nw = c.n_what;
  const int G2 = c.glimpse_size * c.glimpse_size, nsp = nh / 2;
  auto simple = [&](LayerId id, const std::string& name, int fin, int fout, int row0 = 0, bool bias = true) {
    build_layer(h, id, {fin}, {cb1(fout, 0, name + ".w", rm_range(row0, fin), bias ? name + ".b" : "")});
  };
  simple(L_IENC0, "enc.input.l0", P_, nh);
  simple(L_IENC1, "enc.input.l1", nh, nh);
  // discovery RNN in_to_hidden = [input enc nh | conditioning nh | what nw | where 4 | presence 1] (core.py:164-177)
  // slot RNN of both cores (flag transition): VanillaRNN -> nh pre-activation columns from in_to_hidden / hidden_to_hidden
  // (two biases); LSTM -> 4 nh gate columns (i, j, f, o) from the [x | h] rows of w_gates (one bias)
  // GRU -> 3 nh columns [z | r | candidate] from w{z,r,h} / u{z,r} / b{z,r,h} (the candidate's recurrent matrix u_h meets
  // r * h in a second launch, L_*_RNN2).
  const bool RL = c.rnn_cell == RNN_LSTM, RG = c.rnn_cell == RNN_GRU;
  const int fin_d = nh + nh + nw + 4 + 1, fin_p = nw + (nw + 5) + (nw + 5) + nh;
  // column blocks of the RNN pre-activation for a layer whose segments take the given rows of the INPUT weights
  // (rows[s] empty = segment s carries no RNN input; rec_seg = the segment holding h_{k-1}, or -1)
  auto rnn_blocks = [&](const std::string& core, const std::vector<RowMap>& rows, int rec_seg, bool bias, int fin) {
    std::vector<ColBlock> out;
    const int ngate = RG ? 3 : 1;
    for (int g = 0; g < ngate; ++g) {
      const std::string gs = std::string(1, "zrh"[g]);
      const std::string W = RL ? core + ".rnn_lstm.w" : (RG ? core + ".rnn_gru.w" + gs : core + ".rnn.i2h.w");
      const std::string U = RL ? core + ".rnn_lstm.w" : (RG ? core + ".rnn_gru.u" + gs : core + ".rnn.h2h.w");
      ColBlock b;
      b.ncols = RL ? 4 * nh : nh; b.col0 = 0;
      for (size_t sgi = 0; sgi < rows.size(); ++sgi) {
        if ((int)sgi == rec_seg) {
          if (RG && g == 2) b.seg.push_back({"", RowMap()});
          else b.seg.push_back({U, rm_range(RL ? fin : 0, nh)});
        } else if (rows[sgi].empty()) b.seg.push_back({"", RowMap()});
        else b.seg.push_back({W, rows[sgi]});
      }
      if (bias) {
        b.bias_a = RL ? core + ".rnn_lstm.b" : (RG ? core + ".rnn_gru.b" + gs : core + ".rnn.i2h.b");
        b.bias_b = (RL || RG) ? "" : core + ".rnn.h2h.b";
      }
      out.push_back(b);
    }
    return out;
  };
  build_layer(h, L_PREDISC, {nh}, rnn_blocks("disc", {rm_range(0, nh)}, -1, true, fin_d));
  // prior cell on [what, where]_{t-1} (propagate.py:78-81)
  if (c.prior_cell == CELL_LSTM) {  // gates (i, j, f, o) = [what, where | h] w_gates + b_gates in ONE layer (both inputs exist up front)
    ColBlock b;
    b.ncols = 4 * nh; b.col0 = 0;
    b.seg = {{"prop.prior_lstm.w", rm_zrec(nw, nw, 0, -1)}, {"prop.prior_lstm.w", rm_range(nw + 4, nh)}};
    b.bias_a = "prop.prior_lstm.b";
    build_layer(h, L_PRIOR_GRU1, {rec::ZW, nh}, {b});
    build_layer(h, L_PRIOR_GRU2, {nh}, {cb1(16, 0, "prop.prior_lstm.w", rm_none(nh))});  // unused placeholder layer
  } else if (c.prior_cell == CELL_VANILLA) {  // tanh([what, where] W_i + h W_h + b_i + b_h): one layer
    ColBlock b;
    b.ncols = nh; b.col0 = 0;
    b.seg = {{"prop.prior_rnn.i2h.w", rm_zrec(nw, nw, 0, -1)}, {"prop.prior_rnn.h2h.w", rm_range(0, nh)}};
    b.bias_a = "prop.prior_rnn.i2h.b"; b.bias_b = "prop.prior_rnn.h2h.b";
    build_layer(h, L_PRIOR_GRU1, {rec::ZW, nh}, {b});
    build_layer(h, L_PRIOR_GRU2, {nh}, {cb1(16, 0, "prop.prior_rnn.h2h.w", rm_none(nh))});  // unused placeholder layer
  } else {
    const RowMap zx = rm_zrec(nw, nw, 0, -1);
    std::vector<ColBlock> bl;
    const char* g[3] = {"z", "r", "h"};
    for (int i = 0; i < 3; ++i) {
      ColBlock b;
      b.ncols = nh;
      b.col0 = 0;
      b.seg.push_back({std::string("prop.prior_gru.w") + g[i], zx});
      if (i < 2) b.seg.push_back({std::string("prop.prior_gru.u") + g[i], rm_range(0, nh)});
      else b.seg.push_back({"", RowMap()});
      b.bias_a = std::string("prop.prior_gru.b") + g[i];
      bl.push_back(b);
    }
    build_layer(h, L_PRIOR_GRU1, {rec::ZW, nh}, bl);
    simple(L_PRIOR_GRU2, "prop.prior_gru.uh", nh, nh, 0, false);
  }
  // fix-up: uh is a bare matrix (no ".w" suffix) -> handled by simple() through name + ".w"; see alias below
  simple(L_PRIOR_LIN, "prop.prior_linear", nh, 2 * (4 + nw) + 1);
  build_layer(h, L_TAU1, {nh},
              {cb1(128, 0, "prop.where_bias.l0.w", rm_range(0, nh), "prop.where_bias.l0.b"),
               cb1(128, 0, "enc.mask.l0.w", rm_range(0, nh), "enc.mask.l0.b")});
  simple(L_WB2, "prop.where_bias.l1", 128, 4);
  simple(L_MASK2, "enc.mask.l1", 128, G2);
  simple(L_GENC0, "enc.glimpse.l0", G2, nh);
  simple(L_GENC1, "enc.glimpse.l1", nh, nh);
  build_layer(h, L_WHAT_LOC, {nh}, {cb1(nw, 0, "enc.what_head.w", rm_range(0, nh), "enc.what_head.b")});
  simple(L_WHAT_HEAD, "enc.what_head", nh, 2 * nw);
  {
    // the same layer with its output columns INTERLEAVED (loc_c, scale_c adjacent): a discovery slot's what sample
    // loc + scale * eps is then one lane pair of the layer's epilogue (k_linear_what, sqair_glue.hip) instead of 16 rows x nw
    // elements re-derived in every column-tile workgroup of the next slot's fused RNN + tail launch.  Forward-only (inference):
    // the training pass keeps L_WHAT_HEAD and its tape layout.
    std::vector<ColBlock> bl;
    for (int cc = 0; cc < nw; ++cc) {
      bl.push_back(cb1(1, cc, "enc.what_head.w", rm_range(0, nh), "enc.what_head.b"));
      bl.push_back(cb1(1, nw + cc, "enc.what_head.w", rm_range(0, nh), "enc.what_head.b"));
    }
    build_layer(h, L_WHAT_HEAD_I, {nh}, bl);
  }
  // loop-invariant pre-activations of a propagation slot, segments [m1 (nw) | z_{t-1} record | temporal state]
  {
    const int tm1 = nw + (nw + 5);  // rnn input: [loc1 nw | what,where,pres (k-1) | what,where,pres (t-1) | temporal]
    std::vector<ColBlock> bl;
    bl = rnn_blocks("prop", {rm_range(0, nw), rm_zrec(nw, tm1 + nw, tm1, tm1 + nw + 4), rm_range(tm1 + nw + 5, nh)}, -1, true, fin_p);
    ColBlock t1;  // transform input [hidden nh | where_{t-1} 4 | temporal nh] (core.py:325-326)
    t1.ncols = nh; t1.col0 = 0;
    t1.seg = {{"", RowMap()}, {"prop.transform.l0.w", rm_zrec(nw, nh, -1, -1)}, {"prop.transform.l0.w", rm_range(nh + 4, nh)}};
    t1.bias_a = "prop.transform.l0.b";
    bl.push_back(t1);
    ColBlock s1;  // steps predictor input [hidden nh | temporal nh | what nw] (core.py:312-314)
    s1.ncols = nsp; s1.col0 = 0;
    s1.seg = {{"", RowMap()}, {"", RowMap()}, {"prop.steps.l0.w", rm_range(nh, nh)}};
    s1.bias_a = "prop.steps.l0.b";
    bl.push_back(s1);
    if (c.time_cell == CELL_VANILLA) {  // the whole recurrent term of the temporal VanillaRNN is loop-invariant
      ColBlock u;
      u.ncols = nh; u.col0 = 0;
      u.seg = {{"", RowMap()}, {"", RowMap()}, {"prop.temporal_rnn.h2h.w", rm_range(0, nh)}};
      u.bias_a = "prop.temporal_rnn.i2h.b"; u.bias_b = "prop.temporal_rnn.h2h.b";
      bl.push_back(u);
    }
    for (const char* g : {"z", "r"}) {
      if (c.time_cell != CELL_GRU) break;  // (the LSTM's recurrent rows read the HIDDEN half of the state: their own layer below)
      ColBlock u;
      u.ncols = nh; u.col0 = 0;
      u.seg = {{"", RowMap()}, {"", RowMap()}, {std::string("prop.temporal_gru.u") + g, rm_range(0, nh)}};
      u.bias_a = std::string("prop.temporal_gru.b") + g;
      bl.push_back(u);
    }
    build_layer(h, L_PRE, {nw, rec::ZW, nh}, bl);
  }
  {
    // explaining-away + recurrent part of the propagation RNN
    build_layer(h, L_PROP_RNN, {rec::ZW, nh}, rnn_blocks("prop", {rm_zrec(nw, nw + nw, nw, nw + nw + 4), RowMap()}, 1, false, fin_p));
    if (RG) simple(L_PROP_RNN2, "prop.rnn_gru.uh", nh, nh, 0, false);
    else build_layer(h, L_PROP_RNN2, {nh}, {cb1(16, 0, "prop.what_head.w", rm_none(nh))});  // unused placeholder
  }
  // transform hidden layer 1 + (extra columns) the r_k rows of the steps predictor's hidden layer: both consume r_k
  build_layer(h, L_PROP_T1, {nh},
              {cb1(nh, 0, "prop.transform.l0.w", rm_range(0, nh)), cb1(nsp, 0, "prop.steps.l0.w", rm_range(0, nh))});
  simple(L_PROP_T2, "prop.transform.l1", nh, nh);
  simple(L_PROP_T3, "prop.transform.l2", nh, 8);
  if (c.time_cell == CELL_LSTM) {
    // temporal LSTM (core.py:340-341 with time_transition=LSTM): gate pre-activations (i, j, f, o) = [x | h] w_gates + b.
    // L_PROP_GRU1 holds the x rows [hidden nh | where 4 | loc nw | scale nw], L_PROP_GRU2 the h rows + bias (applied to
    // all slots of a frame at once, before the slot loop).
    const std::string w = "prop.temporal_lstm.w";
    const int fin = nh + 4 + 2 * nw;
    ColBlock b;
    b.ncols = 4 * nh; b.col0 = 0;
    b.seg = {{w, rm_range(0, nh)}, {w, rm_range(nh, 4)}, {w, rm_range(nh + 4, 2 * nw)}};
    build_layer(h, L_PROP_GRU1, {nh, 4, 2 * nw}, {b});
    build_layer(h, L_PROP_GRU2, {nh}, {cb1(4 * nh, 0, w, rm_range(fin, nh), "prop.temporal_lstm.b")});
  } else if (c.time_cell == CELL_VANILLA) {
    const std::string w = "prop.temporal_rnn.i2h.w";
    ColBlock b;
    b.ncols = nh; b.col0 = 0;
    b.seg = {{w, rm_range(0, nh)}, {w, rm_range(nh, 4)}, {w, rm_range(nh + 4, 2 * nw)}};
    build_layer(h, L_PROP_GRU1, {nh, 4, 2 * nw}, {b});
    build_layer(h, L_PROP_GRU2, {nh}, {cb1(16, 0, "prop.temporal_rnn.h2h.w", rm_none(nh))});  // unused placeholder layer
  } else {
    // temporal GRU input [hidden nh | where 4 | loc nw | scale nw] (core.py:340-341)
    std::vector<ColBlock> bl;
    const char* g[3] = {"z", "r", "h"};
    for (int i = 0; i < 3; ++i) {
      ColBlock b;
      b.ncols = nh; b.col0 = 0;
      const std::string w = std::string("prop.temporal_gru.w") + g[i];
      b.seg = {{w, rm_range(0, nh)}, {w, rm_range(nh, 4)}, {w, rm_range(nh + 4, 2 * nw)}};
      if (i == 2) b.bias_a = "prop.temporal_gru.bh";
      bl.push_back(b);
    }
    build_layer(h, L_PROP_GRU1, {nh, 4, 2 * nw}, bl);
    simple(L_PROP_GRU2, "prop.temporal_gru.uh", nh, nh, 0, false);
  }
  build_layer(h, L_PROP_HEADS, {nh},
              {cb1(2 * nw, 0, "prop.what_head.w", rm_range(0, nh), "prop.what_head.b"),
               cb1(3 * nw, 0, "prop.gates.w", rm_range(0, nh), "prop.gates.b")});
  {
    // ... and the heads of a propagation slot with the FIVE pre-activations of a `what` element (temporal loc, temporal scale,
    // forget / input / temporal gate: hraw columns c, nw + c, 2 nw + c, 3 nw + c, 4 nw + c) in five adjacent columns, three
    // elements per 16-column tile (column 15 of a tile is empty): the gated what sample (sqair/core.py:336-359) is then computed by
    // the lanes of the layer's epilogue.  Forward-only as well.
    std::vector<ColBlock> bl;
    auto empty = [&]() { ColBlock b; b.ncols = 1; b.col0 = 0; b.seg = {{"", RowMap()}}; return b; };
    for (int t = 0; t < (nw + 2) / 3; ++t) {
      for (int el = 0; el < 3; ++el) {
        const int cc = 3 * t + el;
        for (int g = 0; g < 5; ++g) {
          if (cc >= nw) { bl.push_back(empty()); continue; }
          if (g < 2) bl.push_back(cb1(1, g * nw + cc, "prop.what_head.w", rm_range(0, nh), "prop.what_head.b"));
          else bl.push_back(cb1(1, (g - 2) * nw + cc, "prop.gates.w", rm_range(0, nh), "prop.gates.b"));
        }
      }
      bl.push_back(empty());
    }
    build_layer(h, L_PROP_HEADS_I, {nh}, bl);
  }
  {
    ColBlock b;
    b.ncols = nsp; b.col0 = 0;
    b.seg = {{"prop.steps.l0.w", rm_zrec(nw, -1, 2 * nh, -1)}};  // `what` rows only; consumed by k_slot_tail
    build_layer(h, L_PROP_S1, {rec::ZW}, {b});
  }
  build_layer(h, L_LAT0, {rec::ZW}, {cb1(nh, 0, "seq.latent_enc.l0.w", rm_zrec(nw, nw, 0, -1), "seq.latent_enc.l0.b")});
  simple(L_LAT1, "seq.latent_enc.l1", nh, nh);
  build_layer(h, L_PRED, {nh}, rnn_blocks("disc", {rm_range(nh, nh)}, -1, false, fin_d));
  {
    ColBlock b;  // conditioning state of the recurrent where prior: [init_state 4 | cond nh | e 1] (modules.py:573-576)
    b.ncols = 128; b.col0 = 0;
    b.seg = {{"disc.rn.cond.w", rm_range(0, 4)}, {"disc.rn.cond.w", rm_range(4, nh)}};
    b.bias_a = "disc.rn.cond.b";
    build_layer(h, L_RNCOND, {4, nh}, {b});
  }
  {
    build_layer(h, L_DISC_RNN, {rec::ZW, nh}, rnn_blocks("disc", {rm_zrec(nw, 2 * nh + nw, 2 * nh, 2 * nh + nw + 4), RowMap()}, 1, false, fin_d));
    if (RG) simple(L_DISC_RNN2, "disc.rnn_gru.uh", nh, nh, 0, false);
    else build_layer(h, L_DISC_RNN2, {nh}, {cb1(16, 0, "prop.what_head.w", rm_none(nh))});  // unused placeholder
  }
  build_layer(h, L_DISC_T1, {nh},
              {cb1(nh, 0, "disc.transform.l0.w", rm_range(0, nh), "disc.transform.l0.b"),
               cb1(nsp, 0, "disc.steps.l0.w", rm_range(0, nh), "disc.steps.l0.b")});
  simple(L_DISC_T2, "disc.transform.l1", nh, nh);
  simple(L_DISC_T3, "disc.transform.l2", nh, 8);
  {
    ColBlock b;
    b.ncols = nsp; b.col0 = 0;
    b.seg = {{"disc.steps.l0.w", rm_zrec(nw, -1, nh, -1)}};  // `what` rows only; consumed by k_slot_tail
    build_layer(h, L_DISC_S1, {rec::ZW}, {b});
  }
  build_layer(h, L_DEC0, {rec::ZW}, {cb1(nh, 0, "dec.l0.w", rm_zrec(nw, -1, 0, -1), "dec.l0.b")});
  simple(L_DEC1, "dec.l1", nh, nh);
  simple(L_DEC2, "dec.l2", nh, G2);
}

// Transposed packs for the backward pass: dA_cat [M, 16 kc] = dPre [M, N] W^T with the same zero padding, so the
// gradient of a z-record segment comes out in record order.  Built from the forward plan element by element.
static void build_plan_T(SqairHandle* h) {
  for (int id = 0; id < L_COUNT; ++id) {
    const PackedLayer& L = h->layers[id];
    PackedLayer& T = h->layersT[id];
    T.kc = L.nt;            // reduction over the (padded) forward outputs
    T.nt = L.kc;            // one output tile per forward K-chunk
    T.N = L.kc * 16;
    T.seg_width = {L.N};
    T.w_off = h->packed_w;
    T.b_off = h->packed_b;
    const int64_t nel = (int64_t)T.nt * T.kc * 256;
    h->widx.resize(h->packed_w + nel, -1);
    h->bidx_a.resize(h->packed_b + T.nt * 16, -1);
    h->bidx_b.resize(h->packed_b + T.nt * 16, -1);
    for (int k = 0; k < L.kc * 16; ++k)
      for (int n = 0; n < L.nt * 16; ++n) {
        const int64_t f = L.w_off + ((((int64_t)(n / 16) * L.kc + k / 16) * 64) + ((k % 16) / 4) * 16 + n % 16) * 4 + k % 4;
        const int src = h->widx[f];
        if (src < 0) continue;
        const int64_t t = T.w_off + ((((int64_t)(k / 16) * T.kc + n / 16) * 64) + ((n % 16) / 4) * 16 + k % 16) * 4 + n % 4;
        h->widx[t] = src;
      }
    h->packed_w += nel;
    h->packed_b += T.nt * 16;
  }
}

// ------------------------------------------------------------------------------------------------
// C-ABI: lifetime / introspection
// ------------------------------------------------------------------------------------------------
int sq_allow_big_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -2;
  std::lock_guard<std::mutex> g(mu);
  if (done.count({kernel, dev})) return 0;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    fprintf(stderr, "sqair: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d\n", bytes, dev);
    (void)hipGetLastError();
    return -2;
  }
  done.insert({kernel, dev});
  return 0;
}

extern "C" int sqair_abi_version(void) { return SQAIR_ABI_VERSION; }
// identity of this binary: the hash of the sources it was compiled from (csrc/build.py passes it in) and the build variant.
// The marker string makes the id readable from the file without loading it.
#ifndef SQAIR_BUILD_ID
#define SQAIR_BUILD_ID "0000000000000000"
#endif
#ifndef SQAIR_BUILD_VARIANT
#define SQAIR_BUILD_VARIANT "unknown"
#endif
__attribute__((used)) static const char sq_build_marker[] = "SQAIR_BUILD_ID=" SQAIR_BUILD_ID ";";
extern "C" const char* sqair_build_id(void) { return SQAIR_BUILD_ID; }
extern "C" const char* sqair_build_flags(void) { return SQAIR_BUILD_VARIANT; }

extern "C" int sqair_create(const SqairConfig* cfg, SqairHandle** out) {
  if (cfg == nullptr || out == nullptr) return -1;
  *out = nullptr;
  // The limits of this build (sqair_limits below).  Inside them every value of the reference's flags is accepted
  // (common_model_flags.py:32-56, configs/mlp_mnist_model.py:42-52 take any integer): n_hidden = 32 n_units is padded to the next
  // multiple of 128 with inert units (SqairHandle), frames of any H x W (rows that are not 16-byte multiples are staged through a
  // padded copy), any number of particles.
  if (cfg->n_what < 1 || cfg->n_what > SQ_MAX_NWHAT || cfg->n_steps_per_image < 1 || cfg->n_steps_per_image > SQ_MAXN ||
      cfg->n_hidden < 16 || cfg->n_hidden > SQ_MAX_NHIDDEN || (cfg->n_hidden % 16) != 0 ||  // (n_hidden // 2 is a whole number of units)
      cfg->glimpse_size < 2 || cfg->img_h < 2 || cfg->img_w < 2 || cfg->k_particles < 1 || cfg->k_particles > SQ_MAX_K)
    return -1;
  SqairHandle* h = new SqairHandle();
  h->ucfg = *cfg;
  h->cfg = *cfg;
  // (128, 256 or 512: the slot-tail adjoint indexes its LDS copy of steps.l0 with shifts -- n_hidden / 2 a power of two)
  h->cfg.n_hidden = cfg->n_hidden <= 128 ? 128 : (cfg->n_hidden <= 256 ? 256 : 512);
  {  // the log-probability adjoint stages a row's 2N records, their gradients and the prior statistics in LDS (k_logprob_bwd)
    const int64_t N = cfg->n_steps_per_image;
    const int64_t lds = (4 * N * rec::W + 2 * N * rec::ZW + 2 * N * PS_LD + 2 * (800 + 13 * (N + 1))) * 4;
    if (lds > 150 * 1024) { delete h; return -1; }   // (only the wide build can get here: n_steps_per_image 15, 16 with its 416-float record)
  }
  build_inventory(h);
  // GRU candidate matrices are bare [nh, nh] parameters: give build_plan's simple() a ".w" alias
  if (cfg->prior_cell == CELL_GRU) h->pidx["prop.prior_gru.uh.w"] = h->pidx["prop.prior_gru.uh"];
  if (cfg->rnn_cell == RNN_GRU) {
    h->pidx["prop.rnn_gru.uh.w"] = h->pidx["prop.rnn_gru.uh"];
    h->pidx["disc.rnn_gru.uh.w"] = h->pidx["disc.rnn_gru.uh"];
  }
  if (cfg->time_cell == CELL_GRU) h->pidx["prop.temporal_gru.uh.w"] = h->pidx["prop.temporal_gru.uh"];
  build_plan(h);
  build_plan_T(h);
  *out = h;
  return 0;
}

#ifdef SQAIR_TIMELINE
void sq_tl_forget(const SqairHandle* h);
#endif
extern "C" int sqair_destroy(SqairHandle* h) {
  if (h == nullptr) return 0;
#ifdef SQAIR_TIMELINE
  sq_tl_forget(h);
#endif
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  for (int i = 0; i < 4; ++i) {
    if (h->cap_exec[i]) (void)hipGraphExecDestroy(h->cap_exec[i]);
    if (h->cap_graph[i]) (void)hipGraphDestroy(h->cap_graph[i]);
  }
  sq_chain_destroy(h);
  delete h;
  return 0;
}

extern "C" const char* sqair_last_error(const SqairHandle* h) { return h ? h->err.c_str() : "null handle"; }
// (the caller's inventory: reference shapes, whatever the kernels pad internally)
extern "C" int64_t sqair_param_count(const SqairHandle* h) { return h ? h->n_uparams : -1; }
extern "C" int sqair_param_entries(const SqairHandle* h) { return h ? (int)h->uparams.size() : -1; }
extern "C" int sqair_param_entry(const SqairHandle* h, int i, const char** name, int64_t* offset, int64_t* numel) {
  if (!h || i < 0 || i >= (int)h->uparams.size()) return -1;
  if (name) *name = h->uparams[i].name.c_str();
  if (offset) *offset = h->uparams[i].off;
  if (numel) *numel = h->uparams[i].numel;
  return 0;
}
extern "C" int sqair_get_config(const SqairHandle* h, SqairConfig* out) {
  if (!h || !out) return -1;
  *out = h->ucfg;
  return 0;
}
extern "C" int sqair_noise_width(const SqairHandle* h) { return h ? 4 + h->cfg.n_what + 1 : -1; }


extern "C" int64_t sqair_packed_bytes(const SqairHandle* h) { return h ? packed_layout(h).total * 4 : -1; }

// caller's flat buffer <-> the padded one (padded configurations only): element j of the caller's lives at u2i[j]
__global__ void k_flat_scatter(const float* __restrict__ user, float* __restrict__ padded, const int* __restrict__ u2i, int64_t n SQ_TLP) {
  SQ_TL_SCOPE;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) padded[u2i[i]] = user[i];
}
__global__ void k_flat_gather(const float* __restrict__ padded, float* __restrict__ user, const int* __restrict__ u2i, int64_t n SQ_TLP) {
  SQ_TL_SCOPE;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) user[i] = padded[u2i[i]];
}
// gradient in the padded shapes -> the caller's flat gradient buffer (the padding's entries are dropped)
void sq_flat_gather(const SqairHandle* h, const float* padded_grad, float* user_grad, const void* packed, hipStream_t s) {
  const PackedLayout pl = packed_layout(h);
  SQ_LAUNCH(k_flat_gather, dim3((unsigned)std::min<int64_t>(2048, (h->n_uparams + 255) / 256)), dim3(256), 0, s, padded_grad, user_grad,
            (const int*)packed + pl.um, h->n_uparams);
}

extern "C" int sqair_pack_params(SqairHandle* h, const float* flat, void* packed, void* stream) {
  if (!h || !flat || !packed) return -1;
  hipStream_t s = (hipStream_t)stream;
  const PackedLayout pl = packed_layout(h);
  float* base = (float*)packed;
  int* ibase = (int*)packed;
  if (h->padded && h->plan_uploaded_ptr != packed)
    SQ_CHECK_HIP(hipMemcpyAsync(ibase + pl.um, h->u2i.data(), h->u2i.size() * 4, hipMemcpyHostToDevice, s));
  if (h->plan_uploaded_ptr != packed) {
    SQ_CHECK_HIP(hipMemcpyAsync(ibase + pl.wi, h->widx.data(), h->packed_w * 4, hipMemcpyHostToDevice, s));
    SQ_CHECK_HIP(hipMemcpyAsync(ibase + pl.ba, h->bidx_a.data(), h->packed_b * 4, hipMemcpyHostToDevice, s));
    SQ_CHECK_HIP(hipMemcpyAsync(ibase + pl.bb, h->bidx_b.data(), h->packed_b * 4, hipMemcpyHostToDevice, s));
    SQ_CHECK_HIP(hipMemcpyAsync(ibase + pl.rm, h->rm_pool.data(), h->rm_pool.size() * 4, hipMemcpyHostToDevice, s));
    SQ_CHECK_HIP(hipStreamSynchronize(s));
    h->plan_uploaded_ptr = packed;
  }
  if (h->padded) {  // the padded copy of the flat parameters: zeros + the caller's values in their padded places
    sq_zero_fill(base + pl.fi, h->n_params, s);
    SQ_LAUNCH(k_flat_scatter, dim3((unsigned)std::min<int64_t>(2048, (h->n_uparams + 255) / 256)), dim3(256), 0, s, flat, base + pl.fi,
              (const int*)(ibase + pl.um), h->n_uparams);
    flat = base + pl.fi;
  }
  sq_launch_pack(flat, base + pl.w, ibase + pl.wi, h->packed_w, s);
  sq_launch_pack_bias(flat, base + pl.b, ibase + pl.ba, ibase + pl.bb, h->packed_b, s);
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// workspace carve
// ------------------------------------------------------------------------------------------------
static bool sq_chain_on(const SqairHandle* h, int T, int B);
Workspace sq_carve(const SqairHandle* h, int T, int B, float* base, bool train) {
  const SqairConfig& c = h->cfg;
  const int64_t nh = c.n_hidden, N = c.n_steps_per_image, R = (int64_t)B * c.k_particles, M = R * N;
  const int64_t G2 = c.glimpse_size * c.glimpse_size;
  const int64_t pre_ld = h->layers[L_PRE].nt * 16;
  Workspace w;
  memset(&w, 0, sizeof(w));
  w.train = train; w.chain = sq_chain_on(h, T, B) && h->opt_slot_chain_mode == 1; w.tape = train || sq_chain_on(h, T, B); w.T = T; w.B = B; w.R = (int)R; w.M = (int)M; w.N = (int)N; w.nh = (int)nh;
  const int64_t snh = c.time_cell == CELL_LSTM ? 2 * nh : nh;  // temporal state of a slot: [hidden | cell] for the LSTM
  w.snh = (int)snh;
  const int64_t psnh = c.prior_cell == CELL_LSTM ? 2 * nh : nh;
  w.psnh = (int)psnh;
  int64_t o = 0;
  auto take = [&](int64_t n) {
    float* p = base ? base + o : nullptr;
    o += align64(n);
    return p;
  };
  const bool tape = w.tape;
  const int64_t F = tape ? T : 1;          // per-frame multiplicity
  const int64_t S = tape ? 2 * T * N : 1;  // per-slot multiplicity (x R rows)
  w.ienc_a = take((int64_t)T * B * nh);
  w.ienc_b = take((int64_t)T * B * nh);
  const int64_t rw = sq_rnn_width(c);  // slot-RNN pre-activation width (LSTM: the four gates)
  w.pre_disc = take((int64_t)T * B * rw);
  w.rec_m_all = take((int64_t)(T + 1) * M * rec::W);
  w.temporal_m = take((tape ? T + 1 : 2) * M * snh);
  w.prior_m = take((tape ? T + 1 : 2) * M * psnh);
  w.last_id[0] = take(R);
  w.last_id[1] = take(R);
  w.rec_p_all = take((int64_t)T * M * rec::W);
  w.rec_d_all = take((int64_t)T * M * rec::W);
  w.zero_rec = take(rec::W);
  w.disc_init_rec = take(rec::W);
  w.prop_rnn_init = take(2 * nh);   // [hidden | cell] with an LSTM slot RNN
  w.disc_rnn_init = take(2 * nh);
  w.rn_init_state = take(4);
  w.w3_prop = take(nh * 8 + 8);
  w.w3_disc = take(nh * 8 + 8);
  w.temporal_p = take(F * M * snh);
  w.prior_p = take(F * M * psnh);
  w.pgz = take(F * M * ((c.prior_cell == CELL_LSTM) ? 4 * nh : nh));  // GRU: z gate; LSTM: the four gate pre-activations
  w.pgr = take(F * M * nh);
  w.pghc = take(F * M * nh);
  w.pgrh = take(M * nh);
  w.pgxh = take(M * nh);
  w.pstats = take((int64_t)T * M * PS_LD);
  w.spre = take((int64_t)T * R * 128);
  w.hid1 = take(F * M * 256);
  w.wb = take(F * M * WB_LD);
  w.mask = take(F * M * G2);
  w.g1 = take(F * M * G2);
  w.pea = take(F * M * nh);
  w.peb = take(F * M * nh);
  w.m1 = take(F * M * M1_LD);
  w.pre = take(M * pre_ld);
  w.lea = take(F * M * nh);
  w.leb = take(F * M * nh);
  w.c = take(F * R * nh);
  w.pre_d = take(R * rw);
  w.r = take((tape ? S : 2) * R * nh);
  w.rc = take(c.rnn_cell == RNN_LSTM ? (tape ? S : 2) * R * nh : 64);       // LSTM slot RNN: cell states, laid out like r
  w.rgates = take(c.rnn_cell != RNN_VANILLA ? (tape ? S : 1) * R * rw : 64);  // kept: LSTM gate pre-activations / GRU [z | r | candidate]
  w.t1 = take(S * R * T1_LD);
  w.t2 = take(S * R * nh);
  w.tp = take(S * R * TP_LD);
  w.g2 = take(S * R * G2);
  w.e1 = take(S * R * nh);
  w.e2 = take(S * R * nh);
  w.enc = take(S * R * ENC_LD);
  w.hraw = take(S * R * HRAW_LD);
  w.s1h = take(S * R * S1_LD);
  w.gz = take(S * R * nh);
  w.gr = take(S * R * nh);
  w.ghc = take(S * R * nh);
  w.grh = take((w.chain ? S : 1) * R * nh);   // (the chain hands every slot's GRU internals over in its own buffer)
  w.gxh = take((w.chain ? S : 1) * R * nh);
  // LSTM: recurrent gate pre-activations of all slots of a frame, and the kept gate pre-activations per slot
  w.lpre = take(c.time_cell == CELL_LSTM ? M * 4 * nh : 64);
  w.lgates = take(c.time_cell == CELL_LSTM ? (tape ? (int64_t)T * N : 1) * R * 4 * nh : 64);
  w.src = (int*)take(train ? (int64_t)T * M : 64);
  w.qz = take((int64_t)T * R);
  w.pz = take((int64_t)T * R);
  w.dlp = take((int64_t)T * R);
  w.dll = take((int64_t)T * R);
  w.glimpse = take((int64_t)T * M * G2);
  w.dec_a = take((int64_t)T * M * nh);
  w.dec_b = take((int64_t)T * M * nh);
  w.gen = take(c.sample_from_prior ? (int64_t)T * M * gen::W : 64);
  {
    const int64_t P_ = (int64_t)c.img_h * c.img_w, P4 = (P_ + 3) / 4 * 4;
    w.obs_p = take(P4 != P_ ? (int64_t)T * B * P4 + 16 : 64);   // (+16: the last K chunk of the input encoder may read past the row)
  }
  w.prof_ts = (unsigned long long*)take(5 * PROF_MAX * 2);
  w.chain_ctl = (unsigned*)take(w.chain ? (int64_t)SQ_CHAIN_MAX_LAUNCHES * SQ_CHAIN_CTL_WORDS : 64);
  w.total = o;
  return w;
}
static Workspace carve(const SqairHandle* h, int T, int B, float* base) { return sq_carve(h, T, B, base, false); }

extern "C" int64_t sqair_workspace_bytes(const SqairHandle* h, int T, int B) {
  if (!h || T < 1 || B < 1) return -1;
  return sq_carve(h, T, B, nullptr, false).total * 4;
}
extern "C" int64_t sqair_train_workspace_bytes(const SqairHandle* h, int T, int B) {
  if (!h || T < 1 || B < 1) return -1;
  return sq_carve(h, T, B, nullptr, true).total * 4;
}

// ------------------------------------------------------------------------------------------------
// the launch sequence
// ------------------------------------------------------------------------------------------------
int sq_run(SqairHandle* h, Lin& l, LayerId id, int M, const float* packed, hipStream_t s) {
  const PackedLayer& L = h->layers[id];
  const PackedLayout pl = packed_layout(h);
  if (l.a.nseg != (int)L.seg_width.size()) {
    sq_set_error(h, "internal: segment count mismatch in layer " + std::to_string((int)id));
    return -3;
  }
  for (int i = 0; i < l.a.nseg; ++i)
    if (l.a.seg[i].width != L.seg_width[i]) {
      sq_set_error(h, "internal: segment width mismatch in layer " + std::to_string((int)id));
      return -3;
    }
  l.a.wp = packed + pl.w + L.w_off;
  l.a.wzero = packed + pl.w;
  l.a.bias = packed + pl.b + L.b_off;
  l.a.M = M;
  l.a.N = L.N;
  if (h->dense_log_on) { const int e[4] = {(int)id, M, L.kc * 16, L.N}; h->dense_log.insert(h->dense_log.end(), e, e + 4); }
  if (sq_chain_active(h)) return sq_chain_add_dense(h, l.a, L.kc, L.nt);   // collected into the chain launch (sqair_chain.h)
  const int rc = sq_launch_linear(l.a, L, s);
  if (rc != 0) sq_set_error(h, "internal: A-operand contract (16-byte aligned, ld % 4 == 0) violated in layer " + std::to_string((int)id));
  return rc;
}

// The tail of slot k fused into slot k + 1's VanillaRNN layer (k_rnn_tail, sqair_glue.hip): possible when the layer is exactly
// [z-record (56 -> 4 chunks) | hidden state (nh)] -> nh with nh in {128, 256}.
static bool can_fuse_tail(const SqairHandle* h, LayerId id) {
  const PackedLayer& L = h->layers[id];
  const int nh = h->cfg.n_hidden;
  return h->cfg.rnn_cell == RNN_VANILLA && (nh == 128 || nh == 256) && L.seg_width.size() == 2 && L.seg_width[0] == rec::ZW &&
         L.seg_width[1] == nh && L.kc == 4 + nh / 16 && L.nt == nh / 16 && L.N == nh && h->opt_tail_fusion;
}
static int run_rnn_tail(SqairHandle* h, const TailArgs& ta, Dims d, LayerId id, const float* hid, int hid_ld, const float* add, int add_ld,
                        float* out, int out_ld, const float* packed, hipStream_t s) {
  const PackedLayer& L = h->layers[id];
  const PackedLayout pl = packed_layout(h);
  if (h->dense_log_on) { const int e[4] = {-1 - (int)id, d.R, L.kc * 16, L.N}; h->dense_log.insert(h->dense_log.end(), e, e + 4); }   // (id < 0: with the tail in front)
  if (sq_chain_active(h)) {
    ChainRnn r; memset(&r, 0, sizeof(r));
    r.ta = ta; r.hid = hid; r.hid_ld = hid_ld; r.wp = packed + pl.w + L.w_off; r.bias = packed + pl.b + L.b_off; r.add = add; r.add_ld = add_ld;
    r.out = out; r.out_ld = out_ld; r.n_out = L.N;
    return sq_chain_add_rnn_tail(h, r);
  }
  const int rc = sq_launch_rnn_tail(ta, d, hid, hid_ld, packed + pl.w + L.w_off, packed + pl.b + L.b_off, add, add_ld, out, out_ld, L.N, s, nullptr);
  if (rc != 0) sq_set_error(h, "internal: k_rnn_tail launch rejected");
  return rc;
}

// A slot's what sample in the epilogue of the layer that produces its operands (WhatArgs, sqair_glue.h): the glimpse encoder's
// Gaussian head of a discovery slot (mode 0, pack L_WHAT_HEAD_I) / the temporal cell's heads of a propagation slot (mode 1, pack
// L_PROP_HEADS_I).  Inference passes of the product build on the launch path; the slot's tail then runs with `what_done`.
static bool sq_what_fusion(const SqairHandle* h, bool train, bool chain) {
#ifdef SQAIR_WIDE
  (void)h; (void)train; (void)chain;
  return false;
#else
  // (not in training passes: the layer would also have to write its own output to the tape for the adjoint -- scattered 4-byte stores
  //  in the reference column order -- and the step was slower with it, 7.38 against 7.35 ms)
  return h->opt_what_fusion && !train && !chain && (h->cfg.n_hidden == 128 || h->cfg.n_hidden == 256);
#endif
}
static int run_what(SqairHandle* h, int mode, const float* x, int x_ld, int M, int slot, const float* noise, const float* enc, int enc_ld,
                    const float* rec_prev, float* rec_new, const float* packed, hipStream_t s) {
  const LayerId id = mode == 0 ? L_WHAT_HEAD_I : L_PROP_HEADS_I;
  const PackedLayer& L = h->layers[id];
  const PackedLayout pl = packed_layout(h);
  WhatArgs a; memset(&a, 0, sizeof(a));
  a.mode = mode; a.x = x; a.x_ld = x_ld; a.wp = packed + pl.w + L.w_off; a.wzero = packed + pl.w; a.bias = packed + pl.b + L.b_off;
  a.M = M; a.kc = L.kc; a.n_tiles = L.nt; a.nw = h->cfg.n_what; a.N = h->cfg.n_steps_per_image; a.slot = slot; a.nzw = 4 + h->cfg.n_what + 1;
  a.noise = noise; a.enc = enc; a.enc_ld = enc_ld; a.rec_prev = rec_prev; a.rec_new = rec_new; a.layer_id = (int)id;
  if (h->dense_log_on) { const int e[4] = {(int)id, M, L.kc * 16, L.N}; h->dense_log.insert(h->dense_log.end(), e, e + 4); }
  const int rc = sq_launch_linear_what(a, s);
  if (rc != 0) sq_set_error(h, "internal: k_linear_what launch rejected (layer " + std::to_string((int)id) + ")");
  return rc;
}

// three dependent slot layers: one launch each (a single multi-layer launch with in-launch hand-offs was built and
// measured slower twice in round 1 -- tools/xcd_team.hip, DESIGN.md section 8 -- and left the library)
#define RUN_CHAIN3(l0, id0, l1, id1, l2, id2, M) \
  do {                                            \
    RUN(l0, id0, M);                              \
    RUN(l1, id1, M);                              \
    RUN(l2, id2, M);                              \
  } while (0)

static int emit_crop(SqairHandle* h, const CropArgs& ca, POff po, Dims d, int nslots, hipStream_t s) {
  if (sq_chain_active(h)) return sq_chain_add_crop(h, ca);
  return sq_launch_crop(ca, po, d, nslots, s);
}
static int emit_tail(SqairHandle* h, const TailArgs& ta, Dims d, hipStream_t s) {
  if (sq_chain_active(h)) return sq_chain_add_tail(h, ta);
  return sq_launch_slot_tail(ta, d, s);
}
// The in-launch slot chain (sqair_chain.h) serves the shipped cell configuration -- VanillaRNN slot cell with the tail fused into
// the next slot's layer, GRU temporal cell -- at the row counts where the pass is latency-bound; everything else keeps one launch
// per op.  The workspace is then laid out like the training tape (every slot's activations in their own buffer).
static bool sq_chain_on(const SqairHandle* h, int T, int B) {
#ifdef SQAIR_WIDE
  (void)h; (void)T; (void)B;
  return false;
#else
  const SqairConfig& c = h->cfg;
  return h->opt_slot_chain && c.rnn_cell == RNN_VANILLA && c.time_cell == CELL_GRU && !c.sample_from_prior &&
         c.n_hidden == 256 && can_fuse_tail(h, L_PROP_RNN) && can_fuse_tail(h, L_DISC_RNN) && B * c.k_particles <= 320 && 2 * T <= SQ_CHAIN_MAX_LAUNCHES;
#endif
}
static int emit_latsum(SqairHandle* h, const float* f, const float* rec_p, float* c, Dims d, hipStream_t s) { (void)h; return sq_launch_latent_sum(f, rec_p, c, d, s); }
static int emit_compact(SqairHandle* h, const CompactArgs& ka, POff po, Dims d, hipStream_t s) { (void)h; return sq_launch_compact(ka, po, d, s); }

// rows of n floats -> rows of pitch p4 >= n, zero padded
__global__ void k_pad_rows(const float* __restrict__ src, float* __restrict__ dst, int n, int p4 SQ_TLP) {
  SQ_TL_SCOPE;
  const size_t r = blockIdx.x;
  for (int i = threadIdx.x; i < p4; i += blockDim.x) dst[r * p4 + i] = i < n ? src[r * n + i] : 0.0f;
}
// row r: nseg segments of `padw` floats -> nseg segments of their first `truew` floats, packed
__global__ void k_copy_cols(const float* __restrict__ src, float* __restrict__ dst, int nseg, int truew, int padw SQ_TLP) {
  SQ_TL_SCOPE;
  const size_t r = blockIdx.x;
  for (int i = threadIdx.x; i < nseg * truew; i += blockDim.x) {
    const int sg = i / truew, c = i - sg * truew;
    dst[r * nseg * truew + i] = src[r * nseg * padw + sg * padw + c];
  }
}

// parts: 1 = prologue (workspace clear, initial state, input encoder), 2 = the frame loop, 4 = epilogue (log-probabilities,
// decoder, final state copies).
int sq_forward_impl(SqairHandle* h, const float* flat, const float* packed, const float* obs, const float* noise,
                    int T, int B, int t_offset, const SqairOutputs* outp, float* wsbase, int64_t ws_bytes,
                    hipStream_t s, bool train, int parts) {
  const SqairConfig& c = h->cfg;
  if (!flat || !packed || !obs || !noise || !outp || !wsbase || T < 1 || B < 1) {
    sq_set_error(h, "sqair_forward: null argument or bad T/B");
    return -1;
  }
  if (ws_bytes < sq_carve(h, T, B, nullptr, train).total * 4) {
    sq_set_error(h, "sqair_forward: workspace too small");
    return -1;
  }
  if (c.sample_from_prior && (train || h->gen_noise == nullptr)) {
    sq_set_error(h, "sample_from_prior: inference through sqair_forward / sqair_graph_capture only, after sqair_set_generation_noise");
    return -1;
  }
  flat = sq_flat(h, flat, packed);   // (padded configurations: the copy sqair_pack_params keeps in the packed buffer)
  sq_chain_reset(h);                 // (a pass that ended early between sq_chain_begin and sq_chain_flush left the recorder open)
  const SqairOutputs out = *outp;
  const int nh = c.n_hidden, nw = c.n_what, N = c.n_steps_per_image, K = c.k_particles;
  const int R = B * K, M = R * N, G2 = c.glimpse_size * c.glimpse_size, P_ = c.img_h * c.img_w;
  const int nzw = 4 + nw + 1;
  Dims d = make_dims(c, B);
  const POff po = h->po;
  const Workspace w = sq_carve(h, T, B, wsbase, train);
  // Frames are GEMM A operands (the input encoder) and 16-byte staging units (crop adjoint): their rows must start 16-byte
  // aligned.  H * W not a multiple of 4 (the reference takes any frame size): the pass works on a zero-padded copy with pitch
  // P4 = H * W rounded up to 4, made once in the prologue; otherwise on the caller's buffer as it is.
  const float* const obs_user = obs;
  if ((P_ & 3) != 0) { d.P4 = (P_ + 3) / 4 * 4; obs = w.obs_p; }
  const int PL = d.P4;
  const int pre_ld = h->layers[L_PRE].nt * 16;
  const int RW = rec::W, snh = d.snh, psnh = d.psnh;
  const int rw = sq_rnn_width(c);  // slot-RNN pre-activation width; pre columns: [rnn rw | T1 nh | S1 nh/2 | GRU z, r]
  const PackedLayout pl = packed_layout(h);

  // ---- sequence prologue -----------------------------------------------------------------------
  // The GEMM A-operand contract wants every float it may touch to be finite (padding meets zero weights, but
  // 0 * NaN = NaN) and slot records are read as 56-wide segments before all their fields are written in a frame:
  // clear the caller's (garbage) workspace once per pass
  if (parts & 1) {
    // (clear_each_pass = false: the caller cleared this workspace once with sqair_clear_workspace and reuses it with the
    //  same T and B -- every buffer is then either rewritten by the pass or holds finite values / zeros it never overwrites)
    if (h->clear_each_pass) sq_zero_fill(wsbase, (int64_t)((float*)w.prof_ts - wsbase), s);
    // initial state; discovery starts every frame with presence = 1 (core.py:150) -> disc_init_rec
    sq_launch_init_state(w.rec_m_all, w.state(w.temporal_m, 0, w.snh), w.state(w.prior_m, 0, w.psnh), w.last_id[0], w.disc_init_rec,
                         w.prop_rnn_init, w.disc_rnn_init, w.rn_init_state, w.w3_prop, w.w3_disc,
                         (int)P(h, "prop.transform.l2.w"), (int)P(h, "disc.transform.l2.w"), flat, po, d, s);
    if (w.chain) {
      // hand-off words of the chain launches of this pass -> sentinel; their control blocks -> zero (one launch)
      ChainPoisonList pl; memset(&pl, 0, sizeof(pl));
      const int64_t SR = (int64_t)T * R * N;   // rows of one phase of a slot buffer
      auto slotbuf = [&](float* base, int ld, int width, bool disc_too) {
        pl.r[pl.n++] = ChainPoison{base, disc_too ? 2 * SR : SR, ld, width, 0};
      };
      slotbuf(w.r, nh, nh, true);
      slotbuf(w.t1, T1_LD, h->layers[L_PROP_T1].N, true);
      slotbuf(w.t2, nh, nh, true);
      slotbuf(w.g2, G2, G2, true);
      slotbuf(w.e1, nh, nh, true);
      slotbuf(w.e2, nh, nh, true);
      slotbuf(w.enc, ENC_LD, 2 * nw, true);
      slotbuf(w.hraw, HRAW_LD, h->layers[L_PROP_HEADS].N, false);
      slotbuf(w.gz, nh, nh, false);
      slotbuf(w.grh, nh, nh, false);
      slotbuf(w.gxh, nh, nh, false);
      pl.r[pl.n++] = ChainPoison{w.temporal_p, (int64_t)T * M, w.snh, nh, 0};
      pl.r[pl.n++] = ChainPoison{w.rec_p_all + rec::WHERE, (int64_t)T * M, rec::W, 4, 0};
      pl.r[pl.n++] = ChainPoison{w.rec_d_all + rec::WHERE, (int64_t)T * M, rec::W, 4, 0};
      pl.r[pl.n++] = ChainPoison{w.rec_d_all + rec::PRES, (int64_t)T * M, rec::W, 1, 0};
      if (h->layers[L_DISC_T1].N != h->layers[L_PROP_T1].N) { sq_set_error(h, "slot chain: T1 widths differ"); return -3; }
      sq_chain_poison(pl, w.chain_ctl, SQ_CHAIN_MAX_LAUNCHES * SQ_CHAIN_CTL_WORDS, s);
    }
    if (obs != obs_user) SQ_LAUNCH(k_pad_rows, dim3(T * B), dim3(256), 0, s, obs_user, w.obs_p, P_, PL);
    // input encoder for every frame of every sequence at once (core.py:165, modules.py:100-112)
    Lin a; a.seg(obs, PL, P_).out(w.ienc_a, nh).act(ACT_ELU); RUN(a, L_IENC0, T * B);
    Lin b; b.seg(w.ienc_a, nh, nh).out(w.ienc_b, nh).act(ACT_ELU); RUN(b, L_IENC1, T * B);
    Lin p; p.seg(w.ienc_b, nh, nh).out(w.pre_disc, rw); RUN(p, L_PREDISC, T * B);
  }

  // (the fused launch recomputes a slot's tail in each of the layer's nh / 16 column-tile workgroups: free while the pass is
  // latency-bound (-0.2 ms at 160 rows), even at 320 rows, a loss from 640 on -- 20.5 us against 5.9 + 5.5 us at 1280 rows)
#ifndef SQAIR_TAIL_FUSION_ROWS_DEFAULT
#define SQAIR_TAIL_FUSION_ROWS_DEFAULT 560
#endif
  static const int tail_rows = SQ_KNOB_INT("SQAIR_TAIL_FUSION_ROWS", SQAIR_TAIL_FUSION_ROWS_DEFAULT);
  const bool fuse_prop = d.R <= tail_rows && can_fuse_tail(h, L_PROP_RNN), fuse_disc = d.R <= tail_rows && can_fuse_tail(h, L_DISC_RNN);
  if (w.chain && !(fuse_prop && fuse_disc)) { sq_set_error(h, "slot chain: tail fusion off"); return -3; }
  const bool fw = sq_what_fusion(h, train, w.chain);   // the what sample in the producing layer's epilogue (sqair_glue.h: WhatArgs)
  TailArgs pending_tail; memset(&pending_tail, 0, sizeof(pending_tail));
  for (int t = 0; (parts & 2) && t < T; ++t) {
    const int pp = t & 1, pn = pp ^ 1;
    const float* img = obs + (size_t)t * B * PL;
    const float* nz = noise + (size_t)t * R * 2 * N * nzw;
    const float* rec_prev = w.rec_m_all + (size_t)t * M * RW;
    float* rec_next = w.rec_m_all + (size_t)(t + 1) * M * RW;
    float* rec_p_t = w.rec_p_all + (size_t)t * M * RW;
    float* rec_d_t = w.rec_d_all + (size_t)t * M * RW;
    float* pstats_t = w.pstats + (size_t)t * M * PS_LD;
    float* spre_t = w.spre + (size_t)t * R * 128;
    const float* temporal_prev = w.state(w.temporal_m, t, w.snh);
    const float* prior_prev = w.state(w.prior_m, t, w.psnh);
    float* temporal_p = w.frame(w.temporal_p, (int64_t)M * snh, t);
    const float* tau_prev = temporal_prev + d.toff;  // what the slot networks read as "temporal state" (core.py:284): the
                                                     // GRU state / the CELL half of an LSTM state, row stride snh
    float* prior_p = w.frame(w.prior_p, (int64_t)M * psnh, t);
    float* hid1 = w.frame(w.hid1, (int64_t)M * 256, t);
    float* wb = w.frame(w.wb, (int64_t)M * WB_LD, t);
    float* mask = w.frame(w.mask, (int64_t)M * G2, t);
    float* g1 = w.frame(w.g1, (int64_t)M * G2, t);
    float* pea = w.frame(w.pea, (int64_t)M * nh, t);
    float* peb = w.frame(w.peb, (int64_t)M * nh, t);
    float* m1 = w.frame(w.m1, (int64_t)M * M1_LD, t);
    float* lea = w.frame(w.lea, (int64_t)M * nh, t);
    float* leb = w.frame(w.leb, (int64_t)M * nh, t);
    float* cvec = w.frame(w.c, (int64_t)R * nh, t);

    // ---- A. propagation prior (propagate.py:68-98): GRU over [what, where]_{t-1}, all slots ----
    if (c.prior_cell == CELL_LSTM) {
      float* pg = w.frame(w.pgz, (int64_t)M * 4 * nh, t);
      Lin g; g.seg(rec_prev, RW, rec::ZW).seg(prior_prev, psnh, nh).out(pg, 4 * nh); RUN(g, L_PRIOR_GRU1, M);
      sq_launch_lstm_cell(pg, 4 * nh, prior_prev + nh, psnh, prior_p, psnh, M, nh, s);
      Lin pll; pll.seg(prior_p, psnh, nh).out(pstats_t, PS_LD); RUN(pll, L_PRIOR_LIN, M);
    } else if (c.prior_cell == CELL_VANILLA) {
      Lin g; g.seg(rec_prev, RW, rec::ZW).seg(prior_prev, nh, nh).out(prior_p, nh).act(ACT_TANH); RUN(g, L_PRIOR_GRU1, M);
      Lin pll; pll.seg(prior_p, nh, nh).out(pstats_t, PS_LD); RUN(pll, L_PRIOR_LIN, M);
    } else {
      float* pgz = w.frame(w.pgz, (int64_t)M * nh, t);
      Lin g1l; g1l.seg(rec_prev, RW, rec::ZW).seg(prior_prev, nh, nh).out(pgz, nh)
                 .gru1(prior_prev, nh, w.pgrh, nh, w.pgxh, nh, nh);
      if (train) { g1l.a.o3 = w.frame(w.pgr, (int64_t)M * nh, t); g1l.a.o3_ld = nh; }
      RUN(g1l, L_PRIOR_GRU1, M);
      Lin g2l; g2l.seg(w.pgrh, nh, nh).add(w.pgxh, nh, nh).out(prior_p, nh).gru2(prior_prev, nh, pgz, nh, nh);
      if (train) { g2l.a.o1 = w.frame(w.pghc, (int64_t)M * nh, t); g2l.a.o1_ld = nh; }
      RUN(g2l, L_PRIOR_GRU2, M);
      Lin pll; pll.seg(prior_p, nh, nh).out(pstats_t, PS_LD); RUN(pll, L_PRIOR_LIN, M);
    }
    // ---- B. where-bias MLP and glimpse-mask MLP of every slot (core.py:292, modules.py:350-356) ----
    {
      Lin a; a.seg(tau_prev, snh, nh).out(hid1, 256).act(ACT_ELU); RUN(a, L_TAU1, M);
      Lin b; b.seg(hid1, 256, 128).out(wb, WB_LD); RUN(b, L_WB2, M);
      Lin m; m.seg(hid1 + 128, 256, 128).out(mask, G2).act(ACT_SIGMOID); RUN(m, L_MASK2, M);
    }
    // ---- C. crop #1 at where_{t-1} + bias, masked, encoded -> loc1 (core.py:293-294) ----
    {
      CropArgs ca; memset(&ca, 0, sizeof(ca));
      ca.mode = CROP_PROP1; ca.img = img; ca.mask = c.masked_glimpse ? mask : nullptr; ca.mask_row_mul = N;
      ca.out = g1; ca.out_row_mul = N; ca.rec_prev = rec_prev; ca.wb = wb; ca.wb_ld = WB_LD; ca.flat = flat;
      emit_crop(h, ca, po, d, N, s);
      Lin a; a.seg(g1, G2, G2).out(pea, nh).act(ACT_ELU); RUN(a, L_GENC0, M);
      Lin b; b.seg(pea, nh, nh).out(peb, nh).act(ACT_ELU); RUN(b, L_GENC1, M);
      Lin l; l.seg(peb, nh, nh).out(m1, M1_LD); RUN(l, L_WHAT_LOC, M);
    }
    // ---- D. loop-invariant pre-activations of all slots ----
    {
      Lin p; p.seg(m1, M1_LD, nw).seg(rec_prev, RW, rec::ZW).seg(tau_prev, snh, nh).out(w.pre, pre_ld);
      RUN(p, L_PRE, M);
      if (c.time_cell == CELL_LSTM) {  // recurrent rows of the LSTM gates: h_{t-1} W_h + b of every slot
        Lin q; q.seg(temporal_prev, snh, nh).out(w.lpre, 4 * nh); RUN(q, L_PROP_GRU2, M);
      }
    }
    // ---- E. propagation slots (propagate.py:168-184 static_rnn over PropagationCore) ----
    if (w.chain) sq_chain_begin(h, d, po, wsbase, ws_bytes);
    for (int k = 0; k < N; ++k) {
      const float* pre_k = w.pre + (size_t)k * pre_ld;
      const int pre_rld = N * pre_ld;
      float* r_k = w.rslot(t, 0, k);
      const int rl = w.sld(nh), t1l = w.sld(T1_LD), gl2 = w.sld(G2), el = w.sld(ENC_LD), hl = w.sld(HRAW_LD);
      float* t1 = w.slot(w.t1, T1_LD, t, 0, k);
      float* t2 = w.slot(w.t2, nh, t, 0, k);
      float* g2 = w.slot(w.g2, G2, t, 0, k);
      float* e1 = w.slot(w.e1, nh, t, 0, k);
      float* e2 = w.slot(w.e2, nh, t, 0, k);
      float* enc = w.slot(w.enc, ENC_LD, t, 0, k);
      float* hraw = w.slot(w.hraw, HRAW_LD, t, 0, k);
      float* gz = w.slot(w.gz, nh, t, 0, k);
      {
        Lin a;
        if (k == 0) a.seg(w.zero_rec, 0, rec::ZW).seg(w.prop_rnn_init, 0, nh);
        else a.seg(rec_p_t + (size_t)(k - 1) * RW, N * RW, rec::ZW).seg(w.rslot(t, 0, k - 1), rl, nh);
        if (c.rnn_cell == RNN_LSTM) {
          float* gates = train ? w.slot(w.rgates, 4 * nh, t, 0, k) : w.rgates;
          a.add(pre_k, pre_rld, rw).out(gates, w.sld(4 * nh));
          RUN(a, L_PROP_RNN, R);
          sq_launch_lstm_cell2(gates, w.sld(4 * nh), k == 0 ? w.prop_rnn_init + nh : w.cslot(t, 0, k - 1), k == 0 ? 0 : rl, r_k, rl,
                               w.cslot(t, 0, k), rl, R, nh, s);
        } else if (c.rnn_cell == RNN_GRU) {  // snt.GRU in two launches, [z | r | tanh candidate] kept for the backward pass
          float* g3 = train ? w.slot(w.rgates, 3 * nh, t, 0, k) : w.rgates;
          const int g3l = w.sld(3 * nh);
          const float* hp = k == 0 ? w.prop_rnn_init : w.rslot(t, 0, k - 1);
          const int hpl = k == 0 ? 0 : rl;
          a.add(pre_k, pre_rld, rw).out(g3, g3l).gru1(hp, hpl, w.grh, nh, w.gxh, nh, nh);
          a.a.o3 = g3 + nh; a.a.o3_ld = g3l;
          RUN(a, L_PROP_RNN, R);
          Lin b2; b2.seg(w.grh, nh, nh).add(w.gxh, nh, nh).out(r_k, rl).gru2(hp, hpl, g3, g3l, nh);
          b2.a.o1 = g3 + 2 * nh; b2.a.o1_ld = g3l;
          RUN(b2, L_PROP_RNN2, R);
        } else if (k > 0 && fuse_prop) {  // the previous slot's tail rides in this launch
          const int rc = run_rnn_tail(h, pending_tail, d, L_PROP_RNN, w.rslot(t, 0, k - 1), rl, pre_k, pre_rld, r_k, rl, packed, s);
          if (rc != 0) return rc;
        } else {
          a.add(pre_k, pre_rld, nh).out(r_k, rl).act(ACT_TANH);
          RUN(a, L_PROP_RNN, R);
        }
      }
      {
        // T1 columns [transform hidden 1 (ELU) | steps-predictor hidden pre-activation without `what` (linear)]
        Lin a; a.seg(r_k, rl, nh).add(pre_k + rw, pre_rld, nh + nh / 2).out(t1, t1l).act2(ACT_ELU, ACT_NONE, nh);
        RUN(a, L_PROP_T1, R);
        Lin b; b.seg(t1, t1l, nh).out(t2, rl).act(ACT_ELU); RUN(b, L_PROP_T2, R);
      }
      {
        CropArgs ca; memset(&ca, 0, sizeof(ca));
        ca.mode = CROP_PROP2; ca.img = img; ca.mask = c.masked_glimpse ? mask : nullptr; ca.mask_row_mul = N;
        ca.mask_row_add = k; ca.out = g2; ca.out_row_mul = w.tape ? N : 1; ca.rec_prev = rec_prev; ca.rec_new = rec_p_t;
        ca.t2 = t2; ca.t2_ld = rl; ca.w3 = w.w3_prop; ca.noise = nz; ca.flat = flat; ca.slot = k;
        if (train) { ca.tp_out = w.slot(w.tp, TP_LD, t, 0, k); ca.tp_out_ld = w.sld(TP_LD); }
        emit_crop(h, ca, po, d, 1, s);
      }
      {
        Lin a; a.seg(g2, gl2, G2).out(e1, rl).act(ACT_ELU);
        Lin b; b.seg(e1, rl, nh).out(e2, rl).act(ACT_ELU);
        Lin e; e.seg(e2, rl, nh).out(enc, el).act2(ACT_NONE, ACT_SOFTPLUS_MIN, nw);
        RUN_CHAIN3(a, L_GENC0, b, L_GENC1, e, L_WHAT_HEAD, R);
      }
      if (c.time_cell == CELL_LSTM) {
        float* gates = train ? w.lgates + ((size_t)t * M + k) * 4 * nh : w.lgates;
        const int gld = train ? N * 4 * nh : 4 * nh;
        Lin gl; gl.seg(r_k, rl, nh).seg(rec_p_t + (size_t)k * RW + rec::WHERE, N * RW, 4).seg(enc, el, 2 * nw)
                  .add(w.lpre + (size_t)k * 4 * nh, N * 4 * nh, 4 * nh).out(gates, gld);
        RUN(gl, L_PROP_GRU1, R);
        sq_launch_lstm_cell(gates, gld, tau_prev + (size_t)k * snh, N * snh, temporal_p + (size_t)k * snh, N * snh, R, nh, s);
        if (fw) { const int rc = run_what(h, 1, temporal_p + (size_t)k * snh, N * snh, R, k, nz, enc, el, rec_prev, rec_p_t, packed, s); if (rc != 0) return rc; }
        else { Lin hd; hd.seg(temporal_p + (size_t)k * snh, N * snh, nh).out(hraw, hl); RUN(hd, L_PROP_HEADS, R); }
      } else if (c.time_cell == CELL_VANILLA) {  // tau' = tanh(x W_i + [tau W_h + b, hoisted into `pre`]) in one launch
        Lin g; g.seg(r_k, rl, nh).seg(rec_p_t + (size_t)k * RW + rec::WHERE, N * RW, 4).seg(enc, el, 2 * nw)
                 .add(pre_k + rw + nh + nh / 2, pre_rld, nh).out(temporal_p + (size_t)k * nh, N * nh).act(ACT_TANH);
        RUN(g, L_PROP_GRU1, R);
        if (fw) { const int rc = run_what(h, 1, temporal_p + (size_t)k * nh, N * nh, R, k, nz, enc, el, rec_prev, rec_p_t, packed, s); if (rc != 0) return rc; }
        else { Lin hd; hd.seg(temporal_p + (size_t)k * nh, N * nh, nh).out(hraw, hl); RUN(hd, L_PROP_HEADS, R); }
      } else {
        const float* tau_k = temporal_prev + (size_t)k * nh;
        float* grh_k = w.chain ? w.slot(w.grh, nh, t, 0, k) : w.grh;   // (the chain: every slot's hand-offs in their own buffers)
        float* gxh_k = w.chain ? w.slot(w.gxh, nh, t, 0, k) : w.gxh;
        const int grl = w.chain ? rl : nh;
        Lin g1l; g1l.seg(r_k, rl, nh).seg(rec_p_t + (size_t)k * RW + rec::WHERE, N * RW, 4).seg(enc, el, 2 * nw)
                   .add(pre_k + rw + nh + nh / 2, pre_rld, 2 * nh).out(gz, rl)
                   .gru1(tau_k, N * nh, grh_k, grl, gxh_k, grl, nh);
        if (train) { g1l.a.o3 = w.slot(w.gr, nh, t, 0, k); g1l.a.o3_ld = rl; }
        RUN(g1l, L_PROP_GRU1, R);
        Lin g2l; g2l.seg(grh_k, grl, nh).add(gxh_k, grl, nh).out(temporal_p + (size_t)k * nh, N * nh)
                   .gru2(tau_k, N * nh, gz, rl, nh);
        if (train) { g2l.a.o1 = w.slot(w.ghc, nh, t, 0, k); g2l.a.o1_ld = rl; }
        RUN(g2l, L_PROP_GRU2, R);
        if (fw) { const int rc = run_what(h, 1, temporal_p + (size_t)k * nh, N * nh, R, k, nz, enc, el, rec_prev, rec_p_t, packed, s); if (rc != 0) return rc; }
        else { Lin hd; hd.seg(temporal_p + (size_t)k * nh, N * nh, nh).out(hraw, hl); RUN(hd, L_PROP_HEADS, R); }
      }
      {
        TailArgs ta; memset(&ta, 0, sizeof(ta));
        ta.is_disc = 0; ta.slot = k; ta.hraw = hraw; ta.h_ld = hl; ta.enc = enc; ta.enc_ld = el;
        ta.rec_prev = rec_prev; ta.rec_new = rec_p_t; ta.noise = nz; ta.s1p = t1 + nh; ta.s1p_ld = t1l;
        ta.wp = packed + pl.w + h->layers[L_PROP_S1].w_off; ta.flat = flat;
        ta.w2_off = po.prop_steps_l1_w; ta.b2_off = po.prop_steps_l1_b;
        if (train) { ta.s1h_out = w.slot(w.s1h, S1_LD, t, 0, k); ta.s1h_ld = w.sld(S1_LD); }
        ta.what_done = fw ? 1 : 0;
        if (fuse_prop && k + 1 < N) pending_tail = ta;  // computed inside the next slot's RNN launch
        else emit_tail(h, ta, d, s);
      }
    }
    if (w.chain) {
      const int rc = sq_chain_flush(h, w.chain_ctl + (size_t)(2 * t) * SQ_CHAIN_CTL_WORDS, 2 * t, s);
      if (rc != 0) return rc;
    }
    // ---- generation modes: prior samples of the propagated objects (sqair_modules.py:294-302) ----
    const bool do_generate = c.sample_from_prior && c.generate_after > 0 && (t_offset + t) > c.generate_after;  // seq.py:198-200
    GenArgs ga; memset(&ga, 0, sizeof(ga));
    if (c.sample_from_prior) {
      ga.rec_p = rec_p_t; ga.rec_d = rec_d_t; ga.rec_prev = rec_prev; ga.pstats = pstats_t; ga.ps_ld = PS_LD; ga.spre = spre_t;
      ga.gen_noise = h->gen_noise + (size_t)t * R * 2 * N * nzw; ga.gen = w.gen + (size_t)t * M * gen::W; ga.flat = flat;
      ga.do_generate = do_generate ? 1 : 0; ga.cfg = c;
      sq_launch_generate_prop(ga, po, d, s);
    }
    // ---- F. summary of propagated latents -> discovery conditioning (sqair_modules.py:368-385, :501) ----
    {
      Lin a; a.seg(rec_p_t, RW, rec::ZW).out(lea, nh).act(ACT_ELU); RUN(a, L_LAT0, M);
      Lin b; b.seg(lea, nh, nh).out(leb, nh).act(ACT_ELU); RUN(b, L_LAT1, M);
      emit_latsum(h, leb, rec_p_t, cvec, d, s);
      Lin p; p.seg(cvec, nh, nh).add(w.pre_disc + (size_t)t * B * rw, rw, rw, K).out(w.pre_d, rw); RUN(p, L_PRED, R);
      if (c.rec_where_prior) {
        Lin q; q.seg(w.rn_init_state, 0, 4).seg(cvec, nh, nh).out(spre_t, 128); RUN(q, L_RNCOND, R);
      }
    }
    // ---- G. discovery steps (sqair_modules.py:129-147 static_rnn over DiscoveryCore) ----
    if (w.chain) sq_chain_begin(h, d, po, wsbase, ws_bytes);
    for (int j = 0; j < N; ++j) {
      float* r_j = w.rslot(t, 1, j);
      const int rl = w.sld(nh), t1l = w.sld(T1_LD), gl2 = w.sld(G2), el = w.sld(ENC_LD);
      float* t1 = w.slot(w.t1, T1_LD, t, 1, j);
      float* t2 = w.slot(w.t2, nh, t, 1, j);
      float* g2 = w.slot(w.g2, G2, t, 1, j);
      float* e1 = w.slot(w.e1, nh, t, 1, j);
      float* e2 = w.slot(w.e2, nh, t, 1, j);
      float* enc = w.slot(w.enc, ENC_LD, t, 1, j);
      {
        Lin a;
        if (j == 0) a.seg(w.disc_init_rec, 0, rec::ZW).seg(w.disc_rnn_init, 0, nh);
        else a.seg(rec_d_t + (size_t)(j - 1) * RW, N * RW, rec::ZW).seg(w.rslot(t, 1, j - 1), rl, nh);
        if (c.rnn_cell == RNN_LSTM) {
          float* gates = train ? w.slot(w.rgates, 4 * nh, t, 1, j) : w.rgates;
          a.add(w.pre_d, rw, rw).out(gates, w.sld(4 * nh));
          RUN(a, L_DISC_RNN, R);
          sq_launch_lstm_cell2(gates, w.sld(4 * nh), j == 0 ? w.disc_rnn_init + nh : w.cslot(t, 1, j - 1), j == 0 ? 0 : rl, r_j, rl,
                               w.cslot(t, 1, j), rl, R, nh, s);
        } else if (c.rnn_cell == RNN_GRU) {
          float* g3 = train ? w.slot(w.rgates, 3 * nh, t, 1, j) : w.rgates;
          const int g3l = w.sld(3 * nh);
          const float* hp = j == 0 ? w.disc_rnn_init : w.rslot(t, 1, j - 1);
          const int hpl = j == 0 ? 0 : rl;
          a.add(w.pre_d, rw, rw).out(g3, g3l).gru1(hp, hpl, w.grh, nh, w.gxh, nh, nh);
          a.a.o3 = g3 + nh; a.a.o3_ld = g3l;
          RUN(a, L_DISC_RNN, R);
          Lin b2; b2.seg(w.grh, nh, nh).add(w.gxh, nh, nh).out(r_j, rl).gru2(hp, hpl, g3, g3l, nh);
          b2.a.o1 = g3 + 2 * nh; b2.a.o1_ld = g3l;
          RUN(b2, L_DISC_RNN2, R);
        } else if (j > 0 && fuse_disc) {
          const int rc = run_rnn_tail(h, pending_tail, d, L_DISC_RNN, w.rslot(t, 1, j - 1), rl, w.pre_d, nh, r_j, rl, packed, s);
          if (rc != 0) return rc;
        } else {
          a.add(w.pre_d, nh, nh).out(r_j, rl).act(ACT_TANH);
          RUN(a, L_DISC_RNN, R);
        }
        Lin b; b.seg(r_j, rl, nh).out(t1, t1l).act2(ACT_ELU, ACT_NONE, nh); RUN(b, L_DISC_T1, R);
        Lin cc; cc.seg(t1, t1l, nh).out(t2, rl).act(ACT_ELU); RUN(cc, L_DISC_T2, R);
      }
      {
        CropArgs ca; memset(&ca, 0, sizeof(ca));
        ca.mode = CROP_DISC; ca.img = img; ca.out = g2; ca.out_row_mul = w.tape ? N : 1; ca.rec_new = rec_d_t; ca.t2 = t2;
        ca.t2_ld = rl; ca.w3 = w.w3_disc; ca.noise = nz; ca.flat = flat; ca.slot = j;
        if (train) { ca.tp_out = w.slot(w.tp, TP_LD, t, 1, j); ca.tp_out_ld = w.sld(TP_LD); }
        emit_crop(h, ca, po, d, 1, s);
        Lin a; a.seg(g2, gl2, G2).out(e1, rl).act(ACT_ELU);
        Lin b; b.seg(e1, rl, nh).out(e2, rl).act(ACT_ELU);
        Lin e; e.seg(e2, rl, nh).out(enc, el).act2(ACT_NONE, ACT_SOFTPLUS_MIN, nw);
        if (fw) {   // the Gaussian head writes the slot's what sample itself (`enc` has no other reader in a discovery slot)
          RUN(a, L_GENC0, R);
          RUN(b, L_GENC1, R);
          const int rc = run_what(h, 0, e2, rl, R, j, nz, nullptr, 0, nullptr, rec_d_t, packed, s);
          if (rc != 0) return rc;

        } else {
          RUN_CHAIN3(a, L_GENC0, b, L_GENC1, e, L_WHAT_HEAD, R);
        }
      }
      {
        TailArgs ta; memset(&ta, 0, sizeof(ta));
        ta.is_disc = 1; ta.slot = j; ta.enc = enc; ta.enc_ld = el; ta.rec_prev = rec_prev; ta.rec_new = rec_d_t;
        ta.noise = nz; ta.s1p = t1 + nh; ta.s1p_ld = t1l;
        ta.wp = packed + pl.w + h->layers[L_DISC_S1].w_off; ta.flat = flat;
        ta.w2_off = po.disc_steps_l1_w; ta.b2_off = po.disc_steps_l1_b;
        if (train) { ta.s1h_out = w.slot(w.s1h, S1_LD, t, 1, j); ta.s1h_ld = w.sld(S1_LD); }
        ta.what_done = fw ? 1 : 0;
        if (fuse_disc && j + 1 < N) pending_tail = ta;
        else emit_tail(h, ta, d, s);
      }
    }
    if (w.chain) {
      const int rc = sq_chain_flush(h, w.chain_ctl + (size_t)(2 * t + 1) * SQ_CHAIN_CTL_WORDS, 2 * t + 1, s);
      if (rc != 0) return rc;
    }
    if (do_generate) sq_launch_generate_disc(ga, po, d, s);  // sqair_modules.py:157-170
    // ---- I. merge / compaction (the log-probabilities H and the decoder J are off the recurrence's critical path:
    //      they run once for all T frames after the loop)
    {
      CompactArgs ka; memset(&ka, 0, sizeof(ka));
      ka.rec_p = rec_p_t; ka.rec_d = rec_d_t; ka.rec_prev = rec_prev; ka.temporal_p = temporal_p;
      ka.prior_p = prior_p; ka.last_id_prev = w.last_id[pp]; ka.last_id_next = w.last_id[pn];
      ka.rec_next = rec_next; ka.temporal_next = w.state(w.temporal_m, t + 1, w.snh); ka.prior_next = w.state(w.prior_m, t + 1, w.psnh);
      ka.flat = flat; ka.t = t; ka.out = out;
      ka.src_out = train ? w.src + (size_t)t * M : nullptr;
      emit_compact(h, ka, po, d, s);
    }
  }
  if (!(parts & 4)) return 0;
  // ---- H. log-probabilities of all T frames in one launch (grid R x T) ----
  {
    LogprobArgs la; memset(&la, 0, sizeof(la));
    la.rec_p = w.rec_p_all; la.rec_d = w.rec_d_all; la.rec_prev = w.rec_m_all; la.pstats = w.pstats; la.ps_ld = PS_LD;
    la.spre = w.spre; la.flat = flat; la.t_global = t_offset; la.t = 0; la.n_frames = T; la.qz = w.qz; la.pz = w.pz;
    la.disc_lp = w.dlp; la.out = out; la.cfg = c; la.gen = c.sample_from_prior ? w.gen : nullptr;
    if (sq_launch_logprob(la, po, d, s) != 0) { sq_set_error(h, "sqair_forward: the log-probability launch failed (dynamic LDS limit)"); return -2; }
  }
  // ---- J. decoder of all T frames as three M = T*B'*N row GEMMs + one insert / log-likelihood launch
  //      (modules.py:435-467, seq.py:271-276) ----
  {
    const int MT = T * M;
    const float* rec_all = w.rec_m_all + (size_t)M * RW;  // merged records of frames 0..T-1
    float* gl = (out.glimpse && !train) ? out.glimpse : w.glimpse;
    Lin a; a.seg(rec_all, RW, rec::ZW).out(w.dec_a, nh).act(ACT_ELU); RUN(a, L_DEC0, MT);
    Lin b; b.seg(w.dec_a, nh, nh).out(w.dec_b, nh).act(ACT_ELU); RUN(b, L_DEC1, MT);
    Lin g; g.seg(w.dec_b, nh, nh).out(gl, G2); g.a.scale_ptr = flat + po.dec_output_scale; RUN(g, L_DEC2, MT);
    if (out.glimpse && train) sq_copy(out.glimpse, gl, (int64_t)MT * G2, s);
    InsertArgs ia; memset(&ia, 0, sizeof(ia));
    ia.glimpse = gl; ia.rec = rec_all; ia.rec_ld = RW; ia.img = obs; ia.mean_img = flat + po.dec_mean_img;
    ia.canvas = out.canvas; ia.data_ll = w.dll; ia.qz = w.qz; ia.pz = w.pz; ia.t = 0; ia.n_frames = T; ia.out = out;
    ia.std_fg = c.output_std; ia.std_bg = c.background_std;
    if (sq_launch_insert_loglik(ia, d, s) != 0) { sq_set_error(h, "sqair_forward: the decoder canvas launch failed (dynamic LDS limit)"); return -2; }
  }
  // final recurrent state (for state-level parity checks), in the caller's widths: [hidden | cell] halves without their padding
  const int unh = h->ucfg.n_hidden;
  if (out.final_temporal_state) {
    if (!h->padded) sq_copy(out.final_temporal_state, w.state(w.temporal_m, T, w.snh), (int64_t)M * snh, s);
    else SQ_LAUNCH(k_copy_cols, dim3(M), dim3(256), 0, s, (const float*)w.state(w.temporal_m, T, w.snh), out.final_temporal_state, snh / nh, unh, nh);
  }
  if (out.final_prior_state) {
    if (!h->padded) sq_copy(out.final_prior_state, w.state(w.prior_m, T, w.psnh), (int64_t)M * psnh, s);
    else SQ_LAUNCH(k_copy_cols, dim3(M), dim3(256), 0, s, (const float*)w.state(w.prior_m, T, w.psnh), out.final_prior_state, psnh / nh, unh, nh);
  }
  if (out.final_last_used_id)
    sq_copy(out.final_last_used_id, w.last_id[T & 1], (int64_t)R, s);
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}
static int forward_impl(SqairHandle* h, const float* flat, const float* packed, const float* obs, const float* noise,
                        int T, int B, int t_offset, const SqairOutputs* outp, float* wsbase, int64_t ws_bytes,
                        hipStream_t s) {
  return sq_forward_impl(h, flat, packed, obs, noise, T, B, t_offset, outp, wsbase, ws_bytes, s, false, 7);
}

// Training-mode forward pass: identical launch sequence and results, but every intermediate the backward pass needs
// is kept in the (larger) workspace; sqair_backward consumes it.
// The crop adjoint stages the whole frame in LDS (k_crop_chain_bwd, sqair_bwd.hip): frames beyond SQ_TRAIN_MAX_FRAME_BYTES can be
// evaluated (sqair_forward) but not trained.  Said here, at the entry points, instead of by a failed launch deep in the sweep.
bool sq_trainable_frame(SqairHandle* h) {
  const int64_t bytes = ((int64_t)h->cfg.img_h * h->cfg.img_w + 3) / 4 * 4 * 4;
  if (bytes <= SQ_TRAIN_MAX_FRAME_BYTES) return true;
  sq_set_error(h, "training is limited to frames of at most " + std::to_string(SQ_TRAIN_MAX_FRAME_BYTES / 4) + " pixels (" +
                      std::to_string(h->cfg.img_h) + " x " + std::to_string(h->cfg.img_w) +
                      " given): the crop adjoint stages the frame in LDS; inference (sqair_forward) has no such limit");
  return false;
}
extern "C" int sqair_forward_train(SqairHandle* h, const float* flat_params, const void* packed, const float* obs,
                                   const float* noise, int T, int B, int t_offset, const SqairOutputs* out,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  if (!h) return -1;
  if (!sq_trainable_frame(h)) return -1;
  return sq_forward_impl(h, flat_params, (const float*)packed, obs, noise, T, B, t_offset, out, (float*)workspace,
                         workspace_bytes, (hipStream_t)stream, true, 7);
}

// ------------------------------------------------------------------------------------------------
// workspace clearing policy (see sq_forward_impl's prologue)
extern "C" int sqair_set_workspace_clearing(SqairHandle* h, int each_pass) {
  if (!h) return -1;
  h->clear_each_pass = each_pass != 0;
  return 0;
}
extern "C" int sqair_clear_workspace(SqairHandle* h, void* workspace, int64_t workspace_bytes, int T, int B, int train, void* stream) {
  if (!h || !workspace || T < 1 || B < 1) return -1;
  const int64_t need = train ? sqair_train_workspace_bytes(h, T, B) : sqair_workspace_bytes(h, T, B);
  if (workspace_bytes < need) { sq_set_error(h, "sqair_clear_workspace: workspace too small"); return -1; }
  const Workspace w = sq_carve(h, T, B, (float*)workspace, train != 0);
  sq_zero_fill((float*)workspace, (int64_t)((float*)w.prof_ts - (float*)workspace), (hipStream_t)stream);
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// per-dispatch timeline (sqair_common.h: SQ_TLP / SQ_TL_SCOPE / SQ_LAUNCH; only libsqair_hip_timeline.so records anything)
#ifdef SQAIR_TIMELINE
// Launches reach sq_tl_next through SQ_LAUNCH without a handle, so the ACTIVE recording is one per process -- but it is OWNED by
// the handle that began it: a second handle cannot reset it (-6), its records stay readable by the owner after _end, and the
// owner's captured graphs (whose nodes carry slot addresses frozen at capture time) are dropped whenever a recording begins or
// ends, so that no replay can stamp into a buffer of another recording or one the caller has released.
namespace {
struct TlRecord { std::string kernel; int64_t off; int waves, wgs; };
struct TlState {
  const SqairHandle* owner = nullptr;
  unsigned long long* buf = nullptr; int64_t cap = 0, used = 0; bool on = false, overflow = false, bad_block = false;
  std::vector<TlRecord> rec;
};
TlState g_tl;
void tl_drop_graphs(SqairHandle* h) {
  if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
  if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
  h->graph_nodes = 0;
  for (int i = 0; i < 4; ++i) {
    if (h->cap_exec[i]) { (void)hipGraphExecDestroy(h->cap_exec[i]); h->cap_exec[i] = nullptr; }
    if (h->cap_graph[i]) { (void)hipGraphDestroy(h->cap_graph[i]); h->cap_graph[i] = nullptr; }
  }
}
}  // namespace
SqTl sq_tl_next(const char* kernel, dim3 grid, dim3 block) {
  if (!g_tl.on) return SqTl{nullptr, 0, 0, 0, 0};
  if (block.y != 1 || block.z != 1) { g_tl.bad_block = true; return SqTl{nullptr, 0, 0, 0, 0}; }  // (1-D workgroups only)
  const int64_t wgs = (int64_t)grid.x * grid.y * grid.z;
  // one {start, end} pair per wave; a workgroup's pairs are padded to whole 128-byte lines, so that workgroups on different
  // XCDs (whose L2s are not coherent with each other) never write parts of the same line
  const int64_t nw = (((int64_t)block.x * block.y * block.z + 63) / 64 + 7) / 8 * 8;
  const int64_t need = 2 * wgs * nw;
  if (g_tl.used + need > g_tl.cap) { g_tl.overflow = true; return SqTl{nullptr, 0, 0, 0, 0}; }
  TlRecord r;
  r.kernel = kernel; r.off = g_tl.used; r.waves = (int)(wgs * nw); r.wgs = (int)wgs;
  g_tl.rec.push_back(r);
  SqTl t{g_tl.buf + g_tl.used, grid.x, grid.y, (unsigned)nw, 0};
  g_tl.used += need;
  return t;
}
void sq_tl_forget(const SqairHandle* h) { if (g_tl.owner == h) g_tl = TlState(); }
extern "C" int sqair_timeline_available(void) { return 1; }
extern "C" int sqair_timeline_begin(SqairHandle* h, void* buf, int64_t bytes) {
  if (!h || !buf || bytes < 16 || ((uintptr_t)buf & 15)) return -1;
  if (g_tl.on && g_tl.owner != h) {
    sq_set_error(h, "sqair_timeline_begin: a recording is active on another handle (one recording per process; end it first)");
    return -6;
  }
  tl_drop_graphs(h);
  g_tl = TlState();
  g_tl.owner = h; g_tl.buf = (unsigned long long*)buf; g_tl.cap = bytes / 8; g_tl.on = true;
  return 0;
}
extern "C" int sqair_timeline_count(const SqairHandle* h) {
  if (!h || g_tl.owner != h) return -1;
  if (g_tl.overflow) return -4;   // dispatches since the overflow carry no stamps: a partial timeline must not be read as a whole one
  return (int)g_tl.rec.size();
}
extern "C" int sqair_timeline_end(SqairHandle* h) {
  if (!h) return -1;
  if (g_tl.owner != h) { sq_set_error(h, "sqair_timeline_end: this handle owns no recording"); return -6; }
  g_tl.on = false;
  tl_drop_graphs(h);
  if (g_tl.bad_block) { sq_set_error(h, "sqair_timeline_end: a kernel was launched with a multi-dimensional workgroup (the stamps need 1-D workgroups)"); return -4; }
  if (g_tl.overflow) { sq_set_error(h, "sqair_timeline_end: the stamp buffer was too small"); return -4; }
  return (int)g_tl.rec.size();
}
extern "C" int sqair_timeline_record(const SqairHandle* h, int i, const char** kernel, int64_t* offset_u64, int* waves, int* workgroups) {
  if (!h || g_tl.owner != h || i < 0 || i >= (int)g_tl.rec.size()) return -1;
  if (kernel) *kernel = g_tl.rec[i].kernel.c_str();
  if (offset_u64) *offset_u64 = g_tl.rec[i].off;
  if (waves) *waves = g_tl.rec[i].waves;
  if (workgroups) *workgroups = g_tl.rec[i].wgs;
  return 0;
}
#else
extern "C" int sqair_timeline_available(void) { return 0; }
extern "C" int sqair_timeline_begin(SqairHandle* h, void*, int64_t) {
  if (h) sq_set_error(h, "sqair_timeline_begin: this library was built without -DSQAIR_TIMELINE (use libsqair_hip_timeline.so)");
  return -3;
}
extern "C" int sqair_timeline_count(const SqairHandle*) { return -3; }
extern "C" int sqair_timeline_end(SqairHandle*) { return -3; }
extern "C" int sqair_timeline_record(const SqairHandle*, int, const char**, int64_t*, int*, int*) { return -3; }
#endif

// ------------------------------------------------------------------------------------------------
// documented run-time options (include/sqair_hip.h).  These are API calls of the caller, not environment variables: nothing in
// the environment can change what a pass computes or launches (the measurement knobs of tools/ exist only in -DSQAIR_KNOBS builds).
extern "C" int sqair_set_option(SqairHandle* h, const char* name, int value) {
  if (!h || !name) return -1;
  const std::string n(name);
  if (n == "tail_fusion") { h->opt_tail_fusion = value != 0; return 0; }
  if (n == "what_fusion") { h->opt_what_fusion = value != 0; return 0; }   // (results do not depend on it; drop captured graphs after changing it)
  if (n == "vi_target") {   // 0 = vimco (the reference's make_target), 1 = reinforce (targets.py:78-89)
    if (value != 0 && value != 1) { sq_set_error(h, "sqair_set_option: vi_target is 0 (vimco) or 1 (reinforce)"); return -2; }
    h->opt_vi_target = value;
    return 0;
  }
  if (n == "slot_chain") {
#ifdef SQAIR_WIDE
    if (value != 0) { sq_set_error(h, "sqair_set_option: slot_chain is not available in the wide build"); return -2; }
#endif
    h->opt_slot_chain = value != 0;
    h->opt_slot_chain_mode = value;   // (2: the chain's workspace layout with one launch per op -- a debugging aid)
    return 0;
  }
  if (n == "slot_chain_arena_kb") {   // size of one device arena of chain tables (default 32768): before the first chain launch
    if (sq_chain_set_arena_kb(h, value) != 0) { sq_set_error(h, "sqair_set_option: slot_chain_arena_kb is 64 .. 1048576, set before the first pass"); return -2; }
    return 0;
  }
  sq_set_error(h, "sqair_set_option: unknown option '" + n + "' (known: tail_fusion, what_fusion, slot_chain, slot_chain_arena_kb, vi_target)");
  return -2;
}

// debug mode of the reference (validate_args / allow_nan_stats=False of its distributions, sqair/core.py:226, :261,
// sqair/modules.py:318-320): a finite check of a device tensor that fails through the error channel.
__global__ __launch_bounds__(256) void k_finite_check(const float* __restrict__ x, int64_t n, int* __restrict__ flag SQ_TLP) {
  SQ_TL_SCOPE;
  int bad = 0;
  int64_t first = n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    if (!(fabsf(v) <= 3.4028234664e38f)) { ++bad; if (i < first) first = i; }
  }
  if (bad) {
    atomicAdd(&flag[0], bad);
    atomicMin(&flag[1], (int)(first < 0x7fffffff ? first : 0x7fffffff));
  }
}
extern "C" int sqair_check_finite(SqairHandle* h, const float* x, int64_t n, const char* what, int32_t* flag_dev, void* stream) {
  if (!h || !x || !flag_dev || n < 0) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int init[2] = {0, 0x7fffffff};
  int got[2] = {0, 0};
  SQ_CHECK_HIP(hipMemcpyAsync(flag_dev, init, sizeof(init), hipMemcpyHostToDevice, s));
  if (n > 0) {
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 1024);
    SQ_LAUNCH(k_finite_check, dim3(grid), dim3(256), 0, s, x, n, (int*)flag_dev);
  }
  SQ_CHECK_HIP(hipMemcpyAsync(got, flag_dev, sizeof(got), hipMemcpyDeviceToHost, s));
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  if (got[0] == 0) return 0;
  char msg[256];
  snprintf(msg, sizeof(msg), "non-finite values in %s: %d of %lld, first at flat index %d", what ? what : "tensor", got[0],
           (long long)n, got[1]);
  sq_set_error(h, msg);
  return -5;
}

// validate_args of the reference's Normal distributions (sqair/core.py:226, :261, sqair/modules.py:318-320: every scale must be
// positive -- of EVERY slot, also of the slots the presence mask later removes from the log-weights): the posterior scales of
// `what` and `where` of all propagation and discovery slots of the last pass on this workspace.
__global__ __launch_bounds__(256) void k_scale_check(const float* __restrict__ rec, int64_t n_rec, int nw, int* __restrict__ flag SQ_TLP) {
  SQ_TL_SCOPE;
  const int per = 4 + nw;
  int bad = 0;
  int64_t first = n_rec * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rec * per; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per;
    const int j = (int)(i - r * per);
    const float v = rec[r * rec::W + (j < 4 ? rec::WHERE_SCALE + j : rec::WHAT_SCALE + (j - 4))];
    if (!(v > 0.0f && v <= 3.4028234664e38f)) { ++bad; if (i < first) first = i; }
  }
  if (bad) {
    atomicAdd(&flag[0], bad);
    atomicMin(&flag[1], (int)(first < 0x7fffffff ? first : 0x7fffffff));
  }
}
extern "C" int sqair_check_scales(SqairHandle* h, void* workspace, int T, int B, int train, int32_t* flag_dev, void* stream) {
  if (!h || !workspace || !flag_dev || T < 1 || B < 1) return -1;
  hipStream_t s = (hipStream_t)stream;
  const Workspace w = sq_carve(h, T, B, (float*)workspace, train != 0);
  const int64_t n_rec = (int64_t)T * w.M;
  const char* names[2] = {"propagation", "discovery"};
  const float* recs[2] = {w.rec_p_all, w.rec_d_all};
  for (int ph = 0; ph < 2; ++ph) {
    const int init[2] = {0, 0x7fffffff};
    int got[2] = {0, 0};
    SQ_CHECK_HIP(hipMemcpyAsync(flag_dev, init, sizeof(init), hipMemcpyHostToDevice, s));
    SQ_LAUNCH(k_scale_check, dim3((int)std::min<int64_t>((n_rec * (4 + h->cfg.n_what) + 255) / 256, 1024)), dim3(256), 0, s, recs[ph], n_rec,
              h->cfg.n_what, (int*)flag_dev);
    SQ_CHECK_HIP(hipMemcpyAsync(got, flag_dev, sizeof(got), hipMemcpyDeviceToHost, s));
    SQ_CHECK_HIP(hipStreamSynchronize(s));
    if (got[0] != 0) {
      const int per = 4 + h->cfg.n_what, rec_i = got[1] / per, j = got[1] % per;
      char msg[320];
      snprintf(msg, sizeof(msg), "scale not positive / not finite in the %s posterior: %d values, first at frame %d, row %d, slot %d, %s[%d] "
               "(validate_args: every slot's scale is checked, also slots the presence mask removes)", names[ph], got[0],
               rec_i / w.M, (rec_i % w.M) / w.N, rec_i % w.N, j < 4 ? "where_scale" : "what_scale", j < 4 ? j : j - 4);
      sq_set_error(h, msg);
      return -5;
    }
  }
  return 0;
}

extern "C" int sqair_forward(SqairHandle* h, const float* flat_params, const void* packed, const float* obs,
                             const float* noise, int T, int B, int t_offset, const SqairOutputs* out, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  if (!h) return -1;
  return forward_impl(h, flat_params, (const float*)packed, obs, noise, T, B, t_offset, out, (float*)workspace,
                      workspace_bytes, (hipStream_t)stream);
}

extern "C" int sqair_graph_capture(SqairHandle* h, const float* flat_params, const void* packed, const float* obs,
                                   const float* noise, int T, int B, int t_offset, const SqairOutputs* out,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  if (!h) return -1;
  hipStream_t s = (hipStream_t)stream;
  if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
  if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
  if (h->opt_slot_chain) {  // one eager pass: the chain launches' op tables are uploaded outside the capture (sqair_chain.hip)
    const int rc0 = forward_impl(h, flat_params, (const float*)packed, obs, noise, T, B, t_offset, out, (float*)workspace, workspace_bytes, s);
    if (rc0 != 0) return rc0;
    SQ_CHECK_HIP(hipStreamSynchronize(s));
  }
  SQ_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  int rc = forward_impl(h, flat_params, (const float*)packed, obs, noise, T, B, t_offset, out, (float*)workspace,
                        workspace_bytes, s);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s, &g);
  if (rc != 0) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) {
    sq_set_error(h, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    return -2;
  }
  h->graph = g;
  size_t nn = 0;
  SQ_CHECK_HIP(hipGraphGetNodes(g, nullptr, &nn));
  h->graph_nodes = (int)nn;
  SQ_CHECK_HIP(hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0));
  return 0;
}

// Status of the in-launch slot chain's launches of the last pass on this workspace (synchronises the stream): 0 = every launch
// completed; otherwise the first non-zero status word (1 = a consumer gave up polling for an operand, 3 = the census of the
// workgroups never completed, 4 = more row tiles per XCD than the kernel is laid out for) -- the pass's results are then not valid.
// Returns 0 as well when the chain is off for this shape.
extern "C" int sqair_chain_status(SqairHandle* h, void* workspace, int T, int B, int train, void* stream) {
  if (!h || !workspace || T < 1 || B < 1) return -1;
  const Workspace w = sq_carve(h, T, B, (float*)workspace, train != 0);
  if (!w.chain) return 0;
  std::vector<unsigned> ctl((size_t)SQ_CHAIN_MAX_LAUNCHES * SQ_CHAIN_CTL_WORDS);
  SQ_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  SQ_CHECK_HIP(hipMemcpy(ctl.data(), w.chain_ctl, ctl.size() * 4, hipMemcpyDeviceToHost));
  for (int l = 0; l < 2 * T; ++l) {
    const unsigned st = ctl[(size_t)l * SQ_CHAIN_CTL_WORDS + 9 * 32];
    if (st != 0) {
      sq_set_error(h, "slot chain: launch " + std::to_string(l) + " ended with status " + std::to_string(st));
      return (int)st;
    }
  }
  return 0;
}

extern "C" int sqair_set_generation_noise(SqairHandle* h, const float* gen_noise) {
  if (!h) return -1;
  h->gen_noise = gen_noise;
  return 0;
}

extern "C" int sqair_graph_launch(SqairHandle* h, void* stream) {
  if (!h || !h->graph_exec) {
    sq_set_error(h, "sqair_graph_launch: no captured graph");
    return -1;
  }
  SQ_CHECK_HIP(hipGraphLaunch(h->graph_exec, (hipStream_t)stream));
  return 0;
}
extern "C" int sqair_graph_nodes(const SqairHandle* h) { return h ? h->graph_nodes : -1; }

extern "C" int sqair_elbo(SqairHandle* h, const float* log_w_t, const float* disc_lp_t, int T, int B, float* log_weights,
                          float* elbo_iwae_per_example, float* importance_weights, float* vimco_signal,
                          float* scalars_out, const float* const* iw_means_in, int n_means, float* iw_means_out,
                          void* stream) {
  if (!h || !log_w_t || T < 1 || B < 1 || n_means < 0 || n_means > 8) return -1;
  sq_launch_elbo(log_w_t, disc_lp_t, T, B, h->cfg.k_particles, log_weights, elbo_iwae_per_example, importance_weights,
                 vimco_signal, scalars_out, iw_means_in, n_means, iw_means_out, (hipStream_t)stream, h->opt_vi_target);
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// per-kernel entry points for unit parity tests
// ------------------------------------------------------------------------------------------------
extern "C" int sqair_st_crop(SqairHandle* h, const float* img, const float* where_logits, const float* mask, float* out,
                             int B, void* stream) {
  if (!h || !img || !where_logits || !out || B < 1) return -1;
  const SqairConfig& c = h->cfg;
  Dims d = make_dims(c, B);
  CropArgs ca; memset(&ca, 0, sizeof(ca));
  ca.mode = CROP_PLAIN; ca.img = img; ca.logits = where_logits; ca.mask = mask; ca.mask_row_mul = 1; ca.out = out;
  ca.out_row_mul = 1;
  sq_launch_crop(ca, h->po, d, 1, (hipStream_t)stream);
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int sqair_st_insert_loglik(SqairHandle* h, const float* glimpse, const float* where_logits,
                                      const float* presence, const float* img, const float* mean_img, float* canvas,
                                      float* data_ll, int B, void* stream) {
  if (!h || !glimpse || !where_logits || !presence || !img || !mean_img || !data_ll || B < 1) return -1;
  const SqairConfig& c = h->cfg;
  Dims d = make_dims(c, B);
  InsertArgs ia; memset(&ia, 0, sizeof(ia));
  ia.glimpse = glimpse; ia.where_plain = where_logits; ia.pres_plain = presence; ia.img = img; ia.mean_img = mean_img;
  ia.canvas = canvas; ia.data_ll = data_ll; ia.std_fg = c.output_std; ia.std_bg = c.background_std;
  if (sq_launch_insert_loglik(ia, d, (hipStream_t)stream) != 0) { sq_set_error(h, "sqair_st_insert_loglik: launch failed (dynamic LDS limit)"); return -2; }
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}

// ad-hoc pack of a dense [K,N] matrix on the host plan path: scratch = [idx int32 | packed w | packed b | bidx]
static int adhoc_layer(SqairHandle* h, int Kdim, int Ndim, PackedLayer* L) {
  L->seg_width = {Kdim};
  L->kc = (Kdim + 15) / 16;
  L->nt = (Ndim + 15) / 16;
  L->N = Ndim;
  L->w_off = 0;
  L->b_off = 0;
  (void)h;
  return 0;
}

// fills idx for a dense [K, N] row-major matrix located at flat offset `woff`, columns col0..col0+ncols of a
// matrix with leading dimension ldw, into packed columns n0.. of a layer with `kc` chunks, segment chunk base cbase
static void adhoc_fill(std::vector<int>& idx, int kc, int cbase, int n0, int ncols, int Kdim, int64_t woff, int ldw, int col0) {
  for (int j = 0; j < ncols; ++j) {
    const int n = n0 + j, tile = n / 16, ln = n % 16;
    for (int k = 0; k < Kdim; ++k) {
      const int cch = cbase + k / 16, kin = k % 16;
      const int lane = (kin / 4) * 16 + ln, comp = kin % 4;
      idx[(((int64_t)tile * kc + cch) * 64 + lane) * 4 + comp] = (int)(woff + (int64_t)k * ldw + col0 + j);
    }
  }
}

extern "C" int sqair_linear_test(SqairHandle* h, const float* x, const float* wmat, const float* b, float* y, int M,
                                 int Kdim, int Ndim, int act, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!h || !x || !wmat || !y || !scratch) return -1;
  hipStream_t s = (hipStream_t)stream;
  PackedLayer L;
  adhoc_layer(h, Kdim, Ndim, &L);
  const int64_t nel = (int64_t)L.nt * L.kc * 256, nb = L.nt * 16;
  const int kpad = (Kdim + 3) & ~3;
  if (scratch_bytes < (2 * nel + 2 * nb + 256 + (int64_t)M * kpad) * 4) { sq_set_error(h, "sqair_linear_test: scratch too small"); return -1; }
  std::vector<int> idx(nel, -1), bidx(nb, -1);
  adhoc_fill(idx, L.kc, 0, 0, Ndim, Kdim, 0, Ndim, 0);
  for (int n = 0; n < Ndim; ++n) bidx[n] = b ? n : -1;
  int* d_idx = (int*)scratch;
  float* d_w = (float*)scratch + nel;
  float* d_b = d_w + nel;
  int* d_bidx = (int*)(d_b + nb);
  float* d_zero = (float*)(d_bidx + nb);
  float* d_x = d_zero + 256;  // A-operand contract: rows padded to a multiple of 4 floats, finite
  SQ_CHECK_HIP(hipMemcpyAsync(d_idx, idx.data(), nel * 4, hipMemcpyHostToDevice, s));
  SQ_CHECK_HIP(hipMemcpyAsync(d_bidx, bidx.data(), nb * 4, hipMemcpyHostToDevice, s));
  SQ_CHECK_HIP(hipMemsetAsync(d_zero, 0, (256 + (size_t)M * kpad) * 4, s));
  SQ_CHECK_HIP(hipMemcpy2DAsync(d_x, (size_t)kpad * 4, x, (size_t)Kdim * 4, (size_t)Kdim * 4, M, hipMemcpyDeviceToDevice, s));
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  sq_launch_pack(wmat, d_w, d_idx, nel, s);
  if (b) sq_launch_pack(b, d_b, d_bidx, nb, s);
  else SQ_CHECK_HIP(hipMemsetAsync(d_b, 0, nb * 4, s));
  Lin l;
  l.seg(d_x, kpad, Kdim).out(y, Ndim).act(act);
  l.a.wp = d_w; l.a.wzero = d_zero; l.a.bias = d_b; l.a.M = M; l.a.N = Ndim;
  if (sq_launch_linear(l.a, L, s) != 0) { sq_set_error(h, "sqair_linear_test: A-operand contract violated"); return -5; }
  SQ_CHECK_HIP(hipGetLastError());
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  if (h->debug_reps > 0 && h->debug_graph_nodes > 0) {
    // sqair_debug_linear_graph_time: `nodes` launches of this layer captured as ONE HIP graph -- the same launch over and over
    // (no data dependence), or (dependent) every launch reading what the previous one wrote -- and the graph replayed
    // `reps` times between two HIP events: time per graph NODE (kernel + the dependent-dispatch boundary), on whatever build
    // of the library this is.  The figure bench.py's timeline calls the SLOT of a dense launch, measured without the stamps.
    // dependent: every launch reads what the previous one wrote -- IN PLACE, input columns [0, K) and output columns [0, N) of
    // one buffer of pitch max(K, N) behind the other scratch (a race between the workgroups of a row tile: the values mean
    // nothing, tanh keeps them finite; what is timed is a launch whose operand comes out of the previous launch's stores)
    Lin l2;
    const bool dep = h->debug_graph_dependent;
    if (dep) {
      const int ldz = (std::max(kpad, Ndim) + 3) & ~3;
      float* d_z = d_x + (size_t)M * kpad;
      if (scratch_bytes < (2 * nel + 2 * nb + 256 + (int64_t)M * kpad + (int64_t)M * ldz + 16) * 4) { sq_set_error(h, "sqair_debug_linear_graph_time: scratch too small"); return -1; }
      SQ_CHECK_HIP(hipMemsetAsync(d_z, 0, ((size_t)M * ldz + 16) * 4, s));
      SQ_CHECK_HIP(hipMemcpy2DAsync(d_z, (size_t)ldz * 4, x, (size_t)Kdim * 4, (size_t)Kdim * 4, M, hipMemcpyDeviceToDevice, s));
      l2.seg(d_z, ldz, Kdim).out(d_z, ldz).act(act);
      l2.a.wp = d_w; l2.a.wzero = d_zero; l2.a.bias = d_b; l2.a.M = M; l2.a.N = Ndim;
    }
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    SQ_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < h->debug_graph_nodes; ++i) sq_launch_linear(dep ? l2.a : l.a, L, s);
    SQ_CHECK_HIP(hipStreamEndCapture(s, &g));
    SQ_CHECK_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t ea, eb;
    SQ_CHECK_HIP(hipEventCreate(&ea));
    SQ_CHECK_HIP(hipEventCreate(&eb));
    for (int i = 0; i < 2; ++i) SQ_CHECK_HIP(hipGraphLaunch(ge, s));
    (void)hipEventRecord(ea, s);
    for (int i = 0; i < h->debug_reps; ++i) SQ_CHECK_HIP(hipGraphLaunch(ge, s));
    (void)hipEventRecord(eb, s);
    SQ_CHECK_HIP(hipStreamSynchronize(s));
    float ms = 0.0f;
    SQ_CHECK_HIP(hipEventElapsedTime(&ms, ea, eb));
    h->debug_us = ms * 1e3f / ((float)h->debug_reps * (float)h->debug_graph_nodes);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
  } else if (h->debug_reps > 0) {  // sqair_debug_linear_time: the same launch `reps` times back to back between two events
    hipEvent_t ea, eb;
    SQ_CHECK_HIP(hipEventCreate(&ea));
    SQ_CHECK_HIP(hipEventCreate(&eb));
    for (int i = 0; i < 3; ++i) sq_launch_linear(l.a, L, s);
    (void)hipEventRecord(ea, s);
    for (int i = 0; i < h->debug_reps; ++i) sq_launch_linear(l.a, L, s);
    (void)hipEventRecord(eb, s);
    SQ_CHECK_HIP(hipStreamSynchronize(s));
    float ms = 0.0f;
    SQ_CHECK_HIP(hipEventElapsedTime(&ms, ea, eb));
    h->debug_us = ms * 1e3f / (float)h->debug_reps;
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
  }
  return 0;
}
// measurement helper (tools/time_linear.py): average time of one dense launch of the given shape, launches back to back
extern "C" int sqair_debug_linear_time(SqairHandle* h, const float* x, const float* wmat, const float* b, float* y, int M, int Kdim,
                                       int Ndim, int act, void* scratch, int64_t scratch_bytes, int reps, float* us_out, void* stream) {
  if (!h || !us_out || reps < 1) return -1;
  h->debug_reps = reps;
  const int rc = sqair_linear_test(h, x, wmat, b, y, M, Kdim, Ndim, act, scratch, scratch_bytes, stream);
  h->debug_reps = 0;
  *us_out = h->debug_us;
  return rc;
}

// measurement helper (tools/dense_graph_time.py): the dense launches of the passes issued between (on = 1) and the read-out.
// entry i -> {layer id (-1 - id: the VanillaRNN layer with the slot tail fused in front), rows, K padded, N}; returns the count.
extern "C" int sqair_debug_dense_log(SqairHandle* h, int on) {
  if (!h) return -1;
  if (on) h->dense_log.clear();
  h->dense_log_on = on != 0;
  return (int)(h->dense_log.size() / 4);
}
extern "C" int sqair_debug_dense_log_entry(const SqairHandle* h, int i, int* out4) {
  if (!h || !out4 || i < 0 || (size_t)i * 4 + 3 >= h->dense_log.size()) return -1;
  for (int q = 0; q < 4; ++q) out4[q] = h->dense_log[(size_t)i * 4 + q];
  return 0;
}
// measurement helper (tools/dense_graph_time.py): time per NODE of a HIP graph of `nodes` launches of one dense layer -- the same
// launch again and again, or (dependent != 0) each reading what the previous one wrote -- replayed `replays` times between HIP events
extern "C" int sqair_debug_linear_graph_time(SqairHandle* h, const float* x, const float* wmat, const float* b, float* y, int M, int Kdim,
                                             int Ndim, int act, void* scratch, int64_t scratch_bytes, int nodes, int replays,
                                             int dependent, float* us_out, void* stream) {
  if (!h || !us_out || nodes < 1 || replays < 1) return -1;
  h->debug_reps = replays; h->debug_graph_nodes = nodes; h->debug_graph_dependent = dependent != 0;
  const int rc = sqair_linear_test(h, x, wmat, b, y, M, Kdim, Ndim, act, scratch, scratch_bytes, stream);
  h->debug_reps = 0; h->debug_graph_nodes = 0;
  *us_out = h->debug_us;
  return rc;
}

extern "C" int sqair_gru_test(SqairHandle* h, const float* x, const float* hstate, const float* gru_flat, float* h_out,
                              int M, int Kx, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!h || !x || !hstate || !gru_flat || !h_out || !scratch) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int nh = h->cfg.n_hidden;
  // gru_flat: for g in (z, r, h): w_g [Kx,nh], u_g [nh,nh], b_g [nh]  (the order add_gru() lays them out)
  const int64_t gs = (int64_t)Kx * nh + (int64_t)nh * nh + nh;
  PackedLayer L1, L2;
  L1.seg_width = {Kx, nh};
  L1.kc = (Kx + 15) / 16 + (nh + 15) / 16;
  L1.nt = 3 * nh / 16; L1.N = 3 * nh; L1.w_off = 0; L1.b_off = 0;
  adhoc_layer(h, nh, nh, &L2);
  const int64_t n1 = (int64_t)L1.nt * L1.kc * 256, n2 = (int64_t)L2.nt * L2.kc * 256, nb1 = L1.nt * 16, nb2 = L2.nt * 16;
  const int kpad = (Kx + 3) & ~3;
  const int64_t need = (2 * n1 + 2 * n2 + 2 * nb1 + nb2 + 3 * (int64_t)M * nh + 256 + (int64_t)M * kpad) * 4;
  if (scratch_bytes < need) { sq_set_error(h, "sqair_gru_test: scratch too small"); return -1; }
  std::vector<int> i1(n1, -1), i2(n2, -1), b1(nb1, -1);
  const int xc = (Kx + 15) / 16;
  for (int g = 0; g < 3; ++g) {
    adhoc_fill(i1, L1.kc, 0, g * nh, nh, Kx, g * gs, nh, 0);
    if (g < 2) adhoc_fill(i1, L1.kc, xc, g * nh, nh, nh, g * gs + (int64_t)Kx * nh, nh, 0);
    for (int n = 0; n < nh; ++n) b1[g * nh + n] = (int)(g * gs + (int64_t)Kx * nh + (int64_t)nh * nh + n);
  }
  adhoc_fill(i2, L2.kc, 0, 0, nh, nh, 2 * gs + (int64_t)Kx * nh, nh, 0);
  float* f = (float*)scratch;
  int* d_i1 = (int*)f; f += n1;
  float* d_w1 = f; f += n1;
  int* d_i2 = (int*)f; f += n2;
  float* d_w2 = f; f += n2;
  int* d_b1i = (int*)f; f += nb1;
  float* d_b1 = f; f += nb1;
  float* d_b2 = f; f += nb2;
  float* d_z = f; f += (int64_t)M * nh;
  float* d_rh = f; f += (int64_t)M * nh;
  float* d_xh = f; f += (int64_t)M * nh;
  float* d_zero = f; f += 256;
  float* d_x = f; f += (int64_t)M * kpad;
  SQ_CHECK_HIP(hipMemsetAsync(d_zero, 0, (256 + (size_t)M * kpad) * 4, s));
  SQ_CHECK_HIP(hipMemcpy2DAsync(d_x, (size_t)kpad * 4, x, (size_t)Kx * 4, (size_t)Kx * 4, M, hipMemcpyDeviceToDevice, s));
  SQ_CHECK_HIP(hipMemcpyAsync(d_i1, i1.data(), n1 * 4, hipMemcpyHostToDevice, s));
  SQ_CHECK_HIP(hipMemcpyAsync(d_i2, i2.data(), n2 * 4, hipMemcpyHostToDevice, s));
  SQ_CHECK_HIP(hipMemcpyAsync(d_b1i, b1.data(), nb1 * 4, hipMemcpyHostToDevice, s));
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  sq_launch_pack(gru_flat, d_w1, d_i1, n1, s);
  sq_launch_pack(gru_flat, d_w2, d_i2, n2, s);
  sq_launch_pack(gru_flat, d_b1, d_b1i, nb1, s);
  SQ_CHECK_HIP(hipMemsetAsync(d_b2, 0, nb2 * 4, s));
  Lin g1;
  g1.seg(d_x, kpad, Kx).seg(hstate, nh, nh).out(d_z, nh).gru1(hstate, nh, d_rh, nh, d_xh, nh, nh);
  g1.a.wp = d_w1; g1.a.wzero = d_zero; g1.a.bias = d_b1; g1.a.M = M; g1.a.N = 3 * nh;
  if (sq_launch_linear(g1.a, L1, s) != 0) { sq_set_error(h, "sqair_gru_test: A-operand contract violated"); return -5; }
  Lin g2;
  g2.seg(d_rh, nh, nh).add(d_xh, nh, nh).out(h_out, nh).gru2(hstate, nh, d_z, nh, nh);
  g2.a.wp = d_w2; g2.a.wzero = d_zero; g2.a.bias = d_b2; g2.a.M = M; g2.a.N = nh;
  sq_launch_linear(g2.a, L2, s);
  SQ_CHECK_HIP(hipGetLastError());
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// snt.LSTM step through the launches the forward pass uses: one gate GEMM over [x | h] + the element-wise cell kernel
extern "C" int sqair_lstm_test(SqairHandle* h, const float* x, const float* hstate, const float* cstate, const float* lstm_flat,
                               float* state_out, int M, int Kx, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!h || !x || !hstate || !cstate || !lstm_flat || !state_out || !scratch) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int nh = h->cfg.n_hidden;
  // lstm_flat: w_gates [(Kx + nh), 4 nh] then b_gates [4 nh] (the order of the reference's LSTM variables)
  PackedLayer L;
  L.seg_width = {Kx, nh};
  L.kc = (Kx + 15) / 16 + (nh + 15) / 16;
  L.nt = 4 * nh / 16; L.N = 4 * nh; L.w_off = 0; L.b_off = 0;
  const int64_t nel = (int64_t)L.nt * L.kc * 256, nb = L.nt * 16;
  const int kpad = (Kx + 3) & ~3;
  const int64_t need = (2 * nel + 2 * nb + 256 + (int64_t)M * kpad + (int64_t)M * 4 * nh) * 4;
  if (scratch_bytes < need) { sq_set_error(h, "sqair_lstm_test: scratch too small"); return -1; }
  std::vector<int> idx(nel, -1), bidx(nb, -1);
  adhoc_fill(idx, L.kc, 0, 0, 4 * nh, Kx, 0, 4 * nh, 0);
  adhoc_fill(idx, L.kc, (Kx + 15) / 16, 0, 4 * nh, nh, (int64_t)Kx * 4 * nh, 4 * nh, 0);
  for (int n = 0; n < 4 * nh; ++n) bidx[n] = (int)((int64_t)(Kx + nh) * 4 * nh + n);
  float* f = (float*)scratch;
  int* d_idx = (int*)f; f += nel;
  float* d_w = f; f += nel;
  int* d_bidx = (int*)f; f += nb;
  float* d_b = f; f += nb;
  float* d_zero = f; f += 256;
  float* d_x = f; f += (int64_t)M * kpad;
  float* d_g = f;
  SQ_CHECK_HIP(hipMemsetAsync(d_zero, 0, (256 + (size_t)M * kpad) * 4, s));
  SQ_CHECK_HIP(hipMemcpy2DAsync(d_x, (size_t)kpad * 4, x, (size_t)Kx * 4, (size_t)Kx * 4, M, hipMemcpyDeviceToDevice, s));
  SQ_CHECK_HIP(hipMemcpyAsync(d_idx, idx.data(), nel * 4, hipMemcpyHostToDevice, s));
  SQ_CHECK_HIP(hipMemcpyAsync(d_bidx, bidx.data(), nb * 4, hipMemcpyHostToDevice, s));
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  sq_launch_pack(lstm_flat, d_w, d_idx, nel, s);
  sq_launch_pack(lstm_flat, d_b, d_bidx, nb, s);
  Lin g;
  g.seg(d_x, kpad, Kx).seg(hstate, nh, nh).out(d_g, 4 * nh);
  g.a.wp = d_w; g.a.wzero = d_zero; g.a.bias = d_b; g.a.M = M; g.a.N = 4 * nh;
  if (sq_launch_linear(g.a, L, s) != 0) { sq_set_error(h, "sqair_lstm_test: A-operand contract violated"); return -5; }
  sq_launch_lstm_cell(d_g, 4 * nh, cstate, nh, state_out, 2 * nh, M, nh, s);
  SQ_CHECK_HIP(hipGetLastError());
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

int sq_launch_lstm_cell_bwd(const float* gates, int g_ld, const float* c_prev, int c_ld, const float* d_h, int dh_ld, const float* d_c,
                            int dc_ld, float* d_gates, int dg_ld, float* d_cprev, int dcp_ld, int rows, int nh, hipStream_t s,
                            float* d_gates2 = nullptr, int dg2_ld = 0);
// adjoint of the element-wise LSTM cell: gates [M, 4 nh] (pre-activations i, j, f, o), c_prev, d h', d c' -> d gates, d c_prev
extern "C" int sqair_lstm_cell_bwd_test(SqairHandle* h, const float* gates, const float* c_prev, const float* d_h, const float* d_c,
                                        float* d_gates, float* d_cprev, int M, void* stream) {
  if (!h || !gates || !c_prev || !d_h || !d_c || !d_gates || !d_cprev) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int nh = h->cfg.n_hidden;
  sq_launch_lstm_cell_bwd(gates, 4 * nh, c_prev, nh, d_h, nh, d_c, nh, d_gates, 4 * nh, d_cprev, nh, M, nh, s);
  SQ_CHECK_HIP(hipGetLastError());
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// host-only introspection of the packing plan (tests/test_pack_plan.py emulates the packed GEMMs on
// the CPU from these tables to check the row / column maps without a GPU)
// ------------------------------------------------------------------------------------------------
extern "C" int sqair_debug_layers(const SqairHandle* h) { return h ? (int)L_COUNT : -1; }
// padded inventory: total floats of the padded flat buffer; u2i (optional, n = sqair_param_count entries): where element j of the
// caller's flat buffer lives in it
extern "C" int64_t sqair_debug_padded_count(const SqairHandle* h, int* u2i) {
  if (!h) return -1;
  if (u2i) memcpy(u2i, h->u2i.data(), h->u2i.size() * sizeof(int));
  return h->n_params;
}
extern "C" int sqair_debug_layer(const SqairHandle* h, int id, int* kc, int* nt, int* n, int* nseg, int* seg_widths) {
  if (!h || id < 0 || id >= L_COUNT) return -1;
  const PackedLayer& L = h->layers[id];
  *kc = L.kc; *nt = L.nt; *n = L.N; *nseg = (int)L.seg_width.size();
  for (size_t i = 0; i < L.seg_width.size() && i < 4; ++i) seg_widths[i] = L.seg_width[i];
  return 0;
}
// widx: [nt*kc*256] flat-parameter index per packed weight element; bidx_a/b: [nt*16]
extern "C" int sqair_debug_plan(const SqairHandle* h, int id, int* widx, int* bidx_a, int* bidx_b) {
  if (!h || id < 0 || id >= L_COUNT) return -1;
  const PackedLayer& L = h->layers[id];
  memcpy(widx, h->widx.data() + L.w_off, (size_t)L.nt * L.kc * 256 * sizeof(int));
  memcpy(bidx_a, h->bidx_a.data() + L.b_off, (size_t)L.nt * 16 * sizeof(int));
  memcpy(bidx_b, h->bidx_b.data() + L.b_off, (size_t)L.nt * 16 * sizeof(int));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// dense-layer backward through the MFMA kernels (unit-test entry): y = act(x W + b) was computed forward;
// given dL/dy returns dL/dx (k_linear on the transposed pack), dL/dW and dL/db (k_wgrad).
// ------------------------------------------------------------------------------------------------
int sq_launch_wgrad(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, float* db, int M, int Kdim,
                    int Ndim, int accumulate, hipStream_t s, const int* rowmap = nullptr, const float* alpha_ptr = nullptr);
int sq_launch_insert_bwd_frames(const float* glimpse, const float* rec, int rec_ld, const float* img, const float* mean_img,
                                const float* g_ll, float* d_glimpse, float* d_rec, int d_rec_ld, float* d_mean_rows,
                                float std_fg, float std_bg, int T, Dims d, hipStream_t s, const float* scale = nullptr,
                                float* d_scale = nullptr);
int sq_launch_reduce_rows(const float* rows, float* out, int R, int P, int accumulate, hipStream_t s);
int sq_launch_elbo_bwd(const float* iw, const float* sig, int T, int B, int K, float* g_lw, float* g_dl, hipStream_t s);
int sq_launch_dot_scale(const float* a, const float* b, int64_t n, const float* scale, float* out, hipStream_t s);
int sq_launch_dact(const float* d_out, const float* out, float* d_pre, int64_t n, int act, hipStream_t s);

extern "C" int sqair_linear_bwd_test(SqairHandle* h, const float* x, const float* wmat, const float* y, const float* dy,
                                     float* dx, float* dw, float* db, int M, int Kdim, int Ndim, int act, void* scratch,
                                     int64_t scratch_bytes, void* stream) {
  if (!h || !x || !wmat || !y || !dy || !dx || !dw || !db || !scratch) return -1;
  hipStream_t s = (hipStream_t)stream;
  // transposed layer: K' = Ndim inputs (dpre), N' = Kdim outputs (dx)
  PackedLayer L;
  adhoc_layer(h, Ndim, Kdim, &L);
  const int64_t nel = (int64_t)L.nt * L.kc * 256, nb = L.nt * 16;
  const int npad = (Ndim + 3) & ~3;
  if (scratch_bytes < (2 * nel + nb + 256 + (int64_t)M * npad + (int64_t)M * Ndim) * 4) {
    sq_set_error(h, "sqair_linear_bwd_test: scratch too small");
    return -1;
  }
  std::vector<int> idx(nel, -1);
  for (int j = 0; j < Kdim; ++j) {       // output column j of the transposed layer = input k of the forward layer
    const int tile = j / 16, ln = j % 16;
    for (int kk = 0; kk < Ndim; ++kk) {  // reduction index = forward output n
      const int cch = kk / 16, kin = kk % 16;
      const int lane = (kin / 4) * 16 + ln, comp = kin % 4;
      idx[(((int64_t)tile * L.kc + cch) * 64 + lane) * 4 + comp] = j * Ndim + kk;  // W[j][kk]
    }
  }
  int* d_idx = (int*)scratch;
  float* d_w = (float*)scratch + nel;
  float* d_b = d_w + nel;
  float* d_zero = d_b + nb;
  float* d_dpre_pad = d_zero + 256;
  float* d_dpre = d_dpre_pad + (int64_t)M * npad;
  SQ_CHECK_HIP(hipMemcpyAsync(d_idx, idx.data(), nel * 4, hipMemcpyHostToDevice, s));
  SQ_CHECK_HIP(hipMemsetAsync(d_b, 0, (nb + 256 + (size_t)M * npad) * 4, s));
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  sq_launch_pack(wmat, d_w, d_idx, nel, s);
  sq_launch_dact(dy, y, d_dpre, (int64_t)M * Ndim, act, s);
  sq_launch_wgrad(x, Kdim, d_dpre, Ndim, dw, Ndim, db, M, Kdim, Ndim, 0, s);
  SQ_CHECK_HIP(hipMemcpy2DAsync(d_dpre_pad, (size_t)npad * 4, d_dpre, (size_t)Ndim * 4, (size_t)Ndim * 4, M,
                                hipMemcpyDeviceToDevice, s));
  Lin l;
  l.seg(d_dpre_pad, npad, Ndim).out(dx, Kdim).act(ACT_NONE);
  l.a.wp = d_w; l.a.wzero = d_zero; l.a.bias = d_b; l.a.M = M; l.a.N = Kdim;
  if (sq_launch_linear(l.a, L, s) != 0) { sq_set_error(h, "sqair_linear_bwd_test: A-operand contract violated"); return -5; }
  SQ_CHECK_HIP(hipGetLastError());
  SQ_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// sqair_backward, decoder branch (first slice of the training step).
// Must follow sqair_forward + sqair_elbo on the same workspace (it reads the merged records, decoder activations
// and glimpses of all T frames the forward pass left there).  Computes
//   dL/d(log w_t) from the VIMCO target  ->  insert / log-likelihood adjoint of all T frames  ->  three dense
//   layers backward (dX through the transposed packs, dW / db through k_wgrad over M = T*B'*N rows)
// and writes the gradients of every decoder parameter (dec.mean_img, dec.l0-2.{w,b}, dec.output_scale) into
// flat_grad (other entries untouched) plus the seed gradients on the merged latents d_rec [T, B'*N, 64] (record
// order: where 0:4, what 4:54) that the recurrent part of the backward pass will consume.
// ------------------------------------------------------------------------------------------------
extern "C" int64_t sqair_backward_scratch_bytes(const SqairHandle* h, int T, int B) {
  if (!h || T < 1 || B < 1) return -1;
  const SqairConfig& c = h->cfg;
  const int64_t R = (int64_t)B * c.k_particles, M = R * c.n_steps_per_image, MT = M * T;
  const int64_t G2 = c.glimpse_size * c.glimpse_size, P_ = c.img_h * c.img_w, nh = c.n_hidden;
  return (2 * align64(T * R) + align64(MT * G2) + align64(T * R * P_) + 2 * align64(MT * (nh > G2 ? nh : G2)) +
          align64(MT * 64) + 1024) * 4;
}

extern "C" int sqair_backward_decoder(SqairHandle* h, const float* flat, const void* packedv, const float* obs,
                                      const float* importance_weights, const float* vimco_signal, int T, int B,
                                      void* workspace, int64_t workspace_bytes, void* scratch, int64_t scratch_bytes,
                                      float* flat_grad, float* d_rec_out, void* stream) {
  if (!h || !flat || !packedv || !obs || !importance_weights || !vimco_signal || !workspace || !scratch || !flat_grad) return -1;
  if (workspace_bytes < sqair_workspace_bytes(h, T, B) || scratch_bytes < sqair_backward_scratch_bytes(h, T, B)) {
    sq_set_error(h, "sqair_backward_decoder: workspace / scratch too small");
    return -1;
  }
  if (!sq_unit_frame_ok(h) || !sq_trainable_frame(h)) return -1;
  if (h->padded || rec::ZWP != 64) {
    sq_set_error(h, "sqair_backward_decoder (a partial adjoint kept for unit tests) writes in the product build's own shapes: use sqair_backward in the wide build or with an n_hidden that is padded");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  const float* packed = (const float*)packedv;
  const SqairConfig& c = h->cfg;
  const int nh = c.n_hidden, nw = c.n_what, N = c.n_steps_per_image, K = c.k_particles;
  const int R = B * K, M = R * N, MT = M * T, G2 = c.glimpse_size * c.glimpse_size, P_ = c.img_h * c.img_w;
  Dims d = make_dims(c, B);
  Workspace w = carve(h, T, B, (float*)workspace);
  const PackedLayout pl = packed_layout(h);
  const int RW = rec::W;
  // scratch carve
  float* sc = (float*)scratch;
  const int64_t big = (int64_t)MT * (nh > G2 ? nh : G2);
  float* g_lw = sc; sc += align64((int64_t)T * R);  // every carve 256-byte aligned (GEMM A-operand contract)
  float* g_dl = sc; sc += align64((int64_t)T * R);
  float* d_gl = sc; sc += align64((int64_t)MT * G2);
  float* d_mean_rows = sc; sc += align64((int64_t)T * R * P_);
  float* bufa = sc; sc += align64(big);
  float* bufb = sc; sc += align64(big);
  float* d_rec = sc; sc += align64((int64_t)MT * 64);
  const float* rec_all = w.rec_m_all + (size_t)M * RW;
  const float* gl = w.glimpse;  // forward wrote the decoded glimpses here unless the caller asked for the output tensor
  sq_launch_elbo_bwd(importance_weights, vimco_signal, T, B, K, g_lw, g_dl, s);
  SQ_CHECK_HIP(hipMemsetAsync(d_rec, 0, (size_t)MT * 64 * 4, s));
  if (sq_launch_insert_bwd_frames(gl, rec_all, RW, obs, flat + h->po.dec_mean_img, g_lw, d_gl, d_rec + rec::WHERE, 64, d_mean_rows,
                                  c.output_std, c.background_std, T, d, s) != 0) {
    sq_set_error(h, "the decoder canvas adjoint launch failed (dynamic LDS limit)");
    return -2;
  }
  sq_launch_reduce_rows(d_mean_rows, flat_grad + h->po.dec_mean_img, T * R, P_, 0, s);
  // ---- DEC2: glimpse = scale * (dec_b W2 + b2)
  const float* scale = flat + h->po.dec_output_scale;
  sq_launch_dot_scale(d_gl, gl, (int64_t)MT * G2, scale, flat_grad + h->po.dec_output_scale, s);
  sq_launch_wgrad(w.dec_b, nh, d_gl, G2, flat_grad + P(h, "dec.l2.w"), G2, flat_grad + P(h, "dec.l2.b"), MT, nh, G2, 0, s,
                  nullptr, scale);
  auto dx = [&](LayerId id, const float* dpre, int ld, int width, float* outp, int out_ld, const float* scale_ptr) -> int {
    const PackedLayer& LT = h->layersT[id];
    Lin l;
    l.seg(dpre, ld, width).out(outp, out_ld).act(ACT_NONE);
    l.a.scale_ptr = scale_ptr;
    l.a.wp = packed + pl.w + LT.w_off; l.a.wzero = packed + pl.w; l.a.bias = packed + pl.b + LT.b_off;
    l.a.M = MT; l.a.N = LT.N;
    return sq_launch_linear(l.a, LT, s);
  };
  if (dx(L_DEC2, d_gl, G2, G2, bufa, nh, scale) != 0) { sq_set_error(h, "backward: DEC2 dX contract"); return -5; }
  sq_launch_dact(bufa, w.dec_b, bufb, (int64_t)MT * nh, ACT_ELU, s);            // bufb = dPre of layer 1
  sq_launch_wgrad(w.dec_a, nh, bufb, nh, flat_grad + P(h, "dec.l1.w"), nh, flat_grad + P(h, "dec.l1.b"), MT, nh, nh, 0, s);
  if (dx(L_DEC1, bufb, nh, nh, bufa, nh, nullptr) != 0) { sq_set_error(h, "backward: DEC1 dX contract"); return -5; }
  sq_launch_dact(bufa, w.dec_a, bufb, (int64_t)MT * nh, ACT_ELU, s);            // bufb = dPre of layer 0
  // layer 0 reads the z-record: positions 4..53 are `what` -> rows 0..nw-1 of dec.l0.w
  {
    std::vector<int> rm(rec::ZW, -1);
    for (int i = 0; i < nw; ++i) rm[rec::WHAT + i] = i;
    int* d_rm = (int*)sc;  // 1024-float tail of the scratch
    SQ_CHECK_HIP(hipMemcpyAsync(d_rm, rm.data(), rec::ZW * 4, hipMemcpyHostToDevice, s));
    SQ_CHECK_HIP(hipStreamSynchronize(s));
    sq_launch_wgrad(rec_all, RW, bufb, nh, flat_grad + P(h, "dec.l0.w"), nh, flat_grad + P(h, "dec.l0.b"), MT, rec::ZW, nh, 0, s,
                    d_rm, nullptr);
  }
  // d what of the merged records: accumulate onto the d where the insert adjoint already wrote (disjoint columns)
  {
    const PackedLayer& LT = h->layersT[L_DEC0];
    Lin l;
    l.seg(bufb, nh, nh).out(bufa, 64).act(ACT_NONE);
    l.a.wp = packed + pl.w + LT.w_off; l.a.wzero = packed + pl.w; l.a.bias = packed + pl.b + LT.b_off;
    l.a.M = MT; l.a.N = LT.N;
    l.a.add = d_rec; l.a.add_ld = 64; l.a.add_n = 64;
    if (sq_launch_linear(l.a, LT, s) != 0) { sq_set_error(h, "backward: DEC0 dX contract"); return -5; }
    SQ_CHECK_HIP(hipMemcpyAsync(d_rec, bufa, (size_t)MT * 64 * 4, hipMemcpyDeviceToDevice, s));
  }
  if (d_rec_out) SQ_CHECK_HIP(hipMemcpyAsync(d_rec_out, d_rec, (size_t)MT * 64 * 4, hipMemcpyDeviceToDevice, s));
  SQ_CHECK_HIP(hipGetLastError());
  return 0;
}
