// Internal declarations shared by sqair_api.hip (handle, plan, forward pass) and sqair_train.hip (backward pass).
#pragma once
#include <cstdio>
#include <cstring>
#include <map>

#include "sqair_glue.h"

struct ParamEntry {
  std::string name;
  int64_t off, numel;
  int rows, cols;
};

enum LayerId {
  L_IENC0, L_IENC1, L_PREDISC, L_PRIOR_GRU1, L_PRIOR_GRU2, L_PRIOR_LIN, L_TAU1, L_WB2, L_MASK2, L_GENC0, L_GENC1,
  L_WHAT_LOC, L_WHAT_HEAD, L_PRE, L_PROP_RNN, L_PROP_T1, L_PROP_T2, L_PROP_T3, L_PROP_GRU1, L_PROP_GRU2,
  L_PROP_HEADS, L_PROP_S1, L_LAT0, L_LAT1, L_PRED, L_RNCOND, L_DISC_RNN, L_DISC_T1, L_DISC_T2, L_DISC_T3,
  L_DISC_S1, L_DEC0, L_DEC1, L_DEC2, L_PROP_RNN2, L_DISC_RNN2,
  L_WHAT_HEAD_I, L_PROP_HEADS_I,   // forward-only packs with interleaved output columns (what fusion, sqair_glue.h: WhatArgs)
  L_COUNT
};

struct SqairHandle {
  // cfg is what the kernels run on: n_hidden rounded up to a width the row kernels are written for (a multiple of 128).  The extra
  // hidden units are inert by construction -- zero weights and biases in, zero weights out, zero initial states: ELU(0) = tanh(0)
  // = 0, a GRU / LSTM unit that starts at 0 with zero pre-activations stays at 0 -- so results equal the unpadded model's.  ucfg
  // is the caller's configuration.  `params` is the inventory in the PADDED shapes (everything downstream -- packing plan, POff,
  // the kernels that read small layers straight from the flat buffer -- sees only these); `uparams` is the caller's inventory
  // (reference shapes, what crosses the ABI) and u2i maps each of its elements to its place in the padded flat buffer.  When
  // nothing is padded (`padded` false: n_hidden 128 / 256 / ...) the two coincide and the caller's buffers are used as they are.
  SqairConfig cfg;
  SqairConfig ucfg;
  std::string err;
  std::vector<ParamEntry> params;
  std::map<std::string, int> pidx;
  int64_t n_params = 0;
  std::vector<ParamEntry> uparams;
  int64_t n_uparams = 0;
  std::vector<int> u2i;
  bool padded = false;
  POff po;
  // packing plan
  PackedLayer layers[L_COUNT];
  PackedLayer layersT[L_COUNT];     // transposed packs (dX = dY W^T), K' = padded N, N' = padded concat K
  // weight-gradient plan: per layer, one entry per (column block, segment) with the A-position -> matrix-row map
  struct WgEntry { int n0, ncols, col0, seg; std::string w; int64_t rm_off; };
  struct BgEntry { int n0, ncols, col0; std::string a, b; };
  std::vector<WgEntry> wg[L_COUNT];
  std::vector<BgEntry> bg[L_COUNT];
  std::vector<int> rm_pool;
  std::vector<int> widx;            // per packed weight element: index into flat params or -1
  std::vector<int> bidx_a, bidx_b;  // per packed bias element
  int64_t packed_w = 0, packed_b = 0;
  bool plan_uploaded_to = false;
  const void* plan_uploaded_ptr = nullptr;
  // graph
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int graph_nodes = 0;
  int debug_reps = 0;       // sqair_debug_linear_time
  int debug_graph_nodes = 0;           // sqair_debug_linear_graph_time: launches per captured graph
  bool debug_graph_dependent = false;  // ... chained through two buffers (K == N) instead of repeated
  bool dense_log_on = false;           // sqair_debug_dense_log: {layer id, rows, K (padded to 16 per segment), N} of every dense launch
  std::vector<int> dense_log;          // of the passes issued while it was on (host side only: nothing changes on the device)
  float debug_us = 0.0f;
  bool opt_what_fusion = true;  // sqair_set_option("what_fusion"): the what sample of a slot inside the layer that produces its operands (bit-identical)
  bool opt_tail_fusion = true;  // sqair_set_option("tail_fusion"): the tail of slot k inside slot k + 1's RNN launch (bit-identical either way)
  bool opt_slot_chain = false;  // sqair_set_option("slot_chain"): the slot launches of a frame's propagation / discovery loop as one
                                // persistent launch each (sqair_chain.h; bit-identical; set BEFORE sizing / clearing workspaces)
  int opt_slot_chain_mode = 0;
  int opt_vi_target = 0;        // sqair_set_option("vi_target"): learning signal of sqair_elbo, 0 = VIMCO, 1 = plain REINFORCE
  void* chain = nullptr;        // ChainState (sqair_chain.hip)
  bool clear_each_pass = true;  // zero the caller's workspace at the start of every pass (sqair_set_workspace_clearing)
  const float* gen_noise = nullptr;  // sqair_set_generation_noise
  // generic capture slots (sqair_capture_begin / _end / _launch): any sequence of C-ABI calls as one HIP graph
  hipGraph_t cap_graph[4] = {nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t cap_exec[4] = {nullptr, nullptr, nullptr, nullptr};
};


inline int64_t align64(int64_t x) { return (x + 63) / 64 * 64; }
constexpr int64_t SQ_TRAIN_MAX_FRAME_BYTES = 150 * 1024;   // dynamic LDS the crop adjoint may ask for (sq_allow_big_lds)
bool sq_trainable_frame(SqairHandle* h);                   // false + error text when the handle's frames cannot be trained
// The partial adjoint kept for unit tests (sqair_backward_decoder) takes the caller's frames as they are, and the full adjoint
// kernels stage frames in 16-byte units: it wants H * W to be a multiple of 4 (the full passes stage other frames through a
// padded copy; the per-kernel entry points sqair_st_*_bwd read frames word by word and take any size).
inline bool sq_unit_frame_ok(SqairHandle* h) {
  if (((h->cfg.img_h * h->cfg.img_w) & 3) == 0) return true;
  sq_set_error(h, "this unit entry point needs H * W to be a multiple of 4 (use sqair_forward_train / sqair_backward for other frame sizes)");
  return false;
}
int64_t P(const SqairHandle* h, const std::string& name);  // flat offset of a parameter (aborts on unknown names)
int PC(const SqairHandle* h, const std::string& name);      // its number of columns

// packed buffer = [weights fp32 | biases fp32 | widx int32 | bidx_a | bidx_b | row maps | (padded configurations only:) the flat
// parameters in the padded shapes fp32 | u2i int32], each 256-byte aligned
struct PackedLayout {
  int64_t w, b, wi, ba, bb, rm, fi, um, total;  // offsets in 4-byte words
};
inline PackedLayout packed_layout(const SqairHandle* h) {
  PackedLayout p;
  p.w = 0;
  p.b = align64(p.w + h->packed_w);
  p.wi = align64(p.b + h->packed_b);
  p.ba = align64(p.wi + h->packed_w);
  p.bb = align64(p.ba + h->packed_b);
  p.rm = align64(p.bb + h->packed_b);
  p.fi = align64(p.rm + (int64_t)h->rm_pool.size());
  p.um = align64(p.fi + (h->padded ? h->n_params : 0));
  p.total = align64(p.um + (h->padded ? h->n_uparams : 0));
  return p;
}

struct Lin {
  LinArgs a;
  Lin() {
    memset(&a, 0, sizeof(a));
    a.epi = EPI_ACT;
    a.act_split = 1 << 30;
    a.add_rdiv = 1;
    a.scale = 1.0f;
  }
  Lin& seg(const float* p, int ld, int width, int rdiv = 1) {
    a.seg[a.nseg++] = LinSeg{p, ld, width, rdiv};
    return *this;
  }
  Lin& out(float* p, int ld) { a.out = p; a.out_ld = ld; return *this; }
  Lin& act(int act) { a.act_a = act; return *this; }
  Lin& act2(int a0, int a1, int split) { a.act_a = a0; a.act_b = a1; a.act_split = split; return *this; }
  Lin& add(const float* p, int ld, int n, int rdiv = 1) { a.add = p; a.add_ld = ld; a.add_n = n; a.add_rdiv = rdiv; return *this; }
  Lin& gru1(const float* hprev, int h_ld, float* rh, int rh_ld, float* xh, int xh_ld, int nh) {
    a.epi = EPI_GRU1; a.e0 = hprev; a.e0_ld = h_ld; a.o1 = rh; a.o1_ld = rh_ld; a.o2 = xh; a.o2_ld = xh_ld; a.nh = nh;
    return *this;
  }
  Lin& gru2(const float* hprev, int h_ld, const float* z, int z_ld, int nh) {
    a.epi = EPI_GRU2; a.e0 = hprev; a.e0_ld = h_ld; a.e1 = z; a.e1_ld = z_ld; a.nh = nh;
    return *this;
  }
};



// ------------------------------------------------------------------------------------------------
// workspace.  In training mode (train = true) every intermediate the backward pass needs is kept: per-frame
// buffers become [T][...], per-slot buffers a "tape" [2 phases][T][B'][N slots][width] (slot-inner like the slot
// records, so a slot launch addresses it with row stride N * width and the batched weight-gradient GEMMs see all
// uses of a layer as one long row range).
// ------------------------------------------------------------------------------------------------
constexpr int PROF_MAX = 4096;
// leading dimensions of the activation buffers: room for the widest layer the build accepts (prior statistics 2 (4 + n_what) + 1,
// glimpse-encoder Gaussian 2 n_what, loc1 n_what, raw heads 5 n_what, transform hidden | steps hidden 1.5 n_hidden, steps hidden
// n_hidden / 2), each a multiple of 16
#ifdef SQAIR_WIDE
constexpr int PS_LD = 272, ENC_LD = 256, M1_LD = 128, HRAW_LD = 640, WB_LD = 4, TP_LD = 8, T1_LD = 768, S1_LD = 256;
#else
constexpr int PS_LD = 112, ENC_LD = 112, M1_LD = 64, HRAW_LD = 256, WB_LD = 4, TP_LD = 8, T1_LD = 384, S1_LD = 128;
#endif

struct Workspace {
  bool train;
  bool tape;    // per-frame / per-slot buffers are kept apart ([T] / [2 phases][T][B'][N]): training, or the in-launch slot chain
  bool chain;   // the slot loops run as chain launches (sqair_chain.h)
  unsigned* chain_ctl;   // control blocks of the pass's chain launches
  int T, B, R, M, N, nh, snh, psnh;
  float *ienc_a, *ienc_b, *pre_disc;
  float *rec_m_all, *rec_p_all, *rec_d_all;
  float *temporal_m, *prior_m;                   // train: [T+1][M][snh | nh], else [2][M][snh | nh]
  float *last_id[2];
  float *zero_rec, *disc_init_rec, *prop_rnn_init, *disc_rnn_init, *rn_init_state, *w3_prop, *w3_disc;
  float *temporal_p, *prior_p;                   // per frame
  float *pgz, *pgr, *pghc, *pgrh, *pgxh;         // prior GRU internals, per frame
  float *pstats, *spre;                          // always [T]
  float *hid1, *wb, *mask, *g1, *pea, *peb, *m1, *pre, *lea, *leb, *c, *pre_d;  // per frame
  float *r, *t1, *t2, *tp, *g2, *e1, *e2, *enc, *hraw, *s1h, *gz, *gr, *ghc, *grh, *gxh;  // per slot
  float *rc, *rgates;                            // LSTM / GRU slot RNN: cell states (like r), kept gates [.][4nh | 3nh]
  float *lpre, *lgates;                          // LSTM temporal cell (time_lstm): [M][4nh], kept gates [T][R][N][4nh]
  int* src;                                      // train: compaction source slot [T][R][N]
  float *qz, *pz, *dlp, *dll, *glimpse, *dec_a, *dec_b;
  float* gen;                                    // sample_from_prior: [T][M][64] prior samples + original presences
  float* obs_p;                                  // frames whose H * W is not a multiple of 4: zero-padded copy [T*B][P4] (else unused)
  unsigned long long* prof_ts;
  int64_t total;  // floats

  float* frame(float* base, int64_t per_frame, int t) const { return base + (tape ? (size_t)t * per_frame : 0); }
  float* state(float* base, int t, int width) const { return base + (size_t)(tape ? t : (t & 1)) * M * width; }
  // slot buffer of width W: pointer of (frame t, phase ph, slot k) and its row stride
  float* slot(float* base, int W, int t, int ph, int k) const {
    return base + (tape ? (((size_t)(ph * T + t) * R * N) + k) * W : 0);  // [phase][T][B'][N][W]
  }
  int sld(int W) const { return tape ? N * W : W; }
  float* cslot(int t, int ph, int k) const { return tape ? slot(rc, nh, t, ph, k) : rc + (size_t)(k & 1) * R * nh; }
  float* rslot(int t, int ph, int k) const {  // RNN hidden state: ping-pong over slots when no tape is kept
    return tape ? slot(r, nh, t, ph, k) : r + (size_t)(k & 1) * R * nh;
  }
};
Workspace sq_carve(const SqairHandle* h, int T, int B, float* base, bool train);

// zero fill as a kernel (not a memset node: see sqair_train.hip); p 16-byte aligned
void sq_zero_fill(float* p, int64_t n, hipStream_t s);
void sq_copy(float* dst, const float* src, int64_t n, hipStream_t s);

// the flat parameter buffer the kernels read: the caller's, or (padded configurations) the padded copy sqair_pack_params keeps
// inside the packed buffer
inline const float* sq_flat(const SqairHandle* h, const float* user_flat, const void* packed) {
  return h->padded ? (const float*)packed + packed_layout(h).fi : user_flat;
}
void sq_flat_gather(const SqairHandle* h, const float* padded_grad, float* user_grad, const void* packed, hipStream_t s);
int sq_run(SqairHandle* h, Lin& l, LayerId id, int M, const float* packed, hipStream_t s);
#define RUN(l, id, M)                                   \
  do {                                                  \
    int _rc = sq_run(h, (l), (id), (M), packed, s);     \
    if (_rc != 0) return _rc;                           \
  } while (0)
