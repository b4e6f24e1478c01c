// In-launch slot chain (round 5): the strictly sequential per-slot launches of a frame's propagation / discovery loop
// (sqair/core.py:164-359 via propagate.py:168-184 and sqair_modules.py:129-147) executed as ONE persistent launch per chain.
//
// Hand-off: rows never interact, so a 16-row tile's next layer needs only the column tiles of the SAME row tile.  Activations
// stay plain row-major fp32 where the launch-per-layer path keeps them; every hand-off word is pre-filled with a SENTINEL
// (0xFFFFFFFF, a NaN payload no arithmetic of the pass produces) and the consumer polls its own operand loads (L1-bypassing
// `sc1` buffer loads, served by the XCD's L2) until no word is the sentinel: the data is its own flag, word by word.
// Placement: nothing is assumed about it.  Every workgroup reads HW_REG_XCC_ID, counts itself on THAT XCD's counter and waits
// until the whole grid has been counted (a census: the only requirement is that the 256 workgroups are co-resident); the row
// tiles are then dealt to the XCDs that did get workgroups and each XCD's items to its workgroups by rank -- a static schedule
// every workgroup derives alone.  A row tile's producers and consumers share one L2 by construction.  (Tickets pulled per item
// from per-XCD queues need not even co-residency, and were the first version: the returning atomic and the claim protocol on
// every item's critical path cost more than the census, DESIGN.md.)  Measured first in tools/sentinel_chain.hip.
#pragma once
#include "sqair_glue.h"

constexpr unsigned SQ_SENT = 0xFFFFFFFFu;
enum { COP_DENSE = 0, COP_CROP = 1, COP_RNN_TAIL = 2, COP_TAIL = 3 };
constexpr int SQ_CHAIN_MAX_OPS = 160;
constexpr int SQ_CHAIN_CTL_WORDS = 512;       // control block of one chain launch (zeroed once per pass)
constexpr int SQ_CHAIN_MAX_LAUNCHES = 128;    // control blocks per pass (2 per frame)
constexpr int SQ_CHAIN_SPIN_LIMIT = 1 << 14;  // polls before a consumer gives up and flags the launch (never hang the device)

// the VanillaRNN layer of a slot with the previous slot's tail computed in front of it (k_rnn_tail's arguments)
struct ChainRnn {
  TailArgs ta;
  const float* hid; int hid_ld;
  const float* wp; const float* bias;
  const float* add; int add_ld;
  float* out; int out_ld; int n_out;
};
// A dense layer as the chain's body wants it: everything the launch path derives per launch from LinArgs is derived once on the
// host, every operand inside the workspace is a 32-bit byte offset from its base (one buffer resource, no 64-bit address
// arithmetic on the device), and the virtually concatenated A operand is a table of K chunks.
struct alignas(16) ChChunk { unsigned base, ldb, lim, pad; };   // chunk g of a row: bytes [base + row * ldb + min(16 kq, lim), +16)
constexpr int SQ_CHAIN_MAX_KC = 28;
constexpr unsigned SQ_CHAIN_NONE = 0xFFFFFFFFu;
struct alignas(16) ChDense {     // (16-byte groups: the body fetches the descriptor from LDS with a handful of 128-bit reads)
  int kc_total, nch, M, N;
  int epi, act_a, act_b, act_split;
  int nh, add_n; unsigned add_off; int add_ld;     // add_off = SQ_CHAIN_NONE: no addend
  unsigned wp_lo, wp_hi, wz_lo, wz_hi;             // packed weights / zero block (parameter buffer)
  unsigned bias_lo, bias_hi, out_off; int out_ld;  // packed bias
  unsigned e0_off; int e0_ld; unsigned e1_off; int e1_ld;
  unsigned o1_off; int o1_ld; unsigned o2_off; int o2_ld;   // (SQ_CHAIN_NONE: not kept)
  unsigned o3_off; int o3_ld; float scale; int pad;
  ChChunk chunk[SQ_CHAIN_MAX_KC];
};
struct alignas(16) ChainOp {
  int kind;      // COP_*
  int items;     // dense: column tiles; RNN + tail: column tiles; crop: the 16 rows of a row tile; tail: 1
  int nch;       // dense: K chunks per wave (which instantiation)
  int pad1;
  union {
    ChDense dense;
    CropArgs crop;
    ChainRnn rnn;
    TailArgs tail;
  } u;
};
struct alignas(16) ChainTable {
  int n_ops, n_row_tiles;
  int launch_id;            // ordinal of the chain launch inside its pass (2 t + phase)
  int pad0;
  int staged;               // crop: frame staged in LDS
  int lds_scratch_floats;   // LDS floats ahead of the table's copy (the ops' scratch)
  int pad1, pad2;
  Dims d;
  POff po;
  alignas(16) ChainOp ops[SQ_CHAIN_MAX_OPS];
};

// a strided range of hand-off words to fill with the sentinel before a pass
struct ChainPoison { float* base; int64_t rows; int64_t row_stride; int width; int pad; };
constexpr int SQ_CHAIN_MAX_POISON = 32;
struct ChainPoisonList { int n; int pad; ChainPoison r[SQ_CHAIN_MAX_POISON]; };

struct SqairHandle;
// host recorder (sqair_chain.hip): between sq_chain_begin and sq_chain_flush the slot loop's launches are collected
bool sq_chain_active(const SqairHandle* h);
void sq_chain_reset(SqairHandle* h);   // closes a recorder an earlier pass left open (early return between begin and flush)
void sq_chain_begin(SqairHandle* h, const Dims& d, const POff& po, const float* ws_base, int64_t ws_bytes);
int sq_chain_add_dense(SqairHandle* h, const LinArgs& a, int kc_total, int n_tiles);
int sq_chain_add_crop(SqairHandle* h, const CropArgs& a);
int sq_chain_add_rnn_tail(SqairHandle* h, const ChainRnn& r);
int sq_chain_add_tail(SqairHandle* h, const TailArgs& a);
// uploads the table (cached by content) and launches the chain; `ctl` = this launch's zeroed control block
int sq_chain_flush(SqairHandle* h, unsigned* ctl, int launch_id, hipStream_t s);
int sq_chain_poison(const ChainPoisonList& pl, unsigned* ctl_all, int ctl_words, hipStream_t s);
void sq_chain_destroy(SqairHandle* h);
int sq_chain_set_arena_kb(SqairHandle* h, int kb);
