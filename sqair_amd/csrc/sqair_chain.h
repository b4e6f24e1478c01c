// Layer chains (sqair_chain.hip): several dependent dense layers of a slot in one launch, rows split over per-XCD teams.
#pragma once
#include "sqair_common.h"

enum { SQ_CHAIN_OK = 0, SQ_CHAIN_TIMEOUT = 1, SQ_CHAIN_PLACEMENT = 2 };  // status word of a pass

struct ChainLayer {
  LinArgs a;
  int kc, nt;
};
template <int L>
struct ChainArgs {
  ChainLayer l[L];
  int R;                        // rows of every layer
  unsigned* bar;                // this launch's arrival counters: 8 teams x 64 words (256 B apart), zero on entry
  int* status;                  // the pass's status word (raised, never cleared, by the kernels)
  unsigned long long* prof_ts;  // optional {min start, max end} device-clock slot (as k_linear)
};

// 256 workgroups (one per CU of the MI355X); layers[i] prepared like sq_launch_linear's argument (wp / bias / wzero / M / N set)
int sq_launch_chain(const LinArgs* layers, const PackedLayer* const* packed_layers, int n_layers, int R, unsigned* bar, int* status,
                    unsigned long long* prof_ts, hipStream_t s);
constexpr int SQ_CHAIN_BAR_WORDS = 8 * 64;
