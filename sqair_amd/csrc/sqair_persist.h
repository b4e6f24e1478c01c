// Op list of the XCD-persistent executor (sqair_persist.hip): the frame-loop launches of sq_forward_impl, recorded.
#pragma once
#include <cstring>

#include "sqair_glue.h"

enum XOpType { XOP_LINEAR = 1, XOP_CROP = 2, XOP_TAIL = 3, XOP_LATSUM = 4, XOP_COMPACT = 5 };
struct XLatArgs { const float* f; const float* rec_p; float* c; };
struct XOp {
  int type;
  int kc, nt;        // XOP_LINEAR: K-chunks / N-tiles of the packed layer
  int nslots;        // XOP_CROP: virtual blocks per row (N for the batched crop #1, else 1)
  int sync_after;    // team barrier after the op
  int pad[3];
  union U {
    LinArgs lin;
    CropArgs crop;
    TailArgs tail;
    XLatArgs lat;
    CompactArgs comp;
    U() {}
  } u;
  XOp() { memset(this, 0, sizeof(*this)); }
};
constexpr int XSYNC_WORDS = 64 + 16 * 32;  // [16 team counts | arrivals | abort | ...pad to 64] + one 128-byte line per team

int sq_launch_persistent(const XOp* prog_dev, int n_ops, POff po, Dims d, unsigned* sync, int n_cu, hipStream_t s,
                         unsigned long long* tstamp = nullptr);
