// Arguments of the routed dX GEMM (sqair_linear_dx.hip).
#pragma once
#include <hip/hip_runtime.h>

struct DxRange {
  int n0, n1;                      // output columns [n0, n1) of the transposed layer (positions in the padded concat)
  float* dst; int dst_ld;          // destination of column n: dst[m * dst_ld + (n - n0)]
  float* dst2; int dst2_ld;        // optional second copy of the same values
  const float* add; int add_ld;    // optional addend (may alias dst: accumulate)
  const float* saved; int saved_ld;  // optional saved activation OUTPUT of the layer that produced these inputs
  int act_a, act_b, act_split;     // its activation: act_a for (n - n0) < act_split, act_b otherwise
};
// GRU gate adjoints applied in the epilogue of the dX launch that produces their input (snt.GRU: h' = (1 - z) h + z hc,
// hc = tanh(x W_h + (r h) U_h + b)): mode 1 after the d h' GEMM (g = d h': d_az = g (hc - h) z (1 - z) -> dpre1[:, 0:nh],
// d_ah = g z (1 - hc^2) -> dpre1[:, 2nh:3nh], d h (+)= g (1 - z)), mode 2 after the d(r h) GEMM (g = d(r h): d_ar = g h r (1 - r)
// -> dpre1[:, nh:2nh], d h += g r).  `dup` receives second copies (the hoisted recurrent block of the frame's pre-activation
// gradient).  Formerly two element-wise launches per GRU step (k_gru_bwd_a / _b).
struct DxGru {
  int mode;                          // 0 = none
  const float* g0; int g0_ld;        // z (mode 1) / r (mode 2)
  const float* g1; int g1_ld;        // hc (mode 1)
  const float* hprev; int h_ld;
  float* dpre1; int dp_ld;
  float* d_h; int dh_ld; int acc_dh; // mode 1: accumulate into d_h or overwrite; mode 2 always accumulates
  float* dup; int dup_ld; int dup_h_off;  // mode 1: dup[:, 0:nh] = d_az, dup[:, dup_h_off:] = d_ah if dup_h_off >= 0; mode 2: dup = d_ar
  int nh;
};
struct DxArgs {
  const float* dpre; int ld; int width;  // input gradient [M][width], 16-byte aligned rows (A-operand contract)
  const float* wp;                       // transposed pack of the layer
  const float* wzero;                    // 256 zero floats
  const float* scale_ptr;                // optional device scalar multiplied onto the product
  int M;
  int nranges;
  DxRange r[3];
  DxGru gru;
};
int sq_launch_linear_dx(const DxArgs& a, int kc, int nt, hipStream_t s);
