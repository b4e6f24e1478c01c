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
struct DxArgs {
  const float* dpre; int ld; int width;  // input gradient [M][width], 16-byte aligned rows (A-operand contract)
  const float* wp;                       // transposed pack of the layer
  const float* wzero;                    // 256 zero floats
  const float* scale_ptr;                // optional device scalar multiplied onto the product
  int M;
  int nranges;
  DxRange r[3];
};
int sq_launch_linear_dx(const DxArgs& a, int kc, int nt, hipStream_t s);
