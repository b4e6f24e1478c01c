// Device helpers shared by the dense-layer kernels of sqair_linear.hip: per-lane addressing of the virtually concatenated
// A operand (up to four row-major segments) and the fused epilogue of one output element.
#pragma once
#include "sqair_common.h"

typedef float sq_f32x4 __attribute__((ext_vector_type(4)));

// Per-lane A-operand addressing of up to four input segments.  Deliberately NOT a struct: a select chain over the members
// of a struct returned by value is turned into an indexed load from a scratch copy of it (measured: 160 scratch
// instructions in the chain kernel), so the eleven values live in named locals declared by SQ_XSEGS.
#define SQ_XSEG1(i)                                                                                      \
  const LinSeg& xs_seg##i = a.seg[i];                                                                    \
  const int xs_row##i = xs_seg##i.rmul ? (int)__umulhi((unsigned)arow, xs_seg##i.rmul) : arow;           \
  const float* const xs_rp##i = xs_seg##i.p + (size_t)xs_row##i * xs_seg##i.ld;                          \
  const int xs_lim##i = ((xs_seg##i.width + 3) & ~3) - 4;
// constant indices into a.seg (a runtime index into a by-value kernel argument goes through scratch as well); unused
// segments are value-initialised and never dereferenced: their chunk range is empty
#define SQ_XSEGS(a, arow)                                                                                \
  SQ_XSEG1(0) SQ_XSEG1(1) SQ_XSEG1(2) SQ_XSEG1(3)                                                        \
  const int xs_c1 = (xs_seg0.width + 15) >> 4;                                                           \
  const int xs_c2 = xs_c1 + ((a).nseg > 1 ? (xs_seg1.width + 15) >> 4 : 0);                              \
  const int xs_c3 = xs_c2 + ((a).nseg > 2 ? (xs_seg2.width + 15) >> 4 : 0);                              \
  const int xs_cum1 = (a).nseg > 1 ? xs_c1 : 0x7fffffff, xs_cum2 = (a).nseg > 2 ? xs_c2 : 0x7fffffff,    \
            xs_cum3 = (a).nseg > 3 ? xs_c3 : 0x7fffffff;
#define SQ_XAPTR(g, kq) x_aptr(xs_rp0, xs_rp1, xs_rp2, xs_rp3, xs_cum1, xs_cum2, xs_cum3, xs_lim0, xs_lim1, xs_lim2, xs_lim3, (g), (kq))
__device__ __forceinline__ const float* x_aptr(const float* rp0, const float* rp1, const float* rp2, const float* rp3, int cum1, int cum2,
                                               int cum3, int lim0, int lim1, int lim2, int lim3, int g, int kq) {
  const bool s1 = g >= cum1, s2 = g >= cum2, s3 = g >= cum3;
  const float* rp = s3 ? rp3 : (s2 ? rp2 : (s1 ? rp1 : rp0));
  const int cb = s3 ? cum3 : (s2 ? cum2 : (s1 ? cum1 : 0));
  const int lim = s3 ? lim3 : (s2 ? lim2 : (s1 ? lim1 : lim0));
  return rp + min((g - cb) * 16 + kq * 4, lim);
}
// epilogue of one output element (m, n) with pre-activation sum v (bias and addend already included)
__device__ __forceinline__ void x_epilogue(const LinArgs& a, int m, int n, float v, float p_e0, float p_e1, float p_scale) {
  if (a.epi == EPI_ACT) {
    v = sq_act(v, n < a.act_split ? a.act_a : a.act_b);
    a.out[(size_t)m * a.out_ld + n] = v * a.scale * p_scale;
  } else if (a.epi == EPI_GRU1) {
    const int nh = a.nh;
    if (n < nh) a.out[(size_t)m * a.out_ld + n] = sq_sigmoid(v);
    else if (n < 2 * nh) {
      const float rg = sq_sigmoid(v);
      a.o1[(size_t)m * a.o1_ld + (n - nh)] = rg * p_e0;
      if (a.o3 != nullptr) a.o3[(size_t)m * a.o3_ld + (n - nh)] = rg;
    } else a.o2[(size_t)m * a.o2_ld + (n - 2 * nh)] = v;
  } else {
    const float hc = sq_tanh(v);
    a.out[(size_t)m * a.out_ld + n] = (1.0f - p_e1) * p_e0 + p_e1 * hc;
    if (a.o1 != nullptr) a.o1[(size_t)m * a.o1_ld + n] = hc;
  }
}

