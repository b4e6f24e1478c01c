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
    a.out[(size_t)m * a.out_ld + n] = sq_gru_blend(p_e1, p_e0, hc);
    if (a.o1 != nullptr) a.o1[(size_t)m * a.o1_ld + n] = hc;
  }
}


// Epilogue of FOUR consecutive output elements (m, n0 .. n0 + 3), n0 a multiple of 4, with pre-activation sums v (bias NOT yet
// included: `bias4` = the four biases, one load per lane for the whole tile).  Vector (16-byte) loads and stores wherever the
// four elements lie inside the tensor and its leading dimension keeps rows 16-byte aligned (`vec`: checked once on the host
// for every pointer of the launch); element-wise otherwise (ragged N such as the prior layer's 109 columns, the last float4 of
// the addend).  Same arithmetic per element as x_epilogue.
__device__ __forceinline__ void x_epilogue4(const LinArgs& a, int m, int n0, sq_f32x4 v, const sq_f32x4 bias4, float p_scale, bool vec) {
  float x[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
  const bool full = vec && n0 + 3 < a.N;
  if (a.add != nullptr && n0 < a.add_n) {
    const float* ap = a.add + (size_t)(a.add_rmul ? (int)__umulhi((unsigned)m, a.add_rmul) : m) * a.add_ld + n0;
    if (vec && n0 + 3 < a.add_n) {
      const sq_f32x4 av = *reinterpret_cast<const sq_f32x4*>(ap);
      x[0] += av.x; x[1] += av.y; x[2] += av.z; x[3] += av.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (n0 + j < a.add_n) x[j] += ap[j];
    }
  }
  if (a.epi == EPI_ACT) {
    float* op = a.out + (size_t)m * a.out_ld + n0;
    const float sc = a.scale * p_scale;
    if (a.act_split >= a.N) {
      // one activation for the whole layer (every layer but the two with a Gaussian head): a SCALAR branch on the kernel
      // argument, one case executed.  A per-lane activation code turns the switch into all five cases under lane masks
      // (measured: 0.5 us per float4 of the epilogue, 8.4 of the 31 us of a 128 x 128 x 256 tile)
      switch (a.act_a) {
        case ACT_ELU: _Pragma("unroll") for (int j = 0; j < 4; ++j) x[j] = sq_elu(x[j]); break;
        case ACT_TANH: _Pragma("unroll") for (int j = 0; j < 4; ++j) x[j] = sq_tanh(x[j]); break;
        case ACT_SIGMOID: _Pragma("unroll") for (int j = 0; j < 4; ++j) x[j] = sq_sigmoid(x[j]); break;
        case ACT_SOFTPLUS_MIN: _Pragma("unroll") for (int j = 0; j < 4; ++j) x[j] = sq_softplus(x[j]) + 1e-2f; break;
        default: break;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] *= sc;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = sq_act(x[j], n0 + j < a.act_split ? a.act_a : a.act_b) * sc;
    }
    if (full) *reinterpret_cast<sq_f32x4*>(op) = sq_f32x4{x[0], x[1], x[2], x[3]};
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (n0 + j < a.N) op[j] = x[j];
    }
  } else if (a.epi == EPI_GRU1) {   // nh is a multiple of 64: the four elements lie in one of the three gate blocks
    const int nh = a.nh;
    if (n0 < nh) {
      *reinterpret_cast<sq_f32x4*>(a.out + (size_t)m * a.out_ld + n0) = sq_f32x4{sq_sigmoid(x[0]), sq_sigmoid(x[1]), sq_sigmoid(x[2]), sq_sigmoid(x[3])};
    } else if (n0 < 2 * nh) {
      const sq_f32x4 h = *reinterpret_cast<const sq_f32x4*>(a.e0 + (size_t)m * a.e0_ld + (n0 - nh));
      const sq_f32x4 rg = sq_f32x4{sq_sigmoid(x[0]), sq_sigmoid(x[1]), sq_sigmoid(x[2]), sq_sigmoid(x[3])};
      *reinterpret_cast<sq_f32x4*>(a.o1 + (size_t)m * a.o1_ld + (n0 - nh)) = rg * h;
      if (a.o3 != nullptr) *reinterpret_cast<sq_f32x4*>(a.o3 + (size_t)m * a.o3_ld + (n0 - nh)) = rg;
    } else {
      *reinterpret_cast<sq_f32x4*>(a.o2 + (size_t)m * a.o2_ld + (n0 - 2 * nh)) = sq_f32x4{x[0], x[1], x[2], x[3]};
    }
  } else {
    const sq_f32x4 h = *reinterpret_cast<const sq_f32x4*>(a.e0 + (size_t)m * a.e0_ld + n0);
    const sq_f32x4 z = *reinterpret_cast<const sq_f32x4*>(a.e1 + (size_t)m * a.e1_ld + n0);
    const sq_f32x4 hc = sq_f32x4{sq_tanh(x[0]), sq_tanh(x[1]), sq_tanh(x[2]), sq_tanh(x[3])};
    *reinterpret_cast<sq_f32x4*>(a.out + (size_t)m * a.out_ld + n0) =
        sq_f32x4{sq_gru_blend(z.x, h.x, hc.x), sq_gru_blend(z.y, h.y, hc.y), sq_gru_blend(z.z, h.z, hc.z), sq_gru_blend(z.w, h.w, hc.w)};
    if (a.o1 != nullptr) *reinterpret_cast<sq_f32x4*>(a.o1 + (size_t)m * a.o1_ld + n0) = hc;
  }
}
