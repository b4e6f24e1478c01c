// Per-row spatial-transformer crop, shared by the launch-per-op path (k_crop_row, plain loads) and the XCD-persistent
// executor (L1-bypassing loads of data written earlier in the same launch).  Arithmetic order of the original
// per-sequence k_crop (sqair_glue.hip): results are bit-identical.
#pragma once
#include "sqair_glue.h"

typedef float sq_f32x4 __attribute__((ext_vector_type(4)));
struct LdPlain {
  static __device__ __forceinline__ float f(const float* p) { return *p; }
  static __device__ __forceinline__ sq_f32x4 f4(const float* p) { return *reinterpret_cast<const sq_f32x4*>(p); }
  static __device__ __forceinline__ void f4x4(const float* p0, const float* p1, const float* p2, const float* p3, sq_f32x4& v0,
                                              sq_f32x4& v1, sq_f32x4& v2, sq_f32x4& v3) {
    v0 = f4(p0); v1 = f4(p1); v2 = f4(p2); v3 = f4(p3);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads above the MFMAs (see sqair_linear_kernel.inc)
  }
};

// ---- dense-layer tile shared by the grouped launch (sqair_linear.hip) and the persistent executor -----------------------
// operand addressing shared by both tilings
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
    const float hc = tanhf(v);
    a.out[(size_t)m * a.out_ld + n] = (1.0f - p_e1) * p_e0 + p_e1 * hc;
    if (a.o1 != nullptr) a.o1[(size_t)m * a.o1_ld + n] = hc;
  }
}

// One 16x16 output tile by one workgroup: 4 waves split K, LDS reduce (arithmetic order of k_linear).  LD selects how
// activations are loaded (plain for ordinary launches, L1-bypassing inside the persistent executor).
template <class LD>
__device__ void x_linear_tile(const LinArgs& a, int kc_total, int tile_n, int mbase, int m1, float* red) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int arow = min(mbase + (lane & 15), m1 - 1);
  const int m = mbase + (tid >> 4);
  const int n = tile_n * 16 + (tid & 15);
  const bool live = m < m1 && n < a.N;
  const int mc = min(m, m1 - 1), nc = min(n, a.N - 1);
  const float* pb = a.bias + nc;
  const bool use_add = a.add != nullptr && nc < a.add_n;
  const bool g1 = a.epi == EPI_GRU1 && nc >= a.nh && nc < 2 * a.nh;
  const bool g2 = a.epi == EPI_GRU2;
  const int mcd = a.add_rmul ? (int)__umulhi((unsigned)mc, a.add_rmul) : mc;
  const float* pa = use_add ? a.add + (size_t)mcd * a.add_ld + nc : pb;
  const float* pe0 = g1 ? a.e0 + (size_t)mc * a.e0_ld + (nc - a.nh) : (g2 ? a.e0 + (size_t)mc * a.e0_ld + nc : pb);
  const float* pe1 = g2 ? a.e1 + (size_t)mc * a.e1_ld + nc : pb;
  const float p_bias = *pb;
  float p_add = LD::f(pa);
  const float p_e0 = LD::f(pe0), p_e1 = LD::f(pe1);
  const float p_scale = a.scale_ptr != nullptr ? *a.scale_ptr : 1.0f;
  p_add = use_add ? p_add : 0.0f;
  SQ_XSEGS(a, arow)
  sq_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  const sq_f32x4* __restrict__ wp = reinterpret_cast<const sq_f32x4*>(a.wp) + ((size_t)tile_n * kc_total) * 64 + lane;
  const sq_f32x4* __restrict__ wz = reinterpret_cast<const sq_f32x4*>(a.wzero) + lane;
  const int nmine = (kc_total - wave + 3) >> 2;
  constexpr int NCH = 4;
#pragma unroll 1
  for (int base = 0; base < nmine; base += NCH) {
    sq_f32x4 av[NCH], bv[NCH];
    const float* ap[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const bool valid = base + j < nmine;
      const int g = valid ? wave + 4 * (base + j) : wave;
      ap[j] = SQ_XAPTR(g, kq);
      bv[j] = *(valid ? wp + (size_t)g * 64 : wz);
    }
    LD::f4x4(ap[0], ap[1], ap[2], ap[3], av[0], av[1], av[2], av[3]);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc1, 0, 0, 0);
    }
  }
  float* r = red + wave * 256;
  r[(4 * kq + 0) * 16 + (lane & 15)] = acc0.x + acc1.x;
  r[(4 * kq + 1) * 16 + (lane & 15)] = acc0.y + acc1.y;
  r[(4 * kq + 2) * 16 + (lane & 15)] = acc0.z + acc1.z;
  r[(4 * kq + 3) * 16 + (lane & 15)] = acc0.w + acc1.w;
  __syncthreads();
  if (live) x_epilogue(a, m, n, red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid] + p_bias + p_add, p_e0, p_e1, p_scale);
  __syncthreads();
}


template <class LD>
__device__ void x_crop_row(const CropArgs& a, const POff& po, const Dims& d, int r, int slot, float* smem, bool stage_img) {
  float* coord_s = smem;        // 4
  float* tab_s = smem + 4;      // 2 * 2G
  float* img_s = smem + 4 + 4 * d.G;  // H*W when stage_img: the frame is pulled into LDS WHILE wave 0 computes `where`, so
                                      // the gather below does not start a second memory round trip after it
  const int tid = threadIdx.x, b = r / d.K;
  const int P = d.H * d.W, G = d.G, G2 = d.G * d.G;
  const int mrow_add = a.mask_row_add + (a.mode == CROP_PROP1 ? slot : 0);
  const int orow_add = a.out_row_add + (a.mode == CROP_PROP1 ? slot : 0);
  const float* __restrict__ img = a.img + (size_t)b * P;
  const bool has_mask = a.mask != nullptr;
  const bool fused_tp = a.t2 != nullptr && (a.mode == CROP_PROP2 || a.mode == CROP_DISC);
  constexpr int IPT = 10;  // first 2560 frame pixels: requested before anything else, stored after the where computation
  float v0[IPT];
  if (stage_img) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) v0[q] = img[min(q * 256 + tid, P - 1)];
  }
  constexpr int MPT = 2;  // mask values of this thread's first pixels, requested up front as well
  float mk0[MPT];
#pragma unroll
  for (int q = 0; q < MPT; ++q)
    mk0[q] = has_mask ? LD::f(a.mask + ((size_t)r * a.mask_row_mul + mrow_add) * G2 + min(tid + 256 * q, G2 - 1)) : 1.0f;
  if (tid < 32) {
    const int hl = tid, ci = hl & 3;
    const int per = d.nh / 32;
    float tp_loc = 0.0f, tp_raw = 0.0f;
    float e[4] = {0.0f, 0.0f, 0.0f, 0.0f}, zp = 0.0f, off = 0.0f, chv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, wbv = 0.0f, lg = 0.0f;
    if (a.mode == CROP_PLAIN) {
      lg = LD::f(a.logits + (size_t)r * 4 + ci);
    } else if (a.mode == CROP_PROP1) {
      zp = LD::f(a.rec_prev + ((size_t)r * d.N + slot) * rec::W + rec::WHERE + ci);
      wbv = LD::f(a.wb + ((size_t)r * d.N + slot) * a.wb_ld + ci);
    } else {
      const float* eps = a.noise + (((size_t)r * 2 + (a.mode == CROP_DISC ? 1 : 0)) * d.N + slot) * d.nzw;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) e[jj] = eps[jj];
      if (a.mode == CROP_DISC) {
        off = a.flat[po.disc_scale_offset];
      } else {
        off = a.flat[po.prop_scale_offset];
        zp = LD::f(a.rec_prev + ((size_t)r * d.N + slot) * rec::W + rec::WHERE + ci);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) chv[jj] = tril4(a.flat + po.cholesky, ci, min(jj, ci));
      }
      if (!fused_tp) {
        tp_loc = LD::f(a.tp + (size_t)r * a.tp_ld + ci);
        tp_raw = LD::f(a.tp + (size_t)r * a.tp_ld + 4 + ci);
      }
    }
    if (fused_tp) {
      float part[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      const float* xrow = a.t2 + (size_t)r * a.t2_ld + per * hl;
      const float4* w4 = reinterpret_cast<const float4*>(a.w3) + (size_t)per * hl * 2;
      sq_f32x4 xv[4];  // per / 4 <= 4 (nh <= 512)
      for (int q = 0; q < per / 4; ++q) xv[q] = LD::f4(xrow + 4 * q);
      for (int q = 0; q < per / 4; ++q) {
        const sq_f32x4 x = xv[q];
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const float4 wa = w4[(q * 4 + ii) * 2], wb2 = w4[(q * 4 + ii) * 2 + 1];
          part[0] += xs[ii] * wa.x; part[1] += xs[ii] * wa.y; part[2] += xs[ii] * wa.z; part[3] += xs[ii] * wa.w;
          part[4] += xs[ii] * wb2.x; part[5] += xs[ii] * wb2.y; part[6] += xs[ii] * wb2.z; part[7] += xs[ii] * wb2.w;
        }
      }
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        float v = part[o];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
        part[o] = v + a.w3[d.nh * 8 + o];
      }
      tp_loc = ci == 0 ? part[0] : (ci == 1 ? part[1] : (ci == 2 ? part[2] : part[3]));
      tp_raw = ci == 0 ? part[4] : (ci == 1 ? part[5] : (ci == 2 ? part[6] : part[7]));
      if (a.tp_out != nullptr && hl < 4) {
        a.tp_out[(size_t)r * a.tp_out_ld + ci] = tp_loc;
        a.tp_out[(size_t)r * a.tp_out_ld + 4 + ci] = tp_raw;
      }
    }
    float wl;
    if (a.mode == CROP_PLAIN) {
      wl = lg;
    } else if (a.mode == CROP_PROP1) {
      wl = zp + wbv * 0.1f;
    } else {
      float loc, sc;
      if (a.mode == CROP_DISC) {
        loc = tp_loc;
        sc = sq_softplus(tp_raw + off) + 1e-2f;
        wl = loc + sc * (ci == 0 ? e[0] : (ci == 1 ? e[1] : (ci == 2 ? e[2] : e[3])));
      } else {
        loc = zp + 1.0f * tp_loc;
        sc = sq_softplus(tp_raw + off - 1.0f) + 1e-2f;
        float acc = 0.0f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          if (jj <= ci) acc += (chv[jj] * sc + (jj == ci ? sc : 0.0f)) * e[jj];
        wl = loc + acc;
      }
      if (hl < 4) {
        float* rn = a.rec_new + ((size_t)r * d.N + slot) * rec::W;
        rn[rec::WHERE + ci] = wl;
        rn[rec::WHERE_LOC + ci] = loc;
        rn[rec::WHERE_SCALE + ci] = sc;
      }
    }
    if (hl < 4) coord_s[ci] = (ci & 2) ? tanhf(wl) : fmaxf(sq_sigmoid(wl), 1e-4f);
  }
  if (stage_img) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int idx = q * 256 + tid;
      if (idx < P) img_s[idx] = v0[q];
    }
    for (int idx = 256 * IPT + tid; idx < P; idx += 256) img_s[idx] = img[idx];  // frames larger than 2560 pixels
  }
  __syncthreads();
  const float* __restrict__ src = stage_img ? img_s : img;
  for (int i = tid; i < 2 * G; i += 256) {
    const bool is_y = i >= G;
    const int j = is_y ? i - G : i;
    const float gn = -1.0f + 2.0f * (float)j / (float)(G - 1);
    const float sc = coord_s[is_y ? 1 : 0], tr = coord_s[is_y ? 3 : 2];
    const float L = (float)((is_y ? d.H : d.W) - 1);
    const float x = 0.5f * L * (sc * gn + tr + 1.0f);
    const float x0 = floorf(x);
    tab_s[i * 2 + 0] = x0;
    tab_s[i * 2 + 1] = x - x0;
  }
  __syncthreads();
  for (int pix = tid, q = 0; pix < G2; pix += 256, ++q) {
    const float mk = q < MPT ? mk0[q < MPT ? q : 0] : (has_mask ? LD::f(a.mask + ((size_t)r * a.mask_row_mul + mrow_add) * G2 + pix) : 1.0f);
    const int i = pix / G, j = pix - i * G;
    const float x0f = tab_s[j * 2], wx1 = tab_s[j * 2 + 1];
    const float y0f = tab_s[(G + i) * 2], wy1 = tab_s[(G + i) * 2 + 1];
    const int x0 = (int)x0f, y0 = (int)y0f;
    float v = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int yy = y0 + dy;
      const float wy = dy ? wy1 : 1.0f - wy1;
      if (yy < 0 || yy >= d.H) continue;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int xx = x0 + dx;
        const float wx = dx ? wx1 : 1.0f - wx1;
        if (xx < 0 || xx >= d.W) continue;
        v += wy * wx * src[yy * d.W + xx];
      }
    }
    a.out[((size_t)r * a.out_row_mul + orow_add) * G2 + pix] = has_mask ? v * mk : v;
  }
  __syncthreads();
}

