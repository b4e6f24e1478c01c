// Per-row spatial-transformer crop (k_crop_row).  The load flavour is a template parameter (plain loads in the product;
// an L1-bypassing flavour served the in-launch hand-off experiments of round 1).  Arithmetic order of the original
// per-sequence k_crop: results are bit-identical.
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

template <class LD, bool stage_img>   // stage_img: the frame is copied to LDS (up to SQ_CROP_STAGE_MAX_PIXELS); else its taps are read where they lie
__device__ void x_crop_row(const CropArgs& a, const POff& po, const Dims& d, int r, int slot, float* smem) {
  float* coord_s = smem;        // 4
  float* tab_s = smem + 4;      // 2 * 2G
  float* img_s = smem + 4 + 4 * d.G;  // H*W when stage_img: the frame is pulled into LDS WHILE wave 0 computes `where`, so
                                      // the gather below does not start a second memory round trip after it
  const int tid = threadIdx.x, b = sq_div(r, d.k_mul);
  const int P = d.H * d.W, G = d.G, G2 = d.G * d.G;
  const int mrow_add = a.mask_row_add + (a.mode == CROP_PROP1 ? slot : 0);
  const int orow_add = a.out_row_add + (a.mode == CROP_PROP1 ? slot : 0);
  const float* __restrict__ img = a.img + (size_t)b * d.P4;
  const bool has_mask = a.mask != nullptr;
  const bool fused_tp = a.t2 != nullptr && (a.mode == CROP_PROP2 || a.mode == CROP_DISC);
  // Requests in the order of the kernel's critical path: the operands of the where sample (32 threads: the fused output layer
  // of the transform, the noise, the previous where) go out FIRST, the frame and the mask -- needed only after the where
  // computation -- behind them.  (The frame used to be requested first: the where operands' loads then sat ~400 instructions of
  // address arithmetic into the kernel.)
  constexpr int IPT = 10;  // first 2560 frame pixels: stored to LDS after the where computation
  float v0[IPT];
  constexpr int MPT = 2;   // mask values of this thread's first pixels, requested up front as well
  float mk0[MPT];
  const int hl = tid, ci = hl & 3;
  const int per = d.nh / 32;
  float tp_loc = 0.0f, tp_raw = 0.0f;
  float e[4] = {0.0f, 0.0f, 0.0f, 0.0f}, zp = 0.0f, off = 0.0f, chv[4] = {0.0f, 0.0f, 0.0f, 0.0f}, wbv = 0.0f, lg = 0.0f;
  constexpr int QM = 2;
  const int nq = per / 4;
  sq_f32x4 xv[QM];
  float4 wv[QM][4][2];
  if (tid < 32) {
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
      const float* xrow = a.t2 + (size_t)r * a.t2_ld + per * hl;
      const float4* w4 = reinterpret_cast<const float4*>(a.w3) + (size_t)per * hl * 2;
      // Every load of the layer is requested before the first product (compile-time trip counts, clamped addresses): with the
      // runtime bound per / 4 the two loops below were four dependent memory round trips (activations one by one, then the
      // weights of each group of 4 inputs) on the critical path of every slot.  nh <= 256 => per / 4 <= 2; sums in the same order.
#pragma unroll
      for (int q = 0; q < QM; ++q) {
        const int qc = min(q, nq - 1);
        xv[q] = LD::f4(xrow + 4 * qc);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) { wv[q][ii][0] = w4[(qc * 4 + ii) * 2]; wv[q][ii][1] = w4[(qc * 4 + ii) * 2 + 1]; }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (stage_img) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) v0[q] = img[min(q * 256 + tid, P - 1)];
  }
#pragma unroll
  for (int q = 0; q < MPT; ++q)
    mk0[q] = has_mask ? LD::f(a.mask + ((size_t)r * a.mask_row_mul + mrow_add) * G2 + min(tid + 256 * q, G2 - 1)) : 1.0f;
  __builtin_amdgcn_sched_barrier(0);
  if (tid < 32) {
    if (fused_tp) {
      float part[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      const float* xrow = a.t2 + (size_t)r * a.t2_ld + per * hl;
      const float4* w4 = reinterpret_cast<const float4*>(a.w3) + (size_t)per * hl * 2;
#pragma unroll
      for (int q = 0; q < QM; ++q) {
        if (q < nq) {
          const sq_f32x4 x = xv[q];
          const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) {
            const float4 wa = wv[q][ii][0], wb2 = wv[q][ii][1];
            part[0] += xs[ii] * wa.x; part[1] += xs[ii] * wa.y; part[2] += xs[ii] * wa.z; part[3] += xs[ii] * wa.w;
            part[4] += xs[ii] * wb2.x; part[5] += xs[ii] * wb2.y; part[6] += xs[ii] * wb2.z; part[7] += xs[ii] * wb2.w;
          }
        }
      }
      for (int q = QM; q < nq; ++q) {  // (wider hidden layers than the library is built for)
        const sq_f32x4 x = LD::f4(xrow + 4 * q);
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
        part[o] = sq_half_sum(part[o]) + a.w3[d.nh * 8 + o];
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
    if (hl < 4) coord_s[ci] = (ci & 2) ? tanhf(wl) : fmaxf(sq_sigmoid_geo(wl), 1e-4f);
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
  if (!stage_img) {
    // Large frames (SQ_CROP_STAGE_MAX_PIXELS): the four taps of every glimpse pixel straight from memory (L2: the K particles of
    // a sequence and the slots of a frame keep reading the same 64 KB), PX pixels of a thread in flight at once -- unconditional
    // loads from clamped addresses, taps outside the frame get weight zero (a guarded load is fenced with a full wait)
    constexpr int PX = 2;
    for (int p0 = tid; p0 < G2; p0 += 256 * PX) {
      float tv[PX][4], tw[PX][4], mk[PX];
#pragma unroll
      for (int u = 0; u < PX; ++u) {
        const int pix = min(p0 + 256 * u, G2 - 1);
        mk[u] = has_mask ? LD::f(a.mask + ((size_t)r * a.mask_row_mul + mrow_add) * G2 + pix) : 1.0f;
        const int i = sq_div(pix, d.g_mul), j = pix - i * G;
        const float x0f = tab_s[j * 2], wx1 = tab_s[j * 2 + 1];
        const float y0f = tab_s[(G + i) * 2], wy1 = tab_s[(G + i) * 2 + 1];
        const int x0 = (int)x0f, y0 = (int)y0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int yy = y0 + dy, xx = x0 + dx;
            const bool ok = yy >= 0 && yy < d.H && xx >= 0 && xx < d.W;
            tv[u][dy * 2 + dx] = img[min(max(yy, 0), d.H - 1) * d.W + min(max(xx, 0), d.W - 1)];
            tw[u][dy * 2 + dx] = ok ? (dy ? wy1 : 1.0f - wy1) * (dx ? wx1 : 1.0f - wx1) : 0.0f;
          }
      }
#pragma unroll
      for (int u = 0; u < PX; ++u)
        if (p0 + 256 * u < G2) {
          float v = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) v += tw[u][q] * tv[u][q];
          a.out[((size_t)r * a.out_row_mul + orow_add) * G2 + p0 + 256 * u] = has_mask ? v * mk[u] : v;
        }
    }
  } else {
    // Frame staged in LDS: the same four taps per glimpse pixel out of `img_s`, TWO pixels of a thread at once and without branches
    // (unconditional reads from clamped positions, taps outside the frame get weight zero): a 20 x 20 glimpse is two rounds of 256
    // threads, each a dependent chain table read -> tap reads -> sum -> store that used to run one after the other.  Same products
    // in the same order: a tap outside the frame adds fma(0, x, v) = v where the branchy form skipped it.
    constexpr int PX = MPT;
    for (int p0 = tid, it = 0; p0 < G2; p0 += 256 * PX, ++it) {
      float tv[PX][4], tw[PX][4], mk[PX];
#pragma unroll
      for (int u = 0; u < PX; ++u) {
        const int pix = min(p0 + 256 * u, G2 - 1);
        mk[u] = it == 0 ? mk0[u] : (has_mask ? LD::f(a.mask + ((size_t)r * a.mask_row_mul + mrow_add) * G2 + pix) : 1.0f);
        const int i = sq_div(pix, d.g_mul), j = pix - i * G;
        const float x0f = tab_s[j * 2], wx1 = tab_s[j * 2 + 1];
        const float y0f = tab_s[(G + i) * 2], wy1 = tab_s[(G + i) * 2 + 1];
        const int x0 = (int)x0f, y0 = (int)y0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int yy = y0 + dy, xx = x0 + dx;
            const bool ok = yy >= 0 && yy < d.H && xx >= 0 && xx < d.W;
            tv[u][dy * 2 + dx] = src[min(max(yy, 0), d.H - 1) * d.W + min(max(xx, 0), d.W - 1)];
            tw[u][dy * 2 + dx] = ok ? (dy ? wy1 : 1.0f - wy1) * (dx ? wx1 : 1.0f - wx1) : 0.0f;
          }
      }
#pragma unroll
      for (int u = 0; u < PX; ++u)
        if (p0 + 256 * u < G2) {
          float v = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) v += tw[u][q] * tv[u][q];
          a.out[((size_t)r * a.out_row_mul + orow_add) * G2 + p0 + 256 * u] = has_mask ? v * mk[u] : v;
        }
    }
  }
  __syncthreads();
}

