// Adjoint (backward) kernels of the SQAIR hot path — building blocks of sqair_backward.
// Each one is exported through the C-ABI for unit parity tests against the oracle's autograd.
//
//   k_crop_bwd            d/d(where logits) and d/d(mask) of the spatial-transformer crop
//                         (reference forward: sqair/modules.py:170-227; TF differentiates tf.contrib.resampler
//                         w.r.t. the warp, sqair/model.py:160)
//   k_insert_loglik_bwd   d/d(glimpse), d/d(where logits), d/d(mean image) of the decoder canvas + Gaussian
//                         log-likelihood (reference forward: sqair/modules.py:435-467, sqair/seq.py:271-274)
//   k_elbo_bwd            d(VIMCO target)/d(log w_t), d/d(discrete log prob_t) (sqair/targets.py:62-75,
//                         sqair/model.py:150-158)
//   k_wgrad               dW += A^T dY, db += colsum(dY) on the fp32 matrix cores, written in the reference's
//                         [in, out] layout straight into the flat gradient buffer
#include "sqair_glue.h"
#include "sqair_bwd.h"
#include "sqair_canvas.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <climits>
#include <vector>

typedef float f32x4_b __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// crop backward.  One workgroup per sequence b: the frame is staged in LDS once for its K particles; each
// output pixel contributes 4 partial derivatives (d sx, d sy, d tx, d ty) that are reduced with wave
// shuffles, then across the 4 waves through LDS.
// ------------------------------------------------------------------------------------------------
struct CropBwdArgs {
  const float* img;      // [B,H,W]
  const float* logits;   // [R,4]
  const float* mask;     // optional [R,G*G]
  const float* g_out;    // [R,G*G] upstream gradient of the (masked) glimpse
  float* d_logits;       // [R,4]
  float* d_mask;         // optional [R,G*G]
};

__global__ __launch_bounds__(256) void k_crop_bwd(const CropBwdArgs a, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img_s = smem;  // H*W
  __shared__ float red_s[4][4];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int P = d.H * d.W, G = d.G, G2 = d.G * d.G;
  const float* img = a.img + (size_t)b * d.P4;
  for (int i = tid; i < P; i += 256) img_s[i] = img[i];
  __syncthreads();
  for (int kp = 0; kp < d.K; ++kp) {
    const int r = b * d.K + kp;
    const float l0 = a.logits[(size_t)r * 4 + 0], l1 = a.logits[(size_t)r * 4 + 1];
    const float l2 = a.logits[(size_t)r * 4 + 2], l3 = a.logits[(size_t)r * 4 + 3];
    const float s0 = sq_sigmoid_geo(l0), s1 = sq_sigmoid_geo(l1);
    const float sx = fmaxf(s0, 1e-4f), sy = fmaxf(s1, 1e-4f), tx = tanhf(l2), ty = tanhf(l3);
    const float hx = 0.5f * (float)(d.W - 1), hy = 0.5f * (float)(d.H - 1);
    float dsx = 0.0f, dsy = 0.0f, dtx = 0.0f, dty = 0.0f;
    for (int pix = tid; pix < G2; pix += 256) {
      const int i = sq_div(pix, d.g_mul), j = pix - i * G;
      const float gx = -1.0f + 2.0f * (float)j / (float)(G - 1), gy = -1.0f + 2.0f * (float)i / (float)(G - 1);
      const float x = hx * (sx * gx + tx + 1.0f), y = hy * (sy * gy + ty + 1.0f);
      const float x0f = floorf(x), y0f = floorf(y);
      const int x0 = (int)x0f, y0 = (int)y0f;
      const float wx1 = x - x0f, wy1 = y - y0f;
      float v = 0.0f, dvdx = 0.0f, dvdy = 0.0f;
      float t[2][2];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int yy = y0 + dy, xx = x0 + dx;
          t[dy][dx] = (yy >= 0 && yy < d.H && xx >= 0 && xx < d.W) ? img_s[yy * d.W + xx] : 0.0f;
        }
      v = (1.0f - wy1) * ((1.0f - wx1) * t[0][0] + wx1 * t[0][1]) + wy1 * ((1.0f - wx1) * t[1][0] + wx1 * t[1][1]);
      dvdx = (1.0f - wy1) * (t[0][1] - t[0][0]) + wy1 * (t[1][1] - t[1][0]);
      dvdy = (1.0f - wx1) * (t[1][0] - t[0][0]) + wx1 * (t[1][1] - t[0][1]);
      float g = a.g_out[(size_t)r * G2 + pix];
      if (a.mask != nullptr) {
        const float mk = a.mask[(size_t)r * G2 + pix];
        if (a.d_mask != nullptr) a.d_mask[(size_t)r * G2 + pix] = g * v;
        g *= mk;
      }
      dsx += g * dvdx * hx * gx;
      dtx += g * dvdx * hx;
      dsy += g * dvdy * hy * gy;
      dty += g * dvdy * hy;
    }
    dsx = sq_wave_sum(dsx); dsy = sq_wave_sum(dsy); dtx = sq_wave_sum(dtx); dty = sq_wave_sum(dty);
    if (lane == 0) { red_s[wave][0] = dsx; red_s[wave][1] = dsy; red_s[wave][2] = dtx; red_s[wave][3] = dty; }
    __syncthreads();
    if (tid < 4) {
      const float tot = red_s[0][tid] + red_s[1][tid] + red_s[2][tid] + red_s[3][tid];
      // clip_preserve passes the gradient through the clip (ops.py:33-42): d sx / d l0 = sigmoid'(l0)
      const float dl = tid == 0 ? s0 * (1.0f - s0) : (tid == 1 ? s1 * (1.0f - s1) : (tid == 2 ? 1.0f - tx * tx : 1.0f - ty * ty));
      a.d_logits[(size_t)r * 4 + tid] = tot * dl;
    }
    __syncthreads();
  }
}

extern "C" int sqair_st_crop_bwd(SqairHandle* h, const float* img, const float* where_logits, const float* mask,
                                 const float* g_out, float* d_where_logits, float* d_mask, int B, void* stream) {
  if (!h || !img || !where_logits || !g_out || !d_where_logits || B < 1) return -1;
  SqairConfig c;
  if (sqair_get_config(h, &c) != 0) return -1;
  Dims d = make_dims(c, B);
  CropBwdArgs a{img, where_logits, mask, g_out, d_where_logits, d_mask};
  const size_t shm = (size_t)d.H * d.W * sizeof(float);
  if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_crop_bwd, 150 * 1024) != 0) return -2;
  SQ_LAUNCH(k_crop_bwd, dim3(B), dim3(256), shm, (hipStream_t)stream, a, d);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// insert + log-likelihood backward.  One workgroup per (row b', frame), band by band like the forward kernel
// (sqair_canvas.h): (1) the canvas / mask sum of the band are rebuilt over the slots' boxes; (2) per pixel the
// scalar adjoints g_cv (canvas) and g_ms (written-to mask sum) replace them in LDS and the row's contribution
// to d mean_img is written (summed over rows by a second tiny kernel); (3) per slot, over its box, the adjoints
// are pushed back into the inverse-warp coordinates -> 4 sums per slot; (4) the N glimpses are GATHERED:
// texel (gy, gx) of slot k sums g_cv over the canvas pixels whose bilinear footprint contains it,
// |xg(X) - gx| < 1 and |yg(Y) - gy| < 1, with weight (1 - |xg - gx|)(1 - |yg - gy|) -- the transpose of the
// forward's two-tap interpolation, separable, deterministic; the pixel runs per texel column / row are found
// once per workgroup.  (Rounds 1-2: the pixels SCATTERED into an LDS tile with float atomics, 108 us per step;
// round 3 until the box formulation: every pixel walked every slot, 101 us, and 697 us at 128 x 128.)
// ------------------------------------------------------------------------------------------------
struct InsertBwdArgs {
  const float* glimpse;      // [R,N,G2]
  const float* where;        // [R,N,4]
  const float* pres;         // [R,N]
  const float* img;          // [B,H,W]
  const float* mean_img;     // [H,W]
  const float* g_ll;         // [R] upstream gradient of data_ll
  float* d_glimpse;          // [R,N,G2]
  float* d_where;            // [R,N,4]
  float* d_mean_rows;        // [R,H*W] per-row contribution to d mean_img
  float std_fg, std_bg;
  const float* rec;          // alternatively: merged slot records [.., rec_ld] (where at +0, presence at +54)
  int rec_ld;
  int dw_ld;                 // leading dimension of d_where rows (4 = plain, 64 = gradient records)
  const float* scale;        // optional (training step): the decoder's output scale and its gradient,
  float* d_scale;            //   d scale += sum(d_glimpse * glimpse) / scale (modules.py:144-147) -- both factors are in LDS here
};

// LDS past the canvas block: glimpse gradient [N][G2], pixel runs per texel column / row [N][G] x 2 (first | last << 16),
// per-wave partial sums of the four coordinate gradients [4][N][4]
static inline size_t insert_bwd_lds_floats(const Dims& d, int band_rows) {
  return sq_canvas_lds_floats(d.N, d.G, d.H, d.W, band_rows) + (size_t)d.N * d.G * d.G + 2 * (size_t)d.N * d.G + 16 * (size_t)d.N;
}

template <int PF, int ROWS>
__global__ __launch_bounds__(256) void k_insert_loglik_bwd(const InsertBwdArgs a, const Dims d, const int band_rows SQ_TLP) {
  SQ_TL_SCOPE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.N, G = d.G, G2 = d.G * d.G, H = d.H, W = d.W, P = d.H * d.W;
  const CanvasLds c = sq_canvas_carve(smem, N, G, H, W, band_rows);
  float* dgl_s = c.end;                               // N * G2
  int* xr_s = reinterpret_cast<int*>(dgl_s + N * G2); // N * G
  int* yr_s = xr_s + N * G;                           // N * G
  float* acc_s = reinterpret_cast<float*>(yr_s + N * G);  // 4 waves * N * 4
  const int r = sq_row_of_wg(blockIdx.x, d), tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, tx = tid & 31, ty = tid >> 5;
  const int fr = blockIdx.y;  // frame
  const int b = sq_div(r, d.k_mul);
  const size_t fs = (size_t)fr * d.R * N + (size_t)r * N;  // first slot-row of this (frame, row)
  const size_t frr = (size_t)fr * d.R + r;
  const float gll = a.g_ll[frr];
  const float* __restrict__ img = a.img + ((size_t)fr * d.B + b) * d.P4;
  if (a.rec) sq_canvas_prologue(c, a.glimpse + fs * G2, a.rec + fs * a.rec_ld + rec::WHERE, a.rec_ld, a.rec + fs * a.rec_ld + rec::PRES, a.rec_ld, N, G, H, W);
  else sq_canvas_prologue(c, a.glimpse + fs * G2, a.where + fs * 4, 4, a.pres + fs, 1, N, G, H, W);
  for (int i = tid; i < N * G2; i += 256) dgl_s[i] = 0.0f;
  for (int i = tid; i < 16 * N; i += 256) acc_s[i] = 0.0f;
  // pixel run of texel column gx of slot k: the columns of the box with |xt - gx| < 1 (xt grows with the column); rows alike
  for (int i = tid; i < 2 * N * G; i += 256) {
    const bool is_y = i >= N * G;
    const int q = is_y ? i - N * G : i, k = q / G, g = q - k * G;
    const float* t = is_y ? c.yt + k * H : c.xt + k * W;
    const int lo = c.box[k * 4 + (is_y ? 2 : 0)], hi = c.box[k * 4 + (is_y ? 3 : 1)];
    int below = 0, above = 0;
    for (int j = lo; j <= hi; ++j) {
      const float v = t[j] - (float)g;
      below += v <= -1.0f;
      above += v >= 1.0f;
    }
    (is_y ? yr_s : xr_s)[q] = (lo + below) | ((hi - above) << 16);   // (empty: last < first)
  }
  __syncthreads();
  const float hg = 0.5f * (float)(G - 1);
  const bool one_sd = a.std_fg == a.std_bg;
  const float gll_isd2 = gll / (a.std_fg * a.std_fg), m_bg = sq_sigmoid(-10.0f);
  for (int yb0 = 0; yb0 < H; yb0 += band_rows) {
    const int yb1 = min(H, yb0 + band_rows) - 1, n = (yb1 - yb0 + 1) * W, pix0 = yb0 * W;
    float xv[PF], mv[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int p = tid + q * 256;
      xv[q] = p < n ? img[pix0 + p] : 0.0f;
      mv[q] = p < n ? a.mean_img[pix0 + p] : 0.0f;
    }
    sq_canvas_band<ROWS>(c, yb0, yb1, N, G, H, W);
    // ---- (2) adjoints of the band's pixels.  As in the forward kernel: one likelihood scale when std_fg == std_bg (its powers taken
    // once), and no exponential for a wavefront whose 64 pixels lie outside every glimpse's box (mask sum exactly 0)
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int p = tid + q * 256;
      const float msv = p < n ? c.ms[p] : 0.0f;
      const bool any_on = __builtin_amdgcn_ballot_w64(msv != 0.0f) != 0ull;   // (wave-uniform)
      if (p < n) {
        const float m = any_on ? sq_sigmoid(-10.0f + msv * 20.0f) : m_bg;
        const float mean = mv[q];
        const float cv = c.cv[p] + mean * m;
        const float diff = xv[q] - cv;
        float g_cv, g_m;
        if (one_sd) {
          g_cv = gll_isd2 * diff;
          g_m = g_cv * mean;
        } else {
          const float sd = m * a.std_fg + (1.0f - m) * a.std_bg;
          g_cv = gll * diff / (sd * sd);
          const float g_sd = gll * (diff * diff / (sd * sd * sd) - 1.0f / sd);
          g_m = g_cv * mean + g_sd * (a.std_fg - a.std_bg);
        }
        a.d_mean_rows[frr * P + pix0 + p] = g_cv * m;
        c.cv[p] = g_cv;
        c.ms[p] = g_m * 20.0f * m * (1.0f - m);
      }
    }
    __syncthreads();
    // ---- (3) d/d (sx, sy, tx, ty) of every slot, over its box.  With gx / gy the adjoints of a pixel's glimpse coordinates,
    // d sx = -hg / sx^2 sum gx (Xn - tx), d tx = -hg / sx sum gx (y alike): four running sums per thread, no division
    // per pixel; WG_ROWS patch rows in flight, taps at clamped addresses under validity masks (one pixel at a time
    // with four divisions each this phase was 24 of the kernel's 64 us)
    {
      constexpr int WG_ROWS = 2;   // (4 rows in flight: 123 VGPRs, a workgroup fewer per CU, slower)
      const float inv_w = 2.0f / (float)(W - 1), inv_h = 2.0f / (float)(H - 1);
      for (int k = 0; k < N; ++k) {
        CanvasSlot s;
        if (!sq_canvas_slot(c, k, yb0, yb1, s)) continue;
        const float* gk = c.gl + k * G2;
        const float sx = c.co[k * 4 + 0], sy = c.co[k * 4 + 1], tcx = c.co[k * 4 + 2], tcy = c.co[k * 4 + 3];
        float sgx = 0.0f, sgxx = 0.0f, sgy = 0.0f, sgyy = 0.0f;
        for (int Y0 = s.y0 + ty; Y0 <= s.y1; Y0 += 8 * WG_ROWS)
          for (int X = s.x0 + tx; X <= s.x1; X += 32) {
            const float xg = c.xt[k * W + X];
            const float x0f = floorf(xg);
            const int x0 = (int)x0f;
            const float wx1 = xg - x0f;
            const bool xa_ok = x0 >= 0, xb_ok = x0 + 1 < G;
            const int xa = max(x0, 0), xb = min(x0 + 1, G - 1);
            const float Xn = -1.0f + (float)X * inv_w;
            float gxs[WG_ROWS], gys[WG_ROWS], yns[WG_ROWS];
#pragma unroll
            for (int u = 0; u < WG_ROWS; ++u) {
              const int Y = min(Y0 + 8 * u, s.y1);
              const float yg = c.yt[k * H + Y];
              const float y0f = floorf(yg);
              const int y0 = (int)y0f;
              const float wy1 = yg - y0f;
              const bool ya_ok = y0 >= 0, yb_ok = y0 + 1 < G;
              const float* ra = gk + max(y0, 0) * G;
              const float* rb = gk + min(y0 + 1, G - 1) * G;
              const float v00 = ya_ok && xa_ok ? 1.0f : 0.0f, v01 = ya_ok && xb_ok ? 1.0f : 0.0f;
              const float v10 = yb_ok && xa_ok ? 1.0f : 0.0f, v11 = yb_ok && xb_ok ? 1.0f : 0.0f;
              const float t00 = ra[xa] * v00, t01 = ra[xb] * v01, t10 = rb[xa] * v10, t11 = rb[xb] * v11;
              // d/d xg, d/d yg of (g_cv * bilinear(glimpse) + g_ms * bilinear(ones))
              const float dvdx = (1.0f - wy1) * (t01 - t00) + wy1 * (t11 - t10);
              const float dvdy = (1.0f - wx1) * (t10 - t00) + wx1 * (t11 - t01);
              const float dodx = (1.0f - wy1) * (v01 - v00) + wy1 * (v11 - v10);
              const float dody = (1.0f - wx1) * (v10 - v00) + wx1 * (v11 - v01);
              const int o = (Y - yb0) * W + X;
              const float on = Y0 + 8 * u <= s.y1 ? s.pk : 0.0f;   // (rows past the box: clamped reads, no contribution)
              gxs[u] = on * (c.cv[o] * dvdx + c.ms[o] * dodx);
              gys[u] = on * (c.cv[o] * dvdy + c.ms[o] * dody);
              yns[u] = -1.0f + (float)Y * inv_h;
            }
#pragma unroll
            for (int u = 0; u < WG_ROWS; ++u) {
              sgx += gxs[u];
              sgxx += gxs[u] * (Xn - tcx);
              sgy += gys[u];
              sgyy += gys[u] * (yns[u] - tcy);
            }
          }
        float d0 = -hg / (sx * sx) * sgxx, d1 = -hg / (sy * sy) * sgyy;
        float d2 = -hg / sx * sgx, d3 = -hg / sy * sgy;
        d0 = sq_wave_sum(d0); d1 = sq_wave_sum(d1); d2 = sq_wave_sum(d2); d3 = sq_wave_sum(d3);
        if (lane == 0) {   // one writer per (wave, slot)
          float* ac = acc_s + (wave * N + k) * 4;
          ac[0] += d0; ac[1] += d1; ac[2] += d2; ac[3] += d3;
        }
      }
    }
    // ---- (4) glimpse gradient by gathering (see the header): texel i = (k, gy, gx), rows of this band
    for (int i = tid; i < N * G2; i += 256) {
      const int k = i / G2, q = i - k * G2, gy = q / G, gx = q - gy * G;
      if (c.pres[k] == 0.0f) continue;
      const int xr = xr_s[k * G + gx], yr = yr_s[k * G + gy];
      const int X0 = xr & 0xffff, X1 = xr >> 16, Y0 = max(yr & 0xffff, yb0), Y1 = min(yr >> 16, yb1);
      float acc = dgl_s[i];
      for (int Y = Y0; Y <= Y1; ++Y) {
        const float wy = 1.0f - fabsf(c.yt[k * H + Y] - (float)gy);
        float rowsum = 0.0f;
        for (int X = X0; X <= X1; ++X) rowsum += (1.0f - fabsf(c.xt[k * W + X] - (float)gx)) * c.cv[(Y - yb0) * W + X];
        acc += wy * rowsum;
      }
      dgl_s[i] = acc;
    }
    __syncthreads();  // the next band clears c.cv / c.ms; after the last one: dgl_s / acc_s complete
  }
  float dsc = 0.0f;
  for (int i = tid; i < N * G2; i += 256) {
    const float v = dgl_s[i] * c.pres[i / G2];
    a.d_glimpse[fs * G2 + i] = v;
    dsc += v * c.gl[i];
  }
  if (a.d_scale != nullptr) {  // (was a launch of its own over all of d_glimpse and glimpse: 14 us)
    __shared__ float dsc_s[4];
    dsc = sq_wave_sum(dsc);
    if ((tid & 63) == 0) dsc_s[tid >> 6] = dsc;
    __syncthreads();
    if (tid == 0) unsafeAtomicAdd(a.d_scale, (dsc_s[0] + dsc_s[1] + dsc_s[2] + dsc_s[3]) / a.scale[0]);
  }
  if (tid < N * 4) {
    const int k = tid >> 2, q = tid & 3;
    const float tot = acc_s[(0 * N + k) * 4 + q] + acc_s[(1 * N + k) * 4 + q] + acc_s[(2 * N + k) * 4 + q] + acc_s[(3 * N + k) * 4 + q];
    const float l = a.rec ? a.rec[(fs + k) * a.rec_ld + rec::WHERE + q] : a.where[fs * 4 + tid];
    const float sg = sq_sigmoid_geo(l), th = tanhf(l);
    a.d_where[(fs + k) * a.dw_ld + q] = tot * ((q & 2) ? 1.0f - th * th : sg * (1.0f - sg));
  }
}

// ------------------------------------------------------------------------------------------------
// The same adjoint for frames wider than a wavefront, ROW-WAVE formulation (sqair_canvas.h; forward: k_insert_loglik_rows): a wave
// owns blocks of RB consecutive canvas rows (block-cyclic over the four waves), lane l the CPL adjacent columns from CPL l; column
// taps and their derivative patterns in registers, row taps as broadcast LDS records, canvas / mask sum / pixel adjoints in
// registers, no band, no barrier between prologue and epilogue.  Per row and slot whose box meets it:
// (1) the taps are read once more and give d/d xg, d/d yg of the pixel's canvas value and mask term -> four running sums per slot
//     and lane (sum gx, sum gx Xn, sum gy, sum gy Yn; d sx = -hg / sx^2 (sum gx Xn - tx sum gx), ...);
// (2) the glimpse gradient.  Glimpse row i of a slot collects sa g_cv from the canvas rows whose upper tap is i and sb g_cv from
//     those whose upper tap is i - 1; the tap index grows with the canvas row, so while a wave walks a block it keeps, per slot and
//     COLUMN, just two accumulators (rows i and i + 1: two FMAs per pixel) and FLUSHES one when the index moves on: the finished
//     row is folded over x into the slot's G texel columns -- each lane writes wa acc / wb acc of its columns to a per-wave LDS
//     line, the G lanes that own a texel column sum their pixel run (runs found once per workgroup) -- and added to the
//     workgroup's gradient tile with ONE ds_add_f32 per flush.  LDS float atomics cost ~85 cycles per instruction whatever the
//     lane count (measured: a fold per canvas row through ds_add_f32, ten per row and slot, ran 600 us at 128 x 128; the band
//     kernel 206): the formulation is chosen to need few of them, ~4.5 per slot and block of 8 rows.
// ------------------------------------------------------------------------------------------------
#ifndef SQ_BWD_WPE
#define SQ_BWD_WPE 3   // waves per SIMD the register allocation must leave room for (4: 127 VGPRs with 67 scratch accesses, 169 us; 3: 155 VGPRs, none, 126 us)
#endif
#ifndef SQ_BWD_RB_N
#define SQ_BWD_RB_N 16
#endif
constexpr int SQ_BWD_RB = SQ_BWD_RB_N;   // canvas rows per block
template <int NMAX, int CPL, bool FULLW, bool ONE_SD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SQ_BWD_WPE, 8))) void k_insert_loglik_bwd_rows(const InsertBwdArgs a, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  static_assert(CPL == 2 || CPL == 4, "column pairs");
  constexpr int NT = 256, WAVES = 4, CP = CPL / 2, RB = SQ_BWD_RB, LW = 64 * CPL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.N, G = d.G, G2 = d.G * d.G, H = d.H, W = d.W, P = d.H * d.W;
  const CanvasRowsLds c = sq_canvas_rows_carve(smem, N, G, H, true);
  int* runs_s = reinterpret_cast<int*>(c.end);             // [N][G]   first | last << 16 in-box column whose left tap is texel column j
  float* acc_s = reinterpret_cast<float*>(runs_s + N * G);  // [WAVES][N][4]
  float* un_s = acc_s + ((WAVES * N * 4 + 3) & ~3);         // union: { column records [N][W] float4 } (prologue) | { dgl [N][G2], lines [WAVES][2][LW] }
  float4* xrec_s = reinterpret_cast<float4*>(un_s);
  float* dgl_s = un_s;
  float* line_s = dgl_s + ((N * G2 + 3) & ~3);
  const int r = sq_row_of_wg(blockIdx.x, d), tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = blockIdx.y;  // frame
  const int b = sq_div(r, d.k_mul);
  const size_t fs = (size_t)fr * d.R * N + (size_t)r * N;  // first slot-row of this (frame, row)
  const size_t frr = (size_t)fr * d.R + r;
  const float gll = a.g_ll[frr];
  const char* __restrict__ img = reinterpret_cast<const char*>(a.img + ((size_t)fr * d.B + b) * d.P4);
  const char* __restrict__ mean = reinterpret_cast<const char*>(a.mean_img);
  sq_f2 xn[CP], mn[CP];
  auto request = [&](int yy) {
    const int y = min(yy, H - 1);   // (scalar)
    if (FULLW) {
      const unsigned off = (unsigned)(y * W * 4) + (unsigned)lane * (CPL * 4);
      if (CPL == 2) {
        xn[0] = *reinterpret_cast<const sq_f2*>(img + off);
        mn[0] = *reinterpret_cast<const sq_f2*>(mean + off);
      } else {
        const sq_f4 xx = *reinterpret_cast<const sq_f4*>(img + off), mm = *reinterpret_cast<const sq_f4*>(mean + off);
        xn[0] = xx.xy; xn[CP - 1] = xx.zw; mn[0] = mm.xy; mn[CP - 1] = mm.zw;
      }
    } else {
#pragma unroll
      for (int q = 0; q < CP; ++q) {   // (clamped addresses, unconditional loads)
        const unsigned o0 = (unsigned)(y * W + min(lane * CPL + 2 * q, W - 1)) * 4, o1 = (unsigned)(y * W + min(lane * CPL + 2 * q + 1, W - 1)) * 4;
        xn[q] = sq_f2{*reinterpret_cast<const float*>(img + o0), *reinterpret_cast<const float*>(img + o1)};
        mn[q] = sq_f2{*reinterpret_cast<const float*>(mean + o0), *reinterpret_cast<const float*>(mean + o1)};
      }
    }
  };
  // row of the wave's t-th trip: block-cyclic, RB rows per block
  auto row_of = [&](int t) { return ((t / RB) * WAVES + wave) * RB + (t % RB); };
  request(row_of(0));
  if (a.rec) sq_canvas_rows_prologue<NT, (NMAX * 20 * 20 + NT - 1) / NT, true>(c, a.glimpse + fs * G2, a.rec + fs * a.rec_ld + rec::WHERE, a.rec_ld, a.rec + fs * a.rec_ld + rec::PRES, a.rec_ld, N, G, H, d.g_mul);
  else sq_canvas_rows_prologue<NT, (NMAX * 20 * 20 + NT - 1) / NT, true>(c, a.glimpse + fs * G2, a.where + fs * 4, 4, a.pres + fs, 1, N, G, H, d.g_mul);
  // column records of every (slot, column), computed once per workgroup: {left texel's byte offset in a pair row, wa, wb, {d wa, d wb} as two halves}
  for (int k = 0; k < N; ++k) {
    const float sx = c.co[k * 4 + 0], tx = c.co[k * 4 + 2];
    for (int x = tid; x < W; x += NT) {
      const float g = sq_canvas_coord(x, W, sx, tx, G);
      const CanvasAxisTap t = sq_canvas_axis_tap(g, G);
      float da, db;
      sq_canvas_axis_tap_d(g, G, da, db);
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 dh = {(_Float16)da, (_Float16)db};
      xrec_s[k * W + x] = make_float4(__builtin_bit_cast(float, t.i * 8), t.wa, t.wb, __builtin_bit_cast(float, dh));
    }
  }
  for (int i = tid; i < N * G; i += NT) runs_s[i] = 1;   // (empty: first 1, last 0)
  __syncthreads();
  // runs: the in-box columns are one interval per slot and their tap index grows with the column: both ends of a run have one writer
  for (int i = tid; i < N * W; i += NT) {
    const int k = i / W, x = i - k * W;
    const float4 me = xrec_s[i];
    if (me.y + me.z != 0.0f) {
      const int xi = __builtin_bit_cast(int, me.x) >> 3;
      bool first = x == 0, last = x == W - 1;
      if (!first) { const float4 l = xrec_s[i - 1]; first = (l.y + l.z == 0.0f) || (__builtin_bit_cast(int, l.x) >> 3) != xi; }
      if (!last) { const float4 rr = xrec_s[i + 1]; last = (rr.y + rr.z == 0.0f) || (__builtin_bit_cast(int, rr.x) >> 3) != xi; }
      short* rs = reinterpret_cast<short*>(runs_s + k * G + xi);
      if (first) rs[0] = (short)x;
      if (last) rs[1] = (short)x;
    }
  }
  // this thread's column taps (registers)
  int xo[NMAX][CPL];
  sq_f2 wab[NMAX][CPL];
  unsigned dabh[NMAX][CPL];   // {d wa, d wb} as halves (values 0, +-1)
  float xnrm[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) xnrm[q] = -1.0f + 2.0f * (float)(lane * CPL + q) / (float)(W - 1);
#pragma unroll
  for (int k = 0; k < NMAX; ++k)
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int x = lane * CPL + q;
      float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (k < N && (FULLW || x < W)) t = xrec_s[k * W + x];
      xo[k][q] = __builtin_bit_cast(int, t.x);
      wab[k][q] = sq_f2{t.y, t.z};
      dabh[k][q] = __builtin_bit_cast(unsigned, t.w);
    }
  __syncthreads();   // runs complete; the column records are dead: their storage becomes the gradient tile and the lines
  int runa[NMAX], mrl[NMAX];   // lane j: run of the columns whose left tap is texel column j; longest run (scalar)
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    const int kk = min(k, N - 1);
    runa[k] = (lane < G - 1) ? runs_s[kk * G + lane] : 1;
    const int len = max((runa[k] >> 16) - (runa[k] & 0xffff) + 1, 0);
    mrl[k] = __builtin_amdgcn_readfirstlane((int)sq_wave_max((float)len));
  }
  for (int i = tid; i < N * G2; i += NT) dgl_s[i] = 0.0f;
  if (lane < 32) line_s[wave * 2 * (LW + 16) + 2 * LW + lane] = 0.0f;   // the line's padding: read under a zero weight, so it must be finite
  float sgx[NMAX], sgxx[NMAX], sgy[NMAX], sgyy[NMAX];
  sq_f2 acc_a[NMAX][CP], acc_b[NMAX][CP];   // per column: sa g_cv summed for glimpse row cur, sb g_cv for row cur + 1
  int cur[NMAX];                             // upper tap index the accumulators belong to, -1: none (scalar)
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    sgx[k] = sgxx[k] = sgy[k] = sgyy[k] = 0.0f;
    cur[k] = -1;
#pragma unroll
    for (int q = 0; q < CP; ++q) acc_a[k][q] = acc_b[k][q] = sq_f2{0.0f, 0.0f};
  }
  const float gll_isd2 = gll / (a.std_fg * a.std_fg), m_bg = sq_sigmoid(-10.0f);
  const char* __restrict__ prb = reinterpret_cast<const char*>(c.pr);
  float* lineA = line_s + wave * 2 * (LW + 16);   // {wa acc, wb acc} per column (+ padding: a run's loop may read past the row)
  // (lineA's padding is cleared below, once the column records that share its storage are dead)
  // fold of a finished glimpse row over x and its addition to the gradient tile (see the header): the line holds {wa acc, wb acc} per
  // column; lane j sums both over the run of columns whose left tap is texel column j -- the first sum belongs to texel column j,
  // the second to j + 1 (fetched from the lane below)
  auto flush = [&](int k, int grow, const sq_f2 (&ac)[CP], const sq_f2 (&wq)[CPL], int ra, int nmax) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const float av = (q & 1) ? ac[q / 2].y : ac[q / 2].x;
      *reinterpret_cast<sq_f2*>(lineA + 2 * (lane * CPL + q)) = wq[q] * sq_f2{av, av};
    }
    const int a0 = ra & 0xffff, len = (ra >> 16) - a0;
    const sq_f2* lp = reinterpret_cast<const sq_f2*>(lineA) + a0;
    float sum_a = 0.0f, sum_b = 0.0f;
    for (int t = 0; t < nmax; ++t) {
      const sq_f2 v = lp[min(t, LW + 15 - a0)];   // (the padding covers the runs of the usual glimpse sizes; small glimpses have longer runs)
      const float on = t <= len ? 1.0f : 0.0f;
      sum_a = fmaf(on, v.x, sum_a);
      sum_b = fmaf(on, v.y, sum_b);
    }
    // lane j - 1's second sum: a row shift, lane 16 (first of its DPP row) fetches lane 15's by hand
    float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum_b), 0x111, 0xf, 0xf, true));   // row_shr:1
    const float s15 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sum_b), 15));
    if (lane == 16) up = s15;
    if (lane < G) sq_lds_add(dgl_s + k * G2 + grow * G + lane, sum_a + up);
  };
  // the accumulators of slot k move on by one glimpse row: row cur is finished
  auto advance = [&](int k) {
    flush(k, cur[k], acc_a[k], wab[k], runa[k], mrl[k]);
#pragma unroll
    for (int q = 0; q < CP; ++q) {
      acc_a[k][q] = acc_b[k][q];
      acc_b[k][q] = sq_f2{0.0f, 0.0f};
    }
    cur[k] += 1;
  };
  const int n_trips = (H + WAVES * RB - 1) / (WAVES * RB) * RB;   // (rows past the frame are skipped inside)
  unsigned rm_n = c.rmask[min(row_of(0), H - 1)];
  __syncthreads();   // dgl_s cleared
  for (int t = 0; t < n_trips; ++t) {
    const int y = row_of(t);
    const bool live = y < H;
    sq_f2 xv[CP], mv[CP], cvv[CP], mk[CP], gcv[CP], gm[CP];
#pragma unroll
    for (int q = 0; q < CP; ++q) {
      xv[q] = xn[q];
      mv[q] = mn[q];
    }
    const unsigned rm = live ? __builtin_amdgcn_readfirstlane(rm_n) : 0u;   // slots whose box meets this row
    request(row_of(t + 1));
    rm_n = c.rmask[min(row_of(t + 1), H - 1)];
    if (live) {
      if (rm == 0u) {
#pragma unroll
        for (int q = 0; q < CP; ++q) {
          mk[q] = sq_f2{m_bg, m_bg};
          cvv[q] = mv[q] * mk[q];
        }
      } else {
        float cvs[CPL];
        sq_f2 msw[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          cvs[q] = 0.0f;
          msw[q] = sq_f2{0.0f, 0.0f};
        }
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
          if (!(rm & (1u << k))) continue;   // (scalar bit test)
          const float4 yr = c.yrec[k * H + y];
          const int ro = __builtin_bit_cast(int, yr.x);
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const sq_f4 tp = *reinterpret_cast<const sq_f4a8*>(prb + (xo[k][q] + ro));
            const sq_f2 tt = sq_tap_rows(wab[k][q], tp.xy, tp.zw);
            cvs[q] = fmaf(yr.y, tt.x, cvs[q]);
            cvs[q] = fmaf(yr.z, tt.y, cvs[q]);
          }
          const sq_f2 yzw = {yr.z, yr.w};
#pragma unroll
          for (int q = 0; q < CPL; ++q)
            asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\ts_nop 0" : "+v"(msw[q]) : "v"(yzw), "v"(wab[k][q]));
        }
#pragma unroll
        for (int q = 0; q < CP; ++q) {
          mk[q] = sq_mask_sigmoid2(sq_f2{msw[2 * q].x + msw[2 * q].y, msw[2 * q + 1].x + msw[2 * q + 1].y});
          cvv[q] = sq_fma2(mv[q], mk[q], sq_f2{cvs[2 * q], cvs[2 * q + 1]});
        }
      }
      // pixel adjoints: g_cv (canvas), g_m (mask) and the row's contribution to d mean_img
#pragma unroll
      for (int q = 0; q < CP; ++q) {
        const sq_f2 df = xv[q] - cvv[q];
        if (ONE_SD) {
          gcv[q] = df * sq_f2{gll_isd2, gll_isd2};
          gm[q] = gcv[q] * mv[q];
        } else {
          const sq_f2 sd = mk[q] * a.std_fg + (sq_f2{1.0f, 1.0f} - mk[q]) * a.std_bg;
          gcv[q] = sq_f2{gll * df.x / (sd.x * sd.x), gll * df.y / (sd.y * sd.y)};
          const sq_f2 gsd = {gll * (df.x * df.x / (sd.x * sd.x * sd.x) - 1.0f / sd.x), gll * (df.y * df.y / (sd.y * sd.y * sd.y) - 1.0f / sd.y)};
          gm[q] = gcv[q] * mv[q] + gsd * (a.std_fg - a.std_bg);
        }
        const sq_f2 dm = gcv[q] * mk[q];
        float* o = a.d_mean_rows + frr * P + y * W + lane * CPL + 2 * q;
        if (FULLW) *reinterpret_cast<sq_f2*>(o) = dm;
        else {
          if (lane * CPL + 2 * q < W) o[0] = dm.x;
          if (lane * CPL + 2 * q + 1 < W) o[1] = dm.y;
          if (lane * CPL + 2 * q >= W) gcv[q].x = 0.0f;       // (columns past the frame: nothing flows back)
          if (lane * CPL + 2 * q + 1 >= W) gcv[q].y = 0.0f;
        }
      }
    }
    if (rm != 0u) {
      float gcs[CPL], gms[CPL];   // g_cv and the adjoint of the mask SUM: g_m 20 m (1 - m)
#pragma unroll
      for (int q = 0; q < CP; ++q) {
        const sq_f2 g2 = gm[q] * sq_f2{20.0f, 20.0f} * mk[q] * (sq_f2{1.0f, 1.0f} - mk[q]);
        gcs[2 * q] = gcv[q].x; gcs[2 * q + 1] = gcv[q].y;
        gms[2 * q] = g2.x; gms[2 * q + 1] = g2.y;
      }
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if (!(rm & (1u << k))) continue;
        const float4 yr = c.yrec[k * H + y], y2 = c.yrec2[k * H + y];
        const int ro = __builtin_bit_cast(int, yr.x);
        const int ti = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, y2.w));   // upper tap index of this row (scalar)
        if (cur[k] != ti) {   // the index moved on: finished rows leave the registers (after two steps both accumulators are empty)
          if (cur[k] >= 0)
            for (int st = 0; st < 2 && cur[k] != ti; ++st) advance(k);
          cur[k] = ti;
        }
        const float pdsum = y2.x + y2.y;
        float rowgy = 0.0f;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          const h2 dh = __builtin_bit_cast(h2, dabh[k][q]);
          const sq_f2 dab = {(float)dh.x, (float)dh.y};
          const sq_f4 tp = *reinterpret_cast<const sq_f4a8*>(prb + (xo[k][q] + ro));
          const sq_f2 ab = sq_tap_rows(wab[k][q], tp.xy, tp.zw);    // {A, B}: the two glimpse rows x-interpolated
          const sq_f2 dd = sq_tap_rows(dab, tp.xy, tp.zw);          // {d A / d xg, d B / d xg}
          const float dvdx = fmaf(yr.z, dd.y, yr.y * dd.x), dvdy = fmaf(y2.y, ab.y, y2.x * ab.x);
          const float gx = fmaf(gcs[q], dvdx, gms[q] * (yr.w * (dab.x + dab.y)));
          const float gy = fmaf(gcs[q], dvdy, gms[q] * (pdsum * (wab[k][q].x + wab[k][q].y)));
          sgx[k] += gx;
          sgxx[k] = fmaf(gx, xnrm[q], sgxx[k]);
          rowgy += gy;
        }
        sgy[k] += rowgy;
        sgyy[k] = fmaf(rowgy, y2.z, sgyy[k]);
#pragma unroll
        for (int q = 0; q < CP; ++q) {
          acc_a[k][q] = sq_fma2(sq_f2{yr.y, yr.y}, gcv[q], acc_a[k][q]);
          acc_b[k][q] = sq_fma2(sq_f2{yr.z, yr.z}, gcv[q], acc_b[k][q]);
        }
      }
    }
    if ((t % RB) == RB - 1 || t == n_trips - 1) {   // end of a block: the next block of this wave is 3 RB rows further down
#pragma unroll
      for (int k = 0; k < NMAX; ++k) {
        if (cur[k] < 0) continue;
        advance(k);
        advance(k);
        cur[k] = -1;
      }
    }
  }
  // coordinate gradients: wave sums -> LDS -> four values per slot
  const float hg = 0.5f * (float)(G - 1);
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    if (k >= N) break;
    const float sx = c.co[k * 4 + 0], sy = c.co[k * 4 + 1], tcx = c.co[k * 4 + 2], tcy = c.co[k * 4 + 3];
    const float s0 = sq_wave_sum(sgx[k]), s1 = sq_wave_sum(sgxx[k]), s2 = sq_wave_sum(sgy[k]), s3 = sq_wave_sum(sgyy[k]);
    if (lane == 0) {
      float* ac = acc_s + (wave * N + k) * 4;
      ac[0] = -hg / (sx * sx) * (s1 - tcx * s0);
      ac[1] = -hg / (sy * sy) * (s3 - tcy * s2);
      ac[2] = -hg / sx * s0;
      ac[3] = -hg / sy * s2;
    }
  }
  __syncthreads();   // dgl_s / acc_s complete
  float dsc = 0.0f;
  const float* prf = reinterpret_cast<const float*>(c.pr);
  for (int i = tid; i < N * G2; i += NT) {
    const float v = dgl_s[i];
    a.d_glimpse[fs * G2 + i] = v;
    const int row = sq_div(i, d.g_mul), x = i - row * G, k = sq_div(row, d.g_mul), yy = row - k * G;
    const float gv = yy < G - 1 ? prf[((k * (G - 1) + yy) * G + x) * 2] : prf[((k * (G - 1) + yy - 1) * G + x) * 2 + 1];
    dsc = fmaf(v, gv, dsc);
  }
  if (a.d_scale != nullptr) {
    __shared__ float dsc_s[4];
    dsc = sq_wave_sum(dsc);
    if (lane == 0) dsc_s[wave] = dsc;
    __syncthreads();
    if (tid == 0) unsafeAtomicAdd(a.d_scale, (dsc_s[0] + dsc_s[1] + dsc_s[2] + dsc_s[3]) / a.scale[0]);
  }
  if (tid < N * 4) {
    const int k = tid >> 2, q = tid & 3;
    const float tot = acc_s[(0 * N + k) * 4 + q] + acc_s[(1 * N + k) * 4 + q] + acc_s[(2 * N + k) * 4 + q] + acc_s[(3 * N + k) * 4 + q];
    const float l = a.rec ? a.rec[(fs + k) * a.rec_ld + rec::WHERE + q] : a.where[fs * 4 + tid];
    const float sg = sq_sigmoid_geo(l), th = tanhf(l);
    a.d_where[(fs + k) * a.dw_ld + q] = tot * ((q & 2) ? 1.0f - th * th : sg * (1.0f - sg));
  }
}
static inline size_t insert_bwd_rows_lds_bytes(const Dims& d) {
  const int lw = d.W <= 128 ? 128 : 256;
  const size_t un_a = 4 * (size_t)d.N * d.W, un_b = (size_t)(((size_t)d.N * d.G * d.G + 3) & ~(size_t)3) + 4 * 2 * (size_t)(lw + 16);
  return (sq_canvas_rows_lds_floats(d.N, d.G, d.H, true) + (size_t)d.N * d.G + ((16 * (size_t)d.N + 3) & ~(size_t)3) + (un_a > un_b ? un_a : un_b)) * sizeof(float);
}
// Frames of 65 .. 128 columns with up to 4 slots (BASELINE configs[4]).  The other instantiations were measured and lose to the band
// kernel (1600 workgroups, back to back): 8 slots at 128 x 128 474 us against 396, 4 slots at 96 x 200 (four columns per lane) 419
// against 209, 6 slots there 1120 against 376 -- the per-slot register state (taps, accumulators, running sums) no longer fits.
static inline bool insert_bwd_use_rows(const Dims& d) {
  return d.W > SQ_CANVAS_WIDE && d.W <= 128 && d.N <= 4 && d.G >= 2 && d.G <= 20 && insert_bwd_rows_lds_bytes(d) <= 150 * 1024;
}
template <int NMAX, int CPL, bool FULLW, bool ONE_SD>
static int launch_insert_bwd_rows3(const InsertBwdArgs& a, const Dims& d, dim3 grid, hipStream_t s) {
  const size_t shm = insert_bwd_rows_lds_bytes(d);
  if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_insert_loglik_bwd_rows<NMAX, CPL, FULLW, ONE_SD>, 150 * 1024) != 0) return -2;
  SQ_LAUNCH((k_insert_loglik_bwd_rows<NMAX, CPL, FULLW, ONE_SD>), grid, dim3(256), shm, s, a, d);
  return 0;
}
template <int NMAX, int CPL>
static int launch_insert_bwd_rows2(const InsertBwdArgs& a, const Dims& d, dim3 grid, hipStream_t s) {
  const size_t al = (size_t)CPL * 4 - 1;
  const bool fullw = d.W == 64 * CPL && (((size_t)a.img | (size_t)a.mean_img) & al) == 0 && (d.P4 * 4 & al) == 0 && ((size_t)a.d_mean_rows & 7) == 0;
  const bool one_sd = a.std_fg == a.std_bg;
  if (fullw) return one_sd ? launch_insert_bwd_rows3<NMAX, CPL, true, true>(a, d, grid, s) : launch_insert_bwd_rows3<NMAX, CPL, true, false>(a, d, grid, s);
  return one_sd ? launch_insert_bwd_rows3<NMAX, CPL, false, true>(a, d, grid, s) : launch_insert_bwd_rows3<NMAX, CPL, false, false>(a, d, grid, s);
}
static int launch_insert_bwd_rows(const InsertBwdArgs& a, const Dims& d, dim3 grid, hipStream_t s) {
  return launch_insert_bwd_rows2<4, 2>(a, d, grid, s);
}

// band height and dynamic LDS of k_insert_loglik_bwd
static size_t insert_bwd_lds(const Dims& d, int& band_rows) {
  const bool wide = d.W > SQ_CANVAS_WIDE;
  band_rows = sq_canvas_band_rows(d.H, d.W, wide ? SQ_CANVAS_PF_BWD_W : SQ_CANVAS_PF_BWD);
  const size_t bytes = insert_bwd_lds_floats(d, band_rows) * sizeof(float);
  const void* kern = wide ? (const void*)k_insert_loglik_bwd<SQ_CANVAS_PF_BWD_W, SQ_CANVAS_ROWS_BWD_W> : (const void*)k_insert_loglik_bwd<SQ_CANVAS_PF_BWD, SQ_CANVAS_ROWS_BWD>;
  if (bytes > 48 * 1024 && sq_allow_big_lds(kern, 150 * 1024) != 0) return 0;  // 0 = failed
  return bytes;
}
#define SQ_LAUNCH_INSERT_BWD(grid, shm, s, a, d, band_rows)                                                                         \
  do {                                                                                                                                \
    if ((d).W > SQ_CANVAS_WIDE) SQ_LAUNCH((k_insert_loglik_bwd<SQ_CANVAS_PF_BWD_W, SQ_CANVAS_ROWS_BWD_W>), grid, dim3(256), shm, s, a, d, band_rows); \
    else SQ_LAUNCH((k_insert_loglik_bwd<SQ_CANVAS_PF_BWD, SQ_CANVAS_ROWS_BWD>), grid, dim3(256), shm, s, a, d, band_rows);            \
  } while (0)
__global__ void k_reduce_rows(const float* __restrict__ rows, float* __restrict__ out, int R, int P, int accumulate SQ_TLP) {
  SQ_TL_SCOPE;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float acc = accumulate ? out[p] : 0.0f;
  for (int r = 0; r < R; ++r) acc += rows[(size_t)r * P + p];
  out[p] = acc;
}
// row-chunked variant for the training step: out[p] += sum over this block's rows (float atomics, out pre-zeroed)
__global__ void k_reduce_rows_atomic(const float* __restrict__ rows, float* __restrict__ out, int R, int P, int rows_per_block SQ_TLP) {
  SQ_TL_SCOPE;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float acc = 0.0f;
  for (int r = r0; r < r1; ++r) acc += rows[(size_t)r * P + p];
  unsafeAtomicAdd(out + p, acc);
}

extern "C" int sqair_st_insert_loglik_bwd(SqairHandle* h, const float* glimpse, const float* where_logits,
                                          const float* presence, const float* img, const float* mean_img,
                                          const float* g_data_ll, float* d_glimpse, float* d_where_logits,
                                          float* d_mean_img, void* scratch, int64_t scratch_bytes, int B, void* stream) {
  if (!h || !glimpse || !where_logits || !presence || !img || !mean_img || !g_data_ll || !d_glimpse || !d_where_logits ||
      !scratch || B < 1)
    return -1;
  SqairConfig c;
  if (sqair_get_config(h, &c) != 0) return -1;
  Dims d = make_dims(c, B);
  const int P = d.H * d.W;
  if (scratch_bytes < (int64_t)d.R * P * 4) return -1;
  InsertBwdArgs a{glimpse, where_logits, presence, img, mean_img, g_data_ll, d_glimpse, d_where_logits, (float*)scratch,
                  c.output_std, c.background_std, nullptr, 0, 4};
  if (insert_bwd_use_rows(d)) {
    if (launch_insert_bwd_rows(a, d, dim3(d.R, 1), (hipStream_t)stream) != 0) return -2;
  } else {
    int band_rows;
    const size_t shm = insert_bwd_lds(d, band_rows);
    if (shm == 0) return -2;
    SQ_LAUNCH_INSERT_BWD(dim3(d.R, 1), shm, (hipStream_t)stream, a, d, band_rows);
  }
  if (d_mean_img)   // (NULL: the per-row contributions stay in `scratch`, [R, H * W]; what tools/time_insert.py times)
    SQ_LAUNCH(k_reduce_rows, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)scratch,
                       d_mean_img, d.R, P, 0);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// VIMCO target backward (targets.py:62-75, model.py:150-158): L = mean_{b,k}(-elbo_b - sig_bk * dl_bk) / T with
// sig a stop-gradient  =>  dL/d log_w[t, b, k] = -softmax_k(log_w_b)[k] / (B T),  dL/d disc_lp[t, b, k] = -sig_bk / (B K T)
// ------------------------------------------------------------------------------------------------
__global__ void k_elbo_bwd(const float* __restrict__ iw, const float* __restrict__ sig, int T, int B, int K,
                           float* __restrict__ g_log_w_t, float* __restrict__ g_disc_lp_t SQ_TLP) {
  SQ_TL_SCOPE;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int R = B * K;
  if (i >= T * R) return;
  const int rk = i % R;
  g_log_w_t[i] = -iw[rk] / ((float)B * (float)T);
  g_disc_lp_t[i] = -sig[rk] / ((float)B * (float)K * (float)T);
}

extern "C" int sqair_elbo_bwd(SqairHandle* h, const float* importance_weights, const float* vimco_signal, int T, int B,
                              float* g_log_w_t, float* g_disc_lp_t, void* stream) {
  if (!h || !importance_weights || !vimco_signal || !g_log_w_t || !g_disc_lp_t || T < 1 || B < 1) return -1;
  SqairConfig c;
  if (sqair_get_config(h, &c) != 0) return -1;
  const int n = T * B * c.k_particles;
  SQ_LAUNCH(k_elbo_bwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, importance_weights,
                     vimco_signal, T, B, c.k_particles, g_log_w_t, g_disc_lp_t);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a dense layer on the fp32 matrix cores: dW[k][n] (+)= sum_m A[m][k] dY[m][n], reference
// layout [in, out] with leading dimension ldw; db[n] (+)= sum_m dY[m][n].  One workgroup per 16(k) x 16(n) tile, its
// 4 waves split the M rows (chunks of 4 rows per MFMA: A^T fragment lane l = A[m0 + (l>>4)][k0 + (l&15)],
// dY fragment lane l = dY[m0 + (l>>4)][n0 + (l&15)] — both coalesced 64-byte row pieces), LDS reduce.
// M is large when the tape of a whole step is reduced at once (T*N*B' = 6400 rows), which is how the training
// step uses it: one launch per layer per step, off the critical path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ A, int lda, const float* __restrict__ dY, int ldy,
                                               float* __restrict__ dW, int ldw, float* __restrict__ db, int M, int Kdim,
                                               int Ndim, int accumulate, const int* __restrict__ rowmap,
                                               const float* __restrict__ alpha_ptr SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ float red[4 * 256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int k0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  const int kk = min(k0 + (lane & 15), Kdim - 1), nn = min(n0 + (lane & 15), Ndim - 1);
  const int mq = lane >> 4;
  f32x4_b acc = {0.0f, 0.0f, 0.0f, 0.0f};
  float bsum = 0.0f;
  for (int m0 = wave * 4; m0 < M; m0 += 16 * 4) {  // 4 MFMAs (16 rows) per wave per trip, loads issued together
    float av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + q * 16 + mq;
      const int mc = min(m, M - 1);
      const float x = A[(size_t)mc * lda + kk], y = dY[(size_t)mc * ldy + nn];
      av[q] = m < M ? x : 0.0f;
      bv[q] = m < M ? y : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);
      bsum += bv[q];
    }
  }
  // acc[i] of lane l = dW tile [row k = 4*(l>>4) + i][col n = l & 15]
  float* r = red + wave * 256;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[(4 * mq + i) * 16 + (lane & 15)] = acc[i];
  __syncthreads();
  const float alpha = alpha_ptr != nullptr ? alpha_ptr[0] : 1.0f;
  const float v = (red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid]) * alpha;
  const int k = k0 + (tid >> 4), n = n0 + (tid & 15);
  if (k < Kdim && n < Ndim) {
    const int row = rowmap != nullptr ? rowmap[k] : k;  // A-segment position -> row of the reference matrix (-1: none)
    if (row >= 0) {
      float* p = dW + (size_t)row * ldw + n;
      *p = accumulate ? *p + v : v;
    }
  }
  if (db != nullptr && blockIdx.x == 0) {
    // column sums: lanes with the same (l & 15) across mq and waves
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    __syncthreads();
    if (lane < 16) red[wave * 16 + lane] = bsum;
    __syncthreads();
    if (tid < 16 && n0 + tid < Ndim) {
      const float s = (red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid]) * alpha;
      db[n0 + tid] = accumulate ? db[n0 + tid] + s : s;
    }
  }
}

int sq_launch_wgrad(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, float* db, int M, int Kdim,
                    int Ndim, int accumulate, hipStream_t s, const int* rowmap, const float* alpha_ptr) {
  SQ_LAUNCH(k_wgrad, dim3((Kdim + 15) / 16, (Ndim + 15) / 16), dim3(256), 0, s, A, lda, dY, ldy, dW, ldw, db, M,
                     Kdim, Ndim, accumulate, rowmap, alpha_ptr);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient for the training step: dW[rowmap[k]][n] += alpha * sum_m A[m][k] dY[m][n] (float atomics onto a
// zeroed gradient buffer), db_a[n], db_b[n] += alpha * sum_m dY[m][n].  Grid = (K/32, N/32, M-chunks): every wave
// owns a 32(k) x 32(n) macro tile (4 accumulators: 2 A + 2 dY fragment loads feed 4 MFMAs), the 4 waves of a
// workgroup split the rows of the workgroup's M-chunk and reduce through LDS; the M range is split over
// blockIdx.z so that a layer with few tiles (e.g. 256 x 8) still fills the chip.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wgrad2(const float* __restrict__ A, int lda, const float* __restrict__ dY, int ldy,
                                                float* __restrict__ dW, int ldw, int M, int Kdim, int Ndim,
                                                const int* __restrict__ rowmap, const float* __restrict__ alpha_ptr,
                                                float* __restrict__ db_a, float* __restrict__ db_b, int m_per_wg SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ float red[4 * 1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, mq = lane >> 4;
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int mbeg = blockIdx.z * m_per_wg, mend = min(M, mbeg + m_per_wg);
  const int ka = min(k0 + l15, Kdim - 1), kb = min(k0 + 16 + l15, Kdim - 1);
  const int na = min(n0 + l15, Ndim - 1), nb = min(n0 + 16 + l15, Ndim - 1);
  const bool want_bias = (db_a != nullptr || db_b != nullptr) && blockIdx.x == 0;
  f32x4_b c00 = {0.0f, 0.0f, 0.0f, 0.0f}, c01 = c00, c10 = c00, c11 = c00;
  float bs0 = 0.0f, bs1 = 0.0f;
  for (int mb = mbeg + wave * 16; mb < mend; mb += 64) {
    float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = mb + q * 4 + mq;
      const int mc = min(m, mend - 1);
      const float* ar = A + (size_t)mc * lda;
      const float* yr = dY + (size_t)mc * ldy;
      const float x0 = ar[ka], x1 = ar[kb], y0 = yr[na], y1 = yr[nb];
      const bool ok = m < mend;
      a0[q] = ok ? x0 : 0.0f; a1[q] = ok ? x1 : 0.0f; b0[q] = ok ? y0 : 0.0f; b1[q] = ok ? y1 : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b0[q], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], b1[q], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b0[q], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], b1[q], c11, 0, 0, 0);
      bs0 += b0[q];
      bs1 += b1[q];
    }
  }
  // acc[i] of lane l = tile [row k = 4*(l>>4) + i][col n = l & 15]; tiles ordered (kb, nb) = 00, 01, 10, 11
  float* r = red + wave * 1024;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[0 * 256 + (4 * mq + i) * 16 + l15] = c00[i];
    r[1 * 256 + (4 * mq + i) * 16 + l15] = c01[i];
    r[2 * 256 + (4 * mq + i) * 16 + l15] = c10[i];
    r[3 * 256 + (4 * mq + i) * 16 + l15] = c11[i];
  }
  __syncthreads();
  const float alpha = alpha_ptr != nullptr ? alpha_ptr[0] : 1.0f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = t * 256 + tid;
    const float v = (red[idx] + red[1024 + idx] + red[2048 + idx] + red[3072 + idx]) * alpha;
    const int k = k0 + (t >> 1) * 16 + (tid >> 4), n = n0 + (t & 1) * 16 + (tid & 15);
    if (k < Kdim && n < Ndim) {
      const int row = rowmap != nullptr ? rowmap[k] : k;
      if (row >= 0) unsafeAtomicAdd(dW + (size_t)row * ldw + n, v);
    }
  }
  if (want_bias) {
    bs0 += __shfl_xor(bs0, 16, 64); bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 16, 64); bs1 += __shfl_xor(bs1, 32, 64);
    __syncthreads();
    if (lane < 16) { red[wave * 32 + lane] = bs0; red[wave * 32 + 16 + lane] = bs1; }
    __syncthreads();
    if (tid < 32 && n0 + tid < Ndim) {
      const float sb = (red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid]) * alpha;
      if (db_a != nullptr) unsafeAtomicAdd(db_a + n0 + tid, sb);
      if (db_b != nullptr) unsafeAtomicAdd(db_b + n0 + tid, sb);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same product with a 64(k) x 64(n) macro tile per WAVE (wgrad3_body, sqair_wgrad_kernel.inc): 16 accumulators over the
// wave's own row range of the workgroup's M-chunk, so one 16-byte load of A and one of dY per 4 rows feed 16 MFMAs (k_wgrad2:
// 4 dword loads per 4 MFMAs), the loads run two 16-row steps (128 MFMAs) ahead of their use, and a workgroup issues a quarter
// of k_wgrad2's atomics per flop.  The MFMA tile (i, j) of the macro tile holds k = 4 row + i, n = 4 col + j (interleaved),
// which is what makes the operands of the four tiles one contiguous float4.  The 4 waves' tiles meet in LDS; a thread adds 16
// sums onto the gradient with row-contiguous atomics.  Needs 16-byte aligned operand rows (wgrad3_operands_ok).
// Measured (tools/wgrad_floor.hip, 256 x 256 block): 92 TFLOP/s of the 155 TFLOP/s the fp32 matrix cores deliver at
// M = 51200; at M = 6400 a launch of its own is 20 us for 5 us of MFMAs -- hence the grouped launch below.
// Built with -amdgpu-mfma-vgpr-form: with the accumulators in AGPRs the compiler moved all 64 of them to VGPRs and back
// around every loop trip.
// ------------------------------------------------------------------------------------------------
#define SQ_KWGRAD_NAME k_wgrad3
#define SQ_KWGRAD_BODY wgrad3_body
#include "sqair_wgrad_kernel.inc"
#undef SQ_KWGRAD_NAME
#undef SQ_KWGRAD_BODY
// the same body for the workgroups of k_wgrad_group
#ifndef SQ_WGRAD_GROUP_WAVES
#define SQ_WGRAD_GROUP_WAVES 4
#endif
#define SQ_KWGRAD_BODY wgrad1_body
#define SQ_KWGRAD_WAVES SQ_WGRAD_GROUP_WAVES
#include "sqair_wgrad_kernel.inc"
#undef SQ_KWGRAD_BODY
#undef SQ_KWGRAD_WAVES

// All deferred blocks of a training step in ONE launch (WgradBatch, sqair_bwd.h): the blocks of the ~50 layers are
// independent, each is far too small to fill the chip for long (a 256 x 256 block over 6400 rows is 0.84 GFLOP = 5 us of the
// matrix cores, measured 20 us as its own launch: ramp-up, first-load latency, the LDS meeting and the atomics' drain are
// each as long as the MFMAs), and back to back on one stream those fixed parts added up to 1.2 ms of a 9.9 ms step.  As one
// grid of workgroups (a 64 x 64 tile x m_per_wg rows each) they overlap each other's fixed parts.  The table of blocks
// travels in the kernel arguments (15 KB; 32 KB arguments were checked to work, also inside a captured graph).
//
// Placement.  The tiles of an M-chunk read the same operand rows, and workgroups are handed to the 8 XCDs round-robin: with
// the tile index fastest, the 16 tiles of a 256 x 256 block's chunk sat on 8 XCDs and every L2 pulled its own copy of the rows
// over the fabric (PMC: 1.19 GB per launch at cfg-2, 2.4 GB per step in 0.46 ms -- at the fabric's limit).  Now the host deals
// GROUPS (a chunk's tiles, at most 32 of them) to eight per-XCD queues, always to the queue with the least work so far;
// workgroup b runs entry b / 8 of the queue of XCD b % 8, so a group's workgroups are dispatched back to back onto one XCD and
// walk the rows together: one of them misses, the others hit (0.37 GB per launch).
__global__ __launch_bounds__(64 * SQ_WGRAD_GROUP_WAVES) void k_wgrad_group(const WgGroup g SQ_TLP) {
  SQ_TL_SCOPE;
  extern __shared__ __attribute__((aligned(16))) float red3[];
  const int x = (int)blockIdx.x & 7, pos = (int)blockIdx.x >> 3;
  // the block this queue entry belongs to: all starts of the queue in one batch of scalar loads, then a count (a serial walk
  // was one dependent scalar load per block ahead of the workgroup's first operand request)
  int i = 0;
#pragma unroll
  for (int j = 1; j <= SQ_WG_MAXD; ++j) i += g.begin[x][j] <= pos ? 1 : 0;
  if (i >= g.nd) return;  // past the end of this XCD's queue
  const WgDesc& e = g.d[i];
  const int rel = pos - g.begin[x][i];
  const int m = rel / e.gs, tl = rel - m * e.gs;
  const int q = (int)e.qb[x] + m;          // which of the block's zc * tg groups
  const int zi = q / e.tg, r = q - zi * e.tg;
  const int tile = r * e.gs + tl;
  if (tile >= e.n_tiles) return;           // the last group of a chunk may be short
  wgrad1_body(e.A, e.lda, e.dY, e.ldy, e.dW, e.ldw, e.M, e.Kdim, e.Ndim, e.rowmap, e.alpha_ptr, e.db_a, e.db_b, e.m_per_wg, e.kt, tile, zi, red3);
}
// what the float4 operand loads of wgrad3_body need: 16-byte aligned rows, and a width that is not a multiple of 4 padded
// inside its row
static bool wgrad3_operands_ok(const float* A, int lda, const float* dY, int ldy, int Kdim, int Ndim) {
  static const int use3 = SQ_KNOB_INT("SQAIR_WGRAD3", 1);  // measurement knob: 0 = k_wgrad2 everywhere
  const bool aligned = ((uintptr_t)A % 16 == 0) && ((uintptr_t)dY % 16 == 0) && lda % 4 == 0 && ldy % 4 == 0;
  const bool padded = (Kdim % 4 == 0 || ((Kdim + 3) & ~3) <= lda) && (Ndim % 4 == 0 || ((Ndim + 3) & ~3) <= ldy);
  return use3 > 0 && aligned && padded && Kdim >= 1 && Ndim >= 1;
}
// as its own launch the 64 x 64 macro tile only pays for blocks about that large
static bool wgrad3_eligible(const float* A, int lda, const float* dY, int ldy, int M, int Kdim, int Ndim) {
  return wgrad3_operands_ok(A, lda, dY, ldy, Kdim, Ndim) && Kdim % 4 == 0 && Ndim % 4 == 0 && Kdim >= 48 && Ndim >= 48 && M >= 256;
}
static const size_t WG3_LDS = (4 * 4096 + 256) * sizeof(float);
static const size_t WG1_LDS = SQ_WGRAD_GROUP_WAVES * (4096 + 64) * sizeof(float);  // k_wgrad_group: a tile + its column sums per wave
bool WgradBatch::add(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, int M, int Kdim, int Ndim, const int* rowmap,
                     const float* alpha_ptr, float* db_a, float* db_b) {
  static const bool grouped = SQ_KNOB_INT("SQAIR_WGRAD_GROUP", 1) != 0;
  if (!grouped || M < 1 || !wgrad3_operands_ok(A, lda, dY, ldy, Kdim, Ndim)) return false;
  WgDesc e;
  e.A = A; e.dY = dY; e.dW = dW; e.rowmap = rowmap; e.alpha_ptr = alpha_ptr; e.db_a = db_a; e.db_b = db_b;
  e.lda = lda; e.ldy = ldy; e.ldw = ldw; e.M = M; e.Kdim = Kdim; e.Ndim = Ndim;
  e.kt = (Kdim + 63) / 64; e.n_tiles = e.kt * ((Ndim + 63) / 64);
  blocks.push_back(e);
  return true;
}
// The launch as the host sees it: eight queues (one per XCD) of workgroups.  wg_plan() cuts every block into chunks of about
// `rows` rows and deals the groups.  Chunk length: a workgroup's life is a fixed part (ramp, tile to LDS, sums, atomics,
// turnover: ~8 us) plus ~0.033 us per row (tools/wgrad_floor.hip, stamped launch), so long chunks amortise the fixed part and
// short ones fill the last round.  A list-scheduling model of that (per-XCD queues, 64 slots each) preferred 2560-3200 rows;
// measured (A / B of the training step, chunk length 800 ... 3200) the optimum is ~1280 at cfg-2, cfg-4 and at 256 sequences
// per GPU alike (cfg-2: 800 +22 us, 1280 0, 1600 +3, 2048 +20, 2560 +18, 3200 +29), so 1280 it is and the model is gone.
static const double WG_FIXED_US = 8.0, WG_US_PER_ROW = 0.033;   // only the RATIO matters: the relative cost of groups when dealing
static int wg_plan(std::vector<WgDesc>& bl, size_t i0, int nd, int rows, WgGroup* g, int* grid) {
  double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int len[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nd; ++i) {
    WgDesc& e = g->d[i];
    e = bl[i0 + i];
    e.zc = std::max(1, (e.M + rows / 2) / rows);
    e.m_per_wg = ((e.M + e.zc - 1) / e.zc + 15) / 16 * 16;   // 4 waves x a multiple of 4 rows; the kernel masks a partial last step
    e.zc = (e.M + e.m_per_wg - 1) / e.m_per_wg;
    e.tg = (e.n_tiles + 31) / 32;
    e.gs = (e.n_tiles + e.tg - 1) / e.tg;
  }
  // long workgroups first (stable: blocks of one length keep the order of the backward pass)
  std::stable_sort(g->d, g->d + nd, [](const WgDesc& a, const WgDesc& b) { return a.m_per_wg > b.m_per_wg; });
  for (int i = 0; i < nd; ++i) {
    WgDesc& e = g->d[i];
    const double cost = e.gs * (WG_FIXED_US + WG_US_PER_ROW * e.m_per_wg);
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < e.zc * e.tg; ++q) {
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (load[x] < load[best]) best = x;
      ++cnt[best];
      load[best] += cost;
    }
    int acc = 0;
    for (int x = 0; x < 8; ++x) {
      g->begin[x][i] = len[x];
      e.qb[x] = (unsigned short)acc;
      acc += cnt[x];
      len[x] += cnt[x] * e.gs;
    }
    if (acc > 65535) return -1;
  }
  int longest = 0;
  for (int x = 0; x < 8; ++x) {
    for (int j = nd; j < SQ_WG_MAXD + 8; ++j) g->begin[x][j] = j == nd ? len[x] : INT_MAX;
    longest = std::max(longest, len[x]);
  }
  g->nd = nd;
  *grid = 8 * longest;
  return 0;
}
int WgradBatch::flush(hipStream_t s) {
  static const int rows = SQ_KNOB_INT("SQAIR_WGRAD_ROWS", 1280);  // target rows of a workgroup
  static const bool dump = SQ_KNOB_SET("SQAIR_WGRAD_DUMP");  // print the block table of every flush
  if (WG1_LDS > 65536 && sq_allow_big_lds((const void*)k_wgrad_group, (int)WG1_LDS) != 0) return -2;
  for (size_t i0 = 0; i0 < blocks.size(); i0 += SQ_WG_MAXD) {
    const int nd = (int)std::min(blocks.size() - i0, (size_t)SQ_WG_MAXD);
    static thread_local WgGroup g;   // 15 KB
    int grid = 0;
    if (wg_plan(blocks, i0, nd, rows, &g, &grid) != 0) return -3;
    if (dump) {
      fprintf(stderr, "wgrad launch: %d blocks, chunk length %d, %d workgroups\n", nd, rows, grid);
      for (int i = 0; i < nd; ++i)
        fprintf(stderr, "wgrad block %2d: M %6d K %4d N %4d  tiles %3d  chunks %3d x %4d rows  useful %.2f\n", i, g.d[i].M, g.d[i].Kdim,
                g.d[i].Ndim, g.d[i].n_tiles, g.d[i].zc, g.d[i].m_per_wg, (double)g.d[i].Kdim * g.d[i].Ndim / (4096.0 * g.d[i].n_tiles));
    }
    SQ_LAUNCH(k_wgrad_group, dim3(grid), dim3(64 * SQ_WGRAD_GROUP_WAVES), WG1_LDS, s, g);
  }
  blocks.clear();
  return 0;
}
int sq_launch_wgrad_acc(const float* A, int lda, const float* dY, int ldy, float* dW, int ldw, int M, int Kdim, int Ndim,
                        hipStream_t s, const int* rowmap, const float* alpha_ptr, float* db_a, float* db_b) {
  static const int skip = SQ_KNOB_INT("SQAIR_WGRAD3", 1) < 0;  // measurement knob (knob builds only): no weight gradients at all
  const int wg3_target = 512;
  if (skip) return 0;
  if (wgrad3_eligible(A, lda, dY, ldy, M, Kdim, Ndim)) {
    const int kt = (Kdim + 63) / 64, nt = (Ndim + 63) / 64;
    int zc = wg3_target / (kt * nt);
    const int max_z = (M + 63) / 64;
    if (zc > max_z) zc = max_z;
    if (zc < 1) zc = 1;
    const int m_per_wg = ((M + zc - 1) / zc + 63) / 64 * 64;   // 4 waves x a multiple of 16 rows
    zc = (M + m_per_wg - 1) / m_per_wg;
    if (sq_allow_big_lds((const void*)k_wgrad3, (int)WG3_LDS) != 0) return -2;
    SQ_LAUNCH(k_wgrad3, dim3(kt * nt * zc), dim3(256), WG3_LDS, s, A, lda, dY, ldy, dW, ldw, M, Kdim, Ndim, rowmap, alpha_ptr, db_a,
                       db_b, m_per_wg, kt, kt * nt);
    return 0;
  }
  const int kt = (Kdim + 31) / 32, nt = (Ndim + 31) / 32;
  static const int wg_target = SQ_KNOB_INT("SQAIR_WGRAD_WGS", 2048);  // measurement knob
  int zc = wg_target / (kt * nt);
  const int max_z = (M + 63) / 64;
  if (zc > max_z) zc = max_z;
  if (zc < 1) zc = 1;
  int m_per_wg = ((M + zc - 1) / zc + 63) / 64 * 64;
  zc = (M + m_per_wg - 1) / m_per_wg;
  SQ_LAUNCH(k_wgrad2, dim3(kt, nt, zc), dim3(256), 0, s, A, lda, dY, ldy, dW, ldw, M, Kdim, Ndim, rowmap, alpha_ptr,
                     db_a, db_b, m_per_wg);
  return 0;
}

// batched insert/log-likelihood adjoint over T frames on merged slot records (decoder branch of sqair_backward)
int sq_launch_insert_bwd_frames(const float* glimpse, const float* rec, int rec_ld, const float* img, const float* mean_img,
                                const float* g_ll, float* d_glimpse, float* d_rec, int d_rec_ld, float* d_mean_rows,
                                float std_fg, float std_bg, int T, Dims d, hipStream_t s, const float* scale, float* d_scale) {
  InsertBwdArgs a{glimpse, nullptr, nullptr, img, mean_img, g_ll, d_glimpse, d_rec, d_mean_rows, std_fg, std_bg, rec, rec_ld,
                  d_rec_ld, scale, d_scale};
  if (insert_bwd_use_rows(d)) return launch_insert_bwd_rows(a, d, dim3(d.R, T), s);
  int band_rows;
  const size_t shm = insert_bwd_lds(d, band_rows);
  if (shm == 0) return -2;
  SQ_LAUNCH_INSERT_BWD(dim3(d.R, T), shm, s, a, d, band_rows);
  return 0;
}
int sq_launch_reduce_rows(const float* rows, float* out, int R, int P, int accumulate, hipStream_t s) {
  SQ_LAUNCH(k_reduce_rows, dim3((P + 255) / 256), dim3(256), 0, s, rows, out, R, P, accumulate);
  return 0;
}
int sq_launch_reduce_rows_atomic(const float* rows, float* out, int R, int P, hipStream_t s) {
  const int rpb = 32;
  SQ_LAUNCH(k_reduce_rows_atomic, dim3((P + 255) / 256, (R + rpb - 1) / rpb), dim3(256), 0, s, rows, out, R, P, rpb);
  return 0;
}
int sq_launch_elbo_bwd(const float* iw, const float* sig, int T, int B, int K, float* g_lw, float* g_dl, hipStream_t s) {
  const int n = T * B * K;
  SQ_LAUNCH(k_elbo_bwd, dim3((n + 255) / 256), dim3(256), 0, s, iw, sig, T, B, K, g_lw, g_dl);
  return 0;
}

// d(output_scale) = sum(d_glimpse * glimpse) / scale   (glimpse = scale * raw; modules.py:144-147)
__global__ void k_dot_scale(const float* __restrict__ a, const float* __restrict__ b, int64_t n, const float* __restrict__ scale,
                            float* __restrict__ out SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ float red[16];
  float acc = 0.0f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += a[i] * b[i];
  acc = sq_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int i = 0; i < 16; ++i) t += red[i];
    out[0] = t / scale[0];
  }
}
int sq_launch_dot_scale(const float* a, const float* b, int64_t n, const float* scale, float* out, hipStream_t s) {
  SQ_LAUNCH(k_dot_scale, dim3(1), dim3(1024), 0, s, a, b, n, scale, out);
  return 0;
}

// Elementwise adjoint of the fused activation epilogue: dPre = dOut * act'(.) expressed through the saved OUTPUT
// (elu: out > 0 ? 1 : out + 1; tanh: 1 - out^2; sigmoid: out (1 - out); softplus(x) + c: 1 - exp(-(out - c))).
__global__ void k_dact(const float* __restrict__ d_out, const float* __restrict__ out, float* __restrict__ d_pre, int64_t n,
                       int act SQ_TLP) {
  SQ_TL_SCOPE;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float o = out[i];
  float g = d_out[i];
  switch (act) {
    case ACT_ELU: g *= o > 0.0f ? 1.0f : o + 1.0f; break;
    case ACT_TANH: g *= 1.0f - o * o; break;
    case ACT_SIGMOID: g *= o * (1.0f - o); break;
    case ACT_SOFTPLUS_MIN: g *= 1.0f - expf(-(o - 1e-2f)); break;
    default: break;
  }
  d_pre[i] = g;
}

int sq_launch_dact(const float* d_out, const float* out, float* d_pre, int64_t n, int act, hipStream_t s) {
  SQ_LAUNCH(k_dact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_out, out, d_pre, n, act);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Fused optimiser step on the flat buffers: tf.train.RMSPropOptimizer(lr, momentum=0.9)
// as the reference's driver uses it (sqair/scripts/experiment.py:140): decay 0.9, epsilon 1e-10, ms0 = 1,
//   ms <- rho ms + (1 - rho) g^2 ;  mom <- m mom + lr g / sqrt(ms + eps) ;  theta <- theta - mom.
// One pass over the 2.95 M floats (HBM-bound: 5 streams x 11.8 MB).  grad_scale folds the 1/world of the
// data-parallel all-reduce(sum) into the same pass.
// ------------------------------------------------------------------------------------------------
__global__ void k_rmsprop(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ ms,
                          float* __restrict__ mom, int64_t n, float lr, float rho, float momentum, float eps,
                          float grad_scale SQ_TLP) {
  SQ_TL_SCOPE;
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n && ((reinterpret_cast<uintptr_t>(theta) | reinterpret_cast<uintptr_t>(grad) |
                      reinterpret_cast<uintptr_t>(ms) | reinterpret_cast<uintptr_t>(mom)) & 15) == 0) {
    float4 t = *reinterpret_cast<float4*>(theta + i4);
    const float4 g4 = *reinterpret_cast<const float4*>(grad + i4);
    float4 s = *reinterpret_cast<float4*>(ms + i4);
    float4 m = *reinterpret_cast<float4*>(mom + i4);
    float* tp = &t.x; const float* gp = &g4.x; float* sp = &s.x; float* mp = &m.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = gp[j] * grad_scale;
      sp[j] = rho * sp[j] + (1.0f - rho) * g * g;
      mp[j] = momentum * mp[j] + lr * g / sqrtf(sp[j] + eps);
      tp[j] -= mp[j];
    }
    *reinterpret_cast<float4*>(theta + i4) = t;
    *reinterpret_cast<float4*>(ms + i4) = s;
    *reinterpret_cast<float4*>(mom + i4) = m;
  } else {
    for (int64_t i = i4; i < n && i < i4 + 4; ++i) {
      const float g = grad[i] * grad_scale;
      ms[i] = rho * ms[i] + (1.0f - rho) * g * g;
      mom[i] = momentum * mom[i] + lr * g / sqrtf(ms[i] + eps);
      theta[i] -= mom[i];
    }
  }
}

extern "C" int sqair_rmsprop_step(SqairHandle* h, float* flat_params, const float* flat_grad, float* ms, float* mom,
                                  int64_t n, float lr, float decay, float momentum, float epsilon, float grad_scale,
                                  void* stream) {
  if (!h || !flat_params || !flat_grad || !ms || !mom || n < 1) return -1;
  const int64_t nthreads = (n + 3) / 4;
  SQ_LAUNCH(k_rmsprop, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat_params,
                     flat_grad, ms, mom, n, lr, decay, momentum, epsilon, grad_scale);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// =================================================================================================
// Adjoint kernels of the recurrent part.  Gradients w.r.t. slot quantities live in "gradient records" with the
// layout of the forward slot records (rec::WHERE, WHAT, LOGIT, WHERE_LOC, WHERE_SCALE, WHAT_LOC, WHAT_SCALE), so the
// dX GEMMs of z-record segments (which come out in record order) accumulate straight into them.  Gradients of the
// small parameters are accumulated with float atomics into the flat gradient buffer.
// =================================================================================================

__device__ __forceinline__ float dnormal_dx(float x, float loc, float sc) { return -(x - loc) / (sc * sc); }
__device__ __forceinline__ float dnormal_dsc(float x, float loc, float sc) {
  const float dd = x - loc;
  return dd * dd / (sc * sc * sc) - 1.0f / sc;
}
__device__ __forceinline__ float delu_from_out(float o) { return o > 0.0f ? 1.0f : o + 1.0f; }

// ------------------------------------------------------------------------------------------------
// log-probability adjoint, all T frames in one launch (grid R x T), mirror of k_logprob.
// ------------------------------------------------------------------------------------------------

constexpr int SQ_SMALL_MAX = 1024;
struct SmallParamTab { int n; int src[16]; int len[16]; int dst[16]; int total; };

__global__ __launch_bounds__(256) void k_logprob_bwd(const LogprobBwdArgs a, const POff pc, const SmallParamTab tab, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  // One workgroup of four wavefronts per (row b', frame).  Four things keep its serial chain short:
  // * the ~800 small parameters this kernel reads (where-prior RNN, step-prior MLP, Cholesky factor) are staged in LDS and
  //   their gradients accumulated there (`pc` holds COMPACT offsets into these arrays): reads from the flat buffer
  //   interleaved with atomics on the gradient buffer cannot be hoisted by the compiler and serialised the kernel on L2
  //   latency (1.1 ms in round 1); one coalesced flush of float atomics per workgroup at the end instead;
  // * the row's own inputs -- its N discovery / propagation / merged records and prior statistics -- are staged in LDS too, all
  //   requested in ONE batch, and the record gradients are summed in LDS and flushed at the end (until round 3: read from
  //   global memory where needed, between float atomics on the gradient records the reads have to be ordered against);
  // * everything that is per slot runs on a lane (or a group of four lanes) PER SLOT instead of in a loop over the slots on
  //   lane 0: the where-prior RNN steps (independent given the conditioning), the Cholesky solves, the number-of-steps terms
  //   and the step-prior MLP were ~7000 dependent instructions of one wave, 84 us for a kernel that moves 5 MB;
  // * the independent parts run on different wavefronts: wave 0 the where-prior RNN and the discovery where terms, wave 1 the
  //   step prior and the number-of-steps terms, wave 2 the propagation where terms and Cholesky solves, and all four share
  //   the what terms slot by slot.  Every LDS accumulator has ONE writing wave, so the sums keep a fixed order (the
  //   gradient is bit-reproducible); the two pieces of d e_sum meet in LDS before the prior-logit terms.
  extern __shared__ __attribute__((aligned(16))) float row_s[];
  __shared__ float hid_s[16], gv_s[SQ_MAXN + 1], de_s[2];
  const int r = blockIdx.x, fr = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = d.N, nw = d.nw, RW = rec::W, MW = rec::ZW;
  const size_t fs = (size_t)fr * d.R * N + (size_t)r * N;
  const size_t frr = (size_t)fr * d.R + r;
  float* rd_s = row_s;              // [N][RW] discovery records
  float* rp_s = rd_s + N * RW;      // [N][RW] propagation records
  float* drd_s = rp_s + N * RW;     // the same two as gradients
  float* drp_s = drd_s + N * RW;
  float* rm_s = drp_s + N * RW;     // [N][MW] merged records of the previous frame (z part: all that is used of them), gradient
  float* drm_s = rm_s + N * MW;
  float* ps_s = drm_s + N * MW;     // [N][ps_ld] prior statistics, gradient
  float* dps_s = ps_s + N * a.ps_ld;
  float* fl = dps_s + N * a.ps_ld;  // [tab.total] small parameters, gradient
  float* gl = fl + tab.total;
  sq_wave_stage(rd_s, a.rec_d + fs * RW, N * RW, lane, wave, 4);
  sq_wave_stage(rp_s, a.rec_p + fs * RW, N * RW, lane, wave, 4);
  for (int k = wave; k < N; k += 4) sq_wave_stage(rm_s + k * MW, a.rec_m + (fs + k) * RW, MW, lane);
  sq_wave_stage(ps_s, a.pstats + fs * a.ps_ld, N * a.ps_ld, lane, wave, 4);
#pragma unroll   // (constant indices: the table entries are read from the kernel arguments in one batch, not one dependent scalar load per trip)
  for (int sgi = 0; sgi < 16; ++sgi)
    if (sgi < tab.n) sq_wave_stage(fl + tab.dst[sgi], a.flat + tab.src[sgi], tab.len[sgi], lane, wave, 4);
  for (int i = tid; i < 2 * N * RW; i += 256) drd_s[i] = 0.0f;          // drd_s | drp_s
  for (int i = tid; i < N * MW; i += 256) drm_s[i] = 0.0f;
  for (int i = tid; i < N * a.ps_ld; i += 256) dps_s[i] = 0.0f;
  for (int i = tid; i < tab.total; i += 256) gl[i] = 0.0f;
  const float gw = a.g_lw[frr], gd = a.g_dl[frr];
  __syncthreads();
  const float* flat = fl;
  float* fg = gl;
  const int t_global = a.t_global0 + fr;
  // ---------------- forward quantities needed below: lane k holds the prior logit of slot k; e_sum
  float pl_k = 0.0f, e_sum;
  {
    float e = 0.0f;
    if (lane < N) {
      const float* rm = rm_s + lane * MW;
      const float pres_tm1 = rm[rec::PRES];
      float v = ps_s[lane * a.ps_ld] + a.cfg.prop_prior_step_bias;
      v = pres_tm1 * v + (pres_tm1 - 1.0f) * 88.0f;
      if (a.cfg.prop_prior_type != 0) v = rm[rec::LOGIT] + 0.1f * v;
      pl_k = v;
      e = (sq_sigmoid(v) - 0.5f) / (float)N;
    }
    e_sum = sq_wave_sum(e);
  }
  const float n_disc = sq_wave_sum(lane < N ? rd_s[lane * RW + rec::PRES] : 0.0f);
  const int n = (int)(n_disc + 0.5f);
  // ---------------- wave 0: discovery, recurrent where prior
  if (wave == 0) {
  float d_e = 0.0f;
  if (a.cfg.rec_where_prior) {
    // s = elu(spre + e ce), hs = s h2h + b_h2h + b_i2h  (lanes own s_i for i = lane, lane + 64)
    float sv[2], part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = lane + 64 * q;
      sv[q] = sq_elu(a.spre[frr * 128 + i] + e_sum * flat[pc.rn_cond_w + (4 + d.nh) * 128 + i]);
      for (int jj = 0; jj < 4; ++jj) part[jj] += sv[q] * flat[pc.rn_h2h_w + i * 4 + jj];
    }
    float hs[4], d_hs[4];
    for (int jj = 0; jj < 4; ++jj) hs[jj] = sq_wave_sum(part[jj]) + flat[pc.rn_h2h_b + jj] + flat[pc.rn_i2h_b + jj];
    // the N steps of the RNN only share hs (step j reads the SAMPLED where of step j - 1): lanes 4 j + i work on step j,
    // component i; sums over the components are butterflies inside the group of four
    {
      const int j = min(lane >> 2, N - 1), ci = lane & 3;
      const bool on = lane < 4 * N;
      const float* rd = rd_s + j * RW;
      const float* xp = j == 0 ? flat + pc.rn_init_sample : rd_s + (j - 1) * RW + rec::WHERE;
      float o[4];
      for (int mm = 0; mm < 4; ++mm) {
        float pre = hs[mm];
        for (int i = 0; i < 4; ++i) pre += xp[i] * flat[pc.rn_i2h_w + i * 4 + mm];
        o[mm] = tanhf(pre);
      }
      float loc = flat[pc.rn_readout_b + ci], raw = flat[pc.rn_readout_b + 4 + ci];
      for (int mm = 0; mm < 4; ++mm) {
        loc += o[mm] * flat[pc.rn_readout_w + mm * 8 + ci];
        raw += o[mm] * flat[pc.rn_readout_w + mm * 8 + 4 + ci];
      }
      const float psc = sq_softplus(raw) + 1e-2f;
      const float x = rd[rec::WHERE + ci];
      const float coef = on ? gw * rd[rec::PRES] : 0.0f;
      const float g_loc = coef * (-dnormal_dx(x, loc, psc));
      const float g_raw = coef * dnormal_dsc(x, loc, psc) * sq_sigmoid(raw);
      if (on) {
        atomicAdd(&drd_s[j * RW + rec::WHERE + ci], coef * dnormal_dx(x, loc, psc));
        atomicAdd(&fg[pc.rn_readout_b + ci], g_loc);
        atomicAdd(&fg[pc.rn_readout_b + 4 + ci], g_raw);
        for (int mm = 0; mm < 4; ++mm) {
          atomicAdd(&fg[pc.rn_readout_w + mm * 8 + ci], o[mm] * g_loc);
          atomicAdd(&fg[pc.rn_readout_w + mm * 8 + 4 + ci], o[mm] * g_raw);
        }
      }
      // g_o[m] = sum_i ro_w[m][i] g_loc_i + ro_w[m][4+i] g_raw_i  (over the four lanes of the step)
      float g_pre[4];
      for (int mm = 0; mm < 4; ++mm) {
        float v = flat[pc.rn_readout_w + mm * 8 + ci] * g_loc + flat[pc.rn_readout_w + mm * 8 + 4 + ci] * g_raw;
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        g_pre[mm] = v * (1.0f - o[mm] * o[mm]);
        d_hs[mm] = sq_wave_sum(on && ci == 0 ? g_pre[mm] : 0.0f);
      }
      if (on) {  // ci = input index i of i2h
        float dx = 0.0f;
        for (int mm = 0; mm < 4; ++mm) {
          atomicAdd(&fg[pc.rn_i2h_w + ci * 4 + mm], xp[ci] * g_pre[mm]);
          dx += flat[pc.rn_i2h_w + ci * 4 + mm] * g_pre[mm];
        }
        if (j == 0) atomicAdd(&fg[pc.rn_init_sample + ci], dx);
        else atomicAdd(&drd_s[(j - 1) * RW + rec::WHERE + ci], dx);
      }
    }
    if (lane < 4) {
      atomicAdd(&fg[pc.rn_h2h_b + lane], d_hs[lane]);
      atomicAdd(&fg[pc.rn_i2h_b + lane], d_hs[lane]);
    }
    float de_part = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = lane + 64 * q;
      float g_s = 0.0f;
      for (int jj = 0; jj < 4; ++jj) {
        g_s += flat[pc.rn_h2h_w + i * 4 + jj] * d_hs[jj];
        atomicAdd(&fg[pc.rn_h2h_w + i * 4 + jj], sv[q] * d_hs[jj]);
      }
      const float g_spre = g_s * delu_from_out(sv[q]);
      a.d_spre[frr * 128 + i] = g_spre;
      atomicAdd(&fg[pc.rn_cond_w + (4 + d.nh) * 128 + i], e_sum * g_spre);
      de_part += flat[pc.rn_cond_w + (4 + d.nh) * 128 + i] * g_spre;
    }
    d_e += sq_wave_sum(de_part);
  } else {
    a.d_spre[frr * 128 + lane] = 0.0f;
    a.d_spre[frr * 128 + 64 + lane] = 0.0f;
    if (lane < 4 * N) {
      const int j = lane >> 2, ci = lane & 3;
      const float* rd = rd_s + j * RW;
      atomicAdd(&drd_s[j * RW + rec::WHERE + ci], gw * rd[rec::PRES] * dnormal_dx(rd[rec::WHERE + ci], a.cfg.where_prior_mean[ci], 1.0f));
    }
  }
  if (lane == 0) de_s[0] = d_e;
  // q where of the discovery slots (same gradient elements as the prior above: same wave)
  if (lane < 4 * N) {
    const int j = lane >> 2, ci = lane & 3;
    const float* rd = rd_s + j * RW;
    float* dr = drd_s + j * RW;
    const float cq = -gw * rd[rec::PRES];
    const float x = rd[rec::WHERE + ci], loc = rd[rec::WHERE_LOC + ci], sc = rd[rec::WHERE_SCALE + ci];
    atomicAdd(&dr[rec::WHERE + ci], cq * dnormal_dx(x, loc, sc));
    atomicAdd(&dr[rec::WHERE_LOC + ci], -cq * dnormal_dx(x, loc, sc));
    atomicAdd(&dr[rec::WHERE_SCALE + ci], cq * dnormal_dsc(x, loc, sc));
  }
  }  // wave 0
  // ---------------- all waves: q what, p what of the discovery slots
  for (int j = wave; j < N; j += 4) {
    const float* rd = rd_s + j * RW;
    float* dr = drd_s + j * RW;
    const float pres = rd[rec::PRES];
    const float cq = -gw * pres, cp = gw * pres;
    SQ_WHAT_LANES(wc, lane, nw) {
      const float x = rd[rec::WHAT + wc], loc = rd[rec::WHAT_LOC + wc], sc = rd[rec::WHAT_SCALE + wc];
      atomicAdd(&dr[rec::WHAT + wc], cq * dnormal_dx(x, loc, sc) + cp * (-x));
      atomicAdd(&dr[rec::WHAT_LOC + wc], -cq * dnormal_dx(x, loc, sc));
      atomicAdd(&dr[rec::WHAT_SCALE + wc], cq * dnormal_dsc(x, loc, sc));
    }
  }
  // ---------------- wave 1: number of steps
  if (wave == 1) {
  float d_e = 0.0f;
  if (lane < N) {
    // q_num = log J_n, J from p_j = sigmoid(logit_j): d log J_n / d logit_j = (1 - p_j) for j < n, -p_n for j = n < N
    const int j = lane;
    const float coef = -gw + gd;
    const float pj = rd_s[j * RW + rec::PROB];
    float g = 0.0f;
    if (j < n) g = coef * (1.0f - pj);
    else if (j == n) g = coef * (-pj);
    atomicAdd(&drd_s[j * RW + rec::LOGIT], g);
  }
  // categorical / geometric prior of the number of steps: hidden unit i on lane i, class c on lane c
  if (a.cfg.disc_prior_type == 0) {
    const int N1 = N + 1;
    const float hid = lane < 10 ? sq_elu(e_sum * flat[pc.sp_l0_w + lane] + flat[pc.sp_l0_b + lane]) : 0.0f;
    if (lane < 10) hid_s[lane] = hid;
    __builtin_amdgcn_wave_barrier();   // (one wave: its LDS operations complete in order)
    float lg = -1e30f;
    if (lane < N1) {
      float v = flat[pc.step_prior_bias + lane] + (t_global > 0 ? flat[pc.step_prior_tbias + lane] : 0.0f) + flat[pc.sp_l1_b + lane];
      for (int i = 0; i < 10; ++i) v += hid_s[i] * flat[pc.sp_l1_w + i * N1 + lane];
      lg = sq_elu(v);
    }
    float mx = lg;
    constexpr int CLS = SQ_MAXN + 1 <= 16 ? 16 : 32;   // classes sit on lanes 0 .. N
#pragma unroll
    for (int o = 1; o < CLS; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float ex = lane < N1 ? expf(lg - mx) : 0.0f;
    const float se = sq_wave_sum(ex);
    if (lane < N1) {
      const float gv = gw * ((lane == n ? 1.0f : 0.0f) - ex / se) * delu_from_out(lg);
      gv_s[lane] = gv;
      atomicAdd(&fg[pc.step_prior_bias + lane], gv);
      if (t_global > 0) atomicAdd(&fg[pc.step_prior_tbias + lane], gv);
      atomicAdd(&fg[pc.sp_l1_b + lane], gv);
    }
    __builtin_amdgcn_wave_barrier();
    for (int q = lane; q < 10 * N1; q += 64) atomicAdd(&fg[pc.sp_l1_w + q], hid_s[q / N1] * gv_s[q % N1]);
    float de = 0.0f;
    if (lane < 10) {
      float gh = 0.0f;
      for (int c = 0; c < N1; ++c) gh += flat[pc.sp_l1_w + lane * N1 + c] * gv_s[c];
      const float ghp = gh * delu_from_out(hid);
      atomicAdd(&fg[pc.sp_l0_w + lane], e_sum * ghp);
      atomicAdd(&fg[pc.sp_l0_b + lane], ghp);
      de = flat[pc.sp_l0_w + lane] * ghp;
    }
    d_e += sq_wave_sum(de);
  }
  if (lane == 0) de_s[1] = d_e;
  }  // wave 1
  // ---------------- all waves: q what, p what of the propagated slots
  for (int k = wave; k < N; k += 4) {
    const float* rp = rp_s + k * RW;
    const float* rm = rm_s + k * MW;
    const float* ps = ps_s + k * a.ps_ld;
    float* drp = drp_s + k * RW;
    float* drm = drm_s + k * MW;
    float* dps = dps_s + k * a.ps_ld;
    const float pres = rp[rec::PRES], pres_tm1 = rm[rec::PRES];
    const float m = pres_tm1 * pres;
    const float cq = -gw * m, cp = gw * m;
    SQ_WHAT_LANES(wc, lane, nw) {
      const float x = rp[rec::WHAT + wc];
      const float loc = rp[rec::WHAT_LOC + wc], sc = rp[rec::WHAT_SCALE + wc];
      float ploc = ps[5 + wc];
      if (a.cfg.prop_prior_type == 1) ploc = rm[rec::WHAT + wc];
      else if (a.cfg.prop_prior_type == 2) ploc = rm[rec::WHAT + wc] + 0.1f * ploc;
      const float praw = ps[9 + nw + wc];
      const float psc = sq_softplus(praw) + 1e-2f;
      atomicAdd(&drp[rec::WHAT + wc], cq * dnormal_dx(x, loc, sc) + cp * dnormal_dx(x, ploc, psc));
      atomicAdd(&drp[rec::WHAT_LOC + wc], -cq * dnormal_dx(x, loc, sc));
      atomicAdd(&drp[rec::WHAT_SCALE + wc], cq * dnormal_dsc(x, loc, sc));
      const float g_ploc = -cp * dnormal_dx(x, ploc, psc);
      if (a.cfg.prop_prior_type == 0) dps[5 + wc] = g_ploc;
      else {
        atomicAdd(&drm[rec::WHAT + wc], g_ploc);
        if (a.cfg.prop_prior_type == 2) dps[5 + wc] = 0.1f * g_ploc;
      }
      dps[9 + nw + wc] = cp * dnormal_dsc(x, ploc, psc) * sq_sigmoid(praw);
    }
  }
  // ---------------- wave 2: p where and the MultivariateNormalTriL posterior of where
  if (wave == 2 && lane < 4 * N) {
    const int k = lane >> 2, ci = lane & 3;
    const float* rp = rp_s + k * RW;
    const float* rm = rm_s + k * MW;
    const float* ps = ps_s + k * a.ps_ld;
    float* dps = dps_s + k * a.ps_ld;
    const float cp = gw * rm[rec::PRES] * rp[rec::PRES];
    const float x = rp[rec::WHERE + ci];
    float ploc = ps[1 + ci];
    if (a.cfg.prop_prior_type == 1) ploc = rm[rec::WHERE + ci];
    else if (a.cfg.prop_prior_type == 2) ploc = rm[rec::WHERE + ci] + 0.1f * ploc;
    const float praw = ps[5 + nw + ci];
    const float psc = sq_softplus(praw) + 1e-2f;
    atomicAdd(&drp_s[k * RW + rec::WHERE + ci], cp * dnormal_dx(x, ploc, psc));
    const float g_ploc = -cp * dnormal_dx(x, ploc, psc);
    if (a.cfg.prop_prior_type == 0) dps[1 + ci] = g_ploc;
    else {
      atomicAdd(&drm_s[k * MW + rec::WHERE + ci], g_ploc);
      if (a.cfg.prop_prior_type == 2) dps[1 + ci] = 0.1f * g_ploc;
    }
    dps[5 + nw + ci] = cp * dnormal_dsc(x, ploc, psc) * sq_sigmoid(praw);
  }
  if (wave == 2 && lane < N) {
    const int k = lane;
    const float* rp = rp_s + k * RW;
    const float* rm = rm_s + k * MW;
    float* drp = drp_s + k * RW;
    const float pres = rp[rec::PRES], pres_tm1 = rm[rec::PRES];
    const float cq = -gw * pres_tm1 * pres;
    // L = T * sc[:,None] + diag(sc)
    const float* ch = flat + pc.cholesky;
    float L[4][4], y[4], u[4], dd[4];
    for (int i = 0; i < 4; ++i) {
      const float sci = rp[rec::WHERE_SCALE + i];
      for (int j = 0; j < 4; ++j) L[i][j] = j <= i ? tril4(ch, i, j) * sci + (i == j ? sci : 0.0f) : 0.0f;
      dd[i] = rp[rec::WHERE + i] - rp[rec::WHERE_LOC + i];
    }
    for (int i = 0; i < 4; ++i) {
      float acc = dd[i];
      for (int j = 0; j < i; ++j) acc -= L[i][j] * y[j];
      y[i] = acc / L[i][i];
    }
    for (int i = 3; i >= 0; --i) {  // L^T u = y
      float acc = y[i];
      for (int j = i + 1; j < 4; ++j) acc -= L[j][i] * u[j];
      u[i] = acc / L[i][i];
    }
    for (int i = 0; i < 4; ++i) {
      atomicAdd(&drp[rec::WHERE + i], cq * (-u[i]));
      atomicAdd(&drp[rec::WHERE_LOC + i], cq * u[i]);
      const float sci = rp[rec::WHERE_SCALE + i];
      float dsc = 0.0f;
      for (int j = 0; j <= i; ++j) {
        const float dL = u[i] * y[j] - (i == j ? 1.0f / L[i][i] : 0.0f);
        const float tij = tril4(ch, i, j);
        dsc += dL * (tij + (i == j ? 1.0f : 0.0f));
        const int q = i * 4 + j;  // fill_triangular index -> cholesky_scale element
        atomicAdd(&fg[pc.cholesky + (q < 6 ? 4 + q : 15 - q)], cq * dL * sci);
      }
      atomicAdd(&drp[rec::WHERE_SCALE + i], cq * dsc);
    }
  }
  __syncthreads();
  if (wave == 2 && lane < N) {
    // presence Bernoullis and the prior logit (incl. its path through e_sum: both pieces of d e_sum are in LDS now)
    const int k = lane;
    const float* rp = rp_s + k * RW;
    const float* rm = rm_s + k * MW;
    float* drp = drp_s + k * RW;
    const float pres = rp[rec::PRES], pres_tm1 = rm[rec::PRES];
    const float d_e = (a.cfg.rec_where_prior ? de_s[0] : 0.0f) + (a.cfg.disc_prior_type == 0 ? de_s[1] : 0.0f);
    const float logit = rp[rec::LOGIT];
    atomicAdd(&drp[rec::LOGIT], (-gw + gd) * pres_tm1 * (pres - sq_sigmoid(logit)));
    const float spl = sq_sigmoid(pl_k);
    float g_pl = gw * pres_tm1 * (pres - spl) + d_e * spl * (1.0f - spl) / (float)N;
    if (a.cfg.prop_prior_type != 0) {
      atomicAdd(&drm_s[k * MW + rec::LOGIT], g_pl);
      g_pl *= 0.1f;
    }
    dps_s[k * a.ps_ld] = g_pl * pres_tm1;
  }
  __syncthreads();
  for (int i = tid; i < N * RW; i += 256) {   // (the gradient records are accumulated into: the dX launches add to them too)
    const float gd_ = drd_s[i], gp_ = drp_s[i];
    if (gd_ != 0.0f) unsafeAtomicAdd(a.d_rec_d + fs * RW + i, gd_);
    if (gp_ != 0.0f) unsafeAtomicAdd(a.d_rec_p + fs * RW + i, gp_);
  }
  for (int i = tid; i < N * MW; i += 256) {
    const int k = i / MW, q = i - k * MW;
    const float gm_ = drm_s[i];
    if (gm_ != 0.0f) unsafeAtomicAdd(a.d_rec_m + (fs + k) * RW + q, gm_);
  }
  for (int i = tid; i < N * a.ps_ld; i += 256) a.d_pstats[fs * a.ps_ld + i] = dps_s[i];
#pragma unroll
  for (int sgi = 0; sgi < 16; ++sgi)
    if (sgi < tab.n)
      for (int i = tid; i < tab.len[sgi]; i += 256) {
        const float g = gl[tab.dst[sgi] + i];
        if (g != 0.0f) unsafeAtomicAdd(a.flat_grad + tab.src[sgi] + i, g);
      }
}

int sq_launch_logprob_bwd(const LogprobBwdArgs& a, POff po, Dims d, int T, hipStream_t s) {
  // compact layout of the small parameters (flat offset, length) -> LDS offset
  SmallParamTab tab = {};
  POff cp = po;
  int o = 0;
  auto seg = [&](int src, int len) {
    tab.src[tab.n] = src; tab.len[tab.n] = len; tab.dst[tab.n] = o;
    ++tab.n;
    const int at = o;
    o += len;
    return at;
  };
  const int N1 = d.N + 1;
  cp.rn_readout_w = seg(po.rn_readout_w, 32);
  cp.rn_readout_b = seg(po.rn_readout_b, 8);
  cp.rn_i2h_w = seg(po.rn_i2h_w, 16);
  cp.rn_i2h_b = seg(po.rn_i2h_b, 4);
  cp.rn_h2h_w = seg(po.rn_h2h_w, 512);
  cp.rn_h2h_b = seg(po.rn_h2h_b, 4);
  cp.rn_cond_w = seg(po.rn_cond_w + (4 + d.nh) * 128, 128) - (4 + d.nh) * 128;  // only the e row is touched here
  cp.rn_init_sample = seg(po.rn_init_sample, 4);
  cp.sp_l0_w = seg(po.sp_l0_w, 10);
  cp.sp_l0_b = seg(po.sp_l0_b, 10);
  cp.sp_l1_w = seg(po.sp_l1_w, 10 * N1);
  cp.sp_l1_b = seg(po.sp_l1_b, N1);
  cp.step_prior_bias = seg(po.step_prior_bias, N1);
  cp.step_prior_tbias = seg(po.step_prior_tbias, N1);
  cp.cholesky = seg(po.cholesky, 10);
  if (o > SQ_SMALL_MAX) return -1;
  tab.total = o;
  const size_t shm = (4 * (size_t)d.N * rec::W + 2 * (size_t)d.N * rec::ZW + 2 * (size_t)d.N * a.ps_ld + 2 * (size_t)o) * sizeof(float);
  if (shm > 150 * 1024) return -1;   // (sqair_create has checked this configuration against sq_logprob_bwd_lds_bytes)
  if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_logprob_bwd, 150 * 1024) != 0) return -2;
  SQ_LAUNCH(k_logprob_bwd, dim3(d.R, T), dim3(256), shm, s, a, cp, tab, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// LSTM cell adjoint (mirror of k_lstm_cell): from d h' and d c' to the gate pre-activation gradients (i, j, f, o)
// and d c_prev; c' is recomputed from the kept gate pre-activations.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lstm_cell_bwd(const float* __restrict__ gates, int g_ld, const float* __restrict__ c_prev,
                                                       int c_ld, const float* __restrict__ d_h, int dh_ld,
                                                       const float* __restrict__ d_c, int dc_ld, float* __restrict__ d_gates,
                                                       int dg_ld, float* __restrict__ d_cprev, int dcp_ld, int rows, int nh,
                                                       float* __restrict__ d_gates2, int dg2_ld SQ_TLP) {
  SQ_TL_SCOPE;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * nh) return;
  const int r = e / nh, q = e - r * nh;
  const float* g = gates + (size_t)r * g_ld;
  const float si = sq_sigmoid(g[q]), tj = tanhf(g[nh + q]), sf = sq_sigmoid(g[2 * nh + q] + 1.0f), so = sq_sigmoid(g[3 * nh + q]);
  const float cp = c_prev[(size_t)r * c_ld + q];
  const float tc = tanhf(sf * cp + si * tj);
  const float dh = d_h[(size_t)r * dh_ld + q];
  const float dc = (d_c != nullptr ? d_c[(size_t)r * dc_ld + q] : 0.0f) + dh * so * (1.0f - tc * tc);
  const float gi = dc * tj * si * (1.0f - si), gj = dc * si * (1.0f - tj * tj), gf = dc * cp * sf * (1.0f - sf);
  const float go = dh * tc * so * (1.0f - so);
  float* dg = d_gates + (size_t)r * dg_ld;
  dg[q] = gi; dg[nh + q] = gj; dg[2 * nh + q] = gf; dg[3 * nh + q] = go;
  if (d_gates2 != nullptr) {  // second copy (the slot's block of the loop-invariant pre-activation gradient)
    float* dg2 = d_gates2 + (size_t)r * dg2_ld;
    dg2[q] = gi; dg2[nh + q] = gj; dg2[2 * nh + q] = gf; dg2[3 * nh + q] = go;
  }
  d_cprev[(size_t)r * dcp_ld + q] = dc * sf;
}
int sq_launch_lstm_cell_bwd(const float* gates, int g_ld, const float* c_prev, int c_ld, const float* d_h, int dh_ld, const float* d_c,
                            int dc_ld, float* d_gates, int dg_ld, float* d_cprev, int dcp_ld, int rows, int nh, hipStream_t s,
                            float* d_gates2, int dg2_ld) {
  SQ_LAUNCH(k_lstm_cell_bwd, dim3((rows * nh + 255) / 256), dim3(256), 0, s, gates, g_ld, c_prev, c_ld, d_h, dh_ld, d_c, dc_ld,
                     d_gates, dg_ld, d_cprev, dcp_ld, rows, nh, d_gates2, dg2_ld);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// compaction adjoint: route the gradients of the merged slots of frame t+1 back to their source slots
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact_bwd(const CompactBwdArgs a, const POff po, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  const int r = blockIdx.x, tid = threadIdx.x, N = d.N, RW = rec::W;
  __shared__ int inv_s[2 * SQ_MAXN];  // source slot -> destination (or -1)
  if (tid < 2 * N) inv_s[tid] = -1;
  __syncthreads();
  if (tid < N) inv_s[a.src[(size_t)r * N + tid]] = tid;
  __syncthreads();
  typedef float cf4 __attribute__((ext_vector_type(4)));   // all gradient rows are 16-byte aligned: 16-byte units throughout
  const cf4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  // Three jobs -- (1) gradient records: add the merged slot's record onto the source slot's; (2) recurrent states of the
  // propagated slots: the gradient of the merged slot they became (or zero); (3) newly discovered objects started from the
  // trainable initial states: their gradients summed per row here, over rows / frames after the sweep (atomics on the 2 x nh
  // parameter words from every row serialise in L2) -- and every one of their loads is requested before the first store, so
  // the kernel is two memory round trips (the permutation, then everything) and the write.  (Measured: 5.4 us either way --
  // the per-unit loops it replaces overlapped well enough; what is left is those two cold round trips themselves.)
#ifdef SQAIR_WIDE
  // (the wide build: the same three jobs as plain loops -- sources and destinations are different buffers)
  const int r4 = RW / 4;
  for (int e = tid; e < 2 * N * r4; e += 256) {
    const int sl = e / r4, i = e - sl * r4;
    const int dst = inv_s[sl];
    if (dst >= 0) {
      cf4* tg = reinterpret_cast<cf4*>(sl < N ? a.d_rec_p + ((size_t)r * N + sl) * RW : a.d_rec_d + ((size_t)r * N + (sl - N)) * RW) + i;
      *tg = *tg + reinterpret_cast<const cf4*>(a.d_rec_next + ((size_t)r * N + dst) * RW)[i];
    }
  }
  for (int st = 0; st < 2; ++st) {   // temporal state, prior state
    const int w = st == 0 ? d.snh : d.psnh, w4 = w / 4;
    const float* next = st == 0 ? a.d_temporal_next : a.d_prior_next;
    float* outp = st == 0 ? a.d_temporal_p : a.d_prior_p;
    for (int e = tid; e < N * w4; e += 256) {
      const int sl = e / w4, i = e - sl * w4;
      const int dst = inv_s[sl];
      reinterpret_cast<cf4*>(outp + ((size_t)r * N + sl) * w)[i] = dst >= 0 ? reinterpret_cast<const cf4*>(next + ((size_t)r * N + dst) * w)[i] : zero4;
    }
    for (int i = tid; i < w4; i += 256) {
      cf4 acc = zero4;
      for (int j = 0; j < N; ++j) {
        const int dst = inv_s[N + j];
        if (dst >= 0) acc += reinterpret_cast<const cf4*>(next + ((size_t)r * N + dst) * w)[i];
      }
      reinterpret_cast<cf4*>((st == 0 ? a.d_new_temporal : a.d_new_prior) + (size_t)r * w)[i] = acc;
    }
  }
}
#else
  constexpr int Q1 = 3, Q2 = 4;   // units per thread: 2 N RW / 4 <= 3 x 256 records, N snh / 4 <= 4 x 256 state words (N <= 8, snh <= 512)
  const int r4 = RW / 4;
  cf4 g1[Q1], t1[Q1];
  cf4* tg1[Q1];
#pragma unroll
  for (int q = 0; q < Q1; ++q) {
    const int e = tid + 256 * q;
    tg1[q] = nullptr;
    g1[q] = zero4;
    t1[q] = zero4;
    if (e < 2 * N * r4) {
      const int sl = e / r4, i = e - sl * r4;
      const int dst = inv_s[sl];
      if (dst >= 0) {
        tg1[q] = reinterpret_cast<cf4*>(sl < N ? a.d_rec_p + ((size_t)r * N + sl) * RW : a.d_rec_d + ((size_t)r * N + (sl - N)) * RW) + i;
        g1[q] = reinterpret_cast<const cf4*>(a.d_rec_next + ((size_t)r * N + dst) * RW)[i];
        t1[q] = *tg1[q];
      }
    }
  }
  const int snh = d.snh, psnh = d.psnh;
  cf4 v2[2][Q2], acc3[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {   // temporal state, prior state
    const int w = st == 0 ? snh : psnh, w4 = w / 4;
    const float* next = st == 0 ? a.d_temporal_next : a.d_prior_next;
#pragma unroll
    for (int q = 0; q < Q2; ++q) {
      const int e = tid + 256 * q;
      v2[st][q] = zero4;
      if (e < N * w4) {
        const int sl = e / w4, i = e - sl * w4;
        const int dst = inv_s[sl];
        if (dst >= 0) v2[st][q] = reinterpret_cast<const cf4*>(next + ((size_t)r * N + dst) * w)[i];
      }
    }
    cf4 x3[SQ_MAXN];
#pragma unroll
    for (int j = 0; j < SQ_MAXN; ++j) {
      x3[j] = zero4;
      if (j < N && tid < w4) {
        const int dst = inv_s[N + j];
        if (dst >= 0) x3[j] = reinterpret_cast<const cf4*>(next + ((size_t)r * N + dst) * w)[tid];
      }
    }
    acc3[st] = zero4;
#pragma unroll
    for (int j = 0; j < SQ_MAXN; ++j) acc3[st] += x3[j];
  }
  // ---- stores
#pragma unroll
  for (int q = 0; q < Q1; ++q)
    if (tg1[q] != nullptr) *tg1[q] = t1[q] + g1[q];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int w = st == 0 ? snh : psnh, w4 = w / 4;
    float* outp = st == 0 ? a.d_temporal_p : a.d_prior_p;
#pragma unroll
    for (int q = 0; q < Q2; ++q) {
      const int e = tid + 256 * q;
      if (e < N * w4) {
        const int sl = e / w4, i = e - sl * w4;
        reinterpret_cast<cf4*>(outp + ((size_t)r * N + sl) * w)[i] = v2[st][q];
      }
    }
    if (tid < w4) reinterpret_cast<cf4*>((st == 0 ? a.d_new_temporal : a.d_new_prior) + (size_t)r * w)[tid] = acc3[st];
  }
}
#endif
int sq_launch_compact_bwd(const CompactBwdArgs& a, POff po, Dims d, hipStream_t s) {
  SQ_LAUNCH(k_compact_bwd, dim3(d.R), dim3(256), 0, s, a, po, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// slot-tail adjoint (mirror of k_slot_tail): presence logit -> steps-predictor output / hidden layer -> what,
// then the what-sample adjoint (gated mixture for propagation).  One workgroup (128 threads) per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_slot_tail_bwd(const TailBwdArgs a, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  SQ_PIN8(a.is_disc, a.slot, a.rec_prev, a.rec_new, a.d_rec_new, a.d_rec_prev, a.s1h, a.s1h_ld);
  SQ_PIN8(a.hraw, a.h_ld, a.enc, a.enc_ld, a.noise, a.flat, a.w2_off, a.wwhat_off);
  __shared__ float ds_s[SQ_MAX_NHIDDEN / 2];
  extern __shared__ float wt_s[];  // the `what` rows of steps.l0.w, [nw][nsp + 1] (padded: conflict-free row reads)
  const int r = blockIdx.x, tid = threadIdx.x, nw = d.nw, nsp = d.nh / 2, RW = rec::W;
  // the weight block: 16-byte loads, all of a thread's requests (7 for the 50 x 128 block of the shipped sizes) issued here and
  // written to LDS only AFTER the per-row operands below have been requested too -- one memory round trip for the whole kernel.
  // (Rounds 1-2: the plain copy loop, seven dependent round trips; round 3 at first: 2 x 25 dword loads per thread, whose LDS
  // stores waited for the weights before the per-row operands were even requested: 5.8 us.  LDS-DMA dwords were slower still.)
  typedef float sq_f32x4 __attribute__((ext_vector_type(4), aligned(4)));   // (flat parameter offsets are only 4-byte aligned)
  constexpr int WU = 8;
  const int total4 = (nw * nsp) >> 2;   // nsp is a multiple of 64
  const sq_f32x4* __restrict__ wsrc4 = reinterpret_cast<const sq_f32x4*>(a.flat + a.wwhat_off);
  sq_f32x4 wv[WU];
#pragma unroll
  for (int q = 0; q < WU; ++q) wv[q] = wsrc4[min(tid + 256 * q, total4 - 1)];
  const float* rn = a.rec_new + ((size_t)r * d.N + a.slot) * RW;
  float* drn = a.d_rec_new + ((size_t)r * d.N + a.slot) * RW;
  (void)rn;
  // (one unconditional load from a selected, always valid address: as a guarded load hipcc put a full wait behind it -- the
  // weight block's round trip -- ahead of the per-row requests below)
  const float* prevp = a.is_disc ? (a.slot == 0 ? a.flat : a.rec_new + ((size_t)r * d.N + a.slot - 1) * RW + rec::PRES)
                                 : a.rec_prev + ((size_t)r * d.N + a.slot) * RW + rec::PRES;
  float prev = *prevp;
  if (a.is_disc && a.slot == 0) prev = 1.0f;
  const float d_raw = prev * drn[rec::LOGIT];
  // two threads per what element (c = tid >> 1, each sums half of the hidden units); everything the second phase reads from
  // memory is requested HERE, ahead of the barrier, so that the kernel makes one memory round trip instead of two
  const int c = min(tid >> 1, nw - 1), half = tid & 1;
  const bool lead = (tid >> 1) < nw && half == 0;
  const float q_dw = drn[rec::WHAT + c], q_dloc = drn[rec::WHAT_LOC + c], q_dsc = drn[rec::WHAT_SCALE + c];
  const float eps = a.noise[(((size_t)r * 2 + (a.is_disc ? 1 : 0)) * d.N + a.slot) * d.nzw + 4 + c];
  const float loc2 = a.enc[(size_t)r * a.enc_ld + c], sc2 = a.enc[(size_t)r * a.enc_ld + nw + c];
  float hv5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, wtm1 = 0.0f, q_dprev = 0.0f;
  if (!a.is_disc) {
    const float* hr = a.hraw + (size_t)r * a.h_ld;
#pragma unroll
    for (int g = 0; g < 5; ++g) hv5[g] = hr[g * nw + c];
    wtm1 = a.rec_prev[((size_t)r * d.N + a.slot) * RW + rec::WHAT + c];
    q_dprev = a.d_rec_prev[((size_t)r * d.N + a.slot) * RW + rec::WHAT + c];
  }
  float hv_ = 0.0f, w2_ = 0.0f;
  if (tid < nsp) {
    hv_ = a.s1h[(size_t)r * a.s1h_ld + tid];
    w2_ = a.flat[a.w2_off + tid];
  }
  {
    const int sh = 31 - __clz(nsp);   // nsp = nh / 2 is 64 or 128 (sqair_create)
#pragma unroll
    for (int q = 0; q < WU; ++q) {
      const int e = 4 * (tid + 256 * q);
      if (e < 4 * total4) {
        float* dst = wt_s + (e >> sh) * (nsp + 1) + (e & (nsp - 1));   // (padded rows: conflict-free row reads, scalar stores)
        dst[0] = wv[q].x; dst[1] = wv[q].y; dst[2] = wv[q].z; dst[3] = wv[q].w;
      }
    }
    for (int e4 = tid + 256 * WU; e4 < total4; e4 += 256) {   // (wider layers than the library is built for)
      const sq_f32x4 x = wsrc4[e4];
      float* dst = wt_s + ((4 * e4) >> sh) * (nsp + 1) + ((4 * e4) & (nsp - 1));
      dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
    }
  }
  if (tid < nsp) {
    const float hv = hv_, w2 = w2_;
    const float g = d_raw * w2 * delu_from_out(hv);
    ds_s[tid] = g;
    a.d_s1pre[(size_t)r * a.ds_ld + tid] = g;
    if (a.d_s1pre2 != nullptr) a.d_s1pre2[(size_t)r * a.ds2_ld + tid] = g;
  }
  if (tid == 0) a.d_raw_out[(size_t)r * a.dr_ld] = d_raw;
  __syncthreads();
  float part = 0.0f;
  {
    const float* wrow = wt_s + c * (nsp + 1) + half * (nsp >> 1);
    const float* dsp = ds_s + half * (nsp >> 1);
    for (int i = 0; i < (nsp >> 1); ++i) part += dsp[i] * wrow[i];
  }
  part += __shfl_xor(part, 1, 64);
  if (lead) {
    const float dw = q_dw + part;
    drn[rec::WHAT + c] = dw;  // total gradient of the sample (kept for the batched weight gradients' bookkeeping)
    const float d_loc = dw + q_dloc;
    const float d_sc = dw * eps + q_dsc;
    if (a.is_disc) {
      a.d_enc[(size_t)r * a.de_ld + c] = d_loc;
      // scale = softplus(raw) + 0.01  ->  d raw = d scale * (1 - exp(-(scale - 0.01)))
      a.d_enc[(size_t)r * a.de_ld + nw + c] = a.enc_pre ? d_sc * (1.0f - sq_exp(-(sc2 - 1e-2f))) : d_sc;
    } else {
      const float t_loc = hv5[0], h1 = hv5[1];
      const float t_scale = sq_softplus(h1) + 1e-2f;
      const float s2 = sq_sigmoid(hv5[2]), s3 = sq_sigmoid(hv5[3]), s4 = sq_sigmoid(hv5[4]);
      const float fg = s2 * 0.9999f, ig = s3 * 0.9999f, tg = s4 * 0.9999f;
      const float d_fg = d_loc * wtm1;
      const float d_ig = -d_loc * loc2 - d_sc * sc2;
      const float d_tg = -d_loc * t_loc - d_sc * t_scale;
      a.d_rec_prev[((size_t)r * d.N + a.slot) * RW + rec::WHAT + c] = q_dprev + d_loc * fg;
      a.d_enc[(size_t)r * a.de_ld + c] = d_loc * (1.0f - ig);
      a.d_enc[(size_t)r * a.de_ld + nw + c] = d_sc * (1.0f - ig);
      float* dh = a.d_hraw + (size_t)r * a.dh_ld;
      dh[c] = d_loc * (1.0f - tg);
      dh[nw + c] = d_sc * (1.0f - tg) * sq_sigmoid(h1);
      dh[2 * nw + c] = d_fg * 0.9999f * s2 * (1.0f - s2);
      dh[3 * nw + c] = d_ig * 0.9999f * s3 * (1.0f - s3);
      dh[4 * nw + c] = d_tg * 0.9999f * s4 * (1.0f - s4);
    }
  }
}
int sq_launch_slot_tail_bwd(const TailBwdArgs& a, Dims d, hipStream_t s) {
  const size_t shm = (size_t)d.nw * (d.nh / 2 + 1) * sizeof(float);
  if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_slot_tail_bwd, 150 * 1024) != 0) return -2;
  SQ_LAUNCH(k_slot_tail_bwd, dim3(d.R), dim3(256), shm, s, a, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// crop adjoint inside the recurrence: d glimpse -> d where (all consumers' gradients are already in the gradient
// record) -> adjoint of the where sample -> d(transform output) [R][8], d(previous where), d(mask).
// modes as CropMode.  One workgroup per sequence.
// ------------------------------------------------------------------------------------------------
template <bool STAGED>   // the frame staged in LDS through registers (up to SQ_CROP_STAGE_MAX_PIXELS) or by LDS-DMA (larger frames)
__global__ __launch_bounds__(256) void k_crop_chain_bwd(const CropChainBwdArgs a, const POff po, const Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img_s = smem;
  __shared__ float red_s[4][4];
  __shared__ float dtp_s[8];
  SQ_PIN8(a.mode, a.slot, a.img, a.rec_prev, a.rec_new, a.d_rec_prev, a.d_rec_new, a.wb);
  SQ_PIN8(a.wb_ld, a.mask, a.mask_row_mul, a.mask_row_add, a.d_mask, a.g_out, a.g_row_mul, a.g_row_add);
  SQ_PIN8(a.tp, a.tp_ld, a.noise, a.flat, a.w3, a.t2, a.t2_ld, a.d_t2);
  // one workgroup per particle row (the K particles of a sequence re-stage the same frame from L2: 10 KB at 50x50)
  const int r = sq_row_of_wg(blockIdx.x, d), b = sq_div(r, d.k_mul), tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int slot = a.mode == CROP_PROP1 ? (int)blockIdx.y : a.slot;
  const int P = d.H * d.W, G = d.G, G2 = d.G * d.G, RW = rec::W;
  const float* img = a.img + (size_t)b * d.P4;
  const int madd = a.mask_row_add + (a.mode == CROP_PROP1 ? slot : 0);
  const int gadd = a.g_row_add + (a.mode == CROP_PROP1 ? slot : 0);
  // everything the later phases read from memory is requested up front, next to the frame: the kernel then makes ONE memory
  // round trip (it used to make three: frame | glimpse gradient + mask | the operands of the where-sample adjoint).  The
  // frame's first 1024 16-byte units go out FIRST -- they still have to pass through LDS once they are here -- and the fences keep
  // hipcc from computing every address of the kernel before it issues the first request.
  // Large frames (SQ_CROP_STAGE_MAX_PIXELS; 128 x 128 = 4096 units): every unit on the LDS-DMA path, requested HERE in one batch
  // (16 wave instructions of 1 KB each) -- through registers the units beyond the first 1024 were three more dependent round
  // trips (5.8 us per launch at 128 x 128 against 4.0 at 50 x 50; reading the taps where they lie instead, after the where
  // logits have arrived, was slower still: 6.3 us -- a second COLD round trip, the frame was last touched a forward pass ago)
  if (!STAGED) sq_wave_stage16(img_s, img, d.P4 >> 2, lane, wave, 4);
  const int n4 = STAGED ? d.P4 >> 2 : 1;   // (frames are 16-byte aligned and padded to a multiple of 4 floats: Dims.P4)
  const f32x4_b* __restrict__ s4 = reinterpret_cast<const f32x4_b*>(img);
  f32x4_b fv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) fv[q] = s4[min(tid + 256 * q, n4 - 1)];
  __builtin_amdgcn_sched_barrier(0);
  constexpr int PPT = 4;  // pixels per thread prefetched (covers G * G <= 1024)
  float g_v[PPT], m_v[PPT], dm_v[PPT];
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    const int pix = min(tid + 256 * q, G2 - 1);
    g_v[q] = a.g_out[((size_t)r * a.g_row_mul + gadd) * G2 + pix];
    const size_t mi = ((size_t)r * a.mask_row_mul + madd) * G2 + pix;
    m_v[q] = a.mask != nullptr ? a.mask[mi] : 1.0f;
    dm_v[q] = a.mask != nullptr ? a.d_mask[mi] : 0.0f;
  }
  float wl[4];
  {
    const float* wsrc = a.mode == CROP_PROP1 ? a.rec_prev + ((size_t)r * d.N + slot) * RW + rec::WHERE
                                             : a.rec_new + ((size_t)r * d.N + slot) * RW + rec::WHERE;
    // (unconditional loads from a valid address, then the select: behind the ternary each element was a guarded load with a
    // full wait of its own)
    const bool p1 = a.mode == CROP_PROP1;
    const float* wbp = p1 ? a.wb + ((size_t)r * d.N + slot) * a.wb_ld : wsrc;
    float w0[4], w1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { w0[i] = wsrc[i]; w1[i] = wbp[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) wl[i] = w0[i] + (p1 ? 0.1f * w1[i] : 0.0f);
  }
  // operands of the four where coordinates' adjoint (threads 0..3)
  const int ci = tid & 3;
  float* drn = a.d_rec_new + ((size_t)r * d.N + slot) * RW;
  float q_dw = 0.0f, q_dloc = 0.0f, q_dsc = 0.0f, q_eps[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q_tp = 0.0f, q_off = 0.0f, q_dprev = 0.0f, q_ch[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (tid < 4) {
    q_dprev = a.d_rec_prev[((size_t)r * d.N + slot) * RW + rec::WHERE + ci];
    if (a.mode != CROP_PROP1) {
      q_dw = drn[rec::WHERE + ci]; q_dloc = drn[rec::WHERE_LOC + ci]; q_dsc = drn[rec::WHERE_SCALE + ci];
      const float* eps = a.noise + (((size_t)r * 2 + (a.mode == CROP_DISC ? 1 : 0)) * d.N + slot) * d.nzw;
#pragma unroll
      for (int j = 0; j < 4; ++j) q_eps[j] = eps[j];
      q_tp = a.tp[(size_t)r * a.tp_ld + 4 + ci];
      q_off = a.flat[a.mode == CROP_DISC ? po.disc_scale_offset : po.prop_scale_offset];
      if (a.mode != CROP_DISC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) q_ch[j] = tril4(a.flat + po.cholesky, ci, min(j, ci));
      }
    }
  }
  // operands of the fused output-layer adjoint: this thread's row of w3 and its saved activation
  const bool fuse_t3 = a.d_t2 != nullptr && a.mode != CROP_PROP1;
  f32x4_b w3a = {0.0f, 0.0f, 0.0f, 0.0f}, w3b = w3a;
  float t2v = 0.0f;
  if (fuse_t3 && tid < d.nh) {
    w3a = *reinterpret_cast<const f32x4_b*>(a.w3 + (size_t)tid * 8);
    w3b = *reinterpret_cast<const f32x4_b*>(a.w3 + (size_t)tid * 8 + 4);
    t2v = a.t2[(size_t)r * a.t2_ld + tid];
  }
  if (STAGED) {  // frame -> LDS in 16-byte units, 4 loads per thread in flight at once.  The plain copy loop compiled to three or four
     // dependent round trips (an unrolled trip + remainder loops waiting per element).
    f32x4_b* d4 = reinterpret_cast<f32x4_b*>(img_s);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (tid + 256 * q < n4) d4[tid + 256 * q] = fv[q];
    for (int base = 1024; base < n4; base += 1024) {   // frames beyond 4096 pixels
      f32x4_b v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = s4[min(base + tid + 256 * q, n4 - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (base + tid + 256 * q < n4) d4[base + tid + 256 * q] = v[q];
    }
  }
  __syncthreads();
  {
    const float s0 = sq_sigmoid_geo(wl[0]), s1 = sq_sigmoid_geo(wl[1]);
    const float sx = fmaxf(s0, 1e-4f), sy = fmaxf(s1, 1e-4f), tx = tanhf(wl[2]), ty = tanhf(wl[3]);
    const float hx = 0.5f * (float)(d.W - 1), hy = 0.5f * (float)(d.H - 1);
    float dsx = 0.0f, dsy = 0.0f, dtx = 0.0f, dty = 0.0f;
    auto pixel = [&](const int pix, float g, const float mk, const float dmk) {
      const int i = sq_div(pix, d.g_mul), j = pix - i * G;
      const float gx = -1.0f + 2.0f * (float)j / (float)(G - 1), gy = -1.0f + 2.0f * (float)i / (float)(G - 1);
      const float x = hx * (sx * gx + tx + 1.0f), y = hy * (sy * gy + ty + 1.0f);
      const float x0f = floorf(x), y0f = floorf(y);
      const int x0 = (int)x0f, y0 = (int)y0f;
      const float wx1 = x - x0f, wy1 = y - y0f;
      float t[2][2];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int yy = y0 + dy, xx = x0 + dx;
          t[dy][dx] = (yy >= 0 && yy < d.H && xx >= 0 && xx < d.W) ? img_s[yy * d.W + xx] : 0.0f;
        }
      const float v = (1.0f - wy1) * ((1.0f - wx1) * t[0][0] + wx1 * t[0][1]) + wy1 * ((1.0f - wx1) * t[1][0] + wx1 * t[1][1]);
      const float dvdx = (1.0f - wy1) * (t[0][1] - t[0][0]) + wy1 * (t[1][1] - t[1][0]);
      const float dvdy = (1.0f - wx1) * (t[1][0] - t[0][0]) + wx1 * (t[1][1] - t[0][1]);
      if (a.mask != nullptr) {
        const float dm = dmk + g * v;
        a.d_mask[((size_t)r * a.mask_row_mul + madd) * G2 + pix] = a.mask_dact ? dm * (mk * (1.0f - mk)) : dm;
        g *= mk;
      }
      dsx += g * dvdx * hx * gx;
      dtx += g * dvdx * hx;
      dsy += g * dvdy * hy * gy;
      dty += g * dvdy * hy;
    };
#pragma unroll
    for (int q = 0; q < PPT; ++q)
      if (tid + 256 * q < G2) pixel(tid + 256 * q, g_v[q], m_v[q], dm_v[q]);
    for (int pix = tid + 256 * PPT; pix < G2; pix += 256) {  // glimpses beyond 32 x 32: operands fetched in place
      const size_t mi = ((size_t)r * a.mask_row_mul + madd) * G2 + pix;
      pixel(pix, a.g_out[((size_t)r * a.g_row_mul + gadd) * G2 + pix], a.mask != nullptr ? a.mask[mi] : 1.0f,
            a.mask != nullptr ? a.d_mask[mi] : 0.0f);
    }
    dsx = sq_wave_sum(dsx); dsy = sq_wave_sum(dsy); dtx = sq_wave_sum(dtx); dty = sq_wave_sum(dty);
    if (lane == 0) { red_s[wave][0] = dsx; red_s[wave][1] = dsy; red_s[wave][2] = dtx; red_s[wave][3] = dty; }
    __syncthreads();
    if (tid < 4) {
      const int i = tid;
      const float tot = red_s[0][i] + red_s[1][i] + red_s[2][i] + red_s[3][i];
      const float dl = i == 0 ? s0 * (1.0f - s0) : (i == 1 ? s1 * (1.0f - s1) : (i == 2 ? 1.0f - tx * tx : 1.0f - ty * ty));
      const float g_crop = tot * dl;
      if (a.mode == CROP_PROP1) {
        a.d_rec_prev[((size_t)r * d.N + slot) * RW + rec::WHERE + i] = q_dprev + g_crop;
        a.d_wb[((size_t)r * d.N + slot) * a.wb_ld + i] = 0.1f * g_crop;
      } else {
        const float dW = q_dw + g_crop;  // every other consumer has already accumulated here
        const float d_loc = dW + q_dloc;
        float d_sc = q_dsc;
        float d_raw;
        if (a.mode == CROP_DISC) {
          d_sc += dW * (i == 0 ? q_eps[0] : (i == 1 ? q_eps[1] : (i == 2 ? q_eps[2] : q_eps[3])));
          d_raw = d_sc * sq_sigmoid(q_tp + q_off);
        } else {
          float lin = 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j <= i) lin += (q_ch[j] + (i == j ? 1.0f : 0.0f)) * q_eps[j];
          d_sc += dW * lin;
          d_raw = d_sc * sq_sigmoid(q_tp + q_off - 1.0f);
          a.d_rec_prev[((size_t)r * d.N + slot) * RW + rec::WHERE + i] = q_dprev + d_loc;  // loc = where_{t-1} + transform
        }
        a.d_tp[(size_t)r * a.dtp_ld + i] = d_loc;
        a.d_tp[(size_t)r * a.dtp_ld + 4 + i] = d_raw;
        dtp_s[i] = d_loc;
        dtp_s[4 + i] = d_raw;
        // total gradient of the sample, kept for k_where_param_grads (scale offsets / Cholesky factor: one batched
        // reduction over all uses at the end of the sweep instead of 640-way contended atomics per launch)
        drn[rec::WHERE + i] = dW;
      }
    }
    if (fuse_t3) {  // (wave-uniform) d t2 = d tp W3^T, times elu' from the saved output
      __syncthreads();
      for (int j = tid; j < d.nh; j += 256) {
        if (j >= 256) {
          w3a = *reinterpret_cast<const f32x4_b*>(a.w3 + (size_t)j * 8);
          w3b = *reinterpret_cast<const f32x4_b*>(a.w3 + (size_t)j * 8 + 4);
          t2v = a.t2[(size_t)r * a.t2_ld + j];
        }
        const float g = dtp_s[0] * w3a[0] + dtp_s[1] * w3a[1] + dtp_s[2] * w3a[2] + dtp_s[3] * w3a[3] + dtp_s[4] * w3b[0] +
                        dtp_s[5] * w3b[1] + dtp_s[6] * w3b[2] + dtp_s[7] * w3b[3];
        a.d_t2[(size_t)r * a.dt2_ld + j] = g * delu_from_out(t2v);
      }
    }
  }
}
// gradients of transform.scale_offset (prop / disc) and of the Cholesky factor of the propagation where-posterior over all
// (frame, row, slot) uses: rows of the tapes [T][R][N]; d_tp [2 phases][rows][ld], gradient records hold the total d where
__global__ __launch_bounds__(256) void k_where_param_grads(const float* __restrict__ d_tp, int tp_ld, const float* __restrict__ d_rec_p,
                                                           const float* __restrict__ rec_p, const float* __restrict__ noise, int rows,
                                                           int RN, int N, int nzw, float* __restrict__ flat_grad, POff po SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ float red[4][12];
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
  for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += gridDim.x * blockDim.x) {
    const int t = row / RN, rk = row - t * RN, r = rk / N, k = rk - r * N;
    const float* tp_p = d_tp + (size_t)row * tp_ld;
    const float* tp_d = d_tp + ((size_t)rows + row) * tp_ld;
    acc[10] += tp_p[4] + tp_p[5] + tp_p[6] + tp_p[7];
    acc[11] += tp_d[4] + tp_d[5] + tp_d[6] + tp_d[7];
    const float* eps = noise + ((((size_t)t * (RN / N) + r) * 2 + 0) * N + k) * nzw;
    const float* dr = d_rec_p + (size_t)row * rec::W;
    const float* rp = rec_p + (size_t)row * rec::W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g = dr[rec::WHERE + i] * rp[rec::WHERE_SCALE + i];
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const int q = i * 4 + j;  // fill_triangular index -> element of cholesky_scale
        acc[q < 6 ? 4 + q : 15 - q] += g * eps[j];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = sq_wave_sum(acc[i]);
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 12; ++i) red[threadIdx.x >> 6][i] = acc[i];
  __syncthreads();
  if (threadIdx.x < 12) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    float* dst = threadIdx.x < 10 ? flat_grad + po.cholesky + threadIdx.x
                                  : (threadIdx.x == 10 ? flat_grad + po.prop_scale_offset : flat_grad + po.disc_scale_offset);
    unsafeAtomicAdd(dst, v);
  }
}
int sq_launch_where_param_grads(const float* d_tp, int tp_ld, const float* d_rec_p, const float* rec_p, const float* noise, int T,
                                Dims d, float* flat_grad, POff po, hipStream_t s) {
  const int rows = T * d.R * d.N;
  SQ_LAUNCH(k_where_param_grads, dim3(32), dim3(256), 0, s, d_tp, tp_ld, d_rec_p, rec_p, noise, rows, d.R * d.N, d.N, d.nzw,
                     flat_grad, po);
  return 0;
}
int sq_launch_crop_chain_bwd(const CropChainBwdArgs& a, POff po, Dims d, int nslots, hipStream_t s) {
  if ((d.P4 & 3) != 0) return -1;   // (the frame is staged in 16-byte units: callers pass the padded copy, Dims.P4)
  const size_t shm = (size_t)d.P4 * sizeof(float);
  if (d.H * d.W <= SQ_CROP_STAGE_MAX_PIXELS) SQ_LAUNCH(k_crop_chain_bwd<true>, dim3(d.R, nslots), dim3(256), shm, s, a, po, d);
  else {
    if (shm > 48 * 1024 && sq_allow_big_lds((const void*)k_crop_chain_bwd<false>, 150 * 1024) != 0) return -2;
    SQ_LAUNCH(k_crop_chain_bwd<false>, dim3(d.R, nslots), dim3(256), shm, s, a, po, d);
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// GRU gate adjoints (snt.GRU: z, r gates; hc = tanh(x W_h + (r h) U_h + b_h); h' = (1 - z) h + z hc)
//   stage A: from d h' -> dPre columns [d_az | . | d_ah] and the direct part of d h
//   stage B: after d(r h) = d_ah U_h^T -> d_ar, d h += d(rh) r
// dpre1 is [rows][3 nh] (the dPre of the fused gate GEMM); rows addressed with explicit strides.
// ------------------------------------------------------------------------------------------------
__global__ void k_gru_bwd_a(const float* __restrict__ d_hn, int dhn_ld, const float* __restrict__ z, int z_ld,
                            const float* __restrict__ hc, int hc_ld, const float* __restrict__ hprev, int h_ld,
                            float* __restrict__ dpre1, int dp_ld, float* __restrict__ d_h, int dh_ld, int rows, int nh,
                            int accumulate_dh, float* __restrict__ dup_z, int dup_ld, int dup_h_off SQ_TLP) {
  SQ_TL_SCOPE;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nh) return;
  const int m = i / nh, n = i - m * nh;
  const float g = d_hn[(size_t)m * dhn_ld + n];
  const float zz = z[(size_t)m * z_ld + n], hh = hc[(size_t)m * hc_ld + n], hp = hprev[(size_t)m * h_ld + n];
  const float dz = g * (hh - hp) * zz * (1.0f - zz);
  dpre1[(size_t)m * dp_ld + n] = dz;
  if (dup_z != nullptr) dup_z[(size_t)m * dup_ld + n] = dz;
  const float dc = g * zz * (1.0f - hh * hh);
  dpre1[(size_t)m * dp_ld + 2 * nh + n] = dc;
  if (dup_z != nullptr && dup_h_off >= 0) dup_z[(size_t)m * dup_ld + dup_h_off + n] = dc;
  float* dh = d_h + (size_t)m * dh_ld + n;
  *dh = (accumulate_dh ? *dh : 0.0f) + g * (1.0f - zz);
}
__global__ void k_gru_bwd_b(const float* __restrict__ d_rh, int drh_ld, const float* __restrict__ rg, int r_ld,
                            const float* __restrict__ hprev, int h_ld, float* __restrict__ dpre1, int dp_ld,
                            float* __restrict__ d_h, int dh_ld, int rows, int nh, float* __restrict__ dup_r, int dup_ld SQ_TLP) {
  SQ_TL_SCOPE;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nh) return;
  const int m = i / nh, n = i - m * nh;
  const float g = d_rh[(size_t)m * drh_ld + n], rr = rg[(size_t)m * r_ld + n], hp = hprev[(size_t)m * h_ld + n];
  const float dr = g * hp * rr * (1.0f - rr);
  dpre1[(size_t)m * dp_ld + nh + n] = dr;
  if (dup_r != nullptr) dup_r[(size_t)m * dup_ld + n] = dr;
  d_h[(size_t)m * dh_ld + n] += g * rr;
}
int sq_launch_gru_bwd_a(const float* d_hn, int dhn_ld, const float* z, int z_ld, const float* hc, int hc_ld,
                        const float* hprev, int h_ld, float* dpre1, int dp_ld, float* d_h, int dh_ld, int rows, int nh,
                        int accumulate_dh, hipStream_t s, float* dup_z, int dup_ld, int dup_h_off) {
  SQ_LAUNCH(k_gru_bwd_a, dim3((rows * nh + 255) / 256), dim3(256), 0, s, d_hn, dhn_ld, z, z_ld, hc, hc_ld, hprev,
                     h_ld, dpre1, dp_ld, d_h, dh_ld, rows, nh, accumulate_dh, dup_z, dup_ld, dup_h_off);
  return 0;
}
int sq_launch_gru_bwd_b(const float* d_rh, int drh_ld, const float* rg, int r_ld, const float* hprev, int h_ld,
                        float* dpre1, int dp_ld, float* d_h, int dh_ld, int rows, int nh, hipStream_t s, float* dup_r, int dup_ld) {
  SQ_LAUNCH(k_gru_bwd_b, dim3((rows * nh + 255) / 256), dim3(256), 0, s, d_rh, drh_ld, rg, r_ld, hprev, h_ld,
                     dpre1, dp_ld, d_h, dh_ld, rows, nh, dup_r, dup_ld);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// strided elementwise helpers
// ------------------------------------------------------------------------------------------------
// out[m][n] = (acc ? out : 0) + in[m][n] * act'(saved[m][n])   with per-column-range activations (act_a below split)
__global__ void k_dact2(const float* din, int in_ld, const float* __restrict__ saved, int s_ld,
                        float* dout, int out_ld, int rows, int cols, int act_a, int act_b, int split, int acc SQ_TLP) {
  SQ_TL_SCOPE;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int m = i / cols, n = i - m * cols;
  float g = din[(size_t)m * in_ld + n];
  const int act = n < split ? act_a : act_b;
  if (act != ACT_NONE) {
    const float o = saved[(size_t)m * s_ld + n];
    if (act == ACT_ELU) g *= delu_from_out(o);
    else if (act == ACT_TANH) g *= 1.0f - o * o;
    else if (act == ACT_SIGMOID) g *= o * (1.0f - o);
    else if (act == ACT_SOFTPLUS_MIN) g *= 1.0f - expf(-(o - 1e-2f));
  }
  float* p = dout + (size_t)m * out_ld + n;
  *p = (acc ? *p : 0.0f) + g;
}
int sq_launch_dact2(const float* din, int in_ld, const float* saved, int s_ld, float* dout, int out_ld, int rows, int cols,
                    int act_a, int act_b, int split, int acc, hipStream_t s) {
  SQ_LAUNCH(k_dact2, dim3((rows * cols + 255) / 256), dim3(256), 0, s, din, in_ld, saved, s_ld, dout, out_ld, rows,
                     cols, act_a, act_b, split, acc);
  return 0;
}
// column sums of dY[rows][cols] (+)= into out[cols] (bias gradients)
__global__ void k_colsum(const float* __restrict__ dy, int ld, int rows, int cols, float* __restrict__ out, int rows_per_block SQ_TLP) {
  SQ_TL_SCOPE;
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;  // 4 row partitions
  __shared__ float red[4][64];
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float sacc = 0.0f;
  if (n < cols)
    for (int m = r0 + part; m < r1; m += 4) sacc += dy[(size_t)m * ld + n];
  red[part][threadIdx.x & 63] = sacc;
  __syncthreads();
  if (part == 0 && n < cols) unsafeAtomicAdd(out + n, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
int sq_launch_colsum(const float* dy, int ld, int rows, int cols, float* out, int acc, hipStream_t s) {
  (void)acc;  // always accumulates (float atomics)
  const int rpb = 64;
  SQ_LAUNCH(k_colsum, dim3((cols + 63) / 64, (rows + rpb - 1) / rpb), dim3(256), 0, s, dy, ld, rows, cols, out, rpb);
  return 0;
}
// the same for up to SQ_CS_MAXE independent sums in one launch (ColsumBatch, sqair_bwd.h): a workgroup finds its entry from
// its index (wave-uniform) and runs k_colsum's body on 64 rows x 64 columns of it, each row optionally weighted
__global__ __launch_bounds__(256) void k_colsum_group(const ColsumGroup g SQ_TLP) {
  SQ_TL_SCOPE;
  int ei = 0;
  for (int i = 1; i < g.n; ++i) ei = (int)blockIdx.x >= g.e[i].first ? i : ei;
  const ColsumEntry e = g.e[__builtin_amdgcn_readfirstlane(ei)];
  const int bl = (int)blockIdx.x - e.first, by = bl / e.nbx, bx = bl - by * e.nbx;
  const int n = bx * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;  // 4 row partitions
  __shared__ float red[4][64];
  const int r0 = by * 64, r1 = min(e.rows, r0 + 64);
  float sacc = 0.0f;
  if (n < e.cols) {
    if (e.wt != nullptr)
      for (int m = r0 + part; m < r1; m += 4) sacc += e.dy[(size_t)m * e.ld + n] * e.wt[(size_t)m * e.wld];
    else
      for (int m = r0 + part; m < r1; m += 4) sacc += e.dy[(size_t)m * e.ld + n];
  }
  red[part][threadIdx.x & 63] = sacc;
  __syncthreads();
  if (part == 0 && n < e.cols) unsafeAtomicAdd(e.out + n, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
void ColsumBatch::add(const float* dy, int ld, int rows, int cols, float* out, hipStream_t s, const float* wt, int wld) {
  if (rows <= 0 || cols <= 0) return;
  if (g.n == SQ_CS_MAXE) (void)flush(s);
  ColsumEntry& e = g.e[g.n++];
  e.dy = dy; e.wt = wt; e.out = out; e.ld = ld; e.wld = wld; e.rows = rows; e.cols = cols;
  e.first = total; e.nbx = (cols + 63) / 64;
  total += e.nbx * ((rows + 63) / 64);
}
int ColsumBatch::flush(hipStream_t s) {
  if (g.n > 0) SQ_LAUNCH(k_colsum_group, dim3(total), dim3(256), 0, s, g);
  g.n = 0;
  total = 0;
  return 0;
}
// latent-summary adjoint: d f[(r,k)][n] = d c[r][n] * presence_k ; particle sum: d pre_disc[b][n] = sum_kp d pre_d[b K + kp][n]
__global__ void k_latent_sum_bwd(const float* __restrict__ d_c, const float* __restrict__ rec_p, const float* __restrict__ f_out,
                                 float* __restrict__ d_f, Dims d SQ_TLP) {
  SQ_TL_SCOPE;
  const int rk = blockIdx.x;  // r * N + k;  f_out = elu(.) output of the summed feature: its derivative is applied here
  const float pres = rec_p[(size_t)rk * rec::W + rec::PRES];
  for (int n = threadIdx.x; n < d.nh; n += blockDim.x)
    d_f[(size_t)rk * d.nh + n] = d_c[(size_t)(rk / d.N) * d.nh + n] * pres * delu_from_out(f_out[(size_t)rk * d.nh + n]);
}
int sq_launch_latent_sum_bwd(const float* d_c, const float* rec_p, const float* f_out, float* d_f, Dims d, hipStream_t s) {
  SQ_LAUNCH(k_latent_sum_bwd, dim3(d.R * d.N), dim3(256), 0, s, d_c, rec_p, f_out, d_f, d);
  return 0;
}
// d pre_d[r][n] = sum over the N discovery slots of the RNN pre-activation gradients [R][N][nh]; then over particles
// d_pre_d[r] = sum over the N slots of d_rnn[r][j], d_pre_disc[b] = sum over the K particles of d_pre_d: workgroup = (sequence,
// 64 columns), one wave per particle (K > 16: strided), every load of the pass in flight at once (a thread per (b, column)
// walking its K x N values was K x N dependent round trips: 4.6 us)
__global__ __launch_bounds__(1024) void k_sum_slots(const float* __restrict__ d_rnn, float* __restrict__ d_pre_d, float* __restrict__ d_pre_disc,
                                                   int B, int K, int N, int nh SQ_TLP) {
  SQ_TL_SCOPE;
  __shared__ float part_s[16][64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const int n = blockIdx.y * 64 + lane;
  const bool ok = n < nh;
  float tot = 0.0f;
  for (int kp = wave; kp < K; kp += n_waves) {
    const int r = b * K + kp;
    float v[SQ_MAXN];
#pragma unroll
    for (int j = 0; j < SQ_MAXN; ++j) v[j] = ok && j < N ? d_rnn[((size_t)r * N + j) * nh + n] : 0.0f;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < SQ_MAXN; ++j) acc += v[j];
    if (ok) d_pre_d[(size_t)r * nh + n] = acc;
    tot += acc;
  }
  part_s[wave][lane] = tot;
  __syncthreads();
  if (wave == 0 && ok) {
    float t = 0.0f;
    for (int w = 0; w < n_waves; ++w) t += part_s[w][lane];
    d_pre_disc[(size_t)b * nh + n] = t;
  }
}
int sq_launch_sum_slots(const float* d_rnn, float* d_pre_d, float* d_pre_disc, int B, int K, int N, int nh, hipStream_t s) {
  const int n_waves = K < 16 ? K : 16;
  SQ_LAUNCH(k_sum_slots, dim3(B, (nh + 63) / 64), dim3(64 * n_waves), 0, s, d_rnn, d_pre_d, d_pre_disc, B, K, N, nh);
  return 0;
}
__global__ void k_particle_sum(const float* __restrict__ in, float* __restrict__ out, int B, int K, int nh SQ_TLP) {
  SQ_TL_SCOPE;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nh) return;
  const int b = i / nh, n = i - b * nh;
  float acc = 0.0f;
  for (int kp = 0; kp < K; ++kp) acc += in[(size_t)(b * K + kp) * nh + n];
  out[i] = acc;
}
int sq_launch_particle_sum(const float* in, float* out, int B, int K, int nh, hipStream_t s) {
  SQ_LAUNCH(k_particle_sum, dim3((B * nh + 255) / 256), dim3(256), 0, s, in, out, B, K, nh);
  return 0;
}
// y[m][n] (+)= x[m][n] over a strided sub-block
__global__ void k_axpy2d(const float* __restrict__ x, int x_ld, float* __restrict__ y, int y_ld, int rows, int cols, int acc SQ_TLP) {
  SQ_TL_SCOPE;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int m = i / cols, n = i - m * cols;
  float* p = y + (size_t)m * y_ld + n;
  *p = (acc ? *p : 0.0f) + x[(size_t)m * x_ld + n];
}
int sq_launch_axpy2d(const float* x, int x_ld, float* y, int y_ld, int rows, int cols, int acc, hipStream_t s) {
  SQ_LAUNCH(k_axpy2d, dim3((rows * cols + 255) / 256), dim3(256), 0, s, x, x_ld, y, y_ld, rows, cols, acc);
  return 0;
}
