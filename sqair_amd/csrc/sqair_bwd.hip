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

__global__ __launch_bounds__(256) void k_crop_bwd(const CropBwdArgs a, const Dims d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img_s = smem;  // H*W
  __shared__ float red_s[4][4];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int P = d.H * d.W, G = d.G, G2 = d.G * d.G;
  const float* img = a.img + (size_t)b * P;
  for (int i = tid; i < P; i += 256) img_s[i] = img[i];
  __syncthreads();
  for (int kp = 0; kp < d.K; ++kp) {
    const int r = b * d.K + kp;
    const float l0 = a.logits[(size_t)r * 4 + 0], l1 = a.logits[(size_t)r * 4 + 1];
    const float l2 = a.logits[(size_t)r * 4 + 2], l3 = a.logits[(size_t)r * 4 + 3];
    const float s0 = sq_sigmoid(l0), s1 = sq_sigmoid(l1);
    const float sx = fmaxf(s0, 1e-4f), sy = fmaxf(s1, 1e-4f), tx = tanhf(l2), ty = tanhf(l3);
    const float hx = 0.5f * (float)(d.W - 1), hy = 0.5f * (float)(d.H - 1);
    float dsx = 0.0f, dsy = 0.0f, dtx = 0.0f, dty = 0.0f;
    for (int pix = tid; pix < G2; pix += 256) {
      const int i = pix / G, j = pix - i * G;
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
  Dims d{c.img_h, c.img_w, c.glimpse_size, c.n_steps_per_image, c.n_what, c.n_hidden, c.k_particles,
         B * c.k_particles, B, 4 + c.n_what + 1};
  CropBwdArgs a{img, where_logits, mask, g_out, d_where_logits, d_mask};
  const size_t shm = (size_t)d.H * d.W * sizeof(float);
  static bool big = false;
  if (shm > 48 * 1024 && !big) {
    (void)hipFuncSetAttribute((const void*)k_crop_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipGetLastError();
    big = true;
  }
  hipLaunchKernelGGL(k_crop_bwd, dim3(B), dim3(256), shm, (hipStream_t)stream, a, d);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// insert + log-likelihood backward.  One workgroup per row b'.  Pass over the canvas pixels exactly like
// the forward kernel; per pixel the scalar adjoints g_cv (canvas) and g_ms (written-to mask sum) are pushed
// back into (a) the N glimpses — LDS float atomics on a tile that is written out once, (b) the inverse-warp
// coordinates -> 4 reductions per slot, (c) the mean image (per-row contribution, summed over rows by a
// second tiny kernel so that the result is deterministic).
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
};

__global__ __launch_bounds__(256) void k_insert_loglik_bwd(const InsertBwdArgs a, const Dims d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = d.N, G = d.G, G2 = d.G * d.G, H = d.H, W = d.W, P = d.H * d.W;
  float* gl_s = smem;                 // N * G2   glimpses
  float* dg_s = gl_s + N * G2;        // N * G2   glimpse gradients
  float* xt_s = dg_s + N * G2;        // N * W
  float* yt_s = xt_s + N * W;         // N * H
  float* pres_s = yt_s + N * H;       // N
  float* co_s = pres_s + N;           // N * 4  (sx, sy, tx, ty)
  float* acc_s = co_s + N * 4;        // 4 waves * N * 4
  const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = blockIdx.y;  // frame
  const int b = r / d.K;
  const size_t fs = (size_t)fr * d.R * N + (size_t)r * N;  // first slot-row of this (frame, row)
  const size_t frr = (size_t)fr * d.R + r;
  for (int i = tid; i < N * G2; i += 256) { gl_s[i] = a.glimpse[fs * G2 + i]; dg_s[i] = 0.0f; }
  if (tid < N * 4) {
    const int k = tid >> 2, c = tid & 3;
    const float l = a.rec ? a.rec[(fs + k) * a.rec_ld + rec::WHERE + c] : a.where[fs * 4 + tid];
    co_s[tid] = (tid & 2) ? tanhf(l) : fmaxf(sq_sigmoid(l), 1e-4f);
  }
  if (tid < N) pres_s[tid] = a.rec ? a.rec[(fs + tid) * a.rec_ld + rec::PRES] : a.pres[fs + tid];
  __syncthreads();
  for (int i = tid; i < N * (W + H); i += 256) {
    const int k = i / (W + H), q = i % (W + H);
    const bool is_y = q >= W;
    const int j = is_y ? q - W : q;
    const float sc = co_s[k * 4 + (is_y ? 1 : 0)], tr = co_s[k * 4 + (is_y ? 3 : 2)];
    const float L = (float)((is_y ? H : W) - 1);
    const float cn = -1.0f + 2.0f * (float)j / L;
    const float g = 0.5f * (float)(G - 1) * ((cn - tr) / sc + 1.0f);
    if (is_y) yt_s[k * H + j] = g; else xt_s[k * W + j] = g;
  }
  __syncthreads();
  const float gll = a.g_ll[frr];
  const float* img = a.img + ((size_t)fr * d.B + b) * P;
  float dco[SQ_MAXN][4];
#pragma unroll
  for (int k = 0; k < SQ_MAXN; ++k) dco[k][0] = dco[k][1] = dco[k][2] = dco[k][3] = 0.0f;
  const float hg = 0.5f * (float)(G - 1);
  for (int pix = tid; pix < P; pix += 256) {
    const int Y = pix / W, X = pix - Y * W;
    // ---- forward recompute of canvas / mask sum at this pixel
    float cv = 0.0f, ms = 0.0f;
    for (int k = 0; k < N; ++k) {
      const float pk = pres_s[k];
      if (pk == 0.0f) continue;
      const float xg = xt_s[k * W + X], yg = yt_s[k * H + Y];
      if (!(xg > -1.0f && xg < (float)G && yg > -1.0f && yg < (float)G)) continue;
      const float x0f = floorf(xg), y0f = floorf(yg);
      const int x0 = (int)x0f, y0 = (int)y0f;
      const float wx1 = xg - x0f, wy1 = yg - y0f;
      const float* gk = gl_s + k * G2;
      float v = 0.0f, on = 0.0f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int yy = y0 + dy;
        if (yy < 0 || yy >= G) continue;
        const float wy = dy ? wy1 : 1.0f - wy1;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int xx = x0 + dx;
          if (xx < 0 || xx >= G) continue;
          const float w = wy * (dx ? wx1 : 1.0f - wx1);
          v += w * gk[yy * G + xx];
          on += w;
        }
      }
      cv += v * pk;
      ms += on * pk;
    }
    const float m = sq_sigmoid(-10.0f + ms * 20.0f);
    const float mean = a.mean_img[pix];
    cv += mean * m;
    const float sd = m * a.std_fg + (1.0f - m) * a.std_bg;
    const float diff = img[pix] - cv;
    // ---- adjoints
    const float g_cv = gll * diff / (sd * sd);
    const float g_sd = gll * (diff * diff / (sd * sd * sd) - 1.0f / sd);
    const float g_m = g_cv * mean + g_sd * (a.std_fg - a.std_bg);
    const float g_ms = g_m * 20.0f * m * (1.0f - m);
    a.d_mean_rows[frr * P + pix] = g_cv * m;
    for (int k = 0; k < N; ++k) {
      const float pk = pres_s[k];
      if (pk == 0.0f) continue;
      const float xg = xt_s[k * W + X], yg = yt_s[k * H + Y];
      if (!(xg > -1.0f && xg < (float)G && yg > -1.0f && yg < (float)G)) continue;
      const float x0f = floorf(xg), y0f = floorf(yg);
      const int x0 = (int)x0f, y0 = (int)y0f;
      const float wx1 = xg - x0f, wy1 = yg - y0f;
      const float* gk = gl_s + k * G2;
      float t[2][2], vl[2][2];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int yy = y0 + dy, xx = x0 + dx;
          const bool ok = yy >= 0 && yy < G && xx >= 0 && xx < G;
          vl[dy][dx] = ok ? 1.0f : 0.0f;
          t[dy][dx] = ok ? gk[yy * G + xx] : 0.0f;
          if (ok) {
            const float w = (dy ? wy1 : 1.0f - wy1) * (dx ? wx1 : 1.0f - wx1);
            atomicAdd(&dg_s[k * G2 + yy * G + xx], g_cv * pk * w);
          }
        }
      // d/d xg, d/d yg of (g_cv * bilinear(glimpse) + g_ms * bilinear(ones))
      const float dvdx = (1.0f - wy1) * (t[0][1] - t[0][0]) + wy1 * (t[1][1] - t[1][0]);
      const float dvdy = (1.0f - wx1) * (t[1][0] - t[0][0]) + wx1 * (t[1][1] - t[0][1]);
      const float dodx = (1.0f - wy1) * (vl[0][1] - vl[0][0]) + wy1 * (vl[1][1] - vl[1][0]);
      const float dody = (1.0f - wx1) * (vl[1][0] - vl[0][0]) + wx1 * (vl[1][1] - vl[0][1]);
      const float gx = pk * (g_cv * dvdx + g_ms * dodx), gy = pk * (g_cv * dvdy + g_ms * dody);
      const float sx = co_s[k * 4 + 0], sy = co_s[k * 4 + 1], tx = co_s[k * 4 + 2], ty = co_s[k * 4 + 3];
      const float Xn = -1.0f + 2.0f * (float)X / (float)(W - 1), Yn = -1.0f + 2.0f * (float)Y / (float)(H - 1);
      dco[k][0] += gx * (-hg * (Xn - tx) / (sx * sx));
      dco[k][1] += gy * (-hg * (Yn - ty) / (sy * sy));
      dco[k][2] += gx * (-hg / sx);
      dco[k][3] += gy * (-hg / sy);
    }
  }
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = sq_wave_sum(dco[k][c]);
      if (lane == 0) acc_s[(wave * N + k) * 4 + c] = v;
    }
  __syncthreads();
  for (int i = tid; i < N * G2; i += 256) a.d_glimpse[fs * G2 + i] = dg_s[i];
  if (tid < N * 4) {
    const int k = tid >> 2, c = tid & 3;
    const float tot = acc_s[(0 * N + k) * 4 + c] + acc_s[(1 * N + k) * 4 + c] + acc_s[(2 * N + k) * 4 + c] + acc_s[(3 * N + k) * 4 + c];
    const float l = a.rec ? a.rec[(fs + k) * a.rec_ld + rec::WHERE + c] : a.where[fs * 4 + tid];
    const float sg = sq_sigmoid(l), th = tanhf(l);
    a.d_where[(fs + k) * a.dw_ld + c] = tot * ((c & 2) ? 1.0f - th * th : sg * (1.0f - sg));
  }
}

__global__ void k_reduce_rows(const float* __restrict__ rows, float* __restrict__ out, int R, int P, int accumulate) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float acc = accumulate ? out[p] : 0.0f;
  for (int r = 0; r < R; ++r) acc += rows[(size_t)r * P + p];
  out[p] = acc;
}

extern "C" int sqair_st_insert_loglik_bwd(SqairHandle* h, const float* glimpse, const float* where_logits,
                                          const float* presence, const float* img, const float* mean_img,
                                          const float* g_data_ll, float* d_glimpse, float* d_where_logits,
                                          float* d_mean_img, void* scratch, int64_t scratch_bytes, int B, void* stream) {
  if (!h || !glimpse || !where_logits || !presence || !img || !mean_img || !g_data_ll || !d_glimpse || !d_where_logits ||
      !d_mean_img || !scratch || B < 1)
    return -1;
  SqairConfig c;
  if (sqair_get_config(h, &c) != 0) return -1;
  Dims d{c.img_h, c.img_w, c.glimpse_size, c.n_steps_per_image, c.n_what, c.n_hidden, c.k_particles,
         B * c.k_particles, B, 4 + c.n_what + 1};
  const int P = d.H * d.W;
  if (scratch_bytes < (int64_t)d.R * P * 4) return -1;
  InsertBwdArgs a{glimpse, where_logits, presence, img, mean_img, g_data_ll, d_glimpse, d_where_logits, (float*)scratch,
                  c.output_std, c.background_std, nullptr, 0, 4};
  const size_t shm = ((size_t)2 * d.N * d.G * d.G + (size_t)d.N * (d.W + d.H) + d.N + d.N * 4 + 4 * d.N * 4) * sizeof(float);
  hipLaunchKernelGGL(k_insert_loglik_bwd, dim3(d.R, 1), dim3(256), shm, (hipStream_t)stream, a, d);
  hipLaunchKernelGGL(k_reduce_rows, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)scratch,
                     d_mean_img, d.R, P, 0);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// VIMCO target backward (targets.py:62-75, model.py:150-158): L = mean_{b,k}(-elbo_b - sig_bk * dl_bk) / T with
// sig a stop-gradient  =>  dL/d log_w[t, b, k] = -softmax_k(log_w_b)[k] / (B T),  dL/d disc_lp[t, b, k] = -sig_bk / (B K T)
// ------------------------------------------------------------------------------------------------
__global__ void k_elbo_bwd(const float* __restrict__ iw, const float* __restrict__ sig, int T, int B, int K,
                           float* __restrict__ g_log_w_t, float* __restrict__ g_disc_lp_t) {
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
  hipLaunchKernelGGL(k_elbo_bwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, importance_weights,
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
                                               const float* __restrict__ alpha_ptr) {
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
  hipLaunchKernelGGL(k_wgrad, dim3((Kdim + 15) / 16, (Ndim + 15) / 16), dim3(256), 0, s, A, lda, dY, ldy, dW, ldw, db, M,
                     Kdim, Ndim, accumulate, rowmap, alpha_ptr);
  return 0;
}

// batched insert/log-likelihood adjoint over T frames on merged slot records (decoder branch of sqair_backward)
int sq_launch_insert_bwd_frames(const float* glimpse, const float* rec, int rec_ld, const float* img, const float* mean_img,
                                const float* g_ll, float* d_glimpse, float* d_rec, int d_rec_ld, float* d_mean_rows,
                                float std_fg, float std_bg, int T, Dims d, hipStream_t s) {
  InsertBwdArgs a{glimpse, nullptr, nullptr, img, mean_img, g_ll, d_glimpse, d_rec, d_mean_rows, std_fg, std_bg, rec, rec_ld,
                  d_rec_ld};
  const size_t shm = ((size_t)2 * d.N * d.G * d.G + (size_t)d.N * (d.W + d.H) + d.N + d.N * 4 + 4 * d.N * 4) * sizeof(float);
  hipLaunchKernelGGL(k_insert_loglik_bwd, dim3(d.R, T), dim3(256), shm, s, a, d);
  return 0;
}
int sq_launch_reduce_rows(const float* rows, float* out, int R, int P, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_rows, dim3((P + 255) / 256), dim3(256), 0, s, rows, out, R, P, accumulate);
  return 0;
}
int sq_launch_elbo_bwd(const float* iw, const float* sig, int T, int B, int K, float* g_lw, float* g_dl, hipStream_t s) {
  const int n = T * B * K;
  hipLaunchKernelGGL(k_elbo_bwd, dim3((n + 255) / 256), dim3(256), 0, s, iw, sig, T, B, K, g_lw, g_dl);
  return 0;
}

// d(output_scale) = sum(d_glimpse * glimpse) / scale   (glimpse = scale * raw; modules.py:144-147)
__global__ void k_dot_scale(const float* __restrict__ a, const float* __restrict__ b, int64_t n, const float* __restrict__ scale,
                            float* __restrict__ out) {
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
  hipLaunchKernelGGL(k_dot_scale, dim3(1), dim3(1024), 0, s, a, b, n, scale, out);
  return 0;
}

// Elementwise adjoint of the fused activation epilogue: dPre = dOut * act'(.) expressed through the saved OUTPUT
// (elu: out > 0 ? 1 : out + 1; tanh: 1 - out^2; sigmoid: out (1 - out); softplus(x) + c: 1 - exp(-(out - c))).
__global__ void k_dact(const float* __restrict__ d_out, const float* __restrict__ out, float* __restrict__ d_pre, int64_t n,
                       int act) {
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
  hipLaunchKernelGGL(k_dact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_out, out, d_pre, n, act);
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
                          float grad_scale) {
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
  hipLaunchKernelGGL(k_rmsprop, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat_params,
                     flat_grad, ms, mom, n, lr, decay, momentum, epsilon, grad_scale);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
