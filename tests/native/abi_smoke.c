/* Torch-free user of the C-ABI (include/sqair_hip.h): plain C + the HIP runtime.  Builds random parameters and inputs on the
 * host, runs the T-frame forward pass, the IWAE / VIMCO reductions, a full gradient evaluation and one RMSProp step, and
 * checks that everything is finite, that a second pass reproduces the first bit for bit and that the update moved the
 * parameters.  What a non-Python host of the reference's path (its load() -> Model boundary) would do.
 *   hipcc -x c tests/native/abi_smoke.c -Iinclude -Lsqair_amd -lsqair_hip -Wl,-rpath,$PWD/sqair_amd -o /tmp/abi_smoke   */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sqair_hip.h"

#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 2; } } while (0)
#define CKS(x) do { int r_ = (x); if (r_ != 0) { printf("%s failed (%d): %s\n", #x, r_, sqair_last_error(h)); return 3; } } while (0)

static unsigned long long rs = 88172645463325252ull;
static double urand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }
static double nrand(void) { double u = urand() + 1e-300, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }
static void* dmalloc(size_t bytes) { void* p = NULL; if (hipMalloc(&p, bytes) != hipSuccess) { printf("hipMalloc(%zu) failed\n", bytes); exit(2); } return p; }

int main(void) {
  const int T = 3, B = 4, K = 3, N = 3, H = 50, W = 50, R = B * K;
  SqairConfig cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.img_h = H; cfg.img_w = W; cfg.glimpse_size = 20; cfg.n_steps_per_image = N; cfg.n_what = 50; cfg.n_hidden = 256;
  cfg.k_particles = K; cfg.masked_glimpse = 1; cfg.rec_where_prior = 1; cfg.prop_prior_step_bias = 10.0f;
  cfg.step_success_prob = 0.75f; cfg.output_std = 0.3f; cfg.background_std = 0.3f; cfg.where_prior_mean[0] = cfg.where_prior_mean[1] = -2.0f;
  SqairHandle* h = NULL;
  if (sqair_create(&cfg, &h) != 0 || sqair_abi_version() != SQAIR_ABI_VERSION) { printf("sqair_create failed\n"); return 1; }
  const int64_t np = sqair_param_count(h);
  const int nzw = sqair_noise_width(h);
  printf("parameters %lld in %d entries, noise width %d\n", (long long)np, sqair_param_entries(h), nzw);
  /* host-side random parameters (small weights; the few structural ones set through their names) */
  float* hp = (float*)malloc((size_t)np * 4);
  for (int64_t i = 0; i < np; ++i) hp[i] = (float)(0.03 * nrand());
  for (int e = 0; e < sqair_param_entries(h); ++e) {
    const char* name; int64_t off, n;
    sqair_param_entry(h, e, &name, &off, &n);
    if (strstr(name, "output_scale")) hp[off] = 0.25f;
    if (strstr(name, "mean_img")) for (int64_t i = 0; i < n; ++i) hp[off + i] = 0.1f;
  }
  const size_t n_obs = (size_t)T * B * H * W, n_noise = (size_t)T * R * 2 * N * nzw;
  float* hobs = (float*)malloc(n_obs * 4);
  float* hnoise = (float*)malloc(n_noise * 4);
  for (size_t i = 0; i < n_obs; ++i) hobs[i] = (float)(urand() < 0.1 ? urand() : 0.0);
  for (size_t i = 0; i < n_noise; ++i) hnoise[i] = (i % nzw) == (size_t)(nzw - 1) ? (float)urand() : (float)nrand();
  hipStream_t s; CKH(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* flat = (float*)dmalloc((size_t)np * 4);
  float* grad = (float*)dmalloc((size_t)np * 4);
  float* ms = (float*)dmalloc((size_t)np * 4);
  float* mom = (float*)dmalloc((size_t)np * 4);
  void* packed = dmalloc((size_t)sqair_packed_bytes(h));
  float* obs = (float*)dmalloc(n_obs * 4);
  float* noise = (float*)dmalloc(n_noise * 4);
  const int64_t wsb = sqair_workspace_bytes(h, T, B), twb = sqair_train_workspace_bytes(h, T, B), bwb = sqair_backward_bytes(h, T, B);
  void* ws = dmalloc((size_t)wsb); void* tws = dmalloc((size_t)twb); void* scratch = dmalloc((size_t)bwb);
  float* lw_t = (float*)dmalloc((size_t)T * R * 4); float* dlp_t = (float*)dmalloc((size_t)T * R * 4);
  float* canvas = (float*)dmalloc((size_t)T * R * H * W * 4);
  float* log_w = (float*)dmalloc((size_t)R * 4); float* elbo_b = (float*)dmalloc((size_t)B * 4);
  float* iw = (float*)dmalloc((size_t)R * 4); float* sig = (float*)dmalloc((size_t)R * 4);
  float* scal = (float*)dmalloc(16 * 4); float* means = (float*)dmalloc(8 * 4);
  CKH(hipMemcpy(flat, hp, (size_t)np * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(obs, hobs, n_obs * 4, hipMemcpyHostToDevice));
  CKH(hipMemcpy(noise, hnoise, n_noise * 4, hipMemcpyHostToDevice));
  { float* ones = (float*)malloc((size_t)np * 4); for (int64_t i = 0; i < np; ++i) ones[i] = 1.0f;
    CKH(hipMemcpy(ms, ones, (size_t)np * 4, hipMemcpyHostToDevice)); free(ones); CKH(hipMemset(mom, 0, (size_t)np * 4)); }
  SqairOutputs out; memset(&out, 0, sizeof(out));
  out.log_weights_per_timestep = lw_t; out.discrete_log_prob = dlp_t; out.canvas = canvas;
  CKS(sqair_pack_params(h, flat, packed, s));
  float sc[2][16];
  for (int pass = 0; pass < 2; ++pass) {
    CKS(sqair_forward(h, flat, packed, obs, noise, T, B, 0, &out, ws, wsb, s));
    CKS(sqair_elbo(h, lw_t, dlp_t, T, B, log_w, elbo_b, iw, sig, scal, NULL, 0, means, s));
    CKH(hipStreamSynchronize(s));
    CKH(hipMemcpy(sc[pass], scal, 64, hipMemcpyDeviceToHost));
  }
  printf("elbo_vae %.4f  elbo_iwae %.4f  vimco target %.4f  ess %.3f\n", sc[0][0], sc[0][1], sc[0][2], sc[0][3]);
  if (!isfinite(sc[0][0]) || !isfinite(sc[0][1]) || !isfinite(sc[0][2])) { printf("non-finite objective\n"); return 4; }
  if (memcmp(sc[0], sc[1], 16) != 0) { printf("second pass differs from the first\n"); return 5; }
  /* gradient evaluation + one optimiser step */
  CKS(sqair_forward_train(h, flat, packed, obs, noise, T, B, 0, &out, tws, twb, s));
  CKS(sqair_elbo(h, lw_t, dlp_t, T, B, log_w, elbo_b, iw, sig, scal, NULL, 0, means, s));
  CKS(sqair_backward(h, flat, packed, obs, noise, iw, sig, T, B, 0, tws, twb, scratch, bwb, grad, s));
  CKS(sqair_rmsprop_step(h, flat, grad, ms, mom, np, 1e-4f, 0.9f, 0.9f, 1e-10f, 1.0f, s));
  CKS(sqair_pack_params(h, flat, packed, s));
  CKH(hipStreamSynchronize(s));
  float* hg = (float*)malloc((size_t)np * 4); float* hp2 = (float*)malloc((size_t)np * 4);
  CKH(hipMemcpy(hg, grad, (size_t)np * 4, hipMemcpyDeviceToHost)); CKH(hipMemcpy(hp2, flat, (size_t)np * 4, hipMemcpyDeviceToHost));
  double gmax = 0.0, moved = 0.0; int bad = 0;
  for (int64_t i = 0; i < np; ++i) { if (!isfinite(hg[i]) || !isfinite(hp2[i])) ++bad; if (fabs(hg[i]) > gmax) gmax = fabs(hg[i]); if (fabs(hp2[i] - hp[i]) > moved) moved = fabs(hp2[i] - hp[i]); }
  printf("max |grad| %.4g, largest parameter move %.3g, non-finite entries %d\n", gmax, moved, bad);
  if (bad != 0 || gmax <= 0.0 || moved <= 0.0) return 6;
  sqair_destroy(h);
  printf("abi_smoke OK\n");
  return 0;
}
