// Ablation microbenchmark of the dense-layer kernel in a 1000-node dependent HIP-graph chain (x -> y -> x ...),
// M=160, K=256, N=256: which part of the ~3.4 us in-kernel time is what?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sqair_amd/csrc -o tools/linear_floor tools/linear_floor.hip
#include "sqair_common.h"
#include <chrono>
#include <cstdio>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
void sq_set_error(SqairHandle*, const std::string&) {}

#define SQ_KLINEAR_NAME k_full
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#define SQ_ABL_NO_MFMA
#define SQ_KLINEAR_NAME k_nomfma
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#undef SQ_ABL_NO_MFMA
#define SQ_ABL_NO_A
#define SQ_KLINEAR_NAME k_noa
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#undef SQ_ABL_NO_A
#define SQ_ABL_NO_B
#define SQ_KLINEAR_NAME k_nob
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#define SQ_ABL_NO_A
#define SQ_ABL_NO_EPI
#define SQ_KLINEAR_NAME k_noload
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#undef SQ_ABL_NO_A
#undef SQ_ABL_NO_B
#define SQ_KLINEAR_NAME k_noepi
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#undef SQ_ABL_NO_EPI

#define SQ_ROWTILE_XCD_AFFINITY
#define SQ_KLINEAR_NAME k_aff
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#define SQ_ABL_NO_B
#define SQ_KLINEAR_NAME k_aff_nob
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#undef SQ_ABL_NO_B
#define SQ_PREFETCH_NEXT_W
#define SQ_KLINEAR_NAME k_aff_pf
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#undef SQ_PREFETCH_NEXT_W
#undef SQ_ROWTILE_XCD_AFFINITY
#define SQ_PREFETCH_NEXT_W
#define SQ_KLINEAR_NAME k_full_pf
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME
#undef SQ_PREFETCH_NEXT_W

template <class F>
double time_graph(hipStream_t s, int nodes, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < nodes; ++i) launch(i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t1 = std::chrono::high_resolution_clock::now();
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps / nodes;
}

int main() {
  hipStream_t s; hipStreamCreate(&s);
  const int M = 160, K = 256, N = 256, kc = K / 16, nt = N / 16;
  float *x, *y, *w, *b;
  hipMalloc(&x, 640 * K * 4); hipMalloc(&y, 640 * K * 4); hipMalloc(&w, (256 + 64 * nt * kc * 256) * 4); hipMalloc(&b, N * 4);
  hipMemset(x, 0, 640 * K * 4); hipMemset(y, 0, 640 * K * 4); hipMemset(w, 0, (256 + 64 * nt * kc * 256) * 4); hipMemset(b, 0, N * 4);
  auto mk = [&](const float* in, float* out, int m, int n_tiles_used) {
    LinArgs a = LinArgs();
    a.seg[0] = LinSeg{in, K, K, 1}; a.nseg = 1; a.wp = w + 256; a.wzero = w; a.bias = b; a.out = out; a.out_ld = N;
    a.M = m; a.N = n_tiles_used * 16; a.epi = EPI_ACT; a.act_a = ACT_ELU; a.act_split = 1 << 30; a.scale = 1.0f; a.add_rdiv = 1;
    return a;
  };
  const int NODES = 1000, REPS = 20;
#define RUN(name, kern, m, ntu, statin)                                                                         \
  printf("%-34s %.2f us/node\n", name, time_graph(s, NODES, REPS, [&](int i) {                                    \
    LinArgs a = mk((statin) ? x : ((i & 1) ? y : x), (statin) ? y : ((i & 1) ? x : y), m, ntu);                   \
    hipLaunchKernelGGL(kern<4>, dim3((ntu), ((m) + 15) / 16), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, kc, ntu, a.wzero, a, (unsigned long long*)nullptr); }));
  RUN("full 160x256x256 (160 WG)", k_full, 160, 16, false);
  RUN("full, static input", k_full, 160, 16, true);
  RUN("no MFMA (VALU fma instead)", k_nomfma, 160, 16, false);
  RUN("no A loads", k_noa, 160, 16, false);
  RUN("no B loads", k_nob, 160, 16, false);
  RUN("no epilogue-operand loads", k_noepi, 160, 16, false);
  RUN("no loads at all", k_noload, 160, 16, false);
  RUN("full, 10 WG (M=160, N=16)", k_full, 160, 1, false);
  RUN("full, 40 WG (M=160, N=64)", k_full, 160, 4, false);
  RUN("full, 16 WG (M=16, N=256)", k_full, 16, 16, false);
  RUN("full, 640 WG (M=640)", k_full, 640, 16, true);
  for (int L : {1, 16, 64}) {
    char nm[64]; snprintf(nm, 64, "row-tile XCD affinity, %d w mats", L);
    printf("%-34s %.2f us/node\n", nm, time_graph(s, NODES, REPS, [&](int i) {
      LinArgs a = mk((i & 1) ? y : x, (i & 1) ? x : y, 160, 16);
      a.wp = w + 256 + (size_t)(i % L) * nt * kc * 256;
      hipLaunchKernelGGL(k_aff<4>, dim3(256), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, kc, 16, a.wzero, a, (unsigned long long*)nullptr); }));
  }
  for (int L : {16, 64}) {
    char nm[64]; snprintf(nm, 64, "affinity + next-W prefetch, %d mats", L);
    printf("%-34s %.2f us/node\n", nm, time_graph(s, NODES, REPS, [&](int i) {
      LinArgs a = mk((i & 1) ? y : x, (i & 1) ? x : y, 160, 16);
      a.wp = w + 256 + (size_t)(i % L) * nt * kc * 256;
      a.e1 = w + 256 + (size_t)((i + 1) % L) * nt * kc * 256;
      hipLaunchKernelGGL(k_aff_pf<4>, dim3(256), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, kc, 16, a.wzero, a, (unsigned long long*)nullptr); }));
    snprintf(nm, 64, "n-tile map + next-W prefetch, %d mats", L);
    printf("%-34s %.2f us/node\n", nm, time_graph(s, NODES, REPS, [&](int i) {
      LinArgs a = mk((i & 1) ? y : x, (i & 1) ? x : y, 160, 16);
      a.wp = w + 256 + (size_t)(i % L) * nt * kc * 256;
      a.e1 = w + 256 + (size_t)((i + 1) % L) * nt * kc * 256;
      hipLaunchKernelGGL(k_full_pf<4>, dim3(16, 10), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, kc, 16, a.wzero, a, (unsigned long long*)nullptr); }));
  }
  printf("%-34s %.2f us/node\n", "row-tile XCD affinity, no B loads", time_graph(s, NODES, REPS, [&](int i) {
      LinArgs a = mk((i & 1) ? y : x, (i & 1) ? x : y, 160, 16);
      hipLaunchKernelGGL(k_aff_nob<4>, dim3(256), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, kc, 16, a.wzero, a, (unsigned long long*)nullptr); }));
  // cycle through L different 256 KB weight matrices (the forward pass touches ~12 MB of weights per frame)
  for (int L : {2, 8, 16, 32, 48, 64}) {
    char nm[64]; snprintf(nm, 64, "full, %d rotating weight mats", L);
    printf("%-34s %.2f us/node\n", nm, time_graph(s, NODES, REPS, [&](int i) {
      LinArgs a = mk((i & 1) ? y : x, (i & 1) ? x : y, 160, 16);
      a.wp = w + 256 + (size_t)(i % L) * nt * kc * 256;
      hipLaunchKernelGGL(k_full<4>, dim3(16, 10), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, kc, 16, a.wzero, a, (unsigned long long*)nullptr); }));
  }
  return 0;
}
