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
// (file header above is linear_floor.hip's set of ablated instantiations; this main times the once-per-frame shape)
// Ablations of the split-K dense kernel on the mid-size once-per-frame layers of the pass (PRE: 640 x 384 x 1152, PRIOR_GRU1:
// 640 x 320 x 768), launches back to back in one graph: which part of the ~35 us / ~25 us in-kernel time is what?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isqair_amd/csrc -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/bin/mid_floor tools/mid_floor.hip
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

int main(int argc, char** argv) {
  hipStream_t s; hipStreamCreate(&s);
  const int M = argc > 1 ? atoi(argv[1]) : 640, K = argc > 2 ? atoi(argv[2]) : 384, N = argc > 3 ? atoi(argv[3]) : 1152;
  const int kc = K / 16, nt = N / 16;
  float *x, *y, *w, *b;
  hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&y, (size_t)M * N * 4); hipMalloc(&w, (256 + (size_t)nt * kc * 256) * 4); hipMalloc(&b, N * 4);
  hipMemset(x, 0, (size_t)M * K * 4); hipMemset(y, 0, (size_t)M * N * 4); hipMemset(w, 0, (256 + (size_t)nt * kc * 256) * 4); hipMemset(b, 0, N * 4);
  auto mk = [&](int m, int n_tiles_used) {
    LinArgs a = LinArgs();
    a.seg[0] = LinSeg{x, K, K, 1}; a.nseg = 1; a.wp = w + 256; a.wzero = w; a.bias = b; a.out = y; a.out_ld = N;
    a.M = m; a.N = n_tiles_used * 16; a.epi = EPI_ACT; a.act_a = ACT_NONE; a.act_split = 1 << 30; a.scale = 1.0f; a.add_rdiv = 1;
    return a;
  };
  const int NODES = 200, REPS = 10;
  printf("M %d K %d N %d: %d workgroups, %.2f GFLOP, operand bytes through L1 %.1f MB\n", M, K, N, nt * ((M + 15) / 16), 2.0 * M * K * N * 1e-9,
         (double)nt * ((M + 15) / 16) * 2 * 16 * K * 4e-6);
#define RUN(name, kern, NCHV, m, ntu)                                                                             \
  printf("%-34s %.2f us/node\n", name, time_graph(s, NODES, REPS, [&](int) {                                       \
    LinArgs a = mk(m, ntu);                                                                                        \
    hipLaunchKernelGGL((kern<NCHV, 1, false, false>), dim3((ntu), ((m) + 15) / 16), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, kc, ntu, a.wzero, a, (unsigned long long*)nullptr); }));
#define ALL(NCHV)                                                   \
  RUN("full", k_full, NCHV, M, nt)                                  \
  RUN("no MFMA (VALU fma instead)", k_nomfma, NCHV, M, nt)          \
  RUN("no A loads", k_noa, NCHV, M, nt)                             \
  RUN("no B loads", k_nob, NCHV, M, nt)                             \
  RUN("no epilogue-operand loads", k_noepi, NCHV, M, nt)            \
  RUN("no loads at all", k_noload, NCHV, M, nt)                     \
  RUN("full, half the rows", k_full, NCHV, M / 2, nt)               \
  RUN("full, half the columns", k_full, NCHV, M, nt / 2)            \
  RUN("full, a quarter of both", k_full, NCHV, M / 4, nt / 4)
  const int per_wave = (kc + 3) / 4;
  if (per_wave <= 4) { ALL(4) } else if (per_wave <= 5) { ALL(5) } else if (per_wave <= 6) { ALL(6) } else { ALL(10) }
  return 0;
}
