// Per-node time of the routed dX kernel next to the forward split-K kernel on the same 160 x 256 x 256 shape, back-to-back nodes of
// one graph (the dX epilogue reads its routing table from the by-value argument struct: does that cost anything?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isqair_amd/csrc -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/bin/dx_floor tools/dx_floor.hip
#include "sqair_common.h"
#include <chrono>
#include <cstdio>
#include <cstring>
void sq_set_error(SqairHandle*, const std::string&) {}
#include "sqair_linear_dx.hip"
#define SQ_KLINEAR_NAME k_fwd
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME

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
  float *x, *y, *w, *b, *sv;
  hipMalloc(&x, M * K * 4); hipMalloc(&y, M * N * 4); hipMalloc(&sv, M * N * 4); hipMalloc(&w, (256 + nt * kc * 256) * 4); hipMalloc(&b, N * 4);
  hipMemset(x, 0, M * K * 4); hipMemset(y, 0, M * N * 4); hipMemset(sv, 0, M * N * 4); hipMemset(w, 0, (256 + nt * kc * 256) * 4); hipMemset(b, 0, N * 4);
  const int NODES = 1000, REPS = 20;
  printf("%-44s %.2f us/node\n", "k_linear<4,1> (forward kernel), chain", time_graph(s, NODES, REPS, [&](int i) {
    LinArgs a = LinArgs();
    a.seg[0] = LinSeg{(i & 1) ? y : x, K, K, 1}; a.nseg = 1; a.wp = w + 256; a.wzero = w; a.bias = b; a.out = (i & 1) ? x : y; a.out_ld = N;
    a.M = M; a.N = N; a.epi = EPI_ACT; a.act_a = ACT_ELU; a.act_split = 1 << 30; a.scale = 1.0f; a.add_rdiv = 1;
    hipLaunchKernelGGL((k_fwd<4, 1, false, false>), dim3(nt, M / 16), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, kc, nt, a.wzero, a, (unsigned long long*)nullptr); }));
  for (int variant = 0; variant < 3; ++variant) {
    const char* nm[] = {"k_linear_dx<4>: 1 range, elu' from saved", "k_linear_dx<4>: 1 range, accumulate + saved", "k_linear_dx<4>: 3 ranges"};
    printf("%-44s %.2f us/node\n", nm[variant], time_graph(s, NODES, REPS, [&](int i) {
      DxArgs a; memset(&a, 0, sizeof(a));
      a.dpre = (i & 1) ? y : x; a.ld = K; a.width = K; a.wp = w + 256; a.wzero = w; a.M = M;
      float* out = (i & 1) ? x : y;
      if (variant < 2) {
        a.nranges = 1;
        a.r[0].n0 = 0; a.r[0].n1 = N; a.r[0].dst = out; a.r[0].dst_ld = N; a.r[0].saved = sv; a.r[0].saved_ld = N;
        a.r[0].act_a = ACT_ELU; a.r[0].act_b = ACT_ELU; a.r[0].act_split = 1 << 30;
        if (variant == 1) { a.r[0].add = out; a.r[0].add_ld = N; }
      } else {
        a.nranges = 3;
        for (int q = 0; q < 3; ++q) {
          a.r[q].n0 = q == 0 ? 0 : (q == 1 ? 64 : 128); a.r[q].n1 = q == 0 ? 56 : (q == 1 ? 128 : 256);
          a.r[q].dst = out + a.r[q].n0; a.r[q].dst_ld = N; a.r[q].act_split = 1 << 30;
        }
      }
      hipLaunchKernelGGL((k_linear_dx<4, 0>), dim3(nt, M / 16), dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); }));
  }
  return 0;
}
