// Premise check for a "layer chain" kernel: ONE workgroup per 16-row tile runs L dependent dense layers (256 -> 256,
// ELU) back to back, activations in LDS, every wave streaming its share of each layer's packed weights from L2 --
// against L launches of the production dense kernel.  Rows are independent through a slot's MLP chains, so no
// inter-workgroup synchronisation is needed; the price is that only M/16 = 10 CUs work and each streams ALL the weights
// (256 KB per layer through a 64 B/clk L1 fill path ~ 1.7 us).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isqair_amd/csrc -mllvm -amdgpu-kernarg-preload-count=16 -o tools/bin/chain_floor tools/chain_floor.hip
#include "sqair_common.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
void sq_set_error(SqairHandle*, const std::string&) {}

#define SQ_KLINEAR_NAME k_full
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME

constexpr int KD = 256, ND = 256, KC = KD / 16, NT = ND / 16, ALD = KD + 4;

// WAVES waves per workgroup; wave w owns n-tiles w, w + WAVES, ...  PF: issue the next layer's weight loads before
// this layer's MFMAs (weights do not depend on the activations).
template <int WAVES, bool PF>
__global__ __launch_bounds__(WAVES * 64) void k_chain(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ w, const int L, const int M) {
  constexpr int TPW = NT / WAVES;  // n-tiles per wave
  __shared__ float act[2][16 * ALD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, tile_m = blockIdx.x;
  for (int e = tid; e < 16 * KD / 4; e += WAVES * 64) {
    const int r = e / (KD / 4), c4 = e % (KD / 4);
    const int row = min(tile_m * 16 + r, M - 1);
    *reinterpret_cast<f32x4*>(&act[0][r * ALD + c4 * 4]) = *reinterpret_cast<const f32x4*>(x + (size_t)row * KD + c4 * 4);
  }
  f32x4 bv[TPW][KC];
  auto load_w = [&](int l) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(w + (size_t)l * NT * KC * 256) + lane;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int c = 0; c < KC; ++c) bv[t][c] = wp[((size_t)(wave + t * WAVES) * KC + c) * 64];
  };
  load_w(0);
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    const float* a_in = act[l & 1];
    float* a_out = act[(l & 1) ^ 1];
    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(a_in + (lane & 15) * ALD + c * 16 + kq * 4);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[t][c].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[t][c].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[t][c].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[t][c].w, acc[t], 0, 0, 0);
      }
    }
    if (PF && l + 1 < L) load_w(l + 1);   // registers are free again: the loads fly during the epilogue + barrier
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int n = (wave + t * WAVES) * 16 + (lane & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[t][i];
        v = v > 0.f ? v : expm1f(v);
        a_out[(4 * kq + i) * ALD + n] = v;
      }
    }
    __syncthreads();
    if (!PF && l + 1 < L) load_w(l + 1);
  }
  const float* a_fin = act[L & 1];
  for (int e = tid; e < 16 * ND / 4; e += WAVES * 64) {
    const int r = e / (ND / 4), c4 = e % (ND / 4);
    const int row = tile_m * 16 + r;
    if (row < M) *reinterpret_cast<f32x4*>(y + (size_t)row * ND + c4 * 4) = *reinterpret_cast<const f32x4*>(&a_fin[r * ALD + c4 * 4]);
  }
}

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
  const int M = 160, NL = 48;  // rotating weight matrices (12 MB: the forward pass's per-frame weight footprint)
  float *x, *y, *w, *b, *yref;
  const size_t wl = (size_t)NT * KC * 256;
  hipMalloc(&x, M * KD * 4); hipMalloc(&y, M * KD * 4); hipMalloc(&yref, M * KD * 4); hipMalloc(&w, (256 + NL * wl) * 4); hipMalloc(&b, ND * 4);
  std::vector<float> hx(M * KD), hw(256 + NL * wl, 0.f);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  for (size_t i = 256; i < hw.size(); ++i) hw[i] = ((float)((i * 40503u) % 2001) / 1000.f - 1.0f) * 0.08f;
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemset(b, 0, ND * 4);
  auto lin = [&](const float* in, float* out, int l) {
    LinArgs a = LinArgs();
    a.seg[0] = LinSeg{in, KD, KD, 1}; a.nseg = 1; a.wp = w + 256 + (size_t)l * wl; a.wzero = w; a.bias = b; a.out = out; a.out_ld = ND;
    a.M = M; a.N = ND; a.epi = EPI_ACT; a.act_a = ACT_ELU; a.act_split = 1 << 30; a.scale = 1.0f; a.add_rdiv = 1;
    hipLaunchKernelGGL(k_full<4>, dim3(NT, (M + 15) / 16), dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M,
                       KC, NT, a.wzero, a, (unsigned long long*)nullptr);
  };
  // correctness: 3 chained layers
  lin(x, y, 0); lin(y, yref, 1); lin(yref, y, 2); hipMemcpyAsync(yref, y, M * KD * 4, hipMemcpyDeviceToDevice, s);
  hipLaunchKernelGGL((k_chain<16, false>), dim3((M + 15) / 16), dim3(1024), 0, s, x, y, w + 256, 3, M);
  hipStreamSynchronize(s);
  std::vector<float> h1(M * KD), h2(M * KD);
  hipMemcpy(h1.data(), y, h1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), yref, h2.size() * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (size_t i = 0; i < h1.size(); ++i) { md = std::max(md, (double)std::fabs(h1[i] - h2[i])); mx = std::max(mx, (double)std::fabs(h2[i])); }
  printf("chain vs 3 launches: max |diff| %.3e (max |ref| %.3e) %s\n", md, mx, hipGetLastError() == hipSuccess ? "" : "HIP ERROR");

  const int NODES = 600, REPS = 20;
  for (int L : {1, 2, 3, 4}) {
    printf("L=%d layers per step\n", L);
    printf("  %-44s %.2f us/step\n", "production kernel, L launches", L * time_graph(s, NODES * L, REPS, [&](int i) {
      lin((i & 1) ? y : x, (i & 1) ? x : y, i % NL); }));
#define RUNC(name, W, PF)                                                                                          \
    printf("  %-44s %.2f us/step\n", name, time_graph(s, NODES, REPS, [&](int i) {                                   \
      hipLaunchKernelGGL((k_chain<W, PF>), dim3((M + 15) / 16), dim3(W * 64), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, \
                         w + 256 + (size_t)((i * L) % (NL - 4)) * wl, L, M); }));
    RUNC("chain, 16 waves (1 n-tile each)", 16, false);
    RUNC("chain, 16 waves, next-layer weight prefetch", 16, true);
    RUNC("chain, 8 waves (2 n-tiles each)", 8, false);
    RUNC("chain, 8 waves, next-layer weight prefetch", 8, true);
    RUNC("chain, 4 waves (4 n-tiles each)", 4, false);
    RUNC("chain, 4 waves, next-layer weight prefetch", 4, true);
  }
  return 0;
}
