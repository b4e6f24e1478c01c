// Microbenchmark: what makes the dependent-launch gap of some kernels larger (1.3-1.6 us ahead of k_crop_row, k_crop_chain_bwd,
// k_slot_tail_bwd against 1.1 us ahead of k_linear in the step's timeline)?  A 1000-node dependent chain of
// the same trivial read-modify-write kernel, varied in ONE launch property at a time: LDS (static / dynamic, 10 / 40 KB),
// VGPR allocation, size of the argument block.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gap_floor tools/gap_floor.hip && tools/bin/gap_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Big { int a[240]; };   // 960 bytes

__device__ __forceinline__ void body(const float* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * 1.0001f + 1.0f;
}
__global__ void k_base(const float* __restrict__ in, float* __restrict__ out, int n) { body(in, out, n); }
template <int KB>
__global__ void k_static(const float* __restrict__ in, float* __restrict__ out, int n) {
  __shared__ float s[KB * 256];
  s[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  body(in, out, n);
  if (s[(threadIdx.x + 1) & 255] == 12345.0f) out[0] = 0.0f;
}
__global__ void k_dynamic(const float* __restrict__ in, float* __restrict__ out, int n) {
  extern __shared__ float sd[];
  sd[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  body(in, out, n);
  if (sd[(threadIdx.x + 1) & 255] == 12345.0f) out[0] = 0.0f;
}
__global__ void k_vgpr128(const float* __restrict__ in, float* __restrict__ out, int n) {
  asm volatile("" ::: "v127");
  body(in, out, n);
}
__global__ void k_vgpr250(const float* __restrict__ in, float* __restrict__ out, int n) {
  asm volatile("" ::: "v250");
  body(in, out, n);
}
__global__ void k_sgpr100(const float* __restrict__ in, float* __restrict__ out, int n) {
  asm volatile("" ::: "s99");
  body(in, out, n);
}
__global__ void k_bigargs(const float* __restrict__ in, float* __restrict__ out, int n, Big b) {
  body(in, out, n);
  if (b.a[200] == 12345) out[0] = 0.0f;
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
  hipStream_t s; CK(hipStreamCreate(&s));
  const int n = 160 * 256;
  float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  CK(hipFuncSetAttribute((const void*)k_dynamic, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  Big big{};
  const int NODES = 1000, REPS = 20;
  const dim3 G(160), B(256);
#define SRC ((i & 1) ? b : a)
#define DST ((i & 1) ? a : b)
  for (int pass = 0; pass < 2; ++pass) {
    printf("pass %d: per-node time [us], %d-node dependent graph chain, <<<160, 256>>>\n", pass, NODES);
    printf("  base                   %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_base, G, B, 0, s, SRC, DST, n); }));
    printf("  static LDS 10 KB       %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_static<10>, G, B, 0, s, SRC, DST, n); }));
    printf("  static LDS 40 KB       %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_static<40>, G, B, 0, s, SRC, DST, n); }));
    printf("  dynamic LDS 1 KB       %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_dynamic, G, B, 1024, s, SRC, DST, n); }));
    printf("  dynamic LDS 10 KB      %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_dynamic, G, B, 10240, s, SRC, DST, n); }));
    printf("  dynamic LDS 40 KB      %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_dynamic, G, B, 40960, s, SRC, DST, n); }));
    printf("  dynamic LDS 100 KB     %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_dynamic, G, B, 102400, s, SRC, DST, n); }));
    printf("  128 VGPRs              %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_vgpr128, G, B, 0, s, SRC, DST, n); }));
    printf("  251 VGPRs              %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_vgpr250, G, B, 0, s, SRC, DST, n); }));
    printf("  100 SGPRs              %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_sgpr100, G, B, 0, s, SRC, DST, n); }));
    printf("  + 960 B of arguments   %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_bigargs, G, B, 0, s, SRC, DST, n, big); }));
    printf("  base                   %.3f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_base, G, B, 0, s, SRC, DST, n); }));
  }
  return 0;
}
