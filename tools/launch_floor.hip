// Microbenchmark: cost of a dependent kernel boundary on this MI355X / ROCm stack, in a HIP graph and eagerly.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor tools/launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Big { const float* p[8]; int a[40]; float* out; };

__global__ void k_empty() {}
__global__ void k_args(Big b) { if (b.a[3] == 12345) b.out[0] = 1.0f; }
__global__ void k_chain(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * 1.0001f + 1.0f;
}
// gathers: each thread reads 8 float4 from scattered rows of `in` (like a GEMM A/B fragment fetch)
__global__ void k_gather(const float4* __restrict__ in, float4* __restrict__ out, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 v = in[(i * 17 + j * 1031) % n4];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  out[i % n4] = acc;
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
template <class F>
double time_eager(hipStream_t s, int nodes, int reps, F launch) {
  for (int i = 0; i < nodes; ++i) launch(i);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int r = 0; r < reps; ++r) for (int i = 0; i < nodes; ++i) launch(i);
  hipStreamSynchronize(s);
  auto t1 = std::chrono::high_resolution_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps / nodes;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int n = 160 * 256;
  float *a, *b; CK(hipMalloc(&a, n * 16 * 4)); CK(hipMalloc(&b, n * 16 * 4));
  CK(hipMemset(a, 0, n * 16 * 4)); CK(hipMemset(b, 0, n * 16 * 4));
  Big big{}; big.out = a;
  const int NODES = 1000, REPS = 20;
  printf("per-node time [us], %d-node dependent chain (graph | eager)\n", NODES);
  printf("empty <<<1,64>>>            %.2f | %.2f\n", time_graph(s, NODES, REPS, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }),
         time_eager(s, NODES, REPS, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }));
  printf("empty <<<160,256>>>         %.2f | %.2f\n", time_graph(s, NODES, REPS, [&](int) { hipLaunchKernelGGL(k_empty, dim3(160), dim3(256), 0, s); }),
         time_eager(s, NODES, REPS, [&](int) { hipLaunchKernelGGL(k_empty, dim3(160), dim3(256), 0, s); }));
  printf("368B-arg <<<160,256>>>      %.2f | %.2f\n", time_graph(s, NODES, REPS, [&](int) { hipLaunchKernelGGL(k_args, dim3(160), dim3(256), 0, s, big); }),
         time_eager(s, NODES, REPS, [&](int) { hipLaunchKernelGGL(k_args, dim3(160), dim3(256), 0, s, big); }));
  printf("chain r/w <<<160,256>>>     %.2f | %.2f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_chain, dim3(160), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n); }),
         time_eager(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_chain, dim3(160), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n); }));
  printf("chain r/w <<<1,256>>>       %.2f | %.2f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 256); }),
         time_eager(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 256); }));
  printf("gather 8xfloat4 <<<160,256>>> %.2f | %.2f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_gather, dim3(160), dim3(256), 0, s, (const float4*)((i & 1) ? b : a), (float4*)((i & 1) ? a : b), n * 4); }),
         time_eager(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_gather, dim3(160), dim3(256), 0, s, (const float4*)((i & 1) ? b : a), (float4*)((i & 1) ? a : b), n * 4); }));
  printf("gather 8xfloat4 <<<32,256>>>  %.2f | %.2f\n", time_graph(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_gather, dim3(32), dim3(256), 0, s, (const float4*)((i & 1) ? b : a), (float4*)((i & 1) ? a : b), n * 4); }),
         time_eager(s, NODES, REPS, [&](int i) { hipLaunchKernelGGL(k_gather, dim3(32), dim3(256), 0, s, (const float4*)((i & 1) ? b : a), (float4*)((i & 1) ? a : b), n * 4); }));
  return 0;
}
