// Can heavy independent work hide behind a latency-bound dependent chain inside ONE hipGraph (fork/join capture)?
// chain: 2000 dependent small kernels (160 blocks); side branch: 40 independent "heavy" kernels (2048 blocks, ~25 us each).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/graph_branch tools/graph_branch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_step(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(i * 7 + 3) % n] * 1.0001f + 0.5f;
}
__global__ void k_heavy(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
  for (int r = 0; r < 64; ++r) acc += in[(i + r * 4099) % n];
  out[i % n] = acc;
}
static float time_graph(hipGraphExec_t ge, hipStream_t s) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}
int main() {
  const int n = 160 * 256, nbig = 8 << 20, nodes = 2000, heavy = 40;
  float *a, *b, *c, *d;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, nbig * 4)); CK(hipMalloc(&d, nbig * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, nbig * 4)); CK(hipMemset(d, 0, nbig * 4));
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t g_chain, g_heavy, g_serial, g_par;
  // chain only
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_step, dim3(160), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&g_chain, g, nullptr, nullptr, 0));
  // heavy only
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < heavy; ++i) hipLaunchKernelGGL(k_heavy, dim3(2048), dim3(256), 0, s, c, d, nbig);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&g_heavy, g, nullptr, nullptr, 0));
  // serial: chain then heavy
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_step, dim3(160), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n);
  for (int i = 0; i < heavy; ++i) hipLaunchKernelGGL(k_heavy, dim3(2048), dim3(256), 0, s, c, d, nbig);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&g_serial, g, nullptr, nullptr, 0));
  // parallel: fork heavy onto a second captured stream, join at the end
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s2, fork, 0));
  for (int i = 0; i < heavy; ++i) hipLaunchKernelGGL(k_heavy, dim3(2048), dim3(256), 0, s2, c, d, nbig);
  for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_step, dim3(160), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n);
  CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s, join, 0));
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&g_par, g, nullptr, nullptr, 0));
  // two SYMMETRIC dependent chains (half the rows each) as two branches of one graph, and as two graphs on two streams, against
  // one chain of full-size nodes: does splitting the batch into concurrent half-batches shorten the pass?
  {
    float *a2, *b2; CK(hipMalloc(&a2, n * 4)); CK(hipMalloc(&b2, n * 4)); CK(hipMemset(a2, 0, n * 4)); CK(hipMemset(b2, 0, n * 4));
    hipGraphExec_t g_two, g_half_a, g_half_b;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s2, fork, 0));
    for (int i = 0; i < nodes; ++i) {
      hipLaunchKernelGGL(k_step, dim3(80), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n / 2);
      hipLaunchKernelGGL(k_step, dim3(80), dim3(256), 0, s2, (i & 1) ? b2 : a2, (i & 1) ? a2 : b2, n / 2);
    }
    CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s, join, 0));
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&g_two, g, nullptr, nullptr, 0));
    printf("two half-size chains as branches of one graph: %.3f ms (%.2f us per node pair)\n", time_graph(g_two, s), time_graph(g_two, s) * 1e3 / nodes);
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_step, dim3(80), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n / 2);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&g_half_a, g, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_step, dim3(80), dim3(256), 0, s2, (i & 1) ? b2 : a2, (i & 1) ? a2 : b2, n / 2);
    CK(hipStreamEndCapture(s2, &g)); CK(hipGraphInstantiate(&g_half_b, g, nullptr, nullptr, 0));
    printf("one half-size chain alone                    : %.3f ms\n", time_graph(g_half_a, s));
    CK(hipGraphLaunch(g_half_a, s)); CK(hipGraphLaunch(g_half_b, s2)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    CK(hipEventRecord(e0, s)); CK(hipStreamWaitEvent(s2, e0, 0));
    CK(hipGraphLaunch(g_half_a, s)); CK(hipGraphLaunch(g_half_b, s2));
    CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s, e2, 0)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("two half-size chains as two graphs on two streams: %.3f ms\n", ms);
  }
  printf("chain only   : %.3f ms (%.2f us/node)\n", time_graph(g_chain, s), time_graph(g_chain, s) * 1e3 / nodes);
  printf("heavy only   : %.3f ms (%.1f us each)\n", time_graph(g_heavy, s), time_graph(g_heavy, s) * 1e3 / heavy);
  printf("serial       : %.3f ms\n", time_graph(g_serial, s));
  printf("fork / join  : %.3f ms\n", time_graph(g_par, s));
  return 0;
}
