// Does confining a dependent launch chain to ONE XCD (CU-masked stream) make the kernel boundary cheaper?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cumask_chain tools/cumask_chain.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_step(const float* __restrict__ in, float* __restrict__ out, int n, int* xcc_hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0 && xcc_hist != nullptr) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    atomicAdd(&xcc_hist[x & 0xf], 1);
  }
  if (i < n) out[i] = in[(i * 7 + 3) % n] * 1.0001f + 0.5f;   // reads what the previous launch wrote, scattered
}

static float run_chain(hipStream_t s, float* a, float* b, int n, int blocks, int nodes, int* hist) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n, i == 0 ? hist : nullptr);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / nodes;
}

int main() {
  const int n = 160 * 256, nodes = 2000;
  float *a, *b; int* hist;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&hist, 64)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipMemset(hist, 0, 64));
  printf("unmasked stream, 160 blocks : %.2f us/node\n", run_chain(s0, a, b, n, 160, nodes, hist));
  int h[16]; CK(hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost));
  printf("   xcc histogram:"); for (int i = 0; i < 8; ++i) printf(" %d", h[i]); printf("\n");
  for (int variant = 0; variant < 2; ++variant) {
    // variant 0: CU bit i -> XCD i % 8 (take bits 0, 8, 16, ...); variant 1: XCD = i / 32 (take bits 0..31)
    std::vector<uint32_t> mask(8, 0);
    for (int i = 0; i < 256; ++i) {
      const bool on = variant == 0 ? (i % 8 == 0) : (i < 32);
      if (on) mask[i / 32] |= 1u << (i % 32);
    }
    hipStream_t sm; CK(hipExtStreamCreateWithCUMask(&sm, 8, mask.data()));
    CK(hipMemset(hist, 0, 64));
    const float us = run_chain(sm, a, b, n, 160, nodes, hist);
    CK(hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost));
    printf("CU mask variant %d (32 CUs)   : %.2f us/node   xcc histogram:", variant, us);
    for (int i = 0; i < 8; ++i) printf(" %d", h[i]); printf("\n");
  }
  return 0;
}
