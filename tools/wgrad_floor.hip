// Ablation microbenchmark of the weight-gradient kernel (k_wgrad3): dW[K x N] += A[M x K]^T dY[M x N], back-to-back launches
// on one stream, HIP-event timed.  Which part of a launch is the MFMAs, the operand stream, the atomics?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isqair_amd/csrc -o tools/bin/wgrad_floor tools/wgrad_floor.hip
//   tools/bin/wgrad_floor [M=6400] [K=256] [N=256] [wgs=512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sqair_common.h"   // SQ_TLP / SQ_TL_SCOPE (nothing in the product build)
typedef float f32x4_b __attribute__((ext_vector_type(4)));
// the stamped variant: every wave writes the device wall clock (100 MHz) at five points of its life
__device__ unsigned long long* g_stamps;
#define SQ_ABL_WG_STAMP(i) if ((threadIdx.x & 63) == 0) g_stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64();
#define SQ_KWGRAD_NAME k_stamped
#define SQ_KWGRAD_BODY b_stamped
#include "sqair_wgrad_kernel.inc"
#undef SQ_KWGRAD_NAME
#undef SQ_KWGRAD_BODY
#undef SQ_ABL_WG_STAMP
#define SQ_ABL_WG_STAMP(i)
#define SQ_KWGRAD_NAME k_full
#define SQ_KWGRAD_BODY b_full
#include "sqair_wgrad_kernel.inc"
#undef SQ_KWGRAD_NAME
#define SQ_ABL_WG_NO_ATOMIC
#undef SQ_KWGRAD_BODY
#define SQ_KWGRAD_BODY b_noatomic
#define SQ_KWGRAD_NAME k_noatomic
#include "sqair_wgrad_kernel.inc"
#undef SQ_KWGRAD_NAME
#define SQ_ABL_WG_NO_MFMA
#undef SQ_KWGRAD_BODY
#define SQ_KWGRAD_BODY b_noatomic_nomfma
#define SQ_KWGRAD_NAME k_noatomic_nomfma
#include "sqair_wgrad_kernel.inc"
#undef SQ_KWGRAD_NAME
#undef SQ_ABL_WG_NO_ATOMIC
#undef SQ_KWGRAD_BODY
#define SQ_KWGRAD_BODY b_nomfma
#define SQ_KWGRAD_NAME k_nomfma
#include "sqair_wgrad_kernel.inc"
#undef SQ_KWGRAD_NAME
#undef SQ_ABL_WG_NO_MFMA
#define SQ_ABL_WG_NO_LOAD
#undef SQ_KWGRAD_BODY
#define SQ_KWGRAD_BODY b_noload
#define SQ_KWGRAD_NAME k_noload
#include "sqair_wgrad_kernel.inc"
#undef SQ_KWGRAD_NAME
#undef SQ_ABL_WG_NO_LOAD

// the matrix cores alone: the kernel's 16 accumulators, `iters` x 64 dependent-free MFMAs per wave, nothing else
__global__ __launch_bounds__(256) void k_mfma_only(float* out, int iters) {
  f32x4_b acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = {0.0f, 0.0f, 0.0f, 0.0f};
  float a = (float)threadIdx.x, b = 1.0f / (1.0f + threadIdx.x);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float v = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) v += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (v == 123.456f) out[threadIdx.x] = v;
}

typedef void (*kern_t)(const float*, int, const float*, int, float*, int, int, int, int, const int*, const float*, float*, float*, int,
                       int, int);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 6400, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 256;
  const int wgs = argc > 4 ? atoi(argv[4]) : 512;
  const int NBUF = 8;  // rotate operand buffers so that a launch does not find its rows in L2 / MALL from the previous one
  float *A, *dY, *dW, *db;
  CK(hipMalloc(&A, (size_t)NBUF * M * K * 4)); CK(hipMalloc(&dY, (size_t)NBUF * M * N * 4)); CK(hipMalloc(&dW, (size_t)K * N * 4));
  CK(hipMalloc(&db, N * 4));
  CK(hipMemset(A, 0, (size_t)NBUF * M * K * 4)); CK(hipMemset(dY, 0, (size_t)NBUF * M * N * 4)); CK(hipMemset(dW, 0, (size_t)K * N * 4));
  CK(hipMemset(db, 0, N * 4));
  const int kt = (K + 63) / 64, nt = (N + 63) / 64, n_tiles = kt * nt;
  int zc = wgs / n_tiles; const int max_z = (M + 63) / 64;
  if (zc > max_z) zc = max_z; if (zc < 1) zc = 1;
  const int m_per_wg = ((M + zc - 1) / zc + 63) / 64 * 64;
  zc = (M + m_per_wg - 1) / m_per_wg;
  const int grid = n_tiles * zc;
  const size_t shm = (4 * 4096 + 256) * 4;
  struct V { const char* name; kern_t k; } vs[] = {{"full", k_full}, {"no atomics", k_noatomic}, {"no mfma", k_nomfma}, {"no atomics, no mfma", k_noatomic_nomfma}, {"no loads in the loop", k_noload}};
  printf("M %d K %d N %d: %d tiles x %d chunks of %d rows = %d workgroups (grid %d), %.2f GFLOP, operands %.1f MB\n", M, K, N, n_tiles, zc,
         m_per_wg, n_tiles * zc, grid, 2.0 * M * K * N * 1e-9, (double)M * (K + N) * 4e-6);
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
    const int iters = 400, reps = 20;
    for (int it = -3; it < reps; ++it) {
      if (it == 0) CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_mfma_only, dim3(256 * wg_per_cu), dim3(256), 0, s, dW, iters);
    }
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 256.0 * wg_per_cu * 4 * iters * 64 * 2048.0;
    printf("  MFMA only, %d workgroup(s) per CU: %.1f TFLOP/s\n", wg_per_cu, fl * reps / (ms * 1e-3) * 1e-12);
  }
  for (auto& v : vs) {
    CK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int rot = 0; rot < 2; ++rot) {
      const int reps = 200;
      for (int it = -20; it < reps; ++it) {
        if (it == 0) CK(hipEventRecord(e0, s));
        const int bsel = rot ? ((it + 20) % NBUF) : 0;
        hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), shm, s, A + (size_t)bsel * M * K, K, dY + (size_t)bsel * M * N, N, dW, N, M, K, N, nullptr, nullptr, db,
                           nullptr, m_per_wg, kt, n_tiles);
      }
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("  %-22s %s operands: %7.2f us / launch  (%.1f TFLOP/s)\n", v.name, rot ? "rotating" : "same    ", us, 2.0 * M * K * N / us * 1e-6);
    }
  }
  {  // where a wave's time goes: one stamped launch (after a warm one)
    unsigned long long* st; CK(hipMalloc(&st, (size_t)grid * 4 * 8 * 8)); CK(hipMemset(st, 0, (size_t)grid * 4 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &st, sizeof(st)));
    CK(hipFuncSetAttribute((const void*)k_stamped, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    for (int it = 0; it < 2; ++it)
      hipLaunchKernelGGL(k_stamped, dim3(grid), dim3(256), shm, s, A, K, dY, N, dW, N, M, K, N, nullptr, nullptr, db, nullptr, m_per_wg, kt, n_tiles);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)grid * 32);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    double seg[4] = {0, 0, 0, 0};
    for (int w = 0; w < grid * 4; ++w) {
      const unsigned long long* p = &h[(size_t)w * 8];
      if (p[0] < t0) t0 = p[0];
      if (p[4] > t1) t1 = p[4];
      for (int i = 0; i < 4; ++i) seg[i] += (double)(p[i + 1] - p[i]);
    }
    const double nw = grid * 4.0, tick = 0.01;
    printf("  stamped launch: span %.1f us; mean wave: entry->loads issued %.2f, loop %.2f, tile to LDS + barrier %.2f, sums + atomics %.2f us\n",
           (t1 - t0) * tick, seg[0] / nw * tick, seg[1] / nw * tick, seg[2] / nw * tick, seg[3] / nw * tick);
    // start times of the waves in dispatch order: when does round 2 begin?
    std::vector<double> starts;
    for (int w = 0; w < grid * 4; w += 4) starts.push_back((h[(size_t)w * 8] - t0) * tick);
    printf("  workgroup start (us after the first) at workgroup 0, 1/4, 1/2, 3/4, last: %.1f %.1f %.1f %.1f %.1f\n", starts[0], starts[grid / 4],
           starts[grid / 2], starts[3 * grid / 4], starts[grid - 1]);
  }
  return 0;
}
