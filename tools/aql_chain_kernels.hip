// Device side of tools/aql_chain.cpp (a code object loaded through the HSA runtime, not through HIP):
//   hipcc --offload-arch=gfx950 -O3 --genco -o tools/bin/aql_chain.hsaco tools/aql_chain_kernels.hip
// The dependent 160 x 256 x 256 layer of tools/sentinel_chain.hip (16 x 16 output tile per workgroup, four waves on the K range).
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int KD = 256, ND = 256, KC = KD / 16;
constexpr unsigned SENT = 0xFFFFFFFFu;

__device__ __forceinline__ void tile_mfma(const f32x4 (&av)[4], const f32x4 (&bv)[4], float* red, float* Y, bool sys_store, int nt, int rt) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc, 0, 0, 0);
  }
  float* r = red + wave * 256;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[(4 * kq + i) * 16 + (lane & 15)] = acc[i];
  __syncthreads();
  const float v = tanhf(red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid]);
  float* o = Y + (size_t)(rt * 16 + (tid >> 4)) * ND + nt * 16 + (tid & 15);
  if (sys_store) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(o), "v"(v) : "memory");   // written through: visible to every XCD
  else *o = v;
}

// one launch per layer, ordered by the queue (barrier bit + fences): plain loads and stores
extern "C" __global__ __launch_bounds__(256) void k_layer(const float* __restrict__ X, const float* __restrict__ Wp, float* __restrict__ Y) {
  __shared__ float red[1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const float* rp = X + (size_t)(blockIdx.y * 16 + (lane & 15)) * KD;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((size_t)blockIdx.x * KC) * 64 + lane;
  f32x4 av[4], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = wave + 4 * j;
    av[j] = *reinterpret_cast<const f32x4*>(rp + g * 16 + kq * 4);
    bv[j] = wp[(size_t)g * 64];
  }
  __builtin_amdgcn_sched_barrier(0);
  tile_mfma(av, bv, red, Y, false, blockIdx.x, blockIdx.y);
}

// the same layer ordered by its DATA: X was pre-filled with a sentinel word and is written (through) by the previous layer's
// launch, which may still be running -- the packet carries no barrier bit.  The A-operand loads are system-scope (no stale line of
// this XCD's L2 can satisfy them) and are repeated until no word is the sentinel.  `status`: set when a workgroup gives up.
extern "C" __global__ __launch_bounds__(256) void k_layer_poll(const float* __restrict__ X, const float* __restrict__ Wp, float* __restrict__ Y,
                                                               int* __restrict__ status, int spin_limit, int sleep) {
  __shared__ float red[1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const float* rp = X + (size_t)(blockIdx.y * 16 + (lane & 15)) * KD;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((size_t)blockIdx.x * KC) * 64 + lane;
  f32x4 av[4], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = wp[(size_t)(wave + 4 * j) * 64];
  int spins = 0;
  bool bad;
  do {
    u32x4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* p = rp + (wave + 4 * j) * 16 + kq * 4;
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(a[j]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bad = bad || a[j].x == SENT || a[j].y == SENT || a[j].z == SENT || a[j].w == SENT;
      av[j] = __builtin_bit_cast(f32x4, a[j]);
    }
    bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;   // (the wave moves on together)
    if (bad && sleep > 0) __builtin_amdgcn_s_sleep(8);
  } while (bad && ++spins < spin_limit);
  if (bad && lane == 0) atomicExch(status, 1);   // (gave up: the result is garbage and says so; no wave leaves its workgroup waiting)
  __builtin_amdgcn_sched_barrier(0);
  tile_mfma(av, bv, red, Y, true, blockIdx.x, blockIdx.y);
}

extern "C" __global__ __launch_bounds__(256) void k_fill(unsigned* __restrict__ p, unsigned v, long n, long stride) {   // (stride = threads of the launch)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p + i), "v"(v) : "memory");
}

// (e) the hand-off inside ONE XCD's L2: a 1-D grid whose workgroup b computes row tile (b & 7) + 8 * ((b >> 3) / 16), column tile
// (b >> 3) % 16 -- under the round-robin placement of workgroups on XCDs (observed, not promised) every row tile's producers and
// consumers then share an XCD, and the polls are L2-level loads (sc1: past the L1 only).  Placement is a SPEED assumption here, not a
// correctness one: the stores are written through, and after `fast_polls` misses the polls go to system scope.
// `xcd_hits`: counters of workgroups, see below.
extern "C" __global__ __launch_bounds__(256) void k_layer_poll_xcd(const float* __restrict__ X, const float* __restrict__ Wp, float* __restrict__ Y,
                                                                   int* __restrict__ status, int spin_limit, int fast_polls, int m_tiles,
                                                                   int* __restrict__ xcd_hits) {
  __shared__ float red[1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int b = blockIdx.x, q = b >> 3;
  const int nt = q & 15, rt = (b & 7) + 8 * (q >> 4);
  if (rt >= m_tiles) return;
  const float* rp = X + (size_t)(rt * 16 + (lane & 15)) * KD;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((size_t)nt * KC) * 64 + lane;
  f32x4 av[4], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = wp[(size_t)(wave + 4 * j) * 64];
  int spins = 0;
  bool bad;
  do {
    u32x4 a[4];
    if (spins < fast_polls) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(a[j]) : "v"(rp + (wave + 4 * j) * 16 + kq * 4) : "memory");
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(a[j]) : "v"(rp + (wave + 4 * j) * 16 + kq * 4) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bad = bad || a[j].x == SENT || a[j].y == SENT || a[j].z == SENT || a[j].w == SENT;
      av[j] = __builtin_bit_cast(f32x4, a[j]);
    }
    bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
  } while (bad && ++spins < spin_limit);
  if (bad && lane == 0) atomicExch(status, 1);
  if (tid == 0 && xcd_hits != nullptr) {   // [0] operand there at the first poll (the producer had finished), [1] later, [2] by a system-scope poll
    atomicAdd(xcd_hits + (spins == 0 ? 0 : 1), 1);
    if (spins >= fast_polls) atomicAdd(xcd_hits + 2, 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  tile_mfma(av, bv, red, Y, true, nt, rt);
}

// (f) do packets without the barrier bit overlap at all?  A kernel that only waits `ticks` of the 100 MHz wall clock.
extern "C" __global__ __launch_bounds__(64) void k_delay(int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(4);
}
