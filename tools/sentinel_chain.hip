// Microbenchmark (round 5, stage 0 of the in-launch slot chain): the dependent 160 x 256 x 256 layer chain of
// tools/granule_chain.hip with a hand-off that needs NO layout change and NO placement assumption:
//   * activations stay plain row-major fp32; every hand-off buffer is pre-filled with a SENTINEL word (0xFFFFFFFF, a NaN
//     payload no arithmetic produces); the consumer polls its own A-operand loads (16-byte sc1 loads = L1 bypass, served by
//     the XCD's L2) until no word is the sentinel -- the data is its own flag, word by word (4-byte stores are atomic, so
//     tearing of a 16-byte load is harmless);
//   * teams are formed by what the hardware did, not by what the launch hoped for: every workgroup reads HW_REG_XCC_ID and
//     pulls items (hop, row tile, column tile) from THAT XCD's in-order ticket queue.  Row tiles are bound to XCDs, so a row
//     tile's producers and consumers share one L2 by construction; any number >= 1 of resident workgroups per XCD makes
//     progress (an item's dependencies always hold smaller tickets, i.e. are owned by running workgroups).
// Compared on the same box: one graph node per layer | tagged 8-byte granules on statically placed teams | this.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/sentinel_chain tools/sentinel_chain.hip && tools/bin/sentinel_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int M = 160, KD = 256, ND = 256, MT = M / 16, NT = ND / 16, KC = KD / 16;
constexpr unsigned SENT = 0xFFFFFFFFu;

__device__ __forceinline__ float act(float v) { return tanhf(v); }

// ------------------------------------------------------------------ baseline: one launch per layer
__global__ __launch_bounds__(256) void k_layer(const float* __restrict__ X, const float* __restrict__ Wp, float* __restrict__ Y) {
  __shared__ float red[1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int nt = blockIdx.x, rt = blockIdx.y;
  const float* rp = X + (size_t)(rt * 16 + (lane & 15)) * KD;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((size_t)nt * KC) * 64 + lane;
  f32x4 av[4], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = wave + 4 * j;
    av[j] = *reinterpret_cast<const f32x4*>(rp + g * 16 + kq * 4);
    bv[j] = wp[(size_t)g * 64];
  }
  __builtin_amdgcn_sched_barrier(0);
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
  const float v = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
  Y[(size_t)(rt * 16 + (tid >> 4)) * ND + nt * 16 + (tid & 15)] = act(v);
}

__device__ __forceinline__ u32x4 load16_sc1(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// ------------------------------------------------------------------ sentinel chain on self-assigned XCD teams
// ctl: [8][32] words, word 0 of row x = XCD x's ticket counter; row tile rt belongs to XCD rt % 8 (static here; the library
// version hands row-tile sets out of a global pool so that an XCD without workgroups starves nobody).
// MODE 0: tickets (any residency); MODE 1: fixed item per workgroup from its arrival rank (assumes 16 * tiles residents per XCD)
template <int MODE>
__global__ __launch_bounds__(256) void k_sent(float* bufs, const float* __restrict__ Wp, int layers, int hops, unsigned* ctl, int* status,
                                              unsigned long long* stamps, int* census) {
  __shared__ float red[1024];
  __shared__ unsigned s_item[2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  const int n_t = (int)xcc + 8 < MT ? 2 : 1;
  const unsigned ipx = n_t * 16;
  unsigned* q = ctl + xcc * 32;
  if (tid == 0) {
    s_item[0] = atomicAdd(q, 1u);
    if (census) census[blockIdx.x] = (int)xcc;
  }
  __syncthreads();
  unsigned long long t0 = 0;
  if (tid == 0) t0 = wall_clock64();
  const unsigned rank = s_item[0];
  if (MODE == 1 && rank >= ipx) return;
  int par = 0;
  unsigned done = 0;
  for (unsigned it = 0;; ++it) {
    const unsigned i = MODE == 0 ? s_item[par] : rank + it * ipx;
    const unsigned hop = i / ipx;
    if (hop >= (unsigned)hops) break;
    const unsigned rem = i - hop * ipx;
    const int rt = (int)xcc + 8 * (int)(rem >> 4), nt = rem & 15;
    if (MODE == 0 && tid == 0) s_item[par ^ 1] = atomicAdd(q, 1u);  // the next ticket travels while this item runs
    const int l = hop % layers;
    const float* in = bufs + (size_t)hop * M * KD;
    float* out = bufs + (size_t)(hop + 1) * M * ND;
    const f32x4* wp = reinterpret_cast<const f32x4*>(Wp + (size_t)l * ND * KD) + ((size_t)nt * KC) * 64 + lane;
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = wp[(size_t)(wave + 4 * j) * 64];
    const float* rp = in + (size_t)(rt * 16 + (lane & 15)) * KD + kq * 4;
    f32x4 av[4];
    int spins = 0;
    bool ok;
    do {
      u32x4 g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = load16_sc1(rp + (wave + 4 * j) * 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ok = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ok = ok && g[j].x != SENT && g[j].y != SENT && g[j].z != SENT && g[j].w != SENT;
        av[j] = f32x4{__uint_as_float(g[j].x), __uint_as_float(g[j].y), __uint_as_float(g[j].z), __uint_as_float(g[j].w)};
      }
      ok = __all(ok);
      if (!ok && ++spins > (1 << 18)) { if (lane == 0) atomicExch(status, 1); return; }
    } while (!ok);
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
    for (int ii = 0; ii < 4; ++ii) r[(4 * kq + ii) * 16 + (lane & 15)] = acc[ii];
    __syncthreads();
    const float v = act(red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid]);
    out[(size_t)(rt * 16 + (tid >> 4)) * ND + nt * 16 + (tid & 15)] = v;  // plain store: the line stays in this XCD's L2
    __syncthreads();
    par ^= 1;
    ++done;
  }
  if (stamps && tid == 0) { stamps[blockIdx.x * 2] = wall_clock64() - t0; stamps[blockIdx.x * 2 + 1] = done; }
}

int main(int argc, char** argv) {
  const int layers = 6, passes = argc > 1 ? atoi(argv[1]) : 50;
  const int hops = layers * passes;
  std::vector<float> hx((size_t)M * KD), hw((size_t)layers * KD * ND);
  srand(1);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
  std::vector<float> hwp(hw.size());
  for (int l = 0; l < layers; ++l)
    for (int j = 0; j < NT; ++j)
      for (int c = 0; c < KC; ++c)
        for (int ln = 0; ln < 64; ++ln)
          for (int i = 0; i < 4; ++i)
            hwp[(size_t)l * KD * ND + (((size_t)j * KC + c) * 64 + ln) * 4 + i] = hw[(size_t)l * KD * ND + (size_t)(16 * c + 4 * (ln >> 4) + i) * ND + 16 * j + (ln & 15)];
  float *dx, *dy, *dwp;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dy, hx.size() * 4)); CK(hipMalloc(&dwp, hwp.size() * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwp, hwp.data(), hwp.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  // ---- baseline: graph of `hops` nodes
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int it = 0; it < hops; ++it)
    hipLaunchKernelGGL(k_layer, dim3(NT, MT), dim3(256), 0, s, (it & 1) ? dy : dx, dwp + (size_t)(it % layers) * KD * ND, (it & 1) ? dx : dy);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  std::vector<float> ref((size_t)M * KD);
  CK(hipMemcpy(ref.data(), hops & 1 ? dy : dx, ref.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipEventRecord(ea, s));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(eb, s)); CK(hipStreamSynchronize(s));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, ea, eb));
  printf("graph, one node per layer                     : %.3f us per layer\n", ms * 1e3 / (5.0 * hops));
  // ---- sentinel chain
  const size_t lbuf = (size_t)M * KD;
  float* bufs; unsigned* ctl; int* status; unsigned long long* stamps; int* census;
  CK(hipMalloc(&bufs, (size_t)(hops + 1) * lbuf * 4));
  CK(hipMalloc(&ctl, 8 * 32 * 4)); CK(hipMalloc(&status, 4)); CK(hipMalloc(&stamps, 1024 * 16)); CK(hipMalloc(&census, 1024 * 4));
  for (int rep = 0; rep < 9; ++rep) {
    const int mode = rep < 3 ? 0 : (rep < 6 ? 1 : 0);
    const int grid = rep < 6 ? 256 : 512;  // 512: two workgroups per CU on the tickets (more pullers than items per hop)
    CK(hipMemset(bufs, 0xFF, (size_t)(hops + 1) * lbuf * 4));
    CK(hipMemcpy(bufs, hx.data(), lbuf * 4, hipMemcpyHostToDevice));
    CK(hipMemset(ctl, 0, 8 * 32 * 4)); CK(hipMemset(status, 0, 4)); CK(hipMemset(stamps, 0, 1024 * 16));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ea, s));
    if (mode == 0) hipLaunchKernelGGL(k_sent<0>, dim3(grid), dim3(256), 0, s, bufs, dwp, layers, hops, ctl, status, stamps, census);
    else hipLaunchKernelGGL(k_sent<1>, dim3(grid), dim3(256), 0, s, bufs, dwp, layers, hops, ctl, status, stamps, census);
    CK(hipEventRecord(eb, s)); CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, ea, eb));
    int st = 0;
    CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> hs(1024 * 2);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<int> hc(1024);
    CK(hipMemcpy(hc.data(), census, 1024 * 4, hipMemcpyDeviceToHost));
    unsigned long long mx = 0, items = 0, idle_wg = 0;
    for (int b = 0; b < grid; ++b) { mx = hs[2 * b] > mx ? hs[2 * b] : mx; items += hs[2 * b + 1]; idle_wg += hs[2 * b + 1] == 0; }
    int per_xcd[16] = {0}, rr = 0;
    for (int b = 0; b < grid; ++b) { per_xcd[hc[b] & 15]++; rr += hc[b] == (b & 7); }
    std::vector<float> got(lbuf);
    CK(hipMemcpy(got.data(), bufs + (size_t)hops * lbuf, lbuf * 4, hipMemcpyDeviceToHost));
    double worst = 0; size_t sent = 0;
    for (size_t i = 0; i < lbuf; ++i) {
      unsigned u; memcpy(&u, &got[i], 4);
      sent += u == SENT;
      const double dd = fabs((double)got[i] - (double)ref[i]);
      if (dd > worst) worst = dd;
    }
    printf("sentinel chain, %s, grid %d: %.3f us per layer by events, %.3f by the slowest workgroup's clock; status %d, items %llu (expected %d), "
           "idle workgroups %llu, max|chain-graph| %.3g, unwritten %zu; workgroups per XCD %d %d %d %d %d %d %d %d, b%%8 placement %d/%d\n",
           mode == 0 ? "tickets" : "fixed  ", grid, ms * 1e3 / hops, mx * 0.01 / hops, st, items, hops * MT * NT, idle_wg, worst, sent,
           per_xcd[0], per_xcd[1], per_xcd[2], per_xcd[3], per_xcd[4], per_xcd[5], per_xcd[6], per_xcd[7], rr, grid);
  }
  return 0;
}
