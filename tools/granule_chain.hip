// Microbenchmark: a chain of dependent 160 x 256 x 256 dense layers inside ONE launch, activations handed from layer to layer
// as DATA-TAGGED GRANULES ({value, tag} written by one 8-byte sc1 store, polled with sc1 loads: no flag, no fence, no barrier --
// MI355X_MICROARCH.md "handoff-1to1") against the same chain as one graph node per layer (the production structure).
// Rows never interact in the SQAIR pass, so a 16-row tile's next layer needs only the 16 column tiles of the SAME row tile:
// the "team" of a row tile is its 16 workgroups, and they exchange 16 x 256 activations (32 KB of granules) per layer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/granule_chain tools/granule_chain.hip && tools/bin/granule_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int M = 160, KD = 256, ND = 256, MT = M / 16, NT = ND / 16, KC = KD / 16;

__device__ __forceinline__ float act(float v) { return tanhf(v); }

// ------------------------------------------------------------------ baseline: one launch per layer (row-major activations)
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

// ------------------------------------------------------------------ granule chain
// granule buffer of one layer's OUTPUT: [row tile][chunk = producing column tile][row 16][k 16] x {float value, u32 tag}
// = the order in which the consuming wave's lane (row = l & 15, kq = l >> 4) finds its 4 k-values in 32 contiguous bytes.
struct Granule { float v; unsigned tag; };

__device__ __forceinline__ u32x4 load16_sc1(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store8_sc1(void* p, float v, unsigned tag) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"((unsigned long long)__float_as_uint(v) | ((unsigned long long)tag << 32)) : "memory");
}
__device__ __forceinline__ void store8_plain(void* p, float v, unsigned tag) {  // stays in the XCD's L2 (same-XCD teams only)
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"((unsigned long long)__float_as_uint(v) | ((unsigned long long)tag << 32)) : "memory");
}

// One workgroup = column tile `nt` of row tile `rt`, for every layer of the chain.  `in0` = layer-0 input as granules with tag 1.
// XCDTEAM: 1-D grid of 256 blocks; block b sits on XCD b % 8 (observed dispatch order) and row tile rt's 16 workgroups are all
// given to XCD rt % 8, so the granules travel through ONE L2 (plain stores keep the line there, sc1 loads read it there).
template <bool XCDTEAM>
__global__ __launch_bounds__(256) void k_chain(Granule* bufs, const float* __restrict__ Wp, int layers, int passes, int* status,
                                               unsigned long long* stamps) {
  __shared__ float red[1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  int nt = blockIdx.x, rt = blockIdx.y;
  if (XCDTEAM) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    nt = slot & 15;
    rt = xcd + 8 * (slot >> 4);
    if (rt >= MT) return;
    if (tid == 0) {
      unsigned x;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
      if ((int)(x & 0xf) != xcd) atomicExch(status, 2);
    }
  }
  const size_t lbuf = (size_t)MT * KC * 256;  // granules per layer buffer
  unsigned tag = 1;
  unsigned long long t0 = 0;
  if (stamps && tid == 0) t0 = wall_clock64();
  for (int it = 0; it < passes * layers; ++it) {
    const int l = it % layers;
    const Granule* in = bufs + (size_t)(it % (layers + 1)) * lbuf + ((size_t)rt * KC) * 256;
    Granule* out = bufs + (size_t)((it + 1) % (layers + 1)) * lbuf + ((size_t)rt * KC + nt) * 256;
    // weights of this layer first (they do not depend on the hand-off)
    const f32x4* wp = reinterpret_cast<const f32x4*>(Wp + (size_t)l * ND * KD) + ((size_t)nt * KC) * 64 + lane;
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = wp[(size_t)(wave + 4 * j) * 64];
    // A operand: 4 chunks x (4 granules = 32 bytes) per lane, polled until every tag matches
    f32x4 av[4];
    int spins = 0;
    bool ok;
    do {
      u32x4 g0[4], g1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Granule* p = in + (size_t)(wave + 4 * j) * 256 + (lane & 15) * 16 + kq * 4;
        g0[j] = load16_sc1(p);
        g1[j] = load16_sc1(p + 2);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ok = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ok = ok && g0[j].y == tag && g0[j].w == tag && g1[j].y == tag && g1[j].w == tag;
        av[j] = f32x4{__uint_as_float(g0[j].x), __uint_as_float(g0[j].z), __uint_as_float(g1[j].x), __uint_as_float(g1[j].z)};
      }
      ok = __all(ok);
      if (!ok && ++spins > (1 << 20)) { if (lane == 0) atomicExch(status, 1); return; }
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
    for (int i = 0; i < 4; ++i) r[(4 * kq + i) * 16 + (lane & 15)] = acc[i];
    __syncthreads();
    const float v = act(red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid]);
    ++tag;
    if (XCDTEAM) store8_plain(out + (tid >> 4) * 16 + (tid & 15), v, tag);
    else store8_sc1(out + (tid >> 4) * 16 + (tid & 15), v, tag);  // element (row tid >> 4, column tid & 15 of tile nt) = k index of the next layer
    __syncthreads();
  }
  if (stamps && tid == 0) stamps[rt * NT + nt] = wall_clock64() - t0;
}

int main() {
  const int layers = 6, passes = 100;
  std::vector<float> hx((size_t)M * KD), hw((size_t)layers * KD * ND);
  srand(1);
  for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
  // pack weights: [layer][n tile][k chunk][64 lanes][4]: lane l holds W[16c + 4(l>>4) + i][16j + (l&15)]
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
  // ---- baseline: graph of passes * layers nodes
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int it = 0; it < passes * layers; ++it)
    hipLaunchKernelGGL(k_layer, dim3(NT, MT), dim3(256), 0, s, (it & 1) ? dy : dx, dwp + (size_t)(it % layers) * KD * ND, (it & 1) ? dx : dy);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  std::vector<float> ref((size_t)M * KD);
  CK(hipMemcpy(ref.data(), (passes * layers) & 1 ? dy : dx, ref.size() * 4, hipMemcpyDeviceToHost));
  CK(hipEventRecord(ea, s));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(eb, s)); CK(hipStreamSynchronize(s));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, ea, eb));
  printf("graph, one node per layer : %.3f us per layer\n", ms * 1e3 / (5.0 * passes * layers));
  // ---- granule chain
  const size_t lbuf = (size_t)MT * KC * 256;
  Granule* bufs; int* status; unsigned long long* stamps;
  CK(hipMalloc(&bufs, (layers + 1) * lbuf * sizeof(Granule)));
  CK(hipMalloc(&status, 4)); CK(hipMalloc(&stamps, MT * NT * 8));
  std::vector<Granule> h0((layers + 1) * lbuf, Granule{0.f, 0u});
  for (int rt = 0; rt < MT; ++rt)
    for (int c = 0; c < KC; ++c)
      for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 16; ++k) h0[((size_t)rt * KC + c) * 256 + r * 16 + k] = Granule{hx[(size_t)(rt * 16 + r) * KD + c * 16 + k], 1u};
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipMemcpy(bufs, h0.data(), h0.size() * sizeof(Granule), hipMemcpyHostToDevice));
    CK(hipMemset(status, 0, 4));
    CK(hipEventRecord(ea, s));
    if (rep < 3) hipLaunchKernelGGL(k_chain<false>, dim3(NT, MT), dim3(256), 0, s, bufs, dwp, layers, passes, status, stamps);
    else hipLaunchKernelGGL(k_chain<true>, dim3(256), dim3(256), 0, s, bufs, dwp, layers, passes, status, stamps);
    CK(hipEventRecord(eb, s)); CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, ea, eb));
    int st = 0;
    CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> hs(MT * NT);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (auto v : hs) mx = v > mx ? v : mx;
    printf("granule chain (one launch, %s): %.3f us per layer by events, %.3f us by the slowest workgroup's device clock; status %d\n",
           rep < 3 ? "teams spread over XCDs, sc1 stores" : "one XCD per team, plain stores", ms * 1e3 / (passes * layers), mx * 0.01 / (passes * layers), st);
  }
  // correctness: final activations of the chain vs the graph
  std::vector<Granule> hout(lbuf);
  CK(hipMemcpy(hout.data(), bufs + (size_t)((passes * layers) % (layers + 1)) * lbuf, lbuf * sizeof(Granule), hipMemcpyDeviceToHost));
  double worst = 0;
  size_t bad_tag = 0;
  for (int rt = 0; rt < MT; ++rt)
    for (int c = 0; c < KC; ++c)
      for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 16; ++k) {
          const Granule gq = hout[((size_t)rt * KC + c) * 256 + r * 16 + k];
          const double d = fabs((double)gq.v - (double)ref[(size_t)(rt * 16 + r) * KD + c * 16 + k]);
          worst = d > worst ? d : worst;
          bad_tag += gq.tag != (unsigned)(passes * layers + 1);
        }
  printf("max |chain - graph| = %.3g, wrong tags %zu\n", worst, bad_tag);
  return 0;
}
