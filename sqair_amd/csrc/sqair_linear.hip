// Dense layers of the SQAIR hot path on the fp32 matrix cores of gfx950.
//
// Replaces every snt.Linear / MLP / VanillaRNN / GRU matmul of the reference graph
// (reference: sqair/neural.py:34-116, Sonnet cells via sqair/configs/mlp_mnist_model.py:86-125).
//
// Shape regime: M = B' (160..640) rows, K = 54..2500 (16384 at 128x128), N = 4..1152 — small-M,
// weight-stationary, latency-bound.  Design:
//   * one workgroup = one 16x16 output tile; its 4 wavefronts split K four ways and reduce through
//     LDS, so a 256-deep layer is 16 v_mfma_f32_16x16x4_f32 per wave (~0.2 us of issue) and the
//     grid has M/16 * N/16 workgroups (160 for a 160x256 layer) to spread over the 256 CUs;
//   * weights are pre-packed (sqair_pack_params) in MFMA fragment order: for N-tile j and 16-wide
//     K-chunk c, lane l holds W[16c + 4(l>>4) + i][16j + (l&15)], i = 0..3, as one float4 — a wave
//     reads a contiguous 1 KiB per chunk;
//   * the A operand is a virtual concatenation of up to 4 row-major segments (no concat kernel),
//     each lane loading float4 A[row = l&15][16c + 4(l>>4) .. +3]; chunk c's MFMA i therefore
//     consumes k = 16c + 4(l>>4) + i on both operands (a permutation of k inside the chunk, which
//     the sum does not care about);
//   * blockIdx -> (n-tile fastest): with n_tiles % 8 == 0 a given n-tile, i.e. a given slab of
//     weights, always lands on the same XCD (block b runs on XCD b % 8) and stays in that L2;
//   * epilogue fuses bias, a precomputed partial sum (loop-invariant part of the pre-activation),
//     the activation, and the GRU gate arithmetic.
#include "sqair_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_pack(const float* __restrict__ flat, float* __restrict__ packed, const int* __restrict__ idx,
                       int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int j = idx[i];
    packed[i] = j >= 0 ? flat[j] : 0.0f;
  }
}

__global__ void k_pack_bias(const float* __restrict__ flat, float* __restrict__ packed,
                            const int* __restrict__ idxa, const int* __restrict__ idxb, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int a = idxa[i], b = idxb[i];
    packed[i] = (a >= 0 ? flat[a] : 0.0f) + (b >= 0 ? flat[b] : 0.0f);
  }
}

int sq_launch_pack(const float* flat, float* packed_w, const int* idx, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, s, flat, packed_w, idx, n);
  return 0;
}
int sq_launch_pack_bias(const float* flat, float* packed_b, const int* idxa, const int* idxb, int64_t n,
                        hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_pack_bias, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, flat, packed_b, idxa, idxb, n);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// A-operand contract (checked on the host in sq_launch_linear): every segment base is 16-byte aligned,
// its row stride a multiple of 4 floats, and round_up(width, 4) floats of each row are readable and
// FINITE.  The kernel then needs no guards at all: a lane whose 4 k-positions fall beyond the segment
// width re-reads the last valid float4 of the row (address clamp) and the packed weights there are
// zero.  Guarded / scalar tail loads made hipcc fence every chunk with s_waitcnt vmcnt(0), turning one
// memory round trip into one per chunk (measured 5.5 us per launch, all of it latency).
// ---------------------------------------------------------------------------------------------------

// (SQ_ROWTILE_XCD_AFFINITY was measured: -0.3 us with L2-resident weights, +0.4 us once the weights of a whole
// frame rotate through L2; left off.  tools/linear_floor.hip reproduces both numbers.)
#define SQ_KLINEAR_NAME k_linear
#include "sqair_linear_kernel.inc"
#undef SQ_KLINEAR_NAME

// ---------------------------------------------------------------------------------------------------
// Throughput variant for the batched (M = B'*N = 640-row) layers that run once per frame: the 4 waves of a
// workgroup take 4 consecutive 16-row tiles against the SAME 16-column weight slab (one fetch of the slab
// per workgroup through L1 instead of four, no split-K, no LDS), each wave walking all K-chunks in blocks
// of NCH with every load of a block in flight at once.
// ---------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void k_linear_rows(const LinArgs a, const int kc_total, const int n_tiles,
                                                     unsigned long long* __restrict__ prof_ts) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int tile_n = blockIdx.x;
  const int tile_m = blockIdx.y * 4 + wave;
  (void)n_tiles;
  const int arow = min(tile_m * 16 + (lane & 15), a.M - 1);
  const int kq = lane >> 4;
  unsigned long long t_start = 0;
  if (prof_ts != nullptr && tid == 0) t_start = wall_clock64();

  // epilogue operands of the 4 outputs of this lane: rows 4*kq + i, column lane & 15
  const int n = tile_n * 16 + (lane & 15);
  const int nc = min(n, a.N - 1);
  const float* pb = a.bias + nc;
  const bool use_add = a.add != nullptr && nc < a.add_n;
  const bool g1 = a.epi == EPI_GRU1 && nc >= a.nh && nc < 2 * a.nh;
  const bool g2 = a.epi == EPI_GRU2;
  const float p_bias = *pb;
  float p_scale = *(a.scale_ptr != nullptr ? a.scale_ptr : pb);
  p_scale = a.scale_ptr != nullptr ? p_scale : 1.0f;
  float p_add[4], p_e0[4], p_e1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mc = min(tile_m * 16 + 4 * kq + i, a.M - 1);
    const int mcd = a.add_rmul ? (int)__umulhi((unsigned)mc, a.add_rmul) : mc;
    const float* pa = use_add ? a.add + (size_t)mcd * a.add_ld + nc : pb;
    const float* pe0 = g1 ? a.e0 + (size_t)mc * a.e0_ld + (nc - a.nh) : (g2 ? a.e0 + (size_t)mc * a.e0_ld + nc : pb);
    const float* pe1 = g2 ? a.e1 + (size_t)mc * a.e1_ld + nc : pb;
    p_add[i] = *pa;
    p_e0[i] = *pe0;
    p_e1[i] = *pe1;
  }

  int cum1 = 0x7fffffff, cum2 = 0x7fffffff, cum3 = 0x7fffffff;
#define SQ_ROWOF(sg) ((sg).rmul ? (int)__umulhi((unsigned)arow, (sg).rmul) : arow)
  const float* rp0 = a.seg[0].p + (size_t)SQ_ROWOF(a.seg[0]) * a.seg[0].ld;
  const float* rp1 = rp0; const float* rp2 = rp0; const float* rp3 = rp0;
  int lim0 = ((a.seg[0].width + 3) & ~3) - 4, lim1 = 0, lim2 = 0, lim3 = 0;
  {
    int c = (a.seg[0].width + 15) >> 4;
    if (a.nseg > 1) { cum1 = c; c += (a.seg[1].width + 15) >> 4; rp1 = a.seg[1].p + (size_t)SQ_ROWOF(a.seg[1]) * a.seg[1].ld; lim1 = ((a.seg[1].width + 3) & ~3) - 4; }
    if (a.nseg > 2) { cum2 = c; c += (a.seg[2].width + 15) >> 4; rp2 = a.seg[2].p + (size_t)SQ_ROWOF(a.seg[2]) * a.seg[2].ld; lim2 = ((a.seg[2].width + 3) & ~3) - 4; }
    if (a.nseg > 3) { cum3 = c; rp3 = a.seg[3].p + (size_t)SQ_ROWOF(a.seg[3]) * a.seg[3].ld; lim3 = ((a.seg[3].width + 3) & ~3) - 4; }
  }
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + ((size_t)tile_n * kc_total) * 64 + lane;
  const f32x4* __restrict__ wz = reinterpret_cast<const f32x4*>(a.wzero) + lane;
#pragma unroll 1
  for (int base = 0; base < kc_total; base += NCH) {
    f32x4 av[NCH], bv[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const bool valid = base + j < kc_total;
      const int g = valid ? base + j : 0;
      const bool s1 = g >= cum1, s2 = g >= cum2, s3 = g >= cum3;
      const float* rp = s3 ? rp3 : (s2 ? rp2 : (s1 ? rp1 : rp0));
      const int cb = s3 ? cum3 : (s2 ? cum2 : (s1 ? cum1 : 0));
      const int lim = s3 ? lim3 : (s2 ? lim2 : (s1 ? lim1 : lim0));
      av[j] = *reinterpret_cast<const f32x4*>(rp + min((g - cb) * 16 + kq * 4, lim));
      bv[j] = *(valid ? wp + (size_t)g * 64 : wz);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc1, 0, 0, 0);
    }
  }
  const float accv[4] = {acc0.x + acc1.x, acc0.y + acc1.y, acc0.z + acc1.z, acc0.w + acc1.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = tile_m * 16 + 4 * kq + i;
    if (m < a.M && n < a.N) {
      float v = accv[i] + p_bias + (use_add ? p_add[i] : 0.0f);
      if (a.epi == EPI_ACT) {
        v = sq_act(v, n < a.act_split ? a.act_a : a.act_b);
        a.out[(size_t)m * a.out_ld + n] = v * a.scale * p_scale;
      } else if (a.epi == EPI_GRU1) {
        const int nh = a.nh;
        if (n < nh) a.out[(size_t)m * a.out_ld + n] = sq_sigmoid(v);
        else if (n < 2 * nh) {
          const float rg = sq_sigmoid(v);
          a.o1[(size_t)m * a.o1_ld + (n - nh)] = rg * p_e0[i];
          if (a.o3 != nullptr) a.o3[(size_t)m * a.o3_ld + (n - nh)] = rg;
        }
        else a.o2[(size_t)m * a.o2_ld + (n - 2 * nh)] = v;
      } else {
        const float hc = tanhf(v);
        a.out[(size_t)m * a.out_ld + n] = (1.0f - p_e1[i]) * p_e0[i] + p_e1[i] * hc;
        if (a.o1 != nullptr) a.o1[(size_t)m * a.o1_ld + n] = hc;
      }
    }
  }
  if (prof_ts != nullptr) {
    __syncthreads();
    if (tid == 0) {
      atomicMin(prof_ts, t_start);
      atomicMax(prof_ts + 4096, wall_clock64());
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Wider throughput variant (EPI_ACT layers only): every wave takes ONE 16-row tile against NT consecutive 16-column weight
// slabs, so an activation fragment loaded once feeds NT MFMA chains (k_linear_rows: NT = 1) and the traffic towards L2 per
// MFMA drops from 5 KB to (4 + NT) KB per 4 * NT tiles.  Accumulation order per output element = k_linear_rows'
// (two accumulators, x/z and y/w), i.e. bit-identical results.
// ---------------------------------------------------------------------------------------------------
template <int NCH, int NT>
__global__ __launch_bounds__(256) void k_linear_wide(const LinArgs a, const int kc_total, const int n_tiles,
                                                     unsigned long long* __restrict__ prof_ts) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int tile_n0 = blockIdx.x * NT;
  const int tile_m = blockIdx.y * 4 + wave;
  const int arow = min(tile_m * 16 + (lane & 15), a.M - 1);
  unsigned long long t_start = 0;
  if (prof_ts != nullptr && tid == 0) t_start = wall_clock64();
  float p_scale = *(a.scale_ptr != nullptr ? a.scale_ptr : a.bias);
  p_scale = a.scale_ptr != nullptr ? p_scale : 1.0f;
  int cum1 = 0x7fffffff, cum2 = 0x7fffffff, cum3 = 0x7fffffff;
#define SQ_ROWOF(sg) ((sg).rmul ? (int)__umulhi((unsigned)arow, (sg).rmul) : arow)
  const float* rp0 = a.seg[0].p + (size_t)SQ_ROWOF(a.seg[0]) * a.seg[0].ld;
  const float* rp1 = rp0; const float* rp2 = rp0; const float* rp3 = rp0;
  int lim0 = ((a.seg[0].width + 3) & ~3) - 4, lim1 = 0, lim2 = 0, lim3 = 0;
  {
    int c = (a.seg[0].width + 15) >> 4;
    if (a.nseg > 1) { cum1 = c; c += (a.seg[1].width + 15) >> 4; rp1 = a.seg[1].p + (size_t)SQ_ROWOF(a.seg[1]) * a.seg[1].ld; lim1 = ((a.seg[1].width + 3) & ~3) - 4; }
    if (a.nseg > 2) { cum2 = c; c += (a.seg[2].width + 15) >> 4; rp2 = a.seg[2].p + (size_t)SQ_ROWOF(a.seg[2]) * a.seg[2].ld; lim2 = ((a.seg[2].width + 3) & ~3) - 4; }
    if (a.nseg > 3) { cum3 = c; rp3 = a.seg[3].p + (size_t)SQ_ROWOF(a.seg[3]) * a.seg[3].ld; lim3 = ((a.seg[3].width + 3) & ~3) - 4; }
  }
#undef SQ_ROWOF
  f32x4 acc0[NT], acc1[NT];
  const f32x4* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc0[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc1[t] = acc0[t];
    wp[t] = reinterpret_cast<const f32x4*>(a.wp) + ((size_t)min(tile_n0 + t, n_tiles - 1) * kc_total) * 64 + lane;  // surplus tiles re-read the last slab
  }
  const f32x4* __restrict__ wz = reinterpret_cast<const f32x4*>(a.wzero) + lane;
#pragma unroll 1
  for (int base = 0; base < kc_total; base += NCH) {
    f32x4 av[NCH], bv[NT][NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const bool valid = base + j < kc_total;
      const int g = valid ? base + j : 0;
      const bool s1 = g >= cum1, s2 = g >= cum2, s3 = g >= cum3;
      const float* rp = s3 ? rp3 : (s2 ? rp2 : (s1 ? rp1 : rp0));
      const int cb = s3 ? cum3 : (s2 ? cum2 : (s1 ? cum1 : 0));
      const int lim = s3 ? lim3 : (s2 ? lim2 : (s1 ? lim1 : lim0));
      av[j] = *reinterpret_cast<const f32x4*>(rp + min((g - cb) * 16 + kq * 4, lim));
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t][j] = *(valid ? wp[t] + (size_t)g * 64 : wz);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[t][j].x, acc0[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[t][j].y, acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[t][j].z, acc0[t], 0, 0, 0);
        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[t][j].w, acc1[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = (tile_n0 + t) * 16 + (lane & 15);
    if (tile_n0 + t < n_tiles && n < a.N) {
      const float p_bias = a.bias[n];
      const bool use_add = a.add != nullptr && n < a.add_n;
      const int act = n < a.act_split ? a.act_a : a.act_b;
      const float accv[4] = {acc0[t].x + acc1[t].x, acc0[t].y + acc1[t].y, acc0[t].z + acc1[t].z, acc0[t].w + acc1[t].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = tile_m * 16 + 4 * kq + i;
        if (m < a.M) {
          float p_add = 0.0f;
          if (use_add) {
            const int mcd = a.add_rmul ? (int)__umulhi((unsigned)m, a.add_rmul) : m;
            p_add = a.add[(size_t)mcd * a.add_ld + n];
          }
          const float v = sq_act(accv[i] + p_bias + p_add, act);
          a.out[(size_t)m * a.out_ld + n] = v * a.scale * p_scale;
        }
      }
    }
  }
  if (prof_ts != nullptr) {
    __syncthreads();
    if (tid == 0) {
      atomicMin(prof_ts, t_start);
      atomicMax(prof_ts + 4096, wall_clock64());
    }
  }
}

template <int NCH>
static void launch_nch(const LinArgs& a, const PackedLayer& L, int grid, hipStream_t s, unsigned long long* prof_ts) {
  (void)grid;
  const dim3 g(L.nt, (a.M + 15) / 16);
  switch (a.nseg) {  // the segment count is a template parameter: dead segment-selection code disappears
    case 1: hipLaunchKernelGGL((k_linear<NCH, 1>), g, dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, L.kc, L.nt, a, prof_ts); break;
    case 2: hipLaunchKernelGGL((k_linear<NCH, 2>), g, dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, L.kc, L.nt, a, prof_ts); break;
    case 3: hipLaunchKernelGGL((k_linear<NCH, 3>), g, dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, L.kc, L.nt, a, prof_ts); break;
    default: hipLaunchKernelGGL((k_linear<NCH, 4>), g, dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, L.kc, L.nt, a, prof_ts); break;
  }
}

static unsigned rmul_of(int rdiv) { return rdiv <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)rdiv) + 1u; }

int sq_launch_linear(const LinArgs& a_in, const PackedLayer& L, hipStream_t s, unsigned long long* prof_ts) {
  LinArgs a = a_in;
  for (int i = 0; i < a.nseg; ++i) a.seg[i].rmul = rmul_of(a.seg[i].rdiv);
  a.add_rmul = rmul_of(a.add_rdiv);
  const int mt = (a.M + 15) / 16;
  const int grid = mt * L.nt;
  if (grid <= 0) return 0;
  for (int i = 0; i < a.nseg; ++i) {  // A-operand contract
    const LinSeg& sg = a.seg[i];
    if ((reinterpret_cast<uintptr_t>(sg.p) & 15) != 0 || (sg.ld & 3) != 0 || sg.width < 1 || sg.rdiv < 1) return -5;
  }
  if (a.M >= 2048 || (a.M >= 256 && L.kc * L.nt >= 400 && L.kc <= 32)) {  // big batched once-per-frame layers: throughput variant
    // two column slabs per wave for launches with >= 1024 rows (measured at cfg-2 shapes: the 640-row per-frame layers lose
    // 2 % with it, the 6400-row decoder layers are neutral, and at 256 sequences per GPU -- where the slot layers have 1280
    // rows -- the whole pass gains 10 %; four slabs per wave: +4 % only)
    const int wgs2 = ((L.nt + 1) / 2) * ((mt + 3) / 4);
    if (a.epi == EPI_ACT && L.kc <= 32 && L.nt >= 2 && a.M >= 1024 && wgs2 >= 128) {
      const dim3 g2((L.nt + 1) / 2, (mt + 3) / 4);
      if (L.kc <= 4) hipLaunchKernelGGL((k_linear_wide<4, 2>), g2, dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
      else hipLaunchKernelGGL((k_linear_wide<8, 2>), g2, dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
      return 0;
    }
    const dim3 grid_r(L.nt, (mt + 3) / 4);
    if (L.kc <= 4) hipLaunchKernelGGL(k_linear_rows<4>, grid_r, dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
    else if (L.kc <= 8) hipLaunchKernelGGL(k_linear_rows<8>, grid_r, dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
    else hipLaunchKernelGGL(k_linear_rows<12>, grid_r, dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
    return 0;
  }
  const int per_wave = (L.kc + 3) / 4;
  static const int one_variant = getenv("SQAIR_ONE_LINEAR") ? atoi(getenv("SQAIR_ONE_LINEAR")) : 0;  // experiment knob
  if (one_variant && per_wave <= 8) {
    const dim3 g(L.nt, (a.M + 15) / 16);
    hipLaunchKernelGGL((k_linear<8, 4>), g, dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, L.kc, L.nt, a, prof_ts);
    return 0;
  }
  switch (per_wave) {
    case 1: launch_nch<1>(a, L, grid, s, prof_ts); break;
    case 2: launch_nch<2>(a, L, grid, s, prof_ts); break;
    case 3: launch_nch<3>(a, L, grid, s, prof_ts); break;
    case 4: launch_nch<4>(a, L, grid, s, prof_ts); break;
    case 5: launch_nch<5>(a, L, grid, s, prof_ts); break;
    case 6: launch_nch<6>(a, L, grid, s, prof_ts); break;
    case 7: launch_nch<7>(a, L, grid, s, prof_ts); break;
    case 8: launch_nch<8>(a, L, grid, s, prof_ts); break;
    case 9: launch_nch<9>(a, L, grid, s, prof_ts); break;
    default: launch_nch<10>(a, L, grid, s, prof_ts); break;  // deeper K: blocks of 10 chunks per wave
  }
  return 0;
}
