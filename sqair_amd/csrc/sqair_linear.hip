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

__device__ __forceinline__ f32x4 load_a4(const float* __restrict__ rowp, int kk, int width, bool vec) {
  f32x4 a;
  if (vec && kk + 4 <= width) {
    a = *reinterpret_cast<const f32x4*>(rowp + kk);
  } else {
    a.x = kk + 0 < width ? rowp[kk + 0] : 0.0f;
    a.y = kk + 1 < width ? rowp[kk + 1] : 0.0f;
    a.z = kk + 2 < width ? rowp[kk + 2] : 0.0f;
    a.w = kk + 3 < width ? rowp[kk + 3] : 0.0f;
  }
  return a;
}

// D = K-chunks per wave whose operand loads are issued before the first MFMA.  The layers of this model are
// latency-bound (measured: ~1 us per dependent global access, activations come from another XCD's writes), so the
// kernel is organised as ONE memory round trip: epilogue operands (bias, partial sums, GRU state) and every A / B
// fragment of the wave are requested up front, then the MFMAs drain them in order.
constexpr int LIN_D = 10;

__global__ __launch_bounds__(256) void k_linear(const LinArgs a, const int kc_total, const int n_tiles,
                                                unsigned long long* __restrict__ prof_ts) {
  __shared__ float red[4 * 256];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int tile_n = blockIdx.x % n_tiles;
  const int tile_m = blockIdx.x / n_tiles;
  const int arow = min(tile_m * 16 + (lane & 15), a.M - 1);
  const int kq = lane >> 4;
  unsigned long long t_start = 0;
  if (prof_ts != nullptr && tid == 0) t_start = wall_clock64();

  // ---- epilogue operands, requested first
  const int m = tile_m * 16 + (tid >> 4);
  const int n = tile_n * 16 + (tid & 15);
  const bool live = m < a.M && n < a.N;
  float p_bias = 0.0f, p_add = 0.0f, p_e0 = 0.0f, p_e1 = 0.0f, p_scale = 1.0f;
  if (live) {
    p_bias = a.bias[n];
    if (a.add != nullptr && n < a.add_n) p_add = a.add[(size_t)(m / a.add_rdiv) * a.add_ld + n];
    if (a.epi == EPI_GRU1) {
      if (n >= a.nh && n < 2 * a.nh) p_e0 = a.e0[(size_t)m * a.e0_ld + (n - a.nh)];
    } else if (a.epi == EPI_GRU2) {
      p_e0 = a.e0[(size_t)m * a.e0_ld + n];
      p_e1 = a.e1[(size_t)m * a.e1_ld + n];
    } else if (a.scale_ptr != nullptr) {
      p_scale = a.scale_ptr[0];
    }
  }

  // ---- segment table (chunk ranges are wave-uniform)
  int cum[5];
  const float* rowp[4];
  int width[4];
  bool vec[4];
  cum[0] = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < a.nseg) {
      const LinSeg sg = a.seg[s];
      cum[s + 1] = cum[s] + ((sg.width + 15) >> 4);
      rowp[s] = sg.p + (size_t)(arow / sg.rdiv) * sg.ld;
      width[s] = sg.width;
      vec[s] = ((reinterpret_cast<uintptr_t>(sg.p) & 15) == 0) && ((sg.ld & 3) == 0);
    } else {
      cum[s + 1] = 0x7fffffff;
      rowp[s] = a.seg[0].p;
      width[s] = 0;
      vec[s] = false;
    }
  }

  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + ((size_t)tile_n * kc_total) * 64 + lane;
  const int nmine = (kc_total - wave + 3) >> 2;  // this wave owns global chunks g = wave + 4 i
#pragma unroll 1
  for (int base = 0; base < nmine; base += LIN_D) {
    f32x4 av[LIN_D], bv[LIN_D];
#pragma unroll
    for (int j = 0; j < LIN_D; ++j) {
      if (base + j < nmine) {
        const int g = wave + 4 * (base + j);
        const int s = (g >= cum[1] ? 1 : 0) + (g >= cum[2] ? 1 : 0) + (g >= cum[3] ? 1 : 0);
        const int c = g - cum[s];
        av[j] = load_a4(rowp[s], c * 16 + kq * 4, width[s], vec[s]);
        bv[j] = wp[(size_t)g * 64];
      }
    }
#pragma unroll
    for (int j = 0; j < LIN_D; ++j) {
      if (base + j < nmine) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc1, 0, 0, 0);
      }
    }
  }

  // split-K reduction: acc[i] of lane l is C[row = 4*(l>>4) + i][col = l & 15]
  float* r = red + wave * 256;
  r[(4 * kq + 0) * 16 + (lane & 15)] = acc0.x + acc1.x;
  r[(4 * kq + 1) * 16 + (lane & 15)] = acc0.y + acc1.y;
  r[(4 * kq + 2) * 16 + (lane & 15)] = acc0.z + acc1.z;
  r[(4 * kq + 3) * 16 + (lane & 15)] = acc0.w + acc1.w;
  __syncthreads();
  if (live) {
    float v = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid] + p_bias + p_add;
    if (a.epi == EPI_ACT) {
      v = sq_act(v, n < a.act_split ? a.act_a : a.act_b);
      a.out[(size_t)m * a.out_ld + n] = v * a.scale * p_scale;
    } else if (a.epi == EPI_GRU1) {
      // columns [z | r | x W_h + b_h]   (snt.GRU)
      const int nh = a.nh;
      if (n < nh) a.out[(size_t)m * a.out_ld + n] = sq_sigmoid(v);
      else if (n < 2 * nh) a.o1[(size_t)m * a.o1_ld + (n - nh)] = sq_sigmoid(v) * p_e0;
      else a.o2[(size_t)m * a.o2_ld + (n - 2 * nh)] = v;
    } else {  // EPI_GRU2: h' = (1 - z) h + z tanh(x W_h + (r h) U_h + b_h)
      a.out[(size_t)m * a.out_ld + n] = (1.0f - p_e1) * p_e0 + p_e1 * tanhf(v);
    }
  }
  if (prof_ts != nullptr) {
    __syncthreads();
    if (tid == 0) {
      atomicMin(prof_ts, t_start);
      atomicMax(prof_ts + 4096, wall_clock64());  // end slots follow the PROF_MAX start slots
    }
  }
}

int sq_launch_linear(const LinArgs& a, const PackedLayer& L, hipStream_t s, unsigned long long* prof_ts) {
  const int mt = (a.M + 15) / 16;
  const int grid = mt * L.nt;
  if (grid <= 0) return 0;
  hipLaunchKernelGGL(k_linear, dim3(grid), dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
  return 0;
}
