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
#include <type_traits>
#include "sqair_common.h"
#include "sqair_lin_device.h"
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_pack(const float* __restrict__ flat, float* __restrict__ packed, const int* __restrict__ idx,
                       int64_t n SQ_TLP) {
  SQ_TL_SCOPE;
  // A thread's elements in ONE trip where they fit: all their indices, then all the gathers, then the stores -- unconditional
  // loads from clamped positions (one element per loop trip was index -> gather, two dependent round trips, seven times over
  // for the ~7 M floats of the two packs: 17 us at the end of every training step)
  constexpr int U = 8;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride * U) {
    int j[U];
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = idx[min(i + u * stride, n - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = flat[max(j[u], 0)];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n) packed[i + u * stride] = j[u] >= 0 ? v[u] : 0.0f;
  }
}

__global__ void k_pack_bias(const float* __restrict__ flat, float* __restrict__ packed,
                            const int* __restrict__ idxa, const int* __restrict__ idxb, int64_t n SQ_TLP) {
  SQ_TL_SCOPE;
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
  SQ_LAUNCH(k_pack, dim3(blocks), dim3(256), 0, s, flat, packed_w, idx, n);
  return 0;
}
int sq_launch_pack_bias(const float* flat, float* packed_b, const int* idxa, const int* idxb, int64_t n,
                        hipStream_t s) {
  if (n <= 0) return 0;
  SQ_LAUNCH(k_pack_bias, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, flat, packed_b, idxa, idxb, n);
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
                                                     unsigned long long* __restrict__ prof_ts SQ_TLP) {
  SQ_TL_SCOPE;
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
#undef SQ_ROWOF
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
        const float hc = sq_tanh(v);
        a.out[(size_t)m * a.out_ld + n] = sq_gru_blend(p_e1[i], p_e0[i], hc);
        if (a.o1 != nullptr) a.o1[(size_t)m * a.o1_ld + n] = hc;
      }
    }
  }
  if (prof_ts != nullptr) {
    __syncthreads();
    // (stamped by the last column-tile workgroup of every row tile and by workgroup (0, 0) only: one atomic pair per workgroup
    // serialises thousands of them on one address and made the many-workgroup launches look 2-3x longer than they are)
    if (tid == 0 && (blockIdx.x == gridDim.x - 1 || (blockIdx.x == 0 && blockIdx.y == 0))) {
      atomicMin(prof_ts, t_start);
      atomicMax(prof_ts + 4096, wall_clock64());
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Macro-tile throughput kernel.  Every wave owns MT 16-row tiles x NT 16-column weight slabs (a 16 MT x 16 NT block of
// the output): an activation fragment feeds NT MFMA chains, a weight fragment MT of them, so the bytes a CU pulls out of
// L2 per MFMA fall from 2 KB (k_linear_rows) to (MT + NT) / (MT NT) KB; the 4 waves of a workgroup stack along M and share
// the NT weight slabs through the L1.  What a CU can pull out of L2 (~50 GB/s measured, tools/chain_floor.hip) is the
// roof of the 16x16-tile kernels from M ~ 640 rows up, not the matrix pipe.
// The K loop is software-pipelined: the operand loads of block i + 1 are issued before the MFMAs of block i.
// Accumulation order per output element = k_linear_rows' (chunks in order, accumulator 0 takes the x / z sub-steps,
// accumulator 1 the y / w ones, summed at the end): results are bit-identical across the rows / wide / macro-tile variants,
// so the tile shape is a pure performance choice.  All three epilogues.
// ---------------------------------------------------------------------------------------------------
template <int NCH, int MT, int NT, bool COAL = false>
__global__ __launch_bounds__(256) void k_linear_mt(const LinArgs a, const int kc_total, const int n_tiles,
                                                   unsigned long long* __restrict__ prof_ts SQ_TLP) {
  SQ_TL_SCOPE;
  static_assert(MT == 1 || MT == 2, "row tiles per wave");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  // COAL: the A fragment is LOADED with 4 consecutive lanes on the 64 contiguous bytes of one row's chunk (16 row segments
  // per instruction instead of 64 scattered 16-byte pieces) and brought into the MFMA operand order (lane = row + 16 kq) by
  // four ds_bpermute_b32 -- same values, same arithmetic.
  const int lrow = COAL ? (lane >> 2) : (lane & 15);   // row of the 16-row tile this lane LOADS
  const int lkq = COAL ? (lane & 3) : kq;              // 4-float piece of the 16-wide chunk this lane LOADS
  const int bp_src = (4 * (lane & 15) + kq) * 4;       // byte address for ds_bpermute: the loading lane that holds my operand
  const int tile_n0 = blockIdx.x * NT;
  const int tile_m0 = (blockIdx.y * 4 + wave) * MT;
  unsigned long long t_start = 0;
  if (prof_ts != nullptr && tid == 0) t_start = wall_clock64();
  // segment table (wave-uniform) + per-lane row pointers of the (up to two) row tiles, all in NAMED locals: a select
  // chain over members of the by-value argument struct or over elements of a local array is turned into an indexed load
  // from a scratch copy (seen in this kernel's first version: flat loads, s_waitcnt vmcnt(0) after each of them)
  int cum1 = 0x7fffffff, cum2 = 0x7fffffff, cum3 = 0x7fffffff;
  const int arowA = min(tile_m0 * 16 + lrow, a.M - 1);
  const int arowB = min((tile_m0 + MT - 1) * 16 + lrow, a.M - 1);
#define SQ_ROWOF(sg, r) ((sg).rmul ? (int)__umulhi((unsigned)(r), (sg).rmul) : (r))
  const float* rpA0 = a.seg[0].p + (size_t)SQ_ROWOF(a.seg[0], arowA) * a.seg[0].ld;
  const float* rpB0 = a.seg[0].p + (size_t)SQ_ROWOF(a.seg[0], arowB) * a.seg[0].ld;
  const float *rpA1 = rpA0, *rpA2 = rpA0, *rpA3 = rpA0, *rpB1 = rpB0, *rpB2 = rpB0, *rpB3 = rpB0;
  int lim0 = ((a.seg[0].width + 3) & ~3) - 4, lim1 = 0, lim2 = 0, lim3 = 0;
  {
    int c = (a.seg[0].width + 15) >> 4;
    if (a.nseg > 1) {
      cum1 = c; c += (a.seg[1].width + 15) >> 4; lim1 = ((a.seg[1].width + 3) & ~3) - 4;
      rpA1 = a.seg[1].p + (size_t)SQ_ROWOF(a.seg[1], arowA) * a.seg[1].ld;
      rpB1 = a.seg[1].p + (size_t)SQ_ROWOF(a.seg[1], arowB) * a.seg[1].ld;
    }
    if (a.nseg > 2) {
      cum2 = c; c += (a.seg[2].width + 15) >> 4; lim2 = ((a.seg[2].width + 3) & ~3) - 4;
      rpA2 = a.seg[2].p + (size_t)SQ_ROWOF(a.seg[2], arowA) * a.seg[2].ld;
      rpB2 = a.seg[2].p + (size_t)SQ_ROWOF(a.seg[2], arowB) * a.seg[2].ld;
    }
    if (a.nseg > 3) {
      cum3 = c; lim3 = ((a.seg[3].width + 3) & ~3) - 4;
      rpA3 = a.seg[3].p + (size_t)SQ_ROWOF(a.seg[3], arowA) * a.seg[3].ld;
      rpB3 = a.seg[3].p + (size_t)SQ_ROWOF(a.seg[3], arowB) * a.seg[3].ld;
    }
  }
#undef SQ_ROWOF
  f32x4 acc0[MT][NT], acc1[MT][NT];
  const f32x4* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc0[i][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc1[i][t] = acc0[i][t]; }
    wp[t] = reinterpret_cast<const f32x4*>(a.wp) + ((size_t)min(tile_n0 + t, n_tiles - 1) * kc_total) * 64 + lane;  // surplus slabs re-read the last one
  }
  const f32x4* __restrict__ wz = reinterpret_cast<const f32x4*>(a.wzero) + lane;

  // Operand loads of one block of NCH chunks.  No branches: a chunk beyond the end re-reads A chunk 0 against the packed
  // buffer's block of zero weights (acc + x * 0 = acc exactly, the A operand is finite by contract).
#define SQ_MT_ISSUE(AV, BV, BASE)                                                                   \
  _Pragma("unroll") for (int j = 0; j < NCH; ++j) {                                                 \
    const bool valid = (BASE) + j < kc_total;                                                       \
    const int g = valid ? (BASE) + j : 0;                                                           \
    const bool s1 = g >= cum1, s2 = g >= cum2, s3 = g >= cum3;                                      \
    const int cb = s3 ? cum3 : (s2 ? cum2 : (s1 ? cum1 : 0));                                       \
    const int lim = s3 ? lim3 : (s2 ? lim2 : (s1 ? lim1 : lim0));                                   \
    const int kk = min((g - cb) * 16 + lkq * 4, lim);                                               \
    const float* pA = s3 ? rpA3 : (s2 ? rpA2 : (s1 ? rpA1 : rpA0));                                 \
    AV[j][0] = *reinterpret_cast<const f32x4*>(pA + kk);                                            \
    if (MT > 1) {                                                                                   \
      const float* pB = s3 ? rpB3 : (s2 ? rpB2 : (s1 ? rpB1 : rpB0));                               \
      AV[j][MT - 1] = *reinterpret_cast<const f32x4*>(pB + kk);                                     \
    }                                                                                               \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) BV[j][t] = *(valid ? wp[t] + (size_t)g * 64 : wz); \
  }
#define SQ_MT_CONSUME(AV, BV)                                                                       \
  if (COAL) {                                                                                       \
    _Pragma("unroll") for (int j = 0; j < NCH; ++j) {                                               \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                              \
        AV[j][i].x = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_src, __float_as_int(AV[j][i].x))); \
        AV[j][i].y = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_src, __float_as_int(AV[j][i].y))); \
        AV[j][i].z = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_src, __float_as_int(AV[j][i].z))); \
        AV[j][i].w = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_src, __float_as_int(AV[j][i].w))); \
      }                                                                                             \
    }                                                                                               \
  }                                                                                                 \
  _Pragma("unroll") for (int j = 0; j < NCH; ++j) {                                                 \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                              \
        acc0[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[j][i].x, BV[j][t].x, acc0[i][t], 0, 0, 0); \
        acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[j][i].y, BV[j][t].y, acc1[i][t], 0, 0, 0); \
        acc0[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[j][i].z, BV[j][t].z, acc0[i][t], 0, 0, 0); \
        acc1[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[j][i].w, BV[j][t].w, acc1[i][t], 0, 0, 0); \
      }                                                                                             \
    }                                                                                               \
  }
  f32x4 av0[NCH][MT], bv0[NCH][NT], av1[NCH][MT], bv1[NCH][NT];
  SQ_MT_ISSUE(av0, bv0, 0)
#pragma unroll 1
  for (int base = 0; base < kc_total; base += 2 * NCH) {
    SQ_MT_ISSUE(av1, bv1, base + NCH)
    __builtin_amdgcn_sched_barrier(0);
    SQ_MT_CONSUME(av0, bv0)
    __builtin_amdgcn_sched_barrier(0);
    SQ_MT_ISSUE(av0, bv0, base + 2 * NCH)
    __builtin_amdgcn_sched_barrier(0);
    SQ_MT_CONSUME(av1, bv1)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SQ_MT_ISSUE
#undef SQ_MT_CONSUME

  // Epilogue as ONE rolled loop over the wave's MT * NT * 4 output elements per lane: the sums are parked in LDS (wave-private
  // slab, element-major) so that the loop body -- the three epilogue kinds with their exp / tanh / log1p expansions -- exists
  // once.  Unrolled over the elements this kernel was 27 - 93 KB of code and every launch paid tens of microseconds of cold
  // instruction fetch (measured in the pass: 67 us against 17 us for the same tile shape with a small epilogue).
  constexpr int E = MT * NT * 4;
  __shared__ float epi_s[4][E][64];
  __shared__ float epo_s[4][E][3][64];  // epilogue operands (addend, e0, e1) of the same elements, when the layer has any
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      epi_s[wave][(i * NT + t) * 4 + 0][lane] = acc0[i][t].x + acc1[i][t].x;
      epi_s[wave][(i * NT + t) * 4 + 1][lane] = acc0[i][t].y + acc1[i][t].y;
      epi_s[wave][(i * NT + t) * 4 + 2][lane] = acc0[i][t].z + acc1[i][t].z;
      epi_s[wave][(i * NT + t) * 4 + 3][lane] = acc0[i][t].w + acc1[i][t].w;
    }
  }
  float p_scale = *(a.scale_ptr != nullptr ? a.scale_ptr : a.bias);
  p_scale = a.scale_ptr != nullptr ? p_scale : 1.0f;
  const bool g2 = a.epi == EPI_GRU2;
  const bool has_operands = a.add != nullptr || a.epi != EPI_ACT;  // wave-uniform
  if (has_operands) {
    // all operand loads of the lane's E elements in ONE burst (clamped addresses, dummy = the bias word), parked in LDS for the
    // rolled loop below: guarded loads inside that loop would cost one memory round trip per element
    float qa[E], q0[E], q1[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int q = e & 3, t = (e >> 2) % NT, i = (e >> 2) / NT;
      const int n = min((min(tile_n0 + t, n_tiles - 1)) * 16 + (lane & 15), a.N - 1);
      const int m = min((tile_m0 + i) * 16 + 4 * kq + q, a.M - 1);
      const float* pb = a.bias + n;
      const bool use_add = a.add != nullptr && n < a.add_n;
      const bool g1 = a.epi == EPI_GRU1 && n >= a.nh && n < 2 * a.nh;
      const float* pa = use_add ? a.add + (size_t)(a.add_rmul ? (int)__umulhi((unsigned)m, a.add_rmul) : m) * a.add_ld + n : pb;
      const float* pe0 = g1 ? a.e0 + (size_t)m * a.e0_ld + (n - a.nh) : (g2 ? a.e0 + (size_t)m * a.e0_ld + n : pb);
      const float* pe1 = g2 ? a.e1 + (size_t)m * a.e1_ld + n : pb;
      qa[e] = *pa; q0[e] = *pe0; q1[e] = *pe1;
      if (!use_add) qa[e] = 0.0f;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) { epo_s[wave][e][0][lane] = qa[e]; epo_s[wave][e][1][lane] = q0[e]; epo_s[wave][e][2][lane] = q1[e]; }
  }
#pragma unroll 1
  for (int e = 0; e < E; ++e) {
    const int q = e & 3, t = (e >> 2) % NT, i = (e >> 2) / NT;
    const int n = (tile_n0 + t) * 16 + (lane & 15);
    const int m = (tile_m0 + i) * 16 + 4 * kq + q;
    if (tile_n0 + t < n_tiles && n < a.N && m < a.M) {
      float p_add = 0.0f, p_e0 = 0.0f, p_e1 = 0.0f;
      if (has_operands) { p_add = epo_s[wave][e][0][lane]; p_e0 = epo_s[wave][e][1][lane]; p_e1 = epo_s[wave][e][2][lane]; }
      x_epilogue(a, m, n, epi_s[wave][e][lane] + a.bias[n] + p_add, p_e0, p_e1, p_scale);
    }
  }
  if (prof_ts != nullptr) {
    __syncthreads();
    // (stamped by the last column-tile workgroup of every row tile and by workgroup (0, 0) only: one atomic pair per workgroup
    // serialises thousands of them on one address and made the many-workgroup launches look 2-3x longer than they are)
    if (tid == 0 && (blockIdx.x == gridDim.x - 1 || (blockIdx.x == 0 && blockIdx.y == 0))) {
      atomicMin(prof_ts, t_start);
      atomicMax(prof_ts + 4096, wall_clock64());
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// LDS-tiled throughput kernel for launches with thousands of rows (the decoder's T B' N rows, every layer from ~128 sequences
// per GPU up).  Workgroup tile 128 x 32 TN, 2 x 2 waves, wave tile 64 x 16 TN (4 x TN MFMA tiles: an A fragment feeds TN
// MFMAs, a B fragment 4).  The 128 x 32 activation block of a K step is fetched with 8 lanes on each row's 128 contiguous
// bytes (full cache lines; the MFMA operand order would put 16 different rows into consecutive lanes), staged through a
// padded LDS tile (register-staged double buffer: the global loads of step s + 1 are in flight while step s computes) and
// read back as ds_read_b128 fragments; the weight fragments come straight from the packed buffer (already in operand order,
// 1 KB contiguous per wave).  One accumulator per MFMA tile, chunks in order: results differ from the 16x16-tile kernels in
// the last bits (different summation order), never between two launches of this kernel.
// ---------------------------------------------------------------------------------------------------
template <int TN>
__global__ __launch_bounds__(256) void k_linear_lds(const LinArgs a, const int kc_total, const int n_tiles,
                                                    unsigned long long* __restrict__ prof_ts SQ_TLP) {
  SQ_TL_SCOPE;
  constexpr int LDA = 36;  // 32 floats of a K step + 4 of padding: the 16 rows of a fragment read land on distinct 16-byte slots
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * LDA];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, l15 = lane & 15;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int row0 = blockIdx.y * 128;
  const int tile_n0 = (blockIdx.x * 2 + wave_n) * TN;
  unsigned long long t_start = 0;
  if (prof_ts != nullptr && tid == 0) t_start = wall_clock64();
  // segment table in named locals (see k_linear_mt)
  const float* sp0 = a.seg[0].p; const float* sp1 = a.seg[1].p; const float* sp2 = a.seg[2].p; const float* sp3 = a.seg[3].p;
  const int sl0 = a.seg[0].ld, sl1 = a.seg[1].ld, sl2 = a.seg[2].ld, sl3 = a.seg[3].ld;
  const unsigned sm0 = a.seg[0].rmul, sm1 = a.seg[1].rmul, sm2 = a.seg[2].rmul, sm3 = a.seg[3].rmul;
  const int w0 = a.seg[0].width, w1 = a.seg[1].width, w2 = a.seg[2].width, w3 = a.seg[3].width;
  const int c1 = (w0 + 15) >> 4;
  const int c2 = c1 + (a.nseg > 1 ? (w1 + 15) >> 4 : 0);
  const int c3 = c2 + (a.nseg > 2 ? (w2 + 15) >> 4 : 0);
  const int cum1 = a.nseg > 1 ? c1 : 0x7fffffff, cum2 = a.nseg > 2 ? c2 : 0x7fffffff, cum3 = a.nseg > 3 ? c3 : 0x7fffffff;
  const int lim0 = ((w0 + 3) & ~3) - 4, lim1 = ((w1 + 3) & ~3) - 4, lim2 = ((w2 + 3) & ~3) - 4, lim3 = ((w3 + 3) & ~3) - 4;
  // staging map: thread -> (row tid >> 3 of each 32-row pass, 16 bytes at float (tid & 7) * 4 of the 32-wide K step)
  const int srow = tid >> 3, sk = (tid & 7) * 4;
  const int sch = sk >> 4, skk = sk & 15;  // which of the step's two chunks, offset inside it
  int arow[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) arow[p] = min(row0 + srow + 32 * p, a.M - 1);
  f32x4 ra[4];
#define SQ_LDS_STAGE(STEP)                                                                                   \
  {                                                                                                           \
    const int gq = 2 * (STEP) + sch;                                                                          \
    const int g = gq < kc_total ? gq : 0; /* beyond the end: any finite data (the weights there are zero) */  \
    const bool s1 = g >= cum1, s2 = g >= cum2, s3 = g >= cum3;                                                \
    const float* sp = s3 ? sp3 : (s2 ? sp2 : (s1 ? sp1 : sp0));                                               \
    const int sl = s3 ? sl3 : (s2 ? sl2 : (s1 ? sl1 : sl0));                                                  \
    const unsigned sm = s3 ? sm3 : (s2 ? sm2 : (s1 ? sm1 : sm0));                                             \
    const int cb = s3 ? cum3 : (s2 ? cum2 : (s1 ? cum1 : 0));                                                 \
    const int lim = s3 ? lim3 : (s2 ? lim2 : (s1 ? lim1 : lim0));                                             \
    const int kk = min((g - cb) * 16 + skk, lim);                                                             \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                           \
      const int r = sm ? (int)__umulhi((unsigned)arow[p], sm) : arow[p];                                      \
      ra[p] = *reinterpret_cast<const f32x4*>(sp + (size_t)r * sl + kk);                                      \
    }                                                                                                         \
  }
#define SQ_LDS_WRITE(BUF)                                                                                     \
  _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                               \
    *reinterpret_cast<f32x4*>(&lds[(BUF) * 128 * LDA + (srow + 32 * p) * LDA + sk]) = ra[p];

  f32x4 acc[4][TN];
  const f32x4* wp[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    wp[t] = reinterpret_cast<const f32x4*>(a.wp) + ((size_t)min(tile_n0 + t, n_tiles - 1) * kc_total) * 64 + lane;
  }
  const f32x4* __restrict__ wz = reinterpret_cast<const f32x4*>(a.wzero) + lane;
  const int steps = (kc_total + 1) >> 1;
  SQ_LDS_STAGE(0)
  SQ_LDS_WRITE(0)
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) SQ_LDS_STAGE(s + 1)
    f32x4 bv[2][TN];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int g = 2 * s + c;
#pragma unroll
      for (int t = 0; t < TN; ++t) bv[c][t] = *(g < kc_total ? wp[t] + (size_t)g * 64 : wz);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x4 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const f32x4*>(&lds[buf * 128 * LDA + (wave_m * 64 + i * 16 + l15) * LDA + c * 16 + kq * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bv[c][t].x, acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bv[c][t].y, acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bv[c][t].z, acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bv[c][t].w, acc[i][t], 0, 0, 0);
        }
      }
    }
    if (s + 1 < steps) {
      SQ_LDS_WRITE(buf ^ 1)
    }
    __syncthreads();
  }
#undef SQ_LDS_STAGE
#undef SQ_LDS_WRITE
  // epilogue: rolled loop over the lane's 4 * TN * 4 outputs, sums parked in LDS (two rounds of two row tiles; see k_linear_mt)
  float* epi = lds + wave * (2 * TN * 4 * 64);   // 4 waves x 2 TN x 4 x 64 floats <= 2 x 128 x 36
  float p_scale = *(a.scale_ptr != nullptr ? a.scale_ptr : a.bias);
  p_scale = a.scale_ptr != nullptr ? p_scale : 1.0f;
  const bool g2 = a.epi == EPI_GRU2;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const f32x4 v = acc[half * 2 + ii][t];
        epi[((ii * TN + t) * 4 + 0) * 64 + lane] = v.x;
        epi[((ii * TN + t) * 4 + 1) * 64 + lane] = v.y;
        epi[((ii * TN + t) * 4 + 2) * 64 + lane] = v.z;
        epi[((ii * TN + t) * 4 + 3) * 64 + lane] = v.w;
      }
    }
#pragma unroll 1
    for (int e = 0; e < 2 * TN * 4; ++e) {
      const int q = e & 3, t = (e >> 2) % TN, ii = (e >> 2) / TN;
      const int n = (tile_n0 + t) * 16 + l15;
      const int m = row0 + wave_m * 64 + (half * 2 + ii) * 16 + 4 * kq + q;
      if (tile_n0 + t < n_tiles && n < a.N && m < a.M) {
        const bool use_add = a.add != nullptr && n < a.add_n;
        const bool g1 = a.epi == EPI_GRU1 && n >= a.nh && n < 2 * a.nh;
        float p_add = 0.0f, p_e0 = 0.0f, p_e1 = 0.0f;
        if (use_add) p_add = a.add[(size_t)(a.add_rmul ? (int)__umulhi((unsigned)m, a.add_rmul) : m) * a.add_ld + n];
        if (g1) p_e0 = a.e0[(size_t)m * a.e0_ld + (n - a.nh)];
        if (g2) { p_e0 = a.e0[(size_t)m * a.e0_ld + n]; p_e1 = a.e1[(size_t)m * a.e1_ld + n]; }
        x_epilogue(a, m, n, epi[e * 64 + lane] + a.bias[n] + p_add, p_e0, p_e1, p_scale);
      }
    }
  }
  if (prof_ts != nullptr) {
    __syncthreads();
    // (stamped by the last column-tile workgroup of every row tile and by workgroup (0, 0) only: one atomic pair per workgroup
    // serialises thousands of them on one address and made the many-workgroup launches look 2-3x longer than they are)
    if (tid == 0 && (blockIdx.x == gridDim.x - 1 || (blockIdx.x == 0 && blockIdx.y == 0))) {
      atomicMin(prof_ts, t_start);
      atomicMax(prof_ts + 4096, wall_clock64());
    }
  }
}

#define SQ_T2_NAME k_linear_t2
#include "sqair_linear_t2.inc"
#undef SQ_T2_NAME
template <int NB, int NW>
static void launch_t2(const LinArgs& a, const PackedLayer& L, hipStream_t s) {
  const dim3 g((L.nt + 1) / 2, (a.M + 31) / 32);
  switch (a.nseg) {
    case 1: SQ_LAUNCH((k_linear_t2<NB, 1, NW>), g, dim3(64 * NW), 0, s, a, L.kc, L.nt); break;
    case 2: SQ_LAUNCH((k_linear_t2<NB, 2, NW>), g, dim3(64 * NW), 0, s, a, L.kc, L.nt); break;
    case 3: SQ_LAUNCH((k_linear_t2<NB, 3, NW>), g, dim3(64 * NW), 0, s, a, L.kc, L.nt); break;
    default: SQ_LAUNCH((k_linear_t2<NB, 4, NW>), g, dim3(64 * NW), 0, s, a, L.kc, L.nt); break;
  }
}
// 4 waves per tile; operand blocks of NB chunks, two in flight.  (NW = 8 / 16 waves per tile -- one block per wave, no serial
// chain of round trips -- measured SLOWER: 640 x 1152 x 384 12.1 -> 30.9 us with 16 waves.  The launch is bound by what the
// memory system delivers to scattered 64-byte row pieces, and more waves in flight only lengthen every round trip.)
static void launch_t2_any(const LinArgs& a, const PackedLayer& L, hipStream_t s) {
  const int per = (L.kc + 3) / 4;
  if (per <= 2) launch_t2<2, 4>(a, L, s); else if (per <= 3 || per == 5 || per == 6) launch_t2<3, 4>(a, L, s); else launch_t2<4, 4>(a, L, s);
}

// ---------------------------------------------------------------------------------------------------
// Large-row throughput kernel (launches with thousands of rows whose tiles fill the chip: the once-per-frame layers from ~128
// sequences per GPU up, the decoder, cfg-4): BOTH operands through LDS.
//   * workgroup tile 128 rows x (2 TNW) 16-column tiles (TNW = 4: 128 x 128), 2 x 2 waves, wave tile 64 x 16 TNW: per 16-deep
//     K chunk a wave reads 4 A and TNW B fragments (ds_read_b128) for 16 TNW MFMAs (16x16x4 fp32): with TNW = 4 that is
//     8 KB of LDS reads per 2048 matrix-core cycles and 16 KB of global traffic per workgroup and chunk (8 B/clk/CU; the
//     128 x 64 tile of k_linear_lds with weights straight from global memory needs twice that, which is what capped it at
//     0.40 of the fp32 matrix peak);
//   * global -> LDS without registers (global_load_lds_dwordx4, 1 KB per wave instruction, LDS image = lane order): a weight
//     fragment block of the packed buffer IS lane order; an activation block (16 rows x 64 bytes) is fetched with 4 lanes on
//     each row's 64 contiguous bytes and the 16-byte unit q of row r goes to slot q ^ ((r >> 1) & 3) -- the swizzle is applied
//     to the SOURCE address -- so that the fragment read (16 rows, same q) spreads over all banks;
//   * NST LDS stages: the loads of chunk c + NST are issued during chunk c, and a wave waits only until ITS loads of chunk
//     c + 1 have landed (s_waitcnt vmcnt(loads of the later chunks), then the workgroup barrier);
//   * blockIdx -> tile is XCD-aware: XCD x (= blockIdx % 8) owns a contiguous run of tiles in row-block-major order, i.e. a few
//     row blocks with ALL their column blocks: every activation row is pulled into ONE L2 (the weights, small, into all 8);
//     with the column block fastest across XCDs every L2 would fetch the whole activation matrix;
//   * accumulation order = the other throughput kernels' (chunks in order, components x, y, z, w per accumulator): bit-identical
//     results; the four components are issued as four sweeps over the 4 TNW independent accumulators (a dependent MFMA every
//     16 TNW issues instead of back to back: 40-cycle latency against 32-cycle issue);
//   * epilogue: the rolled loop over LDS-parked sums of k_linear_lds (small code).
// ---------------------------------------------------------------------------------------------------
#ifdef SQAIR_KNOBS
__device__ unsigned long long sq_big_phase[8];   // knob builds: phase stamps of workgroup 0 / wave 0 (tools/time_linear.py prints them)
#define SQ_BIG_STAMP(i) if (blockIdx.x == 8 && threadIdx.x == 0) sq_big_phase[i] = wall_clock64();
// shader-clock cycles of the K loop (s_memtime), next to its wall time: the clock the loop ran at
#define SQ_BIG_CYC0() const unsigned long long cyc0 = __builtin_readcyclecounter();
#define SQ_BIG_CYC1() if (blockIdx.x == 8 && threadIdx.x == 0) sq_big_phase[7] = __builtin_readcyclecounter() - cyc0;
#else
#define SQ_BIG_STAMP(i)
#define SQ_BIG_CYC0()
#define SQ_BIG_CYC1()
#endif
#ifndef SQ_BIG_NST
#define SQ_BIG_NST 4
#endif
// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt = bits 15:14 | 3:0, expcnt 6:4 and lgkmcnt 11:8 left at "no wait"); n is wave-uniform
#define SQ_VMCNT_IMM(n) (((n) & 15) | (((n) >> 4) << 14) | 0x0F70)
template <int I, int N, class F>
__device__ __forceinline__ void sq_static_for(F&& f) {   // f(integral_constant<int, I>) ... f(integral_constant<int, N - 1>), in order
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sq_static_for<I + 1, N>(f);
  }
}
template <int TMW, int TNW>   // wave tile 16 TMW rows x 16 TNW columns; workgroup tile twice that each way
__global__ __launch_bounds__(256) void k_linear_big(const LinArgs a, const int kc_total, const int n_tiles, const int n_colblk,
                                                    const int n_tiles_total, const int vec_ok SQ_TLP) {
  SQ_TL_SCOPE;
  SQ_BIG_STAMP(0)
  constexpr int ROWS = 32 * TMW, A_BYTES = ROWS * 16 * 4, B_BYTES = 2 * TNW * 1024, STAGE = A_BYTES + B_BYTES;
  constexpr int NST = SQ_BIG_NST, AHEAD = NST - 2;   // chunks whose loads may still be in flight while chunk c is multiplied
  constexpr int NPA = (2 * TMW + 3) / 4;       // 16-row activation pieces a wave stages per chunk (pieces wave, wave + 4, ...)
  constexpr int NPB = (2 * TNW + 3) / 4;       // weight tiles a wave stages per chunk
  constexpr int PARK = 4 * (2 * TNW) * 1024;   // epilogue: per wave 2 TNW parked accumulators, 16 bytes per lane each
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE > PARK ? NST * STAGE : PARK];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, l15 = lane & 15;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  // XCD-aware tile order (bijective also when the tile count is not a multiple of 8)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, qn = n_tiles_total >> 3, rn = n_tiles_total & 7;
  if (j >= qn + (xcd < rn ? 1 : 0)) return;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + j;
  const int rowblk = tile / n_colblk, colblk = tile - rowblk * n_colblk;
  const int row0 = rowblk * ROWS;
  const int tile_n0 = colblk * (2 * TNW);
  // segment table in named locals (see k_linear_mt)
  const float* sp0 = a.seg[0].p; const float* sp1 = a.seg[1].p; const float* sp2 = a.seg[2].p; const float* sp3 = a.seg[3].p;
  const int sl0 = a.seg[0].ld, sl1 = a.seg[1].ld, sl2 = a.seg[2].ld, sl3 = a.seg[3].ld;
  const unsigned sm0 = a.seg[0].rmul, sm1 = a.seg[1].rmul, sm2 = a.seg[2].rmul, sm3 = a.seg[3].rmul;
  const int w0 = a.seg[0].width, w1 = a.seg[1].width, w2 = a.seg[2].width, w3 = a.seg[3].width;
  const int c1 = (w0 + 15) >> 4;
  const int c2 = c1 + (a.nseg > 1 ? (w1 + 15) >> 4 : 0);
  const int c3 = c2 + (a.nseg > 2 ? (w2 + 15) >> 4 : 0);
  const int cum1 = a.nseg > 1 ? c1 : 0x7fffffff, cum2 = a.nseg > 2 ? c2 : 0x7fffffff, cum3 = a.nseg > 3 ? c3 : 0x7fffffff;
  const int lim0 = ((w0 + 3) & ~3) - 4, lim1 = ((w1 + 3) & ~3) - 4, lim2 = ((w2 + 3) & ~3) - 4, lim3 = ((w3 + 3) & ~3) - 4;
  // staging map of the activations: wave w issues the 16-row pieces w, w + 4, .. of a chunk; lane -> row (lane >> 2) of
  // the piece, LDS slot lane & 3, which holds the 16-byte unit q = slot ^ ((row >> 1) & 3) of the row's 64 bytes.
  // The per-chunk address work is kept to a handful of instructions (one wave per SIMD issues an instruction every ~4-5 cycles:
  // ~70 instructions of segment selection and 64-bit arithmetic per chunk were 0.3 us of every 1.2 us chunk): the row pointers
  // of the CURRENT segment live in registers and are re-derived only when a chunk crosses a segment boundary (a scalar branch),
  // the weight pointers advance by one fragment block.
  const int srow = lane >> 2;
  const int sq4 = ((lane & 3) ^ ((srow >> 1) & 3)) * 4;
  int arow[NPA];
#pragma unroll
  for (int q = 0; q < NPA; ++q) arow[q] = min(row0 + 16 * (wave + 4 * q) + srow, a.M - 1);
  const float* pa[NPA];                       // row pointers into the current segment
#pragma unroll
  for (int q = 0; q < NPA; ++q) pa[q] = nullptr;
  int seg_c0 = 0, seg_end = 0, seg_lim = 0, seg_i = -1;     // its first chunk, one past its last chunk, clamp of the k offset
  const float* wpt[(2 * TNW + 3) / 4];        // this wave's weight tiles t = wave, wave + 4, ...: pointer to the block of the NEXT chunk
#pragma unroll
  for (int q = 0; q < (2 * TNW + 3) / 4; ++q)
    wpt[q] = a.wp + ((size_t)min(tile_n0 + wave + 4 * q, n_tiles - 1) * kc_total) * 256 + (size_t)lane * 4;
  // issuing the loads of one chunk, in steps the K loop places between its MFMAs:
  //   SEG   -- (rarely) move the row pointers to the next segment of the virtually concatenated operand
  //   A(q)  -- this wave's q-th 16-row activation piece;  B(q) -- its q-th weight tile
  int ld_kk = 0;
  char* ld_st = lds;
#define SQ_BIG_SEG(G, BUF)                                                                                           \
  {                                                                                                                  \
    const int g = (G);                                                                                               \
    if (g >= seg_end) { /* next segment (wave-uniform) */                                                            \
      ++seg_i;                                                                                                       \
      const float* sp = seg_i == 0 ? sp0 : (seg_i == 1 ? sp1 : (seg_i == 2 ? sp2 : sp3));                             \
      const int sl = seg_i == 0 ? sl0 : (seg_i == 1 ? sl1 : (seg_i == 2 ? sl2 : sl3));                                \
      const unsigned sm = seg_i == 0 ? sm0 : (seg_i == 1 ? sm1 : (seg_i == 2 ? sm2 : sm3));                           \
      const int sw = seg_i == 0 ? w0 : (seg_i == 1 ? w1 : (seg_i == 2 ? w2 : w3));                                    \
      seg_c0 = seg_end; seg_end += (sw + 15) >> 4; seg_lim = ((sw + 3) & ~3) - 4;                                     \
      _Pragma("unroll") for (int q = 0; q < NPA; ++q)                                                                 \
        pa[q] = sp + (size_t)(sm ? (int)__umulhi((unsigned)arow[q], sm) : arow[q]) * sl;                              \
    }                                                                                                                \
    ld_kk = min((g - seg_c0) * 16 + sq4, seg_lim);                                                                    \
    ld_st = lds + (BUF) * STAGE;                                                                                      \
  }
  // (a piece index every wave has is not tested: the test would cost an exec-mask save / restore around the load)
#define SQ_BIG_LOAD_A(q)                                                                                              \
  if (4 * (q) + 3 < 2 * TMW || wave + 4 * (q) < 2 * TMW)                                                              \
    __builtin_amdgcn_global_load_lds((glb_ptr)(pa[q] + ld_kk), (lds_ptr)(ld_st + (wave + 4 * (q)) * 1024), 16, 0, 0);
#define SQ_BIG_LOAD_B(q)                                                                                              \
  if (4 * (q) + 3 < 2 * TNW || wave + 4 * (q) < 2 * TNW) {                                                            \
    __builtin_amdgcn_global_load_lds((glb_ptr)wpt[q], (lds_ptr)(ld_st + A_BYTES + (wave + 4 * (q)) * 1024), 16, 0, 0); \
    wpt[q] += 256;                                                                                                    \
  }
#define SQ_BIG_STAGE(G, BUF)                                                                                          \
  {                                                                                                                  \
    SQ_BIG_SEG(G, BUF)                                                                                                \
    _Pragma("unroll") for (int q = 0; q < NPA; ++q) { SQ_BIG_LOAD_A(q) }                                              \
    _Pragma("unroll") for (int q = 0; q < NPB; ++q) { SQ_BIG_LOAD_B(q) }                                              \
  }

  f32x4 acc[TMW][TNW];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int t = 0; t < TNW; ++t) acc[i][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // fragment read offsets (bytes inside a stage): activation row (16 TMW wave_m + 16 i + l15), slot kq ^ ((l15 >> 1) & 3)
  const int a_off = (16 * TMW * wave_m + l15) * 64 + ((kq ^ ((l15 >> 1) & 3)) * 16);
  const int b_off = A_BYTES + wave_n * TNW * 1024 + lane * 16;
  SQ_BIG_STAMP(1)
  // vector-memory instructions this wave issues per chunk (wave-uniform): NL_MAX for the waves that have a piece / tile of every
  // round q, NL_MIN for the others
  constexpr int NL_MAX = NPA + NPB, NL_MIN = (2 * TMW) / 4 + (2 * TNW) / 4;
  const bool ld_max = __builtin_amdgcn_readfirstlane((2 * TMW - wave + 3) / 4 + (2 * TNW - wave + 3) / 4) == NL_MAX;
#pragma unroll
  for (int g0 = 0; g0 < NST; ++g0)
    if (g0 < kc_total) SQ_BIG_STAGE(g0, g0)
  SQ_BIG_STAMP(2)
  SQ_BIG_CYC0()
  // Software pipeline inside the wave: the fragments of chunk c + 1 are read from LDS into a second register set, and the loads
  // of chunk c + NST are issued, BETWEEN the MFMAs of chunk c (which depend on registers only).  With one wave per SIMD --
  // what a launch of <= 256 workgroups gives -- nothing else fills the matrix pipe during the ~45 instructions of address work,
  // the fragment reads and their latency: issued ahead of the MFMAs they were 0.2 - 0.3 us of every chunk (0.33 us per chunk
  // on top of the MFMAs in the fit of tools/big_shapes.sh, for every tile shape; deeper prefetch alone changed nothing).
  // The placement is by hand (the loads sit behind wave-uniform branches, and hipcc does not move MFMAs across basic blocks):
  // an MFMA occupies the pipe for 32 cycles, a step below issues in about as many.
  f32x4 af0[TMW], bv0[TNW], af1[TMW], bv1[TNW];
  constexpr int NM = 4 * TMW * TNW;                 // MFMAs of a chunk, in sweeps over the accumulators (x, y, z, w)
  constexpr int GAP = NM >= 32 ? 2 : 1;             // MFMAs between two steps
#define SQ_BIG_MF(FA, FB, k)                                                                                          \
  {                                                                                                                  \
    const int comp = (k) / (TMW * TNW), i_ = ((k) % (TMW * TNW)) / TNW, t_ = (k) % TNW;                               \
    acc[i_][t_] = __builtin_amdgcn_mfma_f32_16x16x4f32(FB[t_][comp], FA[i_][comp], acc[i_][t_], 0, 0, 0);             \
  }
  // one chunk: (my loads of chunk c + 1 have landed, my reads of chunk c are in registers) -> barrier (everybody's) -> MFMAs of
  // chunk c with, between them, the reads of chunk c + 1 and the loads of chunk c + NST (into the stage chunk c was read from)
  // (k is a constant in every copy of the unrolled loop: the step tests fold)
#define SQ_BIG_CHUNK(FA, FB, GA, GB)                                                                                  \
  {                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    if (c + NST <= kc_total) {                                                                                        \
      if (ld_max) __builtin_amdgcn_s_waitcnt(SQ_VMCNT_IMM(NL_MAX * (NST - 2)));                                       \
      else __builtin_amdgcn_s_waitcnt(SQ_VMCNT_IMM(NL_MIN * (NST - 2)));                                              \
    } else __builtin_amdgcn_s_waitcnt(SQ_VMCNT_IMM(0));   /* the last NST - 1 chunks: everything */                   \
    __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */                                                              \
    __builtin_amdgcn_s_barrier();                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    const int nb = bc + 1 == NST ? 0 : bc + 1;                                                                        \
    const bool rd = c + 1 < kc_total, ld = c + NST < kc_total;                                                        \
    const char* rst = lds + nb * STAGE;                                                                               \
    _Pragma("unroll") for (int k = 0; k < NM; ++k) {                                                                  \
      SQ_BIG_MF(FA, FB, k)                                                                                            \
      const int step = (k % GAP == GAP - 1) ? k / GAP : -1;                                                           \
      if (step >= 0 && step < TMW) { if (rd) GA[step] = *reinterpret_cast<const f32x4*>(rst + a_off + step * 1024); }  \
      else if (step >= TMW && step < TMW + TNW) { if (rd) GB[step - TMW] = *reinterpret_cast<const f32x4*>(rst + b_off + (step - TMW) * 1024); } \
      else if (step == TMW + TNW) { if (ld) SQ_BIG_SEG(c + NST, bc) }                                                 \
      else if (step > TMW + TNW && step <= TMW + TNW + NPA) { if (ld) { SQ_BIG_LOAD_A(step - TMW - TNW - 1) } }       \
      else if (step > TMW + TNW + NPA && step <= TMW + TNW + NPA + NPB) { if (ld) { SQ_BIG_LOAD_B(step - TMW - TNW - NPA - 1) } } \
      if (step >= 0 && step <= TMW + TNW + NPA + NPB) __builtin_amdgcn_sched_barrier(0);                              \
    }                                                                                                                 \
    bc = nb;                                                                                                          \
  }
  static_assert((NM + GAP - 1) / GAP > TMW + TNW + NPA + NPB, "not enough MFMAs of a chunk to place the steps between");
  if (kc_total <= NST) __builtin_amdgcn_s_waitcnt(SQ_VMCNT_IMM(0));   // chunk 0 (all of them if there are that few)
  else if (ld_max) __builtin_amdgcn_s_waitcnt(SQ_VMCNT_IMM(NL_MAX * (NST - 1)));
  else __builtin_amdgcn_s_waitcnt(SQ_VMCNT_IMM(NL_MIN * (NST - 1)));
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  {
    const char* st = lds;
#pragma unroll
    for (int i = 0; i < TMW; ++i) af0[i] = *reinterpret_cast<const f32x4*>(st + a_off + i * 1024);
#pragma unroll
    for (int t = 0; t < TNW; ++t) bv0[t] = *reinterpret_cast<const f32x4*>(st + b_off + t * 1024);
  }
  int bc = 0;   // stage of chunk c
#pragma unroll 1
  for (int c = 0; c < kc_total; ++c) {
    SQ_BIG_CHUNK(af0, bv0, af1, bv1)
    if (++c >= kc_total) break;
    SQ_BIG_CHUNK(af1, bv1, af0, bv0)
  }
#undef SQ_BIG_CHUNK
#undef SQ_BIG_MF
#undef SQ_BIG_STAGE
#undef SQ_BIG_SEG
#undef SQ_BIG_LOAD_A
#undef SQ_BIG_LOAD_B
  __syncthreads();   // the last chunk's fragments have been read by every wave: the stages become the epilogue's parking space
  SQ_BIG_STAMP(3)
  SQ_BIG_CYC1()
  // Epilogue.  The MFMAs above take the WEIGHT fragment as their first operand and the activation fragment as the second, i.e.
  // they compute the transposed tile: lane (kq, l15) of accumulator (i, t) holds C[row 16 i + l15][columns 16 t + 4 kq .. + 3]
  // -- four CONSECUTIVE COLUMNS of one row (the products commute and k is summed in the same order: bit-identical sums).  So
  // every lane owns whole 16-byte pieces of output rows: 16-byte bias / addend loads and output stores, no transposition.
  // What remains is code size: a kernel starts with a cold instruction cache (~1.5 ns per byte of
  // straight-line code executed once), and 16 unrolled copies of the epilogue of one accumulator are ~6 KB.  The accumulators
  // of two row tiles at a time are therefore parked in LDS -- each lane its own 16 bytes per accumulator, read back by the
  // same lane: an indexable register file, no conflicts, no barrier -- and ONE rolled loop body handles all of them.
  sq_f32x4* park = reinterpret_cast<sq_f32x4*>(lds) + wave * (2 * TNW * 64) + lane;   // [2 TNW sums][64 lanes]
  float p_scale = *(a.scale_ptr != nullptr ? a.scale_ptr : a.bias);
  p_scale = a.scale_ptr != nullptr ? p_scale : 1.0f;
  const int wtile_n0 = tile_n0 + wave_n * TNW;
  const int nq = wtile_n0 * 16 + 4 * kq;            // column of this lane's float4 in column tile 0 of the wave
  sq_f32x4 b4[TNW];                                 // this lane's biases (the packed bias is padded to whole tiles)
#pragma unroll
  for (int t = 0; t < TNW; ++t) b4[t] = *reinterpret_cast<const sq_f32x4*>(a.bias + (size_t)min(wtile_n0 + t, n_tiles - 1) * 16 + 4 * kq);
  // fast path (wave-uniform): plain activation layer, one activation code, every float4 inside the tensor, 16-byte aligned rows
  const bool fast = a.epi == EPI_ACT && vec_ok != 0 && a.act_split >= a.N && (a.N & 3) == 0 &&
                    (a.add == nullptr || ((a.add_n & 3) == 0 && a.add_rmul == 0));
  const float sc = a.scale * p_scale;
  const int act = a.act_a;
#pragma unroll
  for (int half = 0; half < (TMW + 1) / 2; ++half) {
    constexpr int LAST = TMW - 2 * ((TMW - 1) / 2);    // row tiles of the last group (an odd TMW ends on a single one)
    const int n_ii = half == (TMW - 1) / 2 ? LAST : 2;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int t = 0; t < TNW; ++t)
        if (2 * half + ii < TMW) park[(ii * TNW + t) * 64] = acc[2 * half + ii][t];
    if (half == 0) { SQ_BIG_STAMP(5) }
    if (half == 1) { SQ_BIG_STAMP(6) }
    const int mh = row0 + wave_m * (16 * TMW) + half * 32 + l15;
    // (TNW accumulators per trip: a lone wave stalls on every LDS / address / store latency of a one-accumulator body --
    // measured 0.23 us per accumulator with an identity activation -- so the trip carries TNW independent chains)
#pragma unroll 1
    for (int ii = 0; ii < n_ii; ++ii) {
      const int m = mh + ii * 16;
      sq_f32x4 v[TNW];
#pragma unroll
      for (int t = 0; t < TNW; ++t) v[t] = park[(ii * TNW + t) * 64];
      if (m < a.M) {
        if (fast) {
          float* orow = a.out + (size_t)m * a.out_ld;
          const float* arow = a.add + (size_t)m * a.add_ld;
          sq_f32x4 x[TNW];
#pragma unroll
          for (int t = 0; t < TNW; ++t) {
            x[t] = v[t] + b4[t];
            if (a.add != nullptr && nq + t * 16 < a.add_n) x[t] += *reinterpret_cast<const sq_f32x4*>(arow + nq + t * 16);
          }
          switch (act) {   // scalar branch on a kernel argument
            case ACT_ELU: _Pragma("unroll") for (int t = 0; t < TNW; ++t) x[t] = sq_f32x4{sq_elu(x[t].x), sq_elu(x[t].y), sq_elu(x[t].z), sq_elu(x[t].w)}; break;
            case ACT_TANH: _Pragma("unroll") for (int t = 0; t < TNW; ++t) x[t] = sq_f32x4{sq_tanh(x[t].x), sq_tanh(x[t].y), sq_tanh(x[t].z), sq_tanh(x[t].w)}; break;
            case ACT_SIGMOID: _Pragma("unroll") for (int t = 0; t < TNW; ++t) x[t] = sq_f32x4{sq_sigmoid(x[t].x), sq_sigmoid(x[t].y), sq_sigmoid(x[t].z), sq_sigmoid(x[t].w)}; break;
            case ACT_SOFTPLUS_MIN: _Pragma("unroll") for (int t = 0; t < TNW; ++t) x[t] = sq_f32x4{sq_softplus(x[t].x), sq_softplus(x[t].y), sq_softplus(x[t].z), sq_softplus(x[t].w)} + 1e-2f; break;
            default: break;
          }
#pragma unroll
          for (int t = 0; t < TNW; ++t)
            if (nq + t * 16 < a.N) *reinterpret_cast<sq_f32x4*>(orow + nq + t * 16) = x[t] * sc;
        } else {
#pragma unroll
          for (int t = 0; t < TNW; ++t)
            if (nq + t * 16 < a.N) x_epilogue4(a, m, nq + t * 16, v[t], b4[t], p_scale, vec_ok != 0);
        }
      }
    }
  }
  SQ_BIG_STAMP(4)
}
#ifdef SQAIR_KNOBS
extern "C" int sqair_debug_big_phases(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sq_big_phase), sizeof(unsigned long long) * 8);
}
#endif
// Tile shape of k_linear_big.  The layer shapes of the pass put between 60 and a few hundred 128 x 64 tiles on 256 CUs, so the
// number of ROUNDS of workgroups decides more than the efficiency of a tile: 270 tiles (cfg-4's 1920 x 362 x 1152) leave 14
// CUs with two tiles while 242 wait (33.2 us; 240 tiles of 96 x 96: 24.5), 80 tiles (2560 x 256 x 256) leave two thirds of the
// chip idle (16.9 us; 160 tiles of 64 x 64: 11.1).  Every shape accumulates a given output in the same order (bit-identical).
// The choice minimises a cost model fitted to tools/big_shapes.sh (14 layer shapes x 8 tile shapes, back to back, MI355X; rms
// 0.5 us on the single-round launches): one round of a wave tile of u = TMW TNW MFMA tiles over kc K chunks takes
//   T1 = 1.92 + 0.297 kc + 0.0736 kc u - 0.069 u (+ 0.175 u with a non-linear activation)   [us]
// and r = ceil(tiles / 256) rounds take T1 (1 + (0.335 + 0.0342 u)(r - 1)) -- a second workgroup on a CU hides part of the
// first one's stalls, less so the larger the tile.  With the four shapes kept, the model's picks sum to 361 us over the 14
// shapes against 353 for the best of all shapes per layer and 388 for 128 x 64 everywhere.
static int pick_big_shape(int M, int n_tiles, int kc, bool nonlinear) {
  static const int cand[4][2] = {{4, 2}, {3, 3}, {3, 2}, {2, 2}};
  int best = 42;
  float best_t = 1e30f;
  for (int i = 0; i < 4; ++i) {
    const int tm = cand[i][0], tn = cand[i][1], u = tm * tn;
    const int tiles = ((M + 32 * tm - 1) / (32 * tm)) * ((n_tiles + 2 * tn - 1) / (2 * tn));
    const float t1 = 1.92f + 0.297f * kc + 0.0736f * kc * u - 0.069f * u + (nonlinear ? 0.175f * u : 0.0f);
    const float t = t1 * (1.0f + (0.335f + 0.0342f * u) * (float)((tiles + 255) / 256 - 1));
    if (t < best_t) { best_t = t; best = 10 * tm + tn; }
  }
  return best;
}

template <int TMW, int TNW>
static void launch_big(const LinArgs& a, const PackedLayer& L, hipStream_t s) {
  const int n_colblk = (L.nt + 2 * TNW - 1) / (2 * TNW), n_rowblk = (a.M + 32 * TMW - 1) / (32 * TMW);
  const int total = n_colblk * n_rowblk;
  // 16-byte epilogue accesses need 16-byte aligned rows in every tensor the epilogue touches
  auto al = [](const void* p, int ld) { return p == nullptr || (((uintptr_t)p & 15) == 0 && (ld & 3) == 0); };
  const int vec = al(a.out, a.out_ld) && al(a.add, a.add_ld) && al(a.e0, a.e0_ld) && al(a.e1, a.e1_ld) && al(a.o1, a.o1_ld) &&
                  al(a.o2, a.o2_ld) && al(a.o3, a.o3_ld) && (a.epi == EPI_ACT || (a.nh & 3) == 0);
  // (measured and dropped: two chunks per stage and barrier, with all fragments of the stage requested before its first MFMA
  // -- the K loop of a lone workgroup stays at 0.66 us per chunk against 0.435 us of MFMAs: so neither the barrier nor the LDS latency
  // ahead of the first MFMA is what it pays)
  // (measured and dropped: four extra LOADER waves per workgroup, one per SIMD, issuing every LDS-DMA load so that the multiply
  // waves only read fragments and issue MFMAs -- a lone workgroup's K loop 10.8 -> 11.4 us, i.e. the load ISSUE is not what
  // it pays either; the 0.23 us per chunk beyond the MFMAs are still unexplained)
  // (measured and dropped: starting the workgroups in odd hardware wave slots half a CU-load of matrix work late, so that their
  // K loops cover the others' epilogues -- 51200 x 256 x 256: 91 -> 84 us at half the computed delay with the 128 x 64 tile,
  // slower in every other combination tried)
  SQ_LAUNCH((k_linear_big<TMW, TNW>), dim3(8 * ((total + 7) / 8)), dim3(256), 0, s, a, L.kc, L.nt, n_colblk, total, vec);
}

// Tile shape of the throughput variants, from measurements of the layer shapes of the pass (tools/time_linear.py, MI355X):
// slabs per wave (1 or 2; two ROW tiles per wave -- template parameter MT = 2 -- measured slower on every shape of the pass and is
// not instantiated) and whether the A fragment is loaded row-contiguously (COAL).  SQAIR_MT="1,NT[,COAL]" overrides.
struct MtShape { int mt, nt, coal; };
static MtShape pick_mt_shape(int M, int n_tiles, int kc) {
  static int ov_mt = -1, ov_nt = -1, ov_coal = -1;
  if (ov_mt < 0) {
    ov_mt = ov_nt = 0;
    const char* e = SQ_KNOB_STR("SQAIR_MT");
    if (e != nullptr) sscanf(e, "%d,%d,%d", &ov_mt, &ov_nt, &ov_coal);
  }
  (void)kc;
  MtShape sh;
  if (M >= 2048) sh = MtShape{1, (M >= 16384 && n_tiles >= 2) ? 2 : 1, 1};
  else if (n_tiles >= 64) sh = MtShape{1, 1, 1};
  else if (n_tiles >= 32) sh = MtShape{1, 2, 0};
  else sh = MtShape{1, 1, 0};
  if (ov_mt > 0 && ov_nt > 0) { sh.mt = ov_mt; sh.nt = ov_nt; }
  if (ov_coal >= 0) sh.coal = ov_coal;
  return sh;
}

template <int NCH, int MT, int NT>
static void launch_mt(const LinArgs& a, const PackedLayer& L, int mt, bool coal, hipStream_t s, unsigned long long* prof_ts) {
  const dim3 g((L.nt + NT - 1) / NT, (mt + 4 * MT - 1) / (4 * MT));
  if (coal) SQ_LAUNCH((k_linear_mt<NCH, MT, NT, true>), g, dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
  else SQ_LAUNCH((k_linear_mt<NCH, MT, NT, false>), g, dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
}

template <int NCH, int NSEG>
static void launch_seg(const LinArgs& a, const PackedLayer& L, hipStream_t s, unsigned long long* prof_ts) {
  const dim3 g(L.nt, (a.M + 15) / 16);
  // the GRU gate epilogues are compiled out of the instantiation the plain layers use (the slot loop rotates through ~10 code
  // objects; the smaller they are, the more of them stay in the instruction cache)
#define SQ_LAUNCH_KL(G, P) SQ_LAUNCH((k_linear<NCH, NSEG, G, P>), g, dim3(256), 0, s, a.seg[0].p, a.wp, a.seg[0].ld, a.seg[0].width, a.seg[0].rmul, a.M, L.kc, L.nt, a.wzero, a, prof_ts)
  // (likewise the device-clock stamps of the profiling pass exist only in the instantiations that pass launches)
  if (prof_ts == nullptr) {
    if (a.epi == EPI_ACT) SQ_LAUNCH_KL(false, false); else SQ_LAUNCH_KL(true, false);
  } else {
    if (a.epi == EPI_ACT) SQ_LAUNCH_KL(false, true); else SQ_LAUNCH_KL(true, true);
  }
#undef SQ_LAUNCH_KL
}
template <int NCH>
static void launch_nch(const LinArgs& a, const PackedLayer& L, int grid, hipStream_t s, unsigned long long* prof_ts) {
  (void)grid;
  switch (a.nseg) {  // the segment count is a template parameter: dead segment-selection code disappears
    case 1: launch_seg<NCH, 1>(a, L, s, prof_ts); break;
    case 2: launch_seg<NCH, 2>(a, L, s, prof_ts); break;
    case 3: launch_seg<NCH, 3>(a, L, s, prof_ts); break;
    default: launch_seg<NCH, 4>(a, L, s, prof_ts); break;
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
  // Launches with many thousands of rows: the throughput variants.  Below that the split-K kernel (one 16 x 16 tile per
  // workgroup, 4 waves on the K range) is the faster one IN THE PASS on every configuration measured -- its time barely moves
  // between 160 and 1920 rows (4.7 -> 5.0 us), while the macro-tile kernels, ahead of it in back-to-back timing of one shape,
  // lost 1 % of the forward pass at 32 sequences per GPU, 11 % at 64 and at cfg-4 (they were used from 256 rows up until round 2).
  // The boundary was 2048 rows until round 5 and is 6000 now (tools/ab_libs.py on product builds with the boundary at 2048 /
  // 4096 / 5200 / 7000 / 20000 / never): 128 sequences per GPU (2560-row layers) forward 5.85 -> 5.43 ms, training 14.00 ->
  // 13.60; 256 sequences (5120 rows) 8.33 -> 8.10 / 21.63 -> 21.37; from 6400 rows on the LDS-tiled kernel is ahead (cfg-2's and
  // cfg-4's once-per-pass layers: 20000 costs cfg-4 0.05 ms, cfg-2 0.007).  Back to back the split-K kernel does
  // 2560 x 256 x 256 in 8.4 us (the LDS-tiled kernel's best tile 11.1), 5120 x 256 x 256 in 15.2 (15.1).
  // Also tried for the 2048+ row launches and dropped: the split-K structure with a 32 x 32 and with a 64 x 64 workgroup tile
  // (half / a quarter of the operand bytes per output tile): 22 - 25 us against 16 - 17 us on 5120 x 256 x 256, back to back.
#ifndef SQAIR_MT_ROWS_DEFAULT
#define SQAIR_MT_ROWS_DEFAULT 6000
#endif
  static const int mt_rows = SQ_KNOB_INT("SQAIR_MT_ROWS", SQAIR_MT_ROWS_DEFAULT);  // measurement knob (tools/build_rev.sh WT <name> -D...)
  // just below 2048 rows only the WIDE layers go to the LDS-tiled kernel (its 128 x 64 tiles then still number ~200):
  // tools/time_linear.py, back to back, 1920 x 362 x 1152 35.7 -> 33.2 us, 1920 x 312 x 768 21.2 -> 19.2 (every 256 / 400-column
  // layer would be twice as slow; at 1280 rows 362 x 1152 gains back to back, 24.3 -> 21.1, but not in the pass, and
  // 312 x 768 loses, 14.7 -> 18.4).  cfg-4 (1920 rows per frame): forward 6.13 -> 6.07 ms, training 14.16 -> 14.12.
  const bool mid_wide = prof_ts == nullptr && L.kc > 4 && a.M >= 1792 && L.nt >= 48;
  if (a.M >= mt_rows || mid_wide) {
    // (all of them accumulate in the same order: the tile shape never changes a result)
    if (L.kc <= 4) {  // K <= 64: one block of loads, nothing to pipeline
      const dim3 grid_r(L.nt, (mt + 3) / 4);
      SQ_LAUNCH(k_linear_rows<4>, grid_r, dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
      return 0;
    }
    // the LDS-tiled kernel pays once its 128 x 64 workgroup tiles fill the chip twice over (measured, tools/time_linear.py:
    // 51200 x 256 x 256 118 -> 106 us, 5120 x 362 x 1152 80 -> 67 us; below that the macro-tile kernel's smaller tiles win)
    // both operands through LDS (k_linear_big): ahead of the two kernels below on every shape of the pass with >= 4 column
    // tiles (tools/time_linear.py, back to back: 5120 x 362 x 1152 70 -> 52 us, 51200 x 256 x 256 113 -> 88, 6400 x 256 x 400
    // 28 -> 25, 5120 x 256 x 256 18 -> 17.4); the 128 x 64 tile (TNW = 2) beats 128 x 128 wherever the tile count is what
    // limits (all of these shapes: 80 - 1600 tiles on 256 CUs)
    static const int big = SQ_KNOB_INT("SQAIR_BIG", 2);  // measurement knob: 0 = off
    if (big > 0 && prof_ts == nullptr && L.nt >= 4) {
#ifdef SQAIR_KNOBS
      if (const char* e = SQ_KNOB_STR("SQAIR_BIG_SHAPE")) {   // "TMW,TNW": forced tile (measurement)
        const int tm = e[0] - '0', tn = e[2] - '0';
#define SQ_BIG_CASE(A, B) if (tm == A && tn == B) { launch_big<A, B>(a, L, s); return 0; }
        SQ_BIG_CASE(1, 2) SQ_BIG_CASE(1, 3) SQ_BIG_CASE(1, 4)
        SQ_BIG_CASE(2, 2) SQ_BIG_CASE(2, 3) SQ_BIG_CASE(2, 4) SQ_BIG_CASE(3, 2) SQ_BIG_CASE(3, 3) SQ_BIG_CASE(3, 4)
        SQ_BIG_CASE(4, 2) SQ_BIG_CASE(4, 3) SQ_BIG_CASE(4, 4)
#undef SQ_BIG_CASE
      }
#endif
      switch (pick_big_shape(a.M, L.nt, L.kc, a.act_a != ACT_NONE)) {
        case 22: launch_big<2, 2>(a, L, s); break;
        case 32: launch_big<3, 2>(a, L, s); break;
        case 33: launch_big<3, 3>(a, L, s); break;
        default: launch_big<4, 2>(a, L, s); break;
      }
      return 0;
    }
    static const int lds_wgs = SQ_KNOB_INT("SQAIR_LDS_WGS", 512);  // measurement knob
    if (((a.M + 127) / 128) * ((L.nt + 3) / 4) >= lds_wgs) {
      SQ_LAUNCH((k_linear_lds<2>), dim3((L.nt + 3) / 4, (a.M + 127) / 128), dim3(256), 0, s, a, L.kc, L.nt, prof_ts);
      return 0;
    }
    const MtShape sh = pick_mt_shape(a.M, L.nt, L.kc);
    if (sh.nt >= 2) launch_mt<4, 1, 2>(a, L, mt, sh.coal != 0, s, prof_ts);
    else launch_mt<4, 1, 1>(a, L, mt, sh.coal != 0, s, prof_ts);
    return 0;
  }
  const int per_wave = (L.kc + 3) / 4;
  // hundreds of rows, several column tiles and a DEEP K (>= 640: the input encoder; in the backward pass the transposes of the
  // wide once-per-frame layers): 32 x 32 tiles with two blocks of operand loads in flight (sqair_linear_t2.inc).  Measured
  // back to back (tools/time_linear.py): 640 x 1152 x 384 17.9 -> 12.1 us, 640 x 768 x 320 17.3 -> 8.8; at K <= 400 the
  // 16 x 16 tile is as fast or faster (640 x 362 x 1152 13.6 / 13.2, 640 x 312 x 768 8.6 / 9.5) and keeps those layers
  static const int t2_rows = SQ_KNOB_INT("SQAIR_T2_ROWS", 512), t2_kc = SQ_KNOB_INT("SQAIR_T2_KC", 40);  // measurement knobs
  if (a.M >= t2_rows && L.nt >= 4 && L.kc >= t2_kc && prof_ts == nullptr) {
    launch_t2_any(a, L, s);
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
