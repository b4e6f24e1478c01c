// In-launch slot chain: device bodies + the persistent kernel + the host recorder (see sqair_chain.h for the protocol).
// The bodies restate k_linear (sqair_linear_kernel.inc), x_crop_row (sqair_rowops.h), tail_body / k_rnn_tail (sqair_glue.hip)
// with the SAME arithmetic in the SAME order -- results are bit-identical to the launch-per-op path -- and one difference: every
// operand another op of the chain may have produced is fetched by a polled, L1-bypassing load.
#include "sqair_chain.h"
#include "sqair_internal.h"

#include <cstring>
#include <list>
#include <vector>

#ifndef SQAIR_WIDE
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int CH_ZLD = 68;   // == ZLD of sqair_glue.hip (z-record tile row pitch in LDS)

// ---- dependent loads: the workspace as one buffer resource; aux 16 = sc1 (bypass this CU's L1, served by the XCD's L2)
#define CH_RSRC __amdgpu_buffer_rsrc_t
// The op table is read from its LDS copy (a scalar load of a table field from memory is a ~0.5 us round trip, and an op body
// makes three or four dependent rounds of them: measured 3.3 - 7 us per item with the table in memory).  LDS reads land in
// VGPRs; CH_UNI puts the fields that steer control flow back into SGPRs (every lane holds the same value).
__device__ __forceinline__ int ch_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned ch_uni(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }   // (the builtin
// returns int: OR-ing it into a 64-bit address sign-extends a low word >= 2^31 over the high word -- a wild pointer for half of
// all buffer addresses)
#define CH_UNI(x) ch_uni(x)
__device__ __forceinline__ unsigned ch_off(const char* wsb, const void* p) { return (unsigned)((const char*)p - wsb); }
__device__ __forceinline__ u32x4_t ch_l4(CH_RSRC rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16); }
__device__ __forceinline__ unsigned ch_l1(CH_RSRC rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 16); }
__device__ __forceinline__ unsigned ch_bad4(u32x4_t v) { return (unsigned)(v.x == SQ_SENT) | (unsigned)(v.y == SQ_SENT) | (unsigned)(v.z == SQ_SENT) | (unsigned)(v.w == SQ_SENT); }
__device__ __forceinline__ f32x4_t ch_f4(u32x4_t v) {
  return f32x4_t{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
#ifdef SQAIR_KNOBS
// per-item trace of the knob build (tools/chain_trace.py): 8 x u64 per item -- {op, tile, sub, xcc, rank, kind} + seven stamps of the
// 100 MHz clock -- in the item's own slot (launch, workgroup, item ordinal): no atomics in the loop
__device__ unsigned long long* g_chain_trace = nullptr;
__device__ unsigned g_chain_trace_cap = 0;
__device__ __forceinline__ unsigned long long ch_clock() { return __builtin_amdgcn_s_memrealtime(); }
constexpr int CH_LDS_TR = 16 * 68 + 64 + 2 * 3072;   // eight stamps of the current item behind the ops' scratch
#define CH_TRACE_T(i) do { extern __shared__ __attribute__((aligned(16))) float ch_smem_[]; \
    if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(ch_smem_ + CH_LDS_TR)[i] = ch_clock(); } while (0)
#else
#define CH_TRACE_T(i)
#endif
// A pointer read from the LDS copy of the table is a generic pointer to the compiler: it is dereferenced with FLAT instructions,
// which count on lgkmcnt as well -- every later wait for an LDS read then also waits for the weight / operand loads in flight
// (measured: the A-operand polls of a dense item went out a memory round trip late).  The bodies therefore cast every such
// pointer to the global address space ONCE (typed pointers: a cast through a generic pointer is folded away again) and all
// accesses through it are global_load / global_store.
typedef __attribute__((address_space(1))) float ch_gf;
typedef const __attribute__((address_space(1))) float ch_gcf;
typedef const __attribute__((address_space(1))) f32x4_t ch_gcf4;
#define CH_GLB(T, p) ((T*)(p))
#define CH_POLL_BEGIN { int ch_spins = 0; bool ch_ok; do { ch_ok = true; unsigned ch_bad = 0;
#define CH_POLL_END(status)                                                                         \
    ch_ok = __all(ch_ok && ch_bad == 0);                                                            \
    if (!ch_ok) {                                                                                   \
      __builtin_amdgcn_s_sleep(1);                                                                  \
      if ((ch_spins & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break; \
    }                                                                                               \
  } while (!ch_ok && ++ch_spins < SQ_CHAIN_SPIN_LIMIT);                                             \
  if (!ch_ok && (threadIdx.x & 63) == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Poll loops written out (polls issued once AHEAD of the item's other preparations, which then run in the shadow of the polls' round
// trip): CH_RETRY(bad, status, REISSUE) at the bottom of `for (int spins = 0;;) { check -> bad; CH_RETRY(...) }`.
#define CH_RETRY(bad, status, REISSUE)                                                                                       \
  if (__all((bad) == 0)) break;                                                                                              \
  if (++spins >= SQ_CHAIN_SPIN_LIMIT ||                                                                                      \
      ((spins & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {               \
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                 \
    break;                                                                                                                   \
  }                                                                                                                          \
  __builtin_amdgcn_s_sleep(1);                                                                                               \
  REISSUE;                                                                                                                   \
  __builtin_amdgcn_sched_barrier(0);

// ------------------------------------------------------------------------------------------------
// dense layer: TN (1..3) consecutive column tiles of ONE row tile -- the A operand is fetched (polled) once for all of them.
// k_linear's arithmetic per 16 x 16 output tile: waves split the K chunks g = wave + 4 j, two accumulators, LDS reduce in wave
// order, the same epilogues.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ch_st(CH_RSRC rs, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)off, 0, 0); }
template <int NCH, int TN>
__device__ __forceinline__ void chain_dense(const ChDense& a, CH_RSRC rs, const int tile_m, const int tile_n0, const int n_tiles, const int M,
                                            float* red, unsigned* status) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, l15 = lane & 15;
  // The A-operand polls go out FIRST: their addresses need nothing but the chunk table (padded to SQ_CHAIN_MAX_KC entries, so the
  // slots past this layer's K meet a valid chunk -- against zero weights), and everything else an item has to prepare (the
  // descriptor's fields, the weight requests, the epilogue's operands) then runs in the shadow of the polls' round trip.  One
  // wave per SIMD hides nothing by itself: with the preparation ahead of the polls an item took 0.4 us longer.
  const unsigned arow = (unsigned)min(tile_m * 16 + l15, M - 1);
  unsigned aoff[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const ChChunk c = a.chunk[wave + 4 * j];
    aoff[j] = c.base + min((unsigned)kq * 16u, c.lim) + arow * c.ldb;
  }
  u32x4_t ar[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) ar[j] = ch_l4(rs, aoff[j]);
  __builtin_amdgcn_sched_barrier(0);
  CH_TRACE_T(1);
  const int kc_total = CH_UNI(a.kc_total), epi = CH_UNI(a.epi), N = CH_UNI(a.N), nh = CH_UNI(a.nh);
  const int act_a = CH_UNI(a.act_a), act_b = CH_UNI(a.act_b), act_split = CH_UNI(a.act_split), add_n = CH_UNI(a.add_n);
  const unsigned add_base = CH_UNI(a.add_off), o3_off = CH_UNI(a.o3_off), o1_off = CH_UNI(a.o1_off);
  const unsigned out_off = a.out_off, o2_off = a.o2_off, e0_off = a.e0_off, e1_off = a.e1_off;
  const int out_ld = a.out_ld, o1_ld = a.o1_ld, o2_ld = a.o2_ld, o3_ld = a.o3_ld, add_ld = a.add_ld, e0_ld = a.e0_ld, e1_ld = a.e1_ld;
  const float scale = a.scale;
  ch_gcf4* __restrict__ wp0 = CH_GLB(ch_gcf4, ((unsigned long long)CH_UNI(a.wp_hi) << 32) | CH_UNI(a.wp_lo)) + lane;
  ch_gcf4* __restrict__ wz = CH_GLB(ch_gcf4, ((unsigned long long)CH_UNI(a.wz_hi) << 32) | CH_UNI(a.wz_lo)) + lane;
  ch_gcf* __restrict__ biasp = CH_GLB(ch_gcf, ((unsigned long long)CH_UNI(a.bias_hi) << 32) | CH_UNI(a.bias_lo));
  const int nmine = (kc_total - wave + 3) >> 2;
  f32x4_t bv[TN][NCH];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    ch_gcf4* wp = wp0 + (size_t)(min(tile_n0 + tn, n_tiles - 1) * kc_total) * 64;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const bool valid = j < nmine;
      const int g = valid ? wave + 4 * j : wave;
#ifdef SQAIR_CHAIN_ABL_NOW
      bv[tn][j] = *wz;          // ablation (timing only): every weight chunk = the (cache-resident) zero block
#else
      bv[tn][j] = *(valid ? wp + (size_t)g * 64 : wz);
#endif
    }
  }
  // epilogue operands
  const int m = tile_m * 16 + (tid >> 4);
  const int mc = min(m, M - 1);
  const bool g2 = epi == EPI_GRU2;
  int ncol[TN];
  float p_bias[TN];
  bool use_add[TN];
  unsigned off_add[TN], off_e0[TN], off_e1[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    ncol[tn] = (tile_n0 + tn) * 16 + (tid & 15);
    const int nc = min(ncol[tn], N - 1);
    p_bias[tn] = biasp[nc];
    use_add[tn] = add_base != SQ_CHAIN_NONE && nc < add_n;
    const bool g1 = epi == EPI_GRU1 && nc >= nh && nc < 2 * nh;
    off_add[tn] = use_add[tn] ? add_base + (unsigned)(mc * add_ld + nc) * 4u : aoff[0];
    off_e0[tn] = g1 ? e0_off + (unsigned)(mc * e0_ld + (nc - nh)) * 4u : (g2 ? e0_off + (unsigned)(mc * e0_ld + nc) * 4u : aoff[0]);
    off_e1[tn] = g2 ? e1_off + (unsigned)(mc * e1_ld + nc) * 4u : aoff[0];
  }
  f32x4_t av[NCH];
  float p_add[TN], p_e0[TN], p_e1[TN];
  CH_TRACE_T(2);
  {
    unsigned xa[TN], x0[TN], x1[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { xa[tn] = ch_l1(rs, off_add[tn]); x0[tn] = ch_l1(rs, off_e0[tn]); x1[tn] = ch_l1(rs, off_e1[tn]); }
    __builtin_amdgcn_sched_barrier(0);
    int spins = 0;
    for (;;) {
      unsigned bad = 0;
#pragma unroll
      for (int j = 0; j < NCH; ++j) { bad |= ch_bad4(ar[j]); av[j] = ch_f4(ar[j]); }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        bad |= (unsigned)(xa[tn] == SQ_SENT) | (unsigned)(x0[tn] == SQ_SENT) | (unsigned)(x1[tn] == SQ_SENT);
        p_add[tn] = __uint_as_float(xa[tn]); p_e0[tn] = __uint_as_float(x0[tn]); p_e1[tn] = __uint_as_float(x1[tn]);
      }
      if (__all(bad == 0)) break;
      if (++spins >= SQ_CHAIN_SPIN_LIMIT ||
          ((spins & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        if ((tid & 63) == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int j = 0; j < NCH; ++j) ar[j] = ch_l4(rs, aoff[j]);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) { xa[tn] = ch_l1(rs, off_add[tn]); x0[tn] = ch_l1(rs, off_e0[tn]); x1[tn] = ch_l1(rs, off_e1[tn]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  CH_TRACE_T(3);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    f32x4_t acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[tn][j].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[tn][j].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[tn][j].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[tn][j].w, acc1, 0, 0, 0);
    }
    float* r = red + tn * 1024 + wave * 256;
    r[(4 * kq + 0) * 16 + l15] = acc0.x + acc1.x;
    r[(4 * kq + 1) * 16 + l15] = acc0.y + acc1.y;
    r[(4 * kq + 2) * 16 + l15] = acc0.z + acc1.z;
    r[(4 * kq + 3) * 16 + l15] = acc0.w + acc1.w;
  }
  CH_TRACE_T(4);
  __syncthreads();
  CH_TRACE_T(5);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = ncol[tn];
    if (m < M && n < N && tile_n0 + tn < n_tiles) {
      const float* rt = red + tn * 1024;
      float v = rt[tid] + rt[256 + tid] + rt[512 + tid] + rt[768 + tid] + p_bias[tn] + (use_add[tn] ? p_add[tn] : 0.0f);
      if (epi == EPI_ACT) {
        if (act_split >= N) v = sq_act(v, act_a);    // one activation for the whole layer: a scalar branch, one case executed
        else v = sq_act(v, n < act_split ? act_a : act_b);
        ch_st(rs, out_off + (unsigned)(m * out_ld + n) * 4u, v * scale * 1.0f);
      } else if (epi == EPI_GRU1) {
        if (n < nh) ch_st(rs, out_off + (unsigned)(m * out_ld + n) * 4u, sq_sigmoid(v));
        else if (n < 2 * nh) {
          const float rg = sq_sigmoid(v);
          ch_st(rs, o1_off + (unsigned)(m * o1_ld + (n - nh)) * 4u, rg * p_e0[tn]);
          if (o3_off != SQ_CHAIN_NONE) ch_st(rs, o3_off + (unsigned)(m * o3_ld + (n - nh)) * 4u, rg);
        }
        else ch_st(rs, o2_off + (unsigned)(m * o2_ld + (n - 2 * nh)) * 4u, v);
      } else {
        const float hc = sq_tanh(v);
        ch_st(rs, out_off + (unsigned)(m * out_ld + n) * 4u, sq_gru_blend(p_e1[tn], p_e0[tn], hc));
        if (o1_off != SQ_CHAIN_NONE) ch_st(rs, o1_off + (unsigned)(m * o1_ld + n) * 4u, hc);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// crop of one particle row with the transform's output layer and the where sample in front (x_crop_row, modes PROP2 / DISC with
// the fused 256 -> 8 layer).  Dependent operand: the transform's hidden layer t2.
// ------------------------------------------------------------------------------------------------
template <bool stage_img>
__device__ __forceinline__ void chain_crop(const CropArgs& a, const POff& po, const Dims& d, CH_RSRC rs, const char* wsb, const int r,
                                           float* smem, unsigned* status) {
  float* coord_s = smem;
  float* tab_s = smem + 4;
  float* img_s = smem + 4 + 4 * d.G;
  const int tid = threadIdx.x, b = sq_div(r, d.k_mul);
  const int P = d.H * d.W, G = d.G, G2 = d.G * d.G;
  const int slot = CH_UNI(a.slot), mode = CH_UNI(a.mode);
  const int mrow_add = a.mask_row_add;
  const int orow_add = a.out_row_add;
  ch_gcf* __restrict__ img = CH_GLB(ch_gcf, a.img) + (size_t)b * d.P4;
  const bool has_mask = a.mask != nullptr;
  ch_gcf* __restrict__ a_mask = CH_GLB(ch_gcf, a.mask);
  ch_gcf* __restrict__ a_noise = CH_GLB(ch_gcf, a.noise);
  ch_gcf* __restrict__ a_flat = CH_GLB(ch_gcf, a.flat);
  ch_gcf* __restrict__ a_rec_prev = CH_GLB(ch_gcf, a.rec_prev);
  ch_gcf* __restrict__ a_w3 = CH_GLB(ch_gcf, a.w3);
  ch_gf* __restrict__ a_out = CH_GLB(ch_gf, a.out);
  ch_gf* __restrict__ a_rec_new = CH_GLB(ch_gf, a.rec_new);
  ch_gf* __restrict__ a_tp_out = CH_GLB(ch_gf, a.tp_out);
  constexpr int IPT = 10;
  float v0[IPT];
  constexpr int MPT = 2;
  float mk0[MPT];
  const int hl = tid, ci = hl & 3;
  const int per = d.nh / 32;
  float tp_loc = 0.0f, tp_raw = 0.0f;
  float e[4] = {0.0f, 0.0f, 0.0f, 0.0f}, zp = 0.0f, off = 0.0f, chv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  constexpr int QM = 2;
  const int nq = per / 4;
  f32x4_t xv[QM];
  f32x4_t wv[QM][4][2];
  if (tid < 32) {
    ch_gcf* eps = a_noise + (((size_t)r * 2 + (mode == CROP_DISC ? 1 : 0)) * d.N + slot) * d.nzw;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) e[jj] = eps[jj];
    if (mode == CROP_DISC) {
      off = a_flat[po.disc_scale_offset];
    } else {
      off = a_flat[po.prop_scale_offset];
      zp = a_rec_prev[((size_t)r * d.N + slot) * rec::W + rec::WHERE + ci];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {   // tril4 (sqair_common.h) on the global pointer
        const int tq = ci * 4 + min(jj, ci);
        chv[jj] = (a_flat + po.cholesky)[tq < 6 ? 4 + tq : 15 - tq];
      }
    }
    ch_gcf4* w4 = (ch_gcf4*)(a_w3) + (size_t)per * hl * 2;
#pragma unroll
    for (int q = 0; q < QM; ++q) {
      const int qc = min(q, nq - 1);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) { wv[q][ii][0] = w4[(qc * 4 + ii) * 2]; wv[q][ii][1] = w4[(qc * 4 + ii) * 2 + 1]; }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (stage_img) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) v0[q] = img[min(q * 256 + tid, P - 1)];
  }
#pragma unroll
  for (int q = 0; q < MPT; ++q)
    mk0[q] = has_mask ? a_mask[((size_t)r * a.mask_row_mul + mrow_add) * G2 + min(tid + 256 * q, G2 - 1)] : 1.0f;
  __builtin_amdgcn_sched_barrier(0);
  if (tid < 32) {
    const unsigned xoff = ch_off(wsb, a.t2) + (unsigned)(r * a.t2_ld + per * hl) * 4u;
    CH_POLL_BEGIN
      u32x4_t xr[QM];
#pragma unroll
      for (int q = 0; q < QM; ++q) xr[q] = ch_l4(rs, xoff + 16u * (unsigned)min(q, nq - 1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < QM; ++q) { ch_bad |= ch_bad4(xr[q]); xv[q] = ch_f4(xr[q]); }
    CH_POLL_END(status)
    float part[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < QM; ++q) {
      if (q < nq) {
        const f32x4_t x = xv[q];
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const f32x4_t wa = wv[q][ii][0], wb2 = wv[q][ii][1];
          part[0] += xs[ii] * wa.x; part[1] += xs[ii] * wa.y; part[2] += xs[ii] * wa.z; part[3] += xs[ii] * wa.w;
          part[4] += xs[ii] * wb2.x; part[5] += xs[ii] * wb2.y; part[6] += xs[ii] * wb2.z; part[7] += xs[ii] * wb2.w;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) part[o] = sq_half_sum(part[o]) + a_w3[d.nh * 8 + o];
    tp_loc = ci == 0 ? part[0] : (ci == 1 ? part[1] : (ci == 2 ? part[2] : part[3]));
    tp_raw = ci == 0 ? part[4] : (ci == 1 ? part[5] : (ci == 2 ? part[6] : part[7]));
    if (a.tp_out != nullptr && hl < 4) {
      a_tp_out[(size_t)r * a.tp_out_ld + ci] = tp_loc;
      a_tp_out[(size_t)r * a.tp_out_ld + 4 + ci] = tp_raw;
    }
    float wl, loc, sc;
    if (mode == CROP_DISC) {
      loc = tp_loc;
      sc = sq_softplus(tp_raw + off) + 1e-2f;
      wl = loc + sc * (ci == 0 ? e[0] : (ci == 1 ? e[1] : (ci == 2 ? e[2] : e[3])));
    } else {
      loc = zp + 1.0f * tp_loc;
      sc = sq_softplus(tp_raw + off - 1.0f) + 1e-2f;
      float acc = 0.0f;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (jj <= ci) acc += (chv[jj] * sc + (jj == ci ? sc : 0.0f)) * e[jj];
      wl = loc + acc;
    }
    if (hl < 4) {
      ch_gf* rn = a_rec_new + ((size_t)r * d.N + slot) * rec::W;
      rn[rec::WHERE + ci] = wl;
      rn[rec::WHERE_LOC + ci] = loc;
      rn[rec::WHERE_SCALE + ci] = sc;
      coord_s[ci] = (ci & 2) ? tanhf(wl) : fmaxf(sq_sigmoid_geo(wl), 1e-4f);
    }
  }
  if (stage_img) {
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int idx = q * 256 + tid;
      if (idx < P) img_s[idx] = v0[q];
    }
    for (int idx = 256 * IPT + tid; idx < P; idx += 256) img_s[idx] = img[idx];
  }
  __syncthreads();
  const float* __restrict__ src = img_s;   // (read only when stage_img)
  for (int i = tid; i < 2 * G; i += 256) {
    const bool is_y = i >= G;
    const int j = is_y ? i - G : i;
    const float gn = -1.0f + 2.0f * (float)j / (float)(G - 1);
    const float sc = coord_s[is_y ? 1 : 0], tr = coord_s[is_y ? 3 : 2];
    const float L = (float)((is_y ? d.H : d.W) - 1);
    const float x = 0.5f * L * (sc * gn + tr + 1.0f);
    const float x0 = floorf(x);
    tab_s[i * 2 + 0] = x0;
    tab_s[i * 2 + 1] = x - x0;
  }
  __syncthreads();
  if (!stage_img) {
    constexpr int PX = 2;
    for (int p0 = tid; p0 < G2; p0 += 256 * PX) {
      float tv[PX][4], tw[PX][4], mk[PX];
#pragma unroll
      for (int u = 0; u < PX; ++u) {
        const int pix = min(p0 + 256 * u, G2 - 1);
        mk[u] = has_mask ? a_mask[((size_t)r * a.mask_row_mul + mrow_add) * G2 + pix] : 1.0f;
        const int i = sq_div(pix, d.g_mul), j = pix - i * G;
        const float x0f = tab_s[j * 2], wx1 = tab_s[j * 2 + 1];
        const float y0f = tab_s[(G + i) * 2], wy1 = tab_s[(G + i) * 2 + 1];
        const int x0 = (int)x0f, y0 = (int)y0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int yy = y0 + dy, xx = x0 + dx;
            const bool ok = yy >= 0 && yy < d.H && xx >= 0 && xx < d.W;
            tv[u][dy * 2 + dx] = img[min(max(yy, 0), d.H - 1) * d.W + min(max(xx, 0), d.W - 1)];
            tw[u][dy * 2 + dx] = ok ? (dy ? wy1 : 1.0f - wy1) * (dx ? wx1 : 1.0f - wx1) : 0.0f;
          }
      }
#pragma unroll
      for (int u = 0; u < PX; ++u)
        if (p0 + 256 * u < G2) {
          float v = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) v += tw[u][q] * tv[u][q];
          a_out[((size_t)r * a.out_row_mul + orow_add) * G2 + p0 + 256 * u] = has_mask ? v * mk[u] : v;
        }
    }
  } else
  for (int pix = tid, q = 0; pix < G2; pix += 256, ++q) {
    const float mk = q < MPT ? mk0[q < MPT ? q : 0] : (has_mask ? a_mask[((size_t)r * a.mask_row_mul + mrow_add) * G2 + pix] : 1.0f);
    const int i = sq_div(pix, d.g_mul), j = pix - i * G;
    const float x0f = tab_s[j * 2], wx1 = tab_s[j * 2 + 1];
    const float y0f = tab_s[(G + i) * 2], wy1 = tab_s[(G + i) * 2 + 1];
    const int x0 = (int)x0f, y0 = (int)y0f;
    float v = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int yy = y0 + dy;
      const float wy = dy ? wy1 : 1.0f - wy1;
      if (yy < 0 || yy >= d.H) continue;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int xx = x0 + dx;
        const float wx = dx ? wx1 : 1.0f - wx1;
        if (xx < 0 || xx >= d.W) continue;
        v += wy * wx * src[yy * d.W + xx];
      }
    }
    a_out[((size_t)r * a.out_row_mul + orow_add) * G2 + pix] = has_mask ? v * mk : v;
  }
}

// ------------------------------------------------------------------------------------------------
// tail of a slot for 16 rows (tail_body of sqair_glue.hip).  Dependent operands: the steps predictor's partial pre-activation
// (t1 columns nh..), the glimpse-encoder Gaussian, the raw heads, the slot's where sample, the previous discovery presence.
// ------------------------------------------------------------------------------------------------
template <bool FULL_Z>
__device__ __forceinline__ void chain_tail(const TailArgs& a, const Dims& d, CH_RSRC rs, const char* wsb, const int row0, const bool STORE,
                                           float* zt, float (*rsum)[16], unsigned* status) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int nw = CH_UNI(d.nw), nsp = CH_UNI(d.nh) / 2;
  const int n_tiles = nsp / 16;
  const bool is_disc = CH_UNI(a.is_disc) != 0;
  const int slot = CH_UNI(a.slot);
  ch_gcf4* wp4 = CH_GLB(ch_gcf4, a.wp);
  ch_gcf* __restrict__ a_flat = CH_GLB(ch_gcf, a.flat);
  ch_gcf* __restrict__ a_noise = CH_GLB(ch_gcf, a.noise);
  ch_gcf* __restrict__ a_rec_prev = CH_GLB(ch_gcf, a.rec_prev);
  ch_gf* __restrict__ a_rec_new = CH_GLB(ch_gf, a.rec_new);
  ch_gf* __restrict__ a_s1h_out = CH_GLB(ch_gf, a.s1h_out);
  f32x4_t bv[2][4];
  float sp[2][4], w2v[2];
  unsigned sp_off[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tile = min(wave + 4 * t, n_tiles - 1);
    const int col = tile * 16 + (lane & 15);
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[t][c] = wp4[(size_t)(tile * 4 + c) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) sp_off[t][i] = ch_off(wsb, a.s1p) + (unsigned)(min(row0 + 4 * kq + i, d.R - 1) * a.s1p_ld + col) * 4u;
    w2v[t] = a_flat[a.w2_off + col];
  }
  const int pr = min(row0 + (tid & 15), d.R - 1);
  const float b2 = a_flat[a.b2_off];
  const float u = a_noise[(((size_t)pr * 2 + (is_disc ? 1 : 0)) * d.N + slot) * d.nzw + 4 + nw];
  const bool prev_dep = is_disc && slot > 0;   // the previous discovery step's presence: written by an earlier op of this chain
  float prev = 1.0f;
  if (!is_disc) prev = a_rec_prev[((size_t)pr * d.N + slot) * rec::W + rec::PRES];
  const unsigned wh_off = ch_off(wsb, a.rec_new) + (unsigned)((pr * d.N + slot) * rec::W + rec::WHERE) * 4u;
  const unsigned prev_off = prev_dep ? ch_off(wsb, a.rec_new) + (unsigned)((pr * d.N + slot - 1) * rec::W + rec::PRES) * 4u : wh_off;
  float whv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  constexpr int EPT = 4;
  const int nel = 16 * nw;
  float v_loc[EPT], v_sc[EPT], v_eps[EPT], v_h[EPT][5], v_tm1[EPT];
  unsigned enc_off[EPT], h_off[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = min(tid + 256 * q, nel - 1);
    const int rr = sq_div(e, d.nw_mul), c = e - rr * nw;
    const int r = min(row0 + rr, d.R - 1);
    enc_off[q] = ch_off(wsb, a.enc) + (unsigned)(r * a.enc_ld + c) * 4u;
    h_off[q] = is_disc ? enc_off[q] : ch_off(wsb, a.hraw) + (unsigned)(r * a.h_ld + c) * 4u;
    v_eps[q] = a_noise[(((size_t)r * 2 + (is_disc ? 1 : 0)) * d.N + slot) * d.nzw + 4 + c];
    v_tm1[q] = is_disc ? 0.0f : a_rec_prev[((size_t)r * d.N + slot) * rec::W + rec::WHAT + c];
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned nwb = (unsigned)nw * 4u;
  CH_POLL_BEGIN
    unsigned xs[2][4], xl[EPT], xc[EPT], xh[EPT][5], xw[4], xp;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) xs[t][i] = ch_l1(rs, sp_off[t][i]);
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      xl[q] = ch_l1(rs, enc_off[q]);
      xc[q] = ch_l1(rs, enc_off[q] + nwb);
#pragma unroll
      for (int g = 0; g < 5; ++g) xh[q][g] = is_disc ? 0u : ch_l1(rs, h_off[q] + (unsigned)g * nwb);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xw[q] = FULL_Z ? ch_l1(rs, wh_off + 4u * q) : 0u;
    xp = ch_l1(rs, prev_off);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) { ch_bad |= (unsigned)(xs[t][i] == SQ_SENT); sp[t][i] = __uint_as_float(xs[t][i]); }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      ch_bad |= (unsigned)(xl[q] == SQ_SENT) | (unsigned)(xc[q] == SQ_SENT);
      v_loc[q] = __uint_as_float(xl[q]); v_sc[q] = __uint_as_float(xc[q]);
#pragma unroll
      for (int g = 0; g < 5; ++g) { ch_bad |= (unsigned)(xh[q][g] == SQ_SENT); v_h[q][g] = __uint_as_float(xh[q][g]); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { ch_bad |= (unsigned)(xw[q] == SQ_SENT); whv[q] = __uint_as_float(xw[q]); }
    ch_bad |= (unsigned)(xp == SQ_SENT);
    if (prev_dep) prev = __uint_as_float(xp);
  CH_POLL_END(status)
  for (int i = tid; i < 16 * CH_ZLD; i += 256) {
    const int c = i % CH_ZLD;
    if (c < rec::WHAT || c >= rec::WHAT + nw) zt[i] = 0.0f;
  }
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = tid + 256 * q;
    if (e < nel) {
      const int rr = sq_div(e, d.nw_mul), c = e - rr * nw;
      float loc, sc;
      if (is_disc) {
        loc = v_loc[q];
        sc = v_sc[q];
      } else {
        const float t_loc = v_h[q][0];
        const float t_scale = sq_softplus(v_h[q][1]) + 1e-2f;
        const float fg = sq_gate(v_h[q][2]);
        const float om_ig = sq_gate_compl(v_h[q][3]), om_tg = sq_gate_compl(v_h[q][4]);   // 1 - input gate, 1 - temporal gate
        loc = sq_mix3(fg, v_tm1[q], om_ig, v_loc[q], om_tg, t_loc);
        sc = sq_mix2(om_ig, v_sc[q], om_tg, t_scale);
      }
      const float what = loc + sc * v_eps[q];
      zt[rr * CH_ZLD + rec::WHAT + c] = what;
      if (STORE && row0 + rr < d.R) {
        ch_gf* rn = a_rec_new + ((size_t)(row0 + rr) * d.N + slot) * rec::W;
        rn[rec::WHAT + c] = what;
        rn[rec::WHAT_LOC + c] = loc;
        rn[rec::WHAT_SCALE + c] = sc;
      }
    }
  }
  __syncthreads();
  f32x4_t acc[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4_t av = *reinterpret_cast<const f32x4_t*>(&zt[(lane & 15) * CH_ZLD + 16 * c + 4 * kq]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[t][c].x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[t][c].y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[t][c].z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[t][c].w, acc[t], 0, 0, 0);
    }
  }
  float part[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v = 0.0f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (wave + 4 * t < n_tiles) {
        const float hv = sq_elu(acc[t][i] + sp[t][i]);
        v += hv * w2v[t];
        if (STORE && a.s1h_out != nullptr && row0 + 4 * kq + i < d.R)
          a_s1h_out[(size_t)(row0 + 4 * kq + i) * a.s1h_ld + (wave + 4 * t) * 16 + (lane & 15)] = hv;
      }
    part[i] = sq_row_sum(v);
  }
  if ((lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rsum[wave][4 * kq + i] = part[i];
  }
  __syncthreads();
  if (tid < 16) {
    const float raw = rsum[0][tid] + rsum[1][tid] + rsum[2][tid] + rsum[3][tid] + b2;
    const float logit = prev * raw + (prev - 1.0f) * 88.0f;
    const float prob = sq_sigmoid(logit);
    const float pres = (u < prob ? 1.0f : 0.0f) * prev;
    if (STORE && row0 + tid < d.R) {
      ch_gf* rn = a_rec_new + ((size_t)(row0 + tid) * d.N + slot) * rec::W;
      rn[rec::PRES] = pres;
      rn[rec::LOGIT] = logit;
      rn[rec::PROB] = prob;
    }
    if (FULL_Z) {
      zt[tid * CH_ZLD + rec::PRES] = pres;
      zt[tid * CH_ZLD + rec::LOGIT] = logit;
#pragma unroll
      for (int q = 0; q < 4; ++q) zt[tid * CH_ZLD + rec::WHERE + q] = whv[q];
    }
  }
}

// the VanillaRNN layer of slot k + 1 with the tail of slot k in front of it (k_rnn_tail)
template <int NH>
__device__ __forceinline__ void chain_rnn(const ChainRnn& c, const Dims& d, CH_RSRC rs, const char* wsb, const int tile_m, const int tile_n,
                                          float* zt, float (*rsum)[16], float* red, unsigned* status) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, l15 = lane & 15;
  const int row0 = tile_m * 16;
  constexpr int KC = 4 + 4 * NH;
  ch_gcf4* __restrict__ wp = CH_GLB(ch_gcf4, c.wp) + ((size_t)tile_n * KC) * 64 + lane;
  const unsigned hoff = ch_off(wsb, c.hid) + (unsigned)(min(row0 + l15, d.R - 1) * c.hid_ld) * 4u;
  f32x4_t bz = wp[(size_t)wave * 64], bh[NH], ah[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) bh[i] = wp[(size_t)(wave + 4 * (i + 1)) * 64];
  const int m = row0 + (tid >> 4), n = tile_n * 16 + (tid & 15);
  const int mc = min(m, d.R - 1), nc = min(n, c.n_out - 1);
  const float p_bias = CH_GLB(ch_gcf, c.bias)[nc];
  const unsigned add_off = ch_off(wsb, c.add) + (unsigned)(mc * c.add_ld + nc) * 4u;
  // The layer's own dependent operands (the previous slot's hidden state, produced ~8 ops ago; the hoisted pre-activation) are
  // requested here and CHECKED after the tail: the tail's poll waits a round trip for operands the op just before this one wrote,
  // and by then these have long arrived -- one round trip for the item instead of two.
  float p_add;
  u32x4_t hr[NH];
  unsigned xa;
#define CH_RNN_ISSUE()                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < NH; ++i) hr[i] = ch_l4(rs, hoff + (unsigned)((wave + 4 * i) * 16 + kq * 4) * 4u); \
  xa = ch_l1(rs, add_off);
  CH_RNN_ISSUE()
  __builtin_amdgcn_sched_barrier(0);
  chain_tail<true>(c.ta, d, rs, wsb, row0, tile_n == 0, zt, rsum, status);
  for (int spins = 0;;) {
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < NH; ++i) { bad |= ch_bad4(hr[i]); ah[i] = ch_f4(hr[i]); }
    bad |= (unsigned)(xa == SQ_SENT);
    p_add = __uint_as_float(xa);
    CH_RETRY(bad, status, CH_RNN_ISSUE())
  }
#undef CH_RNN_ISSUE
  __syncthreads();
  const f32x4_t az = *reinterpret_cast<const f32x4_t*>(&zt[l15 * CH_ZLD + 16 * wave + 4 * kq]);
  f32x4_t acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(az.x, bz.x, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(az.y, bz.y, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(az.z, bz.z, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(az.w, bz.w, acc1, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i].x, bh[i].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i].y, bh[i].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i].z, bh[i].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i].w, bh[i].w, acc1, 0, 0, 0);
  }
  float* r = red + wave * 256;
  r[(4 * kq + 0) * 16 + l15] = acc0.x + acc1.x;
  r[(4 * kq + 1) * 16 + l15] = acc0.y + acc1.y;
  r[(4 * kq + 2) * 16 + l15] = acc0.z + acc1.z;
  r[(4 * kq + 3) * 16 + l15] = acc0.w + acc1.w;
  __syncthreads();
  if (m < d.R && n < c.n_out) CH_GLB(ch_gf, c.out)[(size_t)m * c.out_ld + n] = sq_tanh(red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid] + p_bias + p_add);
}

// ------------------------------------------------------------------------------------------------
// the persistent kernel.  Control block of a launch (zero at launch): word 0 of line x (32 words) = workgroups counted on XCD x;
// line 9 word 0 = status (1 = a consumer gave up polling, 3 = the census never completed).
// ------------------------------------------------------------------------------------------------
constexpr int CH_LDS_ZT = 0, CH_LDS_RS = 16 * CH_ZLD, CH_LDS_RED = CH_LDS_RS + 64, CH_LDS_FLOATS = CH_LDS_RED + 2 * 3072 + 16;
constexpr int CH_MAX_TILES_PER_XCD = 8;

__global__ __launch_bounds__(256) void k_slot_chain(const ChainTable* __restrict__ tab, unsigned* __restrict__ ctl, const float* ws_base,
                                                    const unsigned ws_bytes SQ_TLP) {
  SQ_TL_SCOPE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ unsigned s_cen[10];
  const int tid = threadIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  unsigned* status = ctl + 9 * 32;
  if (tid == 0) s_cen[8] = __hip_atomic_fetch_add(ctl + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  CH_RSRC rs = __builtin_amdgcn_make_buffer_rsrc((void*)ws_base, 0, (int)ws_bytes, 0x00020000);
  const char* wsb = (const char*)ws_base;
  const int n_row_tiles = tab->n_row_tiles, n_ops = tab->n_ops;
  // a failed launch fails the rest of the pass at once (its results are garbage anyway; nobody spins on them)
  const unsigned* pass_status = ctl - (size_t)tab->launch_id * SQ_CHAIN_CTL_WORDS + 9 * 32;
  if (tab->launch_id > 0 && __hip_atomic_load(pass_status + (size_t)(tab->launch_id - 1) * SQ_CHAIN_CTL_WORDS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    if (tid == 0) __hip_atomic_store(status, 6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (n_ops < 1 || n_ops > SQ_CHAIN_MAX_OPS || tab->lds_scratch_floats < CH_LDS_FLOATS || tab->lds_scratch_floats > 40000) {
    if (tid == 0) __hip_atomic_store(status, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // not a table
    return;
  }
  // the table's LDS copy (behind the ops' scratch): requested while the census atomics travel
  const int scratch_floats = tab->lds_scratch_floats;
  const ChainTable* lt = reinterpret_cast<const ChainTable*>(__builtin_assume_aligned(smem + (scratch_floats & ~3), 16));
  {
    const int words4 = ((int)offsetof(ChainTable, ops) + n_ops * (int)sizeof(ChainOp) + 15) / 16;
    const u32x4_t* src = reinterpret_cast<const u32x4_t*>(tab);
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(smem + scratch_floats);
    for (int w0 = tid; w0 < words4; w0 += 1024) {
      u32x4_t v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = src[min(w0 + 256 * q, words4 - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q) if (w0 + 256 * q < words4) dst[w0 + 256 * q] = v[q];
    }
  }
  // census: every workgroup of the grid has counted itself on its XCD (co-residency of the grid is the one requirement)
  if (tid < 64) {
    unsigned c = 0, tot = 0;
    int spins = 0;
    do {
      c = tid < 8 ? __hip_atomic_load(ctl + tid * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      tot = 0;
#pragma unroll
      for (int x = 0; x < 8; ++x) tot += (unsigned)__builtin_amdgcn_readlane((int)c, x);
      if (tot < gridDim.x) __builtin_amdgcn_s_sleep(2);
    } while (tot < gridDim.x && ++spins < SQ_CHAIN_SPIN_LIMIT);
    if (tid < 8) s_cen[tid] = c;
    if (tid == 0) s_cen[9] = tot >= gridDim.x ? 1u : 0u;
  }
  __syncthreads();
  if (CH_UNI(s_cen[9]) == 0u) {
    if (tid == 0) __hip_atomic_store(status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int rank = (int)CH_UNI(s_cen[8]);
  int n_x = 1, n_act = 0, idx_x = 0;
#pragma unroll
  for (int x = 0; x < 8; ++x) {
    const int cx = (int)CH_UNI(s_cen[x]);
    n_act += cx > 0;
    idx_x += (cx > 0 && x < (int)xcc);
    if (x == (int)xcc) n_x = cx;
  }
  // row tiles of this XCD: idx_x, idx_x + n_act, ...
  const int ntl = idx_x < n_row_tiles ? (n_row_tiles - idx_x + n_act - 1) / n_act : 0;
  if (ntl == 0) return;
  // With two workgroups per CU in the grid (sq_chain_flush: where the occupancy allows), an XCD serving ONE row tile keeps
  // one workgroup per CU -- its ops have at most as many items as that, and the workgroups not needed would only add pollers --
  // while an XCD serving several tiles uses both: consecutive ops then run on alternating halves there too.
  {
    const int cus_x = max((int)gridDim.x >> 4, 1);   // workgroups per XCD at one per CU (8 XCDs, 2 per CU in the grid)
    if ((int)gridDim.x >= 512 && ntl == 1 && n_x > cus_x) {
      if (rank >= cus_x) return;
      n_x = cus_x;
    }
  }
  if (ntl > CH_MAX_TILES_PER_XCD) {
    if (tid == 0) __hip_atomic_store(status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  float* zt = smem + CH_LDS_ZT;
  float (*rsum)[16] = reinterpret_cast<float (*)[16]>(smem + CH_LDS_RS);
  const int R = CH_UNI(lt->d.R);
  const bool staged = CH_UNI(lt->staged) != 0;
  // dense items alternate between two LDS meeting areas and end without a barrier (the barrier inside the NEXT dense item
  // orders this item's reads ahead of the one after next's writes); any other item starts behind one when a dense item preceded it
  int red_par = 0;
  bool pending_sync = false;
#ifdef SQAIR_KNOBS
  unsigned tr_items = 0;
#endif
  for (int opi = 0; opi < n_ops; ++opi) {
    const ChainOp& op = lt->ops[opi];
    const int kind = CH_UNI(op.kind), items = CH_UNI(op.items);
    // dense: an item takes TN consecutive column tiles of a row tile, TN such that the XCD's workgroups cover the op in one round
    int tn = 1, groups = items;
    if (kind == COP_DENSE) {
      tn = min(3, (ntl * items + n_x - 1) / n_x);
      groups = (items + tn - 1) / tn;
    }
    const int total = ntl * groups;
    // consecutive ops start at opposite halves of the XCD's workgroups: while one half computes op k the other half has already
    // fetched the weights of its item of op k + 1 and polls
    const int rot = (opi & 1) ? (n_x >> 1) : 0;
    int j = rank - rot;
    if (j < 0) j += n_x;
    for (; j < total; j += n_x) {
      CH_TRACE_T(0);
      int grp = 0, sub = j;
      while (sub >= groups) { sub -= groups; ++grp; }
      const int tile = idx_x + grp * n_act;
      if (kind == COP_DENSE) {
        float* red = smem + CH_LDS_RED + red_par * 3072;
        red_par ^= 1;
        pending_sync = true;
        const int nch = CH_UNI(op.nch);
        if (nch <= 4) {
          if (tn == 1) chain_dense<4, 1>(op.u.dense, rs, tile, sub, items, R, red, status);
          else if (tn == 2) chain_dense<4, 2>(op.u.dense, rs, tile, 2 * sub, items, R, red, status);
          else chain_dense<4, 3>(op.u.dense, rs, tile, 3 * sub, items, R, red, status);
        } else {
          if (tn == 1) chain_dense<7, 1>(op.u.dense, rs, tile, sub, items, R, red, status);
          else if (tn == 2) chain_dense<7, 2>(op.u.dense, rs, tile, 2 * sub, items, R, red, status);
          else chain_dense<7, 3>(op.u.dense, rs, tile, 3 * sub, items, R, red, status);
        }
      } else {
        if (pending_sync) { __syncthreads(); pending_sync = false; }
        float* red = smem + CH_LDS_RED;
        if (kind == COP_CROP) {
          const int r = tile * 16 + sub;
          if (r < R) {
            if (staged) chain_crop<true>(op.u.crop, lt->po, lt->d, rs, wsb, r, smem, status);
            else chain_crop<false>(op.u.crop, lt->po, lt->d, rs, wsb, r, smem, status);
          }
        } else if (kind == COP_RNN_TAIL) {
          chain_rnn<4>(op.u.rnn, lt->d, rs, wsb, tile, sub, zt, rsum, red, status);   // (n_hidden = 256: sq_chain_on)
        } else {
          chain_tail<false>(op.u.tail, lt->d, rs, wsb, tile * 16, true, zt, rsum, status);
        }
      }
#ifdef SQAIR_KNOBS
      if (tid == 0 && g_chain_trace != nullptr) {   // (one record slot per (workgroup, item ordinal): no atomics in the loop)
        const unsigned long long t6 = ch_clock();
        const unsigned idx = ((unsigned)tab->launch_id * 256u + blockIdx.x) * 96u + tr_items++;
        if (idx < g_chain_trace_cap && tr_items <= 96u) {
          unsigned long long* p = g_chain_trace + (size_t)idx * 8;
          const unsigned long long* st = reinterpret_cast<const unsigned long long*>(smem + CH_LDS_TR);
          p[0] = (unsigned long long)opi | ((unsigned long long)(idx_x + grp * n_act) << 16) | ((unsigned long long)sub << 32) |
                 ((unsigned long long)xcc << 48) | ((unsigned long long)(rank & 0xff) << 52) | ((unsigned long long)kind << 60);
          p[1] = st[0]; p[2] = st[1]; p[3] = st[2]; p[4] = st[3]; p[5] = st[4]; p[6] = st[5]; p[7] = t6;
        }
      }
#endif
      if (kind != COP_DENSE) __syncthreads();
    }
  }
}

// sentinel fill of the hand-off words of a pass + zero fill of the control blocks of its chain launches
__global__ __launch_bounds__(256) void k_chain_poison(const ChainPoisonList pl, unsigned* __restrict__ ctl_all, const int ctl_words SQ_TLP) {
  SQ_TL_SCOPE;
  const int ri = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (ri == pl.n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ctl_words; i += stride) ctl_all[i] = 0u;
    return;
  }
  const ChainPoison p = pl.r[ri];
  unsigned* base = reinterpret_cast<unsigned*>(p.base);
  if ((p.width & 3) == 0 && (p.row_stride & 3) == 0) {   // 16-byte units
    const int w4 = p.width >> 2;
    const int64_t n = p.rows * w4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      const int64_t row = i / w4;
      const int c = (int)(i - row * w4);
      *reinterpret_cast<u32x4_t*>(base + row * p.row_stride + 4 * c) = u32x4_t{SQ_SENT, SQ_SENT, SQ_SENT, SQ_SENT};
    }
  } else {
    const int64_t n = p.rows * p.width;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      const int64_t row = i / p.width;
      base[row * p.row_stride + (i - row * p.width)] = SQ_SENT;
    }
  }
}
#endif  // !SQAIR_WIDE

int sq_chain_poison(const ChainPoisonList& pl, unsigned* ctl_all, int ctl_words, hipStream_t s) {
#ifdef SQAIR_WIDE
  (void)pl; (void)ctl_all; (void)ctl_words; (void)s;
  return -1;
#else
  SQ_LAUNCH(k_chain_poison, dim3(512, pl.n + 1), dim3(256), 0, s, pl, ctl_all, ctl_words);
  return 0;
#endif
}

// ------------------------------------------------------------------------------------------------
// host recorder + table cache
// ------------------------------------------------------------------------------------------------
// A cached table is keyed by its CONTENT, which includes the addresses of the caller's frame / noise / parameter buffers: a
// caller that streams batches through fresh buffers makes new tables every pass.  Tables live in device arenas of the handle;
// a table a stream capture has referenced is pinned (a graph holds its address), so is its arena.  When the current arena is
// full the stream is synchronised and the arena recycled if nothing in it is pinned, else a new one is opened -- growth is
// bounded by the number of captured graphs, not by the number of passes.
struct ChainCacheEntry { std::vector<char> host; void* dev; int arena; bool pinned; };
constexpr size_t CH_ARENA_BYTES = 32u << 20;
constexpr int CH_MAX_ARENAS = 64;
struct ChainArena { void* mem = nullptr; size_t used = 0; bool pinned = false; };
struct ChainState {
  bool active = false;
  bool failed = false;
  const char* ws_base = nullptr;
  int64_t ws_bytes = 0;
  ChainTable tab;
  std::list<ChainCacheEntry> cache;   // (a list: the host copies are the sources of asynchronous uploads and must not move)
  std::vector<ChainArena> arenas;
  size_t arena_bytes = CH_ARENA_BYTES;
  int cur_arena = -1;
  int grid = 0, occupancy = 0;
};
static ChainState* cs_of(SqairHandle* h) {
  if (!h->chain) h->chain = new ChainState();
  return (ChainState*)h->chain;
}
int sq_chain_set_arena_kb(SqairHandle* h, int kb) {   // before the first chain launch of the handle only
  ChainState* c = cs_of(h);
  if (kb < 64 || kb > (1 << 20) || !c->arenas.empty()) return -2;
  c->arena_bytes = (size_t)kb << 10;
  return 0;
}
void sq_chain_reset(SqairHandle* h) {
  if (!h->chain) return;
  ChainState* c = (ChainState*)h->chain;
  c->active = false;
  c->failed = false;
}
bool sq_chain_active(const SqairHandle* h) { return h->chain && ((const ChainState*)h->chain)->active; }
void sq_chain_destroy(SqairHandle* h) {
  if (!h->chain) return;
  ChainState* c = (ChainState*)h->chain;
  for (auto& a : c->arenas)
    if (a.mem) (void)hipFree(a.mem);
  delete c;
  h->chain = nullptr;
}
void sq_chain_begin(SqairHandle* h, const Dims& d, const POff& po, const float* ws_base, int64_t ws_bytes) {
  ChainState* c = cs_of(h);
  memset(&c->tab, 0, offsetof(ChainTable, ops));
  c->tab.d = d;
  c->tab.po = po;
  c->tab.staged = d.H * d.W <= SQ_CROP_STAGE_MAX_PIXELS ? 1 : 0;
  c->ws_base = (const char*)ws_base;
  c->ws_bytes = ws_bytes;
  c->active = true;
  c->failed = false;
}
static ChainOp* chain_new_op(SqairHandle* h, int kind, int items) {
  ChainState* c = cs_of(h);
  if (c->tab.n_ops >= SQ_CHAIN_MAX_OPS) { sq_set_error(h, "slot chain: too many ops"); c->failed = true; return nullptr; }
  ChainOp* op = &c->tab.ops[c->tab.n_ops];
  memset(op, 0, sizeof(ChainOp));
  op->kind = kind;
  op->items = items;
  ++c->tab.n_ops;
  return op;
}
// byte offset of a workspace pointer (every operand the chain fetches through its buffer resource must lie inside the workspace)
static bool chain_off(SqairHandle* h, const void* p, unsigned* off) {
  ChainState* c = cs_of(h);
  const int64_t o = (const char*)p - c->ws_base;
  if (o < 0 || o >= c->ws_bytes) { sq_set_error(h, "slot chain: an operand outside the workspace"); c->failed = true; return false; }
  *off = (unsigned)o;
  return true;
}
static void chain_split(const void* p, unsigned* lo, unsigned* hi) {
  const unsigned long long v = (unsigned long long)(uintptr_t)p;
  *lo = (unsigned)v; *hi = (unsigned)(v >> 32);
}
int sq_chain_add_dense(SqairHandle* h, const LinArgs& a, int kc_total, int n_tiles) {
  ChainOp* op = chain_new_op(h, COP_DENSE, n_tiles);
  if (!op) return -3;
  ChDense& dn = op->u.dense;
  ChainState* cst = cs_of(h);
  if (kc_total > SQ_CHAIN_MAX_KC || a.scale_ptr != nullptr || a.add_rdiv > 1) { sq_set_error(h, "slot chain: layer outside the chain's dense body"); cst->failed = true; return -3; }
  chain_split(a.wp, &dn.wp_lo, &dn.wp_hi);
  chain_split(a.wzero, &dn.wz_lo, &dn.wz_hi);
  chain_split(a.bias, &dn.bias_lo, &dn.bias_hi);
  if (a.M != cst->tab.d.R) { sq_set_error(h, "slot chain: a dense op over other rows than the particle rows"); cst->failed = true; return -3; }
  dn.M = a.M; dn.N = a.N; dn.kc_total = kc_total; dn.nch = (kc_total + 3) / 4;
  op->nch = dn.nch;
  dn.epi = a.epi; dn.act_a = a.act_a; dn.act_b = a.act_b; dn.act_split = a.act_split; dn.scale = a.scale; dn.nh = a.nh;
  dn.add_off = dn.o1_off = dn.o2_off = dn.o3_off = SQ_CHAIN_NONE;
  bool ok = chain_off(h, a.out, &dn.out_off);
  dn.out_ld = a.out_ld;
  if (a.add != nullptr) { ok = ok && chain_off(h, a.add, &dn.add_off); dn.add_ld = a.add_ld; dn.add_n = a.add_n; }
  if (a.epi != EPI_ACT) { ok = ok && chain_off(h, a.e0, &dn.e0_off); dn.e0_ld = a.e0_ld; }
  if (a.epi == EPI_GRU2) { ok = ok && chain_off(h, a.e1, &dn.e1_off); dn.e1_ld = a.e1_ld; }
  if (a.o1 != nullptr) { ok = ok && chain_off(h, a.o1, &dn.o1_off); dn.o1_ld = a.o1_ld; }
  if (a.o2 != nullptr) { ok = ok && chain_off(h, a.o2, &dn.o2_off); dn.o2_ld = a.o2_ld; }
  if (a.o3 != nullptr) { ok = ok && chain_off(h, a.o3, &dn.o3_off); dn.o3_ld = a.o3_ld; }
  int g = 0;
  for (int i = 0; i < a.nseg && ok; ++i) {
    const LinSeg& sg = a.seg[i];
    if (sg.rdiv > 1) { sq_set_error(h, "slot chain: row divisors are not supported"); cst->failed = true; return -3; }
    unsigned base = 0;
    ok = chain_off(h, sg.p, &base);
    const int lim = ((sg.width + 3) & ~3) - 4;
    for (int q = 0; q < (sg.width + 15) / 16; ++q, ++g) dn.chunk[g] = ChChunk{base + 64u * q, (unsigned)sg.ld * 4u, (unsigned)(lim * 4 - 64 * q), 0u};
  }
  if (!ok) return -3;
  if (g != kc_total) { sq_set_error(h, "slot chain: chunk count mismatch"); cst->failed = true; return -3; }
  for (; g < SQ_CHAIN_MAX_KC; ++g) dn.chunk[g] = dn.chunk[kc_total - 1];
  return 0;
}
int sq_chain_add_crop(SqairHandle* h, const CropArgs& a) {
  if (!(a.mode == CROP_PROP2 || a.mode == CROP_DISC) || a.t2 == nullptr) { sq_set_error(h, "slot chain: unsupported crop mode"); cs_of(h)->failed = true; return -3; }
  ChainOp* op = chain_new_op(h, COP_CROP, 16);
  if (!op) return -3;
  op->u.crop = a;
  return 0;
}
int sq_chain_add_rnn_tail(SqairHandle* h, const ChainRnn& r) {
  ChainOp* op = chain_new_op(h, COP_RNN_TAIL, (r.n_out + 15) / 16);
  if (!op) return -3;
  op->u.rnn = r;
  return 0;
}
int sq_chain_add_tail(SqairHandle* h, const TailArgs& a) {
  ChainOp* op = chain_new_op(h, COP_TAIL, 1);
  if (!op) return -3;
  op->u.tail = a;
  return 0;
}

int sq_chain_flush(SqairHandle* h, unsigned* ctl, int launch_id, hipStream_t s) {
#ifdef SQAIR_WIDE
  (void)h; (void)ctl; (void)launch_id; (void)s;
  return -1;
#else
  ChainState* c = cs_of(h);
  c->active = false;
  ChainTable& t = c->tab;
  if (c->failed) return -3;
  if (t.n_ops == 0) return 0;
  if (c->ws_bytes >= (int64_t)0xFFFFFFF0ll) { sq_set_error(h, "slot chain: workspace beyond the 4 GB window of the hand-off loads"); return -3; }
  if (t.d.nh != 256) { sq_set_error(h, "slot chain: n_hidden != 256"); return -3; }
  t.n_row_tiles = (t.d.R + 15) / 16;
  t.launch_id = launch_id;
  {
    const int crop_floats = 4 + 4 * t.d.G + (t.staged ? t.d.H * t.d.W : 0);
    t.lds_scratch_floats = ((crop_floats > CH_LDS_FLOATS ? crop_floats : CH_LDS_FLOATS) + 3) / 4 * 4;
  }
  const size_t bytes = (offsetof(ChainTable, ops) + (size_t)t.n_ops * sizeof(ChainOp) + 15) / 16 * 16;
  void* dev = nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  SQ_CHECK_HIP(hipStreamIsCapturing(s, &cap));
  for (auto& e : c->cache)
    if (e.host.size() == bytes && memcmp(e.host.data(), &t, bytes) == 0) {
      dev = e.dev;
      if (cap != hipStreamCaptureStatusNone) { e.pinned = true; c->arenas[e.arena].pinned = true; }   // a graph now holds this address
      break;
    }
  if (!dev) {
    // Tables are uploaded outside stream captures only (an allocation + copy issued while a capture is open on this thread
    // produced a graph that read a garbage table): the capturing entry points run one eager pass first, which leaves every
    // table of the pass resident here.
    if (cap != hipStreamCaptureStatusNone) {
      sq_set_error(h, "slot chain: a table of this pass is not resident -- run the same pass once eagerly before capturing it");
      return -3;
    }
    if (bytes > c->arena_bytes) { sq_set_error(h, "slot chain: table larger than an arena"); return -3; }
    if (c->cur_arena < 0 || c->arenas[c->cur_arena].used + bytes > c->arena_bytes) {
      // the current arena is full: every launch that reads its tables is in stream order behind us -- wait for them, then
      // recycle the first arena no graph refers to (its entries leave the cache), else open another one
      if (c->cur_arena >= 0) SQ_CHECK_HIP(hipStreamSynchronize(s));
      int pick = -1;
      for (int i = 0; i < (int)c->arenas.size() && pick < 0; ++i)
        if (!c->arenas[i].pinned) pick = i;
      if (pick >= 0) {
        c->cache.remove_if([pick](const ChainCacheEntry& e) { return e.arena == pick; });
        c->arenas[pick].used = 0;
      } else {
        if ((int)c->arenas.size() >= CH_MAX_ARENAS) { sq_set_error(h, "slot chain: table arenas exhausted (too many captured graphs on this handle)"); return -3; }
        ChainArena a;
        SQ_CHECK_HIP(hipMalloc(&a.mem, c->arena_bytes));
        c->arenas.push_back(a);
        pick = (int)c->arenas.size() - 1;
      }
      c->cur_arena = pick;
    }
    // The table goes IN the stream of the launch that reads it (a plain hipMemcpy from pageable memory returns once the bytes
    // are in the staging buffer, not once they are on the device: the chain launch behind it read a half-written table).
    ChainArena& ar = c->arenas[c->cur_arena];
    dev = (char*)ar.mem + ar.used;
    ar.used += (bytes + 255) / 256 * 256;
    c->cache.emplace_back();
    ChainCacheEntry& ce = c->cache.back();
    ce.host.assign((const char*)&t, (const char*)&t + bytes);
    ce.dev = dev;
    ce.arena = c->cur_arena;
    ce.pinned = false;
    SQ_CHECK_HIP(hipMemcpyAsync(dev, ce.host.data(), bytes, hipMemcpyHostToDevice, s));
  }
  const int lds = t.lds_scratch_floats * 4 + (int)bytes;
  if (lds > 64 * 1024) {
    if (lds > 150 * 1024 || sq_allow_big_lds((const void*)k_slot_chain, lds) != 0) { sq_set_error(h, "slot chain: LDS"); return -3; }
  }
  // The census needs the whole grid co-resident: one workgroup per CU, or two where the occupancy query says two fit (one
  // polls while the other computes).
  if (c->grid == 0) {
    int dev_id = 0, cus = 0, occ = 0;
    SQ_CHECK_HIP(hipGetDevice(&dev_id));
    SQ_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id));
    SQ_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_slot_chain, 256, (size_t)lds));
    static const int per_cu = SQ_KNOB_INT("SQAIR_CHAIN_WGS_PER_CU", 1);   // (2 measured: 3.3 -> 7.3 ms at cfg-2)
    c->occupancy = occ;
    c->grid = cus * (occ >= per_cu ? per_cu : 1);
    if (occ < 1 || cus < 1) { sq_set_error(h, "slot chain: the kernel does not fit a CU"); return -3; }
#ifdef SQAIR_KNOBS
    fprintf(stderr, "slot chain: %d CUs, occupancy query %d workgroups per CU at %d bytes of LDS, grid %d\n", cus, occ, lds, c->grid);
#endif
  }
  SQ_LAUNCH(k_slot_chain, dim3(c->grid), dim3(256), lds, s, (const ChainTable*)dev, ctl, (const float*)c->ws_base, (unsigned)c->ws_bytes);
  return 0;
#endif
}

#ifdef SQAIR_KNOBS
// knob build only: per-item trace buffer of the chain kernel (8 x u64 per record, (launch * 256 + workgroup) * 96 + item ordinal);
// a null buffer / cap 0 switches it off
extern "C" int sqair_chain_trace(void* buf, unsigned cap_records) {
#ifndef SQAIR_WIDE
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), &buf, sizeof(buf)) != hipSuccess) return -2;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace_cap), &cap_records, 4) != hipSuccess) return -2;
  return 0;
#else
  (void)buf; (void)cap_records;
  return -1;
#endif
}
#endif
