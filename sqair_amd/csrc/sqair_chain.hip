// Layer chains (EXPERIMENTAL, opt-in through sqair_enable_chains; measured SLOWER than one launch per layer at BASELINE's
// batch, see the end of this comment): two or three DEPENDENT dense layers of a slot (glimpse encoder E1 -> E2 -> E3) in ONE
// launch.  Rows (b' = sequence x particle) never interact, so the rows are split over the 8 XCDs: the workgroups that
// the hardware places on XCD x (block b -> XCD b % 8, the dispatch order of this chip) form team x and own rows
// [x * ceil(R/8), ...).  Inside a launch a layer's output is handed to the next layer through that XCD's own L2 (plain
// stores; L1-bypassing `sc1` loads, because a CU's vector L1 is not refreshed by other CUs' stores) behind a per-team
// arrival counter instead of a device-wide kernel boundary.  tools/xcd_team.hip prices it: three dependent 32-row
// 256 x 256 layers cost 6.8 us as one such launch against 10.8 us as three graph nodes.
//
// The placement is a SPEED assumption that is CHECKED, not trusted: every workgroup derives the launch's XCD rotation
// from HW_REG_XCC_ID and raises the pass's status word when two workgroups of a launch disagree (the results of that pass are then reported invalid by
// sqair_chain_status and the caller falls back to one launch per layer); every spin is bounded, a time-out raises the
// same word, nobody hangs.  The tile arithmetic is x_linear_tile's, i.e. k_linear's: results are bit-identical to the
// launch-per-layer path.
//
// Measured (cfg-2, 160 rows, device-clock stamps of workgroup 0, tools/try_chain.py): a three-layer glimpse-encoder chain
// takes 15.4 us in-kernel against 2.8 + 2.3 + 2.2 us for its layers as separate launches (+ 2 x 1.6 us of kernel
// boundary): first layer done after 4.0 us, each team barrier 1.6-2.5 us, each later layer 2.6 us of which ~1.5 us is
// the L1-bypassing load of activations another CU has just written (NOT the 0.3-0.5 us of an ordinary L2 hit; `sc0` and
// `sc1` measure the same).  What was fixed on the way, worth keeping in mind for any kernel with a large argument block:
// 160 scratch instructions from a select chain over a by-value struct (SQ_XSEGS), ~15 serialised scalar-cache misses on
// the ~1 KB argument block (touched up front now), address set-up moved in front of the barrier.  The forward pass takes
// 5.0 ms with the chains against 4.49 ms without: the team barrier costs as much as the kernel boundary it replaces.
#include "sqair_internal.h"
#include "sqair_rowops.h"
#include "sqair_chain.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
// loads of data written earlier in this launch by OTHER workgroups of the team: bypass the L1 (see sqair_persist.hip)
__device__ __forceinline__ float c_ldf(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void c_ld4x4_sc1(const float* p0, const float* p1, const float* p2, const float* p3, f32x4& v0, f32x4& v1,
                                            f32x4& v2, f32x4& v3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %5, off sc1\n\t"
      "global_load_dwordx4 %2, %6, off sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}
struct LdTeam {
  static __device__ __forceinline__ float f(const float* p) { return c_ldf(p); }
  static __device__ __forceinline__ sq_f32x4 f4(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
  }
  static __device__ __forceinline__ void f4x4(const float* p0, const float* p1, const float* p2, const float* p3, sq_f32x4& v0,
                                              sq_f32x4& v1, sq_f32x4& v2, sq_f32x4& v3) {
    c_ld4x4_sc1(p0, p1, p2, p3, v0, v1, v2, v3);
  }
};

// ---- one 16 x 16 output tile in two halves: everything that does NOT depend on the previous layer of the chain (this
// tile's packed weight fragments, bias, pre-activation addend) is requested BEFORE the team barrier, the activation
// loads + MFMAs + epilogue run after it.  Arithmetic order of k_linear / x_linear_tile (bit-identical results).
constexpr int NB = 8;  // weight fragments (16-wide K chunks) per wave held in registers: K <= 512
struct TilePre {
  f32x4 b[NB];
  const float* ap[NB];  // activation addresses of this lane's K chunks
  float bias, add, scale, scale2;
  float* outp;          // EPI_ACT: this thread's output element (nullptr = outside the matrix)
  int act;
  int task;  // -1: nothing prefetched
};
__device__ __forceinline__ bool can_prefetch(const ChainLayer& l) { return ((l.kc + 3) >> 2) <= NB; }

__device__ __forceinline__ void tile_prefetch(const ChainLayer& l, int task, int m0, int m1, TilePre& p) {
  const LinArgs& a = l.a;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile_n = task % l.nt, mbase = m0 + (task / l.nt) * 16;
  const int m = mbase + (tid >> 4), n = tile_n * 16 + (tid & 15);
  const int mc = min(m, m1 - 1), nc = min(n, a.N - 1);
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + ((size_t)tile_n * l.kc) * 64 + lane;
  const f32x4* __restrict__ wz = reinterpret_cast<const f32x4*>(a.wzero) + lane;
  const int nmine = (l.kc - wave + 3) >> 2;
#pragma unroll
  for (int j = 0; j < NB; ++j) p.b[j] = *(j < nmine ? wp + (size_t)(wave + 4 * j) * 64 : wz);
  const float* pb = a.bias + nc;
  const bool use_add = a.add != nullptr && nc < a.add_n;
  const int mcd = a.add_rmul ? (int)__umulhi((unsigned)mc, a.add_rmul) : mc;
  p.bias = *pb;
  const float av = *(use_add ? a.add + (size_t)mcd * a.add_ld + nc : pb);
  p.add = use_add ? av : 0.0f;
  // addresses and epilogue scalars: every read of the argument block happens HERE, i.e. before the team barrier
  const int kq = lane >> 4;
  const int arow = min(mbase + (lane & 15), m1 - 1);
  SQ_XSEGS(a, arow)
#pragma unroll
  for (int j = 0; j < NB; ++j) p.ap[j] = SQ_XAPTR(j < nmine ? wave + 4 * j : wave, kq);  // unused slots re-read chunk `wave` against zero weights
  p.scale = a.scale;
  p.scale2 = a.scale_ptr != nullptr ? *a.scale_ptr : 1.0f;
  p.act = n < a.act_split ? a.act_a : a.act_b;
  p.outp = (m < m1 && n < a.N) ? a.out + (size_t)m * a.out_ld + n : nullptr;
  p.task = task;
}

__device__ __forceinline__ void ld8_sc1(const float* const* ap, f32x4* v) {  // (sc0 measures the same on this chip: tools/xcd_team.hip)
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(ap[0]), "v"(ap[1]), "v"(ap[2]), "v"(ap[3]), "v"(ap[4]), "v"(ap[5]), "v"(ap[6]), "v"(ap[7])
      : "memory");
}

template <bool TEAM>  // TEAM: the activations were written earlier in THIS launch by other workgroups of the team
__device__ __forceinline__ void tile_finish(const ChainLayer& l, int m0, int m1, const TilePre& p, float* red) {
  const LinArgs& a = l.a;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  f32x4 av[NB];
  if (TEAM) {
    ld8_sc1(p.ap, av);
  } else {
#pragma unroll
    for (int j = 0; j < NB; ++j) av[j] = *reinterpret_cast<const f32x4*>(p.ap[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  // k_linear walks the chunks in blocks of NCH per wave with invalid slots multiplying zero weights: adding exact zeros
  // keeps the sums bit-identical whatever the block size
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, p.b[j].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, p.b[j].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, p.b[j].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, p.b[j].w, acc1, 0, 0, 0);
  }
  float* r = red + wave * 256;
  r[(4 * kq + 0) * 16 + (lane & 15)] = acc0.x + acc1.x;
  r[(4 * kq + 1) * 16 + (lane & 15)] = acc0.y + acc1.y;
  r[(4 * kq + 2) * 16 + (lane & 15)] = acc0.z + acc1.z;
  r[(4 * kq + 3) * 16 + (lane & 15)] = acc0.w + acc1.w;
  __syncthreads();
  const float v = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid] + p.bias + p.add;
  if (a.epi == EPI_ACT) {  // x_epilogue's EPI_ACT branch on the prefetched scalars (same operation order)
    if (p.outp != nullptr) *p.outp = sq_act(v, p.act) * p.scale * p.scale2;
  } else {  // GRU epilogues: the generic path
    const int tile_n = p.task % l.nt, mbase = m0 + (p.task / l.nt) * 16;
    const int m = mbase + (tid >> 4), n = tile_n * 16 + (tid & 15);
    const int mc = min(m, m1 - 1), nc = min(n, a.N - 1);
    const bool g1 = a.epi == EPI_GRU1 && nc >= a.nh && nc < 2 * a.nh;
    const float* pe0 = g1 ? a.e0 + (size_t)mc * a.e0_ld + (nc - a.nh) : a.e0 + (size_t)mc * a.e0_ld + nc;
    const float* pe1 = a.epi == EPI_GRU2 ? a.e1 + (size_t)mc * a.e1_ld + nc : pe0;
    const float p_e0 = TEAM ? c_ldf(pe0) : *pe0, p_e1 = TEAM ? c_ldf(pe1) : *pe1;
    if (m < m1 && n < a.N) x_epilogue(a, m, n, v, p_e0, p_e1, a.scale_ptr != nullptr ? *a.scale_ptr : 1.0f);
  }
  __syncthreads();
}

// the team's tiles of one layer; `pre` holds this workgroup's first tile if it was prefetched before the barrier
template <bool TEAM>
__device__ __forceinline__ void run_layer(const ChainLayer& l, int m0, int m1, int rank, int size, TilePre& pre, float* red) {
  const int tasks = ((m1 - m0 + 15) >> 4) * l.nt;
  for (int task = rank; task < tasks; task += size) {
    if (pre.task != task) {
      if (!can_prefetch(l)) {  // deeper K than the register budget: the generic tile
        if (TEAM) x_linear_tile<LdTeam>(l.a, l.kc, task % l.nt, m0 + (task / l.nt) * 16, m1, red);
        else x_linear_tile<LdPlain>(l.a, l.kc, task % l.nt, m0 + (task / l.nt) * 16, m1, red);
        continue;
      }
      tile_prefetch(l, task, m0, m1, pre);
    }
    tile_finish<TEAM>(l, m0, m1, pre, red);
  }
  pre.task = -1;
}
// request the first tile of the NEXT layer (weights, bias, addend: nothing the barrier protects)
__device__ __forceinline__ void prefetch_next(const ChainLayer& l, int m0, int m1, int rank, TilePre& pre) {
  const int tasks = ((m1 - m0 + 15) >> 4) * l.nt;
  if (rank < tasks && can_prefetch(l)) tile_prefetch(l, rank, m0, m1, pre);
}

// arrival of this workgroup at the team's barrier `phase` (1-based); false = timed out (status raised)
__device__ __forceinline__ void team_arrive_wait(unsigned* bar, unsigned target, int* status) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spin = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spin > (1 << 20)) {  // ~30 ms: something is badly wrong; never hang the device
        __hip_atomic_store(status, SQ_CHAIN_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}
}  // namespace

template <int L>
__global__ __launch_bounds__(256) void k_chain(const ChainArgs<L> c) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // 4 x 256 used; SQ_CHAIN_LDS_BYTES requested: one workgroup per CU
  const int tid = threadIdx.x;
  const int xcc = blockIdx.x & 7, rank = blockIdx.x >> 3, size = gridDim.x >> 3;
  // The ~1 KB argument block is read field by field where it is used (~130 scalar loads, most followed by a wait): touch
  // every 64-byte line of it once, up front and all in flight together, so that those loads are scalar-cache hits instead
  // of ~15 serialised misses (measured: 7 us of a 17 us chain)
  {
    const unsigned* ka = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < (int)((sizeof(ChainArgs<L>) + 63) / 64); ++i) acc ^= ka[i * 16];
    if (acc == 0x9e3779b9u && c.R < 0) *c.status = (int)acc;  // never true: keeps the loads alive
  }
  unsigned long long t_start = 0;
  if (c.prof_ts != nullptr && tid == 0) t_start = wall_clock64();
  // Placement check.  What the teams need is that the workgroups with equal b % 8 share an XCD; the dispatcher deals
  // workgroups round-robin over the XCDs but the XCD of workgroup 0 varies from launch to launch, so the test is that the
  // rotation (hardware XCC id - b) mod 8 is the SAME for every workgroup of this launch: each one ORs its rotation bit
  // into a word of the launch's counter block and looks at what was there before.
  if (tid == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    const unsigned rot = ((x & 0xf) - (unsigned)xcc) & 7u;
    const unsigned old = __hip_atomic_fetch_or(c.bar + 7 * 64 + 32, 1u << rot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old & ~(1u << rot)) != 0u) __hip_atomic_store(c.status, SQ_CHAIN_PLACEMENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int rpt = (c.R + 7) >> 3;  // rows per team
  const int m0 = xcc * rpt, m1 = min(c.R, m0 + rpt);
  if (m0 < m1) {
    unsigned* bar = c.bar + xcc * 64;  // one 256-byte block per team: two XCDs must never share an L2 line (128 B)
    TilePre pre;
    pre.task = -1;
    run_layer<false>(c.l[0], m0, m1, rank, size, pre, red);   // inputs come from earlier launches: ordinary loads
    if (c.prof_ts != nullptr && blockIdx.x == 0 && tid == 0) c.prof_ts[2 * 4096] = wall_clock64() - t_start;  // phase stamps of workgroup 0
    prefetch_next(c.l[1], m0, m1, rank, pre);
    team_arrive_wait(bar, 1u * (unsigned)size, c.status);
    if (c.prof_ts != nullptr && blockIdx.x == 0 && tid == 0) c.prof_ts[3 * 4096] = wall_clock64() - t_start;
    run_layer<true>(c.l[1], m0, m1, rank, size, pre, red);
    if (c.prof_ts != nullptr && blockIdx.x == 0 && tid == 0) c.prof_ts[4 * 4096] = wall_clock64() - t_start;
    if (L > 2) {
      prefetch_next(c.l[L > 2 ? 2 : 0], m0, m1, rank, pre);
      team_arrive_wait(bar, 2u * (unsigned)size, c.status);
      run_layer<true>(c.l[L > 2 ? 2 : 0], m0, m1, rank, size, pre, red);
    }
  }
  if (c.prof_ts != nullptr) {
    __syncthreads();
    if (tid == 0) {
      atomicMin(c.prof_ts, t_start);
      atomicMax(c.prof_ts + 4096, wall_clock64());  // end slots follow the PROF_MAX start slots
    }
  }
}

constexpr int SQ_CHAIN_LDS_BYTES = 96 * 1024;  // > half of a CU's 160 KB: at most one workgroup of the chain per CU
static unsigned rmul_of(int rdiv) { return rdiv <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)rdiv) + 1u; }

template <int L>
static int launch(const LinArgs* la, const PackedLayer* const* pl, int R, unsigned* bar, int* status, unsigned long long* prof_ts,
                  hipStream_t s) {
  ChainArgs<L> c;
  for (int i = 0; i < L; ++i) {
    LinArgs a = la[i];
    for (int j = 0; j < a.nseg; ++j) {
      a.seg[j].rmul = rmul_of(a.seg[j].rdiv);
      const LinSeg& sg = a.seg[j];
      if ((reinterpret_cast<uintptr_t>(sg.p) & 15) != 0 || (sg.ld & 3) != 0 || sg.width < 1 || sg.rdiv < 1) return -5;  // A-operand contract
    }
    a.add_rmul = rmul_of(a.add_rdiv);
    c.l[i].a = a; c.l[i].kc = pl[i]->kc; c.l[i].nt = pl[i]->nt;
  }
  c.R = R; c.bar = bar; c.status = status; c.prof_ts = prof_ts;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_chain<L>, hipFuncAttributeMaxDynamicSharedMemorySize, SQ_CHAIN_LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_chain<L>, dim3(256), dim3(256), SQ_CHAIN_LDS_BYTES, s, c);
  return 0;
}

int sq_launch_chain(const LinArgs* layers, const PackedLayer* const* packed_layers, int n_layers, int R, unsigned* bar, int* status,
                    unsigned long long* prof_ts, hipStream_t s) {
  if (n_layers == 2) return launch<2>(layers, packed_layers, R, bar, status, prof_ts, s);
  if (n_layers == 3) return launch<3>(layers, packed_layers, R, bar, status, prof_ts, s);
  return -1;
}
