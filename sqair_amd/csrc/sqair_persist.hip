// XCD-persistent execution of the frame loop.
//
// Rows (b' = sequence x particle) never interact inside the T-frame pass, so a row group can live on ONE XCD for the
// whole pass: its activations are exchanged through that XCD's own L2 (plain stores; L1-bypassing `sc1` loads, because a
// CU's vector L1 is never refreshed by other CUs' stores) and consecutive layers are separated by a per-XCD arrival
// counter instead of a device-wide kernel boundary (~1.5 us + launch ramp).  tools/xcd_team.hip prices it: a 32-row x
// 256 x 256 dependent layer costs 1.75 us this way against 3.6 us as its own graph node.
//
// One launch, one workgroup per CU.  Every workgroup reads HW_REG_XCC_ID and registers with its XCD's team; the teams
// split the row tiles (16 rows) among themselves; then all workgroups walk the same op list — the launch sequence
// sq_forward_impl would have issued for the frame loop, recorded instead of launched — each doing the virtual blocks
// of an op that belong to its team's rows, followed by a team barrier.  No placement assumption is needed for
// correctness: team membership comes from the hardware register, all spins are bounded (a time-out raises the abort
// flag and every workgroup leaves), and ops only ever read rows of their own team.
#include "sqair_internal.h"
#include "sqair_persist.h"
#include "sqair_rowops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- loads of data written earlier in this launch by OTHER workgroups: bypass the L1 ------------------------------
__device__ __forceinline__ float ldf(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte sc1 loads are only reachable through inline asm, which the compiler does not track: the wait is part of the
// same asm statement, so no register copy of a result can be scheduled before the data has landed
__device__ __forceinline__ void ld4x4_sc1(const float* p0, const float* p1, const float* p2, const float* p3, f32x4& v0, f32x4& v1,
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
__device__ __forceinline__ f32x4 ld4_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

struct LdSc1 {
  static __device__ __forceinline__ float f(const float* p) { return ldf(p); }
  static __device__ __forceinline__ sq_f32x4 f4(const float* p) { return ld4_sc1(p); }
  static __device__ __forceinline__ void f4x4(const float* p0, const float* p1, const float* p2, const float* p3, sq_f32x4& v0,
                                              sq_f32x4& v1, sq_f32x4& v2, sq_f32x4& v3) {
    ld4x4_sc1(p0, p1, p2, p3, v0, v1, v2, v3);
  }
};

struct Team {
  int rank, size;      // this workgroup inside its XCD team
  int r0, r1;          // rows b' owned by the team
  unsigned* bar;       // the team's arrival counter
  unsigned phase;      // barriers passed so far
  int* abort_flag;
};

__device__ __forceinline__ bool team_barrier(Team& tm) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
  __syncthreads();
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    ++tm.phase;
    __hip_atomic_fetch_add(tm.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = tm.phase * (unsigned)tm.size;
    int ok = 1, spin = 0;
    while (__hip_atomic_load(tm.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spin > (1 << 21) || ((spin & 1023) == 0 && __hip_atomic_load(tm.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(tm.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

// ---- dense layer over the team's rows [m0, m1) ----------------------------------------------------------------------
// (1) workgroup per 16x16 tile, 4 waves split K (lowest latency; arithmetic order of k_linear): used when the op has
//     at most one tile per workgroup of the team
__device__ void x_linear_wg(const LinArgs& a, int kc_total, int n_tiles, int m0, int m1, const Team& tm, float* red) {
  const int tasks = ((m1 - m0 + 15) >> 4) * n_tiles;
  for (int task = tm.rank; task < tasks; task += tm.size)
    x_linear_tile<LdSc1>(a, kc_total, task % n_tiles, m0 + (task / n_tiles) * 16, m1, red);
}

// (2) wavefront per 16x16 tile, whole K in the wave (no LDS, no workgroup barrier): 4 tiles in flight per workgroup; used
//     when the op has more tiles than the team has workgroups
__device__ void x_linear_wave(const LinArgs& a, int kc_total, int n_tiles, int m0, int m1, const Team& tm) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int mt = (m1 - m0 + 15) >> 4;
  const int tasks = mt * n_tiles;
  const int nwaves = tm.size * 4;
  for (int task = tm.rank + tm.size * wave; task < tasks; task += nwaves) {  // wave w of workgroup r: tasks r + size*w, ...
    const int tile_n = task % n_tiles, mbase = m0 + (task / n_tiles) * 16;
    const int arow = min(mbase + (lane & 15), m1 - 1);
    const int n = tile_n * 16 + (lane & 15);
    const int nc = min(n, a.N - 1);
    const float* pb = a.bias + nc;
    const bool use_add = a.add != nullptr && nc < a.add_n;
    const bool g1 = a.epi == EPI_GRU1 && nc >= a.nh && nc < 2 * a.nh;
    const bool g2 = a.epi == EPI_GRU2;
    const float p_bias = *pb;
    const float p_scale = a.scale_ptr != nullptr ? *a.scale_ptr : 1.0f;
    float p_add[4], p_e0[4], p_e1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // this lane's 4 outputs: rows mbase + 4 kq + i, column n
      const int mc = min(mbase + 4 * kq + i, m1 - 1);
      const int mcd = a.add_rmul ? (int)__umulhi((unsigned)mc, a.add_rmul) : mc;
      const float* pa = use_add ? a.add + (size_t)mcd * a.add_ld + nc : pb;
      const float* pe0 = g1 ? a.e0 + (size_t)mc * a.e0_ld + (nc - a.nh) : (g2 ? a.e0 + (size_t)mc * a.e0_ld + nc : pb);
      const float* pe1 = g2 ? a.e1 + (size_t)mc * a.e1_ld + nc : pb;
      p_add[i] = ldf(pa); p_e0[i] = ldf(pe0); p_e1[i] = ldf(pe1);
    }
    SQ_XSEGS(a, arow)
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + ((size_t)tile_n * kc_total) * 64 + lane;
    const f32x4* __restrict__ wz = reinterpret_cast<const f32x4*>(a.wzero) + lane;
    constexpr int NCH = 8;
#pragma unroll 1
    for (int base = 0; base < kc_total; base += NCH) {
      f32x4 av[NCH], bv[NCH];
      const float* ap[NCH];
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const bool valid = base + j < kc_total;
        const int g = valid ? base + j : 0;
        ap[j] = SQ_XAPTR(g, kq);
        bv[j] = *(valid ? wp + (size_t)g * 64 : wz);
      }
      ld4x4_sc1(ap[0], ap[1], ap[2], ap[3], av[0], av[1], av[2], av[3]);
      ld4x4_sc1(ap[4], ap[5], ap[6], ap[7], av[4], av[5], av[6], av[7]);
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
      const int m = mbase + 4 * kq + i;
      if (m < m1 && n < a.N) x_epilogue(a, m, n, accv[i] + p_bias + (use_add ? p_add[i] : 0.0f), p_e0[i], p_e1[i], p_scale);
    }
  }
}

__device__ __forceinline__ void x_linear(const LinArgs& a, int kc_total, int n_tiles, int m0, int m1, const Team& tm, float* red) {
  const int tasks = ((m1 - m0 + 15) >> 4) * n_tiles;
  if (tasks <= tm.size) x_linear_wg(a, kc_total, n_tiles, m0, m1, tm, red);
  else x_linear_wave(a, kc_total, n_tiles, m0, m1, tm);
}

// ---- tail of a slot for 16 rows (arithmetic of k_slot_tail) -----------------------------------------------------------
__device__ void x_tail16(const TailArgs& a, const Dims& d, int row0, int rlim, float* smem) {  // rows [row0, min(row0 + 16, rlim))
  constexpr int ZLD = 68;
  float* zt = smem;                 // 16 * ZLD
  float* rs = smem + 16 * ZLD;      // 4 * 16
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4;
  const int nw = d.nw, nsp = d.nh / 2;
  const int n_tiles = nsp / 16;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(a.wp);
  f32x4 bv[2][4];
  float sp[2][4], w2v[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tile = min(wave + 4 * t, n_tiles - 1);
    const int col = tile * 16 + (lane & 15);
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[t][c] = wp4[(size_t)(tile * 4 + c) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) sp[t][i] = ldf(a.s1p + (size_t)min(row0 + 4 * kq + i, rlim - 1) * a.s1p_ld + col);
    w2v[t] = a.flat[a.w2_off + col];
  }
  const int pr = min(row0 + (tid & 15), rlim - 1);
  const float b2 = a.flat[a.b2_off];
  const float u = a.noise[(((size_t)pr * 2 + (a.is_disc ? 1 : 0)) * d.N + a.slot) * d.nzw + 4 + nw];
  float prev;
  if (a.is_disc) prev = a.slot == 0 ? 1.0f : ldf(a.rec_new + ((size_t)pr * d.N + a.slot - 1) * rec::W + rec::PRES);
  else prev = ldf(a.rec_prev + ((size_t)pr * d.N + a.slot) * rec::W + rec::PRES);
  constexpr int EPT = 4;
  const int nel = 16 * nw;
  float v_loc[EPT], v_sc[EPT], v_eps[EPT], v_h[EPT][5], v_tm1[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = min(tid + 256 * q, nel - 1);
    const int rr = e / nw, c = e - rr * nw;
    const int r = min(row0 + rr, rlim - 1);
    v_loc[q] = ldf(a.enc + (size_t)r * a.enc_ld + c);
    v_sc[q] = ldf(a.enc + (size_t)r * a.enc_ld + nw + c);
    v_eps[q] = a.noise[(((size_t)r * 2 + (a.is_disc ? 1 : 0)) * d.N + a.slot) * d.nzw + 4 + c];
    if (!a.is_disc) {
      const float* hr = a.hraw + (size_t)r * a.h_ld;
#pragma unroll
      for (int g = 0; g < 5; ++g) v_h[q][g] = ldf(hr + g * nw + c);
      v_tm1[q] = ldf(a.rec_prev + ((size_t)r * d.N + a.slot) * rec::W + rec::WHAT + c);
    } else {
#pragma unroll
      for (int g = 0; g < 5; ++g) v_h[q][g] = 0.0f;
      v_tm1[q] = 0.0f;
    }
  }
  for (int i = tid; i < 16 * ZLD; i += 256) {
    const int c = i % ZLD;
    if (c < rec::WHAT || c >= rec::WHAT + nw) zt[i] = 0.0f;
  }
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = tid + 256 * q;
    if (e < nel) {
      const int rr = e / nw, c = e - rr * nw;
      float loc, sc;
      if (a.is_disc) {
        loc = v_loc[q];
        sc = v_sc[q];
      } else {
        const float t_loc = v_h[q][0];
        const float t_scale = sq_softplus(v_h[q][1]) + 1e-2f;
        const float fg = sq_sigmoid(v_h[q][2]) * 0.9999f;
        const float ig = sq_sigmoid(v_h[q][3]) * 0.9999f;
        const float tg = sq_sigmoid(v_h[q][4]) * 0.9999f;
        loc = fg * v_tm1[q] + (1.0f - ig) * v_loc[q] + (1.0f - tg) * t_loc;
        sc = (1.0f - ig) * v_sc[q] + (1.0f - tg) * t_scale;
      }
      const float what = loc + sc * v_eps[q];
      zt[rr * ZLD + rec::WHAT + c] = what;
      if (row0 + rr < rlim) {
        float* rn = a.rec_new + ((size_t)(row0 + rr) * d.N + a.slot) * rec::W;
        rn[rec::WHAT + c] = what;
        rn[rec::WHAT_LOC + c] = loc;
        rn[rec::WHAT_SCALE + c] = sc;
      }
    }
  }
  __syncthreads();
  f32x4 acc[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(&zt[(lane & 15) * ZLD + 16 * c + 4 * kq]);
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
        if (a.s1h_out != nullptr && row0 + 4 * kq + i < rlim)
          a.s1h_out[(size_t)(row0 + 4 * kq + i) * a.s1h_ld + (wave + 4 * t) * 16 + (lane & 15)] = hv;
      }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    part[i] = v;
  }
  if ((lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rs[wave * 16 + 4 * kq + i] = part[i];
  }
  __syncthreads();
  if (tid < 16 && row0 + tid < rlim) {
    const float raw = rs[0 * 16 + tid] + rs[1 * 16 + tid] + rs[2 * 16 + tid] + rs[3 * 16 + tid] + b2;
    const float logit = prev * raw + (prev - 1.0f) * 88.0f;
    const float prob = sq_sigmoid(logit);
    float* rn = a.rec_new + ((size_t)(row0 + tid) * d.N + a.slot) * rec::W;
    rn[rec::PRES] = (u < prob ? 1.0f : 0.0f) * prev;
    rn[rec::LOGIT] = logit;
    rn[rec::PROB] = prob;
  }
  __syncthreads();
}

// ---- DeepSets summary of one row (k_latent_sum) ---------------------------------------------------------------------
__device__ void x_latsum_row(const XLatArgs& a, const Dims& d, int r) {
  for (int n = threadIdx.x; n < d.nh; n += blockDim.x) {
    float acc = 0.0f;
    for (int k = 0; k < d.N; ++k)
      acc += ldf(a.f + ((size_t)r * d.N + k) * d.nh + n) * ldf(a.rec_p + ((size_t)r * d.N + k) * rec::W + rec::PRES);
    a.c[(size_t)r * d.nh + n] = acc;
  }
}

// ---- slot compaction of one row (k_compact) -------------------------------------------------------------------------
__device__ void x_compact_row(const CompactArgs& a, const POff& po, const Dims& d, int r, float* smem) {
  int* src_s = reinterpret_cast<int*>(smem);   // SQ_MAXN
  float* id_s = smem + SQ_MAXN;                // SQ_MAXN
  const int tid = threadIdx.x;
  const int N = d.N, nh = d.nh;
  if (tid < 64) {
    const int sl = tid;
    const bool in = sl < 2 * N;
    const float last = ldf(a.last_id_prev + r);
    float p = 0.0f, pid = -1.0f;
    if (in) {
      p = sl < N ? ldf(a.rec_p + ((size_t)r * N + sl) * rec::W + rec::PRES) : ldf(a.rec_d + ((size_t)r * N + (sl - N)) * rec::W + rec::PRES);
      if (sl < N) pid = ldf(a.rec_prev + ((size_t)r * N + sl) * rec::W + rec::ID);
    }
    const unsigned long long all = (2 * N >= 64) ? ~0ull : ((1ull << (2 * N)) - 1ull);
    const unsigned long long present = __ballot(in && p != 0.0f) & all;
    const unsigned long long below = (1ull << sl) - 1ull;
    const int n_present = __popcll(present);
    const unsigned long long disc_bits = present >> N;
    float id;
    if (sl < N) id = pid * p - (1.0f - p);
    else {
      const int cum = __popcll(disc_bits & ((2ull << (sl - N)) - 1ull));
      id = ((float)cum + last) * p - (1.0f - p);
    }
    if (in) {
      const int dst = (p != 0.0f) ? __popcll(present & below) : n_present + __popcll(~present & all & below);
      if (dst < N) {
        src_s[dst] = sl;
        id_s[dst] = id;
        if (a.src_out != nullptr) a.src_out[(size_t)r * N + dst] = sl;
      }
    }
    if (sl == 0) a.last_id_next[r] = last + (float)__popcll(disc_bits);
  }
  __syncthreads();
  const size_t tr = (size_t)a.t * d.R + r;
  const int per = rec::W + 2 * nh;
  for (int e = tid; e < N * per; e += 256) {
    const int dst = e / per, i = e - dst * per;
    const int sidx = src_s[dst];
    const bool prop = sidx < N;
    const int ss = prop ? sidx : sidx - N;
    if (i < rec::W) {
      const float* rsrc = (prop ? a.rec_p : a.rec_d) + ((size_t)r * N + ss) * rec::W;
      a.rec_next[((size_t)r * N + dst) * rec::W + i] = (i == rec::ID) ? id_s[dst] : ldf(rsrc + i);
    } else if (i < rec::W + nh) {
      const int q = i - rec::W;
      a.temporal_next[((size_t)r * N + dst) * nh + q] =
          prop ? ldf(a.temporal_p + ((size_t)r * N + ss) * nh + q) : a.flat[po.temporal_init + q];
    } else {
      const int q = i - rec::W - nh;
      a.prior_next[((size_t)r * N + dst) * nh + q] = prop ? ldf(a.prior_p + ((size_t)r * N + ss) * nh + q) : a.flat[po.prior_init + q];
    }
  }
  for (int e = tid; e < N * 64; e += 256) {
    const int dst = e >> 6, c = e & 63;
    const int sidx = src_s[dst];
    const float* rsrc = sidx < N ? a.rec_p + ((size_t)r * N + sidx) * rec::W : a.rec_d + ((size_t)r * N + (sidx - N)) * rec::W;
    const size_t o = tr * N + dst;
    if (c < d.nw) {
      if (a.out.what) a.out.what[o * d.nw + c] = ldf(rsrc + rec::WHAT + c);
      if (a.out.what_loc) a.out.what_loc[o * d.nw + c] = ldf(rsrc + rec::WHAT_LOC + c);
      if (a.out.what_scale) a.out.what_scale[o * d.nw + c] = ldf(rsrc + rec::WHAT_SCALE + c);
    }
    if (c < 4) {
      if (a.out.where) a.out.where[o * 4 + c] = ldf(rsrc + rec::WHERE + c);
      if (a.out.where_loc) a.out.where_loc[o * 4 + c] = ldf(rsrc + rec::WHERE_LOC + c);
      if (a.out.where_scale) a.out.where_scale[o * 4 + c] = ldf(rsrc + rec::WHERE_SCALE + c);
    }
    if (c == 0) {
      if (a.out.presence_prob) a.out.presence_prob[o] = ldf(rsrc + rec::PROB);
      if (a.out.presence) a.out.presence[o] = ldf(rsrc + rec::PRES);
      if (a.out.presence_logit) a.out.presence_logit[o] = ldf(rsrc + rec::LOGIT);
      if (a.out.obj_id) a.out.obj_id[o] = id_s[dst];
    }
  }
  if (tid == 0 && a.out.num_steps_per_sample) {
    float ns = 0.0f;
    for (int dst = 0; dst < N; ++dst) {
      const int sidx = src_s[dst];
      ns += sidx < N ? ldf(a.rec_p + ((size_t)r * N + sidx) * rec::W + rec::PRES) : ldf(a.rec_d + ((size_t)r * N + (sidx - N)) * rec::W + rec::PRES);
    }
    a.out.num_steps_per_sample[tr] = ns;
  }
  __syncthreads();
}

// ---- the persistent kernel ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_xcd_persistent(const XOp* __restrict__ prog, int n_ops, const POff po, const Dims d,
                                                        unsigned* sync /* [16 team counts | gcount | abort | 16 x 32 barrier words] */,
                                                        unsigned long long* tstamp /* optional: 2 stamps per op by (first team, rank 0) */) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // >= 1152 floats used; sized to keep one workgroup per CU
  float* red = dyn;  // GEMM: 1024; tail: 16*68 + 64; crop: 4 + 4G; compact: 16
  __shared__ int s_team[5];
  unsigned* team_count = sync;
  unsigned* gcount = sync + 16;
  int* abort_flag = reinterpret_cast<int*>(sync + 17);
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 0xf;
    const unsigned rank = atomicAdd(&team_count[x], 1u);
    __hip_atomic_fetch_add(gcount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spin = 0, ok = 1;
    while (__hip_atomic_load(gcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
      __builtin_amdgcn_s_sleep(2);
      if (++spin > (1 << 22)) { ok = 0; __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    // row tiles are dealt to the teams that actually exist, in XCC order
    int nteams = 0, my_index = 0, my_size = 0;
    for (int i = 0; i < 16; ++i) {
      const unsigned c = __hip_atomic_load(&team_count[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c > 0) {
        if (i == (int)x) { my_index = nteams; my_size = (int)c; }
        ++nteams;
      }
    }
    // rows are dealt evenly (a team's range need not be a multiple of the 16-row MFMA tile: its last tile is ragged)
    const int base = d.R / nteams, rem = d.R % nteams;
    const int q0 = my_index * base + min(my_index, rem), q1 = q0 + base + (my_index < rem ? 1 : 0);
    s_team[0] = (int)rank; s_team[1] = ok ? my_size : 0; s_team[2] = q0; s_team[3] = q1;
    s_team[4] = (int)x;
  }
  __syncthreads();
  Team tm;
  tm.rank = s_team[0]; tm.size = s_team[1]; tm.r0 = s_team[2]; tm.r1 = s_team[3];
  tm.bar = sync + 64 + 32 * (unsigned)s_team[4];
  tm.phase = 0; tm.abort_flag = abort_flag;
  __syncthreads();
  if (tm.size == 0) return;
  const int rows = tm.r1 - tm.r0;
  // op descriptors travel global -> registers (one dword per thread, requested while the previous op runs) -> LDS ring:
  // fetching them at the point of use would put 2-3 dependent L2 round trips in front of every op
  constexpr int OPW = (int)(sizeof(XOp) / 4);
  static_assert(sizeof(XOp) % 4 == 0 && OPW <= 256, "XOp must fit one dword per thread");
  __shared__ __attribute__((aligned(16))) unsigned op_ring[2][OPW];
  const unsigned* prog_w = reinterpret_cast<const unsigned*>(prog);
  if (threadIdx.x < OPW) op_ring[0][threadIdx.x] = prog_w[threadIdx.x];
  __syncthreads();
  for (int i = 0; i < n_ops; ++i) {
    const XOp& op = *reinterpret_cast<const XOp*>(op_ring[i & 1]);
    unsigned nxt = 0;
    if (i + 1 < n_ops && threadIdx.x < OPW) nxt = prog_w[(size_t)(i + 1) * OPW + threadIdx.x];
    const bool stamp = tstamp != nullptr && tm.rank == 0 && tm.r0 == 0 && threadIdx.x == 0;
    if (stamp) tstamp[2 * i] = wall_clock64();
    if (rows > 0) {
      switch (op.type) {
        case XOP_LINEAR: {
          const int per_r = op.u.lin.M / d.R;  // 1 for slot launches, N for the per-frame batched layers
          x_linear(op.u.lin, op.kc, op.nt, tm.r0 * per_r, tm.r1 * per_r, tm, red);
          break;
        }
        case XOP_CROP: {
          const int tasks = rows * op.nslots;
          for (int task = tm.rank; task < tasks; task += tm.size)
            x_crop_row<LdSc1>(op.u.crop, po, d, tm.r0 + task / op.nslots, op.u.crop.mode == CROP_PROP1 ? task % op.nslots : op.u.crop.slot, red, false);
          break;
        }
        case XOP_TAIL:
          for (int task = tm.rank; task * 16 < rows; task += tm.size) x_tail16(op.u.tail, d, tm.r0 + task * 16, tm.r1, red);
          break;
        case XOP_LATSUM:
          for (int task = tm.rank; task < rows; task += tm.size) x_latsum_row(op.u.lat, d, tm.r0 + task);
          break;
        case XOP_COMPACT:
          for (int task = tm.rank; task < rows; task += tm.size) x_compact_row(op.u.comp, po, d, tm.r0 + task, red);
          break;
        default: break;
      }
    }
    const int sync_after = op.sync_after;
    if (stamp) tstamp[2 * i + 1] = wall_clock64();
    if (threadIdx.x < OPW) op_ring[(i + 1) & 1][threadIdx.x] = nxt;
    if (sync_after) {
      if (!team_barrier(tm)) return;
    } else {
      __syncthreads();
    }
  }
}

int sq_launch_persistent(const XOp* prog_dev, int n_ops, POff po, Dims d, unsigned* sync, int n_cu, hipStream_t s,
                         unsigned long long* tstamp) {
  const size_t shm = 72 * 1024;  // more than half of the 160 KB LDS: at most one workgroup per CU
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_xcd_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    (void)hipGetLastError();
    attr = true;
  }
  hipLaunchKernelGGL(k_xcd_persistent, dim3(n_cu), dim3(256), shm, s, prog_dev, n_ops, po, d, sync, tstamp);
  return 0;
}
