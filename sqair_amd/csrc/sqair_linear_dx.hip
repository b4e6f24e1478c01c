// dX GEMM of the backward pass: out = (dPre W^T [+ addend]) * act'(saved), with the output columns routed to up to
// three destinations.  Same weight-stationary split-K MFMA structure as k_linear (sqair_linear_kernel.inc) on the
// TRANSPOSED packs, but the epilogue does what the reverse sweep would otherwise spend separate launches on:
//   * column ranges of the (virtually concatenated) input gradient go straight to their consumers' gradient
//     buffers (a z-record segment into the gradient record, the hidden-state segment into the RNN accumulator, ...),
//     each either overwriting or accumulating (addend pointer = destination);
//   * the activation derivative of the producing layer is applied from its saved OUTPUT (elu / tanh / sigmoid /
//     softplus), so the result is the next dPre.
// At B' = 160 rows every launch costs ~5 us regardless of its size (dependent-launch boundary + latency), so the
// backward chain is priced in launches: this kernel halves them.
#include "sqair_common.h"
#include "sqair_dx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dx_dact(float g, float o, int act) {
  switch (act) {
    case ACT_ELU: return g * (o > 0.0f ? 1.0f : o + 1.0f);
    case ACT_TANH: return g * (1.0f - o * o);
    case ACT_SIGMOID: return g * (o * (1.0f - o));
    case ACT_SOFTPLUS_MIN: return g * (1.0f - sq_exp(-(o - 1e-2f)));
    default: return g;
  }
}

template <int NCH, int GRU = 0>
__global__ __launch_bounds__(256) void k_linear_dx(const float* __restrict__ dpre0, const float* __restrict__ wp0, const int ld0,
                                                   const int width0, const int M0, const int kc_total,
                                                   const float* __restrict__ wzero0, const DxArgs a SQ_TLP) {
  SQ_TL_SCOPE;
  // leading scalars (copies of a.dpre / a.wp / a.ld / a.width / a.M / a.wzero): preloaded into SGPRs with the launch, so the
  // operand loads are issued before the s_load of the struct has returned; the epilogue operands, which need the struct, are
  // requested right behind them and still ahead of the MFMAs (see sqair_linear_kernel.inc)
  __shared__ float red[4 * 256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile_n = blockIdx.x, tile_m = blockIdx.y;
  const int arow = min(tile_m * 16 + (lane & 15), M0 - 1);
  const int kq = lane >> 4;
  const float* rp = dpre0 + (size_t)arow * ld0;
  const int lim = ((width0 + 3) & ~3) - 4;
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(wp0) + ((size_t)tile_n * kc_total) * 64 + lane;
  const f32x4* __restrict__ wz = reinterpret_cast<const f32x4*>(wzero0) + lane;
  const int nmine = (kc_total - wave + 3) >> 2;  // this wave owns chunks g = wave + 4 i
  f32x4 av[NCH], bv[NCH];
#define SQ_DX_ISSUE(BASE)                                                                    \
  _Pragma("unroll") for (int j = 0; j < NCH; ++j) {                                          \
    const bool valid = (BASE) + j < nmine;                                                   \
    const int g = valid ? wave + 4 * ((BASE) + j) : wave;                                    \
    av[j] = *reinterpret_cast<const f32x4*>(rp + min(g * 16 + kq * 4, lim));                 \
    bv[j] = *(valid ? wp + (size_t)g * 64 : wz);                                             \
  }
#define SQ_DX_MFMA()                                                                         \
  _Pragma("unroll") for (int j = 0; j < NCH; ++j) {                                          \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc0, 0, 0, 0);            \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc1, 0, 0, 0);            \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc0, 0, 0, 0);            \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc1, 0, 0, 0);            \
  }
  SQ_DX_ISSUE(0)
  __builtin_amdgcn_sched_barrier(0);
  // ---- epilogue operands (addend, saved activation)
  const int m = tile_m * 16 + (tid >> 4);
  const int n = tile_n * 16 + (tid & 15);
  const int mc = min(m, a.M - 1);
  // The range of a column is the same for the whole workgroup (range starts are multiples of 16: the launcher checks), so it is
  // picked from the tile index alone: a WAVE-UNIFORM index into the by-value argument struct becomes scalar loads.  With the
  // per-lane column in the comparison every field of the range (pointers, leading dimensions, activation codes) was fetched
  // by a vector load from the argument segment followed by a full wait -- a dozen dependent round trips per launch.
  const int n_tile0 = tile_n * 16;
  int ri_ = 0;
  if (a.nranges > 1 && n_tile0 >= a.r[1].n0) ri_ = 1;
  if (a.nranges > 2 && n_tile0 >= a.r[2].n0) ri_ = 2;
  // the range as a by-value copy from a SCALAR index: every field arrives in one batch of scalar loads.  (As a reference through
  // a per-lane-looking index hipcc kept the struct's address in VGPRs and fetched each field when first used -- readfirstlane,
  // s_load_dword, wait -- under the `live` masks: eight dependent scalar round trips ahead of the epilogue operands' requests.)
  const int ri = __builtin_amdgcn_readfirstlane(ri_);
  const DxRange rg = a.r[ri];
  const bool live = m < a.M && n >= rg.n0 && n < rg.n1;
  const int c = live ? n - rg.n0 : 0;
  // activation codes as scalars NOW: left as `c < split ? rg.act_a : rg.act_b` at the use, the select became a per-lane ADDRESS
  // into the argument segment and a vector load + full wait behind the barrier
  const int act_a = __builtin_amdgcn_readfirstlane(rg.act_a), act_b = __builtin_amdgcn_readfirstlane(rg.act_b),
            act_split = __builtin_amdgcn_readfirstlane(rg.act_split);
  const float* dummy = wzero0;
  const float* pa = (live && rg.add != nullptr) ? rg.add + (size_t)mc * rg.add_ld + c : dummy;
  const float* ps = (live && rg.saved != nullptr) ? rg.saved + (size_t)mc * rg.saved_ld + c : dummy;
  const float p_add = *pa, p_saved = *ps;
  // (per-lane address on purpose: as a wave-uniform load hipcc sinks it to its use behind the barrier, where it costs every
  // launch a memory round trip of its own)
  float p_scale = (a.scale_ptr != nullptr ? a.scale_ptr : dummy + (tid & 15))[0];
  p_scale = a.scale_ptr != nullptr ? p_scale : 1.0f;
  const DxGru gr = a.gru;   // (by value: one batch of scalar loads; field by field they were fetched where first used, the last
                            // ones between the barrier and the stores)
  float q_g0 = 0.0f, q_g1 = 0.0f, q_h = 0.0f, q_dh = 0.0f;
  if (GRU != 0) {  // gate tapes of the GRU adjoint (columns 0 .. nh - 1 of range 0)
    const int cg = min(c, gr.nh - 1);
    q_g0 = gr.g0[(size_t)mc * gr.g0_ld + cg];
    if (GRU == 1) q_g1 = gr.g1[(size_t)mc * gr.g1_ld + cg];
    q_h = gr.hprev[(size_t)mc * gr.h_ld + cg];
    if (GRU == 2 || gr.acc_dh) q_dh = gr.d_h[(size_t)mc * gr.dh_ld + cg];
  }
  __builtin_amdgcn_sched_barrier(0);
  SQ_DX_MFMA()
  if (NCH == 18) {  // only the deepest instantiation loops (K > 1152)
#pragma unroll 1
    for (int base = NCH; base < nmine; base += NCH) {
      SQ_DX_ISSUE(base)
      __builtin_amdgcn_sched_barrier(0);
      SQ_DX_MFMA()
    }
  }
#undef SQ_DX_ISSUE
#undef SQ_DX_MFMA
  float* r = red + wave * 256;
  r[(4 * kq + 0) * 16 + (lane & 15)] = acc0.x + acc1.x;
  r[(4 * kq + 1) * 16 + (lane & 15)] = acc0.y + acc1.y;
  r[(4 * kq + 2) * 16 + (lane & 15)] = acc0.z + acc1.z;
  r[(4 * kq + 3) * 16 + (lane & 15)] = acc0.w + acc1.w;
  __syncthreads();
  if (live) {
    float v = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
    v *= p_scale;
    if (rg.add != nullptr) v += p_add;
    if (rg.saved != nullptr) v = dx_dact(v, p_saved, c < act_split ? act_a : act_b);
    if (GRU == 1) {
      const int nh = gr.nh;
      const float dz = v * (q_g1 - q_h) * q_g0 * (1.0f - q_g0), dc = v * q_g0 * (1.0f - q_g1 * q_g1);
      gr.dpre1[(size_t)m * gr.dp_ld + c] = dz;
      gr.dpre1[(size_t)m * gr.dp_ld + 2 * nh + c] = dc;
      if (gr.dup != nullptr) {
        gr.dup[(size_t)m * gr.dup_ld + c] = dz;
        if (gr.dup_h_off >= 0) gr.dup[(size_t)m * gr.dup_ld + gr.dup_h_off + c] = dc;
      }
      gr.d_h[(size_t)m * gr.dh_ld + c] = (gr.acc_dh ? q_dh : 0.0f) + v * (1.0f - q_g0);
    } else if (GRU == 2) {
      const float dr = v * q_h * q_g0 * (1.0f - q_g0);
      gr.dpre1[(size_t)m * gr.dp_ld + gr.nh + c] = dr;
      if (gr.dup != nullptr) gr.dup[(size_t)m * gr.dup_ld + c] = dr;
      gr.d_h[(size_t)m * gr.dh_ld + c] = q_dh + v * q_g0;
    } else {
      rg.dst[(size_t)m * rg.dst_ld + c] = v;
      if (rg.dst2 != nullptr) rg.dst2[(size_t)m * rg.dst2_ld + c] = v;
    }
  }
}

#define SQ_T2_NAME k_linear_dx_t2
#define SQ_T2_DX 1
#include "sqair_linear_t2.inc"
#undef SQ_T2_DX
#undef SQ_T2_NAME

int sq_launch_linear_dx(const DxArgs& a, int kc, int nt, hipStream_t s) {
  if ((reinterpret_cast<uintptr_t>(a.dpre) & 15) != 0 || (a.ld & 3) != 0 || a.width < 1 || a.nranges < 1 || a.nranges > 3) return -5;
  for (int i = 0; i < a.nranges; ++i)
    if ((a.r[i].n0 & 15) != 0) return -5;  // a column tile belongs to one range
  const dim3 g(nt, (a.M + 15) / 16);
  if (g.x == 0 || g.y == 0) return 0;
  const int per_wave = (kc + 3) / 4;
  if (a.gru.mode != 0) {  // GRU gate adjoints in the epilogue: one range [0, nh), K = nh or the what-head width (<= 16 chunks)
    if (a.nranges != 1 || a.r[0].n0 != 0 || a.r[0].n1 != a.gru.nh || a.r[0].saved != nullptr) return -6;
#define SQ_DXG(NCH, G) SQ_LAUNCH((k_linear_dx<NCH, G>), g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a)
#ifdef SQAIR_WIDE
    // (the wide build: K up to 5 x 128 head columns / 512 hidden units -- deeper instantiations of the same kernel)
    if (per_wave > 8) { if (a.gru.mode == 1) SQ_DXG(18, 1); else SQ_DXG(18, 2); return 0; }
    if (per_wave > 4) { if (a.gru.mode == 1) SQ_DXG(8, 1); else SQ_DXG(8, 2); return 0; }
#else
    if (per_wave > 4) return -6;
#endif
    if (a.gru.mode == 1) SQ_DXG(4, 1); else SQ_DXG(4, 2);
#undef SQ_DXG
    return 0;
  }
  // hundreds of rows and several column tiles (the transposes of the once-per-frame layers): 32 x 32 tiles (sqair_linear_t2.inc)
  static const int t2_rows = SQ_KNOB_INT("SQAIR_T2_ROWS", 512), t2_kc = SQ_KNOB_INT("SQAIR_T2_KC", 40);  // measurement knobs
  if (a.M >= t2_rows && nt >= 4 && kc >= t2_kc) {
    const dim3 g2((nt + 1) / 2, (a.M + 31) / 32);
#define SQ_DXT2(NB, NW) SQ_LAUNCH((k_linear_dx_t2<NB, NW>), g2, dim3(64 * NW), 0, s, a, kc, nt)
    if (per_wave <= 2) SQ_DXT2(2, 4); else if (per_wave <= 3 || per_wave == 5 || per_wave == 6) SQ_DXT2(3, 4); else SQ_DXT2(4, 4);
#undef SQ_DXT2
    return 0;
  }
  switch (per_wave) {
    case 1: SQ_LAUNCH(k_linear_dx<1>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 2: SQ_LAUNCH(k_linear_dx<2>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 3: SQ_LAUNCH(k_linear_dx<3>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 4: SQ_LAUNCH(k_linear_dx<4>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 5: SQ_LAUNCH(k_linear_dx<5>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 6: SQ_LAUNCH(k_linear_dx<6>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 7: SQ_LAUNCH(k_linear_dx<7>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 8: SQ_LAUNCH(k_linear_dx<8>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 9: SQ_LAUNCH(k_linear_dx<9>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    case 10: case 11: case 12:  // K = 768: the GRU gate GEMMs' transposes, all 24 operand loads of a wave in flight at once
      SQ_LAUNCH(k_linear_dx<12>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
    default:                    // K = 1152 (the loop-invariant pre-activation GEMM's transpose) and deeper (looped)
      SQ_LAUNCH(k_linear_dx<18>, g, dim3(256), 0, s, a.dpre, a.wp, a.ld, a.width, a.M, kc, a.wzero, a); break;
  }
  return 0;
}
