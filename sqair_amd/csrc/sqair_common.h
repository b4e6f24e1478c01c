// Internal declarations shared by the translation units of libsqair_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/sqair_hip.h"

#define SQ_CHECK_HIP(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      sq_set_error(h, std::string(#expr) + ": " + hipGetErrorString(_e));                    \
      return -2;                                                                             \
    }                                                                                        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Slot record: everything one object slot carries inside a frame, 168 floats (16-byte multiple).
// The first 56 floats (where | what | presence | logit) are the z-record the recurrences consume;
// weight rows are permuted at pack time so this memory order never has to match the reference's
// concat order.
// ---------------------------------------------------------------------------------------------
namespace rec {
#ifdef SQAIR_WIDE
// the wide build's record (libsqair_hip_wide.so: n_what up to 128): same fields, same order, room for 128 `what` entries
constexpr int NWMAX = 128;      // `what` entries the layout has room for
constexpr int WHERE = 0;        // 4
constexpr int WHAT = 4;         // n_what
constexpr int PRES = 132;
constexpr int LOGIT = 133;
constexpr int ZW = 134;         // width of the z-record segment (K chunks of 16: ZWP = 144 floats are read)
constexpr int ZWP = 144;
constexpr int WHERE_LOC = 144;  // 4
constexpr int WHERE_SCALE = 148; // 4
constexpr int WHAT_LOC = 152;   // 128
constexpr int WHAT_SCALE = 280; // 128
constexpr int PROB = 408;
constexpr int ID = 409;
constexpr int W = 416;
#else
constexpr int NWMAX = 50;
constexpr int WHERE = 0;        // 4
constexpr int WHAT = 4;         // n_what (<= 50 in this layout)
constexpr int PRES = 54;
constexpr int LOGIT = 55;
constexpr int ZW = 56;          // width of the z-record segment
constexpr int ZWP = 64;         // ... padded to whole K chunks of 16
constexpr int WHERE_LOC = 56;   // 4
constexpr int WHERE_SCALE = 60; // 4
constexpr int WHAT_LOC = 64;    // 50
constexpr int WHAT_SCALE = 114; // 50
constexpr int PROB = 164;
constexpr int ID = 165;
constexpr int W = 168;
#endif
// lanes over the `what` entries of a record: one pass in the product build (n_what <= 50 < 64 lanes), a loop in the wide one
#ifdef SQAIR_WIDE
#define SQ_WHAT_LANES(c, lane, nw) for (int c = (lane); c < (nw); c += 64)
#else
#define SQ_WHAT_LANES(c, lane, nw) for (int c = (lane), sq_once_ = 1; sq_once_ != 0 && c < (nw); sq_once_ = 0)
#endif
}  // namespace rec

enum Act { ACT_NONE = 0, ACT_ELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_SOFTPLUS_MIN = 4 };
enum Epi { EPI_ACT = 0, EPI_GRU1 = 1, EPI_GRU2 = 2 };

struct LinSeg {
  const float* p;  // base; row r reads p + (r / rdiv) * ld
  int ld;          // floats between consecutive (r / rdiv); 0 = broadcast one row
  int width;       // true number of inputs taken from this segment (K padded to 16 in the pack)
  int rdiv;
  unsigned rmul;   // filled by the launcher: 0 = rdiv 1, else floor(2^32 / rdiv) + 1 (row / rdiv == umulhi(row, rmul))
};

struct LinArgs {
  LinSeg seg[4];
  int nseg;
  const float* wp;    // packed weights of this layer: [n_tiles][k_chunks][64 lanes][4]
  const float* bias;  // packed bias [n_tiles*16]
  const float* wzero; // 256 zero floats (one chunk of zero weights) inside the packed buffer
  const float* add;   // optional pre-activation addend, applied for n < add_n
  int add_ld, add_rdiv, add_n;
  unsigned add_rmul;
  float* out;
  int out_ld;
  int M, N;
  int epi;
  int act_a, act_b, act_split;  // act_a for n < act_split, act_b otherwise
  const float* scale_ptr;       // optional device scalar multiplied after the activation
  float scale;                  // host scalar multiplied after the activation (1 = none)
  // GRU epilogues
  const float* e0; int e0_ld;   // h  (previous state)
  const float* e1; int e1_ld;   // z  (GRU2)
  float* o1; int o1_ld;         // r*h (GRU1)
  float* o2; int o2_ld;         // x W_h + b_h (GRU1)
  float* o3; int o3_ld;         // optional: r (GRU1) — kept for the backward pass; GRU2 keeps tanh(.) in o1
  int nh;
};

// One packed layer (host-side description)
struct PackedLayer {
  int64_t w_off;   // float offset into the packed buffer
  int64_t b_off;   // float offset of the packed bias
  int kc;          // number of 16-wide K chunks (sum over segments)
  int nt;          // number of 16-wide N tiles
  int N;           // true number of output columns
  std::vector<int> seg_width;  // true widths per segment
};

struct SqairHandle;
void sq_set_error(SqairHandle* h, const std::string& msg);
// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) on the CURRENT device, once per
// (kernel, device): the attribute belongs to the device's copy of the kernel, so a process that drives several devices (one
// handle each) has to set it on each of them.  Returns 0 or -2; the HIP error is not swallowed.
int sq_allow_big_lds(const void* kernel, int bytes);

// launchers (sqair_linear.hip)
// prof_ts (optional): device slot {min start, max end} of the launch in 100 MHz wall-clock ticks
int sq_launch_linear(const LinArgs& a, const PackedLayer& L, hipStream_t s, unsigned long long* prof_ts = nullptr);
int sq_launch_pack(const float* flat, float* packed_w, const int* idx, int64_t n, hipStream_t s);
int sq_launch_pack_bias(const float* flat, float* packed_b, const int* idxa, const int* idxb, int64_t n,
                        hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Per-dispatch timeline (libsqair_hip_timeline.so = this source with -DSQAIR_TIMELINE; bench.py's roofline, tools/timeline.py).
// In that build EVERY kernel carries one more argument (SQ_TLP) and every wave stamps {start, end} on the 100 MHz device wall
// clock (s_memrealtime, chip-wide) into ITS OWN 16-byte slot of a caller-provided buffer -- plain stores, no atomics, nothing
// shared between waves -- so that a replayed graph yields, per node, first-wave start and last-wave end: busy time per kernel,
// gap (dependent launch boundary) between kernels, and their sum = the step.  In the production build the three macros expand to
// nothing and the binary is unchanged.  Kernels are launched through SQ_LAUNCH everywhere.
// ---------------------------------------------------------------------------------------------
#ifdef SQAIR_TIMELINE
struct SqTl { unsigned long long* slot; unsigned gx, gy, nw, pad; };  // slot range + launch geometry
SqTl sq_tl_next(const char* kernel, dim3 grid, dim3 block);  // sqair_api.hip: next slot range of the active recording (or null)
#define SQ_TLP , const SqTl sq_tl
#define SQ_TL_SCOPE SqTlScope sq_tl_scope(sq_tl)
#define SQ_LAUNCH(kern, grid, block, lds, s, ...) \
  hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__, sq_tl_next(#kern, grid, block))
#ifdef __HIPCC__
struct SqTlScope {
  // What keeps the stamped step within ~3 % of the product library's (each item measured on the cfg-2 forward pass, 3.51 ms):
  //  * the end of the wave must not touch memory before its stamp store: grid / block dimensions read through the HIP built-ins
  //    are scalar loads from the dispatch packet which the compiler places at their use -- a memory round trip between the
  //    wave's last instruction and the store when used at the end (3.94 ms), a round trip ahead of the kernel's first operand
  //    loads when used at the start (3.86 ms).  The launch geometry therefore travels in the kernel argument (SqTl, filled by
  //    sq_tl_next on the host) and arrives with the kernel's other arguments (3.70 ms);
  //  * the flat thread id of a multi-dimensional workgroup (threadIdx.x + bdx * ...) made the compiler wait for the kernel's
  //    output stores (s_waitcnt vmcnt(0)) ahead of the stamp code; all kernels use 1-D workgroups, threadIdx.x it is (3.60 ms);
  //  * not the clock reads: the one at the start overlaps the argument loads (+0.03 ms), the one at the end is free next to the
  //    store; nor the number of stores (wave 0 only: -0.01 ms), their cache policy, or lines shared between workgroups.
  const SqTl tl;
  const unsigned long long t0;
  static __device__ __forceinline__ unsigned long long clock() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
  }
  __device__ __forceinline__ explicit SqTlScope(const SqTl tl_) : tl(tl_), t0(__builtin_amdgcn_s_memrealtime()) {}
  __device__ __forceinline__ ~SqTlScope() {
    const uintptr_t b = (uintptr_t)tl.slot;
    if (b == 0) return;
    const unsigned tid = threadIdx.x;   // every kernel of the library uses 1-D workgroups (checked by sq_tl_next)
    if ((tid & 63u) != 0u) return;
    const unsigned wg = blockIdx.x + tl.gx * (blockIdx.y + tl.gy * blockIdx.z);
    // tl.nw = slots per workgroup (its waves, padded to a whole 128-byte line)
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) u64x2 gvec;
    gvec* p = (gvec*)b + ((size_t)wg * tl.nw + (tid >> 6));
    u64x2 v;
    v.x = t0;
    v.y = clock();
    *p = v;
  }
};
#endif
#else
#define SQ_TLP
#define SQ_TL_SCOPE
#ifdef SQAIR_KNOBS
// knob build: SQAIR_SYNC_LAUNCHES=1 names every kernel on stderr before it is launched and waits for it (which launch faults)
#include <cstdio>
#include <cstdlib>
#define SQ_LAUNCH(kern, grid, block, lds, s, ...)                                              \
  do {                                                                                         \
    static const bool sq_sync_ = getenv("SQAIR_SYNC_LAUNCHES") != nullptr;                     \
    if (sq_sync_) { fprintf(stderr, "launch %s\n", #kern); fflush(stderr); }                   \
    hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__);                                \
    if (sq_sync_) (void)hipStreamSynchronize(s);                                               \
  } while (0)
#else
#define SQ_LAUNCH(kern, grid, block, lds, s, ...) hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__)
#endif
#endif

// Measurement knobs (tile shapes, fusion switches, dump files) are read from the environment ONLY in a library built with
// -DSQAIR_KNOBS (tools/: `python sqair_amd/csrc/build.py --knobs` -> tools/bin/libsqair_hip_knobs.so).  The production library
// compiles every knob to its default, so no environment variable can change -- or remove -- work inside a timed region.
#ifdef SQAIR_KNOBS
#define SQ_KNOB_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define SQ_KNOB_SET(name) (getenv(name) != nullptr)
#define SQ_KNOB_STR(name) getenv(name)
#else
#define SQ_KNOB_INT(name, dflt) (dflt)
#define SQ_KNOB_SET(name) false
#define SQ_KNOB_STR(name) ((const char*)nullptr)
#endif

// ---------------------------------------------------------------------------------------------
// device math (exact-ish fp32; no fast-math so that parity with the fp64 oracle holds to ~1e-6)
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// Activations of the dense-layer epilogues and the slot kernels, built on the hardware exp2 / log2 / rcp (1 ulp each) instead
// of the libm expansions: the libm versions (expm1f, tanhf, log1pf, expf: 40-80 instructions apiece, inlined into every
// epilogue) made up almost half of the dense kernel's code, and the kernels of the slot loop are sensitive to their size (a
// dependent node starts with a partly cold instruction cache: 1.2 KB less code in k_linear measured -0.3 us per launch).
// Absolute error <= ~1.7e-7 on outputs of order one (libm: ~6e-8), i.e. at the fp32 rounding level of the GEMM sums feeding them.
// NOT used for the spatial-transformer geometry (to_coords: the shift tanh is multiplied by (W - 1) / 2 pixels and, in the
// decoder's mask, by another factor 20 inside a sigmoid): crop, insert and their adjoints keep libm's tanhf there.
__device__ __forceinline__ float sq_exp(float x) {
  // e^x = 2^(x log2 e); the product is formed in two parts (t + r) so that its rounding error, which exp2 would amplify to
  // |x| * 6e-8 relative, is put back to first order: 2^(t + r) = 2^t (1 + r ln 2)
  // x is clamped to [-104, log(FLT_MAX)] (one v_med3): above it exp2 returns +inf and fma(inf, r ln 2, inf) is NaN whenever
  // r <= 0; far below it x log2 e overflows to -inf and r becomes inf, fma(0, inf, 0) = NaN.  Every user (sigmoid, tanh,
  // softplus, ELU) wants the saturated value: 1 / (1 + 3.4e38) = 0, 1 - 2 / 3.4e38 = 1, e^-104 = 0 in fp32.
  // A NaN argument stays NaN (v_med3 would return the smaller of the two bounds for it, i.e. e^-104: a NaN pre-activation would
  // leave every activation as a finite number and never reach the log-weights, which is what debug mode checks): one compare +
  // select on the way out.
  const float x0 = x;
  x = __builtin_amdgcn_fmed3f(x, -104.0f, 88.72283f);
  const float t = x * 1.44269504088896340736f;
  const float r = fmaf(x, 1.44269504088896340736f, -t) + x * 1.92596299112661746e-8f;
  const float e = __builtin_amdgcn_exp2f(t);
  return x0 != x0 ? x0 : fmaf(e, r * 0.69314718055994530942f, e);
}
__device__ __forceinline__ float sq_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float sq_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + sq_exp(-x)); }
// scale of the spatial transformer (to_coords): libm exponential and an IEEE divide, ~1 ulp.  The compact sigmoid's 1.7e-7
// is a RELATIVE 1e-6 at a scale of 0.16, which the inverse warp of the decoder amplifies (pixel coordinate = 9.5 ((x - t) / s + 1))
// into 6e-5 pixels at the glimpse edge and the likelihood sums over ~400 pixels into 0.04 - 0.4 nats per frame: found as a
// uniform few-percent deviation of whole gradients from the oracle in states with tiny scales (tests/nan_probe.py).
__device__ __forceinline__ float sq_sigmoid_geo(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float sq_tanh(float x) {  // 1 - 2 / (e^{2x} + 1): saturates cleanly (e^{2x} = inf -> 1, 0 -> -1)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(sq_exp(2.0f * x) + 1.0f);
}
__device__ __forceinline__ float sq_log1p(float y) {  // y in [0, 1]: log(u) y / (u - 1) with u = fl(1 + y) undoes the rounding of 1 + y
  const float u = 1.0f + y, d = u - 1.0f;
  return d == 0.0f ? y : sq_log(u) * (y * __builtin_amdgcn_rcpf(d));
}
__device__ __forceinline__ float sq_softplus(float x) { return fmaxf(x, 0.0f) + sq_log1p(sq_exp(-fabsf(x))); }
__device__ __forceinline__ float sq_elu(float x) { return x > 0.0f ? x : sq_exp(x) - 1.0f; }
__device__ __forceinline__ float sq_act(float v, int act) {
  switch (act) {
    case ACT_ELU: return sq_elu(v);
    case ACT_TANH: return sq_tanh(v);
    case ACT_SIGMOID: return sq_sigmoid(v);
    case ACT_SOFTPLUS_MIN: return sq_softplus(v) + 1e-2f;
    default: return v;
  }
}
// Sums of two / three products, association fixed with explicit fused multiply-adds: left to the compiler, `a x + b y` is
// contracted into fma(a, x, b y) or fma(b, y, a x) depending on the surrounding code, and two kernels that compute the same
// quantity (launch-per-op path | in-launch slot chain) would differ in the last bit.
__device__ __forceinline__ float sq_mix2(float a, float x, float b, float y) { return fmaf(b, y, a * x); }
__device__ __forceinline__ float sq_mix3(float a, float x, float b, float y, float c, float z) { return fmaf(c, z, fmaf(b, y, a * x)); }
// GRU state update h' = (1 - z) h + z hc
__device__ __forceinline__ float sq_gru_blend(float z, float h, float hc) { return sq_mix2(1.0f - z, h, z, hc); }
// Gates of the propagated what sample (sqair/core.py:336-359): g = 0.9999 sigmoid(.), and 1 - g with ONE rounding, written out
// as an explicit fused multiply-add: left to the compiler, `1.0f - sigmoid(x) * 0.9999f` is one fma where the product is not
// needed elsewhere and two roundings where it is (e.g. when the gate itself travels through a wave shuffle) -- the slot tail, its
// in-launch restatement and the layer epilogue that can take its place (k_linear_what) must agree to the last bit.
__device__ __forceinline__ float sq_gate(float x) { return sq_sigmoid(x) * 0.9999f; }
__device__ __forceinline__ float sq_gate_compl(float x) { return fmaf(-sq_sigmoid(x), 0.9999f, 1.0f); }
__device__ __forceinline__ float sq_normal_lp(float x, float loc, float scale) {
  const float d = (x - loc) / scale;
  return -0.5f * d * d - logf(scale) - 0.91893853320467274178f;
}
__device__ __forceinline__ float sq_bernoulli_lp(float x, float logit) {
  return -(fmaxf(logit, 0.0f) - logit * x + log1pf(expf(-fabsf(logit))));
}
// tfd.fill_triangular for n = 4: reshape(concat(v[4:], reverse(v)), [4,4]), lower band
__device__ __forceinline__ float tril4(const float* __restrict__ v, int i, int j) {
  const int q = i * 4 + j;
  return q < 6 ? v[4 + q] : v[15 - q];
}
// Pins the scalar loads of by-value kernel-argument fields HERE: hipcc otherwise fetches each field where it is first used --
// one s_load + wait at a time along the kernel's control flow, i.e. a chain of scalar round trips in front of the global loads
// whose addresses they feed.  Listing the fields of a kernel in two or three of these at its top makes them one batch.
#define SQ_PIN1(x) asm volatile("" ::"s"(x))
#define SQ_PIN4(a_, b_, c_, d_) asm volatile("" ::"s"(a_), "s"(b_), "s"(c_), "s"(d_))
#define SQ_PIN8(a_, b_, c_, d_, e_, f_, g_, h_) asm volatile("" ::"s"(a_), "s"(b_), "s"(c_), "s"(d_), "s"(e_), "s"(f_), "s"(g_), "s"(h_))
// n floats from global memory into LDS on the LDS-DMA path (global_load_lds_dword: no registers, nothing waits until the
// barrier ahead of the first reader, so every trip of every array staged this way is in flight together).  Wave `wave` of
// `n_waves` takes the 64-element trips wave, wave + n_waves, ...; lane l of a trip moves element base + l, the trip's LDS
// destination is wave-uniform + 4 l bytes; lanes past n are masked off.
__device__ __forceinline__ void sq_wave_stage(float* lds_dst, const float* __restrict__ src, int n, int lane, int wave = 0, int n_waves = 1) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  for (int base = 64 * wave; base < n; base += 64 * n_waves)
    if (base + lane < n) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + base + lane), (lds_ptr_t)(lds_dst + base), 4, 0, 0);
}
// The same in 16-byte units (global_load_lds_dwordx4: 1 KB per wave instruction): n4 units, src and lds_dst 16-byte aligned.
__device__ __forceinline__ void sq_wave_stage16(float* lds_dst, const float* __restrict__ src, int n4, int lane, int wave = 0, int n_waves = 1) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  for (int base = 64 * wave; base < n4; base += 64 * n_waves)
    if (base + lane < n4) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + 4 * (size_t)(base + lane)), (lds_ptr_t)(lds_dst + 4 * base), 16, 0, 0);
}
// Cross-lane sums on the VALU (every lane of the wave active; the result in every lane of the group).  __shfl_xor is a
// ds_bpermute_b32 -- an LDS-crossbar round trip of ~100 cycles, six of them in a row for one wave sum -- and the small
// per-row kernels of the chain are chains of exactly such sums.  Inside a row of 16 lanes: two quad permutes, the row's half
// mirror and mirror (DPP modifiers on the add's operand); across rows: gfx950's v_permlane16_swap / v_permlane32_swap, which
// with both operands the same value leave [r0 r0 r2 r2] | [r1 r1 r3 r3] resp. [lo lo] | [hi hi] (hipcc pads the
// VALU-write -> permlane hazard itself).  A different summation order than the xor butterfly's, just as fixed.
template <int CTRL>
__device__ __forceinline__ float sq_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sq_row_sum(float v) {   // over the 16 lanes of a DPP row
  v += sq_dpp<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += sq_dpp<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += sq_dpp<0x141>(v);   // row_half_mirror
  v += sq_dpp<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float sq_wave_max(float v) {   // (same network as sq_wave_sum; v of inactive slots: pass -3e38)
  v = fmaxf(v, sq_dpp<0xB1>(v));
  v = fmaxf(v, sq_dpp<0x4E>(v));
  v = fmaxf(v, sq_dpp<0x141>(v));
  v = fmaxf(v, sq_dpp<0x140>(v));
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
  const unsigned u2 = __builtin_bit_cast(unsigned, v);
  const auto r2 = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r2[0]), __builtin_bit_cast(float, (unsigned)r2[1]));
}
__device__ __forceinline__ float sq_read_lane(float v, int lane_uniform) {   // lane index wave-uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_uniform));
}
__device__ __forceinline__ float sq_half_sum(float v) {  // over lanes 0..31 / 32..63
  v = sq_row_sum(v);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float sq_wave_sum(float v) {
  v = sq_half_sum(v);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
#endif
