// Decoder canvas building blocks shared by k_insert_loglik (sqair_glue.hip) and its adjoint k_insert_loglik_bwd
// (sqair_bwd.hip).  reference: AIRDecoder._decode sqair/modules.py:435-467 (inverse spatial transformer of the N
// decoded glimpses onto the canvas + the written-to mask).
//
// A glimpse covers a small axis-aligned BOX of the canvas (the inverse warp is separable and monotone per axis), so the
// canvas is built slot by slot over each slot's box, 256 threads as a 32 x 8 patch, into an LDS band of at most
// 256 PF pixels (whole rows; with PF = 10 the 50 x 50 frame is one band, a 128 x 128 frame eight).  Until round 3
// every pixel walked all N slots: with consecutive pixels on a wavefront's lanes a wave ran a slot's bilinear body
// whenever ANY of its 64 pixels was inside that box -- 2.7x the useful work at 50 x 50, far more at 128 x 128.
#pragma once
#include "sqair_common.h"

// Two tunings (template parameters PF = frame / mean-image values a thread holds per band, so a band has at most 256 PF
// pixels, and ROWS = patch rows a thread has in flight in sq_canvas_band).  Forward: PF 5, ROWS 2 -- 65 VGPRs and 18 KB of LDS
// at 50 x 50 let all of the pass's 1600 workgroups be resident at once (25 -> 21 us; PF 10 / ROWS 4: 108 VGPRs, two rounds).
// Adjoint: PF 10, ROWS 4 -- it holds the glimpse gradient in LDS as well and is resident in two rounds either way, where
// fewer, larger bands win (64 us against 71; 243 against 306 at 128 x 128).
constexpr int SQ_CANVAS_PF_FWD = 5, SQ_CANVAS_ROWS_FWD = 2, SQ_CANVAS_PF_BWD = 10, SQ_CANVAS_ROWS_BWD = 4;
// Frames wider than 64 pixels (BASELINE configs[4]: 128 x 128): band heights that are whole patch trips.  The 32 x 8-thread patch
// walks a box in trips of 8 ROWS rows; with PF 5 a 128-wide band is 10 rows, of which a 16-row trip wastes six (ablation at cfg-5,
// timeline build with early exits: prologue 9 us, pixel phase 25 us, BAND BUILDING 74 of the kernel's 106 us).  PF 4 / ROWS 1:
// 8-row bands, one exact trip; the adjoint PF 8 / ROWS 2: 16-row bands.
constexpr int SQ_CANVAS_WIDE = 64, SQ_CANVAS_PF_FWD_W = 4, SQ_CANVAS_ROWS_FWD_W = 1, SQ_CANVAS_PF_BWD_W = 8, SQ_CANVAS_ROWS_BWD_W = 2;

// rows per band for a W-wide frame: as many whole rows as fit 256 PF pixels (host + device agree through the argument)
static inline int sq_canvas_band_rows(int H, int W, int pf) {
  int rows = 256 * pf / W;
  if (rows < 1) rows = 1;
  return rows < H ? rows : H;
}

struct CanvasLds {
  float* gl;    // [N][G2]  glimpses
  float* xt;    // [N][W]   glimpse x coordinate of every canvas column
  float* yt;    // [N][H]   glimpse y coordinate of every canvas row
  float* pres;  // [N]
  float* co;    // [N][4]   sx, sy, tx, ty
  int* box;     // [N][4]   first / last column, first / last row with a coordinate in (-1, G); empty: last < first
  float* cv;    // [band]   canvas (adjoint kernel: then its gradient)
  float* ms;    // [band]   written-to mask sum (adjoint kernel: then its gradient)
  float* end;
};
__host__ __device__ static inline size_t sq_canvas_lds_floats(int N, int G, int H, int W, int band_rows) {
  return (size_t)N * G * G + (size_t)N * (W + H) + N + 4 * N + 4 * N + 2 * (size_t)band_rows * W;
}
__device__ __forceinline__ CanvasLds sq_canvas_carve(float* smem, int N, int G, int H, int W, int band_rows) {
  CanvasLds c;
  c.gl = smem;
  c.xt = c.gl + N * G * G;
  c.yt = c.xt + N * W;
  c.pres = c.yt + N * H;
  c.co = c.pres + N;
  c.box = reinterpret_cast<int*>(c.co + 4 * N);
  c.cv = reinterpret_cast<float*>(c.box + 4 * N);
  c.ms = c.cv + band_rows * W;
  c.end = c.ms + band_rows * W;
  return c;
}

__device__ __forceinline__ bool sq_canvas_inside(float g, int G) { return g > -1.0f && g < (float)G; }

// glimpses, presences, transform coefficients, the two coordinate tables and the boxes of one row; ends on a barrier.
// where0 / pres0 point at slot 0's four where logits / presence, *_ld floats between slots.
__device__ __forceinline__ void sq_canvas_prologue(const CanvasLds& c, const float* __restrict__ glimpse, const float* __restrict__ where0,
                                                   int where_ld, const float* __restrict__ pres0, int pres_ld, int N, int G, int H, int W) {
  const int tid = threadIdx.x, G2 = G * G;
  sq_wave_stage(c.gl, glimpse, N * G2, tid & 63, tid >> 6, 4);   // (LDS-DMA: lands by the first barrier below)
  if (tid < N * 4) {
    const int k = tid >> 2, q = tid & 3;
    const float l = where0[(size_t)k * where_ld + q];
    c.co[tid] = (q & 2) ? tanhf(l) : fmaxf(sq_sigmoid_geo(l), 1e-4f);
    c.box[tid] = (q & 1) ? -1 : 0;
  }
  if (tid < N) c.pres[tid] = pres0[(size_t)tid * pres_ld];
  __syncthreads();
  for (int i = tid; i < N * (W + H); i += 256) {
    const int k = i / (W + H), q = i - k * (W + H);
    const bool is_y = q >= W;
    const int j = is_y ? q - W : q;
    const float sc = c.co[k * 4 + (is_y ? 1 : 0)], tr = c.co[k * 4 + (is_y ? 3 : 2)];
    const float L = (float)((is_y ? H : W) - 1);
    const float cn = -1.0f + 2.0f * (float)j / L;
    const float g = 0.5f * (float)(G - 1) * ((cn - tr) / sc + 1.0f);
    if (is_y) c.yt[k * H + j] = g; else c.xt[k * W + j] = g;
  }
  __syncthreads();
  // the coordinate grows with the pixel index (scale > 0), so the pixels inside are one run: its two ends have one writer each
  for (int i = tid; i < N * (W + H); i += 256) {
    const int k = i / (W + H), q = i - k * (W + H);
    const bool is_y = q >= W;
    const int j = is_y ? q - W : q, L = is_y ? H : W;
    const float* t = is_y ? c.yt + k * H : c.xt + k * W;
    if (sq_canvas_inside(t[j], G)) {
      if (j == 0 || !sq_canvas_inside(t[j - 1], G)) c.box[k * 4 + (is_y ? 2 : 0)] = j;
      if (j == L - 1 || !sq_canvas_inside(t[j + 1], G)) c.box[k * 4 + (is_y ? 3 : 1)] = j;
    }
  }
  __syncthreads();
}

// bilinear value of glimpse gk and of a glimpse of ones at (xg, yg), both coordinates inside (-1, G): taps outside the glimpse
// are zero.  Branch-free (clamped addresses, zeroed weights): four of these are in flight per thread in sq_canvas_band.
__device__ __forceinline__ void sq_canvas_tap(const float* __restrict__ gk, float xg, float yg, int G, float& v, float& on) {
  const float x0f = floorf(xg), y0f = floorf(yg);
  const int x0 = (int)x0f, y0 = (int)y0f;
  const float fx = xg - x0f, fy = yg - y0f;
  const float wx0 = x0 >= 0 ? 1.0f - fx : 0.0f, wx1 = x0 + 1 < G ? fx : 0.0f;
  const float wy0 = y0 >= 0 ? 1.0f - fy : 0.0f, wy1 = y0 + 1 < G ? fy : 0.0f;
  const int xa = max(x0, 0), xb = min(x0 + 1, G - 1);
  const float* ra = gk + max(y0, 0) * G;
  const float* rb = gk + min(y0 + 1, G - 1) * G;
  v = wy0 * (wx0 * ra[xa] + wx1 * ra[xb]) + wy1 * (wx0 * rb[xa] + wx1 * rb[xb]);
  on = (wy0 + wy1) * (wx0 + wx1);
}

// wave-uniform copies of slot k's presence and of its box clipped to rows [yb0, yb1]; false: nothing of the slot in the band
struct CanvasSlot { float pk; int x0, x1, y0, y1; };
__device__ __forceinline__ bool sq_canvas_slot(const CanvasLds& c, int k, int yb0, int yb1, CanvasSlot& s) {
  s.pk = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c.pres[k])));
  s.x0 = __builtin_amdgcn_readfirstlane(c.box[k * 4 + 0]);
  s.x1 = __builtin_amdgcn_readfirstlane(c.box[k * 4 + 1]);
  s.y0 = max(__builtin_amdgcn_readfirstlane(c.box[k * 4 + 2]), yb0);
  s.y1 = min(__builtin_amdgcn_readfirstlane(c.box[k * 4 + 3]), yb1);
  return s.pk != 0.0f && s.x1 >= s.x0 && s.y1 >= s.y0;
}

// canvas and mask sum of rows [yb0, yb1] into c.cv / c.ms (slots added in index order, as the reference's sum over objects);
// ends on a barrier
template <int ROWS>
__device__ __forceinline__ void sq_canvas_band(const CanvasLds& c, int yb0, int yb1, int N, int G, int H, int W) {
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int n = (yb1 - yb0 + 1) * W;
  for (int i = tid; i < n; i += 256) {
    c.cv[i] = 0.0f;
    c.ms[i] = 0.0f;
  }
  __syncthreads();
  for (int k = 0; k < N; ++k) {
    CanvasSlot s;
    if (!sq_canvas_slot(c, k, yb0, yb1, s)) continue;
    const float* gk = c.gl + k * G * G;
    // ROWS rows of the patch per trip: all their table / glimpse reads first, the read-modify-writes of the band after
    // them (one pixel at a time the reads of a pixel queued behind the writes of the one before: ~6 dependent LDS round trips
    // per pixel)
    for (int Y0 = s.y0 + ty; Y0 <= s.y1; Y0 += 8 * ROWS)
      for (int X = s.x0 + tx; X <= s.x1; X += 32) {
        const float xg = c.xt[k * W + X];
        float v[ROWS], on[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) sq_canvas_tap(gk, xg, c.yt[k * H + min(Y0 + 8 * u, s.y1)], G, v[u], on[u]);
#pragma unroll
        for (int u = 0; u < ROWS; ++u)
          if (Y0 + 8 * u <= s.y1) {
            const int o = (Y0 + 8 * u - yb0) * W + X;
            c.cv[o] += v[u] * s.pk;
            c.ms[o] += on[u] * s.pk;
          }
      }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Round 4, frames wider than a wavefront (BASELINE configs[4], 128 x 128): the ROW-WAVE formulation.  The band builder above
// spends its time in LDS read-modify-writes of the band, a barrier per (band, slot) and ~60 VALU instructions per tap, all of
// which re-derive quantities that depend on the pixel's COLUMN only (x tap, x weights) or on its ROW only (y tap, y weights) --
// and the kernel is bound by VALU issue (a wave64 instruction occupies its SIMD16 for four cycles): 77 lane-operations per pixel
// at 52 us.  Here a wavefront owns whole canvas rows, lane l the CPL adjacent columns CPL l .. CPL l + CPL - 1, so
//   * the column half of every slot's tap -- byte offset of the left texel pair, its two weights -- is computed ONCE per workgroup
//     and stays in registers (3 per slot and column),
//   * the row half -- byte offset of the upper glimpse row, presence-scaled weights -- is a 16-byte record per (slot, row) in LDS
//     which all lanes of the wave read at one address (a broadcast), and "row inside the slot's box" is a SCALAR comparison
//     against row bounds held in SGPRs,
//   * the glimpses lie in LDS as vertical PAIRS {g[y][x], g[y + 1][x]}: one ds_read2_b64 returns the four texels of a tap as two
//     register pairs, which packed fp32 instructions (v_pk_mul / v_pk_fma: two lanes' worth per issue) combine,
//   * canvas and mask sum accumulate in registers: no band, no read-modify-write, no barrier after the prologue.
// Both taps of an axis are taken at i, i + 1 with i clamped to [0, G - 2]; a tap that falls outside the glimpse gets weight
// zero on the texel that stands in for it (the reference's resampler adds nothing for it either).
// ------------------------------------------------------------------------------------------------
typedef float sq_f2 __attribute__((ext_vector_type(2)));
typedef float sq_f4 __attribute__((ext_vector_type(4)));
typedef float sq_f4a8 __attribute__((ext_vector_type(4), aligned(8)));   // four floats at an 8-byte boundary (ds_read2_b64)
struct CanvasAxisTap { int i; float wa, wb; };   // texels i, i + 1 (0 <= i <= G - 2) with weights wa, wb; both 0 outside (-1, G)
__device__ __forceinline__ CanvasAxisTap sq_canvas_axis_tap(float g, int G) {
  CanvasAxisTap t;
  const float f0 = floorf(g);
  const int i0 = (int)f0;
  const float fr = g - f0;
  const bool in = sq_canvas_inside(g, G);
  // i0 == -1: only the right tap (texel 0) exists; i0 == G - 1: only the left tap (texel G - 1)
  t.i = min(max(i0, 0), G - 2);
  const float wl = 1.0f - fr, wr = fr;
  t.wa = !in ? 0.0f : (i0 < 0 ? wr : (i0 > G - 2 ? 0.0f : wl));
  t.wb = !in ? 0.0f : (i0 < 0 ? 0.0f : (i0 > G - 2 ? wl : wr));
  return t;
}
// d wa / d g, d wb / d g of the same tap (the adjoint kernel): -1, +1 inside; +1, 0 at i0 == -1; 0, -1 at i0 == G - 1; 0, 0 outside
__device__ __forceinline__ void sq_canvas_axis_tap_d(float g, int G, float& da, float& db) {
  const int i0 = (int)floorf(g);
  const bool in = sq_canvas_inside(g, G);
  da = !in ? 0.0f : (i0 < 0 ? 1.0f : (i0 > G - 2 ? 0.0f : -1.0f));
  db = !in ? 0.0f : (i0 < 0 ? 0.0f : (i0 > G - 2 ? -1.0f : 1.0f));
}
__device__ __forceinline__ float sq_canvas_coord(int j, int L, float sc, float tr, int G) {   // as sq_canvas_prologue's tables
  const float cn = -1.0f + 2.0f * (float)j / (float)(L - 1);
  return 0.5f * (float)(G - 1) * ((cn - tr) / sc + 1.0f);
}
struct CanvasRowsLds {
  sq_f2* pr;       // [N][G - 1][G]  {g[y][x], g[y + 1][x]}
  float4* yrec;    // [N][H]  {byte offset of pair row i of slot k inside pr, pk wa, pk wb, pk (wa + wb)}; weights 0 outside the box
  float4* yrec2;   // [N][H]  adjoint kernel only: {pk d wa / d g, pk d wb / d g, normalised row coordinate Yn, glimpse row i (int)}
  unsigned* rmask; // [H]  bit k: row inside slot k's box (slot present, weights not both 0)
  float* co;       // [N][4]
  float* pres;     // [N]
  float* end;
};
__host__ __device__ static inline size_t sq_canvas_rows_lds_floats(int N, int G, int H, bool bwd = false) {
  return (size_t)((2 * N * (G - 1) * G + 3) & ~3) + (bwd ? 8 : 4) * (size_t)N * H + H + 4 * N + ((N + 3) & ~3);
}
__device__ __forceinline__ CanvasRowsLds sq_canvas_rows_carve(float* smem, int N, int G, int H, bool bwd = false) {
  CanvasRowsLds c;
  c.pr = reinterpret_cast<sq_f2*>(smem);
  c.yrec = reinterpret_cast<float4*>(smem + ((2 * N * (G - 1) * G + 3) & ~3));
  c.yrec2 = c.yrec + N * H;
  c.rmask = reinterpret_cast<unsigned*>(c.yrec2 + (bwd ? N * H : 0));
  c.co = reinterpret_cast<float*>(c.rmask + H);
  c.pres = c.co + 4 * N;
  c.end = c.pres + ((N + 3) & ~3);
  return c;
}
// glimpse pairs, coefficients, presences, the row records and row masks of one (row, frame); ends on a barrier.  NT threads; GL =
// glimpse values a thread fetches (N G^2 <= NT GL); g_mul = sq_magic(G)
template <int NT, int GL, bool BWD = false>
__device__ __forceinline__ void sq_canvas_rows_prologue(const CanvasRowsLds& c, const float* __restrict__ glimpse, const float* __restrict__ where0,
                                                        int where_ld, const float* __restrict__ pres0, int pres_ld, int N, int G, int H, SqMagic g_mul) {
  const int tid = threadIdx.x, G2 = G * G, n = N * G2;
  float gv[GL];
#pragma unroll
  for (int u = 0; u < GL; ++u) gv[u] = glimpse[min(tid + NT * u, n - 1)];   // (clamped, unconditional: all in flight)
  if (tid < N * 4) {
    const int k = tid >> 2, q = tid & 3;
    const float l = where0[(size_t)k * where_ld + q];
    c.co[tid] = (q & 2) ? tanhf(l) : fmaxf(sq_sigmoid_geo(l), 1e-4f);
  }
  if (tid < N) c.pres[tid] = pres0[(size_t)tid * pres_ld];
  for (int i = tid; i < H; i += NT) c.rmask[i] = 0u;
  __syncthreads();
  for (int k = 0; k < N; ++k) {
    const float pk = c.pres[k], sy = c.co[k * 4 + 1], ty = c.co[k * 4 + 3];
    for (int y = tid; y < H; y += NT) {
      const CanvasAxisTap t = sq_canvas_axis_tap(sq_canvas_coord(y, H, sy, ty, G), G);
      const float sm = pk * (t.wa + t.wb);
      c.yrec[k * H + y] = make_float4(__builtin_bit_cast(float, (k * (G - 1) + t.i) * G * 8), pk * t.wa, pk * t.wb, sm);
      if (BWD) {
        float da, db;
        const float g = sq_canvas_coord(y, H, sy, ty, G);
        sq_canvas_axis_tap_d(g, G, da, db);
        c.yrec2[k * H + y] = make_float4(pk * da, pk * db, -1.0f + 2.0f * (float)y / (float)(H - 1), __builtin_bit_cast(float, t.i));
      }
      if (sm != 0.0f) atomicOr(&c.rmask[y], 1u << k);
    }
  }
  float* prf = reinterpret_cast<float*>(c.pr);
#pragma unroll
  for (int u = 0; u < GL; ++u) {
    const int i = tid + NT * u;
    if (i < n) {
      const int row = sq_div(i, g_mul), x = i - row * G, k = sq_div(row, g_mul), y = row - k * G;   // (no integer division)
      if (y < G - 1) prf[((k * (G - 1) + y) * G + x) * 2] = gv[u];          // upper texel of pair row y
      if (y > 0) prf[((k * (G - 1) + y - 1) * G + x) * 2 + 1] = gv[u];      // lower texel of pair row y - 1
    }
  }
  __syncthreads();
}
#ifndef SQ_ROWS_WPE
// waves per SIMD the register allocation must leave room for.  8 slots x 4 columns per lane keep 96 VGPRs of column taps alone: at
// 4 waves per SIMD (128 VGPRs) that instantiation spilled 27-32 registers to scratch; with room for 2 it takes 158-166 VGPRs, no
// scratch, and is faster (back to back, 1600 rows: 30 x 250 x 8 slots 74 -> 53 us, 100 x 200 x 6 slots 151 -> 96 us)
#define SQ_ROWS_WPE(NMAX, CPL) ((NMAX) * (CPL) <= 8 ? 7 : ((NMAX) * (CPL) >= 32 ? 2 : 4))
#endif
__device__ __forceinline__ sq_f2 sq_fma2(sq_f2 a, sq_f2 b, sq_f2 c) { return __builtin_elementwise_fma(a, b, c); }
// sigmoid(-10 + 20 ms) for two pixels: sq_exp without its clamp and NaN select (the argument 10 - 20 ms lies in [-20 N + 10, 10]: no
// overflow, 2^t underflows to 0 cleanly, a NaN stays a NaN), the same two-part product
__device__ __forceinline__ sq_f2 sq_mask_sigmoid2(sq_f2 ms) {
  const sq_f2 x = sq_fma2(ms, sq_f2{-20.0f, -20.0f}, sq_f2{10.0f, 10.0f});
  const sq_f2 L2E = {1.44269504088896340736f, 1.44269504088896340736f};
  const sq_f2 t = x * L2E;
  const sq_f2 r = sq_fma2(x, L2E, -t) + x * sq_f2{1.92596299112661746e-8f, 1.92596299112661746e-8f};
  const sq_f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  const sq_f2 ee = sq_fma2(e, r * sq_f2{0.69314718055994530942f, 0.69314718055994530942f}, e) + sq_f2{1.0f, 1.0f};
  return sq_f2{__builtin_amdgcn_rcpf(ee.x), __builtin_amdgcn_rcpf(ee.y)};
}
// {wa a0 + wb a1, wa b0 + wb b1} from a tap {a0, b0, a1, b1} and the weight pair {wa, wb} as it lies in its registers: the packed
// instructions pick its low / high half for BOTH lanes (op_sel), where the compiler would keep {wa, wa} and {wb, wb} as two more
// register pairs per slot and column (16 VGPRs more: a wave fewer per SIMD).  (s_nop: the hazard recogniser does not look inside.)
__device__ __forceinline__ sq_f2 sq_tap_rows(sq_f2 w, sq_f2 txy, sq_f2 tzw) {
  sq_f2 t;
  asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\ts_nop 0\n\tv_pk_fma_f32 %0, %1, %3, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\ts_nop 0"
      : "=&v"(t) : "v"(w), "v"(txy), "v"(tzw));
  return t;
}
// fire-and-forget float add on an LDS word (ds_add_f32, no return value)
__device__ __forceinline__ void sq_lds_add(float* p, float v) {
  typedef __attribute__((address_space(3))) float* lds_f;
  __builtin_amdgcn_ds_faddf((lds_f)p, v, 0, 0, false);
}
