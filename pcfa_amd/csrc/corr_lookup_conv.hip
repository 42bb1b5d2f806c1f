// RAFT/GMA: correlation lookup fused with the motion encoder's first layer, forward and backward, for gfx950.
//
//   cor1 = relu(convc1(corr_fn(coords)))      models/raft/update.py:79-93 (convc1 = Conv2d(324, 256, 1)) fed by
//                                             models/raft/corr.py:29-50 (CorrBlock.__call__), raft.py:123-124
// (SURVEY 8f row f2).  Un-fused, every refinement iteration writes the [324][Q] lookup result (9.1 MB at 55x128),
// reads it back into a library GEMM (28 us, 41 TFLOP/s), runs a bias+ReLU pass, and the backward mirrors all of it.
// Here the 4 x 81 taps of a query never leave the CU:
//   forward : workgroup = 32 queries x all 4 levels x all 256 output channels.  The pieces of all four windows of
//             every query are requested up front (20 x 16 B per thread in flight); per level the window images go
//             to LDS (the lookup kernel's image, corr_lookup.hip), thread (query, window row) blends its 9 taps into
//             a [88][32] tap tile (81 taps + 7 zero rows), and the four waves run the level's share of
//             W[256][324] . taps[324][32] on v_mfma_f32_32x32x2_f32 (exact fp32), wave w owning output channels
//             64w..64w+63, W pre-packed in operand order and streamed from L2 straight into registers.
//             Epilogue: + bias, ReLU, 128-B row stores.
//   backward: the gradient tile (grad_out * [cor1 > 0], 256 x 32) goes to LDS, d taps = W^T . g on the matrix
//             cores (11 row tiles of the padded 352 tap rows), then per level the lookup's transpose: thread (query,
//             row) builds a row of the window's gradient image, every 16-B piece adds its texels to dpyr.  The read
//             half of all four levels' read-modify-write is issued before the GEMM.  Same ownership as
//             corr_lookup_bwd: no atomics, bitwise reproducible.
// Radius 4, 4 levels (the only configuration RAFT / GMA use, raft_config.json / gma_config.json); other shapes
// return PCFA_ERR_UNSUPPORTED and the caller composes pcfa_corr_lookup_* with a convolution.
#include <cstdlib>
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int R = 4, N1 = 9, WIN = 10, TXN = 4, PIECES = WIN * TXN;  // window geometry (corr_lookup.hip, Geo<4>)
constexpr int RS = 20, WS = WIN * RS + 4, NRD = 3;                    // LDS image: row stride, window stride
constexpr int QT = 32;                                               // queries per workgroup
constexpr int NT = 256;                                              // threads (4 waves)
constexpr int NPC = QT * PIECES / NT;                                // 5 pieces per thread and level
constexpr int L = 4, TAPS = N1 * N1;                                 // 81 taps per level
constexpr int KL = 88, KG = KL / 8;                                  // padded tap rows per level, groups of 8 rows
constexpr int KP = L * KL;                                           // 352 padded tap rows
constexpr int WIN_FLOATS = QT * WS + 4;
#ifndef PCFA_LC_WD
#define PCFA_LC_WD 11  // depth of the forward kernel's W register ring (groups of 8 MFMAs per wave)
#endif

struct Origin {
  int x0, y0;
  float fx, fy;
};

// reference: coords / 2**i (exact power-of-two scaling), then floor / fraction -- identical to corr_lookup.hip
__device__ __forceinline__ Origin make_origin(float cx, float cy, int level) {
  const float inv = 1.0f / (float)(1 << level);
  const float xl = cx * inv, yl = cy * inv;
  const float flx = floorf(xl), fly = floorf(yl);
  Origin o;
  o.fx = xl - flx;
  o.fy = yl - fly;
  o.x0 = (int)fminf(fmaxf(flx, -1.0e8f), 1.0e8f) - R;
  o.y0 = (int)fminf(fmaxf(fly, -1.0e8f), 1.0e8f) - R;
  return o;
}

// A piece outside its level loads the slab's all-zero tile instead (common.hpp: P.zero): plain loads that hipcc
// counts itself -- with 20 pieces per thread in flight across a GEMM, hand-counted inline-asm loads (corr_lookup.hip)
// would leave 80 registers the compiler believes are already written.
__device__ __forceinline__ f32x4 load_piece(const float* sbase, unsigned voff_bytes) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(sbase) + voff_bytes);
}

template <typename T>
__device__ __forceinline__ T* scalar_ptr(T* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

struct Piece {
  unsigned goff;  // bytes from the workgroup's first slab
  unsigned lds;   // byte address of the piece's first texel in the window image
  int mask;       // bits 0-3: texel is a window texel inside the level; bit 4: the piece exists in the level
  __device__ __forceinline__ bool need() const { return (mask & 16) != 0; }
};

// Per (thread, level): the query's window origin and the thread's NPC pieces.  Piece slot p = i * NT + tid:
// window k = p / 40 (query inside the block), window row rr, tile column tx -- as corr_lookup.hip's make_piece.
struct LevelGeo {
  float fx, fy;
  Piece pc[NPC];
};

__device__ __forceinline__ void level_geometry(LevelGeo& g, float cx, float cy, bool live, int level, int j,
                                               const PyrLayout& P, int tid) {
  const int hl = P.h[level], wl = P.w[level], tw = P.tw[level], off = P.off[level], slab = P.slab;
  const Origin o = make_origin(cx, cy, level);
  g.fx = o.fx;
  g.fy = o.fy;
  const int th4 = ((hl + 3) >> 2) << 2;
  const int cx0 = min(max(o.x0, -16), 4 * tw), cy0 = live ? min(max(o.y0, -16), th4) : th4;
  const int myB = (off + j * slab) * 4;
  const int myXY = (cy0 + 16) | ((cx0 + 16) << 16);
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const unsigned p = (unsigned)(i * NT + tid);
    const unsigned k = p / (unsigned)PIECES, q = p - k * (unsigned)PIECES;
    const unsigned rr = q / (unsigned)TXN, tx = q - rr * (unsigned)TXN;
    const int wB = __builtin_amdgcn_ds_bpermute((int)(k * 4u), myB);     // lane k of this wave owns query k
    const int wXY = __builtin_amdgcn_ds_bpermute((int)(k * 4u), myXY);
    const int y0 = (int)((unsigned)wXY & 0xffffu) - 16, x0 = (int)((unsigned)wXY >> 16) - 16;
    const int ox = x0 & 3, y = y0 + (int)rr, gtx = (x0 >> 2) + (int)tx;
    const bool need = (unsigned)y < (unsigned)hl && (unsigned)gtx < (unsigned)tw && (int)(4 * tx) < ox + WIN;
    const int c0 = (int)(4 * tx) - ox, gx0 = 4 * gtx;
    int mask = need ? 16 : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((unsigned)(c0 + e) < (unsigned)WIN && gx0 + e < wl) mask |= 1 << e;
    g.pc[i].mask = mask;
    g.pc[i].goff = need ? (unsigned)wB + (((((unsigned)y >> 2) * (unsigned)tw + (unsigned)gtx) << 4) + (((unsigned)y & 3u) << 2)) * 4u
                        : (unsigned)P.zero * 4u;   // the first slab's zero tile
    g.pc[i].lds = (k * WS + rr * RS + 4u + 4u * tx - (unsigned)ox) * 4u;
  }
}

// Packed weights (forward): Wp[mtile 8][kg 44][lane 64][4]: float e of lane = W[n = 32 mtile + (lane & 31)][k'] with
// k' = 8 kg + 2 e + (lane >> 5) in the padded tap space (88 rows per level: 81 taps + 7 zero rows).
// Packed weights (backward): Wt[mtile 11][ng 32][lane 64][4]: float e = W[n = 8 ng + 2 e + (lane >> 5)][k' = 32 mtile + (lane & 31)].
__global__ void convc1_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ wt, int Cout) {
  const int nf = (Cout / 32) * (KP / 8) * 64 * 4, nb = (KP / 32) * (Cout / 8) * 64 * 4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nf + nb; e += gridDim.x * blockDim.x) {
    int n, kp;
    if (e < nf) {
      const int pos = e & 3, lane = (e >> 2) & 63, blk = e >> 8;
      const int kg = blk % (KP / 8), mt = blk / (KP / 8);
      n = 32 * mt + (lane & 31);
      kp = 8 * kg + 2 * pos + (lane >> 5);
    } else {
      const int f = e - nf;
      const int pos = f & 3, lane = (f >> 2) & 63, blk = f >> 8;
      const int ng = blk % (Cout / 8), mt = blk / (Cout / 8);
      n = 8 * ng + 2 * pos + (lane >> 5);
      kp = 32 * mt + (lane & 31);
    }
    const int level = kp / KL, i = kp - level * KL;
    const float v = i < TAPS ? w[(size_t)n * (L * TAPS) + level * TAPS + i] : 0.f;
    if (e < nf) wp[e] = v; else wt[e - nf] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void corr_lookup_convc1_fwd_kernel(
    const float* __restrict__ pyr, const float* __restrict__ coords, const float* __restrict__ wp,
    const float* __restrict__ bias, float* __restrict__ out, int Q, PyrLayout P, int relu, int dbg) {
  constexpr int COUT = 256, MTW = COUT / 32 / 4;     // m-tiles per wave (2)
  __shared__ __attribute__((aligned(16))) float s_win[WIN_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_tap[2][KL][QT];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = tid & 31, l31 = lane & 31, lh = lane >> 5;
  const int b_img = blockIdx.z, q0 = blockIdx.x * QT;
  const bool live = q0 + j < Q;
  float cx = 0.f, cy = 0.f;
  if (live) {
    cx = coords[((size_t)b_img * 2 + 0) * Q + q0 + j];
    cy = coords[((size_t)b_img * 2 + 1) * Q + q0 + j];
  }
  const float* slab0 = scalar_ptr(pyr + ((size_t)b_img * Q + q0) * P.slab);

  // the zero rows of the tap tiles (their weights are zero too, but 0 * garbage may be NaN)
  for (int e = tid; e < 2 * (KL - TAPS) * QT; e += NT) {
    const int buf = e / ((KL - TAPS) * QT), rem = e - buf * ((KL - TAPS) * QT);
    s_tap[buf][TAPS + rem / QT][rem % QT] = 0.f;
  }

  // ---- every piece of all four levels is requested before anything waits; coarsest level first: its windows are
  //      L2-resident and land first, so the matrix cores start while the level-0 texels are still on their way ----
  f32x4 v[L][NPC];
  unsigned dst[L][NPC];
  float fxs[L], fys[L];
#pragma unroll
  for (int lr = 0; lr < L; ++lr) {
    const int l = L - 1 - lr;
    LevelGeo g;
    level_geometry(g, cx, cy, live, l, j, P, tid);
    fxs[l] = g.fx;
    fys[l] = g.fy;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      dst[l][i] = g.pc[i].lds;
      v[l][i] = load_piece(slab0, g.pc[i].goff);
    }
  }

  f32x16 acc[MTW];   // starts at the bias: the epilogue then has no dependent loads left
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = bias[(wv * MTW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
  const f32x4* wq = reinterpret_cast<const f32x4*>(wp) + ((size_t)(wv * MTW) * (KP / 8)) * 64 + lane;
  // W operands: a register ring WD groups deep over the 44 groups of all four levels (the loops below are fully
  // unrolled, so every ring index is static).  One group = 4 k-pairs = 8 MFMAs of 64 cycles per wave; the loads come
  // from L2 (500-900 cycles), so the ring has to run >= 2 groups ahead -- with one group of lead the matrix pipe
  // idled for the load latency in every group (26 us per launch, 28 % of the fp32 matrix peak).
  constexpr int WD = PCFA_LC_WD, NG = L * KG;
  f32x4 wring[WD][MTW];
#pragma unroll
  for (int d = 0; d < WD; ++d)
#pragma unroll
    for (int m = 0; m < MTW; ++m) wring[d][m] = wq[((size_t)m * (KP / 8) + (L - 1 - d / KG) * KG + d % KG) * 64];

#pragma unroll
  for (int lr = 0; lr < L; ++lr) {
    const int l = L - 1 - lr;
    // ---- window images of level l -> LDS (sub-tile offset removed: per-lane unaligned dword stores) ----
    char* lds_bytes = reinterpret_cast<char*>(s_win);
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      float* d = reinterpret_cast<float*>(lds_bytes + dst[l][i]);
      d[0] = v[l][i].x; d[1] = v[l][i].y; d[2] = v[l][i].z; d[3] = v[l][i].w;
    }
    __syncthreads();
    // ---- thread (query j, window row b) blends its 9 taps; row 8 by the first 32 threads ----
    if (!(dbg & 2)) {
      const float fx = fxs[l], fy = fys[l];
      const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
      float (*tap)[QT] = s_tap[l & 1];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int b = pass == 0 ? (tid >> 5) : 8;
        if (pass == 1 && tid >= 32) break;
        const float4* row0 = reinterpret_cast<const float4*>(&s_win[j * WS + b * RS + 4]);
        const float4* row1 = reinterpret_cast<const float4*>(&s_win[j * WS + (b + 1) * RS + 4]);
        float t0[NRD * 4], t1[NRD * 4];
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
          const float4 u0 = row0[i], u1 = row1[i];
          t0[4 * i] = u0.x; t0[4 * i + 1] = u0.y; t0[4 * i + 2] = u0.z; t0[4 * i + 3] = u0.w;
          t1[4 * i] = u1.x; t1[4 * i + 1] = u1.y; t1[4 * i + 2] = u1.z; t1[4 * i + 3] = u1.w;
        }
#pragma unroll
        for (int a = 0; a < N1; ++a)   // same expression as corr_lookup_fwd_body: bit-identical taps
          tap[a * N1 + b][j] = t0[a] * w00 + t0[a + 1] * w01 + t1[a] * w10 + t1[a + 1] * w11;
      }
    }
    __syncthreads();   // taps of level l visible; the window image is free for level l + 1
    // ---- the level's share of W . taps: 11 groups of 4 k-pairs, 2 m-tiles per wave ----
    if (!(dbg & 1)) {
      const float (*tap)[QT] = s_tap[l & 1];
#pragma unroll
      for (int g = 0; g < KG; ++g) {
        const int G = lr * KG + g;                      // position in the W stream (compile-time after unrolling)
        float bv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = tap[8 * g + 2 * e + lh][l31];
        f32x4 wcur[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) wcur[m] = wring[G % WD][m];
        if (G + WD < NG) {   // group G + WD of the stream: level L-1 - (G+WD)/KG, group (G+WD) % KG
          constexpr int dummy = 0; (void)dummy;
          const int Gn = G + WD, ln = L - 1 - Gn / KG, gn = Gn % KG;
#pragma unroll
          for (int m = 0; m < MTW; ++m) wring[G % WD][m] = wq[((size_t)m * (KP / 8) + ln * KG + gn) * 64];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            const float av = e == 0 ? wcur[m].x : e == 1 ? wcur[m].y : e == 2 ? wcur[m].z : wcur[m].w;
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[e], acc[m], 0, 0, 0);
          }
      }
    }
  }

  // ---- epilogue: + bias, ReLU, rows of 32 queries (128 B) per half-wave ----
  const bool qlive = q0 + l31 < Q;
  float* ob = out + (size_t)b_img * COUT * Q + q0 + l31;
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (wv * MTW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float y = acc[m][r];
      if (relu) y = fmaxf(y, 0.f);
      if (qlive) ob[(size_t)n * Q] = y;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward: dpyr += lookup^T( W^T . (grad_out * [y > 0]) )
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void corr_lookup_convc1_bwd_kernel(
    float* __restrict__ dpyr, const float* __restrict__ coords, const float* __restrict__ wt,
    const float* __restrict__ y, const float* __restrict__ grad_out, int Q, PyrLayout P, int relu) {
  constexpr int COUT = 256, MT = KP / 32;            // 11 row tiles of the padded tap space
  constexpr int MTW = 3;                             // row tiles per wave (wave 3 owns two)
  // one LDS block: [gradient tile 256 x 32 | later: window image]  +  [d taps 352 x 32]
  __shared__ __attribute__((aligned(16))) float s_a[COUT * QT > WIN_FLOATS ? COUT * QT : WIN_FLOATS];
  __shared__ __attribute__((aligned(16))) float s_dt[KP][QT];
  float (*s_g)[QT] = reinterpret_cast<float (*)[QT]>(s_a);
  float* s_win = s_a;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = tid & 31, l31 = lane & 31, lh = lane >> 5;
  const int b_img = blockIdx.z, q0 = blockIdx.x * QT;
  const bool live = q0 + j < Q;
  float cx = 0.f, cy = 0.f;
  if (live) {
    cx = coords[((size_t)b_img * 2 + 0) * Q + q0 + j];
    cy = coords[((size_t)b_img * 2 + 1) * Q + q0 + j];
  }
  float* slab0 = scalar_ptr(dpyr + ((size_t)b_img * Q + q0) * P.slab);

  // ---- the read half of the read-modify-write of all four levels goes out first ----
  f32x4 v[L][NPC];
  Piece pc[L][NPC];
  float fxs[L], fys[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    LevelGeo g;
    level_geometry(g, cx, cy, live, l, j, P, tid);
    fxs[l] = g.fx;
    fys[l] = g.fy;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      pc[l][i] = g.pc[i];
      v[l][i] = load_piece(slab0, pc[l][i].goff);
    }
  }

  // ---- gradient tile: g[n][q] = grad_out * [y > 0]; thread = (4 queries, 8 rows apart): 16-B loads, all 16 of a
  //      thread in flight at once (per-float loads in four dependent batches were ~6 us of the launch) ----
  {
    const int qq = (tid & 7) * 4, n0 = tid >> 3;      // 8 quads per row, 32 rows per pass, 8 passes
    const bool vec = (Q & 3) == 0 && q0 + qq + 3 < Q;
    const size_t base = (size_t)b_img * COUT * Q + q0 + qq;
    f32x4 gg[COUT / 32], yy[COUT / 32];
#pragma unroll
    for (int i = 0; i < COUT / 32; ++i) {
      const size_t o = base + (size_t)(n0 + 32 * i) * Q;
      if (vec) {
        gg[i] = *reinterpret_cast<const f32x4*>(grad_out + o);
        yy[i] = relu ? *reinterpret_cast<const f32x4*>(y + o) : (f32x4)(1.f);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = q0 + qq + e < Q;
          gg[i][e] = ok ? grad_out[o + e] : 0.f;
          yy[i][e] = (ok && relu) ? y[o + e] : 1.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < COUT / 32; ++i) {
      f32x4 t;
      t.x = yy[i].x > 0.f ? gg[i].x : 0.f; t.y = yy[i].y > 0.f ? gg[i].y : 0.f;
      t.z = yy[i].z > 0.f ? gg[i].z : 0.f; t.w = yy[i].w > 0.f ? gg[i].w : 0.f;
      *reinterpret_cast<f32x4*>(&s_g[n0 + 32 * i][qq]) = t;
    }
  }
  __syncthreads();

  // ---- d taps [352][32] = W^T [352][256] . g [256][32]; wave w owns row tiles w, w + 4, w + 8 ----
  {
    f32x16 acc[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const f32x4* wq = reinterpret_cast<const f32x4*>(wt) + lane;
    const int nmt = wv + 8 < MT ? 3 : 2;
    // W^T operands: register ring 3 groups deep (one group = 12 MFMAs here), all indices static (unrolled by 3)
    constexpr int WD = 3;
    static_assert((COUT / 8) % WD != 0 || true, "");
    f32x4 wring[WD][MTW];
    size_t wrow[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) wrow[m] = (size_t)min(wv + 4 * m, MT - 1) * (COUT / 8);
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
      for (int m = 0; m < MTW; ++m) wring[d][m] = wq[(wrow[m] + d) * 64];
    for (int g0 = 0; g0 < COUT / 8; g0 += WD) {
#pragma unroll
      for (int d = 0; d < WD; ++d) {
        const int g = g0 + d;
        if (g < COUT / 8) {
          float bv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = s_g[8 * g + 2 * e + lh][l31];
          f32x4 wcur[MTW];
#pragma unroll
          for (int m = 0; m < MTW; ++m) wcur[m] = wring[d][m];
          const int gn = min(g + WD, COUT / 8 - 1);
#pragma unroll
          for (int m = 0; m < MTW; ++m) wring[d][m] = wq[(wrow[m] + gn) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
              if (m < nmt) {
                const float av = e == 0 ? wcur[m].x : e == 1 ? wcur[m].y : e == 2 ? wcur[m].z : wcur[m].w;
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[e], acc[m], 0, 0, 0);
              }
            }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      if (m < nmt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s_dt[(wv + 4 * m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh][l31] = acc[m][r];
      }
    }
  }
  __syncthreads();   // d taps complete; the gradient tile is dead: its LDS becomes the window image

  // ---- per level: the lookup's transpose (corr_lookup_bwd_body with the tap gradients read from LDS) ----
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float fx = fxs[l], fy = fys[l];
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = pass == 0 ? (tid >> 5) : 8;
      if (pass == 1 && tid >= 32) break;
      float gc[N1], gp[N1];
#pragma unroll
      for (int a = 0; a < N1; ++a) {
        gc[a] = s_dt[l * KL + a * N1 + row][j];
        gp[a] = row > 0 ? s_dt[l * KL + a * N1 + row - 1][j] : 0.f;
      }
      float4* dst0 = reinterpret_cast<float4*>(&s_win[j * WS + row * RS + 4]);
      float d[NRD * 4];
#pragma unroll
      for (int c = 0; c < NRD * 4; ++c) {
        float s = 0.f;
        if (c < N1) s = gc[c] * w00;
        if (c >= 1 && c <= N1) s = fmaf(gc[c - 1], w01, s);
        if (c < N1) s = fmaf(gp[c], w10, s);
        if (c >= 1 && c <= N1) s = fmaf(gp[c - 1], w11, s);
        d[c] = s;
      }
#pragma unroll
      for (int i = 0; i < NRD; ++i) dst0[i] = make_float4(d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]);
      if (row == N1 - 1) {  // the owner of the last tap row also builds the window's last row
        float4* dst1 = reinterpret_cast<float4*>(&s_win[j * WS + (WIN - 1) * RS + 4]);
#pragma unroll
        for (int c = 0; c < NRD * 4; ++c) {
          float s = 0.f;
          if (c < N1) s = gc[c] * w10;
          if (c >= 1 && c <= N1) s = fmaf(gc[c - 1], w11, s);
          d[c] = s;
        }
#pragma unroll
        for (int i = 0; i < NRD; ++i) dst1[i] = make_float4(d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]);
      }
    }
    __syncthreads();
    const char* lds_bytes = reinterpret_cast<const char*>(s_win);
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const float* src = reinterpret_cast<const float*>(lds_bytes + pc[l][i].lds);
      const float e0 = src[0], e1 = src[1], e2 = src[2], e3 = src[3];
      const int m = pc[l][i].mask;
      f32x4 t = v[l][i];
      t.x += (m & 1) ? e0 : 0.f;
      t.y += (m & 2) ? e1 : 0.f;
      t.z += (m & 4) ? e2 : 0.f;
      t.w += (m & 8) ? e3 : 0.f;
      if (pc[l][i].need()) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(slab0) + pc[l][i].goff) = t;
    }
    __syncthreads();   // image free for the next level
  }
}

bool check_levels(const PyrLayout& P) {
  for (int l = 0; l < P.L; ++l)
    if (P.h[l] < 1 || P.w[l] < 1 || P.h[l] > 32000 || P.w[l] > 32000) return false;
  return true;
}

}  // namespace

extern "C" long long pcfa_lookup_convc1_packed_floats(int Cout) {
  if (Cout != 256) return -1;
  return 2LL * Cout * KP;   // forward packing + backward packing
}

extern "C" int pcfa_lookup_convc1_pack_weights(const float* weight, float* packed, int Cout, int Cin, void* stream) {
  if (!weight || !packed) return PCFA_ERR_INVALID_ARG;
  if (Cout != 256 || Cin != L * TAPS) return PCFA_ERR_UNSUPPORTED;
  pcfa_launch(convc1_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, weight, packed,
              packed + (size_t)Cout * KP, Cout);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_lookup_convc1_fwd(const float* pyr, const float* coords, const float* packed, const float* bias,
                                      float* out, int B, int H, int W, int num_levels, int radius, int Cout, int relu,
                                      void* stream) {
  PyrLayout P;
  if (!pyr || !coords || !packed || !bias || !out || B < 1 || !pcfa_make_layout(P, H, W, num_levels) ||
      !check_levels(P))
    return PCFA_ERR_INVALID_ARG;
  if (num_levels != L || radius != R || Cout != 256) return PCFA_ERR_UNSUPPORTED;
  const int Q = H * W;
  static const int dbg = getenv("PCFA_LC_DBG") ? atoi(getenv("PCFA_LC_DBG")) : 0;   // phase ablation (tools/dev)
  pcfa_launch(corr_lookup_convc1_fwd_kernel, dim3(pcfa_cdiv(Q, QT), 1, B), dim3(NT), 0, (hipStream_t)stream, pyr,
              coords, packed, bias, out, Q, P, relu, dbg);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_lookup_convc1_bwd(float* dpyr, const float* coords, const float* packed, const float* out,
                                      const float* grad_out, int B, int H, int W, int num_levels, int radius,
                                      int Cout, int relu, void* stream) {
  PyrLayout P;
  if (!dpyr || !coords || !packed || !out || !grad_out || B < 1 || !pcfa_make_layout(P, H, W, num_levels) ||
      !check_levels(P))
    return PCFA_ERR_INVALID_ARG;
  if (num_levels != L || radius != R || Cout != 256) return PCFA_ERR_UNSUPPORTED;
  const int Q = H * W;
  pcfa_launch(corr_lookup_convc1_bwd_kernel, dim3(pcfa_cdiv(Q, QT), 1, B), dim3(NT), 0, (hipStream_t)stream, dpyr,
              coords, packed + (size_t)Cout * KP, out, grad_out, Q, P, relu);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
