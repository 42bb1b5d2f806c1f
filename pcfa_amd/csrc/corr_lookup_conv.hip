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
constexpr int NT = 256;                                              // threads that load / run the matrix work (4 waves)
constexpr int NTF = NT + 64;                                         // forward: + one auxiliary wave (tap row 8)
constexpr int NPC = QT * PIECES / NT;                                // 5 pieces per thread and level
constexpr int L = 4, TAPS = N1 * N1;                                 // 81 taps per level
constexpr int KL = 88, KG = KL / 8;                                  // padded tap rows per level, groups of 8 rows
constexpr int KP = L * KL;                                           // 352 padded tap rows
constexpr int WIN_FLOATS = QT * WS + 4;
#ifndef PCFA_LC_WD
#define PCFA_LC_WD 8         // depth of the forward kernel's W register ring (groups of 8 MFMAs per wave)
#endif
#ifndef PCFA_LC_WDB
#define PCFA_LC_WDB 6         // depth of the backward kernel's W^T ring (groups of 12 MFMAs per wave)
#endif
#ifndef PCFA_LC_EARLY
#define PCFA_LC_EARLY 2       // forward: pyramid levels requested up front (the rest ride under the first level's MFMAs)
#endif
#ifndef PCFA_LC_AHEAD
#define PCFA_LC_AHEAD 2       // gradient-tile blocks (32 rows) in flight ahead of the backward GEMM
#endif
#ifndef PCFA_LC_DBG_BUILD
#define PCFA_LC_DBG_BUILD 0  // forward phase ablation (tools/dev): 1 = no MFMAs, 2 = no blend, 4 = W stream from L1 (wrong results),
                             // 8 = per-workgroup phase timestamps instead of results
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
// The explicit global address space matters: through a generic pointer these become flat_load, which also counts
// against lgkmcnt -- every LDS / scalar-load wait behind them then waits for HBM.
__device__ __forceinline__ f32x4 load_piece(const float* sbase, unsigned voff_bytes) {
  typedef const f32x4 __attribute__((address_space(1))) * gptr;
  return *(gptr)(reinterpret_cast<unsigned long long>(sbase) + voff_bytes);
}

__device__ __forceinline__ void store_piece(float* sbase, unsigned voff_bytes, f32x4 t) {
  typedef f32x4 __attribute__((address_space(1))) * gptr;
  *(gptr)(reinterpret_cast<unsigned long long>(sbase) + voff_bytes) = t;
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

// Forward kernel's variant: 8 threads per query (k = tid >> 3), each owning pieces q = sub + 8 i (row q >> 2, tile column
// q & 3) of that query's window -- no divisions, no cross-lane traffic; the origin is recomputed from the query's own
// coordinates.  Same piece addresses and LDS image as level_geometry (which the backward still uses: it needs the masks).
// The first version spent ~800 instructions (4 k cycles at one wave per SIMD) here before the first window load left.
struct FwdPieces {
  unsigned goff[NPC], lds[NPC];
};
__device__ __forceinline__ void fwd_pieces(FwdPieces& g, float kx, float ky, bool klive, int level, int k, int sub,
                                           int hl, int tw, int off, int slab, int zero) {
  static_assert(NPC * 8 == PIECES, "8 threads cover a window's pieces");
  const Origin o = make_origin(kx, ky, level);
  const int th4 = ((hl + 3) >> 2) << 2;
  const int x0 = min(max(o.x0, -16), 4 * tw), y0 = klive ? min(max(o.y0, -16), th4) : th4;
  const unsigned wB = (unsigned)(off + k * slab) * 4u;
  const int ox = x0 & 3;
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int q = sub + 8 * i, rr = q >> 2, tx = q & 3;
    const int y = y0 + rr, gtx = (x0 >> 2) + tx;
    const bool need = (unsigned)y < (unsigned)hl && (unsigned)gtx < (unsigned)tw && 4 * tx < ox + WIN;
    g.goff[i] = need ? wB + (((((unsigned)y >> 2) * (unsigned)tw + (unsigned)gtx) << 4) + (((unsigned)y & 3u) << 2)) * 4u
                     : (unsigned)zero * 4u;   // the first slab's zero tile
    g.lds[i] = (unsigned)(k * WS + rr * RS + 4 + 4 * tx - ox) * 4u;
  }
}

// Backward kernel's variant of fwd_pieces: the same ownership plus the per-texel masks of level_geometry.
struct BwdPieces {
  unsigned goff[NPC], lds[NPC];
  int mask[NPC];   // bits 0-3: texel is a window texel inside the level; bit 4: the piece exists in the level
};
__device__ __forceinline__ void bwd_pieces(BwdPieces& g, float kx, float ky, bool klive, int level, int k, int sub,
                                           int hl, int wl, int tw, int off, int slab, int zero) {
  const Origin o = make_origin(kx, ky, level);
  const int th4 = ((hl + 3) >> 2) << 2;
  const int x0 = min(max(o.x0, -16), 4 * tw), y0 = klive ? min(max(o.y0, -16), th4) : th4;
  const unsigned wB = (unsigned)(off + k * slab) * 4u;
  const int ox = x0 & 3;
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int q = sub + 8 * i, rr = q >> 2, tx = q & 3;
    const int y = y0 + rr, gtx = (x0 >> 2) + tx;
    const bool need = (unsigned)y < (unsigned)hl && (unsigned)gtx < (unsigned)tw && 4 * tx < ox + WIN;
    const int c0 = 4 * tx - ox, gx0 = 4 * gtx;
    int mask = need ? 16 : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((unsigned)(c0 + e) < (unsigned)WIN && gx0 + e < wl) mask |= 1 << e;
    g.mask[i] = mask;
    g.goff[i] = need ? wB + (((((unsigned)y >> 2) * (unsigned)tw + (unsigned)gtx) << 4) + (((unsigned)y & 3u) << 2)) * 4u
                     : (unsigned)zero * 4u;
    g.lds[i] = (unsigned)(k * WS + rr * RS + 4 + 4 * tx - ox) * 4u;
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
// MTW = row tiles (32 output channels) per wave: 2 = one workgroup per query tile owns all 256 channels (220 workgroups
// at 55x128).  (MTW = 1 -- the channels split over two workgroups per tile that share a CU, two waves per SIMD -- was
// measured: both workgroups gather and blend the whole tile, 21.0 us against 19.8.)
// Threads: waves 0-3 load the windows, blend tap rows 0-7 and run the matrix work; wave 4 only blends tap row 8 (9 rows
// x 32 queries = 288 (query, row) items: 256 + 32).
template <int MTW>
__global__ __launch_bounds__(NTF) void corr_lookup_convc1_fwd_kernel(
    const float* __restrict__ pyr, const float* __restrict__ coords, const float* __restrict__ wp,
    const float* __restrict__ bias, float* __restrict__ out, int Q, PyrLayout P, int relu, int /*unused*/) {
  constexpr int dbg = PCFA_LC_DBG_BUILD;             // phase ablation is a build-time switch: a runtime one put 130
                                                     // uniform branches between the MFMAs and the allocator spilled
  constexpr int COUT = 256;
  __shared__ __attribute__((aligned(16))) float s_win2[2][WIN_FLOATS];   // two window images: level i+1 is staged while i is read
  __shared__ __attribute__((aligned(16))) float s_tap[2][KL][QT];
  __shared__ float s_bias[COUT];
  static_assert(COUT == NT, "one bias value per thread");

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = tid & 31, l31 = lane & 31, lh = lane >> 5;
  const int b_img = blockIdx.z, q0 = blockIdx.x * QT;
  const int mt0 = ((int)blockIdx.y * 4 + wv) * MTW;   // this wave's first row tile
  const bool live = q0 + j < Q;
  unsigned long long ts[12] = {};
  int nts = 0;
#define PCFA_LC_STAMP() do { if (dbg & 8) ts[nts++] = __builtin_amdgcn_s_memtime(); } while (0)
  PCFA_LC_STAMP();
  // branch-free head: clamped query indices instead of `if (live)` (the branches kept the kernel-argument loads and
  // the geometry of the four levels from being scheduled together)
  const int kq = tid >> 3, sub = tid & 7;             // piece ownership: 8 threads per query
  const bool klive = q0 + kq < Q;
  const float* cb = coords + (size_t)b_img * 2 * Q;
  const int qj = min(q0 + j, Q - 1), qk = min(q0 + kq, Q - 1);
  const float cx = cb[qj], cy = cb[Q + qj], kx = cb[qk], ky = cb[Q + qk];
  const float bias_mine = bias[min(tid, COUT - 1)];   // -> LDS, read back by the epilogue
  const float* slab0 = scalar_ptr(pyr + ((size_t)b_img * Q + q0) * P.slab);

  if (wv == 4) {
    // ---- auxiliary wave: tap row 8 of every query (lanes 0-31), in step with the main waves' barriers; it also
    //      clears the zero rows 81..87 of both tap tiles once ----
    float fxa[L], fya[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const Origin o = make_origin(cx, cy, l);
      fxa[l] = o.fx;
      fya[l] = o.fy;
    }
    for (int e = lane; e < 2 * (KL - TAPS) * QT; e += 64) {
      const int buf = e / ((KL - TAPS) * QT), r = e - buf * ((KL - TAPS) * QT);
      s_tap[buf][TAPS + r / QT][r % QT] = 0.f;
    }
    auto aux_blend = [&](int i) {
      if (lane < QT) {
        const int l = L - 1 - i;
        const float fx = fxa[l], fy = fya[l];
        const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
        const f32x4* pa = reinterpret_cast<const f32x4*>(&s_win2[i & 1][j * WS + (N1 - 1) * RS + 4]);
        const f32x4* pb = reinterpret_cast<const f32x4*>(&s_win2[i & 1][j * WS + N1 * RS + 4]);
        const f32x4 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
        const float ra[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
        const float rb[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int a = 0; a < N1; ++a)
          s_tap[i & 1][a * N1 + (N1 - 1)][j] = ra[a] * w00 + ra[a + 1] * w01 + rb[a] * w10 + rb[a + 1] * w11;
      }
    };
    if (dbg & 16) {
      __syncthreads();
      return;
    }
    __syncthreads();   // (1) the coarsest level's image is written
    if (!(dbg & 2)) aux_blend(0);
    __syncthreads();   // (2) prologue
#pragma unroll
    for (int i = 0; i < L; ++i) {
      if (i + 1 < L && !(dbg & 2)) aux_blend(i + 1);
      __syncthreads();
    }
    if (dbg & 8) __syncthreads();
    return;
  }

  // ---- every piece of all four levels is requested before anything waits; coarsest level first: its windows are
  //      L2-resident and land first, so the matrix cores start while the level-0 texels are still on their way.
  //      Issue order is pinned (sched_barrier): level 3, the W ring's first groups, then levels 2, 1, 0 -- returns are
  //      in order, so anything issued ahead of the W ring delays the first MFMA. ----
  f32x4 v[L][NPC];
  unsigned dst[L][NPC], goffs[L][NPC];
  float fxs[L], fys[L];
  constexpr int EARLY = PCFA_LC_EARLY;   // levels (coarsest first) whose windows are requested before the first MFMA
  const f32x4* wq = reinterpret_cast<const f32x4*>(wp) + ((size_t)mt0 * (KP / 8)) * 64 + lane;
  // W operands: a register ring WD groups deep over the 44 groups of all four levels (the loops below are fully
  // unrolled, so every ring index is static).  One group = 4 k-pairs = 8 MFMAs of 64 cycles per wave; the loads come
  // from L2 (500-900 cycles), so the ring has to run >= 2 groups ahead -- with one group of lead the matrix pipe
  // idled for the load latency in every group (26 us per launch, 28 % of the fp32 matrix peak).
  constexpr int WD = PCFA_LC_WD, NG = L * KG;
  f32x4 wring[WD][MTW];
  int Ph[L], Ptw[L], Poff[L];     // all kernel-argument loads up front (the fences below would pin them per level)
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Ph[l] = P.h[l];
    Ptw[l] = P.tw[l];
    Poff[l] = P.off[l];
  }
  const int Pslab = P.slab, Pzero = P.zero;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int lr = 0; lr < L; ++lr) {
    const int l = L - 1 - lr;
    FwdPieces g;
    fwd_pieces(g, kx, ky, klive, l, kq, sub, Ph[l], Ptw[l], Poff[l], Pslab, Pzero);
    const Origin oj = make_origin(cx, cy, l);
    fxs[l] = oj.fx;
    fys[l] = oj.fy;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      dst[l][i] = g.lds[i];
      goffs[l][i] = g.goff[i];
      if (lr < EARLY) v[l][i] = load_piece(slab0, g.goff[i]);   // the two fine levels are requested under level 3's MFMAs
    }
    __builtin_amdgcn_sched_barrier(0);
    if (lr == 0) {
#pragma unroll
      for (int d = 0; d < WD; ++d)
#pragma unroll
        for (int m = 0; m < MTW; ++m) wring[d][m] = wq[((size_t)m * (KP / 8) + (L - 1 - d / KG) * KG + d % KG) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  s_bias[tid] = bias_mine;   // (tid < 256 here: the auxiliary wave has left)

  // Pipeline over the levels in stream order i = 0..3 (level L-1-i), ONE barrier per level:
  //   before iteration i : taps(i) are in s_tap[i & 1], the window image of i+1 is in s_win2[(i+1) & 1]
  //   iteration i        : the MFMAs of level i, and BETWEEN them -- a 64-cycle fp32 MFMA leaves ~12 issue slots --
  //                        the blend of level i+1 (one tap column per MFMA group) and the image write of level i+2.
  // The first version ran image write -> barrier -> blend -> barrier -> MFMAs per level with one wave per SIMD, so
  // none of the ~1 us of staging per level overlapped the matrix work.
  auto write_image = [&](int i) {
    const int l = L - 1 - i;
    char* lds_bytes = reinterpret_cast<char*>(s_win2[i & 1]);
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      float* d = reinterpret_cast<float*>(lds_bytes + dst[l][k]);
      d[0] = v[l][k].x; d[1] = v[l][k].y; d[2] = v[l][k].z; d[3] = v[l][k].w;
    }
  };
  // Blend: thread (query j, b8 = tid >> 5) owns tap ROW b8 of its query's window, the nine taps (a, b8), a = 0..8.  They
  // share two window rows: six aligned 16-B LDS reads (the un-fused kernel's scheme, corr_lookup.hip) instead of 4 dword
  // reads per tap -- the first version gave every thread eleven scattered taps (44 dword reads, 77 VALU per level) and
  // the matrix pipe waited ~1-2 k cycles per level for them.  Row 8 belongs to the auxiliary wave.
  struct Rows { f32x4 a[3], b[3]; };
  auto row_read = [&](int i, int b, f32x4 (&dst)[3]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(&s_win2[i & 1][j * WS + b * RS + 4]);
    dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
  };
  auto tap_out = [&](int i, int a, int b, const Rows& R) {   // tap (a, b) of stream level i from rows b, b + 1
    const int l = L - 1 - i;
    const float fx = fxs[l], fy = fys[l];
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    const float t00 = R.a[a >> 2][a & 3], t01 = R.a[(a + 1) >> 2][(a + 1) & 3];
    const float t10 = R.b[a >> 2][a & 3], t11 = R.b[(a + 1) >> 2][(a + 1) & 3];
    s_tap[i & 1][a * N1 + b][j] = t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11;     // as corr_lookup_fwd_body
  };
  const int b8 = tid >> 5;
  static_assert(KG >= N1 + 1, "one tap of the row rides under each of the MFMA groups 1..9");

  if (dbg & 16) {    // arrival time of every level's windows (the image writes wait for them); results are garbage
#pragma unroll
    for (int i = 0; i < L; ++i) {
      write_image(i);
      __builtin_amdgcn_s_waitcnt(0);
      PCFA_LC_STAMP();
    }
    __syncthreads();
    if (live) out[(size_t)b_img * COUT * Q + (size_t)(8 + (tid >> 5)) * Q + q0 + j] = s_win2[0][tid] + s_win2[1][tid];
    if (tid == 0)
      for (int k = 0; k < 5; ++k) out[(size_t)b_img * COUT * Q + (size_t)k * Q + q0] = (float)(ts[k] - ts[0]);
    return;
  }
  write_image(0);
  __syncthreads();
  PCFA_LC_STAMP();   // 1: coarsest level's windows have arrived
  f32x16 acc[MTW];   // starts at the bias (as the un-fused GEMM + bias rounds; the saved output is the ReLU mask)
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = s_bias[(mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
  if (!(dbg & 2)) {
    Rows R0;
    row_read(0, b8, R0.a);
    row_read(0, b8 + 1, R0.b);
#pragma unroll
    for (int a = 0; a < N1; ++a) tap_out(0, a, b8, R0);
  }
  write_image(1);
  __syncthreads();
  PCFA_LC_STAMP();   // 2: prologue done

  // The scheduler is fenced (sched_barrier) at three points of every group.  Left alone it sank each LDS read next to
  // its use, so every tap cost a full LDS round trip during which the wave could not issue the next MFMA, and the
  // matrix pipe idled for more than half of the loop (23 us for 9.4 us of MFMA work).
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const float (*tap)[QT] = s_tap[i & 1];
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = tap[2 * e + lh][l31];     // group 0 of the level: written before the barrier
    Rows R;   // the two window rows of the NEXT level's tap row b8 (read under group 0, one tap per group 1..9)
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const int G = i * KG + g;                      // position in the W stream (compile-time after unrolling)
      // Eight MFMAs, each followed by a small chunk of the other work (a 64-cycle MFMA leaves ~12 issue slots), the
      // scheduler fenced after every pair so the chunks stay where they are put:
      //   after MFMA 0 / 4 (group 0): the next level's two window rows (3 x 16-B LDS reads each)
      //   after MFMA 1: the next group's tap reads      after MFMA 2: the W group WD ahead
      //   after MFMA 5 (groups 1..9): one tap of the next level: blend arithmetic + store
      //   after MFMA 6: (last group) the image of level i + 2
      float bvn[4] = {0.f, 0.f, 0.f, 0.f};
      f32x4 wcur[MTW];
#pragma unroll
      for (int m = 0; m < MTW; ++m) wcur[m] = wring[G % WD][m];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          const int step = (e * MTW + m) * (2 / MTW);   // chunk slots 0..7 of a group (MTW = 1: every other one)
          // tap rows 82..87 of a level are zero rows (81 taps padded to 88): the last group runs its first k-pair only
          if (!(dbg & 1) && !(g == KG - 1 && e >= 1)) {
            const float av = e == 0 ? wcur[m].x : e == 1 ? wcur[m].y : e == 2 ? wcur[m].z : wcur[m].w;
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[e], acc[m], 0, 0, 0);
          }
          if (step == 0 && g == 0 && i + 1 < L && !(dbg & 2)) row_read(i + 1, b8, R.a);
          if (step == 4 && g == 0 && i + 1 < L && !(dbg & 2)) row_read(i + 1, b8 + 1, R.b);
          if (step == (MTW == 1 ? 0 : 1) && g + 1 < KG) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) bvn[ee] = tap[8 * (g + 1) + 2 * ee + lh][l31];
          }
          if (step == 2 && G + WD < NG) {   // group G + WD of the stream: level L-1 - (G+WD)/KG, group (G+WD) % KG
            const int Gn = G + WD, ln = L - 1 - Gn / KG, gn = Gn % KG;
#pragma unroll
            for (int mm = 0; mm < MTW; ++mm)
              wring[G % WD][mm] = wq[((size_t)mm * (KP / 8) + ((dbg & 4) ? (gn & 1) : ln * KG + gn)) * 64];
          }
          if (step == (MTW == 1 ? 2 : 3) && i == 0 && g < (L - EARLY) * NPC) {   // one deferred window piece per group, finer level last
            const int l = L - 1 - EARLY - g / NPC, k = g % NPC;
            v[l][k] = load_piece(slab0, goffs[l][k]);
          }
          if (step == (MTW == 1 ? 4 : 5) && i + 1 < L && g >= 1 && g <= N1 && !(dbg & 2)) tap_out(i + 1, g - 1, b8, R);
          if (step == 6 && i + 2 < L && g == KG - 1 && !(dbg & 64)) write_image(i + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = bvn[e];
    }
    __syncthreads();   // taps(i+1) and image(i+2) visible; everyone is done with taps(i) and image(i+1)
    PCFA_LC_STAMP();   // 3..6: level i done
  }

  // ---- epilogue: + bias, ReLU, rows of 32 queries (128 B) per half-wave ----
  const bool qlive = q0 + l31 < Q;
  float* ob = out + (size_t)b_img * COUT * Q + q0 + l31;
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float y = acc[m][r];
      if (relu) y = fmaxf(y, 0.f);
      if (qlive) ob[(size_t)n * Q] = y;
    }
  if (dbg & 8) {     // timestamps over the first query's column (tools/dev/lc_stamps.py); 100 MHz ticks
    PCFA_LC_STAMP();
    __syncthreads();
    if (tid == 0)
      for (int k = 0; k < 8; ++k) out[(size_t)b_img * COUT * Q + (size_t)k * Q + q0] = (float)(ts[k] - ts[0]);
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
  unsigned long long ts[12] = {};
  int nts = 0;
  constexpr int dbg = PCFA_LC_DBG_BUILD;
  PCFA_LC_STAMP();
  // branch-free head; the gradient tile's loads go out FIRST (they gate the matrix work), the window pieces -- needed
  // only after the GEMM -- behind them.  The first version did the ~800-instruction geometry and the 20 piece loads
  // first, and the in-order return then held the gradient tile behind all of them.
  const int kq = tid >> 3, sub = tid & 7;             // piece ownership: 8 threads per query (as the forward)
  const bool klive = q0 + kq < Q;
  const float* cb = coords + (size_t)b_img * 2 * Q;
  const int qj = min(q0 + j, Q - 1), qk = min(q0 + kq, Q - 1);
  const float cx = cb[qj], cy = cb[Q + qj], kx = cb[qk], ky = cb[Q + qk];
  float* slab0 = scalar_ptr(dpyr + ((size_t)b_img * Q + q0) * P.slab);

  // ---- gradient tile: g[n][q] = grad_out * [y > 0]; thread = (4 queries, 8 rows apart): 16-B loads, all 16 of a
  //      thread in flight at once (per-float loads in four dependent batches were ~6 us of the launch) ----
  const int qq = (tid & 7) * 4, n0 = tid >> 3;      // 8 quads per row, 32 rows per pass, 8 passes
  const size_t gbase = (size_t)b_img * COUT * Q + q0 + qq;
  // workgroup-uniform: every row of the tile is 32 whole, 16-B aligned queries.  (A per-thread test here put a
  // divergent branch in every pass and the loads went out one dependent pair at a time: 9 us before the first MFMA.)
  const bool full = __builtin_amdgcn_readfirstlane((Q & 3) == 0 && q0 + QT <= Q);
  // The tile is streamed in 8 blocks of 32 rows (= 4 GEMM groups each), two blocks ahead of the matrix cores: all 16
  // loads of a thread up front meant 14 MB requested across the chip at once, and the first MFMA waited ~6 us for its
  // whole tile.  Loads are issued unconditionally (a ragged tile reads clamped, in-bounds addresses and ignores the
  // result): behind a branch the compiler can no longer count the loads in flight.
  constexpr int NBLK = COUT / 32, AHEAD = PCFA_LC_AHEAD;
  f32x4 gg[NBLK], yy[NBLK];
  const float* ysrc = relu ? y : grad_out;   // without the ReLU: a second read of the gradient, mask always true
  const size_t glast = (size_t)gridDim.z * COUT * Q - 4;
  auto tile_load = [&](int i) {
    const size_t o = min(gbase + (size_t)(n0 + 32 * i) * Q, glast);
    gg[i] = *reinterpret_cast<const f32x4*>(grad_out + o);
    yy[i] = *reinterpret_cast<const f32x4*>(ysrc + o);
  };
  auto tile_store = [&](int i) {
    f32x4 t;
    t.x = (yy[i].x > 0.f || !relu) ? gg[i].x : 0.f; t.y = (yy[i].y > 0.f || !relu) ? gg[i].y : 0.f;
    t.z = (yy[i].z > 0.f || !relu) ? gg[i].z : 0.f; t.w = (yy[i].w > 0.f || !relu) ? gg[i].w : 0.f;
    *reinterpret_cast<f32x4*>(&s_g[n0 + 32 * i][qq]) = t;
  };
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) tile_load(i);
  __builtin_amdgcn_sched_barrier(0);

  // ---- the read-modify-write of all four levels: its geometry AND its loads ride inside the GEMM (see below); only
  //      the kernel arguments are fetched here.  Done before the GEMM, the geometry's ~600 dependent instructions
  //      (one wave per SIMD) held the first MFMA back by ~5 us. ----
  f32x4 v[L][NPC];
  unsigned pgoff[L][NPC], plds[L][NPC];
  int pmask[L][NPC];
  int Ph[L], Pw[L], Ptw[L], Poff[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Ph[l] = P.h[l];
    Pw[l] = P.w[l];
    Ptw[l] = P.tw[l];
    Poff[l] = P.off[l];
  }
  const int Pslab = P.slab, Pzero = P.zero;
  struct LevelOrg { int x0, y0, ox; unsigned wB; };
  LevelOrg org[L];
  auto level_setup = [&](int l) {
    const Origin o = make_origin(kx, ky, l);
    const int th4 = ((Ph[l] + 3) >> 2) << 2;
    org[l].x0 = min(max(o.x0, -16), 4 * Ptw[l]);
    org[l].y0 = klive ? min(max(o.y0, -16), th4) : th4;
    org[l].ox = org[l].x0 & 3;
    org[l].wB = (unsigned)(Poff[l] + kq * Pslab) * 4u;
  };
  auto piece_setup = [&](int l, int i) {   // as bwd_pieces, one piece
    const int q = sub + 8 * i, rr = q >> 2, tx = q & 3;
    const int hl = Ph[l], wl = Pw[l], tw = Ptw[l], ox = org[l].ox;
    const int y = org[l].y0 + rr, gtx = (org[l].x0 >> 2) + tx;
    const bool need = (unsigned)y < (unsigned)hl && (unsigned)gtx < (unsigned)tw && 4 * tx < ox + WIN;
    const int c0 = 4 * tx - ox, gx0 = 4 * gtx;
    int mask = need ? 16 : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((unsigned)(c0 + e) < (unsigned)WIN && gx0 + e < wl) mask |= 1 << e;
    pmask[l][i] = mask;
    pgoff[l][i] = need ? org[l].wB + (((((unsigned)y >> 2) * (unsigned)tw + (unsigned)gtx) << 4) + (((unsigned)y & 3u) << 2)) * 4u
                       : (unsigned)Pzero * 4u;
    plds[l][i] = (unsigned)(kq * WS + rr * RS + 4 + 4 * tx - ox) * 4u;
  };
  __builtin_amdgcn_sched_barrier(0);
  PCFA_LC_STAMP();   // 1: head done (the gradient tile's loads are in flight)
  if (!full) {   // ragged last tile, or rows that are not 16-B aligned: the whole tile per-float, rolled, up front
#pragma unroll 1
    for (int i = 0; i < NBLK; ++i) {
      const size_t o = gbase + (size_t)(n0 + 32 * i) * Q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = q0 + qq + e < Q;
        const float gv = ok ? grad_out[o + e] : 0.f;
        const float yv = (ok && relu) ? y[o + e] : 1.f;
        s_g[n0 + 32 * i][qq + e] = yv > 0.f ? gv : 0.f;
      }
    }
  }

  // ---- d taps [352][32] = W^T [352][256] . g [256][32]; wave w owns row tiles w, w + 4, w + 8 ----
  {
    f32x16 acc[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const f32x4* wq = reinterpret_cast<const f32x4*>(wt) + lane;
    const int nmt = wv + 8 < MT ? 3 : 2;
    // W^T operands: register ring WD groups deep (one group = 12 MFMAs = 768 cycles here), fully unrolled so every
    // ring index is static; the next group's gradient rows are read one group ahead, and the scheduler is fenced
    // after every MFMA step so that neither the LDS reads nor the L2 loads sink next to their uses (see the forward).
    // The first version (ring of 3, reads at their uses) ran this loop at 54 % of the MFMA rate.
    constexpr int WD = PCFA_LC_WDB, NGB = COUT / 8;
    f32x4 wring[WD][MTW];
    size_t wrow[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) wrow[m] = (size_t)min(wv + 4 * m, MT - 1) * NGB;
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
      for (int m = 0; m < MTW; ++m) wring[d][m] = wq[(wrow[m] + d) * 64];
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NGB; ++g) {
      if (g % 4 == 0) {   // block g / 4 of the tile: registers -> LDS, one barrier, then its four groups
        if (full) tile_store(g / 4);
        __syncthreads();
        if (g == 0) PCFA_LC_STAMP();   // 2: first block of the gradient tile staged
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = s_g[8 * g + 2 * e + lh][l31];
      }
      float bvn[4] = {0.f, 0.f, 0.f, 0.f};
      f32x4 wcur[MTW];
#pragma unroll
      for (int m = 0; m < MTW; ++m) wcur[m] = wring[g % WD][m];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < MTW; ++m) {   // wave 3's third tile is a clamped duplicate, computed and dropped: a
          // wave-dependent branch around every third MFMA cost more than the MFMA (the other waves run 3 anyway)
          const int step = e * MTW + m;
          const float av = e == 0 ? wcur[m].x : e == 1 ? wcur[m].y : e == 2 ? wcur[m].z : wcur[m].w;
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[e], acc[m], 0, 0, 0);
          if (step == 0 && (g + 1) % 4 != 0) {   // the next group's rows, unless they belong to the next block
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) bvn[ee] = s_g[8 * (g + 1) + 2 * ee + lh][l31];
          }
          if (step == 4 && g % 4 == 0 && g / 4 + AHEAD < NBLK) tile_load(g / 4 + AHEAD);
          if (step == 1 && g + WD < NGB) {
#pragma unroll
            for (int mm = 0; mm < MTW; ++mm) wring[g % WD][mm] = wq[(wrow[mm] + g + WD) * 64];
          }
          // The read half of the read-modify-write, one piece per group over the first 20 groups, in the order the
          // scatter needs them.  All 20 up front (44 MB with the gradient tiles, across the chip) held the gradient
          // tile back for 9 us; all 20 right before the GEMM would hold the W ring back instead (returns are in order).
          if (step == 2 && g < L * NPC && g % NPC == 0) level_setup(g / NPC);
          if (step == 3 && g < L * NPC) piece_setup(g / NPC, g % NPC);
          if (step == 5 && g < L * NPC) v[g / NPC][g % NPC] = load_piece(slab0, pgoff[g / NPC][g % NPC]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = bvn[e];
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
  PCFA_LC_STAMP();   // 3: GEMM done

  // ---- per level: the lookup's transpose (corr_lookup_bwd_body with the tap gradients read from LDS) ----
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const Origin oj = make_origin(cx, cy, l);
    const float fx = oj.fx, fy = oj.fy;
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
      const float* src = reinterpret_cast<const float*>(lds_bytes + plds[l][i]);
      const float e0 = src[0], e1 = src[1], e2 = src[2], e3 = src[3];
      const int m = pmask[l][i];
      f32x4 t = v[l][i];
      t.x += (m & 1) ? e0 : 0.f;
      t.y += (m & 2) ? e1 : 0.f;
      t.z += (m & 4) ? e2 : 0.f;
      t.w += (m & 8) ? e3 : 0.f;
      if (m & 16) store_piece(slab0, pgoff[l][i], t);
    }
    __syncthreads();   // image free for the next level
    PCFA_LC_STAMP();   // 4..7: level l scattered
  }
  if (dbg & 8) {     // timestamps over the first floats of the first query's slab (tools/dev/lc_stamps.py --bwd)
    if (tid == 0)
      for (int k = 0; k < 8; ++k) slab0[k] = (float)(ts[k] - ts[0]);
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
  pcfa_launch(corr_lookup_convc1_fwd_kernel<2>, dim3(pcfa_cdiv(Q, QT), 1, B), dim3(NTF), 0, (hipStream_t)stream, pyr,
              coords, packed, bias, out, Q, P, relu, 0);
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
