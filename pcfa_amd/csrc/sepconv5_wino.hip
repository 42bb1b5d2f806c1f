// The SepConvGRU gate convolutions (models/raft/update.py:33-60: Conv2d(c, 128, (1,5)) / ((5,1))) as a 1-D Winograd
// F(2,5) on v_mfma_f32_32x32x2_f32, forward and data gradient, with the four fused GRU epilogues of sepconv5.hip.
//
//   y[n][x] = sum_c sum_t w[n][c][t] in[c][x + t - 2].   Pair j = outputs (2j, 2j+1) from d_i = in[c][2j - 2 + i], i = 0..5:
//   V = B^T d (interpolation points 0, +-1, +-2, inf: the row stage of the F(4x4,3x3) kernel), M_i[n][j] += U_i[n][c] V_i[c][j]
//   (ONE MFMA per point and channel pair), y = A^T M:  6 products per 2 outputs instead of 10 -- 1.67x fewer MFMAs than the
//   direct implicit GEMM of sepconv5.hip.  U = G w is packed once per weight (both directions) in MFMA operand order.
//   1x5: a pair is two neighbouring pixels of a row (lane = pair, float2 operands / stores);
//   5x1: a pair is two neighbouring rows at one x (lane = x, six patch rows).
// Rounding (tools/dev/sepconv5_wino, against fp64 at 256 -> 256 channels): 1.0e-6 rms relative, 7.4e-7 with the channel
// split below (the direct fp32 convolution: 6.4e-7).
//
// Workgroup = KS groups of 2 WN waves.  A group owns every KS-th chunk of 8 input channels (own double-buffered LDS patch,
// shared barriers); wave (wn, wpx) of a group owns 32 output channels x 32 pairs with six accumulators (one per
// point).  After the K loop every wave applies the (linear) output transform and leaves its 32 partial outputs per lane in
// LDS; group g then finalises accumulator rows [16 g / KS, 16 (g+1) / KS): it adds the KS partials in group order
// (deterministic) and runs the epilogue on them, so all groups share the epilogue work and its operand registers
// (requested before the K loop, as in sepconv5.hip).  WN = 2, KS = 2 for 256 output channels (220 workgroups x 8 waves at
// 55x128), WN = 1, KS = 4 for 128 (220 x 8: the plain grid would be one wave per SIMD on 220 CUs).
// Eligible shapes (everything else runs sepconv5.hip's direct kernel): Cin % (16 KS) == 0 with the a|b operand boundary
// on a multiple of 8, Cout % 64 == 0 (WN = 2) or % 32 (WN = 1), W % 128 == 0 (1x5) / W % 64 == 0 (5x1), 16-B aligned
// tensors, output split on a multiple of 32.
#include <cstdlib>
#include "sepconv5.hpp"

// Timing-only ablation builds (tools/dev/sc5w_ablate.sh; results are WRONG with any bit set): PCFA_SC5W_DBG bit 1 = no
// barrier in the K loop, 2 = no input transform (V = d), 4 = no LDS operand reads, 8 = no weight loads after the first
// chunk, 16 = no patch loads / LDS stores after the first chunk.
#ifndef PCFA_SC5W_DBG
#define PCFA_SC5W_DBG 0
#endif
// PCFA_SC5W_WLDS = 1: the U operands reach the waves through LDS.  The two pixel-half waves of a group multiply the same
// 6 KB of U per chunk; instead of each fetching them from L2 into registers (48 KB per CU and chunk), every wave copies
// half of them global -> LDS with three LDS-DMA instructions (global_load_lds_dwordx4: 1 KB each, no VGPR staging) two
// chunks ahead, and reads its 24 operands of the NEXT chunk back with six ds_read_b128 under this chunk's MFMAs.
#ifndef PCFA_SC5W_WLDS
#define PCFA_SC5W_WLDS 1
#endif
#ifndef PCFA_SC5W_STORE_AT
#define PCFA_SC5W_STORE_AT 1   // channel pair after whose MFMAs the next patch is written to LDS (4 = after the last: r04a)
#endif

namespace {
using namespace pcfa_sc5;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CK = 8, STEPS = (CK / 2) * 6;
// 1x5 patch [ch][136]: x0 - 4 .. x0 + 131 (float4 pieces, out-of-image pieces zero)
constexpr int HPX = 128, HRV = HPX / 4 + 2, HCHS = 4 * HRV, HPATCH = CK * HCHS, HNV = CK * HRV;
// 5x1 patch [ch][6 rows][64]: rows 2 rp - 2 .. 2 rp + 3
constexpr int VPX = 64, VCHS = 6 * 64, VPATCH = CK * VCHS, VNV = CK * 6 * 16;

// source of the out-of-image 16-B pieces when the patch is staged by LDS-DMA (a DMA cannot substitute a value: its
// per-lane SOURCE address points here instead)
__device__ __attribute__((aligned(16))) float sc5_zero_piece[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ float wino_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }   // as sepconv5.hip / gru_math.hip

__device__ __forceinline__ const float* wino_plane(const Operand& in, int ci, long long plane) {
  return ci < in.Ca ? in.a + ci * plane : in.b + (ci - in.Ca) * plane;
}

// U[nb][chunk][s / 4][lane][s % 4] = (G w)_i [n = 32 nb + (lane & 31)] [c = 8 chunk + 2 p + (lane >> 5)],  s = 6 p + i:
// a wave fetches the 24 operands of a chunk with six fully coalesced 16-B-per-lane loads (24 dword loads kept the
// texture-address path as busy as the matrix pipe: a wave-wide load costs ~16 cycles of address processing whatever its
// width -- 8 waves x 24 x 16 = 3072 cycles per chunk against 3072 cycles of MFMA).
//   transpose = 0: w'[n][c][t] = w[n][c][t]         (forward: N = Cout, C = Cin of the Conv2d weight [N][C][5])
//   transpose = 1: w'[n][c][t] = w[c][n][4 - t]     (data gradient: N = Cin, C = Cout of the weight [C][N][5])
__global__ void sc5_wino_pack_kernel(const float* __restrict__ w, float* __restrict__ P, int N, int C, int transpose,
                                     long long total) {
  const double G[6][5] = {{0.25, 0., 0., 0., 0.},
                          {-1. / 6, -1. / 6, -1. / 6, -1. / 6, -1. / 6},
                          {-1. / 6, 1. / 6, -1. / 6, 1. / 6, -1. / 6},
                          {1. / 24, 2. / 24, 4. / 24, 8. / 24, 16. / 24},
                          {1. / 24, -2. / 24, 4. / 24, -8. / 24, 16. / 24},
                          {0., 0., 0., 0., 1.}};
  const int nchunk = C / CK;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)((e >> 2) & 63), s = 4 * (int)((e >> 8) % (STEPS / 4)) + (int)(e & 3);
    const long long blk = (e >> 6) / STEPS;
    const int chunk = (int)(blk % nchunk), nb = (int)(blk / nchunk);
    const int n = 32 * nb + (lane & 31), c = chunk * CK + 2 * (s / 6) + (lane >> 5), i = s % 6;
    double u = 0.0;
    if (n < N && c < C)
      for (int t = 0; t < 5; ++t)
        u += G[i][t] * (double)(transpose ? w[((long long)c * N + n) * 5 + (4 - t)] : w[((long long)n * C + c) * 5 + t]);
    P[e] = (float)u;
  }
}

template <bool VERT, int WN, int KS, int MODE>
__global__ __launch_bounds__(128 * WN * KS) void sc5_wino_kernel(Operand in, const float* __restrict__ wp, OutSplit out,
                                                                 int Cout, int H, int W, int tiles_x, GruEpi epi) {
  constexpr int GT = 128 * WN, WPG = 2 * WN;
  constexpr int PATCH = VERT ? VPATCH : HPATCH, NV = VERT ? VNV : HNV, NLOAD = (NV + GT - 1) / GT;
  constexpr int RN = 16 / KS;
  constexpr int WBLK = STEPS * 64;   // floats of one 32-channel block's U operands per chunk (6 KB)
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x / GT);
  float* smem = smem_all + grp * 2 * PATCH;
  float* wlds = smem_all + KS * 2 * PATCH + grp * 2 * WN * WBLK;   // [slot][wn][WBLK] of this group (PCFA_SC5W_WLDS)
  const int tid = threadIdx.x % GT, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave inside the group
  const int wn = WN == 2 ? (wv & 1) : 0, wpx = WN == 2 ? (wv >> 1) : wv;
  // (an XCD-aware tile map -- common.hpp, as in conv3x3.hip -- was measured neutral here: 306.7 vs 305.7 us per GRU update)
  const int ty = blockIdx.x / tiles_x, x0 = (blockIdx.x - ty * tiles_x) * (VERT ? VPX : HPX);   // ty: row (1x5) / row pair (5x1)
  const int nb = blockIdx.y * WN + wn;
  const long long plane = (long long)H * W;
  const int nchunk_all = in.Cin / CK, nchunk = nchunk_all / KS;   // chunks of THIS group: global chunk KS c + grp
  in.a += (long long)blockIdx.z * in.Ca * plane;
  if (in.b) in.b += (long long)blockIdx.z * (in.Cin - in.Ca) * plane;
  out.a += (long long)blockIdx.z * out.Ca * plane;
  if (out.b) out.b += (long long)blockIdx.z * (Cout - out.Ca) * plane;

  // ---- staging: every load unconditional from a clamped address, zero padding applied at the LDS write ------------
  int poff[NLOAD], pch[NLOAD], plds[NLOAD];
  unsigned okm = 0;
#pragma unroll
  for (int k = 0; k < NLOAD; ++k) {
    const int e = min(tid + GT * k, NV - 1);
    if (VERT) {
      const int c = e / 96, rem = e - c * 96, r = rem >> 4, v = rem & 15;
      const int iy = 2 * ty - 2 + r;
      okm |= (unsigned)((int)(iy >= 0) & (int)(iy < H)) << k;
      poff[k] = min(max(iy, 0), H - 1) * W + x0 + 4 * v;
      pch[k] = c;
      plds[k] = c * VCHS + r * 64 + 4 * v;
    } else {
      const int c = e / HRV, v = e - c * HRV;
      const int ix = x0 - 4 + 4 * v;
      okm |= (unsigned)((int)(ix >= 0) & (int)(ix + 3 < W)) << k;
      poff[k] = ty * W + min(max(ix, 0), W - 4);
      pch[k] = c;
      plds[k] = c * HCHS + 4 * v;
    }
  }
  auto load_patch = [&](int chunk, f32x4 (&rr)[NLOAD]) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k)
      rr[k] = *reinterpret_cast<const f32x4*>(wino_plane(in, (KS * chunk + grp) * CK + pch[k], plane) + poff[k]);
  };
  auto store_patch = [&](int buf, const f32x4 (&rr)[NLOAD]) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      const f32x4 t = (okm >> k & 1u) ? rr[k] : zero;
      if (tid + GT * k < NV) *reinterpret_cast<f32x4*>(smem + buf * PATCH + plds[k]) = t;
    }
  };
  // PCFA_SC5W_WLDS: the same pieces by LDS-DMA -- piece e = tid + GT k lies at LDS float 4 e of its buffer (both patch
  // layouts are contiguous in e), so one wave-instruction fills 1 KB; out-of-image pieces read sc5_zero_piece; no VGPR
  // staging, no ds_write pass, nothing for the compiler to wait on in the middle of a chunk
  auto dma_patch = [&](int chunk, int buf) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const float* src = (okm >> k & 1u) ? wino_plane(in, (KS * chunk + grp) * CK + pch[k], plane) + poff[k] : sc5_zero_piece;
      float* dst = smem + buf * PATCH + 4 * (GT * k + 64 * wv);   // wave-uniform; the DMA adds lane * 16 B
      if (tid + GT * k < NV)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  const float* pw = wp + ((long long)nb * nchunk_all * STEPS) * 64;

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  f32x4 ra[NLOAD], rb[NLOAD];
  float wa[STEPS], wb[STEPS];
  if (PCFA_SC5W_WLDS) dma_patch(0, 0);
  else load_patch(0, ra);
  auto load_w = [&](int chunk_global, float (&w)[STEPS]) {
    const f32x4* q4 = reinterpret_cast<const f32x4*>(pw + (long long)chunk_global * STEPS * 64);
#pragma unroll
    for (int sq = 0; sq < STEPS / 4; ++sq) {
      const f32x4 t = q4[sq * 64 + lane];
      w[4 * sq] = t.x;
      w[4 * sq + 1] = t.y;
      w[4 * sq + 2] = t.z;
      w[4 * sq + 3] = t.w;
    }
  };
  // chunk `chunk_global`'s U operands of this wave's channel block -> LDS slot `slot`: this wave's half (three 1-KB pieces)
  auto dma_w = [&](int chunk_global, int slot) {
    const float* src = pw + (long long)chunk_global * WBLK + (3 * wpx * 64 + lane) * 4;
    float* dst = wlds + (slot * WN + wn) * WBLK + 3 * wpx * 256;   // wave-uniform: the DMA adds lane * 16 B itself
#pragma unroll
    for (int k = 0; k < 3; ++k)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 256),
                                       (__attribute__((address_space(3))) void*)(dst + k * 256), 16, 0, 0);
  };
  auto read_w = [&](int slot, int sq, float (&w)[STEPS]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(wlds + (slot * WN + wn) * WBLK + (sq * 64 + lane) * 4);
    w[4 * sq] = t.x;
    w[4 * sq + 1] = t.y;
    w[4 * sq + 2] = t.z;
    w[4 * sq + 3] = t.w;
  };
  if (PCFA_SC5W_WLDS) {
    dma_w(grp, 0);
    dma_w(KS * min(1, nchunk - 1) + grp, 1);
  } else {
    load_w(grp, wa);
  }

  // ---- this lane's two output pixels and the rows of the accumulator tile its group finalises ----------------------
  const int mw = 32 * nb;                       // first output channel of the wave
  const bool rowB = !VERT || 2 * ty + 1 < H;    // 5x1: the pair's second row exists
  const long long pix = VERT ? (long long)(2 * ty) * W + x0 + 32 * wpx + l31 : (long long)ty * W + x0 + 64 * wpx + 2 * l31;
  const long long pixB = VERT ? (rowB ? W : 0) : 1;
  const int RB = grp * RN;
  auto ld2 = [&](const float* p, long long idx) -> f32x2 {
    if (!VERT) return *reinterpret_cast<const f32x2*>(p + idx);
    f32x2 v = {p[idx], p[idx + pixB]};
    return v;
  };
  auto st2 = [&](float* p, long long idx, float a, float b) {
    if (!VERT) {
      f32x2 v = {a, b};
      *reinterpret_cast<f32x2*>(p + idx) = v;
    } else {
      p[idx] = a;
      if (rowB) p[idx + W] = b;
    }
  };
  auto ml_of = [&](int rr) { const int r = RB + rr; return (r & 3) + 8 * (r >> 2) + 4 * lh; };

  // Operands of the fused GRU epilogue, requested before the K loop (see sepconv5.hip).
  const bool apart = mw < epi.C;   // modes 1: z half; modes 3 / 4: the a-part (GRU epilogue) -- wave-uniform
  f32x2 e0[RN], e1[RN], e2[RN], e3[RN], e4[RN];
#pragma unroll
  for (int rr = 0; rr < RN; ++rr) e0[rr] = e1[rr] = e2[rr] = e3[rr] = e4[rr] = (f32x2){0.f, 0.f};
  if (MODE == 1) {
    const long long ia0 = ((long long)blockIdx.z * 2 * epi.C + mw) * plane + pix;
    const long long ic0 = ((long long)blockIdx.z * epi.C + (apart ? mw : mw - epi.C)) * plane + pix;
#pragma unroll
    for (int rr = 0; rr < RN; ++rr) {
      e0[rr] = ld2(epi.p0, ia0 + ml_of(rr) * plane);
      e1[rr] = ld2(epi.p1, ic0 + ml_of(rr) * plane);   // h: used by the r half only (branch-free: valid for both)
    }
  } else if (MODE == 2) {
    const long long ic0 = ((long long)blockIdx.z * epi.C + mw) * plane + pix;
#pragma unroll
    for (int rr = 0; rr < RN; ++rr) {
      e0[rr] = ld2(epi.p0, ic0 + ml_of(rr) * plane);
      e1[rr] = ld2(epi.p1, ic0 + ml_of(rr) * plane);
      e2[rr] = ld2(epi.p2, ic0 + ml_of(rr) * plane);
    }
  } else if (MODE >= 3) {
    if (apart) {
      const long long ic0 = ((long long)blockIdx.z * epi.C + mw) * plane + pix;
#pragma unroll
      for (int rr = 0; rr < RN; ++rr) {
        e0[rr] = ld2(epi.p0, ic0 + ml_of(rr) * plane);
        e1[rr] = ld2(epi.p1, ic0 + ml_of(rr) * plane);
        e2[rr] = ld2(epi.p2, ic0 + ml_of(rr) * plane);
        e3[rr] = ld2(epi.p3, ic0 + ml_of(rr) * plane);
        if (MODE == 3 && epi.p4 != nullptr) e4[rr] = ld2(epi.p4, ic0 + ml_of(rr) * plane);
      }
    }
  }

  if (PCFA_SC5W_WLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed (see `item`)
  else store_patch(0, ra);
  __syncthreads();
  if (PCFA_SC5W_WLDS) {
#pragma unroll
    for (int sq = 0; sq < STEPS / 4; ++sq) read_w(0, sq, wa);
    __syncthreads();      // every wave holds chunk 0's operands: slot 0 may be refilled
  } else {
    load_patch(min(1, nchunk - 1), rb);
  }

  // 1x5: pair j = 32 wpx + l31 of the tile -> patch columns 2 j + 2 .. 2 j + 7, channel 2 p + lh
  // 5x1: x = x0 + 32 wpx + l31, the pair's six rows 64 floats apart
  const int bl = VERT ? lh * VCHS + 32 * wpx + l31 : lh * HCHS + 2 * (32 * wpx + l31) + 2;
  constexpr int CHS = VERT ? VCHS : HCHS;

  auto item = [&](int chunk, const float (&wcur)[STEPS], float (&wnext)[STEPS], f32x4 (&rload)[NLOAD],
                  const f32x4 (&rstore)[NLOAD]) {
    const float* sp = smem + (chunk & 1) * PATCH + bl;
    // slot chunk & 1 held THIS chunk's operands (every wave copied them to registers before the last barrier) and patch
    // buffer (chunk + 1) & 1 was last read in the previous item: both are refilled now and have this whole item to land
    if (PCFA_SC5W_WLDS) {
      if (!(PCFA_SC5W_DBG & 8)) dma_w(KS * min(chunk + 2, nchunk - 1) + grp, chunk & 1);
      if (!(PCFA_SC5W_DBG & 16)) dma_patch(min(chunk + 1, nchunk - 1), (chunk + 1) & 1);
    } else {
      if (!(PCFA_SC5W_DBG & 16)) load_patch(min(chunk + 2, nchunk - 1), rload);
      if (!(PCFA_SC5W_DBG & 8)) load_w(KS * min(chunk + 1, nchunk - 1) + grp, wnext);
    }
    if (PCFA_SC5W_DBG & 8) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) wnext[s] = wcur[s];
    }
    __builtin_amdgcn_sched_barrier(0);
    float d[2][6];
    auto rd = [&](int p, float (&dd)[6]) {
      if (PCFA_SC5W_DBG & 4) {
#pragma unroll
        for (int h = 0; h < 6; ++h) dd[h] = wcur[6 * p + h] + (float)h;
      } else if (VERT) {
#pragma unroll
        for (int h = 0; h < 6; ++h) dd[h] = sp[2 * p * CHS + 64 * h];
      } else {
#pragma unroll
        for (int h = 0; h < 3; ++h) {
          const f32x2 t = *reinterpret_cast<const f32x2*>(sp + 2 * p * CHS + 2 * h);
          dd[2 * h] = t.x;
          dd[2 * h + 1] = t.y;
        }
      }
    };
    rd(0, d[0]);
#pragma unroll
    for (int p = 0; p < CK / 2; ++p) {
      if (p + 1 < CK / 2) rd(p + 1, d[(p + 1) & 1]);
      const float d0 = d[p & 1][0], d1 = d[p & 1][1], d2 = d[p & 1][2], d3 = d[p & 1][3], d4 = d[p & 1][4], d5 = d[p & 1][5];
      float v0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
      const float a1 = fmaf(-4.f, d2, d4), b1 = fmaf(-4.f, d1, d3);
      float v1 = a1 + b1, v2 = a1 - b1;
      const float a2 = d4 - d2, b2 = 2.f * (d3 - d1);
      float v3 = a2 + b2, v4 = a2 - b2;
      float v5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
      if (PCFA_SC5W_DBG & 2) { v0 = d0; v1 = d1; v2 = d2; v3 = d3; v4 = d4; v5 = d5; }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 0], v0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 1], v1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 2], v2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 3], v3, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 4], v4, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 5], v5, acc[5], 0, 0, 0);
      if (PCFA_SC5W_WLDS && !(PCFA_SC5W_DBG & 8)) {   // next chunk's operands (landed before the last barrier): 6 reads over 4 pairs
        if (p < 2) {
          read_w((chunk + 1) & 1, 2 * p, wnext);
          read_w((chunk + 1) & 1, 2 * p + 1, wnext);
        } else {
          read_w((chunk + 1) & 1, p + 2, wnext);
        }
      }
      // the next chunk's patch goes to the other LDS buffer in the shadow of this chunk's MFMAs (nobody reads that buffer
      // since the previous barrier) instead of between the last MFMA and the barrier, where the slowest wave's wait for
      // its load stalled all eight
      if (!PCFA_SC5W_WLDS && p == PCFA_SC5W_STORE_AT && !(PCFA_SC5W_DBG & 16)) store_patch((chunk + 1) & 1, rstore);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!PCFA_SC5W_WLDS && PCFA_SC5W_STORE_AT >= CK / 2 && !(PCFA_SC5W_DBG & 16)) store_patch((chunk + 1) & 1, rstore);
    // The other waves read this wave's DMA pieces after the barrier: the pieces must have LANDED before this wave arrives
    // (the compiler's own waits only protect the issuing wave's reads: without this one, a partner occasionally multiplied
    // with the slot's previous contents -- run-to-run differences in test_gru_step_vs_oracle, batch 2)
    if (PCFA_SC5W_WLDS) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!(PCFA_SC5W_DBG & 1)) __syncthreads();
  };
  for (int chunk = 0; chunk < nchunk; chunk += 2) {   // host: nchunk even
    item(chunk, wa, wb, ra, rb);
    item(chunk + 1, wb, wa, rb, ra);
  }

  // ---- output transform: y(2j) = M0 + M1 + M2 + M3 + M4,  y(2j+1) = M1 - M2 + 2 (M3 - M4) + M5 ---------------------
  // (the last item's barrier has passed: the patches are dead and the partial outputs take their place)
  float* sred = smem_all;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r], s34 = acc[3][r] + acc[4][r],
                d34 = acc[3][r] - acc[4][r];
    sred[((grp * WPG + wv) * 32 + r) * 64 + lane] = acc[0][r] + s12 + s34;
    sred[((grp * WPG + wv) * 32 + 16 + r) * 64 + lane] = fmaf(2.f, d34, d12) + acc[5][r];
  }
  __syncthreads();
  float ya[RN], yb[RN];
#pragma unroll
  for (int rr = 0; rr < RN; ++rr) {
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int g = 0; g < KS; ++g) {   // group order: deterministic
      sa += sred[((g * WPG + wv) * 32 + RB + rr) * 64 + lane];
      sb += sred[((g * WPG + wv) * 32 + 16 + RB + rr) * 64 + lane];
    }
    ya[rr] = sa;
    yb[rr] = sb;
  }

  // ---- epilogues: the arithmetic of sepconv5.hip, on two pixels per accumulator row --------------------------------
  if (MODE == 1) {
    const long long io0 = ((long long)blockIdx.z * epi.C + (apart ? mw : mw - epi.C)) * plane + pix;
#pragma unroll
    for (int rr = 0; rr < RN; ++rr) {
      const long long io = io0 + ml_of(rr) * plane;
      const float sa = wino_sigmoid(ya[rr] + e0[rr].x), sb = wino_sigmoid(yb[rr] + e0[rr].y);
      if (apart) {
        st2(epi.o0, io, sa, sb);
      } else {
        st2(epi.o1, io, sa, sb);
        st2(epi.o2, io, sa * e1[rr].x, sb * e1[rr].y);
      }
    }
  } else if (MODE == 2) {
    const long long i0 = ((long long)blockIdx.z * epi.C + mw) * plane + pix;
#pragma unroll
    for (int rr = 0; rr < RN; ++rr) {
      const long long i = i0 + ml_of(rr) * plane;
      const float qa = tanhf(ya[rr] + e0[rr].x), qb = tanhf(yb[rr] + e0[rr].y);
      st2(epi.o0, i, qa, qb);
      st2(epi.o1, i, (1.f - e2[rr].x) * e1[rr].x + e2[rr].x * qa, (1.f - e2[rr].y) * e1[rr].y + e2[rr].y * qb);
    }
  } else if (MODE == 3 && apart) {   // as gru_gates_bwd_kernel (gru_math.hip), drh = the convolution's output
    const long long i0 = ((long long)blockIdx.z * epi.C + mw) * plane + pix;
    const long long j0 = ((long long)blockIdx.z * 2 * epi.C + mw) * plane + pix;
#pragma unroll
    for (int rr = 0; rr < RN; ++rr) {
      const int ml = ml_of(rr);
      float oa[2], ob[2], oc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float z_ = q ? e0[rr].y : e0[rr].x, r_ = q ? e1[rr].y : e1[rr].x, h_ = q ? e2[rr].y : e2[rr].x,
                    dz_ = q ? e3[rr].y : e3[rr].x, drh_ = q ? yb[rr] : ya[rr];
        oa[q] = dz_ * (1.f - z_) * z_;
        const float dr = drh_ * h_;
        ob[q] = dr * (1.f - r_) * r_;
        float c = drh_ * r_;
        if (epi.p4 != nullptr) c += q ? e4[rr].y : e4[rr].x;
        oc[q] = c;
      }
      st2(epi.o0, j0 + ml * plane, oa[0], oa[1]);
      st2(epi.o1, j0 + (long long)(epi.C + ml) * plane, ob[0], ob[1]);
      st2(epi.o2, i0 + ml * plane, oc[0], oc[1]);
    }
  } else if (MODE == 4 && apart) {   // dh accumulate, then gru_update_bwd_kernel of the previous half-step
    const long long i0 = ((long long)blockIdx.z * epi.C + mw) * plane + pix;
#pragma unroll
    for (int rr = 0; rr < RN; ++rr) {
      const long long i = i0 + ml_of(rr) * plane;
      float oa[2], ob[2], oc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float g_ = (q ? e0[rr].y : e0[rr].x) + (q ? yb[rr] : ya[rr]);
        const float z_ = q ? e1[rr].y : e1[rr].x, q_ = q ? e2[rr].y : e2[rr].x, h_ = q ? e3[rr].y : e3[rr].x;
        oa[q] = g_ * q_ - g_ * h_;
        ob[q] = (g_ * z_) * (1.f - q_ * q_);
        oc[q] = g_ * (1.f - z_);
      }
      st2(epi.o0, i, oa[0], oa[1]);
      st2(epi.o1, i, ob[0], ob[1]);
      st2(epi.o2, i, oc[0], oc[1]);
    }
  } else {
    // plain output / the b-part of modes 3 and 4 (host: the split sits on a multiple of 32, so the wave's channels lie
    // on one side of it and destination, accumulate flag and mask are wave-uniform)
    const bool first = out.b == nullptr || mw < out.Ca;
    float* base = first ? out.a + (long long)mw * plane : out.b + (long long)(mw - out.Ca) * plane;
    const bool accum = first ? out.acc_a != 0 : out.acc_b != 0;
    const bool masked = !first && out.mask_b != nullptr;
    const float* mk = masked ? out.mask_b + (long long)(mw - out.Ca) * plane : base;
#pragma unroll
    for (int rr = 0; rr < RN; ++rr) {
      const int ml = ml_of(rr);
      const long long i = ml * plane + pix;
      float va = ya[rr], vb = yb[rr];
      if (accum) {
        const f32x2 old = ld2(base, i);
        va += old.x;
        vb += old.y;
      }
      if (masked && mw - out.Ca + ml < out.mask_cb) {
        const f32x2 m = ld2(mk, i);
        va = m.x > 0.f ? va : 0.f;
        vb = m.y > 0.f ? vb : 0.f;
      }
      st2(base, i, va, vb);
    }
  }
}

bool wino_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool VERT, int WN, int KS, int MODE>
int wino_run(const Operand& in, const float* w, const OutSplit& out, int B, int Cout, int H, int W, hipStream_t s,
             const GruEpi& epi) {
  constexpr int PATCH = VERT ? VPATCH : HPATCH, WPG = 2 * WN;
  constexpr int SRED = KS * WPG * 32 * 64;
  constexpr int STAGE = 2 * KS * PATCH + (PCFA_SC5W_WLDS ? KS * 2 * WN * STEPS * 64 : 0);   // patches + U slots
  constexpr int LDSF = STAGE > SRED ? STAGE : SRED;
  static bool attr_set = false;   // > 64 KB of dynamic LDS needs the opt-in once per kernel
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)sc5_wino_kernel<VERT, WN, KS, MODE>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDSF * (int)sizeof(float));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tiles_x = W / (VERT ? VPX : HPX);
  dim3 grid((unsigned)(tiles_x * (VERT ? (H + 1) / 2 : H)), (unsigned)(Cout / (32 * WN)), (unsigned)B), block(128 * WN * KS);
  pcfa_launch(sc5_wino_kernel<VERT, WN, KS, MODE>, grid, block, LDSF * sizeof(float), s, in, w, out, Cout, H, W, tiles_x,
              epi);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

template <bool VERT, int MODE>
int wino_pick(bool wide, const Operand& in, const float* w, const OutSplit& out, int B, int Cout, int H, int W,
              hipStream_t s, const GruEpi& epi) {
  if (wide) return wino_run<VERT, 2, 2, MODE>(in, w, out, B, Cout, H, W, s, epi);
  return wino_run<VERT, 1, 4, MODE>(in, w, out, B, Cout, H, W, s, epi);
}

}  // namespace

namespace pcfa_sc5 {

// set < 0: query; 0 / 1: switch the Winograd path off / on (process-wide; PCFA_SEPCONV_WINO=0 sets the initial state)
int sc5_wino_enabled(int set) {
  static int enabled = getenv("PCFA_SEPCONV_WINO") ? atoi(getenv("PCFA_SEPCONV_WINO")) : 1;
  const int prev = enabled;
  if (set >= 0) enabled = set != 0;
  return prev;
}

long long sc5_wino_packed_floats(int Cout, int Cin) {
  if (Cout < 1 || Cin < 1 || Cin % CK != 0) return 0;
  return (long long)((Cout + 31) / 32) * (Cin / CK) * STEPS * 64;
}

int sc5_wino_pack(const float* w, float* packed, int N, int C, int transpose, hipStream_t stream) {
  const long long total = sc5_wino_packed_floats(N, C);
  if (total == 0) return PCFA_OK;
  pcfa_launch(sc5_wino_pack_kernel, dim3(1024), dim3(256), 0, stream, w, packed, N, C, transpose, total);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

// tile shape by output width: 64 channels x 2 K-groups, or 32 channels x 4 K-groups when the grid would not fill the chip
static bool wino_wide(int B, int Cout, int H, int W, int vertical) {
  const long long tiles = (long long)(vertical ? (W / VPX) * ((H + 1) / 2) : (W / HPX) * H) * B;
  return Cout % 64 == 0 && tiles * (Cout / 64) >= 200;
}

// the shape part of the eligibility rule (pointer alignment and the output split are checked per launch)
bool sc5_wino_shape_ok(int B, int Cin, int Ca, int Cout, int H, int W, int vertical) {
  if (!sc5_wino_enabled(-1) || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return false;
  const int ks = wino_wide(B, Cout, H, W, vertical) ? 2 : 4;
  if (Cout % 32 != 0 || Cin % (2 * CK * ks) != 0 || Ca % CK != 0) return false;
  if (W % (vertical ? VPX : HPX) != 0 || B > 65535) return false;
  return (long long)H * W <= (1LL << 30);   // 32-bit offsets inside a channel plane
}

int sc5_wino_launch(const Operand& in, const float* w_wino, const OutSplit& out, int B, int Cout, int H, int W, int vertical,
                    hipStream_t stream, const GruEpi& epi) {
  if (!w_wino || !sc5_wino_shape_ok(B, in.Cin, in.Ca, Cout, H, W, vertical)) return PCFA_SC5_NOT_ELIGIBLE;
  const bool wide = wino_wide(B, Cout, H, W, vertical);
  if (!wino_aligned16(in.a) || (in.b && !wino_aligned16(in.b)) || !wino_aligned16(out.a) || (out.b && !wino_aligned16(out.b)))
    return PCFA_SC5_NOT_ELIGIBLE;
  if (out.b != nullptr && out.Ca % 32 != 0) return PCFA_SC5_NOT_ELIGIBLE;
  if (epi.mode != 0 && (epi.C % 32 != 0)) return PCFA_SC5_NOT_ELIGIBLE;
#define PCFA_WINO(V, M) return wino_pick<V, M>(wide, in, w_wino, out, B, Cout, H, W, stream, epi)
  if (vertical) {
    switch (epi.mode) {
      case 0: PCFA_WINO(true, 0);
      case 1: PCFA_WINO(true, 1);
      case 2: PCFA_WINO(true, 2);
      case 3: PCFA_WINO(true, 3);
      case 4: PCFA_WINO(true, 4);
    }
  } else {
    switch (epi.mode) {
      case 0: PCFA_WINO(false, 0);
      case 1: PCFA_WINO(false, 1);
      case 2: PCFA_WINO(false, 2);
      case 3: PCFA_WINO(false, 3);
      case 4: PCFA_WINO(false, 4);
    }
  }
#undef PCFA_WINO
  return PCFA_SC5_NOT_ELIGIBLE;
}

}  // namespace pcfa_sc5
