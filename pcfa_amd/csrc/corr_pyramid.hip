// RAFT/GMA all-pairs correlation pyramid (build + backward) for gfx950.
//
// Replaces CorrBlock.corr + the avg_pool2d pyramid (reference
// models/raft/corr.py:12-27,52-60) and their autograd backward.
//
// MI355X formulation: average pooling commutes with the inner product, so
//   avg_pool_l(fmap1^T fmap2) == fmap1^T avg_pool_l(fmap2).
// The whole pyramid is therefore ONE fp32-MFMA GEMM
//   pyr[Q x slab] = fmap1^T [Q x D] . f2ext [D x slab] / sqrt(D)
// against fmap2 extended by its pooled copies (f2ext), written straight into
// the slab-per-query layout the lookup kernels read -- no pooling passes over
// the 260 MB volume.  The backward is two split-K GEMMs against the same f2ext
// followed by the (tiny) adjoint of the pooling on a [D x slab] matrix.
//
// GEMM core: 128x128x16 block tile, 4 waves (2x2), each wave 2x2 tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain), LDS double buffer
// fed by register prefetch.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef PCFA_GEMM_BK
#define PCFA_GEMM_BK 16
#endif
#ifndef PCFA_GEMM_STORE_MID
#define PCFA_GEMM_STORE_MID 0          // tools/dev A/B: write the next stage to LDS after MFMA pair PCFA_GEMM_STORE_MID_AT
#endif
#ifndef PCFA_GEMM_STORE_MID_AT
#define PCFA_GEMM_STORE_MID_AT 4
#endif
constexpr int BM = 128, BN = 128, BK = PCFA_GEMM_BK;
constexpr int NLD = BK / 8;   // float4 per thread and operand tile (128 x BK floats over 256 threads)
constexpr int KV = BK / 4;    // float4 per row of a k-contiguous operand tile
constexpr int LDS_KM = 132;  // row stride (floats) when the operand arrives k-major (b128 writes)
constexpr int LDS_MK = 130;  // row stride when transposing on the way in (conflict-free b32 writes)

// Stage one 128(dim) x 16(k) operand tile: global -> registers.
// KMAJ: stored [K][dim] (dim contiguous) ; else stored [dim][K] (k contiguous).
template <bool KMAJ>
__device__ __forceinline__ void tile_load(const float* __restrict__ X, long long ld, int dim,
                                          int d0, int k0, int kend, bool vec, float4 (&r)[NLD]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + 256 * i;
    int k, d;
    if (KMAJ) {
      k = k0 + f / 32;
      d = d0 + (f % 32) * 4;
    } else {
      d = d0 + f / KV;
      k = k0 + (f % KV) * 4;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KMAJ) {
      if (k < kend) {
        const float* p = X + (long long)k * ld + d;
        if (vec && d + 3 < dim) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (d + 0 < dim) v.x = p[0];
          if (d + 1 < dim) v.y = p[1];
          if (d + 2 < dim) v.z = p[2];
          if (d + 3 < dim) v.w = p[3];
        }
      }
    } else {
      if (d < dim) {
        const float* p = X + (long long)d * ld + k;
        if (vec && k + 3 < kend) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (k + 0 < kend) v.x = p[0];
          if (k + 1 < kend) v.y = p[1];
          if (k + 2 < kend) v.z = p[2];
          if (k + 3 < kend) v.w = p[3];
        }
      }
    }
    r[i] = v;
  }
}

// FAST staging (both dims % 4 == 0, 16-B aligned operands, K range % 4 == 0): every load is an unconditional
// float4 from a clamped, always valid address -- no branch between a load and its use, so hipcc keeps the loads
// of stage k+1 in flight across the MFMAs of stage k instead of draining them first (with the predicated loader
// the loop waits with vmcnt(0) BEFORE the MFMA block); out-of-range pieces are zeroed when the registers are
// written to LDS.
template <bool KMAJ>
__device__ __forceinline__ void tile_load_fast(const float* __restrict__ X, long long ld, int dim, int d0, int k0,
                                               int kend, float4 (&r)[NLD]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + 256 * i;
    int k, d;
    if (KMAJ) {
      k = min(k0 + f / 32, kend - 1);
      d = min(d0 + (f % 32) * 4, dim - 4);
      r[i] = *reinterpret_cast<const float4*>(X + (long long)k * ld + d);
    } else {
      d = min(d0 + f / KV, dim - 1);
      k = min(k0 + (f % KV) * 4, kend - 4);
      r[i] = *reinterpret_cast<const float4*>(X + (long long)d * ld + k);
    }
  }
}

template <bool KMAJ>
__device__ __forceinline__ void tile_mask(int dim, int d0, int k0, int kend, float4 (&r)[NLD]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + 256 * i;
    const bool ok = KMAJ ? (k0 + f / 32 < kend && d0 + (f % 32) * 4 < dim)
                         : (d0 + f / KV < dim && k0 + (f % KV) * 4 < kend);
    if (!ok) r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <bool KMAJ>
__device__ __forceinline__ void tile_store(float* __restrict__ s, const float4 (&r)[NLD]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + 256 * i;
    if (KMAJ) {
      const int k = f / 32, d = (f % 32) * 4;
      *reinterpret_cast<float4*>(s + k * LDS_KM + d) = r[i];
    } else {
      const int d = f / KV, k = (f % KV) * 4;
      s[(k + 0) * LDS_MK + d] = r[i].x;
      s[(k + 1) * LDS_MK + d] = r[i].y;
      s[(k + 2) * LDS_MK + d] = r[i].z;
      s[(k + 3) * LDS_MK + d] = r[i].w;
    }
  }
}

// C[m][n] = (sum_k A(m,k) B(k,n)) / div   over k in this block's split.
// grid = (ceil(N/128), ceil(M/128), batch*splits).
// POOL (correlation pyramid forward only): the GEMM runs over the level-0 columns [0, S0) and the tail columns
// [off_tail, slab) (levels >= 3 and the zero tile, taken from f2ext as before); levels 1 and 2 are average-pooled from
// the level-0 accumulators in the epilogue, the way the reference pools the correlation volume
// (models/raft/corr.py:24-27: F.avg_pool2d of the level below) -- 24 % fewer MFMAs than multiplying against the
// pooled copies of fmap2, and the pooled values round like the reference's.  Needs W % 16 == 0 (no x padding inside
// the tiles of levels 1 and 2) and L >= 3; the caller checks.
struct PoolArgs {
  int nb0;        // N-blocks (of BN columns) covering the level-0 columns
  int S0;         // level-0 columns (tiles x 16)
  int off_tail;   // first tail column
  int tw0, tiles0;
  int off1, tw1, h1, w1;
  int off2, tw2, h2, w2;
};
constexpr int SC = BN + 4;   // row stride of the epilogue image of the C tile

// UNPOOL (correlation pyramid backward only, 4 levels, W % 16 == 0): the transposed form of POOL.  The gradient of the
// reference's volume is g0 + up(g1 + up(g2 + up(g3)/4)/4)/4 (autograd of three F.avg_pool2d, models/raft/corr.py:24-27)
// and both backward products only need THAT [Q x level-0] matrix: it is formed piece by piece while the dpyr operand
// is staged -- a 16-B piece of a 4x4 tile plus the 8 B, 4 B and 4 B of its level-1/2/3 parents -- so the GEMMs run
// over the level-0 columns only (25 % fewer MFMAs than multiplying the pooled columns of f2ext) and df2ext's adjoint
// pooling pass disappears.
struct UnpoolArgs {
  int tw0;
  int off1, tw1, h1, w1;
  int off2, tw2, h2, w2;
  int off3, tw3, h3, w3;
};

// tile (ty, tx), tile row y of one query's slab `row`; v = the level-0 piece
__device__ __forceinline__ float4 unpool_piece(const float* __restrict__ row, float4 v, int ty, int tx, int y,
                                               const UnpoolArgs& u) {
  const int y3 = ty >> 1, x3 = tx >> 1;
  const bool ok2 = ty < u.h2 && tx < u.w2, ok3 = ok2 && y3 < u.h3 && x3 < u.w3;
  // clamped addresses, values masked: no branch around a load
  const float g3 = row[ok3 ? u.off3 + (((y3 >> 2) * u.tw3 + (x3 >> 2)) << 4) + ((y3 & 3) << 2) + (x3 & 3) : 0];
  const float g2 = row[ok2 ? u.off2 + (((ty >> 2) * u.tw2 + (tx >> 2)) << 4) + ((ty & 3) << 2) + (tx & 3) : 0];
  const float d2 = ok2 ? g2 + 0.25f * (ok3 ? g3 : 0.f) : 0.f;
  const int Y1 = 2 * ty + (y >> 1), X1 = 2 * tx;
  const bool okY = Y1 < u.h1;
  const float2 g1 = *reinterpret_cast<const float2*>(
      row + (okY ? u.off1 + (((Y1 >> 2) * u.tw1 + (X1 >> 2)) << 4) + ((Y1 & 3) << 2) + (X1 & 3) : 0));
  const float d1a = (okY && X1 < u.w1) ? g1.x + 0.25f * d2 : 0.f;
  const float d1b = (okY && X1 + 1 < u.w1) ? g1.y + 0.25f * d2 : 0.f;
  v.x += 0.25f * d1a;
  v.y += 0.25f * d1a;
  v.z += 0.25f * d1b;
  v.w += 0.25f * d1b;
  return v;
}

// tile_load_fast of the dpyr operand with the un-pooling applied; `dim` / `kend` bound the LEVEL-0 columns.
template <bool KMAJ>
__device__ __forceinline__ void tile_load_unpool(const float* __restrict__ X, long long ld, int dim, int d0, int k0,
                                                 int kend, const UnpoolArgs& u, float4 (&r)[NLD]) {
  static_assert(BK % 16 == 0, "whole 4x4 tiles per stage and query");
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = tid + 256 * i;
    int q, col;   // query row of dpyr, level-0 slab column of the piece
    if (KMAJ) {   // stored [K = query][N = column]
      q = min(k0 + f / 32, kend - 1);
      col = min(d0 + (f % 32) * 4, dim - 4);
    } else {      // stored [N = query][K = column]
      q = min(d0 + f / KV, dim - 1);
      col = min(k0 + (f % KV) * 4, kend - 4);
    }
    const int T0 = col >> 4, y = (col & 15) >> 2;
    const int ty = T0 / u.tw0, tx = T0 - ty * u.tw0;
    const float* row = X + (long long)q * ld;
    r[i] = unpool_piece(row, *reinterpret_cast<const float4*>(row + col), ty, tx, y, u);
  }
}

// SPARSE: the K range of a column block is a list of up to four segments (multiples of BK, ascending, read from
// segs[(batch * gridDim.x + column block) * SEG_INTS]) instead of [0, K): the backward products of the correlation
// pyramid skip the part of dpyr that no lookup window ever touched (corr_window_segments_kernel below).  The splits
// share the ACTIVE steps of a block evenly.
constexpr int SEG_INTS = 10;   // { count, (begin, end) x 4, pad }
template <bool A_KM, bool B_KN, bool FAST, bool POOL, bool UNPOOL = false, bool SPARSE = false>
__device__ __forceinline__ void gemm_f32_mfma_body(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
    int N, int K, long long lda, long long ldb, long long ldc, long long bsA, long long bsB,
    long long bsC, int splits, int kchunk, long long ssC, float div, int vecA, int vecB, const PoolArgs& pool,
    const UnpoolArgs& unpool = UnpoolArgs{}, const int* __restrict__ segs = nullptr) {
  static_assert(!UNPOOL || FAST, "the un-pooling loader is the branch-free one");
  static_assert(!SPARSE || (FAST && !POOL && !UNPOOL), "segment lists ride on the branch-free loader");
  constexpr int SA = A_KM ? LDS_KM : LDS_MK;
  constexpr int SB = B_KN ? LDS_KM : LDS_MK;
  constexpr int STAGE = 2 * BK * SA + 2 * BK * SB;
  constexpr int LDSF = POOL && BM * SC > STAGE ? BM * SC : STAGE;   // the C image reuses the stage buffers
  __shared__ __attribute__((aligned(16))) float smem[LDSF];
  float (*sA)[BK * SA] = reinterpret_cast<float (*)[BK * SA]>(smem);
  float (*sB)[BK * SB] = reinterpret_cast<float (*)[BK * SB]>(smem + 2 * BK * SA);

  const int batch = blockIdx.z / splits;
  const int split = blockIdx.z - batch * splits;
  A += batch * bsA;
  B += batch * bsB;
  C += batch * bsC + split * ssC;
  int kbeg = split * kchunk;
  const int kend = SPARSE ? K : min(K, kbeg + kchunk);
  // XCD-aware tile order (cdna_hip_programming.md T1, bijective form): workgroups are dealt round-robin over the 8
  // XCDs, so consecutive linear ids land on different L2s; remapped, every XCD walks a contiguous band of tile rows
  // and re-reads its A band / the streamed B tiles from its own L2.  Speed only -- any placement is correct.
  int by = blockIdx.y, bx = blockIdx.x;
#ifndef PCFA_GEMM_NO_XCD
  {
    const int nx = gridDim.x, nwg = nx * gridDim.y;
    const int orig = by * nx + bx, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    by = wg / nx;
    bx = wg - by * nx;
  }
#endif
  const int m0 = by * BM;
  int n0 = bx * BN;
  bool pooled = false;
  if (POOL) {   // workgroup-uniform: a level-0 block (pooled epilogue) or a tail block (columns shifted to off_tail)
    if (bx < pool.nb0) {
      pooled = true;
      N = pool.S0;
    } else {
      n0 = pool.off_tail + (bx - pool.nb0) * BN;
    }
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NLD], rb[NLD];
  int nk = (kend - kbeg + BK - 1) / BK;
  int sb[4] = {0, 0, 0, 0}, se[4] = {0, 0, 0, 0}, seg = 0;
  if (SPARSE) {
    const int* sg = segs + ((long long)batch * gridDim.x + bx) * SEG_INTS;
    int total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sb[i] = sg[1 + 2 * i];
      se[i] = sg[2 + 2 * i];
      total += (se[i] - sb[i]) / BK;
    }
    const int t0 = (int)((long long)total * split / splits), t1 = (int)((long long)total * (split + 1) / splits);
    nk = t1 - t0;
    int rem = t0;
    while (seg < 3 && rem >= (se[seg] - sb[seg]) / BK) {
      rem -= (se[seg] - sb[seg]) / BK;
      ++seg;
    }
    kbeg = sb[seg] + rem * BK;   // first stage of this split
  }
  int kcur = kbeg;
  if (nk > 0) {
    if (FAST) {
      tile_load_fast<A_KM>(A, lda, M, m0, kbeg, kend, ra);
      if (UNPOOL) tile_load_unpool<B_KN>(B, ldb, N, n0, kbeg, kend, unpool, rb);
      else tile_load_fast<B_KN>(B, ldb, N, n0, kbeg, kend, rb);
      tile_mask<A_KM>(M, m0, kbeg, kend, ra);
      tile_mask<B_KN>(N, n0, kbeg, kend, rb);
    } else {
      tile_load<A_KM>(A, lda, M, m0, kbeg, kend, vecA, ra);
      tile_load<B_KN>(B, ldb, N, n0, kbeg, kend, vecB, rb);
    }
    tile_store<A_KM>(sA[0], ra);
    tile_store<B_KN>(sB[0], rb);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    int k0 = kbeg + (kt + 1) * BK;
    if (SPARSE) {   // next stage: the following step of the segment, or the first step of the next non-empty segment
      k0 = kcur + BK;
      if (k0 >= se[seg] && seg < 3 && se[seg + 1] > sb[seg + 1]) {
        ++seg;
        k0 = sb[seg];
      }
      kcur = k0;
    }
    if (FAST) {  // unconditional: the stage past the end re-reads the last one (clamped) and is never stored
      tile_load_fast<A_KM>(A, lda, M, m0, min(k0, kend - 4), kend, ra);
      if (UNPOOL) tile_load_unpool<B_KN>(B, ldb, N, n0, min(k0, kend - 4), kend, unpool, rb);
      else tile_load_fast<B_KN>(B, ldb, N, n0, min(k0, kend - 4), kend, rb);
      __builtin_amdgcn_sched_barrier(0);  // keep the loads ABOVE the MFMA block (hipcc otherwise sinks them to their use)
    } else if (more) {
      tile_load<A_KM>(A, lda, M, m0, k0, kend, vecA, ra);
      tile_load<B_KN>(B, ldb, N, n0, k0, kend, vecB, rb);
    }
    const float* a = sA[cur] + wr * 64 + l31;
    const float* b = sB[cur] + wc * 64 + l31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = a[(kk + lh) * SA], a1 = a[(kk + lh) * SA + 32];
      const float b0 = b[(kk + lh) * SB], b1 = b[(kk + lh) * SB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
#if PCFA_GEMM_STORE_MID
      if (FAST && kk == PCFA_GEMM_STORE_MID_AT) {   // the next stage's LDS write in the shadow of this stage's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        tile_mask<A_KM>(M, m0, k0, kend, ra);
        tile_mask<B_KN>(N, n0, k0, kend, rb);
        tile_store<A_KM>(sA[cur ^ 1], ra);
        tile_store<B_KN>(sB[cur ^ 1], rb);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    if (FAST) {  // unconditional (also after the last stage, into the idle buffer): a store under `if (more)`
                 // lets hipcc sink the loads into that branch, i.e. below the MFMAs
#if !PCFA_GEMM_STORE_MID
      tile_mask<A_KM>(M, m0, k0, kend, ra);
      tile_mask<B_KN>(N, n0, k0, kend, rb);
      tile_store<A_KM>(sA[cur ^ 1], ra);
      tile_store<B_KN>(sB[cur ^ 1], rb);
#endif
    } else if (more) {
      tile_store<A_KM>(sA[cur ^ 1], ra);
      tile_store<B_KN>(sB[cur ^ 1], rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // Epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // The reference divides by sqrt(D); when that is a power of two (D = 256 -> 16) multiplying by its reciprocal
  // is the same fp32 result and saves a division sequence per output element.
  int dexp;
  const bool pow2 = frexpf(div, &dexp) == 0.5f;
  const float rdiv = 1.0f / div;
  if (POOL && pooled) {
    // ---- C tile -> LDS (the stage buffers are free: the loop ended on a barrier) ----
    float* sC = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          sC[row * SC + wc * 64 + j * 32 + l31] = pow2 ? acc[i][j][r] * rdiv : acc[i][j][r] / div;
        }
    __syncthreads();
    const int tid = threadIdx.x;
    // level 0: rows of 512 B, 16-B stores
#pragma unroll
    for (int k = 0; k < BM * BN / 4 / 256; ++k) {
      const int idx = tid + 256 * k, row = idx >> 5, c4 = (idx & 31) * 4;
      if (m0 + row < M && n0 + c4 < N)
        *reinterpret_cast<float4*>(&C[(long long)(m0 + row) * ldc + n0 + c4]) =
            *reinterpret_cast<const float4*>(&sC[row * SC + c4]);
    }
    // levels 1 and 2: thread = (query row, 4x4 tile); sums in avg_pool2d's window order, one division by 4 each
#pragma unroll
    for (int k = 0; k < BM * (BN / 16) / 256; ++k) {
      const int item = tid + 256 * k, row = item >> 3, t = item & 7;
      const int T0 = n0 / 16 + t;
      if (m0 + row >= M || T0 >= pool.tiles0) continue;
      const int ty = T0 / pool.tw0, tx = T0 - ty * pool.tw0;
      float v[4][4];
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const float4 q4 = *reinterpret_cast<const float4*>(&sC[row * SC + 16 * t + 4 * y]);
        v[y][0] = q4.x; v[y][1] = q4.y; v[y][2] = q4.z; v[y][3] = q4.w;
      }
      float* crow = C + (long long)(m0 + row) * ldc;
      float l1[2][2];
#pragma unroll
      for (int Y = 0; Y < 2; ++Y) {
#pragma unroll
        for (int X = 0; X < 2; ++X) {
          const float sum = ((v[2 * Y][2 * X] + v[2 * Y][2 * X + 1]) + v[2 * Y + 1][2 * X]) + v[2 * Y + 1][2 * X + 1];
          l1[Y][X] = (2 * ty + Y < pool.h1 && 2 * tx + X < pool.w1) ? sum * 0.25f : 0.f;
        }
        float* d1 = crow + pool.off1 + ((ty >> 1) * pool.tw1 + (tx >> 1)) * 16 + (2 * (ty & 1) + Y) * 4 + 2 * (tx & 1);
        *reinterpret_cast<float2*>(d1) = make_float2(l1[Y][0], l1[Y][1]);
      }
      const float sum2 = ((l1[0][0] + l1[0][1]) + l1[1][0]) + l1[1][1];
      crow[pool.off2 + ((ty >> 2) * pool.tw2 + (tx >> 2)) * 16 + (ty & 3) * 4 + (tx & 3)] =
          (ty < pool.h2 && tx < pool.w2) ? sum2 * 0.25f : 0.f;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + j * 32 + l31;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M) C[(long long)row * ldc + col] = pow2 ? acc[i][j][r] * rdiv : acc[i][j][r] / div;
      }
    }
}

template <bool B_KN>
__global__ __launch_bounds__(256) void gemm_f32_mfma_sparse_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
    int N, int K, long long lda, long long ldb, long long ldc, long long bsA, long long bsB,
    long long bsC, int splits, long long ssC, float div, const int* __restrict__ segs) {
  gemm_f32_mfma_body<false, B_KN, true, false, false, true>(A, B, C, M, N, K, lda, ldb, ldc, bsA, bsB, bsC, splits, 0,
                                                            ssC, div, 1, 1, PoolArgs{}, UnpoolArgs{}, segs);
}

template <bool A_KM, bool B_KN, bool FAST>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
    int N, int K, long long lda, long long ldb, long long ldc, long long bsA, long long bsB,
    long long bsC, int splits, int kchunk, long long ssC, float div, int vecA, int vecB) {
  gemm_f32_mfma_body<A_KM, B_KN, FAST, false>(A, B, C, M, N, K, lda, ldb, ldc, bsA, bsB, bsC, splits, kchunk, ssC, div,
                                              vecA, vecB, PoolArgs{});
}

// The correlation pyramid's forward product with levels 1-2 pooled in the epilogue (see PoolArgs).
__global__ __launch_bounds__(256) void corr_pyramid_pool_gemm_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
    int N, int K, long long lda, long long ldb, long long ldc, long long bsA, long long bsB,
    long long bsC, float div, PoolArgs pool) {
  gemm_f32_mfma_body<true, true, true, true>(A, B, C, M, N, K, lda, ldb, ldc, bsA, bsB, bsC, 1,
                                             ((K + BK - 1) / BK) * BK, 0LL, div, 1, 1, pool);
}

// The two backward products over the level-0 columns with the dpyr operand un-pooled on the way in (see UnpoolArgs).
template <bool B_KN>
__global__ __launch_bounds__(256) void corr_pyramid_unpool_gemm_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
    int N, int K, long long lda, long long ldb, long long ldc, long long bsA, long long bsB,
    long long bsC, int splits, int kchunk, long long ssC, float div, UnpoolArgs unpool) {
  gemm_f32_mfma_body<false, B_KN, true, false, true>(A, B, C, M, N, K, lda, ldb, ldc, bsA, bsB, bsC, splits, kchunk, ssC,
                                                     div, 1, 1, PoolArgs{}, unpool);
}

// dfmap2[b][d][y][x] = sum_s part[s][b][d][tile order of (y, x)]: the split-K reduction of the level-0 product, written
// back in image order (fixed order -> deterministic)
__global__ void splitk_reduce_untile_kernel(const float* __restrict__ part, float* __restrict__ out, int planes, int H,
                                            int W, int tw0, int S0, int splits, long long ss) {
  const long long n = (long long)planes * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long long pl = i / ((long long)W * H);
    const long long src = pl * S0 + ((((y >> 2) * tw0 + (x >> 2)) << 4) + ((y & 3) << 2) + (x & 3));
    float s = part[src];
    for (int k = 1; k < splits; ++k) s += part[(long long)k * ss + src];
    out[i] = s;
  }
}

// out[i] = sum_s partial[s][i]  (fixed order -> deterministic).  No scaling:
// the split GEMMs already divided every partial by sqrt(D).
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                     long long n, int splits, long long ss) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float s = part[i];
    for (int k = 1; k < splits; ++k) s += part[(long long)k * ss + i];
    out[i] = s;
  }
}

// One workgroup per (b, d) plane: f2ext row = the plane and its pooled copies, every level in the
// 4x4-tiled column order of PyrLayout (so the GEMM writes tiled slabs without knowing about tiles);
// tile-padding columns are 0.
__global__ __launch_bounds__(256) void f2ext_fwd_kernel(const float* __restrict__ fmap2,
                                                        float* __restrict__ f2ext, int Q,
                                                        PyrLayout P) {
  extern __shared__ float s_lv[];  // row-major copies of levels >= 1, packed back to back
  const float* src = fmap2 + (size_t)blockIdx.x * Q;
  float* dst = f2ext + (size_t)blockIdx.x * P.slab;
  for (int i = threadIdx.x; i < P.slab; i += blockDim.x) dst[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < Q; i += blockDim.x) {
    const int y = i / P.w[0], x = i - y * P.w[0];
    dst[pcfa_tiled_index(P, 0, y, x)] = src[i];
  }
  int lds_off = 0, prev_off = 0;
  for (int l = 1; l < P.L; ++l) {
    const int hl = P.h[l], wl = P.w[l], wp = P.w[l - 1];
    const float* prev = (l == 1) ? src : (s_lv + prev_off);
    float* cur = s_lv + lds_off;
    for (int i = threadIdx.x; i < hl * wl; i += blockDim.x) {
      const int y = i / wl, x = i - y * wl;
      const float* p = prev + (2 * y) * wp + 2 * x;
      // F.avg_pool2d: sequential sum over the window, then one division
      const float v = (((p[0] + p[1]) + p[wp]) + p[wp + 1]) / 4.0f;
      cur[i] = v;
      dst[pcfa_tiled_index(P, l, y, x)] = v;
    }
    prev_off = lds_off;
    lds_off += hl * wl;
    __syncthreads();
  }
}

// Adjoint of the pooling chain: dfmap2 plane = g_0 + up(g_1)/4 + up(up(g_2)/4)/4 ...
__global__ __launch_bounds__(256) void f2ext_bwd_kernel(const float* __restrict__ df2ext,
                                                        float* __restrict__ dfmap2, int Q,
                                                        PyrLayout P) {
  extern __shared__ float s_lv[];  // row-major gradients of levels >= 1
  const float* src = df2ext + (size_t)blockIdx.x * P.slab;
  float* dst = dfmap2 + (size_t)blockIdx.x * Q;
  int loff[PCFA_MAX_LEVELS];
  {
    int o = 0;
    for (int l = 1; l < P.L; ++l) {
      loff[l] = o;
      o += P.h[l] * P.w[l];
    }
  }
  for (int l = 1; l < P.L; ++l)
    for (int i = threadIdx.x; i < P.h[l] * P.w[l]; i += blockDim.x) {
      const int y = i / P.w[l], x = i - y * P.w[l];
      s_lv[loff[l] + i] = src[pcfa_tiled_index(P, l, y, x)];
    }
  __syncthreads();
  for (int l = P.L - 1; l >= 2; --l) {
    const int hl = P.h[l], wl = P.w[l], wp = P.w[l - 1];
    const float* g = s_lv + loff[l];
    float* gp = s_lv + loff[l - 1];
    for (int i = threadIdx.x; i < 4 * hl * wl; i += blockDim.x) {
      const int y = i / (2 * wl), x = i - y * (2 * wl);
      gp[y * wp + x] += g[(y >> 1) * wl + (x >> 1)] / 4.0f;
    }
    __syncthreads();
  }
  const int W0 = P.w[0];
  for (int i = threadIdx.x; i < Q; i += blockDim.x) {
    const int y = i / W0, x = i - y * W0;
    float v = src[pcfa_tiled_index(P, 0, y, x)];
    if (P.L > 1 && (y >> 1) < P.h[1] && (x >> 1) < P.w[1])
      v += s_lv[loff[1] + (y >> 1) * P.w[1] + (x >> 1)] / 4.0f;
    dst[i] = v;
  }
}

size_t pooled_lds_bytes(const PyrLayout& P) {
  size_t n = 4;
  for (int l = 1; l < P.L; ++l) n += (size_t)P.h[l] * P.w[l];
  return n * sizeof(float);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int BWD_SPLITS = 8;
constexpr int MAX_COORDS = 32;

struct CoordList {
  const float* p[MAX_COORDS];
  int n;
};

// Which part of dpyr can be non-zero?  Every lookup's backward adds into the (2r+2)^2 texels around
// (cx / 2^l, cy / 2^l) of its queries (corr_lookup.hip); with the coordinates of all lookups of this backward pass the
// rows a query row can have touched at level l are  floor(min cy / 2^l) - r - 1 .. floor(max cy / 2^l) + r + 2  (one
// texel of slack on either side).  In the 4x4-tiled slab a range of tile rows is a contiguous range of columns, so
//   segA[b][query block j] : per level, the slab-column segment the 128 queries of block j can have touched
//                            (K segments of  dfmap1 = f2ext . dpyr^T);
//   segB[b][column block j]: the range of query rows that can have touched the block's slab columns
//                            (one K segment of  df2ext = fmap1 . dpyr).
// Two small launches: the rows (workgroup per query row), then the two tables (thread per block).
// Step 1: one workgroup per (batch item, query row): min / max of cy over the row's queries and all lookups -> the
// texel rows of every level the row's windows can have touched (rows table [B][L][H][2] in the workspace).
__global__ __launch_bounds__(256) void corr_window_rows_kernel(CoordList cl, int H, int W, int r, PyrLayout P,
                                                                int* __restrict__ rows) {
  __shared__ float s_lo[4], s_hi[4];
  const int b = blockIdx.y, qy = blockIdx.x, Q = H * W;
  float lo = 3.0e38f, hi = -3.0e38f;
  bool bad = false;
  for (int i = 0; i < cl.n; ++i) {
    const float* cy = cl.p[i] + ((long long)b * 2 + 1) * Q + (long long)qy * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
      const float v = cy[x];
      bad = bad || !(v == v);
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  }
  if (bad) { lo = -3.0e38f; hi = 3.0e38f; }   // NaN coordinates: assume every row
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x < P.L) {
    const int l = threadIdx.x;
    lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    const float inv = 1.0f / (float)(1 << l);
    const float flo = fminf(fmaxf(floorf(lo * inv), -1.0e8f), 1.0e8f), fhi = fminf(fmaxf(floorf(hi * inv), -1.0e8f), 1.0e8f);
    int* o = rows + (((long long)b * P.L + l) * H + qy) * 2;
    o[0] = max((int)flo - r - 1, 0);
    o[1] = min((int)fhi + r + 2, P.h[l] - 1);     // (may be < the low end: nothing touched)
  }
}

// Step 2: the two segment tables from the rows table (thread per 128-wide block).
__global__ __launch_bounds__(256) void corr_window_segments_kernel(const int* __restrict__ rows, int H, int W, PyrLayout P,
                                                                    int nbA, int nbB, int* __restrict__ segA,
                                                                    int* __restrict__ segB) {
  const int b = blockIdx.y, Q = H * W;
  const int* s_rows = rows + (long long)b * P.L * H * 2;   // [L][H][2]
  const int j0 = blockIdx.x * blockDim.x + threadIdx.x;
  // table A: query block j = queries [128 j, 128 j + 127] = query rows qa..qb
  if (j0 < nbA) {
    const int j = j0;
    const int qa = (j * BN) / W, qb = min((j * BN + BN - 1) / W, H - 1);
    int* o = segA + ((long long)b * nbA + j) * SEG_INTS;
    int n = 0;
    for (int l = 0; l < P.L; ++l) {
      int ylo = 1 << 30, yhi = -1;
      for (int qy = qa; qy <= qb; ++qy) {
        const int a = s_rows[(l * H + qy) * 2], c = s_rows[(l * H + qy) * 2 + 1];
        if (a <= c) { ylo = min(ylo, a); yhi = max(yhi, c); }
      }
      if (ylo <= yhi) {
        const int beg = P.off[l] + (ylo >> 2) * P.tw[l] * 16, end = P.off[l] + ((yhi >> 2) + 1) * P.tw[l] * 16;
        if (n > 0 && o[2 * n] == beg) o[2 * n] = end;           // adjacent to the previous segment: merge
        else { o[1 + 2 * n] = beg; o[2 + 2 * n] = end; ++n; }
      }
    }
    o[0] = n;
    for (int i = n; i < 4; ++i) { o[1 + 2 * i] = 0; o[2 + 2 * i] = 0; }
    o[9] = 0;
  }
  // table B: column block j = slab columns [128 j, 128 j + 127]: hull of the query rows that reach any of its tile rows
  if (j0 >= nbA && j0 < nbA + nbB) {
    const int j = j0 - nbA;
    const int ca = j * BN, cb = min(j * BN + BN, P.slab) - 1;
    int qlo = 1 << 30, qhi = -1;
    for (int l = 0; l < P.L; ++l) {
      const int lbeg = P.off[l], lend = (l + 1 < P.L ? P.off[l + 1] : P.zero) - 1;
      const int a = max(ca, lbeg), c = min(cb, lend);
      if (a > c) continue;
      const int ya = ((a - lbeg) / (P.tw[l] * 16)) * 4, yb = ((c - lbeg) / (P.tw[l] * 16)) * 4 + 3;   // texel rows
      for (int qy = 0; qy < H; ++qy) {
        const int ra = s_rows[(l * H + qy) * 2], rc = s_rows[(l * H + qy) * 2 + 1];
        if (ra <= rc && ra <= yb && rc >= ya) { qlo = min(qlo, qy); qhi = max(qhi, qy); }
      }
    }
    int* o = segB + ((long long)b * nbB + j) * SEG_INTS;
    for (int i = 0; i < SEG_INTS; ++i) o[i] = 0;
    if (qlo <= qhi) {
      o[0] = 1;
      o[1] = (qlo * W) / BK * BK;
      o[2] = min(((qhi + 1) * W + BK - 1) / BK * BK, (Q + BK - 1) / BK * BK);
    }
  }
}

int choose_kchunk(int K, int splits) {
  int per = (K + splits - 1) / splits;
  return ((per + BK - 1) / BK) * BK;
}

}  // namespace

extern "C" long long pcfa_corr_slab_floats(int H, int W, int num_levels) {
  PyrLayout P;
  if (!pcfa_make_layout(P, H, W, num_levels)) return -1;
  return P.slab;
}

extern "C" long long pcfa_corr_level_offset(int H, int W, int num_levels, int level, int* h_l,
                                            int* w_l) {
  PyrLayout P;
  if (!pcfa_make_layout(P, H, W, num_levels) || level < 0 || level >= num_levels) return -1;
  if (h_l) *h_l = P.h[level];
  if (w_l) *w_l = P.w[level];
  return P.off[level];
}

extern "C" long long pcfa_corr_tiled_index(int H, int W, int num_levels, int level, int y, int x) {
  PyrLayout P;
  if (!pcfa_make_layout(P, H, W, num_levels) || level < 0 || level >= num_levels || y < 0 ||
      x < 0 || y >= P.h[level] || x >= P.w[level])
    return -1;
  return pcfa_tiled_index(P, level, y, x);
}

extern "C" int pcfa_corr_f2ext_fwd(const float* fmap2, float* f2ext, int B, int D, int H, int W,
                                   int num_levels, void* stream) {
  PyrLayout P;
  if (!fmap2 || !f2ext || B < 1 || D < 1 || !pcfa_make_layout(P, H, W, num_levels))
    return PCFA_ERR_INVALID_ARG;
  for (int l = 0; l < P.L; ++l)
    if (P.h[l] < 1 || P.w[l] < 1) return PCFA_ERR_INVALID_ARG;
  const int Q = H * W;
  const size_t lds = pooled_lds_bytes(P);
  if (lds > 64 * 1024) return PCFA_ERR_UNSUPPORTED;
  pcfa_launch(f2ext_fwd_kernel, dim3(B * D), dim3(256), lds, (hipStream_t)stream, fmap2,
                     f2ext, Q, P);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_corr_pyramid_fwd(const float* fmap1, const float* f2ext, float* pyr, int B,
                                     int D, int H, int W, int num_levels, void* stream) {
  PyrLayout P;
  if (!fmap1 || !f2ext || !pyr || B < 1 || D < 1 || !pcfa_make_layout(P, H, W, num_levels))
    return PCFA_ERR_INVALID_ARG;
  const int Q = H * W, S = P.slab;
  const int vecA = (Q % 4 == 0) && aligned16(fmap1);
  const int vecB = aligned16(f2ext);  // slab % 16 == 0 by construction
  static const bool pool_off = getenv("PCFA_PYRAMID_POOL") && atoi(getenv("PCFA_PYRAMID_POOL")) == 0;   // A/B switch
  if (vecA && vecB && D % 4 == 0 && Q >= 4 && P.L >= 3 && W % 16 == 0 && !pool_off) {
    PoolArgs pa;
    pa.tw0 = P.tw[0];
    pa.tiles0 = ((P.h[0] + 3) / 4) * P.tw[0];
    pa.S0 = pa.tiles0 * 16;
    pa.nb0 = pcfa_cdiv(pa.S0, BN);
    pa.off_tail = P.L > 3 ? P.off[3] : P.zero;
    pa.off1 = P.off[1]; pa.tw1 = P.tw[1]; pa.h1 = P.h[1]; pa.w1 = P.w[1];
    pa.off2 = P.off[2]; pa.tw2 = P.tw[2]; pa.h2 = P.h[2]; pa.w2 = P.w[2];
    if (pa.S0 != P.off[1] || pa.off_tail % 16 != 0 || pa.off_tail > S) return PCFA_ERR_UNSUPPORTED;
    dim3 gridp(pa.nb0 + pcfa_cdiv(S - pa.off_tail, BN), pcfa_cdiv(Q, BM), B);
    pcfa_launch(corr_pyramid_pool_gemm_kernel, gridp, dim3(256), 0, (hipStream_t)stream, fmap1, f2ext, pyr, Q, S, D,
                (long long)Q, (long long)S, (long long)S, (long long)D * Q, (long long)D * S, (long long)Q * S,
                sqrtf((float)D), pa);
    PCFA_LAUNCH_CHECK();
    return PCFA_OK;
  }
  dim3 grid(pcfa_cdiv(S, BN), pcfa_cdiv(Q, BM), B);
#define PCFA_GEMM_ARGS fmap1, f2ext, pyr, Q, S, D, (long long)Q, (long long)S, (long long)S, (long long)D * Q, \
                       (long long)D * S, (long long)Q * S, 1, ((D + BK - 1) / BK) * BK, 0LL, sqrtf((float)D), vecA, vecB
  if (vecA && vecB && D % 4 == 0 && Q >= 4)
    pcfa_launch(gemm_f32_mfma_kernel<true, true, true>, grid, dim3(256), 0, (hipStream_t)stream, PCFA_GEMM_ARGS);
  else
    pcfa_launch(gemm_f32_mfma_kernel<true, true, false>, grid, dim3(256), 0, (hipStream_t)stream, PCFA_GEMM_ARGS);
#undef PCFA_GEMM_ARGS
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" size_t pcfa_corr_pyramid_bwd_workspace_bytes(int B, int D, int H, int W,
                                                        int num_levels) {
  PyrLayout P;
  if (B < 1 || D < 1 || !pcfa_make_layout(P, H, W, num_levels)) return 0;
  const size_t Q = (size_t)H * W, S = P.slab;
  // split-K partials of dfmap1 [splits][B][D][Q], of df2ext [splits][B][D][S], + df2ext [B][D][S]
  return sizeof(float) * ((size_t)BWD_SPLITS * B * D * Q + (size_t)BWD_SPLITS * B * D * S +
                          (size_t)B * D * S);
}

static int pyramid_bwd(const float* dpyr, const float* fmap1, const float* f2ext, float* dfmap1, float* dfmap2,
                       void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int num_levels, void* stream,
                       const int* segA, const int* segB);

extern "C" int pcfa_corr_pyramid_bwd(const float* dpyr, const float* fmap1, const float* f2ext,
                                     float* dfmap1, float* dfmap2, void* workspace,
                                     size_t workspace_bytes, int B, int D, int H, int W,
                                     int num_levels, void* stream) {
  return pyramid_bwd(dpyr, fmap1, f2ext, dfmap1, dfmap2, workspace, workspace_bytes, B, D, H, W, num_levels, stream,
                     nullptr, nullptr);
}

extern "C" size_t pcfa_corr_pyramid_bwd_windows_workspace_bytes(int B, int D, int H, int W, int num_levels) {
  PyrLayout P;
  const size_t base = pcfa_corr_pyramid_bwd_workspace_bytes(B, D, H, W, num_levels);
  if (base == 0 || !pcfa_make_layout(P, H, W, num_levels)) return 0;
  const size_t nbA = (size_t)pcfa_cdiv((long long)H * W, BN), nbB = (size_t)pcfa_cdiv(P.slab, BN);
  return ((base + 15) & ~(size_t)15) + sizeof(int) * (SEG_INTS * B * (nbA + nbB) + 2 * (size_t)B * P.L * H);
}

extern "C" int pcfa_corr_pyramid_bwd_windows(const float* dpyr, const float* fmap1, const float* f2ext, float* dfmap1,
                                             float* dfmap2, void* workspace, size_t workspace_bytes,
                                             const float* const* coords, int n_coords, int radius, int B, int D, int H,
                                             int W, int num_levels, void* stream) {
  PyrLayout P;
  if (!dpyr || !fmap1 || !f2ext || !dfmap1 || !dfmap2 || !workspace || B < 1 || D < 1 || radius < 0 ||
      !pcfa_make_layout(P, H, W, num_levels))
    return PCFA_ERR_INVALID_ARG;
  const size_t base = (pcfa_corr_pyramid_bwd_workspace_bytes(B, D, H, W, num_levels) + 15) & ~(size_t)15;
  const long long Q = (long long)H * W;
  const bool fast = Q % 4 == 0 && aligned16(f2ext) && aligned16(dpyr) && aligned16(fmap1) && Q >= 4;
  // (a segment record holds four levels' ranges: deeper pyramids take the dense products)
  if (!coords || n_coords < 1 || n_coords > MAX_COORDS || !fast || H > 65535 || num_levels > 4)   // no window information:
    return pyramid_bwd(dpyr, fmap1, f2ext, dfmap1, dfmap2, workspace, workspace_bytes, B, D, H, W, num_levels, stream,
                       nullptr, nullptr);                                                   // the dense products
  if (workspace_bytes < pcfa_corr_pyramid_bwd_windows_workspace_bytes(B, D, H, W, num_levels)) return PCFA_ERR_WORKSPACE;
  const int nbA = pcfa_cdiv(Q, BN), nbB = pcfa_cdiv(P.slab, BN);
  int* segA = reinterpret_cast<int*>((char*)workspace + base);
  int* segB = segA + (size_t)SEG_INTS * B * nbA;
  CoordList cl;
  cl.n = n_coords;
  for (int i = 0; i < MAX_COORDS; ++i) cl.p[i] = i < n_coords ? coords[i] : nullptr;
  for (int i = 0; i < n_coords; ++i)
    if (!cl.p[i]) return PCFA_ERR_INVALID_ARG;
  int* rows = segB + (size_t)SEG_INTS * B * nbB;
  pcfa_launch(corr_window_rows_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, cl, H, W, radius, P, rows);
  PCFA_LAUNCH_CHECK();
  pcfa_launch(corr_window_segments_kernel, dim3(pcfa_cdiv(nbA + nbB, 256), B), dim3(256), 0, (hipStream_t)stream,
              (const int*)rows, H, W, P, nbA, nbB, segA, segB);
  PCFA_LAUNCH_CHECK();
  return pyramid_bwd(dpyr, fmap1, f2ext, dfmap1, dfmap2, workspace, base, B, D, H, W, num_levels, stream, segA, segB);
}

static int pyramid_bwd(const float* dpyr, const float* fmap1, const float* f2ext, float* dfmap1, float* dfmap2,
                       void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int num_levels, void* stream,
                       const int* segA, const int* segB) {
  PyrLayout P;
  if (!dpyr || !fmap1 || !f2ext || !dfmap1 || !dfmap2 || !workspace || B < 1 || D < 1 ||
      !pcfa_make_layout(P, H, W, num_levels))
    return PCFA_ERR_INVALID_ARG;
  if (workspace_bytes < pcfa_corr_pyramid_bwd_workspace_bytes(B, D, H, W, num_levels))
    return PCFA_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int Q = H * W, S = P.slab;
  const float div = sqrtf((float)D);
  float* part1 = (float*)workspace;
  float* part2 = part1 + (size_t)BWD_SPLITS * B * D * Q;
  float* df2ext = part2 + (size_t)BWD_SPLITS * B * D * S;

  // Measured and NOT the default: correct (parity-tested with PCFA_PYRAMID_UNPOOL=1) but slower at 55x128 -- 503 / 418 us
  // against 327 / 307 us for the products over all slab columns: four loads and ~40 VALU instructions per 16-B piece
  // land after the MFMA block of every stage.  Building the un-pooled tile once per (query, tile) in LDS is the fix.
  static const bool unpool_on = getenv("PCFA_PYRAMID_UNPOOL") && atoi(getenv("PCFA_PYRAMID_UNPOOL")) == 1;
  if (P.L == 4 && W % 16 == 0 && Q % 4 == 0 && aligned16(f2ext) && aligned16(dpyr) && aligned16(fmap1) && unpool_on) {
    UnpoolArgs ua;
    ua.tw0 = P.tw[0];
    ua.off1 = P.off[1]; ua.tw1 = P.tw[1]; ua.h1 = P.h[1]; ua.w1 = P.w[1];
    ua.off2 = P.off[2]; ua.tw2 = P.tw[2]; ua.h2 = P.h[2]; ua.w2 = P.w[2];
    ua.off3 = P.off[3]; ua.tw3 = P.tw[3]; ua.h3 = P.h[3]; ua.w3 = P.w[3];
    const int S0 = P.off[1];   // level-0 columns (tiles x 16)
    // (a) dfmap1[d][q] = sum over level-0 columns n of fmap2_tiled[d][n] * G0[q][n] / sqrt(D)   (f2ext[:, :S0] IS fmap2)
    {
      const int kchunk = choose_kchunk(S0, BWD_SPLITS);
      dim3 grid(pcfa_cdiv(Q, BN), pcfa_cdiv(D, BM), B * BWD_SPLITS);
      pcfa_launch(corr_pyramid_unpool_gemm_kernel<false>, grid, dim3(256), 0, s, f2ext, dpyr, part1, D, Q, S0,
                  (long long)S, (long long)S, (long long)Q, (long long)D * S, (long long)Q * S, (long long)D * Q,
                  BWD_SPLITS, kchunk, (long long)B * D * Q, div, ua);
      PCFA_LAUNCH_CHECK();
      const long long n = (long long)B * D * Q;
      pcfa_launch(splitk_reduce_kernel, dim3(min(pcfa_cdiv(n, 256), 2048)), dim3(256), 0, s, part1, dfmap1, n,
                  BWD_SPLITS, n);
      PCFA_LAUNCH_CHECK();
    }
    // (b) dfmap2_tiled[d][n] = sum_q fmap1[d][q] * G0[q][n] / sqrt(D), then back to image order
    {
      const int kchunk = choose_kchunk(Q, BWD_SPLITS);
      dim3 grid(pcfa_cdiv(S0, BN), pcfa_cdiv(D, BM), B * BWD_SPLITS);
      pcfa_launch(corr_pyramid_unpool_gemm_kernel<true>, grid, dim3(256), 0, s, fmap1, dpyr, part2, D, S0, Q,
                  (long long)Q, (long long)S, (long long)S0, (long long)D * Q, (long long)Q * S, (long long)D * S0,
                  BWD_SPLITS, kchunk, (long long)B * D * S0, div, ua);
      PCFA_LAUNCH_CHECK();
      const long long n = (long long)B * D * Q;
      pcfa_launch(splitk_reduce_untile_kernel, dim3(min(pcfa_cdiv(n, 256), 2048)), dim3(256), 0, s, part2, dfmap2, B * D,
                  H, W, P.tw[0], S0, BWD_SPLITS, (long long)B * D * S0);
      PCFA_LAUNCH_CHECK();
    }
    return PCFA_OK;
  }
  // (a) dfmap1[d][q] = sum_n f2ext[d][n] * dpyr[q][n] / sqrt(D):  A=[M=D][K=S], B=[N=Q][K=S]
  {
    const int kchunk = choose_kchunk(S, BWD_SPLITS);
    const int vec = aligned16(f2ext) && aligned16(dpyr);
    dim3 grid(pcfa_cdiv(Q, BN), pcfa_cdiv(D, BM), B * BWD_SPLITS);
#define PCFA_GEMM_ARGS f2ext, dpyr, part1, D, Q, S, (long long)S, (long long)S, (long long)Q, (long long)D * S, \
                       (long long)Q * S, (long long)D * Q, BWD_SPLITS, kchunk, (long long)B * D * Q, div, vec, vec
    // FAST: k (= slab index) is the contiguous axis of both operands; slab % 16 == 0 and kchunk % 16 == 0
    if (segA != nullptr)   // only the slab columns some lookup window touched
      pcfa_launch(gemm_f32_mfma_sparse_kernel<false>, grid, dim3(256), 0, s, f2ext, dpyr, part1, D, Q, S, (long long)S,
                  (long long)S, (long long)Q, (long long)D * S, (long long)Q * S, (long long)D * Q, BWD_SPLITS,
                  (long long)B * D * Q, div, segA);
    else if (vec)
      pcfa_launch(gemm_f32_mfma_kernel<false, false, true>, grid, dim3(256), 0, s, PCFA_GEMM_ARGS);
    else
      pcfa_launch(gemm_f32_mfma_kernel<false, false, false>, grid, dim3(256), 0, s, PCFA_GEMM_ARGS);
#undef PCFA_GEMM_ARGS
    PCFA_LAUNCH_CHECK();
    const long long n = (long long)B * D * Q;
    pcfa_launch(splitk_reduce_kernel, dim3(min(pcfa_cdiv(n, 256), 2048)), dim3(256), 0, s,
                       part1, dfmap1, n, BWD_SPLITS, n);
    PCFA_LAUNCH_CHECK();
  }
  // (b) df2ext[d][n] = sum_q fmap1[d][q] * dpyr[q][n] / sqrt(D):  A=[M=D][K=Q], B=[K=Q][N=S]
  {
    const int kchunk = choose_kchunk(Q, BWD_SPLITS);
    const int vecA = (Q % 4 == 0) && aligned16(fmap1);
    const int vecB = aligned16(dpyr);
    dim3 grid(pcfa_cdiv(S, BN), pcfa_cdiv(D, BM), B * BWD_SPLITS);
#define PCFA_GEMM_ARGS fmap1, dpyr, part2, D, S, Q, (long long)Q, (long long)S, (long long)S, (long long)D * Q, \
                       (long long)Q * S, (long long)D * S, BWD_SPLITS, kchunk, (long long)B * D * S, div, vecA, vecB
    if (segB != nullptr)   // only the query rows whose windows reach the column block
      pcfa_launch(gemm_f32_mfma_sparse_kernel<true>, grid, dim3(256), 0, s, fmap1, dpyr, part2, D, S, Q, (long long)Q,
                  (long long)S, (long long)S, (long long)D * Q, (long long)Q * S, (long long)D * S, BWD_SPLITS,
                  (long long)B * D * S, div, segB);
    else if (vecA && vecB && Q >= 4)   // k = query index: contiguous in fmap1 (Q % 4 == 0), row index of dpyr
      pcfa_launch(gemm_f32_mfma_kernel<false, true, true>, grid, dim3(256), 0, s, PCFA_GEMM_ARGS);
    else
      pcfa_launch(gemm_f32_mfma_kernel<false, true, false>, grid, dim3(256), 0, s, PCFA_GEMM_ARGS);
#undef PCFA_GEMM_ARGS
    PCFA_LAUNCH_CHECK();
    const long long n = (long long)B * D * S;
    pcfa_launch(splitk_reduce_kernel, dim3(min(pcfa_cdiv(n, 256), 2048)), dim3(256), 0, s,
                       part2, df2ext, n, BWD_SPLITS, n);
    PCFA_LAUNCH_CHECK();
  }
  // (c) adjoint of the pooling chain
  {
    const size_t lds = pooled_lds_bytes(P);
    if (lds > 64 * 1024) return PCFA_ERR_UNSUPPORTED;
    pcfa_launch(f2ext_bwd_kernel, dim3(B * D), dim3(256), lds, s, df2ext, dfmap2, Q, P);
    PCFA_LAUNCH_CHECK();
  }
  return PCFA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The same GEMM core for GMA's attention products (models/gma/gma.py:34-77 Attention, :79-115
// Aggregate, used models/gma/update.py:128-130):
//   C[b][m][n] = alpha * sum_k A(m,k) B(k,n)        exact fp32 products on v_mfma_f32_32x32x2_f32
// a_kmajor: 0 = A stored [M][K] (k contiguous, lda = row stride), 1 = stored [K][M];  b_kmajor: 0 = B stored [N][K],
// 1 = stored [K][N].  splits > 1: split-K into `workspace` ([splits][batch][M][N] floats) + an ordered reduction
// (deterministic) -- for the N = 128 products of Aggregate, whose 128x128 tiles would otherwise be 55 workgroups.
// ---------------------------------------------------------------------------------------------------------------
extern "C" size_t pcfa_gemm_f32_workspace_bytes(int M, int N, int batch, int splits) {
  if (M < 1 || N < 1 || batch < 1 || splits < 2) return 0;
  return sizeof(float) * (size_t)splits * batch * M * N;
}

extern "C" int pcfa_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long long lda,
                             long long ldb, long long ldc, int a_kmajor, int b_kmajor, int batch, long long bsA,
                             long long bsB, long long bsC, float alpha, int splits, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || batch < 1 || splits < 1 || alpha == 0.f)
    return PCFA_ERR_INVALID_ARG;
  if (a_kmajor == 0 && b_kmajor != 0 && b_kmajor != 1) return PCFA_ERR_INVALID_ARG;
  if (a_kmajor == 1 && b_kmajor == 0) return PCFA_ERR_UNSUPPORTED;   // no call site; not instantiated
  hipStream_t s = (hipStream_t)stream;
  float* dst = C;
  long long bsD = bsC, ssD = 0, ldd = ldc;
  int kchunk = ((K + BK - 1) / BK) * BK;
  if (splits > 1) {
    if (!workspace || workspace_bytes < pcfa_gemm_f32_workspace_bytes(M, N, batch, splits)) return PCFA_ERR_WORKSPACE;
    if (ldc != N || (batch > 1 && bsC != (long long)M * N)) return PCFA_ERR_UNSUPPORTED;   // dense C for the reduction
    dst = (float*)workspace;
    bsD = (long long)M * N;
    ssD = (long long)batch * M * N;
    ldd = N;
    kchunk = choose_kchunk(K, splits);
  }
  const int vecA = aligned16(A) && lda % 4 == 0 && (a_kmajor ? M % 4 == 0 : K % 4 == 0) && bsA % 4 == 0;
  const int vecB = aligned16(B) && ldb % 4 == 0 && (b_kmajor ? N % 4 == 0 : K % 4 == 0) && bsB % 4 == 0;
  const bool fast = vecA && vecB && K % 4 == 0 && M >= 4 && N >= 4 && K >= 4;
  dim3 grid(pcfa_cdiv(N, BN), pcfa_cdiv(M, BM), batch * splits);
  const float div = 1.0f / alpha;
#define PCFA_GEMM_ARGS A, B, dst, M, N, K, lda, ldb, ldd, bsA, bsB, bsD, splits, kchunk, ssD, div, vecA, vecB
#define PCFA_GEMM_GO(AK, BK_)                                                                                \
  do {                                                                                                       \
    if (fast) pcfa_launch(gemm_f32_mfma_kernel<AK, BK_, true>, grid, dim3(256), 0, s, PCFA_GEMM_ARGS);       \
    else pcfa_launch(gemm_f32_mfma_kernel<AK, BK_, false>, grid, dim3(256), 0, s, PCFA_GEMM_ARGS);           \
  } while (0)
  if (a_kmajor && b_kmajor) PCFA_GEMM_GO(true, true);
  else if (!a_kmajor && b_kmajor) PCFA_GEMM_GO(false, true);
  else PCFA_GEMM_GO(false, false);
#undef PCFA_GEMM_GO
#undef PCFA_GEMM_ARGS
  PCFA_LAUNCH_CHECK();
  if (splits > 1) {
    const long long n = (long long)batch * M * N;
    pcfa_launch(splitk_reduce_kernel, dim3(min(pcfa_cdiv(n, 256), 2048)), dim3(256), 0, s, (const float*)workspace, C,
                n, splits, n);
    PCFA_LAUNCH_CHECK();
  }
  return PCFA_OK;
}

