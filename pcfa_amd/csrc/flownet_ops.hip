// FlowNet2's three native operators for gfx950: the FlowNetC correlation layer, Resample2d
// (bilinear backward warp) and ChannelNorm.  The reference ships them as CUDA-only extensions
//   models/FlowNet/correlation_package/correlation_cuda_kernel.cu:74-147 (forward), :150-333 (backward)
//   models/FlowNet/resample2d_package/resample2d_kernel.cu:16-72 (forward), :75-201 (backward)
//   models/FlowNet/channelnorm_package/channelnorm_kernel.cu:18-60 (forward), :63-96 (backward)
// bound as correlation_cuda / resample2d_cuda / channelnorm_cuda (*.cc).
//
// Correlation (kernel_size 1, stride1 1, pad = max_displacement = 20, stride2 2 in FlowNetC, FlowNetC.py:31-35):
//   out[b][tj*21+ti][y][x] = 1/C * sum_c in1[b][c][y][x] * in2[b][c][y + 2(tj-10)][x + 2(ti-10)]   (0 outside)
// Displacements are even, so a pixel only ever meets pixels of ITS OWN column parity: the fast kernels keep the
// columns of every LDS row de-interleaved ([even | odd]) and a thread owns 4 same-parity pixels, which turns the
// 21 horizontal displacements into 24 CONSECUTIVE LDS floats (6 x ds_read_b128 for 84 FMAs).  The reference
// first transposes both inputs into zero-padded channels-last copies (rInput1/2, 2 x 21 MB at 56x128x256); here
// the padding is a predicate on the staging loads and no copy exists.
//   forward : workgroup = 8x32 pixels x 3 displacement rows (12 in2 rows x 72 columns x 8 channels in LDS),
//             thread = 4 pixels x 21 displacements (84 accumulators), next channel chunk prefetched into registers
//             across the FMA block; lane pairs swap halves at the end so that stores are 16-B vectors.
//   backward: both gradients are gathers, gin1[c][p] = 1/C sum_d g[d][p] in2[c][p+2d],
//             gin2[c][p] = 1/C sum_d g[d][p-2d] in1[c][p-2d]; workgroup = 8x32 pixels x 8 channels with the
//             48x72 halo of the other map in LDS, thread = 1 pixel x 8 channels, 441 taps streamed.  No atomics
//             (the reference has none either), bitwise reproducible.
// Other parameter sets (odd kernel_size, any stride2 / pad; stride1 = 1 for the backward) take one-thread-per-
// element kernels that restate the reference loops.
#include "common.hpp"

namespace {

struct FcParams {
  int B, C, H, W, oH, oW, pad, k, md, s1, s2, drad, dsize;
};

bool fc_make_params(FcParams& p, int B, int C, int H, int W, int pad, int k, int md, int s1, int s2) {
  if (B < 1 || C < 1 || H < 1 || W < 1 || pad < 0 || k < 1 || (k & 1) == 0 || md < 0 || s1 < 1 || s2 < 1)
    return false;
  const int kr = (k - 1) / 2, border = kr + md;
  const int nH = H + 2 * pad - 2 * border, nW = W + 2 * pad - 2 * border;
  if (nH < 1 || nW < 1) return false;
  p = FcParams{B, C, H, W, (nH + s1 - 1) / s1, (nW + s1 - 1) / s1, pad, k, md, s1, s2, md / s2, 2 * (md / s2) + 1};
  return true;
}

bool fc_is_flownetc(const FcParams& p) {
  return p.k == 1 && p.s1 == 1 && p.s2 == 2 && p.md == 20 && p.pad == 20;
}

// ---------------------------------------------------------------------------------------------------------
// generic kernels: the reference loops, one thread per output element
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fc_at(const float* __restrict__ img, int H, int W, int y, int x) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.f;
}

__global__ void fcorr_fwd_generic_kernel(const float* __restrict__ in1, const float* __restrict__ in2,
                                         float* __restrict__ out, FcParams p) {
  const long long total = (long long)p.B * p.dsize * p.dsize * p.oH * p.oW;
  const long long step = (long long)gridDim.x * blockDim.x;
  const int kr = (p.k - 1) / 2;
  const float nelems = (float)(p.k * p.k * p.C);
  const size_t plane = (size_t)p.H * p.W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += step) {
    long long t = idx;
    const int x = t % p.oW; t /= p.oW;
    const int y = t % p.oH; t /= p.oH;
    const int tc = t % (p.dsize * p.dsize); t /= (p.dsize * p.dsize);
    const int b = (int)t;
    // positions in un-padded coordinates (padded index - pad)
    const int y1 = y * p.s1 + p.md - p.pad, x1 = x * p.s1 + p.md - p.pad;
    const int y2 = y1 + (tc / p.dsize - p.drad) * p.s2, x2 = x1 + (tc % p.dsize - p.drad) * p.s2;
    float acc = 0.f;
    for (int j = -kr; j <= kr; ++j)
      for (int i = -kr; i <= kr; ++i)
        for (int c = 0; c < p.C; ++c) {
          const float* a = in1 + ((size_t)b * p.C + c) * plane;
          const float* q = in2 + ((size_t)b * p.C + c) * plane;
          acc += fc_at(a, p.H, p.W, y1 + j, x1 + i) * fc_at(q, p.H, p.W, y2 + j, x2 + i);
        }
    out[idx] = acc / nelems;
  }
}

// WHICH = 1: gradient w.r.t. in1 (other = in2, correlation_cuda_kernel.cu:150-241);
// WHICH = 2: w.r.t. in2 (other = in1, :243-333).  stride1 = 1.
template <int WHICH>
__global__ void fcorr_bwd_generic_kernel(const float* __restrict__ other, const float* __restrict__ gout,
                                         float* __restrict__ gin, FcParams p) {
  const long long total = (long long)p.B * p.C * p.H * p.W;
  const long long step = (long long)gridDim.x * blockDim.x;
  const int kr = (p.k - 1) / 2;
  const float nelems = (float)(p.k * p.k * p.C);
  const size_t plane = (size_t)p.H * p.W, oplane = (size_t)p.oH * p.oW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += step) {
    long long t = idx;
    const int x = t % p.W; t /= p.W;
    const int y = t % p.H; t /= p.H;
    const int c = t % p.C; t /= p.C;
    const int b = (int)t;
    const float* oc = other + ((size_t)b * p.C + c) * plane;
    const float* gb = gout + (size_t)b * p.dsize * p.dsize * oplane;
    float s = 0.f;
    for (int tc = 0; tc < p.dsize * p.dsize; ++tc) {
      const int i2 = (tc % p.dsize - p.drad) * p.s2, j2 = (tc / p.dsize - p.drad) * p.s2;
      const int sy = (WHICH == 1) ? 0 : j2, sx = (WHICH == 1) ? 0 : i2;
      int ymin = y + p.pad - kr - p.md - sy, ymax = y + p.pad + kr - p.md - sy;
      int xmin = x + p.pad - kr - p.md - sx, xmax = x + p.pad + kr - p.md - sx;
      if (xmax < 0 || ymax < 0 || xmin >= p.oW || ymin >= p.oH) continue;
      xmin = max(0, xmin); xmax = min(p.oW - 1, xmax);
      ymin = max(0, ymin); ymax = min(p.oH - 1, ymax);
      const float v = (WHICH == 1) ? fc_at(oc, p.H, p.W, y + j2, x + i2) : fc_at(oc, p.H, p.W, y - j2, x - i2);
      const float* g = gb + (size_t)tc * oplane;
      for (int j = ymin; j <= ymax; ++j)
        for (int i = xmin; i <= xmax; ++i) s += g[(size_t)j * p.oW + i] * v;
    }
    gin[idx] = s / nelems;
  }
}

// ---------------------------------------------------------------------------------------------------------
// FlowNetC fast path
// ---------------------------------------------------------------------------------------------------------
constexpr int FH = 8, FW = 32;       // pixel tile
constexpr int FD = 21, FR = 10;      // displacements per axis, radius (in steps of 2 pixels)
constexpr int FTJ = 3;               // displacement rows per forward workgroup (7 groups)
constexpr int FCC = 8;               // channels per LDS chunk
constexpr int FRW = FW + 4 * FR;     // 72 halo columns (36 per parity)
constexpr int FRH = FH + 2 * (FTJ - 1);  // 12 in2 rows per forward workgroup
constexpr int FNT = 8 * FH * FTJ;    // 192 threads

__global__ __launch_bounds__(FNT) void fcorr_fwd_fast_kernel(const float* __restrict__ in1,
                                                            const float* __restrict__ in2,
                                                            float* __restrict__ out, int C, int H, int W) {
  // two stages: the next channel chunk is written while the current one is read (one barrier per chunk)
  __shared__ __attribute__((aligned(16))) float s1[2][FCC][FH][FW];    // row = [16 even | 16 odd]
  __shared__ __attribute__((aligned(16))) float s2[2][FCC][FRH][FRW];  // row = [36 even | 36 odd]

  const int ntx = (W + FW - 1) / FW;
  const int tile_x = blockIdx.x % ntx, g = blockIdx.x / ntx;  // g = displacement-row group
  const int b = blockIdx.z;
  const int y0 = blockIdx.y * FH, x0 = tile_x * FW;
  const int tx = threadIdx.x, ty = threadIdx.y, tz = threadIdx.z;
  const int tid = tx + 8 * ty + 64 * tz;
  const int par = tx >> 2, q = tx & 3;
  const size_t plane = (size_t)H * W;
  const float* p1 = in1 + (size_t)b * C * plane;
  const float* p2 = in2 + (size_t)b * C * plane;
  const int ry0 = y0 + 2 * (FTJ * g - FR);  // global row of in2 region row 0
  const int rx0 = x0 - 2 * FR;              // global column of region column 0 (multiple of 4)

  // Staging plan in 16-B pieces (W % 4 == 0 and 16-B aligned inputs are preconditions of this kernel, so a
  // piece is either inside the image or outside): a chunk is FCC x (8 rows x 8 pieces) of in1 + FCC x (12 rows x
  // 18 pieces) of in2 = 512 + 1728 pieces = 3 + 9 per thread.  Piece (x .. x+3) lands de-interleaved:
  // (x, x+2) -> even half, (x+1, x+3) -> odd half, two 8-B LDS writes.
  constexpr int N1 = FCC * FH * (FW / 4), N2 = FCC * FRH * (FRW / 4);
  constexpr int S1 = (N1 + FNT - 1) / FNT, S2 = (N2 + FNT - 1) / FNT;
  static_assert(N2 % FNT == 0, "in2 pieces must divide evenly");
  int o1[S1], l1[S1], o2[S2], l2[S2];
#pragma unroll
  for (int s = 0; s < S1; ++s) {
    const int e = tid + s * FNT;
    const int c = e / (FH * (FW / 4)), r = (e / (FW / 4)) % FH, m = e % (FW / 4);
    const int gy = y0 + r, gx = x0 + 4 * m;
    const bool live = e < N1;
    o1[s] = (live && gy < H && gx < W) ? (int)(c * plane) + gy * W + gx : -1;
    l1[s] = live ? (c * FH + r) * FW + 2 * m : -1;
  }
#pragma unroll
  for (int s = 0; s < S2; ++s) {
    const int e = tid + s * FNT;
    const int c = e / (FRH * (FRW / 4)), r = (e / (FRW / 4)) % FRH, m = e % (FRW / 4);
    const int gy = ry0 + r, gx = rx0 + 4 * m;
    o2[s] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (int)(c * plane) + gy * W + gx : -1;
    l2[s] = (c * FRH + r) * FRW + 2 * m;
  }

  float acc[4][FD];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int d = 0; d < FD; ++d) acc[p][d] = 0.f;

  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 r1[S1], r2[S2];
  auto fetch = [&](int c0) {
    // branch-free: a dead piece reads offset 0 of the chunk (always mapped) and is zeroed afterwards; a channel
    // tail (C % FCC) is dead through the offset limit
    const float* b1 = p1 + (size_t)c0 * plane;
    const float* b2 = p2 + (size_t)c0 * plane;
    const int climit = (int)((size_t)(C - c0) * plane);  // offsets at or past this belong to channels >= C
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      const bool ok = o1[s] >= 0 && o1[s] < climit;
      const f32x4 t = *reinterpret_cast<const f32x4*>(b1 + (ok ? o1[s] : 0));
      r1[s] = ok ? t : (f32x4)(0.f);
    }
#pragma unroll
    for (int s = 0; s < S2; ++s) {
      const bool ok = o2[s] >= 0 && o2[s] < climit;
      const f32x4 t = *reinterpret_cast<const f32x4*>(b2 + (ok ? o2[s] : 0));
      r2[s] = ok ? t : (f32x4)(0.f);
    }
  };
  auto commit = [&](int buf) {
    float* d1 = &s1[buf][0][0][0];
    float* d2 = &s2[buf][0][0][0];
#pragma unroll
    for (int s = 0; s < S1; ++s)
      if (l1[s] >= 0) {
        *reinterpret_cast<float2*>(d1 + l1[s]) = make_float2(r1[s].x, r1[s].z);
        *reinterpret_cast<float2*>(d1 + l1[s] + FW / 2) = make_float2(r1[s].y, r1[s].w);
      }
#pragma unroll
    for (int s = 0; s < S2; ++s) {
      *reinterpret_cast<float2*>(d2 + l2[s]) = make_float2(r2[s].x, r2[s].z);
      *reinterpret_cast<float2*>(d2 + l2[s] + FRW / 2) = make_float2(r2[s].y, r2[s].w);
    }
  };

  fetch(0);
  commit(0);
  __syncthreads();
  int cur = 0;
  for (int c0 = 0; c0 < C; c0 += FCC) {
    const bool more = c0 + FCC < C;
    if (more) fetch(c0 + FCC);  // in flight across the FMA block below
#pragma unroll 2
    for (int c = 0; c < FCC; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(&s1[cur][c][ty][par * (FW / 2) + 4 * q]);
      const float* row = &s2[cur][c][ty + 2 * tz][par * (FRW / 2) + 4 * q];
      float v[24];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(row + 4 * k);
        v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
      }
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int d = 0; d < FD; ++d) acc[p][d] += av[p] * v[p + d];
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // Thread (par, q) holds pixels x0 + 8q + 2p + par.  Lane pairs (tx, tx^4) swap halves so that the even lane
  // stores x0+8q .. +3 and the odd lane x0+8q+4 .. +7 as one 16-B vector each.
  // the reference divides by C (correlation_cuda_kernel.cu:143); for a power of two the reciprocal gives the same
  // fp32 result and saves 84 division sequences per thread
  const float nelems = (float)C;
  const int gy = y0 + ty;
  const int tj = FTJ * g + tz;
  const int gx = x0 + 8 * q + 4 * par;
  float* ob = out + ((size_t)b * FD * FD + (size_t)tj * FD) * plane + (size_t)gy * W + gx;
  if ((C & (C - 1)) == 0) {
    const float rn = 1.0f / nelems;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int d = 0; d < FD; ++d) acc[p][d] *= rn;
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int d = 0; d < FD; ++d) acc[p][d] /= nelems;
  }
#pragma unroll
  for (int d = 0; d < FD; ++d) {
    const float m0 = acc[0][d], m1 = acc[1][d], m2 = acc[2][d], m3 = acc[3][d];
    // even lane keeps (m0, m1) and needs the partner's (m0, m1); odd lane keeps (m2, m3), needs partner's (m2, m3)
    const float send0 = par ? m0 : m2, send1 = par ? m1 : m3;
    const float recv0 = __shfl_xor(send0, 4), recv1 = __shfl_xor(send1, 4);
    const float4 o = par ? make_float4(recv0, m2, recv1, m3) : make_float4(m0, recv0, m1, recv1);
    if (gy < H && gx < W) *reinterpret_cast<float4*>(ob + (size_t)d * plane) = o;
  }
}

constexpr int BRH = FH + 4 * FR;  // 48 halo rows

// gin[c][p] = 1/C * sum_d G(d) * X[c][p + SIGN*2d]:  SIGN=+1: G = g[d][p], X = in2 (grad in1)
//                                                    SIGN=-1: G = g[d][p-2d], X = in1 (grad in2)
// LDS image of X: [2 channel halves][48 rows][72 columns][4 channels] -- the 8 channels of one tap are two 16-B
// reads, consecutive lanes 16 B apart (conflict-free).
template <int SIGN>
__global__ __launch_bounds__(FH* FW) void fcorr_bwd_fast_kernel(const float* __restrict__ X,
                                                                const float* __restrict__ gout,
                                                                float* __restrict__ gin, int C, int H, int W) {
  static_assert(FCC == 8, "two float4 per tap");
  extern __shared__ __attribute__((aligned(16))) float sx[];  // [2][BRH][FRW][4]
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int ngroups = (C + FCC - 1) / FCC;
  const int b = blockIdx.z / ngroups;
  const int c0 = (blockIdx.z - b * ngroups) * FCC;
  const int y0 = blockIdx.y * FH, x0 = blockIdx.x * FW;
  const int lx = threadIdx.x, ly = threadIdx.y;
  const int tid = lx + FW * ly;
  const int gy = y0 + ly, gx = x0 + lx;
  const bool inside = gy < H && gx < W;
  const size_t plane = (size_t)H * W;
  const float* px = X + ((size_t)b * C + c0) * plane;
  // staging in 16-B pieces (W % 4 == 0, 16-B aligned X: preconditions): 8 ch x 48 rows x 18 pieces = 27 per thread
  constexpr int NP = FCC * BRH * (FRW / 4);
  static_assert(NP % (FH * FW) == 0, "pieces must divide evenly");
  constexpr int BATCH = 9;  // pieces in flight per thread: all loads of a batch are issued before its LDS writes
  static_assert(NP % (FH * FW * BATCH) == 0, "batches must divide evenly");
#pragma unroll 1
  for (int e0 = tid; e0 < NP; e0 += FH * FW * BATCH) {
    f32x4 t[BATCH];
    bool ok[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int e = e0 + k * FH * FW;
      const int c = e / (BRH * (FRW / 4)), r = (e / (FRW / 4)) % BRH, m = e % (FRW / 4);
      const int yy = y0 + r - 2 * FR, xx = x0 + 4 * m - 2 * FR;
      ok[k] = c0 + c < C && yy >= 0 && yy < H && xx >= 0 && xx < W;
      t[k] = *reinterpret_cast<const f32x4*>(px + (ok[k] ? (size_t)c * plane + (size_t)yy * W + xx : 0));
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int e = e0 + k * FH * FW;
      const int c = e / (BRH * (FRW / 4)), r = (e / (FRW / 4)) % BRH, m = e % (FRW / 4);
      float* d = sx + (((c >> 2) * BRH + r) * FRW + 4 * m) * 4 + (c & 3);
      d[0] = ok[k] ? t[k].x : 0.f;
      d[4] = ok[k] ? t[k].y : 0.f;
      d[8] = ok[k] ? t[k].z : 0.f;
      d[12] = ok[k] ? t[k].w : 0.f;
    }
  }
  __syncthreads();

  float acc[FCC];
#pragma unroll
  for (int c = 0; c < FCC; ++c) acc[c] = 0.f;
  const float* gb = gout + (size_t)b * FD * FD * plane;
  const int cy = min(gy, H - 1), cx = min(gx, W - 1);  // threads past the edge compute a clamped pixel, never store
  // The 21 gradient taps of displacement row tj+1 are requested before the FMAs of row tj (one wave per SIMD:
  // nothing else hides the load latency).
  auto load_row = [&](int tj, float (&gv)[FD]) {
    const int dy = 2 * (tj - FR);
    const int sy = (SIGN > 0) ? cy : cy - dy;  // pixel whose gradient row is read
    const bool rowok = sy >= 0 && sy < H;
    const float* grow = gb + (size_t)(tj * FD) * plane + (size_t)(rowok ? sy : 0) * W;
#pragma unroll
    for (int ti = 0; ti < FD; ++ti) {  // branch-free: clamped address, zeroed afterwards
      const int sxp = (SIGN > 0) ? cx : cx - 2 * (ti - FR);
      const bool ok = rowok && sxp >= 0 && sxp < W;
      const float t = grow[(size_t)ti * plane + (ok ? sxp : 0)];
      gv[ti] = ok ? t : 0.f;
    }
  };
  float gcur[FD], gnext[FD];
  load_row(0, gcur);
#pragma unroll 1
  for (int tj = 0; tj < FD; ++tj) {
    load_row(min(tj + 1, FD - 1), gnext);
    const int lr = (SIGN > 0) ? ly + 2 * tj : ly + 4 * FR - 2 * tj;
#pragma unroll
    for (int ti = 0; ti < FD; ++ti) {
      const int lc = (SIGN > 0) ? lx + 2 * ti : lx + 4 * FR - 2 * ti;
      const f32x4* tap = reinterpret_cast<const f32x4*>(sx + (lr * FRW + lc) * 4);
      const f32x4 u0 = tap[0], u1 = tap[BRH * FRW];
      const float gvt = gcur[ti];
      acc[0] += gvt * u0.x; acc[1] += gvt * u0.y; acc[2] += gvt * u0.z; acc[3] += gvt * u0.w;
      acc[4] += gvt * u1.x; acc[5] += gvt * u1.y; acc[6] += gvt * u1.z; acc[7] += gvt * u1.w;
    }
#pragma unroll
    for (int ti = 0; ti < FD; ++ti) gcur[ti] = gnext[ti];
  }
  if (inside) {
    const float nelems = (float)C;
    float* po = gin + ((size_t)b * C + c0) * plane + (size_t)gy * W + gx;
#pragma unroll
    for (int c = 0; c < FCC; ++c)
      if (c0 + c < C) po[(size_t)c * plane] = acc[c] / nelems;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Resample2d (kernel_size 1): out[b][c][y][x] = bilinear(in1[b][c], x + flow_x, y + flow_y), the four neighbour
// indices clamped to the image one by one (resample2d_kernel.cu:44-62).
// ---------------------------------------------------------------------------------------------------------
struct RsTaps {
  int xL, xR, yT, yB;
  float alpha, beta;
};

__device__ __forceinline__ RsTaps rs_taps(float xf, float yf, int h, int w) {
  RsTaps t;
  const float fx = floorf(xf), fy = floorf(yf);
  t.alpha = xf - fx;
  t.beta = yf - fy;
  t.xL = max(min((int)fx, w - 1), 0);
  t.xR = max(min((int)(fx + 1.f), w - 1), 0);
  t.yT = max(min((int)fy, h - 1), 0);
  t.yB = max(min((int)(fy + 1.f), h - 1), 0);
  return t;
}

__global__ void resample2d_fwd_kernel(const float* __restrict__ in1, const float* __restrict__ flow,
                                      float* __restrict__ out, int B, int C, int iH, int iW, int H, int W,
                                      int bilinear) {
  const long long total = (long long)B * H * W;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int x = idx % W, y = (idx / W) % H, b = (int)(idx / ((long long)W * H));
  const size_t plane = (size_t)H * W, iplane = (size_t)iH * iW;
  const float dx = flow[((size_t)b * 2) * plane + (size_t)y * W + x];
  const float dy = flow[((size_t)b * 2 + 1) * plane + (size_t)y * W + x];
  const float xf = (float)x + dx, yf = (float)y + dy;
  const float* src = in1 + (size_t)b * C * iplane;
  float* dst = out + (size_t)b * C * plane + (size_t)y * W + x;
  if (bilinear) {
    // the reference clamps against the OUTPUT size here (resample2d_kernel.cu:50-53)
    const RsTaps t = rs_taps(xf, yf, H, W);
    // weights are formed in double and every term is rounded to float before it is added (:57-60)
    const double a = t.alpha, be = t.beta;
    for (int c = 0; c < C; ++c) {
      const float* s = src + (size_t)c * iplane;
      float val = 0.f;
      val += (float)((1. - a) * (1. - be) * s[(size_t)t.yT * iW + t.xL]);
      val += (float)(a * (1. - be) * s[(size_t)t.yT * iW + t.xR]);
      val += (float)((1. - a) * be * s[(size_t)t.yB * iW + t.xL]);
      val += (float)(a * be * s[(size_t)t.yB * iW + t.xR]);
      dst[(size_t)c * plane] = val;
    }
  } else {
    const int xN = max(min((int)floorf(xf + 0.5f), W - 1), 0);
    const int yN = max(min((int)floorf(yf + 0.5f), H - 1), 0);
    for (int c = 0; c < C; ++c) dst[(size_t)c * plane] = src[(size_t)c * iplane + (size_t)yN * iW + xN];
  }
}

// One thread per output pixel: scatters grad_out into grad_in1 (hardware fp32 atomics, as the reference's
// atomicAdd, resample2d_kernel.cu:113-122) and gathers the flow gradient (:125-201).
__global__ void resample2d_bwd_kernel(const float* __restrict__ in1, const float* __restrict__ flow,
                                      const float* __restrict__ gout, float* __restrict__ gin1,
                                      float* __restrict__ gflow, int B, int C, int iH, int iW, int H, int W) {
  const long long total = (long long)B * H * W;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int x = idx % W, y = (idx / W) % H, b = (int)(idx / ((long long)W * H));
  const size_t plane = (size_t)H * W, iplane = (size_t)iH * iW;
  const float dx = flow[((size_t)b * 2) * plane + (size_t)y * W + x];
  const float dy = flow[((size_t)b * 2 + 1) * plane + (size_t)y * W + x];
  const float xf = (float)x + dx, yf = (float)y + dy;
  // grad_in1: neighbours clamped against the INPUT size, weights from truncation (xf - int(xf), :103-111)
  const RsTaps t1 = rs_taps(xf, yf, iH, iW);
  const float a1 = xf - (float)(int)xf, b1 = yf - (float)(int)yf;
  // grad_flow: neighbours clamped against the flow size, gamma = 1 - frac (:159-170)
  const RsTaps t2 = rs_taps(xf, yf, H, W);
  const float gam_x = 1.f - t2.alpha, gam_y = 1.f - t2.beta;
  const float* src = in1 + (size_t)b * C * iplane;
  const float* g = gout + (size_t)b * C * plane + (size_t)y * W + x;
  float* d1 = gin1 + (size_t)b * C * iplane;
  float gdx = 0.f, gdy = 0.f;
  for (int c = 0; c < C; ++c) {
    const float gv = g[(size_t)c * plane];
    float* d = d1 + (size_t)c * iplane;
    unsafeAtomicAdd(d + (size_t)t1.yT * iW + t1.xL, (1.f - a1) * (1.f - b1) * gv);
    unsafeAtomicAdd(d + (size_t)t1.yT * iW + t1.xR, a1 * (1.f - b1) * gv);
    unsafeAtomicAdd(d + (size_t)t1.yB * iW + t1.xL, (1.f - a1) * b1 * gv);
    unsafeAtomicAdd(d + (size_t)t1.yB * iW + t1.xR, a1 * b1 * gv);
    const float* s = src + (size_t)c * iplane;
    const float iTL = s[(size_t)t2.yT * iW + t2.xL], iTR = s[(size_t)t2.yT * iW + t2.xR];
    const float iBL = s[(size_t)t2.yB * iW + t2.xL], iBR = s[(size_t)t2.yB * iW + t2.xR];
    // channel 0 (d/dx): gamma from the y fraction; channel 1 (d/dy): gamma from the x fraction
    gdx += gam_y * gv * iTR;
    gdx -= gam_y * gv * iTL;
    gdx += (1.f - gam_y) * gv * iBR;
    gdx -= (1.f - gam_y) * gv * iBL;
    gdy += gam_x * gv * iBL;
    gdy -= gam_x * gv * iTL;
    gdy += (1.f - gam_x) * gv * iBR;
    gdy -= (1.f - gam_x) * gv * iTR;
  }
  gflow[((size_t)b * 2) * plane + (size_t)y * W + x] = gdx;
  gflow[((size_t)b * 2 + 1) * plane + (size_t)y * W + x] = gdy;
}

// ---------------------------------------------------------------------------------------------------------
// ChannelNorm: out[b][0][y][x] = sqrt(sum_c in[b][c][y][x]^2)
// ---------------------------------------------------------------------------------------------------------
__global__ void channelnorm_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C,
                                       long long plane) {
  const long long total = (long long)B * plane;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long long b = idx / plane, p = idx - b * plane;
  const float* s = in + (size_t)b * C * plane + p;
  float r = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = s[(size_t)c * plane];
    r += v * v;
  }
  out[idx] = sqrtf(r);
}

__global__ void channelnorm_bwd_kernel(const float* __restrict__ in, const float* __restrict__ out,
                                       const float* __restrict__ gout, float* __restrict__ gin, int B, int C,
                                       long long plane) {
  const long long total = (long long)B * C * plane;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long long b = idx / (C * plane), p = idx % plane;
  // float product divided by (double)(norm + 1e-9), channelnorm_kernel.cu:93
  gin[idx] = (float)((double)(gout[b * plane + p] * in[idx]) / ((double)out[b * plane + p] + 1e-9));
}

// grad_in1 of Resample2d is cleared by a kernel, not hipMemsetAsync: inside torch's stream capture the memset node
// was not replayed with the graph (stale gradients accumulated from replay to replay), a kernel node is.
__global__ void zero_fill_kernel(float* __restrict__ p, long long n) {
  const long long step = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += step) {
    if (i + 3 < n && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (long long k = i; k < n && k < i + 4; ++k) p[k] = 0.f;
    }
  }
}

int blocks_for(long long total, int threads) {
  const long long n = (total + threads - 1) / threads;
  return (int)(n < 65535LL * 32 ? n : 65535LL * 32);
}

}  // namespace

extern "C" int pcfa_flownet_corr_out_size(int H, int W, int pad_size, int kernel_size, int max_displacement,
                                          int stride1, int stride2, int* out_channels, int* oH, int* oW) {
  FcParams p;
  if (!fc_make_params(p, 1, 1, H, W, pad_size, kernel_size, max_displacement, stride1, stride2))
    return PCFA_ERR_INVALID_ARG;
  if (out_channels) *out_channels = p.dsize * p.dsize;
  if (oH) *oH = p.oH;
  if (oW) *oW = p.oW;
  return PCFA_OK;
}

extern "C" int pcfa_flownet_corr_fwd(const float* in1, const float* in2, float* out, int B, int C, int H, int W,
                                     int pad_size, int kernel_size, int max_displacement, int stride1,
                                     int stride2, void* stream) {
  FcParams p;
  if (!in1 || !in2 || !out ||
      !fc_make_params(p, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = W % 4 == 0 && ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2) |
                                        reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (fc_is_flownetc(p) && aligned) {
    dim3 grid(pcfa_cdiv(W, FW) * (FD / FTJ), pcfa_cdiv(H, FH), B), block(8, FH, FTJ);
    pcfa_launch(fcorr_fwd_fast_kernel, grid, block, 0, s, in1, in2, out, C, H, W);
  } else {
    const long long total = (long long)B * p.dsize * p.dsize * p.oH * p.oW;
    pcfa_launch(fcorr_fwd_generic_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in1, in2, out, p);
  }
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_flownet_corr_bwd(const float* in1, const float* in2, const float* grad_out, float* grad_in1,
                                     float* grad_in2, int B, int C, int H, int W, int pad_size, int kernel_size,
                                     int max_displacement, int stride1, int stride2, void* stream) {
  FcParams p;
  if (!in1 || !in2 || !grad_out || !grad_in1 || !grad_in2 ||
      !fc_make_params(p, B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2))
    return PCFA_ERR_INVALID_ARG;
  if (stride1 != 1) return PCFA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = W % 4 == 0 && ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2)) & 15) == 0;
  if (fc_is_flownetc(p) && aligned) {
    static const bool attr_ok = [] {
      const int bytes = FCC * BRH * FRW * (int)sizeof(float);
      return hipFuncSetAttribute((const void*)fcorr_bwd_fast_kernel<+1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 bytes) == hipSuccess &&
             hipFuncSetAttribute((const void*)fcorr_bwd_fast_kernel<-1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 bytes) == hipSuccess;
    }();
    if (!attr_ok) return PCFA_ERR_UNSUPPORTED;
    const size_t lds = (size_t)FCC * BRH * FRW * sizeof(float);
    dim3 grid(pcfa_cdiv(W, FW), pcfa_cdiv(H, FH), B * pcfa_cdiv(C, FCC)), block(FW, FH, 1);
    pcfa_launch(fcorr_bwd_fast_kernel<+1>, grid, block, lds, s, in2, grad_out, grad_in1, C, H, W);
    PCFA_LAUNCH_CHECK();
    pcfa_launch(fcorr_bwd_fast_kernel<-1>, grid, block, lds, s, in1, grad_out, grad_in2, C, H, W);
  } else {
    const long long total = (long long)B * C * H * W;
    pcfa_launch(fcorr_bwd_generic_kernel<1>, dim3(blocks_for(total, 256)), dim3(256), 0, s, in2, grad_out,
                grad_in1, p);
    PCFA_LAUNCH_CHECK();
    pcfa_launch(fcorr_bwd_generic_kernel<2>, dim3(blocks_for(total, 256)), dim3(256), 0, s, in1, grad_out,
                grad_in2, p);
  }
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_resample2d_fwd(const float* in1, const float* flow, float* out, int B, int C, int iH, int iW,
                                   int H, int W, int kernel_size, int bilinear, void* stream) {
  if (!in1 || !flow || !out || B < 1 || C < 1 || iH < 1 || iW < 1 || H < 1 || W < 1) return PCFA_ERR_INVALID_ARG;
  // kernel_size > 1 reads past the image border in the reference (resample2d_kernel.cu:55-62); FlowNet2 uses 1.
  if (kernel_size != 1 || H > iH || W > iW) return PCFA_ERR_UNSUPPORTED;
  const long long total = (long long)B * H * W;
  pcfa_launch(resample2d_fwd_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in1, flow,
              out, B, C, iH, iW, H, W, bilinear);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_resample2d_bwd(const float* in1, const float* flow, const float* grad_out, float* grad_in1,
                                   float* grad_flow, int B, int C, int iH, int iW, int H, int W, int kernel_size,
                                   int bilinear, void* stream) {
  (void)bilinear;  // the reference's backward ignores the flag as well (resample2d_kernel.cu:75-201)
  if (!in1 || !flow || !grad_out || !grad_in1 || !grad_flow || B < 1 || C < 1 || iH < 1 || iW < 1 || H < 1 ||
      W < 1)
    return PCFA_ERR_INVALID_ARG;
  if (kernel_size != 1 || H > iH || W > iW) return PCFA_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long long n1 = (long long)B * C * iH * iW;
  pcfa_launch(zero_fill_kernel, dim3(blocks_for((n1 + 3) / 4, 256)), dim3(256), 0, s, grad_in1, n1);
  PCFA_LAUNCH_CHECK();
  const long long total = (long long)B * H * W;
  pcfa_launch(resample2d_bwd_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in1, flow, grad_out,
              grad_in1, grad_flow, B, C, iH, iW, H, W);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_channelnorm_fwd(const float* in, float* out, int B, int C, long long plane, int norm_deg,
                                    void* stream) {
  if (!in || !out || B < 1 || C < 1 || plane < 1) return PCFA_ERR_INVALID_ARG;
  if (norm_deg != 2) return PCFA_ERR_UNSUPPORTED;  // the reference ignores norm_deg and always computes L2
  pcfa_launch(channelnorm_fwd_kernel, dim3(blocks_for((long long)B * plane, 256)), dim3(256), 0,
              (hipStream_t)stream, in, out, B, C, plane);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_channelnorm_bwd(const float* in, const float* out, const float* grad_out, float* grad_in,
                                    int B, int C, long long plane, int norm_deg, void* stream) {
  if (!in || !out || !grad_out || !grad_in || B < 1 || C < 1 || plane < 1) return PCFA_ERR_INVALID_ARG;
  if (norm_deg != 2) return PCFA_ERR_UNSUPPORTED;
  pcfa_launch(channelnorm_bwd_kernel, dim3(blocks_for((long long)B * C * plane, 256)), dim3(256), 0,
              (hipStream_t)stream, in, out, grad_out, grad_in, B, C, plane);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
