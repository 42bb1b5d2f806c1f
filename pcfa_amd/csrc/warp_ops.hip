// PWC-Net's backward warp as ONE kernel per direction.
//
// Replaces the Python sequence of PWCDCNet.warp (reference models/PWCNet/PWCNet.py:166-206):
//   vgrid = meshgrid + flo;  vx = 2*vgrid_x / max(W-1, 1) - 1;  vy likewise
//   output = grid_sample(x, vgrid)            (bilinear, zero padding, align_corners = False)
//   mask   = grid_sample(ones_like(x), vgrid) >= 0.0001
//   return output * mask
// which the library runs as ~14 launches forward (arange / repeat / cat / normalise, two grid_sampler_2d, compare,
// cast, multiply) and two grid_sampler_2d_backward launches (221 us each at 32 x 96 x 320) plus elementwise
// backward kernels.  Here: forward = one pass (read x where sampled, read flo, write out), backward = one pass that
// scatters grad_x with hardware fp32 atomics (as grid_sampler_2d_backward does) and accumulates grad_flo.
// The coordinate arithmetic repeats the reference's fp32 operation sequence: normalise (x2, /(W-1), -1), then
// grid_sample's un-normalisation ((g + 1) * W - 1) / 2 -- the two do NOT cancel (align_corners mismatch of the
// original PWC-Net code), the sample position is x * W / (W - 1) - 0.5.
#include <cstdlib>
#include "common.hpp"

namespace {

// a * b rounded to fp32 on its own: never contracted into a following add (the scaled flow meets the meshgrid as the
// reference's `up_flow * s` tensor does -- a rounded product).  (__fmul_rn is a plain `*` in this toolchain.)
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

struct WarpTaps {
  int x0, y0;          // north-west tap
  float wx1, wy1;      // weight of the east / south neighbour (ix - x0, iy - y0)
  bool vx0, vx1, vy0, vy1;
};

// Every kernel of this file (forward, fp32-atomic backward, the two fixed-point backward kernels) must form the same sample
// position and the same flow-gradient sums BIT FOR BIT.  With contraction left to the optimiser, the same source expression
// became an fma in one kernel and a multiply + add in another after an unrelated edit (r05: the non-finite flag), and the
// "moves no bit" A/B of the two fixed-point kernels differed in the last place.  So the two shared pieces state their
// roundings explicitly: no contraction inside them, an fma exactly where one is written.
__device__ __forceinline__ float warp_coord(float base, float flow, int size) {
#pragma clang fp contract(off)
  float g = 2.0f * (base + flow);
  g = g / (float)max(size - 1, 1);
  g = g - 1.0f;
  // grid_sampler_unnormalize, align_corners = false: ((g + 1) * size - 1) / 2.  ATen's vectorised CPU kernel (what the port
  // runs) forms it as (g + 1) * (size / 2) - 0.5 with a fused multiply-add: the same value, a power of two apart -- so the
  // fma is written out (an un-fused form misses the port by 1 ulp of the coordinate, 2e-6 of max|x| in the output)
  return fmaf(g + 1.f, (float)size, -1.f) / 2.f;
}

// acc (+/-)= (v * w) * g  as  fma(+/- (v * w), g, acc): one tap's contribution to d out / d ix (or iy)
__device__ __forceinline__ float tap_fma(float acc, float v, float w, float g, bool minus) {
#pragma clang fp contract(off)
  const float t = v * w;
  return fmaf(minus ? -t : t, g, acc);
}

__device__ __forceinline__ WarpTaps warp_taps(float ix, float iy, int H, int W) {
  WarpTaps t;
  const float fx = floorf(ix), fy = floorf(iy);
  t.x0 = (int)fx;
  t.y0 = (int)fy;
  t.wx1 = ix - fx;
  t.wy1 = iy - fy;
  t.vx0 = t.x0 >= 0 && t.x0 < W;
  t.vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
  t.vy0 = t.y0 >= 0 && t.y0 < H;
  t.vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
  return t;
}

// grid = (pixel blocks, channel groups, B); thread = one pixel, channels c = group, group + G, ...
__global__ __launch_bounds__(256) void pwc_warp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ flo,
                                                          float* __restrict__ out, int C, int H, int W,
                                                          float mask_thresh, float fs) {
  const long long plane = (long long)H * W;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int b = blockIdx.z, G = gridDim.y;
  const int py = (int)(p / W), px = (int)(p % W);
  const float* fb = flo + (size_t)b * 2 * plane;
  const float ix = warp_coord((float)px, mul_rounded(fb[p], fs), W), iy = warp_coord((float)py, mul_rounded(fb[plane + p], fs), H);
  const WarpTaps t = warp_taps(ix, iy, H, W);
  // weights as grid_sampler_2d forms them: nw = (ix_se - ix) * (iy_se - iy), ...
  const float ex = (float)(t.x0 + 1) - ix, ey = (float)(t.y0 + 1) - iy;  // ix_se - ix, iy_se - iy
  const float nw = ex * ey, ne = t.wx1 * ey, sw = ex * t.wy1, se = t.wx1 * t.wy1;
  const bool bnw = t.vx0 && t.vy0, bne = t.vx1 && t.vy0, bsw = t.vx0 && t.vy1, bse = t.vx1 && t.vy1;
  float msum = 0.f;  // grid_sample(ones): the in-bounds weights, added in the kernel's tap order
  if (bnw) msum += nw;
  if (bne) msum += ne;
  if (bsw) msum += sw;
  if (bse) msum += se;
  const float m = msum >= mask_thresh ? 1.f : 0.f;
  const int onw = bnw ? t.y0 * W + t.x0 : 0, one = bne ? t.y0 * W + t.x0 + 1 : 0;
  const int osw = bsw ? (t.y0 + 1) * W + t.x0 : 0, ose = bse ? (t.y0 + 1) * W + t.x0 + 1 : 0;
  const float* xb = x + (size_t)b * C * plane;
  float* ob = out + (size_t)b * C * plane + p;
  for (int c = blockIdx.y; c < C; c += G) {
    const float* xc = xb + (size_t)c * plane;
    float v = 0.f;
    const float a = xc[onw], bq = xc[one], cq = xc[osw], d = xc[ose];
    if (bnw) v += a * nw;
    if (bne) v += bq * ne;
    if (bsw) v += cq * sw;
    if (bse) v += d * se;
    ob[(size_t)c * plane] = v * m;
  }
}

// grad_x must be zero on entry (cleared by zero2_kernel below); grad_flo likewise when G > 1.
__global__ __launch_bounds__(256) void pwc_warp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ flo,
                                                          const float* __restrict__ gout, float* __restrict__ gx,
                                                          float* __restrict__ gflo, int C, int H, int W,
                                                          float mask_thresh, float fs) {
  const long long plane = (long long)H * W;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int b = blockIdx.z, G = gridDim.y;
  const int py = (int)(p / W), px = (int)(p % W);
  const float* fb = flo + (size_t)b * 2 * plane;
  const float ix = warp_coord((float)px, mul_rounded(fb[p], fs), W), iy = warp_coord((float)py, mul_rounded(fb[plane + p], fs), H);
  const WarpTaps t = warp_taps(ix, iy, H, W);
  const float ex = (float)(t.x0 + 1) - ix, ey = (float)(t.y0 + 1) - iy;
  const float nw = ex * ey, ne = t.wx1 * ey, sw = ex * t.wy1, se = t.wx1 * t.wy1;
  const bool bnw = t.vx0 && t.vy0, bne = t.vx1 && t.vy0, bsw = t.vx0 && t.vy1, bse = t.vx1 && t.vy1;
  float msum = 0.f;
  if (bnw) msum += nw;
  if (bne) msum += ne;
  if (bsw) msum += sw;
  if (bse) msum += se;
  if (!(msum >= mask_thresh)) return;  // output * 0: no gradient to either input (buffers are zero)
  const int onw = bnw ? t.y0 * W + t.x0 : 0, one = bne ? t.y0 * W + t.x0 + 1 : 0;
  const int osw = bsw ? (t.y0 + 1) * W + t.x0 : 0, ose = bse ? (t.y0 + 1) * W + t.x0 + 1 : 0;
  const float* xb = x + (size_t)b * C * plane;
  float* gb = gx + (size_t)b * C * plane;
  const float* go = gout + (size_t)b * C * plane + p;
  float gix = 0.f, giy = 0.f;
  for (int c = blockIdx.y; c < C; c += G) {
    const float g = go[(size_t)c * plane];
    const float* xc = xb + (size_t)c * plane;
    float* gc = gb + (size_t)c * plane;
    // grid_sampler_2d_backward: scatter into x, gather the grid gradient from the in-bounds taps.  The four tap
    // loads are unconditional (clamped offsets) so that they are in flight together; only the atomics are predicated.
    const float vnw = xc[onw], vne = xc[one], vsw = xc[osw], vse = xc[ose];
    if (bnw) {
      unsafeAtomicAdd(gc + onw, nw * g);
      gix = tap_fma(gix, vnw, ey, g, true);
      giy = tap_fma(giy, vnw, ex, g, true);
    }
    if (bne) {
      unsafeAtomicAdd(gc + one, ne * g);
      gix = tap_fma(gix, vne, ey, g, false);
      giy = tap_fma(giy, vne, t.wx1, g, true);
    }
    if (bsw) {
      unsafeAtomicAdd(gc + osw, sw * g);
      gix = tap_fma(gix, vsw, t.wy1, g, true);
      giy = tap_fma(giy, vsw, ex, g, false);
    }
    if (bse) {
      unsafeAtomicAdd(gc + ose, se * g);
      gix = tap_fma(gix, vse, t.wy1, g, false);
      giy = tap_fma(giy, vse, t.wx1, g, false);
    }
  }
  // d ix / d grid = W / 2 (unnormalize), d grid / d flo = 2 / max(W - 1, 1) (the reference divides, then doubles)
  const float dfx = mul_rounded(2.0f * ((0.5f * (float)W * gix) / (float)max(W - 1, 1)), fs);   // d (fs flo) / d flo
  const float dfy = mul_rounded(2.0f * ((0.5f * (float)H * giy) / (float)max(H - 1, 1)), fs);
  float* gf = gflo + (size_t)b * 2 * plane;
  if (G == 1) {
    gf[p] = dfx;
    gf[plane + p] = dfy;
  } else {
    unsafeAtomicAdd(gf + p, dfx);
    unsafeAtomicAdd(gf + plane + p, dfy);
  }
}

// Deterministic variant of the scatter: contributions are added as fixed-point int64 (integer adds commute and
// associate, so the order in which the atomics land cannot change the sum: two runs give the same bits), the flow
// gradient's channel groups write their partials side by side.  wfinish converts / adds in index order.
// The fixed point is scaled PER CALL: with m = max|grad_out| (found by the kernel that clears the accumulators: block
// maxima, re-reduced by every consumer block -- a maximum does not depend on the order either) the unit is
// 2^(floor(log2 m) - 40), i.e. every addend keeps 40 bits below the largest gradient of the call (fp32 keeps 24 below
// each value: values down to 1.5e-5 of the maximum are resolved as finely as fp32 resolves them, whatever the
// absolute scale -- AEE / npix-scaled gradients of 1e-9 included), and 2^22 addends of maximal size fit an int64.
constexpr int WARP_FIX_BITS = 40;
constexpr int WARP_BMAX = 4096;   // block maxima of |grad_out| (one per block of the clearing kernel)
// A non-finite grad_out has no fixed-point image (fmaxf drops NaN, __double2ll_rn saturates): the call is flagged instead
// -- the scatter kernels run with scale 0 (their sums are not used) and the finish kernel writes NaN into ALL of grad_x and
// grad_flo, so that the optimiser sees the fault as it would after grid_sample's backward (which poisons only the taps of
// the non-finite pixels: a superset here, never finite garbage).
constexpr int WARP_NONFINITE = -(1 << 20);

__device__ __forceinline__ float block_max_256(float m, float* red) {   // 256 threads, result on every thread
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return m;
}

// 2^shift = the fixed-point scale of this call (uniform over the grid: every block reduces the same block maxima)
__device__ __forceinline__ int warp_fix_shift(const float* __restrict__ bmax, int nblk, float* red) {
  float m = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) m = fmaxf(m, bmax[i]);
  m = block_max_256(m, red);
  if (!(m < 3.0e38f)) return WARP_NONFINITE;   // zero_ll_max_kernel stores +inf for a block that saw Inf / NaN (or > 3e38)
  return m > 0.f ? WARP_FIX_BITS - ilogbf(m) : WARP_FIX_BITS;
}

__device__ __forceinline__ void fix_add(long long* p, float v, double scale) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn((double)v * scale));
}

__global__ __launch_bounds__(256) void pwc_warp_bwd_det_kernel(const float* __restrict__ x, const float* __restrict__ flo,
                                                              const float* __restrict__ gout, long long* __restrict__ gxi,
                                                              float* __restrict__ gfpart, const float* __restrict__ bmax,
                                                              int nblk, int C, int H, int W, float mask_thresh,
                                                              float fs) {
  __shared__ float red[4];
  const double scale = ldexp(1.0, warp_fix_shift(bmax, nblk, red));
  const long long plane = (long long)H * W;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int b = blockIdx.z, G = gridDim.y, B = gridDim.z;
  const int py = (int)(p / W), px = (int)(p % W);
  const float* fb = flo + (size_t)b * 2 * plane;
  float* gf = gfpart + ((size_t)blockIdx.y * B + b) * 2 * plane;   // this channel group's partial flow gradient
  const float ix = warp_coord((float)px, mul_rounded(fb[p], fs), W), iy = warp_coord((float)py, mul_rounded(fb[plane + p], fs), H);
  const WarpTaps t = warp_taps(ix, iy, H, W);
  const float ex = (float)(t.x0 + 1) - ix, ey = (float)(t.y0 + 1) - iy;
  const float nw = ex * ey, ne = t.wx1 * ey, sw = ex * t.wy1, se = t.wx1 * t.wy1;
  const bool bnw = t.vx0 && t.vy0, bne = t.vx1 && t.vy0, bsw = t.vx0 && t.vy1, bse = t.vx1 && t.vy1;
  float msum = 0.f;
  if (bnw) msum += nw;
  if (bne) msum += ne;
  if (bsw) msum += sw;
  if (bse) msum += se;
  if (!(msum >= mask_thresh)) {   // output * 0: no gradient to either input
    gf[p] = 0.f;
    gf[plane + p] = 0.f;
    return;
  }
  const int onw = bnw ? t.y0 * W + t.x0 : 0, one = bne ? t.y0 * W + t.x0 + 1 : 0;
  const int osw = bsw ? (t.y0 + 1) * W + t.x0 : 0, ose = bse ? (t.y0 + 1) * W + t.x0 + 1 : 0;
  const float* xb = x + (size_t)b * C * plane;
  long long* gb = gxi + (size_t)b * C * plane;
  const float* go = gout + (size_t)b * C * plane + p;
  float gix = 0.f, giy = 0.f;
  for (int c = blockIdx.y; c < C; c += G) {
    const float g = go[(size_t)c * plane];
    const float* xc = xb + (size_t)c * plane;
    long long* gc = gb + (size_t)c * plane;
    const float vnw = xc[onw], vne = xc[one], vsw = xc[osw], vse = xc[ose];
    if (bnw) {
      fix_add(gc + onw, nw * g, scale);
      gix = tap_fma(gix, vnw, ey, g, true);
      giy = tap_fma(giy, vnw, ex, g, true);
    }
    if (bne) {
      fix_add(gc + one, ne * g, scale);
      gix = tap_fma(gix, vne, ey, g, false);
      giy = tap_fma(giy, vne, t.wx1, g, true);
    }
    if (bsw) {
      fix_add(gc + osw, sw * g, scale);
      gix = tap_fma(gix, vsw, t.wy1, g, true);
      giy = tap_fma(giy, vsw, ex, g, false);
    }
    if (bse) {
      fix_add(gc + ose, se * g, scale);
      gix = tap_fma(gix, vse, t.wy1, g, false);
      giy = tap_fma(giy, vse, t.wx1, g, false);
    }
  }
  gf[p] = 2.0f * ((0.5f * (float)W * gix) / (float)max(W - 1, 1));
  gf[plane + p] = 2.0f * ((0.5f * (float)H * giy) / (float)max(H - 1, 1));
}

// The same scatter through an LDS window.  The cost of the kernel above is its global atomics, and an atomic instruction
// costs by the cache lines it touches, not by its lanes (profiles/r04_warp_bwd_scatter_ablation.txt: a constant flow runs
// 6-8x faster than a textured one).  Here a workgroup owns a 16 x 16 pixel tile; its taps land in a 32 x 32 texel window
// around the tile centre's target, kept in LDS for WCH channels at a time (ds_add_u64: integers, so any order gives the
// same bits as the kernel above); the window is then flushed with one global atomic per NON-ZERO texel, consecutive lanes
// on consecutive texels of a row (2 cache lines per 32 lanes).  Taps outside the window (a flow that tears the tile
// apart) go to global memory directly, as above.
constexpr int WT = 16, WWIN = 32, WCH = 4;
__global__ __launch_bounds__(256) void pwc_warp_bwd_det_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ flo, const float* __restrict__ gout, long long* __restrict__ gxi,
    float* __restrict__ gfpart, const float* __restrict__ bmax, int nblk, int C, int H, int W, float mask_thresh, float fs,
    int tiles_x) {
  __shared__ float red[4];
  __shared__ unsigned long long win[WCH][WWIN * WWIN];
  __shared__ int s_org[2];
  const double scale = ldexp(1.0, warp_fix_shift(bmax, nblk, red));
  const long long plane = (long long)H * W;
  const int tid = threadIdx.x;
  const int tby = blockIdx.x / tiles_x, tbx = blockIdx.x - tby * tiles_x;
  const int py0 = tby * WT + (tid >> 4), px0 = tbx * WT + (tid & 15);
  const bool inside = py0 < H && px0 < W;
  const int py = min(py0, H - 1), px = min(px0, W - 1);
  const long long p = (long long)py * W + px;
  const int b = blockIdx.z, G = gridDim.y, B = gridDim.z;
  const float* fb = flo + (size_t)b * 2 * plane;
  float* gf = gfpart + ((size_t)blockIdx.y * B + b) * 2 * plane;
  const float ix = warp_coord((float)px, mul_rounded(fb[p], fs), W), iy = warp_coord((float)py, mul_rounded(fb[plane + p], fs), H);
  const WarpTaps t = warp_taps(ix, iy, H, W);
  const float ex = (float)(t.x0 + 1) - ix, ey = (float)(t.y0 + 1) - iy;
  const float nw = ex * ey, ne = t.wx1 * ey, sw = ex * t.wy1, se = t.wx1 * t.wy1;
  const bool bnw = t.vx0 && t.vy0, bne = t.vx1 && t.vy0, bsw = t.vx0 && t.vy1, bse = t.vx1 && t.vy1;
  float msum = 0.f;
  if (bnw) msum += nw;
  if (bne) msum += ne;
  if (bsw) msum += sw;
  if (bse) msum += se;
  const bool act = inside && msum >= mask_thresh;   // else output * 0: no gradient to either input
  if (tid == (WT / 2) * WT + WT / 2) {   // window origin: the centre pixel's target, centred (clamped so that the window
    s_org[0] = min(max(t.y0 - (WWIN - WT) / 2 - WT / 2 + 1, -1), max(H - WWIN + 1, -1));   // overlaps the image where it can)
    s_org[1] = min(max(t.x0 - (WWIN - WT) / 2 - WT / 2 + 1, -1), max(W - WWIN + 1, -1));
  }
  for (int e = tid; e < WCH * WWIN * WWIN; e += 256) (&win[0][0])[e] = 0ull;
  __syncthreads();
  const int wy0 = s_org[0], wx0 = s_org[1];
  // taps as window cells (or -1: outside the window -> global atomic)
  const int ly = t.y0 - wy0, lx = t.x0 - wx0;
  const bool iny0 = ly >= 0 && ly < WWIN, iny1 = ly + 1 >= 0 && ly + 1 < WWIN;
  const bool inx0 = lx >= 0 && lx < WWIN, inx1 = lx + 1 >= 0 && lx + 1 < WWIN;
  const int cnw = (iny0 && inx0) ? ly * WWIN + lx : -1, cne = (iny0 && inx1) ? ly * WWIN + lx + 1 : -1;
  const int csw = (iny1 && inx0) ? (ly + 1) * WWIN + lx : -1, cse = (iny1 && inx1) ? (ly + 1) * WWIN + lx + 1 : -1;
  const int onw = bnw ? t.y0 * W + t.x0 : 0, one = bne ? t.y0 * W + t.x0 + 1 : 0;
  const int osw = bsw ? (t.y0 + 1) * W + t.x0 : 0, ose = bse ? (t.y0 + 1) * W + t.x0 + 1 : 0;
  const float* xb = x + (size_t)b * C * plane;
  long long* gb = gxi + (size_t)b * C * plane;
  const float* go = gout + (size_t)b * C * plane + p;
  float gix = 0.f, giy = 0.f;
  auto put = [&](int j, int cell, long long* gaddr, float v) {
    const unsigned long long q = (unsigned long long)__double2ll_rn((double)v * scale);
    if (cell >= 0) atomicAdd(&win[j][cell], q);
    else atomicAdd(reinterpret_cast<unsigned long long*>(gaddr), q);
  };
  for (int c0 = blockIdx.y; c0 < C; c0 += G * WCH) {   // (workgroup-uniform trip count: barriers inside)
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      const int c = c0 + j * G;
      if (c < C && act) {
        const float g = go[(size_t)c * plane];
        const float* xc = xb + (size_t)c * plane;
        long long* gc = gb + (size_t)c * plane;
        const float vnw = xc[onw], vne = xc[one], vsw = xc[osw], vse = xc[ose];
        if (bnw) {
          put(j, cnw, gc + onw, nw * g);
          gix = tap_fma(gix, vnw, ey, g, true);
          giy = tap_fma(giy, vnw, ex, g, true);
        }
        if (bne) {
          put(j, cne, gc + one, ne * g);
          gix = tap_fma(gix, vne, ey, g, false);
          giy = tap_fma(giy, vne, t.wx1, g, true);
        }
        if (bsw) {
          put(j, csw, gc + osw, sw * g);
          gix = tap_fma(gix, vsw, t.wy1, g, true);
          giy = tap_fma(giy, vsw, ex, g, false);
        }
        if (bse) {
          put(j, cse, gc + ose, se * g);
          gix = tap_fma(gix, vse, t.wy1, g, false);
          giy = tap_fma(giy, vse, t.wx1, g, false);
        }
      }
    }
    __syncthreads();
    // flush (and clear for the next pass): one global atomic per non-zero texel, rows of the window = runs of texels
    for (int e = tid; e < WCH * WWIN * WWIN; e += 256) {
      const int j = e / (WWIN * WWIN), cell = e - j * (WWIN * WWIN);
      const unsigned long long v = win[j][cell];
      if (v != 0ull) {
        win[j][cell] = 0ull;
        const int yy = wy0 + cell / WWIN, xx = wx0 + cell % WWIN;   // (a non-zero cell was hit by a valid tap: inside the image)
        atomicAdd(reinterpret_cast<unsigned long long*>(gb + (size_t)(c0 + j * G) * plane + (size_t)yy * W + xx), v);
      }
    }
    __syncthreads();
  }
  if (inside) {
    gf[p] = act ? 2.0f * ((0.5f * (float)W * gix) / (float)max(W - 1, 1)) : 0.f;
    gf[plane + p] = act ? 2.0f * ((0.5f * (float)H * giy) / (float)max(H - 1, 1)) : 0.f;
  }
}

__global__ __launch_bounds__(256) void pwc_warp_finish_kernel(const long long* __restrict__ gxi,
                                                              const float* __restrict__ gfpart,
                                                              const float* __restrict__ bmax, int nblk,
                                                              float* __restrict__ gx, float* __restrict__ gflo,
                                                              long long nx, long long nf, int G, float fs) {
  __shared__ float red[4];
  const int shift = warp_fix_shift(bmax, nblk, red);
  const long long step = (long long)gridDim.x * blockDim.x;
  if (shift == WARP_NONFINITE) {
    const float qnan = __int_as_float(0x7fc00000);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nx + nf; i += step)
      (i < nx ? gx[i] : gflo[i - nx]) = qnan;
    return;
  }
  const double inv = ldexp(1.0, -shift);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nx + nf; i += step) {
    if (i < nx) {
      gx[i] = (float)((double)gxi[i] * inv);
    } else {
      const long long j = i - nx;
      float s = gfpart[j];
      for (int g = 1; g < G; ++g) s += gfpart[j + (long long)g * nf];   // channel groups in index order
      gflo[j] = mul_rounded(s, fs);   // gradient of the scaled flow times the scale, as autograd's mul backward
    }
  }
}

// clears the accumulators and leaves max|g| of this block's share of grad_out (same element count) in bmax[blockIdx.x]
__global__ __launch_bounds__(256) void zero_ll_max_kernel(long long* __restrict__ a, const float* __restrict__ g,
                                                          float* __restrict__ bmax, long long n) {
  __shared__ float red[4];
  const long long step = (long long)gridDim.x * blockDim.x;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    a[i] = 0;
    const float v = fabsf(g[i]);
    m = (v <= 3.0e38f) ? fmaxf(m, v) : __int_as_float(0x7f800000);   // NaN fails the comparison too: +inf = the flag
  }
  m = block_max_256(m, red);
  if (threadIdx.x == 0) bmax[blockIdx.x] = m;
}

__global__ void zero2_kernel(float* __restrict__ a, long long na, float* __restrict__ b, long long nb) {
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += step) {
    if (i < na) a[i] = 0.f;
    else b[i - na] = 0.f;
  }
}

int channel_groups(long long plane, int C) {
  int g = 1;
  while (plane * g < 65536 && 2 * g <= C / 4 && g < 32) g *= 2;
  return g;
}

}  // namespace

extern "C" int pcfa_pwc_warp_fwd(const float* x, const float* flo, float* out, int B, int C, int H, int W,
                                 float mask_threshold, float flow_scale, void* stream) {
  if (!x || !flo || !out || B < 1 || C < 1 || H < 1 || W < 1) return PCFA_ERR_INVALID_ARG;
  const long long plane = (long long)H * W;
  dim3 grid(pcfa_cdiv(plane, 256), channel_groups(plane, C), B);
  pcfa_launch(pwc_warp_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, flo, out, C, H, W, mask_threshold, flow_scale);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" size_t pcfa_pwc_warp_bwd_det_workspace_bytes(int B, int C, int H, int W) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return 0;
  const long long plane = (long long)H * W;
  return (size_t)B * C * plane * sizeof(long long) + (size_t)channel_groups(plane, C) * B * 2 * plane * sizeof(float) +
         WARP_BMAX * sizeof(float);
}

extern "C" int pcfa_pwc_warp_bwd_det(const float* x, const float* flo, const float* grad_out, float* grad_x,
                                     float* grad_flo, void* workspace, size_t workspace_bytes, int B, int C, int H,
                                     int W, float mask_threshold, float flow_scale, void* stream) {
  if (!x || !flo || !grad_out || !grad_x || !grad_flo || !workspace || B < 1 || C < 1 || H < 1 || W < 1)
    return PCFA_ERR_INVALID_ARG;
  if (workspace_bytes < pcfa_pwc_warp_bwd_det_workspace_bytes(B, C, H, W)) return PCFA_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 7) return PCFA_ERR_INVALID_ARG;
  const long long plane = (long long)H * W;
  hipStream_t s = (hipStream_t)stream;
  const long long nx = (long long)B * C * plane, nf = (long long)B * 2 * plane;
  const int G = channel_groups(plane, C);
  long long* gxi = (long long*)workspace;
  float* gfpart = (float*)(gxi + nx);
  float* bmax = gfpart + (size_t)G * nf;
  const int nblk = (int)min((nx + 255) / 256, (long long)WARP_BMAX);
  pcfa_launch(zero_ll_max_kernel, dim3(nblk), dim3(256), 0, s, gxi, grad_out, bmax, nx);
  PCFA_LAUNCH_CHECK();
  // PCFA_WARP_SCATTER=global: one global atomic per tap (the r03 kernel; dev A/B, read once)
  static const bool lds_window = !(getenv("PCFA_WARP_SCATTER") && getenv("PCFA_WARP_SCATTER")[0] == 'g');
  if (lds_window && plane >= 256) {   // (tiny planes: the window's clear / flush passes cost more than they save)
    const int tiles_x = pcfa_cdiv(W, WT), tiles_y = pcfa_cdiv(H, WT);
    dim3 grid((unsigned)(tiles_x * tiles_y), G, B);
    pcfa_launch(pwc_warp_bwd_det_lds_kernel, grid, dim3(256), 0, s, x, flo, grad_out, gxi, gfpart, (const float*)bmax, nblk,
                C, H, W, mask_threshold, flow_scale, tiles_x);
  } else {
    dim3 grid(pcfa_cdiv(plane, 256), G, B);
    pcfa_launch(pwc_warp_bwd_det_kernel, grid, dim3(256), 0, s, x, flo, grad_out, gxi, gfpart, (const float*)bmax, nblk, C,
                H, W, mask_threshold, flow_scale);
  }
  PCFA_LAUNCH_CHECK();
  pcfa_launch(pwc_warp_finish_kernel, dim3((int)min((nx + nf + 255) / 256, 4096LL)), dim3(256), 0, s,
              (const long long*)gxi, (const float*)gfpart, (const float*)bmax, nblk, grad_x, grad_flo, nx, nf, G, flow_scale);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_pwc_warp_bwd(const float* x, const float* flo, const float* grad_out, float* grad_x,
                                 float* grad_flo, int B, int C, int H, int W, float mask_threshold, float flow_scale,
                                 void* stream) {
  if (!x || !flo || !grad_out || !grad_x || !grad_flo || B < 1 || C < 1 || H < 1 || W < 1)
    return PCFA_ERR_INVALID_ARG;
  const long long plane = (long long)H * W;
  hipStream_t s = (hipStream_t)stream;
  const long long na = (long long)B * C * plane, nb = (long long)B * 2 * plane;
  long long zb = (na + nb + 255) / 256;
  if (zb > 4096) zb = 4096;
  pcfa_launch(zero2_kernel, dim3((int)zb), dim3(256), 0, s, grad_x, na, grad_flo, nb);
  PCFA_LAUNCH_CHECK();
  dim3 grid(pcfa_cdiv(plane, 256), channel_groups(plane, C), B);
  pcfa_launch(pwc_warp_bwd_kernel, grid, dim3(256), 0, s, x, flo, grad_out, grad_x, grad_flo, C, H, W,
              mask_threshold, flow_scale);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
