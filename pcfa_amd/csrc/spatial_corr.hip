// PWC-Net cost volume: spatial correlation sampler forward/backward for gfx950.
//
// Replaces spatial_correlation_sampler_backend.forward/backward (reference
// models/PWCNet/cpu_spatial_correlation_sampler-0.3.0/Correlation_Module/
// correlation.cpp:9-37,39-73,75-124,126-178; binding correlation_sampler.cpp:58-112).
//
//   out[b][ph][pw][h][w] = sum_{c,i,j} in1[b][c][u+i*dil][v+j*dil] *
//                                      in2[b][c][u+i*dil+sU][v+j*dil+sV]
//   u = h*dH - padH, v = w*dW - padW, sU = (ph - (patchH-1)/2)*dil_patchH (same for V),
//   terms with either operand outside the image dropped.
//
// Fast path (what PWC-Net uses: k=1, stride 1, pad 0, patch PxP, dil_patch 1):
//   forward : workgroup = 8x32 output pixels, wave z = patch row; a thread owns
//             4 consecutive pixels x P shifts in registers and streams channel
//             chunks of both feature maps through LDS (in2 tile carries the halo;
//             16-B pieces, next chunk prefetched into registers, two LDS stages),
//             so every input byte is read from HBM once per tile and written
//             outputs are 16-B vectors.
//   backward: both gradients are gathers with the same shape,
//               gin1[c][p] = sum_d g[d][p]   * in2[c][p+d]
//               gin2[c][p] = sum_d g[d][p-d] * in1[c][p-d]
//             a thread keeps its P*P gradient taps in registers and walks the
//             channels through LDS -- no atomics, bitwise reproducible (the CPU
//             reference accumulates serially, correlation.cpp:148).
// Every other parameter set takes the generic one-thread-per-element kernels.
#include "common.hpp"

namespace {

constexpr int TH = 8, TW = 32;  // output-pixel tile of the fast path
constexpr int CC = 8;           // channels per LDS chunk (backward)
constexpr int FC = 16;          // channels per LDS chunk (forward: half as many latency-bound round trips)

// Preconditions (checked by the host): W % 4 == 0 and 16-B aligned in1 / in2, so the tiles are staged in 16-B
// pieces that lie either inside or outside the image.  Two LDS stages: the next channel chunk is fetched into
// registers before the FMAs of the current one and written to the other stage after them (one barrier per chunk;
// the coarse pyramid levels are a single workgroup walking up to 25 chunks, i.e. a pure latency chain).
// THT = tile rows: 8 for the fine levels; 2 for the coarse ones, where 8-row tiles leave 1-30 workgroups walking a
// serial channel loop on as many CUs (LDS-read bound): four times as many, four times lighter workgroups.
template <int PS, int THT>
__global__ __launch_bounds__(8 * THT * PS) void scorr_fwd_fast_kernel(
    const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out, int C,
    int H, int W) {
  constexpr int R = (PS - 1) / 2;
  constexpr int HW2 = TW + 2 * R;  // in2 tile width  (40 for PS=9)
  constexpr int HH2 = THT + 2 * R;  // in2 tile height (16)
  static_assert(R % 4 == 0 && HW2 % 4 == 0, "halo must keep the 16-B pieces aligned");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float s1[2][FC][THT][TW];
  __shared__ __attribute__((aligned(16))) float s2[2][FC][HH2][HW2];

  const int b = blockIdx.z;
  const int y0 = blockIdx.y * THT, x0 = blockIdx.x * TW;
  const int tx = threadIdx.x, ty = threadIdx.y, tz = threadIdx.z;  // quad, row, patch row
  const int tid = tx + 8 * ty + 8 * THT * tz;
  constexpr int NT = 8 * THT * PS;
  const size_t plane = (size_t)H * W;
  const float* p1 = in1 + (size_t)b * C * plane;
  const float* p2 = in2 + (size_t)b * C * plane;

  // staging plan (fixed per thread): pieces of the in1 tile and of the in2 tile with its halo
  constexpr int N1 = FC * THT * (TW / 4), N2 = FC * HH2 * (HW2 / 4);
  constexpr int S1 = (N1 + NT - 1) / NT, S2 = (N2 + NT - 1) / NT;
  int o1[S1], d1[S1], o2[S2], d2[S2];  // global offset inside the chunk (-1: outside), LDS float index (-1: none)
#pragma unroll
  for (int k = 0; k < S1; ++k) {
    const int e = tid + k * NT;
    const int c = e / (THT * (TW / 4)), r = (e / (TW / 4)) % THT, m = e % (TW / 4);
    const int gy = y0 + r, gx = x0 + 4 * m;
    o1[k] = (e < N1 && gy < H && gx < W) ? (int)(c * plane) + gy * W + gx : -1;
    d1[k] = e < N1 ? (c * THT + r) * TW + 4 * m : -1;
  }
#pragma unroll
  for (int k = 0; k < S2; ++k) {
    const int e = tid + k * NT;
    const int c = e / (HH2 * (HW2 / 4)), r = (e / (HW2 / 4)) % HH2, m = e % (HW2 / 4);
    const int gy = y0 + r - R, gx = x0 + 4 * m - R;
    o2[k] = (e < N2 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (int)(c * plane) + gy * W + gx : -1;
    d2[k] = e < N2 ? (c * HH2 + r) * HW2 + 4 * m : -1;
  }

  float acc[4][PS];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int d = 0; d < PS; ++d) acc[p][d] = 0.f;

  f32x4 r1[S1], r2[S2];
  auto fetch = [&](int c0) {  // branch-free: a dead piece reads offset 0 of the chunk and is zeroed
    const int climit = (int)((size_t)(C - c0) * plane);
    const float* b1 = p1 + (size_t)c0 * plane;
    const float* b2 = p2 + (size_t)c0 * plane;
#pragma unroll
    for (int k = 0; k < S1; ++k) {
      const bool ok = o1[k] >= 0 && o1[k] < climit;
      const f32x4 t = *reinterpret_cast<const f32x4*>(b1 + (ok ? o1[k] : 0));
      r1[k] = ok ? t : (f32x4)(0.f);
    }
#pragma unroll
    for (int k = 0; k < S2; ++k) {
      const bool ok = o2[k] >= 0 && o2[k] < climit;
      const f32x4 t = *reinterpret_cast<const f32x4*>(b2 + (ok ? o2[k] : 0));
      r2[k] = ok ? t : (f32x4)(0.f);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int k = 0; k < S1; ++k)
      if (d1[k] >= 0) *reinterpret_cast<f32x4*>(&s1[buf][0][0][0] + d1[k]) = r1[k];
#pragma unroll
    for (int k = 0; k < S2; ++k)
      if (d2[k] >= 0) *reinterpret_cast<f32x4*>(&s2[buf][0][0][0] + d2[k]) = r2[k];
  };

  fetch(0);
  commit(0);
  __syncthreads();
  int cur = 0;
  for (int c0 = 0; c0 < C; c0 += FC) {
    const bool more = c0 + FC < C;
    if (more) fetch(c0 + FC);
#pragma unroll 2
    for (int c = 0; c < FC; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(&s1[cur][c][ty][4 * tx]);
      float v2[4 + 2 * R];
#pragma unroll
      for (int k = 0; k < (4 + 2 * R) / 4; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(&s2[cur][c][ty + tz][4 * tx + 4 * k]);
        v2[4 * k] = t.x; v2[4 * k + 1] = t.y; v2[4 * k + 2] = t.z; v2[4 * k + 3] = t.w;
      }
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int d = 0; d < PS; ++d) acc[p][d] += av[p] * v2[p + d];
    }
    if (more) commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  const int gy = y0 + ty, gx = x0 + 4 * tx;
  if (gy >= H || gx >= W) return;
  float* o = out + (((size_t)b * PS + tz) * PS) * plane + (size_t)gy * W + gx;
  const bool vec = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
#pragma unroll
  for (int d = 0; d < PS; ++d) {
    float* od = o + (size_t)d * plane;
    if (vec) {
      *reinterpret_cast<float4*>(od) = make_float4(acc[0][d], acc[1][d], acc[2][d], acc[3][d]);
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) od[p] = acc[p][d];
    }
  }
}

// gin[c][p] = sum_{dy,dx} G(dy,dx) * X[c][p + SIGN*(dy,dx)]
//   SIGN=+1: G = g[d][p]       X = in2   (gradient w.r.t. in1)
//   SIGN=-1: G = g[d][p - d]   X = in1   (gradient w.r.t. in2)
template <int PS, int SIGN>
__global__ __launch_bounds__(TH* TW) void scorr_bwd_fast_kernel(
    const float* __restrict__ X, const float* __restrict__ gout, float* __restrict__ gin, int C,
    int H, int W) {
  constexpr int R = (PS - 1) / 2;
  constexpr int HW2 = TW + 2 * R, HH2 = TH + 2 * R;
  constexpr int S2 = HW2 + 1;
  __shared__ float sx[CC][HH2][S2];

  const int ngroups = (C + CC - 1) / CC;
  const int b = blockIdx.z / ngroups;
  const int c0 = (blockIdx.z - b * ngroups) * CC;
  const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int lx = threadIdx.x, ly = threadIdx.y;
  const int tid = lx + TW * ly;
  constexpr int NT = TH * TW;
  const int gy = y0 + ly, gx = x0 + lx;
  const bool inside = gy < H && gx < W;
  const size_t plane = (size_t)H * W;

  // Every load below is branch-free (clamped address, value zeroed afterwards) and the staging loop issues a batch
  // of loads before its LDS writes: with predicated loads the 81 gradient taps and the 20 staging iterations were
  // each a dependent global round trip, and the kernel was a 17-20 us latency chain.
  float G[PS][PS];
  const float* gb = gout + (size_t)b * PS * PS * plane;
  const int cgy = min(gy, H - 1), cgx = min(gx, W - 1);
#pragma unroll
  for (int i = 0; i < PS; ++i)
#pragma unroll
    for (int j = 0; j < PS; ++j) {
      const int sy = (SIGN > 0) ? cgy : cgy - (i - R), sx_ = (SIGN > 0) ? cgx : cgx - (j - R);
      const bool ok = inside && sy >= 0 && sy < H && sx_ >= 0 && sx_ < W;
      const float v = gb[(size_t)(i * PS + j) * plane + (ok ? (size_t)sy * W + sx_ : 0)];
      G[i][j] = ok ? v : 0.f;
    }

  // The channels are independent in both gradients, so blockIdx.z also splits them: a workgroup stages
  // ONE chunk of CC channels (no serial channel loop -- the small pyramid levels have only 1-4 pixel tiles).
  const float* px = X + (size_t)b * C * plane;
  float* po = gin + (size_t)b * C * plane;
  {
    constexpr int NE = CC * HH2 * HW2, BATCH = 10;
    static_assert(NE % (NT * BATCH) == 0, "staging batches must divide evenly");
#pragma unroll 1
    for (int e0 = tid; e0 < NE; e0 += NT * BATCH) {
      float t[BATCH];
      bool ok[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int e = e0 + k * NT;
        const int c = e / (HH2 * HW2), r = (e / HW2) % HH2, x = e % HW2;
        const int yy = y0 + r - R, xx = x0 + x - R;
        ok[k] = c0 + c < C && yy >= 0 && yy < H && xx >= 0 && xx < W;
        t[k] = px[ok[k] ? (size_t)(c0 + c) * plane + (size_t)yy * W + xx : 0];
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int e = e0 + k * NT;
        const int c = e / (HH2 * HW2), r = (e / HW2) % HH2, x = e % HW2;
        sx[c][r][x] = ok[k] ? t[k] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < CC; ++c) {
      if (c0 + c >= C) break;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < PS; ++i)
#pragma unroll
        for (int j = 0; j < PS; ++j) {
          const int r = (SIGN > 0) ? (ly + i) : (ly + 2 * R - i);
          const int x = (SIGN > 0) ? (lx + j) : (lx + 2 * R - j);
          s += G[i][j] * sx[c][r][x];
        }
      if (inside) po[(size_t)(c0 + c) * plane + (size_t)gy * W + gx] = s;
    }
  }
}

struct ScParams {
  int B, C, iH, iW, oH, oW;
  int kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW;
};

__global__ void scorr_fwd_generic_kernel(const float* __restrict__ in1,
                                         const float* __restrict__ in2, float* __restrict__ out,
                                         ScParams p) {
  const long long total = (long long)p.B * p.patchH * p.patchW * p.oH * p.oW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    long long t = idx;
    const int w = t % p.oW; t /= p.oW;
    const int h = t % p.oH; t /= p.oH;
    const int pw = t % p.patchW; t /= p.patchW;
    const int ph = t % p.patchH; t /= p.patchH;
    const int b = (int)t;
    const int u = -p.padH + h * p.dH, v = -p.padW + w * p.dW;
    const int sU = (ph - (p.patchH - 1) / 2) * p.dpH, sV = (pw - (p.patchW - 1) / 2) * p.dpW;
    const size_t plane = (size_t)p.iH * p.iW;
    const float* a = in1 + (size_t)b * p.C * plane;
    const float* c2 = in2 + (size_t)b * p.C * plane;
    float s = 0.f;
    for (int c = 0; c < p.C; ++c)
      for (int i = 0; i < p.kH; ++i) {
        const int i1 = u + i * p.dilH, i2 = i1 + sU;
        if (i1 < 0 || i1 >= p.iH || i2 < 0 || i2 >= p.iH) continue;
        for (int j = 0; j < p.kW; ++j) {
          const int j1 = v + j * p.dilW, j2 = j1 + sV;
          if (j1 < 0 || j1 >= p.iW || j2 < 0 || j2 >= p.iW) continue;
          s += a[c * plane + (size_t)i1 * p.iW + j1] * c2[c * plane + (size_t)i2 * p.iW + j2];
        }
      }
    out[idx] = s;
  }
}

// WHICH = 1: gradient w.r.t. in1 (other = in2), WHICH = 2: w.r.t. in2 (other = in1).
template <int WHICH>
__global__ void scorr_bwd_generic_kernel(const float* __restrict__ other,
                                         const float* __restrict__ gout, float* __restrict__ gin,
                                         ScParams p) {
  const long long total = (long long)p.B * p.C * p.iH * p.iW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const size_t plane = (size_t)p.iH * p.iW, oplane = (size_t)p.oH * p.oW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    long long t = idx;
    const int x = t % p.iW; t /= p.iW;
    const int y = t % p.iH; t /= p.iH;
    const int c = t % p.C; t /= p.C;
    const int b = (int)t;
    const float* oc = other + ((size_t)b * p.C + c) * plane;
    const float* gb = gout + (size_t)b * p.patchH * p.patchW * oplane;
    float s = 0.f;
    for (int ph = 0; ph < p.patchH; ++ph) {
      const int sU = (ph - (p.patchH - 1) / 2) * p.dpH;
      for (int pw = 0; pw < p.patchW; ++pw) {
        const int sV = (pw - (p.patchW - 1) / 2) * p.dpW;
        for (int i = 0; i < p.kH; ++i) {
          // position in in1 (i1) and in2 (i2) for this tap
          const int i1 = (WHICH == 1) ? y : y - sU;
          const int i2 = i1 + sU;
          if (i1 < 0 || i1 >= p.iH || i2 < 0 || i2 >= p.iH) continue;
          const int hn = i1 + p.padH - i * p.dilH;
          if (hn < 0 || hn % p.dH != 0) continue;
          const int h = hn / p.dH;
          if (h >= p.oH) continue;
          for (int j = 0; j < p.kW; ++j) {
            const int j1 = (WHICH == 1) ? x : x - sV;
            const int j2 = j1 + sV;
            if (j1 < 0 || j1 >= p.iW || j2 < 0 || j2 >= p.iW) continue;
            const int wn = j1 + p.padW - j * p.dilW;
            if (wn < 0 || wn % p.dW != 0) continue;
            const int w = wn / p.dW;
            if (w >= p.oW) continue;
            const float g = gb[(size_t)(ph * p.patchW + pw) * oplane + (size_t)h * p.oW + w];
            const float o = (WHICH == 1) ? oc[(size_t)i2 * p.iW + j2] : oc[(size_t)i1 * p.iW + j1];
            s += g * o;
          }
        }
      }
    }
    gin[idx] = s;
  }
}

bool make_params(ScParams& p, int B, int C, int iH, int iW, int kH, int kW, int patchH,
                 int patchW, int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH,
                 int dW) {
  if (B < 1 || C < 1 || iH < 1 || iW < 1 || kH < 1 || kW < 1 || patchH < 1 || patchW < 1 ||
      padH < 0 || padW < 0 || dilH < 1 || dilW < 1 || dpH < 1 || dpW < 1 || dH < 1 || dW < 1)
    return false;
  const int dkH = (kH - 1) * dilH + 1, dkW = (kW - 1) * dilW + 1;
  const int nH = iH + 2 * padH - dkH, nW = iW + 2 * padW - dkW;
  if (nH < 0 || nW < 0) return false;
  p = ScParams{B, C, iH, iW, nH / dH + 1, nW / dW + 1, kH, kW, patchH, patchW, padH, padW,
               dilH, dilW, dpH, dpW, dH, dW};
  return true;
}

bool is_fast(const ScParams& p, int ps) {
  return p.kH == 1 && p.kW == 1 && p.dH == 1 && p.dW == 1 && p.padH == 0 && p.padW == 0 &&
         p.dpH == 1 && p.dpW == 1 && p.patchH == ps && p.patchW == ps;
}

}  // namespace

extern "C" int pcfa_spatial_corr_out_size(int iH, int iW, int kH, int kW, int padH, int padW,
                                          int dilH, int dilW, int dH, int dW, int* oH, int* oW) {
  ScParams p;
  if (!make_params(p, 1, 1, iH, iW, kH, kW, 1, 1, padH, padW, dilH, dilW, 1, 1, dH, dW))
    return PCFA_ERR_INVALID_ARG;
  if (oH) *oH = p.oH;
  if (oW) *oW = p.oW;
  return PCFA_OK;
}

extern "C" int pcfa_spatial_corr_fwd(const float* in1, const float* in2, float* out, int B, int C,
                                     int iH, int iW, int kH, int kW, int patchH, int patchW,
                                     int padH, int padW, int dilH, int dilW, int dil_patchH,
                                     int dil_patchW, int dH, int dW, void* stream) {
  ScParams p;
  if (!in1 || !in2 || !out ||
      !make_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH,
                   dil_patchW, dH, dW))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = iW % 4 == 0 && ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2)) & 15) == 0;
  if (is_fast(p, 9) && aligned) {
    if ((long long)iH * iW <= 48 * 160) {  // coarse levels: 2-row tiles
      dim3 grid(pcfa_cdiv(iW, TW), pcfa_cdiv(iH, 2), B), block(8, 2, 9);
      pcfa_launch(scorr_fwd_fast_kernel<9, 2>, grid, block, 0, s, in1, in2, out, C, iH, iW);
    } else {
      dim3 grid(pcfa_cdiv(iW, TW), pcfa_cdiv(iH, TH), B), block(8, 8, 9);
      pcfa_launch(scorr_fwd_fast_kernel<9, TH>, grid, block, 0, s, in1, in2, out, C, iH, iW);
    }
  } else {
    const long long total = (long long)B * patchH * patchW * p.oH * p.oW;
    const int blocks = (int)((total + 255) / 256 < 65535LL * 16 ? (total + 255) / 256 : 65535LL * 16);
    pcfa_launch(scorr_fwd_generic_kernel, dim3(blocks), dim3(256), 0, s, in1, in2, out, p);
  }
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_spatial_corr_bwd(const float* in1, const float* in2, const float* grad_out,
                                     float* grad_in1, float* grad_in2, int B, int C, int iH,
                                     int iW, int kH, int kW, int patchH, int patchW, int padH,
                                     int padW, int dilH, int dilW, int dil_patchH, int dil_patchW,
                                     int dH, int dW, void* stream) {
  ScParams p;
  if (!in1 || !in2 || !grad_out || !grad_in1 || !grad_in2 ||
      !make_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH,
                   dil_patchW, dH, dW))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (is_fast(p, 9)) {
    dim3 grid(pcfa_cdiv(iW, TW), pcfa_cdiv(iH, TH), B * pcfa_cdiv(C, CC)), block(TW, TH, 1);
    pcfa_launch(scorr_bwd_fast_kernel<9, +1>, grid, block, 0, s, in2, grad_out, grad_in1,
                       C, iH, iW);
    PCFA_LAUNCH_CHECK();
    pcfa_launch(scorr_bwd_fast_kernel<9, -1>, grid, block, 0, s, in1, grad_out, grad_in2,
                       C, iH, iW);
  } else {
    const long long total = (long long)B * C * iH * iW;
    const int blocks = (int)((total + 255) / 256 < 65535LL * 16 ? (total + 255) / 256 : 65535LL * 16);
    pcfa_launch(scorr_bwd_generic_kernel<1>, dim3(blocks), dim3(256), 0, s, in2, grad_out,
                       grad_in1, p);
    PCFA_LAUNCH_CHECK();
    pcfa_launch(scorr_bwd_generic_kernel<2>, dim3(blocks), dim3(256), 0, s, in1, grad_out,
                       grad_in2, p);
  }
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
