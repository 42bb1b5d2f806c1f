// RAFT/GMA correlation-pyramid lookup, forward and backward, for gfx950.
//
// Replaces CorrBlock.__call__ (reference models/raft/corr.py:29-50) and the
// grid_sample inside bilinear_sampler (models/raft/utils/utils.py:57-71).
//
// Work decomposition (wave64): one workgroup = 64 consecutive queries x one
// pyramid level, (2r+1) waves.  All (2r+1)^2 taps of a (query, level) share
// one fractional offset, so the workgroup
//   0. derives each query's window origin + fractions from coords,
//   A. gathers the (2r+2)^2 texel windows into LDS with lanes running ALONG the
//      window rows (7 row segments per wave-load instead of 64 scattered lines),
//   B. lets thread (query, b) blend two LDS rows into the 2r+1 taps of row b and
//      store them with lanes running along the query index (256-B stores).
// The backward is the exact transpose with the same ownership, so the
// read-modify-write of dpyr needs no atomics and is deterministic.
#include "common.hpp"

namespace {

constexpr int QB = 64;  // queries per workgroup

template <int R>
struct LookupShared {
  static constexpr int N1 = 2 * R + 1;
  static constexpr int WIN = 2 * R + 2;
  static constexpr int WSTRIDE = WIN * WIN + 1;  // odd -> conflict-free per-query stride
  static constexpr int GSTRIDE = N1 * N1 + ((N1 * N1) % 2 == 0 ? 1 : 0);
  int x0[QB];
  int y0[QB];
  float fx[QB];
  float fy[QB];
};

template <int R>
__device__ __forceinline__ void lookup_origins(const float* __restrict__ coords, int b_img, int Q,
                                               int q0, int level, int* sx0, int* sy0, float* sfx,
                                               float* sfy) {
  const int t = threadIdx.x;
  if (threadIdx.y == 0) {
    int q = q0 + t;
    float cx = 0.f, cy = 0.f;
    if (q < Q) {
      cx = coords[((size_t)b_img * 2 + 0) * Q + q];
      cy = coords[((size_t)b_img * 2 + 1) * Q + q];
    }
    // reference: coords / 2**i  (exact power-of-two scaling)
    const float inv = 1.0f / (float)(1 << level);
    const float xl = cx * inv, yl = cy * inv;
    const float flx = floorf(xl), fly = floorf(yl);
    sfx[t] = xl - flx;
    sfy[t] = yl - fly;
    sx0[t] = (int)fminf(fmaxf(flx, -1.0e8f), 1.0e8f) - R;
    sy0[t] = (int)fminf(fmaxf(fly, -1.0e8f), 1.0e8f) - R;
  }
}

template <int R>
__global__ __launch_bounds__(QB*(2 * R + 1)) void corr_lookup_fwd_kernel(
    const float* __restrict__ pyr, const float* __restrict__ coords, float* __restrict__ out,
    int Q, PyrLayout P) {
  using S = LookupShared<R>;
  constexpr int N1 = S::N1, WIN = S::WIN, WS = S::WSTRIDE;
  __shared__ int s_x0[QB], s_y0[QB];
  __shared__ float s_fx[QB], s_fy[QB];
  __shared__ float s_win[QB * WS];

  const int level = blockIdx.y;
  const int b_img = blockIdx.z;
  const int q0 = blockIdx.x * QB;
  const int hl = P.h[level], wl = P.w[level], off = P.off[level];
  const int tid = threadIdx.y * QB + threadIdx.x;
  constexpr int NT = QB * N1;

  lookup_origins<R>(coords, b_img, Q, q0, level, s_x0, s_y0, s_fx, s_fy);
  __syncthreads();

  // Phase A: window gather, lanes along (row, col) of the window.
  const float* base = pyr + (size_t)b_img * Q * P.slab + off;
  for (int e = tid; e < QB * WIN * WIN; e += NT) {
    const int ql = e / (WIN * WIN);
    const int rc = e - ql * (WIN * WIN);
    const int r = rc / WIN;
    const int c = rc - r * WIN;
    const int q = q0 + ql;
    const int x = s_x0[ql] + c;
    const int y = s_y0[ql] + r;
    float v = 0.f;
    if (q < Q && x >= 0 && x < wl && y >= 0 && y < hl)
      v = base[(size_t)q * P.slab + (size_t)y * wl + x];
    s_win[ql * WS + rc] = v;
  }
  __syncthreads();

  // Phase B: thread (query, b) produces the 2r+1 taps a = 0..2r of window row b.
  const int ql = threadIdx.x, b = threadIdx.y;
  const int q = q0 + ql;
  if (q >= Q) return;
  const float fx = s_fx[ql], fy = s_fy[ql];
  const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy);
  const float w10 = (1.f - fx) * fy, w11 = fx * fy;
  float t0[WIN], t1[WIN];
#pragma unroll
  for (int c = 0; c < WIN; ++c) {
    t0[c] = s_win[ql * WS + b * WIN + c];
    t1[c] = s_win[ql * WS + (b + 1) * WIN + c];
  }
  const int C = P.L * N1 * N1;
  float* o = out + ((size_t)b_img * C + (size_t)level * N1 * N1 + b) * Q + q;
#pragma unroll
  for (int a = 0; a < N1; ++a) {
    const float v = t0[a] * w00 + t0[a + 1] * w01 + t1[a] * w10 + t1[a + 1] * w11;
    o[(size_t)a * N1 * Q] = v;
  }
}

template <int R>
__global__ __launch_bounds__(QB*(2 * R + 1)) void corr_lookup_bwd_kernel(
    float* __restrict__ dpyr, const float* __restrict__ coords,
    const float* __restrict__ grad_out, int Q, PyrLayout P) {
  using S = LookupShared<R>;
  constexpr int N1 = S::N1, WIN = S::WIN, GS = S::GSTRIDE;
  __shared__ int s_x0[QB], s_y0[QB];
  __shared__ float s_fx[QB], s_fy[QB];
  __shared__ float s_g[QB * GS];

  const int level = blockIdx.y;
  const int b_img = blockIdx.z;
  const int q0 = blockIdx.x * QB;
  const int hl = P.h[level], wl = P.w[level], off = P.off[level];
  const int tid = threadIdx.y * QB + threadIdx.x;
  constexpr int NT = QB * N1;

  lookup_origins<R>(coords, b_img, Q, q0, level, s_x0, s_y0, s_fx, s_fy);

  // Phase A': coalesced read of the (2r+1)^2 tap gradients of every query.
  {
    const int ql = threadIdx.x, b = threadIdx.y;
    const int q = q0 + ql;
    const int C = P.L * N1 * N1;
    const float* g = grad_out + ((size_t)b_img * C + (size_t)level * N1 * N1 + b) * Q + q;
#pragma unroll
    for (int a = 0; a < N1; ++a) s_g[ql * GS + b * N1 + a] = (q < Q) ? g[(size_t)a * N1 * Q] : 0.f;
  }
  __syncthreads();

  // Phase B': every window texel gathers its <= 4 taps, then owns its RMW.
  float* base = dpyr + (size_t)b_img * Q * P.slab + off;
  for (int e = tid; e < QB * WIN * WIN; e += NT) {
    const int ql = e / (WIN * WIN);
    const int rc = e - ql * (WIN * WIN);
    const int r = rc / WIN;
    const int c = rc - r * WIN;
    const int q = q0 + ql;
    const int x = s_x0[ql] + c;
    const int y = s_y0[ql] + r;
    if (!(q < Q && x >= 0 && x < wl && y >= 0 && y < hl)) continue;
    const float fx = s_fx[ql], fy = s_fy[ql];
    const float* g = s_g + ql * GS;
    float acc = 0.f;
    // tap (a, b) touches texels (r, c) in {b, b+1} x {a, a+1}
    if (r < N1) {
      const float wy = 1.f - fy;
      if (c < N1) acc += g[r * N1 + c] * ((1.f - fx) * wy);
      if (c > 0) acc += g[r * N1 + c - 1] * (fx * wy);
    }
    if (r > 0) {
      const float wy = fy;
      if (c < N1) acc += g[(r - 1) * N1 + c] * ((1.f - fx) * wy);
      if (c > 0) acc += g[(r - 1) * N1 + c - 1] * (fx * wy);
    }
    float* p = base + (size_t)q * P.slab + (size_t)y * wl + x;
    *p += acc;
  }
}

template <int R>
int launch_fwd(const float* pyr, const float* coords, float* out, int B, int Q,
               const PyrLayout& P, hipStream_t s) {
  dim3 grid(pcfa_cdiv(Q, QB), P.L, B), block(QB, 2 * R + 1, 1);
  hipLaunchKernelGGL(corr_lookup_fwd_kernel<R>, grid, block, 0, s, pyr, coords, out, Q, P);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
template <int R>
int launch_bwd(float* dpyr, const float* coords, const float* go, int B, int Q,
               const PyrLayout& P, hipStream_t s) {
  dim3 grid(pcfa_cdiv(Q, QB), P.L, B), block(QB, 2 * R + 1, 1);
  hipLaunchKernelGGL(corr_lookup_bwd_kernel<R>, grid, block, 0, s, dpyr, coords, go, Q, P);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

bool check_levels(const PyrLayout& P) {
  for (int l = 0; l < P.L; ++l)
    if (P.h[l] < 1 || P.w[l] < 1) return false;
  return true;
}

}  // namespace

extern "C" int pcfa_corr_lookup_fwd(const float* pyr, const float* coords, float* out, int B,
                                    int H, int W, int num_levels, int radius, void* stream) {
  PyrLayout P;
  if (!pyr || !coords || !out || B < 1 || !pcfa_make_layout(P, H, W, num_levels) ||
      !check_levels(P))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int Q = H * W;
  switch (radius) {
    case 1: return launch_fwd<1>(pyr, coords, out, B, Q, P, s);
    case 2: return launch_fwd<2>(pyr, coords, out, B, Q, P, s);
    case 3: return launch_fwd<3>(pyr, coords, out, B, Q, P, s);
    case 4: return launch_fwd<4>(pyr, coords, out, B, Q, P, s);
    default: return PCFA_ERR_UNSUPPORTED;
  }
}

extern "C" int pcfa_corr_lookup_bwd(float* dpyr, const float* coords, const float* grad_out,
                                    int B, int H, int W, int num_levels, int radius,
                                    void* stream) {
  PyrLayout P;
  if (!dpyr || !coords || !grad_out || B < 1 || !pcfa_make_layout(P, H, W, num_levels) ||
      !check_levels(P))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int Q = H * W;
  switch (radius) {
    case 1: return launch_bwd<1>(dpyr, coords, grad_out, B, Q, P, s);
    case 2: return launch_bwd<2>(dpyr, coords, grad_out, B, Q, P, s);
    case 3: return launch_bwd<3>(dpyr, coords, grad_out, B, Q, P, s);
    case 4: return launch_bwd<4>(dpyr, coords, grad_out, B, Q, P, s);
    default: return PCFA_ERR_UNSUPPORTED;
  }
}
