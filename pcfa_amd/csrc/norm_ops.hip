// InstanceNorm2d (no affine parameters, batch statistics) fused with the ReLU that follows it, and the residual
// relu(x + y), for the RAFT / GMA feature encoder.
//
// Replaces, per normalisation layer of models/raft/extractor.py:6-58,118-157 (norm_fn='instance': 15 layers per
// encoder pass, both images batched), the library's three forward passes (batch_norm_collect_statistics,
// batch_norm_transform_input, clamp_min) and two backward passes (relu backward, batch_norm_backward with ONE
// workgroup per plane: 128 workgroups on 256 CUs at the 64-channel stage) by two launches each:
//   stats : grid (chunks, planes) -- per-chunk partial sums in fp64, one pair per workgroup, no atomics;
//   apply : grid (chunks, planes) -- every workgroup re-adds its plane's partials in index order (bitwise
//           reproducible, no finalize launch), then streams its chunk:
//             forward   y  = relu?((x - mean) * rstd)                      rstd = 1/sqrt(var_biased + eps)
//             backward  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * (xhat > 0) with ReLU, else dy
// Both are HBM streams: forward reads x twice (second time from L2 / MALL) and writes y; backward reads x and dy
// twice and writes dx.  Algorithmic bytes: forward 2 tensors, backward 3 tensors.
#include <cstdlib>
#include "common.hpp"

namespace {

constexpr int NT = 256;
constexpr int MAX_CHUNKS = 64;

struct NormPlan {
  int chunks;
  long long chunk_len;  // floats, multiple of 4
};

NormPlan norm_plan(int planes, long long plane) {
  long long want = (2048 + planes - 1) / planes;        // aim at >= 2048 workgroups
  const long long by_size = (plane + 4095) / 4096;      // but at least 4096 floats per chunk
  if (want > by_size) want = by_size;
  if (want > MAX_CHUNKS) want = MAX_CHUNKS;
  if (want < 1) want = 1;
  long long len = (plane + want - 1) / want;
  len = (len + 3) & ~3LL;
  NormPlan p;
  p.chunk_len = len;
  p.chunks = (int)((plane + len - 1) / len);
  return p;
}

__device__ __forceinline__ double2 block_sum2(double a, double b) {
  __shared__ double sa[NT / 64], sb[NT / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sa[wave] = a;
    sb[wave] = b;
  }
  __syncthreads();
  double ra = 0., rb = 0.;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {  // fixed order
    ra += sa[w];
    rb += sb[w];
  }
  return make_double2(ra, rb);
}

// the plane's totals from the per-chunk partials, same order in every workgroup
__device__ __forceinline__ double2 plane_totals(const double2* __restrict__ part, int chunks) {
  double a = 0., b = 0.;
  for (int k = 0; k < chunks; ++k) {
    const double2 p = part[k];
    a += p.x;
    b += p.y;
  }
  return make_double2(a, b);
}

template <bool BWD>
__global__ __launch_bounds__(NT) void instnorm_stats_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dy,
                                                           const float* __restrict__ mean_rstd,
                                                           double2* __restrict__ part, long long plane,
                                                           long long chunk_len, int relu, int vec) {
  const int pl = blockIdx.y, ck = blockIdx.x;
  const long long beg = ck * chunk_len;
  const long long end = min(plane, beg + chunk_len);
  const float* xp = x + (size_t)pl * plane;
  const float* gp = BWD ? dy + (size_t)pl * plane : nullptr;
  float mean = 0.f, rstd = 1.f;
  if (BWD) {
    mean = mean_rstd[2 * pl];
    rstd = mean_rstd[2 * pl + 1];
  }
  double a = 0., b = 0.;
  auto take = [&](float xv, float gv) {
    if (BWD) {
      const float xh = (xv - mean) * rstd;
      const float g = (relu && !(xh > 0.f)) ? 0.f : gv;
      a += (double)g;
      b += (double)(g * xh);
    } else {
      a += (double)xv;
      b += (double)xv * (double)xv;
    }
  };
  if (vec) {
    for (long long i = beg + 4 * threadIdx.x; i < end; i += 4 * NT) {  // chunk_len % 4 == 0, plane % 4 == 0
      const float4 xv = *reinterpret_cast<const float4*>(xp + i);
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BWD) gv = *reinterpret_cast<const float4*>(gp + i);
      take(xv.x, gv.x); take(xv.y, gv.y); take(xv.z, gv.z); take(xv.w, gv.w);
    }
  } else {
    for (long long i = beg + threadIdx.x; i < end; i += NT) take(xp[i], BWD ? gp[i] : 0.f);
  }
  const double2 r = block_sum2(a, b);
  if (threadIdx.x == 0) part[(size_t)pl * gridDim.x + ck] = r;
}

template <bool BWD>
__global__ __launch_bounds__(NT) void instnorm_apply_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dy,
                                                           float* __restrict__ out,
                                                           float* __restrict__ mean_rstd,
                                                           const double2* __restrict__ part, long long plane,
                                                           long long chunk_len, float eps, int relu, int vec) {
  const int pl = blockIdx.y, ck = blockIdx.x;
  const double2 tot = plane_totals(part + (size_t)pl * gridDim.x, gridDim.x);
  float mean, rstd, m1 = 0.f, m2 = 0.f;
  if (BWD) {
    mean = mean_rstd[2 * pl];
    rstd = mean_rstd[2 * pl + 1];
    m1 = (float)(tot.x / (double)plane);
    m2 = (float)(tot.y / (double)plane);
  } else {
    const double mu = tot.x / (double)plane;
    double var = tot.y / (double)plane - mu * mu;  // biased, as nn.InstanceNorm2d
    if (var < 0.) var = 0.;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (ck == 0 && threadIdx.x == 0) {
      mean_rstd[2 * pl] = mean;
      mean_rstd[2 * pl + 1] = rstd;
    }
  }
  const long long beg = ck * chunk_len;
  const long long end = min(plane, beg + chunk_len);
  const float* xp = x + (size_t)pl * plane;
  const float* gp = BWD ? dy + (size_t)pl * plane : nullptr;
  float* op = out + (size_t)pl * plane;
  auto f = [&](float xv, float gv) -> float {
    const float xh = (xv - mean) * rstd;
    if (BWD) {
      const float g = (relu && !(xh > 0.f)) ? 0.f : gv;
      return rstd * (g - m1 - xh * m2);
    }
    return relu ? fmaxf(xh, 0.f) : xh;
  };
  if (vec) {
    for (long long i = beg + 4 * threadIdx.x; i < end; i += 4 * NT) {
      const float4 xv = *reinterpret_cast<const float4*>(xp + i);
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BWD) gv = *reinterpret_cast<const float4*>(gp + i);
      *reinterpret_cast<float4*>(op + i) = make_float4(f(xv.x, gv.x), f(xv.y, gv.y), f(xv.z, gv.z), f(xv.w, gv.w));
    }
  } else {
    for (long long i = beg + threadIdx.x; i < end; i += NT) op[i] = f(xp[i], BWD ? gp[i] : 0.f);
  }
}

// ---- one launch per direction for planes that fit a workgroup's registers (r05) -----------------------------------------
// The two-launch form above reads x twice (statistics, then apply) and pays two launch boundaries on planes of 28 KB (the
// 55x128 stage: 7-9 us for both launches) and 112 KB (110x256).  Here ONE workgroup owns a plane: every thread keeps its
// VPT float4 pieces in registers, the block reduces the fp64 partial sums in a fixed order (bit-reproducible), and the
// normalised values leave from the registers -- one read and one write of the tensor, one launch.  The 220x512 stage
// (450 KB per plane: 110 registers per thread at 1024 threads) stays on the two-launch form.
template <int NTF>
__device__ __forceinline__ double2 block_sum2_n(double a, double b, double* sa, double* sb) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sa[wave] = a;
    sb[wave] = b;
  }
  __syncthreads();
  double ra = 0., rb = 0.;
#pragma unroll
  for (int w = 0; w < NTF / 64; ++w) {  // fixed order
    ra += sa[w];
    rb += sb[w];
  }
  return make_double2(ra, rb);
}

template <int NTF, int VPT, bool BWD>
__global__ __launch_bounds__(NTF) void instnorm_plane_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ out, float* __restrict__ mean_rstd,
                                                            long long plane, float eps, int relu) {
  __shared__ double sa[NTF / 64], sb[NTF / 64];
  const int pl = blockIdx.x, tid = threadIdx.x;
  const float* xp = x + (size_t)pl * plane;
  const float* gp = BWD ? dy + (size_t)pl * plane : nullptr;
  float* op = out + (size_t)pl * plane;
  const long long n4 = plane >> 2;   // host: plane % 4 == 0, 16-B aligned, n4 <= NTF * VPT
  float4 xv[VPT], gv[BWD ? VPT : 1];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const long long j = min((long long)(i * NTF + tid), n4 - 1);   // clamped: surplus lanes re-read the last piece (not summed)
    xv[i] = reinterpret_cast<const float4*>(xp)[j];
    if (BWD) gv[i] = reinterpret_cast<const float4*>(gp)[j];
  }
  float mean = 0.f, rstd = 1.f;
  if (BWD) {
    mean = mean_rstd[2 * pl];
    rstd = mean_rstd[2 * pl + 1];
  }
  double a = 0., b = 0.;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    if ((long long)(i * NTF + tid) < n4) {
      const float xs[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
      float gs[4] = {0.f, 0.f, 0.f, 0.f};
      if (BWD) { gs[0] = gv[i].x; gs[1] = gv[i].y; gs[2] = gv[i].z; gs[3] = gv[i].w; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (BWD) {
          const float xh = (xs[k] - mean) * rstd;
          const float g = (relu && !(xh > 0.f)) ? 0.f : gs[k];
          a += (double)g;
          b += (double)(g * xh);
        } else {
          a += (double)xs[k];
          b += (double)xs[k] * (double)xs[k];
        }
      }
    }
  }
  const double2 tot = block_sum2_n<NTF>(a, b, sa, sb);
  float m1 = 0.f, m2 = 0.f;
  if (BWD) {
    m1 = (float)(tot.x / (double)plane);
    m2 = (float)(tot.y / (double)plane);
  } else {
    const double mu = tot.x / (double)plane;
    double var = tot.y / (double)plane - mu * mu;  // biased, as nn.InstanceNorm2d
    if (var < 0.) var = 0.;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (tid == 0) {
      mean_rstd[2 * pl] = mean;
      mean_rstd[2 * pl + 1] = rstd;
    }
  }
  auto f = [&](float xs, float gs) -> float {
    const float xh = (xs - mean) * rstd;
    if (BWD) {
      const float g = (relu && !(xh > 0.f)) ? 0.f : gs;
      return rstd * (g - m1 - xh * m2);
    }
    return relu ? fmaxf(xh, 0.f) : xh;
  };
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const long long j = (long long)(i * NTF + tid);
    if (j < n4) {
      const float4 gq = BWD ? gv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(op)[j] = make_float4(f(xv[i].x, gq.x), f(xv[i].y, gq.y), f(xv[i].z, gq.z), f(xv[i].w, gq.w));
    }
  }
}

// planes of at most 7168 floats: 256 threads x 7 pieces; at most 28672: 1024 x 7.  0 = two-launch form.
// PCFA_INSTNORM_FUSED=0 switches the one-launch form off (A/B, read once).
int plane_kernel_threads(long long plane, int planes, bool vec) {
  static const bool on = !(getenv("PCFA_INSTNORM_FUSED") && atoi(getenv("PCFA_INSTNORM_FUSED")) == 0);
  if (!on || !vec || planes < 96) return 0;   // few planes: one workgroup each would leave the chip empty
  if (plane <= 4LL * 256 * 7) return 256;
  if (plane <= 4LL * 1024 * 7) return 1024;
  return 0;
}

// out = relu(a + b); backward of both operands: g * (out > 0) (pcfa_relu_bwd).
__global__ __launch_bounds__(NT) void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ out, long long n, int vec) {
  const long long step = (long long)gridDim.x * NT;
  if (vec) {
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += step) {
      const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
      reinterpret_cast<float4*>(out)[i] =
          make_float4(fmaxf(u.x + v.x, 0.f), fmaxf(u.y + v.y, 0.f), fmaxf(u.z + v.z, 0.f), fmaxf(u.w + v.w, 0.f));
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * NT + threadIdx.x; i < n; i += step)
      out[i] = fmaxf(a[i] + b[i], 0.f);
  } else {
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += step) out[i] = fmaxf(a[i] + b[i], 0.f);
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" size_t pcfa_instnorm_workspace_bytes(int planes, long long plane) {
  if (planes < 1 || plane < 1) return 0;
  const NormPlan p = norm_plan(planes, plane);
  return (size_t)planes * p.chunks * sizeof(double2);
}

extern "C" int pcfa_instnorm_fwd(const float* x, float* y, float* mean_rstd, void* workspace, int planes,
                                 long long plane, float eps, int relu, void* stream) {
  if (!x || !y || !mean_rstd || !workspace || planes < 1 || plane < 1 || !(eps >= 0.f)) return PCFA_ERR_INVALID_ARG;
  if (!aligned16(workspace)) return PCFA_ERR_WORKSPACE;
  const NormPlan p = norm_plan(planes, plane);
  const int vec = plane % 4 == 0 && aligned16(x) && aligned16(y);
  hipStream_t s = (hipStream_t)stream;
  if (const int nt = plane_kernel_threads(plane, planes, vec != 0)) {
    if (nt == 256)
      pcfa_launch(instnorm_plane_kernel<256, 7, false>, dim3(planes), dim3(256), 0, s, x, (const float*)nullptr, y, mean_rstd,
                  plane, eps, relu);
    else
      pcfa_launch(instnorm_plane_kernel<1024, 7, false>, dim3(planes), dim3(1024), 0, s, x, (const float*)nullptr, y,
                  mean_rstd, plane, eps, relu);
    PCFA_LAUNCH_CHECK();
    return PCFA_OK;
  }
  dim3 grid(p.chunks, planes);
  double2* part = (double2*)workspace;
  pcfa_launch(instnorm_stats_kernel<false>, grid, dim3(NT), 0, s, x, (const float*)nullptr,
              (const float*)nullptr, part, plane, p.chunk_len, relu, vec);
  PCFA_LAUNCH_CHECK();
  pcfa_launch(instnorm_apply_kernel<false>, grid, dim3(NT), 0, s, x, (const float*)nullptr, y, mean_rstd,
              (const double2*)part, plane, p.chunk_len, eps, relu, vec);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_instnorm_bwd(const float* x, const float* mean_rstd, const float* grad_out, float* grad_x,
                                 void* workspace, int planes, long long plane, int relu, void* stream) {
  if (!x || !mean_rstd || !grad_out || !grad_x || !workspace || planes < 1 || plane < 1)
    return PCFA_ERR_INVALID_ARG;
  if (!aligned16(workspace)) return PCFA_ERR_WORKSPACE;
  const NormPlan p = norm_plan(planes, plane);
  const int vec = plane % 4 == 0 && aligned16(x) && aligned16(grad_out) && aligned16(grad_x);
  hipStream_t s = (hipStream_t)stream;
  if (const int nt = plane_kernel_threads(plane, planes, vec != 0)) {
    if (nt == 256)
      pcfa_launch(instnorm_plane_kernel<256, 7, true>, dim3(planes), dim3(256), 0, s, x, grad_out, grad_x, (float*)mean_rstd,
                  plane, 0.f, relu);
    else
      pcfa_launch(instnorm_plane_kernel<1024, 7, true>, dim3(planes), dim3(1024), 0, s, x, grad_out, grad_x,
                  (float*)mean_rstd, plane, 0.f, relu);
    PCFA_LAUNCH_CHECK();
    return PCFA_OK;
  }
  dim3 grid(p.chunks, planes);
  double2* part = (double2*)workspace;
  pcfa_launch(instnorm_stats_kernel<true>, grid, dim3(NT), 0, s, x, grad_out, mean_rstd, part, plane, p.chunk_len,
              relu, vec);
  PCFA_LAUNCH_CHECK();
  pcfa_launch(instnorm_apply_kernel<true>, grid, dim3(NT), 0, s, x, grad_out, grad_x, (float*)mean_rstd,
              (const double2*)part, plane, p.chunk_len, 0.f, relu, vec);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_add_relu_fwd(const float* a, const float* b, float* out, long long n, void* stream) {
  if (!a || !b || !out || n < 1) return PCFA_ERR_INVALID_ARG;
  const int vec = aligned16(a) && aligned16(b) && aligned16(out);
  const long long work = vec ? (n + 3) / 4 : n;
  long long blocks = (work + NT - 1) / NT;
  if (blocks > 256 * 32) blocks = 256 * 32;
  pcfa_launch(add_relu_kernel, dim3((int)blocks), dim3(NT), 0, (hipStream_t)stream, a, b, out, n, vec);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
