// RAFT / GMA convex upsampling, forward and backward, for gfx950.
//
// Replaces RAFT.upsample_flow (models/raft/raft.py:72-83; models/gma/network.py likewise):
//     mask = softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
//     up   = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
//     out  = sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(N, 2, 8H, 8W)
// which the library runs as softmax + unfold (im2col) + broadcast multiply + reduction + permute copy over
// [N,2,9,8,8,H,W] temporaries (26 MB each at 55x128) -- about 250 us per closure forward + backward.
// Here: one streaming pass per direction.  Thread (w, i) of workgroup (h, 64 columns) owns the 8 output pixels
// (8h + i, 8w .. 8w + 7): per pixel it reads the 9 logits mask[k*64 + i*8 + j][h][w] (coalesced along w), forms the
// softmax exactly as above (max, exp, sum in k order, divide) and the 9-term sum in k order, and stores two float4 per
// channel (a wave writes 2 KB contiguous).  Backward: same ownership; d logits = p (dp - sum p dp) with
// dp_k = sum_c g_c up_ck, and the flow gradient goes through T[c][k][h][w] = sum_ij p_k g_c (summed over j in the thread,
// over i in LDS in index order) followed by a 9-tap gather (no atomics: bitwise reproducible).
#include "common.hpp"

namespace {

constexpr int UW = 64;   // columns per workgroup
constexpr int UT = UW * 8;   // threads: (w, i)

__device__ __forceinline__ void flow_taps(const float* __restrict__ fl, long long plane, int H, int W, int h, int w,
                                          float (&up)[2][9]) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int y = h + k / 3 - 1, x = w + k % 3 - 1;
    const bool ok = y >= 0 && y < H && x >= 0 && x < W;
    const long long o = (long long)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
    up[0][k] = ok ? 8.f * fl[o] : 0.f;            // 8 * flow first, as the reference forms it
    up[1][k] = ok ? 8.f * fl[plane + o] : 0.f;
  }
}

__device__ __forceinline__ void softmax9(const float (&x)[9], float (&p)[9]) {
  float m = x[0];
#pragma unroll
  for (int k = 1; k < 9; ++k) m = fmaxf(m, x[k]);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    p[k] = expf(x[k] - m);
    s += p[k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) p[k] = p[k] / s;
}

__global__ __launch_bounds__(UT) void convex_upsample_fwd_kernel(const float* __restrict__ flow,
                                                                 const float* __restrict__ mask,
                                                                 float* __restrict__ out, int H, int W) {
  const int tid = threadIdx.x, wl = tid & (UW - 1), i = tid / UW;
  const int h = blockIdx.y, w = blockIdx.x * UW + wl, n = blockIdx.z;
  if (w >= W) return;
  const long long plane = (long long)H * W;
  const float* fl = flow + (long long)n * 2 * plane;
  const float* mk = mask + (long long)n * 576 * plane + (long long)h * W + w;
  float up[2][9];
  flow_taps(fl, plane, H, W, h, w, up);
  float o[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x[9], p[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) x[k] = mk[(long long)(k * 64 + i * 8 + j) * plane];
    softmax9(x, p);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      a += p[k] * up[0][k];
      b += p[k] * up[1][k];
    }
    o[0][j] = a;
    o[1][j] = b;
  }
  const long long oplane = 64 * plane;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float4* dst = reinterpret_cast<float4*>(out + ((long long)n * 2 + c) * oplane + (long long)(8 * h + i) * (8 * W) + 8 * w);
    dst[0] = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
    dst[1] = make_float4(o[c][4], o[c][5], o[c][6], o[c][7]);
  }
}

// grad_mask and T[n][c*9 + k][h][w]
__global__ __launch_bounds__(UT) void convex_upsample_bwd_kernel(const float* __restrict__ flow,
                                                                 const float* __restrict__ mask,
                                                                 const float* __restrict__ gout,
                                                                 float* __restrict__ gmask, float* __restrict__ T,
                                                                 int H, int W) {
  __shared__ float s_t[8][18][UW];
  const int tid = threadIdx.x, wl = tid & (UW - 1), i = tid / UW;
  const int h = blockIdx.y, w = blockIdx.x * UW + wl, n = blockIdx.z;
  const bool live = w < W;
  const int wc = min(w, W - 1);
  const long long plane = (long long)H * W;
  const float* fl = flow + (long long)n * 2 * plane;
  const float* mk = mask + (long long)n * 576 * plane + (long long)h * W + wc;
  float* gm = gmask + (long long)n * 576 * plane + (long long)h * W + wc;
  float up[2][9];
  flow_taps(fl, plane, H, W, h, wc, up);
  const long long oplane = 64 * plane;
  float g[2][8];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float4* src =
        reinterpret_cast<const float4*>(gout + ((long long)n * 2 + c) * oplane + (long long)(8 * h + i) * (8 * W) + 8 * wc);
    const float4 a = src[0], b = src[1];
    g[c][0] = a.x; g[c][1] = a.y; g[c][2] = a.z; g[c][3] = a.w;
    g[c][4] = b.x; g[c][5] = b.y; g[c][6] = b.z; g[c][7] = b.w;
  }
  float t[18];
#pragma unroll
  for (int q = 0; q < 18; ++q) t[q] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float x[9], p[9], dp[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) x[k] = mk[(long long)(k * 64 + i * 8 + j) * plane];
    softmax9(x, p);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      dp[k] = g[0][j] * up[0][k] + g[1][j] * up[1][k];
      dot += p[k] * dp[k];
      t[k] += p[k] * g[0][j];
      t[9 + k] += p[k] * g[1][j];
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < 9; ++k) gm[(long long)(k * 64 + i * 8 + j) * plane] = p[k] * (dp[k] - dot);
    }
  }
#pragma unroll
  for (int q = 0; q < 18; ++q) s_t[i][q][wl] = t[q];
  __syncthreads();
  // thread (wl, i) finishes entries q = i, i + 8, i + 16 of column wl: sum over the 8 sub-rows in index order
  for (int q = i; q < 18; q += 8) {
    float s = 0.f;
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) s += s_t[ii][q][wl];
    if (live) T[((long long)n * 18 + q) * plane + (long long)h * W + w] = s;
  }
}

// grad_flow[c][y][x] = 8 * sum_k T[c][k][y - ky + 1][x - kx + 1]
__global__ void convex_upsample_gflow_kernel(const float* __restrict__ T, float* __restrict__ gflow, int N, int H, int W) {
  const long long plane = (long long)H * W, total = (long long)N * 2 * plane;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(e % W), y = (int)((e / W) % H);
    const long long nc = e / plane;          // n * 2 + c
    const long long n = nc >> 1, c = nc & 1;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int hy = y - (k / 3) + 1, wx = x - (k % 3) + 1;
      if (hy >= 0 && hy < H && wx >= 0 && wx < W) s += T[((n * 18) + c * 9 + k) * plane + (long long)hy * W + wx];
    }
    gflow[e] = 8.f * s;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pcfa_convex_upsample_fwd(const float* flow, const float* mask, float* out, int N, int H, int W,
                                        void* stream) {
  if (!flow || !mask || !out || N < 1 || H < 1 || W < 1 || N > 65535 || H > 65535) return PCFA_ERR_INVALID_ARG;
  if (!aligned16(out)) return PCFA_ERR_INVALID_ARG;   // rows of 8W floats: 32-B pieces
  pcfa_launch(convex_upsample_fwd_kernel, dim3(pcfa_cdiv(W, UW), H, N), dim3(UT), 0, (hipStream_t)stream, flow, mask, out,
              H, W);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" long long pcfa_convex_upsample_workspace_floats(int N, int H, int W) {
  if (N < 1 || H < 1 || W < 1) return -1;
  return 18LL * N * H * W;
}

extern "C" int pcfa_convex_upsample_bwd(const float* flow, const float* mask, const float* grad_out, float* grad_flow,
                                        float* grad_mask, float* workspace, int N, int H, int W, void* stream) {
  if (!flow || !mask || !grad_out || !grad_flow || !grad_mask || !workspace || N < 1 || H < 1 || W < 1 || N > 65535 ||
      H > 65535 || !aligned16(grad_out))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  pcfa_launch(convex_upsample_bwd_kernel, dim3(pcfa_cdiv(W, UW), H, N), dim3(UT), 0, s, flow, mask, grad_out, grad_mask,
              workspace, H, W);
  PCFA_LAUNCH_CHECK();
  const long long total = 2LL * N * H * W;
  pcfa_launch(convex_upsample_gflow_kernel, dim3((unsigned)min((total + 255) / 256, 2048LL)), dim3(256), 0, s, workspace,
              grad_flow, N, H, W);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

// ---- nn.Upsample(scale_factor = f, mode = 'bilinear') (align_corners = False), times a constant --------------------
// PWC-Net's `20 * self.upsample(flow2)` (models/PWCNet/PWCNet.py:73,321; f = 4) and its backward.  The library's
// backward (upsample_bilinear2d_backward_out_frame) scatters with fp32 atomics; here it is a gather over the <= 2f
// output rows and columns whose interpolation window reaches the input pixel -- bit-reproducible.  Index arithmetic as
// in ATen (area_pixel_compute_source_index): src = (1/f) * (dst + 0.5) - 0.5, clamped at 0; i1 = (int)src;
// i1p = i1 < size - 1; l1 = src - i1; l0 = 1 - l1;  out = l0y (l0x v00 + l1x v01) + l1y (l0x v10 + l1x v11).
namespace {

struct BilinTap {
  int i1, ip;
  float l0, l1;
};

__device__ __forceinline__ BilinTap bilin_tap(int dst, float rscale, int size) {
  float src = rscale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  BilinTap t;
  t.i1 = (int)src;
  t.ip = t.i1 < size - 1 ? 1 : 0;
  t.l1 = src - (float)t.i1;
  t.l0 = 1.f - t.l1;
  return t;
}

__global__ __launch_bounds__(256) void upsample_bilinear_fwd_kernel(const float* __restrict__ in,
                                                                    float* __restrict__ out, int H, int W, int f,
                                                                    float rscale, float mul) {
  const int OW = W * f, OH = H * f;
  const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
  if (X >= OW) return;
  const float* ip = in + (size_t)blockIdx.z * H * W;
  const BilinTap ty = bilin_tap(Y, rscale, H), tx = bilin_tap(X, rscale, W);
  const float* r0 = ip + (size_t)ty.i1 * W + tx.i1;
  const float* r1 = r0 + (size_t)ty.ip * W;
  const float v = ty.l0 * (tx.l0 * r0[0] + tx.l1 * r0[tx.ip]) + ty.l1 * (tx.l0 * r1[0] + tx.l1 * r1[tx.ip]);
  out[((size_t)blockIdx.z * OH + Y) * OW + X] = mul * v;
}

__global__ __launch_bounds__(256) void upsample_bilinear_bwd_kernel(const float* __restrict__ gout,
                                                                    float* __restrict__ gin, int H, int W, int f,
                                                                    float rscale, float mul) {
  const int OW = W * f, OH = H * f;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const float* gp = gout + (size_t)blockIdx.z * OH * OW;
  // outputs whose window can reach input index i: f (i - 0.5) - 0.5 < dst < f (i + 1.5) - 0.5 (one of slack each side;
  // the weight below is exact, and zero outside)
  const int y0 = max(0, f * y - (f + 1) / 2 - 1), y1 = min(OH - 1, f * y + f + (f + 1) / 2 + 1);
  const int x0 = max(0, f * x - (f + 1) / 2 - 1), x1 = min(OW - 1, f * x + f + (f + 1) / 2 + 1);
  float s = 0.f;
  for (int Y = y0; Y <= y1; ++Y) {
    const BilinTap ty = bilin_tap(Y, rscale, H);
    const float wy = (ty.i1 == y ? ty.l0 : 0.f) + (ty.i1 + ty.ip == y ? ty.l1 : 0.f);
    if (wy == 0.f) continue;
    float r = 0.f;
    for (int X = x0; X <= x1; ++X) {
      const BilinTap tx = bilin_tap(X, rscale, W);
      const float wx = (tx.i1 == x ? tx.l0 : 0.f) + (tx.i1 + tx.ip == x ? tx.l1 : 0.f);
      r += wx * gp[(size_t)Y * OW + X];
    }
    s += wy * r;
  }
  gin[((size_t)blockIdx.z * H + y) * W + x] = mul * s;
}

}  // namespace

extern "C" int pcfa_upsample_bilinear_fwd(const float* in, float* out, int planes, int H, int W, int factor, float mul,
                                          void* stream) {
  if (!in || !out || planes < 1 || H < 1 || W < 1 || factor < 1) return PCFA_ERR_INVALID_ARG;
  if (planes > 65535 || (long long)H * factor > 65535) return PCFA_ERR_UNSUPPORTED;
  dim3 grid(pcfa_cdiv((long long)W * factor, 256), H * factor, planes);
  pcfa_launch(upsample_bilinear_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W, factor,
              1.0f / (float)factor, mul);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_upsample_bilinear_bwd(const float* grad_out, float* grad_in, int planes, int H, int W, int factor,
                                          float mul, void* stream) {
  if (!grad_out || !grad_in || planes < 1 || H < 1 || W < 1 || factor < 1) return PCFA_ERR_INVALID_ARG;
  if (planes > 65535 || H > 65535) return PCFA_ERR_UNSUPPORTED;
  dim3 grid(pcfa_cdiv(W, 256), H, planes);
  pcfa_launch(upsample_bilinear_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad_out, grad_in, H, W, factor,
              1.0f / (float)factor, mul);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
