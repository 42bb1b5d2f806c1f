// k x k / stride 1 / "same" convolution with 1..4 INPUT channels, bias and optional ReLU fused; forward only.
//
// Replaces, for the frozen-weight attack, convf1 of the RAFT / GMA motion encoder (reference
// models/raft/update.py:79-101: relu(Conv2d(2, 128, 7, padding=3)(flow))) -- 12 library launches of 15-20 us plus
// 12 bias+ReLU passes per closure.  Its input is the current flow estimate, which the reference detaches every
// iteration (raft.py:122-123), so no data gradient exists; the wrapper refuses inputs that require one.
// With two input channels the layer is 98 multiplies per output: far too thin for the matrix cores, and writing the
// 128-channel output (3.6 MB at 55 x 128) is the only real traffic -- an HBM-bound stream.
//   workgroup = 64 consecutive pixels (one per lane) x 8 waves.  The flat input range the tile's taps can touch
//   ([p0 - r(W+1), p0 + 63 + r(W+1)] per channel) and ALL weights go to LDS once; a lane gathers its Cin*k*k taps
//   into registers (horizontal wrap-around masked), wave w then produces output channels w, w+8, ... with the
//   weights as broadcast 16-B LDS reads and stores 64 consecutive floats per channel.
#include "common.hpp"

namespace {

constexpr int FI_PX = 64, FI_WAVES = 8;  // 8 waves: the Cin*k*k taps of a lane live in registers (<= 256 VGPRs)

template <int CIN, int KS>
__global__ __launch_bounds__(FI_PX* FI_WAVES) void conv_fewin_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int N, int H, int W, int relu) {
  constexpr int R = KS / 2, T = CIN * KS * KS;  // taps per output
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [N][T] weights (padded to 16 B), then [CIN][span] inputs
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const long long plane = (long long)H * W;
  const long long p0 = (long long)blockIdx.x * FI_PX;
  const int halo = R * (W + 1);
  const int span = FI_PX + 2 * halo;
  float* wl = lds;
  float* xl = lds + (((size_t)N * T + 3) & ~(size_t)3);

  // weights: a flat copy of w ([N][T] contiguous) in 16-B pieces, all loads of a batch issued before its LDS
  // writes (a plain copy loop is one global round trip per element: 25 of them made the kernel 20 us)
  {
    constexpr int NTH = FI_PX * FI_WAVES, BATCH = 8;
    const int n4 = (N * T) / 4;  // whole float4s (the pointer is 16-B aligned: checked by the host)
    for (int e0 = threadIdx.x; e0 < n4; e0 += NTH * BATCH) {
      f32x4 t[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int e = e0 + k * NTH;
        t[k] = reinterpret_cast<const f32x4*>(w)[e < n4 ? e : 0];
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int e = e0 + k * NTH;
        if (e < n4) reinterpret_cast<f32x4*>(wl)[e] = t[k];
      }
    }
    for (int e = 4 * n4 + threadIdx.x; e < N * T; e += NTH) wl[e] = w[e];
  }
  // inputs: flat range [p0 - halo, p0 + 63 + halo] of every channel, zero outside the image
  const float* xb = x + (size_t)b * CIN * plane;
  for (int e = threadIdx.x; e < CIN * span; e += FI_PX * FI_WAVES) {
    const int c = e / span, i = e - c * span;
    const long long q = p0 - halo + i;
    xl[e] = (q >= 0 && q < plane) ? xb[(size_t)c * plane + q] : 0.f;
  }
  __syncthreads();

  const long long p = p0 + lane;
  const int xx = (int)(p % W);  // (rows outside the image are zero in the flat range; columns wrap and are masked)
  float v[T];
#pragma unroll
  for (int c = 0; c < CIN; ++c)
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int dx = kx - R;
        const bool ok = xx + dx >= 0 && xx + dx < W;
        const float t = xl[c * span + halo + lane + (ky - R) * W + dx];
        v[(c * KS + ky) * KS + kx] = ok ? t : 0.f;
      }
  if (p >= plane) return;
  float* ob = out + (size_t)b * N * plane + p;
  for (int o = wave; o < N; o += FI_WAVES) {
    const float* wo = wl + (size_t)o * T;  // broadcast reads (all lanes, same address)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four partial sums in a fixed pattern
#pragma unroll
    for (int i = 0; i + 3 < T; i += 4) {
      s0 += wo[i] * v[i];
      s1 += wo[i + 1] * v[i + 1];
      s2 += wo[i + 2] * v[i + 2];
      s3 += wo[i + 3] * v[i + 3];
    }
#pragma unroll
    for (int i = T & ~3; i < T; ++i) s0 += wo[i] * v[i];
    float s = (s0 + s1) + (s2 + s3) + (bias ? bias[o] : 0.f);
    if (relu) s = fmaxf(s, 0.f);
    ob[(size_t)o * plane] = s;
  }
}

template <int CIN, int KS>
int launch_fewin(const float* x, const float* w, const float* bias, float* out, int B, int N, int H, int W,
                 int relu, hipStream_t s) {
  constexpr int T = CIN * KS * KS;
  const long long plane = (long long)H * W;
  const size_t span = FI_PX + 2 * (size_t)(KS / 2) * (W + 1);
  const size_t bytes = ((((size_t)N * T + 3) & ~(size_t)3) + CIN * span) * sizeof(float);
  if (reinterpret_cast<uintptr_t>(w) & 15) return PCFA_ERR_UNSUPPORTED;
  if (bytes > 150 * 1024) return PCFA_ERR_UNSUPPORTED;
  static size_t granted = 0;  // per template instance
  if (bytes > granted) {
    if (hipFuncSetAttribute((const void*)conv_fewin_fwd_kernel<CIN, KS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return PCFA_ERR_UNSUPPORTED;
    granted = bytes;
  }
  dim3 grid(pcfa_cdiv(plane, FI_PX), B), block(FI_PX * FI_WAVES);
  pcfa_launch(conv_fewin_fwd_kernel<CIN, KS>, grid, block, bytes, s, x, w, bias, out, N, H, W, relu);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

}  // namespace

extern "C" int pcfa_conv_fewin_fwd(const float* x, const float* w, const float* bias, float* out, int B, int Cin,
                                   int N, int H, int W, int ksize, int relu, void* stream) {
  if (!x || !w || !out || B < 1 || N < 1 || H < 1 || W < 1) return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (ksize == 7 && Cin == 2) return launch_fewin<2, 7>(x, w, bias, out, B, N, H, W, relu, s);
  if (ksize == 7 && Cin == 1) return launch_fewin<1, 7>(x, w, bias, out, B, N, H, W, relu, s);
  if (ksize == 3 && Cin == 2) return launch_fewin<2, 3>(x, w, bias, out, B, N, H, W, relu, s);
  if (ksize == 5 && Cin == 2) return launch_fewin<2, 5>(x, w, bias, out, B, N, H, W, relu, s);
  return PCFA_ERR_UNSUPPORTED;
}
