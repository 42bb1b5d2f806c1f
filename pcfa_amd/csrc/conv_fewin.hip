// k x k / stride 1 / "same" convolution with 1..4 INPUT channels, bias and optional ReLU fused; forward only.
//
// Replaces, for the frozen-weight attack, convf1 of the RAFT / GMA motion encoder (reference
// models/raft/update.py:79-101: relu(Conv2d(2, 128, 7, padding=3)(flow))) -- 12 library launches of 15-20 us plus
// 12 bias+ReLU passes per closure.  Its input is the current flow estimate, which the reference detaches every
// iteration (raft.py:122-123), so no data gradient exists; the wrapper refuses inputs that require one.
// With two input channels the layer is 98 multiplies per output: far too thin for the matrix cores, and writing the
// 128-channel output (3.6 MB at 55 x 128) is the only real traffic -- an HBM-bound stream.
//   workgroup = 64 consecutive pixels x 4 waves.  The flat input range the tile's taps can touch
//   ([p0 - r(W+1), p0 + 63 + r(W+1)] per channel) and ALL weights go to LDS once; the layer then is the GEMM
//   out[N x 64] = W[N x T] . V[T x 64] (T = Cin*k*k taps, V gathered from the flat range with the horizontal
//   wrap-around masked) on v_mfma_f32_32x32x2_f32, wave w owning 32 output channels; bias + ReLU in the epilogue.
#include "common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FI_PX = 64;      // pixels per workgroup: two 32-wide MFMA column tiles
constexpr int FI_THREADS = 256;  // 4 waves, wave w owns output rows 32w .. 32w+31 of a 128-row block

// out[N x 64 pixels] = W[N x T] . V[T x 64], T = Cin*k*k taps -- an im2col-in-LDS GEMM on v_mfma_f32_32x32x2_f32
// (exact fp32 products, fp32 accumulation).  The first version gave every lane the taps of its pixel in registers
// and read the weights as wave-wide broadcasts; a broadcast read still costs the full LDS bandwidth and the kernel
// sat at 19 us.  As MFMA operands both matrices are ordinary per-lane reads: A = W[32w + (lane & 31)][2s + (lane >> 5)],
// B = tap (2s + (lane >> 5)) of pixel (32j + (lane & 31)).
template <int CIN, int KS>
__global__ __launch_bounds__(FI_THREADS) void conv_fewin_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int N, int H, int W, int relu) {
  constexpr int R = KS / 2, T = CIN * KS * KS;
  constexpr int STEPS = (T + 1) / 2;     // k pairs
  constexpr int TS = (T + 1) | 1;        // odd row stride > T: conflict-free A reads, zero column at index T
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [npad][TS] weights, then [CIN][span] inputs
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const long long plane = (long long)H * W;
  const long long p0 = (long long)blockIdx.x * FI_PX;
  const int halo = R * (W + 1);
  const int span = FI_PX + 2 * halo;
  const int npad = (N + 127) & ~127;
  float* wl = lds;
  float* xl = lds + (size_t)npad * TS;

  // Staging.  Every global load of the workgroup (weights in 16-B pieces, inputs as dwords) is issued before the
  // first LDS write: a copy loop that waits per element is one global round trip per iteration, and 14 of them were
  // 3/4 of this kernel's time in its first versions.  Preconditions (host): w 16-B aligned.
  constexpr int WQ = 16;   // 16-B weight pieces per thread in flight: covers N*T <= 16384 floats per pass
  constexpr int XQ = 12;   // input dwords per thread in flight: covers Cin*span <= 3072 per pass
  const int total = N * T, n4 = total / 4;
  for (int e = threadIdx.x; e < (npad * TS + 3) / 4; e += FI_THREADS)
    reinterpret_cast<f32x4*>(wl)[e] = (f32x4)(0.f);  // rows >= N and the stride padding stay zero
  const float* xb = x + (size_t)b * CIN * plane;
  for (int w0 = 0, x0 = 0; w0 < n4 || x0 < CIN * span; w0 += WQ * FI_THREADS, x0 += XQ * FI_THREADS) {
    f32x4 tw[WQ];
    float tx[XQ];
#pragma unroll
    for (int k = 0; k < WQ; ++k) {
      const int e = w0 + threadIdx.x + k * FI_THREADS;
      tw[k] = reinterpret_cast<const f32x4*>(w)[e < n4 ? e : 0];
    }
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int e = x0 + threadIdx.x + k * FI_THREADS;
      const int c = e / span, i = e - c * span;
      const long long q = p0 - halo + i;
      const bool ok = e < CIN * span && q >= 0 && q < plane;
      const float t = xb[ok ? (size_t)c * plane + q : 0];
      tx[k] = ok ? t : 0.f;
    }
    __syncthreads();  // (first pass: the zero fill above is complete)
#pragma unroll
    for (int k = 0; k < WQ; ++k) {
      const int e = w0 + threadIdx.x + k * FI_THREADS;
      if (e < n4) {
        const int f = 4 * e;
        const float tv[4] = {tw[k].x, tw[k].y, tw[k].z, tw[k].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) wl[f + u + ((f + u) / T) * (TS - T)] = tv[u];
      }
    }
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int e = x0 + threadIdx.x + k * FI_THREADS;
      if (e < CIN * span) xl[e] = tx[k];
    }
  }
  for (int f = 4 * n4 + threadIdx.x; f < total; f += FI_THREADS) wl[f + (f / T) * (TS - T)] = w[f];
  __syncthreads();

  // column validity of this lane's two pixels: bit kx set <=> 0 <= x + kx - R < W (rows outside the image read zeros
  // from the flat range; columns wrap into the neighbouring row and are masked)
  unsigned colmask[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int xx = (int)((p0 + 32 * j + l31) % W);
    unsigned m = 0;
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
      if (xx + kx - R >= 0 && xx + kx - R < W) m |= 1u << kx;
    colmask[j] = m;
  }

  for (int nb = 0; nb < npad; nb += 128) {
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* arow = wl + (size_t)(nb + 32 * wave + l31) * TS + lh;
    // operands of step s+1 are read before the MFMAs of step s (the compiler otherwise waits for each step's three
    // LDS reads right in front of its two MFMAs)
    auto operands = [&](int s, float& a, float (&v)[2]) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int c0 = k0 / (KS * KS), r0 = k0 % (KS * KS), c1 = k1 / (KS * KS), r1 = k1 % (KS * KS);
      const int off0 = c0 * span + (r0 / KS - R) * W + (r0 % KS - R);
      const int off1 = c1 * span + (r1 / KS - R) * W + (r1 % KS - R);
      const bool live1 = k1 < T;  // odd T: the last odd tap does not exist
      const int off = lh ? (live1 ? off1 : off0) : off0;
      const int kx = lh ? (r1 % KS) : (r0 % KS);
      a = arow[k0];  // + lh folded into arow; rows / columns beyond (N, T) are zero in LDS
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float t = xl[halo + 32 * j + l31 + off];
        v[j] = (!((colmask[j] >> kx) & 1u) || (lh && !live1)) ? 0.f : t;
      }
    };
    // bias of this lane's 16 output rows, requested before the MFMA loop (32 dependent loads in the epilogue, each
    // waited for, cost more than the whole GEMM)
    float brow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = nb + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float t = bias ? bias[o < N ? o : 0] : 0.f;
      brow[r] = (bias && o < N) ? t : 0.f;
    }
    float a_cur, v_cur[2];
    operands(0, a_cur, v_cur);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      float a_nxt = 0.f, v_nxt[2] = {0.f, 0.f};
      if (s + 1 < STEPS) operands(s + 1, a_nxt, v_nxt);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, v_cur[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, v_cur[1], acc[1], 0, 0, 0);
      a_cur = a_nxt;
      v_cur[0] = v_nxt[0];
      v_cur[1] = v_nxt[1];
    }
    // C/D layout of the 32x32 tile: column = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long long p = p0 + 32 * j + l31;
      if (p >= plane) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = nb + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (o < N) {
          float v = acc[j][r] + brow[r];
          if (relu) v = fmaxf(v, 0.f);
          out[((size_t)b * N + o) * plane + p] = v;
        }
      }
    }
  }
}

template <int CIN, int KS>
int launch_fewin(const float* x, const float* w, const float* bias, float* out, int B, int N, int H, int W,
                 int relu, hipStream_t s) {
  constexpr int T = CIN * KS * KS, TS = (T + 1) | 1;
  const long long plane = (long long)H * W;
  const size_t span = FI_PX + 2 * (size_t)(KS / 2) * (W + 1);
  const size_t npad = ((size_t)N + 127) & ~(size_t)127;
  const size_t bytes = (npad * TS + CIN * span) * sizeof(float);
  if (bytes > 150 * 1024) return PCFA_ERR_UNSUPPORTED;
  static size_t granted = 0;  // per template instance
  if (bytes > granted) {
    if (hipFuncSetAttribute((const void*)conv_fewin_fwd_kernel<CIN, KS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return PCFA_ERR_UNSUPPORTED;
    granted = bytes;
  }
  dim3 grid(pcfa_cdiv(plane, FI_PX), B), block(FI_THREADS);
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
