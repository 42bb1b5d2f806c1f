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

// ---------------------------------------------------------------------------------------------------------------
// Packed-weight variant (pcfa_conv_fewin_pack + pcfa_conv_fewin_packed_fwd): the weights of a frozen layer are laid out
// once in MFMA operand order -- P[32-row block nb][step group sg][lane][4]: float e of a lane = W[32 nb + (lane & 31)]
// [tap 2 (4 sg + e) + (lane >> 5)], zero beyond (N, T) -- and go from L2 straight into registers with ceil(STEPS/4)
// coalesced 16-B loads per lane, all in flight before the first MFMA.  The kernel above stages 50 KB of weights per
// workgroup through LDS (zero fill, 16-B loads, four de-strided dword LDS writes per piece): that staging, not the
// 49 MFMAs per wave, was most of its 16 us.  Only the flat input range remains in LDS (7 KB), tiles are 32 pixels
// (220 workgroups at 55x128 instead of 110).
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, int KS>
__global__ void conv_fewin_pack_kernel(const float* __restrict__ w, float* __restrict__ P, int N, long long total) {
  constexpr int T = CIN * KS * KS, STEPS = (T + 1) / 2, SG = (STEPS + 3) / 4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long long blk = e >> 8;   // nb * SG + sg
    const int sg = (int)(blk % SG), nb = (int)(blk / SG);
    const int n = 32 * nb + (lane & 31), tap = 2 * (4 * sg + q) + (lane >> 5);
    P[e] = (n < N && tap < T) ? w[(long long)n * T + tap] : 0.f;
  }
}

constexpr int FP_PX = 32;   // pixels per workgroup of the packed variant: one MFMA column tile

template <int CIN, int KS>
__global__ __launch_bounds__(FI_THREADS) void conv_fewin_packed_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ P, const float* __restrict__ bias,
    float* __restrict__ out, int N, int H, int W, int relu) {
  constexpr int R = KS / 2, T = CIN * KS * KS;
  constexpr int STEPS = (T + 1) / 2, SG = (STEPS + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float xl[];   // [CIN][span] flat input range
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const long long plane = (long long)H * W;
  const long long p0 = (long long)blockIdx.x * FP_PX;
  const int halo = R * (W + 1);
  const int span = FP_PX + 2 * halo;
  const float* xb = x + (size_t)b * CIN * plane;
  // inputs: every dword of the range requested before the first LDS write (clamped addresses, values masked)
  constexpr int XQ = 8;
  for (int x0 = 0; x0 < CIN * span; x0 += XQ * FI_THREADS) {
    float tx[XQ];
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int e = x0 + threadIdx.x + k * FI_THREADS;
      const int c = e / span, i = e - c * span;
      const long long q = p0 - halo + i;
      const bool ok = e < CIN * span && q >= 0 && q < plane;
      const float t = xb[ok ? (size_t)c * plane + q : 0];
      tx[k] = ok ? t : 0.f;
    }
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
      const int e = x0 + threadIdx.x + k * FI_THREADS;
      if (e < CIN * span) xl[e] = tx[k];
    }
  }
  const int xx = (int)((p0 + l31) % W);
  unsigned colmask = 0;
#pragma unroll
  for (int kx = 0; kx < KS; ++kx)
    if (xx + kx - R >= 0 && xx + kx - R < W) colmask |= 1u << kx;
  const int nblocks = (N + 127) / 128;
  for (int nb = 0; nb < nblocks; ++nb) {
    const int row0 = 128 * nb + 32 * wave;
    // A operands of all steps + the bias of this lane's 16 rows: in flight across the barrier below
    f32x4 areg[SG];
    const f32x4* pa = reinterpret_cast<const f32x4*>(P) + ((size_t)(4 * nb + wave) * SG) * 64 + lane;
#pragma unroll
    for (int sg = 0; sg < SG; ++sg) areg[sg] = pa[(size_t)sg * 64];
    float brow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float t = bias ? bias[o < N ? o : 0] : 0.f;
      brow[r] = (bias && o < N) ? t : 0.f;
    }
    if (nb == 0) __syncthreads();   // input range visible
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto tap_of = [&](int s) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int c0 = k0 / (KS * KS), r0 = k0 % (KS * KS), c1 = k1 / (KS * KS), r1 = k1 % (KS * KS);
      const int off0 = c0 * span + (r0 / KS - R) * W + (r0 % KS - R);
      const int off1 = c1 * span + (r1 / KS - R) * W + (r1 % KS - R);
      const bool live1 = k1 < T;
      const int off = lh ? (live1 ? off1 : off0) : off0;
      const int kx = lh ? (r1 % KS) : (r0 % KS);
      const float t = xl[halo + l31 + off];
      return (!((colmask >> kx) & 1u) || (lh && !live1)) ? 0.f : t;
    };
    constexpr int PRE = 4;   // B operands read this many MFMAs ahead
    float v[STEPS];
#pragma unroll
    for (int s = 0; s < PRE && s < STEPS; ++s) v[s] = tap_of(s);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s + PRE < STEPS) v[s + PRE] = tap_of(s + PRE);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 a4 = areg[s >> 2];
      const float a = (s & 3) == 0 ? a4.x : (s & 3) == 1 ? a4.y : (s & 3) == 2 ? a4.z : a4.w;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[s], acc, 0, 0, 0);
    }
    const long long p = p0 + l31;
    if (p < plane) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (o < N) {
          float y = acc[r] + brow[r];
          if (relu) y = fmaxf(y, 0.f);
          out[((size_t)b * N + o) * plane + p] = y;
        }
      }
    }
  }
}

template <int CIN, int KS>
long long packed_floats_fewin(int N) {
  constexpr int T = CIN * KS * KS, STEPS = (T + 1) / 2, SG = (STEPS + 3) / 4;
  return (long long)((N + 127) / 128) * 4 * SG * 64 * 4;
}

template <int CIN, int KS>
int pack_fewin(const float* w, float* P, int N, hipStream_t s) {
  const long long total = packed_floats_fewin<CIN, KS>(N);
  pcfa_launch(conv_fewin_pack_kernel<CIN, KS>, dim3((unsigned)min((total + 255) / 256, 1024LL)), dim3(256), 0, s, w, P, N,
              total);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

template <int CIN, int KS>
int launch_fewin_packed(const float* x, const float* P, const float* bias, float* out, int B, int N, int H, int W,
                        int relu, hipStream_t s) {
  const long long plane = (long long)H * W;
  const size_t span = FP_PX + 2 * (size_t)(KS / 2) * (W + 1);
  const size_t bytes = CIN * span * sizeof(float);
  if (bytes > 60 * 1024) return PCFA_ERR_UNSUPPORTED;
  dim3 grid(pcfa_cdiv(plane, FP_PX), B), block(FI_THREADS);
  pcfa_launch(conv_fewin_packed_fwd_kernel<CIN, KS>, grid, block, bytes, s, x, P, bias, out, N, H, W, relu);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
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

#define PCFA_FEWIN_DISPATCH(CALL)                              \
  if (ksize == 7 && Cin == 2) return CALL(2, 7);               \
  if (ksize == 7 && Cin == 1) return CALL(1, 7);               \
  if (ksize == 3 && Cin == 2) return CALL(2, 3);               \
  if (ksize == 5 && Cin == 2) return CALL(2, 5);

extern "C" long long pcfa_conv_fewin_packed_floats(int Cin, int N, int ksize) {
  if (N < 1) return -1;
#define PCFA_FEWIN_CALL(C, K) packed_floats_fewin<C, K>(N)
  PCFA_FEWIN_DISPATCH(PCFA_FEWIN_CALL)
#undef PCFA_FEWIN_CALL
  return -1;
}

extern "C" int pcfa_conv_fewin_pack(const float* w, float* packed, int Cin, int N, int ksize, void* stream) {
  if (!w || !packed || N < 1) return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
#define PCFA_FEWIN_CALL(C, K) pack_fewin<C, K>(w, packed, N, s)
  PCFA_FEWIN_DISPATCH(PCFA_FEWIN_CALL)
#undef PCFA_FEWIN_CALL
  return PCFA_ERR_UNSUPPORTED;
}

extern "C" int pcfa_conv_fewin_packed_fwd(const float* x, const float* packed, const float* bias, float* out, int B,
                                          int Cin, int N, int H, int W, int ksize, int relu, void* stream) {
  if (!x || !packed || !out || B < 1 || N < 1 || H < 1 || W < 1 || (reinterpret_cast<uintptr_t>(packed) & 15))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
#define PCFA_FEWIN_CALL(C, K) launch_fewin_packed<C, K>(x, packed, bias, out, B, N, H, W, relu, s)
  PCFA_FEWIN_DISPATCH(PCFA_FEWIN_CALL)
#undef PCFA_FEWIN_CALL
  return PCFA_ERR_UNSUPPORTED;
}
#undef PCFA_FEWIN_DISPATCH
