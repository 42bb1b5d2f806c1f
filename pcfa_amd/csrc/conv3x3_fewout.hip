// 3x3 / stride 1 / pad 1 convolution with 1..4 OUTPUT channels (frozen weights), forward and data gradient.
//
// Replaces the library convolution for the flow-prediction layers: FlowHead.conv2 of RAFT / GMA
// (models/raft/update.py:6-14, Conv2d(256, 2, 3, padding=1): 24 launches per closure, 31 us each as a Winograd
// kernel padded to a 32-channel output tile), predict_flow of PWC-Net (models/PWCNet/PWCNet.py:37-38) and of
// FlowNet2 (models/FlowNet/submodules.py:33-34).  With two output channels the layer is a stream over its input:
// 7.2 MB in, 56 KB out at 256 x 55 x 128 -- HBM-bound, 0.9 us at 8 TB/s; the matrix cores have nothing to do.
//   forward : workgroup = 64 consecutive pixels x 16 waves; wave w reduces channels w, w+16, ... with the nine taps
//             of its pixel in registers and WAVE-UNIFORM weights (scalar loads), partial sums meet in LDS and are
//             added in wave order (bitwise reproducible), + bias.
//   backward: grad_x[c][p] = sum_{o,ky,kx} w[o][c][ky][kx] * grad_out[o][p + (1-ky, 1-kx)]: a lane keeps the 9*N
//             gradient taps of its pixel in registers, a wave walks its channels with scalar weights and streams
//             grad_x out (64 consecutive floats per store instruction).
#include "common.hpp"

namespace {

constexpr int FO_PX = 64;     // pixels per workgroup (one per lane)
constexpr int FO_WAVES = 8;   // channel groups per workgroup (backward)
constexpr int FO_FWD_WAVES = 16;  // forward: 16 waves x 8 channels in flight = 72 loads per lane

template <int N>
__global__ __launch_bounds__(FO_PX* FO_FWD_WAVES) void conv3x3_fewout_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int K, int H, int W, int splits, int kper) {
  __shared__ float red[FO_FWD_WAVES][N][FO_PX];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // blockIdx.y = b * splits + split: with splits > 1 the workgroup reduces channels [kbeg, kend) only and `out` is
  // the partial-sum workspace [splits][B][N][plane] (no bias), summed in split order by fewout_reduce_kernel
  const int nb = gridDim.y / splits;
  const int b = blockIdx.y / splits, split = blockIdx.y - b * splits;
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const long long plane = (long long)H * W;
  const long long p = (long long)blockIdx.x * FO_PX + lane;
  const bool live = p < plane;
  const int y = live ? (int)(p / W) : 0, xx = live ? (int)(p % W) : 0;
  // tap offsets and validity of this lane's 3x3 neighbourhood
  int off[9];
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xq = xx + kx - 1;
      ok[ky * 3 + kx] = live && yy >= 0 && yy < H && xq >= 0 && xq < W;
      off[ky * 3 + kx] = ok[ky * 3 + kx] ? yy * W + xq : 0;
    }
  float acc[N];
#pragma unroll
  for (int o = 0; o < N; ++o) acc[o] = 0.f;
  const float* xb = x + (size_t)b * K * plane;
  constexpr int U = 8;  // channels in flight per wave: 72 loads per lane (the loop is a latency chain otherwise)
  for (int c0 = kbeg + wave; c0 < kend; c0 += U * FO_FWD_WAVES) {
    float v[U][9];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * FO_FWD_WAVES;  // wave-uniform
      const float* xc = xb + (size_t)(c < kend ? c : c0) * plane;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float t = xc[off[k]];
        v[u][k] = ok[k] ? t : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * FO_FWD_WAVES;
      if (c < kend) {  // uniform branch
#pragma unroll
        for (int o = 0; o < N; ++o) {
          const float* wc = w + ((size_t)o * K + c) * 9;  // wave-uniform address: scalar loads
#pragma unroll
          for (int k = 0; k < 9; ++k) acc[o] += wc[k] * v[u][k];
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < N; ++o) red[wave][o][lane] = acc[o];
  __syncthreads();
  for (int e = threadIdx.x; e < N * FO_PX; e += FO_PX * FO_FWD_WAVES) {
    const int o = e / FO_PX, l = e % FO_PX;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < FO_FWD_WAVES; ++g) s += red[g][o][l];
    const long long q = (long long)blockIdx.x * FO_PX + l;
    if (q < plane) {
      if (splits > 1) out[(((size_t)split * nb + b) * N + o) * plane + q] = s;
      else out[((size_t)b * N + o) * plane + q] = s + (bias ? bias[o] : 0.f);
    }
  }
}

// Same operator, rows staged through LDS (used when every channel plane is 16-B aligned: H*W % 4 == 0 and an aligned
// x).  The nine taps of a pixel lie in three flat 66-float segments [p0 + dy*W - 1, p0 + dy*W + 64] of the channel
// plane -- contiguous whatever the image width, because a 64-pixel tile is a flat pixel range and wrapped columns are
// masked.  A wave fetches the three segments of one channel with ONE 16-B-per-lane load (54 lanes) instead of nine
// dword loads: a wave-wide load costs ~16 cycles of address processing whatever its width, which is what bounded the
// kernel above (15.5 us at 256 x 55 x 128).  Each wave stages its own channels in a private LDS slice (no barrier).
constexpr int FO_SEG = 72;   // floats per staged segment (18 pieces)
constexpr int FO_UL = 4;     // channels in flight per wave (8 measured slower)

template <int N>
__global__ __launch_bounds__(FO_PX* FO_FWD_WAVES) void conv3x3_fewout_fwd_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int K, int H, int W, int splits, int kper) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float red[FO_FWD_WAVES][N][FO_PX];
  __shared__ __attribute__((aligned(16))) float seg[FO_FWD_WAVES][FO_UL][3][FO_SEG];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nb = gridDim.y / splits;
  const int b = blockIdx.y / splits, split = blockIdx.y - b * splits;
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const long long plane = (long long)H * W;
  const long long p0 = (long long)blockIdx.x * FO_PX;
  const long long p = p0 + lane;
  const bool live = p < plane;
  const int y = live ? (int)(p / W) : 0, xx = live ? (int)(p % W) : 0;
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xq = xx + kx - 1;
      ok[ky * 3 + kx] = live && yy >= 0 && yy < H && xq >= 0 && xq < W;
    }
  // this lane's piece of the staging load: segment r = lane / 18, piece m = lane % 18 (lanes >= 54 idle)
  const int lr = lane / 18, lm = lane - lr * 18;
  long long s4[3];
  int dsh[3];  // wave-uniform: offset of the segment's first float inside its aligned 72-float window
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const long long st = p0 + (long long)(r - 1) * W - 1;
    s4[r] = st & ~3LL;  // (two's complement: rounds towards -inf)
    dsh[r] = (int)(st - s4[r]);
  }
  const long long myidx = (lr == 0 ? s4[0] : (lr == 1 ? s4[1] : s4[2])) + 4 * lm;
  const bool myok = lane < 54 && myidx >= 0 && myidx + 3 < plane;
  const long long myoff = myok ? myidx : 0;

  float acc[N];
#pragma unroll
  for (int o = 0; o < N; ++o) acc[o] = 0.f;
  const float* xb = x + (size_t)b * K * plane;
  float* mine = &seg[wave][0][0][0];
  for (int c0 = kbeg + wave; c0 < kend; c0 += FO_UL * FO_FWD_WAVES) {
    f32x4 t[FO_UL];
#pragma unroll
    for (int u = 0; u < FO_UL; ++u) {
      const int c = c0 + u * FO_FWD_WAVES;  // wave-uniform
      const float* xc = xb + (size_t)(c < kend ? c : c0) * plane;
      const f32x4 q = *reinterpret_cast<const f32x4*>(xc + myoff);
      t[u] = myok ? q : (f32x4)(0.f);
    }
#pragma unroll
    for (int u = 0; u < FO_UL; ++u)
      if (lane < 54) *reinterpret_cast<f32x4*>(mine + (u * 3 + lr) * FO_SEG + 4 * lm) = t[u];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < FO_UL; ++u) {
      const int c = c0 + u * FO_FWD_WAVES;
      if (c < kend) {  // uniform branch
        float v[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float q = mine[(u * 3 + r) * FO_SEG + dsh[r] + lane + kx];
            v[3 * r + kx] = ok[3 * r + kx] ? q : 0.f;
          }
#pragma unroll
        for (int o = 0; o < N; ++o) {
          const float* wc = w + ((size_t)o * K + c) * 9;  // wave-uniform address: scalar loads
#pragma unroll
          for (int k = 0; k < 9; ++k) acc[o] += wc[k] * v[k];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next batch
  }
#pragma unroll
  for (int o = 0; o < N; ++o) red[wave][o][lane] = acc[o];
  __syncthreads();
  for (int e = threadIdx.x; e < N * FO_PX; e += FO_PX * FO_FWD_WAVES) {
    const int o = e / FO_PX, l = e % FO_PX;
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < FO_FWD_WAVES; ++g) sum += red[g][o][l];
    const long long q = p0 + l;
    if (q < plane) {
      if (splits > 1) out[(((size_t)split * nb + b) * N + o) * plane + q] = sum;
      else out[((size_t)b * N + o) * plane + q] = sum + (bias ? bias[o] : 0.f);
    }
  }
}

// out[b][o][p] = bias[o] + sum over splits (in order) of part[split][b][o][p]
__global__ void fewout_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                     float* __restrict__ out, long long n, long long plane, int N, int splits) {
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
    out[i] = s + (bias ? bias[(i / plane) % N] : 0.f);
  }
}

template <int N>
__global__ __launch_bounds__(FO_PX* FO_WAVES) void conv3x3_fewout_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ w, const float* __restrict__ addend, float* __restrict__ gx,
    int K, int H, int W, int csplit) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y / csplit, part = blockIdx.y % csplit;
  const long long plane = (long long)H * W;
  const long long p = (long long)blockIdx.x * FO_PX + lane;
  const bool live = p < plane;
  const int y = live ? (int)(p / W) : 0, xx = live ? (int)(p % W) : 0;
  // g[o][ky*3+kx] = grad_out[o][y + 1 - ky][x + 1 - kx]
  float g[N][9];
  const float* gb = gout + (size_t)b * N * plane;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + 1 - ky, xq = xx + 1 - kx;
      const bool ok = live && yy >= 0 && yy < H && xq >= 0 && xq < W;
      const int o_ = ok ? yy * W + xq : 0;
#pragma unroll
      for (int o = 0; o < N; ++o) {
        const float t = gb[(size_t)o * plane + o_];
        g[o][ky * 3 + kx] = ok ? t : 0.f;
      }
    }
  float* xb = gx + (size_t)b * K * plane;
  const float* ab = addend != nullptr ? addend + (size_t)b * K * plane : nullptr;   // another consumer's gradient of x
  constexpr int U = 4;
  const int cstep = FO_WAVES * csplit;
  for (int c0 = part * FO_WAVES + wave; c0 < K; c0 += U * cstep) {
    float s[U], a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * cstep;  // wave-uniform
      a[u] = (ab != nullptr && c < K && live) ? ab[(size_t)c * plane + p] : 0.f;   // requested ahead of the products
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * cstep;  // wave-uniform
      s[u] = 0.f;
      if (c < K) {
#pragma unroll
        for (int o = 0; o < N; ++o) {
          const float* wc = w + ((size_t)o * K + c) * 9;  // wave-uniform address: scalar loads
#pragma unroll
          for (int k = 0; k < 9; ++k) s[u] += wc[k] * g[o][k];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * cstep;
      if (c < K && live) xb[(size_t)c * plane + p] = s[u] + a[u];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(K, N <= 4, kernel 4, stride 2, padding 1) and its data gradient (frozen weights): PWC-Net's `deconv`
// layers (models/PWCNet/PWCNet.py:42-43; deconv6..2 on the 2-channel flow and upfeat6..3 on the 529..661-channel
// decoder output, :107-147, :259-304).  out[o][Y][X] = bias[o] + sum_{c,ky,kx} x[c][y][x] w[c][o][ky][kx] with
// Y = 2y - 1 + ky, X = 2x - 1 + kx: the 2x2 output block of input pixel (y, x) reads the pixel's 3x3 neighbourhood,
// four taps per output -- the same stream over the input as conv3x3_fewout (the library ran these as implicit-GEMM /
// GEMM + Col2Im2dU / first-call "naive" kernels chosen per process).  Lane = input pixel, waves split the channels,
// wave-uniform weights through scalar loads, partial sums meet in LDS and are added in wave order (bit-reproducible).
// Output class (a, b) = (Y & 1, X & 1); its two row taps are (dy, ky) = (-t, 1 + 2t) for a = 0 and (1 - t, 2t) for
// a = 1, t = 0, 1 (likewise for columns).
template <int N>
__global__ __launch_bounds__(FO_PX* FO_FWD_WAVES) void deconv4s2_fewout_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int K, int H, int W, int splits, int kper) {
  __shared__ float red[FO_FWD_WAVES][4 * N][FO_PX];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nb = gridDim.y / splits;
  const int b = blockIdx.y / splits, split = blockIdx.y - b * splits;
  const int kbeg = split * kper, kend = min(K, kbeg + kper);
  const long long plane = (long long)H * W;
  const long long p = (long long)blockIdx.x * FO_PX + lane;
  const bool live = p < plane;
  const int y = live ? (int)(p / W) : 0, xx = live ? (int)(p % W) : 0;
  int off[9];
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = y + ky - 1, xq = xx + kx - 1;
      ok[ky * 3 + kx] = live && yy >= 0 && yy < H && xq >= 0 && xq < W;
      off[ky * 3 + kx] = ok[ky * 3 + kx] ? yy * W + xq : 0;
    }
  float acc[4][N];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int o = 0; o < N; ++o) acc[q][o] = 0.f;
  const float* xb = x + (size_t)b * K * plane;
  constexpr int U = 4;  // channels in flight per wave
  for (int c0 = kbeg + wave; c0 < kend; c0 += U * FO_FWD_WAVES) {
    float v[U][9];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * FO_FWD_WAVES;  // wave-uniform
      const float* xc = xb + (size_t)(c < kend ? c : c0) * plane;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float t = xc[off[k]];
        v[u][k] = ok[k] ? t : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * FO_FWD_WAVES;
      if (c < kend) {  // uniform branch
#pragma unroll
        for (int o = 0; o < N; ++o) {
          const float* wc = w + ((size_t)c * N + o) * 16;  // wave-uniform address: scalar loads
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
              for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                for (int tx = 0; tx < 2; ++tx) {
                  const int dy = a ? 1 - ty : -ty, ky = a ? 2 * ty : 1 + 2 * ty;
                  const int dx = bb ? 1 - tx : -tx, kx = bb ? 2 * tx : 1 + 2 * tx;
                  acc[a * 2 + bb][o] += wc[ky * 4 + kx] * v[u][(dy + 1) * 3 + dx + 1];
                }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int o = 0; o < N; ++o) red[wave][q * N + o][lane] = acc[q][o];
  __syncthreads();
  const long long oplane = 4 * plane;
  for (int e = threadIdx.x; e < 4 * N * FO_PX; e += FO_PX * FO_FWD_WAVES) {
    const int qo = e / FO_PX, l = e % FO_PX;
    const int q = qo / N, o = qo - q * N;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < FO_FWD_WAVES; ++g) s += red[g][qo][l];
    const long long pp = (long long)blockIdx.x * FO_PX + l;
    if (pp < plane) {
      const int py = (int)(pp / W), px = (int)(pp % W);
      const long long oi = (long long)(2 * py + (q >> 1)) * (2 * W) + 2 * px + (q & 1);
      if (splits > 1) out[(((size_t)split * nb + b) * N + o) * oplane + oi] = s;
      else out[((size_t)b * N + o) * oplane + oi] = s + (bias ? bias[o] : 0.f);
    }
  }
}

// grad_x[c][y][x] = sum_{o,ky,kx} w[c][o][ky][kx] * grad_out[o][2y - 1 + ky][2x - 1 + kx]: a lane keeps the 16 N
// gradient taps of its pixel in registers, waves walk the channels with scalar weights and stream grad_x.
template <int N>
__global__ __launch_bounds__(FO_PX* FO_WAVES) void deconv4s2_fewout_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ w, float* __restrict__ gx, int K, int H, int W,
    int csplit) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y / csplit, part = blockIdx.y % csplit;
  const long long plane = (long long)H * W;
  const long long p = (long long)blockIdx.x * FO_PX + lane;
  const bool live = p < plane;
  const int y = live ? (int)(p / W) : 0, xx = live ? (int)(p % W) : 0;
  const int OH = 2 * H, OW = 2 * W;
  float g[N][16];
  const float* gb = gout + (size_t)b * N * 4 * plane;
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      const int yy = 2 * y - 1 + ky, xq = 2 * xx - 1 + kx;
      const bool ok = live && yy >= 0 && yy < OH && xq >= 0 && xq < OW;
      const long long o_ = ok ? (long long)yy * OW + xq : 0;
#pragma unroll
      for (int o = 0; o < N; ++o) {
        const float t = gb[(size_t)o * 4 * plane + o_];
        g[o][ky * 4 + kx] = ok ? t : 0.f;
      }
    }
  float* xb = gx + (size_t)b * K * plane;
  constexpr int U = 4;
  const int cstep = FO_WAVES * csplit;
  for (int c0 = part * FO_WAVES + wave; c0 < K; c0 += U * cstep) {
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * cstep;  // wave-uniform
      s[u] = 0.f;
      if (c < K) {
#pragma unroll
        for (int o = 0; o < N; ++o) {
          const float* wc = w + ((size_t)c * N + o) * 16;  // wave-uniform address: scalar loads
#pragma unroll
          for (int k = 0; k < 16; ++k) s[u] += wc[k] * g[o][k];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * cstep;
      if (c < K && live) xb[(size_t)c * plane + p] = s[u];
    }
  }
}

}  // namespace

// Channel splits of the forward: few pixels (coarse pyramid levels: 2 workgroups at 1024 x 7 x 16) cannot fill the
// chip with one workgroup per 64 pixels, so the channel sum is split until ~256 workgroups exist (>= 64 channels each;
// FlowNet2 closure: 920 -> 480 us over its 23 predict_flow layers).
static int fewout_splits(int B, int K, long long plane) {
  const long long tiles = (long long)pcfa_cdiv(plane, FO_PX) * B;
  long long want = (256 + tiles - 1) / tiles;
  const long long by_k = K / 64 > 0 ? K / 64 : 1;
  if (want > by_k) want = by_k;
  if (want > 64) want = 64;
  return want < 4 ? 1 : (int)want;  // the reduce launch costs ~5 us: not worth it below a 4-fold split
}

extern "C" size_t pcfa_conv3x3_fewout_workspace_bytes(int B, int K, int N, int H, int W) {
  if (B < 1 || K < 1 || N < 1 || N > 4 || H < 1 || W < 1) return 0;
  const long long plane = (long long)H * W;
  const int splits = fewout_splits(B, K, plane);
  return splits > 1 ? (size_t)splits * B * N * plane * sizeof(float) : 0;
}

extern "C" int pcfa_conv3x3_fewout_fwd(const float* x, const float* w, const float* bias, float* out,
                                       void* workspace, int B, int K, int N, int H, int W, void* stream) {
  if (!x || !w || !out || B < 1 || K < 1 || H < 1 || W < 1) return PCFA_ERR_INVALID_ARG;
  if (N < 1 || N > 4) return PCFA_ERR_UNSUPPORTED;
  const long long plane = (long long)H * W;
  const int splits = fewout_splits(B, K, plane);
  if (splits > 1 && !workspace) return PCFA_ERR_WORKSPACE;
  const int kper = ((K + splits - 1) / splits + FO_FWD_WAVES - 1) / FO_FWD_WAVES * FO_FWD_WAVES;
  float* dst = splits > 1 ? (float*)workspace : out;
  dim3 grid(pcfa_cdiv(plane, FO_PX), B * splits), block(FO_PX * FO_FWD_WAVES);
  hipStream_t s = (hipStream_t)stream;
  const bool wide = plane % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (wide) {
    switch (N) {
      case 1: pcfa_launch(conv3x3_fewout_fwd_lds_kernel<1>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
      case 2: pcfa_launch(conv3x3_fewout_fwd_lds_kernel<2>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
      case 3: pcfa_launch(conv3x3_fewout_fwd_lds_kernel<3>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
      default: pcfa_launch(conv3x3_fewout_fwd_lds_kernel<4>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
    }
  } else {
    switch (N) {
      case 1: pcfa_launch(conv3x3_fewout_fwd_kernel<1>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
      case 2: pcfa_launch(conv3x3_fewout_fwd_kernel<2>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
      case 3: pcfa_launch(conv3x3_fewout_fwd_kernel<3>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
      default: pcfa_launch(conv3x3_fewout_fwd_kernel<4>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
    }
  }
  PCFA_LAUNCH_CHECK();
  if (splits > 1) {
    const long long n = (long long)B * N * plane;
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    pcfa_launch(fewout_reduce_kernel, dim3((int)blocks), dim3(256), 0, s, (const float*)workspace, bias, out, n, plane,
                N, splits);
    PCFA_LAUNCH_CHECK();
  }
  return PCFA_OK;
}

extern "C" int pcfa_conv3x3_fewout_bwd(const float* grad_out, const float* w, const float* addend, float* grad_x, int B,
                                       int K, int N, int H, int W, void* stream) {
  if (!grad_out || !w || !grad_x || B < 1 || K < 1 || H < 1 || W < 1) return PCFA_ERR_INVALID_ARG;
  if (N < 1 || N > 4) return PCFA_ERR_UNSUPPORTED;
  const long long plane = (long long)H * W;
  // channels are independent in the data gradient: split them over blockIdx.y until the grid fills the chip
  const int tiles = pcfa_cdiv(plane, FO_PX) * B;
  int csplit = 1;
  while (tiles * csplit < 512 && FO_WAVES * csplit * 2 <= K) csplit *= 2;
  dim3 grid(pcfa_cdiv(plane, FO_PX), B * csplit), block(FO_PX * FO_WAVES);
  hipStream_t s = (hipStream_t)stream;
  switch (N) {
    case 1: pcfa_launch(conv3x3_fewout_bwd_kernel<1>, grid, block, 0, s, grad_out, w, addend, grad_x, K, H, W, csplit); break;
    case 2: pcfa_launch(conv3x3_fewout_bwd_kernel<2>, grid, block, 0, s, grad_out, w, addend, grad_x, K, H, W, csplit); break;
    case 3: pcfa_launch(conv3x3_fewout_bwd_kernel<3>, grid, block, 0, s, grad_out, w, addend, grad_x, K, H, W, csplit); break;
    default: pcfa_launch(conv3x3_fewout_bwd_kernel<4>, grid, block, 0, s, grad_out, w, addend, grad_x, K, H, W, csplit); break;
  }
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

// ---- ConvTranspose2d(K, N, 4, stride 2, padding 1), N <= 4 ------------------------------------------------------------
static int deconv_splits(int B, int K, long long plane) {
  const long long tiles = (long long)pcfa_cdiv(plane, FO_PX) * B;
  long long want = (256 + tiles - 1) / tiles;
  const long long by_k = K / 64 > 0 ? K / 64 : 1;
  if (want > by_k) want = by_k;
  if (want > 64) want = 64;
  return want < 4 ? 1 : (int)want;
}

extern "C" size_t pcfa_deconv4s2_fewout_workspace_bytes(int B, int K, int N, int H, int W) {
  if (B < 1 || K < 1 || N < 1 || N > 4 || H < 1 || W < 1) return 0;
  const long long plane = (long long)H * W;
  const int splits = deconv_splits(B, K, plane);
  return splits > 1 ? (size_t)splits * B * N * 4 * plane * sizeof(float) : 0;
}

extern "C" int pcfa_deconv4s2_fewout_fwd(const float* x, const float* w, const float* bias, float* out,
                                         void* workspace, int B, int K, int N, int H, int W, void* stream) {
  if (!x || !w || !out || B < 1 || K < 1 || H < 1 || W < 1) return PCFA_ERR_INVALID_ARG;
  if (N < 1 || N > 4) return PCFA_ERR_UNSUPPORTED;
  const long long plane = (long long)H * W;
  const int splits = deconv_splits(B, K, plane);
  if (splits > 1 && !workspace) return PCFA_ERR_WORKSPACE;
  const int kper = ((K + splits - 1) / splits + FO_FWD_WAVES - 1) / FO_FWD_WAVES * FO_FWD_WAVES;
  float* dst = splits > 1 ? (float*)workspace : out;
  dim3 grid(pcfa_cdiv(plane, FO_PX), B * splits), block(FO_PX * FO_FWD_WAVES);
  hipStream_t s = (hipStream_t)stream;
  switch (N) {
    case 1: pcfa_launch(deconv4s2_fewout_fwd_kernel<1>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
    case 2: pcfa_launch(deconv4s2_fewout_fwd_kernel<2>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
    case 3: pcfa_launch(deconv4s2_fewout_fwd_kernel<3>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
    default: pcfa_launch(deconv4s2_fewout_fwd_kernel<4>, grid, block, 0, s, x, w, bias, dst, K, H, W, splits, kper); break;
  }
  PCFA_LAUNCH_CHECK();
  if (splits > 1) {
    const long long n = (long long)B * N * 4 * plane;
    long long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    pcfa_launch(fewout_reduce_kernel, dim3((int)blocks), dim3(256), 0, s, (const float*)workspace, bias, out, n,
                4 * plane, N, splits);
    PCFA_LAUNCH_CHECK();
  }
  return PCFA_OK;
}

extern "C" int pcfa_deconv4s2_fewout_bwd(const float* grad_out, const float* w, float* grad_x, int B, int K, int N,
                                         int H, int W, void* stream) {
  if (!grad_out || !w || !grad_x || B < 1 || K < 1 || H < 1 || W < 1) return PCFA_ERR_INVALID_ARG;
  if (N < 1 || N > 4) return PCFA_ERR_UNSUPPORTED;
  const long long plane = (long long)H * W;
  const int tiles = pcfa_cdiv(plane, FO_PX) * B;
  int csplit = 1;
  while (tiles * csplit < 512 && FO_WAVES * csplit * 2 <= K) csplit *= 2;
  dim3 grid(pcfa_cdiv(plane, FO_PX), B * csplit), block(FO_PX * FO_WAVES);
  hipStream_t s = (hipStream_t)stream;
  switch (N) {
    case 1: pcfa_launch(deconv4s2_fewout_bwd_kernel<1>, grid, block, 0, s, grad_out, w, grad_x, K, H, W, csplit); break;
    case 2: pcfa_launch(deconv4s2_fewout_bwd_kernel<2>, grid, block, 0, s, grad_out, w, grad_x, K, H, W, csplit); break;
    case 3: pcfa_launch(deconv4s2_fewout_bwd_kernel<3>, grid, block, 0, s, grad_out, w, grad_x, K, H, W, csplit); break;
    default: pcfa_launch(deconv4s2_fewout_bwd_kernel<4>, grid, block, 0, s, grad_out, w, grad_x, K, H, W, csplit); break;
  }
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
