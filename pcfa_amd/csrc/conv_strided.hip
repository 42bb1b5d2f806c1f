// Stride-2 convolutions of the encoders on the fp32 matrix cores of gfx950 (direct implicit GEMM, no im2col).
//
// Replaces, for frozen weights: the 7x7 / stride-2 stem `Conv2d(3, 64, 7, stride=2, padding=3)` (reference
// models/raft/extractor.py:118, models/gma/extractor.py:118) and the 3x3 / stride-2 first convolution of the
// down-sampling residual blocks (extractor.py:23-58 with stride=2; 64 -> 96 and 96 -> 128), which the library path runs
// as a Winograd / implicit-GEMM kernel plus layout transposes.
//
// MI355X formulation: out[n][pixel] = sum_k Wp[n][k] . X[k][pixel] on v_mfma_f32_32x32x2_f32 (exact fp32 products,
// fp32 accumulation) with the OUTPUT CHANNELS as the M operand and the PIXELS as N: a lane's 16 accumulators are 16
// channels of ONE pixel, so a store instruction writes 32 consecutive pixels of a channel row (128 B) and the
// epilogue needs no LDS transpose.  A workgroup (4 waves) owns one output row segment; the input patch of a channel
// chunk is staged in LDS once and read for every tap.  Stride 2 would make neighbouring lanes read every other float
// (2-way bank conflict): the patch is stored DE-INTERLEAVED (even input columns in the first half of a row, odd in
// the second), so tap q of pixel i sits at half (q + LP - pad) & 1, index i + ((q + LP - pad) >> 1) -- consecutive lanes,
// consecutive banks.  The two k of an MFMA step (lane halves) are two input channels, or for the 3-channel stem two
// window rows; that stride is padded to 32 mod 64 floats so the halves land on disjoint banks.  Weights are packed
// once in operand order (one coalesced 256-B load per step and wave) and requested a whole chunk ahead.
#include <cstdlib>
#include "common.hpp"

#ifndef PCFA_S2_DBG
#define PCFA_S2_DBG 0   // timing-only ablation builds (tools/dev): 1 no MFMA, 2 no patch loads, 4 no weight refills, 8 no LDS stores,
                        // 16 no LDS operand reads, 32 no barrier
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int rup(int a, int b) { return (a + b - 1) / b * b; }
constexpr int to32mod64(int a) { return (a % 64 <= 32) ? a + (32 - a % 64) : a + (96 - a % 64); }

// KH x KW window, stride 2, padding K / 2.  CK input channels per chunk.  PAIRROW: the two k of a step are window
// rows 2pp, 2pp + 1 of one channel (odd channel counts: the stem), else channels 2cp, 2cp + 1 of one tap.
// WN waves side by side over 32-channel blocks (they share the staged patch), WPX over pixel groups of MB * 32.
template <int KH_, int KW_, int CK_, bool PAIRROW_, int WN_, int WPX_, int MB_>
struct S2Cfg {
  static constexpr int KH = KH_, KW = KW_, CK = CK_, WN = WN_, MB = MB_;
  static constexpr bool PAIRROW = PAIRROW_;
  static constexpr int WP = WPX_, PXT = WP * MB * 32, NT = 64 * WN * WP;
  static constexpr int PAD = KH / 2, PADW = KW / 2, LP = 4;
  static constexpr int ROWF = rup(2 * (PXT - 1) + KW - PADW + LP, 4);   // input floats staged per patch row
  static constexpr int RV = ROWF / 4, HALF = ROWF / 2;
  static constexpr int ROWS = PAIRROW ? rup(KH, 2) : KH;
  static constexpr int RS = PAIRROW ? to32mod64(ROWF) : ROWF;
  static constexpr int CHS = PAIRROW ? ROWS * RS : to32mod64(ROWS * RS);
  static constexpr int PATCH = CK * CHS;
  static constexpr int STEPS = PAIRROW ? CK * (ROWS / 2) * KW : (CK / 2) * KH * KW;
  static constexpr int DELTA = PAIRROW ? RS : CHS;                      // LDS distance of the second k of a step
  static constexpr int NV = CK * KH * RV, NLOAD = (NV + NT - 1) / NT;   // staging: float4 pieces per chunk, per thread
  static constexpr int PRE = 4;                                        // LDS operand reads run this many steps ahead
  static constexpr int WAVES = PAIRROW ? 2 : (WN >= 2 ? 4 : 3);                        // waves per SIMD the register budget is held to
  static_assert(PAIRROW || CK % 2 == 0, "channel pairs");
  // chunks of the K loop; even unless the layer is a single chunk (two register sets of weights alternate)
  static constexpr int nchunk(int Cin) { return PAIRROW ? 1 : rup((Cin + CK - 1) / CK, 2); }
  static constexpr int step_offset(int s) {
    const int q = s % KW, qq = q + LP - PADW;
    const int col = (qq & 1) * HALF + (qq >> 1);
    if (PAIRROW) {
      const int c = s / ((ROWS / 2) * KW), pp = (s / KW) % (ROWS / 2);
      return c * CHS + 2 * pp * RS + col;
    }
    const int cp = s / (KH * KW), p = (s / KW) % KH;
    return 2 * cp * CHS + p * RS + col;
  }
};

template <class C>
__global__ void conv_s2_pack_kernel(const float* __restrict__ w, float* __restrict__ P, int N, int Cin, int nchunk,
                                    long long total) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63), s = (int)((e >> 6) % C::STEPS);
    const long long blk = (e >> 6) / C::STEPS;
    const int chunk = (int)(blk % nchunk), nb = (int)(blk / nchunk);
    const int n = 32 * nb + (lane & 31), lh = lane >> 5, q = s % C::KW;
    int c, p;
    if (C::PAIRROW) {
      c = chunk * C::CK + s / ((C::ROWS / 2) * C::KW);
      p = 2 * ((s / C::KW) % (C::ROWS / 2)) + lh;
    } else {
      c = chunk * C::CK + 2 * (s / (C::KH * C::KW)) + lh;
      p = (s / C::KW) % C::KH;
    }
    P[e] = (n < N && c < Cin && p < C::KH) ? w[(((long long)n * Cin + c) * C::KH + p) * C::KW + q] : 0.f;
  }
}

// A workgroup walks `rpw` consecutive output rows of its (pixel segment, channel block): the (row, chunk) items form
// one sequence whose next patch is always in flight under the current MFMAs, so only the first item of a workgroup
// pays the global latency, and the single-chunk stem keeps its 84 weight registers for all rows.
template <class C, int ACT>
__global__ __launch_bounds__(C::NT) __attribute__((amdgpu_waves_per_eu(C::WAVES, C::WAVES))) void conv_s2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int Cin, int N, int H, int W, int Ho, int Wo, int tiles_x,
                                                          int rpw, float slope) {
  __shared__ __attribute__((aligned(16))) float smem[2 * C::PATCH + C::HALF + 4];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wv % C::WN, wpx = wv / C::WN;
  const int rblk = blockIdx.x / tiles_x, ox0 = (blockIdx.x - rblk * tiles_x) * C::PXT;
  const int oy0 = rblk * rpw, rows = min(rpw, Ho - oy0);
  const int nb = blockIdx.y * C::WN + wn;
  const long long plane = (long long)H * W;
  x += (long long)blockIdx.z * Cin * plane;
  const int nchunk = C::nchunk(Cin);

  // ---- staging: float4 piece e = tid + NT i of a chunk's patch [channel][row][RV]; (channel, row, column) are
  //      recomputed from e at the load and at the store (a few multiplies per piece) instead of living in registers ----
  auto load_patch = [&](int oy, int chunk, float4 (&rr)[C::NLOAD], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int i = 0; i < C::NLOAD; ++i) {
      const int e = min(tid + C::NT * i, C::NV - 1);
      const int j = e / C::RV, v = e - j * C::RV, c = j / C::KH, r = j - c * C::KH;
      const int ch = chunk * C::CK + c, iy = 2 * oy + r - C::PAD, ix = 2 * ox0 - C::LP + 4 * v;
      okm |= (unsigned)((int)(ix >= 0) & (int)(ix + 3 < W) & (int)(ch < Cin) & (int)(iy >= 0) & (int)(iy < H)) << i;
      // uniform base + 32-bit lane offset (Cin * H * W < 2^31 is checked by the host): one address register per load
      rr[i] = *reinterpret_cast<const float4*>(
          x + (unsigned)(min(ch, Cin - 1) * (int)plane + min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 4)));
    }
  };
  auto store_patch = [&](int buf, const float4 (&rr)[C::NLOAD], unsigned okm) {
#pragma unroll
    for (int i = 0; i < C::NLOAD; ++i) {
      const int e = tid + C::NT * i;
      const int j = e / C::RV, v = e - j * C::RV, c = j / C::KH, r = j - c * C::KH;
      // surplus slots of the last pass write a pad area: no branch in the loop
      const int dst = e < C::NV ? buf * C::PATCH + c * C::CHS + r * C::RS + 2 * v : 2 * C::PATCH;
      const float4 t = (okm >> i & 1u) ? rr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float2*>(smem + dst) = make_float2(t.x, t.z);             // even input columns
      *reinterpret_cast<float2*>(smem + dst + C::HALF) = make_float2(t.y, t.w);   // odd input columns
    }
  };
  const float* pw = wp + ((long long)nb * nchunk * C::STEPS) * 64;   // wave-uniform; + lane at the loads

  f32x16 acc[C::MB];
#pragma unroll
  for (int m = 0; m < C::MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  float4 ra[C::NLOAD], rb[C::NLOAD];
  unsigned oka, okb;
  float wa[C::STEPS], wb[C::STEPS];
  if (C::PAIRROW)   // the pad row of every channel is read against zero weights: it must hold finite numbers
    for (int e = tid; e < 2 * C::PATCH; e += C::NT) smem[e] = 0.f;
  load_patch(oy0, 0, ra, oka);
#pragma unroll
  for (int s = 0; s < C::STEPS; ++s) wa[s] = pw[s * 64 + lane];
  if (C::PAIRROW) __syncthreads();
  store_patch(0, ra, oka);
  __syncthreads();
  const int bl = wpx * C::MB * 32 + l31 + lh * C::DELTA;

  // One item = (row, chunk).  No branch between a load and its use (a conditional prefetch makes the compiler wait for
  // it at once): the last item requests itself again and writes it to the buffer nobody reads.  The weights of the
  // next item are requested at the top of an item into the OTHER register set (two items per trip, no copies).
  // The patch of item it + 2 is requested at the top of item it and written to LDS at the end of item it + 1: two items
  // of flight time with two register sets (an item is ~1200 cycles, an L2 miss more).
  int row = 0, chunk = 0;
  auto advance = [&](int& r, int& c) {   // the item after (r, c); the last item repeats itself
    int nr = r, nc = c + 1;
    if (nc == nchunk) { nc = 0; nr = r + 1; }
    if (nr < rows) { r = nr; c = nc; }
  };
  if (!C::PAIRROW) {
    int r1 = 0, c1 = 0;
    advance(r1, c1);
    load_patch(oy0 + r1, c1, rb, okb);
  }
  auto item = [&](int it, const float (&wcur)[C::STEPS], float (&wnext)[C::STEPS], float4 (&rload)[C::NLOAD],
                  unsigned& okload, const float4 (&rstore)[C::NLOAD], const unsigned& okstore) {
    int nrow = row, nchk = chunk;
    advance(nrow, nchk);
    int r2 = nrow, c2 = nchk;
    if (!C::PAIRROW) advance(r2, c2);   // the stem (84 weight registers) keeps one patch set: one item of flight time
    const float* sp = smem + (it & 1) * C::PATCH + bl;
    if (!(PCFA_S2_DBG & 2)) load_patch(oy0 + r2, c2, rload, okload);
    if (!C::PAIRROW && !(PCFA_S2_DBG & 4)) {
      const float* qn = pw + (long long)nchk * C::STEPS * 64;
#pragma unroll
      for (int s = 0; s < C::STEPS; ++s) wnext[s] = qn[s * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    // pixel operands: LDS reads run PRE steps ahead of their MFMA through a small register ring; the fence after every
    // step keeps the compiler from hoisting all STEPS reads to the top of the item (and spilling)
    float ring[C::PRE][C::MB];
#pragma unroll
    for (int s = 0; s < C::PRE && s < C::STEPS; ++s)
#pragma unroll
      for (int m = 0; m < C::MB; ++m) ring[s][m] = (PCFA_S2_DBG & 16) ? 1.f : sp[32 * m + C::step_offset(s)];
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s) {
      float cur[C::MB];
#pragma unroll
      for (int m = 0; m < C::MB; ++m) cur[m] = ring[s % C::PRE][m];
      if (s + C::PRE < C::STEPS)
#pragma unroll
        for (int m = 0; m < C::MB; ++m)
          ring[s % C::PRE][m] = (PCFA_S2_DBG & 16) ? 1.f : sp[32 * m + C::step_offset(s + C::PRE)];
#pragma unroll
      for (int m = 0; m < C::MB; ++m)
        if (PCFA_S2_DBG & 1)
          acc[m][s & 15] += wcur[s] * cur[m];
        else
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[s], cur[m], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(PCFA_S2_DBG & 8)) store_patch((it + 1) & 1, rstore, okstore);
    if (!(PCFA_S2_DBG & 32)) __syncthreads();
    if (chunk == nchunk - 1) {
      // ---- epilogue: lane = pixel, register r = channel 8 (r >> 2) + 4 lh + (r & 3) of the wave's block ----
      // uniform row base per register (scalar arithmetic) + one 32-bit lane offset: no per-register address pairs
      const int howo = Ho * Wo;
      float* ob = out + ((long long)blockIdx.z * N + 32 * nb) * howo + (long long)(oy0 + row) * Wo + ox0 + wpx * C::MB * 32;
      const int loff = 4 * lh * howo + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nu = 8 * (r >> 2) + (r & 3);                  // + 4 lh: the lane part
        const int n = 32 * nb + nu + 4 * lh;
        const float bvr = bias != nullptr ? bias[min(n, N - 1)] : 0.f;
        float* orow = ob + (long long)nu * howo;
#pragma unroll
        for (int m = 0; m < C::MB; ++m) {
          float y = acc[m][r] + bvr;
          if (ACT == 1) y = fmaxf(y, 0.f);
          if (ACT == 2) y = y > 0.f ? y : y * slope;
          if (ox0 + (wpx * C::MB + m) * 32 + l31 < Wo && n < N) orow[loff + 32 * m] = y;
          acc[m][r] = 0.f;
        }
      }
    }
    row = nrow;
    chunk = nchk;
  };
  const int nitems = rows * nchunk;
  if (C::PAIRROW) {   // one chunk per row, the weights stay
    for (int it = 0; it < nitems; ++it) item(it, wa, wa, ra, oka, ra, oka);
  } else {   // nchunk is even (the packing pads it): items come in pairs
    for (int it = 0; it < nitems; it += 2) {
      item(it, wa, wb, ra, oka, rb, okb);
      item(it + 1, wb, wa, rb, okb, ra, oka);
    }
  }
}

typedef S2Cfg<7, 7, 3, true, 2, 2, 2> StemCfg;    // 3 -> N, 7x7: 64 channels x 128 pixels per workgroup
// C -> N, 3x3: WN waves = WN 32-channel blocks on one patch of 32 MB pixels (1 or 2 waves along the pixels for few blocks)
#ifndef PCFA_S2_CK
#define PCFA_S2_CK 4
#endif
template <int WN, int MB, int WP = (WN == 1 ? 4 : WN == 2 ? 2 : 1)>
using Res3 = S2Cfg<3, 3, PCFA_S2_CK, false, WN, WP, MB>;
typedef Res3<4, 1> Res3Cfg;   // the weight packing does not depend on WN / MB beyond the block padding to 4

template <class C>
long long packed_floats_t(int Cin, int N) {
  const long long nchunk = C::nchunk(Cin), nblk = rup((N + 31) / 32, C::WN);
  return nblk * nchunk * C::STEPS * 64;
}

template <class C>
int pack_t(const float* w, float* packed, int Cin, int N, hipStream_t s) {
  const long long total = packed_floats_t<C>(Cin, N);
  pcfa_launch(conv_s2_pack_kernel<C>, dim3((unsigned)min((total + 255) / 256, 4096LL)), dim3(256), 0, s, w, packed, N,
              Cin, C::nchunk(Cin), total);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

template <class C>
int fwd_t(const float* x, const float* packed, const float* bias, float* out, int B, int Cin, int N, int H, int W, int act,
          float slope, hipStream_t s) {
  const int Ho = (H + 2 * C::PAD - C::KH) / 2 + 1, Wo = (W + 2 * C::PADW - C::KW) / 2 + 1;
  const int tiles_x = pcfa_cdiv(Wo, C::PXT), nby = rup((N + 31) / 32, C::WN) / C::WN;
  // rows per workgroup: as many as still leave >= 6 workgroups per CU (dev override PCFA_S2_RPW)
  static const int rpw_env = getenv("PCFA_S2_RPW") ? atoi(getenv("PCFA_S2_RPW")) : 0;
  int rpw = 1;
  while (rpw < 16 && (long long)tiles_x * pcfa_cdiv(Ho, rpw * 2) * nby * B >= 6 * 256) rpw *= 2;
  if (rpw_env > 0) rpw = rpw_env;
  dim3 grid((unsigned)(tiles_x * pcfa_cdiv(Ho, rpw)), (unsigned)nby, (unsigned)B);
#define PCFA_S2_GO(A_)                                                                                              \
  pcfa_launch(conv_s2_fwd_kernel<C, A_>, grid, dim3(C::NT), 0, s, x, packed, bias, out, Cin, N, H, W, Ho, Wo, tiles_x, \
              rpw, slope)
  if (act == 1) PCFA_S2_GO(1); else if (act == 2) PCFA_S2_GO(2); else PCFA_S2_GO(0);
#undef PCFA_S2_GO
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

bool is_stem(int Cin, int ksize) { return ksize == 7 && Cin == 3; }

}  // namespace

extern "C" {

int pcfa_conv_s2_supported(int Cin, int N, int ksize, int H, int W) {
  if (Cin < 1 || N < 1 || H < 2 || W < 4 || W % 4 != 0) return 0;
  if ((long long)Cin * H * W > 0x7fffffffLL || (long long)N * H * W > 0x7fffffffLL) return 0;
  return is_stem(Cin, ksize) || ksize == 3;
}

long long pcfa_conv_s2_packed_floats(int Cin, int N, int ksize) {
  if (is_stem(Cin, ksize)) return packed_floats_t<StemCfg>(Cin, N);
  if (ksize == 3) return packed_floats_t<Res3Cfg>(Cin, N);
  return 0;
}

int pcfa_conv_s2_pack(const float* w, float* packed, int Cin, int N, int ksize, void* stream) {
  if (w == nullptr || packed == nullptr) return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (is_stem(Cin, ksize)) return pack_t<StemCfg>(w, packed, Cin, N, s);
  if (ksize == 3) return pack_t<Res3Cfg>(w, packed, Cin, N, s);
  return PCFA_ERR_INVALID_ARG;
}

int pcfa_conv_s2_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int Cin, int N, int H,
                     int W, int ksize, int act, float slope, void* stream) {
  if (x == nullptr || packed == nullptr || out == nullptr || B < 1 || act < 0 || act > 2) return PCFA_ERR_INVALID_ARG;
  if (!pcfa_conv_s2_supported(Cin, N, ksize, H, W)) return PCFA_ERR_UNSUPPORTED;
  if (((uintptr_t)x & 15) != 0) return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (is_stem(Cin, ksize)) return fwd_t<StemCfg>(x, packed, bias, out, B, Cin, N, H, W, act, slope, s);
  // as many channel blocks per workgroup as the layer has (up to four; they share the staged patch), 32 pixels per
  // wave (64 measured 2-5 % slower at the encoder shapes, one-wave workgroups with private patches 10-70 % slower)
  static const int wn_env = getenv("PCFA_S2_WN") ? atoi(getenv("PCFA_S2_WN")) : 0;   // dev override
  const int nblk = (N + 31) / 32, wn = wn_env ? wn_env : (nblk >= 4 ? 4 : nblk);
#define PCFA_S2_RES(WN_) return fwd_t<Res3<WN_, 1>>(x, packed, bias, out, B, Cin, N, H, W, act, slope, s)
  if (wn == 4) PCFA_S2_RES(4);
  if (wn == 3) PCFA_S2_RES(3);
  if (wn == 2) PCFA_S2_RES(2);
  PCFA_S2_RES(1);
#undef PCFA_S2_RES
}

}  // extern "C"
