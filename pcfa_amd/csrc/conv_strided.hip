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
// DS: the workgroup also evaluates a 1x1 / stride-2 convolution of the same input (the residual blocks' downsample,
// extractor.py:44-46): its input pixel is the centre tap of the 3x3 window, so it costs CK / 2 more steps per chunk on
// operands already in registers, into a second accumulator.
template <int KH_, int KW_, int CK_, bool PAIRROW_, int WN_, int WPX_, int MB_, bool DS_ = false>
struct S2Cfg {
  static constexpr bool DS = DS_;
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
  static constexpr int STEPS3 = PAIRROW ? CK * (ROWS / 2) * KW : (CK / 2) * KH * KW;
  static constexpr int STEPS = STEPS3 + (DS ? CK / 2 : 0);             // steps >= STEPS3: the 1x1 convolution's
  static_assert(!DS || (!PAIRROW && KH == 3 && KW == 3), "the fused 1x1 rides on the 3x3 channel-pair layout");
  static constexpr int DELTA = PAIRROW ? RS : CHS;                      // LDS distance of the second k of a step
  static constexpr int NV = CK * KH * RV, NLOAD = (NV + NT - 1) / NT;   // staging: float4 pieces per chunk, per thread
#ifndef PCFA_S2_PRE
#define PCFA_S2_PRE 4
#endif
  static constexpr int PRE = PCFA_S2_PRE;                              // LDS operand reads run this many steps ahead
  static constexpr int WAVES = PAIRROW ? 2 : (WN >= 2 && !DS ? 4 : 3);                        // waves per SIMD the register budget is held to
  static_assert(PAIRROW || CK % 2 == 0, "channel pairs");
  // chunks of the K loop; even unless the layer is a single chunk (two register sets of weights alternate)
  static constexpr int nchunk(int Cin) { return PAIRROW ? 1 : rup((Cin + CK - 1) / CK, 2); }
  static constexpr int step_offset(int s) {
    if (s >= STEPS3) return step_offset(((s - STEPS3) * KH + PAD) * KW + PADW);   // centre tap of channel pair s - STEPS3
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
__global__ void conv_s2_pack_kernel(const float* __restrict__ w, const float* __restrict__ wd, float* __restrict__ P, int N,
                                    int Cin, int nchunk, long long total) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63), s = (int)((e >> 6) % C::STEPS);
    const long long blk = (e >> 6) / C::STEPS;
    const int chunk = (int)(blk % nchunk), nb = (int)(blk / nchunk);
    const int n = 32 * nb + (lane & 31), lh = lane >> 5, q = s % C::KW;
    int c, p;
    if (s >= C::STEPS3) {   // wd[N][Cin] of the fused 1x1
      c = chunk * C::CK + 2 * (s - C::STEPS3) + lh;
      P[e] = (n < N && c < Cin) ? wd[(long long)n * Cin + c] : 0.f;
      continue;
    }
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
__global__ __launch_bounds__(C::NT) __attribute__((amdgpu_waves_per_eu(C::WAVES, C::WAVES))) void conv_s2_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ out,
    const float* __restrict__ bias_d, float* __restrict__ out_d, int Cin, int N, int H, int W, int Ho, int Wo, int tiles_x,
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

  f32x16 acc[C::MB], accd[C::DS ? C::MB : 1];
#pragma unroll
  for (int m = 0; m < C::MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[m][r] = 0.f;
      if (C::DS) accd[m][r] = 0.f;
    }

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
        else if (C::DS && s >= C::STEPS3)
          accd[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[s], cur[m], accd[m], 0, 0, 0);
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
        if (C::DS) {   // the 1x1 convolution's output: bias only (its normalisation follows in the caller)
          const float bdr = bias_d != nullptr ? bias_d[min(n, N - 1)] : 0.f;
          float* drow = out_d + (orow - out);
#pragma unroll
          for (int m = 0; m < C::MB; ++m) {
            if (ox0 + (wpx * C::MB + m) * 32 + l31 < Wo && n < N) drow[loff + 32 * m] = accd[m][r] + bdr;
            accd[m][r] = 0.f;
          }
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

#ifndef PCFA_S2_STEM_MB
#define PCFA_S2_STEM_MB 2
#endif
#ifndef PCFA_S2_STEM_WN
#define PCFA_S2_STEM_WN 2
#endif
typedef S2Cfg<7, 7, 3, true, PCFA_S2_STEM_WN, 4 / PCFA_S2_STEM_WN, PCFA_S2_STEM_MB> StemCfg;    // 3 -> N, 7x7: 64 channels x 128 pixels per workgroup
// C -> N, 3x3: WN waves = WN 32-channel blocks on one patch of 32 MB pixels (1 or 2 waves along the pixels for few blocks)
#ifndef PCFA_S2_CK
#define PCFA_S2_CK 4
#endif
template <int WN, int MB, bool DS = false, int WP = (WN == 1 ? 4 : WN == 2 ? 2 : 1)>
using Res3 = S2Cfg<3, 3, PCFA_S2_CK, false, WN, WP, MB, DS>;
typedef Res3<4, 1> Res3Cfg;        // the weight packing does not depend on WN / MB beyond the block padding to 4
typedef Res3<4, 1, true> Res3DsCfg;

template <class C>
long long packed_floats_t(int Cin, int N) {
  const long long nchunk = C::nchunk(Cin), nblk = rup((N + 31) / 32, C::WN);
  return nblk * nchunk * C::STEPS * 64;
}

template <class C>
int pack_t(const float* w, const float* wd, float* packed, int Cin, int N, hipStream_t s) {
  const long long total = packed_floats_t<C>(Cin, N);
  pcfa_launch(conv_s2_pack_kernel<C>, dim3((unsigned)min((total + 255) / 256, 4096LL)), dim3(256), 0, s, w, wd, packed, N,
              Cin, C::nchunk(Cin), total);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

template <class C>
int fwd_t(const float* x, const float* packed, const float* bias, float* out, const float* bias_d, float* out_d, int B,
          int Cin, int N, int H, int W, int act, float slope, hipStream_t s) {
  const int Ho = (H + 2 * C::PAD - C::KH) / 2 + 1, Wo = (W + 2 * C::PADW - C::KW) / 2 + 1;
  const int tiles_x = pcfa_cdiv(Wo, C::PXT), nby = rup((N + 31) / 32, C::WN) / C::WN;
  // rows per workgroup: as many as still leave >= 6 workgroups per CU (dev override PCFA_S2_RPW)
  static const int rpw_env = getenv("PCFA_S2_RPW") ? atoi(getenv("PCFA_S2_RPW")) : 0;
  int rpw = 1;
  while (rpw < 16 && (long long)tiles_x * pcfa_cdiv(Ho, rpw * 2) * nby * B >= 6 * 256) rpw *= 2;
  if (rpw_env > 0) rpw = rpw_env;
  dim3 grid((unsigned)(tiles_x * pcfa_cdiv(Ho, rpw)), (unsigned)nby, (unsigned)B);
#define PCFA_S2_GO(A_)                                                                                              \
  pcfa_launch(conv_s2_fwd_kernel<C, A_>, grid, dim3(C::NT), 0, s, x, packed, bias, out, bias_d, out_d, Cin, N, H, W, Ho, \
              Wo, tiles_x, rpw, slope)
  if (act == 1) PCFA_S2_GO(1); else if (act == 2) PCFA_S2_GO(2); else PCFA_S2_GO(0);
#undef PCFA_S2_GO
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Data gradient of the 3x3 / stride-2 convolution: dx[c][y][x] = sum_n sum_pq g[n][oy][ox] w[n][c][p][q] with
// y = 2 oy - 1 + p, x = 2 ox - 1 + q.  By output parity (a, b) = (y & 1, x & 1) it is four stride-1 convolutions of g:
//   a = 0: p = 1 (oy = i)            a = 1: p = 0 (oy = i + 1), p = 2 (oy = i)          (y = 2 i + a; same for b, q, j)
// so every (p, q) feeds exactly one parity class: a wave keeps the four classes of its 32 coarse pixels (two fine rows
// x 64 fine pixels) in four accumulators and issues ONE MFMA per (channel pair, p, q) -- the same 9 MFMAs per channel
// pair as the forward, no zero taps, nothing scattered.  The two x-classes of a pixel sit in one lane, so the epilogue
// writes float2 (256 B per instruction and channel row).  g needs no de-interleave (stride 1 on the coarse grid).
// Layout as the forward: channels of dx = M (WN blocks of 32 per workgroup share the patch), coarse pixels = N.
// ---------------------------------------------------------------------------------------------------------------
// DS: the gradient of the fused 1x1 / stride-2 convolution (dx[c][2 i][2 j] += sum_n gd[n][i][j] wd[n][c]) rides along:
// a second patch region holds row i of gd, CK / 2 more steps per chunk feed the (0, 0) class.
template <int WN_, int WP_, bool DS_ = false>
struct S2BwdCfg {
  static constexpr bool DS = DS_;
  static constexpr int WN = WN_, WP = WP_, NT = 64 * WN * WP, CK = 4, STEPS3 = (CK / 2) * 9, STEPS = STEPS3 + (DS ? CK / 2 : 0);
  static constexpr int PXC = 32 * WP;                       // coarse pixels per workgroup
  static constexpr int RV = PXC / 4 + 1, RSB = 4 * RV;      // float4 per patch row (one more for ox = j + 1), row stride
  static constexpr int CHS = 2 * RSB;                       // [channel][row i, i + 1][RSB]
  static constexpr int PATCH = (DS ? 2 : 1) * CK * CHS;     // DS: + [channel][row i of gd, unused][RSB]
  static constexpr int NV3 = CK * 2 * RV, NV = NV3 + (DS ? CK * RV : 0), NLOAD = (NV + NT - 1) / NT;
  static constexpr int PRE = 4;
  static constexpr int WAVES = 3;
  static constexpr int step_offset(int s) {                 // LDS offset of (pair, p, q): channel 2 pair, row (p == 0), column (q == 0)
    if (s >= STEPS3) return CK * CHS + 2 * (s - STEPS3) * CHS;
    const int pair = s / 9, p = (s / 3) % 3, q = s % 3;
    return 2 * pair * CHS + (p == 0 ? RSB : 0) + (q == 0 ? 1 : 0);
  }
  static constexpr int step_class(int s) {                  // accumulator 2 a + b
    if (s >= STEPS3) return 0;
    const int p = (s / 3) % 3, q = s % 3;
    return 2 * (p == 1 ? 0 : 1) + (q == 1 ? 0 : 1);
  }
};

// packed[cb][chunk][step][lane]: A operand of step (pair, p, q): w[n = chunk CK + 2 pair + (lane >> 5)][c = 32 cb + (lane & 31)][p][q]
template <bool DS>
__global__ void conv_s2_bwd_pack_kernel(const float* __restrict__ w, const float* __restrict__ wd, float* __restrict__ P,
                                        int N, int Cin, int nchunk, long long total) {
  constexpr int STEPS = DS ? 20 : 18, CK = 4;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63), s = (int)((e >> 6) % STEPS);
    const long long blk = (e >> 6) / STEPS;
    const int chunk = (int)(blk % nchunk), cb = (int)(blk / nchunk);
    const int c = 32 * cb + (lane & 31);
    if (s >= 18) {
      const int n = chunk * CK + 2 * (s - 18) + (lane >> 5);
      P[e] = (n < N && c < Cin) ? wd[(long long)n * Cin + c] : 0.f;
      continue;
    }
    const int n = chunk * CK + 2 * (s / 9) + (lane >> 5), p = (s / 3) % 3, q = s % 3;
    P[e] = (n < N && c < Cin) ? w[(((long long)n * Cin + c) * 3 + p) * 3 + q] : 0.f;
  }
}

template <class C>
__global__ __launch_bounds__(C::NT) __attribute__((amdgpu_waves_per_eu(C::WAVES, C::WAVES))) void conv_s2_bwd_kernel(
    const float* __restrict__ g, const float* __restrict__ gd, const float* __restrict__ wp, float* __restrict__ dx,
    int Cin, int N, int H, int W, int Ho, int Wo, int tiles_x) {
  __shared__ __attribute__((aligned(16))) float smem[2 * C::PATCH + 4];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wv % C::WN, wpx = wv / C::WN;
  const int i = blockIdx.x / tiles_x, j0 = (blockIdx.x - i * tiles_x) * C::PXC;   // coarse row, first coarse column
  const int cb = blockIdx.y * C::WN + wn;
  const int gplane = Ho * Wo;
  g += (long long)blockIdx.z * N * gplane;
  if (C::DS) gd += (long long)blockIdx.z * N * gplane;
  const int nchunk = (N + C::CK - 1) / C::CK + (((N + C::CK - 1) / C::CK) & 1);   // even (two weight register sets)

  auto load_patch = [&](int chunk, float4 (&rr)[C::NLOAD], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int k = 0; k < C::NLOAD; ++k) {
      const int e = min(tid + C::NT * k, C::NV - 1);
      const bool third = C::DS && e >= C::NV3;                 // a piece of gd's row i
      const int jj = (third ? e - C::NV3 : e) / C::RV, v = (third ? e - C::NV3 : e) - jj * C::RV;
      const int c = third ? jj : jj >> 1, r = third ? 0 : jj & 1;
      const int n = chunk * C::CK + c, oy = i + r, ox = j0 + 4 * v;
      okm |= (unsigned)((int)(n < N) & (int)(oy < Ho) & (int)(ox + 3 < Wo)) << k;
      rr[k] = *reinterpret_cast<const float4*>((third ? gd : g) +
                                               (unsigned)(min(n, N - 1) * gplane + min(oy, Ho - 1) * Wo + min(ox, Wo - 4)));
    }
  };
  auto store_patch = [&](int buf, const float4 (&rr)[C::NLOAD], unsigned okm) {
#pragma unroll
    for (int k = 0; k < C::NLOAD; ++k) {
      const int e = tid + C::NT * k;
      const bool third = C::DS && e >= C::NV3;
      const int jj = (third ? e - C::NV3 : e) / C::RV, v = (third ? e - C::NV3 : e) - jj * C::RV;
      const int c = third ? C::CK + jj : jj >> 1, r = third ? 0 : jj & 1;
      const float4 t = (okm >> k & 1u) ? rr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < C::NV) *reinterpret_cast<float4*>(smem + buf * C::PATCH + c * C::CHS + r * C::RSB + 4 * v) = t;
    }
  };
  const float* pw = wp + ((long long)cb * nchunk * C::STEPS) * 64;   // wave-uniform; + lane at the loads

  f32x16 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  float4 ra[C::NLOAD], rb[C::NLOAD];
  unsigned oka, okb;
  float wa[C::STEPS], wb[C::STEPS];
  load_patch(0, ra, oka);
#pragma unroll
  for (int s = 0; s < C::STEPS; ++s) wa[s] = pw[s * 64 + lane];
  store_patch(0, ra, oka);
  __syncthreads();
  load_patch(min(1, nchunk - 1), rb, okb);
  const int bl = wpx * 32 + l31 + lh * C::CHS;

  auto item = [&](int chunk, const float (&wcur)[C::STEPS], float (&wnext)[C::STEPS], float4 (&rload)[C::NLOAD],
                  unsigned& okload, const float4 (&rstore)[C::NLOAD], const unsigned& okstore) {
    const float* sp = smem + (chunk & 1) * C::PATCH + bl;
    load_patch(min(chunk + 2, nchunk - 1), rload, okload);
    const float* qn = pw + (long long)min(chunk + 1, nchunk - 1) * C::STEPS * 64;
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s) wnext[s] = qn[s * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    float ring[C::PRE];
#pragma unroll
    for (int s = 0; s < C::PRE; ++s) ring[s] = sp[C::step_offset(s)];
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s) {
      const float cur = ring[s % C::PRE];
      if (s + C::PRE < C::STEPS) ring[s % C::PRE] = sp[C::step_offset(s + C::PRE)];
      acc[C::step_class(s)] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[s], cur, acc[C::step_class(s)], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_patch((chunk + 1) & 1, rstore, okstore);
    __syncthreads();
  };
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
    item(chunk, wa, wb, ra, oka, rb, okb);
    item(chunk + 1, wb, wa, rb, okb, ra, oka);
  }

  // ---- epilogue: lane = coarse pixel j, register r = channel 8 (r >> 2) + 4 lh + (r & 3); the two x-classes of a
  //      pixel are neighbours in the row: float2 stores ----
  const int plane = H * W;
  float* ob = dx + ((long long)blockIdx.z * Cin + 32 * cb) * plane + 2 * (j0 + wpx * 32);
  const int jx = j0 + wpx * 32 + l31;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int y = 2 * i + a;
    if (y < H && 2 * jx < W) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nu = 8 * (r >> 2) + (r & 3);
        if (32 * cb + nu + 4 * lh < Cin)
          *reinterpret_cast<float2*>(ob + (long long)nu * plane + (unsigned)(4 * lh * plane + y * W + 2 * l31)) =
              make_float2(acc[2 * a][r], acc[2 * a + 1][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Data gradient of the 7x7 / stride-2 stem (3 <- N channels).  With y = 2 i + a, x = 2 j + b:
//   dx[c][2 i + a][2 j + b] = sum_n sum_{di, dj in -1..2} g[n][i + di][j + dj] w[n][c][a + 3 - 2 di][b + 3 - 2 dj]
// (weights outside 0..6 are zero): a stride-1 4x4-tap convolution of g onto 12 "phase channels" (c, a, b).  Twelve rows
// are too few for the 32x32 tiles, so this one runs on v_mfma_f32_16x16x4_f32: M = 16 = (c, a, b) + 4 idle rows,
// N = 16 coarse pixels, K = 4 channels of g at one tap.  Its D layout puts row 4 (lane >> 4) + r in register r of a
// lane: lane group = colour channel c, register = parity class, so a lane stores float2 (x = 2 j, 2 j + 1) for the two
// fine rows -- no scatter, no col2im buffer (the library path: a GEMM into a 147 x pixels matrix + Col2Im2dU).
// A wave owns 64 coarse pixels of one coarse row (four pixel blocks share each weight operand), a workgroup four rows;
// the weight operand stream (16 taps x N / 4) is packed once in lane order and requested a chunk ahead.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct StemBwd {
  static constexpr int CK = 8, KG = CK / 4, TAPS = 16, STEPS = KG * TAPS;   // chunk: 8 channels of g = 2 k-groups x 16 taps
  static constexpr int MB = 4, PXC = 16 * MB, TR = 4, NT = 256;             // wave: 64 coarse pixels; workgroup: 4 coarse rows
  static constexpr int RV = (PXC + 8) / 4, RS = 4 * RV;                     // patch row: columns j0 - 4 .. j0 + 67
  static constexpr int PROWS = TR + 3;                                      // coarse rows i0 - 1 .. i0 + 5
  static constexpr int CHS = (PROWS * RS + 31) / 32 * 32 + 16;              // = 16 (mod 32): the four k lanes groups of a
  static constexpr int PATCH = CK * CHS;                                    //   ds_read_b32 half land on disjoint banks
  static constexpr int NV = CK * PROWS * RV, NLOAD = (NV + NT - 1) / NT;
  static constexpr int PRE = 3;
  static constexpr int step_offset(int s) {   // (k-group, tap (di + 1) * 4 + (dj + 1)): channel 4 kg, row di + 1, column dj + 4
    const int kg = s / TAPS, t = s % TAPS;
    return 4 * kg * CHS + (t / 4) * RS + (t % 4) + 3;
  }
};

// packed[chunk][step][lane]: A operand: row m = lane & 15 = 4 c + 2 a + b, k = lane >> 4: n = 8 chunk + 4 kg + k
__global__ void conv_s2_stem_bwd_pack_kernel(const float* __restrict__ w, float* __restrict__ P, int N, long long total) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63), s = (int)((e >> 6) % StemBwd::STEPS), chunk = (int)((e >> 6) / StemBwd::STEPS);
    const int m = lane & 15, c = m >> 2, a = (m >> 1) & 1, b = m & 1;
    const int kg = s / 16, t = s % 16, di = t / 4 - 1, dj = t % 4 - 1;
    const int n = 8 * chunk + 4 * kg + (lane >> 4), p = a + 3 - 2 * di, q = b + 3 - 2 * dj;
    P[e] = (c < 3 && n < N && p >= 0 && p < 7 && q >= 0 && q < 7) ? w[(((long long)n * 3 + c) * 7 + p) * 7 + q] : 0.f;
  }
}

__global__ __launch_bounds__(StemBwd::NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_s2_stem_bwd_kernel(
    const float* __restrict__ g, const float* __restrict__ wp, float* __restrict__ dx, int N, int H, int W, int Ho, int Wo,
    int tiles_x) {
  typedef StemBwd C;
  __shared__ __attribute__((aligned(16))) float smem[2 * C::PATCH + 4];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rblk = blockIdx.x / tiles_x, j0 = (blockIdx.x - rblk * tiles_x) * C::PXC, i0 = rblk * C::TR;
  const int gplane = Ho * Wo;
  g += (long long)blockIdx.z * N * gplane;
  const int nchunk = ((N + C::CK - 1) / C::CK + 1) & ~1;   // even: two weight register sets alternate

  auto load_patch = [&](int chunk, float4 (&rr)[C::NLOAD], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int k = 0; k < C::NLOAD; ++k) {
      const int e = min(tid + C::NT * k, C::NV - 1);
      const int jj = e / C::RV, v = e - jj * C::RV, c = jj / C::PROWS, r = jj - c * C::PROWS;
      const int n = chunk * C::CK + c, oy = i0 - 1 + r, ox = j0 - 4 + 4 * v;
      okm |= (unsigned)((int)(n < N) & (int)(oy >= 0) & (int)(oy < Ho) & (int)(ox >= 0) & (int)(ox + 3 < Wo)) << k;
      rr[k] = *reinterpret_cast<const float4*>(
          g + (unsigned)(min(n, N - 1) * gplane + min(max(oy, 0), Ho - 1) * Wo + min(max(ox, 0), Wo - 4)));
    }
  };
  auto store_patch = [&](int buf, const float4 (&rr)[C::NLOAD], unsigned okm) {
#pragma unroll
    for (int k = 0; k < C::NLOAD; ++k) {
      const int e = tid + C::NT * k;
      const int jj = e / C::RV, v = e - jj * C::RV, c = jj / C::PROWS, r = jj - c * C::PROWS;
      const float4 t = (okm >> k & 1u) ? rr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < C::NV) *reinterpret_cast<float4*>(smem + buf * C::PATCH + c * C::CHS + r * C::RS + 4 * v) = t;
    }
  };

  f32x4v acc[C::MB];
#pragma unroll
  for (int m = 0; m < C::MB; ++m) acc[m] = f32x4v{0.f, 0.f, 0.f, 0.f};

  float4 ra[C::NLOAD], rb[C::NLOAD];
  unsigned oka, okb;
  float wa[C::STEPS], wb[C::STEPS];
  load_patch(0, ra, oka);
#pragma unroll
  for (int s = 0; s < C::STEPS; ++s) wa[s] = wp[s * 64 + lane];
  store_patch(0, ra, oka);
  __syncthreads();
  load_patch(1, rb, okb);
  const int bl = lg * C::CHS + wv * C::RS + l15;   // k lane group -> channel, wave -> row of the tile, lane -> pixel

  auto item = [&](int chunk, const float (&wcur)[C::STEPS], float (&wnext)[C::STEPS], float4 (&rload)[C::NLOAD],
                  unsigned& okload, const float4 (&rstore)[C::NLOAD], const unsigned& okstore) {
    const float* sp = smem + (chunk & 1) * C::PATCH + bl;
    load_patch(min(chunk + 2, nchunk - 1), rload, okload);
    const float* qn = wp + (long long)min(chunk + 1, nchunk - 1) * C::STEPS * 64;
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s) wnext[s] = qn[s * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    float ring[C::PRE][C::MB];
#pragma unroll
    for (int s = 0; s < C::PRE; ++s)
#pragma unroll
      for (int m = 0; m < C::MB; ++m) ring[s][m] = sp[16 * m + C::step_offset(s)];
#pragma unroll
    for (int s = 0; s < C::STEPS; ++s) {
      float cur[C::MB];
#pragma unroll
      for (int m = 0; m < C::MB; ++m) cur[m] = ring[s % C::PRE][m];
      if (s + C::PRE < C::STEPS)
#pragma unroll
        for (int m = 0; m < C::MB; ++m) ring[s % C::PRE][m] = sp[16 * m + C::step_offset(s + C::PRE)];
#pragma unroll
      for (int m = 0; m < C::MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[s], cur[m], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_patch((chunk + 1) & 1, rstore, okstore);
    __syncthreads();
  };
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
    item(chunk, wa, wb, ra, oka, rb, okb);
    item(chunk + 1, wb, wa, rb, okb, ra, oka);
  }

  // ---- epilogue: lane group = colour channel, register 2 a + b = parity class: float2 per fine row ----
  const int i = i0 + wv;
  if (lg < 3) {
    float* ob = dx + ((long long)blockIdx.z * 3 + lg) * H * W;
#pragma unroll
    for (int m = 0; m < C::MB; ++m) {
      const int j = j0 + 16 * m + l15;
      if (2 * j < W) {
        if (2 * i < H) *reinterpret_cast<float2*>(ob + (long long)(2 * i) * W + 2 * j) = make_float2(acc[m][0], acc[m][1]);
        if (2 * i + 1 < H)
          *reinterpret_cast<float2*>(ob + (long long)(2 * i + 1) * W + 2 * j) = make_float2(acc[m][2], acc[m][3]);
      }
    }
  }
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
  if (is_stem(Cin, ksize)) return pack_t<StemCfg>(w, nullptr, packed, Cin, N, s);
  if (ksize == 3) return pack_t<Res3Cfg>(w, nullptr, packed, Cin, N, s);
  return PCFA_ERR_INVALID_ARG;
}

long long pcfa_conv_s2_ds_packed_floats(int Cin, int N) { return packed_floats_t<Res3DsCfg>(Cin, N); }

int pcfa_conv_s2_ds_pack(const float* w, const float* wd, float* packed, int Cin, int N, void* stream) {
  if (w == nullptr || wd == nullptr || packed == nullptr) return PCFA_ERR_INVALID_ARG;
  return pack_t<Res3DsCfg>(w, wd, packed, Cin, N, (hipStream_t)stream);
}

int pcfa_conv_s2_ds_fwd(const float* x, const float* packed, const float* bias, float* out, const float* bias_d,
                        float* out_d, int B, int Cin, int N, int H, int W, int act, float slope, void* stream) {
  if (x == nullptr || packed == nullptr || out == nullptr || out_d == nullptr || B < 1 || act < 0 || act > 2)
    return PCFA_ERR_INVALID_ARG;
  if (!pcfa_conv_s2_supported(Cin, N, 3, H, W)) return PCFA_ERR_UNSUPPORTED;
  if (((uintptr_t)x & 15) != 0) return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (N + 31) / 32, wn = nblk >= 4 ? 4 : nblk;
#define PCFA_S2_RES(WN_) return fwd_t<Res3<WN_, 1, true>>(x, packed, bias, out, bias_d, out_d, B, Cin, N, H, W, act, slope, s)
  if (wn == 4) PCFA_S2_RES(4);
  if (wn == 3) PCFA_S2_RES(3);
  if (wn == 2) PCFA_S2_RES(2);
  PCFA_S2_RES(1);
#undef PCFA_S2_RES
}

static long long s2_bwd_floats(int Cin, int N, int steps) {
  const long long nchunk = (N + 3) / 4 + (((N + 3) / 4) & 1), nblk = rup((Cin + 31) / 32, 4);
  return nblk * nchunk * steps * 64;
}
static long long stem_bwd_floats(int N) {
  return (long long)(((N + StemBwd::CK - 1) / StemBwd::CK + 1) & ~1) * StemBwd::STEPS * 64;
}
long long pcfa_conv_s2_bwd_packed_floats(int Cin, int N, int ksize) {
  if (is_stem(Cin, ksize)) return stem_bwd_floats(N);
  return ksize == 3 ? s2_bwd_floats(Cin, N, 18) : 0;
}
long long pcfa_conv_s2_ds_bwd_packed_floats(int Cin, int N) { return s2_bwd_floats(Cin, N, 20); }

int pcfa_conv_s2_bwd_supported(int Cin, int N, int ksize, int H, int W) {
  if (!(ksize == 3 || is_stem(Cin, ksize)) || Cin < 1 || N < 1 || H < 2 || W < 8 || W % 8 != 0) return 0;    // Wo % 4 == 0: float4 rows of g
  if ((long long)Cin * H * W > 0x7fffffffLL || (long long)N * H * W > 0x7fffffffLL) return 0;
  return 1;
}

int pcfa_conv_s2_bwd_pack(const float* w, float* packed, int Cin, int N, int ksize, void* stream) {
  if (w == nullptr || packed == nullptr) return PCFA_ERR_INVALID_ARG;
  if (is_stem(Cin, ksize)) {
    const long long total = stem_bwd_floats(N);
    pcfa_launch(conv_s2_stem_bwd_pack_kernel, dim3((unsigned)min((total + 255) / 256, 4096LL)), dim3(256), 0,
                (hipStream_t)stream, w, packed, N, total);
    PCFA_LAUNCH_CHECK();
    return PCFA_OK;
  }
  if (ksize != 3) return PCFA_ERR_UNSUPPORTED;
  const long long total = pcfa_conv_s2_bwd_packed_floats(Cin, N, ksize);
  pcfa_launch(conv_s2_bwd_pack_kernel<false>, dim3((unsigned)min((total + 255) / 256, 4096LL)), dim3(256), 0,
              (hipStream_t)stream, w, (const float*)nullptr, packed, N, Cin, (int)((N + 3) / 4 + (((N + 3) / 4) & 1)), total);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

int pcfa_conv_s2_ds_bwd_pack(const float* w, const float* wd, float* packed, int Cin, int N, void* stream) {
  if (w == nullptr || wd == nullptr || packed == nullptr) return PCFA_ERR_INVALID_ARG;
  const long long total = pcfa_conv_s2_ds_bwd_packed_floats(Cin, N);
  pcfa_launch(conv_s2_bwd_pack_kernel<true>, dim3((unsigned)min((total + 255) / 256, 4096LL)), dim3(256), 0,
              (hipStream_t)stream, w, wd, packed, N, Cin, (int)((N + 3) / 4 + (((N + 3) / 4) & 1)), total);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

static int s2_bwd_run(const float* grad_out, const float* grad_out_d, const float* packed, float* grad_x, int B, int Cin,
                      int N, int H, int W, int ksize, void* stream) {
  if (grad_out == nullptr || packed == nullptr || grad_x == nullptr || B < 1) return PCFA_ERR_INVALID_ARG;
  if (!pcfa_conv_s2_bwd_supported(Cin, N, ksize, H, W)) return PCFA_ERR_UNSUPPORTED;
  if (((uintptr_t)grad_out & 15) != 0 || ((uintptr_t)grad_out_d & 15) != 0 || ((uintptr_t)grad_x & 7) != 0)
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int Ho = (H - 1) / 2 + 1, Wo = W / 2, nblk = (Cin + 31) / 32;
  const int rows = (H + 1) / 2;                       // coarse rows i = y >> 1
#define PCFA_S2_BWD1(WN_, WP_, DS_)                                                                                    \
  {                                                                                                                    \
    typedef S2BwdCfg<WN_, WP_, DS_> C;                                                                                 \
    const int tiles_x = pcfa_cdiv(Wo, C::PXC);                                                                         \
    pcfa_launch(conv_s2_bwd_kernel<C>, dim3((unsigned)(tiles_x * rows), (unsigned)pcfa_cdiv(nblk, WN_), (unsigned)B),   \
                dim3(C::NT), 0, s, grad_out, grad_out_d, packed, grad_x, Cin, N, H, W, Ho, Wo, tiles_x);               \
    PCFA_LAUNCH_CHECK();                                                                                               \
    return PCFA_OK;                                                                                                    \
  }
#define PCFA_S2_BWD(WN_, WP_)                                   \
  {                                                             \
    if (grad_out_d != nullptr) PCFA_S2_BWD1(WN_, WP_, true)     \
    PCFA_S2_BWD1(WN_, WP_, false)                               \
  }
  if (nblk >= 4) PCFA_S2_BWD(4, 1)
  if (nblk == 3) PCFA_S2_BWD(3, 1)
  if (nblk == 2) PCFA_S2_BWD(2, 2)
  PCFA_S2_BWD(1, 4)
#undef PCFA_S2_BWD
#undef PCFA_S2_BWD1
}

int pcfa_conv_s2_bwd(const float* grad_out, const float* packed, float* grad_x, int B, int Cin, int N, int H, int W,
                     int ksize, void* stream) {
  if (is_stem(Cin, ksize)) {
    if (grad_out == nullptr || packed == nullptr || grad_x == nullptr || B < 1) return PCFA_ERR_INVALID_ARG;
    if (!pcfa_conv_s2_bwd_supported(Cin, N, ksize, H, W)) return PCFA_ERR_UNSUPPORTED;
    if (((uintptr_t)grad_out & 15) != 0 || ((uintptr_t)grad_x & 7) != 0) return PCFA_ERR_INVALID_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = W / 2, rows = (H + 1) / 2, tiles_x = pcfa_cdiv(Wo, StemBwd::PXC);
    pcfa_launch(conv_s2_stem_bwd_kernel, dim3((unsigned)(tiles_x * pcfa_cdiv(rows, StemBwd::TR)), 1u, (unsigned)B),
                dim3(StemBwd::NT), 0, (hipStream_t)stream, grad_out, packed, grad_x, N, H, W, Ho, Wo, tiles_x);
    PCFA_LAUNCH_CHECK();
    return PCFA_OK;
  }
  return s2_bwd_run(grad_out, nullptr, packed, grad_x, B, Cin, N, H, W, ksize, stream);
}

int pcfa_conv_s2_ds_bwd(const float* grad_out, const float* grad_out_d, const float* packed, float* grad_x, int B, int Cin,
                        int N, int H, int W, void* stream) {
  if (grad_out_d == nullptr) return PCFA_ERR_INVALID_ARG;
  return s2_bwd_run(grad_out, grad_out_d, packed, grad_x, B, Cin, N, H, W, 3, stream);
}

int pcfa_conv_s2_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int Cin, int N, int H,
                     int W, int ksize, int act, float slope, void* stream) {
  if (x == nullptr || packed == nullptr || out == nullptr || B < 1 || act < 0 || act > 2) return PCFA_ERR_INVALID_ARG;
  if (!pcfa_conv_s2_supported(Cin, N, ksize, H, W)) return PCFA_ERR_UNSUPPORTED;
  if (((uintptr_t)x & 15) != 0) return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (is_stem(Cin, ksize)) return fwd_t<StemCfg>(x, packed, bias, out, nullptr, nullptr, B, Cin, N, H, W, act, slope, s);
  // as many channel blocks per workgroup as the layer has (up to four; they share the staged patch), 32 pixels per
  // wave (64 measured 2-5 % slower at the encoder shapes, one-wave workgroups with private patches 10-70 % slower)
  static const int wn_env = getenv("PCFA_S2_WN") ? atoi(getenv("PCFA_S2_WN")) : 0;   // dev override
  const int nblk = (N + 31) / 32, wn = wn_env ? wn_env : (nblk >= 4 ? 4 : nblk);
#define PCFA_S2_RES(WN_) \
  return fwd_t<Res3<WN_, 1>>(x, packed, bias, out, nullptr, nullptr, B, Cin, N, H, W, act, slope, s)
  if (wn == 4) PCFA_S2_RES(4);
  if (wn == 3) PCFA_S2_RES(3);
  if (wn == 2) PCFA_S2_RES(2);
  PCFA_S2_RES(1);
#undef PCFA_S2_RES
}

}  // extern "C"
