// 3x3 / stride 1 / pad 1 convolution as Winograd F(4x4, 3x3) on the fp32 matrix cores of gfx950.
//
// Same operator and call sites as conv3x3.hip (RAFT / GMA encoders and update block: models/raft/extractor.py:23-58,
// update.py:6-16,79-101; PWC-Net conv(): PWCNet.py:29-35), forward and data gradient.  conv3x3.hip runs F(2x2, 3x3):
// 16 products per 4 outputs, 2.25x fewer multiplies than the direct form.  F(4x4, 3x3) needs 36 products per 16
// outputs -- 4x fewer than direct, 1.78x fewer matrix instructions than F(2x2, 3x3):
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 6x6 input tile -> Y: 4x4 output tile, interpolation points
//                                                  0, +-1, +-2, inf (Lavin & Gray 2016, the standard matrices)
// Price: the transform constants reach 8, so fp32 rounding error grows from ~3.5e-7 to ~2e-6 relative per layer
// (measured against fp64; direct fp32: 2.2e-7) -- two orders of magnitude inside the closure-level parity bar.
//
// At RAFT's 55x128 feature maps there are only 448 tiles of 4x4 pixels: 14 workgroup-sized groups of 32 tiles x
// (Cout / 32) channel blocks = 56-112 workgroups for 256 CUs.  The input channels are therefore SPLIT over
// `ksplit` workgroups (blockIdx.z) that write transformed partial outputs into a caller-provided workspace, and a
// streaming kernel adds the partials in index order (deterministic), then bias / activation / mask / addend.  Layers
// with enough tiles (the encoders' 220x512 and 110x256 maps) run with ksplit = 1 and finish in the kernel's epilogue.
//
// Workgroup = 12 waves, 32 tiles (2 tile rows x 16 tile columns = 8 x 64 output pixels) x 32 output channels.
// Six transform rows do not divide over four SIMDs (six waves land 2-2-1-1 and the two loaded SIMDs set the pace), so
// the workgroup's input channels are split once more INSIDE it: waves 0-5 take the even chunks, waves 6-11 the odd
// ones -- three waves on every SIMD, which also hide each other's LDS / L2 waits.  Wave (g, w) owns row w of the 6x6
// transform (xi = 6w .. 6w+5) on group g's chunks: per chunk of 8 input channels the 10 x 72 patch goes global ->
// registers -> LDS as aligned 16-B pieces (own buffers per group); every lane builds the six A operands of its own
// MFMAs for (tile l31, channel pair kp + lane/32) from four patch rows (row stage with wave-uniform coefficients,
// column stage 13 operations), the U operands come from L2 straight into registers (24 B per lane and channel pair);
// 24 MFMAs per wave and chunk, one barrier per chunk.  Group 1's sums meet group 0's in the epilogue's LDS image.
#include "common.hpp"
#include "conv3x3_f43.hpp"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 8;                 // input channels per chunk
constexpr int TRB = 2, TCB = 16;      // tile rows / columns per workgroup
constexpr int TB = TRB * TCB;         // 32 tiles
constexpr int PR = 4 * TRB + 2;       // 10 patch rows
constexpr int NPIECE = 4 * TCB / 4 + 2;   // 18 16-B pieces per patch row: global columns x0-4 .. x0+67 (LDS 1 .. 72)
constexpr int PCP = 80;               // LDS row stride in floats (multiple of 16: the b128 operand reads are conflict-free)
constexpr int RAW = KC * PR * PCP;    // 6400 floats per buffer
constexpr int NW = 6, NG = 2;         // transform rows (waves per group), channel groups (NGT <= NG per instance)
constexpr int NTG = 64 * NW;          // 384 threads per group (768 per workgroup with two groups)
constexpr int RAW_LOADS = (KC * PR * NPIECE + NTG - 1) / NTG;   // 4 pieces per thread and chunk
constexpr int MS = TB + 1;
constexpr int EPI = NW * 4 * 16 * MS;  // epilogue image [6 rows][4 output columns][16 channels][MS]
constexpr int LDSF = NG * 2 * RAW > EPI ? NG * 2 * RAW : EPI;
#ifndef PCFA_F43_DBG
#define PCFA_F43_DBG 0   // timing-only ablation builds (tools/dev/build_variant.sh): 1 no MFMAs, 2 no operand build,
#endif                   // 4 no U loads, 8 no epilogue, 16 no patch staging -- results are garbage

// U = G g G^T, [32-channel block nb][chunk c][wave w][pair kpi 4]{[lane 64][4], [lane 64][2]}: the six operands
// U[xi = 6 w + j][k = 8 c + 2 kpi + (lane >> 5)][n = 32 nb + (lane & 31)], j = 0..5, of a lane and channel pair are one
// 16-B and one 8-B piece, and each piece of all 64 lanes is contiguous: every load instruction of a wave is fully
// coalesced.  (With the 24 floats of a lane adjacent, a wave's load touched 24 cache lines for 1 KB of payload and
// the operand stream alone took 16 us of a 45 us launch.)
__global__ void f43_pack_kernel(const float* __restrict__ w, float* __restrict__ P, int Cout, int Cin, int backward,
                                int K, int N, int nchunk, long long total) {
  const float G[6][3] = {{0.25f, 0.f, 0.f},
                         {-1.f / 6, -1.f / 6, -1.f / 6},
                         {-1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6},
                         {1.f / 24, -1.f / 12, 1.f / 6},
                         {0.f, 0.f, 1.f}};
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int wv = (int)((e / (24 * 64)) % NW), inw = (int)(e % (24 * 64));
    const int kpi_ = inw / 384, inp = inw - 384 * kpi_;            // pair, then 256 floats [lane][4] + 128 floats [lane][2]
    const int lane = inp < 256 ? inp >> 2 : (inp - 256) >> 1;
    const int pos = 6 * kpi_ + (inp < 256 ? (inp & 3) : 4 + ((inp - 256) & 1));
    const long long blk = e / (24 * 64 * NW);   // nb * nchunk + c
    const int c = (int)(blk % nchunk), nb = (int)(blk / nchunk);
    const int kpi = pos / 6, j = pos - 6 * kpi, i = wv;
    const int k = 8 * c + 2 * kpi + (lane >> 5), n = 32 * nb + (lane & 31);
    float acc = 0.f;
    if (n < N && k < K) {
      // (G g G^T)[i][j] = sum_pq G[i][p] g[p][q] G[j][q], accumulated in double and rounded once
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float g = backward ? w[(((long long)k * Cin + n) * 3 + (2 - p)) * 3 + (2 - q)]
                                   : w[(((long long)n * Cin + k) * 3 + p) * 3 + q];
          s += (double)G[i][p] * (double)g * (double)G[j][q];
        }
      acc = (float)s;
    }
    P[e] = acc;
  }
}

// ksplit > 1: `out` is the workspace [ksplit][B][N][H][W] (partials, no bias / activation); ksplit == 1: the result.
// NGT = channel groups inside the workgroup (2: twelve waves, three per SIMD, 168 registers; 1: six waves, 2-2-1-1
// over the SIMDs but 256 registers, so the U operands can run UA = 2 channel pairs ahead).
template <int ACT, bool PARTIAL, int NGT = 2, int UA = 1>
__global__ __launch_bounds__(NTG * NGT) void conv3x3_f43_kernel(
    const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
    const float* __restrict__ mask, const float* __restrict__ addend, float* __restrict__ out, int K, int N, int H,
    int W, int blocks_x, int ksplit, int B, float slope, int mask_n) {
  __shared__ __attribute__((aligned(16))) float smem[LDSF];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int NT = NTG * NGT;
  const int wv12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = (NGT > 1 && wv12 >= NW) ? 1 : 0, wave = wv12 - grp * NW;      // channel group, transform row
  const int gt = tid - grp * NTG;                                   // thread index inside the group
  const int l31 = lane & 31, lh = lane >> 5;
  const int by = blockIdx.x / blocks_x, bx = blockIdx.x - by * blocks_x;
  const int y0 = by * (4 * TRB), x0 = bx * (4 * TCB);
  const int nby = blockIdx.y, n0 = nby * 32;
  const int b_img = blockIdx.z / ksplit, ks = blockIdx.z - b_img * ksplit;
  const long long plane = (long long)H * W;
  x += (long long)b_img * K * plane;

  const int nchunk = (K + KC - 1) / KC;
  const int cb = (int)((long long)nchunk * ks / ksplit), ce = (int)((long long)nchunk * (ks + 1) / ksplit);
  // this group's chunks: cb + grp, cb + grp + 2, ...; both groups run the same number of iterations (the shorter one
  // repeats its last chunk into a dead accumulator-free pass: see `live` below) so that the barriers match
  const int niter = (ce - cb + NGT - 1) / NGT;
  const int nmine = (ce - cb - grp + NGT - 1) / NGT;

  // ---- staging: piece e = gt + NTG i of the chunk's patch: (channel, row, piece column) ----
  unsigned psrc[RAW_LOADS];
  int pdst[RAW_LOADS];
  bool pok[RAW_LOADS];
  unsigned pchs = 0;   // the four pieces' channels inside the chunk, one byte each
#pragma unroll
  for (int i = 0; i < RAW_LOADS; ++i) {
    const int e = gt + NTG * i;
    const int ch = e / (PR * NPIECE), rem = e - ch * (PR * NPIECE);
    pchs |= (unsigned)min(ch, KC - 1) << (8 * i);
    const int r = rem / NPIECE, p = rem - r * NPIECE;
    const int yy = y0 - 1 + r, xx = x0 - 4 + 4 * p;
    const bool in = e < KC * PR * NPIECE;
    pok[i] = in && yy >= 0 && yy < H && xx >= 0 && xx + 3 < W;
    // LDS column of global column x is x - x0 + 5: a tile's six input columns start at 4 tc + 4 (16-B aligned), so a
    // piece lands one float off alignment and is written as four dwords
    pdst[i] = in ? (ch * PR + r) * PCP + 4 * p + 1 : (PR - 1) * PCP + 4 * NPIECE + 4;   // past the patch: pad cells nobody reads
    // channel folded into the offset (ch < 8: fits); chunks whose channels run past K clamp per element below
    psrc[i] = (unsigned)(min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 4)) + (unsigned)(min(ch, KC - 1) * (int)plane);
  }
  const bool kfull = (K % KC) == 0;
  // always a chunk of this workgroup's range (a group that has run out repeats one; `live` keeps it out of the sums)
  auto chunk_of = [&](int it) { return min(cb + grp + NGT * max(min(it, nmine - 1), 0), ce - 1); };
  auto load_raw_half = [&](int c, int h, float4 (&rr)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = 2 * h + q;
      if (kfull) {
        rr[q] = *reinterpret_cast<const float4*>(x + (long long)c * KC * plane + psrc[i]);
      } else {   // the last chunk runs past K: clamp the channel per piece (its LDS cells are zeroed at the store)
        const int chl = (int)((pchs >> (8 * i)) & 255u);
        const int over = max(c * KC + chl - (K - 1), 0);
        rr[q] = *reinterpret_cast<const float4*>(x + (long long)c * KC * plane + psrc[i] - (long long)over * plane);
      }
    }
  };
  auto store_raw_half = [&](int c, int buf, int h, const float4 (&rr)[2]) {
    float* s = smem + (grp * 2 + buf) * RAW;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = 2 * h + q;
      bool ok = pok[i];
      if (!kfull) ok = ok && c * KC + (int)((pchs >> (8 * i)) & 255u) < K;
      const float4 t = ok ? rr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      float* d = s + pdst[i];
      d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    }
  };
  const float* pu = U + ((long long)nby * nchunk * NW + wave) * (24 * 64);

  // ---- row w of B^T: four patch rows and their coefficients (wave-uniform) ----
  //   0: 4 d0 - 5 d2 + d4        1: -4 d1 - 4 d2 + d3 + d4     2: 4 d1 - 4 d2 - d3 + d4
  //   3: -2 d1 - d2 + 2 d3 + d4  4: 2 d1 - d2 - 2 d3 + d4      5: 4 d1 - 5 d3 + d5
  int r0, r1, r2, r3;
  float a0, a1, a2, a3;
  switch (wave) {
    case 0: r0 = 0; r1 = 2; r2 = 4; r3 = 4; a0 = 4.f; a1 = -5.f; a2 = 1.f; a3 = 0.f; break;
    case 1: r0 = 1; r1 = 2; r2 = 3; r3 = 4; a0 = -4.f; a1 = -4.f; a2 = 1.f; a3 = 1.f; break;
    case 2: r0 = 1; r1 = 2; r2 = 3; r3 = 4; a0 = 4.f; a1 = -4.f; a2 = -1.f; a3 = 1.f; break;
    case 3: r0 = 1; r1 = 2; r2 = 3; r3 = 4; a0 = -2.f; a1 = -1.f; a2 = 2.f; a3 = 1.f; break;
    case 4: r0 = 1; r1 = 2; r2 = 3; r3 = 4; a0 = 2.f; a1 = -1.f; a2 = -2.f; a3 = 1.f; break;
    default: r0 = 1; r1 = 3; r2 = 5; r3 = 5; a0 = 4.f; a1 = -5.f; a2 = 1.f; a3 = 0.f; break;
  }
  const int tr = l31 >> 4, tc = l31 & 15;
  // the tile's six input columns are LDS columns 4 tc + 4 .. 4 tc + 9: one aligned 16-B and one 8-B read per row
  // (indices in float4 units so that the compiler sees the alignment: it split unproven 16-B reads into dword pairs,
  // and the LDS -- 2-way conflicts included -- then cost as many cycles as the matrix pipe)
  static_assert(PCP % 4 == 0 && RAW % 4 == 0, "rows and buffers are whole float4s");
  const int q4 = (lh * PR + 4 * tr) * (PCP / 4) + tc + 1;
  const int o0 = q4 + r0 * (PCP / 4), o1 = q4 + r1 * (PCP / 4), o2 = q4 + r2 * (PCP / 4), o3 = q4 + r3 * (PCP / 4);

  f32x16 acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // A operands of one channel pair: row stage over the four patch rows (one 4-B, one aligned 16-B, one 4-B read per
  // row), then the column stage.
  auto operands = [&](const float4* p, float (&v)[6]) {   // p: the channel pair's patch, float4 units
    float t[6];
    {
      const float4 m = p[o0];
      const float2 e = *reinterpret_cast<const float2*>(p + o0 + 1);
      t[0] = a0 * m.x; t[1] = a0 * m.y; t[2] = a0 * m.z; t[3] = a0 * m.w; t[4] = a0 * e.x; t[5] = a0 * e.y;
    }
    {
      const float4 m = p[o1];
      const float2 e = *reinterpret_cast<const float2*>(p + o1 + 1);
      t[0] = fmaf(a1, m.x, t[0]); t[1] = fmaf(a1, m.y, t[1]); t[2] = fmaf(a1, m.z, t[2]);
      t[3] = fmaf(a1, m.w, t[3]); t[4] = fmaf(a1, e.x, t[4]); t[5] = fmaf(a1, e.y, t[5]);
    }
    {
      const float4 m = p[o2];
      const float2 e = *reinterpret_cast<const float2*>(p + o2 + 1);
      t[0] = fmaf(a2, m.x, t[0]); t[1] = fmaf(a2, m.y, t[1]); t[2] = fmaf(a2, m.z, t[2]);
      t[3] = fmaf(a2, m.w, t[3]); t[4] = fmaf(a2, e.x, t[4]); t[5] = fmaf(a2, e.y, t[5]);
    }
    {
      const float4 m = p[o3];
      const float2 e = *reinterpret_cast<const float2*>(p + o3 + 1);
      t[0] = fmaf(a3, m.x, t[0]); t[1] = fmaf(a3, m.y, t[1]); t[2] = fmaf(a3, m.z, t[2]);
      t[3] = fmaf(a3, m.w, t[3]); t[4] = fmaf(a3, e.x, t[4]); t[5] = fmaf(a3, e.y, t[5]);
    }
    v[0] = fmaf(4.f, t[0], fmaf(-5.f, t[2], t[4]));
    const float p1 = fmaf(-4.f, t[2], t[4]), q1 = fmaf(-4.f, t[1], t[3]);
    v[1] = p1 + q1;
    v[2] = p1 - q1;
    const float p2 = t[4] - t[2], q2 = 2.f * (t[3] - t[1]);
    v[3] = p2 + q2;
    v[4] = p2 - q2;
    v[5] = fmaf(4.f, t[1], fmaf(-5.f, t[3], t[5]));
  };
  struct UPair { float4 a; float2 b; };
  auto load_u = [&](int c, int kpi) {
    const float* q = pu + (long long)c * (NW * 24 * 64) + kpi * 384;
    return UPair{reinterpret_cast<const float4*>(q)[lane], reinterpret_cast<const float2*>(q + 256)[lane]};
  };

  // ---- main loop: the patch of this group's next chunk is loaded during the current one (registers), written to the
  //      group's other LDS buffer after the MFMAs, one barrier per chunk (all twelve waves).  The U operands of a
  //      channel pair are requested one pair ahead (the first pair of a chunk: before the previous chunk's barrier);
  //      the scheduler is fenced behind the requests -- left alone it sank every load next to its use and each pair
  //      waited for L2. ----
  // Register budget: three waves per SIMD leave 168 registers per lane, 96 of them accumulators -- a spilled value in
  // this loop is a scratch (global) access whose wait also waits for every load in flight.  The U operands of a pair
  // are requested ONE pair ahead (two pairs ahead needs six more registers and spilled: 36 -> 46 us), the next chunk's
  // patch travels in two halves of two pieces.
  if (niter > 0) {
    float4 rh[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      load_raw_half(chunk_of(0), h, rh);
      store_raw_half(chunk_of(0), 0, h, rh);
    }
  }
  auto upair = [&](int it, int kpi) {   // pair kpi (may run past 3: the following chunks) of iteration it
    const int itn = it + (kpi >> 2);
    return load_u(chunk_of(itn), kpi & 3);
  };
  UPair uq[UA + 1];
#pragma unroll
  for (int a = 0; a < UA; ++a) uq[a] = upair(0, a);
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    const int cn = chunk_of(it + 1);
    const bool live = it < nmine;          // wave-uniform: the group with one chunk fewer idles through the last pass
    const bool more = it + 1 < niter && !(PCFA_F43_DBG & 16);
    const float4* s = reinterpret_cast<const float4*>(smem) + (grp * 2 + (it & 1)) * (RAW / 4);
    float4 rh[2];
#pragma unroll
    for (int kpi = 0; kpi < KC / 2; ++kpi) {
      uq[UA] = (PCFA_F43_DBG & 4) ? uq[0] : upair(it, kpi + UA);
      if (more && (kpi & 1) == 0) load_raw_half(cn, kpi >> 1, rh);
      __builtin_amdgcn_sched_barrier(0);
      const UPair u0 = uq[0];
      if (live) {
        float v[6];
        if (PCFA_F43_DBG & 2) {
#pragma unroll
          for (int j = 0; j < 6; ++j) v[j] = u0.a.x + (float)(j + kpi);
        } else {
          operands(s + 2 * kpi * (PR * PCP / 4), v);
        }
        if (PCFA_F43_DBG & 1) {
#pragma unroll
          for (int j = 0; j < 6; ++j) acc[j][0] += v[j] * u0.b.y;
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], u0.a.x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1], u0.a.y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], u0.a.z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3], u0.a.w, acc[3], 0, 0, 0);
          acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[4], u0.b.x, acc[4], 0, 0, 0);
          acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[5], u0.b.y, acc[5], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more && (kpi & 1) == 1) store_raw_half(cn, (it + 1) & 1, kpi >> 1, rh);
#pragma unroll
      for (int a = 0; a < UA; ++a) uq[a] = uq[a + 1];
    }
    __syncthreads();
  }
  if (PCFA_F43_DBG & 8) {
    if (tid == 0) out[blockIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][4] + acc[5][5];
    return;
  }

  // ---- epilogue.  Column stage of A^T . A in registers (six accumulators of one (tile, channel) live in one lane):
  //      s[c'] = sum_j M[w][j] A[j][c'], then [6 rows][4 columns] through LDS, thread (tile, channel) applies the row
  //      stage and finishes the 4 x 4 pixels.  Group 1 adds its sums onto group 0's image (fixed order) first: one
  //      partial output per workgroup (a slot per group was measured: twice the workspace traffic, slower). ----
  float* sY = smem;   // [(w * 4 + c') * 16 + channel][MS]
  const float* pin = PARTIAL ? nullptr : bias;
  float* ob = out + ((long long)(PARTIAL ? ks * B + b_img : b_img) * N) * plane;
  if (!PARTIAL) {
    if (mask != nullptr) mask += (long long)b_img * N * plane;
    if (addend != nullptr) addend += (long long)b_img * N * plane;
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int g = 0; g < NGT; ++g) {
      if (grp == g && (l31 >> 4) == pass) {
        const int cl = l31 & 15;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
          const float d12 = m1 - m2, s12 = m1 + m2, d34 = m3 - m4, s34 = m3 + m4;
          const int tile = (r & 3) + 8 * (r >> 2) + 4 * lh;
          float* d = &sY[((wave * 4) * 16 + cl) * MS + tile];
          const float y0_ = m0 + s12 + s34, y1_ = fmaf(2.f, d34, d12), y2_ = fmaf(4.f, s34, s12),
                      y3_ = fmaf(8.f, d34, d12) + m5;
          if (g == 0) {
            d[0] = y0_; d[16 * MS] = y1_; d[32 * MS] = y2_; d[48 * MS] = y3_;
          } else {
            d[0] += y0_; d[16 * MS] += y1_; d[32 * MS] += y2_; d[48 * MS] += y3_;
          }
        }
      }
      __syncthreads();
    }
    for (int e = tid; e < 16 * TB; e += NT) {
      const int tile = e & 31, cl = e >> 5;
      const int n = n0 + pass * 16 + cl;
      const int oy = y0 + 4 * (tile >> 4), ox = x0 + 4 * (tile & 15);
      float yv[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float q[6];
#pragma unroll
        for (int w6 = 0; w6 < 6; ++w6) q[w6] = sY[((w6 * 4 + c) * 16 + cl) * MS + tile];
        const float d12 = q[1] - q[2], s12 = q[1] + q[2], d34 = q[3] - q[4], s34 = q[3] + q[4];
        yv[0][c] = q[0] + s12 + s34;
        yv[1][c] = fmaf(2.f, d34, d12);
        yv[2][c] = fmaf(4.f, s34, s12);
        yv[3][c] = fmaf(8.f, d34, d12) + q[5];
      }
      if (n < N && ox < W) {
        const float bv = (!PARTIAL && pin != nullptr) ? pin[n] : 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (oy + p < H) {
            const long long oo = (long long)n * plane + (long long)(oy + p) * W + ox;
            float4 y = make_float4(yv[p][0] + bv, yv[p][1] + bv, yv[p][2] + bv, yv[p][3] + bv);
            if (!PARTIAL) {
              if (ACT == 1) {
                y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
              } else if (ACT == 2) {
                y.x = y.x > 0.f ? y.x : y.x * slope; y.y = y.y > 0.f ? y.y : y.y * slope;
                y.z = y.z > 0.f ? y.z : y.z * slope; y.w = y.w > 0.f ? y.w : y.w * slope;
              }
              const float ms = ACT == 0 ? slope : 0.f;   // mask factor where the producer's output is not positive
              auto masked = [&](float v, float m) { return m > 0.f ? v : (ms == 0.f ? 0.f : v * ms); };
              if (mask != nullptr && mask_n == 0) {
                const float4 mk = *reinterpret_cast<const float4*>(mask + oo);
                y.x = masked(y.x, mk.x); y.y = masked(y.y, mk.y); y.z = masked(y.z, mk.z); y.w = masked(y.w, mk.w);
              }
              if (addend != nullptr) {
                const float4 ad = *reinterpret_cast<const float4*>(addend + oo);
                y.x += ad.x; y.y += ad.y; y.z += ad.z; y.w += ad.w;
              }
              if (mask != nullptr && mask_n > 0 && n < mask_n) {   // channel prefix, after the addend (conv3x3.hip)
                const float4 mk = *reinterpret_cast<const float4*>(mask + oo);
                y.x = masked(y.x, mk.x); y.y = masked(y.y, mk.y); y.z = masked(y.z, mk.z); y.w = masked(y.w, mk.w);
              }
            }
            *reinterpret_cast<float4*>(ob + oo) = y;
          }
        }
      }
    }
    __syncthreads();
  }
}

// out = act(sum_ks partial[ks] + bias) [* (mask > 0)] [+ addend], partials added in index order.
// FAMILY only names the instance (43: partials of this file's kernel, 23: of conv3x3.hip's K-sliced F(2x2,3x3) kernel), so
// that a kernel trace books the finish launch under the family whose work it completes (bench.py TRACED).
template <int ACT, int FAMILY>
__global__ __launch_bounds__(256) void f43_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                         const float* __restrict__ mask,
                                                         const float* __restrict__ addend, float* __restrict__ out,
                                                         int ksplit, long long total4, long long plane4, int N,
                                                         float slope, int mask_n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    float4 y = reinterpret_cast<const float4*>(part)[i];
    for (int k = 1; k < ksplit; ++k) {
      const float4 p = reinterpret_cast<const float4*>(part)[i + (long long)k * total4];
      y.x += p.x; y.y += p.y; y.z += p.z; y.w += p.w;
    }
    if (bias != nullptr) {
      const float bv = bias[(int)((i / plane4) % N)];
      y.x += bv; y.y += bv; y.z += bv; y.w += bv;
    }
    if (ACT == 1) {
      y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
    } else if (ACT == 2) {
      y.x = y.x > 0.f ? y.x : y.x * slope; y.y = y.y > 0.f ? y.y : y.y * slope;
      y.z = y.z > 0.f ? y.z : y.z * slope; y.w = y.w > 0.f ? y.w : y.w * slope;
    }
    const float ms = ACT == 0 ? slope : 0.f;
    auto masked = [&](float v, float m) { return m > 0.f ? v : (ms == 0.f ? 0.f : v * ms); };
    if (mask != nullptr && mask_n == 0) {
      const float4 mk = reinterpret_cast<const float4*>(mask)[i];
      y.x = masked(y.x, mk.x); y.y = masked(y.y, mk.y); y.z = masked(y.z, mk.z); y.w = masked(y.w, mk.w);
    }
    if (addend != nullptr) {
      const float4 ad = reinterpret_cast<const float4*>(addend)[i];
      y.x += ad.x; y.y += ad.y; y.z += ad.z; y.w += ad.w;
    }
    if (mask != nullptr && mask_n > 0 && (int)((i / plane4) % N) < mask_n) {
      const float4 mk = reinterpret_cast<const float4*>(mask)[i];
      y.x = masked(y.x, mk.x); y.y = masked(y.y, mk.y); y.z = masked(y.z, mk.z); y.w = masked(y.w, mk.w);
    }
    reinterpret_cast<float4*>(out)[i] = y;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

long long pcfa_f43_packed_floats(int K, int N) {
  return 36LL * (((long long)K + KC - 1) / KC * KC) * (((long long)N + 31) / 32 * 32);
}

int pcfa_f43_pack(const float* w, float* packed, int Cout, int Cin, int backward, hipStream_t s) {
  const int K = backward ? Cout : Cin, N = backward ? Cin : Cout;
  const int nchunk = (K + KC - 1) / KC;
  const long long total = pcfa_f43_packed_floats(K, N);
  pcfa_launch(f43_pack_kernel, dim3((unsigned)min((total + 255) / 256, 8192LL)), dim3(256), 0, s, w, packed, Cout, Cin,
              backward, K, N, nchunk, total);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

// How many workgroups share the input channels of one output block: enough to give every CU one workgroup, never
// fewer than two chunks per workgroup.
int pcfa_f43_ksplit(int B, int K, int N, int H, int W) {
  const long long nwg = (long long)pcfa_cdiv(W, 4 * TCB) * pcfa_cdiv(H, 4 * TRB) * pcfa_cdiv(N, 32) * B;
  const int nchunk = (K + KC - 1) / KC;
  if (nwg >= 192) return 1;
  int ks = (int)(256 / nwg);
  ks = min(ks, nchunk / 2);
  return max(ks, 1);
}

bool pcfa_f43_supported(int B, int K, int N, int H, int W) {
  return W % 4 == 0 && W >= 8 && H >= 1 && (long long)K * H * W < 0x7fffffffLL && B < 16384 &&
         pcfa_f43_packed_floats(K, N) < 0x7fffffffLL;
}

size_t pcfa_f43_workspace_bytes(int B, int K, int N, int H, int W) {
  if (!pcfa_f43_supported(B, K, N, H, W)) return 0;
  const int ks = pcfa_f43_ksplit(B, K, N, H, W);
  return ks > 1 ? (size_t)ks * B * N * H * W * sizeof(float) : 0;
}

int pcfa_f43_run(const float* x, const float* packed, const float* bias, const float* mask, const float* addend,
                 float* out, int B, int K, int N, int H, int W, int act, float slope, int mask_n, void* workspace,
                 size_t workspace_bytes, hipStream_t s) {
  if (!pcfa_f43_supported(B, K, N, H, W)) return PCFA_ERR_UNSUPPORTED;
  if (!aligned16(x) || !aligned16(out) || !aligned16(packed) || (mask && !aligned16(mask)) ||
      (addend && !aligned16(addend)))
    return PCFA_ERR_UNSUPPORTED;
  const int ksplit = pcfa_f43_ksplit(B, K, N, H, W);
  const int blocks_x = pcfa_cdiv(W, 4 * TCB), blocks_y = pcfa_cdiv(H, 4 * TRB);
  static const int six = getenv("PCFA_F43_WAVES") ? atoi(getenv("PCFA_F43_WAVES")) == 6 : 0;   // dev A/B (tools/dev)
  dim3 grid((unsigned)(blocks_x * blocks_y), (unsigned)pcfa_cdiv(N, 32), (unsigned)(B * ksplit)), block(six ? NTG : 2 * NTG);
  if (ksplit == 1) {
#define PCFA_F43_DIRECT(A_)                                                                                          \
  do {                                                                                                               \
    if (six) pcfa_launch(conv3x3_f43_kernel<A_, false, 1, 2>, grid, block, 0, s, x, packed, bias, mask, addend, out, \
                         K, N, H, W, blocks_x, 1, B, slope, mask_n);                                                 \
    else pcfa_launch(conv3x3_f43_kernel<A_, false>, grid, block, 0, s, x, packed, bias, mask, addend, out, K, N, H,  \
                     W, blocks_x, 1, B, slope, mask_n);                                                              \
  } while (0)
    if (act == 1) PCFA_F43_DIRECT(1); else if (act == 2) PCFA_F43_DIRECT(2); else PCFA_F43_DIRECT(0);
#undef PCFA_F43_DIRECT
    PCFA_LAUNCH_CHECK();
    return PCFA_OK;
  }
  const size_t need = (size_t)ksplit * B * N * H * W * sizeof(float);
  if (!workspace || workspace_bytes < need || !aligned16(workspace)) return PCFA_ERR_INVALID_ARG;
  float* part = (float*)workspace;
  if (six)
    pcfa_launch(conv3x3_f43_kernel<0, true, 1, 2>, grid, block, 0, s, x, packed, (const float*)nullptr,
                (const float*)nullptr, (const float*)nullptr, part, K, N, H, W, blocks_x, ksplit, B, 0.f, 0);
  else
    pcfa_launch(conv3x3_f43_kernel<0, true>, grid, block, 0, s, x, packed, (const float*)nullptr, (const float*)nullptr,
                (const float*)nullptr, part, K, N, H, W, blocks_x, ksplit, B, 0.f, 0);
  PCFA_LAUNCH_CHECK();
  return pcfa_f43_finish(part, bias, mask, addend, out, ksplit, B, N, H, W, act, slope, mask_n, s);
}

// The finish pass alone: out = epilogue(sum of `ksplit` partial outputs [ksplit][B][N][H][W], in index order).  Also
// serves conv3x3.hip's F(2x2,3x3) kernel when it slices K over workgroups.  H * W % 4 == 0.
int pcfa_f43_finish(const float* part, const float* bias, const float* mask, const float* addend, float* out, int ksplit,
                    int B, int N, int H, int W, int act, float slope, int mask_n, hipStream_t s, int family) {
  if (((long long)H * W) % 4 != 0 || !aligned16(part) || !aligned16(out) || (mask && !aligned16(mask)) ||
      (addend && !aligned16(addend)))
    return PCFA_ERR_UNSUPPORTED;
  const long long total4 = (long long)B * N * H * W / 4, plane4 = (long long)H * W / 4;
  const dim3 fg((unsigned)min((total4 + 255) / 256, 2048LL)), fb(256);
#define PCFA_F43_FINISH(A_, F_) \
  pcfa_launch(f43_finish_kernel<A_, F_>, fg, fb, 0, s, part, bias, mask, addend, out, ksplit, total4, plane4, N, slope, mask_n)
  if (family == 23) {
    if (act == 1) PCFA_F43_FINISH(1, 23); else if (act == 2) PCFA_F43_FINISH(2, 23); else PCFA_F43_FINISH(0, 23);
  } else {
    if (act == 1) PCFA_F43_FINISH(1, 43); else if (act == 2) PCFA_F43_FINISH(2, 43); else PCFA_F43_FINISH(0, 43);
  }
#undef PCFA_F43_FINISH
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
