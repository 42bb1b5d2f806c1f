// 3x3 / stride 1 / pad 1 convolution, forward and data gradient, as Winograd F(2x2, 3x3) on the fp32 matrix cores of
// gfx950.
//
// Replaces, for the frozen-weight attack, the 3x3 convolutions of the RAFT/GMA update block (reference
// models/raft/update.py:6-16 FlowHead, :79-101 BasicMotionEncoder; models/gma/update.py likewise), which the library
// path runs as a VALU Winograd kernel at 85-98 effective TFLOP/s.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 4x4 input tile d -> 2x2 output tile Y
// The 16 element-wise products over (tiles x Cin x Cout) are 16 independent GEMMs
//   M_xi [tiles x Cout] = V_xi [tiles x Cin] . U_xi [Cin x Cout],   xi = 0..15,
// and run on v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation): 2.25x fewer multiplies than the direct
// form at the matrix-pipe rate.  U = G g G^T is computed once per weight tensor (pcfa_conv3x3_pack_weights); the data
// gradient is the same operator on grad_out with the flipped / transposed weights (second packing).
//
// Workgroup = 4 waves: 32 tiles (4 tile rows x 8 tile columns = 8 x 16 output pixels) x 32 output channels.
// Per chunk of 8 input channels the 10 x 18 input patch goes global -> registers -> LDS; wave w owns xi = 4w..4w+3
// (row w of the transform) and every lane builds the A operands of its own MFMAs from two patch rows, the U
// operands come from L2 straight into registers; 16 MFMAs per wave and chunk, one barrier per chunk.
// Epilogue: accumulators -> LDS in passes of 16 channels, thread (channel, tile) applies A^T . A, adds the bias,
// optionally ReLU, stores 2x2 pixels.
#include "common.hpp"
#include "conv3x3_f43.hpp"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TC = 8;                            // tile columns per workgroup (16 output columns)
constexpr int PC = 2 * TC + 2;                   // input patch columns (18)
constexpr int PCP = 20;                          // padded patch row stride
constexpr int KC = 8;                            // input channels per chunk
constexpr int CB = 64;                           // packing granularity of the output channels (U row padding)

// Packed weights: U = G g G^T with g = w[n][k] (forward) or the flipped w[k][n] (data gradient), laid out in the
// order the kernel consumes it -- [32-channel block nb][8-channel chunk c][wave w][lane][16]: the 16 floats of a
// lane are its B operands of one chunk, U[xi = 4w + a][k = 8c + 2kpi + (lane >> 5)][n = 32nb + (lane & 31)] at
// position 4a + kpi, so a wave fetches them with four coalesced 16-B loads per lane instead of sixteen dword loads
// (load issue was 21-31 % of the loop, tools/dev/conv3x3_stamps.py).  Rows k >= K and columns n >= N are zero.
__global__ void conv3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ P, int Cout, int Cin,
                                    int backward, int K, int N, int nchunk, long long total) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int pos = (int)(e & 15), lane = (int)((e >> 4) & 63), wv = (int)((e >> 10) & 3);
    const long long blk = e >> 12;  // nb * nchunk + c
    const int c = (int)(blk % nchunk), nb = (int)(blk / nchunk);
    const int a = pos >> 2, kpi = pos & 3;
    const int i = wv, j = a;  // xi = 4 i + j
    const int k = 8 * c + 2 * kpi + (lane >> 5), n = 32 * nb + (lane & 31);
    float g[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float v = 0.f;
        if (n < N && k < K) v = backward ? w[(((long long)k * Cin + n) * 3 + (2 - p)) * 3 + (2 - q)]
                                         : w[(((long long)n * Cin + k) * 3 + p) * 3 + q];
        g[p][q] = v;
      }
    // row i of G g (1 x 3), then column j of (.) G^T
    float r[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
      r[q] = i == 0 ? g[0][q] : i == 1 ? 0.5f * (g[0][q] + g[1][q] + g[2][q])
                              : i == 2 ? 0.5f * (g[0][q] - g[1][q] + g[2][q]) : g[2][q];
    P[e] = j == 0 ? r[0] : j == 1 ? 0.5f * (r[0] + r[1] + r[2]) : j == 2 ? 0.5f * (r[0] - r[1] + r[2]) : r[2];
  }
}

// Diagnostic stamps (tools/dev, -DPCFA_C3_STAMPS): per workgroup and wave, sums of the s_memtime deltas between the
// five points of a loop iteration, written to the floats just past the output tensor's end is NOT acceptable, so
// they go to a __device__ buffer of their own that a dev tool copies back.  Compiled out of the product build.
#ifdef PCFA_C3_STAMPS
__device__ unsigned long long c3_stamp_sums[8];
__device__ __forceinline__ unsigned long long c3_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define C3_STAMP(k)                                                                              \
  do {                                                                                           \
    const unsigned long long now_ = c3_now();                                                    \
    if (k != 0) stamp_acc[k] += now_ - stamp_prev;                                               \
    stamp_prev = now_;                                                                           \
  } while (0)
#else
#define C3_STAMP(k) do {} while (0)
#endif

// x [B][K][H][W], U = packed weights (above), out [B][N][H][W]; grid = (tile blocks, Npad / 32, B).
// CBT = output channels per workgroup: 64 (128 accumulator registers per lane, one block per CU) or 32 (64
// accumulators, 76.8 KB LDS: two blocks per CU overlap each other's staging / transform / barrier phases).
// MT = 32-tile groups per workgroup (4 waves each): MT = 2 -> 64 tiles (16 x 16 output pixels), 8 waves sharing one
// U slice, i.e. half the L2 -> LDS weight traffic per multiply.
// KFULL: K % 8 == 0 -- the staging loop then carries no channel bookkeeping at all (per-thread base pointers plus
// one scalar chunk offset; the generic variant clamps and masks the channel index of every element).
// NRING: depth of the register ring (2: <= 168 registers, 3 waves per SIMD -- large grids; 3: loads two chunks
// ahead, 2 waves per SIMD -- small grids, where a CU holds one or two workgroups anyway and latency is all).
// A second, independent convolution over the same image size served by the same launch (blockIdx.y >= nb0): the
// motion encoder's convc2 (256 -> 192, 336 workgroups) and convf2 (128 -> 64, 112 workgroups) are independent, and
// launched separately each pays its own partial last round of workgroups on the 256 CUs; together the small
// problem's workgroups fill the large one's tail (dispatch order: x fastest, then y).  nb0 = gridDim.y: none.
struct Second {
  const float* x;
  const float* U;
  const float* bias;
  float* out;
  int K, N, nb0;
  int mask_n;   // epilogue mask mode of the FIRST problem: 0 = every channel, before the addend; > 0 = channels < mask_n only,
                // AFTER the addend (the dense-block backward: the producer's LeakyReLU backward once its gradient is complete)
  int ksl;      // K slices over workgroups (blockIdx.z = image * ksl + slice): slice s convolves chunks [s cper, (s+1) cper)
  int cper;     // and writes its raw sums to out[(s * images + image)] -- a partial output for f43_finish_kernel (small
                // maps: a handful of workgroups walking the whole K is a latency chain of ~1 us per chunk)
  PcfaXcdMap xcd;   // a != 0: 1-D grid, XCD k owns a sub-rectangle of (tile block, channel block) (common.hpp)
};

// KS = 2: in-workgroup split-K for launches of at most one workgroup per CU (e.g. conv 256 -> 126 at 55x128: 224
// workgroups, one wave per SIMD, 37 us for 17 us of MFMA work).  Two groups of four waves run the same pipeline on
// the even / odd input-channel chunks (own LDS patch buffers, shared barriers); the second group's accumulators are
// added through LDS in fixed order (deterministic) and the first group runs the epilogue.  K % 16 == 0.
template <int ACT, int CBT, int MT, bool KFULL, int NRING, int KS = 1>  // ACT: 0 none, 1 ReLU, 2 LeakyReLU(slope)
__global__ __launch_bounds__(256 * MT * KS) __attribute__((amdgpu_waves_per_eu(KS == 2 ? 2 : (MT == 1 && NRING == 2 && CBT == 32 ? 3 : (CBT == 64 ? 1 : 2))))) void conv3x3_winograd_kernel(
    const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
    const float* __restrict__ mask, const float* __restrict__ addend, float* __restrict__ out, int K, int N, int Npad,
    int H, int W, int blocks_x, float slope, Second second) {
  constexpr int NT = 256 * MT;                         // threads
  constexpr int TR = 4 * MT, TB = TR * TC;             // tile rows / tiles per workgroup
  constexpr int PR = 2 * TR + 2;                       // input patch rows
  constexpr int RAW = KC * PR * PCP;
  constexpr int RAW_LOADS = (KC * PR * PC + NT - 1) / NT;
  constexpr int MS = TB + 1;                           // epilogue image: [16][16 channels][MS]
  // Only the raw input patch lives in LDS (double-buffered).  Wave w owns xi = 4w..4w+3, i.e. ROW w of the 4x4
  // transform: every lane builds the A operands of its own MFMAs -- V[w][0..3] of (tile l31, channel kp + lh) --
  // from two patch rows (8 LDS reads, 8 adds), so there is no V image, no transform phase and ONE barrier per chunk.
  // The U operands go from L2 straight into registers (every U element is consumed by exactly one wave).
  constexpr int NB = CBT / 32;
  static_assert(KS == 1 || (KS == 2 && MT == 1 && CBT == 32), "split-K rides on the 256-thread tile");
  constexpr int EPI = 16 * 16 * (TB + 1);               // epilogue image
  constexpr int RED = KS == 2 ? 4 * 4 * 16 * 64 : 0;    // second group's accumulators (64 KB)
  constexpr int LDS0 = 2 * KS * RAW > EPI ? 2 * KS * RAW : EPI;
  constexpr int LDSF = RED > LDS0 ? RED : LDS0;
  __shared__ __attribute__((aligned(16))) float smem[LDSF];
  const int kgrp = KS == 1 ? 0 : (int)(threadIdx.x >> 8);   // which half of the chunks this wave group owns
  float* sRaw = smem + kgrp * 2 * RAW;   // [2][RAW] per group

  const int tid = KS == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 255);
  const int lane = tid & 63, wave = (tid >> 6) & 3, mt = tid >> 8;  // wave: xi group, mt: tile group
  const int l31 = lane & 31, lh = lane >> 5;
  int tileblk = blockIdx.x, nby = blockIdx.y;
  if (second.xcd.a != 0 && !pcfa_xcd_item(second.xcd, (int)blockIdx.x, tileblk, nby)) return;   // (dead: whole workgroup)
  const int by = tileblk / blocks_x, bx = tileblk - by * blocks_x;
  const int y0 = by * (2 * TR), x0 = bx * (2 * TC);  // first output pixel of the block
  if (nby >= second.nb0) {   // workgroup-uniform: this workgroup belongs to the second problem
    nby -= second.nb0;
    x = second.x;
    U = second.U;
    bias = second.bias;
    out = second.out;
    K = second.K;
    N = second.N;
    mask = nullptr;
    addend = nullptr;
  }
  const int mask_n = second.mask_n;
  const float mslope = ACT == 0 ? slope : 0.f;   // data gradients run without an activation: `slope` is the mask's
  const int n0 = nby * CBT;
  if (n0 >= N) return;  // (the packing pads N to 64: a 32-channel block may lie entirely in the padding)
  const long long plane = (long long)H * W;
  const int ksl = second.ksl, img = blockIdx.z / ksl, slice = blockIdx.z - img * ksl;   // ksl = 1: img = blockIdx.z
  x += (long long)img * K * plane;
  out += ((long long)slice * (gridDim.z / ksl) + img) * N * plane;
  if (mask != nullptr) mask += (long long)img * N * plane;   // same shape as `out`
  if (addend != nullptr) addend += (long long)img * N * plane;

  // ---- per-thread constants of the staging loads (chunk-invariant) ----
  unsigned praw[RAW_LOADS];      // chunk 0 source of the patch element (clamped into the image), floats from x
  int rdst[RAW_LOADS];           // sRaw index, -1 = no element
  int rch[RAW_LOADS];
  bool rok[RAW_LOADS];
#pragma unroll
  for (int i = 0; i < RAW_LOADS; ++i) {
    const int e = tid + NT * i;
    const int ch = e / (PR * PC), rem = e - ch * (PR * PC);
    const int r = rem / PC, c = rem - r * PC;
    const int yy = y0 - 1 + r, xx = x0 - 1 + c;
    rch[i] = min(ch, KC - 1);
    rdst[i] = e < KC * PR * PC ? (ch * PR + r) * PCP + c : PC;   // past the patch: a padding cell of row 0 (never
                                                                   // read) -- an `if` here was a branch per store
    rok[i] = e < KC * PR * PC && yy >= 0 && yy < H && xx >= 0 && xx < W;
    praw[i] = (unsigned)(min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1) + (KFULL ? rch[i] * (int)plane : 0));
  }
  // this lane's 16 B operands of chunk c: packed[((nb * nchunk + c) * 4 + wave) * 64 + lane][16]
  static_assert(CBT == 32, "the packed weight layout is per 32-channel block");
  const int nchunk = (K + KC - 1) / KC;
  const float* pu = U + (((long long)nby * nchunk * 4 + wave) * 64 + lane) * 16;

  // register ring: the global loads of a chunk are issued NRING-1 chunks before its MFMAs
  float ring_raw[NRING][RAW_LOADS];
  float4 ring_u[NRING][4];
  auto load_u = [&](int c0, float4 (&ru)[4]) {
    const float4* q = reinterpret_cast<const float4*>(pu + (long long)(c0 / KC) * (4 * 64 * 16));
#pragma unroll
    for (int i = 0; i < 4; ++i) ru[i] = q[i];
  };
  auto load_raw = [&](int c0, float (&rraw)[RAW_LOADS]) {
    if (KFULL) {
      const unsigned xo = (unsigned)(c0 * (int)plane);  // wave-uniform chunk offset
#pragma unroll
      for (int i = 0; i < RAW_LOADS; ++i)
#ifdef PCFA_C3_TIMING_NO_RAW  // timing-only build (tools/dev)
        rraw[i] = (float)(praw[i] + xo);
#else
        rraw[i] = x[praw[i] + xo];
#endif
    } else {
#pragma unroll
      for (int i = 0; i < RAW_LOADS; ++i)
        rraw[i] = x[praw[i] + (unsigned)(min(c0 + rch[i], K - 1) * (int)plane)];  // channels >= K: zeroed at the LDS write
    }
  };
  auto store_chunk = [&](int c0, int buf, const float (&rraw)[RAW_LOADS]) {
    float* sRawb = sRaw + buf * RAW;
#pragma unroll
    for (int i = 0; i < RAW_LOADS; ++i)
      sRawb[rdst[i]] = (rok[i] && (KFULL || c0 + rch[i] < K)) ? rraw[i] : 0.f;
  };
  f32x16 acc[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // MFMAs of the chunk whose patch sits in sRaw[buf]; ru = the chunk's U operands.
  // Row w of B^T d uses patch rows (ra, rb) with signs (sa, sb): 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3.
  const int ra = wave == 0 ? 0 : 1, rb = wave == 3 ? 3 : 2;
  const float sa = wave == 2 ? -1.f : 1.f, sb = (wave == 1 || wave == 2) ? 1.f : -1.f;
  const int m_tile = mt * 32 + l31;
  const float* prow = sRaw + (lh * PR + 2 * (m_tile >> 3)) * PCP + 2 * (m_tile & 7);
  auto load_chunk = [&](int c0, float (&rraw)[RAW_LOADS], float4 (&ru)[4]) {
    load_u(c0, ru);
    load_raw(c0, rraw);
  };
  // mid1 / mid2 run after the first / second channel pair's MFMAs, fenced so they stay there: the LDS write of the
  // next chunk's patch and the global loads of the chunk after it ride in the shadow of this chunk's MFMAs instead of
  // following them (at one or two waves per SIMD -- every launch at 55x128 -- nothing else hid that tail).
  auto mfma_chunk = [&](int buf, const float4 (&ru)[4], auto&& mid1, auto&& mid2) {
    // the patch rows of channel pair kp+2 are requested BEFORE the MFMAs of pair kp are issued: an in-order wave
    // otherwise starts the LDS reads only after its fourth MFMA has left the issue stage and the matrix pipe
    // idles for the LDS latency in every pair (stamps: 2000 cycles for 1024 cycles of MFMA work)
    const float* p0 = prow + buf * RAW;
    float2 a0 = *reinterpret_cast<const float2*>(p0 + ra * PCP), a1 = *reinterpret_cast<const float2*>(p0 + ra * PCP + 2);
    float2 b0 = *reinterpret_cast<const float2*>(p0 + rb * PCP), b1 = *reinterpret_cast<const float2*>(p0 + rb * PCP + 2);
#pragma unroll
    for (int kp = 0; kp < KC; kp += 2) {
      float2 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (kp + 2 < KC) {
        const float* p = p0 + (kp + 2) * (PR * PCP);
        na0 = *reinterpret_cast<const float2*>(p + ra * PCP);
        na1 = *reinterpret_cast<const float2*>(p + ra * PCP + 2);
        nb0 = *reinterpret_cast<const float2*>(p + rb * PCP);
        nb1 = *reinterpret_cast<const float2*>(p + rb * PCP + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      const float t0 = fmaf(sb, b0.x, sa * a0.x), t1 = fmaf(sb, b0.y, sa * a0.y);
      const float t2 = fmaf(sb, b1.x, sa * a1.x), t3 = fmaf(sb, b1.y, sa * a1.y);
      const float av[4] = {t0 - t2, t1 + t2, t2 - t1, t1 - t3};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float bv = kp == 0 ? ru[a].x : kp == 2 ? ru[a].y : kp == 4 ? ru[a].z : ru[a].w;
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv, acc[a][0], 0, 0, 0);
      }
      if (kp == 0) {
        mid1();
        __builtin_amdgcn_sched_barrier(0);
      } else if (kp == 2) {
        mid2();
        __builtin_amdgcn_sched_barrier(0);
      }
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
  };

  // Ring slot c % NRING holds chunk c: its patch is written to LDS one iteration before its MFMAs, its U operands
  // stay in registers until the MFMAs have consumed them; the slot is reloaded (chunk c + NRING) right after.
  // local chunk cl of this wave group = global chunk cl * KS + kgrp
  const int cbeg = ksl > 1 ? slice * second.cper : 0, cend = ksl > 1 ? min(nchunk, cbeg + second.cper) : nchunk;
  const int nchunk_g = (cend - cbeg) / KS;
  auto gc0 = [&](int cl) { return (cbeg + cl * KS + kgrp) * KC; };
  const int lastc = gc0(nchunk_g - 1);
#ifdef PCFA_C3_STAMPS
  unsigned long long stamp_prev = 0, stamp_acc[5] = {0, 0, 0, 0, 0};
#endif
#pragma unroll
  for (int j = 0; j < NRING; ++j) load_chunk(min(gc0(j), lastc), ring_raw[j], ring_u[j]);
  store_chunk(gc0(0), 0, ring_raw[0]);
  __syncthreads();
  for (int cbase = 0; cbase < nchunk_g; cbase += NRING) {
#pragma unroll
    for (int j = 0; j < NRING; ++j) {
      const int c = cbase + j;
      if (c < nchunk_g) {
        const int cur = c & 1, nxt = cur ^ 1;
        C3_STAMP(0);
        // the patch of chunk c+1 was requested one iteration ago and goes to the LDS buffer chunk c-1 used (free since
        // the last barrier); the raw slot j is free again once its patch is in LDS, the U slot j after the last MFMA
        // The U slot that is refilled here is the one chunk c - 1 multiplied with (it gets chunk c - 1 + NRING), NOT this
        // chunk's: a load whose destination registers are the B operands of MFMAs that are still queued on the matrix pipe
        // waits for them -- with three waves per SIMD sharing the pipe the four U loads in front of the barrier took 853 of
        // a chunk's 5645 cycles to ISSUE at 220x512 (r05 stamps); one channel pair into the next chunk those MFMAs are done.
        mfma_chunk(cur, ring_u[j],
                   [&] {
                     store_chunk(gc0(c + 1), nxt, ring_raw[(j + 1) % NRING]);   // (past the end: never read)
                     load_u(min(gc0(c - 1 + NRING), lastc), ring_u[(j + NRING - 1) % NRING]);
                   },
                   [&] { load_raw(min(gc0(c + NRING), lastc), ring_raw[j]); });
        __builtin_amdgcn_sched_barrier(0);
        C3_STAMP(1);
        C3_STAMP(2);
        C3_STAMP(3);
        __syncthreads();  // patch c+1 visible; everyone done with the patch of chunk c
        C3_STAMP(4);
      }
    }
  }

#ifdef PCFA_C3_STAMPS
  if (lane == 0 && wave == 0) {
    for (int k = 1; k < 5; ++k) atomicAdd(&c3_stamp_sums[k], stamp_acc[k]);
    atomicAdd(&c3_stamp_sums[0], (unsigned long long)nchunk);
  }
#endif
  if (KS == 2) {   // second group's accumulators -> LDS -> added by the first group (fixed order: deterministic)
    float* red = smem;
    if (kgrp == 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 4 + a) * 16 + r) * 64 + lane] = acc[a][0][r];
    }
    __syncthreads();
    if (kgrp == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][0][r] += red[((wave * 4 + a) * 16 + r) * 64 + lane];
    }
    __syncthreads();   // the image below reuses the same LDS
  }
  const bool worker = kgrp == 0;   // the second group only keeps the barriers of the epilogue company
  // ---- epilogue: passes of 16 output channels through LDS (the image reuses the patch buffers) ----
  static_assert(EPI == 16 * 16 * MS, "epilogue image size");
  float* sM = smem;  // [16 xi][16 channels][MS]
  const int e_tile = tid % TB, e_cl = tid / TB;  // thread (tile, channel) and channel + 8
  const int e_tr = e_tile >> 3, e_tc = e_tile & 7;
  const int oy = y0 + 2 * e_tr, ox = x0 + 2 * e_tc;
#pragma unroll
  for (int pass = 0; pass < CBT / 16; ++pass) {
    const int nb = pass >> 1, half = pass & 1;
    // deferred-ReLU mask of this pass's outputs, requested before the LDS round trip instead of on the store path
    float mk[2][4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
    float ad[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // gradient arriving on the residual path
    if (addend != nullptr && worker) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int n = n0 + pass * 16 + e_cl + 8 * q;
        if (n < N && oy < H && ox < W) {
          const float* ap = addend + (long long)n * plane + (long long)oy * W + ox;
          ad[q][0] = ap[0];
          if (ox + 1 < W) ad[q][1] = ap[1];
          if (oy + 1 < H) {
            ad[q][2] = ap[W];
            if (ox + 1 < W) ad[q][3] = ap[W + 1];
          }
        }
      }
    }
    if (mask != nullptr && worker) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int n = n0 + pass * 16 + e_cl + 8 * q;
        if (n < N && (mask_n == 0 || n < mask_n) && oy < H && ox < W) {
          const float* mp = mask + (long long)n * plane + (long long)oy * W + ox;
          mk[q][0] = mp[0];
          if (ox + 1 < W) mk[q][1] = mp[1];
          if (oy + 1 < H) {
            mk[q][2] = mp[W];
            if (ox + 1 < W) mk[q][3] = mp[W + 1];
          }
        }
      }
    }
    if ((l31 >> 4) == half && worker) {
      const int cl = l31 & 15;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float* m = &sM[((wave * 4 + a) * 16 + cl) * MS];
#pragma unroll
        for (int r = 0; r < 16; ++r) m[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[a][nb][r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int cl = e_cl + 8 * q;
      const int n = n0 + pass * 16 + cl;
      float m[16];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) m[xi] = sM[(xi * 16 + cl) * MS + e_tile];
      // T = A^T M (2 x 4), Y = T A (2 x 2)
      float t0[4], t1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t0[j] = m[j] + m[4 + j] + m[8 + j];
        t1[j] = m[4 + j] - m[8 + j] - m[12 + j];
      }
      const float bv = (bias != nullptr && n < N) ? bias[n] : 0.f;
      float y00 = t0[0] + t0[1] + t0[2] + bv, y01 = t0[1] - t0[2] - t0[3] + bv;
      float y10 = t1[0] + t1[1] + t1[2] + bv, y11 = t1[1] - t1[2] - t1[3] + bv;
      if (ACT == 1) {
        y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f);
      } else if (ACT == 2) {  // torch: x > 0 ? x : x * negative_slope
        y00 = y00 > 0.f ? y00 : y00 * slope; y01 = y01 > 0.f ? y01 : y01 * slope;
        y10 = y10 > 0.f ? y10 : y10 * slope; y11 = y11 > 0.f ? y11 : y11 * slope;
      }
      if (n < N && oy < H && ox < W && worker) {
        const long long oo = (long long)n * plane + (long long)oy * W + ox;
        // data gradient w.r.t. a (Leaky)ReLU output: the deferred activation backward of the producer, factor 1 where its
        // output is positive, else the producer's slope (0: ReLU, written as an exact zero)
        auto masked = [&](float v, float m) { return m > 0.f ? v : (mslope == 0.f ? 0.f : v * mslope); };
        if (mask != nullptr && mask_n == 0) {
          y00 = masked(y00, mk[q][0]); y01 = masked(y01, mk[q][1]); y10 = masked(y10, mk[q][2]); y11 = masked(y11, mk[q][3]);
        }
        if (addend != nullptr) {   // the other consumer's gradient of the same tensor, summed here instead of by autograd
          y00 += ad[q][0]; y01 += ad[q][1]; y10 += ad[q][2]; y11 += ad[q][3];
        }
        if (mask != nullptr && mask_n > 0 && n < mask_n) {   // the gradient is complete only with the addend
          y00 = masked(y00, mk[q][0]); y01 = masked(y01, mk[q][1]); y10 = masked(y10, mk[q][2]); y11 = masked(y11, mk[q][3]);
        }
        float* o = out + oo;
        o[0] = y00;
        if (ox + 1 < W) o[1] = y01;
        if (oy + 1 < H) {
          o[W] = y10;
          if (ox + 1 < W) o[W + 1] = y11;
        }
      }
    }
    __syncthreads();
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static long long f23_packed_floats(int K, int N) {
  return 16LL * (((long long)K + KC - 1) / KC * KC) * (((long long)N + CB - 1) / CB * CB);
}

// [F(2x2,3x3) packing | F(4x4,3x3) packing]: both transforms of the same weights, one buffer per direction
extern "C" long long pcfa_conv3x3_packed_floats(int K, int N) {
  if (K < 1 || N < 1) return -1;
  return f23_packed_floats(K, N) + pcfa_f43_packed_floats(K, N);
}

extern "C" int pcfa_conv3x3_pack_weights(const float* w, float* fwd_packed, float* bwd_packed, int Cout, int Cin,
                                         void* stream) {
  if (!w || (!fwd_packed && !bwd_packed) || Cout < 1 || Cin < 1) return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  for (int dir = 0; dir < 2; ++dir) {
    float* dst = dir == 0 ? fwd_packed : bwd_packed;
    if (!dst) continue;
    const int K = dir == 0 ? Cin : Cout, N = dir == 0 ? Cout : Cin;
    const int nchunk = (K + KC - 1) / KC;
    const long long total = f23_packed_floats(K, N);
    pcfa_launch(conv3x3_pack_kernel, dim3((unsigned)min((total + 255) / 256, 8192LL)), dim3(256), 0, s, w, dst, Cout,
                Cin, dir, K, N, nchunk, total);
    PCFA_LAUNCH_CHECK();
    const int rc = pcfa_f43_pack(w, dst + total, Cout, Cin, dir, s);
    if (rc != PCFA_OK) return rc;
  }
  return PCFA_OK;
}

// dx = dy * (out > 0 ? 1 : slope): backward of LeakyReLU from its OUTPUT (same sign as the input for slope > 0)
__global__ void leaky_relu_bwd_kernel(const float* __restrict__ out, const float* __restrict__ g, float* __restrict__ gx,
                                      float slope, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    gx[i] = out[i] > 0.f ? g[i] : g[i] * slope;
}

#ifdef PCFA_C3_STAMPS
extern "C" __attribute__((visibility("default"))) int dev_c3_stamps(unsigned long long* host8, int reset) {
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(c3_stamp_sums), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(c3_stamp_sums), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

extern "C" int pcfa_leaky_relu_bwd(const float* out, const float* grad_out, float* grad_x, float slope, long long n,
                                   void* stream) {
  if (!out || !grad_out || !grad_x || n < 1) return PCFA_ERR_INVALID_ARG;
  pcfa_launch(leaky_relu_bwd_kernel, dim3((unsigned)min((n + 255) / 256, 4096LL)), dim3(256), 0, (hipStream_t)stream,
              out, grad_out, grad_x, slope, n);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

static int conv3x3_launch(const float* x, const float* packed, const float* bias, const float* mask, float* out, int B,
                          int K, int N, int H, int W, int act, float slope, void* stream, const float* x2 = nullptr,
                          const float* packed2 = nullptr, const float* bias2 = nullptr, float* out2 = nullptr,
                          int K2 = 0, int N2 = 0, const float* addend = nullptr, int mask_n = 0, int ksl = 1);

extern "C" int pcfa_conv3x3_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int K,
                                int N, int H, int W, int relu, void* stream) {
  return conv3x3_launch(x, packed, bias, nullptr, out, B, K, N, H, W, relu ? 1 : 0, 0.f, stream);
}

extern "C" int pcfa_conv3x3_act_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int K,
                                    int N, int H, int W, int act, float slope, void* stream) {
  return conv3x3_launch(x, packed, bias, nullptr, out, B, K, N, H, W, act, slope, stream);
}

extern "C" int pcfa_conv3x3_masked_fwd(const float* x, const float* packed, const float* mask, float* out, int B, int K,
                                       int N, int H, int W, void* stream) {
  if (!mask) return PCFA_ERR_INVALID_ARG;
  return conv3x3_launch(x, packed, nullptr, mask, out, B, K, N, H, W, 0, 0.f, stream);
}

extern "C" int pcfa_conv3x3_act_fwd_pair(const float* x, const float* packed, const float* bias, float* out, int K, int N,
                                         const float* x2, const float* packed2, const float* bias2, float* out2, int K2,
                                         int N2, int H, int W, int act, float slope, void* stream) {
  if (!x2 || !packed2 || !out2 || K2 < 1 || N2 < 1 || !aligned16(packed2)) return PCFA_ERR_INVALID_ARG;
  if ((K % KC == 0) != (K2 % KC == 0)) return PCFA_ERR_UNSUPPORTED;   // one kernel instance serves both
  if (f23_packed_floats(K2, N2) > 0x7fffffffLL || (long long)K2 * H * W > 0x7fffffffLL) return PCFA_ERR_UNSUPPORTED;
  return conv3x3_launch(x, packed, bias, nullptr, out, 1, K, N, H, W, act, slope, stream, x2, packed2, bias2, out2, K2,
                        N2);
}

extern "C" int pcfa_conv3x3_fused_bwd(const float* g, const float* packed_bwd, const float* mask, const float* addend,
                                      float* grad_in, int B, int K, int N, int H, int W, void* stream) {
  return conv3x3_launch(g, packed_bwd, nullptr, mask, grad_in, B, K, N, H, W, 0, 0.f, stream, nullptr, nullptr, nullptr,
                        nullptr, 0, 0, addend);
}

// K slices of the F(2x2,3x3) kernel over workgroups (Second::ksl), 1 = none.  Small single-image maps (PWC-Net's levels 6-4:
// 6x20 .. 24x80) are 3..48 workgroups that walk 16-79 chunks one after the other -- a ~1 us per chunk latency chain on an
// empty chip; `ksl` slices turn it into ksl x the workgroups with a ksl x shorter chain, and f43_finish_kernel adds the
// partial outputs in index order (deterministic).  PCFA_CONV3X3_KSL=0 switches it off, =n forces n (dev A/B).
static int f23_kslices(int B, int K, int N, int H, int W) {
  static const int env = getenv("PCFA_CONV3X3_KSL") ? atoi(getenv("PCFA_CONV3X3_KSL")) : -1;
  if (env == 0 || ((long long)H * W) % 4 != 0) return 1;
  const int nchunk = (K + KC - 1) / KC;
  const long long nwg = (long long)pcfa_cdiv(W, 2 * TC) * pcfa_cdiv(H, 8) * ((N + 31) / 32) * B;
  long long want = env > 0 ? env : min(256 / max(nwg, 1LL), (long long)nchunk / 2);
  if (env < 0 && (B > 2 || (long long)H * W > 2048 || nwg > 128)) want = 1;   // (B = 2: PWC-Net's two pyramids in one batch)
  if (want < 2 || nchunk < 2) return 1;
  want = min(want, (long long)nchunk);
  const int cper = (int)((nchunk + want - 1) / want);
  return (nchunk + cper - 1) / cper;   // every slice holds at least one chunk
}

// Which transform serves a shape.  F(4x4,3x3) needs 1.78x fewer matrix instructions than F(2x2,3x3), at 6x its rounding
// error (2e-6 against 3.5e-7 relative per layer).
//   r05 policy: F(4x4,3x3) serves ONLY PWC-Net's single-image, many-input-channel decoder shapes (the B == 1 block below),
//   where it is worth 1.0 ms of a 4.8 ms closure and the GPU gradient measures CLOSER to the float64 port than the fp32
//   port does (profiles/r05/fp64_arbiter.json).  RAFT / GMA never take it any more: on their shapes it bought nothing at
//   the step level (bench at driver settings, back to back: r04 policy 7.44, F(2x2,3x3) only 7.45, large maps only 7.46
//   attack steps/s -- three waves per SIMD leave its loop 168 registers, so the U operand stream runs one channel pair
//   ahead and the channel-split path pays for its partial outputs), while the same arbiter puts the gradient of the r04
//   policy 1.1-2.3x farther from float64 than the port's and the F(2x2,3x3)-only build level with the port
//   (geometric mean over ten trajectory points: port 5.9e-4, F(2x2,3x3) 6.3e-4, r04 policy 8.3e-4).
// PCFA_CONV3X3_ALGO = f23 | f43 | r04 overrides (read once; A/B of tools/parity_arbiter.py and tools/dev): r04 = the policy
// of round 4 (additionally: the 55x128 two-way channel split at K N >= 192 x 256 and every >= 100000-pixel map).
static bool use_f43(int B, int K, int N, int H, int W) {
  static const int forced = [] {
    const char* e = getenv("PCFA_CONV3X3_ALGO");
    if (e == nullptr) return 0;
    if (e[0] == 'r') return 4;
    return e[1] == '4' ? 43 : 23;
  }();
  if (!pcfa_f43_supported(B, K, N, H, W)) return false;
  if (forced == 43 || forced == 23) return forced == 43;
  // Single images with many input channels (PWC-Net's dense decoder blocks, PWCNet.py:110-158: 117..629 -> 128..32 at
  // 96x320 .. 6x20): the F(2x2,3x3) kernel has no channel split, so a small map is a handful of workgroups each
  // walking the whole K (81 us at 533 -> 64 on 24x80); F(4x4,3x3) splits K over workgroups (27 us).  Thresholds from
  // tools/dev/conv3x3_shapes_ab.py (every conv3x3 shape of a PWC-Net closure, both algorithms): -1.0 ms per closure.
  if (B == 1) {
    const long long px = (long long)H * W;
    // levels 6-4 (6x20 .. 24x80): the F(2x2,3x3) kernel with K sliced over workgroups (f23_kslices; r04) beats the
    // channel-split F(4x4,3x3) path on every such shape but one (tools/dev/conv_ksl_ab.sh: 21-30 us -> 16-23 us)
    if (px <= 2048 && f23_kslices(B, K, N, H, W) > 1) return false;
    if (px <= 4096 && K >= 176) return true;
    if (px > 4096 && px <= 16384 && K >= 384) return true;
    if (px >= 16384 && px < 100000 && (long long)K * N >= 15000) return true;
  }
  if (forced != 4) return false;
  // ---- round 4's additional cases (PCFA_CONV3X3_ALGO=r04 only) ----
  if (H < 24 || W < 64 || K < 16) return false;
  const int ks = pcfa_f43_ksplit(B, K, N, H, W);
  if (ks > 1) return ks == 2 && (long long)K * N >= 192LL * 256;
  return (long long)H * W >= 100000;
}

extern "C" int pcfa_conv3x3_algo(int B, int K, int N, int H, int W) {
  if (B < 1 || K < 1 || N < 1 || H < 1 || W < 1) return 0;
  return use_f43(B, K, N, H, W) ? 43 : 23;
}

extern "C" size_t pcfa_conv3x3_workspace_bytes(int B, int K, int N, int H, int W) {
  if (B < 1 || K < 1 || N < 1 || H < 1 || W < 1) return 0;
  // (an upper bound over both paths: a misaligned view falls from F(4x4,3x3) through to the sliced F(2x2,3x3) kernel)
  const int ksl = f23_kslices(B, K, N, H, W);
  const size_t sliced = ksl > 1 ? (size_t)ksl * B * N * H * W * sizeof(float) : 0;
  return max(use_f43(B, K, N, H, W) ? pcfa_f43_workspace_bytes(B, K, N, H, W) : (size_t)0, sliced);
}

extern "C" int pcfa_conv3x3_run(const float* x, const float* packed, const float* bias, const float* mask,
                                const float* addend, float* out, int B, int K, int N, int H, int W, int act,
                                float slope, int mask_channels, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !packed || !out || B < 1 || K < 1 || N < 1 || H < 1 || W < 1 || act < 0 || act > 2 || mask_channels < 0 ||
      mask_channels > N || (mask_channels > 0 && (!mask || act != 0)) || (mask && act == 2))
    return PCFA_ERR_INVALID_ARG;
  if (use_f43(B, K, N, H, W)) {
    const int rc = pcfa_f43_run(x, packed + f23_packed_floats(K, N), bias, mask, addend, out, B, K, N, H, W, act,
                                slope, mask_channels, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc != PCFA_ERR_UNSUPPORTED) return rc;   // (misaligned views fall through to the F(2x2,3x3) kernel)
  }
  const int ksl = f23_kslices(B, K, N, H, W);
  if (ksl > 1 && workspace && aligned16(workspace) && aligned16(out) && (!mask || aligned16(mask)) &&
      (!addend || aligned16(addend)) && workspace_bytes >= (size_t)ksl * B * N * H * W * sizeof(float)) {
    float* part = (float*)workspace;
    const int rc = conv3x3_launch(x, packed, nullptr, nullptr, part, B, K, N, H, W, 0, 0.f, stream, nullptr, nullptr,
                                  nullptr, nullptr, 0, 0, nullptr, 0, ksl);
    if (rc != PCFA_OK) return rc;
    return pcfa_f43_finish(part, bias, mask, addend, out, ksl, B, N, H, W, act, slope, mask_channels, (hipStream_t)stream,
                           23);
  }
  return conv3x3_launch(x, packed, bias, mask, out, B, K, N, H, W, act, slope, stream, nullptr, nullptr, nullptr,
                        nullptr, 0, 0, addend, mask_channels);
}

static int conv3x3_launch(const float* x, const float* packed, const float* bias, const float* mask, float* out, int B,
                          int K, int N, int H, int W, int act, float slope, void* stream, const float* x2,
                          const float* packed2, const float* bias2, float* out2, int K2, int N2, const float* addend,
                          int mask_n, int ksl) {
  if (act < 0 || act > 2 || mask_n < 0 || (mask_n > 0 && (act != 0 || !mask))) return PCFA_ERR_INVALID_ARG;
  if (!x || !packed || !out || B < 1 || K < 1 || N < 1 || H < 1 || W < 1 || !aligned16(packed))
    return PCFA_ERR_INVALID_ARG;
  const int Npad = (N + CB - 1) / CB * CB;
  const int blocks_x = pcfa_cdiv(W, 2 * TC), blocks_y = pcfa_cdiv(H, 8);
  const long long gx = (long long)blocks_x * blocks_y;
  // 32-bit element offsets inside one image and inside the packed weights
  if (gx > 0x7fffffffLL || B > 65535 || Npad / CB > 65535 || (long long)K * H * W > 0x7fffffffLL ||
      f23_packed_floats(K, N) > 0x7fffffffLL)
    return PCFA_ERR_UNSUPPORTED;
  dim3 grid((unsigned)gx, Npad / CB, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  // 64-channel blocks only when there are enough of them to fill the chip twice over; otherwise 32-channel blocks
  // (twice the workgroups, two resident per CU)
  int mt = 1;
  if (const char* e = getenv("PCFA_CONV3X3_MT")) mt = atoi(e) == 2 ? 2 : 1;  // A/B switch for tools/dev
  if (mt == 2) {
    const int by2 = pcfa_cdiv(H, 16);
    grid.x = (unsigned)(blocks_x * by2);
    block.x = 512;
  }
  grid.y = Npad / 32;
  const int nchunk_all = (K + KC - 1) / KC, cper = (nchunk_all + ksl - 1) / ksl;
  if (ksl > 1) {   // K slices over workgroups: raw partial sums (the caller runs the finish pass)
    if (x2 != nullptr || mt != 1 || bias || mask || addend || act != 0 || (cper * (ksl - 1)) >= nchunk_all) return PCFA_ERR_INVALID_ARG;
    grid.z = (unsigned)(B * ksl);
    if (grid.z > 65535) return PCFA_ERR_UNSUPPORTED;
  }
  Second second{x2, packed2, bias2, out2, K2, N2, (int)grid.y, mask_n, ksl, cper, PcfaXcdMap{0, 0, 0, 0}};
  if (x2 != nullptr) {
    if (mt == 2) return PCFA_ERR_UNSUPPORTED;
    grid.y += (unsigned)((N2 + CB - 1) / CB * CB / 32);
    if (grid.y > 65535) return PCFA_ERR_UNSUPPORTED;
  }
  // XCD-aware placement (common.hpp): x = tile blocks (the input planes follow them), y = channel blocks (the U slices
  // follow them).  Measured (tools/bench_conv3x3.py, on / off): 64 -> 64 at 220x512 117.9 / 124.4 us (batch 2), 61.3 / 63.1
  // (batch 1); neutral within 1-2 % on the 110x256 and 55x128 maps -- so only grids of >= 1024 workgroups take it.
  // PCFA_XCD_MAP=0: plain 2-D grid (A/B).
  const long long nwg = (long long)grid.x * grid.y * grid.z;   // (live workgroups: the XCD map below pads the grid)
  static const bool xcd_on = !(getenv("PCFA_XCD_MAP") && atoi(getenv("PCFA_XCD_MAP")) == 0);
  if (xcd_on && mt == 1 && (long long)grid.x * grid.y >= 1024) {
    second.xcd = pcfa_xcd_pick((int)grid.x, (int)grid.y, 4.0 * K * H * W, 4.0 * 16 * K * 32 * grid.y);
    if (second.xcd.a != 0) {
      grid.x = pcfa_xcd_grid(second.xcd);
      grid.y = 1;
    }
  }
#define PCFA_C3_ARGS grid, block, 0, s, x, packed, bias, mask, addend, out, K, N, Npad, H, W, blocks_x, slope, second
#define PCFA_C3_LAUNCH(MT_, KF_, NR_)                                                              \
  do {                                                                                             \
    if (act == 1) pcfa_launch(conv3x3_winograd_kernel<1, 32, MT_, KF_, NR_>, PCFA_C3_ARGS);        \
    else if (act == 2) pcfa_launch(conv3x3_winograd_kernel<2, 32, MT_, KF_, NR_>, PCFA_C3_ARGS);   \
    else pcfa_launch(conv3x3_winograd_kernel<0, 32, MT_, KF_, NR_>, PCFA_C3_ARGS);                 \
  } while (0)
  const bool kfull = K % KC == 0;
  // at most one workgroup per CU: split K inside the workgroup (8 waves, two per SIMD)
  static const int ks_env = getenv("PCFA_CONV3X3_KS") ? atoi(getenv("PCFA_CONV3X3_KS")) : 0;   // tuning override (1 / 2)
  if (ksl == 1 && mt == 1 && x2 == nullptr && K % (2 * KC) == 0 && (ks_env ? ks_env == 2 : nwg <= 256)) {
    block.x = 512;
    if (act == 1) pcfa_launch(conv3x3_winograd_kernel<1, 32, 1, true, 2, 2>, PCFA_C3_ARGS);
    else if (act == 2) pcfa_launch(conv3x3_winograd_kernel<2, 32, 1, true, 2, 2>, PCFA_C3_ARGS);
    else pcfa_launch(conv3x3_winograd_kernel<0, 32, 1, true, 2, 2>, PCFA_C3_ARGS);
    PCFA_LAUNCH_CHECK();
    return PCFA_OK;
  }
  int deep = 0;  // measured: the 3-deep ring (2 waves per SIMD) is never faster, also not on small grids
  if (const char* e = getenv("PCFA_CONV3X3_RING")) deep = atoi(e) == 3;  // A/B switch for tools/dev
  if (mt == 2) {
    if (kfull) PCFA_C3_LAUNCH(2, true, 2); else PCFA_C3_LAUNCH(2, false, 2);
  } else if (deep) {
    if (kfull) PCFA_C3_LAUNCH(1, true, 3); else PCFA_C3_LAUNCH(1, false, 3);
  } else {
    if (kfull) PCFA_C3_LAUNCH(1, true, 2); else PCFA_C3_LAUNCH(1, false, 2);
  }
#undef PCFA_C3_LAUNCH
#undef PCFA_C3_ARGS
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
