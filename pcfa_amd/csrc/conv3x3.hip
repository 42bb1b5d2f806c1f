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
// Workgroup = 4 waves: 32 tiles (4 tile rows x 8 tile columns = 8 x 16 output pixels) x 64 output channels.
// Per chunk of 8 input channels: the 10 x 18 input patch goes global -> registers -> LDS, thread (tile, channel)
// transforms its 4x4 patch to V[16][8][32] in LDS, wave w multiplies xi = 4w..4w+3 (4 MFMAs per xi and 32-channel
// half: 32 per wave and chunk, 128 accumulator registers), the next chunk's patch and U slice (32 KB) prefetched in
// registers meanwhile.  Epilogue: accumulators -> LDS in four passes of 16 channels, thread (channel, tile) applies
// A^T . A, adds the bias, optionally ReLU, stores 2x2 pixels.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TC = 8;                            // tile columns per workgroup (16 output columns)
constexpr int PC = 2 * TC + 2;                   // input patch columns (18)
constexpr int PCP = 20;                          // padded patch row stride
constexpr int KC = 8;                            // input channels per chunk
constexpr int CB = 64;                           // packing granularity of the output channels (U row padding)

// U[xi][k][n_pad] = (G g G^T)[xi] with g = w[n][k] (forward) or the flipped w[k][n] (data gradient);
// columns n >= N are zero.
__global__ void conv3x3_pack_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin,
                                    int backward, int K, int N, int Npad) {
  const long long total = (long long)K * Npad;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(e / Npad), n = (int)(e - (long long)k * Npad);
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float v = 0.f;
        if (n < N) v = backward ? w[(((long long)k * Cin + n) * 3 + (2 - i)) * 3 + (2 - j)]
                                : w[(((long long)n * Cin + k) * 3 + i) * 3 + j];
        g[i][j] = v;
      }
    float t[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      t[0][j] = g[0][j];
      t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
      t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
      t[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]),
                  u3 = t[i][2];
      U[((long long)(4 * i + 0) * K + k) * Npad + n] = u0;
      U[((long long)(4 * i + 1) * K + k) * Npad + n] = u1;
      U[((long long)(4 * i + 2) * K + k) * Npad + n] = u2;
      U[((long long)(4 * i + 3) * K + k) * Npad + n] = u3;
    }
  }
}

// x [B][K][H][W], U [16][K][Npad], out [B][N][H][W]; grid = (tile blocks, Npad / CB, B).
// CBT = output channels per workgroup: 64 (128 accumulator registers per lane, one block per CU) or 32 (64
// accumulators, 76.8 KB LDS: two blocks per CU overlap each other's staging / transform / barrier phases).
// MT = 32-tile groups per workgroup (4 waves each): MT = 2 -> 64 tiles (16 x 16 output pixels), 8 waves sharing one
// U slice, i.e. half the L2 -> LDS weight traffic per multiply.
// KFULL: K % 8 == 0 -- the staging loop then carries no channel bookkeeping at all (per-thread base pointers plus
// one scalar chunk offset; the generic variant clamps and masks the channel index of every element).
template <bool RELU, int CBT, int MT, bool KFULL>
__global__ __launch_bounds__(256 * MT) __attribute__((amdgpu_waves_per_eu(MT == 1 ? 3 : 2))) void conv3x3_winograd_kernel(
    const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
    float* __restrict__ out, int K, int N, int Npad, int H, int W, int blocks_x) {
  constexpr int NT = 256 * MT;                         // threads
  constexpr int TR = 4 * MT, TB = TR * TC;             // tile rows / tiles per workgroup
  constexpr int PR = 2 * TR + 2;                       // input patch rows
  constexpr int RAW = KC * PR * PCP;
  constexpr int RAW_LOADS = (KC * PR * PC + NT - 1) / NT;
  constexpr int MS = TB + 1;                           // epilogue image: [16][16 channels][MS]
  // double-buffered staging of the patch and of V; the U operands go from L2 straight into registers (every U
  // element is consumed by exactly one wave -- xi = 4w..4w+3 -- so an LDS round trip buys nothing)
  constexpr int VSZ = 16 * KC * TB, NB = CBT / 32, NU = 4 * (KC / 2) * NB;  // U dwords per lane and chunk
  constexpr int EPI = 16 * 16 * (TB + 1);                                   // epilogue image
  constexpr int LDSF = 2 * (RAW + VSZ) > 2 * RAW + EPI ? 2 * (RAW + VSZ) : 2 * RAW + EPI;
  __shared__ __attribute__((aligned(16))) float smem[LDSF];
  float* sRaw = smem;               // [2][RAW]
  float* sV = smem + 2 * RAW;       // [2][VSZ]

  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, mt = tid >> 8;  // wave: xi group, mt: tile group
  const int l31 = lane & 31, lh = lane >> 5;
  const int by = blockIdx.x / blocks_x, bx = blockIdx.x - by * blocks_x;
  const int y0 = by * (2 * TR), x0 = bx * (2 * TC);  // first output pixel of the block
  const int n0 = blockIdx.y * CBT;
  if (n0 >= N) return;  // (the packing pads N to 64: a 32-channel block may lie entirely in the padding)
  const long long plane = (long long)H * W;
  x += (long long)blockIdx.z * K * plane;
  out += (long long)blockIdx.z * N * plane;

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
    rdst[i] = e < KC * PR * PC ? (ch * PR + r) * PCP + c : -1;
    rok[i] = e < KC * PR * PC && yy >= 0 && yy < H && xx >= 0 && xx < W;
    praw[i] = (unsigned)(min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1) + (KFULL ? rch[i] * (int)plane : 0));
  }
  // U operand of MFMA (a, kp, b): U[xi = 4*wave + a][k = c0 + kp + lh][n0 + 32 b + l31]
  unsigned pu[4];  // floats from U
#pragma unroll
  for (int a = 0; a < 4; ++a) pu[a] = (unsigned)(((4 * wave + a) * K + lh) * Npad + n0 + l31);

  // register ring: the global loads of a chunk are issued NRING-1 chunks before its MFMAs
  constexpr int NRING = 2;
  float ring_raw[NRING][RAW_LOADS];
  float ring_u[NRING][NU];
  auto load_chunk = [&](int c0, float (&rraw)[RAW_LOADS], float (&ru)[NU]) {
    if (KFULL) {
      const unsigned xo = (unsigned)(c0 * (int)plane), uo = (unsigned)(c0 * Npad);  // wave-uniform chunk offsets
#pragma unroll
      for (int i = 0; i < RAW_LOADS; ++i) rraw[i] = x[praw[i] + xo];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kp = 0; kp < KC; kp += 2)
#pragma unroll
          for (int b = 0; b < NB; ++b) ru[(a * (KC / 2) + kp / 2) * NB + b] = U[pu[a] + uo + (unsigned)(kp * Npad + 32 * b)];
    } else {
#pragma unroll
      for (int i = 0; i < RAW_LOADS; ++i)
        rraw[i] = x[praw[i] + (unsigned)(min(c0 + rch[i], K - 1) * (int)plane)];  // channels >= K: zeroed at the LDS write
      // rows k >= K: any finite value will do (their V operand is zero); clamp the row index
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int kp = 0; kp < KC; kp += 2)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            ru[(a * (KC / 2) + kp / 2) * NB + b] = U[pu[a] + (unsigned)((min(c0 + kp + lh, K - 1) - lh) * Npad + 32 * b)];
    }
  };
  auto store_chunk = [&](int c0, int buf, const float (&rraw)[RAW_LOADS]) {
    float* sRaw = smem + buf * RAW;
#pragma unroll
    for (int i = 0; i < RAW_LOADS; ++i)
      if (rdst[i] >= 0) sRaw[rdst[i]] = (rok[i] && (KFULL || c0 + rch[i] < K)) ? rraw[i] : 0.f;
  };
  // thread (tile, channel) of the input transform
  const int t_tile = tid % TB, t_ch = tid / TB;
  const int t_tr = t_tile >> 3, t_tc = t_tile & 7;
  auto transform = [&](int buf) {
    const float* p = &sRaw[buf * RAW + (t_ch * PR + 2 * t_tr) * PCP + 2 * t_tc];
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = p[i * PCP + j];
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0][j] - d[2][j];
      t[1][j] = d[1][j] + d[2][j];
      t[2][j] = d[2][j] - d[1][j];
      t[3][j] = d[1][j] - d[3][j];
    }
    float* v = &sV[buf * VSZ + t_ch * TB + t_tile];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[(4 * i + 0) * (KC * TB)] = t[i][0] - t[i][2];
      v[(4 * i + 1) * (KC * TB)] = t[i][1] + t[i][2];
      v[(4 * i + 2) * (KC * TB)] = t[i][2] - t[i][1];
      v[(4 * i + 3) * (KC * TB)] = t[i][1] - t[i][3];
    }
  };

  f32x16 acc[4][NB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // MFMAs of the chunk staged in `buf` for the xi-quarter a of this wave; ru = the chunk's U operands
  auto mfma_quarter = [&](int buf, int a, const float (&ru)[NU]) {
    const int xi = wave * 4 + a;
    const float* vp = &sV[buf * VSZ + (xi * KC + lh) * TB + mt * 32 + l31];
#pragma unroll
    for (int kp = 0; kp < KC; kp += 2) {
      const float av = vp[kp * TB];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ru[(a * (KC / 2) + kp / 2) * NB + b], acc[a][b], 0, 0, 0);
    }
  };

  // Pipeline: while the matrix pipe works on chunk c (V, U in buffer c&1), the same waves write chunk c+1 to the
  // other buffer, transform it, and have chunk c+2 in flight from global memory.  MFMAs are issued on both sides of
  // the first barrier so that the barrier wait overlaps matrix work.
  // Ring slot c % NRING holds chunk c: its patch is written to LDS one iteration before its MFMAs, its U operands
  // stay in registers until the MFMAs have consumed them; the slot is reloaded (chunk c + NRING) right after.
  const int nchunk = (K + KC - 1) / KC;
  const int lastc = (nchunk - 1) * KC;
#pragma unroll
  for (int j = 0; j < NRING; ++j) load_chunk(min(j * KC, lastc), ring_raw[j], ring_u[j]);
  store_chunk(0, 0, ring_raw[0]);
  __syncthreads();
  transform(0);
  __syncthreads();
  for (int cbase = 0; cbase < nchunk; cbase += NRING) {
#pragma unroll
    for (int j = 0; j < NRING; ++j) {
      const int c = cbase + j;
      if (c < nchunk) {
        const int cur = c & 1, nxt = cur ^ 1;
        store_chunk((c + 1) * KC, nxt, ring_raw[(j + 1) % NRING]);  // patch of chunk c+1 (past the end: zeros)
        mfma_quarter(cur, 0, ring_u[j]);
        mfma_quarter(cur, 1, ring_u[j]);
        __syncthreads();                 // patch c+1 is in LDS
        if (c + 1 < nchunk) transform(nxt);
        mfma_quarter(cur, 2, ring_u[j]);
        mfma_quarter(cur, 3, ring_u[j]);
        __builtin_amdgcn_sched_barrier(0);
        load_chunk(min((c + NRING) * KC, lastc), ring_raw[j], ring_u[j]);  // slot j is free again
        __syncthreads();                 // V of chunk c+1 complete; everyone done with the buffers of chunk c
      }
    }
  }

  // ---- epilogue: four passes of 16 output channels through LDS (the image reuses sV + sU) ----
  static_assert(EPI == 16 * 16 * MS, "epilogue image size");
  float* sM = sV;  // [16 xi][16 channels][MS]
  const int e_tile = tid % TB, e_cl = tid / TB;  // thread (tile, channel) and channel + 8
  const int e_tr = e_tile >> 3, e_tc = e_tile & 7;
  const int oy = y0 + 2 * e_tr, ox = x0 + 2 * e_tc;
#pragma unroll
  for (int pass = 0; pass < CBT / 16; ++pass) {
    const int nb = pass >> 1, half = pass & 1;
    if ((l31 >> 4) == half) {
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
      if (RELU) {
        y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f);
      }
      if (n < N && oy < H && ox < W) {
        float* o = out + (long long)n * plane + (long long)oy * W + ox;
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

extern "C" long long pcfa_conv3x3_packed_floats(int K, int N) {
  if (K < 1 || N < 1) return -1;
  return 16LL * K * (((long long)N + CB - 1) / CB * CB);
}

extern "C" int pcfa_conv3x3_pack_weights(const float* w, float* fwd_packed, float* bwd_packed, int Cout, int Cin,
                                         void* stream) {
  if (!w || (!fwd_packed && !bwd_packed) || Cout < 1 || Cin < 1) return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (fwd_packed) {
    const int K = Cin, N = Cout, Npad = (N + CB - 1) / CB * CB;
    pcfa_launch(conv3x3_pack_kernel, dim3((unsigned)min(((long long)K * Npad + 255) / 256, 4096LL)), dim3(256), 0, s, w,
                fwd_packed, Cout, Cin, 0, K, N, Npad);
    PCFA_LAUNCH_CHECK();
  }
  if (bwd_packed) {
    const int K = Cout, N = Cin, Npad = (N + CB - 1) / CB * CB;
    pcfa_launch(conv3x3_pack_kernel, dim3((unsigned)min(((long long)K * Npad + 255) / 256, 4096LL)), dim3(256), 0, s, w,
                bwd_packed, Cout, Cin, 1, K, N, Npad);
    PCFA_LAUNCH_CHECK();
  }
  return PCFA_OK;
}

extern "C" int pcfa_conv3x3_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int K,
                                int N, int H, int W, int relu, void* stream) {
  if (!x || !packed || !out || B < 1 || K < 1 || N < 1 || H < 1 || W < 1 || !aligned16(packed))
    return PCFA_ERR_INVALID_ARG;
  const int Npad = (N + CB - 1) / CB * CB;
  const int blocks_x = pcfa_cdiv(W, 2 * TC), blocks_y = pcfa_cdiv(H, 8);
  const long long gx = (long long)blocks_x * blocks_y;
  // 32-bit element offsets inside one image and inside the packed weights
  if (gx > 0x7fffffffLL || B > 65535 || Npad / CB > 65535 || (long long)K * H * W > 0x7fffffffLL ||
      16LL * K * Npad > 0x7fffffffLL)
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
#define PCFA_C3_ARGS grid, block, 0, s, x, packed, bias, out, K, N, Npad, H, W, blocks_x
#define PCFA_C3_LAUNCH(MT_, KF_)                                                                   \
  do {                                                                                             \
    if (relu) pcfa_launch(conv3x3_winograd_kernel<true, 32, MT_, KF_>, PCFA_C3_ARGS);              \
    else pcfa_launch(conv3x3_winograd_kernel<false, 32, MT_, KF_>, PCFA_C3_ARGS);                  \
  } while (0)
  const bool kfull = K % KC == 0;
  if (mt == 2) {
    if (kfull) PCFA_C3_LAUNCH(2, true); else PCFA_C3_LAUNCH(2, false);
  } else {
    if (kfull) PCFA_C3_LAUNCH(1, true); else PCFA_C3_LAUNCH(1, false);
  }
#undef PCFA_C3_LAUNCH
#undef PCFA_C3_ARGS
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
