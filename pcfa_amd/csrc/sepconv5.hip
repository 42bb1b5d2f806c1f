// 5-tap separable convolution of the RAFT/GMA SepConvGRU gates for gfx950.
//
// Replaces the six Conv2d(c, 128, (1,5), padding=(0,2)) / ((5,1), padding=(2,0)) of
// reference models/raft/update.py:33-60 (models/gma/update.py:33-60), forward and data gradient, which
// the library path runs as im2col + GEMM + (col2im in the backward).
//
// MI355X formulation: implicit GEMM  out[Cout x pixels] = Wp[Cout x (t,ci)] . X[(t,ci) x pixels]  on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation).  A pixel tile is 64 consecutive x of one image
// row, so the five shifted operand rows of a channel are five views of ONE halo'd LDS row (1x5) or five staged rows
// (5x1): the 5x im2col expansion never exists in memory.  The operand may be the channel concatenation of two
// tensors ([h | motion features]), read in place: no torch.cat.  The data gradient is the same operator with
// the taps flipped and the channel roles swapped (pcfa_sepconv5_pack_weights emits both packings).
//
// Block = 4 waves, 64(out-channels) x 64(pixels) tile, one 32x32 MFMA tile per wave; K streamed in chunks of
// 8 input channels x 5 taps through an LDS double buffer.  The global loads of a stage are issued three stages
// before its MFMAs (ring of four register sets): with one or two workgroups per CU nothing else hides the
// L2/HBM latency.
#include <cstdlib>
#include "sepconv5.hpp"

#ifndef PCFA_SC5_PRE
#define PCFA_SC5_PRE 4   // LDS operand reads issued this many MFMAs ahead of their use
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64, TN = 64, KC = 8, TAPS = 5;
constexpr int NR = 4;                       // register ring: global loads run NR-1 stages ahead of the MFMAs
constexpr int A_STRIDE = TM;                // sA[t][ci][m]
constexpr int A_TILE = TAPS * KC * A_STRIDE;
constexpr int A_VEC = A_TILE / 4;           // float4 per stage (640)
constexpr int A_REGS = (A_VEC + 255) / 256;
// 1x5: sB[ci][4 + (x - x0)], x - x0 in [-2, 66): the 64 interior floats sit 16-B aligned, 2 halo floats either side
constexpr int BH_ROW = 72, BH_X0 = 4;
constexpr int B_TILE_H = KC * BH_ROW;
// 5x1: sB[ci][t][n]
constexpr int BV_ROW = TN;
constexpr int B_TILE_V = KC * TAPS * BV_ROW;
constexpr int BV_VEC = B_TILE_V / 4;
constexpr int BV_REGS = (BV_VEC + 255) / 256;

using namespace pcfa_sc5;   // Operand, OutSplit, GruEpi (sepconv5.hpp)

__device__ __forceinline__ float sc5_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }   // as gru_math.hip

__device__ __forceinline__ const float* channel_plane(const Operand& in, int ci, long long plane) {
  return ci < in.Ca ? in.a + ci * plane : in.b + (ci - in.Ca) * plane;
}

__device__ __forceinline__ f32x4 load4(const float* p, int n_valid, bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (vec && n_valid >= 4) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else if (n_valid > 0) {
    v.x = p[0];
    if (n_valid > 1) v.y = p[1];
    if (n_valid > 2) v.z = p[2];
    if (n_valid > 3) v.w = p[3];
  }
  return v;
}

__device__ __forceinline__ f32x4 keep_if(bool ok, f32x4 v) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  return ok ? v : zero;
}

// FAST: Cin % (KC * NR) == 0, Cout % 4 == 0, W % 4 == 0 and 16-B aligned operands.  Every global load is then an
// unconditional float4/dword from a clamped (always valid) address and zero padding is applied when the value
// is written to LDS -- no branch sits between a load and its use, so the loads really stay in flight across
// stages.  The generic variant keeps predicated element-wise loads for ragged shapes.
// KS = 2: in-workgroup split-K.  Two groups of four waves run the same pipeline on the two halves of the input
// channels (own LDS stages, shared barriers) and the second group's accumulators are added through LDS at the end, in
// a fixed order.  For launches whose grid is smaller than the chip (the q convolutions: 220 workgroups = one 4-wave
// workgroup on 220 of 256 CUs, one wave per SIMD and nothing to hide a barrier or an LDS wait behind) this doubles the
// waves per CU without changing the tile count.
template <bool VERT, bool FAST, int KS = 1>
__global__ __launch_bounds__(256 * KS) void sepconv5_kernel(Operand in, const float* __restrict__ wp,
                                                            OutSplit out, int Cout, int H, int W,
                                                            int tiles_x, int vec_w, int vec_x, GruEpi epi) {
  static_assert(KS == 1 || FAST, "split-K rides on the branch-free staging path");
  __shared__ __attribute__((aligned(16))) float sA_[KS][2][A_TILE];
  __shared__ __attribute__((aligned(16))) float sB_[KS][2][VERT ? B_TILE_V : B_TILE_H];
  const int kgrp = KS == 1 ? 0 : (int)(threadIdx.x >> 8);   // which half of K this wave group owns
  float (*sA)[A_TILE] = sA_[kgrp];
  float (*sB)[VERT ? B_TILE_V : B_TILE_H] = sB_[kgrp];

  const long long plane = (long long)H * W;
  const int y = blockIdx.x / tiles_x;
  const int x0 = (blockIdx.x - y * tiles_x) * TN;
  const int m0 = blockIdx.y * TM;
  in.a += (long long)blockIdx.z * in.Ca * plane;
  if (in.b) in.b += (long long)blockIdx.z * (in.Cin - in.Ca) * plane;
  out.a += (long long)blockIdx.z * out.Ca * plane;
  if (out.b) out.b += (long long)blockIdx.z * (Cout - out.Ca) * plane;

  const int tid = threadIdx.x & 255;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // 1x5 loader roles: threads [0,128) fetch the aligned interior float4 (ci = tid/16, x = x0 + 4*(tid%16)),
  // threads [128,160) one halo float each (ci = (tid-128)/4, offsets -2,-1,64,65).
  // (Every thread issues both loads -- the upper threads repeat addresses of the lower ones -- so that no
  // branch surrounds a load; only the named threads write to LDS.)
  const int ici = (tid & 127) >> 4, idx = (tid & 15) * 4;
  const int hci = (tid & 31) >> 2, hq = tid & 3;
  const int hdx = hq < 2 ? hq - 2 : TN + hq - 2;

  // FAST-path per-thread constants: clamped source offsets and the zero-padding predicates
  // (stage-invariant because Cin % KC == 0).
  long long a_off[A_REGS];
  long long bv_off[BV_REGS];
  int bv_ci[BV_REGS];
  bool bv_ok[BV_REGS];
#pragma unroll
  for (int i = 0; i < A_REGS; ++i) {
    const int f = tid + 256 * i;
    const int m = min(m0 + (f % (TM / 4)) * 4, Cout - 4), row = min(f / (TM / 4), TAPS * KC - 1);
    a_off[i] = ((long long)(row / KC) * in.Cin + row % KC) * Cout + m;  // rows >= Cout are never stored
  }
#pragma unroll
  for (int i = 0; i < BV_REGS; ++i) {
    const int f = tid + 256 * i;
    const int x = x0 + (f % (TN / 4)) * 4, row = min(f / (TN / 4), TAPS * KC - 1);
    const int yy = y + row % TAPS - 2;
    bv_ci[i] = row / TAPS;
    bv_ok[i] = yy >= 0 && yy < H && x < W;
    bv_off[i] = (long long)min(max(yy, 0), H - 1) * W + min(x, W - 4);
  }
  const long long i_off = (long long)y * W + min(x0 + idx, W - 4);
  const long long h_off = (long long)y * W + min(max(x0 + hdx, 0), W - 1);
  const bool i_ok = x0 + idx < W, h_ok = x0 + hdx >= 0 && x0 + hdx < W;

  auto load_stage = [&](f32x4 (&ra)[A_REGS], f32x4 (&rb)[BV_REGS], float& halo, int c0) {
    if (FAST) {
#pragma unroll
      for (int i = 0; i < A_REGS; ++i)
        ra[i] = *reinterpret_cast<const f32x4*>(wp + a_off[i] + (long long)c0 * Cout);
      if (VERT) {
#pragma unroll
        for (int i = 0; i < BV_REGS; ++i)
          rb[i] = *reinterpret_cast<const f32x4*>(channel_plane(in, c0 + bv_ci[i], plane) + bv_off[i]);
      } else {
        rb[0] = *reinterpret_cast<const f32x4*>(channel_plane(in, c0 + ici, plane) + i_off);
        halo = channel_plane(in, c0 + hci, plane)[h_off];
      }
    } else {
#pragma unroll
    for (int i = 0; i < A_REGS; ++i) {
      const int f = tid + 256 * i;
      const int m = m0 + (f % (TM / 4)) * 4, row = f / (TM / 4);
      const int t = row / KC, ci = c0 + row % KC;
      const bool ok = f < A_VEC && ci < in.Cin;
      ra[i] = load4(wp + ((long long)t * in.Cin + ci) * Cout + m, ok ? Cout - m : 0, vec_w);
    }
    if (VERT) {
#pragma unroll
      for (int i = 0; i < BV_REGS; ++i) {
        const int f = tid + 256 * i;
        const int x = x0 + (f % (TN / 4)) * 4, row = f / (TN / 4);
        const int ci = c0 + row / TAPS, yy = y + row % TAPS - 2;
        const bool ok = f < BV_VEC && ci < in.Cin && yy >= 0 && yy < H;
        const int cic = min(ci, in.Cin - 1), yc = min(max(yy, 0), H - 1);
        rb[i] = load4(channel_plane(in, cic, plane) + (long long)yc * W + x, ok ? W - x : 0, vec_x);
      }
    } else {
      const int ci = c0 + ici, x = x0 + idx;
      rb[0] = load4(channel_plane(in, min(ci, in.Cin - 1), plane) + (long long)y * W + x,
                     ci < in.Cin ? W - x : 0, vec_x);
      const int cih = c0 + hci, xh = x0 + hdx;
      halo = 0.f;
      if (cih < in.Cin && xh >= 0 && xh < W) halo = channel_plane(in, cih, plane)[(long long)y * W + xh];
    }
    }
  };

  auto store_stage = [&](const f32x4 (&ra)[A_REGS], const f32x4 (&rb)[BV_REGS], const float& halo, int buf) {
#pragma unroll
    for (int i = 0; i < A_REGS; ++i) {
      const int f = tid + 256 * i;
      if (f < A_VEC) *reinterpret_cast<f32x4*>(&sA[buf][f * 4]) = ra[i];
    }
    if (VERT) {
#pragma unroll
      for (int i = 0; i < BV_REGS; ++i) {
        const int f = tid + 256 * i;
        if (f < BV_VEC)
          *reinterpret_cast<f32x4*>(&sB[buf][f * 4]) = FAST ? keep_if(bv_ok[i], rb[i]) : rb[i];
      }
    } else {
      if (tid < 128)
        *reinterpret_cast<f32x4*>(&sB[buf][ici * BH_ROW + BH_X0 + idx]) = FAST ? keep_if(i_ok, rb[0]) : rb[0];
      if (tid < 32) sB[buf][hci * BH_ROW + BH_X0 + hdx] = (!FAST || h_ok) ? halo : 0.f;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // Operands of the fused GRU epilogue, requested before the K loop: read after it (16 x 3..5 dependent loads per
  // lane, one wave per SIMD) they sat on the tail of every workgroup and cost more than the launches they replace.
  float e0[16], e1[16], e2[16], e3[16], e4[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) e0[r] = e1[r] = e2[r] = e3[r] = e4[r] = 0.f;
  {
    const int emw = m0 + wr * 32;
    const bool apart = emw < epi.C;   // modes 3 / 4: this wave's channels are the a-part
    if (epi.mode != 0 && kgrp == 0 && (epi.mode < 3 || apart)) {
      const int ex = min(x0 + wc * 32 + l31, W - 1);
      const long long epix = (long long)y * W + ex;
      const bool is_z = epi.mode == 1 && apart;
      // index of channel emw in a [B][C] tensor (r-half of mode 1: channel emw - C) and in mode 1's [B][2C] addend
      const long long ic0 = ((long long)blockIdx.z * epi.C + (epi.mode == 1 && !apart ? emw - epi.C : emw)) * plane + epix;
      const long long ia0 = epi.mode == 1 ? ((long long)blockIdx.z * 2 * epi.C + emw) * plane + epix : ic0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (emw + ml < Cout) {
          const long long ic = ic0 + ml * plane;
          e0[r] = epi.p0[epi.mode <= 2 ? ia0 + ml * plane : ic];
          if (!is_z) e1[r] = epi.p1[ic];
          if (epi.mode >= 2) e2[r] = epi.p2[ic];
          if (epi.mode >= 3) e3[r] = epi.p3[ic];
          if (epi.mode == 3 && epi.p4 != nullptr) e4[r] = epi.p4[ic];
        }
      }
    }
  }

  // The operands of MFMA i + PRE are read from LDS before MFMA i is issued, and the scheduler is fenced per MFMA:
  // left alone it reused one register pair for all reads, so every pair of MFMAs waited for its own LDS round trip
  // (read -> wait -> 2 MFMAs -> read ...: ~65 idle cycles per pair at one or two waves per SIMD).
  auto compute_stage = [&](int buf, auto&& mid1, auto&& mid2) {
    const float* ap = sA[buf] + lh * A_STRIDE + wr * 32 + l31;
    const float* bp = sB[buf] + (VERT ? lh * TAPS * BV_ROW : lh * BH_ROW + BH_X0 - 2) + wc * 32 + l31;
    constexpr int NM = TAPS * KC / 2, PRE = PCFA_SC5_PRE;
    float av[NM], bv[NM];
    auto rd = [&](int i) {
      const int t = i / (KC / 2), s = 2 * (i % (KC / 2));
      av[i] = ap[(t * KC + s) * A_STRIDE];
      bv[i] = VERT ? bp[s * TAPS * BV_ROW + t * BV_ROW] : bp[s * BH_ROW + t];
    };
#pragma unroll
    for (int i = 0; i < PRE; ++i) rd(i);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (i + PRE < NM) rd(i + PRE);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
      if (i == 1) mid1();          // global loads of a later stage / LDS write of the next one: in the MFMA shadow
      if (i == NM / 2) mid2();
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nstage = (in.Cin + KC - 1) / KC / KS;   // stages of this wave group (host: Cin % (KC * NR * KS) == 0)
  const int cbase = kgrp * nstage * KC;             // its first input channel
  f32x4 ring_a[NR][A_REGS], ring_b[NR][BV_REGS];
  float ring_h[NR];
  if (FAST) {
    // nstage % NR == 0: the unrolled body has no branch at all.  Stages past the end re-load the last one
    // (clamped) and store it to the idle LDS buffer; nothing reads it.
    const int last = cbase + (nstage - 1) * KC;
#pragma unroll
    for (int j = 0; j < NR - 1; ++j) load_stage(ring_a[j], ring_b[j], ring_h[j], min(cbase + j * KC, last));
    store_stage(ring_a[0], ring_b[0], ring_h[0], 0);
    __syncthreads();
    for (int s0 = 0; s0 < nstage; s0 += NR) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        compute_stage(j & 1,
                      [&] { load_stage(ring_a[(j + NR - 1) % NR], ring_b[(j + NR - 1) % NR], ring_h[(j + NR - 1) % NR],
                                       min(cbase + (s0 + j + NR - 1) * KC, last)); },
                      [&] { store_stage(ring_a[(j + 1) % NR], ring_b[(j + 1) % NR], ring_h[(j + 1) % NR], (j + 1) & 1); });
        __syncthreads();
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NR - 1; ++j)
      if (j < nstage) load_stage(ring_a[j], ring_b[j], ring_h[j], j * KC);
    store_stage(ring_a[0], ring_b[0], ring_h[0], 0);
    __syncthreads();
    for (int s0 = 0; s0 < nstage; s0 += NR) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {  // NR is even: LDS buffer of stage s0+j is j & 1
        const int st = s0 + j;
        if (st < nstage) {
          if (st + NR - 1 < nstage)
            load_stage(ring_a[(j + NR - 1) % NR], ring_b[(j + NR - 1) % NR], ring_h[(j + NR - 1) % NR],
                       (st + NR - 1) * KC);
          compute_stage(j & 1, [] {}, [] {});
          if (st + 1 < nstage)
            store_stage(ring_a[(j + 1) % NR], ring_b[(j + 1) % NR], ring_h[(j + 1) % NR], (j + 1) & 1);
          __syncthreads();
        }
      }
    }
  }

  if (KS == 2) {   // second half of K -> LDS -> added by the first group (fixed order: deterministic)
    float* red = &sA_[0][0][0];                     // 4 waves x 16 x 64 floats = 16 KB <= the first group's A stages
    if (kgrp == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kgrp == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[(wave * 16 + r) * 64 + lane];
  }

  // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const int x = x0 + wc * 32 + l31;
  if (x < W) {
    const long long pix = (long long)y * W + x;
    const int mw = m0 + wr * 32;  // first output channel of this wave's 32x32 tile
    if (epi.mode == 1) {
      const bool is_z = mw < epi.C;
      const long long io0 = ((long long)blockIdx.z * epi.C + (is_z ? mw : mw - epi.C)) * plane + pix;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (mw + ml < Cout) {
          const float sg = sc5_sigmoid(acc[r] + e0[r]);
          const long long io = io0 + ml * plane;
          if (is_z) {
            epi.o0[io] = sg;
          } else {
            epi.o1[io] = sg;
            epi.o2[io] = sg * e1[r];
          }
        }
      }
    } else if (epi.mode == 2) {
      const long long i0 = ((long long)blockIdx.z * epi.C + mw) * plane + pix;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (mw + ml < Cout) {
          const long long i = i0 + ml * plane;
          const float qq = tanhf(acc[r] + e0[r]);
          const float zz = e2[r], hh = e1[r];
          epi.o0[i] = qq;
          epi.o1[i] = (1.f - zz) * hh + zz * qq;
        }
      }
    } else if (epi.mode == 3 && mw < epi.C) {   // as gru_gates_bwd_kernel (gru_math.hip), drh = acc
      const long long i0 = ((long long)blockIdx.z * epi.C + mw) * plane + pix;
      const long long j0 = ((long long)blockIdx.z * 2 * epi.C + mw) * plane + pix;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (mw + ml < epi.C) {
          const float z_ = e0[r], r_ = e1[r], h_ = e2[r], dz_ = e3[r], drh_ = acc[r];
          const float a = dz_ * (1.f - z_) * z_;
          const float dr = drh_ * h_;
          const float b = dr * (1.f - r_) * r_;
          float c = drh_ * r_;
          if (epi.p4 != nullptr) c += e4[r];
          epi.o0[j0 + ml * plane] = a;
          epi.o1[j0 + (long long)(epi.C + ml) * plane] = b;
          epi.o2[i0 + ml * plane] = c;
        }
      }
    } else if (epi.mode == 4 && mw < epi.C) {   // dh accumulate, then gru_update_bwd_kernel of the previous half-step
      const long long i0 = ((long long)blockIdx.z * epi.C + mw) * plane + pix;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (mw + ml < epi.C) {
          const float g_ = e0[r] + acc[r];
          const float z_ = e1[r], q_ = e2[r], h_ = e3[r];
          const long long i = i0 + ml * plane;
          epi.o0[i] = g_ * q_ - g_ * h_;
          epi.o1[i] = (g_ * z_) * (1.f - q_ * q_);
          epi.o2[i] = g_ * (1.f - z_);
        }
      }
    } else if ((out.Ca & 31) == 0 || out.b == nullptr) {
      // the wave's 32 channels lie on one side of the split: destination and accumulate flag are wave-uniform
      const bool first = out.b == nullptr || mw < out.Ca;
      float* base = (first ? out.a + (long long)mw * plane : out.b + (long long)(mw - out.Ca) * plane) + pix;
      if (!first && out.mask_b != nullptr) {   // last write of the b part: apply the deferred ReLU mask
        const float* mk = out.mask_b + (long long)(mw - out.Ca) * plane + pix;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (mw + ml < Cout) {
            const float v = out.acc_b ? base[ml * plane] + acc[r] : acc[r];
            base[ml * plane] = (mw - out.Ca + ml >= out.mask_cb || mk[ml * plane] > 0.f) ? v : 0.f;
          }
        }
      } else if (first ? out.acc_a : out.acc_b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (mw + ml < Cout) base[ml * plane] += acc[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (mw + ml < Cout) base[ml * plane] = acc[r];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < Cout) {
          const bool first = m < out.Ca;
          float* o = (first ? out.a + (long long)m * plane : out.b + (long long)(m - out.Ca) * plane) + pix;
          float v = (first ? out.acc_a : out.acc_b) ? *o + acc[r] : acc[r];
          if (!first && out.mask_b != nullptr && m - out.Ca < out.mask_cb &&
              !(out.mask_b[(long long)(m - out.Ca) * plane + pix] > 0.f))
            v = 0.f;
          *o = v;
        }
      }
    }
  }
}

// w [Cout][Cin][5] (a Conv2d (1,5) or (5,1) weight, flattened) ->
//   fwd[t][ci][co] = w[co][ci][t]          (forward operator)
//   bwd[t][co][ci] = w[co][ci][4 - t]      (data gradient = the same operator on grad_out)
__global__ void sepconv5_pack_kernel(const float* __restrict__ w, float* __restrict__ fwd,
                                     float* __restrict__ bwd, int Cout, int Cin) {
  const long long n = (long long)Cout * Cin * TAPS;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % TAPS);
    const int ci = (int)((i / TAPS) % Cin);
    const int co = (int)(i / ((long long)TAPS * Cin));
    const float v = w[i];
    if (fwd) fwd[((long long)t * Cin + ci) * Cout + co] = v;
    if (bwd) bwd[((long long)(TAPS - 1 - t) * Cout + co) * Cin + ci] = v;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pcfa_sepconv5_algo(int use_winograd) { return sc5_wino_enabled(use_winograd); }

extern "C" int pcfa_sepconv5_uses_winograd(int B, int Ca, int Cb, int Cout, int H, int W, int vertical) {
  return sc5_wino_shape_ok(B, Ca + Cb, Ca, Cout, H, W, vertical) ? 1 : 0;
}

extern "C" long long pcfa_sepconv5_packed_floats(int Cout, int Cin) {
  if (Cout < 1 || Cin < 1) return 0;
  return (long long)Cout * Cin * TAPS + sc5_wino_packed_floats(Cout, Cin);
}

extern "C" int pcfa_sepconv5_pack_weights(const float* w, float* fwd_packed, float* bwd_packed, int Cout,
                                          int Cin, void* stream) {
  if (!w || (!fwd_packed && !bwd_packed) || Cout < 1 || Cin < 1) return PCFA_ERR_INVALID_ARG;
  const long long n = (long long)Cout * Cin * TAPS;
  pcfa_launch(sepconv5_pack_kernel, dim3((unsigned)min((n + 255) / 256, (long long)1024)), dim3(256), 0,
              (hipStream_t)stream, w, fwd_packed, bwd_packed, Cout, Cin);
  PCFA_LAUNCH_CHECK();
  // the Winograd-domain weights of sepconv5_wino.hip follow the direct packing (nothing when the shape is not eligible)
  if (fwd_packed) {
    const int e = sc5_wino_pack(w, fwd_packed + n, Cout, Cin, 0, (hipStream_t)stream);
    if (e != PCFA_OK) return e;
  }
  if (bwd_packed) {
    const int e = sc5_wino_pack(w, bwd_packed + n, Cin, Cout, 1, (hipStream_t)stream);
    if (e != PCFA_OK) return e;
  }
  return PCFA_OK;
}

static int sepconv5_launch(const float* in_a, int Ca, const float* in_b, int Cb, const float* w_packed,
                           OutSplit out, int B, int Cout, int H, int W, int vertical, void* stream,
                           GruEpi epi = GruEpi{0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});

extern "C" int pcfa_sepconv5_fwd(const float* in_a, int Ca, const float* in_b, int Cb,
                                 const float* w_packed, float* out, int B, int Cout, int H, int W,
                                 int vertical, void* stream) {
  if (!out) return PCFA_ERR_INVALID_ARG;
  return sepconv5_launch(in_a, Ca, in_b, Cb, w_packed, OutSplit{out, nullptr, Cout, 0, 0, nullptr, 0}, B, Cout, H, W, vertical,
                         stream);
}

extern "C" int pcfa_sepconv5_fwd_split(const float* in_a, int Ca, const float* in_b, int Cb, const float* w_packed,
                                       float* out_a, int Cout_a, int accumulate_a, float* out_b, int accumulate_b,
                                       int B, int Cout, int H, int W, int vertical, void* stream) {
  if (!out_a || Cout_a < 1 || Cout_a > Cout || (Cout_a < Cout && !out_b)) return PCFA_ERR_INVALID_ARG;
  return sepconv5_launch(in_a, Ca, in_b, Cb, w_packed,
                         OutSplit{out_a, Cout_a < Cout ? out_b : nullptr, Cout_a, accumulate_a != 0, accumulate_b != 0,
                                  nullptr, 0},
                         B, Cout, H, W, vertical, stream);
}

extern "C" int pcfa_sepconv5_fwd_split_masked(const float* in_a, int Ca, const float* in_b, int Cb,
                                              const float* w_packed, float* out_a, int Cout_a, int accumulate_a,
                                              float* out_b, int accumulate_b, const float* mask_b, int mask_channels,
                                              int B, int Cout, int H, int W, int vertical, void* stream) {
  if (!out_a || Cout_a < 1 || Cout_a >= Cout || !out_b || !mask_b || mask_channels < 0 ||
      mask_channels > Cout - Cout_a)
    return PCFA_ERR_INVALID_ARG;
  return sepconv5_launch(in_a, Ca, in_b, Cb, w_packed,
                         OutSplit{out_a, out_b, Cout_a, accumulate_a != 0, accumulate_b != 0, mask_b, mask_channels},
                         B, Cout, H, W, vertical, stream);
}

extern "C" int pcfa_sepconv5_gru_gates_fwd(const float* h, int C, const float* rest, int Cr, const float* w_packed,
                                           const float* add_zr, float* z, float* r, float* rh, int B, int H, int W,
                                           int vertical, void* stream) {
  if (!h || !add_zr || !z || !r || !rh || C < 32 || C % 32 != 0) return PCFA_ERR_INVALID_ARG;
  return sepconv5_launch(h, C, rest, Cr, w_packed, OutSplit{z, nullptr, 2 * C, 0, 0, nullptr, 0}, B, 2 * C, H, W, vertical,
                         stream, GruEpi{1, C, add_zr, h, nullptr, nullptr, nullptr, z, r, rh});
}

extern "C" int pcfa_sepconv5_gru_update_fwd(const float* rh, int C, const float* rest, int Cr, const float* w_packed,
                                            const float* add_q, const float* z, const float* h, float* q, float* hnew,
                                            int B, int H, int W, int vertical, void* stream) {
  if (!rh || !add_q || !z || !h || !q || !hnew || C < 32 || C % 32 != 0) return PCFA_ERR_INVALID_ARG;
  return sepconv5_launch(rh, C, rest, Cr, w_packed, OutSplit{q, nullptr, C, 0, 0, nullptr, 0}, B, C, H, W, vertical,
                         stream, GruEpi{2, C, add_q, h, z, nullptr, nullptr, q, hnew, nullptr});
}

extern "C" int pcfa_sepconv5_gru_gates_bwd(const float* dqc, int C, int Cr, const float* w_packed_bwd, const float* z,
                                           const float* r, const float* h, const float* dz, const float* dh_in,
                                           float* dzr, float* dh, float* d_rest, int accumulate_rest, int B, int H, int W,
                                           int vertical, void* stream) {
  if (!dqc || !z || !r || !h || !dz || !dzr || !dh || !d_rest || C < 32 || C % 32 != 0 || Cr < 1)
    return PCFA_ERR_INVALID_ARG;
  return sepconv5_launch(dqc, C, nullptr, 0, w_packed_bwd, OutSplit{dh, d_rest, C, 0, accumulate_rest != 0, nullptr, 0}, B,
                         C + Cr, H, W, vertical, stream, GruEpi{3, C, z, r, h, dz, dh_in, dzr, dzr, dh});
}

extern "C" int pcfa_sepconv5_gru_update_bwd(const float* dzr, int C, int Cr, const float* w_packed_bwd, const float* dh_acc,
                                            const float* z_prev, const float* q_prev, const float* h_prev, float* dz_prev,
                                            float* dqc_prev, float* dh_prev, float* d_rest, int B, int H, int W,
                                            int vertical, void* stream) {
  if (!dzr || !dh_acc || !z_prev || !q_prev || !h_prev || !dz_prev || !dqc_prev || !dh_prev || !d_rest || C < 32 ||
      C % 32 != 0 || Cr < 1)
    return PCFA_ERR_INVALID_ARG;
  return sepconv5_launch(dzr, 2 * C, nullptr, 0, w_packed_bwd, OutSplit{dh_prev, d_rest, C, 0, 1, nullptr, 0}, B, C + Cr, H,
                         W, vertical, stream, GruEpi{4, C, dh_acc, z_prev, q_prev, h_prev, nullptr, dz_prev, dqc_prev, dh_prev});
}

static int sepconv5_launch(const float* in_a, int Ca, const float* in_b, int Cb, const float* w_packed,
                           OutSplit out, int B, int Cout, int H, int W, int vertical, void* stream, GruEpi epi) {
  if (!in_a || !w_packed || Ca < 1 || Cb < 0 || (Cb > 0 && !in_b) || B < 1 || Cout < 1 ||
      H < 1 || W < 1)
    return PCFA_ERR_INVALID_ARG;
  const int tiles_x = pcfa_cdiv(W, TN);
  const long long gx = (long long)tiles_x * H;
  if (gx > 0x7fffffffLL || B > 65535 || pcfa_cdiv(Cout, TM) > 65535) return PCFA_ERR_UNSUPPORTED;
  Operand in{in_a, Cb > 0 ? in_b : nullptr, Ca, Ca + Cb};
  {   // 1-D Winograd F(2,5) where the shape allows it (sepconv5_wino.hip); its weights sit behind the direct packing
    const float* w_wino = sc5_wino_packed_floats(Cout, Ca + Cb) > 0 ? w_packed + (long long)TAPS * (Ca + Cb) * Cout : nullptr;
    const int e = sc5_wino_launch(in, w_wino, out, B, Cout, H, W, vertical, (hipStream_t)stream, epi);
    if (e != PCFA_SC5_NOT_ELIGIBLE) return e;
  }
  const int vec_w = (Cout % 4 == 0) && aligned16(w_packed);
  const int vec_x = (W % 4 == 0) && aligned16(in_a) && (Cb == 0 || aligned16(in_b));
  dim3 grid((unsigned)gx, pcfa_cdiv(Cout, TM), B), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool fast = vec_w && vec_x && (Ca + Cb) % (KC * NR) == 0 && Cout >= 4 && W >= 4;
  // fewer workgroups than CUs (x 1.25): split K inside the workgroup (8 waves per tile)
  static const int ks_env = getenv("PCFA_SEPCONV_KS") ? atoi(getenv("PCFA_SEPCONV_KS")) : 0;   // tuning override
  const long long nwg = gx * pcfa_cdiv(Cout, TM) * B;
  const bool split2 = fast && (Ca + Cb) % (KC * NR * 2) == 0 && (ks_env ? ks_env == 2 : nwg <= 320);
  if (split2) {
    dim3 block2(512);
    if (vertical) pcfa_launch(sepconv5_kernel<true, true, 2>, grid, block2, 0, s, in, w_packed, out, Cout, H, W, tiles_x, vec_w, vec_x, epi);
    else pcfa_launch(sepconv5_kernel<false, true, 2>, grid, block2, 0, s, in, w_packed, out, Cout, H, W, tiles_x, vec_w, vec_x, epi);
    PCFA_LAUNCH_CHECK();
    return PCFA_OK;
  }
#define PCFA_SEPCONV5(V, F) \
  pcfa_launch(sepconv5_kernel<V, F>, grid, block, 0, s, in, w_packed, out, Cout, H, W, tiles_x, vec_w, vec_x, epi)
  if (vertical) {
    if (fast) PCFA_SEPCONV5(true, true); else PCFA_SEPCONV5(true, false);
  } else {
    if (fast) PCFA_SEPCONV5(false, true); else PCFA_SEPCONV5(false, false);
  }
#undef PCFA_SEPCONV5
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
