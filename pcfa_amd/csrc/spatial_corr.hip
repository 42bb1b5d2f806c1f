// PWC-Net cost volume: spatial correlation sampler forward/backward for gfx950.
//
// Replaces spatial_correlation_sampler_backend.forward/backward (reference
// models/PWCNet/cpu_spatial_correlation_sampler-0.3.0/Correlation_Module/
// correlation.cpp:9-37,39-73,75-124,126-178; binding correlation_sampler.cpp:58-112).
//
//   out[b][ph][pw][h][w] = sum_{c,i,j} in1[b][c][u+i*dil][v+j*dil] *
//                                      in2[b][c][u+i*dil+sU][v+j*dil+sV]
//   u = h*dH - padH, v = w*dW - padW, sU = (ph - (patchH-1)/2)*dil_patchH (same for V),
//   terms with either operand outside the image dropped.
//
// Fast path (what PWC-Net uses: k=1, stride 1, pad 0, patch 9x9, dil_patch 1, W % 4 == 0):
//   The five cost volumes of a PWC-Net forward are 0.28 GFLOP on 27 MB -- each launch is a latency chain, not a
//   stream, and the coarse levels (6x20 ... 24x80 pixels, 96-196 channels) have almost no pixels to spread over
//   256 CUs.  Both directions are therefore built for parallelism inside the workgroup and ONE round trip to
//   memory: every load of a workgroup is issued before its first LDS write ("single shot": all channels of the
//   tile, or as many as fit 144 KB), one barrier, compute from LDS, store.
//   forward : workgroup = TH x TW output pixels (4x32, 2x32 or 2x16: the smallest levels take the smallest tile)
//             x NS channel slices.  A thread owns 4 consecutive pixels x 9 horizontal shifts of one patch row
//             and walks the channels of its slice (1 + 3 aligned ds_read_b128 per 36 FMAs); the NS partial sums
//             meet in LDS and are added in slice order, so the result does not depend on timing.  An optional
//             epilogue (scale, LeakyReLU) serves PWC-Net's `leaky_relu(corr / C)` in the same pass.
//   backward: ONE launch for both gradients (blockIdx.z selects which).  With e = d + 4 in [0,9)^2 both are the
//             same gather   gin[c][p] = sum_e G[e][p] * X[c][p + e - 4]   with
//               gin1: X = in2, G[e][p] = g[e][p]
//               gin2: X = in1, G[e][p] = g[8 - e][p + e - 4]   (the same taps, seen from the other image)
//             so only the staging of G differs.  Workgroup = 2x32 pixels x 32 channels: the 81 x 64 gradient taps
//             and the 10x40 halo of 32 channels go to LDS once; a thread owns 4 pixels x 2 channels, reads one
//             row of taps (9 x b128) per patch row and three aligned b128 of X per channel.  No atomics: bitwise
//             reproducible (the CPU reference accumulates serially, correlation.cpp:148).  LeakyReLU's mask and the
//             1/C of PWC-Net's correlate() can be applied to the taps while they are staged.
// Every other parameter set takes the generic one-thread-per-element kernels.
#include <cstdlib>
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SC_LDS_BUDGET = 144 * 1024;  // bytes of dynamic LDS a forward workgroup may stage into

// XCD-aware workgroup -> tile map.  Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has its
// own L2: with the plain (x, y) grid the tiles that share a halo (4 of the 12 x 40 staged in2 rows / columns belong to the
// neighbours) sat on different XCDs, so every L2 fetched its own copy -- rocprofv3 counted 30.1 MB of L2 misses per launch for
// the 7.9 MB of level-2 inputs (forward) and 137.6 MB for the backward's 25.7 MB (profiles/r05/pwc_traffic_before_xcd_map.json).
// Here XCD k owns the contiguous band of tiles [k * chunk, (k + 1) * chunk) (row-major: x neighbours adjacent, rows stacked),
// the launch is a 1-D grid of 8 * chunk * nz workgroups, and a workgroup whose tile falls past the end leaves at once.
struct XcdTile {
  int tile, z;
  bool live;
};
__device__ __forceinline__ XcdTile xcd_tile(int ntiles, int nz) {
  const int chunk = (ntiles + 7) >> 3;
  const int lin = blockIdx.x, k = lin & 7, r = lin >> 3;   // r-th workgroup of XCD k
  XcdTile t;
  t.z = r % nz;                                           // z fastest: the workgroups of one tile (gradient side, channel
  t.tile = k * chunk + r / nz;                            // group) follow each other on one XCD and share its taps
  t.live = t.tile < ntiles;
  return t;
}
inline unsigned xcd_grid(int ntiles, int nz) { return 8u * (unsigned)((ntiles + 7) / 8) * (unsigned)nz; }

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int TH, int TW>
__global__ __launch_bounds__(576) void scorr9_fwd_kernel(const float* __restrict__ in1,
                                                         const float* __restrict__ in2,
                                                         float* __restrict__ out, int C, int H, int W, int NS,
                                                         int CR, float scale, float slope, int dbg, int tiles_x,
                                                         int ntiles, int B) {
  constexpr int PS = 9, R = 4;
  constexpr int QW = TW / 4;                 // pixel quads per tile row
  constexpr int HW2 = TW + 2 * R, HH2 = TH + 2 * R;
  constexpr int P1 = TH * QW;                // 16-B pieces of the in1 tile (per channel)
  constexpr int P2 = HH2 * (HW2 / 4);        // ... of the in2 tile with its halo
  constexpr int PPC = P1 + P2;               // pieces per channel
  constexpr int PCF = 4 * PPC;               // floats per channel in LDS: [in1 tile | in2 halo tile]
  constexpr int TPS = QW * TH * PS;          // threads per channel slice
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const XcdTile xt = xcd_tile(ntiles, B);
  if (!xt.live) return;   // (whole workgroup, before any barrier)
  const int b = xt.z;
  const int y0 = (xt.tile / tiles_x) * TH, x0 = (xt.tile % tiles_x) * TW;
  const int tid = threadIdx.x, NT = blockDim.x;
  const size_t plane = (size_t)H * W;
  const float* p1 = in1 + (size_t)b * C * plane;
  const float* p2 = in2 + (size_t)b * C * plane;

  const int sl = tid / TPS, t = tid - sl * TPS;
  const int q = t % QW, r = (t / QW) % TH, ph = t / (QW * TH);

  float acc[4][PS];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int d = 0; d < PS; ++d) acc[p][d] = 0.f;

  // Staging plan, fixed per thread: thread = (channel lane cl, piece pp of the per-channel pattern), so a piece's
  // row / column / validity are decoded ONCE and walking the channels is a pointer increment (the first version
  // decoded every piece from a flat index: ~50 integer instructions per 16-B piece, and the kernel was bound by
  // that arithmetic, not by memory -- 7-19 us with loads, FMAs and stores all switched off).
  const int lanes = NT / PPC;                // channels staged per pass (NT >= PPC for every tile shape)
  const int cl = tid / PPC, pp = tid - cl * PPC;
  bool pvalid;
  const float* pbase;
  {
    int gy, gx;
    if (pp < P1) {
      gy = y0 + pp / QW; gx = x0 + 4 * (pp % QW); pbase = p1;
    } else {
      const int m = pp - P1;
      gy = y0 + m / (HW2 / 4) - R; gx = x0 + 4 * (m % (HW2 / 4)) - R; pbase = p2;
    }
    pvalid = cl < lanes && gy >= 0 && gy < H && gx >= 0 && gx < W && !(dbg & 1);
    if (pvalid) pbase += (size_t)gy * W + gx;
  }
  const size_t lstep = (size_t)lanes * plane;

  for (int c0 = 0; c0 < C; c0 += CR) {
    const int cr = min(CR, C - c0);
    if (c0 > 0) __syncthreads();             // the previous round's readers are done
    // ---- stage cr channels: every piece is 16 B, fully inside or fully outside the image ----
    if (cl < lanes) {
      constexpr int U = 12;
      const float* gp = pbase + (size_t)(c0 + cl) * plane;
      float* lp = lds + cl * PCF + 4 * pp;
#pragma unroll 1
      for (int cb = cl; cb < cr; cb += lanes * U) {
        f32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const bool ok = pvalid && cb + k * lanes < cr;
          v[k] = *reinterpret_cast<const f32x4*>(ok ? gp : pbase);   // branch-free: a dead piece re-reads a live address
          if (!ok) v[k] = (f32x4)(0.f);
          gp += lstep;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          if (cb + k * lanes < cr) *reinterpret_cast<f32x4*>(lp) = v[k];
          lp += lanes * PCF;
        }
      }
    }
    __syncthreads();
    // ---- this slice's channels of the round ----
    if (sl < NS && !(dbg & 2)) {
#pragma unroll 2
      for (int c = sl; c < cr; c += NS) {
        const float* ch = lds + c * PCF;
        const f32x4 a = *reinterpret_cast<const f32x4*>(ch + (r * QW + q) * 4);
        const float* row = ch + 4 * P1 + (r + ph) * HW2 + 4 * q;
        float v2[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const f32x4 tt = *reinterpret_cast<const f32x4*>(row + 4 * k);
          v2[4 * k] = tt.x; v2[4 * k + 1] = tt.y; v2[4 * k + 2] = tt.z; v2[4 * k + 3] = tt.w;
        }
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int d = 0; d < PS; ++d) acc[p][d] += av[p] * v2[p + d];
      }
    }
  }

  // ---- the NS partial sums meet in LDS and are added in slice order ----
  constexpr int NO = PS * PS * TH * QW;      // float4 outputs of the tile
  float* ob = out + (size_t)b * PS * PS * plane;
  if (NS == 1) {
    const int gy = y0 + r, gx = x0 + 4 * q;
    if (tid < TPS && gy < H && gx < W && (!(dbg & 4) || acc[0][0] == 123.f)) {
#pragma unroll
      for (int d = 0; d < PS; ++d) {
        f32x4 v = {acc[0][d], acc[1][d], acc[2][d], acc[3][d]};
        v *= scale;
        if (slope != 1.f) {
          v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
          v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
        }
        *reinterpret_cast<f32x4*>(ob + (size_t)(ph * PS + d) * plane + (size_t)gy * W + gx) = v;
      }
    }
    return;
  }
  __syncthreads();
  if (sl < NS) {
#pragma unroll
    for (int d = 0; d < PS; ++d) {
      const int o = ((ph * PS + d) * TH + r) * QW + q;
      const f32x4 v = {acc[0][d], acc[1][d], acc[2][d], acc[3][d]};
      *reinterpret_cast<f32x4*>(lds + 4 * (sl * NO + o)) = v;
    }
  }
  __syncthreads();
  for (int o = tid; o < NO; o += NT) {
    f32x4 v = *reinterpret_cast<const f32x4*>(lds + 4 * o);
    for (int s2 = 1; s2 < NS; ++s2) v += *reinterpret_cast<const f32x4*>(lds + 4 * (s2 * NO + o));
    v *= scale;
    if (slope != 1.f) {
      v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
    }
    const int d = o / (TH * QW), rem = o - d * (TH * QW);
    const int gy = y0 + rem / QW, gx = x0 + 4 * (rem % QW);
    if (gy < H && gx < W && (!(dbg & 4) || v.x == 123.f))
      *reinterpret_cast<f32x4*>(ob + (size_t)d * plane + (size_t)gy * W + gx) = v;
  }
}

struct FwdPlan {
  int th, tw, ns, cr;
  size_t lds;
};

// Tile and slice count for one level: the largest tile that still gives the chip >= 200 workgroups, otherwise the
// smallest one; as many channel slices as 576 threads hold (each slice keeps >= 4 channels); rounds of equal size.
FwdPlan plan_fwd(int B, int C, int H, int W) {
  const int th[3] = {4, 2, 2}, tw[3] = {32, 32, 16}, nsmax[3] = {2, 4, 8};
  int k = 2;
  for (int i = 0; i < 3; ++i)
    if ((long long)pcfa_cdiv(H, th[i]) * pcfa_cdiv(W, tw[i]) * B >= 200) { k = i; break; }
  FwdPlan P;
  // tuning overrides (tools/dev only), read ONCE per process: plan_fwd runs on every launch
  static const int env_tile = getenv("PCFA_SC_TILE") ? atoi(getenv("PCFA_SC_TILE")) : -1;
  static const int env_ns = getenv("PCFA_SC_NS") ? atoi(getenv("PCFA_SC_NS")) : -1;
  static const int env_cr = getenv("PCFA_SC_CR") ? atoi(getenv("PCFA_SC_CR")) : -1;
  if (env_tile >= 0 && env_tile < 3) k = env_tile;
  P.th = th[k]; P.tw = tw[k];
  P.ns = nsmax[k];
  while (P.ns > 1 && C / P.ns < 4) P.ns >>= 1;
  if (env_ns > 0) P.ns = env_ns < nsmax[k] ? env_ns : nsmax[k];
  const int pcf = P.th * P.tw + (P.th + 8) * (P.tw + 8);
  int crmax = SC_LDS_BUDGET / (pcf * 4);
  if (env_cr > 0) crmax = env_cr < crmax ? env_cr : crmax;
  const int rounds = pcfa_cdiv(C, crmax);
  P.cr = pcfa_cdiv(pcfa_cdiv(C, rounds), P.ns) * P.ns;
  if (P.cr > crmax) P.cr = crmax / P.ns * P.ns;
  const size_t stage = (size_t)P.cr * pcf * 4;
  const size_t red = P.ns > 1 ? (size_t)P.ns * 81 * P.th * P.tw * 4 : 0;
  P.lds = stage > red ? stage : red;
  return P;
}

template <int TH, int TW>
int launch_fwd(const FwdPlan& P, const float* in1, const float* in2, float* out, int B, int C, int H, int W,
               float scale, float slope, hipStream_t s) {
  static size_t granted = 0;   // per template instance; the attribute only ever grows
  if (P.lds > granted) {
    hipError_t e = hipFuncSetAttribute((const void*)scorr9_fwd_kernel<TH, TW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds);
    if (e != hipSuccess) return (int)e;
    granted = P.lds;
  }
  const int tps = (TW / 4) * TH * 9;
  const int tiles_x = pcfa_cdiv(W, TW), ntiles = tiles_x * pcfa_cdiv(H, TH);
  dim3 grid(xcd_grid(ntiles, B)), block(tps * P.ns);
  static const int dbg = getenv("PCFA_SC_DBG") ? atoi(getenv("PCFA_SC_DBG")) : 0;   // phase ablation (tools/dev)
  pcfa_launch(scorr9_fwd_kernel<TH, TW>, grid, block, P.lds, s, in1, in2, out, C, H, W, P.ns, P.cr, scale, slope,
              dbg, tiles_x, ntiles, B);
  return PCFA_OK;
}

int scorr9_forward(const float* in1, const float* in2, float* out, int B, int C, int H, int W, float scale,
                   float slope, hipStream_t s) {
  const FwdPlan P = plan_fwd(B, C, H, W);
  if (P.th == 4) return launch_fwd<4, 32>(P, in1, in2, out, B, C, H, W, scale, slope, s);
  if (P.tw == 32) return launch_fwd<2, 32>(P, in1, in2, out, B, C, H, W, scale, slope, s);
  return launch_fwd<2, 16>(P, in1, in2, out, B, C, H, W, scale, slope, s);
}

// ---------------------------------------------------------------------------------------------------------------
// backward (both gradients, one launch)
// ---------------------------------------------------------------------------------------------------------------
constexpr int BTW = 32, BCG = 32;   // pixel tile width, channels per workgroup

// BTH = tile rows (2: 64 pixels, 2 channels per thread; 4: 128 pixels, 4 channels per thread -- half the workgroups,
// the taps staged once per 128 pixels: the fine levels)
template <int BTH>
__global__ __launch_bounds__(256) void scorr9_bwd_kernel(const float* __restrict__ in1,
                                                         const float* __restrict__ in2,
                                                         const float* __restrict__ gout,
                                                         const float* __restrict__ fwd_out,
                                                         float* __restrict__ gin1, float* __restrict__ gin2,
                                                         int C, int H, int W, float gscale, float slope, int dbg,
                                                         int tiles_x, int ntiles, int nz) {
  constexpr int PS = 9, R = 4;
  constexpr int QW = BTW / 4, HW2 = BTW + 2 * R, HH2 = BTH + 2 * R;
  constexpr int NT = 256;
  constexpr int BCPT = BCG * QW * BTH / NT;      // channels per thread
  static_assert(QW * BTH * (BCG / BCPT) == NT, "thread map");
  constexpr int GW = BTW + 2 * R;   // tap rows carry the halo columns: grad_in2 reads them shifted by ex (below)
  __shared__ __attribute__((aligned(16))) float Gs[PS * PS][BTH][GW];
  __shared__ __attribute__((aligned(16))) float Xs[BCG][HH2][HW2];

  const int ngroups = (C + BCG - 1) / BCG;
  const XcdTile xt = xcd_tile(ntiles, nz);
  if (!xt.live) return;   // (whole workgroup, before any barrier)
  int z = xt.z;
  const int which = z & 1; z >>= 1;               // 0: gradient w.r.t. in1, 1: w.r.t. in2
  const int b = z / ngroups, c0 = (z - b * ngroups) * BCG;
  const int y0 = (xt.tile / tiles_x) * BTH, x0 = (xt.tile % tiles_x) * BTW;
  const int tid = threadIdx.x;
  const size_t plane = (size_t)H * W;
  const float* X = (which ? in1 : in2) + (size_t)b * C * plane;
  const float* gb = gout + (size_t)b * PS * PS * plane;
  const float* fb = fwd_out ? fwd_out + (size_t)b * PS * PS * plane : nullptr;
  float* po = (which ? gin2 : gin1) + (size_t)b * C * plane;

  // Staging, decoded once per thread and advanced by pointer increments (see the forward kernel): all loads of the
  // halo tile are in flight before the first LDS write, the gradient taps follow in batches.
  {
    // halo tile of BCG channels: thread = (channel lane, 16-B piece of the 10 x 40 pattern)
    constexpr int XP = HH2 * (HW2 / 4);            // 100 / 120 pieces per channel
    constexpr int XL = NT / XP;                    // 2 channel lanes
    constexpr int XU = BCG / XL;                   // 16 loads per thread
    static_assert(XL * XU == BCG, "halo staging");
    const int xl = tid / XP, xp = tid - xl * XP;
    const int xgy = y0 + xp / (HW2 / 4) - R, xgx = x0 + 4 * (xp % (HW2 / 4)) - R;
    const bool xvalid = xl < XL && xgy >= 0 && xgy < H && xgx >= 0 && xgx < W && !(dbg & 1);
    const float* xb = X + (xvalid ? (size_t)xgy * W + xgx : 0);
    f32x4 v[XU];
#pragma unroll
    for (int k = 0; k < XU; ++k) {
      const int c = c0 + xl + k * XL;
      const bool ok = xvalid && c < C;
      v[k] = *reinterpret_cast<const f32x4*>(ok ? xb + (size_t)c * plane : xb);
      if (!ok) v[k] = (f32x4)(0.f);
    }
    // ---- the 81 x (2 x 32) gradient taps ----
    if (dbg & 8) {
    } else if (which == 0) {
      // thread = (plane lane, 16-B piece of the 2 x 32 tile): 16 planes per pass
      constexpr int GP = BTH * QW, GL = NT / GP, GU = (PS * PS + GL - 1) / GL;   // 16, 16, 6  /  32, 8, 11
      const int gl = tid / GP, gp_ = tid - gl * GP;
      const int ggy = y0 + gp_ / QW, ggx = x0 + 4 * (gp_ % QW);
      const bool gvalid = ggy < H && ggx < W;
      const size_t goff = gvalid ? (size_t)ggy * W + ggx : 0;
      f32x4 g[GU], f[GU];
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int d = gl + k * GL;
        const size_t off = goff + (size_t)(d < PS * PS ? d : 0) * plane;
        g[k] = *reinterpret_cast<const f32x4*>(gb + off);
        f[k] = fb ? *reinterpret_cast<const f32x4*>(fb + off) : (f32x4)(1.f);
      }
#pragma unroll
      for (int k = 0; k < GU; ++k) {
        const int d = gl + k * GL;
        f32x4 t = g[k] * gscale;
        if (fb) {
          t.x = f[k].x > 0.f ? t.x : t.x * slope; t.y = f[k].y > 0.f ? t.y : t.y * slope;
          t.z = f[k].z > 0.f ? t.z : t.z * slope; t.w = f[k].w > 0.f ? t.w : t.w * slope;
        }
        if (d < PS * PS) *reinterpret_cast<f32x4*>(&Gs[d][gp_ / QW][4 * (gp_ % QW)]) = gvalid ? t : (f32x4)(0.f);
      }
    } else {
      // G[e][p] = g[8 - e][p + e - 4]: tap plane e is plane 8 - e of grad_out read at rows r + ey - 4 and columns
      // x + ex - 4.  The shift by ex is NOT applied here: the aligned 40-float segment [x0 - 4, x0 + 36) of every row
      // is staged as ten 16-B pieces (Gs[e][r][0..39]) and the FMA loop reads G at column 4q + ex + p.  (The first
      // version staged the shifted rows float by float: 42 dword loads per thread with the mask re-read, half of every
      // launch -- profiles/r02_scorr_microbench_and_ablation.txt.)
      constexpr int NPI = PS * PS * BTH * (GW / 4);            // 1620 pieces for the 2-row tile
      constexpr int UG = (NPI + NT - 1) / NT;                  // 7 per thread
      f32x4 g[UG], f[UG];
      bool okg[UG];
#pragma unroll
      for (int k = 0; k < UG; ++k) {
        const int e = tid + k * NT;
        const int d = e / (BTH * (GW / 4)), rem = e - d * (BTH * (GW / 4));
        const int rr = rem / (GW / 4), m = rem - rr * (GW / 4);
        const int ey = d / PS;
        const int gy = y0 + rr + ey - R, gx = x0 - R + 4 * m;
        okg[k] = e < NPI && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t off = okg[k] ? (size_t)(PS * PS - 1 - d) * plane + (size_t)(gy * W + gx) : 0;
        g[k] = *reinterpret_cast<const f32x4*>(gb + off);
        f[k] = fb ? *reinterpret_cast<const f32x4*>(fb + off) : (f32x4)(1.f);
      }
#pragma unroll
      for (int k = 0; k < UG; ++k) {
        const int e = tid + k * NT;
        f32x4 t = g[k] * gscale;
        if (fb) {
          t.x = f[k].x > 0.f ? t.x : t.x * slope; t.y = f[k].y > 0.f ? t.y : t.y * slope;
          t.z = f[k].z > 0.f ? t.z : t.z * slope; t.w = f[k].w > 0.f ? t.w : t.w * slope;
        }
        if (e < NPI) *reinterpret_cast<f32x4*>(&Gs[0][0][0] + 4 * e) = okg[k] ? t : (f32x4)(0.f);
      }
    }
    if (xl < XL) {
#pragma unroll
      for (int k = 0; k < XU; ++k)
        *reinterpret_cast<f32x4*>(&Xs[xl + k * XL][0][0] + 4 * xp) = v[k];
    }
  }
  __syncthreads();

  // ---- thread = 4 pixels x BCPT channels ----
  const int q = tid % QW, r = (tid / QW) % BTH, sub = tid / (QW * BTH);
  float acc[BCPT][4];
#pragma unroll
  for (int cc = 0; cc < BCPT; ++cc)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[cc][p] = 0.f;
#pragma unroll 1
  for (int ey = 0; ey < ((dbg & 2) ? 0 : PS); ++ey) {
    f32x4 G[PS];
    if (which == 0) {
#pragma unroll
      for (int ex = 0; ex < PS; ++ex) G[ex] = *reinterpret_cast<const f32x4*>(&Gs[ey * PS + ex][r][4 * q]);
    } else {   // shifted by ex: unaligned, four dword reads per tap
#pragma unroll
      for (int ex = 0; ex < PS; ++ex) {
        const float* gp4 = &Gs[ey * PS + ex][r][4 * q + ex];
        G[ex] = f32x4{gp4[0], gp4[1], gp4[2], gp4[3]};
      }
    }
#pragma unroll
    for (int cc = 0; cc < BCPT; ++cc) {
      const float* row = &Xs[sub * BCPT + cc][r + ey][4 * q];
      float x[12];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f32x4 tt = *reinterpret_cast<const f32x4*>(row + 4 * k);
        x[4 * k] = tt.x; x[4 * k + 1] = tt.y; x[4 * k + 2] = tt.z; x[4 * k + 3] = tt.w;
      }
#pragma unroll
      for (int ex = 0; ex < PS; ++ex) {
        acc[cc][0] += G[ex].x * x[ex];
        acc[cc][1] += G[ex].y * x[ex + 1];
        acc[cc][2] += G[ex].z * x[ex + 2];
        acc[cc][3] += G[ex].w * x[ex + 3];
      }
    }
  }
  const int gy = y0 + r, gx = x0 + 4 * q;
  if (gy < H && gx < W && (!(dbg & 4) || acc[0][0] == 123.f)) {
#pragma unroll
    for (int cc = 0; cc < BCPT; ++cc) {
      const int c = c0 + sub * BCPT + cc;
      if (c < C) {
        const f32x4 v = {acc[cc][0], acc[cc][1], acc[cc][2], acc[cc][3]};
        *reinterpret_cast<f32x4*>(po + (size_t)c * plane + (size_t)gy * W + gx) = v;
      }
    }
  }
}

int scorr9_backward(const float* in1, const float* in2, const float* gout, const float* fwd_out, float* gin1,
                    float* gin2, int B, int C, int H, int W, float gscale, float slope, hipStream_t s) {
  // 4-row tiles (half the workgroups, taps staged once per 128 pixels) measured SLOWER at every PWC level
  // (103 KB of LDS = one workgroup per CU, nothing overlaps its load phase): kept for tuning only
  static const int dbg = getenv("PCFA_SC_DBG") ? atoi(getenv("PCFA_SC_DBG")) : 0;   // phase ablation (tools/dev)
  bool tall = false;
  static const int env_bth = getenv("PCFA_SC_BTH") ? atoi(getenv("PCFA_SC_BTH")) : 0;   // tuning override (tools/dev), read once
  if (env_bth) tall = env_bth == 4;
  const int tiles_x = pcfa_cdiv(W, BTW), nz = 2 * B * pcfa_cdiv(C, BCG);
  if (tall) {
    const int ntiles = tiles_x * pcfa_cdiv(H, 4);
    pcfa_launch(scorr9_bwd_kernel<4>, dim3(xcd_grid(ntiles, nz)), dim3(256), 0, s, in1, in2, gout, fwd_out, gin1, gin2, C, H,
                W, gscale, slope, dbg, tiles_x, ntiles, nz);
  } else {
    const int ntiles = tiles_x * pcfa_cdiv(H, 2);
    pcfa_launch(scorr9_bwd_kernel<2>, dim3(xcd_grid(ntiles, nz)), dim3(256), 0, s, in1, in2, gout, fwd_out, gin1, gin2, C, H,
                W, gscale, slope, dbg, tiles_x, ntiles, nz);
  }
  return PCFA_OK;
}

struct ScParams {
  int B, C, iH, iW, oH, oW;
  int kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW;
};

__global__ void scorr_fwd_generic_kernel(const float* __restrict__ in1,
                                         const float* __restrict__ in2, float* __restrict__ out,
                                         ScParams p) {
  const long long total = (long long)p.B * p.patchH * p.patchW * p.oH * p.oW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    long long t = idx;
    const int w = t % p.oW; t /= p.oW;
    const int h = t % p.oH; t /= p.oH;
    const int pw = t % p.patchW; t /= p.patchW;
    const int ph = t % p.patchH; t /= p.patchH;
    const int b = (int)t;
    const int u = -p.padH + h * p.dH, v = -p.padW + w * p.dW;
    const int sU = (ph - (p.patchH - 1) / 2) * p.dpH, sV = (pw - (p.patchW - 1) / 2) * p.dpW;
    const size_t plane = (size_t)p.iH * p.iW;
    const float* a = in1 + (size_t)b * p.C * plane;
    const float* c2 = in2 + (size_t)b * p.C * plane;
    float s = 0.f;
    for (int c = 0; c < p.C; ++c)
      for (int i = 0; i < p.kH; ++i) {
        const int i1 = u + i * p.dilH, i2 = i1 + sU;
        if (i1 < 0 || i1 >= p.iH || i2 < 0 || i2 >= p.iH) continue;
        for (int j = 0; j < p.kW; ++j) {
          const int j1 = v + j * p.dilW, j2 = j1 + sV;
          if (j1 < 0 || j1 >= p.iW || j2 < 0 || j2 >= p.iW) continue;
          s += a[c * plane + (size_t)i1 * p.iW + j1] * c2[c * plane + (size_t)i2 * p.iW + j2];
        }
      }
    out[idx] = s;
  }
}

// WHICH = 1: gradient w.r.t. in1 (other = in2), WHICH = 2: w.r.t. in2 (other = in1).
template <int WHICH>
__global__ void scorr_bwd_generic_kernel(const float* __restrict__ other,
                                         const float* __restrict__ gout, float* __restrict__ gin,
                                         ScParams p) {
  const long long total = (long long)p.B * p.C * p.iH * p.iW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const size_t plane = (size_t)p.iH * p.iW, oplane = (size_t)p.oH * p.oW;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += stride) {
    long long t = idx;
    const int x = t % p.iW; t /= p.iW;
    const int y = t % p.iH; t /= p.iH;
    const int c = t % p.C; t /= p.C;
    const int b = (int)t;
    const float* oc = other + ((size_t)b * p.C + c) * plane;
    const float* gb = gout + (size_t)b * p.patchH * p.patchW * oplane;
    float s = 0.f;
    for (int ph = 0; ph < p.patchH; ++ph) {
      const int sU = (ph - (p.patchH - 1) / 2) * p.dpH;
      for (int pw = 0; pw < p.patchW; ++pw) {
        const int sV = (pw - (p.patchW - 1) / 2) * p.dpW;
        for (int i = 0; i < p.kH; ++i) {
          // position in in1 (i1) and in2 (i2) for this tap
          const int i1 = (WHICH == 1) ? y : y - sU;
          const int i2 = i1 + sU;
          if (i1 < 0 || i1 >= p.iH || i2 < 0 || i2 >= p.iH) continue;
          const int hn = i1 + p.padH - i * p.dilH;
          if (hn < 0 || hn % p.dH != 0) continue;
          const int h = hn / p.dH;
          if (h >= p.oH) continue;
          for (int j = 0; j < p.kW; ++j) {
            const int j1 = (WHICH == 1) ? x : x - sV;
            const int j2 = j1 + sV;
            if (j1 < 0 || j1 >= p.iW || j2 < 0 || j2 >= p.iW) continue;
            const int wn = j1 + p.padW - j * p.dilW;
            if (wn < 0 || wn % p.dW != 0) continue;
            const int w = wn / p.dW;
            if (w >= p.oW) continue;
            const float g = gb[(size_t)(ph * p.patchW + pw) * oplane + (size_t)h * p.oW + w];
            const float o = (WHICH == 1) ? oc[(size_t)i2 * p.iW + j2] : oc[(size_t)i1 * p.iW + j1];
            s += g * o;
          }
        }
      }
    }
    gin[idx] = s;
  }
}

bool make_params(ScParams& p, int B, int C, int iH, int iW, int kH, int kW, int patchH,
                 int patchW, int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH,
                 int dW) {
  if (B < 1 || C < 1 || iH < 1 || iW < 1 || kH < 1 || kW < 1 || patchH < 1 || patchW < 1 ||
      padH < 0 || padW < 0 || dilH < 1 || dilW < 1 || dpH < 1 || dpW < 1 || dH < 1 || dW < 1)
    return false;
  const int dkH = (kH - 1) * dilH + 1, dkW = (kW - 1) * dilW + 1;
  const int nH = iH + 2 * padH - dkH, nW = iW + 2 * padW - dkW;
  if (nH < 0 || nW < 0) return false;
  p = ScParams{B, C, iH, iW, nH / dH + 1, nW / dW + 1, kH, kW, patchH, patchW, padH, padW,
               dilH, dilW, dpH, dpW, dH, dW};
  return true;
}

bool is_fast(const ScParams& p, int ps) {
  return p.kH == 1 && p.kW == 1 && p.dH == 1 && p.dW == 1 && p.padH == 0 && p.padW == 0 &&
         p.dpH == 1 && p.dpW == 1 && p.patchH == ps && p.patchW == ps;
}

}  // namespace

extern "C" int pcfa_spatial_corr_out_size(int iH, int iW, int kH, int kW, int padH, int padW,
                                          int dilH, int dilW, int dH, int dW, int* oH, int* oW) {
  ScParams p;
  if (!make_params(p, 1, 1, iH, iW, kH, kW, 1, 1, padH, padW, dilH, dilW, 1, 1, dH, dW))
    return PCFA_ERR_INVALID_ARG;
  if (oH) *oH = p.oH;
  if (oW) *oW = p.oW;
  return PCFA_OK;
}

extern "C" int pcfa_spatial_corr_fwd(const float* in1, const float* in2, float* out, int B, int C,
                                     int iH, int iW, int kH, int kW, int patchH, int patchW,
                                     int padH, int padW, int dilH, int dilW, int dil_patchH,
                                     int dil_patchW, int dH, int dW, void* stream) {
  ScParams p;
  if (!in1 || !in2 || !out ||
      !make_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH,
                   dil_patchW, dH, dW))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = iW % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (is_fast(p, 9) && aligned) {
    const int rc = scorr9_forward(in1, in2, out, B, C, iH, iW, 1.f, 1.f, s);
    if (rc != PCFA_OK) return rc;
  } else {
    const long long total = (long long)B * patchH * patchW * p.oH * p.oW;
    const int blocks = (int)((total + 255) / 256 < 65535LL * 16 ? (total + 255) / 256 : 65535LL * 16);
    pcfa_launch(scorr_fwd_generic_kernel, dim3(blocks), dim3(256), 0, s, in1, in2, out, p);
  }
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_spatial_corr_bwd(const float* in1, const float* in2, const float* grad_out,
                                     float* grad_in1, float* grad_in2, int B, int C, int iH,
                                     int iW, int kH, int kW, int patchH, int patchW, int padH,
                                     int padW, int dilH, int dilW, int dil_patchH, int dil_patchW,
                                     int dH, int dW, void* stream) {
  ScParams p;
  if (!in1 || !in2 || !grad_out || !grad_in1 || !grad_in2 ||
      !make_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH,
                   dil_patchW, dH, dW))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = iW % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2) | reinterpret_cast<uintptr_t>(grad_out) |
        reinterpret_cast<uintptr_t>(grad_in1) | reinterpret_cast<uintptr_t>(grad_in2)) & 15) == 0;
  if (is_fast(p, 9) && aligned) {
    const int rc = scorr9_backward(in1, in2, grad_out, nullptr, grad_in1, grad_in2, B, C, iH, iW, 1.f, 1.f, s);
    if (rc != PCFA_OK) return rc;
  } else {
    const long long total = (long long)B * C * iH * iW;
    const int blocks = (int)((total + 255) / 256 < 65535LL * 16 ? (total + 255) / 256 : 65535LL * 16);
    pcfa_launch(scorr_bwd_generic_kernel<1>, dim3(blocks), dim3(256), 0, s, in2, grad_out,
                       grad_in1, p);
    PCFA_LAUNCH_CHECK();
    pcfa_launch(scorr_bwd_generic_kernel<2>, dim3(blocks), dim3(256), 0, s, in1, grad_out,
                       grad_in2, p);
  }
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

// PWC-Net's use of the sampler, fused: out = leaky_relu(correlate(in1, in2)) with correlate = 9x9 cost volume / C
// (models/PWCNet/PWCNet.py:45-58 followed by self.leakyRELU at :249,264,278,292,308).  scale = 1/C and the
// LeakyReLU slope ride in the forward epilogue; the backward applies mask * scale to the gradient taps while it
// stages them (fwd_out = the forward's output, whose sign is the mask).  iW % 4 == 0 and 16-B aligned pointers.
extern "C" int pcfa_cost_volume9_fwd(const float* in1, const float* in2, float* out, int B, int C, int iH, int iW,
                                     float scale, float slope, void* stream) {
  if (!in1 || !in2 || !out || B < 1 || C < 1 || iH < 1 || iW < 1) return PCFA_ERR_INVALID_ARG;
  if (iW % 4 != 0 ||
      ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2) | reinterpret_cast<uintptr_t>(out)) & 15))
    return PCFA_ERR_UNSUPPORTED;
  const int rc = scorr9_forward(in1, in2, out, B, C, iH, iW, scale, slope, (hipStream_t)stream);
  if (rc != PCFA_OK) return rc;
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_cost_volume9_bwd(const float* in1, const float* in2, const float* fwd_out, const float* grad_out,
                                     float* grad_in1, float* grad_in2, int B, int C, int iH, int iW, float scale,
                                     float slope, void* stream) {
  if (!in1 || !in2 || !fwd_out || !grad_out || !grad_in1 || !grad_in2 || B < 1 || C < 1 || iH < 1 || iW < 1)
    return PCFA_ERR_INVALID_ARG;
  if (iW % 4 != 0 ||
      ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2) | reinterpret_cast<uintptr_t>(fwd_out) |
        reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_in1) |
        reinterpret_cast<uintptr_t>(grad_in2)) & 15))
    return PCFA_ERR_UNSUPPORTED;
  const int rc = scorr9_backward(in1, in2, grad_out, fwd_out, grad_in1, grad_in2, B, C, iH, iW, scale, slope,
                                 (hipStream_t)stream);
  if (rc != PCFA_OK) return rc;
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
