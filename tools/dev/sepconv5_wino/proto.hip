// PROTOTYPE (tools/dev, not part of libpcfa_hip.so): the 1x5 gate convolution of SepConvGRU (update.py:33-60) as a 1-D
// Winograd F(2,5) on v_mfma_f32_32x32x2_f32 -- 6 products per 2 outputs instead of 10.  Same interpolation points as the
// F(4x4,3x3) kernel (0, +-1, +-2, inf), so B^T is that kernel's row stage; plain output, no GRU epilogue.  Question it
// answers: what does the transform leave of the 1.67x fewer MFMAs at the RAFT shape (256 -> 256 channels, 55x128)?
//   y[n][x] = sum_c sum_t w[n][c][t] in[c][x + t - 2];   pair j = outputs 2j, 2j+1 from d_i = in[c][2j - 2 + i], i = 0..5
//   V = B^T d,  M_i[n][j] += U_i[n][c] V_i[c][j] (one MFMA per point and channel pair),  y = A^T M
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int CK = 8, STEPS = (CK / 2) * 6, PXT = 128, RVP = PXT / 4 + 2, RSP = 4 * RVP;   // patch row: x0 - 4 .. x0 + 131
constexpr int CHS = RSP, PATCH = CK * CHS;
constexpr int NV = CK * RVP, NLOAD = (NV + 255) / 256;
constexpr int VCHS = 6 * 64, VPATCH = CK * VCHS, VNV = CK * 6 * 16, VNLOAD = (VNV + 255) / 256;   // 5x1: [ch][6 rows][64 px]

__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ P, int N, int C, int nchunk, long long total) {
  const float G[6][5] = {{0.25f, 0.f, 0.f, 0.f, 0.f},
                         {-1.f / 6, -1.f / 6, -1.f / 6, -1.f / 6, -1.f / 6},
                         {-1.f / 6, 1.f / 6, -1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 2.f / 24, 4.f / 24, 8.f / 24, 16.f / 24},
                         {1.f / 24, -2.f / 24, 4.f / 24, -8.f / 24, 16.f / 24},
                         {0.f, 0.f, 0.f, 0.f, 1.f}};
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63), s = (int)((e >> 6) % STEPS);
    const long long blk = (e >> 6) / STEPS;
    const int chunk = (int)(blk % nchunk), nb = (int)(blk / nchunk);
    const int n = 32 * nb + (lane & 31), c = chunk * CK + 2 * (s / 6) + (lane >> 5), i = s % 6;
    double u = 0.0;
    if (n < N && c < C)
      for (int t = 0; t < 5; ++t) u += (double)G[i][t] * (double)w[((long long)n * C + c) * 5 + t];
    P[e] = (float)u;
  }
}

__global__ __launch_bounds__(256) void wino15_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                     float* __restrict__ out, int C, int N, int H, int W) {
  __shared__ __attribute__((aligned(16))) float smem[2 * PATCH];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wv & 1, wpx = wv >> 1;
  const int tiles_x = W / PXT;
  const int y = blockIdx.x / tiles_x, x0 = (blockIdx.x - y * tiles_x) * PXT;
  const int nb = blockIdx.y * 2 + wn;
  const int plane = H * W, nchunk = C / CK;

  auto load_patch = [&](int chunk, float4 (&rr)[NLOAD], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const int e = min(tid + 256 * k, NV - 1);
      const int c = e / RVP, v = e - c * RVP;
      const int ix = x0 - 4 + 4 * v;
      okm |= (unsigned)((int)(ix >= 0) & (int)(ix + 3 < W)) << k;
      rr[k] = *reinterpret_cast<const float4*>(x + (unsigned)((chunk * CK + c) * plane + y * W + min(max(ix, 0), W - 4)));
    }
  };
  auto store_patch = [&](int buf, const float4 (&rr)[NLOAD], unsigned okm) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const int e = tid + 256 * k;
      const int c = e / RVP, v = e - c * RVP;
      const float4 t = (okm >> k & 1u) ? rr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NV) *reinterpret_cast<float4*>(smem + buf * PATCH + c * CHS + 4 * v) = t;
    }
  };
  const float* pw = wp + ((long long)nb * nchunk * STEPS) * 64;

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 ra[NLOAD], rb[NLOAD];
  unsigned oka, okb;
  float wa[STEPS], wb[STEPS];
  load_patch(0, ra, oka);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) wa[s] = pw[s * 64 + lane];
  store_patch(0, ra, oka);
  __syncthreads();
  load_patch(min(1, nchunk - 1), rb, okb);
  // lane: pair j = 32 wpx + l31 of the tile -> patch columns 2 j + 2 .. 2 j + 7 (x = x0 + 2 j - 2 ..), channel 2 p + lh
  const int bl = lh * CHS + 2 * (32 * wpx + l31) + 2;

  auto item = [&](int chunk, const float (&wcur)[STEPS], float (&wnext)[STEPS], float4 (&rload)[NLOAD], unsigned& okload,
                  const float4 (&rstore)[NLOAD], const unsigned& okstore) {
    const float* sp = smem + (chunk & 1) * PATCH + bl;
    load_patch(min(chunk + 2, nchunk - 1), rload, okload);
    const float* qn = pw + (long long)min(chunk + 1, nchunk - 1) * STEPS * 64;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) wnext[s] = qn[s * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    float2 d[2][3];
#pragma unroll
    for (int h = 0; h < 3; ++h) d[0][h] = *reinterpret_cast<const float2*>(sp + 2 * h);
#pragma unroll
    for (int p = 0; p < CK / 2; ++p) {
      if (p + 1 < CK / 2)
#pragma unroll
        for (int h = 0; h < 3; ++h) d[(p + 1) & 1][h] = *reinterpret_cast<const float2*>(sp + 2 * (p + 1) * CHS + 2 * h);
      const float d0 = d[p & 1][0].x, d1 = d[p & 1][0].y, d2 = d[p & 1][1].x, d3 = d[p & 1][1].y, d4 = d[p & 1][2].x,
                  d5 = d[p & 1][2].y;
      // B^T d: the row stage of the F(4x4,3x3) kernel
      const float v0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
      const float a1 = fmaf(-4.f, d2, d4), b1 = fmaf(-4.f, d1, d3);
      const float v1 = a1 + b1, v2 = a1 - b1;
      const float a2 = d4 - d2, b2 = 2.f * (d3 - d1);
      const float v3 = a2 + b2, v4 = a2 - b2;
      const float v5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 0], v0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 1], v1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 2], v2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 3], v3, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 4], v4, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 5], v5, acc[5], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_patch((chunk + 1) & 1, rstore, okstore);
    __syncthreads();
  };
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
    item(chunk, wa, wb, ra, oka, rb, okb);
    item(chunk + 1, wb, wa, rb, okb, ra, oka);
  }

  // y(2j) = M0 + M1 + M2 + M3 + M4,  y(2j+1) = M1 - M2 + 2 (M3 - M4) + M5
  float* ob = out + (long long)(32 * nb) * plane + (long long)y * W + x0 + 2 * (32 * wpx);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int nu = 8 * (r >> 2) + (r & 3);
    const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r], s34 = acc[3][r] + acc[4][r],
                d34 = acc[3][r] - acc[4][r];
    const float ye = acc[0][r] + s12 + s34, yo = fmaf(2.f, d34, d12) + acc[5][r];
    if (32 * nb + nu + 4 * lh < N)
      *reinterpret_cast<float2*>(ob + (long long)nu * plane + (unsigned)(4 * lh * plane + 2 * l31)) = make_float2(ye, yo);
  }
}
// The same kernel with the input channels split over two groups of four waves (even / odd chunks): two waves per SIMD at
// the RAFT shape, where the plain grid has 880 waves for 1024 SIMDs.  Partial outputs meet in LDS after the output
// transform (it is linear), group 0 stores.
__global__ __launch_bounds__(512) void wino15_ks2_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                     float* __restrict__ out, int C, int N, int H, int W) {
  __shared__ __attribute__((aligned(16))) float smem_all[4 * PATCH + 4 * 32 * 64];
  const int grp = threadIdx.x >> 8;
  float* smem = smem_all + grp * 2 * PATCH;
  float* sred = smem_all + 4 * PATCH;
  const int tid = threadIdx.x & 255, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave inside the group
  const int wn = wv & 1, wpx = wv >> 1;
  const int tiles_x = W / PXT;
  const int y = blockIdx.x / tiles_x, x0 = (blockIdx.x - y * tiles_x) * PXT;
  const int nb = blockIdx.y * 2 + wn;
  const int plane = H * W, nchunk = C / CK / 2;   // chunks of THIS group: global chunk 2 c + grp

  auto load_patch = [&](int chunk, float4 (&rr)[NLOAD], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const int e = min(tid + 256 * k, NV - 1);
      const int c = e / RVP, v = e - c * RVP;
      const int ix = x0 - 4 + 4 * v;
      okm |= (unsigned)((int)(ix >= 0) & (int)(ix + 3 < W)) << k;
      rr[k] = *reinterpret_cast<const float4*>(x + (unsigned)(((2 * chunk + grp) * CK + c) * plane + y * W + min(max(ix, 0), W - 4)));
    }
  };
  auto store_patch = [&](int buf, const float4 (&rr)[NLOAD], unsigned okm) {
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const int e = tid + 256 * k;
      const int c = e / RVP, v = e - c * RVP;
      const float4 t = (okm >> k & 1u) ? rr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NV) *reinterpret_cast<float4*>(smem + buf * PATCH + c * CHS + 4 * v) = t;
    }
  };
  const float* pw = wp + ((long long)nb * (2 * nchunk) * STEPS) * 64;

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 ra[NLOAD], rb[NLOAD];
  unsigned oka, okb;
  float wa[STEPS], wb[STEPS];
  load_patch(0, ra, oka);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) wa[s] = pw[((long long)grp * STEPS + s) * 64 + lane];
  store_patch(0, ra, oka);
  __syncthreads();
  load_patch(min(1, nchunk - 1), rb, okb);
  // lane: pair j = 32 wpx + l31 of the tile -> patch columns 2 j + 2 .. 2 j + 7 (x = x0 + 2 j - 2 ..), channel 2 p + lh
  const int bl = lh * CHS + 2 * (32 * wpx + l31) + 2;

  auto item = [&](int chunk, const float (&wcur)[STEPS], float (&wnext)[STEPS], float4 (&rload)[NLOAD], unsigned& okload,
                  const float4 (&rstore)[NLOAD], const unsigned& okstore) {
    const float* sp = smem + (chunk & 1) * PATCH + bl;
    load_patch(min(chunk + 2, nchunk - 1), rload, okload);
    const float* qn = pw + (long long)(2 * min(chunk + 1, nchunk - 1) + grp) * STEPS * 64;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) wnext[s] = qn[s * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    float2 d[2][3];
#pragma unroll
    for (int h = 0; h < 3; ++h) d[0][h] = *reinterpret_cast<const float2*>(sp + 2 * h);
#pragma unroll
    for (int p = 0; p < CK / 2; ++p) {
      if (p + 1 < CK / 2)
#pragma unroll
        for (int h = 0; h < 3; ++h) d[(p + 1) & 1][h] = *reinterpret_cast<const float2*>(sp + 2 * (p + 1) * CHS + 2 * h);
      const float d0 = d[p & 1][0].x, d1 = d[p & 1][0].y, d2 = d[p & 1][1].x, d3 = d[p & 1][1].y, d4 = d[p & 1][2].x,
                  d5 = d[p & 1][2].y;
      // B^T d: the row stage of the F(4x4,3x3) kernel
      const float v0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
      const float a1 = fmaf(-4.f, d2, d4), b1 = fmaf(-4.f, d1, d3);
      const float v1 = a1 + b1, v2 = a1 - b1;
      const float a2 = d4 - d2, b2 = 2.f * (d3 - d1);
      const float v3 = a2 + b2, v4 = a2 - b2;
      const float v5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 0], v0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 1], v1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 2], v2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 3], v3, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 4], v4, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 5], v5, acc[5], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_patch((chunk + 1) & 1, rstore, okstore);
    __syncthreads();
  };
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
    item(chunk, wa, wb, ra, oka, rb, okb);
    item(chunk + 1, wb, wa, rb, okb, ra, oka);
  }

  // y(2j) = M0 + M1 + M2 + M3 + M4,  y(2j+1) = M1 - M2 + 2 (M3 - M4) + M5; group 1 hands its part to group 0 through LDS
  float* ob = out + (long long)(32 * nb) * plane + (long long)y * W + x0 + 2 * (32 * wpx);
  float ye[16], yo[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r], s34 = acc[3][r] + acc[4][r],
                d34 = acc[3][r] - acc[4][r];
    ye[r] = acc[0][r] + s12 + s34;
    yo[r] = fmaf(2.f, d34, d12) + acc[5][r];
  }
  if (grp == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sred[(wv * 32 + r) * 64 + lane] = ye[r];
      sred[(wv * 32 + 16 + r) * 64 + lane] = yo[r];
    }
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nu = 8 * (r >> 2) + (r & 3);
      const float e_ = ye[r] + sred[(wv * 32 + r) * 64 + lane], o_ = yo[r] + sred[(wv * 32 + 16 + r) * 64 + lane];
      if (32 * nb + nu + 4 * lh < N)
        *reinterpret_cast<float2*>(ob + (long long)nu * plane + (unsigned)(4 * lh * plane + 2 * l31)) = make_float2(e_, o_);
    }
  }
}
// The 5x1 (vertical) form: pair = output rows 2 r, 2 r + 1 at one x; lane = x, d_i = in[c][2 r - 2 + i][x] (six patch rows);
// workgroup = one row pair x 64 pixels x 64 channels, input channels split over two wave groups as above.
__global__ __launch_bounds__(512) void wino51_ks2_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                     float* __restrict__ out, int C, int N, int H, int W) {
  __shared__ __attribute__((aligned(16))) float smem_all[4 * VPATCH + 4 * 32 * 64];
  const int grp = threadIdx.x >> 8;
  float* smem = smem_all + grp * 2 * VPATCH;
  float* sred = smem_all + 4 * VPATCH;
  const int tid = threadIdx.x & 255, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave inside the group
  const int wn = wv & 1, wpx = wv >> 1;
  const int tiles_x = W / 64;
  const int rp = blockIdx.x / tiles_x, x0 = (blockIdx.x - rp * tiles_x) * 64;   // row pair: outputs 2 rp, 2 rp + 1
  const int nb = blockIdx.y * 2 + wn;
  const int plane = H * W, nchunk = C / CK / 2;   // chunks of THIS group: global chunk 2 c + grp

  auto load_patch = [&](int chunk, float4 (&rr)[VNLOAD], unsigned& okm) {
    okm = 0;
#pragma unroll
    for (int k = 0; k < VNLOAD; ++k) {
      const int e = min(tid + 256 * k, VNV - 1);
      const int c = e / 96, rem = e - c * 96, r = rem >> 4, v = rem & 15;
      const int iy = 2 * rp - 2 + r;
      okm |= (unsigned)((int)(iy >= 0) & (int)(iy < H)) << k;
      rr[k] = *reinterpret_cast<const float4*>(x + (unsigned)(((2 * chunk + grp) * CK + c) * plane + min(max(iy, 0), H - 1) * W + x0 + 4 * v));
    }
  };
  auto store_patch = [&](int buf, const float4 (&rr)[VNLOAD], unsigned okm) {
#pragma unroll
    for (int k = 0; k < VNLOAD; ++k) {
      const int e = tid + 256 * k;
      const int c = e / 96, rem = e - c * 96, r = rem >> 4, v = rem & 15;
      const float4 t = (okm >> k & 1u) ? rr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < VNV) *reinterpret_cast<float4*>(smem + buf * VPATCH + c * VCHS + r * 64 + 4 * v) = t;
    }
  };
  const float* pw = wp + ((long long)nb * (2 * nchunk) * STEPS) * 64;

  f32x16 acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 ra[VNLOAD], rb[VNLOAD];
  unsigned oka, okb;
  float wa[STEPS], wb[STEPS];
  load_patch(0, ra, oka);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) wa[s] = pw[((long long)grp * STEPS + s) * 64 + lane];
  store_patch(0, ra, oka);
  __syncthreads();
  load_patch(min(1, nchunk - 1), rb, okb);
  // lane: x = x0 + 32 wpx + l31, channel 2 p + lh; the six rows of the pair are 64 floats apart
  const int bl = lh * VCHS + 32 * wpx + l31;

  auto item = [&](int chunk, const float (&wcur)[STEPS], float (&wnext)[STEPS], float4 (&rload)[VNLOAD], unsigned& okload,
                  const float4 (&rstore)[VNLOAD], const unsigned& okstore) {
    const float* sp = smem + (chunk & 1) * VPATCH + bl;
    load_patch(min(chunk + 2, nchunk - 1), rload, okload);
    const float* qn = pw + (long long)(2 * min(chunk + 1, nchunk - 1) + grp) * STEPS * 64;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) wnext[s] = qn[s * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    float d[2][6];
#pragma unroll
    for (int h = 0; h < 6; ++h) d[0][h] = sp[64 * h];
#pragma unroll
    for (int p = 0; p < CK / 2; ++p) {
      if (p + 1 < CK / 2)
#pragma unroll
        for (int h = 0; h < 6; ++h) d[(p + 1) & 1][h] = sp[2 * (p + 1) * VCHS + 64 * h];
      const float d0 = d[p & 1][0], d1 = d[p & 1][1], d2 = d[p & 1][2], d3 = d[p & 1][3], d4 = d[p & 1][4], d5 = d[p & 1][5];
      // B^T d: the row stage of the F(4x4,3x3) kernel
      const float v0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
      const float a1 = fmaf(-4.f, d2, d4), b1 = fmaf(-4.f, d1, d3);
      const float v1 = a1 + b1, v2 = a1 - b1;
      const float a2 = d4 - d2, b2 = 2.f * (d3 - d1);
      const float v3 = a2 + b2, v4 = a2 - b2;
      const float v5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 0], v0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 1], v1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 2], v2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 3], v3, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 4], v4, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[6 * p + 5], v5, acc[5], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    store_patch((chunk + 1) & 1, rstore, okstore);
    __syncthreads();
  };
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
    item(chunk, wa, wb, ra, oka, rb, okb);
    item(chunk + 1, wb, wa, rb, okb, ra, oka);
  }

  // y(2j) = M0 + M1 + M2 + M3 + M4,  y(2j+1) = M1 - M2 + 2 (M3 - M4) + M5; group 1 hands its part to group 0 through LDS
  float* ob = out + (long long)(32 * nb) * plane + (long long)(2 * rp) * W + x0 + 32 * wpx;
  float ye[16], yo[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r], s34 = acc[3][r] + acc[4][r],
                d34 = acc[3][r] - acc[4][r];
    ye[r] = acc[0][r] + s12 + s34;
    yo[r] = fmaf(2.f, d34, d12) + acc[5][r];
  }
  if (grp == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sred[(wv * 32 + r) * 64 + lane] = ye[r];
      sred[(wv * 32 + 16 + r) * 64 + lane] = yo[r];
    }
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nu = 8 * (r >> 2) + (r & 3);
      const float e_ = ye[r] + sred[(wv * 32 + r) * 64 + lane], o_ = yo[r] + sred[(wv * 32 + 16 + r) * 64 + lane];
      if (32 * nb + nu + 4 * lh < N) {
        float* o = ob + (long long)nu * plane + (unsigned)(4 * lh * plane + l31);
        o[0] = e_;
        if (2 * rp + 1 < H) o[W] = o_;
      }
    }
  }
}
}  // namespace

extern "C" {
__attribute__((visibility("default"))) long long wino15_packed_floats(int C, int N) {
  return (long long)((N + 31) / 32 + 1) / 2 * 2 * (C / CK) * STEPS * 64;
}
__attribute__((visibility("default"))) int wino15_pack(const float* w, float* packed, int C, int N, void* stream) {
  const long long total = wino15_packed_floats(C, N);
  hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w, packed, N, C, C / CK, total);
  return (int)hipGetLastError();
}
__attribute__((visibility("default"))) int wino51_run(const float* x, const float* packed, float* out, int C, int N, int H,
                                                      int W, void* stream) {
  if (C % 32 != 0 || N % 64 != 0 || W % 64 != 0) return -2;
  hipLaunchKernelGGL(wino51_ks2_kernel, dim3((unsigned)((H + 1) / 2 * (W / 64)), (unsigned)(N / 64)), dim3(512), 0,
                     (hipStream_t)stream, x, packed, out, C, N, H, W);
  return (int)hipGetLastError();
}
// x [C][H][W], out [N][H][W]; C % 16 == 0, N % 64 == 0, W % 128 == 0
__attribute__((visibility("default"))) int wino15_run(const float* x, const float* packed, float* out, int C, int N, int H,
                                                      int W, void* stream) {
  if (C % 16 != 0 || N % 64 != 0 || W % PXT != 0) return -2;
  static const int ks2 = getenv("WINO_KS2") ? atoi(getenv("WINO_KS2")) : 0;
  if (ks2 && C % 32 == 0)
    hipLaunchKernelGGL(wino15_ks2_kernel, dim3((unsigned)(H * (W / PXT)), (unsigned)(N / 64)), dim3(512), 0,
                       (hipStream_t)stream, x, packed, out, C, N, H, W);
  else
    hipLaunchKernelGGL(wino15_kernel, dim3((unsigned)(H * (W / PXT)), (unsigned)(N / 64)), dim3(256), 0, (hipStream_t)stream,
                       x, packed, out, C, N, H, W);
  return (int)hipGetLastError();
}
}
