// Vector math of the L-BFGS iteration that drives the PCFA attack, for gfx950.
//
// Replaces, inside torch.optim.LBFGS.step (the optimiser the reference calls at attack_PCFA.py:97,114,382,388
// with max_iter=10, no line search), the memory update and the two-loop recursion:
//     y = g - g_prev;  s = t*d;  ys = y.s;  H = ys / y.y                                   (pcfa_lbfgs_pair)
//     q = -g;  for i = m-1..0: al_i = ro_i (s_i.q);  q -= al_i y_i
//     r = H q; for i = 0..m-1: be_i = ro_i (y_i.r);  r += (al_i - be_i) s_i;  d = r        (pcfa_lbfgs_direction)
// torch runs this as 4m+3 separate vector kernels of 10.8 MB each (dot = 2 launches), launched from Python: at
// m = 100 that is 7 ms per iteration next to a 21 ms closure.  Here the SAME sequence of operations runs as 2m+1
// launches: launch k applies the update whose coefficient follows from launch k-1's dot product and, in the same
// sweep over the vector, accumulates the next dot product.  Every block re-derives the coefficient from the
// previous launch's per-block partial sums (summed in index order: deterministic, identical in every block), so
// there is no finalize launch, no atomics and no host round trip anywhere in the recursion.  HBM-bound:
// 4 vector passes (43 MB at n = 2.7 M) per launch.
#include "common.hpp"

namespace {

constexpr int LB_THREADS = 256;
constexpr int LB_BLOCKS = 1024;  // = partial sums per dot product

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over the block, same value returned to every thread.
__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // s_red may still be read from a previous call
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < LB_THREADS / 64; ++w) t += s_red[w];
  return t;
}

// Total of the previous launch's LB_BLOCKS block partials, in a fixed order.
__device__ __forceinline__ float total_of(const float* __restrict__ partials, float* s_red) {
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < LB_BLOCKS / LB_THREADS; ++k) v += partials[threadIdx.x + k * LB_THREADS];
  return block_sum(v, s_red);
}

enum { LB_FIRST = 0, LB_LOOP1 = 1, LB_TURN = 2, LB_LOOP2 = 3, LB_LAST = 4 };

// One launch of the recursion on the vector x (= q, then r, finally d), see the header comment.
//   FIRST : x = -g                                   ; dot(b, x)        b = s_{m-1}
//   LOOP1 : al_i = ro_i * prev ; x -= al_i * a       ; dot(b, x)        a = y_i, b = s_{i-1}
//   TURN  : al_0 = ro_0 * prev ; x -= al_0 * a ; x *= H ; dot(b, x)     a = b = y_0
//   LOOP2 : be = ro_i * prev ; x += (al_i - be) * a  ; dot(b, x)        a = s_i, b = y_{i+1}
//   LAST  : be = ro_i * prev ; x += (al_i - be) * a                     a = s_{m-1}
template <int MODE>
__global__ __launch_bounds__(LB_THREADS) void lbfgs_step_kernel(
    float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ prev_partials, float* __restrict__ partials, const float* __restrict__ ro_i,
    float* __restrict__ al_i, const float* __restrict__ H, long long n) {
  __shared__ float s_red[LB_THREADS / 64];
  float c = 0.f, h = 1.f;
  if (MODE != LB_FIRST) {
    const float prev = total_of(prev_partials, s_red);
    const float v = prev * ro_i[0];  // torch: old_stps[i].dot(q) * ro[i]
    if (MODE == LB_LOOP1 || MODE == LB_TURN) {
      if (blockIdx.x == 0 && threadIdx.x == 0) al_i[0] = v;
      c = -v;
    } else {
      c = al_i[0] - v;
    }
    if (MODE == LB_TURN) h = H[0];
  }
  float acc = 0.f;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * LB_THREADS;
  for (long long i = (long long)blockIdx.x * LB_THREADS + threadIdx.x; i < n4; i += stride) {
    float4 xv;
    if (MODE == LB_FIRST) {
      const float4 gv = reinterpret_cast<const float4*>(g)[i];
      xv = make_float4(-gv.x, -gv.y, -gv.z, -gv.w);
    } else {
      xv = reinterpret_cast<const float4*>(x)[i];
      const float4 av = reinterpret_cast<const float4*>(a)[i];
      xv.x = fmaf(c, av.x, xv.x); xv.y = fmaf(c, av.y, xv.y); xv.z = fmaf(c, av.z, xv.z); xv.w = fmaf(c, av.w, xv.w);
      if (MODE == LB_TURN) { xv.x *= h; xv.y *= h; xv.z *= h; xv.w *= h; }
    }
    reinterpret_cast<float4*>(x)[i] = xv;
    if (MODE != LB_LAST) {
      const float4 bv = reinterpret_cast<const float4*>(b)[i];
      acc += xv.x * bv.x + xv.y * bv.y + xv.z * bv.z + xv.w * bv.w;
    }
  }
  // tail (n % 4 elements), by the first threads of block 0
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    float xv;
    if (MODE == LB_FIRST) {
      xv = -g[i];
    } else {
      xv = fmaf(c, a[i], x[i]);
      if (MODE == LB_TURN) xv *= h;
    }
    x[i] = xv;
    if (MODE != LB_LAST) acc += xv * b[i];
  }
  if (MODE != LB_LAST) {
    const float t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// y = g - g_prev ; s = t * d ; partial sums of y.s and y.y ; optionally g_prev = g.
__global__ __launch_bounds__(LB_THREADS) void lbfgs_pair_kernel(
    const float* __restrict__ g, float* __restrict__ g_prev, const float* __restrict__ d, float t,
    float* __restrict__ y, float* __restrict__ s, float* __restrict__ partials, int update_prev, long long n) {
  __shared__ float s_red[LB_THREADS / 64];
  float ys = 0.f, yy = 0.f;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * LB_THREADS;
  for (long long i = (long long)blockIdx.x * LB_THREADS + threadIdx.x; i < n4; i += stride) {
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    const float4 pv = reinterpret_cast<const float4*>(g_prev)[i];
    const float4 dv = reinterpret_cast<const float4*>(d)[i];
    const float4 yv = make_float4(gv.x - pv.x, gv.y - pv.y, gv.z - pv.z, gv.w - pv.w);
    const float4 sv = make_float4(dv.x * t, dv.y * t, dv.z * t, dv.w * t);
    reinterpret_cast<float4*>(y)[i] = yv;
    reinterpret_cast<float4*>(s)[i] = sv;
    if (update_prev) reinterpret_cast<float4*>(g_prev)[i] = gv;
    ys += yv.x * sv.x + yv.y * sv.y + yv.z * sv.z + yv.w * sv.w;
    yy += yv.x * yv.x + yv.y * yv.y + yv.z * yv.z + yv.w * yv.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float gv = g[i], yv = gv - g_prev[i], sv = d[i] * t;
    y[i] = yv;
    s[i] = sv;
    if (update_prev) g_prev[i] = gv;
    ys += yv * sv;
    yy += yv * yv;
  }
  const float tys = block_sum(ys, s_red);
  const float tyy = block_sum(yy, s_red);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = tys;
    partials[LB_BLOCKS + blockIdx.x] = tyy;
  }
}

// scal = { y.s, y.y, 1 / y.s, y.s / y.y }
__global__ __launch_bounds__(LB_THREADS) void lbfgs_pair_final_kernel(const float* __restrict__ partials,
                                                                      float* __restrict__ scal) {
  __shared__ float s_red[LB_THREADS / 64];
  const float ys = total_of(partials, s_red);
  const float yy = total_of(partials + LB_BLOCKS, s_red);
  if (threadIdx.x == 0) {
    scal[0] = ys;
    scal[1] = yy;
    scal[2] = 1.0f / ys;
    scal[3] = ys / yy;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" size_t pcfa_lbfgs_workspace_floats(void) { return 2 * LB_BLOCKS; }

extern "C" int pcfa_lbfgs_pair(const float* g, float* g_prev, const float* d, float t, float* y_out, float* s_out,
                               float* scal4, float* workspace, int update_prev, long long n, void* stream) {
  if (!g || !g_prev || !d || !y_out || !s_out || !scal4 || !workspace || n < 1) return PCFA_ERR_INVALID_ARG;
  if (!aligned16(g) || !aligned16(g_prev) || !aligned16(d) || !aligned16(y_out) || !aligned16(s_out))
    return PCFA_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  pcfa_launch(lbfgs_pair_kernel, dim3(LB_BLOCKS), dim3(LB_THREADS), 0, s, g, g_prev, d, t, y_out, s_out, workspace,
              update_prev, n);
  PCFA_LAUNCH_CHECK();
  pcfa_launch(lbfgs_pair_final_kernel, dim3(1), dim3(LB_THREADS), 0, s, (const float*)workspace, scal4);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_lbfgs_direction(const float* g, const float* S, const float* Y, const float* ro, const float* H,
                                    float* al, float* d, float* workspace, int first, int count, int capacity,
                                    long long ld, long long n, void* stream) {
  if (!g || !S || !Y || !ro || !H || !al || !d || !workspace || count < 1 || capacity < count || first < 0 ||
      first >= capacity || n < 1 || ld < n || (ld & 3))
    return PCFA_ERR_INVALID_ARG;
  if (!aligned16(g) || !aligned16(S) || !aligned16(Y) || !aligned16(d)) return PCFA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(LB_BLOCKS), block(LB_THREADS);
  auto slot = [&](int k) { return (long long)((first + k) % capacity); };  // k-th oldest pair
  auto Sk = [&](int k) { return S + slot(k) * ld; };
  auto Yk = [&](int k) { return Y + slot(k) * ld; };
  float* part[2] = {workspace, workspace + LB_BLOCKS};
  int cur = 0;
  const int m = count;
  const float* none = nullptr;
  pcfa_launch(lbfgs_step_kernel<LB_FIRST>, grid, block, 0, st, d, g, none, Sk(m - 1), none, part[cur], none,
              (float*)nullptr, none, n);
  PCFA_LAUNCH_CHECK();
  for (int i = m - 1; i >= 1; --i) {
    pcfa_launch(lbfgs_step_kernel<LB_LOOP1>, grid, block, 0, st, d, none, Yk(i), Sk(i - 1), (const float*)part[cur],
                part[cur ^ 1], ro + slot(i), al + i, none, n);
    PCFA_LAUNCH_CHECK();
    cur ^= 1;
  }
  pcfa_launch(lbfgs_step_kernel<LB_TURN>, grid, block, 0, st, d, none, Yk(0), Yk(0), (const float*)part[cur],
              part[cur ^ 1], ro + slot(0), al + 0, H, n);
  PCFA_LAUNCH_CHECK();
  cur ^= 1;
  for (int i = 0; i + 1 < m; ++i) {
    pcfa_launch(lbfgs_step_kernel<LB_LOOP2>, grid, block, 0, st, d, none, Sk(i), Yk(i + 1), (const float*)part[cur],
                part[cur ^ 1], ro + slot(i), al + i, none, n);
    PCFA_LAUNCH_CHECK();
    cur ^= 1;
  }
  pcfa_launch(lbfgs_step_kernel<LB_LAST>, grid, block, 0, st, d, none, Sk(m - 1), none, (const float*)part[cur],
              (float*)nullptr, ro + slot(m - 1), al + (m - 1), none, n);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
