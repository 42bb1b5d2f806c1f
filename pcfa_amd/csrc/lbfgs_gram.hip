// L-BFGS direction in the coefficient space of the history ("Gram form"), for gfx950.
//
// Replaces, like lbfgs.hip, the memory update and the two-loop recursion of torch.optim.LBFGS.step (the optimiser
// the reference calls at attack_PCFA.py:97,114,382,388: max_iter=10, history_size=100, no line search).  lbfgs.hip
// runs the recursion as 2m+1 dependent sweeps over the 10.8 MB vector (4 passes each): at the steady-state history
// m = 100 that is 8.6 GB of HBM traffic per iteration, 1.6 ms, 16 ms of a 166 ms attack step.  The recursion is
// linear algebra in span{g, s_i, y_i}: with the inner products
//     SY[i][j] = s_i.y_j (i <= j),   YY[i][j] = y_i.y_j,   Sg[i] = s_i.g,   Yg[i] = y_i.g
// the two loops are two triangular substitutions on m numbers,
//     loop 1:  R al = -Sg            (R = upper triangle of SY; al_i = ro_i s_i.q of the recursion)
//     loop 2:  be_i = ro_i (gamma (-Yg_i - sum_j YY[i][j] al_j) + sum_{j<i} SY[j][i] (al_j - be_j))
//     d = -gamma g - gamma sum_j al_j y_j + sum_j (al_j - be_j) s_j
// so an iteration needs the history exactly TWICE: one sweep that forms the inner products of the new gradient and
// of the new pair with every stored vector (and writes the new pair), one sweep that forms d -- 4.3 GB instead of 8.6.
// The substitutions run in fp64 in one workgroup.  Everything stays on the device: the curvature test y.s > 1e-10 of
// the optimiser is taken by the coefficient kernel, which commits the candidate pair to the ring itself.
//
// Rounding: not the optimiser's operation order (a dot product against the running vector q becomes a combination of
// stored inner products), so iterates differ from torch.optim.LBFGS in the last bits of d; the inner products are
// fp32 products summed in fp32 over 8 elements per lane and a wave, then in fp64 over the workgroups in index order
// (deterministic, no atomics).  tests/test_gpu_parity.py::test_lbfgs_matches_torch_optimizer states the tolerance.
#include "common.hpp"

namespace {

constexpr int GR_THREADS = 256;
constexpr int GR_WAVES = GR_THREADS / 64;
constexpr int GR_CHUNK4 = 2 * GR_THREADS;  // float4 groups per workgroup: 2 per thread = 8 floats per lane
constexpr int GR_MAX_ROWS = 129;           // capacity + 1 (the coefficient kernel keeps an m x m fp64 matrix in LDS)

// ---- device-resident optimiser state ----------------------------------------------------------------------------
struct GramHeader {
  int first, count, accepted, rows;
  float H, cg, ys, yy;
  float gtd, dmax, pad0, pad1;
};
__host__ __device__ inline size_t gs_align(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline size_t gs_off_cS(int rows) { return gs_align(sizeof(GramHeader)); }
__host__ __device__ inline size_t gs_off_cY(int rows) { return gs_off_cS(rows) + gs_align((size_t)rows * 4); }
__host__ __device__ inline size_t gs_off_red(int rows) { return gs_off_cY(rows) + gs_align((size_t)rows * 4); }
__host__ __device__ inline size_t gs_off_SY(int rows) { return gs_off_red(rows) + gs_align((size_t)rows * 4 * 8); }
__host__ __device__ inline size_t gs_off_YY(int rows) { return gs_off_SY(rows) + gs_align((size_t)rows * rows * 8); }
__host__ __device__ inline size_t gs_bytes(int rows) { return gs_off_YY(rows) + gs_align((size_t)rows * rows * 8); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(x);
}
// Sum over the 64 lanes; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xb1, 0xf>(v);   // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e, 0xf>(v);   // quad_perm:[2,3,0,1]
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  return fmaf(a.w, b.w, acc);
}
__device__ __forceinline__ void axpy4(float c, const float4& a, float4& x) {
  x.x = fmaf(c, a.x, x.x);
  x.y = fmaf(c, a.y, x.y);
  x.z = fmaf(c, a.z, x.z);
  x.w = fmaf(c, a.w, x.w);
}

// ---- sweep 1: new pair + inner products ---------------------------------------------------------------------------
// y = g - g_prev, s = t d go to the candidate row (first + count) % rows, g_prev = g; for every live row r and the
// candidate: partial[(r*4 + k) * nblk + block], k = {S_r.g, S_r.y, Y_r.g, Y_r.y}.
__global__ __launch_bounds__(GR_THREADS) void gram_pass_kernel(
    const float* __restrict__ g, float* __restrict__ g_prev, const float* __restrict__ d, float t,
    float* __restrict__ S, float* __restrict__ Y, const GramHeader* __restrict__ hdr, float* __restrict__ partial,
    int rows, long long ld4, int nblk) {
  extern __shared__ float4 s_acc[];  // [GR_WAVES][rows]
  const int first = hdr->first, count = hdr->count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long i0 = (long long)blockIdx.x * GR_CHUNK4 + threadIdx.x, i1 = i0 + GR_THREADS;
  const bool ok0 = i0 < ld4, ok1 = i1 < ld4;
  const long long c0 = ok0 ? i0 : ld4 - 1, c1 = ok1 ? i1 : ld4 - 1;  // clamped: loads are unconditional
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* p4 = reinterpret_cast<float4*>(g_prev);
  const float4* d4 = reinterpret_cast<const float4*>(d);
  float4 ga = g4[c0], gb = g4[c1];
  const float4 pa = p4[c0], pb = p4[c1];
  const float4 da = d4[c0], db = d4[c1];
  float4 ya = make_float4(ga.x - pa.x, ga.y - pa.y, ga.z - pa.z, ga.w - pa.w);
  float4 yb = make_float4(gb.x - pb.x, gb.y - pb.y, gb.z - pb.z, gb.w - pb.w);
  float4 sa = make_float4(da.x * t, da.y * t, da.z * t, da.w * t);
  float4 sb = make_float4(db.x * t, db.y * t, db.z * t, db.w * t);
  const int crow = (first + count) % rows;
  const long long ld = ld4;  // row pitch in float4
  float4* Sc = reinterpret_cast<float4*>(S) + (long long)crow * ld;
  float4* Yc = reinterpret_cast<float4*>(Y) + (long long)crow * ld;
  if (ok0) { Sc[i0] = sa; Yc[i0] = ya; p4[i0] = ga; }
  if (ok1) { Sc[i1] = sb; Yc[i1] = yb; p4[i1] = gb; }
  if (!ok0) { ga = z4; ya = z4; sa = z4; }   // lanes past the end contribute nothing
  if (!ok1) { gb = z4; yb = z4; sb = z4; }

  for (int k = 0; k < count; ++k) {
    int r = first + k;
    if (r >= rows) r -= rows;
    const float4* Sr = reinterpret_cast<const float4*>(S) + (long long)r * ld;
    const float4* Yr = reinterpret_cast<const float4*>(Y) + (long long)r * ld;
    const float4 s0 = Sr[c0], s1 = Sr[c1], y0 = Yr[c0], y1 = Yr[c1];
    float a0 = dot4(s1, gb, dot4(s0, ga, 0.f));
    float a1 = dot4(s1, yb, dot4(s0, ya, 0.f));
    float a2 = dot4(y1, gb, dot4(y0, ga, 0.f));
    float a3 = dot4(y1, yb, dot4(y0, ya, 0.f));
    a0 = wave_sum_lane63(a0);
    a1 = wave_sum_lane63(a1);
    a2 = wave_sum_lane63(a2);
    a3 = wave_sum_lane63(a3);
    if (lane == 63) s_acc[wave * rows + r] = make_float4(a0, a1, a2, a3);
  }
  {
    float a0 = wave_sum_lane63(dot4(sb, gb, dot4(sa, ga, 0.f)));
    float a1 = wave_sum_lane63(dot4(sb, yb, dot4(sa, ya, 0.f)));
    float a2 = wave_sum_lane63(dot4(yb, gb, dot4(ya, ga, 0.f)));
    float a3 = wave_sum_lane63(dot4(yb, yb, dot4(ya, ya, 0.f)));
    if (lane == 63) s_acc[wave * rows + crow] = make_float4(a0, a1, a2, a3);
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= count; k += GR_THREADS) {
    int r = first + k;
    if (r >= rows) r -= rows;
    float4 v = s_acc[r];
#pragma unroll
    for (int w = 1; w < GR_WAVES; ++w) {
      const float4 u = s_acc[w * rows + r];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    float* p = partial + (long long)(r * 4) * nblk + blockIdx.x;
    p[0] = v.x;
    p[nblk] = v.y;
    p[2LL * nblk] = v.z;
    p[3LL * nblk] = v.w;
  }
}

__device__ __forceinline__ double wave_sum_all_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// red[r*4 + k] = sum over the workgroups of sweep 1, in index order per lane, fp64.  One workgroup per ring row,
// one wave per kind.
__global__ __launch_bounds__(GR_THREADS) void gram_reduce_kernel(const float* __restrict__ partial,
                                                                 const GramHeader* __restrict__ hdr,
                                                                 double* __restrict__ red, int rows, int nblk) {
  const int r = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int k = r - hdr->first;
  if (k < 0) k += rows;
  double v = 0.0;
  if (k <= hdr->count) {
    const float* p = partial + (long long)(r * 4 + wave) * nblk;
    for (int b = lane; b < nblk; b += 64) v += (double)p[b];
    v = wave_sum_all_f64(v);
  }
  if (lane == 0) red[r * 4 + wave] = v;
}

// ---- coefficients ---------------------------------------------------------------------------------------------------
// Takes the optimiser's curvature decision, commits the candidate pair and solves the two substitutions.
// One workgroup of 1024 threads: all of them stage the matrix and form the matrix-vector product, wave 0 runs the two
// sequential substitutions with the running right-hand side in registers (lane k owns rows k and k + 64) and the pivot
// broadcast through v_readlane -- one step is a readlane pair, a multiply and two FMAs on LDS operands that were
// requested a step earlier.
constexpr int GC_THREADS = 1024;

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__global__ __launch_bounds__(GC_THREADS) void gram_coeff_kernel(unsigned char* __restrict__ state, int rows, int cap,
                                                                int have_candidate) {
  extern __shared__ double s_mem[];
  GramHeader* hdr = reinterpret_cast<GramHeader*>(state);
  float* cS = reinterpret_cast<float*>(state + gs_off_cS(rows));
  float* cY = reinterpret_cast<float*>(state + gs_off_cY(rows));
  const double* red = reinterpret_cast<const double*>(state + gs_off_red(rows));
  double* SY = reinterpret_cast<double*>(state + gs_off_SY(rows));
  double* YY = reinterpret_cast<double*>(state + gs_off_YY(rows));
  __shared__ int s_first, s_count;
  __shared__ float s_H;
  const int tid = threadIdx.x;
  int first = hdr->first, count = hdr->count;
  if (have_candidate) {
    const int c = (first + count) % rows;
    const double ys = red[c * 4 + 1], yy = red[c * 4 + 3];
    const bool accept = (double)(float)ys > 1e-10;   // torch: `if ys > 1e-10` on the fp32 dot product
    if (accept) {
      // inner products of the new pair with every stored vector: column c of SY, row/column c of YY
      for (int k = tid; k < count; k += GC_THREADS) {
        int r = first + k;
        if (r >= rows) r -= rows;
        SY[r * rows + c] = red[r * 4 + 1];
        const double v = red[r * 4 + 3];
        YY[r * rows + c] = v;
        YY[c * rows + r] = v;
      }
      if (tid == 0) {
        SY[c * rows + c] = ys;
        YY[c * rows + c] = yy;
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (accept) {
        if (count == cap) first = (first + 1) % rows; else count += 1;
        hdr->H = (float)ys / (float)yy;   // H_diag = ys / y.dot(y), fp32 as in the optimiser
      }
      hdr->first = first;
      hdr->count = count;
      hdr->accepted = accept ? 1 : 0;
      hdr->ys = (float)ys;
      hdr->yy = (float)yy;
      s_first = first;
      s_count = count;
      s_H = hdr->H;
    }
  } else if (tid == 0) {
    s_first = first;
    s_count = count;
    s_H = hdr->H;
  }
  __syncthreads();
  first = s_first;
  const int m = s_count;
  const double gamma = (double)s_H;
  if (tid == 0) hdr->cg = -s_H;
  if (m == 0) return;

  // LDS: G[m][m] (upper triangle of SY in live order, row pitch m), then al, de, u, rdiag
  double* G = s_mem;
  double* al = G + (size_t)m * m;
  double* de = al + m;
  double* u = de + m;
  double* rdiag = u + m;
  auto ring = [&](int k) { int r = first + k; return r >= rows ? r - rows : r; };
  {
    constexpr int NE = (GR_MAX_ROWS * GR_MAX_ROWS + GC_THREADS - 1) / GC_THREADS;   // 17 elements per thread at most
    double v[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = tid + q * GC_THREADS;
      const int ec = e < m * m ? e : 0;
      const int i = ec / m, j = ec - i * m;
      v[q] = SY[ring(i) * rows + ring(j)];
    }
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = tid + q * GC_THREADS;
      if (e < m * m) {
        const int i = e / m, j = e - i * m;
        G[e] = (j >= i) ? v[q] : 0.0;
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < m; k += GC_THREADS) rdiag[k] = 1.0 / G[k * m + k];
  __syncthreads();

  const int lane = tid & 63;
  const int k0 = lane, k1 = lane + 64;
  const int c0 = k0 < m ? k0 : 0, c1 = k1 < m ? k1 : 0;     // clamped rows: LDS loads are unconditional
  constexpr int PF = 4;                                       // operands of a step are requested PF steps ahead
  if (tid < 64) {
    // loop 1 (newest -> oldest): R al = -Sg, column-oriented back substitution
    double r0 = k0 < m ? -red[ring(k0) * 4 + 0] : 0.0;
    double r1 = k1 < m ? -red[ring(k1) * 4 + 0] : 0.0;
    double g0[PF], g1[PF], rd[PF];
#pragma unroll
    for (int s_ = 0; s_ < PF; ++s_) {
      const int ii = m - 1 - s_ > 0 ? m - 1 - s_ : 0;
      g0[s_] = G[c0 * m + ii]; g1[s_] = G[c1 * m + ii]; rd[s_] = rdiag[ii];
    }
    for (int ib = m - 1; ib >= 0; ib -= PF) {
#pragma unroll
      for (int s_ = 0; s_ < PF; ++s_) {
        const int i = ib - s_;
        if (i >= 0) {                                          // uniform
          const double ri = (i >= 64) ? readlane_f64(r1, i - 64) : readlane_f64(r0, i);
          const double a = ri * rd[s_];
          al[i] = a;                                           // every lane, same value
          r0 = (k0 < i) ? r0 - g0[s_] * a : r0;
          r1 = (k1 < i) ? r1 - g1[s_] * a : r1;
        }
        const int in = i - PF > 0 ? i - PF : 0;
        g0[s_] = G[c0 * m + in]; g1[s_] = G[c1 * m + in]; rd[s_] = rdiag[in];
      }
    }
  }
  __syncthreads();
  // u_k = gamma * (-Yg_k - sum_j YY[k][j] al_j): 8 threads per row, every load of a thread in flight at once
  {
    const int k = tid >> 3, part = tid & 7;
    const int rk = ring(k < m ? k : 0);
    constexpr int NJ = (GR_MAX_ROWS + 7) / 8;                  // 17 >= ceil(m / 8)
    double yv[NJ];
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
      const int j = part + 8 * q;
      yv[q] = YY[rk * rows + ring(j < m ? j : 0)];
    }
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
      const int j = part + 8 * q;
      acc += (j < m) ? yv[q] * al[j < m ? j : 0] : 0.0;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (k < m && part == 0) u[k] = gamma * (-red[rk * 4 + 2] - acc);
  }
  __syncthreads();
  if (tid < 64) {
    // loop 2 (oldest -> newest): be_i = ro_i (u_i + sum_{j<i} SY[j][i] de_j), de_i = al_i - be_i
    double a0 = k0 < m ? u[k0] : 0.0;
    double a1 = k1 < m ? u[k1] : 0.0;
    double g0[PF], g1[PF], rd[PF], ali[PF];
#pragma unroll
    for (int s_ = 0; s_ < PF; ++s_) {
      const int ii = s_ < m ? s_ : m - 1;
      g0[s_] = G[ii * m + c0]; g1[s_] = G[ii * m + c1]; rd[s_] = rdiag[ii]; ali[s_] = al[ii];
    }
    for (int ib = 0; ib < m; ib += PF) {
#pragma unroll
      for (int s_ = 0; s_ < PF; ++s_) {
        const int i = ib + s_;
        if (i < m) {                                           // uniform
          const double ai = (i >= 64) ? readlane_f64(a1, i - 64) : readlane_f64(a0, i);
          const double dl = ali[s_] - ai * rd[s_];
          de[i] = dl;
          a0 = (k0 > i && k0 < m) ? a0 + g0[s_] * dl : a0;
          a1 = (k1 > i && k1 < m) ? a1 + g1[s_] * dl : a1;
        }
        const int in = i + PF < m ? i + PF : m - 1;
        g0[s_] = G[in * m + c0]; g1[s_] = G[in * m + c1]; rd[s_] = rdiag[in]; ali[s_] = al[in];
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < m; k += GC_THREADS) {
    const int rk = ring(k);
    cS[rk] = (float)de[k];
    cY[rk] = (float)(-gamma * al[k]);
  }
}

// ---- sweep 2: the direction ---------------------------------------------------------------------------------------
// d = cg g + sum over live rows (cS[r] S_r + cY[r] Y_r), newest pair first; partials of g.d and max|d|.
__global__ __launch_bounds__(GR_THREADS) void gram_direction_kernel(
    const float* __restrict__ g, const float* __restrict__ S, const float* __restrict__ Y,
    const unsigned char* __restrict__ state, float* __restrict__ d, float* __restrict__ partial, int rows,
    long long ld4, int nblk) {
  __shared__ float s_red[2 * GR_WAVES];
  const GramHeader* hdr = reinterpret_cast<const GramHeader*>(state);
  const float* cS = reinterpret_cast<const float*>(state + gs_off_cS(rows));
  const float* cY = reinterpret_cast<const float*>(state + gs_off_cY(rows));
  const int first = hdr->first, count = hdr->count;
  const float cg = hdr->cg;
  const long long i0 = (long long)blockIdx.x * GR_CHUNK4 + threadIdx.x, i1 = i0 + GR_THREADS;
  const bool ok0 = i0 < ld4, ok1 = i1 < ld4;
  const long long c0 = ok0 ? i0 : ld4 - 1, c1 = ok1 ? i1 : ld4 - 1;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4 ga = g4[c0], gb = g4[c1];
  float4 xa = make_float4(cg * ga.x, cg * ga.y, cg * ga.z, cg * ga.w);
  float4 xb = make_float4(cg * gb.x, cg * gb.y, cg * gb.z, cg * gb.w);
  // newest pair first: the rows sweep 1 read last are the ones still in the Infinity Cache
  for (int k = count - 1; k >= 0; --k) {
    int r = first + k;
    if (r >= rows) r -= rows;
    const float4* Sr = reinterpret_cast<const float4*>(S) + (long long)r * ld4;
    const float4* Yr = reinterpret_cast<const float4*>(Y) + (long long)r * ld4;
    const float4 s0 = Sr[c0], s1 = Sr[c1], y0 = Yr[c0], y1 = Yr[c1];
    const float a = cS[r], b = cY[r];
    axpy4(b, y0, xa);
    axpy4(b, y1, xb);
    axpy4(a, s0, xa);
    axpy4(a, s1, xb);
  }
  float gtd = 0.f, mx = 0.f;
  if (ok0) {
    reinterpret_cast<float4*>(d)[i0] = xa;
    gtd = dot4(ga, xa, gtd);
    mx = fmaxf(fmaxf(fabsf(xa.x), fabsf(xa.y)), fmaxf(fabsf(xa.z), fabsf(xa.w)));
  }
  if (ok1) {
    reinterpret_cast<float4*>(d)[i1] = xb;
    gtd = dot4(gb, xb, gtd);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(xb.x), fabsf(xb.y)), fmaxf(fabsf(xb.z), fabsf(xb.w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gtd += __shfl_xor(gtd, o, 64);
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_red[wave] = gtd;
    s_red[GR_WAVES + wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < GR_WAVES; ++w) {
      a += s_red[w];
      b = fmaxf(b, s_red[GR_WAVES + w]);
    }
    partial[blockIdx.x] = a;
    partial[nblk + blockIdx.x] = b;
  }
}

__global__ __launch_bounds__(GR_THREADS) void gram_direction_final_kernel(const float* __restrict__ partial,
                                                                          unsigned char* __restrict__ state,
                                                                          float* __restrict__ out2, int nblk) {
  __shared__ double s_a[GR_WAVES];
  __shared__ float s_b[GR_WAVES];
  double a = 0.0;
  float b = 0.f;
  for (int i = threadIdx.x; i < nblk; i += GR_THREADS) {
    a += (double)partial[i];
    b = fmaxf(b, partial[nblk + i]);
  }
  a = wave_sum_all_f64(a);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
  if ((threadIdx.x & 63) == 0) {
    s_a[threadIdx.x >> 6] = a;
    s_b[threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    float mx = 0.f;
#pragma unroll
    for (int w = 0; w < GR_WAVES; ++w) {
      t += s_a[w];
      mx = fmaxf(mx, s_b[w]);
    }
    GramHeader* hdr = reinterpret_cast<GramHeader*>(state);
    hdr->gtd = (float)t;
    hdr->dmax = mx;
    out2[0] = (float)t;
    out2[1] = mx;
  }
}

__global__ void gram_reset_kernel(unsigned char* __restrict__ state, int rows) {
  GramHeader* hdr = reinterpret_cast<GramHeader*>(state);
  hdr->first = 0;
  hdr->count = 0;
  hdr->accepted = 0;
  hdr->rows = rows;
  hdr->H = 1.f;
  hdr->cg = -1.f;
  hdr->ys = hdr->yy = hdr->gtd = hdr->dmax = 0.f;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int nblocks_for(long long ld) { return (int)((ld / 4 + GR_CHUNK4 - 1) / GR_CHUNK4); }

}  // namespace

extern "C" size_t pcfa_lbfgs_gram_state_bytes(int capacity) {
  if (capacity < 1 || capacity + 1 > GR_MAX_ROWS) return 0;
  return gs_bytes(capacity + 1);
}

extern "C" size_t pcfa_lbfgs_gram_workspace_bytes(int capacity, long long ld) {
  if (capacity < 1 || capacity + 1 > GR_MAX_ROWS || ld < 4 || (ld & 3)) return 0;
  const size_t nblk = (size_t)nblocks_for(ld);
  return ((size_t)(capacity + 1) * 4 * nblk + 2 * nblk) * sizeof(float);
}

extern "C" int pcfa_lbfgs_gram_reset(void* state, int capacity, void* stream) {
  if (!state || capacity < 1 || capacity + 1 > GR_MAX_ROWS) return PCFA_ERR_INVALID_ARG;
  pcfa_launch(gram_reset_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned char*)state, capacity + 1);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_lbfgs_gram_update(const float* g, float* g_prev, const float* d, float t, float* S, float* Y,
                                      void* state, void* workspace, int capacity, long long ld, void* stream) {
  if (!g || !g_prev || !d || !S || !Y || !state || !workspace || capacity < 1 || ld < 4 || (ld & 3))
    return PCFA_ERR_INVALID_ARG;
  if (capacity + 1 > GR_MAX_ROWS) return PCFA_ERR_UNSUPPORTED;
  if (!aligned16(g) || !aligned16(g_prev) || !aligned16(d) || !aligned16(S) || !aligned16(Y) || !aligned16(state))
    return PCFA_ERR_INVALID_ARG;
  const int rows = capacity + 1, nblk = nblocks_for(ld);
  hipStream_t st = (hipStream_t)stream;
  unsigned char* sb = (unsigned char*)state;
  float* partial = (float*)workspace;
  pcfa_launch(gram_pass_kernel, dim3(nblk), dim3(GR_THREADS), (size_t)GR_WAVES * rows * sizeof(float4), st, g, g_prev,
              d, t, S, Y, (const GramHeader*)sb, partial, rows, ld / 4, nblk);
  PCFA_LAUNCH_CHECK();
  pcfa_launch(gram_reduce_kernel, dim3(rows), dim3(GR_THREADS), 0, st, (const float*)partial, (const GramHeader*)sb,
              (double*)(sb + gs_off_red(rows)), rows, nblk);
  PCFA_LAUNCH_CHECK();
  const size_t lds = ((size_t)capacity * capacity + 4 * (size_t)capacity) * sizeof(double);
  static size_t granted = 0;   // the attribute only ever grows
  if (lds > granted) {
    hipError_t e = hipFuncSetAttribute((const void*)gram_coeff_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    granted = lds;
  }
  pcfa_launch(gram_coeff_kernel, dim3(1), dim3(GC_THREADS), lds, st, sb, rows, capacity, 1);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_lbfgs_gram_direction(const float* g, const float* S, const float* Y, void* state, float* d,
                                         float* out_gtd_dmax, void* workspace, int capacity, long long ld,
                                         void* stream) {
  if (!g || !S || !Y || !state || !d || !out_gtd_dmax || !workspace || capacity < 1 || ld < 4 || (ld & 3))
    return PCFA_ERR_INVALID_ARG;
  if (capacity + 1 > GR_MAX_ROWS) return PCFA_ERR_UNSUPPORTED;
  if (!aligned16(g) || !aligned16(S) || !aligned16(Y) || !aligned16(d) || !aligned16(state))
    return PCFA_ERR_INVALID_ARG;
  const int rows = capacity + 1, nblk = nblocks_for(ld);
  hipStream_t st = (hipStream_t)stream;
  unsigned char* sb = (unsigned char*)state;
  float* partial = (float*)workspace + (size_t)rows * 4 * nblk;
  pcfa_launch(gram_direction_kernel, dim3(nblk), dim3(GR_THREADS), 0, st, g, S, Y, (const unsigned char*)sb, d,
              partial, rows, ld / 4, nblk);
  PCFA_LAUNCH_CHECK();
  pcfa_launch(gram_direction_final_kernel, dim3(1), dim3(GR_THREADS), 0, st, (const float*)partial, sb, out_gtd_dmax,
              nblk);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
