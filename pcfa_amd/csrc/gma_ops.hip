// GMA attention: row softmax of the [N x N] similarity matrix, forward and backward, for gfx950.
//
// Replaces `sim.softmax(dim=-1)` of models/gma/gma.py:34-77 (Attention.forward) and its autograd backward
// (SURVEY 8f row f1).  The matrix is 7040 x 7040 fp32 = 198 MB at 436x1024: the library runs three passes over it in
// the forward (max, sum, normalise) and reads three tensors in the backward; here a row (28 KB) lives in the registers
// of one workgroup: ONE read and ONE write of the matrix per direction, in place if the caller wishes.
//   forward : y = exp(x - max_j x) / sum_j exp(x - max_j x)
//   backward: gx = y * (gy - sum_j gy_j y_j)
// Rows of up to 256 threads x 8 x 4 = 8192 columns stay in registers; longer rows take a three-sweep fallback.
#include "common.hpp"

namespace {

constexpr int NT = 256, NV = 8;   // threads per row, float4 per thread held in registers

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float u = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, u) : v + u;
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();                       // red[] may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];   // fixed order: deterministic
  return r;
}

template <bool BWD>
__global__ __launch_bounds__(NT) void softmax_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, long long rows, int cols,
                                                          int vec_ok) {
  __shared__ float red[NT / 64];
  const long long row = blockIdx.x;
  const float* pa = a + row * cols;                       // forward: x; backward: y
  const float* pb = BWD ? b + row * cols : nullptr;       // backward: gy
  float* po = out + row * cols;
  const int nv4 = cols >> 2;
  if (vec_ok && nv4 <= NT * NV && (cols & 3) == 0) {
    float4 va[NV], vb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = threadIdx.x + NT * i;
      const bool ok = c < nv4;
      va[i] = ok ? reinterpret_cast<const float4*>(pa)[c]
                 : (BWD ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY));
      if (BWD) vb[i] = ok ? reinterpret_cast<const float4*>(pb)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!BWD) {
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < NV; ++i) m = fmaxf(fmaxf(fmaxf(m, va[i].x), fmaxf(va[i].y, va[i].z)), va[i].w);
      m = block_reduce(m, red, true);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        va[i].x = expf(va[i].x - m); va[i].y = expf(va[i].y - m);
        va[i].z = expf(va[i].z - m); va[i].w = expf(va[i].w - m);
        sum += (va[i].x + va[i].y) + (va[i].z + va[i].w);
      }
      sum = block_reduce(sum, red, false);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = threadIdx.x + NT * i;
        if (c < nv4)
          reinterpret_cast<float4*>(po)[c] = make_float4(va[i].x * inv, va[i].y * inv, va[i].z * inv, va[i].w * inv);
      }
    } else {
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        dot += (va[i].x * vb[i].x + va[i].y * vb[i].y) + (va[i].z * vb[i].z + va[i].w * vb[i].w);
      dot = block_reduce(dot, red, false);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = threadIdx.x + NT * i;
        if (c < nv4)
          reinterpret_cast<float4*>(po)[c] = make_float4(va[i].x * (vb[i].x - dot), va[i].y * (vb[i].y - dot),
                                                         va[i].z * (vb[i].z - dot), va[i].w * (vb[i].w - dot));
      }
    }
    return;
  }
  // generic fallback: three sweeps over the row
  if (!BWD) {
    float m = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += NT) m = fmaxf(m, pa[c]);
    m = block_reduce(m, red, true);
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += NT) sum += expf(pa[c] - m);
    sum = block_reduce(sum, red, false);
    const float inv = 1.0f / sum;
    for (int c = threadIdx.x; c < cols; c += NT) po[c] = expf(pa[c] - m) * inv;
  } else {
    float dot = 0.f;
    for (int c = threadIdx.x; c < cols; c += NT) dot += pa[c] * pb[c];
    dot = block_reduce(dot, red, false);
    for (int c = threadIdx.x; c < cols; c += NT) po[c] = pa[c] * (pb[c] - dot);
  }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pcfa_softmax_rows_fwd(const float* x, float* y, long long rows, int cols, void* stream) {
  if (!x || !y || rows < 1 || cols < 1 || rows > 0x7fffffffLL) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(x) && al16(y) && (cols & 3) == 0;
  pcfa_launch(softmax_rows_kernel<false>, dim3((unsigned)rows), dim3(NT), 0, (hipStream_t)stream, x,
              (const float*)nullptr, y, rows, cols, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_softmax_rows_bwd(const float* y, const float* grad_y, float* grad_x, long long rows, int cols,
                                     void* stream) {
  if (!y || !grad_y || !grad_x || rows < 1 || cols < 1 || rows > 0x7fffffffLL) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(y) && al16(grad_y) && al16(grad_x) && (cols & 3) == 0;
  pcfa_launch(softmax_rows_kernel<true>, dim3((unsigned)rows), dim3(NT), 0, (hipStream_t)stream, y, grad_y, grad_x,
              rows, cols, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
