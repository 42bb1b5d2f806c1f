// Fused gate arithmetic of RAFT/GMA's SepConvGRU for gfx950.
//
// Replaces the elementwise chain between the gate convolutions (reference
// models/raft/update.py:45-60, identical models/gma/update.py:51-66):
//     z = sigmoid(convz(hx));  r = sigmoid(convr(hx));  q = tanh(convq(cat[r*h, x]))
//     h = (1 - z) * h + z * q
// torch runs it as ~10 forward and ~18 backward launches per GRU half-step, each a few
// microseconds of launch floor on 3.6 MB tensors; here it is two streaming kernels forward
// and two backward (16-B accesses, grid-stride, HBM-bound).
#include "common.hpp"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename F>
__device__ __forceinline__ void for_each4(long long n, F f) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = gid; i < n4; i += stride) f(i, true);
  // tail (n not a multiple of 4): one scalar element per thread of the first block
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) f((n4 << 2) + threadIdx.x, false);
}

#define LD4(p, i) (*reinterpret_cast<const float4*>((p) + ((i) << 2)))
#define ST4(p, i, v) (*reinterpret_cast<float4*>((p) + ((i) << 2)) = (v))

__global__ void gru_gates_fwd_kernel(const float* __restrict__ zc, const float* __restrict__ rc,
                                     const float* __restrict__ h, float* __restrict__ z,
                                     float* __restrict__ r, float* __restrict__ rh, long long n) {
  for_each4(n, [&](long long i, bool vec) {
    if (vec) {
      const float4 a = LD4(zc, i), b = LD4(rc, i), hh = LD4(h, i);
      float4 zz, rr, o;
      zz.x = sigmoidf_(a.x); zz.y = sigmoidf_(a.y); zz.z = sigmoidf_(a.z); zz.w = sigmoidf_(a.w);
      rr.x = sigmoidf_(b.x); rr.y = sigmoidf_(b.y); rr.z = sigmoidf_(b.z); rr.w = sigmoidf_(b.w);
      o.x = rr.x * hh.x; o.y = rr.y * hh.y; o.z = rr.z * hh.z; o.w = rr.w * hh.w;
      ST4(z, i, zz); ST4(r, i, rr); ST4(rh, i, o);
    } else {
      const float zz = sigmoidf_(zc[i]), rr = sigmoidf_(rc[i]);
      z[i] = zz; r[i] = rr; rh[i] = rr * h[i];
    }
  });
}

// dzc = dz * (1 - z) * z ; dr = drh * h ; drc = dr * (1 - r) * r ; dh = drh * r
__global__ void gru_gates_bwd_kernel(const float* __restrict__ z, const float* __restrict__ r,
                                     const float* __restrict__ h, const float* __restrict__ dz,
                                     const float* __restrict__ drh, float* __restrict__ dzc,
                                     float* __restrict__ drc, float* __restrict__ dh, long long n) {
  auto one = [](float z_, float r_, float h_, float dz_, float drh_, float& a, float& b, float& c) {
    a = dz_ * (1.f - z_) * z_;
    const float dr = drh_ * h_;
    b = dr * (1.f - r_) * r_;
    c = drh_ * r_;
  };
  for_each4(n, [&](long long i, bool vec) {
    if (vec) {
      const float4 zz = LD4(z, i), rr = LD4(r, i), hh = LD4(h, i), gz = LD4(dz, i), gr = LD4(drh, i);
      float4 a, b, c;
      one(zz.x, rr.x, hh.x, gz.x, gr.x, a.x, b.x, c.x);
      one(zz.y, rr.y, hh.y, gz.y, gr.y, a.y, b.y, c.y);
      one(zz.z, rr.z, hh.z, gz.z, gr.z, a.z, b.z, c.z);
      one(zz.w, rr.w, hh.w, gz.w, gr.w, a.w, b.w, c.w);
      ST4(dzc, i, a); ST4(drc, i, b); ST4(dh, i, c);
    } else {
      one(z[i], r[i], h[i], dz[i], drh[i], dzc[i], drc[i], dh[i]);
    }
  });
}

__global__ void gru_update_fwd_kernel(const float* __restrict__ z, const float* __restrict__ qc,
                                      const float* __restrict__ h, float* __restrict__ q,
                                      float* __restrict__ hnew, long long n) {
  for_each4(n, [&](long long i, bool vec) {
    if (vec) {
      const float4 zz = LD4(z, i), c = LD4(qc, i), hh = LD4(h, i);
      float4 qq, o;
      qq.x = tanhf(c.x); qq.y = tanhf(c.y); qq.z = tanhf(c.z); qq.w = tanhf(c.w);
      o.x = (1.f - zz.x) * hh.x + zz.x * qq.x;
      o.y = (1.f - zz.y) * hh.y + zz.y * qq.y;
      o.z = (1.f - zz.z) * hh.z + zz.z * qq.z;
      o.w = (1.f - zz.w) * hh.w + zz.w * qq.w;
      ST4(q, i, qq); ST4(hnew, i, o);
    } else {
      const float qq = tanhf(qc[i]);
      q[i] = qq;
      hnew[i] = (1.f - z[i]) * h[i] + z[i] * qq;
    }
  });
}

// dz = g*q - g*h ; dqc = (g*z) * (1 - q*q) ; dh = g * (1 - z)
__global__ void gru_update_bwd_kernel(const float* __restrict__ z, const float* __restrict__ q,
                                      const float* __restrict__ h, const float* __restrict__ g,
                                      float* __restrict__ dz, float* __restrict__ dqc,
                                      float* __restrict__ dh, long long n) {
  auto one = [](float z_, float q_, float h_, float g_, float& a, float& b, float& c) {
    a = g_ * q_ - g_ * h_;
    b = (g_ * z_) * (1.f - q_ * q_);
    c = g_ * (1.f - z_);
  };
  for_each4(n, [&](long long i, bool vec) {
    if (vec) {
      const float4 zz = LD4(z, i), qq = LD4(q, i), hh = LD4(h, i), gg = LD4(g, i);
      float4 a, b, c;
      one(zz.x, qq.x, hh.x, gg.x, a.x, b.x, c.x);
      one(zz.y, qq.y, hh.y, gg.y, a.y, b.y, c.y);
      one(zz.z, qq.z, hh.z, gg.z, a.z, b.z, c.z);
      one(zz.w, qq.w, hh.w, gg.w, a.w, b.w, c.w);
      ST4(dz, i, a); ST4(dqc, i, b); ST4(dh, i, c);
    } else {
      one(z[i], q[i], h[i], g[i], dz[i], dqc[i], dh[i]);
    }
  });
}

inline int blocks_for(long long n) {
  long long b = ((n >> 2) + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int pcfa_gru_gates_fwd(const float* zc, const float* rc, const float* h, float* z, float* r,
                                  float* rh, long long n, void* stream) {
  if (!zc || !rc || !h || !z || !r || !rh || n < 1) return PCFA_ERR_INVALID_ARG;
  if (!(al16(zc) && al16(rc) && al16(h) && al16(z) && al16(r) && al16(rh))) return PCFA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, zc,
                     rc, h, z, r, rh, n);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_gru_gates_bwd(const float* z, const float* r, const float* h, const float* dz,
                                  const float* drh, float* dzc, float* drc, float* dh, long long n,
                                  void* stream) {
  if (!z || !r || !h || !dz || !drh || !dzc || !drc || !dh || n < 1) return PCFA_ERR_INVALID_ARG;
  if (!(al16(z) && al16(r) && al16(h) && al16(dz) && al16(drh) && al16(dzc) && al16(drc) && al16(dh)))
    return PCFA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, r,
                     h, dz, drh, dzc, drc, dh, n);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_gru_update_fwd(const float* z, const float* qc, const float* h, float* q,
                                   float* hnew, long long n, void* stream) {
  if (!z || !qc || !h || !q || !hnew || n < 1) return PCFA_ERR_INVALID_ARG;
  if (!(al16(z) && al16(qc) && al16(h) && al16(q) && al16(hnew))) return PCFA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gru_update_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, qc,
                     h, q, hnew, n);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_gru_update_bwd(const float* z, const float* q, const float* h, const float* g,
                                   float* dz, float* dqc, float* dh, long long n, void* stream) {
  if (!z || !q || !h || !g || !dz || !dqc || !dh || n < 1) return PCFA_ERR_INVALID_ARG;
  if (!(al16(z) && al16(q) && al16(h) && al16(g) && al16(dz) && al16(dqc) && al16(dh)))
    return PCFA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gru_update_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, q,
                     h, g, dz, dqc, dh, n);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}
