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

// f(i, true) handles floats 4i..4i+3 with 16-B accesses, f(i, false) the single float i.  `vec_ok` = every
// pointer of the call is 16-B aligned (always true for the BASELINE shapes; ragged test shapes take the scalar loop).
template <typename F>
__device__ __forceinline__ void for_each4(long long n, int vec_ok, F f) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (!vec_ok) {
    for (long long i = gid; i < n; i += stride) f(i, false);
    return;
  }
  const long long n4 = n >> 2;
  for (long long i = gid; i < n4; i += stride) f(i, true);
  // tail (n not a multiple of 4): one scalar element per thread of the first block
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) f((n4 << 2) + threadIdx.x, false);
}

#define LD4(p, i) (*reinterpret_cast<const float4*>((p) + ((i) << 2)))
#define ST4(p, i, v) (*reinterpret_cast<float4*>((p) + ((i) << 2)) = (v))

// channel of flat NCHW index e (plane = H*W, C channels); bias pointers may be null
__device__ __forceinline__ float bias_at(const float* __restrict__ b, long long e, int plane, int C) {
  return b ? b[(e / plane) % C] : 0.f;
}

__device__ __forceinline__ float4 add4(float4 a, const float* __restrict__ p, long long i) {
  if (p) {
    const float4 b = LD4(p, i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  return a;
}

__global__ void gru_gates_fwd_kernel(const float* __restrict__ zc, const float* __restrict__ rc,
                                     const float* __restrict__ h, const float* __restrict__ bz,
                                     const float* __restrict__ br, const float* __restrict__ az,
                                     const float* __restrict__ ar, float* __restrict__ z,
                                     float* __restrict__ r, float* __restrict__ rh, long long n,
                                     int plane, int C, int vec_ok) {
  for_each4(n, vec_ok, [&](long long i, bool vec) {
    if (vec) {
      const float4 a = add4(LD4(zc, i), az, i), b = add4(LD4(rc, i), ar, i), hh = LD4(h, i);
      const long long e = i << 2;
      float4 zz, rr, o;
      zz.x = sigmoidf_(a.x + bias_at(bz, e, plane, C)); zz.y = sigmoidf_(a.y + bias_at(bz, e + 1, plane, C));
      zz.z = sigmoidf_(a.z + bias_at(bz, e + 2, plane, C)); zz.w = sigmoidf_(a.w + bias_at(bz, e + 3, plane, C));
      rr.x = sigmoidf_(b.x + bias_at(br, e, plane, C)); rr.y = sigmoidf_(b.y + bias_at(br, e + 1, plane, C));
      rr.z = sigmoidf_(b.z + bias_at(br, e + 2, plane, C)); rr.w = sigmoidf_(b.w + bias_at(br, e + 3, plane, C));
      o.x = rr.x * hh.x; o.y = rr.y * hh.y; o.z = rr.z * hh.z; o.w = rr.w * hh.w;
      ST4(z, i, zz); ST4(r, i, rr); ST4(rh, i, o);
    } else {
      const float zz = sigmoidf_(zc[i] + (az ? az[i] : 0.f) + bias_at(bz, i, plane, C));
      const float rr = sigmoidf_(rc[i] + (ar ? ar[i] : 0.f) + bias_at(br, i, plane, C));
      z[i] = zz; r[i] = rr; rh[i] = rr * h[i];
    }
  });
}

// dzc = dz * (1 - z) * z ; dr = drh * h ; drc = dr * (1 - r) * r ; dh = drh * r
__global__ void gru_gates_bwd_kernel(const float* __restrict__ z, const float* __restrict__ r,
                                     const float* __restrict__ h, const float* __restrict__ dz,
                                     const float* __restrict__ drh, const float* dh_in,  // dh_in may alias dh
                                     float* __restrict__ dzc,
                                     float* __restrict__ drc, float* dh, long long n, int vec_ok) {
  // dh_in (nullable): gradient that already reached h on another path; added here instead of by a separate kernel
  auto one = [](float z_, float r_, float h_, float dz_, float drh_, float& a, float& b, float& c) {
    a = dz_ * (1.f - z_) * z_;
    const float dr = drh_ * h_;
    b = dr * (1.f - r_) * r_;
    c = drh_ * r_;
  };
  for_each4(n, vec_ok, [&](long long i, bool vec) {
    if (vec) {
      const float4 zz = LD4(z, i), rr = LD4(r, i), hh = LD4(h, i), gz = LD4(dz, i), gr = LD4(drh, i);
      float4 a, b, c;
      one(zz.x, rr.x, hh.x, gz.x, gr.x, a.x, b.x, c.x);
      one(zz.y, rr.y, hh.y, gz.y, gr.y, a.y, b.y, c.y);
      one(zz.z, rr.z, hh.z, gz.z, gr.z, a.z, b.z, c.z);
      one(zz.w, rr.w, hh.w, gz.w, gr.w, a.w, b.w, c.w);
      if (dh_in) {
        const float4 p = LD4(dh_in, i);
        c.x += p.x; c.y += p.y; c.z += p.z; c.w += p.w;
      }
      ST4(dzc, i, a); ST4(drc, i, b); ST4(dh, i, c);
    } else {
      float c;
      one(z[i], r[i], h[i], dz[i], drh[i], dzc[i], drc[i], c);
      dh[i] = dh_in ? c + dh_in[i] : c;
    }
  });
}

__global__ void gru_update_fwd_kernel(const float* __restrict__ z, const float* __restrict__ qc,
                                      const float* __restrict__ h, const float* __restrict__ bq,
                                      const float* __restrict__ aq, float* __restrict__ q,
                                      float* __restrict__ hnew, long long n, int plane, int C, int vec_ok) {
  for_each4(n, vec_ok, [&](long long i, bool vec) {
    if (vec) {
      const float4 zz = LD4(z, i), c = add4(LD4(qc, i), aq, i), hh = LD4(h, i);
      const long long e = i << 2;
      float4 qq, o;
      qq.x = tanhf(c.x + bias_at(bq, e, plane, C)); qq.y = tanhf(c.y + bias_at(bq, e + 1, plane, C));
      qq.z = tanhf(c.z + bias_at(bq, e + 2, plane, C)); qq.w = tanhf(c.w + bias_at(bq, e + 3, plane, C));
      o.x = (1.f - zz.x) * hh.x + zz.x * qq.x;
      o.y = (1.f - zz.y) * hh.y + zz.y * qq.y;
      o.z = (1.f - zz.z) * hh.z + zz.z * qq.z;
      o.w = (1.f - zz.w) * hh.w + zz.w * qq.w;
      ST4(q, i, qq); ST4(hnew, i, o);
    } else {
      const float qq = tanhf(qc[i] + (aq ? aq[i] : 0.f) + bias_at(bq, i, plane, C));
      q[i] = qq;
      hnew[i] = (1.f - z[i]) * h[i] + z[i] * qq;
    }
  });
}

// dz = g*q - g*h ; dqc = (g*z) * (1 - q*q) ; dh = g * (1 - z)
__global__ void gru_update_bwd_kernel(const float* __restrict__ z, const float* __restrict__ q,
                                      const float* __restrict__ h, const float* __restrict__ g,
                                      float* __restrict__ dz, float* __restrict__ dqc,
                                      float* __restrict__ dh, long long n, int vec_ok) {
  auto one = [](float z_, float q_, float h_, float g_, float& a, float& b, float& c) {
    a = g_ * q_ - g_ * h_;
    b = (g_ * z_) * (1.f - q_ * q_);
    c = g_ * (1.f - z_);
  };
  for_each4(n, vec_ok, [&](long long i, bool vec) {
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

// out = relu(x + bias[c])  /  gx = gout * (out > 0)   (conv bias add + ReLU of the update block in one pass)
__global__ void bias_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                     float* __restrict__ out, long long n, int plane, int C, int vec_ok) {
  for_each4(n, vec_ok, [&](long long i, bool vec) {
    if (vec) {
      const float4 a = LD4(x, i);
      const long long e = i << 2;
      float4 o;
      o.x = fmaxf(a.x + bias_at(bias, e, plane, C), 0.f);
      o.y = fmaxf(a.y + bias_at(bias, e + 1, plane, C), 0.f);
      o.z = fmaxf(a.z + bias_at(bias, e + 2, plane, C), 0.f);
      o.w = fmaxf(a.w + bias_at(bias, e + 3, plane, C), 0.f);
      ST4(out, i, o);
    } else {
      out[i] = fmaxf(x[i] + bias_at(bias, i, plane, C), 0.f);
    }
  });
}

__global__ void relu_bwd_kernel(const float* __restrict__ out, const float* __restrict__ gout,
                                float* __restrict__ gx, long long n, int vec_ok) {
  for_each4(n, vec_ok, [&](long long i, bool vec) {
    if (vec) {
      const float4 o = LD4(out, i), g = LD4(gout, i);
      float4 r;
      r.x = o.x > 0.f ? g.x : 0.f; r.y = o.y > 0.f ? g.y : 0.f;
      r.z = o.z > 0.f ? g.z : 0.f; r.w = o.w > 0.f ? g.w : 0.f;
      ST4(gx, i, r);
    } else {
      gx[i] = out[i] > 0.f ? gout[i] : 0.f;
    }
  });
}

// gx = gout * [out > 0] ; gx2 = gx * [out2 > 0]: the backward of relu(a + relu(.)) for both of its consumers in one pass
// (ResidualBlock's output ReLU and the ReLU of its second convolution, models/raft/extractor.py:50-58)
__global__ void relu_bwd2_kernel(const float* __restrict__ out, const float* __restrict__ out2,
                                 const float* __restrict__ gout, float* __restrict__ gx, float* __restrict__ gx2,
                                 long long n, int vec_ok) {
  for_each4(n, vec_ok, [&](long long i, bool vec) {
    if (vec) {
      const float4 o = LD4(out, i), p = LD4(out2, i), g = LD4(gout, i);
      float4 r, t;
      r.x = o.x > 0.f ? g.x : 0.f; r.y = o.y > 0.f ? g.y : 0.f;
      r.z = o.z > 0.f ? g.z : 0.f; r.w = o.w > 0.f ? g.w : 0.f;
      t.x = p.x > 0.f ? r.x : 0.f; t.y = p.y > 0.f ? r.y : 0.f;
      t.z = p.z > 0.f ? r.z : 0.f; t.w = p.w > 0.f ? r.w : 0.f;
      ST4(gx, i, r); ST4(gx2, i, t);
    } else {
      const float r = out[i] > 0.f ? gout[i] : 0.f;
      gx[i] = r;
      gx2[i] = out2[i] > 0.f ? r : 0.f;
    }
  });
}

inline int blocks_for(long long n) {
  long long b = ((n >> 2) + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }  // nullptr counts as aligned

}  // namespace

extern "C" int pcfa_gru_gates_fwd(const float* zc, const float* rc, const float* h, const float* bias_z,
                                  const float* bias_r, const float* add_z, const float* add_r, float* z,
                                  float* r, float* rh, long long n, int plane, int channels,
                                  void* stream) {
  if (!zc || !rc || !h || !z || !r || !rh || n < 1 || plane < 1 || channels < 1) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(zc) && al16(rc) && al16(h) && al16(z) && al16(r) && al16(rh) && al16(add_z) && al16(add_r);
  pcfa_launch(gru_gates_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, zc,
                     rc, h, bias_z, bias_r, add_z, add_r, z, r, rh, n, plane, channels, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_gru_gates_bwd_acc(const float* z, const float* r, const float* h, const float* dz,
                                      const float* drh, const float* dh_in, float* dzc, float* drc, float* dh,
                                      long long n, void* stream) {
  if (!z || !r || !h || !dz || !drh || !dzc || !drc || !dh || n < 1) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(z) && al16(r) && al16(h) && al16(dz) && al16(drh) && al16(dzc) && al16(drc) && al16(dh) &&
                     al16(dh_in);
  pcfa_launch(gru_gates_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, r,
                     h, dz, drh, dh_in, dzc, drc, dh, n, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_gru_gates_bwd(const float* z, const float* r, const float* h, const float* dz,
                                  const float* drh, float* dzc, float* drc, float* dh, long long n,
                                  void* stream) {
  return pcfa_gru_gates_bwd_acc(z, r, h, dz, drh, nullptr, dzc, drc, dh, n, stream);
}

extern "C" int pcfa_gru_update_fwd(const float* z, const float* qc, const float* h, const float* bias_q,
                                   const float* add_q, float* q, float* hnew, long long n, int plane,
                                   int channels, void* stream) {
  if (!z || !qc || !h || !q || !hnew || n < 1 || plane < 1 || channels < 1) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(z) && al16(qc) && al16(h) && al16(q) && al16(hnew) && al16(add_q);
  pcfa_launch(gru_update_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, qc,
                     h, bias_q, add_q, q, hnew, n, plane, channels, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_gru_update_bwd(const float* z, const float* q, const float* h, const float* g,
                                   float* dz, float* dqc, float* dh, long long n, void* stream) {
  if (!z || !q || !h || !g || !dz || !dqc || !dh || n < 1) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(z) && al16(q) && al16(h) && al16(g) && al16(dz) && al16(dqc) && al16(dh);
  pcfa_launch(gru_update_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, z, q,
                     h, g, dz, dqc, dh, n, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_bias_relu_fwd(const float* x, const float* bias, float* out, long long n, int plane,
                                  int channels, void* stream) {
  if (!x || !out || n < 1 || plane < 1 || channels < 1) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(x) && al16(out);
  pcfa_launch(bias_relu_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, bias,
                     out, n, plane, channels, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_relu_bwd(const float* out, const float* grad_out, float* grad_x, long long n,
                             void* stream) {
  if (!out || !grad_out || !grad_x || n < 1) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(out) && al16(grad_out) && al16(grad_x);
  pcfa_launch(relu_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, out, grad_out,
                     grad_x, n, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_relu_bwd2(const float* out, const float* out2, const float* grad_out, float* grad_x, float* grad_x2,
                              long long n, void* stream) {
  if (!out || !out2 || !grad_out || !grad_x || !grad_x2 || n < 1) return PCFA_ERR_INVALID_ARG;
  const int vec_ok = al16(out) && al16(out2) && al16(grad_out) && al16(grad_x) && al16(grad_x2);
  pcfa_launch(relu_bwd2_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, out, out2, grad_out, grad_x,
              grad_x2, n, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

// out = srcs[0] + srcs[1] + ... + srcs[n-1] (n <= 16) in index order: the gradient of a tensor that the twelve
// refinement iterations all read (the hoisted gate pre-activations of SepConvGRU, models/raft/update.py:45-60) summed by
// ONE launch reading every contribution once, instead of n-1 accumulate kernels of 3 tensor passes each.
namespace {
struct SumSrcs {
  const float* p[16];
};
__global__ void sum_n_kernel(SumSrcs s, int n, float* __restrict__ out, long long numel, int vec_ok) {
  for_each4(numel, vec_ok, [&](long long i, bool vec) {
    if (vec) {
      float4 a = LD4(s.p[0], i);
      for (int k = 1; k < n; ++k) {
        const float4 b = LD4(s.p[k], i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      ST4(out, i, a);
    } else {
      float a = s.p[0][i];
      for (int k = 1; k < n; ++k) a += s.p[k][i];
      out[i] = a;
    }
  });
}
// coords1_new = coords1 + delta ; flow_new = coords1_new - coords0   (models/raft/raft.py:122-137: the two
// coordinate updates of a refinement iteration, one launch instead of an add and a subtract)
__global__ void flow_step_kernel(const float* __restrict__ c1, const float* __restrict__ d,
                                 const float* __restrict__ c0, float* __restrict__ c1n, float* __restrict__ fl,
                                 long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = c1[i] + d[i];
    c1n[i] = v;
    fl[i] = v - c0[i];
  }
}
}  // namespace

extern "C" int pcfa_flow_step(const float* coords1, const float* delta, const float* coords0, float* coords1_new,
                              float* flow_new, long long n, void* stream) {
  if (!coords1 || !delta || !coords0 || !coords1_new || !flow_new || n < 1) return PCFA_ERR_INVALID_ARG;
  pcfa_launch(flow_step_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, coords1, delta, coords0,
              coords1_new, flow_new, n);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

extern "C" int pcfa_sum_n(const float* const* srcs, int n, float* out, long long numel, void* stream) {
  if (!srcs || !out || n < 1 || n > 16 || numel < 1) return PCFA_ERR_INVALID_ARG;
  SumSrcs s;
  int vec_ok = al16(out);
  for (int k = 0; k < 16; ++k) {
    s.p[k] = k < n ? srcs[k] : srcs[0];
    if (!s.p[k]) return PCFA_ERR_INVALID_ARG;
    vec_ok = vec_ok && al16(s.p[k]);
  }
  pcfa_launch(sum_n_kernel, dim3(blocks_for(numel)), dim3(256), 0, (hipStream_t)stream, s, n, out, numel, vec_ok);
  PCFA_LAUNCH_CHECK();
  return PCFA_OK;
}

