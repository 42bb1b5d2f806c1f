// Diagnostic build of the correlation-lookup kernels (NOT part of libpcfa_hip.so): compiles
// pcfa_amd/csrc/corr_lookup.hip a second time with PCFA_LOOKUP_DEV, which adds the in-kernel-stamped
// kernels (cdna_hip_programming.md 7, "In-kernel stamps") behind dev_* entry points that
// tools/dev/lookup_stamps.py drives on the GPU box.
#define PCFA_LOOKUP_DEV 1
#include "../../pcfa_amd/csrc/corr_lookup.hip"

PcfaTimingState& pcfa_timing_state() {
  static thread_local PcfaTimingState s;
  return s;
}

#define DEV_API extern "C" __attribute__((visibility("default")))

DEV_API int dev_lookup_fwd_stamped(const float* pyr, const float* coords, float* out, int B, int H, int W,
                                   int num_levels, unsigned long long* stamps, void* stream) {
  PyrLayout P;
  if (!pcfa_make_layout(P, H, W, num_levels)) return -1;
  const int Q = H * W;
  dim3 grid(pcfa_cdiv(Q, PCFA_LOOKUP_QB), P.L, B), block(64, Geo<4, PCFA_LOOKUP_QB>::NW, 1);
  hipLaunchKernelGGL(corr_lookup_fwd_stamped_kernel<4>, grid, block, 0, (hipStream_t)stream, pyr, coords, out, Q, P,
                     stamps);
  return (int)hipGetLastError();
}

DEV_API int dev_lookup_bwd_stamped(float* dpyr, const float* coords, const float* grad_out, int B, int H, int W,
                                   int num_levels, unsigned long long* stamps, void* stream) {
  PyrLayout P;
  if (!pcfa_make_layout(P, H, W, num_levels)) return -1;
  const int Q = H * W;
  dim3 grid(pcfa_cdiv(Q, PCFA_LOOKUP_QB), P.L, B), block(64, Geo<4, PCFA_LOOKUP_QB>::NW, 1);
  hipLaunchKernelGGL(corr_lookup_bwd_stamped_kernel<4>, grid, block, 0, (hipStream_t)stream, dpyr, coords, grad_out,
                     Q, P, stamps);
  return (int)hipGetLastError();
}

DEV_API int dev_lookup_fwd_store(int mode, const float* pyr, const float* coords, float* out, int B, int H, int W,
                                 int num_levels, void* stream) {
  PyrLayout P;
  if (!pcfa_make_layout(P, H, W, num_levels)) return -1;
  const int Q = H * W;
  dim3 grid(pcfa_cdiv(Q, PCFA_LOOKUP_QB), P.L, B), block(64, Geo<4, PCFA_LOOKUP_QB>::NW, 1);
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(corr_lookup_fwd_store_kernel<0>, grid, block, 0, s, pyr, coords, out, Q, P);
  if (mode == 1) hipLaunchKernelGGL(corr_lookup_fwd_store_kernel<1>, grid, block, 0, s, pyr, coords, out, Q, P);
  if (mode == 2) hipLaunchKernelGGL(corr_lookup_fwd_store_kernel<2>, grid, block, 0, s, pyr, coords, out, Q, P);
  return (int)hipGetLastError();
}

DEV_API int dev_stamp_slots() { return STAMP_SLOTS; }
DEV_API int dev_waves() { return Geo<4, PCFA_LOOKUP_QB>::NW; }
DEV_API int dev_qb() { return PCFA_LOOKUP_QB; }
