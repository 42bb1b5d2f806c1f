// Diagnostic build of the correlation-lookup kernels (NOT part of libpcfa_hip.so): compiles
// pcfa_amd/csrc/corr_lookup.hip a second time with PCFA_LOOKUP_DEV, which adds the in-kernel-stamped
// forward kernel (cdna_hip_programming.md 7, "In-kernel stamps") and experimental variants, behind
// dev_* entry points that tools/dev/lookup_stamps.py drives on the GPU box.
#define PCFA_LOOKUP_DEV 1
#include "../../pcfa_amd/csrc/corr_lookup.hip"

PcfaTimingState& pcfa_timing_state() {
  static thread_local PcfaTimingState s;
  return s;
}

extern "C" __attribute__((visibility("default"))) int dev_lookup_fwd_stamped(
    const float* pyr, const float* coords, float* out, int B, int H, int W, int num_levels,
    unsigned long long* stamps, void* stream) {
  PyrLayout P;
  if (!pcfa_make_layout(P, H, W, num_levels)) return -1;
  const int Q = H * W;
  dim3 grid(pcfa_cdiv(Q, QB), P.L, B), block(QB, 9, 1);
  hipLaunchKernelGGL(corr_lookup_fwd_stamped_kernel<4>, grid, block, 0, (hipStream_t)stream, pyr, coords, out, Q,
                     QB, P, stamps);
  return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int dev_stamp_slots() { return STAMP_SLOTS; }
