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

template <int VAR>
static int launch_var(bool stamped, const float* pyr, const float* coords, float* out, int B, int Q, const PyrLayout& P,
                      unsigned long long* stamps, hipStream_t s) {
  dim3 grid(pcfa_cdiv(Q, QB), P.L, B), block(QB, 9, 1);
  if (stamped)
    hipLaunchKernelGGL((corr_lookup_fwd_var_kernel<VAR, true>), grid, block, 0, s, pyr, coords, out, Q, QB, P, stamps);
  else
    hipLaunchKernelGGL((corr_lookup_fwd_var_kernel<VAR, false>), grid, block, 0, s, pyr, coords, out, Q, QB, P, stamps);
  return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int dev_lookup_fwd_var(
    int var, int stamped, const float* pyr, const float* coords, float* out, int B, int H, int W, int num_levels,
    unsigned long long* stamps, void* stream) {
  PyrLayout P;
  if (!pcfa_make_layout(P, H, W, num_levels)) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int Q = H * W;
  switch (var) {
    case 0: return launch_var<0>(stamped, pyr, coords, out, B, Q, P, stamps, s);
    case 1: return launch_var<1>(stamped, pyr, coords, out, B, Q, P, stamps, s);
    case 2: return launch_var<2>(stamped, pyr, coords, out, B, Q, P, stamps, s);
    case 3: return launch_var<3>(stamped, pyr, coords, out, B, Q, P, stamps, s);
  }
  return -2;
}

extern "C" __attribute__((visibility("default"))) int dev_stamp_slots() { return STAMP_SLOTS; }
