// Shared host/device helpers for libpcfa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pcfa_hip.h"

#define PCFA_WAVE 64

#define PCFA_LAUNCH_CHECK()                          \
  do {                                               \
    hipError_t _e = hipGetLastError();               \
    if (_e != hipSuccess) return (int)_e;            \
  } while (0)

// Pyramid layout shared by every corr kernel (passed by value as a kernel arg).
struct PyrLayout {
  int L;
  int h[PCFA_MAX_LEVELS];
  int w[PCFA_MAX_LEVELS];
  int off[PCFA_MAX_LEVELS];
  int slab;  // floats per query row, multiple of 4
};

static inline bool pcfa_make_layout(PyrLayout& P, int H, int W, int L) {
  if (L < 1 || L > PCFA_MAX_LEVELS || H < 1 || W < 1) return false;
  P.L = L;
  long long off = 0;
  int h = H, w = W;
  for (int l = 0; l < PCFA_MAX_LEVELS; ++l) {
    if (l < L) {
      P.h[l] = h;
      P.w[l] = w;
      P.off[l] = (int)off;
      off += (long long)h * w;
      h /= 2;
      w /= 2;
    } else {
      P.h[l] = 0;
      P.w[l] = 0;
      P.off[l] = (int)off;
    }
  }
  off = (off + 3) & ~3LL;
  if (off > 0x7fffffffLL) return false;
  P.slab = (int)off;
  return true;
}

static inline int pcfa_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
