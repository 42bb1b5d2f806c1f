// Cost of the "last arriver reduces" hand-off a cross-workgroup split-K would need (tools/dev, not product code):
// every workgroup writes a 16 KB partial tile, fences, bumps its tile's counter; the last of KS arrivers reads the KS
// partials back and writes the 16 KB result.  Variants: 0 = partial write only, 1 = + __threadfence + atomic,
// 2 = + the last arriver's read-back and sum.  A delay loop stands in for the K loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k(float* ws, unsigned* counters, float* out, int KS, int variant, int spin) {
  __shared__ unsigned s_prev;
  const int tile = blockIdx.x / KS, slice = blockIdx.x % KS, tid = threadIdx.x;
  float acc[16];
  for (int r = 0; r < 16; ++r) acc[r] = (float)(tid + r);
  for (int i = 0; i < spin; ++i)
    for (int r = 0; r < 16; ++r) acc[r] = fmaf(acc[r], 1.0001f, 0.5f);
  float* mine = ws + ((size_t)tile * KS + slice) * 4096;
  for (int r = 0; r < 16; ++r) mine[r * 256 + tid] = acc[r];
  if (variant == 0) return;
  __threadfence();
  __syncthreads();
  if (tid == 0) s_prev = atomicAdd(&counters[tile], 1u);
  __syncthreads();
  if (s_prev != (unsigned)(KS - 1)) return;
  if (tid == 0) counters[tile] = 0;
  if (variant == 1) return;
  __threadfence();
  for (int r = 0; r < 16; ++r) {
    float s = 0.f;
    for (int k2 = 0; k2 < KS; ++k2) s += ws[((size_t)tile * KS + k2) * 4096 + r * 256 + tid];
    out[(size_t)tile * 4096 + r * 256 + tid] = s;
  }
}

int main() {
  const int tiles = 440, KS = 4;
  float *ws, *out;
  unsigned* cnt;
  hipMalloc(&ws, (size_t)tiles * KS * 4096 * 4);
  hipMalloc(&out, (size_t)tiles * 4096 * 4);
  hipMalloc(&cnt, tiles * 4);
  hipMemset(cnt, 0, tiles * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int spin : {0, 2000}) {
    for (int variant = 0; variant < 3; ++variant) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles * KS), dim3(256), 0, 0, ws, cnt, out, KS, variant, spin);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(tiles * KS), dim3(256), 0, 0, ws, cnt, out, KS, variant, spin);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("spin %5d variant %d: %.2f us per launch (1760 workgroups)\n", spin, variant, ms * 1000 / 20);
    }
  }
  // correctness of variant 2
  hipLaunchKernelGGL(k, dim3(tiles * KS), dim3(256), 0, 0, ws, cnt, out, KS, 2, 0);
  hipDeviceSynchronize();
  float h[4];
  hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  printf("out[0..3] = %g %g %g %g (expect %g ...)\n", h[0], h[1], h[2], h[3], 4.0f * 0);
  return 0;
}
