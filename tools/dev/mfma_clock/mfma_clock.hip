// Which fp32 MFMA shape sustains the higher rate on THIS box?  Register-only loops on pseudo-random operands (the matrix
// pipe's clock depends on what it multiplies), four independent accumulators per wave, 2048 workgroups x 4 waves (two waves
// per SIMD resident).  v_mfma_f32_32x32x2_f32: 4096 flop / 64 cycles; v_mfma_f32_16x16x4_f32: 2048 flop / 32 cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_clock/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  const unsigned seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  float a0 = (float)(seed & 0xffff) * (1.0f / 65536.f) - 0.5f, b0 = (float)((seed >> 16) & 0xffff) * (1.0f / 65536.f) - 0.5f;
  float a1 = b0 * 0.75f + 0.1f, b1 = a0 * 0.5f - 0.2f;
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c3, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  } else {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int i = 0; i < iters; ++i) {   // 8 x 2048 flop = the 32x32 loop's work per iteration
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, c7, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
  }
  if (s == 123456.789f) out[0] = s;
}

template <int SHAPE>
double run(float* d, int blocks, int iters, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<SHAPE><<<blocks, 256>>>(d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<SHAPE><<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * 4 * 4096 * reps;
  return flop / (ms * 1e-3) / 1e12;
}

int main() {
  float* d;
  hipMalloc(&d, 1024);
  for (int blocks : {1024, 2048, 4096})
    for (int rep = 0; rep < 2; ++rep)
      printf("blocks %d: 32x32x2 %.1f TFLOP/s   16x16x4 %.1f TFLOP/s\n", blocks, run<32>(d, blocks, 2000, 20),
             run<16>(d, blocks, 2000, 20));
  return 0;
}
