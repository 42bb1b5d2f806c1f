// Residency census for gfx950: how many workgroups of (threads, LDS bytes) does one CU hold at once?
// Each workgroup stamps s_memrealtime at start/end and spins ~3 us; the host counts, per CU, the
// maximum number of workgroups whose lifetimes overlap.  Diagnostic tool (tools/dev), not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

extern __shared__ char dyn_lds[];

__global__ void census_kernel(unsigned long long* rec, int lds_bytes, int spin_ticks) {
  unsigned long long t0, t1;
  unsigned hwid, xcc;
  asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\t"
               "s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(hwid), "=s"(xcc)::"memory");
  // touch the LDS so the allocation is real
  for (int i = threadIdx.x * 4; i < lds_bytes; i += blockDim.x * 4) *(volatile int*)(dyn_lds + i) = i;
  __syncthreads();
  do {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __builtin_amdgcn_s_sleep(8);
  } while ((long long)(t1 - t0) < spin_ticks);
  if (threadIdx.x == 0) {
    rec[blockIdx.x * 3 + 0] = t0;
    rec[blockIdx.x * 3 + 1] = t1;
    rec[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | hwid;
  }
}

template <int LDS_FLOATS, int BY>
__global__ __launch_bounds__(64 * BY) void census_static_kernel(unsigned long long* rec, int spin_ticks) {
  __shared__ __attribute__((aligned(16))) float s_lds[LDS_FLOATS];
  unsigned long long t0, t1;
  unsigned hwid, xcc;
  asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\t"
               "s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(hwid), "=s"(xcc)::"memory");
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int i = tid; i < LDS_FLOATS; i += 64 * BY) *(volatile float*)(s_lds + i) = (float)i;
  __syncthreads();
  do {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __builtin_amdgcn_s_sleep(8);
  } while ((long long)(t1 - t0) < spin_ticks);
  const int g = blockIdx.y * gridDim.x + blockIdx.x;
  if (tid == 0) {
    rec[g * 3 + 0] = t0;
    rec[g * 3 + 1] = t1;
    rec[g * 3 + 2] = ((unsigned long long)xcc << 32) | hwid;
  }
}

template <int LDS_FLOATS, int NW, bool BOUNDED>
__global__ __launch_bounds__(BOUNDED ? 64 * NW : 1024) void census_static1d_kernel(unsigned long long* rec, int spin_ticks) {
  __shared__ __attribute__((aligned(16))) float s_lds[LDS_FLOATS];
  unsigned long long t0, t1;
  unsigned hwid, xcc;
  asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\t"
               "s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(hwid), "=s"(xcc)::"memory");
  const int tid = threadIdx.x;
  for (int i = tid; i < LDS_FLOATS; i += 64 * NW) *(volatile float*)(s_lds + i) = (float)i;
  __syncthreads();
  do {
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __builtin_amdgcn_s_sleep(8);
  } while ((long long)(t1 - t0) < spin_ticks);
  const int g = blockIdx.y * gridDim.x + blockIdx.x;
  if (tid == 0) {
    rec[g * 3 + 0] = t0;
    rec[g * 3 + 1] = t1;
    rec[g * 3 + 2] = ((unsigned long long)xcc << 32) | hwid;
  }
}

static void report(const char* name, std::vector<unsigned long long>& h, int nwg, int waves) {
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int g = 0; g < nwg; ++g) {
    const unsigned long long key = ((h[g * 3 + 2] >> 32) << 8) | ((h[g * 3 + 2] >> 8) & 0xFF);
    ev[key].push_back({h[g * 3 + 0], +1});
    ev[key].push_back({h[g * 3 + 1], -1});
    tmin = std::min(tmin, h[g * 3 + 0]);
    tmax = std::max(tmax, h[g * 3 + 1]);
  }
  int best = 0;
  for (auto& kv : ev) {
    std::sort(kv.second.begin(), kv.second.end());
    int cur = 0;
    for (auto& e : kv.second) { cur += e.second; best = std::max(best, cur); }
  }
  printf("%-40s nwg %5d  max_WG/CU %d  waves/CU %d  span %.2f us (%zu CUs)\n", name, nwg, best, best * waves, (tmax - tmin) / 100.0, ev.size());
}

template <int LDS_FLOATS, int BY>
static void run_static(const char* name, unsigned long long* d, std::vector<unsigned long long>& h, dim3 grid) {
  const int nwg = grid.x * grid.y;
  hipMemset(d, 0, nwg * 3 * 8);
  hipLaunchKernelGGL((census_static_kernel<LDS_FLOATS, BY>), grid, dim3(64, BY, 1), 0, 0, d, 300);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d, nwg * 3 * 8, hipMemcpyDeviceToHost);
  report(name, h, nwg, BY);
}

template <int LDS_FLOATS, int NW, bool BOUNDED>
static void run_static1d(const char* name, unsigned long long* d, std::vector<unsigned long long>& h, dim3 grid) {
  const int nwg = grid.x * grid.y;
  (void)hipMemset(d, 0, nwg * 3 * 8);
  hipLaunchKernelGGL((census_static1d_kernel<LDS_FLOATS, NW, BOUNDED>), grid, dim3(64 * NW, 1, 1), 0, 0, d, 300);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), d, nwg * 3 * 8, hipMemcpyDeviceToHost);
  report(name, h, nwg, NW);
}

int main() {
  const int nwg = 1024;
  unsigned long long* d;
  hipMalloc(&d, nwg * 3 * 8);
  std::vector<unsigned long long> h(nwg * 3);
  const int threads[] = {256, 512, 576, 640, 1024};
  const int ldsk[] = {0, 16, 32, 40, 48, 53, 60, 64, 66, 72, 80};
  hipFuncSetAttribute((const void*)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("%8s %8s %10s %10s %12s\n", "threads", "lds_KiB", "max_WG/CU", "waves/CU", "span_us");
  for (int t : threads)
    for (int lk : ldsk) {
      const int lds = lk * 1024;
      hipMemset(d, 0, nwg * 3 * 8);
      hipLaunchKernelGGL(census_kernel, dim3(nwg), dim3(t), lds, 0, d, lds, 300);
      if (hipDeviceSynchronize() != hipSuccess) { printf("%8d %8d launch failed\n", t, lk); hipGetLastError(); continue; }
      hipMemcpy(h.data(), d, nwg * 3 * 8, hipMemcpyDeviceToHost);
      std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
      unsigned long long tmin = ~0ull, tmax = 0;
      for (int g = 0; g < nwg; ++g) {
        const unsigned long long key = ((h[g * 3 + 2] >> 32) << 8) | ((h[g * 3 + 2] >> 8) & 0xFF);
        ev[key].push_back({h[g * 3 + 0], +1});
        ev[key].push_back({h[g * 3 + 1], -1});
        tmin = std::min(tmin, h[g * 3 + 0]);
        tmax = std::max(tmax, h[g * 3 + 1]);
      }
      int best = 0;
      for (auto& kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0;
        for (auto& e : kv.second) { cur += e.second; best = std::max(best, cur); }
      }
      printf("%8d %8d %10d %10d %12.2f   (%zu CUs)\n", t, lk, best, best * ((t + 63) / 64), (tmax - tmin) / 100.0, ev.size());
    }
  run_static<16896, 9>("static 67584 B, block (64,9), grid 110x4", d, h, dim3(110, 4));
  run_static<16896, 9>("static 67584 B, block (64,9), grid 1024", d, h, dim3(1024, 1));
  run_static<16896, 8>("static 67584 B, block (64,8), grid 110x4", d, h, dim3(110, 4));
  run_static<16384, 9>("static 65536 B, block (64,9), grid 110x4", d, h, dim3(110, 4));
  run_static<16128, 9>("static 64512 B, block (64,9), grid 110x4", d, h, dim3(110, 4));
  run_static<12288, 9>("static 49152 B, block (64,9), grid 110x4", d, h, dim3(110, 4));
  run_static<8448, 9>("static 33792 B, block (64,9), grid 220x4", d, h, dim3(220, 4));
  run_static<8448, 5>("static 33792 B, block (64,5), grid 220x4", d, h, dim3(220, 4));
  run_static<8448, 4>("static 33792 B, block (64,4), grid 220x4", d, h, dim3(220, 4));
  run_static<16896, 4>("static 67584 B, block (64,4), grid 110x4", d, h, dim3(110, 4));
  run_static1d<16896, 9, true>("static 67584 B, block 576 1-D, bounded, 110x4", d, h, dim3(110, 4));
  run_static1d<16896, 9, false>("static 67584 B, block 576 1-D, unbounded, 110x4", d, h, dim3(110, 4));
  run_static1d<16896, 10, true>("static 67584 B, block 640 1-D, bounded, 110x4", d, h, dim3(110, 4));
  run_static1d<16896, 12, true>("static 67584 B, block 768 1-D, bounded, 110x4", d, h, dim3(110, 4));
  run_static1d<16896, 8, true>("static 67584 B, block 512 1-D, bounded, 110x4", d, h, dim3(110, 4));
  run_static1d<8448, 9, true>("static 33792 B, block 576 1-D, bounded, 220x4", d, h, dim3(220, 4));
  // dynamic, 2-D block
  for (int by : {8, 9}) {
    (void)hipMemset(d, 0, 440 * 3 * 8);
    hipLaunchKernelGGL(census_kernel, dim3(440), dim3(64, by, 1), 67584, 0, d, 67584, 300);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, 440 * 3 * 8, hipMemcpyDeviceToHost);
    report(by == 8 ? "dynamic 67584 B, block (64,8), 440" : "dynamic 67584 B, block (64,9), 440", h, 440, by);
  }
  for (int t : {576, 512}) {
    (void)hipMemset(d, 0, 440 * 3 * 8);
    hipLaunchKernelGGL(census_kernel, dim3(440), dim3(t, 1, 1), 67584, 0, d, 67584, 300);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, 440 * 3 * 8, hipMemcpyDeviceToHost);
    report(t == 576 ? "dynamic 67584 B, block 576 1-D, 440" : "dynamic 67584 B, block 512 1-D, 440", h, 440, t / 64);
  }
  return 0;
}
