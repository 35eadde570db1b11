// Do two HIP streams of one process run small kernels side by side on this box?  (tools/: a
// measurement, not part of the library.)  k_spin keeps `waves` one-wave workgroups busy for about
// `us` microseconds each; A on one stream and B on another, against both on one stream.
//   hipcc --offload-arch=gfx950 -O2 tools/stream_overlap.hip -o /tmp/stream_overlap && /tmp/stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_spin(long long ticks, unsigned long long* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned long long)(wall_clock64() - t0);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  unsigned long long* d;
  CK(hipMalloc(&d, 64));
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const long long ticks = (long long)rate_khz * 100 / 1000;  // 100 us
  hipStream_t a, b, hi, lo;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  int least, greatest;
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
  CK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least));
  printf("wall clock %d kHz, priorities least %d greatest %d\n", rate_khz, least, greatest);
  auto run = [&](hipStream_t s1, hipStream_t s2, int n1, int n2, int reps) {
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
      (void)hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < 4; ++k) {
        hipLaunchKernelGGL(k_spin, dim3(n1), dim3(64), 0, s1, ticks, d);
        hipLaunchKernelGGL(k_spin, dim3(n2), dim3(64), 0, s2, ticks, d + 1);
      }
      (void)hipDeviceSynchronize();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      best = us < best ? us : best;
    }
    return best;
  };
  for (int n : {100, 4096, 100000}) {
    printf("%6d + %6d one-wave workgroups of 100 us, 4 launches each: one stream %.0f us, two streams %.0f us, "
           "high + low priority %.0f us\n", n, n, run(a, a, n, n, 5), run(a, b, n, n, 5), run(hi, lo, n, n, 5));
  }
  printf("%6d + %6d: one stream %.0f us, two streams %.0f us, high + low %.0f us\n", 100, 100000,
         run(a, a, 100, 100000, 5), run(a, b, 100, 100000, 5), run(hi, lo, 100, 100000, 5));
  return 0;
}
