// Does global_load_lds (LDS DMA) overlap with VALU work of the same wave on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_overlap.hip -o /tmp/lds_dma && /tmp/lds_dma
// Each of 1024 waves (4 per CU at 40 KB LDS) repeats: issue 32 KB of 16-byte LDS-DMA loads,
// run N dependent-free FMAs, wait for the loads.  Prints the time for loads only, FMAs only
// and both: "both" near max(...) = asynchronous, near the sum = serialised.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(64) k(const double* __restrict__ src, double* out, int iters,
                                        int n_fma, int do_load, size_t stride_doubles) {
  __shared__ __attribute__((aligned(16))) double buf[4096 + 900];  // 32 KB + pad -> 4 waves per CU
  const int lane = threadIdx.x;
  double a0 = lane, a1 = 1.5, a2 = 2.5, a3 = 3.5;
  const double* p = src + (size_t)blockIdx.x * stride_doubles;
  for (int it = 0; it < iters; ++it) {
    if (do_load) {
      for (int q = 0; q < 32; ++q)
        __builtin_amdgcn_global_load_lds(p + (size_t)it * 4096 + q * 128 + lane * 2, buf + q * 128, 16,
                                         0, 0);
    }
    for (int k = 0; k < n_fma; ++k) {
      a0 = __builtin_fma(a0, 1.0000001, 0.5);
      a1 = __builtin_fma(a1, 1.0000001, 0.5);
      a2 = __builtin_fma(a2, 1.0000001, 0.5);
      a3 = __builtin_fma(a3, 1.0000001, 0.5);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    a0 += buf[lane];
  }
  out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3;
}

int main() {
  const int waves = 1024, iters = 200;
  const size_t stride = (size_t)iters * 4096;  // doubles per wave: 200 x 32 KB = 6.5 MB
  double *src, *out;
  hipMalloc(&src, waves * stride * sizeof(double));
  hipMalloc(&out, waves * 64 * sizeof(double));
  hipMemset(src, 0, waves * stride * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int n_fma : {0, 400, 800, 1600}) {
    for (int do_load : {0, 1}) {
      if (!n_fma && !do_load) continue;
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, src, out, iters, n_fma, do_load, stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("n_fma %5d x4  load %d : %7.3f ms  (%.1f GB -> %.2f TB/s)\n", n_fma, do_load, best,
             do_load ? waves * stride * 8 / 1e9 : 0.0, do_load ? waves * stride * 8 / 1e9 / best : 0.0);
    }
  }
  return 0;
}
