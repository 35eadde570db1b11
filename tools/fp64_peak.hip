// Calibration: the FP64 vector issue rate the chip sustains on a pure stream of
// independent v_fma_f64 (8 chains per lane, 4 or 2 waves per SIMD), and on the
// mul/fma/add mix of the objective kernel's loop body.  Prints wave-instructions per second
// and the implied fraction of the 2.4 GHz peak (256 CUs x 4 SIMDs x 2.4e9 / 4).
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o /tmp/fp64_peak && /tmp/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MIX>
__global__ void __launch_bounds__(64) k(double* out, int iters, double a, double b) {
  double x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3 + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MIX == 0) x[j] = __builtin_fma(x[j], a, b);
        else if (MIX == 1) x[j] = (r & 1) ? x[j] * a : __builtin_fma(x[j], a, b);
        else x[j] = (r % 3 == 0) ? x[j] + b : ((r % 3 == 1) ? x[j] * a : __builtin_fma(x[j], a, b));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

// three distinct VGPR-pair sources per FMA (the objective kernel's row updates look like
// this), against fma(x, uniform, uniform) above
__global__ void __launch_bounds__(64) k3(double* out, const double* in, int iters) {
  double x[8], y[8], z[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    x[j] = threadIdx.x * 1e-3 + j;
    y[j] = in[threadIdx.x + 64 * j];
    z[j] = in[threadIdx.x + 64 * (8 + j)];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], y[(j + r) & 7], z[(j + 3 * r) & 7]);
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

void run3(int waves_per_simd) {
  const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
  double *d, *in;
  hipMalloc(&d, (size_t)blocks * 64 * sizeof(double));
  hipMalloc(&in, 64 * 16 * sizeof(double));
  double h[64 * 16];
  for (int i = 0; i < 64 * 16; ++i) h[i] = (i % 2) ? 0.999999 : 1e-9;
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k3, dim3(blocks), dim3(64), 0, 0, d, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * iters * 64.0;
    if (rep == 2)
      printf("%-22s %d waves/SIMD: %7.2f ms  %.1f G wave-instr/s = %.1f%% of the 2.4 GHz peak\n",
             "fma, 3 VGPR sources", waves_per_simd, ms, winstr / (ms * 1e-3) / 1e9,
             winstr / (ms * 1e-3) / 614.4e9 * 100);
  }
  hipFree(d);
  hipFree(in);
}

template <int MIX>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
  double* d;
  hipMalloc(&d, (size_t)blocks * 64 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999999, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * iters * 64.0;
    if (rep == 2)
      printf("%-22s %d waves/SIMD: %7.2f ms  %.1f G wave-instr/s = %.1f%% of the 2.4 GHz peak\n", name,
             waves_per_simd, ms, winstr / (ms * 1e-3) / 1e9, winstr / (ms * 1e-3) / 614.4e9 * 100);
  }
  hipFree(d);
}

int main() {
  run<0>("fma only", 4);
  run<0>("fma only", 2);
  run<0>("fma only", 1);
  run<1>("mul/fma 1:1", 4);
  run<2>("add/mul/fma 1:1:1", 4);
  run3(4);
  run3(2);
  return 0;
}
