// kernels_util.hip -- layout helpers: LDS-tiled transposes and a flat copy.
// HBM-bound byte movers: 64x64 (32x32 for 16-byte elements) tiles staged through LDS so both the read and the
// write side are contiguous 512 B (f64) / 64 B (u8) segments per wave-row.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace nghmm {

namespace {

// out[c][r] = in[r][c]; in is rows x cols.  Block = 64x4 threads, tile 64x64,
// LDS row padded by one element (guide: Guideline 4) to avoid bank conflicts on
// the transposed read.
template <typename T, int TILE>
__global__ void __launch_bounds__(256)
k_transpose(const T* __restrict__ in, T* __restrict__ out, uint64_t rows, uint64_t cols) {
  __shared__ T tile[TILE][TILE + 1];
  constexpr int STEP = 256 / TILE;
  const uint64_t tiles_c = (cols + TILE - 1) / TILE;
  const uint64_t c0 = ((uint64_t)blockIdx.x % tiles_c) * TILE;
  const uint64_t r0 = ((uint64_t)blockIdx.x / tiles_c) * TILE;
  const int tx = threadIdx.x % TILE;
  const int ty = threadIdx.x / TILE;
  for (int j = ty; j < TILE; j += STEP) {
    const uint64_t r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) tile[j][tx] = in[r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < TILE; j += STEP) {
    const uint64_t c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) out[c * rows + r] = tile[tx][j];
  }
}

__global__ void __launch_bounds__(256)
k_copy_f64(const double* __restrict__ in, double* __restrict__ out, uint64_t n) {
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n;
       c += (uint64_t)gridDim.x * blockDim.x)
    out[c] = in[c];
}

// printf("%f") of values in [0, 1] (posteriors): always "d.dddddd", 8 characters, followed
// by a tab or, after the last value of a row, a newline -- 9 bytes per value, so every
// value's place in the text is known and one thread formats one value.  The digits are
// those of glibc: the exact binary value rounded to 6 decimals, ties to even.  With
// p = fl(v * 1e6) and err = fma(v, 1e6, -p) the exact product is p + err; p - floor(p) - 0.5
// is exact and, unless zero, larger than |err|, so its sign decides; on zero err decides,
// and a true tie goes to the even integer.  A value outside [0, 1] raises *bad.
__global__ void __launch_bounds__(256)
k_format_fixed6(const double* __restrict__ in, uint64_t rows, uint64_t cols,
                char* __restrict__ out, int* __restrict__ bad) {
  const uint64_t n = rows * cols;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const double v = in[k];
    if (!(v >= 0.0 && v <= 1.0)) *bad = 1;
    const double p = v * 1e6;
    const double err = __builtin_fma(v, 1e6, -p);
    const double n0 = __builtin_floor(p);
    const double d = (p - n0) - 0.5;
    uint32_t q = (uint32_t)n0;
    if (d > 0 || (d == 0 && (err > 0 || (err == 0 && (q & 1u))))) ++q;
    char* o = out + k * 9;
    uint32_t frac = q % 1000000u;
    o[0] = (char)('0' + q / 1000000u);
    o[1] = '.';
#pragma unroll
    for (int j = 7; j >= 2; --j) {
      o[j] = (char)('0' + frac % 10u);
      frac /= 10u;
    }
    o[8] = ((k + 1) % cols == 0) ? '\n' : '\t';
  }
}

template <typename T, int TILE>
void launch_transpose(hipStream_t st, const T* in, T* out, uint64_t rows, uint64_t cols) {
  if (rows == 0 || cols == 0) return;
  // 1-D grid: a 5M-site dimension would overflow gridDim.y
  dim3 grid((unsigned)(((cols + TILE - 1) / TILE) * ((rows + TILE - 1) / TILE)));
  hipLaunchKernelGGL((k_transpose<T, TILE>), grid, dim3(256), 0, st, in, out, rows, cols);
}

}  // namespace

void launch_transpose_f64(hipStream_t st, const double* in, double* out, uint64_t rows,
                          uint64_t cols) {
  launch_transpose<double, 64>(st, in, out, rows, cols);
}

void launch_transpose_u8(hipStream_t st, const uint8_t* in, uint8_t* out, uint64_t rows,
                         uint64_t cols) {
  launch_transpose<uint8_t, 64>(st, in, out, rows, cols);
}

void launch_transpose_pairs_f64(hipStream_t st, const double* in, double* out, uint64_t rows,
                                uint64_t cols) {
  launch_transpose<double2, 32>(st, reinterpret_cast<const double2*>(in),
                            reinterpret_cast<double2*>(out), rows, cols);
}

void launch_format_fixed6(hipStream_t st, const double* in, uint64_t rows, uint64_t cols, char* out,
                          int* bad) {
  if (rows == 0 || cols == 0) return;
  uint64_t blocks = (rows * cols + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_format_fixed6, dim3((unsigned)blocks), dim3(256), 0, st, in, rows, cols, out,
                     bad);
}

void launch_copy_f64(hipStream_t st, const double* in, double* out, uint64_t n) {
  if (n == 0) return;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n);
}

}  // namespace nghmm
