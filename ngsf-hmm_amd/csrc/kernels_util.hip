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

void launch_copy_f64(hipStream_t st, const double* in, double* out, uint64_t n) {
  if (n == 0) return;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n);
}

}  // namespace nghmm
