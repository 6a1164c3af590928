// Feasibility probe: would a symmetric (upper-half) SELL image speed up the lattice SpMV?  Rows are 64-row slices with
// W stored k-columns.  Variant FULL reads 27 own columns; variant HALF reads 14 own columns and 13 columns of the rows
// i - off (the transposed entries live in the upper halves of earlier rows: re-reads of data other waves stream anyway --
// L2 / Infinity-Cache hits if the reuse distance fits).  x gathers at the 27 lattice offsets in both variants.
//   hipcc --offload-arch=gfx950 -O3 tools/sym_spmv_probe.hip -o tools/build/sym_spmv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <bool HALF>
__global__ void __launch_bounds__(256) probe(const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y, long n,
                                             int nx, int nxy) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const long s = row >> 6;
  const int lane = row & 63;
  double acc = 0.0;
  int k = 0;
  constexpr int W = HALF ? 14 : 27;
  // own columns: offsets 0 .. (upper half) or all 27
#pragma unroll
  for (int dz = HALF ? 0 : -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int off = dz * nxy + dy * nx + dx;
        if (HALF && (dz == 0) && (dy < 0 || (dy == 0 && dx < 0))) continue;
        long c = row + off;
        c = c < 0 ? 0 : (c >= n ? n - 1 : c);
        acc = fma(val[(s * W + k) * 64 + lane], x[c], acc);
        ++k;
      }
  if (HALF) {
    int kk = 1;   // slot of the mirrored entry in the neighbour's upper list
#pragma unroll
    for (int dz = 0; dz <= 1; ++dz)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          if ((dz == 0) && (dy < 0 || (dy == 0 && dx <= 0))) continue;
          const int off = dz * nxy + dy * nx + dx;
          long j = row - off;                      // the row whose upper entry (j, row) is our lower entry
          j = j < 0 ? 0 : j;
          acc = fma(val[((j >> 6) * W + kk) * 64 + (j & 63)], x[j], acc);
          ++kk;
        }
  }
  y[row] = acc;
}

int main() {
  const int nx = 216, nxy = 216 * 216;
  const long n = (long)nxy * 216;
  const long ns = (n + 63) / 64;
  double *val, *x, *y;
  hipMalloc(&val, ns * 27 * 64 * sizeof(double));
  hipMalloc(&x, n * sizeof(double));
  hipMalloc(&y, n * sizeof(double));
  hipMemset(val, 0, ns * 27 * 64 * sizeof(double));
  hipMemset(x, 0, n * sizeof(double));
  const int grid = (int)((n + 255) / 256);
  for (int rep = 0; rep < 3; ++rep)
    for (int half = 0; half < 2; ++half) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) {
        if (half) probe<true><<<grid, 256>>>(val, x, y, n, nx, nxy);
        else probe<false><<<grid, 256>>>(val, x, y, n, nx, nxy);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.3f ms per product\n", half ? "upper half + mirrored re-reads" : "full rows", ms / 10);
    }
  return 0;
}
