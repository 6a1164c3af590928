// Read / write bandwidth of consecutive 2 GiB allocations of one process: do some regions of the device memory stream slower?
// (the CG product of the bench matrix runs 0.49 or 0.54 ms depending on which allocation holds the values: tools/placement_probe.py)
// build: hipcc --offload-arch=gfx950 -O3 tools/placement_bw.hip -o tools/build/placement_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) read_kernel(const double2* __restrict__ p, size_t n, double* out) {
  double a = 0.0, b = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = p[i];
    a += v.x;
    b += v.y;
  }
  if (a + b == 123.456) out[0] = a;   // never true: keeps the loads
}
__global__ void __launch_bounds__(256) write_kernel(double2* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}

int main(int argc, char** argv) {
  const int nchunk = argc > 1 ? atoi(argv[1]) : 12;
  const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(double2);
  std::vector<double2*> p(nchunk);
  double* out;
  hipMalloc((void**)&out, 8);
  for (int c = 0; c < nchunk; ++c) {
    if (hipMalloc((void**)&p[c], bytes) != hipSuccess) {
      printf("chunk %d: allocation failed\n", c);
      return 1;
    }
    hipMemset(p[c], 0, bytes);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int c = 0; c < nchunk; ++c) {
    float best_r = 1e9f, best_w = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      float ms;
      hipEventRecord(e0);
      read_kernel<<<2048, 256>>>(p[c], n, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best_r) best_r = ms;
      hipEventRecord(e0);
      write_kernel<<<2048, 256>>>(p[c], n);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best_w) best_w = ms;
    }
    printf("chunk %2d at %p: read %.0f GB/s  write %.0f GB/s\n", c, (void*)p[c], bytes / best_r / 1e6, bytes / best_w / 1e6);
  }
  return 0;
}
