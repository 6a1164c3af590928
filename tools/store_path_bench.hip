// Per-CU store-path rate on MI355X: B workgroups of 256 threads (one per CU while B <= 256) stream 8-byte-per-lane or
// 16-byte-per-lane coalesced stores into private regions; reports bytes per shader cycle per workgroup.  Question behind
// it: is the ~10 B/cycle/CU seen in the assembly copy-out phase a chip-wide (HBM) share or a per-CU limit?
//   hipcc --offload-arch=gfx950 -O3 tools/store_path_bench.hip -o gpurun_out/store_path_bench && gpurun_out/store_path_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W>   // doubles per lane and store
__global__ void __launch_bounds__(256) store_kernel(double* out, size_t per_block, int reps, unsigned long long* cyc) {
  double* base = out + (size_t)blockIdx.x * per_block;
  const int t = threadIdx.x;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)t * W; i + W <= per_block; i += 256 * W) {
      if (W == 1) base[i] = (double)r;
      else {
        double2 v = {(double)r, (double)i};
        *reinterpret_cast<double2*>(base + i) = v;
      }
    }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cyc[blockIdx.x] = c1 - c0;
}

int main() {
  const size_t per_block = 4u << 20;   // doubles per block per pass = 32 MB (far beyond L2 share)
  const int maxb = 512;
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, (size_t)maxb * per_block * sizeof(double));
  hipMalloc(&cyc, maxb * sizeof(unsigned long long));
  for (int W = 1; W <= 2; ++W)
    for (int nb : {1, 2, 8, 32, 64, 128, 256, 512}) {
      for (int it = 0; it < 2; ++it) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        if (W == 1) store_kernel<1><<<nb, 256>>>(out, per_block, 1, cyc);
        else store_kernel<2><<<nb, 256>>>(out, per_block, 1, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it == 0) continue;
        std::vector<unsigned long long> h(nb);
        hipMemcpy(h.data(), cyc, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= nb;
        printf("%2d B/lane, %3d workgroups: %.2f B/cycle/workgroup (in-kernel), aggregate %.0f GB/s (event time %.3f ms)\n", 8 * W, nb,
               per_block * 8.0 / mean, nb * per_block * 8.0 / (ms * 1e6), ms);
      }
    }
  return 0;
}
