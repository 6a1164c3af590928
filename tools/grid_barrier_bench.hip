// Cost of a software grid barrier on MI355X (8 XCDs, one L2 each): the number a one-launch-per-chunk CG would pay
// per phase instead of a ~5 us kernel launch.  Every block is co-resident (grid <= 8 blocks per CU), the spin is
// bounded (a stuck barrier ends the kernel and is reported), nothing else is measured.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_bench.hip -o /tmp/gbb && /tmp/gbb
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <bool FENCE, bool TOUCH>
__global__ void __launch_bounds__(256) barrier_kernel(unsigned* cnt, int iters, unsigned* fail, double* buf, long n) {
  double acc = 0.0;
  __shared__ int stop;
  if (threadIdx.x == 0) stop = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if (TOUCH) {   // each phase writes and reads 8 B per thread of a vector another block wrote in the previous phase
      const long i = ((long)blockIdx.x * 256 + threadIdx.x + (long)it * 4099 * 256) % n;
      acc += buf[i];
      buf[i] = acc * 0.5 + 1.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (FENCE) __threadfence();   // release what this block wrote to every other XCD
      const unsigned target = (unsigned)(it + 1) * gridDim.x;
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        // a block that cannot be scheduled next to the others would never arrive: give up everywhere, at once
        if (++spins > 200000 || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          atomicAdd(fail, 1u);
          stop = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (stop) break;
  }
  if (TOUCH && acc == 12345.678) buf[0] = acc;
}

template <bool FENCE, bool TOUCH>
static int run(int grid, int iters, unsigned* d_cnt, unsigned* d_fail, double* buf, long n, const char* what) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipMemset(d_cnt, 0, sizeof(unsigned)));
    CHK(hipEventRecord(e0, 0));
    barrier_kernel<FENCE, TOUCH><<<grid, 256>>>(d_cnt, iters, d_fail, buf, n);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  unsigned fail = 0;
  CHK(hipMemcpy(&fail, d_fail, sizeof(unsigned), hipMemcpyDeviceToHost));
  printf("%-34s grid %5d: %7.2f us per barrier%s\n", what, grid, ms * 1e3 / iters, fail ? "  (STUCK barriers reported)" : "");
  return 0;
}

int main() {
  unsigned *d_cnt, *d_fail;
  double* buf;
  const long n = 1306368;   // a rank's rows at N = 8
  CHK(hipMalloc(&d_cnt, sizeof(unsigned)));
  CHK(hipMalloc(&d_fail, sizeof(unsigned)));
  CHK(hipMalloc(&buf, n * sizeof(double)));
  CHK(hipMemset(d_fail, 0, sizeof(unsigned)));
  CHK(hipMemset(buf, 0, n * sizeof(double)));
  const int iters = 2000;
  for (int grid : {256, 1024, 2048}) {
    if (run<false, false>(grid, iters, d_cnt, d_fail, buf, n, "counter only")) return 1;
    if (run<true, false>(grid, iters, d_cnt, d_fail, buf, n, "with __threadfence")) return 1;
    if (run<true, true>(grid, iters, d_cnt, d_fail, buf, n, "fence + 8 B/thread of vector traffic")) return 1;
  }
  // reference: an empty kernel launch pair back to back
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < 1000; ++i) barrier_kernel<false, false><<<256, 256>>>(d_cnt, 0, d_fail, buf, n);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel, back-to-back launches:      %7.2f us per launch\n", ms);
  return 0;
}
