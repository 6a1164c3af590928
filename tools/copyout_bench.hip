// What is the ceiling of the assembly kernels' SKELETON on MI355X?  Every atomics-free assembly kernel of this library is "clear an LDS
// image, (integrate), barrier, copy the image out as one contiguous piece of the value array": the row-run kernels reach 4.0 TB/s in
// that skeleton alone, plain stores 5.8 TB/s.  This bench runs the skeleton with nothing else, persistent workgroups walking pieces
// w, w + grid, ... of PIECE doubles each, in variants:
//   mode 0  registers -> global (no LDS, no barrier)
//   mode 1  clear LDS, barrier, LDS -> global, barrier                      (the kernels' skeleton)
//   mode 2  as 1, the clear of piece t+1 merged into the copy of piece t (one barrier per piece)
//   mode 3  as 1 with `spin` dependent FMAs per lane between the barriers    (a stand-in for the integration)
//   nt = 1  non-temporal stores
//   hipcc --offload-arch=gfx950 -O3 tools/copyout_bench.hip -o gpurun_out/copyout_bench && gpurun_out/copyout_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double __attribute__((ext_vector_type(2), aligned(8))) d2u;

template <int MODE, bool NT>
__global__ void __launch_bounds__(256) skeleton_kernel(double* __restrict__ out, int piece, int npieces, int spin) {
  extern __shared__ double img[];
  const int tid = threadIdx.x;
  double seed = 1.0 + tid;
  if (MODE == 2) {
    for (int i = tid; i < piece; i += 256) img[i] = 0.0;
    __syncthreads();
  }
  for (int w = blockIdx.x; w < npieces; w += gridDim.x) {
    double* __restrict__ dst = out + (size_t)w * piece;
    if (MODE == 0) {
      for (int i = 2 * tid; i + 1 < piece; i += 512) {
        d2u v;
        v.x = seed;
        v.y = (double)i;
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2u*>(dst + i));
        else *reinterpret_cast<d2u*>(dst + i) = v;
      }
      continue;
    }
    if (MODE != 2) {
      for (int i = tid; i < piece; i += 256) img[i] = 0.0;
      __syncthreads();
    }
    if (MODE == 3) {
      double a = seed;
      for (int k = 0; k < spin; ++k) a = fma(a, 1.0000001, 1e-9);
      if (a == 12345.678) img[tid] = a;   // (keeps the loop)
      seed = a;
    }
    if (MODE != 0) {
      if (tid < 64) img[tid] += seed;     // something to copy
      __syncthreads();
    }
    for (int i = 2 * tid; i + 1 < piece; i += 512) {
      d2u v;
      v.x = img[i];
      v.y = img[i + 1];
      if (MODE == 2) {
        d2u z;
        z.x = z.y = 0.0;
        *reinterpret_cast<d2u*>(img + i) = z;
      }
      if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2u*>(dst + i));
      else *reinterpret_cast<d2u*>(dst + i) = v;
    }
    __syncthreads();
  }
}

template <int MODE, bool NT>
static void run(double* out, int piece, int npieces, int wgs_per_cu, int spin, const char* name) {
  const size_t lds = (size_t)piece * sizeof(double);
  hipFuncSetAttribute(reinterpret_cast<const void*>(skeleton_kernel<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const int grid = 256 * wgs_per_cu;
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    skeleton_kernel<MODE, NT><<<grid, 256, MODE == 0 ? 0 : lds>>>(out, piece, npieces, spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  const double gb = (double)piece * npieces * 8.0 / 1e9;
  printf("%-44s piece %5.1f KB  %d WG/CU  spin %4d : %.3f ms  %.2f TB/s\n", name, piece * 8.0 / 1024, wgs_per_cu, spin, best, gb / best);
  fflush(stdout);
}

int main() {
  const size_t total = (size_t)10 << 30;   // 10 GiB target
  double* out;
  if (hipMalloc(&out, total) != hipSuccess) return 1;
  hipMemset(out, 0, total);
  for (int piece : {2304, 3600}) {   // 18 KB (the average run of the 64^3 K), 28.8 KB (its largest)
    const int np = (int)(total / 8 / piece);
    for (int w : {4, 5}) {
      run<0, false>(out, piece, np, w, 0, "0 registers -> global");
      run<0, true>(out, piece, np, w, 0, "0 registers -> global, non-temporal");
      run<1, false>(out, piece, np, w, 0, "1 clear, barrier, LDS -> global, barrier");
      run<1, true>(out, piece, np, w, 0, "1 ..., non-temporal");
      run<2, false>(out, piece, np, w, 0, "2 clear merged into the copy");
      run<2, true>(out, piece, np, w, 0, "2 ..., non-temporal");
    }
    for (int spin : {100, 300, 1000}) run<3, false>(out, piece, np, 4, spin, "3 skeleton + dependent FMAs");
    for (int w : {8, 2}) run<1, false>(out, piece, np, w, 0, "1 skeleton");
  }
  hipFree(out);
  return 0;
}
