// Micro-benchmark: what does HBM deliver on MI355X for streams that mix reads and writes?  Phase 1 of the tiled SpMV moves
// 2.74 GB of reads and 1.38 GB of writes per iteration (2:1); the 8 TB/s roofline is a read-or-write peak, the achievable rate
// of a mix is what the kernel should be compared with.  Every kernel streams 16 bytes per lane and access, grid-stride, over
// arrays of 2 GiB each; "r:w" = arrays read : arrays written per element.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NR, int NW>
__global__ void __launch_bounds__(256) k_mix(float4 const* __restrict__ a, float4 const* __restrict__ b, float4 const* __restrict__ c, float4* __restrict__ x,
                                             float4* __restrict__ y, size_t n, float4* sink)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float4 acc = {0, 0, 0, 0};
  for (; i < n; i += stride) {
    float4 v = {1, 2, 3, 4};
    if (NR >= 1) { float4 t = a[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (NR >= 2) { float4 t = b[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (NR >= 3) { float4 t = c[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (NW >= 1) x[i] = v;
    if (NW >= 2) y[i] = v;
    if (NW == 0) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  }
  if (NW == 0 && acc.x == 12345.678f) *sink = acc;
}

template <int NR, int NW>
int run(char const* name, float4* p[5], size_t n, float4* sink, int grid)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NR, NW>), grid, 256, 0, 0, p[0], p[1], p[2], p[3], p[4], n, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  double const gb = (double)(NR + NW) * n * 16 / 1e9;
  printf("%-28s r:w %d:%d  %7.3f ms  %8.1f GB/s total  (read %7.1f, write %7.1f)\n", name, NR, NW, best, gb / best * 1e3, NR * n * 16 / 1e6 / best, NW * n * 16 / 1e6 / best);
  return 0;
}

int main()
{
  size_t const bytes = (size_t)2 << 30, n = bytes / 16;
  float4* p[5];
  for (int k = 0; k < 5; ++k) { CK(hipMalloc(&p[k], bytes)); CK(hipMemset(p[k], 0, bytes)); }
  float4* sink;
  CK(hipMalloc(&sink, 64));
  for (int grid : {256 * 8, 256 * 32}) {
    printf("grid %d x 256 threads\n", grid);
    if (run<1, 0>("read", p, n, sink, grid)) return 1;
    if (run<2, 0>("read two streams", p, n, sink, grid)) return 1;
    if (run<0, 1>("write", p, n, sink, grid)) return 1;
    if (run<1, 1>("copy", p, n, sink, grid)) return 1;
    if (run<2, 1>("triad (phase-1 mix)", p, n, sink, grid)) return 1;
    if (run<3, 1>("3 reads, 1 write", p, n, sink, grid)) return 1;
    if (run<1, 2>("1 read, 2 writes", p, n, sink, grid)) return 1;
  }
  return 0;
}
