// Does the physical placement of a buffer change what a plain stream over it reaches?  N buffers of SIZE bytes are allocated and KEPT (so each gets
// other physical memory), each is read (16 B per lane, grid-stride) and written a few times; then the pieces of the slowest and the fastest buffer
// are timed one by one.  hipcc --offload-arch=gfx950 -O3 -o placement_bench placement_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_read(u32x4 const* p, size_t n, unsigned* sink)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i + 3 * stride < n; i += 4 * stride) {
    u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride), c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  for (; i < n; i += stride) acc ^= p[i].x;
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) k_write(u32x4* p, size_t n, unsigned v)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  u32x4 const w = {v, v + 1, v + 2, v + 3};
  for (; i < n; i += stride) p[i] = w;
}
static double time_kernel(bool rd, u32x4* p, size_t n16, unsigned* sink, int reps)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  int const grid = 256 * 16;
  if (rd) k_read<<<grid, 256>>>(p, n16, sink); else k_write<<<grid, 256>>>(p, n16, 1u);
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) { if (rd) k_read<<<grid, 256>>>(p, n16, sink); else k_write<<<grid, 256>>>(p, n16, (unsigned)r); }
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return (double)n16 * 16.0 * reps / (ms * 1e-3) / 1e12;
}
int main(int argc, char** argv)
{
  int const N = argc > 1 ? atoi(argv[1]) : 12;
  size_t const size = (size_t)(argc > 2 ? atoi(argv[2]) : 2048) << 20;
  unsigned* sink; CK(hipMalloc(&sink, 4));
  std::vector<u32x4*> bufs(N);
  std::vector<double> rd(N), wr(N);
  for (int i = 0; i < N; ++i) { CK(hipMalloc(&bufs[i], size)); CK(hipMemset(bufs[i], 1, size)); }
  CK(hipDeviceSynchronize());
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < N; ++i) {
      rd[i] = time_kernel(true, bufs[i], size / 16, sink, 5);
      wr[i] = time_kernel(false, bufs[i], size / 16, sink, 5);
      printf("pass %d buffer %2d @ %p: read %.3f TB/s  write %.3f TB/s\n", pass, i, (void*)bufs[i], rd[i], wr[i]);
    }
  int const lo = (int)(std::min_element(rd.begin(), rd.end()) - rd.begin()), hi = (int)(std::max_element(rd.begin(), rd.end()) - rd.begin());
  size_t const piece = (size_t)128 << 20;
  for (int which : {lo, hi}) {
    printf("pieces of buffer %d (128 MiB each, read TB/s):", which);
    for (size_t off = 0; off + piece <= size; off += piece) printf(" %.2f", time_kernel(true, bufs[which] + off / 16, piece / 16, sink, 10));
    printf("\n");
  }
  return 0;
}
