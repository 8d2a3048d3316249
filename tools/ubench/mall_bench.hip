// Micro-benchmark: does a buffer that is written by one kernel and read by the next stay in the 256 MB Infinity Cache (MALL) of
// MI355X, and does it survive a stream of other data passing through in between?  The two-phase SpMV writes 1.24 GB of partials
// in phase 1 and reads them in phase 2: if the work were cut into destination stripes whose partials fit the MALL, the hand-off
// would not have to touch HBM.  Each round: write W bytes (k_write), optionally stream S bytes of other data (k_stream, plain or
// non-temporal loads), read the W bytes back (k_read).  Reported: time of the write and of the read-back, as GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_write(float4* p, size_t n, float v)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = float4{v, v, v, v};
}
template <bool NT>
__global__ void __launch_bounds__(256) k_read(float4 const* p, size_t n, float4* sink)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float4 acc = {0, 0, 0, 0};
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (; i < n; i += stride) {
    f4 t = NT ? __builtin_nontemporal_load(reinterpret_cast<f4 const*>(p) + i) : reinterpret_cast<f4 const*>(p)[i];
    acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
  }
  if (acc.x == 12345.678f) *sink = acc;
}

int main()
{
  size_t const big = (size_t)4 << 30;
  float4 *buf, *other, *sink;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&other, big)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, big)); CK(hipMemset(other, 0, big));
  hipEvent_t ev[4];
  for (int k = 0; k < 4; ++k) CK(hipEventCreate(&ev[k]));
  int const grid = 256 * 16;
  printf("%10s %12s %6s | %10s %10s   (GB/s of the W bytes)\n", "W MB", "stream MB", "nt", "write", "read-back");
  for (size_t wmb : {32, 64, 128, 160, 192, 256, 384, 1024}) {
    for (size_t smb : {0, 256, 512}) {
      for (int nt = 0; nt < 2; ++nt) {
        if (smb == 0 && nt) continue;
        size_t const nw = wmb * ((size_t)1 << 20) / 16, ns = smb * ((size_t)1 << 20) / 16;
        float best_w = 1e30f, best_r = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(ev[0]));
          hipLaunchKernelGGL(k_write, grid, 256, 0, 0, buf, nw, (float)rep);
          CK(hipEventRecord(ev[1]));
          if (ns) {
            if (nt) hipLaunchKernelGGL(k_read<true>, grid, 256, 0, 0, (float4 const*)other, ns, sink);
            else    hipLaunchKernelGGL(k_read<false>, grid, 256, 0, 0, (float4 const*)other, ns, sink);
          }
          CK(hipEventRecord(ev[2]));
          hipLaunchKernelGGL(k_read<false>, grid, 256, 0, 0, (float4 const*)buf, nw, sink);
          CK(hipEventRecord(ev[3]));
          CK(hipEventSynchronize(ev[3]));
          float tw, tr;
          CK(hipEventElapsedTime(&tw, ev[0], ev[1])); CK(hipEventElapsedTime(&tr, ev[2], ev[3]));
          if (rep > 0) { if (tw < best_w) best_w = tw; if (tr < best_r) best_r = tr; }
        }
        double const gb = (double)nw * 16 / 1e9;
        printf("%10zu %12zu %6s | %10.0f %10.0f\n", wmb, smb, smb ? (nt ? "yes" : "no") : "-", gb / best_w * 1e3, gb / best_r * 1e3);
      }
    }
  }
  return 0;
}
