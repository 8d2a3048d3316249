// Micro-benchmark: random 4-byte gather throughput on MI355X as a function of the table size (L1 / L2 / MALL / HBM
// resident) and load flavour.  Used to set the ceiling for the PageRank cold-source gathers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>  // 0 plain, 1 nontemporal, 2 index-stream + gather (idx read from memory)
__global__ void __launch_bounds__(1024, 8) k_gather(float const* x, uint32_t mask, int const* idx, int64_t n_per_thread, float* out)
{
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t nthreads = gridDim.x * blockDim.x;
  float acc = 0;
  for (int64_t it = 0; it < n_per_thread; it += 8) {
    uint32_t i[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 2) i[k] = (uint32_t)idx[(it + k) * (int64_t)nthreads + tid] & mask;
      else i[k] = hash32(tid * 2654435761u + (uint32_t)(it + k) * 40503u) & mask;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += (MODE == 1) ? __builtin_nontemporal_load(x + i[k]) : x[i[k]];
  }
  if (acc == 12345.678f) out[tid] = acc;
}

// same number of lanes, but lanes (4j..4j+3) read four consecutive floats of one random 16-byte slot
__global__ void __launch_bounds__(1024, 8) k_gather_quad(float const* x, uint32_t mask, int64_t n_per_thread, float* out)
{
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0;
  for (int64_t it = 0; it < n_per_thread; it += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      uint32_t i = (hash32((tid >> 2) * 2654435761u + (uint32_t)(it + k) * 40503u) & mask & ~3u) | (tid & 3);
      acc += x[i];
    }
  }
  if (acc == 12345.678f) out[tid] = acc;
}

int main()
{
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  printf("device %s CUs %d LDS/block %zu clock %d kHz\n", p.name, p.multiProcessorCount, p.sharedMemPerBlock, p.clockRate);
  size_t maxn = (size_t)1 << 28;  // 1 GiB of floats
  float* x; CK(hipMalloc(&x, maxn * 4)); CK(hipMemset(x, 0, maxn * 4));
  float* out; CK(hipMalloc(&out, 1 << 24));
  int grid = p.multiProcessorCount * 2, block = 1024;
  int64_t nthreads = (int64_t)grid * block;
  int64_t per = 128;
  int* idx; CK(hipMalloc(&idx, nthreads * per * 4));
  { std::vector<int> h(nthreads * per); uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int)(s >> 4); } CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int lg = 13; lg <= 28; ++lg) {
    uint32_t mask = (1u << lg) - 1;
    float ms[4] = {0, 0, 0, 0};
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_gather<0>, grid, block, 0, 0, x, mask, idx, per, out);
        if (mode == 1) hipLaunchKernelGGL(k_gather<1>, grid, block, 0, 0, x, mask, idx, per, out);
        if (mode == 2) hipLaunchKernelGGL(k_gather<2>, grid, block, 0, 0, x, mask, idx, per, out);
        if (mode == 3) hipLaunchKernelGGL(k_gather_quad, grid, block, 0, 0, x, mask, per, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
    }
    double g = (double)nthreads * per / 1e9;
    printf("table %8.2f MiB : plain %7.1f  nontemporal %7.1f  idx-stream+gather %7.1f  quad-coalesced %7.1f  Ggathers/s\n",
           (double)(1u << lg) * 4 / 1048576.0, g / (ms[0] * 1e-3), g / (ms[1] * 1e-3), g / (ms[2] * 1e-3), g / (ms[3] * 1e-3));
  }
  return 0;
}
