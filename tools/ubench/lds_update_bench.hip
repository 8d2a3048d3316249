// LDS scatter-update throughput on gfx950: which instruction form can accumulate random-address partials fastest?
//   f32 atomic (ds_add_f32), u32 atomic (ds_add_u32), u64 atomic (ds_add_u64), returning f32 atomic, plain read/add/write
//   (racy: rate reference only), and gather-only (ds_read_b32).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_update_bench lds_update_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int ROWS = 8192;
constexpr int BLOCK = 512;

template <int MODE>
__global__ void __launch_bounds__(BLOCK) k(uint16_t const* idx, float const* val, int n_per_block, float* out)
{
  __shared__ unsigned long long acc64[ROWS];  // 64 KiB
  float* accf = reinterpret_cast<float*>(acc64);
  uint32_t* accu = reinterpret_cast<uint32_t*>(acc64);
  for (int i = threadIdx.x; i < ROWS; i += BLOCK) acc64[i] = 0;
  __syncthreads();
  uint16_t const* ip = idx + (size_t)blockIdx.x * n_per_block;
  float const* vp    = val + (size_t)blockIdx.x * n_per_block;
  float g = 0;
  for (int s = 8 * threadIdx.x; s < n_per_block; s += 8 * BLOCK) {
    uint4 d = *reinterpret_cast<uint4 const*>(ip + s);
    float4 p0 = *reinterpret_cast<float4 const*>(vp + s), p1 = *reinterpret_cast<float4 const*>(vp + s + 4);
    uint32_t w4[4] = {d.x, d.y, d.z, d.w};
    float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      uint32_t i = (kk & 1) ? (w4[kk >> 1] >> 16) : (w4[kk >> 1] & 0xFFFFu);
      if (MODE == 0) atomicAdd(&accf[i], v[kk]);
      if (MODE == 1) atomicAdd(&accu[i], __float_as_uint(v[kk]) >> 8);
      if (MODE == 2) atomicAdd(&acc64[i], (unsigned long long)__float_as_uint(v[kk]));
      if (MODE == 3) g += atomicAdd(&accf[i], v[kk]);
      if (MODE == 4) accf[i] += v[kk];
      if (MODE == 5) g += accf[i] * v[kk];
    }
  }
  __syncthreads();
  float s = g;
  for (int i = threadIdx.x; i < ROWS; i += BLOCK) s += accf[i] + (float)accu[2 * i + 1];
  if (s == 123.456f) out[blockIdx.x] = s;
}

int main()
{
  int const blocks = 1024, n_per_block = 1 << 17;  // 128 Mi updates per launch
  size_t n = (size_t)blocks * n_per_block;
  std::vector<uint16_t> hi(n);
  uint32_t st = 12345;
  for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; hi[i] = (uint16_t)((st >> 12) % ROWS); }
  uint16_t* di; float* dv; float* dout;
  hipMalloc(&di, n * 2); hipMalloc(&dv, n * 4); hipMalloc(&dout, blocks * 4);
  hipMemcpy(di, hi.data(), n * 2, hipMemcpyHostToDevice);
  hipMemset(dv, 0, n * 4);
  char const* names[6] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_add_rtn_f32", "read+add+write (racy)", "ds_read_b32 gather"};
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](int mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, blocks, BLOCK, 0, 0, di, dv, n_per_block, dout); break;
        case 1: hipLaunchKernelGGL(k<1>, blocks, BLOCK, 0, 0, di, dv, n_per_block, dout); break;
        case 2: hipLaunchKernelGGL(k<2>, blocks, BLOCK, 0, 0, di, dv, n_per_block, dout); break;
        case 3: hipLaunchKernelGGL(k<3>, blocks, BLOCK, 0, 0, di, dv, n_per_block, dout); break;
        case 4: hipLaunchKernelGGL(k<4>, blocks, BLOCK, 0, 0, di, dv, n_per_block, dout); break;
        case 5: hipLaunchKernelGGL(k<5>, blocks, BLOCK, 0, 0, di, dv, n_per_block, dout); break;
      }
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-26s %8.3f ms  %8.1f G updates/s  (stream %.0f GB/s)\n", names[mode], ms, n / ms / 1e6, n * 6.0 / ms / 1e6);
  };
  for (int m = 0; m < 6; ++m) run(m);
  return 0;
}
