// On-device RMAT edge-list generator (benchmark input; include/cugraph_amd/extensions.h).
// Algorithm: cpp/src/generators/generate_rmat_edgelist.cuh:85-103 (no clip-and-flip, no scramble).
// RNG: counter-based splitmix64 on (seed, edge, bit) compared against 32-bit integer thresholds --
// bit-identical to oracle/oracle.c:orc_rmat, so the GPU path and the CPU oracle see the same edge list.
#include "common.hpp"

namespace cga {
namespace {

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t counter)
{
  uint64_t z = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
  z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void k_rmat(int scale, uint64_t first_edge, uint64_t num_edges, uint32_t t_ab, uint32_t t_an, uint32_t t_cn, uint64_t seed,
                       int32_t* src, int32_t* dst)
{
  uint64_t k      = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; k < num_edges; k += stride) {
    uint64_t i = first_edge + k;
    int32_t s = 0, d = 0;
    for (int bit = scale - 1; bit >= 0; --bit) {
      uint64_t z  = splitmix64_at(seed, i * 64ull + (uint64_t)bit);
      uint32_t r0 = (uint32_t)(z >> 32), r1 = (uint32_t)z;
      int sb      = r0 > t_ab;
      int db      = r1 > (sb ? t_cn : t_an);
      s |= sb << bit;
      d |= db << bit;
    }
    src[k] = s;
    dst[k] = d;
  }
}

uint32_t prob_to_u32(double p)
{
  if (p <= 0.0) return 0u;
  if (p >= 1.0) return 0xFFFFFFFFu;
  return (uint32_t)(p * 4294967296.0);
}

}  // namespace
}  // namespace cga

using namespace cga;

extern "C" cugraph_error_code_t cugraph_amd_generate_rmat_edgelist(const cugraph_resource_handle_t* handle, size_t scale, size_t first_edge,
                                                                   size_t num_edges, double a, double b, double c, uint64_t seed,
                                                                   cugraph_type_erased_device_array_view_t* src,
                                                                   cugraph_type_erased_device_array_view_t* dst, cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    auto s = V(src);
    auto d = V(dst);
    CGA_EXPECTS(s && d && s->type == INT32 && d->type == INT32, CUGRAPH_INVALID_INPUT, "src / dst must be INT32 views");
    CGA_EXPECTS(s->size >= num_edges && d->size >= num_edges, CUGRAPH_INVALID_INPUT, "src / dst views are smaller than num_edges");
    CGA_EXPECTS(scale >= 1 && scale <= 30, CUGRAPH_INVALID_INPUT, "Invalid input argument: scale too large for vertex_t.");
    CGA_EXPECTS(a >= 0.0 && b >= 0.0 && c >= 0.0 && a + b + c <= 1.0, CUGRAPH_INVALID_INPUT,
                "Invalid input argument: a, b, c should be non-negative and a + b + c should not be larger than 1.0.");
    double a_plus_b = a + b;
    double a_norm   = a_plus_b > 0.0 ? a / a_plus_b : 0.0;
    double c_norm   = (1.0 - a_plus_b) > 0.0 ? c / (1.0 - a_plus_b) : 0.0;
    HIP_TRY(hipSetDevice(h.device));
    if (num_edges > 0)
      hipLaunchKernelGGL(k_rmat, grid_for((int64_t)num_edges, kBlock, 16384), kBlock, 0, h.stream, (int)scale, (uint64_t)first_edge,
                         (uint64_t)num_edges, prob_to_u32(a_plus_b), prob_to_u32(a_norm), prob_to_u32(c_norm), seed, s->as<int32_t>(),
                         d->as<int32_t>());
    h.sync();
  });
}
