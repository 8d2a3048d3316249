// On-device RMAT edge-list generator (benchmark input; include/cugraph_amd/extensions.h).
// Algorithm: cpp/src/generators/generate_rmat_edgelist.cuh:85-103 (no clip-and-flip, no scramble).
// RNG: counter-based splitmix64 on (seed, edge, bit) compared against 32-bit integer thresholds --
// bit-identical to oracle/oracle.c:orc_rmat, so the GPU path and the CPU oracle see the same edge list.
#include "common.hpp"

#include "cugraph_c/graph_generators.h"

#include <cmath>

namespace cga {
namespace {

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t counter)
{
  uint64_t z = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
  z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Vertex-id permutation of the Graph500 generator (cpp/src/generators/scramble.cuh:41-67 uses the same construction): two
// rounds of (add, multiply by an odd constant, reverse the low lgN bits) -- each step is a bijection on lgN-bit values.
__device__ __forceinline__ int32_t scramble_id(int32_t value, int lgN)
{
  uint32_t const c0 = 282475248u, c1 = 2617694917u;
  uint32_t v = (uint32_t)value;
  v += c0 + c1;
  v *= (c0 | 0x11493211u);
  v = __brev(v) >> (32 - lgN);
  v *= (c1 | 0x02C843A5u);
  v = __brev(v) >> (32 - lgN);
  return (int32_t)v;
}

__global__ void k_rmat(int scale, uint64_t first_edge, uint64_t num_edges, uint32_t t_ab, uint32_t t_an, uint32_t t_cn, uint64_t seed,
                       int32_t* src, int32_t* dst, int clip_and_flip = 0, int scramble = 0)
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
      if (clip_and_flip && s == d && !sb && db) { sb = 1; db = 0; }  // generate_rmat_edgelist.cuh:90-97: keep the lower triangle
      s |= sb << bit;
      d |= db << bit;
    }
    if (scramble) { s = scramble_id(s, scale); d = scramble_id(d, scale); }
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

// ------------------------------------------------------------------------------------------------------------------
// The reference's generator API: cugraph_rng_state_*, cugraph_generate_rmat_edgelist,
// cugraph_generate_edge_weights, cugraph_coo_*.  Replaces cpp/src/c_api/random.cpp, cpp/src/c_api/graph_generators.cpp:24-330
// (cugraph::generate_rmat_edgelist, cpp/src/generators/generate_rmat_edgelist.cuh:38-112).  The reference draws its uniforms
// from raft::random (not vendored, so its streams cannot be reproduced); here the state is (seed, number of edges drawn so
// far) of the counter-based generator above, so a fresh state with seed s yields exactly the edge list of
// cugraph_amd_generate_rmat_edgelist(seed = s) and of the CPU oracle.
namespace cga {
struct rng_state_t {
  uint64_t seed{0};
  uint64_t drawn{0};
};
namespace {
template <typename T>
__global__ void k_uniform_weights(T* w, uint64_t n, uint64_t seed, uint64_t first, double lo, double hi)
{
  uint64_t k      = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; k < n; k += stride) {
    uint64_t const z = splitmix64_at(seed ^ 0x5851F42D4C957F2Dull, first + k);
    double const u   = (double)(z >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
    w[k]             = (T)(lo + (hi - lo) * u);
  }
}
}  // namespace
}  // namespace cga

extern "C" cugraph_error_code_t cugraph_rng_state_create(const cugraph_resource_handle_t* handle, uint64_t seed, cugraph_rng_state_t** state,
                                                         cugraph_error_t** error)
{
  if (state) *state = nullptr;
  return guarded(error, [&] {
    (void)H(handle);
    CGA_EXPECTS(state != nullptr, CUGRAPH_INVALID_INPUT, "state is NULL");
    *state = reinterpret_cast<cugraph_rng_state_t*>(new rng_state_t{seed, 0});
  });
}
extern "C" void cugraph_rng_state_free(cugraph_rng_state_t* p) { delete reinterpret_cast<rng_state_t*>(p); }

extern "C" cugraph_error_code_t cugraph_generate_rmat_edgelist(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state, size_t scale,
                                                               size_t num_edges, double a, double b, double c, bool_t clip_and_flip,
                                                               bool_t scramble_vertex_ids, cugraph_coo_t** result, cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(rng_state != nullptr && result != nullptr, CUGRAPH_INVALID_INPUT, "rng_state / result is NULL");
    CGA_EXPECTS(scale >= 1 && scale <= 30, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "Invalid input argument: scale too large for the 32-bit vertex type of this build.");
    CGA_EXPECTS(a >= 0.0 && b >= 0.0 && c >= 0.0 && a + b + c <= 1.0, CUGRAPH_INVALID_INPUT,
                "Invalid input argument: a, b, c should be non-negative and a + b + c should not be larger than 1.0.");
    rng_state_t& st = *reinterpret_cast<rng_state_t*>(rng_state);
    HIP_TRY(hipSetDevice(h.device));
    auto coo = std::make_unique<coo_t>();
    coo->src = new device_array_t(num_edges, INT32);
    coo->dst = new device_array_t(num_edges, INT32);
    double const a_plus_b = a + b;
    double const a_norm   = a_plus_b > 0.0 ? a / a_plus_b : 0.0;
    double const c_norm   = (1.0 - a_plus_b) > 0.0 ? c / (1.0 - a_plus_b) : 0.0;
    if (num_edges > 0)
      hipLaunchKernelGGL(k_rmat, grid_for((int64_t)num_edges, kBlock, 16384), kBlock, 0, h.stream, (int)scale, st.drawn, (uint64_t)num_edges,
                         prob_to_u32(a_plus_b), prob_to_u32(a_norm), prob_to_u32(c_norm), st.seed, coo->src->buf.as<int32_t>(),
                         coo->dst->buf.as<int32_t>(), clip_and_flip == TRUE ? 1 : 0, scramble_vertex_ids == TRUE ? 1 : 0);
    h.sync();
    st.drawn += num_edges;
    *result = reinterpret_cast<cugraph_coo_t*>(coo.release());
  });
}

extern "C" cugraph_error_code_t cugraph_generate_edge_weights(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state, cugraph_coo_t* coo,
                                                              cugraph_data_type_id_t dtype, double minimum_weight, double maximum_weight,
                                                              cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(rng_state != nullptr && coo != nullptr, CUGRAPH_INVALID_INPUT, "rng_state / coo is NULL");
    CGA_EXPECTS(dtype == FLOAT32 || dtype == FLOAT64, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "weights must be FLOAT32 or FLOAT64");
    rng_state_t& st = *reinterpret_cast<rng_state_t*>(rng_state);
    coo_t& c        = *reinterpret_cast<coo_t*>(coo);
    size_t const n  = c.src ? c.src->size : 0;
    HIP_TRY(hipSetDevice(h.device));
    delete c.wgt;
    c.wgt = new device_array_t(n, dtype);
    if (n > 0) {
      int const g = grid_for((int64_t)n, kBlock, 16384);
      if (dtype == FLOAT32) hipLaunchKernelGGL(k_uniform_weights<float>, g, kBlock, 0, h.stream, c.wgt->buf.as<float>(), (uint64_t)n, st.seed, st.drawn, minimum_weight, maximum_weight);
      else hipLaunchKernelGGL(k_uniform_weights<double>, g, kBlock, 0, h.stream, c.wgt->buf.as<double>(), (uint64_t)n, st.seed, st.drawn, minimum_weight, maximum_weight);
    }
    h.sync();
    st.drawn += n;
  });
}

static cugraph_type_erased_device_array_view_t* coo_view(device_array_t* a)
{
  return a ? reinterpret_cast<cugraph_type_erased_device_array_view_t*>(a->new_view()) : nullptr;
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_coo_get_sources(cugraph_coo_t* coo) { return coo_view(reinterpret_cast<coo_t*>(coo)->src); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_coo_get_destinations(cugraph_coo_t* coo) { return coo_view(reinterpret_cast<coo_t*>(coo)->dst); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_coo_get_edge_weights(cugraph_coo_t* coo) { return coo_view(reinterpret_cast<coo_t*>(coo)->wgt); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_coo_get_edge_id(cugraph_coo_t* coo) { return coo_view(reinterpret_cast<coo_t*>(coo)->ids); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_coo_get_edge_type(cugraph_coo_t* coo) { return coo_view(reinterpret_cast<coo_t*>(coo)->types); }
extern "C" void cugraph_coo_free(cugraph_coo_t* coo) { delete reinterpret_cast<coo_t*>(coo); }
extern "C" size_t cugraph_coo_list_size(const cugraph_coo_list_t* coo_list) { return reinterpret_cast<coo_list_t const*>(coo_list)->list.size(); }
extern "C" cugraph_coo_t* cugraph_coo_list_element(cugraph_coo_list_t* coo_list, size_t index)
{  // borrowed: the list owns its elements (cpp/src/c_api/graph_generators.cpp: cugraph_coo_list_element)
  auto& l = reinterpret_cast<coo_list_t*>(coo_list)->list;
  return index < l.size() ? reinterpret_cast<cugraph_coo_t*>(l[index]) : nullptr;
}
extern "C" void cugraph_coo_list_free(cugraph_coo_list_t* coo_list) { delete reinterpret_cast<coo_list_t*>(coo_list); }

// n_edgelists RMAT lists (cugraph::generate_rmat_edgelists, cpp/src/generators/generate_rmat_edgelist.cuh:114-190): list i has
// scale s_i drawn from [min_scale, max_scale] (UNIFORM) or min_scale + floor(range * Exp(4)) mod range (POWER_LAW) and -- as the
// reference passes it -- s_i * edge_factor edges, (a, b, c) = (0.57, 0.19, 0.19) for POWER_LAW edges, (0.25, 0.25, 0.25) for UNIFORM
extern "C" cugraph_error_code_t cugraph_generate_rmat_edgelists(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state, size_t n_edgelists,
                                                                size_t min_scale, size_t max_scale, size_t edge_factor,
                                                                cugraph_generator_distribution_t size_distribution,
                                                                cugraph_generator_distribution_t edge_distribution, bool_t clip_and_flip,
                                                                bool_t scramble_vertex_ids, cugraph_coo_list_t** result, cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    (void)H(handle);
    CGA_EXPECTS(rng_state != nullptr && result != nullptr, CUGRAPH_INVALID_INPUT, "rng_state / result is NULL");
    CGA_EXPECTS(min_scale > 0, CUGRAPH_INVALID_INPUT, "minimum graph scale is 1.");
    CGA_EXPECTS(max_scale >= min_scale, CUGRAPH_INVALID_INPUT, "Invalid input argument: max_scale is smaller than min_scale.");
    CGA_EXPECTS(max_scale <= 30, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "Invalid input argument: scale too large for vertex_t.");
    rng_state_t& st = *reinterpret_cast<rng_state_t*>(rng_state);
    double const a = edge_distribution == UNIFORM ? 0.25 : 0.57, b = edge_distribution == UNIFORM ? 0.25 : 0.19, c = b;
    auto host_u01 = [&](uint64_t k) {  // the same counter-based stream as the device kernels, on the host
      uint64_t z = (st.seed ^ 0xD1B54A32D192ED03ull) + (k + 1) * 0x9E3779B97F4A7C15ull;
      z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    };
    auto out = std::make_unique<coo_list_t>();
    size_t const range = max_scale - min_scale;
    for (size_t i = 0; i < n_edgelists; ++i) {
      double const u = host_u01(st.drawn++);
      size_t scale;
      if (size_distribution == UNIFORM) scale = min_scale + std::min<size_t>(range, (size_t)(u * (double)(range + 1)));
      else scale = range == 0 ? min_scale : min_scale + (size_t)((double)range * (-std::log(1.0 - u) / 4.0)) % range;
      cugraph_coo_t* one       = nullptr;
      cugraph_error_t* err     = nullptr;
      cugraph_error_code_t const code = cugraph_generate_rmat_edgelist(handle, rng_state, scale, scale * edge_factor, a, b, c, clip_and_flip, scramble_vertex_ids, &one, &err);
      if (code != CUGRAPH_SUCCESS) {
        std::string msg = err ? cugraph_error_message(err) : "cugraph_generate_rmat_edgelist failed";
        cugraph_error_free(err);
        throw api_error(code, msg);
      }
      out->list.push_back(reinterpret_cast<coo_t*>(one));
    }
    *result = reinterpret_cast<cugraph_coo_list_t*>(out.release());
  });
}

namespace cga {
namespace {
__global__ void k_iota_i32(int32_t* out, uint64_t n)
{
  uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  for (; k < n; k += stride) out[k] = (int32_t)k;
}
__global__ void k_uniform_i32(int32_t* out, uint64_t n, uint64_t seed, uint64_t first, int32_t lo, uint64_t span)
{
  uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  for (; k < n; k += stride) {
    uint64_t const z = splitmix64_at(seed ^ 0xA0761D6478BD642Full, first + k);
    out[k]           = (int32_t)((int64_t)lo + (int64_t)(((z >> 32) * span) >> 32));  // uniform in [lo, lo + span), span <= 2^32
  }
}
}  // namespace
}  // namespace cga

// edge ids 0 .. E-1 in the vertex type (cpp/src/c_api/graph_generators.cpp: cugraph_generate_edge_ids; detail::sequence_fill).
// multi_gpu: every rank numbers from base_edge_id = sum of the lower ranks' sizes; a one-rank handle starts at 0 either way.
extern "C" cugraph_error_code_t cugraph_generate_edge_ids(const cugraph_resource_handle_t* handle, cugraph_coo_t* coo, bool_t multi_gpu, cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(coo != nullptr, CUGRAPH_INVALID_INPUT, "coo is NULL");
    CGA_EXPECTS(multi_gpu == FALSE || h.comm_size == 1, CUGRAPH_NOT_IMPLEMENTED,
                "cugraph_generate_edge_ids(multi_gpu = TRUE) needs a multi-rank resource handle; this library exchanges data through the host layer (cugraph_amd/mg.py)");
    coo_t& c       = *reinterpret_cast<coo_t*>(coo);
    size_t const n = c.src ? c.src->size : 0;
    CGA_EXPECTS(n < ((size_t)1 << 31), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "edge ids do not fit the 32-bit edge type of this build");
    HIP_TRY(hipSetDevice(h.device));
    delete c.ids;
    c.ids = new device_array_t(n, INT32);
    if (n > 0) hipLaunchKernelGGL(k_iota_i32, grid_for((int64_t)n, kBlock, 16384), kBlock, 0, h.stream, c.ids->buf.as<int32_t>(), (uint64_t)n);
    h.sync();
  });
}

// uniform INT32 edge types in [min_edge_type, max_edge_type] (cugraph_generate_edge_types; detail::uniform_random_fill)
extern "C" cugraph_error_code_t cugraph_generate_edge_types(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state, cugraph_coo_t* coo,
                                                            int32_t min_edge_type, int32_t max_edge_type, cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(rng_state != nullptr && coo != nullptr, CUGRAPH_INVALID_INPUT, "rng_state / coo is NULL");
    CGA_EXPECTS(max_edge_type >= min_edge_type, CUGRAPH_INVALID_INPUT, "Invalid input argument: max_edge_type is smaller than min_edge_type.");
    rng_state_t& st = *reinterpret_cast<rng_state_t*>(rng_state);
    coo_t& c        = *reinterpret_cast<coo_t*>(coo);
    size_t const n  = c.src ? c.src->size : 0;
    HIP_TRY(hipSetDevice(h.device));
    delete c.types;
    c.types = new device_array_t(n, INT32);
    uint64_t const span = (uint64_t)((int64_t)max_edge_type - (int64_t)min_edge_type + 1);  // up to 2^32: 64 bits
    if (n > 0)
      hipLaunchKernelGGL(k_uniform_i32, grid_for((int64_t)n, kBlock, 16384), kBlock, 0, h.stream, c.types->buf.as<int32_t>(), (uint64_t)n, st.seed, st.drawn,
                         min_edge_type, span);
    h.sync();
    st.drawn += n;
  });
}
