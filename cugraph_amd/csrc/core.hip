// Handle, error and type-erased array entry points of the drop-in libcugraph_c.so.
// Replaces cpp/src/c_api/resource_handle.cpp:11-39, cpp/src/c_api/error.cpp, cpp/src/c_api/array.cpp.
#include "common.hpp"
#include "comm.hpp"

using namespace cga;

extern "C" const char* cugraph_amd_version(void) { return "cugraph_amd 0.1.0 (gfx950, HIP, no Thrust/CUB)"; }

// ------------------------------------------------------------------------------------------ errors
extern "C" const char* cugraph_error_message(const cugraph_error_t* error)
{
  return error ? reinterpret_cast<err_obj_t const*>(error)->message.c_str() : nullptr;
}
extern "C" void cugraph_error_free(cugraph_error_t* error) { delete reinterpret_cast<err_obj_t*>(error); }

// ------------------------------------------------------------------------------------------ handle
extern "C" cugraph_resource_handle_t* cugraph_create_resource_handle(void* raft_handle)
{
  // A non-NULL argument is a raft::handle_t* in the reference (resource_handle.hpp:12-25), i.e. the object that carries the
  // communicator.  RAFT does not exist on this platform; the library's own communicator (cugraph_amd_comm_create, comm.hpp) takes
  // its place: the handle then reports that communicator's rank / size and the MG entry points become collective.
  comm_t* comm = nullptr;
  if (raft_handle != nullptr) {
    comm = static_cast<comm_t*>(raft_handle);
    if (comm->magic != kCommMagic) return nullptr;
  }
  try {
    auto h = std::make_unique<handle_t>();
    if (comm) { h->comm = comm; h->rank = comm->rank; h->comm_size = comm->size; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); return nullptr; }
    HIP_TRY(hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    h->num_cus       = prop.multiProcessorCount;
    h->lds_per_block = prop.sharedMemPerBlock;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = h->stream;
    HIP_TRY(hipHostMalloc(&h->pinned, 65536, hipHostMallocDefault));
    return reinterpret_cast<cugraph_resource_handle_t*>(h.release());
  } catch (...) {
    return nullptr;
  }
}

extern "C" int cugraph_resource_handle_get_comm_size(const cugraph_resource_handle_t* handle)
{
  return handle ? reinterpret_cast<handle_t const*>(handle)->comm_size : 0;
}
extern "C" int cugraph_resource_handle_get_rank(const cugraph_resource_handle_t* handle)
{
  return handle ? reinterpret_cast<handle_t const*>(handle)->rank : 0;
}

static void free_timers(handle_t* h)
{
  for (auto& kv : h->timers)
    for (auto& ev : kv.second.events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  h->timers.clear();
}

extern "C" void cugraph_free_resource_handle(cugraph_resource_handle_t* handle)
{
  if (!handle) return;
  auto* h = reinterpret_cast<handle_t*>(handle);
  (void)hipStreamSynchronize(h->stream);
  if (h->own_stream && h->own_stream != h->stream) (void)hipStreamSynchronize(h->own_stream);
  // device blocks that outlive the handle (graphs, plans, results freed later) must not record events on its stream any more
  pool_forget_stream(h->stream);
  pool_forget_stream(h->own_stream);
  free_timers(h);
  if (h->side_stream) {
    (void)hipStreamSynchronize(h->side_stream);
    pool_forget_stream(h->side_stream);
    (void)hipStreamDestroy(h->side_stream);
    (void)hipEventDestroy(h->side_ev[0]);
    (void)hipEventDestroy(h->side_ev[1]);
  }
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

// The library then enqueues everything on the caller's stream (e.g. torch's current stream, on which RCCL collectives are
// ordered too): no host synchronisation is needed between a collective and the kernels that consume its result.
extern "C" cugraph_error_code_t cugraph_amd_handle_set_stream(const cugraph_resource_handle_t* handle, void* hip_stream, cugraph_error_t** error)
{
  return guarded(error, [&] {
    auto* h = const_cast<handle_t*>(&H(handle));
    h->sync();
    if (hip_stream == nullptr) { h->stream = h->own_stream; h->stream_borrowed = false; }
    else { h->stream = static_cast<hipStream_t>(hip_stream); h->stream_borrowed = true; }
  });
}

extern "C" cugraph_error_code_t cugraph_amd_handle_sync(const cugraph_resource_handle_t* handle, cugraph_error_t** error)
{
  return guarded(error, [&] { H(handle).sync(); });
}

extern "C" void cugraph_amd_kernel_timing_enable(const cugraph_resource_handle_t* handle, bool_t on)
{
  if (handle) const_cast<handle_t*>(reinterpret_cast<handle_t const*>(handle))->timing = on == TRUE;
}
// One event pair around a whole REGION of work on the handle's stream (bench.py: the timed iterations with the per-launch events off,
// so that no hipEventCreate / hipEventRecord sits between the launches the wall clock measures).  get(family) then returns 1 and the region's ms.
extern "C" void cugraph_amd_kernel_timing_region_begin(const cugraph_resource_handle_t* handle, const char* family)
{
  if (!handle || !family) return;
  auto* h = const_cast<handle_t*>(reinterpret_cast<handle_t const*>(handle));
  hipEvent_t start = nullptr, stop = nullptr;
  if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) return;
  (void)hipEventRecord(start, h->stream);
  h->timers[family].events.emplace_back(start, stop);
}
extern "C" void cugraph_amd_kernel_timing_region_end(const cugraph_resource_handle_t* handle, const char* family)
{
  if (!handle || !family) return;
  auto* h = const_cast<handle_t*>(reinterpret_cast<handle_t const*>(handle));
  auto it = h->timers.find(family);
  if (it == h->timers.end() || it->second.events.empty()) return;
  (void)hipEventRecord(it->second.events.back().second, h->stream);
}
extern "C" void cugraph_amd_kernel_timing_reset(const cugraph_resource_handle_t* handle)
{
  if (!handle) return;
  auto* h = const_cast<handle_t*>(reinterpret_cast<handle_t const*>(handle));
  (void)hipStreamSynchronize(h->stream);
  free_timers(h);
}
extern "C" cugraph_error_code_t cugraph_amd_kernel_timing_get(const cugraph_resource_handle_t* handle, const char* family,
                                                              size_t* launches, double* total_ms, cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    h.sync();
    size_t n = 0;
    double ms = 0.0;
    auto it = h.timers.find(family ? family : "");
    if (it != h.timers.end()) {
      for (auto const& ev : it->second.events) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, ev.first, ev.second));
        ms += t;
        ++n;
      }
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
  });
}

extern "C" int cugraph_amd_set_pagerank_hot_tile(const cugraph_resource_handle_t* handle, int n_entries)
{
  if (!handle) return -1;
  auto* h  = const_cast<handle_t*>(reinterpret_cast<handle_t const*>(handle));
  int prev = h->pagerank_hot_tile;
  h->pagerank_hot_tile = n_entries;
  return prev;
}

extern "C" void cugraph_amd_last_traversal_stats(const cugraph_resource_handle_t* handle, cugraph_amd_traversal_stats_t* out)
{
  if (handle && out) *out = reinterpret_cast<handle_t const*>(handle)->last_stats;
}

// ----------------------------------------------------------------------------------- device arrays
extern "C" cugraph_error_code_t cugraph_type_erased_device_array_create(const cugraph_resource_handle_t* handle, size_t n_elems,
                                                                        cugraph_data_type_id_t dtype,
                                                                        cugraph_type_erased_device_array_t** array,
                                                                        cugraph_error_t** error)
{
  if (array) *array = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(array != nullptr, CUGRAPH_INVALID_INPUT, "array is NULL");
    CGA_EXPECTS(dtype_size(dtype) != 0, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "unsupported data type");
    HIP_TRY(hipSetDevice(h.device));
    *array = reinterpret_cast<cugraph_type_erased_device_array_t*>(new device_array_t(n_elems, dtype));
  });
}

extern "C" cugraph_error_code_t cugraph_type_erased_device_array_create_from_view(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* view,
  cugraph_type_erased_device_array_t** array, cugraph_error_t** error)
{
  if (array) *array = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    auto v            = V(view);
    CGA_EXPECTS(array != nullptr && v != nullptr, CUGRAPH_INVALID_INPUT, "array / view is NULL");
    auto a = std::make_unique<device_array_t>(v->size, v->type);
    if (v->size > 0) HIP_TRY(hipMemcpyAsync(a->buf.ptr, v->data, v->size * dtype_size(v->type), hipMemcpyDeviceToDevice, h.stream));
    h.sync();
    *array = reinterpret_cast<cugraph_type_erased_device_array_t*>(a.release());
  });
}

extern "C" void cugraph_type_erased_device_array_free(cugraph_type_erased_device_array_t* p)
{
  delete reinterpret_cast<device_array_t*>(p);
}

// array.h:85 / array.cpp:105 (declared by the reference, compiled out there because rmm::device_buffer cannot give up its
// pointer).  Here an array's storage is one hipMalloc block of its own: the block leaves the library's cache for good and the
// caller frees it with hipFree; the array object stays valid as an EMPTY array (size 0) until cugraph_type_erased_device_array_free.
extern "C" void* cugraph_type_erased_device_array_release(cugraph_type_erased_device_array_t* p)
{
  if (!p) return nullptr;
  auto* a   = reinterpret_cast<device_array_t*>(p);
  void* ptr = a->buf.ptr;
  a->buf.ptr = nullptr; a->buf.bytes = 0; a->buf.granted = 0;  // ownership moves to the caller: not returned to the pool
  a->size = 0;
  return ptr;
}

extern "C" cugraph_type_erased_device_array_view_t* cugraph_type_erased_device_array_view(cugraph_type_erased_device_array_t* array)
{
  if (!array) return nullptr;
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(reinterpret_cast<device_array_t*>(array)->new_view());
}

extern "C" cugraph_error_code_t cugraph_type_erased_device_array_view_as_type(cugraph_type_erased_device_array_t* array,
                                                                              cugraph_data_type_id_t dtype,
                                                                              cugraph_type_erased_device_array_view_t** result_view,
                                                                              cugraph_error_t** error)
{
  if (result_view) *result_view = nullptr;
  return guarded(error, [&] {
    auto a = reinterpret_cast<device_array_t*>(array);
    CGA_EXPECTS(a != nullptr && result_view != nullptr, CUGRAPH_INVALID_INPUT, "array / result_view is NULL");
    // array.cpp:187-208: reinterpretation is allowed between types of the same width only
    CGA_EXPECTS(dtype_size(dtype) == dtype_size(a->type), CUGRAPH_INVALID_INPUT, "Could not treat type erased device array as the requested type");
    *result_view = reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new device_array_view_t{a->buf.ptr, a->size, dtype});
  });
}

extern "C" cugraph_type_erased_device_array_view_t* cugraph_type_erased_device_array_view_create(void* pointer, size_t n_elems,
                                                                                                cugraph_data_type_id_t dtype)
{
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new device_array_view_t{pointer, n_elems, dtype});
}
extern "C" void cugraph_type_erased_device_array_view_free(cugraph_type_erased_device_array_view_t* p)
{
  delete reinterpret_cast<device_array_view_t*>(p);
}
extern "C" size_t cugraph_type_erased_device_array_view_size(const cugraph_type_erased_device_array_view_t* p)
{
  return p ? V(p)->size : 0;
}
extern "C" cugraph_data_type_id_t cugraph_type_erased_device_array_view_type(const cugraph_type_erased_device_array_view_t* p)
{
  return p ? V(p)->type : NTYPES;
}
extern "C" const void* cugraph_type_erased_device_array_view_pointer(const cugraph_type_erased_device_array_view_t* p)
{
  return p ? V(p)->data : nullptr;
}

// ------------------------------------------------------------------------------------- host arrays
extern "C" cugraph_error_code_t cugraph_type_erased_host_array_create(const cugraph_resource_handle_t*, size_t n_elems,
                                                                      cugraph_data_type_id_t dtype,
                                                                      cugraph_type_erased_host_array_t** array,
                                                                      cugraph_error_t** error)
{
  if (array) *array = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(array != nullptr, CUGRAPH_INVALID_INPUT, "array is NULL");
    CGA_EXPECTS(dtype_size(dtype) != 0, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "unsupported data type");
    void* mem = std::malloc(n_elems * dtype_size(dtype) + 1);  // malloc: cugraph_type_erased_host_array_release hands it to a C caller
    if (!mem) throw std::bad_alloc();
    auto a  = new host_array_t{host_array_t::storage_t(static_cast<uint8_t*>(mem)), n_elems, dtype};
    *array  = reinterpret_cast<cugraph_type_erased_host_array_t*>(a);
  });
}
extern "C" void cugraph_type_erased_host_array_free(cugraph_type_erased_host_array_t* p) { delete reinterpret_cast<host_array_t*>(p); }
// array.h:207 / array.cpp:200: the caller takes the storage over and frees it with free(); the array object stays valid, empty
extern "C" void* cugraph_type_erased_host_array_release(cugraph_type_erased_host_array_t* p)
{
  if (!p) return nullptr;
  auto* a = reinterpret_cast<host_array_t*>(p);
  a->size = 0;
  return a->data.release();
}
extern "C" cugraph_type_erased_host_array_view_t* cugraph_type_erased_host_array_view(cugraph_type_erased_host_array_t* array)
{
  if (!array) return nullptr;
  auto a = reinterpret_cast<host_array_t*>(array);
  return reinterpret_cast<cugraph_type_erased_host_array_view_t*>(new host_array_view_t{a->data.get(), a->size, a->type});
}
extern "C" cugraph_type_erased_host_array_view_t* cugraph_type_erased_host_array_view_create(void* pointer, size_t n_elems,
                                                                                            cugraph_data_type_id_t dtype)
{
  return reinterpret_cast<cugraph_type_erased_host_array_view_t*>(new host_array_view_t{pointer, n_elems, dtype});
}
extern "C" void cugraph_type_erased_host_array_view_free(cugraph_type_erased_host_array_view_t* p)
{
  delete reinterpret_cast<host_array_view_t*>(p);
}
extern "C" size_t cugraph_type_erased_host_array_size(const cugraph_type_erased_host_array_view_t* p)
{
  return p ? reinterpret_cast<host_array_view_t const*>(p)->size : 0;
}
extern "C" cugraph_data_type_id_t cugraph_type_erased_host_array_type(const cugraph_type_erased_host_array_view_t* p)
{
  return p ? reinterpret_cast<host_array_view_t const*>(p)->type : NTYPES;
}
extern "C" void* cugraph_type_erased_host_array_pointer(const cugraph_type_erased_host_array_view_t* p)
{
  return p ? reinterpret_cast<host_array_view_t const*>(p)->data : nullptr;
}
extern "C" cugraph_error_code_t cugraph_type_erased_host_array_view_copy(const cugraph_resource_handle_t*,
                                                                         cugraph_type_erased_host_array_view_t* dst,
                                                                         const cugraph_type_erased_host_array_view_t* src,
                                                                         cugraph_error_t** error)
{
  return guarded(error, [&] {
    auto d = reinterpret_cast<host_array_view_t*>(dst);
    auto s = reinterpret_cast<host_array_view_t const*>(src);
    CGA_EXPECTS(d && s, CUGRAPH_INVALID_INPUT, "dst / src is NULL");
    CGA_EXPECTS(d->size == s->size && d->type == s->type, CUGRAPH_INVALID_INPUT, "Type and size of src and dst must match");
    std::memcpy(d->data, s->data, s->size * dtype_size(s->type));
  });
}

// ------------------------------------------------------------------------------------------ copies
extern "C" cugraph_error_code_t cugraph_type_erased_device_array_view_copy_from_host(const cugraph_resource_handle_t* handle,
                                                                                     cugraph_type_erased_device_array_view_t* dst,
                                                                                     const byte_t* h_src, cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    auto d            = V(dst);
    CGA_EXPECTS(d != nullptr, CUGRAPH_INVALID_INPUT, "dst is NULL");
    size_t bytes = d->size * dtype_size(d->type);
    if (bytes == 0) return;
    CGA_EXPECTS(h_src != nullptr, CUGRAPH_INVALID_INPUT, "h_src is NULL");
    HIP_TRY(hipMemcpyAsync(d->data, h_src, bytes, hipMemcpyHostToDevice, h.stream));
    h.sync();  // h_src may be pageable and short-lived
  });
}

extern "C" cugraph_error_code_t cugraph_type_erased_device_array_view_copy_to_host(const cugraph_resource_handle_t* handle,
                                                                                   byte_t* h_dst,
                                                                                   const cugraph_type_erased_device_array_view_t* src,
                                                                                   cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    auto s            = V(src);
    CGA_EXPECTS(s != nullptr, CUGRAPH_INVALID_INPUT, "src is NULL");
    size_t bytes = s->size * dtype_size(s->type);
    if (bytes == 0) return;
    CGA_EXPECTS(h_dst != nullptr, CUGRAPH_INVALID_INPUT, "h_dst is NULL");
    HIP_TRY(hipMemcpyAsync(h_dst, s->data, bytes, hipMemcpyDeviceToHost, h.stream));
    h.sync();  // array.cpp:348-352 synchronises too
  });
}

extern "C" cugraph_error_code_t cugraph_type_erased_device_array_view_copy(const cugraph_resource_handle_t* handle,
                                                                           cugraph_type_erased_device_array_view_t* dst,
                                                                           const cugraph_type_erased_device_array_view_t* src,
                                                                           cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    auto d = V(dst);
    auto s = V(src);
    CGA_EXPECTS(d && s, CUGRAPH_INVALID_INPUT, "dst / src is NULL");
    CGA_EXPECTS(d->type == s->type, CUGRAPH_INVALID_INPUT, "type of src and dst must match");
    CGA_EXPECTS(d->size == s->size, CUGRAPH_INVALID_INPUT, "size of src and dst must match");
    size_t bytes = s->size * dtype_size(s->type);
    if (bytes == 0) return;
    // `dst` is the caller's array (pylibcugraph: a cupy array it has just created -- and zero-filled -- on ITS stream, the legacy default
    // stream; utils.pyx:162-196).  The reference orders its copy behind that work implicitly (RAFT's stream is a blocking stream); this
    // library's stream is non-blocking, so the order is made explicit: whatever the caller queued on the default stream runs first.
    // Found with two ranks sharing one GPU: the zero-fill of the result array landed AFTER the copy once in a few calls (round 4).
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpyAsync(d->data, s->data, bytes, hipMemcpyDeviceToDevice, h.stream));
    h.sync();  // callers free the source right after (pylibcugraph utils.pyx:162-196); see SURVEY section 9.8
  });
}

// cpp/src/c_api/dlpack_interop.cpp:25-95 (called by pylibcugraph/utils.pyx:127-133 for every array that enters the library)
#include "cugraph_c/dlpack_interop.h"
extern "C" cugraph_error_code_t cugraph_data_type_id_from_dlpack(const DLDataType* dlpack_dtype, cugraph_data_type_id_t* dtype, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(dlpack_dtype != nullptr, CUGRAPH_INVALID_INPUT, "dlpack_dtype cannot be NULL");
    CGA_EXPECTS(dtype != nullptr, CUGRAPH_INVALID_INPUT, "dtype cannot be NULL");
    CGA_EXPECTS(dlpack_dtype->lanes == 1, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "vectorized DLPack types (lanes > 1) are not supported");
    unsigned const code = dlpack_dtype->code, bits = dlpack_dtype->bits;
    CGA_EXPECTS(code == kDLInt || code == kDLUInt || code == kDLFloat || code == kDLBool, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "unsupported DLPack type code");
    int t = -1;
    if (code == kDLInt) t = bits == 8 ? INT8 : bits == 16 ? INT16 : bits == 32 ? INT32 : bits == 64 ? INT64 : -1;
    else if (code == kDLUInt) t = bits == 8 ? UINT8 : bits == 16 ? UINT16 : bits == 32 ? UINT32 : bits == 64 ? UINT64 : -1;
    else if (code == kDLFloat) t = bits == 32 ? FLOAT32 : bits == 64 ? FLOAT64 : -1;
    else t = bits == 8 ? BOOL : -1;
    CGA_EXPECTS(t >= 0, CUGRAPH_INVALID_INPUT, "unsupported DLPack dtype bit width for the given type code");
    *dtype = (cugraph_data_type_id_t)t;
  });
}

// ------------------------------------------------------------------------------------------------------------------
// caching device-memory pool behind dev_buf (see common.hpp)
#include <map>
#include <mutex>
namespace cga {
namespace {
// The stream of the API call this host thread is executing (set by H() at every entry point).  A freed block remembers the
// stream it was last used on together with an event recorded behind that use; a reuse from ANOTHER stream (a second handle in
// the same process, or a handle whose stream was swapped with cugraph_amd_handle_set_stream) waits for the event first.  A reuse
// on the same stream is ordered by the stream itself and costs nothing.
thread_local hipStream_t tl_stream = nullptr;
thread_local bool tl_stream_known  = false;

struct free_block_t {
  void* ptr;
  hipStream_t stream;
  hipEvent_t ev;  // nullptr: the freeing thread had no API stream (static destruction, host-only paths)
};

struct pool_t {
  std::mutex m;
  std::multimap<std::pair<int, size_t>, free_block_t> free_blocks;  // (device, size) -> block
  std::vector<hipEvent_t> spare_events;
  size_t cached{0};
  size_t max_cached{(size_t)128 << 30};  // a Louvain sweep at RMAT-26 recycles ~40 GB of sort buffers: a 32 GB cap made every sweep hipFree /
                                        // hipMalloc them (9.0 s instead of 2.3 s per run; 3.9 s with 96 GB: round 3); what other allocators of the process
                                        // need can be returned with cugraph_amd_memory_pool_trim (an eager trim at the end of every
                                        // graph / plan construction was tried: the hipFree of tens of GB stalls for seconds now and then)
  bool enabled{true};
  bool debug{false};
  pool_t()
  {
    if (char const* e = getenv("CUGRAPH_AMD_POOL")) enabled = atoi(e) != 0;
    if (char const* e = getenv("CUGRAPH_AMD_POOL_MAX_GB")) max_cached = (size_t)std::max(0.0, atof(e)) << 30;
    else {  // never more than 45 % of the device
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) max_cached = std::min(max_cached, total_b / 20 * 9);
      else (void)hipGetLastError();
    }
    debug = getenv("CUGRAPH_AMD_POOL_DEBUG") != nullptr;
  }
  hipEvent_t get_event()
  {
    if (!spare_events.empty()) { hipEvent_t e = spare_events.back(); spare_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
  }
  void put_event(hipEvent_t e)
  {
    if (!e) return;
    if (spare_events.size() < 256) spare_events.push_back(e); else (void)hipEventDestroy(e);
  }
  void drop_locked(std::multimap<std::pair<int, size_t>, free_block_t>::iterator it)
  {
    if (it->second.ev) { (void)hipEventSynchronize(it->second.ev); put_event(it->second.ev); }
    (void)hipFree(it->second.ptr);
    cached -= it->first.second;
    free_blocks.erase(it);
  }
  void trim_locked(size_t keep)
  {  // largest blocks first
    while (cached > keep && !free_blocks.empty()) drop_locked(std::prev(free_blocks.end()));
  }
  size_t trim_blocks_above_locked(size_t block_bytes)
  {
    size_t freed = 0;
    for (auto it = free_blocks.begin(); it != free_blocks.end();) {
      auto cur = it++;
      if (cur->first.second > block_bytes) { freed += cur->first.second; drop_locked(cur); }
    }
    return freed;
  }
};
pool_t& pool()
{
  static pool_t* p = new pool_t();  // leaked on purpose: buffers may be released during static destruction
  return *p;
}
size_t round_size(size_t n)
{
  if (n <= 256) return 256;
  if (n < ((size_t)1 << 20)) {  // next power of two
    size_t r = 256;
    while (r < n) r <<= 1;
    return r;
  }
  size_t const g = (size_t)2 << 20;  // 2 MiB granules
  return (n + g - 1) / g * g;
}
}  // namespace

void pool_set_stream(hipStream_t s) noexcept { tl_stream = s; tl_stream_known = true; }
void pool_forget_stream(hipStream_t s) noexcept
{  // the stream is about to be destroyed (it has been synchronised): later frees of this thread have no stream to order against ...
  if (s == nullptr || (tl_stream_known && tl_stream == s)) { tl_stream = nullptr; tl_stream_known = false; }  // (nullptr: "this thread's frees have no stream now")
  if (s == nullptr) return;
  // ... and the cached blocks that were freed on it are plainly free now: their events go back to the spare list (an event that outlives the stream it was
  // recorded on made a later hipEventRecord on ANOTHER stream of the device fault inside the runtime's walk over the active streams -- round 6: a second
  // resource handle of a communicator created, used and destroyed between two traversals)
  pool_t& p = pool();
  std::lock_guard<std::mutex> lock(p.m);
  for (auto& kv : p.free_blocks)
    if (kv.second.stream == s) {
      if (kv.second.ev) { (void)hipEventSynchronize(kv.second.ev); p.put_event(kv.second.ev); kv.second.ev = nullptr; }
      kv.second.stream = nullptr;
    }
}

void* pool_alloc(size_t n_bytes, size_t* granted)
{
  pool_t& p = pool();
  int dev   = 0;
  (void)hipGetDevice(&dev);
  size_t const want = p.enabled ? round_size(n_bytes) : n_bytes;
  if (p.enabled) {
    free_block_t blk{nullptr, nullptr, nullptr};
    {
      std::lock_guard<std::mutex> lock(p.m);
      auto it = p.free_blocks.lower_bound({dev, want});
      // best fit within 25 % slack; large requests (>= 256 MiB) take a cached block of up to twice their size: the levels of a
      // Louvain run and the passes of a graph build ask for a sequence of shrinking multi-GB buffers, and a miss there means a
      // synchronous hipMalloc of gigabytes (and, once the cache is over its cap, hipFrees) in the middle of the algorithm
      size_t const slack = want >= ((size_t)256 << 20) ? want : want / 4;
      if (it != p.free_blocks.end() && it->first.first == dev && it->first.second <= want + slack) {
        blk      = it->second;
        *granted = it->first.second;
        p.cached -= it->first.second;
        p.free_blocks.erase(it);
      }
    }
    if (blk.ptr) {
      if (blk.ev) {
        // last used on another stream (or the requester's stream is unknown): order the reuse behind that use
        if (!tl_stream_known || blk.stream != tl_stream) (void)hipEventSynchronize(blk.ev);
        std::lock_guard<std::mutex> lock(p.m);
        p.put_event(blk.ev);
      }
      return blk.ptr;
    }
  }
  void* ptr    = nullptr;
  hipError_t e = hipMalloc(&ptr, want);
  if (e != hipSuccess && p.enabled) {  // give the cached blocks back to the driver and try once more
    (void)hipGetLastError();
    { std::lock_guard<std::mutex> lock(p.m); p.trim_locked(0); }
    e = hipMalloc(&ptr, want);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw api_error(CUGRAPH_ALLOC_ERROR, "hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
  }
  *granted = want;
  return ptr;
}

void pool_free(void* ptr, size_t granted) noexcept
{
  if (!ptr) return;
  pool_t& p = pool();
  if (!p.enabled || granted > p.max_cached) { (void)hipFree(ptr); return; }
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, ptr) == hipSuccess) dev = attr.device; else (void)hipGetLastError();
  free_block_t blk{ptr, tl_stream, nullptr};
  if (p.debug && tl_stream_known && hipStreamQuery(tl_stream) == hipErrorNotReady)
    fprintf(stderr, "[pool] block %p (%zu bytes) freed while its stream still has work queued (reuse is ordered by the recorded event)\n", ptr, granted);
  (void)hipGetLastError();
  std::lock_guard<std::mutex> lock(p.m);
  if (tl_stream_known) {
    blk.ev = p.get_event();
    if (blk.ev && hipEventRecord(blk.ev, tl_stream) != hipSuccess) { (void)hipGetLastError(); p.put_event(blk.ev); blk.ev = nullptr; (void)hipDeviceSynchronize(); }
  }
  p.free_blocks.insert({{dev, granted}, blk});
  p.cached += granted;
  if (p.cached > p.max_cached) p.trim_locked(p.max_cached / 2);
}

// Returns the cached blocks above `block_bytes` to the driver (cugraph_amd_memory_pool_trim_large): for callers that share the
// device with another allocator and want the sort buffers of a finished graph / plan construction back without losing the small
// and medium blocks the per-call paths (BFS / SSSP state, result columns) recycle.  Not done eagerly: hipFree of tens of GB takes
// seconds now and then (graph build 0.18 s -> 5.4 s in one of five runs when every build ended with it).
size_t pool_release_large_blocks(size_t block_bytes) noexcept
{
  pool_t& p = pool();
  std::lock_guard<std::mutex> lock(p.m);
  return p.trim_blocks_above_locked(block_bytes);
}
}  // namespace cga

// releases every cached device block to the driver (returns the number of bytes released)
extern "C" size_t cugraph_amd_memory_pool_trim(void)
{
  cga::pool_t& p = cga::pool();
  std::lock_guard<std::mutex> lock(p.m);
  size_t const had = p.cached;
  p.trim_locked(0);
  return had;
}
extern "C" size_t cugraph_amd_memory_pool_trim_large(size_t block_bytes) { return cga::pool_release_large_blocks(block_bytes); }
extern "C" size_t cugraph_amd_memory_pool_cached_bytes(void)
{
  cga::pool_t& p = cga::pool();
  std::lock_guard<std::mutex> lock(p.m);
  return p.cached;
}
