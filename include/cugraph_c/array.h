/* Type-erased device / host arrays and views.  Replaces cpp/include/cugraph_c/array.h:42-326
 * (impl cpp/src/c_api/array.cpp).  Views BORROW a pointer (the caller, e.g. a cupy / torch tensor,
 * keeps ownership); arrays OWN device memory.  Result getters of the algorithms return a NEW heap
 * view on every call (cpp/src/c_api/array.hpp:70-73) which the caller frees.
 * copy_to_host synchronises (array.cpp:348-352); device->device view_copy also synchronises here
 * (the reference relies on stream-ordered RMM frees). */
#pragma once
#include <cugraph_c/export.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_type_erased_device_array_t;
typedef struct { int32_t align_; } cugraph_type_erased_device_array_view_t;
typedef struct { int32_t align_; } cugraph_type_erased_host_array_t;
typedef struct { int32_t align_; } cugraph_type_erased_host_array_view_t;

CUGRAPH_EXPORT cugraph_error_code_t cugraph_type_erased_device_array_create(
  const cugraph_resource_handle_t* handle, size_t n_elems, cugraph_data_type_id_t dtype,
  cugraph_type_erased_device_array_t** array, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_type_erased_device_array_create_from_view(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* view,
  cugraph_type_erased_device_array_t** array, cugraph_error_t** error);
CUGRAPH_EXPORT void cugraph_type_erased_device_array_free(cugraph_type_erased_device_array_t* p);
/* cpp/include/cugraph_c/array.h:85 (declared by the reference, compiled out there: cpp/src/c_api/array.cpp:105).  Gives up the
 * array's device storage: the caller owns the returned pointer and frees it with hipFree(); `p` stays valid as an empty array
 * and is still freed with cugraph_type_erased_device_array_free. */
CUGRAPH_EXPORT void* cugraph_type_erased_device_array_release(cugraph_type_erased_device_array_t* p);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_type_erased_device_array_view(
  cugraph_type_erased_device_array_t* array);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_type_erased_device_array_view_as_type(
  cugraph_type_erased_device_array_t* array, cugraph_data_type_id_t dtype,
  cugraph_type_erased_device_array_view_t** result_view, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_type_erased_device_array_view_create(
  void* pointer, size_t n_elems, cugraph_data_type_id_t dtype);
CUGRAPH_EXPORT void cugraph_type_erased_device_array_view_free(cugraph_type_erased_device_array_view_t* p);
CUGRAPH_EXPORT size_t cugraph_type_erased_device_array_view_size(const cugraph_type_erased_device_array_view_t* p);
CUGRAPH_EXPORT cugraph_data_type_id_t cugraph_type_erased_device_array_view_type(
  const cugraph_type_erased_device_array_view_t* p);
CUGRAPH_EXPORT const void* cugraph_type_erased_device_array_view_pointer(
  const cugraph_type_erased_device_array_view_t* p);

CUGRAPH_EXPORT cugraph_error_code_t cugraph_type_erased_host_array_create(
  const cugraph_resource_handle_t* handle, size_t n_elems, cugraph_data_type_id_t dtype,
  cugraph_type_erased_host_array_t** array, cugraph_error_t** error);
CUGRAPH_EXPORT void cugraph_type_erased_host_array_free(cugraph_type_erased_host_array_t* p);
/* array.h:207 / array.cpp:200: same for host arrays; the storage comes from malloc(), the caller frees it with free(). */
CUGRAPH_EXPORT void* cugraph_type_erased_host_array_release(cugraph_type_erased_host_array_t* p);
CUGRAPH_EXPORT cugraph_type_erased_host_array_view_t* cugraph_type_erased_host_array_view(
  cugraph_type_erased_host_array_t* array);
CUGRAPH_EXPORT cugraph_type_erased_host_array_view_t* cugraph_type_erased_host_array_view_create(
  void* pointer, size_t n_elems, cugraph_data_type_id_t dtype);
CUGRAPH_EXPORT void cugraph_type_erased_host_array_view_free(cugraph_type_erased_host_array_view_t* p);
CUGRAPH_EXPORT size_t cugraph_type_erased_host_array_size(const cugraph_type_erased_host_array_view_t* p);
CUGRAPH_EXPORT cugraph_data_type_id_t cugraph_type_erased_host_array_type(
  const cugraph_type_erased_host_array_view_t* p);
CUGRAPH_EXPORT void* cugraph_type_erased_host_array_pointer(const cugraph_type_erased_host_array_view_t* p);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_type_erased_host_array_view_copy(
  const cugraph_resource_handle_t* handle, cugraph_type_erased_host_array_view_t* dst,
  const cugraph_type_erased_host_array_view_t* src, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_type_erased_device_array_view_copy_from_host(
  const cugraph_resource_handle_t* handle, cugraph_type_erased_device_array_view_t* dst,
  const byte_t* h_src, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_type_erased_device_array_view_copy_to_host(
  const cugraph_resource_handle_t* handle, byte_t* h_dst,
  const cugraph_type_erased_device_array_view_t* src, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_type_erased_device_array_view_copy(
  const cugraph_resource_handle_t* handle, cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* src, cugraph_error_t** error);
#ifdef __cplusplus
}
#endif
