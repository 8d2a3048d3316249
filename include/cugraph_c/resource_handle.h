/* Resource handle.  Replaces cpp/include/cugraph_c/resource_handle.h:20-70
 * (impl cpp/src/c_api/resource_handle.cpp:11-39).
 *
 * cugraph_create_resource_handle(NULL) makes the library allocate its own single-GPU context on the
 * current HIP device (one non-blocking HIP stream + scratch), exactly what pylibcugraph passes for SG
 * (python/pylibcugraph/pylibcugraph/resource_handle.pyx).  A non-NULL argument is a raft::handle_t*
 * in the reference; this library cannot consume one and returns NULL for it.  One handle per OS
 * thread / GPU; calls on one handle are serial and blocking at return for host-visible results. */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/export.h>
#include <cugraph_c/types.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct cugraph_resource_handle_ { int32_t align_; } cugraph_resource_handle_t;
CUGRAPH_EXPORT cugraph_resource_handle_t* cugraph_create_resource_handle(void* raft_handle);
CUGRAPH_EXPORT int cugraph_resource_handle_get_comm_size(const cugraph_resource_handle_t* handle);
CUGRAPH_EXPORT int cugraph_resource_handle_get_rank(const cugraph_resource_handle_t* handle);
CUGRAPH_EXPORT void cugraph_free_resource_handle(cugraph_resource_handle_t* handle);
#ifdef __cplusplus
}
#endif
