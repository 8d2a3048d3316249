/* Error codes and error object.  Replaces cpp/include/cugraph_c/error.h:16-47
 * (impl cpp/src/c_api/error.cpp).  Every entry point returns a code and writes *error (heap object,
 * caller frees with cugraph_error_free, NULL-safe).  C++ exceptions never cross this boundary. */
#pragma once
#include <cugraph_c/export.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum cugraph_error_code_ {
  CUGRAPH_SUCCESS = 0,
  CUGRAPH_UNKNOWN_ERROR,
  CUGRAPH_INVALID_HANDLE,
  CUGRAPH_ALLOC_ERROR,
  CUGRAPH_INVALID_INPUT,
  CUGRAPH_NOT_IMPLEMENTED,
  CUGRAPH_UNSUPPORTED_TYPE_COMBINATION
} cugraph_error_code_t;
typedef struct cugraph_error_ { int32_t align_; } cugraph_error_t;
CUGRAPH_EXPORT const char* cugraph_error_message(const cugraph_error_t* error);
CUGRAPH_EXPORT void cugraph_error_free(cugraph_error_t* error);
#ifdef __cplusplus
}
#endif
