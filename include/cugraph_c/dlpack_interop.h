/* DLPack dtype -> cugraph_data_type_id_t.  Replaces cpp/include/cugraph_c/dlpack_interop.h:16-43
 * (impl cpp/src/c_api/dlpack_interop.cpp:25-95; called by python/pylibcugraph/pylibcugraph/utils.pyx:127-133).
 * DLPack itself (dlpack/dlpack.h, dmlc/dlpack, pulled by cpp/cmake/thirdparty/get_dlpack.cmake) is a third-party header that
 * this tree does not vendor: it is included when the build provides it; otherwise the two ABI-stable declarations this
 * entry point needs (DLDataTypeCode, DLDataType -- DLPack ABI, unchanged since v0.2) are declared here. */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/export.h>
#include <cugraph_c/types.h>
#if defined(__has_include)
#if __has_include(<dlpack/dlpack.h>)
#include <dlpack/dlpack.h>
#endif
#endif
#ifndef DLPACK_DLPACK_H_ /* include guard of dlpack/dlpack.h */
#ifndef CUGRAPH_C_DLPACK_SUBSET_H_
#define CUGRAPH_C_DLPACK_SUBSET_H_
typedef enum {
  kDLInt = 0U, kDLUInt = 1U, kDLFloat = 2U, kDLOpaqueHandle = 3U, kDLBfloat = 4U, kDLComplex = 5U, kDLBool = 6U
} DLDataTypeCode;
typedef struct {
  uint8_t code;   /* DLDataTypeCode */
  uint8_t bits;
  uint16_t lanes;
} DLDataType;
#endif
#endif
#ifdef __cplusplus
extern "C" {
#endif
/* lanes != 1 or a code other than int / uint / float / bool: CUGRAPH_UNSUPPORTED_TYPE_COMBINATION; an unsupported bit width:
 * CUGRAPH_INVALID_INPUT; NULL arguments: CUGRAPH_INVALID_INPUT */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_data_type_id_from_dlpack(const DLDataType* dlpack_dtype, cugraph_data_type_id_t* dtype,
                                                                     cugraph_error_t** error);
#ifdef __cplusplus
}
#endif
