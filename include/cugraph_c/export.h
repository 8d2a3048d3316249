/* Symbol visibility for the drop-in libcugraph_c.so (replaces cpp/include/cugraph_c/export.h). */
#pragma once
#if defined(__GNUC__)
#define CUGRAPH_EXPORT __attribute__((visibility("default")))
#else
#define CUGRAPH_EXPORT
#endif
