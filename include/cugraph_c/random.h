/* Random-number state object.  Replaces cpp/include/cugraph_c/random.h:16-44 (impl cpp/src/c_api/random.cpp).
 * Here a state is (seed, draws so far) of a counter-based generator: a fresh state with seed s reproduces the benchmark
 * generator of include/cugraph_amd/extensions.h and the CPU oracle (the reference's raft::random streams are not
 * reproducible outside RAFT). */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/export.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_rng_state_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_rng_state_create(const cugraph_resource_handle_t* handle, uint64_t seed,
                                                             cugraph_rng_state_t** state, cugraph_error_t** error);
CUGRAPH_EXPORT void cugraph_rng_state_free(cugraph_rng_state_t* p);
#ifdef __cplusplus
}
#endif
