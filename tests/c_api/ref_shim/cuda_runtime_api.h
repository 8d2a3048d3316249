/* TEST INFRASTRUCTURE.  The reference's cpp/tests/c_api/c_test_utils.h:11 includes <cuda_runtime_api.h> although the plain-C
 * tests on the PageRank / BFS / SSSP path call no CUDA runtime function: an empty stand-in is all they need to compile against
 * this library on a ROCm box. */
#pragma once
