/* TEST INFRASTRUCTURE.  The reference's cpp/tests/c_api/mg_test_utils.h:9 includes <mpi.h> for macros (C_MPI_TRY ...) that the plain-C
 * multi-GPU tests themselves never expand: its ranks are started by mpirun, this library's by tests/c_api/ref_mg_test_shim.c.  The few
 * names those macro bodies mention are all the tests need to compile on a box without MPI. */
#pragma once
#define MPI_SUCCESS 0
#define MPI_MAX_ERROR_STRING 256
int MPI_Error_string(int errorcode, char* string, int* resultlen);
