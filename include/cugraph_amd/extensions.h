/* Entry points of libcugraph_c.so that have NO counterpart in the reference's C API.  They exist for
 * the benchmark / test harness only (on-device RMAT input, per-iteration stepping so that bench.py can
 * time exactly K power iterations, HIP-event timing of the dominant kernel).  Everything an existing
 * cuGraph caller needs is in include/cugraph_c/.
 */
#pragma once
#include <cugraph_c/array.h>
#include <cugraph_c/centrality_algorithms.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/graph_generators.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Library / build identification ("cugraph_amd <ver> gfx950"). */
CUGRAPH_EXPORT const char* cugraph_amd_version(void);

/* RMAT edge list on the device.  Algorithm of cpp/src/generators/generate_rmat_edgelist.cuh:85-103
 * (no clip-and-flip, no scramble) with a counter-based splitmix64 RNG, so edge i is a pure function of
 * (seed, i): any rank can generate any slice [first_edge, first_edge + n).  Bit-identical to
 * oracle/oracle.c:orc_rmat.  src/dst are INT32 views of size >= n. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_generate_rmat_edgelist(
  const cugraph_resource_handle_t* handle, size_t scale, size_t first_edge, size_t num_edges, double a,
  double b, double c, uint64_t seed, cugraph_type_erased_device_array_view_t* src,
  cugraph_type_erased_device_array_view_t* dst, cugraph_error_t** error);

/* PageRank as an explicit plan: create (out-weight sums, initial vector), step n power iterations,
 * finish (result object as cugraph_pagerank returns it).  cugraph_pagerank* are implemented on top of
 * exactly these calls; stepping never synchronises with the host unless epsilon > 0. */
typedef struct { int32_t align_; } cugraph_amd_pagerank_plan_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_plan_create(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values,
  const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha,
  cugraph_amd_pagerank_plan_t** plan, cugraph_error_t** error);
/* Runs up to max_iterations more iterations; stops early when the L1 change drops below epsilon
 * (epsilon <= 0: never, and no host synchronisation at all).  *iterations_done = iterations run by this
 * call, *converged = stopped on epsilon. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_plan_step(
  cugraph_amd_pagerank_plan_t* plan, double epsilon, size_t max_iterations, size_t* iterations_done,
  bool_t* converged, cugraph_error_t** error);
/* Optional, before the first step: the plan times itself (about ten iterations each) on `placements` fresh allocations of its large streamed arrays
 * -- same contents, other physical pages -- and keeps the fastest; *ms_per_iteration = the kept one's time (0: nothing to tune for this plan / graph).
 * On MI355X the same plan runs in a +-2.5 % band depending on where its arrays lie; a solver that steps the plan hundreds
 * of times gets ~2 % for ~0.15 s at RMAT-26, a one-shot cugraph_pagerank would lose -- hence an explicit call (CUGRAPH_AMD_PR_PLACEMENT_TRIALS=n makes
 * every plan creation do it).  The result vector is unaffected bit for bit. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_plan_tune(cugraph_amd_pagerank_plan_t* plan, size_t placements,
                                                                  double* ms_per_iteration, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_plan_result(
  cugraph_amd_pagerank_plan_t* plan, size_t total_iterations, bool_t converged,
  cugraph_centrality_result_t** result, cugraph_error_t** error);
CUGRAPH_EXPORT void cugraph_amd_pagerank_plan_free(cugraph_amd_pagerank_plan_t* plan);


/* Multi-GPU PageRank, one process per GPU (1-D partition by destination; design in DESIGN.md section 5).
 * `graph` is this rank's LOCAL CSC: rows [0, n_local_rows) are the destinations this rank owns, numbered by descending
 * in-degree; column ids are COMPACT: 0 .. ncols-1 = the distinct sources its edges reference, hottest first (the graph
 * must be created with max(ncols, n_local_rows) vertices).  Per iteration the host layer runs ONE sparse all-to-all
 * (torch.distributed.all_to_all_single = RCCL over xGMI), then reduce_scalars(), then local_step().
 *   send_index[k]   INT32, k < sum(send_counts): local row whose x = pr / out_w is the k-th value sent (grouped by peer)
 *   send_counts[s], recv_counts[s]  number of values sent to / received from rank s (host arrays; even for fp32)
 *   col_pos[c]      INT32, c < ncols: ELEMENT offset in `recv` of the value of column c
 *   send / recv     device views (weight type): for every peer s, in rank order, its values followed by a 32-byte tail
 *                   holding the sender's (L1 change, dangling mass, max |x|) as three doubles + 8 bytes of padding.
 * cugraph_amd/mg.py is the host layer. */
typedef struct { int32_t align_; } cugraph_amd_pagerank_mg_plan_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg_plan_create(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, size_t n_local_rows, size_t global_num_vertices, int comm_rank,
  int comm_size, const cugraph_type_erased_device_array_view_t* out_weight_sums_local,
  const cugraph_type_erased_device_array_view_t* initial_local, const cugraph_type_erased_device_array_view_t* send_index,
  const size_t* send_counts, const size_t* recv_counts, const cugraph_type_erased_device_array_view_t* col_pos,
  cugraph_type_erased_device_array_view_t* send, cugraph_type_erased_device_array_view_t* recv, double alpha,
  cugraph_amd_pagerank_mg_plan_t** plan, cugraph_error_t** error);
/* messages <- x of the initial vector + (0, partial dangling mass, max |x|) (iteration-0 state) */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg_plan_start(cugraph_amd_pagerank_mg_plan_t* plan, cugraph_error_t** error);
/* folds the comm_size (L1 change, dangling, max |x|) tails found in recv, in rank order; read_back = TRUE also
 * returns them to the host (synchronises) -- needed only when epsilon > 0 */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg_plan_reduce_scalars(cugraph_amd_pagerank_mg_plan_t* plan, bool_t read_back,
                                                                                double* diff, double* dangling, cugraph_error_t** error);
/* one power iteration on the local rows: unpack recv, tiled SpMV, new pr, next messages packed into send; blocks until done */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg_plan_local_step(cugraph_amd_pagerank_mg_plan_t* plan, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg_plan_values(cugraph_amd_pagerank_mg_plan_t* plan,
                                                                        cugraph_type_erased_device_array_view_t* out_local,
                                                                        cugraph_error_t** error);
CUGRAPH_EXPORT void cugraph_amd_pagerank_mg_plan_free(cugraph_amd_pagerank_mg_plan_t* plan);

/* Device primitives of the library behind the C ABI: the multi-GPU host layer (cugraph_amd/mg.py) builds its partition and its
 * exchange plan from these two instead of torch.sort / torch.unique / torch.argsort / torch.cumsum (rocPRIM).
 * sort_pairs: stable LSD radix sort of n (key, value) pairs in place on the bits [bit_lo, bit_hi) of the keys (vals may be NULL);
 * exclusive_scan: out[i] = sum of in[0 .. i) (in == out allowed).  Device pointers, n < 2^32; both block until done. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_sort_pairs_u64_u32(const cugraph_resource_handle_t* handle, uint64_t* keys, uint32_t* vals, size_t n,
                                                                   int bit_lo, int bit_hi, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_exclusive_scan_u32(const cugraph_resource_handle_t* handle, const uint32_t* in, uint32_t* out, size_t n,
                                                                   cugraph_error_t** error);

/* 2-D layout of the same iteration (the reference's scheme: cpp/include/cugraph/graph_view.hpp:159-216, partition_manager.hpp:42-51,
 * prims/update_edge_src_dst_property.cuh:550-579, prims/detail/per_v_transform_reduce_e.cuh:3390-3406): P = R x C ranks, rank = c * R + r;
 * vertex partitions of rows_per_partition rows (position p of the global degree order -> partition p % P, row p / P); rank (r, c) owns
 * partition c * R + r and stores the edges with source in partitions [c * R, (c + 1) * R) -- local column (q_src % R) * L + row -- and
 * destination in partitions {i * R + r} -- local row (q_dst / R) * L + row.  `graph` = that block (CSC, renumber = FALSE,
 * max(block_rows, block_cols) vertices).  Caller-owned device buffers (weight type): x_own [L] (out: this rank's x = pr / out_w,
 * the input of the column group's all-gather), x_cols [block_cols = R * L] (in: the gathered x), y_part [block_rows = C * L]
 * (out: partial row sums, the input of the row group's reduce-scatter), y_own [L] (in: the reduced owned rows), triple [4 doubles]
 * (out: this rank's L1 change, dangling mass, max |x|).  owned_rows <= L = the rows of this rank's partition that are vertices (the last
 * partitions are padded when V is not a multiple of P): padded rows hold 0 and stay out of the dangling mass and the L1 change.  One iteration = all-gather(x_own -> x_cols), spmv, reduce-scatter(y_part ->
 * y_own), epilogue, all-gather(triple) + set_scalars.  cugraph_amd/mg.py: MGPageRank2D is the host layer. */
typedef struct { int32_t align_; } cugraph_amd_pagerank_mg2d_plan_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_create(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, size_t rows_per_partition, size_t owned_rows, size_t block_rows, size_t block_cols,
  size_t global_num_vertices, const cugraph_type_erased_device_array_view_t* out_weight_sums_own, const cugraph_type_erased_device_array_view_t* initial_own,
  cugraph_type_erased_device_array_view_t* x_own, const cugraph_type_erased_device_array_view_t* x_cols, cugraph_type_erased_device_array_view_t* y_part,
  const cugraph_type_erased_device_array_view_t* y_own, cugraph_type_erased_device_array_view_t* triple, double alpha, cugraph_amd_pagerank_mg2d_plan_t** plan,
  cugraph_error_t** error);
/* x_own <- x of the initial vector, triple <- (0, partial dangling mass, max |x|) */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_start(cugraph_amd_pagerank_mg2d_plan_t* plan, cugraph_error_t** error);
/* folds comm_size gathered triples (4 doubles each, device) in rank order into the iteration's constants; read_back as for the 1-D plan */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_set_scalars(cugraph_amd_pagerank_mg2d_plan_t* plan, const void* gathered_triples,
                                                                               int comm_size, bool_t read_back, double* diff, double* dangling,
                                                                               cugraph_error_t** error);
/* y_part <- (local block) x (alpha * x_cols) */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_spmv(cugraph_amd_pagerank_mg2d_plan_t* plan, cugraph_error_t** error);
/* pr <- base + y_own, x_own <- pr / out_w, triple <- this rank's scalars */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_epilogue(cugraph_amd_pagerank_mg2d_plan_t* plan, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_values(cugraph_amd_pagerank_mg2d_plan_t* plan,
                                                                          cugraph_type_erased_device_array_view_t* out_own, cugraph_error_t** error);
CUGRAPH_EXPORT void cugraph_amd_pagerank_mg2d_plan_free(cugraph_amd_pagerank_mg2d_plan_t* plan);

/* Partitioned BFS / SSSP: the per-rank engine (cugraph_amd/csrc/traversal_mg.hip; host layer cugraph_amd/mg_traversal.py).
 * Replaces the multi_gpu = true halves of cpp/src/traversal/bfs_impl.cuh:133-870, sssp_impl.cuh:169-566 and of
 * prims/transform_reduce_if_v_frontier_outgoing_e_by_dst.cuh:617-1127 (the (dst, payload) shuffle to the dst owner is the
 * all-to-all the host layer runs between expand and apply).
 *   Vertices: position p in the global degree order -> rank p % comm_size, local row p / comm_size; a vertex is named by its
 *   compact global id g = owner * rows_per_rank + row (rows_per_rank a multiple of 64, the same on every rank).
 *   offsets / indices / weights   CSR of the out-edges of the local rows (device, borrowed), destinations as compact global ids
 *   row_vertex                    external id of every local row (device, borrowed)
 *   mode                          0 = BFS (tuples of 2 int32: row at the owner, parent external id)
 *                                 1 = SSSP, float weights (3 int32: row, distance bits, parent external id + 1)
 *   send                          device buffer for capacity_tuples tuples, >= min(comm_size * rows_per_rank, n_edges)
 * One level: expand (send <- candidates grouped by owner, send_counts[r] tuples for rank r) -> all-to-all -> apply (n_next =
 * size of the next local frontier; the search ends when it is 0 on every rank).  BFS additionally all-gathers
 * frontier_bits after reset and after every apply and hands the result to merge_visited. */
typedef struct { int32_t align_; } cugraph_amd_traversal_mg_plan_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_create(
  const cugraph_resource_handle_t* handle, const int32_t* offsets, const int32_t* indices, const float* weights, size_t n_rows,
  size_t n_edges, size_t rows_per_rank, int comm_rank, int comm_size, const int32_t* row_vertex, int mode, int32_t* send,
  size_t capacity_tuples, cugraph_amd_traversal_mg_plan_t** plan, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_reset(cugraph_amd_traversal_mg_plan_t* plan, const int32_t* source_rows,
                                                                        size_t n_sources, double cutoff, bool_t compute_predecessors,
                                                                        cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_expand(cugraph_amd_traversal_mg_plan_t* plan, size_t* send_counts,
                                                                         cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_apply(cugraph_amd_traversal_mg_plan_t* plan, const int32_t* recv,
                                                                        size_t n_tuples, uint32_t level, size_t* n_next,
                                                                        cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_frontier_bits(cugraph_amd_traversal_mg_plan_t* plan, const uint32_t** bits,
                                                                                cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_merge_visited(cugraph_amd_traversal_mg_plan_t* plan, const uint32_t* gathered,
                                                                                cugraph_error_t** error);
/* SSSP near / far windows on a partitioned plan (sssp_impl.cuh:376-561): rows whose distance drops to a value at or beyond `hi` wait in a
 * per-rank far pile instead of joining the next frontier (reset sets hi = +inf: no window).  All ranks together: rounds until the frontier is
 * empty everywhere, far_stats on every rank (entries still beyond the window, their smallest distance), the next bound from the global
 * minimum, advance (splits the pile: the rows below the new bound become the frontier).  The result does not depend on the windows. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_sssp_set_window(cugraph_amd_traversal_mg_plan_t* plan, double hi, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_sssp_far_stats(cugraph_amd_traversal_mg_plan_t* plan, size_t* n_far, double* min_distance,
                                                                                 cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_sssp_advance(cugraph_amd_traversal_mg_plan_t* plan, double hi_new, size_t* n_frontier,
                                                                               cugraph_error_t** error);
/* BFS, direction-optimising (replaces the bottom-up branch of bfs_impl.cuh:587-805 for the partitioned case): set_bottom_up hands over the
 * in-edges of the local rows -- in_offsets [n_rows + 1], in_indices = in-neighbours as compact global ids in ascending order of their
 * EXTERNAL id (the first frontier member of a row is then the minimum-external-id parent, the rule of the top-down levels), over-allocated
 * by at least 8 entries -- and ext_of_g [comm_size * rows_per_rank] = external id of a compact global id (device, borrowed).  A bottom_up
 * level scans the unvisited local rows against `front`, the all-gathered frontier bitmap of the previous level (comm_size * rows_per_rank
 * bits): no candidate exchange, the discoveries are local; frontier_bits / merge_visited follow as after apply.  last_degree_sums: out- and
 * in-degree sums of the vertices the last apply / bottom_up discovered on this rank (the two quantities of the direction heuristic). */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_set_bottom_up(cugraph_amd_traversal_mg_plan_t* plan, const int32_t* in_offsets,
                                                                                const int32_t* in_indices, const int32_t* ext_of_g,
                                                                                cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_bottom_up(cugraph_amd_traversal_mg_plan_t* plan, const uint32_t* front,
                                                                            uint32_t level, size_t* n_found, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_last_degree_sums(cugraph_amd_traversal_mg_plan_t* plan,
                                                                                   unsigned long long* out_edges, unsigned long long* in_edges,
                                                                                   cugraph_error_t** error);
/* distances of the local rows (BFS: int32, INT32_MAX unreached; SSSP: float, FLT_MAX unreached) and predecessors (external ids, -1) */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_traversal_mg_plan_results(cugraph_amd_traversal_mg_plan_t* plan, void* distances,
                                                                          int32_t* predecessors, cugraph_error_t** error);
/* the buffers handed to merge_visited outlive the call (persistent exchange windows): it then returns without synchronising */
CUGRAPH_EXPORT void cugraph_amd_traversal_mg_plan_keep_buffers(cugraph_amd_traversal_mg_plan_t* plan, bool_t on);
/* Moves a plan to another resource handle of the same device (later calls queue on that handle's stream and read back through its pinned page).  The
 * library's own multi-GPU BFS / SSSP do this when a graph's cached plan is used through a different handle than the one that built it. */
CUGRAPH_EXPORT void cugraph_amd_traversal_mg_plan_rebind(cugraph_amd_traversal_mg_plan_t* plan, const cugraph_resource_handle_t* handle);
CUGRAPH_EXPORT void cugraph_amd_traversal_mg_plan_free(cugraph_amd_traversal_mg_plan_t* plan);

/* One-node communicator of the library (cugraph_amd/csrc/comm.hpp): one process per GPU, every rank's device windows mapped into every
 * rank through HIP IPC, data moved by direct peer writes over xGMI, completion by sequence-number flags -- no collective launches on
 * the per-iteration paths.  It plays the part of the raft::handle_t with NCCL comms that the reference's
 * cugraph_create_resource_handle(void* raft_handle) takes (cpp/src/c_api/resource_handle.cpp:11-39): pass the communicator as that
 * pointer and the handle reports its rank / size; cugraph_graph_create_mg, cugraph_pagerank, cugraph_bfs, cugraph_sssp and
 * cugraph_louvain on such a handle are COLLECTIVE (every rank calls them in the same order with its slice, as in the reference).
 *   session   names the job on this node (a POSIX shared-memory segment carries the bootstrap): unique per job, the same on all ranks
 *   rank/size one process per rank; several ranks may share one GPU (tests), at most 64 ranks
 * The calling thread's current HIP device is the rank's GPU.  Free the communicator after the handles / graphs that use it. */
typedef struct { int32_t align_; } cugraph_amd_comm_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_comm_create(const char* session, int rank, int size, cugraph_amd_comm_t** comm,
                                                            cugraph_error_t** error);
CUGRAPH_EXPORT void cugraph_amd_comm_free(cugraph_amd_comm_t* comm);
/* the host half of the bootstrap alone (shared-memory session, barrier, all-gather of small host payloads); needs no GPU */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_comm_host_selftest(const char* session, int rank, int size, int rounds, cugraph_error_t** error);
/* host-side collectives of the bootstrap segment for the harness (timing barriers, max-over-ranks): a barrier, and an all-gather of at
 * most 4096 bytes per rank (out holds size * bytes) */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_comm_host_barrier(cugraph_amd_comm_t* comm, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_comm_host_allgather(cugraph_amd_comm_t* comm, const void* in, size_t bytes, void* out,
                                                                    cugraph_error_t** error);
CUGRAPH_EXPORT int cugraph_amd_comm_rank(const cugraph_amd_comm_t* comm);
CUGRAPH_EXPORT int cugraph_amd_comm_size(const cugraph_amd_comm_t* comm);
/* Collective self-check of every primitive (all-gather, all-to-all-v, integer / double all-reduce against closed forms) and timing of
 * the two on the per-iteration path: out[0] = microseconds per device barrier, out[1] = GB/s of peer pushes issued by this rank
 * (n_words 4-byte words to every peer per iteration), out[2] = 1 when the ranks sit on different GPUs.  `handle` must have been
 * created on a communicator. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_comm_selftest(const cugraph_resource_handle_t* handle, size_t n_words, int iterations,
                                                              double* out, cugraph_error_t** error);

/* MatrixMarket coordinate file -> device edge list (the format of the reference's datasets/karate.mtx etc.; conventions of
 * cpp/tests/utilities/matrix_market_file_utilities.cu: 0-based ids, `pattern` -> weight 1, a `symmetric` file's off-diagonal
 * entries are mirrored, vertex count = matrix dimension).  The result is a cugraph_coo_t (INT32 ids, FLOAT32 weights). */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_read_matrix_market(const cugraph_resource_handle_t* handle, const char* path,
                                                                   cugraph_coo_t** result, size_t* num_vertices, bool_t* is_symmetric,
                                                                   bool_t* has_weights, cugraph_error_t** error);

/* Makes the library enqueue its work on `hip_stream` (a hipStream_t; NULL = back to the handle's own stream).  With the
 * stream the caller's collectives run on (torch's current stream for RCCL), stream order replaces host synchronisation:
 * cugraph_amd_pagerank_mg_plan_local_step then returns without waiting. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_handle_set_stream(const cugraph_resource_handle_t* handle, void* hip_stream,
                                                                  cugraph_error_t** error);

/* Blocks until everything queued on the handle's stream has finished. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_handle_sync(const cugraph_resource_handle_t* handle,
                                                            cugraph_error_t** error);

/* HIP-event timing of the dominant kernels on the handle's own stream.  When enabled, every launch of
 * the named kernel family is bracketed by hipEventRecord on that stream; get() synchronises and returns
 * launch count and summed milliseconds since the last reset.  family: "pagerank_spmv", "bfs_expand",
 * "sssp_relax". */
CUGRAPH_EXPORT void cugraph_amd_kernel_timing_enable(const cugraph_resource_handle_t* handle, bool_t on);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_kernel_timing_get(
  const cugraph_resource_handle_t* handle, const char* family, size_t* launches, double* total_ms,
  cugraph_error_t** error);
CUGRAPH_EXPORT void cugraph_amd_kernel_timing_reset(const cugraph_resource_handle_t* handle);
/* One event pair around a whole region of work on the handle's stream, independent of kernel_timing_enable: begin() records the start event,
 * end() the stop event; cugraph_amd_kernel_timing_get(family) then reports them like one launch.  bench.py brackets its timed iterations this
 * way, with the per-launch events off: nothing but the kernels sits between the two events or inside the wall clock. */
CUGRAPH_EXPORT void cugraph_amd_kernel_timing_region_begin(const cugraph_resource_handle_t* handle, const char* family);
CUGRAPH_EXPORT void cugraph_amd_kernel_timing_region_end(const cugraph_resource_handle_t* handle, const char* family);

/* Graph introspection used by the tests (sizes, degree-class boundaries of the CSC/CSR row schedule). */
CUGRAPH_EXPORT size_t cugraph_amd_graph_num_vertices(const cugraph_graph_t* graph);
CUGRAPH_EXPORT size_t cugraph_amd_graph_num_edges(const cugraph_graph_t* graph);
/* multi-GPU graph: edges of this rank's PageRank partition (0 before the first PageRank call has built it); otherwise all edges */
CUGRAPH_EXPORT size_t cugraph_amd_graph_num_local_edges(const cugraph_graph_t* graph);

/* Hypersparse rows (DCSR; DCSC for the transposed orientation): the CSR + DCSR hybrid the reference builds for edge partitions whose rows are mostly
 * empty (compress_hypersparse_offsets, cpp/src/structure/detail/structure_utils.cuh:139-195; read through dcs_nzd_vertices / major_hypersparse_first,
 * cpp/include/cugraph/edge_partition_device_view.cuh:43-58, 820-835).  Rows [0, first_row) keep one offset each; a row >= first_row is stored only
 * when it has an edge.  The library puts the local block of its 2-D multi-GPU PageRank layout into this form by itself; this call does it to an
 * orientation of a single-GPU graph (internal row ids; transposed = TRUE: the CSC orientation, built when missing) -- for graphs created with
 * renumber = FALSE over a sparse id range, and for the tests.  PageRank's re-blocking, the degree calls and cugraph_decompress_to_edgelist walk the
 * hybrid form directly; any other algorithm re-inflates the orientation to plain offsets on its first use. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_graph_compress_hypersparse(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                                           bool_t transposed, size_t first_row, cugraph_error_t** error);
/* The row storage of an orientation as it is now: *first_row / *num_nzd (0 / 0 for a plain orientation), and borrowed device views (valid until the
 * graph changes form or is freed; free them with cugraph_type_erased_device_array_view_free) of the INT32 nzd rows [num_nzd] and of the INT32 offsets
 * [first_row + num_nzd + 1, or V + 1 for a plain orientation].  *is_hypersparse = FALSE for a plain orientation. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_amd_graph_hypersparse_view(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                                       bool_t transposed, bool_t* is_hypersparse, size_t* first_row,
                                                                       size_t* num_nzd, cugraph_type_erased_device_array_view_t** nzd_rows,
                                                                       cugraph_type_erased_device_array_view_t** offsets, cugraph_error_t** error);

/* Tuning knob for the PageRank SpMV: number of x entries staged in LDS per workgroup (0 = off).
 * Default is chosen from the graph size; set before plan creation.  Returns the previous value. */
CUGRAPH_EXPORT int cugraph_amd_set_pagerank_hot_tile(const cugraph_resource_handle_t* handle, int n_entries);

/* Traversal statistics of the last cugraph_bfs / cugraph_sssp on this handle: number of levels or bucket
 * steps, edges inspected (relaxations), vertices reached.  cugraph_louvain reports its work in the same struct: steps = sweeps,
 * edges_inspected / vertices_reached = sum over the sweeps of the level's edges / vertices, edges_of_reached = sum over the
 * levels of the level's edges (one contraction each). */
typedef struct {
  uint64_t steps;
  uint64_t edges_inspected;
  uint64_t vertices_reached;
  uint64_t edges_of_reached;
  uint64_t probes;  /* cugraph_sssp: relaxations that went on to probe the destination's tentative distance (the others were dropped by the
                     * L2-resident distance filter); 0 elsewhere */
} cugraph_amd_traversal_stats_t;
CUGRAPH_EXPORT void cugraph_amd_last_traversal_stats(const cugraph_resource_handle_t* handle,
                                                     cugraph_amd_traversal_stats_t* out);
/* Device memory of the library comes from a process-wide caching pool (csrc/common.hpp: dev_buf): freed blocks are kept and
 * reused (hipMalloc / hipFree are synchronous and slow at graph sizes).  _trim returns every cached block to the driver and
 * reports how many bytes that were; environment: CUGRAPH_AMD_POOL=0 (off), CUGRAPH_AMD_POOL_MAX_GB (cache cap, default 128 GB and at most 45 % of the device).
 * _trim_large returns only the cached blocks above block_bytes (the sort buffers of a finished graph / plan construction) and keeps
 * the small blocks the per-call paths recycle.  A cached block is reused across streams only behind an event recorded at its free. */
CUGRAPH_EXPORT size_t cugraph_amd_memory_pool_trim(void);
CUGRAPH_EXPORT size_t cugraph_amd_memory_pool_trim_large(size_t block_bytes);
CUGRAPH_EXPORT size_t cugraph_amd_memory_pool_cached_bytes(void);

#ifdef __cplusplus
}
#endif
