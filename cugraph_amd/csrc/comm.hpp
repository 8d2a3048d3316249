// One-node communicator of the library: one process per GPU, device memory of every rank mapped into every rank (HIP IPC), data moved
// by direct peer WRITES over xGMI, completion signalled by sequence numbers in peer-mapped flag words.
//
// What it replaces: raft::comms (NCCL) inside the reference's raft::handle_t (cpp/src/c_api/resource_handle.cpp:11-39,
// cpp/include/cugraph/partition_manager.hpp:42-51,165-178) for the collectives of SURVEY.md section 8e -- the x exchange of a PageRank
// iteration (prims/update_edge_src_dst_property.cuh:550-579), the frontier exchange of a BFS / SSSP level
// (prims/transform_reduce_if_v_frontier_outgoing_e_by_dst.cuh:981-1074), scalar all-reduces (host_scalar_allreduce) and the edge
// shuffle of cugraph_graph_create_mg (c_api/graph_mg.cpp:140).
//
// Why not ring collectives: xGMI on an MI355X node is a full mesh of point-to-point links (7 per GPU); a ring is bound by ONE link,
// a push where every rank writes each peer's share straight into that peer's receive window uses all seven at once, needs no
// staging copy on either side (the producing kernel's stores ARE the transfer) and no collective launch.  Ordering is by
// monotone sequence numbers: a rank's k-th signal on a channel stores k into flags[channel][rank] of every peer (system-scope
// release after a system-scope fence), a wait spins (one wavefront, s_sleep, bounded by a wall-clock timeout) until all peers'
// words are >= k.  Everything is enqueued on the handle's stream; steady-state iterations never synchronise with the host.
// The same code runs with several ranks SHARING one GPU (tests on a one-GPU box): IPC handles map the same physical memory.
//
// Bootstrap: a POSIX shared-memory segment named by the session (one node), holding a sense-reversing host barrier and one
// 4 KiB slot per rank (IPC handles, counts).  Host-side collectives (host_allgather) are used at construction time only.
#pragma once

#include "common.hpp"

#include <atomic>

namespace cga {

constexpr int kCommMaxRanks  = 64;
constexpr int kCommChannels  = 64;       // independent signal sequences (a plan takes one or two)
constexpr uint32_t kCommMagic = 0x43474331u;  // "CGC1"
constexpr size_t kCommSlotBytes = 4096;

struct comm_shm_t {  // lives in the POSIX shm segment
  std::atomic<uint32_t> ready;
  uint32_t size;
  std::atomic<uint32_t> bar_count;
  std::atomic<uint32_t> bar_gen;
  std::atomic<uint32_t> abort_flag;
  std::atomic<uint32_t> attached;
  uint32_t pid0;  // process id of the rank 0 that created the segment: a segment whose creator is gone is a crashed job's (comm.hip: attach_session)
  uint32_t pad0_;
  uint64_t pidns0;  // inode of the creator's PID namespace (/proc/self/ns/pid; 0: unknown): pid0 means something only to a rank in the SAME namespace
                    // (one container per GPU sharing /dev/shm: the creator's pid is invisible there and must not be read as "gone")
  uint32_t pad_[6];
  unsigned char slots[kCommMaxRanks][kCommSlotBytes];
};

// A symmetric allocation: one device block per rank at the same offset of a symmetric ARENA, every block mapped into every process.
struct comm_window_t {
  std::vector<void*> peer;    // peer[r]: address of rank r's block in THIS process (peer[rank] = local block)
  std::vector<size_t> bytes;  // bytes[r] (the same on every rank: the largest request)
  void* local{nullptr};
  int arena{-1};
  size_t offset{0};
  template <typename T> T* at(int r) const { return static_cast<T*>(peer[r]); }
};

// Device memory is exported / imported through HIP IPC ONCE per arena and never handed back while the communicator lives: on this
// driver (ROCm 7.2, dmabuf IPC) exporting a fresh allocation that reuses the virtual address of a freed, previously exported one fails
// with "invalid argument" or -- for uncached memory -- silently yields the handle of the FREED memory (round 4: the second large window
// of a process received its peers' pushes nowhere; tools/gpu_r4c.sh).  Windows are carved out of the arenas stack-wise; every rank
// allocates the same (largest) size at the same offset, so all ranks take the same decisions without talking.
struct comm_arena_t {
  void* local{nullptr};
  size_t bytes{0}, top{0};
  std::vector<void*> peer;
  std::vector<std::pair<size_t, size_t>> live;  // (offset, size) of the windows carved out, in order; size 0 = freed, waiting for the ones above it
};

// one launch that copies words[k] 4-byte words from src[k] to dst[k] (usually a peer's window) for k < n
struct comm_push_desc_t {
  void* dst[kCommMaxRanks];
  void const* src[kCommMaxRanks];
  int64_t words[kCommMaxRanks];
  int n;
};

struct comm_t {
  uint32_t magic{kCommMagic};  // first member: cugraph_create_resource_handle recognises a communicator by it
  int rank{0}, size{1}, device{0};
  bool has_device{false};
  bool multi_device{false};    // some peer sits on another GPU (windows are then fine-grained allocations)
  std::string session;
  comm_shm_t* shm{nullptr};
  int shm_fd{-1};
  double timeout_s{60.0};
  // device-side signalling
  std::vector<comm_arena_t> arenas;
  comm_window_t* flags{nullptr};        // per rank: uint64 flags[kCommChannels][kCommMaxRanks]
  uint64_t** d_peer_flags{nullptr};     // device array [size] of the peers' flag blocks
  uint32_t* err_word{nullptr};          // host-mapped: bit 0 = a wait timed out
  uint64_t seq[kCommChannels]{};        // next sequence number per channel (all ranks advance in lockstep)
  int next_channel{2};                  // 0 = barrier, 1 = build-time collectives; plans allocate from 2
  std::vector<int> free_channels;       // returned by channel_free: seq[channel] stays monotone, so a channel can be handed out again
  uint64_t wall_ticks_per_s{100000000ull};

  // ---- host side (construction time)
  void host_barrier();
  void host_allgather(void const* in, size_t bytes, void* out);  // bytes <= kCommSlotBytes per rank
  comm_window_t* window_create(size_t bytes);                    // collective
  void window_free(comm_window_t* w);                            // collective
  // (collective like everything here: every rank allocates and returns channels in the same order, so all ranks draw the same numbers)
  int channel_alloc()
  {
    if (!free_channels.empty()) { int const c = free_channels.back(); free_channels.pop_back(); return c; }
    CGA_EXPECTS(next_channel < kCommChannels, CUGRAPH_UNKNOWN_ERROR, "communicator: out of signal channels");
    return next_channel++;
  }
  void channel_free(int channel) { if (channel >= 2) free_channels.push_back(channel); }
  // ---- device side (stream-ordered)
  uint64_t signal(hipStream_t s, int channel);           // returns the sequence number it stored
  void wait(hipStream_t s, int channel, uint64_t seq);   // until every rank's word on `channel` is >= seq
  void device_barrier(hipStream_t s) { wait(s, 0, signal(s, 0)); }
  void push_multi(hipStream_t s, comm_push_desc_t const& d);
  void check(char const* where) const;                   // throws when a wait timed out (call after a stream synchronisation)
  // ---- build-time data collectives on device buffers (each ends with a stream synchronisation)
  // every rank sends send[off[r] .. off[r] + counts[r]) (elements of elem bytes) to rank r; returns what it received, grouped by sender
  void all_to_all_v(handle_t const& h, void const* send, std::vector<int64_t> const& send_counts, size_t elem, dev_buf& recv, std::vector<int64_t>& recv_counts);
  void all_reduce_sum_u32(handle_t const& h, uint32_t* data, int64_t n);
  void all_reduce_sum_f64(handle_t const& h, double* data, int64_t n);  // folded in rank order: the same bits on every rank
  void all_gather(handle_t const& h, void const* in, size_t bytes, void* out);  // out[r * bytes ..) <- rank r's block (device pointers)
  ~comm_t();
};

comm_t* comm_create(char const* session, int rank, int size, double timeout_s);

inline comm_t* handle_comm(handle_t const& h) { return static_cast<comm_t*>(h.comm); }

}  // namespace cga
