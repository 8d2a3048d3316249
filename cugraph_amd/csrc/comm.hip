// One-node communicator: HIP IPC windows, direct peer writes, sequence-number flags (design: comm.hpp).
// Replaces raft::comms / NCCL behind the reference's resource handle for the collectives of SURVEY.md section 8e
// (cpp/src/c_api/resource_handle.cpp:11-39, cpp/include/cugraph/partition_manager.hpp:42-51).
#include "comm.hpp"

#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstring>
#include <new>

namespace cga {
namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

std::string shm_name(std::string const& session)
{
  std::string n = "/cga_";
  for (char c : session) n.push_back((isalnum((unsigned char)c) || c == '_' || c == '-') ? c : '_');
  return n;
}

// ---------------------------------------------------------------------------------------------- device side
// k-th signal of `rank` on `channel`: every peer's flags[channel][rank] <- seq, after everything this stream wrote before.
__global__ void k_comm_signal(uint64_t* const* peer_flags, int size, int rank, int channel, uint64_t seq)
{
  int const r = threadIdx.x;
  __threadfence_system();
  if (r < size) {
    uint64_t* f = peer_flags[r] + (size_t)channel * kCommMaxRanks + rank;
    __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// one lane per peer spins until that peer's word has reached seq; bounded by the wall clock (a dead peer must not hang the GPU)
__global__ void k_comm_wait(uint64_t const* my_flags, int size, int channel, uint64_t seq, long long timeout_ticks, uint32_t* err)
{
  int const r = threadIdx.x;
  if (r < size) {
    uint64_t const* f  = my_flags + (size_t)channel * kCommMaxRanks + r;
    long long const t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > timeout_ticks) {
        __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __threadfence_system();
}

// dst[i] = src[i], 4-byte words (dst usually lives in a peer's memory: coalesced stores over xGMI)
__global__ void __launch_bounds__(256) k_push_u32(uint32_t* __restrict__ dst, uint32_t const* __restrict__ src, int64_t n)
{
  int64_t i            = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    uint32_t const a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_push_multi(comm_push_desc_t d)
{
  int const k          = blockIdx.y;
  uint32_t* dst        = static_cast<uint32_t*>(d.dst[k]);
  uint32_t const* src  = static_cast<uint32_t const*>(d.src[k]);
  int64_t const n      = d.words[k];
  int64_t i            = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    uint32_t const a = src[i], b = src[i + stride], c = src[i + 2 * stride], e = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = e;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

template <typename T>
__global__ void __launch_bounds__(256) k_fold_chunks(T const* staged /*[size][chunk]*/, int size, int64_t chunk, int64_t count, T* out)
{
  int64_t i            = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    T acc = staged[i];
    for (int s = 1; s < size; ++s) acc += staged[(int64_t)s * chunk + i];  // rank order: the same bits on every rank
    out[i] = acc;
  }
}

void push_words(hipStream_t s, void* dst, void const* src, int64_t n_words)
{
  if (n_words <= 0) return;
  hipLaunchKernelGGL(k_push_u32, grid_for((n_words + 3) / 4, 256, 4096), 256, 0, s, static_cast<uint32_t*>(dst), static_cast<uint32_t const*>(src), n_words);
}

// Windows are written by OTHER processes (peers on other GPUs over xGMI; in the one-GPU tests, kernels of another process) and then read
// by local kernels.  Peers on other GPUs: fine-grained memory (coherent with remote writes).  All ranks on ONE GPU: ordinary device memory
// (same physical L2s; kernel boundaries write back / invalidate them).  CUGRAPH_AMD_COMM_WINDOWS = cached | finegrained | uncached overrides.
void* raw_alloc(comm_t const& c, size_t bytes)
{
  void* p = nullptr;
  static int const forced = [] {
    char const* e = getenv("CUGRAPH_AMD_COMM_WINDOWS");
    std::string const k = e ? e : "";
    return k == "cached" ? 0 : k == "finegrained" ? 1 : k == "uncached" ? 2 : -1;
  }();
  int const kind = forced >= 0 ? forced : (c.multi_device ? 1 : 0);
  auto alloc = [&]() {
    return kind == 0 ? hipMalloc(&p, bytes) : hipExtMallocWithFlags(&p, bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
  };
  hipError_t e = alloc();
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    pool_release_large_blocks(0);  // the library's own cache may be sitting on the memory
    e = alloc();
  }
  HIP_TRY(e);
  return p;
}

struct win_slot_t {
  hipIpcMemHandle_t handle;
  uint64_t bytes;
  uint64_t pid;
};

}  // namespace

// ---------------------------------------------------------------------------------------------- host side
void comm_t::host_barrier()
{
  if (size == 1) return;
  uint32_t const gen = shm->bar_gen.load(std::memory_order_acquire);
  if (shm->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)size) {
    shm->bar_count.store(0, std::memory_order_relaxed);
    shm->bar_gen.fetch_add(1, std::memory_order_release);
    return;
  }
  double const t0 = now_s();
  int spins       = 0;
  while (shm->bar_gen.load(std::memory_order_acquire) == gen) {
    if (shm->abort_flag.load(std::memory_order_relaxed)) throw api_error(CUGRAPH_UNKNOWN_ERROR, "communicator: a peer aborted");
    if (++spins > 200) { sched_yield(); }
    if ((spins & 1023) == 0 && now_s() - t0 > timeout_s) {
      shm->abort_flag.store(1, std::memory_order_relaxed);
      throw api_error(CUGRAPH_UNKNOWN_ERROR, "communicator: host barrier timed out on rank " + std::to_string(rank) + " (a peer is missing or has failed)");
    }
  }
}

void comm_t::host_allgather(void const* in, size_t bytes, void* out)
{
  CGA_EXPECTS(bytes <= kCommSlotBytes, CUGRAPH_UNKNOWN_ERROR, "communicator: host_allgather payload exceeds a slot");
  if (size == 1) { std::memcpy(out, in, bytes); return; }
  std::memcpy(shm->slots[rank], in, bytes);
  host_barrier();
  for (int r = 0; r < size; ++r) std::memcpy(static_cast<char*>(out) + (size_t)r * bytes, shm->slots[r], bytes);
  host_barrier();  // nobody overwrites a slot before everybody has read it
}

namespace {
// exports `local` (a fresh allocation of `bytes`) and maps every peer's counterpart; collective
void exchange_mappings(comm_t& c, void* local, size_t bytes, std::vector<void*>& peer)
{
  peer.assign(c.size, nullptr);
  peer[c.rank] = local;
  if (c.size == 1) return;
  win_slot_t mine{};
  hipError_t e = hipIpcGetMemHandle(&mine.handle, local);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c.shm->abort_flag.store(1, std::memory_order_relaxed);  // the peers are waiting in a barrier: fail them now, not after the timeout
    throw api_error(CUGRAPH_UNKNOWN_ERROR, std::string("communicator: hipIpcGetMemHandle failed (") + hipGetErrorString(e) + ") for " + std::to_string(bytes) +
                                             " bytes at " + std::to_string((uintptr_t)local) + " on rank " + std::to_string(c.rank));
  }
  mine.bytes = bytes;
  mine.pid   = (uint64_t)getpid();
  std::vector<win_slot_t> all(c.size);
  c.host_allgather(&mine, sizeof(mine), all.data());
  for (int r = 0; r < c.size; ++r) {
    if (r == c.rank) continue;
    CGA_EXPECTS(all[r].pid != mine.pid, CUGRAPH_INVALID_INPUT, "communicator: two ranks in one process (one process per rank is required: HIP IPC maps ANOTHER process's memory)");
    void* p = nullptr;
    e       = hipIpcOpenMemHandle(&p, all[r].handle, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      c.shm->abort_flag.store(1, std::memory_order_relaxed);
      throw api_error(CUGRAPH_UNKNOWN_ERROR, std::string("communicator: hipIpcOpenMemHandle failed (") + hipGetErrorString(e) + ") for rank " + std::to_string(r) + "'s block of " +
                                               std::to_string(all[r].bytes) + " bytes on rank " + std::to_string(c.rank));
    }
    peer[r] = p;
  }
  c.host_barrier();
}
}  // namespace

comm_window_t* comm_t::window_create(size_t bytes)
{
  HIP_TRY(hipSetDevice(device));
  // every rank carves the LARGEST request out of the same place: the arenas look the same everywhere
  uint64_t mine = (std::max<size_t>(bytes, 256) + 4095) / 4096 * 4096;
  std::vector<uint64_t> all(size);
  host_allgather(&mine, sizeof(mine), all.data());
  size_t b = 0;
  for (auto x : all) b = std::max<size_t>(b, (size_t)x);
  int a = -1;
  for (size_t k = 0; k < arenas.size(); ++k)
    if (arenas[k].top + b <= arenas[k].bytes) { a = (int)k; break; }
  if (a < 0) {
    comm_arena_t ar;
    ar.bytes = std::max<size_t>(b, (size_t)256 << 20);
    ar.local = raw_alloc(*this, ar.bytes);
    exchange_mappings(*this, ar.local, ar.bytes, ar.peer);
    arenas.push_back(std::move(ar));
    a = (int)arenas.size() - 1;
  }
  comm_arena_t& ar = arenas[a];
  {  // the arenas must look the same on every rank: a rank that created or freed windows in another order (a destructor that ran at another
     // time, an error path taken on one rank only) would carve this one out of another place and its peers would push into the wrong
     // memory -- silently.  The place is agreed on here, where a host all-gather runs anyway.
    uint64_t const place[2] = {(uint64_t)a, (uint64_t)ar.top};
    std::vector<uint64_t> places((size_t)2 * size);
    host_allgather(place, sizeof(place), places.data());
    for (int r = 0; r < size; ++r)
      CGA_EXPECTS(places[2 * r] == place[0] && places[2 * r + 1] == place[1], CUGRAPH_UNKNOWN_ERROR,
                  "communicator: the ranks' window stacks differ (rank " + std::to_string(r) + " would place this window at arena " + std::to_string(places[2 * r]) + " + " +
                    std::to_string(places[2 * r + 1]) + ", rank " + std::to_string(rank) + " at arena " + std::to_string(place[0]) + " + " + std::to_string(place[1]) +
                    "): windows, plans and graphs must be created and freed in the same order on every rank");
  }
  auto w    = std::make_unique<comm_window_t>();
  w->arena  = a;
  w->offset = ar.top;
  w->peer.assign(size, nullptr);
  w->bytes.assign(size, b);
  for (int r = 0; r < size; ++r) w->peer[r] = static_cast<char*>(ar.peer[r]) + ar.top;
  w->local = w->peer[rank];
  ar.live.emplace_back(ar.top, b);
  ar.top += b;
  return w.release();
}

void comm_t::window_free(comm_window_t* w)
{
  if (!w) return;
  {  // every rank frees the SAME window (the all-gather is also the barrier: nobody is still writing into a block that is about to be handed
     // out again); a mismatch is reported, not repaired -- the arenas of the ranks no longer agree
    uint64_t const place[2] = {(uint64_t)w->arena, (uint64_t)w->offset};
    std::vector<uint64_t> places((size_t)2 * size);
    host_allgather(place, sizeof(place), places.data());
    bool same = true;
    for (int r = 0; r < size; ++r) same = same && places[2 * r] == place[0] && places[2 * r + 1] == place[1];
    if (!same) {
      if (shm) shm->abort_flag.store(1, std::memory_order_relaxed);
      fprintf(stderr, "cugraph_amd communicator: rank %d frees the window at arena %d + %zu while a peer frees another one: windows must be freed in the same order on every rank\n",
              rank, w->arena, w->offset);
    }
  }
  comm_arena_t& ar = arenas[w->arena];
  for (auto& lv : ar.live)
    if (lv.first == w->offset && lv.second != 0) { lv.second = 0; break; }
  while (!ar.live.empty() && ar.live.back().second == 0) {  // give back what is on top of the stack
    ar.top = ar.live.back().first;
    ar.live.pop_back();
  }
  delete w;
}

void comm_t::push_multi(hipStream_t s, comm_push_desc_t const& d)
{
  int64_t biggest = 0;
  for (int k = 0; k < d.n; ++k) biggest = std::max(biggest, d.words[k]);
  if (biggest <= 0 || d.n <= 0) return;
  dim3 const grid((unsigned)grid_for((biggest + 3) / 4, 256, 2048), (unsigned)d.n);
  hipLaunchKernelGGL(k_push_multi, grid, dim3(256), 0, s, d);
}

uint64_t comm_t::signal(hipStream_t s, int channel)
{
  uint64_t const k = ++seq[channel];
  hipLaunchKernelGGL(k_comm_signal, 1, 64, 0, s, (uint64_t* const*)d_peer_flags, size, rank, channel, k);
  return k;
}

void comm_t::wait(hipStream_t s, int channel, uint64_t k)
{
  long long const ticks = (long long)(timeout_s * (double)wall_ticks_per_s);
  hipLaunchKernelGGL(k_comm_wait, 1, 64, 0, s, (uint64_t const*)flags->local, size, channel, k, ticks, err_word);
}

void comm_t::check(char const* where) const
{
  if (err_word && *reinterpret_cast<volatile uint32_t*>(err_word) != 0)
    throw api_error(CUGRAPH_UNKNOWN_ERROR, std::string("communicator: a peer did not signal within the timeout (") + where + ", rank " + std::to_string(rank) + ")");
}

void comm_t::all_to_all_v(handle_t const& h, void const* send, std::vector<int64_t> const& send_counts, size_t elem, dev_buf& recv,
                          std::vector<int64_t>& recv_counts)
{
  CGA_EXPECTS((int)send_counts.size() == size && elem % 4 == 0, CUGRAPH_INVALID_INPUT, "communicator: all_to_all_v counts / element size");
  recv_counts.assign(size, 0);
  int64_t total_send = 0;
  for (auto c : send_counts) total_send += c;
  if (size == 1) {
    recv_counts[0] = send_counts[0];
    recv.alloc(std::max<size_t>((size_t)total_send * elem, 4));
    if (total_send > 0) HIP_TRY(hipMemcpyAsync(recv.ptr, send, (size_t)total_send * elem, hipMemcpyDeviceToDevice, h.stream));
    h.sync();
    return;
  }
  std::vector<int64_t> M((size_t)size * size);
  host_allgather(send_counts.data(), (size_t)size * sizeof(int64_t), M.data());
  int64_t total_recv = 0;
  for (int s = 0; s < size; ++s) { recv_counts[s] = M[(size_t)s * size + rank]; total_recv += recv_counts[s]; }
  comm_window_t* w = window_create(std::max<size_t>((size_t)total_recv * elem, 4));
  int64_t soff = 0;
  for (int r = 0; r < size; ++r) {
    int64_t roff = 0;  // where my segment starts in rank r's window: behind the segments of the lower ranks
    for (int s = 0; s < rank; ++s) roff += M[(size_t)s * size + r];
    push_words(h.stream, static_cast<char*>(w->peer[r]) + (size_t)roff * elem, static_cast<char const*>(send) + (size_t)soff * elem,
               send_counts[r] * (int64_t)(elem / 4));
    soff += send_counts[r];
  }
  wait(h.stream, 1, signal(h.stream, 1));
  recv.alloc(std::max<size_t>((size_t)total_recv * elem, 4));
  if (total_recv > 0) HIP_TRY(hipMemcpyAsync(recv.ptr, w->local, (size_t)total_recv * elem, hipMemcpyDeviceToDevice, h.stream));
  h.sync();
  check("all_to_all_v");
  window_free(w);
}

namespace {
template <typename T>
void all_reduce_sum(comm_t& c, handle_t const& h, T* data, int64_t n)
{
  if (c.size == 1 || n <= 0) return;
  int const P         = c.size;
  int64_t const chunk = ((n + P - 1) / P + 3) / 4 * 4;  // rank j reduces elements [j * chunk, (j + 1) * chunk)
  int64_t const wpe   = (int64_t)(sizeof(T) / 4);        // words per element
  auto count_of       = [&](int j) { return std::max<int64_t>(0, std::min(chunk, n - (int64_t)j * chunk)); };
  // window of every rank: stage[P][chunk] (row s = rank s's contribution to MY chunk) followed by gathered[P * chunk]
  comm_window_t* w = c.window_create((size_t)2 * P * chunk * sizeof(T));
  for (int j = 0; j < P; ++j) push_words(h.stream, w->at<T>(j) + (int64_t)c.rank * chunk, data + (int64_t)j * chunk, count_of(j) * wpe);
  c.wait(h.stream, 1, c.signal(h.stream, 1));
  int64_t const my_first = (int64_t)c.rank * chunk, my_cnt = count_of(c.rank);
  T* gathered            = w->at<T>(c.rank) + (int64_t)P * chunk;
  if (my_cnt > 0) {
    hipLaunchKernelGGL(k_fold_chunks<T>, grid_for(my_cnt, 256, 4096), 256, 0, h.stream, (T const*)w->at<T>(c.rank), P, chunk, my_cnt, gathered + my_first);
    for (int r = 0; r < P; ++r)
      if (r != c.rank) push_words(h.stream, w->at<T>(r) + (int64_t)P * chunk + my_first, gathered + my_first, my_cnt * wpe);
  }
  c.wait(h.stream, 1, c.signal(h.stream, 1));
  HIP_TRY(hipMemcpyAsync(data, gathered, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, h.stream));
  h.sync();
  c.check("all_reduce");
  c.window_free(w);
}
}  // namespace

void comm_t::all_reduce_sum_u32(handle_t const& h, uint32_t* data, int64_t n) { all_reduce_sum<uint32_t>(*this, h, data, n); }
void comm_t::all_reduce_sum_f64(handle_t const& h, double* data, int64_t n) { all_reduce_sum<double>(*this, h, data, n); }

void comm_t::all_gather(handle_t const& h, void const* in, size_t bytes, void* out)
{
  CGA_EXPECTS(bytes % 4 == 0, CUGRAPH_INVALID_INPUT, "communicator: all_gather moves 4-byte words");
  if (size == 1) {
    if (bytes) HIP_TRY(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, h.stream));
    h.sync();
    return;
  }
  comm_window_t* w = window_create(std::max<size_t>((size_t)size * bytes, 4));
  for (int r = 0; r < size; ++r) push_words(h.stream, static_cast<char*>(w->peer[r]) + (size_t)rank * bytes, in, (int64_t)(bytes / 4));
  wait(h.stream, 1, signal(h.stream, 1));
  if (bytes) HIP_TRY(hipMemcpyAsync(out, w->local, (size_t)size * bytes, hipMemcpyDeviceToDevice, h.stream));
  h.sync();
  check("all_gather");
  window_free(w);
}

comm_t::~comm_t()
{
  if (has_device) {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
  }
  if (flags) {
    for (int r = 0; r < size; ++r)
      if (r != rank && flags->peer[r]) (void)hipIpcCloseMemHandle(flags->peer[r]);
    (void)hipFree(flags->local);
    delete flags;
  }
  for (auto& ar : arenas) {
    for (int r = 0; r < size; ++r)
      if (r != rank && ar.peer[r]) (void)hipIpcCloseMemHandle(ar.peer[r]);
    (void)hipFree(ar.local);
  }
  if (d_peer_flags) (void)hipFree(d_peer_flags);
  if (err_word) (void)hipHostFree(err_word);
  if (shm) {
    uint32_t const left = shm->attached.fetch_sub(1) - 1;
    munmap(shm, sizeof(comm_shm_t));
    if (left == 0 || rank == 0) shm_unlink(shm_name(session).c_str());
  }
  if (shm_fd >= 0) close(shm_fd);
  magic = 0;
}

// the host half of the bootstrap: attaches to (rank 0: creates) the session's shared-memory segment; no HIP call
static std::unique_ptr<comm_t> attach_session(char const* session, int rank, int size, double timeout_s)
{
  CGA_EXPECTS(session != nullptr && session[0] != 0, CUGRAPH_INVALID_INPUT, "communicator: empty session name");
  CGA_EXPECTS(size >= 1 && size <= kCommMaxRanks && rank >= 0 && rank < size, CUGRAPH_INVALID_INPUT, "communicator: rank / size out of range (at most 64 ranks on one node)");
  auto c       = std::make_unique<comm_t>();
  c->rank      = rank;
  c->size      = size;
  c->session   = session;
  c->timeout_s = timeout_s > 0 ? timeout_s : 60.0;
  if (char const* e = getenv("CUGRAPH_AMD_COMM_TIMEOUT_S")) c->timeout_s = std::max(1.0, atof(e));
  std::string const name = shm_name(session);
  double const t0        = now_s();
  // A crashed job may have left a segment of this name behind.  Rank 0 unlinks and re-creates it; a rank that starts BEFORE rank 0 must not
  // settle on the old one: it trusts a segment only while the process that created it is alive (pid0), and while it waits in the first
  // barrier it keeps checking that the name still leads to the file it mapped -- if rank 0 has re-created the session meanwhile, it moves over.
  auto pid_namespace = [] {  // inode of this process's PID namespace (0: not available)
    struct stat st;
    return stat("/proc/self/ns/pid", &st) == 0 ? (uint64_t)st.st_ino : (uint64_t)0;
  };
  auto still_linked = [&](int fd) {
    struct stat a, b;
    std::string const path = "/dev/shm" + name;
    return fstat(fd, &a) == 0 && stat(path.c_str(), &b) == 0 && a.st_ino == b.st_ino && a.st_dev == b.st_dev;
  };
  if (rank == 0) {
    shm_unlink(name.c_str());  // a stale segment of a crashed job with the same session name
    c->shm_fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    CGA_EXPECTS(c->shm_fd >= 0, CUGRAPH_UNKNOWN_ERROR, "communicator: shm_open(" + name + ") failed: " + strerror(errno));
    CGA_EXPECTS(ftruncate(c->shm_fd, sizeof(comm_shm_t)) == 0, CUGRAPH_UNKNOWN_ERROR, "communicator: ftruncate failed");
    void* m = mmap(nullptr, sizeof(comm_shm_t), PROT_READ | PROT_WRITE, MAP_SHARED, c->shm_fd, 0);
    CGA_EXPECTS(m != MAP_FAILED, CUGRAPH_UNKNOWN_ERROR, "communicator: mmap failed");
    c->shm = static_cast<comm_shm_t*>(m);
    std::memset(m, 0, sizeof(comm_shm_t));
    c->shm->size = (uint32_t)size;
    c->shm->pid0 = (uint32_t)getpid();
    c->shm->pidns0 = pid_namespace();
    c->shm->ready.store(kCommMagic, std::memory_order_release);
    c->shm->attached.fetch_add(1);
    c->host_barrier();
    return c;
  }
  for (;;) {
    auto expired = [&] { return now_s() - t0 >= c->timeout_s; };
    c->shm_fd = shm_open(name.c_str(), O_RDWR, 0600);
    if (c->shm_fd >= 0) {
      struct stat st;
      if (!(fstat(c->shm_fd, &st) == 0 && (size_t)st.st_size >= sizeof(comm_shm_t))) { close(c->shm_fd); c->shm_fd = -1; }
    }
    if (c->shm_fd < 0) {
      CGA_EXPECTS(!expired(), CUGRAPH_UNKNOWN_ERROR, "communicator: rank 0 never created the session " + name);
      usleep(2000);
      continue;
    }
    void* m = mmap(nullptr, sizeof(comm_shm_t), PROT_READ | PROT_WRITE, MAP_SHARED, c->shm_fd, 0);
    CGA_EXPECTS(m != MAP_FAILED, CUGRAPH_UNKNOWN_ERROR, "communicator: mmap failed");
    c->shm = static_cast<comm_shm_t*>(m);
    auto drop = [&] { munmap(c->shm, sizeof(comm_shm_t)); c->shm = nullptr; close(c->shm_fd); c->shm_fd = -1; };
    bool stale = false;
    while (c->shm->ready.load(std::memory_order_acquire) != kCommMagic) {
      if (!still_linked(c->shm_fd)) { stale = true; break; }
      CGA_EXPECTS(!expired(), CUGRAPH_UNKNOWN_ERROR, "communicator: session " + name + " was never initialised");
      usleep(1000);
    }
    if (!stale) {
      // the creator is gone: a crashed job's segment.  "Gone" can only be read off the pid by a rank that shares the creator's PID namespace; across
      // namespaces (a container per GPU over one /dev/shm) the pid is invisible although its process lives, and the segment is trusted as long as the
      // name still leads to it (still_linked, re-checked inside the first barrier's wait): rank 0 unlinks a crashed job's segment before it creates its own
      uint32_t const creator = c->shm->pid0;
      uint64_t const ns = c->shm->pidns0, my_ns = pid_namespace();
      bool const same_ns = ns != 0 && ns == my_ns;
      stale = creator == 0 || (same_ns && kill((pid_t)creator, 0) != 0 && errno == ESRCH) || !still_linked(c->shm_fd);
      // an abort flag that is set before this rank has even attached is a crashed job's (rank 0 creates a session with the flag clear)
      stale = stale || c->shm->abort_flag.load(std::memory_order_relaxed) != 0;
    }
    if (stale) {
      drop();
      CGA_EXPECTS(!expired(), CUGRAPH_UNKNOWN_ERROR, "communicator: only a stale segment of the session " + name + " was found (rank 0 never re-created it)");
      usleep(5000);
      continue;
    }
    CGA_EXPECTS(c->shm->size == (uint32_t)size, CUGRAPH_INVALID_INPUT, "communicator: ranks disagree on the communicator size");
    c->shm->attached.fetch_add(1);
    // the first barrier, with the freshness check inside its wait
    uint32_t const gen = c->shm->bar_gen.load(std::memory_order_acquire);
    if (c->shm->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)size) {
      c->shm->bar_count.store(0, std::memory_order_relaxed);
      c->shm->bar_gen.fetch_add(1, std::memory_order_release);
      return c;
    }
    int spins = 0;
    while (c->shm->bar_gen.load(std::memory_order_acquire) == gen) {
      if (c->shm->abort_flag.load(std::memory_order_relaxed)) {
        // (a crashed job leaves its abort flag set: a segment the name no longer leads to is that job's, not a peer's verdict -- move over to rank 0's new one)
        if (!still_linked(c->shm_fd)) { stale = true; break; }
        throw api_error(CUGRAPH_UNKNOWN_ERROR, "communicator: a peer aborted");
      }
      if (++spins > 200) sched_yield();
      if ((spins & 1023) == 0) {
        if (!still_linked(c->shm_fd)) { stale = true; break; }  // rank 0 has re-created the session under our feet
        if (expired()) {
          c->shm->abort_flag.store(1, std::memory_order_relaxed);
          throw api_error(CUGRAPH_UNKNOWN_ERROR, "communicator: host barrier timed out on rank " + std::to_string(rank) + " (a peer is missing or has failed)");
        }
      }
    }
    if (!stale) return c;
    drop();
  }
  return c;
}

comm_t* comm_create(char const* session, int rank, int size, double timeout_s)
{
  auto c = attach_session(session, rank, size, timeout_s);
  HIP_TRY(hipGetDevice(&c->device));
  c->has_device = true;
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) c->wall_ticks_per_s = (uint64_t)khz * 1000ull;
  else (void)hipGetLastError();
  // which GPU is each rank on?
  char bus[64] = {0};
  HIP_TRY(hipDeviceGetPCIBusId(bus, sizeof(bus), c->device));
  std::vector<char> all((size_t)size * sizeof(bus));
  c->host_allgather(bus, sizeof(bus), all.data());
  for (int r = 0; r < size; ++r)
    if (std::strncmp(bus, all.data() + (size_t)r * sizeof(bus), sizeof(bus)) != 0) c->multi_device = true;
  HIP_TRY(hipHostMalloc((void**)&c->err_word, 64, hipHostMallocMapped));
  std::memset(c->err_word, 0, 64);
  // the flag words: zeroed BEFORE anybody can map them
  size_t const fbytes = (size_t)kCommChannels * kCommMaxRanks * sizeof(uint64_t);
  {
    auto w   = std::make_unique<comm_window_t>();
    w->local = raw_alloc(*c, fbytes);
    HIP_TRY(hipMemset(w->local, 0, fbytes));
    HIP_TRY(hipDeviceSynchronize());
    w->bytes.assign(size, fbytes);
    exchange_mappings(*c, w->local, fbytes, w->peer);
    c->flags = w.release();
  }
  HIP_TRY(hipMalloc((void**)&c->d_peer_flags, (size_t)size * sizeof(uint64_t*)));
  HIP_TRY(hipMemcpy(c->d_peer_flags, c->flags->peer.data(), (size_t)size * sizeof(uint64_t*), hipMemcpyHostToDevice));
  return c.release();
}

}  // namespace cga

// ------------------------------------------------------------------------------------------------ C ABI
using namespace cga;

extern "C" cugraph_error_code_t cugraph_amd_comm_create(const char* session, int rank, int size, cugraph_amd_comm_t** comm, cugraph_error_t** error)
{
  if (comm) *comm = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(comm != nullptr, CUGRAPH_INVALID_INPUT, "comm is NULL");
    int ndev = 0;
    CGA_EXPECTS(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0, CUGRAPH_UNKNOWN_ERROR, "communicator: no HIP device");
    *comm = reinterpret_cast<cugraph_amd_comm_t*>(comm_create(session, rank, size, 0.0));
  });
}

extern "C" void cugraph_amd_comm_free(cugraph_amd_comm_t* comm)
{
  auto* c = reinterpret_cast<comm_t*>(comm);
  if (c && c->magic == kCommMagic) delete c;
}

// The bootstrap without a GPU (CPU tests): attach, then `rounds` barriers and all-gathers whose contents are checked.
extern "C" cugraph_error_code_t cugraph_amd_comm_host_selftest(const char* session, int rank, int size, int rounds, cugraph_error_t** error)
{
  return guarded(error, [&] {
    auto c = attach_session(session, rank, size, 0.0);
    std::vector<uint64_t> all((size_t)size * 8);
    for (int k = 0; k < rounds; ++k) {
      uint64_t mine[8];
      for (int j = 0; j < 8; ++j) mine[j] = (uint64_t)rank * 1000003ull + (uint64_t)k * 17ull + (uint64_t)j;
      c->host_allgather(mine, sizeof(mine), all.data());
      for (int r = 0; r < size; ++r)
        for (int j = 0; j < 8; ++j)
          CGA_EXPECTS(all[(size_t)r * 8 + j] == (uint64_t)r * 1000003ull + (uint64_t)k * 17ull + (uint64_t)j, CUGRAPH_UNKNOWN_ERROR, "host_allgather delivered wrong data");
      c->host_barrier();
    }
  });
}

extern "C" cugraph_error_code_t cugraph_amd_comm_host_barrier(cugraph_amd_comm_t* comm, cugraph_error_t** error)
{
  return guarded(error, [&] {
    auto* c = reinterpret_cast<comm_t*>(comm);
    CGA_EXPECTS(c != nullptr && c->magic == kCommMagic, CUGRAPH_INVALID_INPUT, "not a communicator");
    c->host_barrier();
  });
}

extern "C" cugraph_error_code_t cugraph_amd_comm_host_allgather(cugraph_amd_comm_t* comm, const void* in, size_t bytes, void* out, cugraph_error_t** error)
{
  return guarded(error, [&] {
    auto* c = reinterpret_cast<comm_t*>(comm);
    CGA_EXPECTS(c != nullptr && c->magic == kCommMagic && in != nullptr && out != nullptr, CUGRAPH_INVALID_INPUT, "not a communicator / NULL buffer");
    c->host_allgather(in, bytes, out);
  });
}

extern "C" int cugraph_amd_comm_rank(const cugraph_amd_comm_t* comm) { return comm ? reinterpret_cast<comm_t const*>(comm)->rank : 0; }
extern "C" int cugraph_amd_comm_size(const cugraph_amd_comm_t* comm) { return comm ? reinterpret_cast<comm_t const*>(comm)->size : 0; }

namespace {
__global__ void k_selftest_fill(uint32_t* p, int64_t n, uint32_t tag)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = tag * 1000003u + (uint32_t)i;
}
// exactly representable, and their sum depends on the order: +2^60, 1, -2^60, 1, ... (in rank order: 2^60 + 1 rounds to 2^60)
__host__ __device__ inline double selftest_value(int rank, int64_t i)
{
  double const big = 1152921504606846976.0;
  int const k      = (rank + (int)(i % 4)) % 4;
  return k == 0 ? big : k == 2 ? -big : 1.0;
}
__global__ void k_selftest_fill_f64(double* p, int64_t n, int rank)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = selftest_value(rank, i);
}
}  // namespace

// Exercises every primitive against closed-form expectations and times the two that sit on the per-iteration path.
// out[0] = microseconds per device barrier (signal + wait across all ranks), out[1] = GB/s of peer pushes issued by this rank,
// out[2] = 1 when peers sit on different GPUs.  Collective.
extern "C" cugraph_error_code_t cugraph_amd_comm_selftest(const cugraph_resource_handle_t* handle, size_t n_words, int iterations, double* out,
                                                          cugraph_error_t** error)
{
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    comm_t* cp        = handle_comm(h);
    CGA_EXPECTS(cp != nullptr, CUGRAPH_INVALID_INPUT, "selftest: the handle was not created on a communicator");
    comm_t& c    = *cp;
    int const P  = c.size, me = c.rank;
    int64_t const n = std::max<int64_t>((int64_t)n_words, 64);
    HIP_TRY(hipSetDevice(h.device));
    // all_gather
    {
      dvec<uint32_t> in((size_t)n), outv((size_t)n * P);
      hipLaunchKernelGGL(k_selftest_fill, grid_for(n), 256, 0, h.stream, in.data(), n, (uint32_t)me + 1);
      c.all_gather(h, in.data(), (size_t)n * 4, outv.data());
      std::vector<uint32_t> host((size_t)n * P);
      HIP_TRY(hipMemcpy(host.data(), outv.data(), host.size() * 4, hipMemcpyDeviceToHost));
      for (int r = 0; r < P; ++r)
        for (int64_t i = 0; i < n; i += std::max<int64_t>(1, n / 257))
          CGA_EXPECTS(host[(size_t)r * n + i] == (uint32_t)(r + 1) * 1000003u + (uint32_t)i, CUGRAPH_UNKNOWN_ERROR, "selftest: all_gather delivered wrong data");
    }
    // all_to_all_v: rank s sends (s + r + 1) * 5 words tagged 100 * s + r to rank r
    {
      std::vector<int64_t> sc(P), rc;
      int64_t tot = 0;
      for (int r = 0; r < P; ++r) { sc[r] = (int64_t)(me + r + 1) * 5; tot += sc[r]; }
      dvec<uint32_t> send((size_t)tot);
      int64_t off = 0;
      for (int r = 0; r < P; ++r) {
        hipLaunchKernelGGL(k_selftest_fill, 1, 64, 0, h.stream, send.data() + off, sc[r], (uint32_t)(100 * me + r));
        off += sc[r];
      }
      dev_buf recv;
      c.all_to_all_v(h, send.data(), sc, 4, recv, rc);
      int64_t rt = 0;
      for (int s = 0; s < P; ++s) { CGA_EXPECTS(rc[s] == (int64_t)(s + me + 1) * 5, CUGRAPH_UNKNOWN_ERROR, "selftest: all_to_all_v counts"); rt += rc[s]; }
      std::vector<uint32_t> host((size_t)rt);
      HIP_TRY(hipMemcpy(host.data(), recv.ptr, host.size() * 4, hipMemcpyDeviceToHost));
      int64_t at = 0;
      for (int s = 0; s < P; ++s)
        for (int64_t i = 0; i < rc[s]; ++i, ++at)
          CGA_EXPECTS(host[at] == (uint32_t)(100 * s + me) * 1000003u + (uint32_t)i, CUGRAPH_UNKNOWN_ERROR, "selftest: all_to_all_v delivered wrong data");
    }
    // all_reduce (integer and double; a length that is not a multiple of the rank count).  Integer: rank r contributes 1 << r everywhere,
    // so a wrong sum names the ranks whose contribution is missing or stale
    {
      int64_t const m = n + 3;
      dvec<uint32_t> a((size_t)m);
      dvec<double> d((size_t)m);
      fill_u32(h, a.data(), m, 1u << (me % 31));
      hipLaunchKernelGGL(k_selftest_fill_f64, grid_for(m), 256, 0, h.stream, d.data(), m, me);
      c.all_reduce_sum_u32(h, a.data(), m);
      c.all_reduce_sum_f64(h, d.data(), m);
      std::vector<uint32_t> ha((size_t)m);
      std::vector<double> hd((size_t)m);
      HIP_TRY(hipMemcpy(ha.data(), a.data(), ha.size() * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(hd.data(), d.data(), hd.size() * 8, hipMemcpyDeviceToHost));
      uint32_t want_u = 0;
      for (int r = 0; r < P; ++r) want_u += 1u << (r % 31);
      int64_t bad_u = 0, bad_d = 0, first_u = -1, first_d = -1;
      for (int64_t i = 0; i < m; ++i) {
        if (ha[i] != want_u) { if (bad_u++ == 0) first_u = i; }
        double want = selftest_value(0, i);
        for (int r = 1; r < P; ++r) want += selftest_value(r, i);  // the library's fold order
        if (hd[i] != want) { if (bad_d++ == 0) first_d = i; }
      }
      if (bad_u || bad_d) {
        char msg[512];
        snprintf(msg, sizeof(msg), "selftest: all_reduce wrong on rank %d of %d (n = %lld): integer %lld wrong, first at %lld (got 0x%x, want 0x%x); double %lld wrong, first at %lld (got %.17g)",
                 me, P, (long long)m, (long long)bad_u, (long long)first_u, first_u >= 0 ? ha[first_u] : 0u, want_u, (long long)bad_d, (long long)first_d, first_d >= 0 ? hd[first_d] : 0.0);
        throw api_error(CUGRAPH_UNKNOWN_ERROR, msg);
      }
    }
    // timing: device barriers back to back; pushes of n words to every peer
    double bar_us = 0.0, gbps = 0.0;
    int const it = std::max(iterations, 1);
    {
      c.device_barrier(h.stream);
      h.sync();
      c.host_barrier();
      double const t0 = now_s();
      for (int k = 0; k < it; ++k) c.device_barrier(h.stream);
      h.sync();
      bar_us = (now_s() - t0) / it * 1e6;
      c.check("selftest barriers");
    }
    {
      comm_window_t* w = c.window_create((size_t)n * 4 * P);
      dvec<uint32_t> src((size_t)n);
      hipLaunchKernelGGL(k_selftest_fill, grid_for(n), 256, 0, h.stream, src.data(), n, 7u);
      c.device_barrier(h.stream);
      h.sync();
      c.host_barrier();
      double const t0 = now_s();
      for (int k = 0; k < it; ++k) {
        for (int r = 0; r < P; ++r) push_words(h.stream, w->at<uint32_t>(r) + (int64_t)me * n, src.data(), n);
        c.device_barrier(h.stream);
      }
      h.sync();
      gbps = (double)n * 4.0 * P * it / (now_s() - t0) / 1e9;
      c.check("selftest pushes");
      c.window_free(w);
    }
    if (out) { out[0] = bar_us; out[1] = gbps; out[2] = c.multi_device ? 1.0 : 0.0; }
  });
}
