// Device building blocks written directly for gfx950 (wave64): fills, reductions, prefix sums, histogram,
// stable LSD radix sort, gathers.  These replace the thrust::/cub:: call sites the reference's graph
// construction sits on (renumber_edgelist_impl.cuh:425-829,
// structure/detail/structure_utils.cuh:297-464).  No Thrust / CUB / rocPRIM / hipCUB.
#include "common.hpp"
#include <mutex>

namespace cga {

namespace {

constexpr int WAVE = 64;

template <typename T>
__global__ void k_fill(T* p, int64_t n, T v)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

__global__ void k_iota(int32_t* p, int64_t n, int32_t first)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = first + (int32_t)i;
}

__device__ __forceinline__ int32_t wave_min(int32_t v)
{
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int32_t wave_max(int32_t v)
{
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

__global__ void k_minmax(int32_t const* p, int64_t n, int32_t* out /* [min,max] */)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int32_t mn = INT32_MAX, mx = INT32_MIN;
  for (; i < n; i += stride) { int32_t v = p[i]; mn = min(mn, v); mx = max(mx, v); }
  mn = wave_min(mn);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { atomicMin(out, mn); atomicMax(out + 1, mx); }
}

__global__ void k_count_negative(int32_t const* p, int64_t n, unsigned long long* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned c     = 0;
  for (; i < n; i += stride) c += p[i] < 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

// ---------------------------------------------------------------------------------- prefix sum
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS   = 8;
constexpr int SCAN_TILE    = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
  for (int o = 1; o < WAVE; o <<= 1) {
    uint32_t t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns the exclusive prefix and
// writes the block total to *total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total)
{
  __shared__ uint32_t wsum[SCAN_THREADS / WAVE];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = wave_inclusive_scan(v, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < SCAN_THREADS / WAVE; ++k) {
    uint32_t s = wsum[k];
    if (k < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void k_scan_tile_sums(uint32_t const* in, int64_t n, uint32_t* sums)
{
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s   = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k)
    if (base + k < n) s += in[base + k];
  uint32_t tot;
  (void)block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ void k_scan_tile_apply(uint32_t const* in, uint32_t* out, int64_t n, uint32_t const* tile_base)
{
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0u;
    s += v[k];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(s, &tot) + (tile_base ? tile_base[blockIdx.x] : 0u);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
}

// ----------------------------------------------------------------------------------- histogram
// counts[map(keys[i])] += 1, map(k) = rank ? rank[k - vmin] : k.  Power-law inputs put millions of increments on a few
// counters, and same-address global atomics serialise in one L2 channel (k_degree_compact: 100 ms at RMAT-26 with one
// atomic per edge).  Every workgroup therefore counts its slice in a direct-mapped LDS cache (LDS atomics on one address
// run at LDS speed); a key that loses its slot to another key goes straight to the global counter -- those are the cold
// keys, which do not contend -- and the cache is flushed with one global atomic per occupied slot.
constexpr int HC_SLOTS = 4096;
constexpr int64_t HC_CHUNK = 65536;  // keys per workgroup
__global__ void __launch_bounds__(256) k_histogram(int32_t const* keys, int64_t n, int64_t vmin, uint32_t const* rank, uint32_t* counts)
{
  __shared__ uint32_t s_key[HC_SLOTS];
  __shared__ uint32_t s_cnt[HC_SLOTS];
  for (int t = threadIdx.x; t < HC_SLOTS; t += blockDim.x) { s_key[t] = 0xFFFFFFFFu; s_cnt[t] = 0; }
  __syncthreads();
  int64_t const b = (int64_t)blockIdx.x * HC_CHUNK, e = b + HC_CHUNK < n ? b + HC_CHUNK : n;
  for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    uint32_t const k    = rank ? rank[(int64_t)keys[i] - vmin] : (uint32_t)keys[i];
    uint32_t const slot = (k * 2654435761u) >> 20;  // 12 bits
    uint32_t const prev = atomicCAS(&s_key[slot], 0xFFFFFFFFu, k);
    if (prev == 0xFFFFFFFFu || prev == k) atomicAdd(&s_cnt[slot], 1u);
    else atomicAdd(&counts[k], 1u);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < HC_SLOTS; t += blockDim.x)
    if (s_cnt[t]) atomicAdd(&counts[s_key[t]], s_cnt[t]);
}

// Large inputs (10^9 keys over 10^7..10^8 counters): the cold keys above -- most of them -- are one random read-modify-write of
// a 64-byte line each (73 ms at RMAT-26).  histogram_i32_mapped therefore first groups the keys by their top 16 bits (two
// passes of the radix partition below), after which a chunk of consecutive keys maps into a narrow window of counters: the
// window is counted in LDS (32 Ki counters) and flushed once; only keys beyond the window (sparse regions) fall back to a
// global atomic, and those now land in lines the neighbouring chunks keep in L2.
constexpr int HW_WINDOW    = 32768;
constexpr int64_t HW_CHUNK = 131072;
__global__ void k_hist_widen(int32_t const* keys, int64_t n, int64_t vmin, uint64_t* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (uint64_t)((int64_t)keys[i] - vmin);
}
__global__ void __launch_bounds__(1024) k_histogram_window(uint64_t const* keys, int64_t n, int low_bits, uint32_t const* rank, uint32_t* counts)
{
  extern __shared__ uint32_t s_win[];
  for (int t = threadIdx.x; t < HW_WINDOW; t += blockDim.x) s_win[t] = 0;
  __syncthreads();
  int64_t const b = (int64_t)blockIdx.x * HW_CHUNK, e = b + HW_CHUNK < n ? b + HW_CHUNK : n;
  // keys ascend in their bits >= low_bits; map() is monotone: nothing in the chunk maps below the first key's group start
  uint64_t const k0   = (keys[b] >> low_bits) << low_bits;
  uint32_t const base = rank ? rank[k0] : (uint32_t)k0;
  for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    uint64_t const kk  = keys[i];
    uint32_t const k   = rank ? rank[kk] : (uint32_t)kk;
    uint32_t const off = k - base;
    if (off < (uint32_t)HW_WINDOW) atomicAdd(&s_win[off], 1u);
    else atomicAdd(&counts[k], 1u);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < HW_WINDOW; t += blockDim.x) {
    uint32_t const c = s_win[t];
    if (c) atomicAdd(&counts[(size_t)base + t], c);
  }
}

// ---------------------------------------------------------------------------------- radix sort
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES   = RS_THREADS / WAVE;
constexpr int RS_PER_WAVE_ITERS = 16;
constexpr int RS_WAVE_CHUNK = WAVE * RS_PER_WAVE_ITERS;  // 1024 keys per wave
constexpr int RS_TILE    = RS_WAVE_CHUNK * RS_WAVES;     // 4096 keys per block
constexpr int RS_BINS    = 256;

// lanes of the wavefront that hold the same digit as this lane (among the valid ones): 8 ballots
__device__ __forceinline__ uint64_t rs_match_digit(uint32_t d, bool valid)
{
  uint64_t peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    bool const bit     = (d >> b) & 1u;
    uint64_t const bal = __ballot(bit);
    peers &= bit ? bal : ~bal;
  }
  return peers;
}

// Digit counts per tile.  One LDS atomic per (wavefront, distinct digit) instead of one per key: the sort keys of a degree-sorted
// graph put most of a tile into one or two bins of the high digits, and same-address LDS atomics serialise.
__global__ void __launch_bounds__(RS_THREADS)
k_rs_hist(uint64_t const* keys, int64_t n, int shift, uint32_t mask, uint32_t* hist, int nblocks)
{
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  int const lane = threadIdx.x & 63;
  int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int it = 0; it < RS_TILE / RS_THREADS; ++it) {
    int64_t idx       = base + it * RS_THREADS + threadIdx.x;
    bool const valid  = idx < n;
    uint32_t const d  = valid ? (uint32_t)(keys[idx] >> shift) & mask : 0u;
    // skewed digit (many lanes share the first lane's): one atomic per distinct digit; flat digit: plain atomics are cheaper
    uint32_t const d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
    if (__popcll(__ballot(valid && d == d0)) >= 8) {
      uint64_t const pe = rs_match_digit(d, valid);
      if (valid && lane == __ffsll((unsigned long long)pe) - 1) atomicAdd(&h[d], (uint32_t)__popcll(pe));
    } else if (valid) {
      atomicAdd(&h[d], 1u);
    }
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one 8-bit digit.  The tile (RS_TILE keys) is first sorted by digit INSIDE LDS -- rank of a key = start
// of its digit in the tile + keys of that digit in earlier wavefronts + rank inside its wavefront -- and then written out in
// tile order, so consecutive threads write consecutive addresses within each digit's run (16 keys = 128 B on average) instead
// of one isolated 8-byte store per key (the first version: 1.6 TB/s at RMAT-26).  A wavefront's 1024 keys are read ONCE, all
// 16 loads in flight, and stay in registers; the rank inside the wavefront comes from the ballot match (no per-key LDS atomic).
__global__ void __launch_bounds__(RS_THREADS)
k_rs_scatter(uint64_t const* keys_in, uint32_t const* vals_in, uint64_t* keys_out, uint32_t* vals_out,
             int64_t n, int shift, uint32_t mask, uint32_t const* offs, int nblocks)
{
  __shared__ uint32_t cnt[RS_WAVES][RS_BINS];    // keys of (wave, digit); after the scan: tile-local position of the first of them
  __shared__ uint32_t lstart[RS_BINS];           // tile-local start of the digit
  __shared__ uint32_t gbase[RS_BINS];            // global start of the digit's run of this tile
  __shared__ uint32_t wsum[RS_WAVES];
  __shared__ uint64_t skey[RS_TILE];
  __shared__ uint32_t sval[RS_TILE];
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) cnt[w][threadIdx.x] = 0;
  __syncthreads();
  int64_t const tile0 = (int64_t)blockIdx.x * RS_TILE;
  int64_t const chunk = tile0 + (int64_t)wave * RS_WAVE_CHUNK;
  uint64_t key[RS_PER_WAVE_ITERS];
  uint32_t val[RS_PER_WAVE_ITERS];
  uint32_t dr[RS_PER_WAVE_ITERS];  // digit | rank among the wavefront's keys of that digit << 8
#pragma unroll
  for (int it = 0; it < RS_PER_WAVE_ITERS; ++it) {
    int64_t const idx = chunk + it * WAVE + lane;
    key[it] = idx < n ? keys_in[idx] : 0ull;
    val[it] = (vals_in && idx < n) ? vals_in[idx] : 0u;
  }
  uint64_t const lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  volatile uint32_t* my_cnt = cnt[wave];
#pragma unroll
  for (int it = 0; it < RS_PER_WAVE_ITERS; ++it) {
    bool const valid  = chunk + it * WAVE + lane < n;
    uint32_t const d  = (uint32_t)(key[it] >> shift) & mask;
    uint64_t const pe = rs_match_digit(d, valid);
    uint32_t const before = my_cnt[d];  // every lane reads before the group's first lane writes (LDS operations of a wavefront execute in order)
    uint32_t const rank   = before + (uint32_t)__popcll(pe & lt_mask);
    if (valid && (pe & lt_mask) == 0) my_cnt[d] = before + (uint32_t)__popcll(pe);
    dr[it] = d | (rank << 8);
  }
  __syncthreads();
  {  // thread d: digit d.  Exclusive scan of the digit totals over the workgroup (RS_THREADS == RS_BINS)
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) tot += cnt[w][threadIdx.x];
    uint32_t inc = tot;
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    uint32_t run          = before + inc - tot;
    lstart[threadIdx.x]   = run;
    gbase[threadIdx.x]    = offs[(int64_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      uint32_t const c     = cnt[w][threadIdx.x];
      cnt[w][threadIdx.x]  = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RS_PER_WAVE_ITERS; ++it) {
    if (chunk + it * WAVE + lane < n) {
      uint32_t const pos = cnt[wave][dr[it] & 0xFFu] + (dr[it] >> 8);
      skey[pos]          = key[it];
      if (vals_in) sval[pos] = val[it];
    }
  }
  __syncthreads();
  int64_t const left = n - tile0;
  uint32_t const nt  = left < (int64_t)RS_TILE ? (uint32_t)left : (uint32_t)RS_TILE;
  for (uint32_t t = threadIdx.x; t < nt; t += RS_THREADS) {
    uint64_t const key_t = skey[t];
    uint32_t const d     = (uint32_t)(key_t >> shift) & mask;
    uint32_t const pos   = gbase[d] + (t - lstart[d]);
    keys_out[pos]        = key_t;
    if (vals_in) vals_out[pos] = sval[t];
  }
}

template <typename T>
__global__ void k_gather(T const* src, uint32_t const* idx, T* out, int64_t n)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = src[idx[i]];
}

}  // namespace

void fill_i32(handle_t const& h, int32_t* p, int64_t n, int32_t v)
{
  if (n > 0) hipLaunchKernelGGL(k_fill<int32_t>, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, p, n, v);
}
void fill_u32(handle_t const& h, uint32_t* p, int64_t n, uint32_t v)
{
  if (n > 0) hipLaunchKernelGGL(k_fill<uint32_t>, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, p, n, v);
}
void fill_f32(handle_t const& h, float* p, int64_t n, float v)
{
  if (n > 0) hipLaunchKernelGGL(k_fill<float>, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, p, n, v);
}
void fill_f64(handle_t const& h, double* p, int64_t n, double v)
{
  if (n > 0) hipLaunchKernelGGL(k_fill<double>, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, p, n, v);
}
void iota_i32(handle_t const& h, int32_t* p, int64_t n, int32_t first)
{
  if (n > 0) hipLaunchKernelGGL(k_iota, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, p, n, first);
}

void minmax_i32(handle_t const& h, int32_t const* p, int64_t n, int32_t* mn, int32_t* mx)
{
  dvec<int32_t> out(2);
  int32_t init[2] = {INT32_MAX, INT32_MIN};
  HIP_TRY(hipMemcpyAsync(out.data(), init, sizeof(init), hipMemcpyHostToDevice, h.stream));
  h.sync();  // `init` is a stack buffer
  hipLaunchKernelGGL(k_minmax, grid_for(n, kBlock, 2048), kBlock, 0, h.stream, p, n, out.data());
  int32_t r[2];
  h.read_back(r, out.data(), 2);
  *mn = r[0];
  *mx = r[1];
}

int64_t count_negative_i32(handle_t const& h, int32_t const* ids, int64_t n)
{
  if (n == 0) return 0;
  dvec<unsigned long long> out(1);
  HIP_TRY(hipMemsetAsync(out.data(), 0, 8, h.stream));
  hipLaunchKernelGGL(k_count_negative, grid_for(n, kBlock, 2048), kBlock, 0, h.stream, ids, n, out.data());
  unsigned long long r;
  h.read_back(&r, out.data(), 1);
  return (int64_t)r;
}

void exclusive_scan_u32(handle_t const& h, uint32_t const* in, uint32_t* out, int64_t n)
{
  if (n <= 0) return;
  int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (ntiles == 1) {
    hipLaunchKernelGGL(k_scan_tile_apply, 1, SCAN_THREADS, 0, h.stream, in, out, n, (uint32_t const*)nullptr);
    return;
  }
  dvec<uint32_t> sums(ntiles);
  hipLaunchKernelGGL(k_scan_tile_sums, (int)ntiles, SCAN_THREADS, 0, h.stream, in, n, sums.data());
  exclusive_scan_u32(h, sums.data(), sums.data(), ntiles);
  hipLaunchKernelGGL(k_scan_tile_apply, (int)ntiles, SCAN_THREADS, 0, h.stream, in, out, n,
                     (uint32_t const*)sums.data());
  h.sync();  // `sums` is freed on return
}

void histogram_i32(handle_t const& h, int32_t const* keys, int64_t n, uint32_t* counts, int64_t range)
{
  histogram_i32_mapped(h, keys, n, 0, nullptr, counts, range);
}

void histogram_i32_mapped(handle_t const& h, int32_t const* keys, int64_t n, int64_t vmin, uint32_t const* rank, uint32_t* counts,
                          int64_t range)
{
  if (n <= 0) return;
  char const* mode = getenv("CUGRAPH_AMD_HISTOGRAM");  // "direct" / "partition" force a path (tests, A/B runs)
  bool partition = mode && mode[0] == 'p' ? range > 0 : (n >= ((int64_t)1 << 24) && range >= ((int64_t)1 << 20) && !(mode && mode[0] == 'd'));
  if (h.lds_per_block < (size_t)HW_WINDOW * sizeof(uint32_t) + 1024) partition = false;  // the counting window must fit one workgroup's LDS
  if (!partition) {
    hipLaunchKernelGGL(k_histogram, (int)((n + HC_CHUNK - 1) / HC_CHUNK), 256, 0, h.stream, keys, n, vmin, rank, counts);
    return;
  }
  dvec<uint64_t> wide((size_t)n), tmp((size_t)n);
  hipLaunchKernelGGL(k_hist_widen, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, keys, n, vmin, wide.data());
  int bits = 0;
  while (bits < 63 && ((uint64_t)(range - 1) >> bits) != 0) ++bits;
  int const low_bits = bits > 16 ? bits - 16 : 0;
  radix_sort_u64_u32(h, wide.data(), nullptr, tmp.data(), nullptr, n, low_bits, bits);
  {  // per device (the attribute belongs to the function ON a device), guarded for callers on several host threads
    static std::mutex attr_m;
    static std::vector<bool> attr_done;
    std::lock_guard<std::mutex> lock(attr_m);
    if ((int)attr_done.size() <= h.device) attr_done.resize(h.device + 1, false);
    if (!attr_done[h.device]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(k_histogram_window), hipFuncAttributeMaxDynamicSharedMemorySize, HW_WINDOW * 4));
      attr_done[h.device] = true;
    }
  }
  hipLaunchKernelGGL(k_histogram_window, (int)((n + HW_CHUNK - 1) / HW_CHUNK), 1024, HW_WINDOW * sizeof(uint32_t), h.stream,
                     (uint64_t const*)wide.data(), n, low_bits, rank, counts);
  h.sync();  // temporaries die here
}

void radix_sort_u64_u32(handle_t const& h, uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp,
                        uint32_t* vals_tmp, int64_t n, int bit_lo, int bit_hi)
{
  if (n <= 1 || bit_hi <= bit_lo) return;
  int64_t nblocks64 = (n + RS_TILE - 1) / RS_TILE;
  CGA_EXPECTS(nblocks64 < (int64_t)1 << 30, CUGRAPH_UNKNOWN_ERROR, "radix sort: input too large");
  int nblocks = (int)nblocks64;
  dvec<uint32_t> hist((size_t)nblocks * RS_BINS);
  uint64_t* kin  = keys;
  uint64_t* kout = keys_tmp;
  uint32_t* vin  = vals;
  uint32_t* vout = vals_tmp;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    int bits      = bit_hi - shift < 8 ? bit_hi - shift : 8;
    uint32_t mask = (1u << bits) - 1u;
    hipLaunchKernelGGL(k_rs_hist, nblocks, RS_THREADS, 0, h.stream, (uint64_t const*)kin, n, shift, mask,
                       hist.data(), nblocks);
    exclusive_scan_u32(h, hist.data(), hist.data(), (int64_t)nblocks * RS_BINS);
    hipLaunchKernelGGL(k_rs_scatter, nblocks, RS_THREADS, 0, h.stream, (uint64_t const*)kin,
                       (uint32_t const*)vin, kout, vout, n, shift, mask, (uint32_t const*)hist.data(), nblocks);
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  if (kin != keys) {
    HIP_TRY(hipMemcpyAsync(keys, kin, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, h.stream));
    if (vals) HIP_TRY(hipMemcpyAsync(vals, vin, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, h.stream));
  }
  h.sync();  // `hist` is freed on return
}

// ONE stable 8-bit pass, stream-ordered: no synchronisation, no copy back -- (keys_out, vals_out) hold the result; `hist` is the caller's
// scratch of radix_pass_scratch(n) words (it must outlive the launches).  For callers inside a latency-sensitive loop (the SSSP hub round).
size_t radix_pass_scratch(int64_t n) { return (size_t)((n + RS_TILE - 1) / RS_TILE) * RS_BINS + 1; }
void radix_pass_u64_u32(handle_t const& h, uint64_t const* keys_in, uint32_t const* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift, int bits,
                        uint32_t* hist)
{
  if (n <= 0) return;
  int const nblocks   = (int)((n + RS_TILE - 1) / RS_TILE);
  uint32_t const mask = (1u << bits) - 1u;
  hipLaunchKernelGGL(k_rs_hist, nblocks, RS_THREADS, 0, h.stream, keys_in, n, shift, mask, hist, nblocks);
  exclusive_scan_u32(h, hist, hist, (int64_t)nblocks * RS_BINS);
  hipLaunchKernelGGL(k_rs_scatter, nblocks, RS_THREADS, 0, h.stream, keys_in, vals_in, keys_out, vals_out, n, shift, mask, (uint32_t const*)hist, nblocks);
}

void gather_b32(handle_t const& h, uint32_t const* src, uint32_t const* idx, uint32_t* out, int64_t n)
{
  if (n > 0) hipLaunchKernelGGL(k_gather<uint32_t>, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, src, idx, out, n);
}
void gather_b64(handle_t const& h, uint64_t const* src, uint32_t const* idx, uint64_t* out, int64_t n)
{
  if (n > 0) hipLaunchKernelGGL(k_gather<uint64_t>, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, src, idx, out, n);
}

namespace {
template <typename T>
__global__ void k_count_negative_fp(T const* v, int64_t n, unsigned long long* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  unsigned c = 0;
  for (; i < n; i += stride) c += v[i] < T(0);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}
template <typename T>
int64_t count_negative_fp(handle_t const& h, T const* v, int64_t n)
{
  if (n <= 0) return 0;
  dvec<unsigned long long> c(1);
  HIP_TRY(hipMemsetAsync(c.data(), 0, 8, h.stream));
  hipLaunchKernelGGL(k_count_negative_fp<T>, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, v, n, c.data());
  unsigned long long r = 0;
  h.read_back(&r, c.data(), 1);
  return (int64_t)r;
}
}  // namespace
int64_t count_negative_f32(handle_t const& h, float const* v, int64_t n) { return count_negative_fp<float>(h, v, n); }
int64_t count_negative_f64(handle_t const& h, double const* v, int64_t n) { return count_negative_fp<double>(h, v, n); }

}  // namespace cga

// ---- the two primitives the host layer's multi-GPU plan construction is built from (cugraph_amd/mg.py): the library's own
// stable LSD radix sort and its exclusive scan behind the C ABI, so that no rocPRIM / Thrust kernel (torch.sort, torch.unique,
// torch.argsort, torch.cumsum) sits on the product's path.  Device pointers; blocking at return.
extern "C" cugraph_error_code_t cugraph_amd_sort_pairs_u64_u32(const cugraph_resource_handle_t* handle, uint64_t* keys, uint32_t* vals, size_t n, int bit_lo,
                                                              int bit_hi, cugraph_error_t** error)
{
  return cga::guarded(error, [&] {
    cga::handle_t const& h = cga::H(handle);
    CGA_EXPECTS(keys != nullptr || n == 0, CUGRAPH_INVALID_INPUT, "sort_pairs: keys is NULL");
    CGA_EXPECTS(n < ((size_t)1 << 32) && bit_lo >= 0 && bit_hi <= 64 && bit_lo <= bit_hi, CUGRAPH_INVALID_INPUT, "sort_pairs: at most 2^32 - 1 pairs, bits in [0, 64]");
    if (n <= 1 || bit_hi == bit_lo) return;
    HIP_TRY(hipSetDevice(h.device));
    cga::dvec<uint64_t> kt(n);
    cga::dvec<uint32_t> vt(vals ? n : 0);
    cga::radix_sort_u64_u32(h, keys, vals, kt.data(), vals ? vt.data() : nullptr, (int64_t)n, bit_lo, bit_hi);
    h.sync();
  });
}
extern "C" cugraph_error_code_t cugraph_amd_exclusive_scan_u32(const cugraph_resource_handle_t* handle, const uint32_t* in, uint32_t* out, size_t n,
                                                              cugraph_error_t** error)
{
  return cga::guarded(error, [&] {
    cga::handle_t const& h = cga::H(handle);
    CGA_EXPECTS((in != nullptr && out != nullptr) || n == 0, CUGRAPH_INVALID_INPUT, "exclusive_scan: in / out is NULL");
    HIP_TRY(hipSetDevice(h.device));
    cga::exclusive_scan_u32(h, in, out, (int64_t)n);
    h.sync();
  });
}
