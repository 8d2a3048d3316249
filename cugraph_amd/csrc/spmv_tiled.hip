// Column-tiled two-phase pull-SpMV (see spmv_tiled.hpp for the design and the reference call sites it replaces).
#include "spmv_tiled.hpp"
#include "wave_ops.hpp"

#include <algorithm>
#include <type_traits>
#include <utility>

namespace cga {

namespace {

// =================================================================================================
// build: CSC -> tiles
// =================================================================================================
// Sort key of every edge, one wavefront per CSC row: source tile << 47 | destination row << 16 | tile-local source.  Only the tile
// bits are sorted on (stable: the CSC order -- destination, then source -- survives inside a tile); everything k_tile_emit needs
// travels in the key, so it gathers nothing (it used to chase positions -> rows / indices -> xcol: four random reads per edge).
constexpr int TK_TILE_SHIFT = 47, TK_ROW_SHIFT = 16;
// The walk is over the STORED rows of the orientation (rows_view_t, common.hpp): plain CSC, or the CSC + DCSC hybrid of the 2-D multi-GPU block.
__global__ void k_tile_keys(rows_view_t rv, int32_t const* indices, int32_t const* xcol, uint32_t T, uint64_t* keys, uint32_t* vals)
{
  int64_t const wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int const lane = threadIdx.x & 63;
  for (int64_t k = wave; k < rv.n_stored; k += nwaves) {
    uint32_t const b = (uint32_t)rv.offsets[k], e = (uint32_t)rv.offsets[k + 1];  // (edge positions are unsigned 32-bit words: graphs of 2^31 edges and more)
    int32_t const v  = rv.row_of(k);
    for (uint64_t p = (uint64_t)b + lane; p < e; p += 64) {
      uint32_t const c = xcol ? (uint32_t)xcol[indices[p]] : (uint32_t)indices[p];
      uint32_t const J = c / T;
      keys[p] = ((uint64_t)J << TK_TILE_SHIFT) | ((uint64_t)(uint32_t)v << TK_ROW_SHIFT) | (uint64_t)(c - J * T);
      if (vals) vals[p] = (uint32_t)p;
    }
  }
}

__global__ void k_mark_sources(int32_t const* indices, int64_t ne, uint32_t* live)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < ne; i += stride) live[indices[i]] = 1u;  // same value from every writer
}
__global__ void k_xcol(uint32_t const* live, uint32_t const* rank, int64_t nv, int32_t* xcol)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nv; i += stride) xcol[i] = live[i] ? (int32_t)rank[i] : -1;
}

__global__ void k_gather_u32(uint32_t const* src, uint32_t const* idx, int64_t n, uint32_t* out)
{
  int64_t const i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

// first[key] = first position holding that key in a sorted key array (entries of absent keys keep the pre-filled value)
__global__ void k_first_of_key(uint64_t const* keys, int64_t n, int shift, uint32_t* first)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    if (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) first[keys[i] >> shift] = (uint32_t)i;
}

// tiled position k (sorted by source tile, then CSC order) -> padded position, 16-bit source, run-start flag
template <typename WB>
__global__ void k_tile_emit(uint64_t const* keys, uint32_t const* vals, int64_t ne, WB const* w_in, uint32_t const* tile_off, uint32_t const* tile_off_pad,
                            uint16_t* src16, WB* w_out, uint32_t* bits, uint32_t* flag32, uint32_t* dsts)
{
  int64_t k      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; k < ne; k += stride) {
    uint64_t const key = keys[k];
    uint32_t const J   = (uint32_t)(key >> TK_TILE_SHIFT);
    uint32_t const d   = (uint32_t)(key >> TK_ROW_SHIFT) & 0x7FFFFFFFu;
    uint32_t const pk  = tile_off_pad[J] + ((uint32_t)k - tile_off[J]);
    src16[pk]          = (uint16_t)(key & 0xFFFFu);
    if (w_in) w_out[pk] = w_in[vals[k]];
    bool const flag = (uint32_t)k == tile_off[J] || ((uint32_t)(keys[k - 1] >> TK_ROW_SHIFT) & 0x7FFFFFFFu) != d;
    flag32[k] = flag ? 1u : 0u;
    dsts[k]   = d;
    if (flag) atomicOr(&bits[pk >> 5], 1u << (pk & 31));
  }
}

__global__ void k_run_dst(uint32_t const* flag32, uint32_t const* ord, uint32_t const* dsts, int64_t ne, uint32_t* run_dst, uint32_t* cnt_dst)
{
  int64_t k      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; k < ne; k += stride)
    if (flag32[k]) {
      uint32_t d      = dsts[k];
      run_dst[ord[k]] = d;
      atomicAdd(&cnt_dst[d], 1u);
    }
}

// cut[q] = smallest row d with cost(d) >= q * total / nq, cost(d) = 6 * (runs of rows < d) + 16 * d
__global__ void k_cost_cuts(uint32_t const* csum, int64_t nv, uint64_t total, int nq, uint32_t* cut)
{
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint64_t target = total * (uint64_t)q / (uint64_t)nq;  // total < 2^36, q < 2^12
  int64_t lo = 0, hi = nv;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    uint64_t c  = 6ull * csum[mid] + 16ull * (uint64_t)mid;
    if (c < target) lo = mid + 1; else hi = mid;
  }
  cut[q] = (uint32_t)lo;
}

// out[0] = smallest d in [0, n] with csum[d] == total (csum non-decreasing, csum[n] == total): rows >= d have no run
__global__ void k_first_rowless(uint32_t const* csum, int64_t n, uint32_t total, uint32_t* out)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (csum[mid] < total) lo = mid + 1; else hi = mid;
  }
  out[0] = (uint32_t)lo;
}

__device__ __forceinline__ uint32_t tile_of_row(uint32_t const* tile_row0, int nI, uint32_t d)
{  // largest I with tile_row0[I] <= d
  int lo = 0, hi = nI;  // invariant: tile_row0[lo] <= d < tile_row0[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (tile_row0[mid] <= d) lo = mid; else hi = mid;
  }
  return (uint32_t)lo;
}

__global__ void k_run_keys(uint32_t const* run_dst, int64_t n_runs, uint32_t const* tile_row0, int nI, uint64_t* keys, uint32_t* vals)
{
  int64_t q      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; q < n_runs; q += stride) {
    keys[q] = tile_of_row(tile_row0, nI, run_dst[q]);
    vals[q] = (uint32_t)q;
  }
}

// run starts inside every work item (cost estimate of the phase-1 schedule)
__global__ void k_item_runs(int32_t const* item_tile, uint32_t const* item_end, int n_items, uint32_t const* tile_off, uint32_t const* tile_off_pad,
                            uint32_t const* ord, uint32_t* runs)
{
  int const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  uint32_t const J = (uint32_t)item_tile[i];
  uint64_t const s = (uint64_t)i * TP_ITEM, e = min((uint64_t)item_end[i], s + TP_ITEM);
  uint32_t r = 0;
  if (e > s) {
    uint32_t const ks = tile_off[J] + ((uint32_t)s - tile_off_pad[J]);
    r = ord[ks + (uint32_t)(e - s)] - ord[ks];
  }
  runs[i] = r;
}

__global__ void k_wave_desc(int32_t const* item_tile, uint32_t const* item_end, int n_items,
                            uint32_t const* tile_off, uint32_t const* tile_off_pad, uint32_t const* flag32, uint32_t const* ord,
                            uint32_t const* run_dst, uint32_t const* tile_row0, int nI, int64_t n_runs, tiled_wave_t* waves,
                            uint32_t* call, uint64_t* keys, uint32_t* vals)
{
  int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_items * TP_WAVES) return;
  int item = (int)(idx / TP_WAVES), w = (int)(idx % TP_WAVES);
  uint32_t J   = (uint32_t)item_tile[item];
  uint32_t end = item_end[item];
  uint64_t es64 = (uint64_t)item * TP_ITEM + (uint64_t)w * TP_WLEN;
  tiled_wave_t d{0, 0, 0, 0};
  uint32_t c_all = 0;
  uint64_t key = (uint64_t)nI;  // no head: dummy region
  if (es64 < end) {
    uint32_t es = (uint32_t)es64;
    d.es        = es;
    d.ee        = (uint32_t)min((uint64_t)end, es64 + TP_WLEN);
    uint32_t k  = tile_off[J] + (es - tile_off_pad[J]);
    d.rank      = ord[k];
    c_all       = ord[k + (d.ee - es)] - d.rank;  // run starts inside [es, ee)
    if (!flag32[k]) key = tile_of_row(tile_row0, nI, run_dst[d.rank - 1]);  // the run open at es is run (rank - 1)
  }
  waves[idx]          = d;
  call[idx]           = c_all;
  keys[n_runs + idx] = key;
  vals[n_runs + idx] = (uint32_t)(n_runs + idx);
}

// block starts: run q opens a new slot block when its slot does not continue the previous run's (rpos: run q at [q + 1])
__global__ void k_block_flags(uint32_t const* rpos, int64_t n_runs, uint32_t* flag)
{
  int64_t q      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; q < n_runs; q += stride) flag[q] = (q == 0 || rpos[q + 1] != rpos[q] + 1u) ? 1u : 0u;
}

__global__ void k_block_delta(uint32_t const* rpos, uint32_t const* flag, uint32_t const* bidx, int64_t n_runs, uint32_t* delta1)
{
  int64_t q      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; q < n_runs; q += stride)
    if (flag[q]) delta1[bidx[q] + 1] = rpos[q + 1] - (uint32_t)q;  // modulo 2^32
}

__global__ void k_pack_flags(uint32_t const* flag, int64_t n_runs, uint32_t* gbits, int64_t n_words)
{
  int64_t w      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; w < n_words; w += stride) {
    uint32_t word = 0;
    for (int b = 0; b < 32; ++b) {
      int64_t const q = 32 * w + b;
      if (q < n_runs && flag[q]) word |= 1u << b;
    }
    gbits[w] = word;
  }
}

// the per-wavefront records phase 1 reads (layout: spmv_tiled.hpp)
__global__ void k_wave_records(tiled_wave_t const* waves, uint32_t const* call, uint32_t const* rpos, uint32_t const* bidx, uint32_t const* gbits,
                               int64_t n_waves, uint32_t* rec)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_waves * TP_REC_DWORDS; i += stride) {
    int64_t const w     = i / TP_REC_DWORDS;
    int const L         = (int)(i % TP_REC_DWORDS);
    tiled_wave_t const d = waves[w];
    uint32_t const c    = call[w];
    uint32_t v          = 0;
    if (L < 32) {
      uint32_t const t0 = 32u * (uint32_t)L;
      if (t0 < c) {
        uint64_t const r0 = (uint64_t)d.rank + t0;
        uint64_t const two = ((uint64_t)gbits[(r0 >> 5) + 1] << 32) | gbits[r0 >> 5];
        v = (uint32_t)(two >> (r0 & 31));
        if (c - t0 < 32u) v &= (1u << (c - t0)) - 1u;
      }
    } else if (L == TP_REC_NVAL) v = d.ee - d.es;
    else if (L == TP_REC_RANK) v = d.rank;
    else if (L == TP_REC_HEAD) v = d.head_slot;
    else if (L == TP_REC_TAIL) v = c ? rpos[d.rank + c] : d.head_slot;
    else if (L == TP_REC_BLK) v = bidx[d.rank];
    rec[i] = v;
  }
}

__global__ void k_assign_slots(uint64_t const* keys, uint32_t const* vals, int64_t n, int64_t n_runs, uint32_t const* shift,
                               uint32_t const* run_dst, uint32_t const* tile_row0, int nI, uint32_t* rpos, tiled_wave_t* waves,
                               uint16_t* dstl16)
{
  int64_t s      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; s < n; s += stride) {
    uint32_t I    = (uint32_t)keys[s];
    uint32_t idx  = vals[s];
    uint32_t slot = (uint32_t)s + shift[I];  // modulo 2^32
    if ((int64_t)idx < n_runs) {
      rpos[idx + 1] = slot;
      dstl16[slot] = (uint16_t)(run_dst[idx] - tile_row0[I]);
    } else {
      tiled_wave_t& wd = waves[idx - n_runs];
      wd.head_slot     = slot;
      if ((int)I < nI) dstl16[slot] = (uint16_t)(run_dst[wd.rank - 1] - tile_row0[I]);
    }
  }
}

// (Slot blocks aligned to whole cache lines -- every (destination tile, source tile) block of the partial buffer padded to 16 / 32 / 64 slots -- were built in
// round 5: identical bits, phase 1 +4 %, phase 2 +23 ... +144 % for the padding; removed in round 6, numbers in profiles/r5e_block_align_s26.txt.)

// 8 tile-local destinations (< 4096 each) -> 3 dwords; slot groups of 8 never straddle a region (regions are multiples of 8 slots)
__global__ void k_pack_dstl12(uint16_t const* d16, int64_t n_groups, uint32_t* out)
{
  int64_t g      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; g < n_groups; g += stride) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = d16[8 * g + k] & 0xFFFu;
    out[3 * g]     = v[0] | (v[1] << 12) | (v[2] << 24);
    out[3 * g + 1] = (v[2] >> 8) | (v[3] << 4) | (v[4] << 16) | (v[5] << 28);
    out[3 * g + 2] = (v[5] >> 4) | (v[6] << 8) | (v[7] << 20);
  }
}

int bits_for_u(uint64_t max_value)
{
  int b = 0;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b < 1 ? 1 : b;
}

template <typename T>
std::vector<T> to_host(handle_t const& h, T const* dev, size_t n)
{
  std::vector<T> out(n);
  if (n) HIP_TRY(hipMemcpyAsync(out.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost, h.stream));
  h.sync();
  return out;
}
template <typename T>
void to_device(handle_t const& h, dvec<T>& dev, std::vector<T> const& host)
{
  dev.resize_discard(host.size() ? host.size() : 1);
  if (host.size()) HIP_TRY(hipMemcpyAsync(dev.data(), host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, h.stream));
  h.sync();  // `host` may be a temporary
}

// first-occurrence table of a sorted key array over keys [0, nkeys]; absent keys take the next present key's position
std::vector<uint32_t> key_starts(handle_t const& h, uint64_t const* keys, int64_t n, int64_t nkeys, int shift = 0)
{
  dvec<uint32_t> first(nkeys + 1);
  fill_u32(h, first.data(), nkeys + 1, 0xFFFFFFFFu);
  if (n > 0) hipLaunchKernelGGL(k_first_of_key, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, keys, n, shift, first.data());
  std::vector<uint32_t> f = to_host(h, first.data(), (size_t)nkeys + 1);
  f[nkeys] = (uint32_t)n;
  for (int64_t k = nkeys - 1; k >= 0; --k)
    if (f[k] == 0xFFFFFFFFu) f[k] = f[k + 1];
  return f;
}

}  // namespace

namespace {
// wmax[I] = max over the rows of destination tile I of sum |w| of the row's in-edges (w == nullptr: the in-degree); one workgroup per tile,
// one wavefront per row at a time
template <typename WB>
__global__ void __launch_bounds__(256) k_tile_wmax(rows_view_t rv, WB const* w, uint32_t const* tile_row0, double* wmax)
{
  __shared__ double red[4];
  int const I = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the tile's rows as a range of STORED rows (a hypersparse orientation stores only the rows that have an edge; the others add nothing to a maximum)
  uint32_t const r0 = (uint32_t)rv.lower_bound(tile_row0[I]), r1 = (uint32_t)rv.lower_bound(tile_row0[I + 1]);
  int32_t const* const offsets = rv.offsets;
  double best = 0.0;
  if (w == nullptr) {
    for (uint32_t r = r0 + threadIdx.x; r < r1; r += 256) best = fmax(best, (double)((uint32_t)offsets[r + 1] - (uint32_t)offsets[r]));
  } else {
    for (uint32_t r = r0 + wave; r < r1; r += 4) {
      double s = 0.0;
      for (uint64_t p = (uint64_t)(uint32_t)offsets[r] + lane; p < (uint32_t)offsets[r + 1]; p += 64) s += fabs((double)w[p]);
      s    = group_sum(s, 64);
      best = fmax(best, s);
    }
  }
  for (int o = 32; o > 0; o >>= 1) best = fmax(best, __shfl_xor(best, o));
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) wmax[I] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
}  // namespace

int tiled_default_T(handle_t const& h, size_t vsize, int64_t nv)
{
  // LDS = tile (T values) + one TP_SUB-entry staging row per wavefront + a few static words
  int64_t const lds_max = ((int64_t)h.lds_per_block / TP_WG_PER_CU - 1536) / (int64_t)vsize - (int64_t)TP_WAVES * TP_STAGE;
  int64_t T = h.pagerank_hot_tile > 0 ? h.pagerank_hot_tile : lds_max;
  T = std::min<int64_t>({T, lds_max, 65536});
  T = std::min<int64_t>(T, (std::max<int64_t>(nv, 1) + 255) / 256 * 256);  // small graphs: one small tile
  T = std::max<int64_t>(T / 256 * 256, 256);
  return (int)T;
}

// max over rows of sum |w|
template <typename WB>
__global__ void k_row_abs_max(int32_t const* offsets, WB const* w, int64_t nv, unsigned long long* out)  // (nv = stored rows)
{
  int64_t wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int lane       = threadIdx.x & 63;
  double best    = 0;
  for (int64_t v = wave; v < nv; v += nwaves) {
    double s = 0;
    for (uint64_t p = (uint64_t)(uint32_t)offsets[v] + lane; p < (uint32_t)offsets[v + 1]; p += 64) s += fabs((double)w[p]);
    s    = group_sum(s, 64);
    best = fmax(best, s);
  }
  if (lane == 0) atomicMax(out, (unsigned long long)__double_as_longlong(best));  // non-negative doubles order like their bit patterns
}

void build_tiled_csc(handle_t const& h, int64_t nv, int64_t n_dst, int64_t ne, orientation_t const& csc, bool has_weights, size_t vsize, int T,
                     tiled_csc_t& t, bool compact_columns, uint32_t const* live_hint)
{
  // nv = number of column (source) ids and of CSC rows; n_dst <= nv = rows that can have in-edges and get an epilogue
  // (single GPU: n_dst = nv; multi-GPU: the local rows, while the columns span the whole graph)
  size_t const wsize = has_weights ? vsize : 0;
  rows_view_t const rows = rows_view(csc, nv);  // plain CSC, or its hypersparse form (walked natively: no offsets array is rebuilt)
  CGA_EXPECTS(nv < ((int64_t)1 << 31) && ne <= kMaxGraphEdges - 2 * (int64_t)TP_ITEM * 1024, CUGRAPH_UNKNOWN_ERROR, "tiled SpMV: graph too large for 32-bit positions");
  t       = tiled_csc_t{};
  build_trace tr(h, "tiled");
  dvec<uint32_t> live_rank;  // compact columns: live_rank[r] = number of live sources with an id < r
  t.T     = T;
  t.nv    = n_dst;
  t.ne    = ne;
  t.ncols = nv;
  if (compact_columns && ne > 0) {  // columns = the sources that occur, in id order
    CGA_EXPECTS(nv == n_dst, CUGRAPH_UNKNOWN_ERROR, "tiled SpMV: compact columns need a square local graph");
    live_rank.resize_discard((size_t)nv + 1);
    dvec<uint32_t> live_own(live_hint ? 1 : (size_t)nv + 1);
    dvec<uint32_t>& rank = live_rank;
    if (!live_hint) {  // mark the sources that occur: one random 4-byte store per edge (16 ms at RMAT-26)
      HIP_TRY(hipMemsetAsync(live_own.data(), 0, ((size_t)nv + 1) * sizeof(uint32_t), h.stream));
      hipLaunchKernelGGL(k_mark_sources, grid_for(ne, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)csc.indices.data(), ne, live_own.data());
    }
    uint32_t const* const live_p = live_hint ? live_hint : live_own.data();
    exclusive_scan_u32(h, live_p, rank.data(), nv + 1);
    uint32_t nc = 0;
    h.read_back(&nc, rank.data() + nv, 1);
    t.ncols = nc;
    t.xcol.resize_discard((size_t)nv);
    hipLaunchKernelGGL(k_xcol, grid_for(nv, kBlock, 8192), kBlock, 0, h.stream, live_p, (uint32_t const*)rank.data(), nv, t.xcol.data());
    h.sync();
  }
  int32_t const* xcol = t.xcol.size() ? t.xcol.data() : nullptr;
  t.nJ    = (int)std::max<int64_t>(1, (t.ncols + T - 1) / T);
  int const nJ = t.nJ;

  tr.step("live columns");
  // ---- edges ordered by (source tile, destination, source): stable sort of the CSC positions by source tile
  dvec<uint64_t> keys, keys_tmp;
  dvec<uint32_t> vals, vals_tmp, flag32, ord, dsts;
  std::vector<uint32_t> tile_off(nJ + 1, 0), tile_off_pad(nJ + 1, 0);
  if (ne > 0) {
    CGA_EXPECTS(nJ <= (1 << (64 - TK_TILE_SHIFT)) && T <= (1 << TK_ROW_SHIFT), CUGRAPH_UNKNOWN_ERROR, "tiled SpMV: too many source tiles for the packed sort key");
    keys.resize_discard(ne); keys_tmp.resize_discard(ne);
    if (has_weights) { vals.resize_discard(ne); vals_tmp.resize_discard(ne); }  // the edge position only has to travel when weights follow it
    uint32_t* const vp  = has_weights ? vals.data() : nullptr;
    uint32_t* const vtp = has_weights ? vals_tmp.data() : nullptr;
    hipLaunchKernelGGL(k_tile_keys, grid_for(rows.n_stored * 16, kBlock, 8192), kBlock, 0, h.stream, rows, (int32_t const*)csc.indices.data(), xcol, (uint32_t)T,
                       keys.data(), vp);
    radix_sort_u64_u32(h, keys.data(), vp, keys_tmp.data(), vtp, ne, TK_TILE_SHIFT, TK_TILE_SHIFT + bits_for_u((uint64_t)nJ - 1));
    keys_tmp = dvec<uint64_t>(); vals_tmp = dvec<uint32_t>();
    tile_off = key_starts(h, keys.data(), ne, nJ, TK_TILE_SHIFT);
  }
  tr.step("sort by source tile");
  {  // every tile starts on a work-item boundary: item i covers padded positions [i * TP_ITEM, (i + 1) * TP_ITEM)
    uint64_t pos = 0;
    for (int J = 0; J < nJ; ++J) {
      pos += ((uint64_t)(tile_off[J + 1] - tile_off[J]) + TP_ITEM - 1) / TP_ITEM * TP_ITEM;
      CGA_EXPECTS(pos + TP_ITEM < ((uint64_t)1 << 32), CUGRAPH_UNKNOWN_ERROR, "tiled SpMV: padded edge count overflows 32 bits");
      tile_off_pad[J + 1] = (uint32_t)pos;
    }
  }
  t.ne_pad = tile_off_pad[nJ];
  size_t const epad = (size_t)t.ne_pad + TP_WLEN + 64;  // tail loads of the last wavefront stay in bounds
  t.src16.resize_discard(epad);
  HIP_TRY(hipMemsetAsync(t.src16.data(), 0, epad * sizeof(uint16_t), h.stream));
  t.bits.resize_discard(epad / 32 + 2);
  HIP_TRY(hipMemsetAsync(t.bits.data(), 0, (epad / 32 + 2) * sizeof(uint32_t), h.stream));
  if (has_weights) {
    t.weights.alloc(epad * wsize);
    HIP_TRY(hipMemsetAsync(t.weights.ptr, 0, epad * wsize, h.stream));
  }
  dvec<uint32_t> d_tile_off, d_tile_off_pad;
  to_device(h, d_tile_off, tile_off);
  to_device(h, d_tile_off_pad, tile_off_pad);

  tr.step("edge arrays");
  // ---- runs
  dvec<uint32_t> run_dst, cnt_dst(n_dst + 1);
  HIP_TRY(hipMemsetAsync(cnt_dst.data(), 0, (n_dst + 1) * sizeof(uint32_t), h.stream));
  if (ne > 0) {
    flag32.resize_discard(ne + 1); ord.resize_discard(ne + 1); dsts.resize_discard(ne);
    HIP_TRY(hipMemsetAsync(flag32.data() + ne, 0, sizeof(uint32_t), h.stream));
    int const g = grid_for(ne, kBlock, 8192);
    uint32_t const* const vk = has_weights ? vals.data() : nullptr;
    if (!has_weights)
      hipLaunchKernelGGL(k_tile_emit<uint32_t>, g, kBlock, 0, h.stream, (uint64_t const*)keys.data(), vk, ne, (uint32_t const*)nullptr,
                         (uint32_t const*)d_tile_off.data(), (uint32_t const*)d_tile_off_pad.data(), t.src16.data(), (uint32_t*)nullptr, t.bits.data(),
                         flag32.data(), dsts.data());
    else if (wsize == 4)
      hipLaunchKernelGGL(k_tile_emit<uint32_t>, g, kBlock, 0, h.stream, (uint64_t const*)keys.data(), vk, ne, csc.weights.as<uint32_t const>(),
                         (uint32_t const*)d_tile_off.data(), (uint32_t const*)d_tile_off_pad.data(), t.src16.data(), t.weights.as<uint32_t>(), t.bits.data(),
                         flag32.data(), dsts.data());
    else
      hipLaunchKernelGGL(k_tile_emit<uint64_t>, g, kBlock, 0, h.stream, (uint64_t const*)keys.data(), vk, ne, csc.weights.as<uint64_t const>(),
                         (uint32_t const*)d_tile_off.data(), (uint32_t const*)d_tile_off_pad.data(), t.src16.data(), t.weights.as<uint64_t>(), t.bits.data(),
                         flag32.data(), dsts.data());
    exclusive_scan_u32(h, flag32.data(), ord.data(), ne + 1);
    uint32_t p32 = 0;
    h.read_back(&p32, ord.data() + ne, 1);
    t.n_runs = p32;
    keys = dvec<uint64_t>(); vals = dvec<uint32_t>();
    run_dst.resize_discard(t.n_runs);
    hipLaunchKernelGGL(k_run_dst, g, kBlock, 0, h.stream, (uint32_t const*)flag32.data(), (uint32_t const*)ord.data(), (uint32_t const*)dsts.data(), ne,
                       run_dst.data(), cnt_dst.data());
    h.sync();
    dsts = dvec<uint32_t>();
  }

  tr.step("runs");
  // ---- destination tiles: equal cost (6 bytes per partial + 16 bytes per row), at most TP2_ROWS rows
  std::vector<uint32_t> row0;
  {
    dvec<uint32_t> csum(n_dst + 1);
    exclusive_scan_u32(h, cnt_dst.data(), csum.data(), n_dst + 1);
    uint64_t total = 6ull * (uint64_t)t.n_runs + 16ull * (uint64_t)n_dst;
    int nq = (int)std::min<uint64_t>(4096, std::max<uint64_t>(1, total / 32768));
    dvec<uint32_t> cut(nq);
    hipLaunchKernelGGL(k_cost_cuts, (nq + 255) / 256, 256, 0, h.stream, (uint32_t const*)csum.data(), n_dst, total, nq, cut.data());
    std::vector<uint32_t> c = to_host(h, cut.data(), (size_t)nq);
    c.push_back((uint32_t)n_dst);
    // rows >= n_act have no in-edge (with degree-sorted ids: the zero-in-degree suffix, 60 % of an RMAT-26 graph): a tile
    // boundary is forced there so that a plan may leave those rows out of its per-iteration epilogue (tiled_const_rows)
    dvec<uint32_t> d_nact(1);
    hipLaunchKernelGGL(k_first_rowless, 1, 64, 0, h.stream, (uint32_t const*)csum.data(), n_dst, (uint32_t)t.n_runs, d_nact.data());
    uint32_t nact = 0;
    h.read_back(&nact, d_nact.data(), 1);
    t.n_act = nact;
    c.push_back(nact);
    std::sort(c.begin(), c.end());
    uint32_t prev = 0;
    row0.push_back(0);
    for (uint32_t b : c) {
      if (b <= prev) continue;
      while (b - prev > (uint32_t)TP2_ROWS) { prev += TP2_ROWS; row0.push_back(prev); }
      row0.push_back(b);
      prev = b;
    }
    if (row0.size() == 1) row0.push_back((uint32_t)n_dst);  // n_dst == 0
    t.nI_act = (int)(std::lower_bound(row0.begin(), row0.end(), nact) - row0.begin());  // tiles [0, nI_act) cover rows [0, n_act)
  }
  t.nI = (int)row0.size() - 1;
  to_device(h, t.tile_row0, row0);
  t.c0 = t.n_act;  // first column of a row >= n_act (identity columns: the row itself)
  if (live_rank.size()) {
    uint32_t c0 = 0;
    h.read_back(&c0, live_rank.data() + t.n_act, 1);
    t.c0 = c0;
    live_rank = dvec<uint32_t>();
  }
  t.tile_wmax.resize_discard((size_t)std::max(t.nI, 1));
  HIP_TRY(hipMemsetAsync(t.tile_wmax.data(), 0, (size_t)std::max(t.nI, 1) * sizeof(double), h.stream));
  if (ne > 0 && t.nI > 0) {
    if (!has_weights) hipLaunchKernelGGL(k_tile_wmax<uint32_t>, t.nI, 256, 0, h.stream, rows, (uint32_t const*)nullptr, (uint32_t const*)t.tile_row0.data(), t.tile_wmax.data());
    else if (wsize == 4) hipLaunchKernelGGL(k_tile_wmax<float>, t.nI, 256, 0, h.stream, rows, csc.weights.as<float const>(), (uint32_t const*)t.tile_row0.data(), t.tile_wmax.data());
    else hipLaunchKernelGGL(k_tile_wmax<double>, t.nI, 256, 0, h.stream, rows, csc.weights.as<double const>(), (uint32_t const*)t.tile_row0.data(), t.tile_wmax.data());
  }

  tr.step("destination tiles");
  // ---- phase-1 work items (source tile order = hottest tiles first)
  std::vector<int32_t> item_tile;
  std::vector<uint32_t> item_end;  // end of the real edges of the item's tile (padded position)
  for (int J = 0; J < nJ; ++J) {
    uint32_t const e = tile_off_pad[J] + (tile_off[J + 1] - tile_off[J]);
    for (uint32_t s = tile_off_pad[J]; s < tile_off_pad[J + 1]; s += TP_ITEM) {
      item_tile.push_back(J);
      item_end.push_back(e);
    }
  }
  t.n_items = (int)item_tile.size();
  int64_t const n_waves = (int64_t)t.n_items * TP_WAVES;
  to_device(h, t.item_tile, item_tile);
  dvec<tiled_wave_t> waves(n_waves > 0 ? n_waves : 1);
  dvec<uint32_t> call(n_waves > 0 ? n_waves : 1);  // run starts inside each wavefront's range
  t.wrec.resize_discard((size_t)(n_waves > 0 ? n_waves : 1) * TP_REC_DWORDS);
  HIP_TRY(hipMemsetAsync(t.wrec.data(), 0, (size_t)(n_waves > 0 ? n_waves : 1) * TP_REC_DWORDS * sizeof(uint32_t), h.stream));

  tr.step("work items");
  // ---- phase-1 chunks: up to TP_CHUNK consecutive items of one source tile, handed out dynamically LARGEST FIRST
  // (long chunks keep an LDS tile for many items; the one-item chunks of the cold tiles fill the tail evenly)
  {
    size_t const lds = ((size_t)T + (size_t)TP_WAVES * TP_STAGE) * vsize;
    int const per_cu = std::max<int>(1, std::min<int>(2, (int)((h.lds_per_block - 1024) / (lds + 64))));
    int const max_wg = h.num_cus * per_cu;
    char const* env_chunk = getenv("CUGRAPH_AMD_TP_CHUNK");  // tests: multi-item chunks on small graphs
    int const chunk  = env_chunk ? std::max(1, atoi(env_chunk)) : std::max(1, std::min<int>(TP_CHUNK, t.n_items / (max_wg * 4)));  // small graphs: more, shorter chunks
    // the hottest tiles span thousands of items: their first items go out in LONG chunks (a workgroup reloads the 126 KiB
    // tile once per chunk, with nothing else in flight), the rest in `chunk`-item pieces that balance the tail
    int const big        = std::max(chunk, TP_CHUNK_BIG);
    double const frac    = 0.55;
    int64_t big_budget   = t.n_items / (max_wg * 4) >= TP_CHUNK ? (int64_t)(frac * t.n_items) : 0;
    // the items of the coldest tiles (the last `tail_frac` of all items; an item there costs about twice a hot one: ten times the
    // runs and partial stores) go out in short chunks, so the workgroups finish within one short chunk of each other
    double const tail_frac = TP_TAIL_FRAC;
    int tail_chunk         = std::max(1, std::min(chunk, TP_TAIL_CHUNK));
    int const tail_first   = t.n_items / (max_wg * 4) >= TP_CHUNK ? (int)((1.0 - tail_frac) * t.n_items) : t.n_items;
    // STATIC PREFIX (sticky tiles): every workgroup first walks a private, contiguous range of work items -- equal shares, by
    // estimated cost, of the first `static_frac` of the total cost -- so that it loads a hot source tile ONCE (a 126 KiB tile
    // load drains the workgroup's memory pipeline: with every chunk drawn from one global queue a workgroup reloaded tile 0
    // every 32 items, 13.7 tile loads per workgroup and 0.44 GB of x per iteration at RMAT-26); the cold remainder -- hundreds
    // of tiles of a few items each, one tile load per chunk whoever takes it -- stays dynamic and evens out the finish.
    // Cost of an item = its edges + TP_RUN_COST x its run starts (a cold item stores ten times the partials of a hot one and
    // takes about twice as long).
    char const* env_sf = getenv("CUGRAPH_AMD_TP_STATIC_FRAC");
    // Default by the number of work items per workgroup (measured): RMAT-26 (257 items per workgroup):
    // the x-tile reloads are a small share and the dynamic queue balances better -- off (0.85: +8 %, 0.5: neutral); RMAT-24 (64):
    // 0.7 -> phase 1 -17 %; RMAT-22 (16): 0.85-0.95 with 2-item dynamic chunks -> -13 %.
    double const items_per_wg = (double)t.n_items / (double)max_wg;
    double const static_frac  = env_sf ? atof(env_sf) : (items_per_wg >= 160.0 ? 0.0 : items_per_wg >= 24.0 ? 0.7 : 0.9);
    double const run_cost    = 1.3;
    std::vector<int32_t> cb;              // [4 * n_chunks]: (unused, first item, end item, source tile); static chunks first, grouped by workgroup
    std::vector<int32_t> wg_static;       // [2 * n_wg]: (first static chunk, end static chunk) of every workgroup
    int static_items = 0;
    bool const use_static = static_frac > 0.0 && ne > 0 && (items_per_wg >= 2.0 || getenv("CUGRAPH_AMD_TP_STATIC_FORCE") != nullptr);
    if (use_static && !env_chunk && items_per_wg < 24.0) tail_chunk = std::max(1, tail_chunk / 2);  // few items per workgroup: finer dynamic tail
    t.n_wg = use_static ? max_wg : 0;
    if (use_static) {
      dvec<uint32_t> d_item_end, d_item_runs((size_t)t.n_items);
      to_device(h, d_item_end, item_end);
      hipLaunchKernelGGL(k_item_runs, grid_for(t.n_items, kBlock), kBlock, 0, h.stream, (int32_t const*)t.item_tile.data(), (uint32_t const*)d_item_end.data(), t.n_items,
                         (uint32_t const*)d_tile_off.data(), (uint32_t const*)d_tile_off_pad.data(), (uint32_t const*)ord.data(), d_item_runs.data());
      std::vector<uint32_t> item_runs = to_host(h, d_item_runs.data(), (size_t)t.n_items);
      std::vector<double> cum(t.n_items + 1, 0.0);
      for (int i = 0; i < t.n_items; ++i) {
        double const runs = (double)item_runs[i];
        uint32_t const first = (uint32_t)i * (uint32_t)TP_ITEM;
        double const edges   = (double)std::min<uint32_t>((uint32_t)TP_ITEM, item_end[i] > first ? item_end[i] - first : 0u);
        cum[i + 1] = cum[i] + edges + run_cost * runs;
      }
      double const share = static_frac * cum[t.n_items] / max_wg;
      wg_static.assign((size_t)2 * max_wg, 0);
      int i = 0;
      for (int w = 0; w < max_wg; ++w) {
        int j = i;
        while (j < t.n_items && cum[j + 1] <= share * (w + 1)) ++j;
        wg_static[2 * w] = (int32_t)(cb.size() / 4);
        for (int k = i; k < j;) {  // one chunk per source tile of the range
          int e = k + 1;
          while (e < j && item_tile[e] == item_tile[k]) ++e;
          cb.push_back(0); cb.push_back(k); cb.push_back(e); cb.push_back(item_tile[k]);
          k = e;
        }
        wg_static[2 * w + 1] = (int32_t)(cb.size() / 4);
        i = j;
      }
      static_items = i;
    }
    t.n_static_chunks = (int)(cb.size() / 4);
    std::vector<std::pair<int32_t, int32_t>> ch;  // dynamic part: (first item, items)
    for (int i = static_items; i < t.n_items;) {
      int j = i + 1;
      while (j < t.n_items && item_tile[j] == item_tile[i]) ++j;  // [i, j) = the (remaining) items of one tile
      int k = i;
      while (!use_static && big > chunk && big_budget >= big && j - k >= 2 * big) { ch.push_back({k, big}); k += big; big_budget -= big; }
      while (k < j) { int const n = std::min((use_static || k >= tail_first) ? tail_chunk : chunk, j - k); ch.push_back({k, n}); k += n; }
      i = j;
    }
    std::stable_sort(ch.begin(), ch.end(), [](auto const& x, auto const& y) { return x.second > y.second; });
    for (auto const& c : ch) { cb.push_back(0); cb.push_back(c.first); cb.push_back(c.first + c.second); cb.push_back(item_tile[c.first]); }
    t.n_chunks = (int)(cb.size() / 4);
    if (cb.empty()) cb.assign(4, 0);
    if (!use_static) t.n_wg = std::max(1, std::min<int>(t.n_chunks, max_wg));
    if (wg_static.empty()) wg_static.assign((size_t)2 * std::max(t.n_wg, 1), 0);
    to_device(h, t.wg_static, wg_static);
    if (getenv("CUGRAPH_AMD_TILED_DEBUG"))
      fprintf(stderr, "[tiled build] chunks: %d static (items [0, %d) of %d, %d workgroups), %d dynamic\n", t.n_static_chunks, static_items, t.n_items, t.n_wg,
              t.n_chunks - t.n_static_chunks);
    to_device(h, t.chunk_begin, cb);
  }

  tr.step("chunks");
  // ---- slots: runs and wave heads ordered by (destination tile, source tile, destination)
  int64_t const n_el = t.n_runs + n_waves;
  std::vector<uint32_t> region_off(t.nI + 2, 0);
  if (n_el > 0) {
    dvec<uint32_t> d_item_end;
    to_device(h, d_item_end, item_end);
    keys.resize_discard(n_el); keys_tmp.resize_discard(n_el); vals.resize_discard(n_el); vals_tmp.resize_discard(n_el);
    if (t.n_runs > 0)
      hipLaunchKernelGGL(k_run_keys, grid_for(t.n_runs, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)run_dst.data(), t.n_runs,
                         (uint32_t const*)t.tile_row0.data(), t.nI, keys.data(), vals.data());
    hipLaunchKernelGGL(k_wave_desc, grid_for(n_waves, kBlock), kBlock, 0, h.stream, (int32_t const*)t.item_tile.data(),
                       (uint32_t const*)d_item_end.data(), t.n_items, (uint32_t const*)d_tile_off.data(), (uint32_t const*)d_tile_off_pad.data(),
                       (uint32_t const*)flag32.data(), (uint32_t const*)ord.data(), (uint32_t const*)run_dst.data(), (uint32_t const*)t.tile_row0.data(), t.nI,
                       t.n_runs, waves.data(), call.data(), keys.data(), vals.data());
    radix_sort_u64_u32(h, keys.data(), vals.data(), keys_tmp.data(), vals_tmp.data(), n_el, 0, bits_for_u((uint64_t)t.nI));
    std::vector<uint32_t> first = key_starts(h, keys.data(), n_el, (int64_t)t.nI + 1);  // regions 0..nI (nI = dummy)
    std::vector<uint32_t> shift(t.nI + 1);
    for (int I = 0; I <= t.nI; ++I) {
      region_off[I + 1] = region_off[I] + ((first[I + 1] - first[I] + 7u) & ~7u);
      shift[I]          = region_off[I] - first[I];
    }
    t.n_slots = region_off[t.nI + 1];
    size_t const spad = (size_t)t.n_slots + 64;
    t.dstl16.resize_discard(spad);
    HIP_TRY(hipMemsetAsync(t.dstl16.data(), 0, spad * sizeof(uint16_t), h.stream));
    // (the edge arrays -- src16, weights -- are addressed from a 64-bit per-wavefront base: only their POSITIONS must fit 32 bits, checked above)
    CGA_EXPECTS(((uint64_t)t.n_runs + 512) * 4 < ((uint64_t)1 << 32) &&
                  ((uint64_t)t.n_slots + 64) * vsize < ((uint64_t)1 << 32) && (uint64_t)(n_waves + 2) * TP_REC_DWORDS * 4 < ((uint64_t)1 << 32),
                CUGRAPH_UNKNOWN_ERROR, "tiled SpMV: an array exceeds the 32-bit byte-offset addressing of phase 1");
    // slot of every run (build-time only): [0] = dummy (the "run" before run 0), run q at [q + 1], zero padding behind
    dvec<uint32_t> rpos((size_t)t.n_runs + 512);
    HIP_TRY(hipMemsetAsync(rpos.data(), 0, ((size_t)t.n_runs + 512) * sizeof(uint32_t), h.stream));
    dvec<uint32_t> d_shift;
    to_device(h, d_shift, shift);
    hipLaunchKernelGGL(k_assign_slots, grid_for(n_el, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), (uint32_t const*)vals.data(), n_el,
                       t.n_runs, (uint32_t const*)d_shift.data(), (uint32_t const*)run_dst.data(), (uint32_t const*)t.tile_row0.data(), t.nI, rpos.data(),
                       waves.data(), t.dstl16.data());
    h.sync();
    keys = dvec<uint64_t>(); keys_tmp = dvec<uint64_t>(); vals = dvec<uint32_t>(); vals_tmp = dvec<uint32_t>();
    // ---- slot blocks: inside one (destination tile, source tile) block the slots follow the run order, so phase 1 needs
    // 4 bytes per block (slot - run index) and one bit per run instead of 4 bytes per run
    dvec<uint32_t> bflag((size_t)t.n_runs + 1), bidx((size_t)t.n_runs + 1);
    HIP_TRY(hipMemsetAsync(bflag.data() + t.n_runs, 0, sizeof(uint32_t), h.stream));
    if (t.n_runs > 0) hipLaunchKernelGGL(k_block_flags, grid_for(t.n_runs, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)rpos.data(), t.n_runs, bflag.data());
    exclusive_scan_u32(h, bflag.data(), bidx.data(), t.n_runs + 1);
    uint32_t nb = 0;
    h.read_back(&nb, bidx.data() + t.n_runs, 1);
    t.n_blocks = nb;
    t.delta1.resize_discard((size_t)nb + 1 + 64 * (TP_NDREG + 1));
    HIP_TRY(hipMemsetAsync(t.delta1.data(), 0, ((size_t)nb + 1 + 64 * (TP_NDREG + 1)) * sizeof(uint32_t), h.stream));
    int64_t const n_words = t.n_runs / 32 + 4;
    dvec<uint32_t> gbits(n_words);
    if (t.n_runs > 0)
      hipLaunchKernelGGL(k_block_delta, grid_for(t.n_runs, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)rpos.data(), (uint32_t const*)bflag.data(),
                         (uint32_t const*)bidx.data(), t.n_runs, t.delta1.data());
    hipLaunchKernelGGL(k_pack_flags, grid_for(n_words, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)bflag.data(), t.n_runs, gbits.data(), n_words);
    hipLaunchKernelGGL(k_wave_records, grid_for(n_waves * TP_REC_DWORDS, kBlock, 8192), kBlock, 0, h.stream, (tiled_wave_t const*)waves.data(),
                       (uint32_t const*)call.data(), (uint32_t const*)rpos.data(), (uint32_t const*)bidx.data(), (uint32_t const*)gbits.data(), n_waves,
                       t.wrec.data());
    h.sync();
    if (TP2_ROWS <= 4096) {
      int64_t const n_groups = (int64_t)(spad + 7) / 8;
      t.dstl12.resize_discard((size_t)n_groups * 3 + 16);
      HIP_TRY(hipMemsetAsync(t.dstl12.data(), 0, ((size_t)n_groups * 3 + 16) * sizeof(uint32_t), h.stream));
      hipLaunchKernelGGL(k_pack_dstl12, grid_for(n_groups - 1, kBlock, 8192), kBlock, 0, h.stream, (uint16_t const*)t.dstl16.data(), n_groups - 1, t.dstl12.data());
      h.sync();
      t.dstl16 = dvec<uint16_t>();
    }
  } else {
    t.n_slots = 0;
    t.dstl16.resize_discard(64);
    t.delta1.resize_discard(1 + 64 * (TP_NDREG + 1));
    HIP_TRY(hipMemsetAsync(t.delta1.data(), 0, (1 + 64 * (TP_NDREG + 1)) * sizeof(uint32_t), h.stream));
  }
  to_device(h, t.region_off, region_off);

  tr.step("slots");
  // ---- bound used by the fixed-point accumulation of phase 2
  t.wmax = (double)csc.max_degree;
  if (has_weights && ne > 0) {
    dvec<unsigned long long> mx(1);
    HIP_TRY(hipMemsetAsync(mx.data(), 0, sizeof(unsigned long long), h.stream));
    int const g = grid_for(nv * 16, kBlock, 8192);
    if (vsize == 4) hipLaunchKernelGGL(k_row_abs_max<float>, g, kBlock, 0, h.stream, rows.offsets, csc.weights.as<float const>(), rows.n_stored, mx.data());
    else            hipLaunchKernelGGL(k_row_abs_max<double>, g, kBlock, 0, h.stream, rows.offsets, csc.weights.as<double const>(), rows.n_stored, mx.data());
    unsigned long long bits = 0;
    h.read_back(&bits, mx.data(), 1);
    std::memcpy(&t.wmax, &bits, sizeof(double));
  }
  t.built = true;
  h.sync();
  if (getenv("CUGRAPH_AMD_TILED_DEBUG"))
    fprintf(stderr, "[tiled build] T %d nJ %d nI %d items %d chunks %d wg %d | ne %lld ne_pad %lld runs %lld slots %lld blocks %lld | rows %lld cols %lld\n", t.T,
            t.nJ, t.nI, t.n_items, t.n_chunks, t.n_wg, (long long)t.ne, (long long)t.ne_pad, (long long)t.n_runs, (long long)t.n_slots,
            (long long)t.n_blocks, (long long)t.nv, (long long)t.ncols);
  if (getenv("CUGRAPH_AMD_TILED_DEBUG")) fprintf(stderr, "[tiled build] rows with in-edges end at %lld (tiles [0, %d)), first column of the rest %lld\n", (long long)t.n_act, t.nI_act, (long long)t.c0);
}

namespace {

// =================================================================================================
// per-iteration scalars: fixed-order fp64 reduction of the per-destination-tile (L1 change, dangling mass, max |x|)
// partials; also picks the fixed-point scale of the next phase 2
// =================================================================================================
template <typename WT>
struct fin_args {
  double const* partials{nullptr};  // [n][3]; nullptr = nothing to finish
  int n{0};
  pr_scalars<WT>* scal{nullptr};
  double* totals{nullptr};  // multi-GPU: this rank's (diff, dangling, xmax) go here instead of into scal
  WT alpha{0};
  int64_t nv_global{0};
  int personalized{0};
  double wmax{0};
  // rows left out of the epilogue (tiled_const_rows): their analytic share; init_prev >= 0: iteration-0 fold, base_prev := init_prev
  double cr_rows{0}, cr_dangling{0}, cr_max_inv_outw{0};
  double init_prev{-1};
};

// executed by ONE workgroup of `nthreads` threads; scratch = 3 * nthreads doubles of LDS
template <typename WT>
__device__ __forceinline__ void finish_scalars(fin_args<WT> const& f, double* scratch, int tid, int nthreads)
{
  double* r0 = scratch;
  double* r1 = scratch + nthreads;
  double* r2 = scratch + 2 * nthreads;
  double d0 = 0, d1 = 0, d2 = 0;
  for (int i = tid; i < f.n; i += nthreads) { d0 += f.partials[3 * i]; d1 += f.partials[3 * i + 1]; d2 = fmax(d2, f.partials[3 * i + 2]); }
  r0[tid] = d0; r1[tid] = d1; r2[tid] = d2;
  __syncthreads();
  for (int s = nthreads >> 1; s > 0; s >>= 1) {
    if (tid < s) { r0[tid] += r0[tid + s]; r1[tid] += r1[tid + s]; r2[tid] = fmax(r2[tid], r2[tid + s]); }
    __syncthreads();
  }
  if (tid == 0) {
    double diff = r0[0], dang = r1[0], xmax = r2[0];
    if (f.cr_rows > 0 && f.init_prev < 0) {  // the rows without in-edges were not visited: pr' = base for each of them
      WT const b = f.scal->base, bp = f.scal->base_prev;
      diff += f.cr_rows * (double)fabs(b - bp);
      dang += f.cr_dangling * (double)b;
      xmax = fmax(xmax, fabs((double)b) * f.cr_max_inv_outw);
    }
    if (f.init_prev >= 0) f.scal->base = (WT)f.init_prev;  // becomes base_prev: the rows' value before the first iteration
    if (f.totals) { f.totals[0] = diff; f.totals[1] = dang; f.totals[2] = xmax; }  // multi-GPU: this rank's share, folded with the peers' later
    else tiled_write_scalars<WT>(f.scal, diff, dang, xmax, f.alpha, f.nv_global, f.personalized, f.wmax);
  }
  __syncthreads();
}

template <typename WT>
__global__ void __launch_bounds__(TP2_BLOCK) k_tiled_finish(fin_args<WT> f)
{
  __shared__ double scratch[3 * TP2_BLOCK];
  finish_scalars<WT>(f, scratch, threadIdx.x, TP2_BLOCK);
}

// multi-GPU: folds the per-rank (diff, dangling, xmax) triples found at stride `stride_bytes` in the all-gathered buffer
template <typename WT>
__global__ void k_tiled_scalars_from_ranks(unsigned char const* recv, size_t first_off, size_t stride_bytes, int nranks, fin_args<WT> f)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double diff = 0, dang = 0, xmax = 0;
  for (int r = 0; r < nranks; ++r) {
    double const* t = reinterpret_cast<double const*>(recv + first_off + (size_t)r * stride_bytes);
    diff += t[0]; dang += t[1]; xmax = fmax(xmax, t[2]);
  }
  tiled_write_scalars<WT>(f.scal, diff, dang, xmax, f.alpha, f.nv_global, f.personalized, f.wmax);
}

// iteration-0 state: x = pr / out_w, partial dangling mass and max |x| per block
template <typename WT>
__global__ void __launch_bounds__(256) k_tiled_prologue(WT const* pr, WT const* outw, int32_t const* xcol, WT* x, int64_t nv, double* partials)
{
  __shared__ double red[8];
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double dang = 0.0, xmax = 0.0;
  for (; i < nv; i += stride) {
    WT p = pr[i], ow = outw[i];
    WT xv = p / (ow == WT(0) ? WT(1) : ow);
    if (xcol) { int32_t const c = xcol[i]; if (c >= 0) x[c] = xv; }
    else x[i] = xv;
    xmax = fmax(xmax, fabs((double)xv));
    if (ow == WT(0)) dang += (double)p;
  }
  dang = group_sum(dang, 64);
  for (int o = 32; o > 0; o >>= 1) xmax = fmax(xmax, __shfl_xor(xmax, o));
  if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = dang; red[2 * (threadIdx.x >> 6) + 1] = xmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[3 * blockIdx.x]     = 0.0;
    partials[3 * blockIdx.x + 1] = red[0] + red[2] + red[4] + red[6];
    partials[3 * blockIdx.x + 2] = fmax(fmax(red[1], red[3]), fmax(red[5], red[7]));
  }
}

// =================================================================================================
// phase 1
// =================================================================================================
template <typename WT>
struct p1_args {
  uint16_t const* src16;
  uint8_t const* bits;
  WT const* weights;
  uint32_t const* delta1;  // slot - run index of slot block b at [b + 1], padded
  uint32_t const* wrec;    // per-wavefront records (TP_REC_DWORDS dwords each, layout in spmv_tiled.hpp)
  int32_t const* chunk_begin;  // [n_chunks][4] (unused, first work item, end work item, source tile); items of a chunk share one source tile
  int n_chunks;
  int n_static_chunks;         // chunks [0, n_static_chunks) are pre-assigned: workgroup b owns [wg_static[2b], wg_static[2b + 1])
  int32_t const* wg_static;
  uint32_t* counter;           // cursor of the dynamic chunks: 0 on entry, reset by phase 2
  int T;
  WT const* x;
  WT* part;
  WT alpha;
  uint32_t pmask, plog, chunk, ncols;
  fin_args<WT> fin;  // scalars of the PREVIOUS iteration, folded by workgroup 0 before it starts streaming
  unsigned long long* dbg{nullptr};  // CUGRAPH_AMD_TILED_DEBUG: per-workgroup (cycles, tile loads)
};

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// 32-bit byte offsets from a wave-uniform base: the load/store takes the SGPR-base + VGPR-offset form (one VGPR and no
// 64-bit address arithmetic per access).  build_tiled_csc checks that every array stays below 4 GiB.
template <typename T>
__device__ __forceinline__ T ld32(void const* base, uint32_t byte_off)
{
  return *reinterpret_cast<T const*>(static_cast<char const*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st32(void* base, uint32_t byte_off, T v)
{
  *reinterpret_cast<T*>(static_cast<char*>(base) + byte_off) = v;
}


// ---- vector-memory operations of phase 1's item loop and its explicit wait.
// gfx9 has ONE counter for loads and stores (vmcnt, retired in order).  The loop is rotated so that its back edge follows
// the one explicit vmcnt(0) of an iteration: nothing is in flight across the back edge, so the compiler's wait insertion
// (which merges states conservatively at joins: it put vmcnt(1) in front of the first use of an item's edge data and
// vmcnt(0) between the partial stores of the long-run path in the previous, un-rotated loop -- measured: the partial
// stores cost 0.36 of phase 1's 1.04 ms at RMAT-26) finds nothing pending at the loop header and adds no wait of its own
// in the steady state.  (An inline-asm form of the same operations, invisible to that pass, was built in round 2 and rejected: the compiler is free to
// copy a register whose asm load is still in flight.)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vm_ld128(u32x4_t& d, void const* base, uint32_t off) { d = *reinterpret_cast<u32x4_t const*>(static_cast<char const*>(base) + off); }
__device__ __forceinline__ void vm_ld128_o16(u32x4_t& d, void const* base, uint32_t off) { d = *reinterpret_cast<u32x4_t const*>(static_cast<char const*>(base) + off + 16); }
__device__ __forceinline__ void vm_ld16u(uint32_t& d, void const* base, uint32_t off) { d = *reinterpret_cast<uint16_t const*>(static_cast<char const*>(base) + off); }
__device__ __forceinline__ void vm_ld32(uint32_t& d, void const* base, uint32_t off) { d = *reinterpret_cast<uint32_t const*>(static_cast<char const*>(base) + off); }
template <typename V>
__device__ __forceinline__ void vm_st(void* base, uint32_t off, V v)
{
  *reinterpret_cast<V*>(static_cast<char*>(base) + off) = v;
}
__device__ __forceinline__ void vm_wait0() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0); lgkmcnt / expcnt untouched

// LDS access by ABSOLUTE byte address (address space 3): `ds_read_b32 v, vaddr` with nothing added.  Through a generic pointer
// (`xs + offset`) the compiler adds the symbol's address first -- a link-time constant that happens to be 0: the dynamic LDS of these kernels
// starts at LDS address 0, they declare no static LDS; k_tiled_phase1 traps if that ever changes -- which costs a `v_add_u32 v, 0, v` per
// access: 16 per lane and work item in the gathers of phase 1.
template <typename T> __device__ __forceinline__ T lds_ld(uint32_t a) { return *reinterpret_cast<__attribute__((address_space(3))) T const*>((uintptr_t)a); }
template <typename T> __device__ __forceinline__ void lds_st(uint32_t a, T v) { *reinterpret_cast<__attribute__((address_space(3))) T*>((uintptr_t)a) = v; }

// LDS byte offset of 16-bit tile-local index number HALF of w, scaled by the element size, in ONE VALU op (SDWA word select
// + shift); the x tile starts at LDS address 0
template <int HALF, int SHIFT>
__device__ __forceinline__ uint32_t idx_offset(uint32_t w)
{
  uint32_t r;
  if constexpr (HALF == 0) asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(w), "n"(SHIFT));
  else asm("v_lshlrev_b32_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(w), "n"(SHIFT));
  return r;
}
// ~0 when bit K of v is set, else 0 (one op; kept opaque so that the select below stays a bit mask)
template <int K>
__device__ __forceinline__ uint32_t bit_fill(uint32_t v)
{
  uint32_t m;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(v), "n"(K));
  return m;
}
template <int... K, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, K...>, F&& f)
{
  (f(std::integral_constant<int, K>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct p1_regs {  // one work item's data for one lane: TP_EPL consecutive edges + one dword of the wavefront's record
  u32x4_t id[TP_EPL / 8];
  uint32_t fl;   // TP_EPL run-start bits
  uint32_t rec;  // dword min(lane, TP_REC_DWORDS - 1) of the record (only lanes < TP_REC_DWORDS are ever read)
  uint32_t es;   // first edge position of the wavefront's share (wave-uniform)
};
static_assert(TP_EPL == 16, "p1_load issues two 16-byte index loads and one 16-bit flag load per lane");
// straight-line (4 loads, no branch; lanes past the record re-read its last dword)
template <typename WT>
__device__ __forceinline__ void p1_load(p1_args<WT> const& a, int item, int wave, int lane, p1_regs& r)
{  // arrays are over-allocated and zero padded: no bounds checks
  r.es = (uint32_t)item * (uint32_t)TP_ITEM + (uint32_t)wave * TP_WLEN;
  uint32_t const e = r.es + (uint32_t)TP_EPL * (uint32_t)lane;
  // (the base is the wavefront's share -- wave-uniform, a 64-bit scalar add -- and the lane offset a constant: byte offsets of the whole
  // array would pass 2^32 from 2^31 edges on)
  uint16_t const* const wbase = a.src16 + r.es;
  vm_ld128(r.id[0], wbase, (uint32_t)(2 * TP_EPL) * (uint32_t)lane);
  vm_ld128_o16(r.id[1], wbase, (uint32_t)(2 * TP_EPL) * (uint32_t)lane);
  (void)e;
  vm_ld16u(r.fl, a.bits + (r.es >> 3), (uint32_t)(TP_EPL / 8) * (uint32_t)lane);  // (TP_EPL consecutive edges per lane = TP_EPL / 8 bytes of the bitmap)
  vm_ld32(r.rec, a.wrec + ((size_t)item * TP_WAVES + (size_t)wave) * TP_REC_DWORDS, 4u * (uint32_t)min(lane, TP_REC_DWORDS - 1));
}

// Run ordinal n within the wavefront's range: n = 0 is the run that was already open at `es` (its partial goes to the
// head slot), n >= 1 is run (rank - 1 + n), whose slot is (rank - 1 + n) + delta1[blk + #block starts among the first n runs
// of the range] (record bits 0 .. n - 1).
struct p1_runs {  // what the bitmap and the record of one work item say about its runs
  uint32_t f, ex_c, c_all;
  uint32_t dreg[TP_NDREG];  // delta1[blk + 64 k + lane]: the block deltas this item's runs can need (lane = block relative to blk)
  uint32_t slot_tail;       // slot of the run still open at the end of the range
  uint32_t es, ee, rank, head_slot, blk;
};

__device__ __forceinline__ void p1_counts(int lane, p1_regs const& rg, p1_runs& q)
{
  q.es = rg.es; q.ee = rg.es + rdl(rg.rec, TP_REC_NVAL); q.rank = rdl(rg.rec, TP_REC_RANK); q.head_slot = rdl(rg.rec, TP_REC_HEAD);
  q.slot_tail = rdl(rg.rec, TP_REC_TAIL); q.blk = rdl(rg.rec, TP_REC_BLK);
  uint32_t const e     = q.es + (uint32_t)TP_EPL * (uint32_t)lane;
  uint32_t const nval  = min((uint32_t)TP_EPL, q.ee > e ? q.ee - e : 0u);
  q.f                  = rg.fl & ((1u << nval) - 1u);
  uint32_t const nf    = __popc(q.f);
  uint32_t const c_inc = wave_inclusive_sum_u32(nf);
  q.ex_c               = c_inc - nf;
  q.c_all              = (uint32_t)__builtin_amdgcn_readlane((int)c_inc, 63);
}

// The slot of run ordinal n >= 1 is (rank - 1 + n) + delta1[blk + B(n)], B(n) = number of block starts among the record bits
// [0, n): a wavefront's runs fall into few consecutive slot blocks (5 on average at RMAT-26, one per ~24 runs in the coldest
// tiles), so instead of one delta load per 64 runs the item requests delta1[blk .. blk + 64 * TP_NDREG) ONCE, lane-wise (one
// or two coalesced loads, one or two registers), an item ahead; the write-out picks a run's delta from lane B(n) of that
// register with ds_bpermute (LDS crossbar, no memory access).  Items with more blocks fall back to a direct load.
template <typename WT>
__device__ __forceinline__ void p1_issue_slots(p1_args<WT> const& a, int lane, p1_regs const& rg, p1_runs& q)
{
#pragma unroll
  for (int k = 0; k < TP_NDREG; ++k) {
    q.dreg[k] = 0;
    if (k == 0 || q.c_all > (uint32_t)(64 * k)) vm_ld32(q.dreg[k], a.delta1 + q.blk, 4u * ((uint32_t)(64 * k) + (uint32_t)lane));  // (block starts <= run starts)
  }
}
// delta of run ordinal n = 64 * j + lane (j wave-uniform); `base` = B(64 j) on entry, B(64 (j + 1)) on exit (scalar popcounts)
template <typename WT>
__device__ __forceinline__ uint32_t p1_delta_of_group(p1_args<WT> const& a, p1_runs const& q, uint32_t rec, uint32_t j, uint32_t& base)
{
  uint32_t const lo = rdl(rec, 2u * j), hi = rdl(rec, 2u * j + 1u);
  uint32_t const b  = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, base));  // B(n) of this lane's run
  base += (uint32_t)__builtin_popcount(lo) + (uint32_t)__builtin_popcount(hi);               // >= every lane's b
  uint32_t d = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b << 2), (int)q.dreg[0]);
#pragma unroll
  for (int k = 1; k < TP_NDREG; ++k)
    if (base >= (uint32_t)(64 * k)) {  // wave-uniform: some lane of the group may need blocks [64 k, 64 k + 64)
      uint32_t const dk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b << 2), (int)q.dreg[k]);
      d = (b >> 6) == (uint32_t)k ? dk : d;
    }
  if (base >= (uint32_t)(64 * TP_NDREG)) {  // rare: more slot blocks than the prefetched registers cover
    uint32_t direct;
    vm_ld32(direct, a.delta1, 4u * (q.blk + b));
    vm_wait0();
    d = b >= (uint32_t)(64 * TP_NDREG) ? direct : d;
  }
  return d;
}

// run totals of the lanes selected by `mine` -> staging area, in run order, starting at ordinal base_c
// start[k] != 0 <=> a run starts at element k (it closes the run that ran up to element k - 1).  The totals are staged without
// the carry of the previous lanes; the lane's FIRST staged total is patched afterwards (LDS operations of one wavefront
// execute in order), which keeps the select out of the 16 predicated stores.
template <typename WT>
__device__ __forceinline__ void p1_stage(WT* stage, p1_runs const& q, WT const (&r)[TP_EPL], WT carry_in, uint32_t base_c, bool mine)
{
  if (mine && q.f) {
    WT* p             = stage + (q.ex_c - base_c);
    WT* const p_first = p;
    static_for<TP_EPL>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (q.f & (1u << k)) {  // (the flag word is re-tested here rather than kept as 16 masks: registers)
        if constexpr (k == 0) *p = WT(0);
        else *p = r[k - 1];
        ++p;
      }
    });
    *p_first += carry_in;
  }
}

// staging area -> partial buffer: lane i takes staged totals i, 64 + i, ... (coalesced); run ordinals [n_lo, n_hi), staged at
// [0, n_hi - n_lo).  The caller has waited for q.dreg.
template <typename WT>
__device__ __forceinline__ void p1_writeout(p1_args<WT> const& a, WT const* stage, int lane, uint32_t rec, p1_runs const& q, uint32_t n_lo, uint32_t n_hi)
{
  static_assert(TP_WLEN <= 1024, "a wavefront's share of an item starts at most 1024 runs: 16 groups of 64 ordinals, 32 record dwords");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint32_t const rm1 = q.rank - 1u;  // run index of ordinal n = rm1 + n (modulo 2^32)
  uint32_t base = 0;                 // B(64 jj): block starts among the record bits below the group
#pragma unroll
  for (int jj = 0; jj < TP_WLEN / 64; ++jj) {
    if ((uint32_t)(64 * jj) < n_hi) {  // wave-uniform
      uint32_t const d = p1_delta_of_group<WT>(a, q, rec, (uint32_t)jj, base);
      uint32_t const n = (uint32_t)lane + 64u * (uint32_t)jj;
      uint32_t slot    = d + (rm1 + 64u * (uint32_t)jj) + (uint32_t)lane;
      if (jj == 0) slot = lane == 0 ? q.head_slot : slot;
      if (n >= n_lo && n < n_hi) vm_st(a.part, slot * (uint32_t)sizeof(WT), stage[n - n_lo]);
    }
  }
  __builtin_amdgcn_wave_barrier();  // the staging area is reused by the next item
}

struct p1_pending {  // what compute leaves for the (later) write-out of the same item
  uint32_t base_c, count;
};

// Values: LDS gathers, in-lane segmented sum, wave64 segmented scan; run totals staged in LDS.  Returns (lane 63) the total
// of the run still open at the end of the range -- or of the whole range when no run starts in it.
template <typename WT, bool WEIGHTED>
__device__ __forceinline__ WT p1_compute(p1_args<WT> const& a, WT const* xs, WT* stage, int lane, p1_regs const& rg, p1_runs& q, p1_pending& pend)
{
  uint32_t const e = q.es + (uint32_t)TP_EPL * (uint32_t)lane;
  WT r[TP_EPL];  // values, then in place: running sum since the last run start at or before element k
  static_for<TP_EPL>([&](auto kc) {  // positions past `ee` hold zero-padded (valid) indices; their values are masked below
    constexpr int k  = decltype(kc)::value;
    uint32_t const w = k % 8 < 2 ? rg.id[k / 8].x : k % 8 < 4 ? rg.id[k / 8].y : k % 8 < 6 ? rg.id[k / 8].z : rg.id[k / 8].w;
    constexpr int sh = sizeof(WT) == 4 ? 2 : 3;
    r[k] = lds_ld<WT>(idx_offset<(k & 1), sh>(w));  // (the tile sits at LDS address 0)
  });
  if constexpr (WEIGHTED) {
#pragma unroll
    for (int k = 0; k < TP_EPL; ++k) r[k] *= ld32<WT>(a.weights + q.es, ((uint32_t)TP_EPL * (uint32_t)lane + (uint32_t)k) * (uint32_t)sizeof(WT));
  }
  if (e + TP_EPL > q.ee) {
    uint32_t const nval = q.ee > e ? q.ee - e : 0u;
#pragma unroll
    for (int k = 0; k < TP_EPL; ++k) r[k] = (uint32_t)k < nval ? r[k] : WT(0);
  }
  pend.base_c = 0;
  pend.count  = 0;
  if (q.c_all == 0) {  // no run starts in the wavefront's edges (inside a long run): plain sum (goes to the head slot)
    WT t = r[0];
#pragma unroll
    for (int k = 1; k < TP_EPL; ++k) t += r[k];
    return wave_sum_to_lane63(t);
  }
  // in-lane segmented sum: r[k] = v[k] + (run start at k ? 0 : r[k-1]); the select is a bit mask (0 / ~0 from the flag bit)
  uint32_t const nflags = ~q.f;
  static_for<TP_EPL>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k >= 1) {
      uint32_t const cont = bit_fill<k>(nflags);  // ~0 when element k continues the run of element k - 1, 0 when a run starts at k
      if constexpr (sizeof(WT) == 4) r[k] += __uint_as_float(__float_as_uint(r[k - 1]) & cont);
      else r[k] += __longlong_as_double(__double_as_longlong(r[k - 1]) & (long long)(int)cont);
    }
  });
  WT s       = r[TP_EPL - 1];
  uint32_t c = __popc(q.f);
  wave_seg_scan(s, c);
  WT const carry_in = dpp_val<0x138, 0xF>(s);  // wave_shr:1: segmented sum up to the previous lane (0 in lane 0; the range starts with an empty carry)
  if (q.c_all <= (uint32_t)TP_STAGE) {
    p1_stage<WT>(stage, q, r, carry_in, 0u, true);
    pend.count = q.c_all;
  } else {  // more run totals than the staging area holds: lanes 0-31 (at most TP_STAGE runs) are written out right away
    uint32_t const half = (uint32_t)__builtin_amdgcn_readlane((int)c, 31);
    p1_stage<WT>(stage, q, r, carry_in, 0u, lane < 32);
    vm_wait0();  // the delta entries of this item were requested just before this compute (rare path: a full stall)
    p1_writeout<WT>(a, stage, lane, rg.rec, q, 0u, half);
    p1_stage<WT>(stage, q, r, carry_in, half, lane >= 32);
    pend.base_c = half;
    pend.count  = q.c_all - half;
  }
  return s;
}

#ifndef P1_ISSUE_PRIO
#define P1_ISSUE_PRIO 2  // wavefront priority of phase 1's issue part (k_tiled_phase1: body)
#endif
struct p1_iter {  // position in this workgroup's item sequence (all fields wave-uniform); item < 0: exhausted
  int item, end, pos, tile;
};

template <typename WT, bool WEIGHTED, bool DBG = false>
__global__ void __launch_bounds__(TP_BLOCK, 4) k_tiled_phase1(p1_args<WT> a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WT* xs = reinterpret_cast<WT*>(smem);
  // ring of four chunk descriptors (chunk id, first item, end item, source tile) behind tile + staging; the tile sits at LDS address 0
  int4* s_chunk = reinterpret_cast<int4*>(smem + ((size_t)a.T + (size_t)TP_WAVES * TP_STAGE) * sizeof(WT));
  int const tid = threadIdx.x, lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem != 0u) __builtin_trap();  // lds_ld / lds_st / idx_offset address the tile from LDS address 0
  WT* stage = xs + a.T + wave * TP_STAGE;

  if (blockIdx.x == 0 && a.fin.partials) finish_scalars<WT>(a.fin, reinterpret_cast<double*>(smem), tid, TP_BLOCK);

  // Work is handed out dynamically in CHUNKS (a few consecutive work items of one source tile; hottest tiles first, the
  // small cold tiles last); chunk ids are fetched three chunks ahead so the item sequence is known two items ahead.
  //
  unsigned long long const t_start = wall_clock64();
  int n_tiles = 0;
  // thread 0 draws the next chunk and stages its descriptor in LDS: nothing in the item loop depends on a global load that
  // was just issued (a dependent load is followed by vmcnt(0), which also waits for every prefetch and store in flight)
  int my_next = 0, my_end = 0;  // thread 0: this workgroup's private (static) chunks
  if (tid == 0) { my_next = a.wg_static[2 * blockIdx.x]; my_end = a.wg_static[2 * blockIdx.x + 1]; }
  auto draw = [&](int slot) {
    int const cid = my_next < my_end ? my_next++ : a.n_static_chunks + (int)atomicAdd(a.counter, 1u);
    int4 c{cid, 0, 0, 0};
    if (cid < a.n_chunks) c = reinterpret_cast<int4 const*>(a.chunk_begin)[cid];
    c.x = cid;
    s_chunk[slot] = c;
  };
  if (tid == 0) { draw(0); draw(1); draw(2); }
  __syncthreads();
  auto enter = [&](p1_iter& x) {  // (readfirstlane: the item sequence is wave-uniform, and the compiler should know it -- scalar
    int4 const c = s_chunk[x.pos & 3];  // branches keep the loop's control flow, and with it the wait insertion, simple)
    int const cid = __builtin_amdgcn_readfirstlane(c.x);
    if (cid >= a.n_chunks) { x.item = -1; }
    else { x.item = __builtin_amdgcn_readfirstlane(c.y); x.end = __builtin_amdgcn_readfirstlane(c.z); x.tile = __builtin_amdgcn_readfirstlane(c.w); }
  };
  auto advance = [&](p1_iter& x) {
    if (x.item < 0) return;
    if (++x.item >= x.end) { ++x.pos; enter(x); }
  };
  p1_iter I{0, 0, 0, 0};
  enter(I);
  if (tid == 0) draw(3);  // read only after the next chunk-transition barrier
  int curJ = -1;
  p1_regs rA, rB;
  p1_runs qA, qB;
  p1_iter Jt = I;
  // loop state carried from an item's compute to its store phase (wave-uniform except `tail`, which lives in lane 63)
  p1_pending pend{0, 0};
  WT tail   = WT(0);
  bool busy = false;

  // stage the source tile of item I (first item of a chunk: every wavefront has passed the chunk-transition barrier), then
  // LDS gathers + scans of its edges: run totals staged in LDS, no use of global memory besides the tile itself
  auto compute_item = [&](p1_regs& rg, p1_runs& q) {
    int const J = I.tile;
    if (J != curJ) {
      // x is allocated (and zero-filled) up to nJ * T elements; indices are clamped instead of guarded so that the loads
      // stay straight-line (8 in flight per thread)
      using vec4 = typename std::conditional<sizeof(WT) == 4, float4, double4>::type;
      vec4 const* src = reinterpret_cast<vec4 const*>(a.x + (size_t)J * a.T);
      vec4* dst       = reinterpret_cast<vec4*>(xs);
      int const n4    = a.T / 4;
      // (round 6: the same fill through LDS-DMA -- global_load_lds_dwordx4, 1 KiB per wavefront instruction, alpha applied in place afterwards -- is
      // bit-identical and NEUTRAL, 0.9211 against 0.9209 ms alternated on one plan, profiles/r6o_p1_dma_fill.txt: the fill is one round trip either way)
      for (int i0 = 0; i0 < n4; i0 += 8 * TP_BLOCK) {
        vec4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[min(i0 + j * TP_BLOCK + tid, n4 - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j].x *= a.alpha; v[j].y *= a.alpha; v[j].z *= a.alpha; v[j].w *= a.alpha;
          dst[min(i0 + j * TP_BLOCK + tid, n4 - 1)] = v[j];  // the clamped duplicates write the same value
        }
      }
      __syncthreads();
      curJ = J;
      ++n_tiles;
    }
    pend.base_c = 0; pend.count = 0;
    tail = WT(0);
    busy = q.es < q.ee;  // wave-uniform; an empty share (tail of a tile) has nothing to compute or store
    if (busy) tail = p1_compute<WT, WEIGHTED>(a, xs, stage, lane, rg, q, pend);
  };

  // Software pipeline around the ONE memory counter of gfx9 (vmcnt counts loads AND stores and retires in order).  Steady
  // state, per iteration (item i = I was computed in the previous iteration; its edge data sits in `cur`, item i+1's in `nxt`):
  //     counts(i+1); stores(i); request slot deltas(i+1) and edge data(i+2)      <- everything is ISSUED here ...
  //     [chunk transition: barrier, next chunk id, source tile of item i+1]
  //     compute(i+1)                                                              <- ... has a whole compute to complete ...
  //     vmcnt(0)                                                                  <- ... and is waited for here, once.
  // The back edge follows the wait, so no operation is in flight across it (see the note on vm_ld*).
  if (I.item >= 0) {
    p1_load<WT>(a, I.item, wave, lane, rA);
    vm_wait0();
    p1_counts(lane, rA, qA);
    p1_issue_slots<WT>(a, lane, rA, qA);
    advance(Jt);
    p1_load<WT>(a, max(Jt.item, 0), wave, lane, rB);  // unconditional (a dummy re-read of item 0 when there is no next item)
    compute_item(rA, qA);
    vm_wait0();
  }
  auto body = [&](p1_regs& cur, p1_runs& qc, p1_regs& nxt, p1_runs& qn) {
    // The issue part of an iteration -- stores of item i, slot deltas of item i + 1, edge data of item i + 2: ~30 memory instructions -- runs at raised
    // wavefront priority: the SIMD's other three wavefronts are mostly inside a compute (VALU 60 % / LDS 40 % busy), and at equal priority the issuing
    // wavefront's requests leave interleaved with their arithmetic, i.e. late.  Fresh processes of the driver's line, alternating, 2 x 8 pairs: 1.341 ->
    // 1.318-1.323 ms at RMAT-26, phase 1 0.934 -> 0.913 (profiles/r6an_ab_fresh.txt, r6ao_ab_fresh.txt; levels 1 / 2 / 3 alike; the same around phase 2's
    // loads: no effect).  Same instructions, same arithmetic: bit-identical results.
    __builtin_amdgcn_s_setprio(P1_ISSUE_PRIO);
    bool const have_next = Jt.item >= 0;
    if (have_next) p1_counts(lane, nxt, qn);
    if (busy) {
      if (pend.count) {
        int wl = lane;
        asm volatile("" : "+v"(wl));  // (the kernel sits AT its 128-register budget: keep the lane-derived LDS addresses of the write-out from
        p1_writeout<WT>(a, stage, wl, cur.rec, qc, pend.base_c, pend.base_c + pend.count);  // being hoisted out of the item loop and spilled)
      }
      if (lane == 63) vm_st(a.part, qc.slot_tail * (uint32_t)sizeof(WT), tail);  // slot_tail = head slot when no run starts in the range
    }
    if (have_next) p1_issue_slots<WT>(a, lane, nxt, qn);
    p1_iter K = Jt;
    advance(K);
    p1_load<WT>(a, max(K.item, 0), wave, lane, cur);  // `cur` is free: data(i+2); unconditional
    int const old_pos = I.pos;
    I  = Jt;
    Jt = K;
    busy = false;
    __builtin_amdgcn_s_setprio(0);
    if (I.item >= 0) {
      if (I.pos != old_pos) {  // next item belongs to another chunk: all wavefronts are done with the tile
        __syncthreads();
        if (tid == 0) draw((I.pos + 3) & 3);
      }
      compute_item(nxt, qn);
    }
    // every path through an iteration ends here (no early exit: a loop exit that bypasses the wait would reach the loop
    // header's join with operations in flight, and the compiler would then guard every later use with its own vmcnt(0))
    vm_wait0();
  };
  while (I.item >= 0) {  // (the second call is a no-op apart from a dummy load when the first one consumed the last item)
    body(rA, qA, rB, qB);
    body(rB, qB, rA, qA);
  }
  if constexpr (DBG) {
    __syncthreads();
    if (tid == 0) { a.dbg[2 * blockIdx.x] = wall_clock64() - t_start; a.dbg[2 * blockIdx.x + 1] = (unsigned long long)n_tiles; }
  }
}

// =================================================================================================
// phase 2 + PageRank epilogue
// =================================================================================================
template <typename WT>
struct p2_args {
  WT const* part;
  uint16_t const* dstl16;
  uint32_t const* dstl12;  // non-null: 12-bit packed destinations (see tiled_csc_t)
  uint32_t const* tile_row0;
  uint32_t const* region_off;
  int nI;
  tiled_epilogue<WT> e;
  uint32_t* counters;
  double const* tile_wmax{nullptr};  // tiled_csc_t::tile_wmax (fp32: the fixed-point scale of the tile)
  int stagger{0};                    // start offset between the co-resident workgroups of the first generation (k_tiled_phase2)
};

// fp32 partials are accumulated as signed 64-bit fixed point (value * 2^k of the tile, rounded to nearest: tiled_to_fixed in spmv_tiled.hpp):
// LDS integer atomics run at full rate on gfx950 while ds_add_f32 is ~5x slower (tools/ubench/lds_update_bench.hip), and integer addition is
// associative, so the result does not depend on the order in which wavefronts reach a row (bit-reproducible).
template <typename WT> struct p2_acc { using type = WT; };
template <> struct p2_acc<float> { using type = unsigned long long; };

template <typename WT, bool PERS>
__global__ void __launch_bounds__(TP2_BLOCK) k_tiled_phase2(p2_args<WT> a)
{
  using ACC = typename p2_acc<WT>::type;
  constexpr int RPT = TP2_ROWS / TP2_BLOCK;  // rows per thread in the epilogue
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  ACC* acc = reinterpret_cast<ACC*>(smem2);  // [TP2_ROWS]
  __shared__ double red[3 * (TP2_BLOCK / 64)];
  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int const I = (int)blockIdx.x;  // destination tile, or (I >= nI_act) block of tiled_const_rows
  if (a.e.cr.nI_act > 0 && I >= a.e.cr.nI_act) {  // tiled_const_rows: x of the live columns whose rows have no in-edge
    int const cblock = I - a.e.cr.nI_act;
    WT const b       = a.e.scal->base;
    int64_t const n  = a.e.cr.n_cols;
    WT* const xo           = a.e.x_next + a.e.cr.c0;
    int32_t const* const ci = a.e.cr.col_idx;
    for (int64_t j = (int64_t)cblock * TP2_CONST_COLS + tid, k = 0; k < 8 && j < n; ++k, j += TP2_BLOCK) {
      WT const ow = a.e.cr.outw_c[j];
      WT const xv = b / (ow == WT(0) ? WT(1) : ow);
      if (ci) a.e.x_next[ci[j]] = xv; else xo[j] = xv;
    }
    return;
  }
  // The first generation of workgroups (four per CU, dispatched together) starts a quarter of a workgroup's life apart: left in lockstep the
  // four share their latency-bound prologue / epilogue (40 % of a workgroup's 52 us) instead of hiding it behind each other's streaming loop,
  // and the lockstep survives for several generations because the tiles have equal cost.  a.stagger = units of s_sleep(127) (~3.4 us) per
  // quarter; 0 on grids too small for it to pay (tiled_phase2).  Measured: profiles/r6q_phase2_stagger.txt.
  if (a.stagger > 0 && blockIdx.x < 1024u) {
    int const q = (int)((blockIdx.x >> 8) & 3u) * a.stagger;
    for (int i = 0; i < q; ++i) __builtin_amdgcn_s_sleep(127);
  }
  uint32_t const row0 = a.tile_row0[I], nrows = a.tile_row0[I + 1] - row0;
  uint32_t const s0 = a.region_off[I], s1 = a.region_off[I + 1];
  tiled_epilogue<WT> const& e = a.e;
  pr_scalars<WT> const sc     = *e.scal;

  // the epilogue's inputs do not depend on the accumulation: request them first
  WT old[RPT], ow[RPT];
  int32_t col[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    uint32_t i = tid + j * TP2_BLOCK;
    bool const in = i < nrows && e.raw_y == nullptr;  // (plain SpMV mode has no epilogue)
    old[j] = (e.need_diff && in) ? e.pr[(size_t)row0 + i] : WT(0);
    ow[j]  = in ? e.outw[(size_t)row0 + i] : WT(1);
    col[j] = (e.xcol && in) ? e.xcol[(size_t)row0 + i] : (int32_t)(row0 + i);
  }
  for (uint32_t i = tid; i < nrows; i += TP2_BLOCK) acc[i] = ACC(0);
  // fp32: the tile's fixed-point scale (spmv_tiled.hpp, tiled_to_fixed): its partials and row sums are bounded by fx_unit * tile_wmax[I]
  double fx_scale = 1.0, fx_inv = 1.0;
  if constexpr (sizeof(WT) == 4) tiled_tile_scale(sc.fx_unit * a.tile_wmax[I], &fx_scale, &fx_inv);
  __syncthreads();
  // 8 slots per thread and step: every load of a step is issued before the first partial of the step is added (44 bytes per lane in flight,
  // 32 wavefronts per CU)
  for (uint32_t s = s0 + 8 * tid; s < s1; s += 8 * TP2_BLOCK) {
    uint32_t idx8[8];
    WT v[8];
    uint32_t w0, w1, w2, w3 = 0;
    if (a.dstl12) {  // wave-uniform: 8 slots = 3 dwords of 12-bit tile-local rows
      typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
      u32x3_t const w = __builtin_nontemporal_load(reinterpret_cast<u32x3_t const*>(a.dstl12 + 3u * (s >> 3)));  // streamed once per iteration
      w0 = w.x; w1 = w.y; w2 = w.z;
    } else {  // 16-bit destinations: 8 slots = 16 bytes (tiles of <= 4096 rows only reach here with CUGRAPH_AMD_TILED_DSTL16)
      uint4 const d = *reinterpret_cast<uint4 const*>(a.dstl16 + s);
      w0 = d.x; w1 = d.y; w2 = d.z; w3 = d.w;
    }
    if constexpr (sizeof(WT) == 4) {
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      f32x4_t const q0 = __builtin_nontemporal_load(reinterpret_cast<f32x4_t const*>(a.part + s)), q1 = __builtin_nontemporal_load(reinterpret_cast<f32x4_t const*>(a.part + s + 4));
      v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double2 const p = *reinterpret_cast<double2 const*>(a.part + s + 2 * k);
        v[2 * k] = p.x; v[2 * k + 1] = p.y;
      }
    }
    if (a.dstl12) {
      idx8[0] = w0 & 0xFFFu; idx8[1] = (w0 >> 12) & 0xFFFu; idx8[2] = (w0 >> 24) | ((w1 & 0xFu) << 8); idx8[3] = (w1 >> 4) & 0xFFFu;
      idx8[4] = (w1 >> 16) & 0xFFFu; idx8[5] = (w1 >> 28) | ((w2 & 0xFFu) << 4); idx8[6] = (w2 >> 8) & 0xFFFu; idx8[7] = w2 >> 20;
    } else {
      uint32_t const w4[4] = {w0, w1, w2, w3};
#pragma unroll
      for (int k = 0; k < 8; ++k) idx8[k] = (k & 1) ? (w4[k >> 1] >> 16) : (w4[k >> 1] & 0xFFFFu);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (sizeof(WT) == 4) atomicAdd(&acc[idx8[k]], tiled_to_fixed(v[k], fx_scale));  // ds_add_u64; padding slots hold 0
      else atomicAdd(&acc[idx8[k]], v[k]);                                                      // ds_add_f64
    }
  }
  __syncthreads();
  if (e.raw_y) {  // (workgroup-uniform) plain SpMV: the row sums and nothing else -- the 2-D multi-GPU layout reduces them over ranks first
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      uint32_t i = tid + j * TP2_BLOCK;
      if (i < nrows) {
        WT sum;
        if constexpr (sizeof(WT) == 4) sum = (WT)((double)(long long)acc[i] * fx_inv);
        else sum = acc[i];
        e.raw_y[(size_t)row0 + i] = sum;
      }
    }
    if (tid == 0 && I == 0) a.counters[0] = 0;  // rewind phase 1's chunk cursor
    return;
  }

  // Every row's values first, every store afterwards: a store between the rows makes the compiler drain the memory counter (vmcnt(0)) in front of
  // the next row's first use of a register loaded at the top of the kernel -- one store round trip per row, eight in a row per thread (ISA of
  // round 5).  With the stores at the end the only wait of the epilogue is the one for its inputs, which arrived long ago.
  double diff = 0.0, dang = 0.0, xmax = 0.0;
  WT val[RPT], xn[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    uint32_t i = tid + j * TP2_BLOCK;
    val[j] = WT(0); xn[j] = WT(0);
    if (i < nrows) {
      WT sum;
      if constexpr (sizeof(WT) == 4) sum = (WT)((double)(long long)acc[i] * fx_inv);
      else sum = acc[i];
      WT vv = sc.base + sum;
      if constexpr (PERS) vv += sc.pers_factor * e.pers[(size_t)row0 + i];
      val[j] = vv;
      xn[j]  = vv / (ow[j] == WT(0) ? WT(1) : ow[j]);
      if (e.need_diff) diff += (double)fabs(vv - old[j]);
      xmax = fmax(xmax, fabs((double)xn[j]));
      if (ow[j] == WT(0)) dang += (double)vv;
    }
  }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    uint32_t i = tid + j * TP2_BLOCK;
    if (i < nrows) {
      if (e.write_pr) e.pr[(size_t)row0 + i] = val[j];
      if (col[j] >= 0) e.x_next[col[j]] = xn[j];  // sources without out-edges have no column
    }
  }
  diff = group_sum(diff, 64);
  dang = group_sum(dang, 64);
  for (int o = 32; o > 0; o >>= 1) xmax = fmax(xmax, __shfl_xor(xmax, o));
  if (lane == 0) { red[3 * wave] = diff; red[3 * wave + 1] = dang; red[3 * wave + 2] = xmax; }
  __syncthreads();
  if (tid == 0) {  // folded by the next phase-1 launch (or k_tiled_finish): no device-scope fence per workgroup here
    if (I == 0) a.counters[0] = 0;  // phase 1 is over: rewind its chunk cursor for the next iteration
    double d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
    for (int k = 0; k < TP2_BLOCK / 64; ++k) { d0 += red[3 * k]; d1 += red[3 * k + 1]; d2 = fmax(d2, red[3 * k + 2]); }
    e.partials[3 * I]     = d0;
    e.partials[3 * I + 1] = d1;
    e.partials[3 * I + 2] = d2;
  }
}

template <typename K>
void ensure_max_lds(K kernel, int bytes)
{
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes - 1024));  // the kernels also hold a few static LDS words
}

template <typename WT>
fin_args<WT> make_fin(tiled_epilogue<WT> const& e, int n)
{
  fin_args<WT> f;
  f.partials = e.partials; f.n = n; f.scal = e.scal; f.totals = e.totals; f.alpha = e.alpha; f.nv_global = e.nv_global;
  f.personalized = e.pers != nullptr;
  f.wmax = e.wmax;
  if (e.cr.nI_act > 0) { f.cr_rows = (double)e.cr.n_rows; f.cr_dangling = (double)e.cr.n_dangling; f.cr_max_inv_outw = e.cr.max_inv_outw; }
  return f;
}

}  // namespace

template <typename WT>
void tiled_phase1(handle_t const& h, tiled_csc_t const& t, WT const* x, WT alpha, WT* part, uint32_t* counters, tiled_x_map<WT> const& map,
                  tiled_epilogue<WT> const* pending)
{
  if (t.n_items == 0) {
    if (pending) tiled_finish<WT>(h, *pending, tiled_fold_count(t, *pending));
    return;
  }
  p1_args<WT> a;
  a.src16     = t.src16.data();
  a.bits      = reinterpret_cast<uint8_t const*>(t.bits.data());
  a.weights   = t.weights.ptr ? t.weights.as<WT const>() : nullptr;
  a.delta1    = t.delta1.data();
  a.wrec      = t.wrec.data();
  a.chunk_begin = t.chunk_begin.data();
  a.n_chunks    = t.n_chunks;
  a.n_static_chunks = t.n_static_chunks;
  a.wg_static   = t.wg_static.data();
  a.counter     = counters;
  a.T         = t.T;
  a.x         = x;
  a.part      = part;
  a.alpha     = alpha;
  a.pmask = map.pmask; a.plog = map.plog; a.chunk = map.chunk; a.ncols = map.ncols;
  if (pending) a.fin = make_fin<WT>(*pending, tiled_fold_count(t, *pending));
  size_t const lds = std::max<size_t>(((size_t)t.T + (size_t)TP_WAVES * TP_STAGE) * sizeof(WT) + 64, 3 * TP_BLOCK * sizeof(double));
  bool const w     = a.weights != nullptr;
  static bool attr_done[4] = {false, false, false, false};
  static int dbg_calls = getenv("CUGRAPH_AMD_TILED_DEBUG") ? 2 : 0;
  auto launch = [&](auto kernel, int slot) {
    if (!attr_done[slot]) { ensure_max_lds(kernel, (int)h.lds_per_block); attr_done[slot] = true; }
    timed_launch tl(h, "pagerank_spmv");
    hipLaunchKernelGGL(kernel, t.n_wg, TP_BLOCK, lds, h.stream, a);
  };
  if (dbg_calls > 0 && !w && sizeof(WT) == 4) {  // instrumented variant: per-workgroup wall time / tile loads to stderr
    --dbg_calls;
    size_t const n = (size_t)t.n_wg * 2;
    dvec<unsigned long long> dbg(n);
    a.dbg = dbg.data();
    ensure_max_lds(k_tiled_phase1<WT, false, true>, (int)h.lds_per_block);
    hipLaunchKernelGGL((k_tiled_phase1<WT, false, true>), t.n_wg, TP_BLOCK, lds, h.stream, a);
    std::vector<unsigned long long> d = to_host(h, dbg.data(), n);
    std::vector<unsigned long long> v;
    double tiles = 0;
    for (int b = 0; b < t.n_wg; ++b) { v.push_back(d[2 * b]); tiles += (double)d[2 * b + 1]; }
    std::sort(v.begin(), v.end());
    fprintf(stderr, "[tiled phase1 dbg] %d workgroups, %d items, %d chunks; wall ticks (100 MHz) min %llu p10 %llu p50 %llu p90 %llu max %llu; tile loads per workgroup %.1f\n",
            t.n_wg, t.n_items, t.n_chunks, v[0], v[t.n_wg / 10], v[t.n_wg / 2], v[t.n_wg * 9 / 10], v[t.n_wg - 1], tiles / t.n_wg);
    return;
  }
  if (w) launch(k_tiled_phase1<WT, true>, 0); else launch(k_tiled_phase1<WT, false>, 1);
}

template <typename WT>
void tiled_phase2(handle_t const& h, tiled_csc_t const& t, WT const* part, tiled_epilogue<WT> const& e, uint32_t* counters)
{
  p2_args<WT> a;
  a.part       = part;
  a.dstl16     = t.dstl16.size() ? t.dstl16.data() : nullptr;
  a.dstl12     = t.dstl12.size() ? t.dstl12.data() : nullptr;
  a.tile_row0  = t.tile_row0.data();
  a.region_off = t.region_off.data();
  a.nI         = t.nI;
  a.e          = e;
  a.counters   = counters;
  using ACC = typename p2_acc<WT>::type;
  size_t const lds = (size_t)TP2_ROWS * sizeof(ACC);
  a.tile_wmax = t.tile_wmax.data();
  int grid = t.nI;
  int const n_const = e.cr.nI_act > 0 ? (int)((e.cr.n_cols + TP2_CONST_COLS - 1) / TP2_CONST_COLS) : 0;
  if (e.cr.nI_act > 0) grid = e.cr.nI_act + n_const;
  auto launch = [&](auto kernel) {
    static bool attr_done = false;  // (one flag per instantiation of this lambda = per kernel)
    if (!attr_done && lds > 48 * 1024) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_done = true; }
    timed_launch tl(h, "pagerank_reduce");
    hipLaunchKernelGGL(kernel, grid, TP2_BLOCK, lds, h.stream, a);
  };
  char const* const stag_env = getenv("CUGRAPH_AMD_P2_STAGGER");  // (read per launch: tools/variant_ab.py alternates it on one plan)
  a.stagger = stag_env ? std::max(0, std::min(64, atoi(stag_env))) : (grid >= TP2_STAGGER_MIN_GRID ? 6 : 0);
  if (e.pers != nullptr) launch(k_tiled_phase2<WT, true>); else launch(k_tiled_phase2<WT, false>);
}

template <typename WT>
void tiled_finish(handle_t const& h, tiled_epilogue<WT> const& e, int n_partials, double init_prev, hipStream_t stream)
{
  fin_args<WT> f = make_fin<WT>(e, n_partials);
  f.init_prev    = init_prev;
  hipLaunchKernelGGL(k_tiled_finish<WT>, 1, TP2_BLOCK, 0, stream ? stream : h.stream, f);
}

template <typename WT>
int tiled_prologue(handle_t const& h, tiled_csc_t const& t, WT const* pr, WT const* outw, WT* x, int64_t nv, double* partials, int32_t const* xcol_override)
{
  int const grid = std::max(1, std::min(t.nI, grid_for(nv, 256, 1024)));
  int32_t const* xc = xcol_override ? xcol_override : (t.xcol.size() ? (int32_t const*)t.xcol.data() : (int32_t const*)nullptr);
  hipLaunchKernelGGL(k_tiled_prologue<WT>, grid, 256, 0, h.stream, pr, outw, xc, x, nv, partials);
  return grid;
}

template <typename WT>
void tiled_scalars_from_ranks(handle_t const& h, tiled_epilogue<WT> const& e, void const* recv, size_t first_off, size_t stride_bytes, int nranks)
{
  hipLaunchKernelGGL(k_tiled_scalars_from_ranks<WT>, 1, 64, 0, h.stream, static_cast<unsigned char const*>(recv), first_off, stride_bytes, nranks,
                     make_fin<WT>(e, 0));
}

#define CGA_INSTANTIATE_TILED(WT)                                                                                                          \
  template void tiled_phase1<WT>(handle_t const&, tiled_csc_t const&, WT const*, WT, WT*, uint32_t*, tiled_x_map<WT> const&, tiled_epilogue<WT> const*); \
  template void tiled_phase2<WT>(handle_t const&, tiled_csc_t const&, WT const*, tiled_epilogue<WT> const&, uint32_t*);                                  \
  template void tiled_finish<WT>(handle_t const&, tiled_epilogue<WT> const&, int, double, hipStream_t);                                                           \
  template int tiled_prologue<WT>(handle_t const&, tiled_csc_t const&, WT const*, WT const*, WT*, int64_t, double*, int32_t const*);          \
  template void tiled_scalars_from_ranks<WT>(handle_t const&, tiled_epilogue<WT> const&, void const*, size_t, size_t, int);
CGA_INSTANTIATE_TILED(float)
CGA_INSTANTIATE_TILED(double)

}  // namespace cga
