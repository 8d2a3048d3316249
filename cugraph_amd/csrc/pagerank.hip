// PageRank for gfx950: one fused pull-SpMV kernel per power iteration.
//
// Replaces:
//   cugraph_pagerank* C API                       cpp/src/c_api/pagerank.cpp:247, :316, :378, :466
//   cugraph::pagerank / detail::pagerank          cpp/src/link_analysis/pagerank_impl.cuh:406-447, 39-330
//   per_v_transform_reduce_incoming_e (4 kernels) cpp/include/cugraph/prims/detail/per_v_transform_reduce_e.cuh:252/389/500/688
//   update_edge_src_property (SG copy)            cpp/include/cugraph/prims/update_edge_src_dst_property.cuh:656-659
//   transform_reduce_v x2 + transform + copy      pagerank_impl.cuh:225-251, 311-318
//   compute_out_degrees / compute_out_weight_sums cpp/src/structure/graph_view_impl.cuh:207-238
//
// The reference runs, per iteration, 4 V-length passes + a V-length copy + up to 4 SpMV kernels on 4
// streams + 2 host-synchronising scalar reductions.  Here an iteration is ONE persistent kernel + a
// 1-block finisher, with no host synchronisation unless epsilon > 0:
//
//   k_spmv:  for every destination row (CSC, degree-descending schedule)
//              y      = base + sum_{in-edges} x[src] * [w *] alpha         (x = pr / out_w of the previous iterate)
//              pr[v]  = y (+ personalization term)                          -> |y - pr_old| accumulated (L1 change)
//              x'[v]  = y / (out_w[v] == 0 ? 1 : out_w[v])                  -> next iteration's gather vector
//              dangling' += out_w[v] == 0 ? y : 0
//            Row classes by in-degree: >= 4096 one workgroup per row, 64..4095 one wavefront per row,
//            16..63 / 4..15 / 0..3 sixteen / four / one lane(s) per row, so lanes of a wave always read
//            consecutive index words (coalesced HBM stream) and reduce with wave64 cross-lane shuffles.
//            The first `hot` entries of x (the highest-degree vertices after renumbering: >50 % of all
//            gathers on RMAT) are staged in LDS once per workgroup; only colder sources go to L2/MALL.
//   k_finish: fixed-order fp64 reduction of the per-workgroup (diff, dangling) partials -> next `base`.
//
// Loop order and stopping rule follow pagerank_impl.cuh:224-329 exactly (see oracle/oracle.c).
#include "common.hpp"
#include "spmv_tiled.hpp"
#include "comm.hpp"
#include "mg_graph.hpp"
#include "wave_ops.hpp"

#include <numeric>
#include <type_traits>

namespace cga {

namespace {

constexpr int PR_BLOCK = 1024;  // 16 wavefronts
constexpr int PR_WAVES = PR_BLOCK / 64;

template <typename WT>
struct spmv_args {
  int32_t const* offsets;
  int32_t const* indices;
  WT const* weights;      // or nullptr
  int32_t const* row_order;  // or nullptr (identity)
  int64_t nv;
  int64_t seg0, seg1, seg2, seg3;  // schedule positions: [0,seg0) WG/row, [seg0,seg1) wave/row, [seg1,seg2) 16 lanes, [seg2,seg3) 4 lanes, rest 1 lane
  WT const* x;            // gather vector (previous iterate / out_w)
  WT* x_next;
  WT* pr;                 // in: previous iterate, out: new iterate
  WT const* outw;
  WT const* pers;         // dense normalised personalization or nullptr
  pr_scalars<WT> const* scal;
  double* partials;       // [gridDim.x][2] = (diff, dangling)
  WT alpha;
  int hot;                // entries of x staged in LDS
};

template <typename WT, bool WEIGHTED>
struct gatherer {
  spmv_args<WT> const& a;
  WT const* xs;  // LDS tile
  __device__ __forceinline__ WT val(int32_t i) const { return i < a.hot ? xs[i] : a.x[i]; }
  // sum over edges e = first, first + stride, ... < end
  __device__ __forceinline__ WT strided(uint32_t first, uint32_t end, uint32_t stride) const
  {
    WT acc = 0;
    uint32_t e = first;  // unsigned: e + 3 * stride stays below 2^32 for any INT32 edge count
    for (; e + 3 * stride < end; e += 4 * stride) {
      int32_t i0 = a.indices[e], i1 = a.indices[e + stride], i2 = a.indices[e + 2 * stride], i3 = a.indices[e + 3 * stride];
      WT v0 = val(i0), v1 = val(i1), v2 = val(i2), v3 = val(i3);
      if constexpr (WEIGHTED) {
        v0 *= a.weights[e]; v1 *= a.weights[e + stride]; v2 *= a.weights[e + 2 * stride]; v3 *= a.weights[e + 3 * stride];
      }
      acc = fma(v0, a.alpha, acc); acc = fma(v1, a.alpha, acc); acc = fma(v2, a.alpha, acc); acc = fma(v3, a.alpha, acc);
    }
    for (; e < end; e += stride) {
      WT v = val(a.indices[e]);
      if constexpr (WEIGHTED) v *= a.weights[e];
      acc = fma(v, a.alpha, acc);
    }
    return acc;
  }
};

template <typename WT, bool PERS>
__device__ __forceinline__ void row_epilogue(spmv_args<WT> const& a, pr_scalars<WT> const& sc, int32_t v, WT sum, double& diff,
                                             double& dang)
{
  WT y = sc.base + sum;
  if constexpr (PERS) y += sc.pers_factor * a.pers[v];
  WT old = a.pr[v];
  WT ow  = a.outw[v];
  a.pr[v]     = y;
  a.x_next[v] = y / (ow == WT(0) ? WT(1) : ow);
  diff += (double)fabs(y - old);
  if (ow == WT(0)) dang += (double)y;
}

template <typename WT, bool WEIGHTED, bool PERS>
__global__ void __launch_bounds__(PR_BLOCK) k_spmv(spmv_args<WT> a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WT* xs       = reinterpret_cast<WT*>(smem);
  double* wred = reinterpret_cast<double*>(smem + (size_t)a.hot * sizeof(WT));  // [PR_WAVES][2] + row partials [PR_WAVES]
  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int i = tid; i < a.hot; i += PR_BLOCK) xs[i] = a.x[i];
  __syncthreads();

  pr_scalars<WT> const sc = *a.scal;
  gatherer<WT, WEIGHTED> g{a, xs};
  double diff = 0.0, dang = 0.0;

  // ---- class H: one workgroup per row
  {
    WT* rowp = reinterpret_cast<WT*>(wred + 2 * PR_WAVES);
    for (int64_t r = blockIdx.x; r < a.seg0; r += gridDim.x) {
      int32_t v = a.row_order ? a.row_order[r] : (int32_t)r;
      int32_t b = a.offsets[v], e = a.offsets[v + 1];
      WT s = group_sum(g.strided(b + tid, e, PR_BLOCK), 64);
      if (lane == 0) rowp[wave] = s;
      __syncthreads();
      if (tid == 0) {
        WT t = 0;
#pragma unroll
        for (int k = 0; k < PR_WAVES; ++k) t += rowp[k];
        row_epilogue<WT, PERS>(a, sc, v, t, diff, dang);
      }
      __syncthreads();
    }
  }
  int64_t const gwave  = (int64_t)blockIdx.x * PR_WAVES + wave;
  int64_t const nwaves = (int64_t)gridDim.x * PR_WAVES;
  // ---- class M: one wavefront per row
  for (int64_t r = a.seg0 + gwave; r < a.seg1; r += nwaves) {
    int32_t v = a.row_order ? a.row_order[r] : (int32_t)r;
    int32_t b = a.offsets[v], e = a.offsets[v + 1];
    WT s = group_sum(g.strided(b + lane, e, 64), 64);
    if (lane == 0) row_epilogue<WT, PERS>(a, sc, v, s, diff, dang);
  }
  // ---- class L16: 16 lanes per row
  for (int64_t r = a.seg1 + gwave * 4 + (lane >> 4); r < a.seg2; r += nwaves * 4) {
    int32_t v = a.row_order ? a.row_order[r] : (int32_t)r;
    int32_t b = a.offsets[v], e = a.offsets[v + 1];
    WT s = group_sum(g.strided(b + (lane & 15), e, 16), 16);
    if ((lane & 15) == 0) row_epilogue<WT, PERS>(a, sc, v, s, diff, dang);
  }
  // ---- class L4: 4 lanes per row
  for (int64_t r = a.seg2 + gwave * 16 + (lane >> 2); r < a.seg3; r += nwaves * 16) {
    int32_t v = a.row_order ? a.row_order[r] : (int32_t)r;
    int32_t b = a.offsets[v], e = a.offsets[v + 1];
    WT s = group_sum(g.strided(b + (lane & 3), e, 4), 4);
    if ((lane & 3) == 0) row_epilogue<WT, PERS>(a, sc, v, s, diff, dang);
  }
  // ---- class L1: one lane per row (degree 0..3)
  for (int64_t r = a.seg3 + gwave * 64 + lane; r < a.nv; r += nwaves * 64) {
    int32_t v = a.row_order ? a.row_order[r] : (int32_t)r;
    int32_t b = a.offsets[v], e = a.offsets[v + 1];
    WT s = 0;
    for (int32_t p = b; p < e; ++p) {
      WT t = g.val(a.indices[p]);
      if constexpr (WEIGHTED) t *= a.weights[p];
      s = fma(t, a.alpha, s);
    }
    row_epilogue<WT, PERS>(a, sc, v, s, diff, dang);
  }

  // ---- fixed-order block reduction of the two scalars
  diff = group_sum(diff, 64);
  dang = group_sum(dang, 64);
  __syncthreads();
  if (lane == 0) { wred[2 * wave] = diff; wred[2 * wave + 1] = dang; }
  __syncthreads();
  if (tid == 0) {
    double d0 = 0, d1 = 0;
#pragma unroll
    for (int k = 0; k < PR_WAVES; ++k) { d0 += wred[2 * k]; d1 += wred[2 * k + 1]; }
    a.partials[2 * blockIdx.x]     = d0;
    a.partials[2 * blockIdx.x + 1] = d1;
  }
}


// =================================================================================================
// Edge-balanced ("flat") SpMV: y[row] = sum_{in-edges} x[src] * [w *] alpha for rows [0, n_nonempty).
//
// Every wavefront owns a contiguous range of the CSC edge array (same number of edges for every wave,
// whatever the degree distribution) and streams it with 16-byte index loads: lane l of a wave reads edges
// [base + 256 g + 4 l, +4) for g = 0..3, i.e. each load instruction covers 1 KiB of consecutive addresses.
// Row boundaries come from a bitmap (bit e = "edge e starts a row", 1 bit per edge, 3 % of the index
// bytes) instead of per-row offset loads, so there is no load -> load dependency per row: the only
// dependent chain is index -> gather.  Sums are formed by an in-lane pass over the lane's 4 edges and a
// wave64 segmented scan (shuffles) across lanes; a completed row is stored by the lane that sees the next
// row start.  The two partial rows at the ends of a wave's range go to side arrays and are stitched by
// k_flat_fixup in a fixed order, so the result is deterministic (no floating-point atomics).
// Requires: non-empty rows are exactly ids [0, n_nonempty) in edge order -- true for the degree-sorted
// numbering.  Other graphs use k_spmv above.
// =================================================================================================

constexpr int FL_BLOCK = 1024;
constexpr int FL_WAVES = FL_BLOCK / 64;
constexpr int FL_SUB   = 256;           // edges per wave per sub-chunk (4 per lane)
constexpr int FL_UNROLL = 4;
constexpr int FL_CHUNK = FL_SUB * FL_UNROLL;

template <typename WT>
struct flat_args {
  int32_t const* indices;
  WT const* weights;          // or nullptr
  uint8_t const* bits;        // row-start bitmap, byte view
  uint32_t const* wave_rank;  // [n_waves + 1]: number of row starts in [0, wave range start)
  int64_t ne;
  int64_t range_len;          // edges per wave, multiple of FL_CHUNK
  WT const* x;
  WT* y;                      // [n_nonempty]
  WT* head_sum;               // [n_waves] partial of the row running at the wave's range start
  uint8_t* has_flag;          // [n_waves] wave range contains a row start
  WT alpha;
  int hot;
  // multi-GPU: column id c = local_index * P + rank (global degree order); the all-gathered x is blocked by rank:
  // address = (c & pmask) * chunk + (c >> plog).  Single GPU: identity.
  uint32_t pmask, plog, chunk;
};

template <typename WT, bool WEIGHTED, bool MG = false>
__global__ void __launch_bounds__(FL_BLOCK, (sizeof(WT) == 4 && !WEIGHTED) ? 8 : 4) k_spmv_flat(flat_args<WT> a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WT* xs        = reinterpret_cast<WT*>(smem);
  int const tid = threadIdx.x, lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: range bounds live in SGPRs
  {  // stage the hot tile with 16-byte loads (a.hot is a multiple of 4; x is 16-byte aligned)
    using vec4 = typename std::conditional<sizeof(WT) == 4, float4, double4>::type;
    vec4 const* src = reinterpret_cast<vec4 const*>(a.x);
    vec4* dst       = reinterpret_cast<vec4*>(xs);
    if constexpr (MG) {
      for (int i = tid; i < a.hot; i += FL_BLOCK) xs[i] = a.x[((uint32_t)i & a.pmask) * a.chunk + ((uint32_t)i >> a.plog)];
    } else {
      for (int i = tid; i < a.hot / 4; i += FL_BLOCK) dst[i] = src[i];
    }
    if (tid == 0 && a.hot == 0) xs[0] = WT(0);
  }
  __syncthreads();

  int64_t const w  = (int64_t)blockIdx.x * FL_WAVES + wave;
  int64_t const es = w * a.range_len;
  if (es >= a.ne) {
    if (lane == 0) { a.head_sum[w] = WT(0); a.has_flag[w] = 0; }
    return;
  }
  int64_t const ee = min(es + a.range_len, a.ne);
  // row closed by the n-th row start met in this range is row (rank - 1 + n); n = 0 is the partial head
  int64_t const row0 = (int64_t)a.wave_rank[w] - 1;
  int32_t const hot_last = a.hot > 0 ? a.hot - 1 : 0;
  char const* const xbytes = reinterpret_cast<char const*>(a.x);
  uint32_t closed = 0;  // row starts met so far (wave-uniform)
  WT carry        = 0;  // running sum of the row open at the current position (wave-uniform)

  for (int64_t base = es; base < ee; base += FL_CHUNK) {
    int4 id[FL_UNROLL];
    uint32_t fl[FL_UNROLL];
    WT v[FL_UNROLL][4];
    // indices / bitmap / weights are over-allocated by kEdgePad (zero filled), so every load below is in bounds
    // and UNCONDITIONAL: all four 1 KiB index loads and the bitmap bytes are in flight together.
#pragma unroll
    for (int g = 0; g < FL_UNROLL; ++g) {
      int64_t e = base + g * FL_SUB + 4 * lane;
      id[g]     = *reinterpret_cast<int4 const*>(a.indices + e);
      fl[g]     = ((uint32_t)a.bits[e >> 3] >> (e & 4)) & 0xFu;
    }
#pragma unroll
    for (int g = 0; g < FL_UNROLL; ++g) {
      int64_t e    = base + g * FL_SUB + 4 * lane;
      int32_t i[4] = {id[g].x, id[g].y, id[g].z, id[g].w};
      WT cold[4], hotv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // cold sources: exec-masked global gather (no flat loads, no waits inside)
        cold[k] = WT(0);
        // 32-bit byte offset from a wave-uniform base: one VGPR per address (SGPR base + VGPR offset form)
        if (i[k] >= a.hot) {
          uint32_t c = (uint32_t)i[k];
          if constexpr (MG) c = (c & a.pmask) * a.chunk + (c >> a.plog);
          cold[k] = *reinterpret_cast<WT const*>(xbytes + c * (uint32_t)sizeof(WT));
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) hotv[k] = xs[min(i[k], hot_last)];  // LDS read with a clamped (always valid) index
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        WT t = i[k] < a.hot ? hotv[k] : cold[k];
        if constexpr (WEIGHTED) t *= a.weights[e + k];
        v[g][k] = (e + k < ee) ? t * a.alpha : WT(0);
      }
      if (e + 3 >= ee) {  // bits past the range end belong to the next wave
        uint32_t keep = e >= ee ? 0u : (1u << (ee - e)) - 1u;
        fl[g] &= keep;
      }
    }
#pragma unroll
    for (int g = 0; g < FL_UNROLL; ++g) {
      uint32_t const f = fl[g];
      if (__ballot(f != 0) == 0) {  // no row starts in these 256 edges (inside a long row): plain wave sum
        WT t = wave_sum_to_lane63((v[g][0] + v[g][1]) + (v[g][2] + v[g][3]));
        carry += read_lane63(t);
        continue;
      }
      // in-lane: r_k = running sum since the last row start at or before element k
      WT const r0 = v[g][0];
      WT const r1 = (f & 2u) ? v[g][1] : r0 + v[g][1];
      WT const r2 = (f & 4u) ? v[g][2] : r1 + v[g][2];
      WT const r3 = (f & 8u) ? v[g][3] : r2 + v[g][3];
      // wave64 segmented scan of (tail = r3, number of row starts in the lane)
      uint32_t const nf = __popc(f);
      WT s       = r3;
      uint32_t c = nf;
      wave_seg_scan(s, c);
      WT ex_s         = dpp_val<0x138, 0xF>(s);   // wave_shr:1 (lane 0 reads 0)
      uint32_t ex_c   = dpp_u32<0x138, 0xF>(c);
      WT const carry_in = ex_c ? ex_s : ex_s + carry;
      if (f) {  // emit the rows that end inside this lane: the row closed by a start at element k ran up to element k-1
        uint32_t ord = closed + ex_c;  // ordinal of this lane's first row start within the wave's range
        WT before[4] = {WT(0), r0, r1, r2};
        bool first   = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((f >> k) & 1u) {
            WT total = first ? carry_in + before[k] : before[k];
            first    = false;
            if (ord == 0) a.head_sum[w] = total;  // the row running at the range start: partial
            else a.y[row0 + ord] = total;
            ++ord;
          }
        }
      }
      uint32_t const c_last = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
      WT const s_last       = read_lane63(s);
      carry                 = c_last ? s_last : s_last + carry;
      closed += c_last;
    }
  }
  if (lane == 0) {
    if (closed == 0) { a.head_sum[w] = carry; a.has_flag[w] = 0; }
    else { a.y[row0 + closed] = carry; a.has_flag[w] = 1; }  // row opened here (may continue in later waves)
  }
}

// Adds to the row left open at the end of wave w's range the heads of the following waves.
template <typename WT>
__global__ void k_flat_fixup(uint32_t const* wave_rank, WT const* head_sum, uint8_t const* has_flag, int64_t n_waves, WT* y)
{
  int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (w >= n_waves || !has_flag[w]) return;
  WT acc = 0;
  bool any = false;
  for (int64_t k = w + 1; k < n_waves; ++k) {
    acc += head_sum[k];
    any = true;
    if (has_flag[k]) break;
  }
  if (any) y[(int64_t)wave_rank[w + 1] - 1] += acc;
}

// wave_rank[w] = number of rows whose first edge lies before w * range_len (rows [0, n_nonempty) are non-empty,
// so offsets is strictly increasing there)
__global__ void k_wave_ranks(int32_t const* offsets, int64_t n_nonempty, int64_t range_len, int64_t n_waves, uint32_t* wave_rank)
{
  int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (w > n_waves) return;
  int64_t es = w * range_len;
  int64_t lo = 0, hi = n_nonempty;  // first row with offsets[row] >= es
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)offsets[mid] < es) lo = mid + 1; else hi = mid;
  }
  wave_rank[w] = (uint32_t)lo;
}

__global__ void k_rowstart_bits(int32_t const* offsets, int64_t n_nonempty, uint32_t* bits)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_nonempty; i += stride) {
    uint32_t e = (uint32_t)offsets[i];
    atomicOr(&bits[e >> 5], 1u << (e & 31));
  }
}

// V-length epilogue of the flat path: pr <- base + y (+ personalization), x' <- pr / out_w, L1 change and
// dangling mass partials (same arithmetic as row_epilogue above).
template <typename WT, bool PERS>
__global__ void __launch_bounds__(256) k_pr_epilogue(WT const* y, int64_t n_nonempty, int64_t nv, WT* pr, WT* x_next, WT const* outw,
                                                     WT const* pers, pr_scalars<WT> const* scal, double* partials)
{
  __shared__ double red[8];
  pr_scalars<WT> const sc = *scal;
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double diff = 0.0, dang = 0.0;
  for (; i < nv; i += stride) {
    WT sum = i < n_nonempty ? y[i] : WT(0);
    WT val = sc.base + sum;
    if constexpr (PERS) val += sc.pers_factor * pers[i];
    WT old = pr[i], ow = outw[i];
    pr[i]     = val;
    x_next[i] = val / (ow == WT(0) ? WT(1) : ow);
    diff += (double)fabs(val - old);
    if (ow == WT(0)) dang += (double)val;
  }
  diff = group_sum(diff, 64);
  dang = group_sum(dang, 64);
  int const wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[2 * wv] = diff; red[2 * wv + 1] = dang; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x]     = red[0] + red[2] + red[4] + red[6];
    partials[2 * blockIdx.x + 1] = red[1] + red[3] + red[5] + red[7];
  }
}

// x = pr / (outw == 0 ? 1 : outw); partial dangling sums (iteration 0 state)
template <typename WT>
__global__ void __launch_bounds__(256) k_prologue(WT const* pr, WT const* outw, WT* x, int64_t nv, double* partials)
{
  __shared__ double red[4];
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double dang    = 0.0;
  for (; i < nv; i += stride) {
    WT p = pr[i], ow = outw[i];
    x[i] = p / (ow == WT(0) ? WT(1) : ow);
    if (ow == WT(0)) dang += (double)p;
  }
  dang = group_sum(dang, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dang;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x]     = 0.0;
    partials[2 * blockIdx.x + 1] = red[0] + red[1] + red[2] + red[3];
  }
}

template <typename WT>
__global__ void __launch_bounds__(256) k_finish(double const* partials, int n, pr_scalars<WT>* scal, WT alpha, WT one_minus_alpha,
                                                int64_t nv, int personalized)
{
  __shared__ double r0[256], r1[256];
  double d0 = 0, d1 = 0;
  for (int i = threadIdx.x; i < n; i += 256) { d0 += partials[2 * i]; d1 += partials[2 * i + 1]; }
  r0[threadIdx.x] = d0; r1[threadIdx.x] = d1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    WT dangling       = (WT)r1[0];
    WT factor         = dangling * alpha + one_minus_alpha;
    scal->dangling    = dangling;
    scal->diff        = (WT)r0[0];
    scal->pers_factor = factor;
    scal->base        = personalized ? WT(0) : factor / (WT)nv;
  }
}

// ---- out-weight sums --------------------------------------------------------------------------
template <typename WT>
__global__ void k_outdeg_from_offsets(int32_t const* offsets, int64_t nv, WT* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nv; i += stride) out[i] = (WT)(offsets[i + 1] - offsets[i]);
}
template <typename WT>
__global__ void k_u32_to_wt(uint32_t const* c, int64_t nv, WT* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nv; i += stride) out[i] = (WT)c[i];
}
template <typename WT>
__global__ void k_rowsum_weights(int32_t const* offsets, WT const* w, int64_t nv, WT* out)
{
  int64_t wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int lane       = threadIdx.x & 63;
  for (int64_t v = wave; v < nv; v += nwaves) {
    double s = 0;
    for (int32_t p = offsets[v] + lane; p < offsets[v + 1]; p += 64) s += (double)w[p];
    s = group_sum(s, 64);
    if (lane == 0) out[v] = (WT)s;
  }
}
template <typename WT>
__global__ void k_scatter_add_weights(int32_t const* indices, WT const* w, int64_t ne, double* acc)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < ne; i += stride) atomicAdd(&acc[indices[i]], (double)w[i]);
}
template <typename WT>
__global__ void k_f64_to_wt(double const* c, int64_t nv, WT* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nv; i += stride) out[i] = (WT)c[i];
}

// dense[ids[i]] (+)= vals[i]
template <typename WT, bool ADD>
__global__ void k_scatter_pairs(int32_t const* ids, WT const* vals, int64_t n, WT* dense, WT scale)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    if constexpr (ADD) atomicAdd(&dense[ids[i]], vals[i] * scale);
    else dense[ids[i]] = vals[i];
  }
}

template <typename WT>
__global__ void k_sum(WT const* v, int64_t n, double* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double s       = 0;
  for (; i < n; i += stride) s += (double)v[i];
  s = group_sum(s, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

template <typename WT> void fill_wt(handle_t const& h, WT* p, int64_t n, WT v);
template <> void fill_wt<float>(handle_t const& h, float* p, int64_t n, float v) { fill_f32(h, p, n, v); }
template <> void fill_wt<double>(handle_t const& h, double* p, int64_t n, double v) { fill_f64(h, p, n, v); }

// rows [n_act, n_act + n) have no in-edge: out-weight sums of their live columns in column order, how many of them are dangling,
// max 1 / out-weight (red[0] = count, red[1] = bits of the max, a non-negative double)
// flags[i] = v[i] != 0 for i < n, flags[n] = 0
template <typename WT>
__global__ void k_nonzero_flags(WT const* v, int64_t n, uint32_t* flags)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i <= n; i += stride) flags[i] = (i < n && v[i] != WT(0)) ? 1u : 0u;
}

template <typename WT>
__global__ void k_const_rows_setup(WT const* outw, int32_t const* xcol, int64_t n, int64_t n_act, int64_t c0, WT* outw_c, unsigned long long* red)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long dangling = 0;
  double best = 0.0;
  for (; i < n; i += stride) {
    WT const ow     = outw[i];
    int64_t const c = xcol ? (int64_t)xcol[i] : n_act + i;
    if (c >= 0) outw_c[c - c0] = ow;
    dangling += ow == WT(0) ? 1ull : 0ull;
    best = fmax(best, 1.0 / (double)(ow == WT(0) ? WT(1) : ow));
  }
  for (int o = 32; o > 0; o >>= 1) { dangling += __shfl_xor(dangling, o); best = fmax(best, __shfl_xor(best, o)); }
  if ((threadIdx.x & 63) == 0) {
    if (dangling) atomicAdd(&red[0], dangling);
    atomicMax(&red[1], (unsigned long long)__double_as_longlong(best));
  }
}

template <typename WT>
__global__ void k_fill_from_base_prev(WT* out, int64_t n, pr_scalars<WT> const* scal)
{
  WT const v     = scal->base_prev;
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = v;
}

}  // namespace

// ----------------------------------------------------------------------------------------- plan
struct pagerank_plan_base {
  virtual ~pagerank_plan_base() = default;
  virtual void step(double epsilon, size_t max_iterations, size_t* done, bool* converged) = 0;
  virtual centrality_result_t* result(size_t total_iterations, bool converged)          = 0;
  // cugraph_amd_pagerank_plan_tune: time the plan on `placements` placements of its streamed arrays, keep the fastest; the milliseconds per iteration of
  // the kept one (0: this kind of plan has nothing to tune)
  virtual double tune(int /*placements*/) { return 0.0; }
};

template <typename WT>
struct pagerank_plan : pagerank_plan_base {
  handle_t const& h;
  graph_t& g;
  WT alpha;
  bool personalized{false};
  dvec<WT> pr, x0, x1, outw_own, pers;
  WT const* outw{nullptr};
  dvec<pr_scalars<WT>> scal;
  dvec<double> partials;
  int grid{0};
  int hot{0};
  size_t lds_bytes{0};
  int cur{0};
  // edge-balanced path
  bool flat{false};
  dvec<WT> yv, head_sum;
  dvec<uint8_t> has_flag;
  dvec<uint32_t> wave_rank;
  int64_t range_len{0}, n_waves{0};
  int flat_grid{0}, epi_grid{0};
  size_t flat_lds{0};
  // column-tiled two-phase path (spmv_tiled.hpp): the default
  bool tiled{false};
  bool const force_diff{getenv("CUGRAPH_AMD_PAGERANK_DIFF") != nullptr};          // measurement switches, read once per plan
  bool const force_write_pr{getenv("CUGRAPH_AMD_PAGERANK_WRITE_PR") != nullptr};
  std::shared_ptr<tiled_csc_t> tc;
  dvec<WT> part;
  dvec<double> tpartials;  // [max(nI, 1024)][3]
  dvec<uint32_t> counters;

  pagerank_plan(handle_t const& h_, graph_t& g_, double alpha_) : h(h_), g(g_), alpha((WT)alpha_) {}

  // (ext ids, values) -> dense vector with `fill` elsewhere; INVALID_INPUT on ids that are not vertices
  void pairs_to_dense(device_array_view_t const* ids, device_array_view_t const* vals, WT* dense, WT fill, char const* what)
  {
    CGA_EXPECTS(ids->size == vals->size, CUGRAPH_INVALID_INPUT, std::string(what) + ": vertices and values differ in size");
    fill_wt<WT>(h, dense, g.nv, fill);
    if (ids->size == 0) return;
    dvec<int32_t> tmp(ids->size);
    HIP_TRY(hipMemcpyAsync(tmp.data(), ids->data, ids->size * 4, hipMemcpyDeviceToDevice, h.stream));
    renumber_ext_to_int(h, g, tmp.data(), (int64_t)ids->size);
    CGA_EXPECTS(count_negative_i32(h, tmp.data(), (int64_t)ids->size) == 0, CUGRAPH_INVALID_INPUT,
                std::string(what) + ": found a vertex id that is not in the graph");
    hipLaunchKernelGGL((k_scatter_pairs<WT, false>), grid_for(ids->size, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)tmp.data(),
                       vals->as<WT const>(), (int64_t)ids->size, dense, WT(1));
    h.sync();
  }

  void compute_out_weight_sums()
  {
    if (!g.out_weight_sums_valid) {
      g.out_weight_sums.alloc((size_t)(g.nv > 0 ? g.nv : 1) * sizeof(WT));
      WT* ow = g.out_weight_sums.as<WT>();
      if (g.nv > 0) {
        if (!g.has_weights) {
          if (g.csr.built && !g.csr.dcs.active()) {
            hipLaunchKernelGGL(k_outdeg_from_offsets<WT>, grid_for(g.nv, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)g.csr.offsets.data(), g.nv, ow);
          } else {
            dvec<uint32_t> c(g.nv);
            HIP_TRY(hipMemsetAsync(c.data(), 0, g.nv * 4, h.stream));
            histogram_i32(h, g.csc.indices.data(), g.ne, c.data(), g.nv);
            hipLaunchKernelGGL(k_u32_to_wt<WT>, grid_for(g.nv, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)c.data(), g.nv, ow);
            h.sync();
          }
        } else if (g.csr.built && !g.csr.dcs.active()) {
          hipLaunchKernelGGL(k_rowsum_weights<WT>, grid_for(g.nv * 16, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)g.csr.offsets.data(),
                             g.csr.weights.as<WT const>(), g.nv, ow);
        } else {
          dvec<double> acc(g.nv);
          HIP_TRY(hipMemsetAsync(acc.data(), 0, g.nv * 8, h.stream));
          if (g.ne > 0)
            hipLaunchKernelGGL(k_scatter_add_weights<WT>, grid_for(g.ne, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)g.csc.indices.data(),
                               g.csc.weights.as<WT const>(), g.ne, acc.data());
          hipLaunchKernelGGL(k_f64_to_wt<WT>, grid_for(g.nv, kBlock, 4096), kBlock, 0, h.stream, (double const*)acc.data(), g.nv, ow);
          h.sync();
        }
      }
      g.out_weight_sums_valid = true;
    }
    outw = g.out_weight_sums.as<WT const>();
  }

  // The edge-balanced kernel applies when ids are degree-sorted (non-empty rows = id prefix in edge order).
  void setup_flat()
  {
    orientation_t& o = g.csc;
    char const* env  = getenv("CUGRAPH_AMD_PAGERANK_KERNEL");  // "flat" / "rows" select the single-pass kernels (testing / profiling)
    std::string const kern = env ? env : "tiled";
    tiled = g.ne > 0 && kern != "flat" && kern != "rows";
    CGA_EXPECTS(tiled || g.ne <= kMaxSignedEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "PageRank: the single-pass kernels address edges with signed 32-bit positions (fewer than 2^31 edges)");
    if (tiled) {
      int const T = tiled_default_T(h, sizeof(WT), g.nv);
      if (!o.tiled || o.tiled->T != T || getenv("CUGRAPH_AMD_TILED_REBUILD")) {  // (the env: parameter sweeps on one graph, tools/plan_sweep.py)
        auto t = std::make_shared<tiled_csc_t>();
        try {
          bool const compact = true;  // sources without out-edges get no column (spmv_tiled.hpp: xcol)
          dvec<uint32_t> live;  // unweighted graph: a vertex has an out-edge <=> its out-degree (= out-weight sum, cached on the graph) is non-zero
          if (compact && !g.has_weights) {
            compute_out_weight_sums();
            live.resize_discard((size_t)g.nv + 1);
            hipLaunchKernelGGL(k_nonzero_flags<WT>, grid_for(g.nv + 1, kBlock, 4096), kBlock, 0, h.stream, g.out_weight_sums.template as<WT const>(), g.nv, live.data());
          }
          build_tiled_csc(h, g.nv, g.nv, g.ne, o, g.has_weights, sizeof(WT), T, *t, compact, live.size() ? live.data() : nullptr);
          o.tiled = t;
        } catch (api_error const& e) {
          // the re-blocked arrays are addressed with 32-bit byte offsets; a graph that outgrows them (or the memory for
          // the build temporaries) still gets an answer from the single-pass kernels
          if (g.ne > kMaxSignedEdges) throw;  // (the single-pass kernels keep signed 32-bit edge positions: no answer from them here)
          if (e.code == CUGRAPH_ALLOC_ERROR || std::string(e.what()).find("tiled SpMV") != std::string::npos) tiled = false;
          else throw;
        }
      }
    }
    if (tiled) {
      tc = o.tiled;
      part.resize_discard((size_t)tc->n_slots + 64);
      HIP_TRY(hipMemsetAsync(part.data(), 0, ((size_t)tc->n_slots + 64) * sizeof(WT), h.stream));
      tpartials.resize_discard((size_t)3 * std::max(tc->nI, 1024));
      counters.resize_discard(4);
      HIP_TRY(hipMemsetAsync(counters.data(), 0, 4 * sizeof(uint32_t), h.stream));
      // the gather vectors are read tile-wise: pad them to whole tiles (zero tail)
      size_t const nx = (size_t)tc->nJ * tc->T + 8;
      x0.resize_discard(nx); x1.resize_discard(nx);
      HIP_TRY(hipMemsetAsync(x0.data(), 0, nx * sizeof(WT), h.stream));
      HIP_TRY(hipMemsetAsync(x1.data(), 0, nx * sizeof(WT), h.stream));
      h.sync();
      if (getenv("CUGRAPH_AMD_TILED_DEBUG"))  // where the streamed arrays sit (run-to-run spread of phase 1: placement?)
        fprintf(stderr, "[tiled plan] src16 %p bits %p wrec %p delta1 %p dstl12 %p part %p x0 %p x1 %p pr %p\n", (void*)tc->src16.data(), (void*)tc->bits.data(),
                (void*)tc->wrec.data(), (void*)tc->delta1.data(), (void*)tc->dstl12.data(), (void*)part.data(), (void*)x0.data(), (void*)x1.data(), (void*)pr.data());
      return;
    }
    inflate_offsets(h, o, g.nv);  // the single-pass kernels read plain offsets
    flat = o.row_order.size() == 0 && g.ne > 0 && kern != "rows";
    if (!flat) return;
    int64_t const nnz_rows = o.seg[4];
    if (o.rowstart_bits.size() == 0) {
      size_t words = (size_t)((g.ne + kEdgePad) / 32 + 2);
      o.rowstart_bits.resize_discard(words);
      HIP_TRY(hipMemsetAsync(o.rowstart_bits.data(), 0, words * 4, h.stream));
      hipLaunchKernelGGL(k_rowstart_bits, grid_for(nnz_rows, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), nnz_rows,
                         o.rowstart_bits.data());
    }
    flat_lds      = (size_t)std::max(hot, 4) * sizeof(WT);
    // fp32 unweighted fits 64 VGPRs (2 workgroups of 16 waves per CU); the other variants get 128 VGPRs, 1 per CU
    int per_cu    = (flat_lds <= 80 * 1024 && sizeof(WT) == 4 && !g.has_weights) ? 2 : 1;
    flat_grid     = h.num_cus * per_cu;
    int64_t waves = (int64_t)flat_grid * FL_WAVES;
    range_len     = ((g.ne + waves - 1) / waves + FL_CHUNK - 1) / FL_CHUNK * FL_CHUNK;
    n_waves       = (g.ne + range_len - 1) / range_len;
    flat_grid     = (int)((n_waves + FL_WAVES - 1) / FL_WAVES);
    n_waves       = (int64_t)flat_grid * FL_WAVES;  // trailing waves own an empty range
    yv.resize_discard((size_t)nnz_rows + 1);
    head_sum.resize_discard((size_t)n_waves);
    has_flag.resize_discard((size_t)n_waves);
    wave_rank.resize_discard((size_t)n_waves + 1);
    hipLaunchKernelGGL(k_wave_ranks, grid_for(n_waves + 1, kBlock), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), nnz_rows, range_len, n_waves,
                       wave_rank.data());
    epi_grid = std::min(grid_for(g.nv, 256, 2048), 2048);
    h.sync();
  }

  template <bool WEIGHTED>
  void launch_flat(WT const* xcur)
  {
    orientation_t const& o = g.csc;
    flat_args<WT> a;
    a.indices   = o.indices.data();
    a.weights   = g.has_weights ? o.weights.as<WT const>() : nullptr;
    a.bits      = reinterpret_cast<uint8_t const*>(o.rowstart_bits.data());
    a.wave_rank = wave_rank.data();
    a.ne        = g.ne;
    a.range_len = range_len;
    a.x         = xcur;
    a.y         = yv.data();
    a.head_sum  = head_sum.data();
    a.has_flag  = has_flag.data();
    a.alpha     = alpha;
    a.hot       = hot;
    timed_launch t(h, "pagerank_spmv");
    hipLaunchKernelGGL((k_spmv_flat<WT, WEIGHTED>), flat_grid, FL_BLOCK, flat_lds, h.stream, a);
  }

  void iterate_flat()
  {
    WT const* xcur = cur == 0 ? x0.data() : x1.data();
    WT* xnext      = cur == 0 ? x1.data() : x0.data();
    if (g.has_weights) launch_flat<true>(xcur); else launch_flat<false>(xcur);
    hipLaunchKernelGGL(k_flat_fixup<WT>, grid_for(n_waves, kBlock), kBlock, 0, h.stream, (uint32_t const*)wave_rank.data(), (WT const*)head_sum.data(),
                       (uint8_t const*)has_flag.data(), n_waves, yv.data());
    if (personalized)
      hipLaunchKernelGGL((k_pr_epilogue<WT, true>), epi_grid, 256, 0, h.stream, (WT const*)yv.data(), g.csc.seg[4], g.nv, pr.data(), xnext, outw,
                         (WT const*)pers.data(), (pr_scalars<WT> const*)scal.data(), partials.data());
    else
      hipLaunchKernelGGL((k_pr_epilogue<WT, false>), epi_grid, 256, 0, h.stream, (WT const*)yv.data(), g.csc.seg[4], g.nv, pr.data(), xnext, outw,
                         (WT const*)nullptr, (pr_scalars<WT> const*)scal.data(), partials.data());
    hipLaunchKernelGGL(k_finish<WT>, 1, 256, 0, h.stream, (double const*)partials.data(), epi_grid, scal.data(), alpha, (WT)(1.0 - (double)alpha),
                       g.nv, personalized ? 1 : 0);
  }

  bool pending_finish{false};  // tiled: phase 2 ran, its scalar partials are not folded into `scal` yet
  // rows without in-edges stay out of the per-iteration epilogue (spmv_tiled.hpp: tiled_const_rows) unless the plan is
  // personalized or starts from a user vector
  tiled_const_rows<WT> crows;
  dvec<WT> outw_c;
  bool const_rows_stale{false};  // pr[n_act ..) has to be filled with the rows' common value before it is read
  void setup_const_rows(bool allowed)
  {
    crows = tiled_const_rows<WT>{};
    if (!allowed || !tiled || getenv("CUGRAPH_AMD_PAGERANK_ALL_ROWS") || tc->n_act >= g.nv || tc->nI_act <= 0) return;
    int64_t const n = g.nv - tc->n_act;
    crows.n_rows = n;
    crows.c0     = tc->c0;
    crows.n_cols = tc->ncols - tc->c0;
    outw_c.resize_discard((size_t)std::max<int64_t>(crows.n_cols, 1));
    dvec<unsigned long long> red(2);
    HIP_TRY(hipMemsetAsync(red.data(), 0, 2 * sizeof(unsigned long long), h.stream));
    hipLaunchKernelGGL(k_const_rows_setup<WT>, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, outw + tc->n_act,
                       tc->xcol.size() ? (int32_t const*)tc->xcol.data() + tc->n_act : (int32_t const*)nullptr, n, (int64_t)tc->n_act, (int64_t)tc->c0, outw_c.data(),
                       red.data());
    unsigned long long r[2];
    h.read_back(r, red.data(), 2);
    crows.n_dangling = (int64_t)r[0];
    std::memcpy(&crows.max_inv_outw, &r[1], sizeof(double));
    crows.outw_c = outw_c.data();
    crows.nI_act = tc->nI_act;
  }
  void materialize_const_rows()
  {
    if (!const_rows_stale) return;
    hipLaunchKernelGGL(k_fill_from_base_prev<WT>, grid_for(crows.n_rows, kBlock, 4096), kBlock, 0, h.stream, pr.data() + tc->n_act, crows.n_rows,
                       (pr_scalars<WT> const*)scal.data());
    const_rows_stale = false;
  }

  tiled_epilogue<WT> tiled_epi(WT* xnext)
  {
    tiled_epilogue<WT> e;
    e.nv = g.nv; e.pr = pr.data(); e.x_next = xnext; e.outw = outw; e.pers = personalized ? pers.data() : nullptr;
    e.scal = scal.data(); e.partials = tpartials.data(); e.totals = nullptr; e.alpha = alpha; e.nv_global = g.nv;
    e.wmax = tc->wmax;
    e.xcol = tc->xcol.size() ? tc->xcol.data() : nullptr;
    e.cr   = crows;
    return e;
  }
  void iterate_tiled(bool need_diff, bool last_of_call)
  {
    WT const* xcur = cur == 0 ? x0.data() : x1.data();
    WT* xnext      = cur == 0 ? x1.data() : x0.data();
    tiled_epilogue<WT> e = tiled_epi(xnext);
    e.need_diff = need_diff || force_diff;
    // pr is this plan's result buffer, not its iteration state (that is x): an iteration whose L1 change is not wanted and that is
    // followed by another one in the same call leaves pr alone.  Rows of dangling vertices have no x: they need pr itself, but only
    // to be summed, which happens in registers.  (CUGRAPH_AMD_PAGERANK_WRITE_PR=1 writes it every iteration.)
    e.write_pr = e.need_diff || last_of_call || force_write_pr;
    tiled_phase1<WT>(h, *tc, xcur, alpha, part.data(), counters.data(), tiled_x_map<WT>{}, pending_finish ? &e : nullptr);
    tiled_phase2<WT>(h, *tc, (WT const*)part.data(), e, counters.data());
    pending_finish   = true;
    const_rows_stale = crows.nI_act > 0;
  }
  bool const_rows_allowed{false};
  bool stepped{false};  // step() ran: the iteration state is the caller's
  void tiled_initial_state()
  {
    int64_t const nv = g.nv;
    setup_const_rows(const_rows_allowed);
    int n = tiled_prologue<WT>(h, *tc, (WT const*)pr.data(), outw, x0.data(), nv, tpartials.data());
    // the rows without in-edges start from the uniform value (there is no user vector when they are left out)
    tiled_finish<WT>(h, tiled_epi(nullptr), n, crows.nI_act > 0 ? (double)(WT(1) / (WT)nv) : -1.0);
    cur              = 0;
    pending_finish   = false;
    const_rows_stale = false;
    h.sync();
  }

  // Placement trials (cugraph_amd_pagerank_plan_tune; CUGRAPH_AMD_PR_PLACEMENT_TRIALS=n runs them inside every plan creation).  The SAME plan (same
  // kernels, same bytes) on other physical pages runs anywhere in a +-2.5 % band on this part (RMAT-26: phase 1 0.92-0.99 ms, phase 2 0.40-0.45 ms;
  // DESIGN.md section 3.1, profiles/r6c_*, r6d_*, r6e_*, r6t_*, r6u_*).  Where the streamed arrays lie decides how their streams meet in the memory
  // system; an offset inside one allocation changes nothing (measured), other pages do, and user space cannot choose pages -- but it can SAMPLE them:
  // the plan times itself on `placements` fresh allocations of its large streamed arrays (contents copied device to device) and keeps the fastest.
  // Same data, same kernels, same bits; only the addresses differ.  It costs ~10 iterations + one copy per placement, so it is the caller's decision
  // (a solver that runs the plan hundreds of times; a one-shot cugraph_pagerank of 20 iterations would lose): NOT done by default.  The re-blocked
  // arrays are cached on the graph, so later plans of the graph start from the kept placement.  Only before the first step(): the trial iterations run on
  // the plan's own vectors, which are set back to the initial state afterwards (the initial vector itself is never written by them).
  template <typename T>
  static dvec<T> clone_dvec(handle_t const& h, dvec<T> const& a)
  {
    dvec<T> b(a.size());
    if (a.size()) HIP_TRY(hipMemcpyAsync(b.data(), a.data(), a.size() * sizeof(T), hipMemcpyDeviceToDevice, h.stream));
    return b;
  }
  double tune(int placements) override
  {
    HIP_TRY(hipSetDevice(h.device));
    CGA_EXPECTS(!stepped, CUGRAPH_INVALID_INPUT, "pagerank plan: tune() must come before the first step()");
    if (!tiled || placements <= 1 || g.ne == 0 || force_diff || force_write_pr) return 0.0;
    bool const trace = getenv("CUGRAPH_AMD_PR_PLACEMENT_TRACE") != nullptr;
    constexpr int kWarm = 2, kTimed = 8;
    hipEvent_t ev[2 * kTimed + 1];
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    auto free_events = [&] { for (auto& e : ev) (void)hipEventDestroy(e); };
    struct times_t { double p1, p2; double total() const { return p1 + p2; } };
    // one event in front of every launch of the timed iterations: phase 1 of iteration i = ev[2i] .. ev[2i + 1] (the launch gap rides with it), phase 2 =
    // ev[2i + 1] .. ev[2i + 2]
    auto time_plan = [&]() -> times_t {
      for (int i = 0; i < kWarm; ++i) { iterate_tiled(false, false); cur ^= 1; }
      for (int i = 0; i < kTimed; ++i) {
        WT const* xcur = cur == 0 ? x0.data() : x1.data();
        WT* xnext      = cur == 0 ? x1.data() : x0.data();
        tiled_epilogue<WT> e = tiled_epi(xnext);
        e.need_diff = false; e.write_pr = false;
        HIP_TRY(hipEventRecord(ev[2 * i], h.stream));
        tiled_phase1<WT>(h, *tc, xcur, alpha, part.data(), counters.data(), tiled_x_map<WT>{}, pending_finish ? &e : nullptr);
        HIP_TRY(hipEventRecord(ev[2 * i + 1], h.stream));
        tiled_phase2<WT>(h, *tc, (WT const*)part.data(), e, counters.data());
        pending_finish = true;
        cur ^= 1;
      }
      HIP_TRY(hipEventRecord(ev[2 * kTimed], h.stream));
      HIP_TRY(hipEventSynchronize(ev[2 * kTimed]));
      times_t t{0, 0};
      for (int i = 0; i < kTimed; ++i) {
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, ev[2 * i], ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&b, ev[2 * i + 1], ev[2 * i + 2]));
        t.p1 += a / kTimed; t.p2 += b / kTimed;
      }
      return t;
    };
    // The arrays an iteration streams, in three groups: A = what phase 1 reads (src16, bits, weights, delta1, wrec), B = the partial buffer (phase 1 writes
    // it, phase 2 reads it), C = the destinations phase 2 reads (dstl12 / dstl16)
    struct placement_t {
      dvec<uint16_t> src16; dvec<uint32_t> bits; dev_buf weights; dvec<uint32_t> delta1, wrec, dstl12; dvec<uint16_t> dstl16; dvec<WT> part;
      unsigned groups{0};
    };
    enum : unsigned { GA = 1, GB = 2, GC = 4 };
    auto clone = [&](unsigned groups) {
      placement_t p;
      p.groups = groups;
      if (groups & GA) {
        p.src16 = clone_dvec(h, tc->src16); p.bits = clone_dvec(h, tc->bits); p.delta1 = clone_dvec(h, tc->delta1); p.wrec = clone_dvec(h, tc->wrec);
        if (tc->weights.ptr) {
          p.weights.alloc(tc->weights.bytes);
          HIP_TRY(hipMemcpyAsync(p.weights.ptr, tc->weights.ptr, tc->weights.bytes, hipMemcpyDeviceToDevice, h.stream));
        }
      }
      if (groups & GB) p.part = clone_dvec(h, part);
      if (groups & GC) { p.dstl12 = clone_dvec(h, tc->dstl12); p.dstl16 = clone_dvec(h, tc->dstl16); }
      return p;
    };
    auto swap_in = [&](placement_t& p) {
      if (p.groups & GA) { std::swap(tc->src16, p.src16); std::swap(tc->bits, p.bits); std::swap(tc->weights, p.weights); std::swap(tc->delta1, p.delta1); std::swap(tc->wrec, p.wrec); }
      if (p.groups & GB) std::swap(part, p.part);
      if (p.groups & GC) { std::swap(tc->dstl12, p.dstl12); std::swap(tc->dstl16, p.dstl16); }
    };
    size_t const bytes_a = tc->src16.buf.bytes + tc->bits.buf.bytes + tc->weights.bytes + tc->delta1.buf.bytes + tc->wrec.buf.bytes;
    size_t const bytes_b = part.buf.bytes, bytes_c = tc->dstl12.buf.bytes + tc->dstl16.buf.bytes;
    (void)time_plan();  // (the first timing after a build is 2-5 % slow whatever the placement: clocks, first touches)
    times_t best = time_plan();
    auto where = [&] {
      char buf[256];
      snprintf(buf, sizeof buf, " | part %p (%zu of %zu granted) src16 %p (%zu of %zu) dstl12 %p (%zu of %zu)", (void*)part.data(), part.buf.bytes, part.buf.granted, (void*)tc->src16.data(),
               tc->src16.buf.bytes, tc->src16.buf.granted, (void*)tc->dstl12.data(), tc->dstl12.buf.bytes, tc->dstl12.buf.granted);
      return std::string(buf);
    };
    if (trace) fprintf(stderr, "[pagerank plan] placement 0: phase 1 %.4f + phase 2 %.4f = %.4f ms per iteration%s\n", best.p1, best.p2, best.total(), where().c_str());
    std::vector<placement_t> losers;  // kept until the end: a block handed back to the pool would be the next trial's "fresh" allocation
    size_t held = 0;
    // the schedule: the first placements re-roll everything; the later ones one group at a time, judged by the phase(s) that stream it.  What the traces
    // say (profiles/r6v_placement_groups.txt): the partial buffer (B) decides most -- a "bad" one costs phase 1 0.02 ms AND phase 2 0.025 ms at RMAT-26 --,
    // phase 1's read streams (A) are bimodal as well (0.93 / 0.98 ms), C is worth up to 0.01 ms of phase 2; allocations made late, with much memory
    // held, are mostly bad (14 placements are no better than 8)
    auto groups_of = [&](int t) -> unsigned {
      if (t <= 4) return GA | GB | GC;
      switch ((t - 5) % 3) { case 0: return GA; case 1: return GC; default: return GB; }
    };
    try {
      for (int t = 1; t < placements; ++t) {
        unsigned const gr = groups_of(t);
        size_t const need = ((gr & GA) ? bytes_a : 0) + ((gr & GB) ? bytes_b : 0) + ((gr & GC) ? bytes_c : 0);
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if (held + 2 * need > total_b / 4) break;  // the trials never hold more than a quarter of the device
        placement_t p = clone(gr);
        swap_in(p);  // p now holds the previous best's arrays of these groups
        times_t const ms = time_plan();
        // a group is judged by the phases that stream it: A by phase 1, C by phase 2, B and whole placements by the iteration
        bool const better = gr == GA ? ms.p1 < best.p1 : gr == GC ? ms.p2 < best.p2 : ms.total() < best.total();
        if (trace)
          fprintf(stderr, "[pagerank plan] placement %d (%s%s%s): phase 1 %.4f + phase 2 %.4f = %.4f ms per iteration%s\n", t, (gr & GA) ? "A" : "", (gr & GB) ? "B" : "",
                  (gr & GC) ? "C" : "", ms.p1, ms.p2, ms.total(), (std::string(better ? "  (kept)" : "") + where()).c_str());
        if (better) best = ms; else swap_in(p);  // p holds the loser either way
        held += need;
        losers.push_back(std::move(p));
      }
    } catch (api_error const& e) {  // no memory for another copy: the best placement so far stays
      if (e.code != CUGRAPH_ALLOC_ERROR) { free_events(); throw; }
    }
    h.sync();
    free_events();
    losers.clear();
    tiled_initial_state();  // the trial iterations ran on the plan's own vectors
    return best.total();
  }

  void flush_tiled_scalars()
  {
    if (!pending_finish) return;
    tiled_epilogue<WT> const e = tiled_epi(nullptr);
    tiled_finish<WT>(h, e, tiled_fold_count(*tc, e));
    pending_finish = false;
  }

  void launch_finish()
  {
    hipLaunchKernelGGL(k_finish<WT>, 1, 256, 0, h.stream, (double const*)partials.data(), grid, scal.data(), alpha, (WT)(1.0 - (double)alpha),
                       g.nv, personalized ? 1 : 0);
  }

  void create(device_array_view_t const* ow_v, device_array_view_t const* ow_s, device_array_view_t const* ig_v,
              device_array_view_t const* ig_s, device_array_view_t const* p_v, device_array_view_t const* p_s)
  {
    HIP_TRY(hipSetDevice(h.device));
    // (round 5: the column-tiled plan addresses its edge arrays from 64-bit per-wavefront bases, so it takes graphs of 2^31 edges and more --
    // RMAT-27 on one 288 GB GPU; the single-pass comparison kernels keep signed 32-bit positions and are not offered there)
    CGA_EXPECTS(g.ne <= kMaxGraphEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "PageRank: graphs of more than 2^32 - 4097 edges are not supported");
    ensure_orientation(h, g, true, /*dcs_aware=*/true);  // PageRank pulls over CSC (plain or hypersparse rows: the re-blocking walks either form)
    int64_t const nv = g.nv;
    size_t const n1  = (size_t)(nv > 0 ? nv : 1);
    pr.resize_discard(n1); x0.resize_discard(n1); x1.resize_discard(n1);
    scal.resize_discard(1);

    // persistent launch geometry: 1 or 2 workgroups of 1024 threads per CU depending on the LDS tile
    int hot_req = h.pagerank_hot_tile;
    if (hot_req < 0) hot_req = (int)(65536 / sizeof(WT));  // 64 KiB tile -> 2 workgroups per CU
    size_t red_bytes = (2 * PR_WAVES) * sizeof(double) + PR_WAVES * sizeof(double);
    size_t max_tile  = (h.lds_per_block > red_bytes + 1024 ? h.lds_per_block - red_bytes - 1024 : 0) / sizeof(WT);
    hot = (int)std::min<int64_t>({(int64_t)hot_req, (int64_t)max_tile, nv});
    hot &= ~3;
    lds_bytes       = (size_t)hot * sizeof(WT) + red_bytes;
    int blocks_per_cu = lds_bytes <= 80 * 1024 ? 2 : 1;
    grid            = h.num_cus * blocks_per_cu;
    partials.resize_discard((size_t)2 * std::max(grid, 2048));
    build_trace tr(h, "plan");
    tr.step("vectors");
    setup_flat();
    tr.step("re-blocked structure");

    if (ow_s) {
      outw_own.resize_discard(n1);
      pairs_to_dense(ow_v, ow_s, outw_own.data(), WT(0), "precomputed_vertex_out_weight");
      outw = outw_own.data();
    } else {
      compute_out_weight_sums();
    }
    tr.step("out-weight sums");
    if (ig_s) {
      pairs_to_dense(ig_v, ig_s, pr.data(), WT(0), "initial_guess");  // not renormalised: pagerank_impl.cuh:427-432
    } else {
      fill_wt<WT>(h, pr.data(), nv, nv > 0 ? WT(1) / (WT)nv : WT(0));  // pagerank_impl.cuh:422-426
    }
    if (p_s) {
      personalized = true;
      CGA_EXPECTS(p_v->size == p_s->size, CUGRAPH_INVALID_INPUT, "personalization: vertices and values differ in size");
      dvec<double> sum(1);
      HIP_TRY(hipMemsetAsync(sum.data(), 0, 8, h.stream));
      if (p_s->size > 0)
        hipLaunchKernelGGL(k_sum<WT>, grid_for(p_s->size, kBlock, 1024), kBlock, 0, h.stream, p_s->as<WT const>(), (int64_t)p_s->size, sum.data());
      double s;
      h.read_back(&s, sum.data(), 1);
      CGA_EXPECTS((WT)s > WT(0), CUGRAPH_INVALID_INPUT, "Invalid input argument: sum of personalization values should be positive.");
      pers.resize_discard(n1);
      fill_wt<WT>(h, pers.data(), nv, WT(0));
      dvec<int32_t> tmp(p_v->size > 0 ? p_v->size : 1);
      HIP_TRY(hipMemcpyAsync(tmp.data(), p_v->data, p_v->size * 4, hipMemcpyDeviceToDevice, h.stream));
      renumber_ext_to_int(h, g, tmp.data(), (int64_t)p_v->size);
      CGA_EXPECTS(count_negative_i32(h, tmp.data(), (int64_t)p_v->size) == 0, CUGRAPH_INVALID_INPUT,
                  "personalization: found a vertex id that is not in the graph");
      if (p_v->size > 0)
        hipLaunchKernelGGL((k_scatter_pairs<WT, true>), grid_for(p_v->size, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)tmp.data(),
                           p_s->as<WT const>(), (int64_t)p_v->size, pers.data(), WT(1) / (WT)s);
      h.sync();
    }
    // iteration-0 state: x = pr / out_w, dangling mass, base
    if (tiled) {
      const_rows_allowed = !personalized && !ig_s;
      tiled_initial_state();
      if (char const* env = getenv("CUGRAPH_AMD_PR_PLACEMENT_TRIALS")) (void)tune(atoi(env));  // (default: the caller asks for it, cugraph_amd_pagerank_plan_tune)
      return;
    }
    int pgrid = std::min(grid_for(nv, 256, 2048), 2048);
    hipLaunchKernelGGL(k_prologue<WT>, pgrid, 256, 0, h.stream, (WT const*)pr.data(), outw, x0.data(), nv, partials.data());
    int keep = grid;
    grid     = pgrid;
    launch_finish();
    grid = keep;
    cur  = 0;
    h.sync();
  }

  template <bool WEIGHTED, bool PERS>
  void launch_spmv(spmv_args<WT> const& a)
  {
    timed_launch t(h, "pagerank_spmv");
    hipLaunchKernelGGL((k_spmv<WT, WEIGHTED, PERS>), grid, PR_BLOCK, lds_bytes, h.stream, a);
  }

  void step(double epsilon, size_t max_iterations, size_t* done, bool* converged) override
  {
    HIP_TRY(hipSetDevice(h.device));
    stepped = true;
    static bool attr_set[4] = {false, false, false, false};
    auto set_attr = [&](auto kernel, int slot) {
      if (!attr_set[slot]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_per_block));
        attr_set[slot] = true;
      }
    };
    set_attr(k_spmv<WT, false, false>, 0); set_attr(k_spmv<WT, true, false>, 1);
    set_attr(k_spmv<WT, false, true>, 2);  set_attr(k_spmv<WT, true, true>, 3);
    orientation_t const& o = g.csc;
    size_t it = 0;
    bool conv = false;
    WT const eps = (WT)epsilon;
    static bool flat_attr[2] = {false, false};
    if (flat) {
      if (!flat_attr[0]) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(k_spmv_flat<WT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_per_block)); flat_attr[0] = true; }
      if (!flat_attr[1]) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(k_spmv_flat<WT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds_per_block)); flat_attr[1] = true; }
    }
    while (it < max_iterations) {
      if (tiled || flat) {
        if (tiled) iterate_tiled(epsilon > 0.0, it + 1 == max_iterations); else iterate_flat();  // epsilon == 0: nobody looks at the L1 change
        cur ^= 1;
        ++it;
        if (epsilon > 0.0) {
          if (tiled) flush_tiled_scalars();
          pr_scalars<WT> sc;
          h.read_back(&sc, scal.data(), 1);
          if (sc.diff < eps) { conv = true; break; }
        }
        continue;
      }
      spmv_args<WT> a;
      a.offsets   = o.offsets.data();
      a.indices   = o.indices.data();
      a.weights   = g.has_weights ? o.weights.as<WT const>() : nullptr;
      a.row_order = o.row_order.size() ? o.row_order.data() : nullptr;
      a.nv        = g.nv;
      a.seg0 = o.seg[0]; a.seg1 = o.seg[1]; a.seg2 = o.seg[2]; a.seg3 = o.seg[3];
      a.x         = cur == 0 ? x0.data() : x1.data();
      a.x_next    = cur == 0 ? x1.data() : x0.data();
      a.pr        = pr.data();
      a.outw      = outw;
      a.pers      = personalized ? pers.data() : nullptr;
      a.scal      = scal.data();
      a.partials  = partials.data();
      a.alpha     = alpha;
      a.hot       = hot;
      if (g.has_weights) { if (personalized) launch_spmv<true, true>(a); else launch_spmv<true, false>(a); }
      else               { if (personalized) launch_spmv<false, true>(a); else launch_spmv<false, false>(a); }
      launch_finish();
      cur ^= 1;
      ++it;
      if (epsilon > 0.0) {  // pagerank_impl.cuh:320-326: diff < eps -> stop (one 16-byte read-back per iteration)
        pr_scalars<WT> s;
        h.read_back(&s, scal.data(), 1);
        if (s.diff < eps) { conv = true; break; }
      }
    }
    if (tiled) flush_tiled_scalars();
    *done      = it;
    *converged = conv;
  }

  centrality_result_t* result(size_t total_iterations, bool converged) override
  {
    auto ids  = std::make_unique<device_array_t>((size_t)g.nv, g.vertex_type);
    auto vals = std::make_unique<device_array_t>((size_t)g.nv, g.weight_type);
    if (tiled) { flush_tiled_scalars(); materialize_const_rows(); }
    if (g.nv > 0) {
      HIP_TRY(hipMemcpyAsync(ids->buf.ptr, g.number_map.data(), g.nv * 4, hipMemcpyDeviceToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(vals->buf.ptr, pr.data(), g.nv * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    }
    h.sync();
    auto* r = new centrality_result_t{ids.release(), vals.release(), total_iterations, converged};
    outer_replace_ids(h, g, r->vertex_ids);
    return r;
  }
};


// =================================================================================================
// Multi-GPU PageRank, one process per GPU, re-designed for a full-mesh xGMI node.
// 1-D partition by destination: this rank owns n_rows destination vertices and ALL their in-edges, so the pull-SpMV needs
// no partial-result reduction.  The only data-path exchange is a SPARSE all-to-all of x = pr / out_w: every rank receives
// exactly the source values its local edges reference (vertices without out-edges -- more than half of an RMAT graph --
// are never sent; a low-degree source travels only to the few ranks that hold one of its out-edges), as one
// point-to-point message per peer, so all seven xGMI links of a GPU carry traffic at once (a ring all-gather of the whole
// vector would be bound by ONE link and move (P-1)/P * 4V bytes into every GPU).  The host layer (cugraph_amd/mg.py)
// issues that collective (torch.distributed all_to_all_single = RCCL) on the two device buffers this plan exposes.
// The three scalars of an iteration (L1 change, dangling mass, max |x|) ride in a 32-byte tail behind every message, so
// there is no separate all-reduce: every rank folds the P triples in rank order (deterministic).
// Local column ids are COMPACT (0 .. ncols-1 = the distinct sources this rank references, hottest first), so the
// column-tiled kernels run unchanged on x_compact[c] = recv[col_pos[c]].
// The reference instead uses a 2-D partition with a row broadcast + column reduce + 2 scalar all-reduces per iteration
// (update_edge_src_dst_property.cuh:550-579, per_v_transform_reduce_e.cuh:3390-3406).
// =================================================================================================
template <typename WT>
__global__ void k_mg_unpack(WT const* recv, int32_t const* col_pos, int64_t ncols, WT* x)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < ncols; i += stride) x[i] = recv[col_pos[i]];
}

// send[seg_off[s] + j] = x_own[send_index[first[s] + j]]; the tail of every segment <- totals (3 doubles)
template <typename WT>
__global__ void k_mg_pack(WT const* x_own, int32_t const* send_index, int64_t n_send, int64_t const* first /*[P+1] value index*/,
                          int64_t const* seg_off /*[P] element offset of segment s in send*/, int comm_size, double const* totals, WT* send)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = i; k < n_send; k += stride) {
    int s = 0;
    while (s + 1 < comm_size && k >= first[s + 1]) ++s;
    send[seg_off[s] + (k - first[s])] = x_own[send_index[k]];
  }
  if (i < comm_size) {
    double* tail = reinterpret_cast<double*>(send + seg_off[i] + (first[i + 1] - first[i]));
    tail[0] = totals[0]; tail[1] = totals[1]; tail[2] = totals[2]; tail[3] = 0.0;
  }
}

template <typename WT>
__global__ void k_mg_fold_tails(WT const* recv, int64_t const* tail_off /*[P] element offset of the tail of segment s*/, int comm_size,
                                pr_scalars<WT>* scal, WT alpha, int64_t nv_global, double wmax)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double diff = 0, dang = 0, xmax = 0;
  for (int r = 0; r < comm_size; ++r) {
    double const* t = reinterpret_cast<double const*>(recv + tail_off[r]);
    diff += t[0]; dang += t[1]; xmax = fmax(xmax, t[2]);
  }
  tiled_write_scalars<WT>(scal, diff, dang, xmax, alpha, nv_global, 0, wmax);
}

struct pagerank_mg_plan_base {
  handle_t const* hp{nullptr};  // the entry points name its stream to the memory pool (frees / reuses are ordered on it)
  virtual ~pagerank_mg_plan_base() = default;
  virtual void start()                                           = 0;
  virtual void reduce_scalars(bool read_back, double* diff, double* dangling) = 0;
  virtual void local_step()                                      = 0;
  virtual void values(device_array_view_t const* out)            = 0;
};

template <typename WT>
struct pagerank_mg_plan : pagerank_mg_plan_base {
  static constexpr size_t kTailBytes = 32;
  static constexpr int64_t kTail     = (int64_t)(kTailBytes / sizeof(WT));  // tail length in elements
  handle_t const& h;
  graph_t& g;
  WT alpha;
  int64_t n_rows, nv_global, ncols{0}, n_send{0}, n_recv{0};
  int rank, size;
  dvec<WT> pr, outw, part, x_own, x_compact;
  dvec<int32_t> send_index, col_pos;
  dvec<int64_t> d_first, d_send_off, d_recv_tail;
  WT* send{nullptr};  // caller-owned exchange buffers (torch tensors on the host side)
  WT* recv{nullptr};
  dvec<pr_scalars<WT>> scal;
  dvec<double> tpartials, totals;
  dvec<uint32_t> counters;
  std::shared_ptr<tiled_csc_t> tc;

  pagerank_mg_plan(handle_t const& h_, graph_t& g_, double alpha_, int64_t n_rows_, int64_t nv_global_, int rank_, int size_)
    : h(h_), g(g_), alpha((WT)alpha_), n_rows(n_rows_), nv_global(nv_global_), rank(rank_), size(size_)
  {
    hp = &h_;
  }

  void create(device_array_view_t const* outw_local, device_array_view_t const* init_local, device_array_view_t const* send_index_v,
              size_t const* send_counts, size_t const* recv_counts, device_array_view_t const* col_pos_v, device_array_view_t const* send_v,
              device_array_view_t const* recv_v)
  {
    HIP_TRY(hipSetDevice(h.device));
    CGA_EXPECTS(size >= 1 && send_counts && recv_counts, CUGRAPH_INVALID_INPUT, "multi-GPU PageRank: comm_size / counts");
    std::vector<int64_t> first(size + 1, 0), send_off(size, 0), recv_tail(size, 0);
    int64_t so = 0, ro = 0;
    for (int s = 0; s < size; ++s) {
      first[s + 1] = first[s] + (int64_t)send_counts[s];
      send_off[s]  = so;
      so += (int64_t)send_counts[s] + kTail;
      ro += (int64_t)recv_counts[s];
      recv_tail[s] = ro;
      ro += kTail;
      CGA_EXPECTS(((int64_t)send_counts[s] * (int64_t)sizeof(WT)) % 8 == 0 && ((int64_t)recv_counts[s] * (int64_t)sizeof(WT)) % 8 == 0, CUGRAPH_INVALID_INPUT,
                  "multi-GPU PageRank: per-peer counts must keep the 8-byte alignment of the tails (even counts for fp32)");
    }
    n_send = first[size];
    n_recv = ro - (int64_t)size * kTail;
    CGA_EXPECTS(send_v && recv_v && send_v->type == g.weight_type && recv_v->type == g.weight_type && (int64_t)send_v->size == so &&
                  (int64_t)recv_v->size == ro,
                CUGRAPH_INVALID_INPUT, "multi-GPU PageRank: send / recv must hold the per-peer values plus a 32-byte tail per peer");
    CGA_EXPECTS(send_index_v && send_index_v->type == INT32 && (int64_t)send_index_v->size == n_send, CUGRAPH_INVALID_INPUT,
                "multi-GPU PageRank: send_index must be INT32 with sum(send_counts) entries");
    CGA_EXPECTS(col_pos_v && col_pos_v->type == INT32, CUGRAPH_INVALID_INPUT, "multi-GPU PageRank: col_pos must be INT32");
    ncols = (int64_t)col_pos_v->size;
    CGA_EXPECTS(ncols <= g.nv && n_rows <= g.nv, CUGRAPH_INVALID_INPUT, "multi-GPU PageRank: the local graph must have max(columns, local rows) vertices");
    send = send_v->as<WT>();
    recv = recv_v->as<WT>();
    CGA_EXPECTS(outw_local != nullptr && (int64_t)outw_local->size == n_rows && outw_local->type == g.weight_type, CUGRAPH_INVALID_INPUT,
                "multi-GPU PageRank: out_weight_sums must have one weight-typed value per local row");
    ensure_orientation(h, g, true);
    orientation_t& o = g.csc;
    CGA_EXPECTS(o.seg[4] <= n_rows, CUGRAPH_INVALID_INPUT, "multi-GPU PageRank: edges point to rows outside the local range");
    size_t const n1 = (size_t)(n_rows > 0 ? n_rows : 1);
    pr.resize_discard(n1); outw.resize_discard(n1); x_own.resize_discard(n1);
    scal.resize_discard(1); totals.resize_discard(4);
    HIP_TRY(hipMemsetAsync(scal.data(), 0, sizeof(pr_scalars<WT>), h.stream));
    HIP_TRY(hipMemsetAsync(totals.data(), 0, 4 * sizeof(double), h.stream));
    HIP_TRY(hipMemsetAsync(send, 0, (size_t)so * sizeof(WT), h.stream));
    HIP_TRY(hipMemsetAsync(recv, 0, (size_t)ro * sizeof(WT), h.stream));
    if (n_rows > 0) HIP_TRY(hipMemcpyAsync(outw.data(), outw_local->data, n_rows * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    if (init_local) {
      CGA_EXPECTS((int64_t)init_local->size == n_rows && init_local->type == g.weight_type, CUGRAPH_INVALID_INPUT, "initial guess: one value per local row");
      if (n_rows > 0) HIP_TRY(hipMemcpyAsync(pr.data(), init_local->data, n_rows * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    } else {
      fill_wt<WT>(h, pr.data(), n_rows, WT(1) / (WT)nv_global);
    }
    send_index.resize_discard(n_send > 0 ? n_send : 1);
    col_pos.resize_discard(ncols > 0 ? ncols : 1);
    if (n_send > 0) HIP_TRY(hipMemcpyAsync(send_index.data(), send_index_v->data, n_send * 4, hipMemcpyDeviceToDevice, h.stream));
    if (ncols > 0) HIP_TRY(hipMemcpyAsync(col_pos.data(), col_pos_v->data, ncols * 4, hipMemcpyDeviceToDevice, h.stream));
    auto upload = [&](dvec<int64_t>& d, std::vector<int64_t> const& v) {
      d.resize_discard(v.size());
      HIP_TRY(hipMemcpyAsync(d.data(), v.data(), v.size() * sizeof(int64_t), hipMemcpyHostToDevice, h.stream));
      h.sync();
    };
    upload(d_first, first); upload(d_send_off, send_off); upload(d_recv_tail, recv_tail);
    // compact column ids (hottest sources first): the column-tiled re-blocking applies unchanged
    int const T = tiled_default_T(h, sizeof(WT), std::max<int64_t>(ncols, 1));
    if (!o.tiled || o.tiled->T != T || o.tiled->nv != n_rows) {
      auto t = std::make_shared<tiled_csc_t>();
      build_tiled_csc(h, g.nv, n_rows, g.ne, o, g.has_weights, sizeof(WT), T, *t);
      o.tiled = t;
    }
    tc = o.tiled;
    part.resize_discard((size_t)tc->n_slots + 64);
    HIP_TRY(hipMemsetAsync(part.data(), 0, ((size_t)tc->n_slots + 64) * sizeof(WT), h.stream));
    size_t const nx = (size_t)tc->nJ * tc->T + 8;
    x_compact.resize_discard(nx);
    HIP_TRY(hipMemsetAsync(x_compact.data(), 0, nx * sizeof(WT), h.stream));
    counters.resize_discard(4);
    HIP_TRY(hipMemsetAsync(counters.data(), 0, 4 * sizeof(uint32_t), h.stream));
    tpartials.resize_discard((size_t)3 * std::max(tc->nI, 1024));
    h.sync();
  }

  tiled_epilogue<WT> epi()
  {
    tiled_epilogue<WT> e;
    e.nv = n_rows; e.pr = pr.data(); e.x_next = x_own.data(); e.outw = outw.data(); e.pers = nullptr; e.scal = scal.data();
    e.partials = tpartials.data(); e.totals = totals.data(); e.alpha = alpha; e.nv_global = nv_global; e.wmax = tc->wmax;
    return e;
  }

  void pack()
  {
    int const grid = std::max(1, std::min(grid_for(std::max<int64_t>(n_send, size), kBlock, 2048), 2048));
    hipLaunchKernelGGL(k_mg_pack<WT>, grid, kBlock, 0, h.stream, (WT const*)x_own.data(), (int32_t const*)send_index.data(), n_send,
                       (int64_t const*)d_first.data(), (int64_t const*)d_send_off.data(), size, (double const*)totals.data(), send);
  }

  void start() override
  {  // x_own <- x of the initial vector, totals <- (0, partial dangling mass, max |x|); messages packed
    HIP_TRY(hipSetDevice(h.device));
    int n = tiled_prologue<WT>(h, *tc, (WT const*)pr.data(), (WT const*)outw.data(), x_own.data(), n_rows, tpartials.data());
    tiled_finish<WT>(h, epi(), n);
    pack();
    h.sync();
  }

  void reduce_scalars(bool read_back, double* diff, double* dangling) override
  {
    hipLaunchKernelGGL(k_mg_fold_tails<WT>, 1, 64, 0, h.stream, (WT const*)recv, (int64_t const*)d_recv_tail.data(), size, scal.data(), alpha, nv_global,
                       tc->wmax);
    if (read_back) {
      pr_scalars<WT> sc;
      h.read_back(&sc, scal.data(), 1);
      if (diff) *diff = (double)sc.diff;
      if (dangling) *dangling = (double)sc.dangling;
    }
  }

  void local_step() override
  {
    HIP_TRY(hipSetDevice(h.device));
    if (ncols > 0)
      hipLaunchKernelGGL(k_mg_unpack<WT>, grid_for(ncols, kBlock, 4096), kBlock, 0, h.stream, (WT const*)recv, (int32_t const*)col_pos.data(), ncols,
                         x_compact.data());
    tiled_epilogue<WT> e = epi();
    tiled_phase1<WT>(h, *tc, (WT const*)x_compact.data(), alpha, part.data(), counters.data(), tiled_x_map<WT>{}, nullptr);
    tiled_phase2<WT>(h, *tc, (WT const*)part.data(), e, counters.data());
    tiled_finish<WT>(h, e, tc->nI);  // this rank's (diff, dangling, xmax)
    pack();
    if (!h.stream_borrowed) h.sync();  // the host layer issues the next all-to-all on its own stream -- unless it shares ours
  }

  void values(device_array_view_t const* out) override
  {
    CGA_EXPECTS(out != nullptr && (int64_t)out->size == n_rows && out->type == g.weight_type, CUGRAPH_INVALID_INPUT, "values: one weight-typed value per local row");
    if (n_rows > 0) HIP_TRY(hipMemcpyAsync(out->data, pr.data(), n_rows * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    h.sync();
  }
};

// =================================================================================================
// 2-D layout (north_star / the reference's own scheme; cpp/include/cugraph/graph_view.hpp:159-216,
// partition_manager.hpp:42-51): P = R x C ranks, rank = c * R + r.  Vertex partitions 0 .. P-1 of L rows each (position p of the
// global degree order -> partition p % P, row p / P); rank (r, c) OWNS partition c * R + r and STORES the edges whose source lies
// in the R partitions of its column group [c * R, (c + 1) * R) and whose destination lies in the C partitions {i * R + r}:
// local column id = (q_src % R) * L + row, local row id = (q_dst / R) * L + row.  Per iteration (host layer: cugraph_amd/mg.py,
// MGPageRank2D): all-gather of x over the column group (update_edge_src_dst_property.cuh:550-579) -> spmv() = plain tiled SpMV
// of the local block (phase 2 in raw mode: no epilogue) -> reduce-scatter of the C partial row blocks over the row group
// (per_v_transform_reduce_e.cuh:3390-3406) -> epilogue() on the owned L rows -> all-gather of the (L1 change, dangling, max|x|)
// triples, folded in rank order by set_scalars().  Built to be measured against the 1-D sparse all-to-all on real hardware;
// DESIGN.md section 5 has the byte counts of both.
// =================================================================================================
// iteration-0 state of an owned slice: x = pr / out_w, per-block (0, dangling mass, max |x|) partials
template <typename WT>
__global__ void __launch_bounds__(256) k_tiled_prologue_plain(WT const* pr, WT const* outw, WT* x, int64_t n, double* partials)
{
  __shared__ double red[2][4];
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double dang = 0.0, xmax = 0.0;
  for (; i < n; i += stride) {
    WT const p = pr[i], ow = outw[i];
    WT const xv = p / (ow == WT(0) ? WT(1) : ow);
    x[i] = xv;
    xmax = fmax(xmax, fabs((double)xv));
    if (ow == WT(0)) dang += (double)p;
  }
  dang = group_sum(dang, 64);
  for (int o = 32; o > 0; o >>= 1) xmax = fmax(xmax, __shfl_xor(xmax, o));
  int const wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = dang; red[1][wave] = xmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[3 * blockIdx.x]     = 0.0;
    partials[3 * blockIdx.x + 1] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partials[3 * blockIdx.x + 2] = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
  }
}

template <typename WT>
__global__ void __launch_bounds__(256) k_mg2d_epilogue(WT const* y_own, WT const* outw, WT* pr, WT* x_own, int64_t n, pr_scalars<WT> const* scal, double* partials)
{
  __shared__ double red[3][4];
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  WT const base  = scal->base;
  double diff = 0.0, dang = 0.0, xmax = 0.0;
  for (; i < n; i += stride) {
    WT const old = pr[i], ow = outw[i];
    WT const val = base + y_own[i];
    WT const xn  = val / (ow == WT(0) ? WT(1) : ow);
    pr[i]    = val;
    x_own[i] = xn;
    diff += (double)fabs(val - old);
    xmax = fmax(xmax, fabs((double)xn));
    if (ow == WT(0)) dang += (double)val;
  }
  diff = group_sum(diff, 64);
  dang = group_sum(dang, 64);
  for (int o = 32; o > 0; o >>= 1) xmax = fmax(xmax, __shfl_xor(xmax, o));
  int const wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = diff; red[1][wave] = dang; red[2][wave] = xmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[3 * blockIdx.x]     = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partials[3 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    partials[3 * blockIdx.x + 2] = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
  }
}

struct pagerank_mg2d_plan_base {
  handle_t const* hp{nullptr};
  virtual ~pagerank_mg2d_plan_base() = default;
  virtual void start()                                                          = 0;
  virtual void set_scalars(void const* gathered, int nranks, bool read_back, double* diff, double* dangling) = 0;
  virtual void spmv()                                                           = 0;
  virtual void epilogue()                                                       = 0;
  virtual void values(device_array_view_t const* out)                           = 0;
};

template <typename WT>
struct pagerank_mg2d_plan : pagerank_mg2d_plan_base {
  handle_t const& h;
  graph_t& g;
  WT alpha;
  int64_t L, n_own, n_block_rows, n_block_cols, nv_global;  // n_own <= L: owned rows that are vertices (the last partitions are padded)
  dvec<WT> pr, outw, part, x_pad;
  WT* x_own{nullptr};    // [L]        caller-owned: this rank's x = pr / out_w (input of the column all-gather)
  WT const* x_cols{nullptr};  // [R * L]  caller-owned: the gathered x of the column group
  WT* y_part{nullptr};   // [C * L]    caller-owned: partial row sums of the local block (input of the row reduce-scatter)
  WT const* y_own{nullptr};   // [L]      caller-owned: the reduced rows this rank owns
  double* triple{nullptr};    // [4]      caller-owned: this rank's (L1 change, dangling mass, max |x|, 0)
  dvec<pr_scalars<WT>> scal;
  dvec<double> tpartials;
  dvec<uint32_t> counters;
  std::shared_ptr<tiled_csc_t> tc;
  int epi_grid{1};

  pagerank_mg2d_plan(handle_t const& h_, graph_t& g_, double alpha_, int64_t L_, int64_t n_own_, int64_t rows_, int64_t cols_, int64_t nv_global_)
    : h(h_), g(g_), alpha((WT)alpha_), L(L_), n_own(n_own_), n_block_rows(rows_), n_block_cols(cols_), nv_global(nv_global_)
  {
    hp = &h_;
  }

  void create(device_array_view_t const* outw_own, device_array_view_t const* init_own, device_array_view_t const* x_own_v, device_array_view_t const* x_cols_v,
              device_array_view_t const* y_part_v, device_array_view_t const* y_own_v, device_array_view_t const* triple_v)
  {
    HIP_TRY(hipSetDevice(h.device));
    CGA_EXPECTS(n_own >= 0 && n_own <= L, CUGRAPH_INVALID_INPUT, "2-D multi-GPU PageRank: owned rows must not exceed rows_per_partition");
    auto typed = [&](device_array_view_t const* v, int64_t n) { return v != nullptr && v->type == g.weight_type && (int64_t)v->size == n; };
    CGA_EXPECTS(typed(outw_own, L) && typed(x_own_v, L) && typed(y_own_v, L) && typed(x_cols_v, n_block_cols) && typed(y_part_v, n_block_rows), CUGRAPH_INVALID_INPUT,
                "2-D multi-GPU PageRank: out_weight_sums / x_own / y_own need L values, x_cols R * L, y_part C * L, all of the weight type");
    CGA_EXPECTS(triple_v != nullptr && triple_v->type == FLOAT64 && triple_v->size == 4, CUGRAPH_INVALID_INPUT, "2-D multi-GPU PageRank: triple must hold 4 doubles");
    CGA_EXPECTS(n_block_rows <= g.nv && n_block_cols <= g.nv, CUGRAPH_INVALID_INPUT, "2-D multi-GPU PageRank: the local graph must have max(block rows, block columns) vertices");
    x_own = x_own_v->as<WT>(); x_cols = x_cols_v->as<WT const>(); y_part = y_part_v->as<WT>(); y_own = y_own_v->as<WT const>();
    triple = triple_v->as<double>();
    ensure_orientation(h, g, true, /*dcs_aware=*/true);  // (the block of the library's 2-D layout arrives in hypersparse form: mg_pagerank2d_part)
    orientation_t& o = g.csc;
    CGA_EXPECTS(o.seg[4] <= n_block_rows, CUGRAPH_INVALID_INPUT, "2-D multi-GPU PageRank: edges point to rows outside the local block");
    size_t const n1 = (size_t)(L > 0 ? L : 1);
    pr.resize_discard(n1); outw.resize_discard(n1);
    scal.resize_discard(1);
    HIP_TRY(hipMemsetAsync(scal.data(), 0, sizeof(pr_scalars<WT>), h.stream));
    HIP_TRY(hipMemsetAsync(triple, 0, 4 * sizeof(double), h.stream));
    if (L > 0) HIP_TRY(hipMemcpyAsync(outw.data(), outw_own->data, L * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    if (init_own) {
      CGA_EXPECTS(typed(init_own, L), CUGRAPH_INVALID_INPUT, "initial guess: one value per owned row");
      if (L > 0) HIP_TRY(hipMemcpyAsync(pr.data(), init_own->data, L * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    } else {
      fill_wt<WT>(h, pr.data(), n_own, WT(1) / (WT)nv_global);
    }
    // padded rows [n_own, L) are not vertices: they hold 0 and stay out of the dangling mass / L1 change (their x_own is 0 as well, so
    // the all-gathered column block reads 0 where no vertex is)
    if (L > n_own) {
      HIP_TRY(hipMemsetAsync(pr.data() + n_own, 0, (size_t)(L - n_own) * sizeof(WT), h.stream));
      HIP_TRY(hipMemsetAsync(x_own + n_own, 0, (size_t)(L - n_own) * sizeof(WT), h.stream));
    }
    int const T = tiled_default_T(h, sizeof(WT), std::max<int64_t>(n_block_cols, 1));
    if (!o.tiled || o.tiled->T != T || o.tiled->nv != n_block_rows) {
      auto t = std::make_shared<tiled_csc_t>();
      build_tiled_csc(h, g.nv, n_block_rows, g.ne, o, g.has_weights, sizeof(WT), T, *t);
      o.tiled = t;
    }
    tc = o.tiled;
    part.resize_discard((size_t)tc->n_slots + 64);
    HIP_TRY(hipMemsetAsync(part.data(), 0, ((size_t)tc->n_slots + 64) * sizeof(WT), h.stream));
    size_t const nx = (size_t)tc->nJ * tc->T + 8;  // the gather vector is read tile-wise: padded copy of x_cols
    x_pad.resize_discard(nx);
    HIP_TRY(hipMemsetAsync(x_pad.data(), 0, nx * sizeof(WT), h.stream));
    counters.resize_discard(4);
    HIP_TRY(hipMemsetAsync(counters.data(), 0, 4 * sizeof(uint32_t), h.stream));
    epi_grid = std::max(1, std::min(grid_for(n_own, 256, 1024), 1024));
    tpartials.resize_discard((size_t)3 * std::max({tc->nI, 1024, epi_grid}));
    h.sync();
  }

  tiled_epilogue<WT> epi()
  {
    tiled_epilogue<WT> e;
    e.nv = L; e.pr = pr.data(); e.x_next = x_own; e.outw = outw.data(); e.pers = nullptr; e.scal = scal.data();
    e.partials = tpartials.data(); e.totals = triple; e.alpha = alpha; e.nv_global = nv_global; e.wmax = tc->wmax;
    return e;
  }
  bool in_library_loop{false};  // pagerank_mgc2d_plan: exchange and compute share the handle's stream, nothing to synchronise between the phases
  void done() { if (!h.stream_borrowed && !in_library_loop) h.sync(); }  // the host layer's collectives run on its own stream -- unless it shares ours

  void start() override
  {  // x_own <- x of the initial vector, triple <- (0, partial dangling mass, max |x|)
    HIP_TRY(hipSetDevice(h.device));
    int const grid = std::max(1, std::min(grid_for(n_own, 256, 1024), 1024));
    hipLaunchKernelGGL(k_tiled_prologue_plain<WT>, grid, 256, 0, h.stream, (WT const*)pr.data(), (WT const*)outw.data(), x_own, n_own, tpartials.data());
    tiled_finish<WT>(h, epi(), grid);
    h.sync();
  }
  void set_scalars(void const* gathered, int nranks, bool read_back, double* diff, double* dangling) override
  {
    tiled_scalars_from_ranks<WT>(h, epi(), gathered, 0, 4 * sizeof(double), nranks);
    if (read_back) {
      pr_scalars<WT> sc;
      h.read_back(&sc, scal.data(), 1);
      if (diff) *diff = (double)sc.diff;
      if (dangling) *dangling = (double)sc.dangling;
    }
  }
  void spmv() override
  {
    HIP_TRY(hipSetDevice(h.device));
    if (n_block_cols > 0) HIP_TRY(hipMemcpyAsync(x_pad.data(), x_cols, n_block_cols * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    tiled_epilogue<WT> e = epi();
    e.nv    = n_block_rows;
    e.raw_y = y_part;
    tiled_phase1<WT>(h, *tc, (WT const*)x_pad.data(), alpha, part.data(), counters.data(), tiled_x_map<WT>{}, nullptr);
    tiled_phase2<WT>(h, *tc, (WT const*)part.data(), e, counters.data());
    done();
  }
  void epilogue() override
  {
    hipLaunchKernelGGL(k_mg2d_epilogue<WT>, epi_grid, 256, 0, h.stream, y_own, (WT const*)outw.data(), pr.data(), x_own, n_own, (pr_scalars<WT> const*)scal.data(),
                       tpartials.data());
    tiled_finish<WT>(h, epi(), epi_grid);  // this rank's (L1 change, dangling, max |x|) -> triple
    done();
  }
  void values(device_array_view_t const* out) override
  {
    CGA_EXPECTS(out != nullptr && (int64_t)out->size == L && out->type == g.weight_type, CUGRAPH_INVALID_INPUT, "values: one weight-typed value per owned row");
    if (L > 0) HIP_TRY(hipMemcpyAsync(out->data, pr.data(), L * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    h.sync();
  }
};

// =================================================================================================
// Multi-GPU PageRank behind the reference's own entry points: cugraph_pagerank on a graph from cugraph_graph_create_mg (a handle on the
// library's communicator, comm.hpp).  Same 1-D partition and the same tiled kernels as pagerank_mg_plan above; what changes is the exchange:
//   * the epilogue's x' is PUSHED: one kernel gathers x_own[send_index[k]] and stores it straight into each peer's receive window
//     through the peer-mapped pointers (xGMI stores; all links at once), the three scalars go into a [P][4] window the same way, then ONE
//     signal; the consumer waits on the sequence number.  No pack into a send buffer, no collective launch, no host in the loop;
//   * the receive window IS the gather vector: columns are numbered (owner, position) at partition time (mg_graph.hip), so there is no
//     unpack pass.  (Measured on the partition itself, RMAT-24 / 8 ranks: the owner-grouped numbering costs 16 % more partial sums,
//     +10 MB of the 138 MB a rank moves per iteration; the unpack pass it removes moved 42 MB: DESIGN.md section 5.)
//   * windows are double-buffered by iteration parity: a rank can be at most one push ahead of the slowest peer (its next iteration
//     waits for every peer's signal), so the buffer it overwrites has been consumed everywhere;
//   * the single-GPU plan's shortcuts apply: rows without in-edges stay out of the epilogue (their share of the scalars is analytic and
//     added to this rank's triple), fixed-count iterations neither read the previous iterate nor write pr.
// The iteration loop is this file's step(): Python is not in it.  Replaces the multi_gpu = true path of detail::pagerank
// (cpp/src/link_analysis/pagerank_impl.cuh:224-329) with update_edge_src_property's row broadcast (prims/update_edge_src_dst_property.cuh:550-579).
// =================================================================================================
// dst[i] = x_own[idx[i]] for the k-range of peer blockIdx.y: four independent gathers per thread (the index lists are ascending, so the
// gathers walk x_own forwards), coalesced stores into the peer's window
template <typename WT>
__global__ void __launch_bounds__(256) k_mgc_push(WT const* x_own, int32_t const* send_index, int64_t const* begin /*[P]*/, int64_t const* end /*[P]*/, WT* const* peer_x /*[P]*/,
                                                  int64_t const* dst_off /*[P]*/)
{
  int const r      = blockIdx.y;
  int64_t const k0 = begin[r], n = end[r] - k0;
  WT* dst          = peer_x[r] + dst_off[r];
  int32_t const* idx = send_index + k0;
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    int32_t const a = idx[i], b = idx[i + stride], c = idx[i + 2 * stride], d = idx[i + 3 * stride];
    WT const va = x_own[a], vb = x_own[b], vc = x_own[c], vd = x_own[d];
    dst[i] = va; dst[i + stride] = vb; dst[i + 2 * stride] = vc; dst[i + 3 * stride] = vd;
  }
  for (; i < n; i += stride) dst[i] = x_own[idx[i]];
}

// this rank's (L1 change, dangling mass, max |x|) -> slot `rank` of every peer's [P][4] window, then the iteration's signal: everything
// this stream wrote before (the pushed x) is released with it (comm.hpp: k-th signal = sequence number k in flags[channel][rank])
__global__ void k_mgc_scalars_signal(double const* totals, double* const* peer_s, int rank, int P, uint64_t* const* peer_flags, int channel, uint64_t seq)
{
  int const r = threadIdx.x;
  if (r < P) {
    double* d = peer_s[r] + 4 * rank;
    d[0] = totals[0]; d[1] = totals[1]; d[2] = totals[2]; d[3] = 0.0;
  }
  __threadfence_system();
  if (r < P) __hip_atomic_store(peer_flags[r] + (size_t)channel * kCommMaxRanks + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// waits until every rank's k-th push has landed (one lane per peer, bounded by the wall clock), then folds the P triples in rank order
// into the constants of the next iteration
template <typename WT>
__global__ void k_mgc_wait_fold(uint64_t const* my_flags, int channel, uint64_t seq, long long timeout_ticks, uint32_t* err, double const* triples /*[P][4]*/, int P,
                                pr_scalars<WT>* scal, WT alpha, int64_t nv_global, double wmax, int personalized)
{
  int const r = threadIdx.x;
  if (r < P) {
    uint64_t const* f  = my_flags + (size_t)channel * kCommMaxRanks + r;
    long long const t0 = wall_clock64();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > timeout_ticks) { __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
  }
  __threadfence_system();
  __syncthreads();
  if (r == 0) {
    double diff = 0, dang = 0, xmax = 0;
    for (int k = 0; k < P; ++k) {
      double const* t = triples + 4 * k;
      diff += __hip_atomic_load(t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      dang += __hip_atomic_load(t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      xmax = fmax(xmax, __hip_atomic_load(t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    }
    tiled_write_scalars<WT>(scal, diff, dang, xmax, alpha, nv_global, personalized, wmax);
  }
}

template <typename WT>
__global__ void k_fill_from_scal(WT* out, int64_t n, pr_scalars<WT> const* scal, int use_prev)
{
  WT const v     = use_prev ? scal->base_prev : scal->base;
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = v;
}

// external id -> local row of the ids this rank owns (-1 elsewhere)
__global__ void k_mgc_row_of(int32_t const* local_vertices, int64_t n_rows, int64_t vmin, int32_t* rowof)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) rowof[(int64_t)local_vertices[i] - vmin] = (int32_t)i;
}
// the (vertex, value) pairs of ALL ranks ([P][stride], counts[r] valid entries each): the pairs of the rows this rank owns go into `dense`
// (ADD: accumulated, scaled); *matched counts them -- over all ranks every pair must have found its owner
template <typename WT, bool ADD>
__global__ void k_mgc_pairs_to_rows(int32_t const* ids, WT const* vals, int64_t stride, int64_t const* counts, int P, int64_t vmin, int64_t vrange, int32_t const* rowof,
                                    WT* dense, WT scale, unsigned long long* matched)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < stride * P; i += (int64_t)gridDim.x * blockDim.x) {
    int const r     = (int)(i / stride);
    int64_t const k = i - (int64_t)r * stride;
    if (k >= counts[r]) continue;
    int64_t const x = (int64_t)ids[i] - vmin;
    if (x < 0 || x >= vrange) continue;
    int32_t const row = rowof[x];
    if (row < 0) continue;
    if constexpr (ADD) atomicAdd(&dense[row], vals[i] * scale);
    else dense[row] = vals[i];
    atomicAdd(matched, 1ull);
  }
}
__global__ void k_mark_rows(int32_t const* rows, int64_t n, uint32_t* flags)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) flags[rows[i]] = 1u;
}
template <typename T>
std::vector<T> mgc_to_host(handle_t const& h, T const* dev, size_t n)
{
  std::vector<T> out(n);
  if (n) HIP_TRY(hipMemcpyAsync(out.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost, h.stream));
  h.sync();
  return out;
}
template <typename T>
void mgc_to_device(handle_t const& h, dvec<T>& dev, std::vector<T> const& host)
{
  dev.resize_discard(host.size() ? host.size() : 1);
  if (host.size()) HIP_TRY(hipMemcpyAsync(dev.data(), host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, h.stream));
  h.sync();
}
// one rank: row r's value goes straight to where the rank's own window wants it
__global__ void k_self_columns(int32_t const* rows, int64_t n, int64_t first_col, int32_t* xcol)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) xcol[rows[i]] = (int32_t)(first_col + i);
}
// the referenced rows >= n_act, compacted: where their value goes (the row itself, or its window column when xcol is given) and their out-weight
template <typename WT>
__global__ void k_live_const_rows(uint32_t const* flags, uint32_t const* rank, int64_t n_act, int64_t n_rows, int32_t const* xcol, WT const* outw, int32_t* col_idx, WT* outw_c)
{
  int64_t r = n_act + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
    if (flags[r]) {
      uint32_t const j = rank[r] - rank[n_act];
      col_idx[j] = xcol ? xcol[r] : (int32_t)r;
      outw_c[j]  = outw[r];
    }
}

template <typename WT>
struct pagerank_mgc_plan : pagerank_plan_base {
  handle_t const& h;
  graph_t& g;
  comm_t& c;
  mg_pagerank_part_t* part{nullptr};
  WT alpha;
  int P, me;
  int64_t n_rows{0}, nv_global{0};
  dvec<WT> pr, x_own, partial, outw_live, pers, outw_own;
  dvec<int32_t> xcol_self, live_idx;
  bool personalized{false}, has_guess{false};
  bool direct{false};  // one rank: the epilogue writes x straight into the (own) gather window -- nothing to push
  WT const* outw{nullptr};
  dvec<pr_scalars<WT>> scal;
  dvec<double> tpartials, totals;
  dvec<uint32_t> counters;
  std::shared_ptr<tiled_csc_t> tc;
  tiled_const_rows<WT> crows;
  comm_window_t* xwin[2]{nullptr, nullptr};  // the gather vector of an iteration = what the peers pushed (double-buffered)
  comm_window_t* swin[2]{nullptr, nullptr};  // [P][4] doubles: every rank's (L1 change, dangling mass, max |x|)
  dvec<WT*> d_peer_x[2];
  dvec<double*> d_peer_s[2];
  dvec<int64_t> d_first, d_dst_off;
  int channel{0};
  uint64_t seq_base{0};  // c.seq[channel] when the plan took the channel: its k-th push carries sequence number seq_base + k
  uint64_t pushes{0}, folds{0};  // push #n fills buffer (n - 1) & 1 everywhere; fold #n waits for it
  size_t iterations{0};
  double last_diff{0};
  int64_t biggest_all{0};  // the longest per-peer send list
  // (The exchange in two chunks on a side stream -- hot rows pushed behind phase 2 of the hot destination tiles, phase 1 of the receiver split by source
  // tile -- was built in round 5, bit-identical, and cost more on one GPU (+4.5 %: two launches per phase over a third of a rank's share cannot fill the
  // chip) than the 0.07 ms of wire it could hide on eight; removed in round 6, numbers in profiles/r5r_*, r5s_*)

  pagerank_mgc_plan(handle_t const& h_, graph_t& g_, double alpha_) : h(h_), g(g_), c(*g_.mg->comm), alpha((WT)alpha_), P(g_.mg->comm->size), me(g_.mg->comm->rank) {}

  ~pagerank_mgc_plan() override
  {  // collective, like the constructor: every rank frees its plans in the same order
    try {
      (void)hipStreamSynchronize(h.stream);
      for (int b = 1; b >= 0; --b) { if (swin[b]) c.window_free(swin[b]); if (xwin[b]) c.window_free(xwin[b]); }
      if (channel >= 2) c.channel_free(channel);  // (64 channels per communicator: a loop of personalized PageRanks used to run out after ~60 calls)
    } catch (...) {
    }
  }

  // where iteration results go: next = the buffer the NEXT push fills
  tiled_epilogue<WT> epi()
  {
    tiled_epilogue<WT> e;
    e.nv = n_rows; e.pr = pr.data(); e.outw = outw; e.pers = personalized ? pers.data() : nullptr; e.scal = scal.data();
    e.partials = tpartials.data(); e.totals = totals.data(); e.alpha = alpha; e.nv_global = nv_global; e.wmax = tc->wmax;
    e.cr = crows;
    if (direct) { e.x_next = static_cast<WT*>(xwin[pushes & 1]->local); e.xcol = xcol_self.data(); }
    else e.x_next = x_own.data();
    return e;
  }

  // Collective.  Any rank may name any vertex (the reference shuffles such pairs to the owning GPU by vertex partition,
  // cpp/src/c_api/pagerank.cpp:140-196 shuffle_ext_vertex_value_pairs_to_local_gpu_by_vertex_partitioning): every rank's pairs are
  // all-gathered (padded to the longest list) and each rank keeps the pairs of the rows it owns.  ADD: duplicates accumulate; normalise:
  // values are divided by the sum of all values (which must be positive: the personalization vector, pagerank_impl.cuh:136-146).  A pair nobody owns is not a vertex of the graph: INVALID_INPUT on every rank.  Returns the sum of ALL values, added
  // in rank order (the same bits everywhere).
  template <bool ADD>
  double pairs_to_rows(device_array_view_t const* ids, device_array_view_t const* vals, WT* dense, WT fill, bool normalise, char const* what)
  {
    CGA_EXPECTS(ids->size == vals->size, CUGRAPH_INVALID_INPUT, std::string(what) + ": vertices and values differ in size");
    mg_graph_t& mg = *g.mg;
    fill_wt<WT>(h, dense, n_rows, fill);
    int64_t const mine = (int64_t)ids->size;
    std::vector<int64_t> counts(P);
    c.host_allgather(&mine, sizeof(mine), counts.data());
    int64_t stride = 0, total = 0;
    for (auto x : counts) { stride = std::max(stride, x); total += x; }
    dvec<double> dsum(1);
    HIP_TRY(hipMemsetAsync(dsum.data(), 0, sizeof(double), h.stream));
    if (mine > 0) hipLaunchKernelGGL(k_sum<WT>, grid_for(mine, kBlock, 1024), kBlock, 0, h.stream, vals->as<WT const>(), mine, dsum.data());
    double my_sum = 0;
    h.read_back(&my_sum, dsum.data(), 1);
    std::vector<double> sums(P);
    c.host_allgather(&my_sum, sizeof(my_sum), sums.data());
    double all_sum = 0;
    for (double x : sums) all_sum += x;
    if (normalise) CGA_EXPECTS((WT)all_sum > WT(0), CUGRAPH_INVALID_INPUT, "Invalid input argument: sum of personalization values should be positive.");
    WT const scale = normalise ? WT(1) / (WT)all_sum : WT(1);
    if (total == 0) return all_sum;
    size_t const w = (size_t)stride;
    dvec<int32_t> ids_in(w), ids_all(w * P);
    dvec<WT> vals_in(w), vals_all(w * P);
    HIP_TRY(hipMemsetAsync(ids_in.data(), 0, w * 4, h.stream));
    HIP_TRY(hipMemsetAsync(vals_in.data(), 0, w * sizeof(WT), h.stream));
    if (mine > 0) {
      HIP_TRY(hipMemcpyAsync(ids_in.data(), ids->data, (size_t)mine * 4, hipMemcpyDeviceToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(vals_in.data(), vals->data, (size_t)mine * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    }
    c.all_gather(h, ids_in.data(), w * 4, ids_all.data());
    c.all_gather(h, vals_in.data(), w * sizeof(WT), vals_all.data());
    dvec<int32_t> rowof((size_t)std::max<int64_t>(mg.vrange, 1));
    fill_i32(h, rowof.data(), std::max<int64_t>(mg.vrange, 1), -1);
    if (n_rows > 0) hipLaunchKernelGGL(k_mgc_row_of, grid_for(n_rows, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)part->local_vertices.data(), n_rows, mg.vmin, rowof.data());
    dvec<int64_t> d_counts(P);
    dvec<unsigned long long> d_matched(1);
    HIP_TRY(hipMemcpyAsync(d_counts.data(), counts.data(), (size_t)P * sizeof(int64_t), hipMemcpyHostToDevice, h.stream));
    HIP_TRY(hipMemsetAsync(d_matched.data(), 0, sizeof(unsigned long long), h.stream));
    hipLaunchKernelGGL((k_mgc_pairs_to_rows<WT, ADD>), grid_for(stride * P, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)ids_all.data(), (WT const*)vals_all.data(), stride,
                       (int64_t const*)d_counts.data(), P, mg.vmin, mg.vrange, (int32_t const*)rowof.data(), dense, scale, d_matched.data());
    unsigned long long matched = 0;
    h.read_back(&matched, d_matched.data(), 1);
    std::vector<unsigned long long> all_matched(P);
    c.host_allgather(&matched, sizeof(matched), all_matched.data());
    unsigned long long found = 0;
    for (auto x : all_matched) found += x;
    CGA_EXPECTS(found == (unsigned long long)total, CUGRAPH_INVALID_INPUT, std::string(what) + ": found a vertex id that is not in the graph");
    return all_sum;
  }

  void create(device_array_view_t const* ow_v = nullptr, device_array_view_t const* ow_s = nullptr, device_array_view_t const* ig_v = nullptr,
              device_array_view_t const* ig_s = nullptr, device_array_view_t const* p_v = nullptr, device_array_view_t const* p_s = nullptr)
  {
    HIP_TRY(hipSetDevice(h.device));
    part      = &mg_pagerank_part(h, g);  // collective on first use
    n_rows    = part->n_rows;
    nv_global = part->nv_global;
    direct    = P == 1 && getenv("CUGRAPH_AMD_MG_PUSH_SELF") == nullptr;
    graph_t& lg      = *reinterpret_cast<graph_t*>(part->local);
    CGA_EXPECTS(lg.weight_type == g.weight_type, CUGRAPH_UNKNOWN_ERROR, "multi-GPU PageRank: local graph of another weight type");
    ensure_orientation(h, lg, true);
    orientation_t& o = lg.csc;
    CGA_EXPECTS(o.seg[4] <= n_rows, CUGRAPH_UNKNOWN_ERROR, "multi-GPU PageRank: edges point to rows outside the local range");
    int const T = tiled_default_T(h, sizeof(WT), std::max<int64_t>(part->ncols, 1));
    if (!o.tiled || o.tiled->T != T || o.tiled->nv != n_rows) {
      auto t = std::make_shared<tiled_csc_t>();
      build_tiled_csc(h, lg.nv, n_rows, lg.ne, o, lg.has_weights, sizeof(WT), T, *t);
      o.tiled = t;
    }
    tc = o.tiled;
    size_t const n1 = (size_t)std::max<int64_t>(n_rows, 1);
    pr.resize_discard(n1); x_own.resize_discard(n1);
    outw = part->outw_local.template as<WT const>();
    scal.resize_discard(1); totals.resize_discard(4); counters.resize_discard(4);
    partial.resize_discard((size_t)tc->n_slots + 64);
    tpartials.resize_discard((size_t)3 * std::max(tc->nI, 1024));
    HIP_TRY(hipMemsetAsync(partial.data(), 0, ((size_t)tc->n_slots + 64) * sizeof(WT), h.stream));
    HIP_TRY(hipMemsetAsync(counters.data(), 0, 4 * sizeof(uint32_t), h.stream));
    HIP_TRY(hipMemsetAsync(totals.data(), 0, 4 * sizeof(double), h.stream));
    HIP_TRY(hipMemsetAsync(x_own.data(), 0, n1 * sizeof(WT), h.stream));
    fill_wt<WT>(h, pr.data(), n_rows, nv_global > 0 ? WT(1) / (WT)nv_global : WT(0));  // pagerank_impl.cuh:422-426
    // the optional (vertices, values) arguments, as in the single-GPU plan (pagerank_plan::create); collective: every rank passes all or none
    if (ow_s) {
      outw_own.resize_discard(n1);
      (void)pairs_to_rows<false>(ow_v, ow_s, outw_own.data(), WT(0), false, "precomputed_vertex_out_weight");
      outw = outw_own.data();
    }
    if (ig_s) {
      has_guess = true;
      (void)pairs_to_rows<false>(ig_v, ig_s, pr.data(), WT(0), false, "initial_guess");  // not renormalised: pagerank_impl.cuh:427-432
    }
    if (p_s) {
      personalized = true;
      pers.resize_discard(n1);
      (void)pairs_to_rows<true>(p_v, p_s, pers.data(), WT(0), true, "personalization");
    }
    pr_scalars<WT> s0{};
    s0.base = nv_global > 0 ? WT(1) / (WT)nv_global : WT(0);  // the first fold makes it base_prev: what the rows hold before iteration 1
    HIP_TRY(hipMemcpyAsync(scal.data(), &s0, sizeof(s0), hipMemcpyHostToDevice, h.stream));
    h.sync();
    if (direct) {  // the row -> window column map of the rank's own segment
      xcol_self.resize_discard(n1);
      fill_i32(h, xcol_self.data(), (int64_t)n1, -1);
      if (part->n_send > 0)
        hipLaunchKernelGGL(k_self_columns, grid_for(part->n_send, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)part->send_index.data(), part->n_send, part->dst_off[0], xcol_self.data());
    }
    // rows without in-edges leave the per-iteration epilogue (spmv_tiled.hpp: tiled_const_rows): their share of the scalars is analytic, and
    // only those of them that SOME rank references (an entry of send_index) get their x = base / out_w written at all
    if (!getenv("CUGRAPH_AMD_PAGERANK_ALL_ROWS") && !personalized && !has_guess && tc->n_act < n_rows && tc->nI_act > 0) {
      int64_t const n = n_rows - tc->n_act;
      crows.n_rows = n; crows.c0 = 0; crows.nI_act = tc->nI_act;
      dvec<unsigned long long> red(2);
      dvec<WT> scratch((size_t)n);
      HIP_TRY(hipMemsetAsync(red.data(), 0, 2 * sizeof(unsigned long long), h.stream));
      hipLaunchKernelGGL(k_const_rows_setup<WT>, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, outw + tc->n_act, (int32_t const*)nullptr, n, (int64_t)tc->n_act, (int64_t)tc->n_act,
                         scratch.data(), red.data());
      unsigned long long r[2];
      h.read_back(r, red.data(), 2);
      crows.n_dangling = (int64_t)r[0];
      std::memcpy(&crows.max_inv_outw, &r[1], sizeof(double));
      dvec<uint32_t> flags((size_t)n_rows + 1), rank((size_t)n_rows + 1);
      HIP_TRY(hipMemsetAsync(flags.data(), 0, ((size_t)n_rows + 1) * 4, h.stream));
      if (part->n_send > 0) hipLaunchKernelGGL(k_mark_rows, grid_for(part->n_send, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)part->send_index.data(), part->n_send, flags.data());
      exclusive_scan_u32(h, flags.data(), rank.data(), n_rows + 1);
      uint32_t ends[2];
      HIP_TRY(hipMemcpyAsync(&ends[0], rank.data() + tc->n_act, 4, hipMemcpyDeviceToHost, h.stream));
      HIP_TRY(hipMemcpyAsync(&ends[1], rank.data() + n_rows, 4, hipMemcpyDeviceToHost, h.stream));
      h.sync();
      int64_t const n_live = (int64_t)ends[1] - (int64_t)ends[0];
      live_idx.resize_discard((size_t)std::max<int64_t>(n_live, 1));
      outw_live.resize_discard((size_t)std::max<int64_t>(n_live, 1));
      if (n_live > 0)
        hipLaunchKernelGGL(k_live_const_rows<WT>, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)flags.data(), (uint32_t const*)rank.data(), (int64_t)tc->n_act, n_rows,
                           direct ? (int32_t const*)xcol_self.data() : (int32_t const*)nullptr, outw, live_idx.data(), outw_live.data());
      h.sync();
      crows.n_cols  = n_live;
      crows.col_idx = live_idx.data();
      crows.outw_c  = outw_live.data();
    }
    // windows (collective): zeroed before anybody may push into them
    channel         = c.channel_alloc();
    seq_base        = c.seq[channel];
    size_t const nx = (size_t)tc->nJ * tc->T + 8;  // the gather vector is read tile-wise
    for (int b = 0; b < 2; ++b) {
      xwin[b] = c.window_create(nx * sizeof(WT));
      swin[b] = c.window_create((size_t)P * 4 * sizeof(double));
      HIP_TRY(hipMemsetAsync(xwin[b]->local, 0, nx * sizeof(WT), h.stream));
      HIP_TRY(hipMemsetAsync(swin[b]->local, 0, (size_t)P * 4 * sizeof(double), h.stream));
      d_peer_x[b].resize_discard(P); d_peer_s[b].resize_discard(P);
      HIP_TRY(hipMemcpyAsync(d_peer_x[b].data(), xwin[b]->peer.data(), (size_t)P * sizeof(void*), hipMemcpyHostToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(d_peer_s[b].data(), swin[b]->peer.data(), (size_t)P * sizeof(void*), hipMemcpyHostToDevice, h.stream));
    }
    d_first.resize_discard(P + 1); d_dst_off.resize_discard(P);
    HIP_TRY(hipMemcpyAsync(d_first.data(), part->send_first.data(), (size_t)(P + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h.stream));
    HIP_TRY(hipMemcpyAsync(d_dst_off.data(), part->dst_off.data(), (size_t)P * sizeof(int64_t), hipMemcpyHostToDevice, h.stream));
    h.sync();
    for (int r = 0; r < P; ++r) biggest_all = std::max(biggest_all, part->send_first[r + 1] - part->send_first[r]);
    c.host_barrier();
    // iteration-0 state: x = pr / out_w of every owned row, this rank's (0, dangling mass, max |x|); first push
    tiled_epilogue<WT> const e0 = epi();
    int const n = tiled_prologue<WT>(h, *tc, (WT const*)pr.data(), outw, e0.x_next, n_rows, tpartials.data(), direct ? (int32_t const*)xcol_self.data() : (int32_t const*)nullptr);
    tiled_finish<WT>(h, e0, n, (double)s0.base);
    push();
    h.sync();
    c.check("multi-GPU PageRank: first exchange");
  }

  void push_part(hipStream_t s, int b, int64_t const* begin, int64_t const* end, int64_t const* dst, int64_t biggest)
  {
    if (direct || biggest <= 0) return;
    dim3 const grid((unsigned)std::max(1, std::min(grid_for((biggest + 3) / 4, 256, 4096), 4096)), (unsigned)P);
    hipLaunchKernelGGL(k_mgc_push<WT>, grid, dim3(256), 0, s, (WT const*)x_own.data(), (int32_t const*)part->send_index.data(), begin, end, (WT* const*)d_peer_x[b].data(), dst);
  }

  // the whole x, then the scalars and the push's signal
  void push()
  {
    int const b         = (int)(pushes & 1);
    hipStream_t const s = h.stream;
    push_part(s, b, d_first.data(), d_first.data() + 1, d_dst_off.data(), biggest_all);
    uint64_t const k = ++c.seq[channel];
    hipLaunchKernelGGL(k_mgc_scalars_signal, 1, 64, 0, s, (double const*)totals.data(), (double* const*)d_peer_s[b].data(), me, P, (uint64_t* const*)c.d_peer_flags, channel, k);
    ++pushes;
  }

  // waits for the latest push of every rank and folds the P scalar triples into the constants of the next iteration
  void fold(bool read_back)
  {
    if (folds < pushes) {
      int const b = (int)(folds & 1);
      long long const ticks = (long long)(c.timeout_s * (double)c.wall_ticks_per_s);
      // (the channel's sequence numbers continue where its previous owner stopped: channels are handed out again, comm_t::channel_free)
      hipLaunchKernelGGL(k_mgc_wait_fold<WT>, 1, 64, 0, h.stream, (uint64_t const*)c.flags->local, channel, seq_base + folds + 1, ticks, c.err_word, (double const*)swin[b]->local, P, scal.data(),
                         alpha, nv_global, tc->wmax, personalized ? 1 : 0);
      ++folds;
    }
    if (read_back) {
      pr_scalars<WT> sc;
      h.read_back(&sc, scal.data(), 1);
      c.check("multi-GPU PageRank");
      last_diff = (double)sc.diff;
    }
  }

  void iterate(bool need_diff, bool last_of_call)
  {
    int const b          = (int)((folds - 1) & 1);
    tiled_epilogue<WT> e = epi();
    e.need_diff          = need_diff;
    e.write_pr           = need_diff || last_of_call;
    tiled_phase1<WT>(h, *tc, (WT const*)xwin[b]->local, alpha, partial.data(), counters.data(), tiled_x_map<WT>{}, nullptr);
    tiled_phase2<WT>(h, *tc, (WT const*)partial.data(), e, counters.data());
    tiled_finish<WT>(h, e, tiled_fold_count(*tc, e));  // this rank's triple (with the analytic share of the rows left out)
    push();
  }

  void step(double epsilon, size_t max_iterations, size_t* done, bool* converged) override
  {
    HIP_TRY(hipSetDevice(h.device));
    bool const track = epsilon > 0.0;
    size_t it = 0;
    bool conv = false;
    while (it < max_iterations) {
      fold(track);
      if (track && iterations > 0 && last_diff < epsilon) { conv = true; break; }  // pagerank_impl.cuh:320-326, on the GLOBAL L1 change
      iterate(track, it + 1 == max_iterations);
      ++iterations; ++it;
    }
    if (track && !conv) { fold(true); conv = last_diff < epsilon; }  // did the last allowed iteration converge?
    *done      = it;
    *converged = conv;
  }

  centrality_result_t* result(size_t total_iterations, bool converged) override
  {
    auto ids  = std::make_unique<device_array_t>((size_t)n_rows, INT32);
    auto vals = std::make_unique<device_array_t>((size_t)n_rows, g.weight_type);
    if (crows.nI_act > 0)  // the rows without in-edges hold the base of the last iteration (base_prev once its scalars have been folded)
      hipLaunchKernelGGL(k_fill_from_scal<WT>, grid_for(crows.n_rows, kBlock, 4096), kBlock, 0, h.stream, pr.data() + tc->n_act, crows.n_rows,
                         (pr_scalars<WT> const*)scal.data(), (folds == pushes && iterations > 0) ? 1 : 0);
    if (n_rows > 0) {
      HIP_TRY(hipMemcpyAsync(ids->buf.ptr, part->local_vertices.data(), (size_t)n_rows * 4, hipMemcpyDeviceToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(vals->buf.ptr, pr.data(), (size_t)n_rows * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    }
    h.sync();
    c.check("multi-GPU PageRank result");
    device_array_t* idp = ids.release();
    outer_replace_ids(h, g, idp);  // INT64 ids: the caller's id space (no-op otherwise)
    return new centrality_result_t{idp, vals.release(), total_iterations, converged};
  }
};

// =================================================================================================
// The reference's 2-D layout behind cugraph_graph_create_mg + cugraph_pagerank (round 6; CUGRAPH_AMD_MG_LAYOUT=2d on every rank): the block of
// mg_pagerank2d_part_t (mg_graph.hpp), the kernels of pagerank_mg2d_plan, and the exchange on the library's communicator --
//   column group {c * R + r'}: every member PUSHES its x_own [L] into slot r of the members' x windows [R * L]      (the row broadcast of
//                              update_edge_src_property, prims/update_edge_src_dst_property.cuh:550-579)
//   row group {c' * R + r}:    every member pushes chunk c' of its partial rows [C * L] into slot c of member c''s y window [C * L]; the owner adds its C
//                              slots in slot order (the column reduction of prims/detail/per_v_transform_reduce_e.cuh:3390-3406; fixed order = the same bits in every run)
//   all ranks:                 the scalar triples, into slot `rank` of everybody's [P][4] window, folded in rank order
// all as peer-mapped stores over xGMI (comm_t::push_multi), one signal + wait per exchange step, no host in the loop.  One buffer per window suffices: a
// rank overwrites a peer's x / scalar slot only after its own epilogue, i.e. after every rank has signalled the y exchange of the iteration that read them,
// and a y slot only after the x exchange of the next iteration, which every rank signals after its epilogue has consumed the y window.
// Per rank and iteration (R - 1 + C - 1) * L values travel -- 4 * V / 8 * 4 B = 134 MB at RMAT-26 on 2 x 4 ranks against the 1-D
// layout's sparse exchange (DESIGN.md section 5 has both byte counts).  Plain PageRank only: a call with precomputed out-weights, an initial guess or a
// personalization vector runs on the 1-D partition.
// =================================================================================================
template <typename WT>
__global__ void __launch_bounds__(256) k_mgc2d_reduce_rows(WT const* ywin, int C, int64_t L, WT* y_own)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < L; i += (int64_t)gridDim.x * blockDim.x) {
    WT sum = ywin[i];
    for (int cc = 1; cc < C; ++cc) sum += ywin[(int64_t)cc * L + i];
    y_own[i] = sum;
  }
}

template <typename WT>
struct pagerank_mgc2d_plan : pagerank_plan_base {
  handle_t const& h;
  graph_t& g;
  comm_t& c;
  mg_pagerank2d_part_t* part{nullptr};
  double alpha;
  std::unique_ptr<pagerank_mg2d_plan<WT>> inner;
  dvec<WT> x_own, y_part, y_own;
  dvec<double> triple;
  comm_window_t* xwin{nullptr};  // [R * L] the gathered x of the column group
  comm_window_t* ywin{nullptr};  // [C * L] the row group's partial sums of the owned rows, one slot per sender
  comm_window_t* swin{nullptr};  // [P][4] doubles
  int channel{0};
  uint64_t pending_seq{0};  // the signal of the latest x / scalar push (0: none outstanding)
  bool folded{true};        // its scalars have been folded into the iteration constants
  size_t iterations{0};
  double last_diff{0};

  pagerank_mgc2d_plan(handle_t const& h_, graph_t& g_, double alpha_) : h(h_), g(g_), c(*g_.mg->comm), alpha(alpha_) {}
  ~pagerank_mgc2d_plan() override
  {  // collective, like create(): every rank frees in the same order
    try {
      (void)hipStreamSynchronize(h.stream);
      inner.reset();
      if (swin) c.window_free(swin);
      if (ywin) c.window_free(ywin);
      if (xwin) c.window_free(xwin);
      if (channel >= 2) c.channel_free(channel);
    } catch (...) {
    }
  }

  void create()
  {
    HIP_TRY(hipSetDevice(h.device));
    part = &mg_pagerank2d_part(h, g);
    int const R = part->R, C = part->C, P = part->P;
    int64_t const L = part->L;
    CGA_EXPECTS(R + P <= kCommMaxRanks, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "2-D multi-GPU PageRank: too many ranks for one push");
    x_own.resize_discard((size_t)L); y_part.resize_discard((size_t)C * L); y_own.resize_discard((size_t)L); triple.resize_discard(4);
    HIP_TRY(hipMemsetAsync(x_own.data(), 0, (size_t)L * sizeof(WT), h.stream));
    HIP_TRY(hipMemsetAsync(y_part.data(), 0, (size_t)C * L * sizeof(WT), h.stream));
    HIP_TRY(hipMemsetAsync(y_own.data(), 0, (size_t)L * sizeof(WT), h.stream));
    channel = c.channel_alloc();
    xwin    = c.window_create((size_t)R * L * sizeof(WT));
    ywin    = c.window_create((size_t)C * L * sizeof(WT));
    swin    = c.window_create((size_t)P * 4 * sizeof(double));
    HIP_TRY(hipMemsetAsync(xwin->local, 0, (size_t)R * L * sizeof(WT), h.stream));
    HIP_TRY(hipMemsetAsync(ywin->local, 0, (size_t)C * L * sizeof(WT), h.stream));
    HIP_TRY(hipMemsetAsync(swin->local, 0, (size_t)P * 4 * sizeof(double), h.stream));
    h.sync();
    c.host_barrier();  // nobody pushes into a window that is still being cleared
    graph_t& block = G(part->local);
    cugraph_data_type_id_t const wt = sizeof(WT) == 8 ? FLOAT64 : FLOAT32;
    inner = std::make_unique<pagerank_mg2d_plan<WT>>(h, block, alpha, L, part->n_own, (int64_t)C * L, (int64_t)R * L, part->nv_global);
    inner->in_library_loop = true;
    device_array_view_t ow{part->outw_own.ptr, (size_t)L, wt}, xo{x_own.data(), (size_t)L, wt}, xc{xwin->local, (size_t)R * L, wt}, yp{y_part.data(), (size_t)C * L, wt},
      yo{y_own.data(), (size_t)L, wt}, tv{triple.data(), 4, FLOAT64};
    // (an unweighted multi-GPU graph computes in fp32: its block is created without weights and reports FLOAT32)
    inner->create(&ow, nullptr, &xo, &xc, &yp, &yo, &tv);
    inner->start();  // x_own, triple of the initial vector
    push_x_and_scalars();
  }

  void push_x_and_scalars()
  {
    comm_push_desc_t d{};
    int const R = part->R, P = part->P;
    int64_t const L = part->L;
    int n = 0;
    for (int rr = 0; rr < R; ++rr, ++n) {  // my x into slot r of every member of my column group (myself included)
      int const peer = part->c * R + rr;
      d.dst[n] = static_cast<char*>(xwin->peer[peer]) + (size_t)part->r * L * sizeof(WT);
      d.src[n] = x_own.data();
      d.words[n] = (int64_t)(L * sizeof(WT) / 4);
    }
    for (int p = 0; p < P; ++p, ++n) {
      d.dst[n] = static_cast<char*>(swin->peer[p]) + (size_t)part->rank * 4 * sizeof(double);
      d.src[n] = triple.data();
      d.words[n] = 8;
    }
    d.n = n;
    c.push_multi(h.stream, d);
    pending_seq = c.signal(h.stream, channel);
    folded      = false;
  }
  void fold(bool read_back)
  {
    if (!folded) {
      c.wait(h.stream, channel, pending_seq);
      double diff = 0, dang = 0;
      inner->set_scalars(swin->local, part->P, read_back, &diff, &dang);
      if (read_back) { c.check("2-D multi-GPU PageRank"); last_diff = diff; }
      folded = true;
    } else if (read_back) {  // (folded already: only the L1 change is wanted)
      pr_scalars<WT> sc;
      h.read_back(&sc, inner->scal.data(), 1);
      last_diff = (double)sc.diff;
    }
  }
  void iterate()
  {
    int const R = part->R, C = part->C;
    int64_t const L = part->L;
    inner->spmv();  // y_part <- block x (alpha * x window)
    comm_push_desc_t d{};
    for (int cc = 0; cc < C; ++cc) {  // chunk cc of my partial rows into slot c of the owner's y window
      int const peer = cc * R + part->r;
      d.dst[cc] = static_cast<char*>(ywin->peer[peer]) + (size_t)part->c * L * sizeof(WT);
      d.src[cc] = y_part.data() + (size_t)cc * L;
      d.words[cc] = (int64_t)(L * sizeof(WT) / 4);
    }
    d.n = C;
    c.push_multi(h.stream, d);
    c.wait(h.stream, channel, c.signal(h.stream, channel));
    hipLaunchKernelGGL(k_mgc2d_reduce_rows<WT>, grid_for(L, 256, 2048), 256, 0, h.stream, (WT const*)ywin->local, C, L, y_own.data());
    inner->epilogue();  // pr, x_own, triple
    push_x_and_scalars();
  }

  void step(double epsilon, size_t max_iterations, size_t* done, bool* converged) override
  {
    HIP_TRY(hipSetDevice(h.device));
    bool const track = epsilon > 0.0;
    size_t it = 0;
    bool conv = false;
    while (it < max_iterations) {
      fold(track);
      if (track && iterations > 0 && last_diff < epsilon) { conv = true; break; }  // pagerank_impl.cuh:320-326, on the GLOBAL L1 change
      iterate();
      ++iterations; ++it;
    }
    if (track && !conv) { fold(true); conv = last_diff < epsilon; }
    *done      = it;
    *converged = conv;
  }

  centrality_result_t* result(size_t total_iterations, bool converged) override
  {
    int64_t const n = part->n_own, L = part->L;
    auto ids  = std::make_unique<device_array_t>((size_t)n, INT32);
    auto vals = std::make_unique<device_array_t>((size_t)n, g.weight_type);
    dvec<WT> all((size_t)L);
    cugraph_data_type_id_t const wt = sizeof(WT) == 8 ? FLOAT64 : FLOAT32;
    device_array_view_t av{all.data(), (size_t)L, wt};
    inner->values(&av);
    if (n > 0) {
      HIP_TRY(hipMemcpyAsync(ids->buf.ptr, part->local_vertices.data(), (size_t)n * 4, hipMemcpyDeviceToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(vals->buf.ptr, all.data(), (size_t)n * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
    }
    h.sync();
    c.check("2-D multi-GPU PageRank result");
    device_array_t* idp = ids.release();
    outer_replace_ids(h, g, idp);
    return new centrality_result_t{idp, vals.release(), total_iterations, converged};
  }
};

namespace {

void check_pair_types(graph_t const& g, device_array_view_t const* v, device_array_view_t const* s, char const* vmsg, char const* smsg)
{
  if (v == nullptr && s == nullptr) return;
  CGA_EXPECTS(v != nullptr && s != nullptr, CUGRAPH_INVALID_INPUT, std::string(vmsg) + " (vertices and values must both be given)");
  CGA_EXPECTS(g.api_vertex_type() == v->type, CUGRAPH_INVALID_INPUT, vmsg);
  CGA_EXPECTS(g.weight_type == s->type, CUGRAPH_INVALID_INPUT, smsg);
}

pagerank_plan_base* make_plan(cugraph_resource_handle_t const* handle, cugraph_graph_t* graph,
                              cugraph_type_erased_device_array_view_t const* ow_v, cugraph_type_erased_device_array_view_t const* ow_s,
                              cugraph_type_erased_device_array_view_t const* ig_v, cugraph_type_erased_device_array_view_t const* ig_s,
                              cugraph_type_erased_device_array_view_t const* p_v, cugraph_type_erased_device_array_view_t const* p_s, double alpha,
                              bool do_expensive_check = false)
{
  handle_t const& h = H(handle);
  graph_t& g        = GM(graph);
  // pagerank_impl.cuh:78-88 (a multi-GPU graph checks it with the other rank-local arguments below: every rank fails or none does)
  if (!g.mg) CGA_EXPECTS(alpha >= 0.0 && alpha <= 1.0, CUGRAPH_INVALID_INPUT, "Invalid input argument: alpha should be in [0.0, 1.0].");
  if (g.mg) {  // a graph from cugraph_graph_create_mg on a communicator handle: collective, every rank gets its owned vertices back
    CGA_EXPECTS(handle_comm(h) == g.mg->comm, CUGRAPH_INVALID_HANDLE, "multi-GPU PageRank: the handle is not on the communicator the graph was created on");
    mg_agree(g, [&] {  // rank-local argument checks: every rank fails or none does
      CGA_EXPECTS(alpha >= 0.0 && alpha <= 1.0, CUGRAPH_INVALID_INPUT, "Invalid input argument: alpha should be in [0.0, 1.0].");
      check_pair_types(g, V(ow_v), V(ow_s), "vertex type of graph and precomputed_vertex_out_weight_vertices must match",
                       "vertex type of graph and precomputed_vertex_out_weight_sums must match");
      check_pair_types(g, V(ig_v), V(ig_s), "vertex type of graph and initial_guess_vertices must match", "vertex type of graph and initial_guess_values must match");
      check_pair_types(g, V(p_v), V(p_s), "vertex type of graph and personalization_vector must match", "vertex type of graph and personalization_vector must match");
    }, "cugraph_pagerank");
    mg_agree_same(g, &alpha, sizeof(alpha), "cugraph_pagerank (alpha)");
    // INT64 ids of a multi-GPU graph: the vertex columns become compact int32 ids (outer_ids.hip); ids that are no vertices map to -1 and are
    // rejected by create() exactly as unknown int32 ids are; the result's vertex column goes back through outer_replace_ids (pagerank_mgc_plan::result)
    vertex_column_in mc_ow, mc_ig, mc_p;
    device_array_view_t const* mow = mc_ow.get(h, g, V(ow_v), "precomputed_vertex_out_weight_vertices");
    device_array_view_t const* mig = mc_ig.get(h, g, V(ig_v), "initial_guess_vertices");
    device_array_view_t const* mpv = mc_p.get(h, g, V(p_v), "personalization_vector");
    {  // the layout: the ranks must agree (a collective call); the optional arguments run on the 1-D partition
      char const* env_l = getenv("CUGRAPH_AMD_MG_LAYOUT");
      int64_t const want2d = env_l != nullptr && std::string(env_l) == "2d" ? 1 : 0;
      int64_t const plain  = (mow == nullptr && mig == nullptr && mpv == nullptr) ? 1 : 0;
      int64_t const hi = mg_host_max(g, want2d * 2 + plain), lo = -mg_host_max(g, -(want2d * 2 + plain));
      CGA_EXPECTS(hi == lo, CUGRAPH_INVALID_INPUT, "multi-GPU PageRank: the ranks disagree on CUGRAPH_AMD_MG_LAYOUT or on which optional arguments they pass");
      if (want2d && plain) {
        if (g.weight_type == FLOAT64) {
          auto p = std::make_unique<pagerank_mgc2d_plan<double>>(h, g, alpha);
          p->create();
          return p.release();
        }
        auto p = std::make_unique<pagerank_mgc2d_plan<float>>(h, g, alpha);
        p->create();
        return p.release();
      }
    }
    if (g.weight_type == FLOAT64) {
      auto p = std::make_unique<pagerank_mgc_plan<double>>(h, g, alpha);
      p->create(mow, V(ow_s), mig, V(ig_s), mpv, V(p_s));
      return p.release();
    }
    auto p = std::make_unique<pagerank_mgc_plan<float>>(h, g, alpha);
    p->create(mow, V(ow_s), mig, V(ig_s), mpv, V(p_s));
    return p.release();
  }
  CGA_EXPECTS(p_v == nullptr || V(p_v)->size > 0, CUGRAPH_INVALID_INPUT,
              "Invalid input argument: if personalizations.has_value() is true, the input personalization vector size should not be 0.");
  if (do_expensive_check) {  // pagerank_impl.cuh:90-117
    auto negatives = [&](device_array_view_t const* v) -> int64_t {
      if (v == nullptr || v->size == 0) return 0;
      return v->type == FLOAT64 ? count_negative_f64(h, v->as<double const>(), (int64_t)v->size) : count_negative_f32(h, v->as<float const>(), (int64_t)v->size);
    };
    CGA_EXPECTS(negatives(V(ow_s)) == 0, CUGRAPH_INVALID_INPUT, "Invalid input argument: outgoing edge weight sum values should be non-negative.");
    if (g.has_weights && g.ne > 0) {
      orientation_t const& o = g.csc.built ? g.csc : g.csr;
      device_array_view_t wv{o.weights.ptr, (size_t)g.ne, g.weight_type};
      CGA_EXPECTS(negatives(&wv) == 0, CUGRAPH_INVALID_INPUT, "Invalid input argument: input edge weights should have non-negative values.");
    }
  }
  // messages as cpp/src/c_api/pagerank.cpp:260-296
  check_pair_types(g, V(ow_v), V(ow_s), "vertex type of graph and precomputed_vertex_out_weight_vertices must match",
                   "vertex type of graph and precomputed_vertex_out_weight_sums must match");
  check_pair_types(g, V(ig_v), V(ig_s), "vertex type of graph and initial_guess_vertices must match",
                   "vertex type of graph and initial_guess_values must match");
  check_pair_types(g, V(p_v), V(p_s), "vertex type of graph and personalization_vector must match",
                   "vertex type of graph and personalization_vector must match");
  // INT64 / sparse external ids: the vertex columns become compact int32 ids (outer_ids.hip); ids that are not vertices map
  // to -1 and are rejected by create() exactly as unknown int32 ids are
  vertex_column_in c_ow, c_ig, c_p;
  device_array_view_t const* ow = c_ow.get(h, g, V(ow_v), "precomputed_vertex_out_weight_vertices");
  device_array_view_t const* ig = c_ig.get(h, g, V(ig_v), "initial_guess_vertices");
  device_array_view_t const* pv = c_p.get(h, g, V(p_v), "personalization_vector");
  if (g.weight_type == FLOAT64) {
    auto p = std::make_unique<pagerank_plan<double>>(h, g, alpha);
    p->create(ow, V(ow_s), ig, V(ig_s), pv, V(p_s));
    return p.release();
  }
  auto p = std::make_unique<pagerank_plan<float>>(h, g, alpha);
  p->create(ow, V(ow_s), ig, V(ig_s), pv, V(p_s));
  return p.release();
}

cugraph_error_code_t run_pagerank(cugraph_resource_handle_t const* handle, cugraph_graph_t* graph,
                                  cugraph_type_erased_device_array_view_t const* ow_v, cugraph_type_erased_device_array_view_t const* ow_s,
                                  cugraph_type_erased_device_array_view_t const* ig_v, cugraph_type_erased_device_array_view_t const* ig_s,
                                  cugraph_type_erased_device_array_view_t const* p_v, cugraph_type_erased_device_array_view_t const* p_s, double alpha,
                                  double epsilon, size_t max_iterations, bool must_converge, cugraph_centrality_result_t** result,
                                  cugraph_error_t** error, bool do_expensive_check = false)
{
  if (result) *result = nullptr;
  bool converged = false;
  cugraph_error_code_t rc = guarded(error, [&] {
    CGA_EXPECTS(result != nullptr, CUGRAPH_INVALID_INPUT, "result is NULL");
    if (graph != nullptr && reinterpret_cast<graph_t*>(graph)->mg) {  // collective call: the scalar arguments are checked and compared on every rank together
      graph_t& gm = *reinterpret_cast<graph_t*>(graph);
      mg_agree(gm, [&] { CGA_EXPECTS(epsilon >= 0.0, CUGRAPH_INVALID_INPUT, "Invalid input argument: epsilon should be non-negative."); }, "cugraph_pagerank");
      uint64_t scalars[2] = {0, (uint64_t)max_iterations};
      std::memcpy(&scalars[0], &epsilon, sizeof(double));
      mg_agree_same(gm, scalars, sizeof(scalars), "cugraph_pagerank (epsilon, max_iterations)");
    }
    CGA_EXPECTS(epsilon >= 0.0, CUGRAPH_INVALID_INPUT, "Invalid input argument: epsilon should be non-negative.");  // pagerank_impl.cuh:88
    std::unique_ptr<pagerank_plan_base> plan(make_plan(handle, graph, ow_v, ow_s, ig_v, ig_s, p_v, p_s, alpha, do_expensive_check));
    size_t done = 0;
    // pagerank_impl.cuh:224-329: the loop body runs at least once; `converged` = iter < max_iterations
    plan->step(epsilon, max_iterations > 0 ? max_iterations : 1, &done, &converged);
    converged = done < max_iterations;
    *result   = reinterpret_cast<cugraph_centrality_result_t*>(plan->result(done, converged));
  });
  if (rc == CUGRAPH_SUCCESS && must_converge && !converged) {  // pagerank.cpp:306-313
    if (error) *error = reinterpret_cast<cugraph_error_t*>(new err_obj_t{"PageRank failed to converge."});
    return CUGRAPH_UNKNOWN_ERROR;
  }
  return rc;
}

}  // namespace
}  // namespace cga

using namespace cga;

extern "C" cugraph_error_code_t cugraph_pagerank(CUGRAPH_PAGERANK_COMMON_ARGS, double alpha, double epsilon, size_t max_iterations,
                                                 bool_t do_expensive_check, cugraph_centrality_result_t** result, cugraph_error_t** error)
{
  return run_pagerank(handle, graph, precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums, initial_guess_vertices,
                      initial_guess_values, nullptr, nullptr, alpha, epsilon, max_iterations, true, result, error, do_expensive_check == TRUE);
}
extern "C" cugraph_error_code_t cugraph_pagerank_allow_nonconvergence(CUGRAPH_PAGERANK_COMMON_ARGS, double alpha, double epsilon,
                                                                      size_t max_iterations, bool_t do_expensive_check, cugraph_centrality_result_t** result,
                                                                      cugraph_error_t** error)
{
  return run_pagerank(handle, graph, precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums, initial_guess_vertices,
                      initial_guess_values, nullptr, nullptr, alpha, epsilon, max_iterations, false, result, error, do_expensive_check == TRUE);
}
extern "C" cugraph_error_code_t cugraph_personalized_pagerank(CUGRAPH_PAGERANK_COMMON_ARGS,
                                                              const cugraph_type_erased_device_array_view_t* personalization_vertices,
                                                              const cugraph_type_erased_device_array_view_t* personalization_values, double alpha,
                                                              double epsilon, size_t max_iterations, bool_t do_expensive_check, cugraph_centrality_result_t** result,
                                                              cugraph_error_t** error)
{
  return run_pagerank(handle, graph, precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums, initial_guess_vertices,
                      initial_guess_values, personalization_vertices, personalization_values, alpha, epsilon, max_iterations, true, result, error,
                      do_expensive_check == TRUE);
}
extern "C" cugraph_error_code_t cugraph_personalized_pagerank_allow_nonconvergence(
  CUGRAPH_PAGERANK_COMMON_ARGS, const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha, double epsilon, size_t max_iterations,
  bool_t do_expensive_check, cugraph_centrality_result_t** result, cugraph_error_t** error)
{
  return run_pagerank(handle, graph, precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums, initial_guess_vertices,
                      initial_guess_values, personalization_vertices, personalization_values, alpha, epsilon, max_iterations, false, result, error,
                      do_expensive_check == TRUE);
}

// ---- result accessors (cpp/src/c_api/centrality_result.cpp) -------------------------------------
extern "C" cugraph_type_erased_device_array_view_t* cugraph_centrality_result_get_vertices(cugraph_centrality_result_t* result)
{
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(reinterpret_cast<centrality_result_t*>(result)->vertex_ids->new_view());
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_centrality_result_get_values(cugraph_centrality_result_t* result)
{
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(reinterpret_cast<centrality_result_t*>(result)->values->new_view());
}
extern "C" size_t cugraph_centrality_result_get_num_iterations(cugraph_centrality_result_t* result)
{
  return reinterpret_cast<centrality_result_t*>(result)->num_iterations;
}
extern "C" bool_t cugraph_centrality_result_converged(cugraph_centrality_result_t* result)
{
  return reinterpret_cast<centrality_result_t*>(result)->converged ? TRUE : FALSE;
}
extern "C" void cugraph_centrality_result_free(cugraph_centrality_result_t* result)
{
  auto r = reinterpret_cast<centrality_result_t*>(result);
  if (!r) return;
  delete r->vertex_ids;
  delete r->values;
  delete r;
}

// ---- plan API (include/cugraph_amd/extensions.h) -------------------------------------------------
extern "C" cugraph_error_code_t cugraph_amd_pagerank_plan_create(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, const cugraph_type_erased_device_array_view_t* ow_v,
  const cugraph_type_erased_device_array_view_t* ow_s, const cugraph_type_erased_device_array_view_t* ig_v,
  const cugraph_type_erased_device_array_view_t* ig_s, const cugraph_type_erased_device_array_view_t* p_v,
  const cugraph_type_erased_device_array_view_t* p_s, double alpha, cugraph_amd_pagerank_plan_t** plan, cugraph_error_t** error)
{
  if (plan) *plan = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr, CUGRAPH_INVALID_INPUT, "plan is NULL");
    *plan = reinterpret_cast<cugraph_amd_pagerank_plan_t*>(make_plan(handle, graph, ow_v, ow_s, ig_v, ig_s, p_v, p_s, alpha));
  });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_plan_step(cugraph_amd_pagerank_plan_t* plan, double epsilon, size_t max_iterations,
                                                               size_t* iterations_done, bool_t* converged, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr, CUGRAPH_INVALID_INPUT, "plan is NULL");
    size_t done = 0;
    bool conv   = false;
    reinterpret_cast<pagerank_plan_base*>(plan)->step(epsilon, max_iterations, &done, &conv);
    if (iterations_done) *iterations_done = done;
    if (converged) *converged = conv ? TRUE : FALSE;
  });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_plan_tune(cugraph_amd_pagerank_plan_t* plan, size_t placements, double* ms_per_iteration,
                                                               cugraph_error_t** error)
{
  if (ms_per_iteration) *ms_per_iteration = 0.0;
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr, CUGRAPH_INVALID_INPUT, "plan is NULL");
    double const ms = reinterpret_cast<pagerank_plan_base*>(plan)->tune((int)std::min<size_t>(placements, 64));
    if (ms_per_iteration) *ms_per_iteration = ms;
  });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_plan_result(cugraph_amd_pagerank_plan_t* plan, size_t total_iterations, bool_t converged,
                                                                 cugraph_centrality_result_t** result, cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && result != nullptr, CUGRAPH_INVALID_INPUT, "plan / result is NULL");
    *result = reinterpret_cast<cugraph_centrality_result_t*>(
      reinterpret_cast<pagerank_plan_base*>(plan)->result(total_iterations, converged == TRUE));
  });
}
extern "C" void cugraph_amd_pagerank_plan_free(cugraph_amd_pagerank_plan_t* plan) { delete reinterpret_cast<pagerank_plan_base*>(plan); }

// ---- multi-GPU plan API (include/cugraph_amd/extensions.h) ---------------------------------------
namespace {
// a stepping / free entry point of a multi-GPU plan: NULL is reported, and the plan's stream is named to the memory pool (the handle may
// have borrowed another stream since the plan was created; blocks freed by this call must be ordered on the stream the kernels ran on)
pagerank_mg_plan_base* P1D(cugraph_amd_pagerank_mg_plan_t* plan)
{
  CGA_EXPECTS(plan != nullptr, CUGRAPH_INVALID_INPUT, "plan is NULL");
  auto* p = reinterpret_cast<pagerank_mg_plan_base*>(plan);
  pool_set_stream(p->hp->stream);
  return p;
}
pagerank_mg2d_plan_base* P2D(cugraph_amd_pagerank_mg2d_plan_t* plan)
{
  CGA_EXPECTS(plan != nullptr, CUGRAPH_INVALID_INPUT, "plan is NULL");
  auto* p = reinterpret_cast<pagerank_mg2d_plan_base*>(plan);
  pool_set_stream(p->hp->stream);
  return p;
}
}  // namespace

extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg_plan_create(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                                    size_t n_local_rows, size_t global_num_vertices, int comm_rank, int comm_size,
                                                                    const cugraph_type_erased_device_array_view_t* out_weight_sums_local,
                                                                    const cugraph_type_erased_device_array_view_t* initial_local,
                                                                    const cugraph_type_erased_device_array_view_t* send_index, const size_t* send_counts,
                                                                    const size_t* recv_counts, const cugraph_type_erased_device_array_view_t* col_pos,
                                                                    cugraph_type_erased_device_array_view_t* send, cugraph_type_erased_device_array_view_t* recv,
                                                                    double alpha, cugraph_amd_pagerank_mg_plan_t** plan, cugraph_error_t** error)
{
  if (plan) *plan = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr, CUGRAPH_INVALID_INPUT, "plan is NULL");
    handle_t const& h = H(handle);
    graph_t& g        = G(graph);
    if (g.weight_type == FLOAT64) {
      auto p = std::make_unique<pagerank_mg_plan<double>>(h, g, alpha, (int64_t)n_local_rows, (int64_t)global_num_vertices, comm_rank, comm_size);
      p->create(V(out_weight_sums_local), V(initial_local), V(send_index), send_counts, recv_counts, V(col_pos), V(send), V(recv));
      *plan = reinterpret_cast<cugraph_amd_pagerank_mg_plan_t*>(static_cast<pagerank_mg_plan_base*>(p.release()));
    } else {
      auto p = std::make_unique<pagerank_mg_plan<float>>(h, g, alpha, (int64_t)n_local_rows, (int64_t)global_num_vertices, comm_rank, comm_size);
      p->create(V(out_weight_sums_local), V(initial_local), V(send_index), send_counts, recv_counts, V(col_pos), V(send), V(recv));
      *plan = reinterpret_cast<cugraph_amd_pagerank_mg_plan_t*>(static_cast<pagerank_mg_plan_base*>(p.release()));
    }
  });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg_plan_start(cugraph_amd_pagerank_mg_plan_t* plan, cugraph_error_t** error)
{
  return guarded(error, [&] { P1D(plan)->start(); });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg_plan_reduce_scalars(cugraph_amd_pagerank_mg_plan_t* plan, bool_t read_back, double* diff,
                                                                            double* dangling, cugraph_error_t** error)
{
  return guarded(error, [&] {
    P1D(plan)->reduce_scalars(read_back == TRUE, diff, dangling);
  });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg_plan_local_step(cugraph_amd_pagerank_mg_plan_t* plan, cugraph_error_t** error)
{
  return guarded(error, [&] { P1D(plan)->local_step(); });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg_plan_values(cugraph_amd_pagerank_mg_plan_t* plan,
                                                                    cugraph_type_erased_device_array_view_t* out_local, cugraph_error_t** error)
{
  return guarded(error, [&] { P1D(plan)->values(V(out_local)); });
}
extern "C" void cugraph_amd_pagerank_mg_plan_free(cugraph_amd_pagerank_mg_plan_t* plan)
{
  if (!plan) return;
  auto* p = reinterpret_cast<pagerank_mg_plan_base*>(plan);
  pool_set_stream(p->hp->stream);
  delete p;
}

// ---- 2-D layout (see pagerank_mg2d_plan)
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_create(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, size_t rows_per_partition, size_t owned_rows, size_t block_rows, size_t block_cols,
  size_t global_num_vertices, const cugraph_type_erased_device_array_view_t* out_weight_sums_own, const cugraph_type_erased_device_array_view_t* initial_own,
  cugraph_type_erased_device_array_view_t* x_own, const cugraph_type_erased_device_array_view_t* x_cols, cugraph_type_erased_device_array_view_t* y_part,
  const cugraph_type_erased_device_array_view_t* y_own, cugraph_type_erased_device_array_view_t* triple, double alpha, cugraph_amd_pagerank_mg2d_plan_t** plan,
  cugraph_error_t** error)
{
  if (plan) *plan = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    graph_t& g        = G(graph);
    CGA_EXPECTS(plan != nullptr, CUGRAPH_INVALID_INPUT, "plan is NULL");
    CGA_EXPECTS(alpha >= 0.0 && alpha <= 1.0, CUGRAPH_INVALID_INPUT, "Invalid input argument: alpha should be in [0.0, 1.0].");
    if (g.weight_type == FLOAT64) {
      auto p = std::make_unique<pagerank_mg2d_plan<double>>(h, g, alpha, (int64_t)rows_per_partition, (int64_t)owned_rows, (int64_t)block_rows, (int64_t)block_cols, (int64_t)global_num_vertices);
      p->create(V(out_weight_sums_own), V(initial_own), V(x_own), V(x_cols), V(y_part), V(y_own), V(triple));
      *plan = reinterpret_cast<cugraph_amd_pagerank_mg2d_plan_t*>(static_cast<pagerank_mg2d_plan_base*>(p.release()));
    } else {
      auto p = std::make_unique<pagerank_mg2d_plan<float>>(h, g, alpha, (int64_t)rows_per_partition, (int64_t)owned_rows, (int64_t)block_rows, (int64_t)block_cols, (int64_t)global_num_vertices);
      p->create(V(out_weight_sums_own), V(initial_own), V(x_own), V(x_cols), V(y_part), V(y_own), V(triple));
      *plan = reinterpret_cast<cugraph_amd_pagerank_mg2d_plan_t*>(static_cast<pagerank_mg2d_plan_base*>(p.release()));
    }
  });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_start(cugraph_amd_pagerank_mg2d_plan_t* plan, cugraph_error_t** error)
{
  return guarded(error, [&] { P2D(plan)->start(); });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_set_scalars(cugraph_amd_pagerank_mg2d_plan_t* plan, const void* gathered_triples, int comm_size,
                                                                          bool_t read_back, double* diff, double* dangling, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(gathered_triples != nullptr && comm_size >= 1, CUGRAPH_INVALID_INPUT, "set_scalars: gathered triples / comm_size");
    P2D(plan)->set_scalars(gathered_triples, comm_size, read_back == TRUE, diff, dangling);
  });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_spmv(cugraph_amd_pagerank_mg2d_plan_t* plan, cugraph_error_t** error)
{
  return guarded(error, [&] { P2D(plan)->spmv(); });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_epilogue(cugraph_amd_pagerank_mg2d_plan_t* plan, cugraph_error_t** error)
{
  return guarded(error, [&] { P2D(plan)->epilogue(); });
}
extern "C" cugraph_error_code_t cugraph_amd_pagerank_mg2d_plan_values(cugraph_amd_pagerank_mg2d_plan_t* plan, cugraph_type_erased_device_array_view_t* out_own,
                                                                     cugraph_error_t** error)
{
  return guarded(error, [&] { P2D(plan)->values(V(out_own)); });
}
extern "C" void cugraph_amd_pagerank_mg2d_plan_free(cugraph_amd_pagerank_mg2d_plan_t* plan)
{
  if (!plan) return;
  auto* p = reinterpret_cast<pagerank_mg2d_plan_base*>(plan);
  pool_set_stream(p->hp->stream);
  delete p;
}
