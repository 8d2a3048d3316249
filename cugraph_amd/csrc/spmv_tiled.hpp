// Column-tiled two-phase pull-SpMV for gfx950: every gather of the source-vertex vector is served from LDS.
//
// Replaces the reference's per_v_transform_reduce_incoming_e kernels on the PageRank path
// (cpp/include/cugraph/prims/detail/per_v_transform_reduce_e.cuh:252/389/500/688) -- see DESIGN.md, "tiled SpMV".
//
// Why: on MI355X a random 4-byte gather that misses LDS costs one L2 (or Infinity-Cache / HBM) request per edge and
// the texture-addresser path tops out near 260 G requests/s even when the whole vector is L2-resident (measured,
// tools/ubench/gather_bench.hip) -- 3x below what an HBM-bound edge stream needs.  So the edge list is re-blocked once
// per graph:
//   * sources are cut into tiles of T consecutive ids (T * sizeof(weight) = 128 KiB of LDS);
//   * edges are ordered by (source tile J, destination, source); a maximal group of edges with equal (J, destination)
//     is a RUN; per edge we keep a 16-bit tile-local source id and one bit "starts a run";
//   * phase 1 (k_tiled_phase1): a workgroup stages x[J*T, (J+1)*T) * alpha in LDS, streams its share of tile J's
//     edges (2 bytes + 1 bit each), forms the run sums with an in-lane pass + wave64 DPP segmented scan, and stores
//     one partial per run into a slot of the partial buffer (slot = run index + a per-block delta: inside one
//     (destination tile, source tile) block the slots follow the run order, so 4 bytes per block + 1 bit per run
//     replace a 4-byte slot per run);
//   * destinations are cut into tiles I (<= 4096 rows, equal cost); the slots of all runs whose destination lies in I
//     form the contiguous REGION I (ordered by J, then destination);
//   * phase 2 (k_tiled_phase2): one workgroup per destination tile streams its region (4-byte partial + 16-bit
//     tile-local destination), accumulates in LDS (ds_add), and runs the fused PageRank epilogue for its rows
//     (new pr, next x = pr / out_w, L1 change, dangling mass); workgroup 0 of the next phase 1 folds the per-tile
//     scalar partials in a fixed order into the next iteration's constants (no extra launch, no device-scope fences).
//     fp32 partials are accumulated in 64-bit fixed point: LDS integer atomics run at full rate on gfx950 (ds_add_f32
//     is ~5x slower) and make the result independent of the accumulation order.
// HBM traffic per iteration = 2.125 E + 10 P + 16 V bytes + per-wavefront records + block deltas + tile reloads
// (P = number of runs; RMAT-22: P = 0.15 E, RMAT-26: 0.29 E; measured 7.1 GB at RMAT-26), all of it streaming; nothing is
// gathered from global memory.  Single-GPU plans number only the sources that have out-edges as columns (xcol).
#pragma once

#include "common.hpp"

namespace cga {

template <typename WT>
struct pr_scalars {  // device-resident PageRank loop state
  WT base;         // (alpha * dangling + (1 - alpha)) / V, or 0 when personalized
  WT pers_factor;  // alpha * dangling + (1 - alpha)
  WT dangling;
  WT diff;
  int32_t fx_k;   // (round 1-4 layout: one global fixed-point scale; kept for the record, phase 2 now derives a scale per destination tile)
  double fx_inv;  // 2^-fx_k
  double fx_unit; // alpha * max|x|: a partial or a row sum of destination tile I is bounded by fx_unit * tile_wmax[I] (phase 2's fixed-point scale)
  WT base_prev;   // tiled_const_rows: the value the rows without in-edges hold BEFORE the iteration that uses `base`
};

// Rows without in-edges (ids >= tiled_csc_t::n_act) have pr' = base in every iteration: a plan that is neither personalized nor
// started from a user vector leaves them out of the per-iteration epilogue.  Their share of the iteration's scalars is
// analytic (n_rows * |base - base_prev|, n_dangling * base, base * max_inv_outw), the x of their live columns
// [c0, c0 + n_cols) is base / outw_c[.] (written by a few extra workgroups of phase 2), and pr itself is materialized when the
// result is read.  Saves 16 B x 60 % of the rows of an RMAT-26 graph per iteration.
template <typename WT>
struct tiled_const_rows {
  int nI_act{0};             // 0 = off (every destination tile runs its epilogue)
  int64_t n_rows{0};         // rows >= n_act
  int64_t n_dangling{0};     // of those, rows without out-edges
  double max_inv_outw{0};    // max over those rows of 1 / (outw == 0 ? 1 : outw)
  WT const* outw_c{nullptr}; // out-weight sums of the live columns >= c0, in column order
  int64_t c0{0}, n_cols{0};
  int32_t const* col_idx{nullptr};  // non-null: the j-th live column is written to x_next[col_idx[j]] instead of x_next[c0 + j] (multi-GPU
                                    // plans: only the rows some rank references get a value, wherever the exchange wants it)
};

constexpr int TP_BLOCK = 1024;               // phase-1 workgroup: 16 wavefronts sharing one LDS tile
constexpr int TP_WAVES = TP_BLOCK / 64;
constexpr int TP_EPL = 16;                   // consecutive edges per lane per work item (two 16-byte loads)
constexpr int TP_WG_PER_CU = 1;              // the source tile takes ~126 KiB of the 160 KiB LDS
#ifndef CGA_TP_STAGE
#define CGA_TP_STAGE 512
#endif
static_assert(CGA_TP_STAGE >= 512, "the long-run path stages the run totals of 32 lanes x 16 edges at a time");
constexpr int TP_STAGE = CGA_TP_STAGE;                // per-wavefront LDS staging entries (run totals awaiting the coalesced write-out)
constexpr int TP_NDREG = 2;                  // registers of prefetched slot-block deltas per lane: 64 * TP_NDREG blocks per wavefront and item
constexpr int TP_SUB   = 64 * TP_EPL;        // edges per wavefront per work item (TP_EPL consecutive edges per lane)
constexpr int TP_WLEN  = TP_SUB;
constexpr int TP_ITEM  = TP_WLEN * TP_WAVES;  // edges per work item
constexpr double TP_TAIL_FRAC = 0.15;        // share of the items (coldest tiles) handed out in chunks of TP_TAIL_CHUNK items
constexpr int TP_TAIL_CHUNK = 8;
constexpr int TP_CHUNK = 16;                 // max work items per dynamically scheduled chunk (256 Ki edges); chunks are handed out largest first
constexpr int TP_CHUNK_BIG = 32;             // chunk length for the first part of a tile that spans many chunks (see build_tiled_csc)
#ifndef CGA_TP2_BLOCK
#define CGA_TP2_BLOCK 512
#endif
#ifndef CGA_TP2_ROWS
#define CGA_TP2_ROWS 4096
#endif
constexpr int TP2_BLOCK = CGA_TP2_BLOCK;     // phase-2 workgroup
constexpr int TP2_ROWS  = CGA_TP2_ROWS;      // max destination rows per phase-2 tile (64-bit LDS accumulators: 32 KiB at 4096)
constexpr int TP2_CONST_COLS = TP2_BLOCK * 8;  // columns per tiled_const_rows block of phase 2
constexpr int TP2_STAGGER_MIN_GRID = 4096;    // phase 2 staggers its first generation of workgroups on grids of at least four generations

struct tiled_wave_t {  // build-time description of one wavefront's share of a work item
  uint32_t es, ee;     // padded edge positions [es, ee); es = item * TP_ITEM + wave * TP_WLEN
  uint32_t rank;       // number of run starts before es (global run ordinal of the first run that starts in the range)
  uint32_t head_slot;  // slot for the partial of the run already open at es (dummy slot when there is none)
};

// What phase 1 reads per wavefront and work item: ONE record of TP_REC_DWORDS dwords, dword L loaded by lane L.
//   dwords 0..31   block-start bits: bit t = run (rank + t), the (t + 1)-th run that starts in the range, is the first run of
//                  a slot block (slot - run index is constant inside a block; zero beyond the runs of the range)
//   dword 32..     TP_REC_NVAL = ee - es, TP_REC_RANK, TP_REC_HEAD (head slot), TP_REC_TAIL (slot of the run still open at
//                  ee; the head slot when no run starts in the range), TP_REC_BLK (index into delta1 of the block of run rank - 1)
// Slot of run r = r + delta1[1 + block(r)]: 4 bytes per BLOCK (runs of one source tile falling into one destination tile)
// instead of 4 bytes per run.
constexpr int TP_REC_DWORDS = 40;
constexpr int TP_REC_NVAL = 32, TP_REC_RANK = 33, TP_REC_HEAD = 34, TP_REC_TAIL = 35, TP_REC_BLK = 36;

struct tiled_csc_t {
  bool built{false};
  int T{0};    // sources per tile
  int nJ{0};   // source tiles
  int nI{0};   // destination tiles
  int n_items{0};
  int n_wg{0};  // phase-1 workgroups
  int n_chunks{0};
  int n_static_chunks{0};     // chunks [0, n_static_chunks) are pre-assigned to workgroups (wg_static), the rest are drawn dynamically
  int64_t nv{0}, ne{0}, ne_pad{0}, n_runs{0}, n_slots{0}, n_blocks{0};
  int64_t n_act{0};  // rows >= n_act have no in-edge; a destination-tile boundary is forced there
  int nI_act{0};     // destination tiles [0, nI_act) cover rows [0, n_act)
  int64_t c0{0};     // columns >= c0 belong to rows >= n_act (columns are monotone in the row id)
  double wmax{0};  // max over destinations of sum |w| of the in-edges (in-degree when unweighted): bounds a row sum by alpha * max|x| * wmax
  dvec<uint16_t> src16;       // [ne_pad + pad] tile-local source id
  dvec<uint32_t> bits;        // [ne_pad / 32 + pad] bit p = edge position p starts a run
  dev_buf weights;            // [ne_pad + pad] or empty
  dvec<uint32_t> delta1;      // [n_blocks + 1 + pad] slot - run index of block b at [b + 1]
  dvec<int32_t> item_tile;    // [n_items] source tile of work item
  dvec<uint32_t> wrec;        // [n_items * TP_WAVES][TP_REC_DWORDS] per-wavefront records (see above)
  dvec<int32_t> chunk_begin;  // [n_chunks][4] (unused, first item, end item, source tile) of each chunk (<= TP_CHUNK items of one source tile), largest first
  dvec<int32_t> wg_static;    // [n_wg][2] (first, end) static chunk of each phase-1 workgroup
  dvec<uint32_t> tile_row0;   // [nI + 1] destination tile boundaries
  dvec<double> tile_wmax;     // [nI] max over the tile's rows of sum |w| of the row's in-edges (in-degree when unweighted): bounds the tile's partials and row sums
  dvec<uint32_t> region_off;  // [nI + 2] slot range of region I (multiples of 8); region nI = dummy
  dvec<uint16_t> dstl16;      // [n_slots + pad] tile-local destination of slot (16 bits per slot: tiles of more than 4096 rows)
  dvec<uint32_t> dstl12;      // [n_slots / 8 * 3 + pad] the same packed to 12 bits per slot, 8 slots = 3 dwords (tiles of <= 4096 rows:
                              // 0.15 GB less to read per iteration at RMAT-26); exactly one of the two is kept
  // Compact column ids (single-GPU plans): sources WITHOUT out-edges are never gathered, so they get no column; column of a
  // live source = number of live sources with a smaller id (monotone, so rows of one destination tile map to consecutive
  // columns and the epilogue's x writes stay coalesced).  At RMAT-26 two thirds of the vertices have no out-edge: the tiles
  // become 3x denser (longer runs, a third of the tile loads).  Empty = identity (multi-GPU: columns are compact already).
  dvec<int32_t> xcol;         // [nv] row -> column, -1 = no out-edges
  int64_t ncols{0};           // number of columns (= nv when xcol is empty)
};

// Re-blocks the CSC orientation of g (offsets / indices / weights) into tiles of T sources.
// `vsize` = sizeof(value type); weights (if any) have the same type.
// nv = number of column (source) ids = CSC rows; n_dst <= nv = leading rows that may have in-edges (single GPU: nv).
// compact_columns: renumber the sources that have out-edges densely (see tiled_csc_t::xcol); needs nv == n_dst.
void build_tiled_csc(handle_t const& h, int64_t nv, int64_t n_dst, int64_t ne, orientation_t const& csc, bool has_weights, size_t vsize, int T,
                     tiled_csc_t& t, bool compact_columns = false, uint32_t const* live_hint = nullptr);  // live_hint[v] != 0 <=> v has an
                                                                                                         // out-edge ([nv + 1], last entry 0): saves the marking pass

template <typename WT>
struct tiled_x_map {  // where x[c] lives; single GPU: identity.  Multi-GPU all-gather buffer: (c & pmask) * chunk + (c >> plog)
  uint32_t pmask{0}, plog{0}, chunk{0}, ncols{0};
  bool mg{false};
};

template <typename WT>
struct tiled_epilogue {
  int64_t nv{0};         // rows (local rows in the multi-GPU case)
  WT* pr{nullptr};       // in: previous iterate, out: new iterate
  WT* x_next{nullptr};   // pr / out_w (next gather vector; the multi-GPU send chunk)
  int32_t const* xcol{nullptr};  // tiled_csc_t::xcol: where row r's x goes in x_next (nullptr: x_next[r])
  bool write_pr{true};           // false: the new iterate is kept as x_next only (an iteration that is followed by another one inside the
                                 // same step() call and is not asked for its L1 change: nobody reads pr before it is overwritten)
  bool need_diff{true};          // false: the L1 change is not wanted (fixed iteration count): the previous iterate is not read
                                 // (-0.11 GB per iteration at RMAT-26, phase 2 0.436 -> 0.418 ms).  Deriving the row -> column map from one
                                 // bit per row instead of reading xcol was tried with it and lost (+0.012 ms: the lookups sit at the end)
  WT* raw_y{nullptr};            // non-null: plain SpMV -- phase 2 stores y[row] = sum of the row's partials and nothing else (no base term, no
                                 // division, no scalars): the 2-D multi-GPU layout reduces the partial rows over a column of ranks first
  WT const* outw{nullptr};
  WT const* pers{nullptr};  // dense normalised personalization or nullptr
  pr_scalars<WT>* scal{nullptr};
  double* partials{nullptr};  // [max(nI, 1024)][3] per-destination-tile (L1 change, dangling mass, max |x_next|)
  double* totals{nullptr};    // multi-GPU: the folded (diff, dangling, xmax) of this rank are written here instead of into scal
  WT alpha{0};
  int64_t nv_global{0};
  double wmax{0};             // tiled_csc_t::wmax
  tiled_const_rows<WT> cr;    // nI_act == 0: off
};

// phase 1: part[slot of run] = sum over the run's edges of alpha * x[src] (* w).  counters[0] = chunk cursor (0 on entry;
// phase 2 rewinds it).  `pending` != nullptr: the scalars of the
// previous phase 2 have not been folded yet -- workgroup 0 does it first (saves a launch per iteration).
template <typename WT>
void tiled_phase1(handle_t const& h, tiled_csc_t const& t, WT const* x, WT alpha, WT* part, uint32_t* counters, tiled_x_map<WT> const& map,
                  tiled_epilogue<WT> const* pending);

// phase 2 + fused PageRank epilogue; leaves per-tile scalar partials in e.partials (fold them with the next phase 1 or tiled_finish)
template <typename WT>
void tiled_phase2(handle_t const& h, tiled_csc_t const& t, WT const* part, tiled_epilogue<WT> const& e, uint32_t* counters);

// folds e.partials[0 .. n_partials) in a fixed order into e.scal (or e.totals).  init_prev >= 0: this is the fold of the
// iteration-0 state (tiled_prologue visited every row); scal->base_prev becomes init_prev, the rows' initial value
template <typename WT>
void tiled_finish(handle_t const& h, tiled_epilogue<WT> const& e, int n_partials, double init_prev = -1.0, hipStream_t stream = nullptr);

// number of per-tile scalar triples phase 2 leaves for the fold
template <typename WT>
inline int tiled_fold_count(tiled_csc_t const& t, tiled_epilogue<WT> const& e) { return e.cr.nI_act > 0 ? e.cr.nI_act : t.nI; }

// iteration-0 state: x = pr / out_w plus per-block (0, dangling, max |x|) partials; returns the number of partial triples
template <typename WT>
int tiled_prologue(handle_t const& h, tiled_csc_t const& t, WT const* pr, WT const* outw, WT* x, int64_t nv, double* partials,
                   int32_t const* xcol_override = nullptr);  // honours t.xcol (or the given row -> column map, -1 = no column)

// multi-GPU: e.scal <- fold of the per-rank (diff, dangling, xmax) triples at recv + first_off + r * stride_bytes
template <typename WT>
void tiled_scalars_from_ranks(handle_t const& h, tiled_epilogue<WT> const& e, void const* recv, size_t first_off, size_t stride_bytes, int nranks);

int tiled_default_T(handle_t const& h, size_t weight_size, int64_t nv);

// 2^k with alpha * xmax * wmax * 2^k < 2^61 (every row sum of phase 2 fits a signed 64-bit accumulator with room to spare)
__device__ __forceinline__ void tiled_fixed_point_scale(double bound, int32_t* k, double* inv)
{
  int kk = 0;
  if (bound > 0.0 && bound < 1.0e300) {
    int e;
    (void)frexp(bound, &e);  // bound = m * 2^e, 0.5 <= m < 1  =>  bound < 2^e
    kk = 61 - e;
  }
  kk   = max(-900, min(900, kk));
  *k   = kk;
  *inv = ldexp(1.0, -kk);
}

// the iteration's scalars -> constants of the next iteration (base term, personalization factor, fixed-point scale)
template <typename WT>
__device__ __forceinline__ void tiled_write_scalars(pr_scalars<WT>* scal, double diff, double dang, double xmax, WT alpha, int64_t nv_global,
                                                    int personalized, double wmax)
{
  scal->base_prev   = scal->base;  // what the iteration that just ended wrote into the rows without in-edges
  WT dangling       = (WT)dang;
  WT factor         = dangling * alpha + (WT)(1.0 - (double)alpha);
  scal->dangling    = dangling;
  scal->diff        = (WT)diff;
  scal->pers_factor = factor;
  scal->base        = personalized ? WT(0) : factor / (WT)nv_global;
  tiled_fixed_point_scale((double)alpha * xmax * wmax, &scal->fx_k, &scal->fx_inv);
  scal->fx_unit = (double)alpha * xmax;
}

// Phase 2's fixed point, round 5.  A partial v of destination tile I satisfies |v| <= B = alpha * max|x| * (max over the tile's rows of sum |w|),
// and so does every row sum.  With 2^k chosen so that B * 2^k < 2^50, n = round(v * 2^k) is an integer below 2^50 in magnitude and
//     t = fma((double)v, 2^k, 1.5 * 2^52)
// is exactly 1.5 * 2^52 + n (one rounding, to nearest even, at unit 1): the integer is the difference of the BIT PATTERNS of t and of
// 1.5 * 2^52 -- whose low dword is zero, so the conversion is v_cvt_f64_f32 + v_fma_f64 + one 32-bit subtract instead of the 32 VALU
// instructions of the shift-based conversion of rounds 1-4 (a CU then reduced 22 GB/s of partials, i.e. phase 2 needed the whole chip to keep
// up with the HBM; profiles/r5a_overlap_first_sweep.txt).  Integer accumulation stays order-independent (bit-reproducible); the scale is per
// destination tile, so the cold tiles (small in-degrees) keep more fraction bits than the single global scale gave them.
constexpr double kFxMagic          = 6755399441055744.0;       // 1.5 * 2^52
constexpr unsigned long long kFxMagicBits = 0x4338000000000000ull;
__device__ __forceinline__ void tiled_tile_scale(double bound, double* scale, double* inv)
{
  // 2^kk, kk = 50 - e with bound = m * 2^e, 0.5 <= m < 1 (frexp), straight from the exponent field: every phase-2 workgroup starts with this
  // (the frexp / ldexp library calls of round 5 sat on its critical path; same values bit for bit, denormal bounds clamp to 2^900 as before)
  int kk = 0;
  if (bound > 0.0 && bound < 1.0e300) kk = 50 - ((int)((unsigned long long)__double_as_longlong(bound) >> 52) - 1022);
  kk     = max(-900, min(900, kk));
  *scale = __longlong_as_double((long long)(1023 + kk) << 52);
  *inv   = __longlong_as_double((long long)(1023 - kk) << 52);
}
__device__ __forceinline__ unsigned long long tiled_to_fixed(float v, double scale)
{
  return (unsigned long long)__double_as_longlong(fma((double)v, scale, kFxMagic)) - kFxMagicBits;
}

}  // namespace cga
