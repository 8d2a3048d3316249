// The bottom-up BFS level shared by the single-GPU driver (traversal.hip) and the partitioned engine (traversal_mg.hip).
#pragma once
#include "traversal_common.hpp"

namespace cga {
namespace {

// Bottom-up level.  in_offsets / in_indices = the orientation whose rows are DESTINATIONS (CSC; the CSR itself when the
// graph is symmetric).  front = frontier bitmap of the current level; next (fully rewritten) = vertices found.
#ifndef CGA_BU_GROUPS
#define CGA_BU_GROUPS 4
#endif
constexpr int BU_GROUPS     = CGA_BU_GROUPS;   // 64-vertex groups a wavefront keeps in flight
constexpr int BU_CHUNK      = 4;   // independent probes per lane and step (one 16-byte neighbour load)
constexpr int BU_LANE_MAX   = 64;  // neighbours a lane scans on its own; the rest of a still unsettled row is scanned by the whole wave
// the next BU_CHUNK neighbour ids of a row as ONE 16-byte load per lane (4-byte aligned: gfx950 global loads take it): a lane's
// neighbours are consecutive, but to the memory pipeline every dword load of a row start is a separate random access, and
// those, not bytes, bound a bottom-up level (~65 G random accesses/s beyond the Infinity Cache, tools/ubench/gather_bench.hip).
// Entries past `left` (the row's remaining length; the index array is padded) come back as -1.
typedef int32_t bu_i32x4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void bu_load_chunk(int32_t const* indices, eoff_t pos, int32_t left, int32_t (&u)[BU_CHUNK])
{
  static_assert(BU_CHUNK == 4, "one dwordx4 load per chunk");
  bu_i32x4 v = {-1, -1, -1, -1};
  if (left > 0) v = *reinterpret_cast<bu_i32x4 const*>(indices + pos);
  u[0] = left > 0 ? v.x : -1; u[1] = left > 1 ? v.y : -1; u[2] = left > 2 ? v.z : -1; u[3] = left > 3 ? v.w : -1;
}

template <bool PROF>
__global__ void __launch_bounds__(TV_BLOCK) k_bfs_bottom_up(int32_t const* in_offsets, int32_t const* in_indices, int32_t const* out_offsets,
                                                            int64_t nv, uint32_t* vis, uint32_t const* front, uint32_t* next, int32_t* dist,
                                                            int32_t* pred, int32_t next_depth, counters_t* cnt, unsigned long long* prof,
                                                            int32_t const* parent_label = nullptr)  // pred[v] = parent_label ? parent_label[parent] : parent
{
  // PROF (CUGRAPH_AMD_BFS_PROFILE): wall ticks per wavefront in the three parts of an iteration, tail steps, long rows and their steps
  unsigned long long pt[3] = {0, 0, 0}, pn[3] = {0, 0, 0};
  // A level is a chain of dependent memory round trips per 64-vertex group (visited word -> offsets -> neighbour ids -> frontier
  // bits): with one group per wavefront at a time the level was latency-bound (0.9 ms at RMAT-24 whatever the frontier).  So a
  // wavefront walks BU_GROUPS groups at once through the common part -- the first BU_CHUNK neighbours of every unvisited
  // vertex, which settle most of them because in-neighbours are sorted hubs first -- and only then finishes the rows that
  // are still open, group by group.
  int const lane       = threadIdx.x & 63;
  int64_t const gwave  = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t const nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int64_t const ngroup = (nv + 63) >> 6;
  unsigned long long inspected = 0, acc_out = 0, acc_in = 0;
  uint32_t found_total = 0;
  int32_t od0[BU_GROUPS], od1[BU_GROUPS];  // out-degree bounds of the vertices discovered in the previous iteration (summed one iteration late)
#pragma unroll
  for (int g = 0; g < BU_GROUPS; ++g) { od0[g] = 0; od1[g] = 0; }
  for (int64_t grp0 = gwave * BU_GROUPS; grp0 < ngroup; grp0 += nwaves * BU_GROUPS) {
    unsigned long long const tk0 = PROF ? wall_clock64() : 0;
    bool unvisited[BU_GROUPS], found[BU_GROUPS], open_row[BU_GROUPS];
    eoff_t b[BU_GROUPS];  // first in-edge of the lane's vertex (unsigned 32-bit position)
    int32_t e[BU_GROUPS], parent[BU_GROUPS], scanned[BU_GROUPS];  // e = the row's LENGTH (a row has fewer than 2^31 edges)
    int32_t u[BU_GROUPS][BU_CHUNK];
    uint32_t word[BU_GROUPS];
#pragma unroll
    for (int g = 0; g < BU_GROUPS; ++g) {  // visited words and row bounds are requested together (the bounds of the visited vertices
      int64_t const grp = grp0 + g, v = grp * 64 + lane;  // too: 64 consecutive offsets are one cheap coalesced load, a dependent step is not)
      word[g] = grp < ngroup ? vis[(grp * 2) + (lane >> 5)] : 0xFFFFFFFFu;
      b[g] = 0; e[g] = 0; parent[g] = -1; found[g] = false; scanned[g] = 0;
      if (v < nv) { b[g] = eoff(in_offsets, v); e[g] = (int32_t)(eoff(in_offsets, v + 1) - b[g]); }
    }
#pragma unroll
    for (int g = 0; g < BU_GROUPS; ++g) {
      int64_t const v = (grp0 + g) * 64 + lane;
      unvisited[g] = v < nv && !((word[g] >> (lane & 31)) & 1u);
      acc_out += (unsigned long long)(uint32_t)(od1[g] - od0[g]);  // requested an iteration ago: complete by the time `word` is (loads retire in order)
    }
#pragma unroll
    for (int g = 0; g < BU_GROUPS; ++g) {
      open_row[g] = unvisited[g] && e[g] > 0;
#if defined(CGA_BU_ABL) && CGA_BU_ABL == 2  // timing experiment: no neighbour load either
      u[g][0] = open_row[g] ? 0 : -1; u[g][1] = u[g][2] = u[g][3] = -1;
#else
      bu_load_chunk(in_indices, b[g], open_row[g] ? e[g] : 0, u[g]);
#endif
    }
#pragma unroll
    for (int g = 0; g < BU_GROUPS; ++g) {
      int first = BU_CHUNK;
#pragma unroll
      for (int k = BU_CHUNK - 1; k >= 0; --k)
#if defined(CGA_BU_ABL) && CGA_BU_ABL == 1  // timing experiment: no frontier probe (every neighbour "hits")
        if (u[g][k] >= 0) first = k;
#else
        if (u[g][k] >= 0 && ((front[u[g][k] >> 5] >> (u[g][k] & 31)) & 1u)) first = k;  // ascending ids: the first hit is the minimum
#endif
      if (open_row[g]) {
        int32_t const deg = e[g];
        if (first < BU_CHUNK) {
#pragma unroll
          for (int k = 0; k < BU_CHUNK; ++k) if (k == first) parent[g] = u[g][k];
          found[g] = true; open_row[g] = false; scanned[g] = first + 1;
        } else {
          scanned[g] = min(deg, BU_CHUNK);
          if (scanned[g] >= deg) open_row[g] = false;
        }
      }
    }
    // the rows that are still open: further neighbours per lane up to BU_LANE_MAX.  A step is a dependent round trip (neighbour
    // ids -> frontier bits) whatever the number of lanes still scanning, and in the mid-degree range nearly every group has a
    // few lanes that scan to the end (vertices that are reached a level later): the BU_GROUPS groups therefore take their steps
    // TOGETHER, two 16-byte chunks per lane and step -- eight times the requests in flight of a group-by-group walk
    unsigned long long const tk1 = PROF ? wall_clock64() : 0;
    for (;;) {
      if (PROF) ++pn[0];
      bool act[BU_GROUPS];
      bool any = false;
#pragma unroll
      for (int g = 0; g < BU_GROUPS; ++g) { act[g] = open_row[g] && scanned[g] < BU_LANE_MAX; any |= act[g]; }
      if (!__ballot(any)) break;
      int32_t w[BU_GROUPS][2 * BU_CHUNK];
#pragma unroll
      for (int g = 0; g < BU_GROUPS; ++g) {
        int32_t const left = act[g] ? e[g] - scanned[g] : 0;
        int32_t lo[BU_CHUNK], hi[BU_CHUNK];
        bu_load_chunk(in_indices, b[g] + (eoff_t)scanned[g], left, lo);
        bu_load_chunk(in_indices, b[g] + (eoff_t)scanned[g] + BU_CHUNK, left - BU_CHUNK, hi);
#pragma unroll
        for (int k = 0; k < BU_CHUNK; ++k) { w[g][k] = lo[k]; w[g][BU_CHUNK + k] = hi[k]; }
      }
#pragma unroll
      for (int g = 0; g < BU_GROUPS; ++g) {
        int first = 2 * BU_CHUNK;
#pragma unroll
        for (int k = 2 * BU_CHUNK - 1; k >= 0; --k)
          if (w[g][k] >= 0 && ((front[w[g][k] >> 5] >> (w[g][k] & 31)) & 1u)) first = k;
        if (act[g]) {
          int32_t const deg = e[g];
          if (first < 2 * BU_CHUNK) {
#pragma unroll
            for (int k = 0; k < 2 * BU_CHUNK; ++k) if (k == first) parent[g] = w[g][k];
            found[g] = true; open_row[g] = false; scanned[g] += first + 1;
          } else {
            scanned[g] = min(deg, scanned[g] + 2 * BU_CHUNK);
            if (scanned[g] >= deg) open_row[g] = false;
          }
        }
      }
    }
    unsigned long long const tk2 = PROF ? wall_clock64() : 0;
    // rows longer than BU_LANE_MAX with nothing found so far (rare): the wavefront strides the row, 64 neighbours per step, and
    // stops at the first hit
#pragma unroll
    for (int g = 0; g < BU_GROUPS; ++g) {
      inspected += (unsigned long long)scanned[g];
      uint64_t cm = __ballot(open_row[g]);
      while (cm) {
        int src = __ffsll((unsigned long long)cm) - 1;
        cm &= cm - 1;
        if (PROF) ++pn[1];
        eoff_t const row = (eoff_t)__shfl((int)b[g], src);                       // the row's first edge
        int32_t const first = __shfl(scanned[g], src), len = __shfl(e[g], src);  // scan positions [first, len) of the row
        bool hit = false;
        int32_t p = first, par = -1;
        for (; p < len && !hit; p += 64) {
          if (PROF) ++pn[2];
          int32_t q = p + lane, x = -1;
          bool h    = false;
          if (q < len) { x = in_indices[row + (eoff_t)q]; h = ((front[x >> 5] >> (x & 31)) & 1u) != 0; }
          uint64_t hm = __ballot(h);
          hit         = hm != 0;
          if (hit) par = __shfl(x, __ffsll((unsigned long long)hm) - 1);  // lowest lane = smallest position = smallest id
        }
        if (lane == src) { found[g] = hit; parent[g] = par; inspected += (unsigned long long)(min(p, len) - first); }
      }
    }
    // results: nothing below waits for memory (the visited words are still in registers; the out-degrees of the discovered
    // vertices -- the top-down cost of the next level -- are requested for all groups and summed after the loop)
#pragma unroll
    for (int g = 0; g < BU_GROUPS; ++g) {
      int64_t const v = (grp0 + g) * 64 + lane;
      od0[g] = 0; od1[g] = 0;
      if (found[g]) { od0[g] = out_offsets[v]; od1[g] = out_offsets[v + 1]; }  // (differences of the words are taken modulo 2^32: a degree fits)
    }
#pragma unroll
    for (int g = 0; g < BU_GROUPS; ++g) {
      int64_t const grp = grp0 + g, v = grp * 64 + lane;
      uint64_t const fm = __ballot(found[g]);
      if (grp < ngroup && (lane & 31) == 0) {  // lanes 0 and 32 hold the two visited words of the group; this wavefront is their only writer
        uint32_t const bits = lane ? (uint32_t)(fm >> 32) : (uint32_t)fm;
        next[grp * 2 + (lane >> 5)] = bits;
        if (bits) vis[grp * 2 + (lane >> 5)] = word[g] | bits;
      }
      if (found[g]) {
        dist[v] = next_depth;
        if (pred) pred[v] = parent_label ? parent_label[parent[g]] : parent[g];
        acc_in += (unsigned long long)e[g];
      }
      found_total += (uint32_t)__popcll(fm);
    }
    if (PROF) { unsigned long long const tk3 = wall_clock64(); pt[0] += tk1 - tk0; pt[1] += tk2 - tk1; pt[2] += tk3 - tk2; }
  }
  if (PROF && lane == 0) {
    for (int k = 0; k < 3; ++k) { atomicAdd(prof + k, pt[k]); atomicAdd(prof + 3 + k, pn[k]); }
    atomicMax(prof + 6, pt[0] + pt[1] + pt[2]);
    atomicMax(prof + 7, pt[2]);
  }
#pragma unroll
  for (int g = 0; g < BU_GROUPS; ++g) acc_out += (unsigned long long)(uint32_t)(od1[g] - od0[g]);
  for (int o = 32; o > 0; o >>= 1) { inspected += __shfl_xor(inspected, o); acc_out += __shfl_xor(acc_out, o); acc_in += __shfl_xor(acc_in, o); }
  if (lane == 0) {
    counter_sums_t* r = cnt_replica(cnt);
    if (found_total) atomicAdd(&r->n_found, (unsigned long long)found_total);
    if (inspected) atomicAdd(&r->edges, inspected);
    if (acc_out | acc_in) { atomicAdd(&r->out_edges, acc_out); atomicAdd(&r->in_edges, acc_in); }
  }
}

// set bits of `bits` -> a list of their positions (n_big doubles as the queue cursor of this kernel).  Every workgroup owns one contiguous
// range of words, counts its bits, reserves its piece of the queue with ONE atomic and fills it: with one atomic per wavefront-chunk the
// 8 Ki same-address atomics of a 16 M-vertex bitmap took 100 us whatever the bitmap held (round 4).
__global__ void __launch_bounds__(TV_BLOCK) k_bfs_bitmap_to_queue(uint32_t const* bits, int64_t nwords, int32_t* q, counters_t* cnt)
{
  __shared__ uint32_t s_wave[TV_BLOCK / 64];
  __shared__ uint32_t s_base;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t const per = ((nwords + gridDim.x - 1) / gridDim.x + TV_BLOCK - 1) / TV_BLOCK * TV_BLOCK;  // words per workgroup, whole rounds
  int64_t const w0 = (int64_t)blockIdx.x * per, w1 = w0 + per < nwords ? w0 + per : nwords;
  uint32_t mine = 0;
  for (int64_t i = w0 + threadIdx.x; i < w1; i += TV_BLOCK) mine += __popc(bits[i]);
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if (lane == 0) s_wave[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int k = 0; k < TV_BLOCK / 64; ++k) t += s_wave[k];
    s_base = t ? atomicAdd(&cnt->n_big, t) : 0u;
  }
  __syncthreads();
  uint32_t base = s_base;  // running cursor of the workgroup (uniform)
  for (int64_t r0 = w0; r0 < w1; r0 += TV_BLOCK) {  // one round = TV_BLOCK words; the waves of a round take consecutive pieces
    int64_t const i = r0 + threadIdx.x;
    uint32_t w      = i < w1 ? bits[i] : 0u;
    uint32_t c      = __popc(w), total;
    uint32_t ex     = wave_excl_scan(c, lane, &total);
    __syncthreads();
    if (lane == 0) s_wave[wave] = total;
    __syncthreads();
    uint32_t before = 0, round_total = 0;
    for (int k = 0; k < TV_BLOCK / 64; ++k) { before += k < wave ? s_wave[k] : 0u; round_total += s_wave[k]; }
    uint32_t at = base + before + ex;
    while (w) {
      int b = __ffs((int)w) - 1;
      w &= w - 1;
      q[at++] = (int32_t)(i * 32 + b);
    }
    base += round_total;
  }
}

}  // namespace
}  // namespace cga
