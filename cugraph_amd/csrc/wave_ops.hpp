// wave64 cross-lane helpers for gfx950, built on DPP (data-parallel primitives: one VALU op per step instead of an
// LDS-crossbar ds_bpermute plus address arithmetic).  Row = 16 lanes.  Encodings: row_shr:n = 0x110+n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143, wave_shr:1 = 0x138 (gfx9 family incl. gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace cga {

template <typename WT>
__device__ __forceinline__ WT group_sum(WT v, int width)
{
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t src)
{  // lanes without a valid source (or outside ROW_MASK) read 0 = identity of the operators below
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, 0xF, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_val(float src) { return __uint_as_float(dpp_u32<CTRL, ROW_MASK>(__float_as_uint(src))); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_val(double src)
{
  unsigned long long b = (unsigned long long)__double_as_longlong(src);
  uint32_t lo = dpp_u32<CTRL, ROW_MASK>((uint32_t)b), hi = dpp_u32<CTRL, ROW_MASK>((uint32_t)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// inclusive segmented scan over the 64 lanes: s = sum of the lane values back to (and including) the nearest lane
// whose count c is non-zero; c = inclusive sum of counts.  Operator (left (+) right) = (right.c ? right.s : left.s + right.s).
template <typename WT, int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_step(WT& s, uint32_t& c)
{
  WT ts       = dpp_val<CTRL, ROW_MASK>(s);
  uint32_t tc = dpp_u32<CTRL, ROW_MASK>(c);
  s           = c == 0 ? s + ts : s;
  c += tc;
}
template <typename WT>
__device__ __forceinline__ void wave_seg_scan(WT& s, uint32_t& c)
{
  seg_step<WT, 0x111, 0xF>(s, c);  // row_shr:1
  seg_step<WT, 0x112, 0xF>(s, c);  // row_shr:2
  seg_step<WT, 0x114, 0xF>(s, c);  // row_shr:4
  seg_step<WT, 0x118, 0xF>(s, c);  // row_shr:8
  seg_step<WT, 0x142, 0xA>(s, c);  // row_bcast:15 -> rows 1, 3
  seg_step<WT, 0x143, 0xC>(s, c);  // row_bcast:31 -> rows 2, 3
}
// inclusive prefix sum over the 64 lanes (lane 63 ends up with the wave total)
template <typename WT>
__device__ __forceinline__ WT wave_sum_to_lane63(WT s)
{
  s += dpp_val<0x111, 0xF>(s);
  s += dpp_val<0x112, 0xF>(s);
  s += dpp_val<0x114, 0xF>(s);
  s += dpp_val<0x118, 0xF>(s);
  s += dpp_val<0x142, 0xA>(s);
  s += dpp_val<0x143, 0xC>(s);
  return s;
}
__device__ __forceinline__ uint32_t wave_inclusive_sum_u32(uint32_t c)
{
  c += dpp_u32<0x111, 0xF>(c);
  c += dpp_u32<0x112, 0xF>(c);
  c += dpp_u32<0x114, 0xF>(c);
  c += dpp_u32<0x118, 0xF>(c);
  c += dpp_u32<0x142, 0xA>(c);
  c += dpp_u32<0x143, 0xC>(c);
  return c;
}
template <typename WT> __device__ __forceinline__ WT read_lane63(WT v);
template <> __device__ __forceinline__ float read_lane63<float>(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63)); }
template <> __device__ __forceinline__ double read_lane63<double>(double v)
{
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

}  // namespace cga
