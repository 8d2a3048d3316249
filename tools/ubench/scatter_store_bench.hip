// Micro-benchmark: what does a partial-buffer store cost on MI355X as a function of the length and alignment of the pieces?
// Phase 1 of the tiled SpMV writes 1.24 GB per iteration as runs of consecutive 4-byte slots; a (destination tile, source tile)
// block is one contiguous piece, pieces of different blocks are written at very different times.  Each wavefront here writes
// `piece` consecutive floats per store instruction group (64 lanes cover 64 / piece pieces), piece start = random position
// (aligned to `align` floats) in a buffer of `total` floats; every float of the buffer is written exactly once per pass when
// pieces tile the buffer (permuted piece order), so the bytes are the same for every variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// piece q (of n_pieces, each `piece` floats, laid out at q * stride + shift) is written by lanes; order of pieces = bit-mixed
__global__ void __launch_bounds__(256) k_store(float* out, uint32_t n_pieces, uint32_t log_pieces, uint32_t piece, uint32_t stride, uint32_t shift, uint32_t mul)
{
  uint64_t const t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t const nthreads = (uint64_t)gridDim.x * blockDim.x;
  uint64_t const total = (uint64_t)n_pieces * piece;
  for (uint64_t e = t; e < total; e += nthreads) {
    uint32_t const q = (uint32_t)(e / piece), k = (uint32_t)(e % piece);
    uint32_t const p = (q * mul) & (n_pieces - 1);  // odd multiplier: a permutation of the pieces (n_pieces is a power of two)
    out[(uint64_t)p * stride + shift + k] = (float)k;
  }
}

int main()
{
  size_t const bytes = (size_t)1 << 31;  // 2 GiB buffer, 1.07 GB written per pass (stride = 2 * piece leaves gaps when asked)
  float* buf;
  CK(hipMalloc(&buf, bytes + 4096));
  CK(hipMemset(buf, 0, bytes + 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("piece_floats  stride  shift  order      GB/s (bytes written / time)\n");
  for (uint32_t piece : {4u, 8u, 16u, 24u, 32u, 64u, 128u, 256u, 1024u}) {
    for (int variant = 0; variant < 4; ++variant) {
      // 0: dense tiling, scattered order   1: dense tiling, sequential order   2: misaligned by 3 floats, scattered   3: padded to 2x (every other piece a hole), scattered
      if (piece % 16 && variant == 3) continue;
      uint32_t const stride = variant == 3 ? 2 * piece : piece;
      uint64_t const total_floats = (bytes / 4) / (variant == 3 ? 2 : 1) / 2;  // 1.07 GB written
      uint32_t n_pieces = 1;
      uint32_t lg = 0;
      while ((uint64_t)(n_pieces * 2ull) * piece <= total_floats) { n_pieces *= 2; ++lg; }
      uint32_t const mul = variant == 1 ? 1u : 2654435761u;
      uint32_t const shift = variant == 2 ? 3u : 0u;
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_store, 256 * 8, 256, 0, 0, buf, n_pieces, lg, piece, stride, shift, mul);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      double const gb = (double)n_pieces * piece * 4 / 1e9;
      printf("%6u %9u %5u  %-10s %8.1f  (%.2f GB in %.3f ms)\n", piece, stride, shift, variant == 1 ? "sequential" : variant == 3 ? "scatter+pad" : "scattered", gb / (ms * 1e-3), gb, ms);
    }
  }
  return 0;
}
