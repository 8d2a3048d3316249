#!/usr/bin/env bash
# Compile the reference's OWN CPU reference functions, from the sources where they lie under
# /root/reference, into oracle/_ref/libref.so (git-ignored; travels to the GPU box as a built .so).
# These three templates are what the reference's gtest suite uses as its definition of correctness:
#   pagerank_reference  cpp/tests/link_analysis/pagerank_test.cpp:33-121
#   bfs_reference       cpp/tests/traversal/bfs_test.cpp:32-70
#   sssp_reference      cpp/tests/traversal/sssp_test.cpp:33-75
# The test .cpp files as a whole need gtest/raft/rmm/CCCL (not vendored), so the reference's own
# build system is NOT run; instead the translation unit below #includes each function's line range
# straight from the reference file (via the preprocessor on an awk-extracted temporary that is deleted
# after compilation -- no reference source is kept in this repository) behind a 10-line shim that
# supplies ASSERT_TRUE and cugraph::invalid_vertex_id.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${CUGRAPH_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/cpp/tests" ]; then
  echo "build_ref.sh: $REF not present; keeping any prebuilt $OUT/libref.so" >&2
  exit 0
fi
mkdir -p "$OUT"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT

# extract "template <...>\nvoid NAME(" ... up to the first line that is exactly "}"
extract() { # file function_name out
  awk -v fn="$2" '
    /^template </ { hold=$0; held=1; next }
    held && $0 ~ "^void " fn "\\(" { print hold; print; on=1; held=0; next }
    { held=0 }
    on { print; if ($0 == "}") { exit } }
  ' "$1" > "$3"
  test -s "$3" || { echo "build_ref.sh: could not find $2 in $1" >&2; exit 1; }
}
extract "$REF/cpp/tests/link_analysis/pagerank_test.cpp" pagerank_reference "$TMP/pagerank_reference.inc"
extract "$REF/cpp/tests/traversal/bfs_test.cpp"          bfs_reference      "$TMP/bfs_reference.inc"
extract "$REF/cpp/tests/traversal/sssp_test.cpp"         sssp_reference     "$TMP/sssp_reference.inc"

cat > "$TMP/ref_tu.cpp" <<'CPP'
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <iterator>
#include <limits>
#include <numeric>
#include <optional>
#include <queue>
#include <tuple>
#include <vector>
// ---- shim for the two non-standard names the extracted functions use
static thread_local int g_ref_failed = 0;
#define ASSERT_TRUE(c) do { if (!(c)) { g_ref_failed = 1; return; } } while (0)
namespace cugraph { template <typename T> struct invalid_vertex_id { static constexpr T value = T(-1); }; }
// ---- the reference's functions, verbatim from /root/reference at build time
#include "pagerank_reference.inc"
#include "bfs_reference.inc"
#include "sssp_reference.inc"
// ---- C entry points (int32 vertices; CSR/CSC offsets widened by the caller to int64 -> edge_t=int64)
extern "C" {
int ref_pagerank_f32(int64_t nv, const int64_t* off, const int32_t* idx, const float* w, int64_t n_pers,
                     const int32_t* pv, const float* pval, float* pr, float alpha, float eps,
                     int64_t max_iter, int has_guess)
{
  g_ref_failed = 0;
  pagerank_reference<int32_t, int64_t, float, float>(
    off, idx, w ? std::make_optional<float const*>(w) : std::nullopt,
    n_pers ? std::make_optional<int32_t const*>(pv) : std::nullopt,
    n_pers ? std::make_optional<float const*>(pval) : std::nullopt,
    n_pers ? std::make_optional<int32_t>((int32_t)n_pers) : std::nullopt, pr, (int32_t)nv, alpha, eps,
    (size_t)max_iter, has_guess != 0);
  return g_ref_failed;
}
int ref_pagerank_f64(int64_t nv, const int64_t* off, const int32_t* idx, const double* w, int64_t n_pers,
                     const int32_t* pv, const double* pval, double* pr, double alpha, double eps,
                     int64_t max_iter, int has_guess)
{
  g_ref_failed = 0;
  pagerank_reference<int32_t, int64_t, double, double>(
    off, idx, w ? std::make_optional<double const*>(w) : std::nullopt,
    n_pers ? std::make_optional<int32_t const*>(pv) : std::nullopt,
    n_pers ? std::make_optional<double const*>(pval) : std::nullopt,
    n_pers ? std::make_optional<int32_t>((int32_t)n_pers) : std::nullopt, pr, (int32_t)nv, alpha, eps,
    (size_t)max_iter, has_guess != 0);
  return g_ref_failed;
}
void ref_bfs(int64_t nv, const int64_t* off, const int32_t* idx, int32_t* dist, int32_t* pred,
             int32_t source, int32_t depth_limit)
{
  bfs_reference<int32_t, int64_t>(off, idx, dist, pred, (int32_t)nv, source, depth_limit);
}
void ref_sssp_f32(int64_t nv, const int64_t* off, const int32_t* idx, const float* w, float* dist,
                  int32_t* pred, int32_t source, float cutoff)
{
  sssp_reference<int32_t, int64_t, float>(off, idx, w, dist, pred, (int32_t)nv, source, cutoff);
}
void ref_sssp_f64(int64_t nv, const int64_t* off, const int32_t* idx, const double* w, double* dist,
                  int32_t* pred, int32_t source, double cutoff)
{
  sssp_reference<int32_t, int64_t, double>(off, idx, w, dist, pred, (int32_t)nv, source, cutoff);
}
}
CPP
g++ -O2 -std=c++17 -fPIC -shared -I"$TMP" -o "$OUT/libref.so" "$TMP/ref_tu.cpp"
echo "built $OUT/libref.so from $REF"
