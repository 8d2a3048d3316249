// MatrixMarket coordinate files -> device edge list (the on-disk format the reference's datasets and
// C++ tests use: datasets/*.mtx, cpp/tests/utilities/matrix_market_file_utilities.cu read_edgelist_from_matrix_market_file).
// Same conventions as that reader: 1-based ids become 0-based, `pattern` files get weight 1, a `symmetric` file stores one
// triangle and every off-diagonal entry is mirrored, the vertex count is the matrix dimension (isolated vertices survive).
// Host-side parse (one pass over the text, strtol / strtod), then one upload; the result is a cugraph_coo_t.
#include "common.hpp"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cugraph_amd/extensions.h"

using namespace cga;

namespace {

std::string lower(std::string s)
{
  for (auto& c : s) c = (char)std::tolower((unsigned char)c);
  return s;
}

}  // namespace

extern "C" cugraph_error_code_t cugraph_amd_read_matrix_market(const cugraph_resource_handle_t* handle, const char* path, cugraph_coo_t** result,
                                                               size_t* num_vertices, bool_t* is_symmetric, bool_t* has_weights,
                                                               cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(path != nullptr && result != nullptr, CUGRAPH_INVALID_INPUT, "path / result is NULL");
    FILE* f = std::fopen(path, "rb");
    CGA_EXPECTS(f != nullptr, CUGRAPH_INVALID_INPUT, std::string("cannot open ") + path);
    std::string text;
    {
      char buf[1 << 16];
      size_t got;
      while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
      std::fclose(f);
    }
    char const* p   = text.c_str();
    char const* end = p + text.size();
    auto next_line  = [&](char const* q) { while (q < end && *q != '\n') ++q; return q < end ? q + 1 : end; };
    // banner: %%MatrixMarket matrix coordinate <field> <symmetry>
    char const* eol = next_line(p);
    std::string banner = lower(std::string(p, eol));
    CGA_EXPECTS(banner.rfind("%%matrixmarket", 0) == 0, CUGRAPH_INVALID_INPUT, "not a MatrixMarket file (banner missing)");
    CGA_EXPECTS(banner.find("matrix") != std::string::npos && banner.find("coordinate") != std::string::npos, CUGRAPH_INVALID_INPUT,
                "only 'matrix coordinate' MatrixMarket files are supported");
    bool const pattern = banner.find("pattern") != std::string::npos;
    CGA_EXPECTS(pattern || banner.find("real") != std::string::npos || banner.find("integer") != std::string::npos, CUGRAPH_INVALID_INPUT,
                "MatrixMarket field must be real, integer or pattern");
    bool const symmetric = banner.find("symmetric") != std::string::npos && banner.find("skew") == std::string::npos;
    CGA_EXPECTS(symmetric || banner.find("general") != std::string::npos, CUGRAPH_INVALID_INPUT, "MatrixMarket symmetry must be general or symmetric");
    p = eol;
    while (p < end && (*p == '%' || *p == '\n' || *p == '\r')) p = next_line(p);
    char* q        = nullptr;
    long long rows = std::strtoll(p, &q, 10);
    p              = q;
    long long cols = std::strtoll(p, &q, 10);
    p              = q;
    long long nnz  = std::strtoll(p, &q, 10);
    CGA_EXPECTS(q != p && rows >= 0 && cols >= 0 && nnz >= 0, CUGRAPH_INVALID_INPUT, "malformed MatrixMarket size line");
    p = q;
    CGA_EXPECTS(rows == cols, CUGRAPH_INVALID_INPUT, "the matrix of a graph must be square");
    CGA_EXPECTS(rows < (long long)INT32_MAX && nnz < (long long)INT32_MAX / 2, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "matrix too large for 32-bit ids");
    std::vector<int32_t> s, d;
    std::vector<float> w;
    s.reserve((size_t)nnz * (symmetric ? 2 : 1));
    d.reserve(s.capacity());
    w.reserve(s.capacity());
    for (long long k = 0; k < nnz; ++k) {
      long long i = std::strtoll(p, &q, 10);
      CGA_EXPECTS(q != p, CUGRAPH_INVALID_INPUT, "MatrixMarket file ends before its declared number of entries");
      p           = q;
      long long j = std::strtoll(p, &q, 10);
      CGA_EXPECTS(q != p, CUGRAPH_INVALID_INPUT, "malformed MatrixMarket entry");
      p        = q;
      double v = 1.0;
      if (!pattern) {
        v = std::strtod(p, &q);
        CGA_EXPECTS(q != p, CUGRAPH_INVALID_INPUT, "malformed MatrixMarket entry (value missing)");
        p = q;
      }
      CGA_EXPECTS(i >= 1 && i <= rows && j >= 1 && j <= cols, CUGRAPH_INVALID_INPUT, "MatrixMarket index out of range");
      s.push_back((int32_t)(i - 1)); d.push_back((int32_t)(j - 1)); w.push_back((float)v);
      if (symmetric && i != j) { s.push_back((int32_t)(j - 1)); d.push_back((int32_t)(i - 1)); w.push_back((float)v); }
    }
    HIP_TRY(hipSetDevice(h.device));
    size_t const n = s.size();
    auto coo       = std::make_unique<coo_t>();
    coo->src       = new device_array_t(n, INT32);
    coo->dst       = new device_array_t(n, INT32);
    coo->wgt       = new device_array_t(n, FLOAT32);
    if (n > 0) {
      HIP_TRY(hipMemcpyAsync(coo->src->buf.ptr, s.data(), n * 4, hipMemcpyHostToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(coo->dst->buf.ptr, d.data(), n * 4, hipMemcpyHostToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(coo->wgt->buf.ptr, w.data(), n * 4, hipMemcpyHostToDevice, h.stream));
    }
    h.sync();  // the host vectors go out of scope
    if (num_vertices) *num_vertices = (size_t)rows;
    if (is_symmetric) *is_symmetric = symmetric ? TRUE : FALSE;
    if (has_weights) *has_weights = pattern ? FALSE : TRUE;
    *result = reinterpret_cast<cugraph_coo_t*>(coo.release());
  });
}
