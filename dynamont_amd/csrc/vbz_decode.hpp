// vbz_decode.hpp -- POD5 VBZ signal chunks (see vbz_decode.cpp)
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "zstd_dl.hpp"

namespace dynvbz {

dynzstd::Zstd& zstd();
// n int16 samples from an svb16 + zigzag + delta stream; false = truncated input
bool svb16_decode(const uint8_t* buf, size_t bytes, uint32_t n, int16_t* out);
// one VBZ chunk (zstd frame around svb16); tmp = scratch reused across calls of one thread
bool decode_chunk(const void* blob, size_t blob_bytes, uint32_t samples, int16_t* out, std::vector<uint8_t>& tmp, std::string& err);

}  // namespace dynvbz
