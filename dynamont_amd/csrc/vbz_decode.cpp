// vbz_decode.cpp -- POD5 signal chunks, decoded natively.
//
// The reference reads raw signal through ONT's pod5 package (src/dynamont/pod5_io.py:1-16: `record.signal` /
// `record.signal_pa`), whose C++ core decompresses the signal table's VBZ chunks. That package is absent here; the
// vendor-free reader (dynamont_amd/pod5_native.py) decodes a chunk with NumPy at ~40 Msamples/s per core -- 13x slower than
// one MI355X aligns. This file is the same decode as a tight loop, so that the asynchronous engine's helper threads
// can decode the chunks of a batch straight into the pinned staging buffer (dyn_batch_align_vbz_async).
//
// VBZ (POD5 format specification, "signal compression"): zstd( svb16( zigzag( delta( int16 samples ))))
//   delta    d[0] = x[0], d[i] = x[i] - x[i-1]          (mod 2^16)
//   zigzag   z = (d << 1) ^ (d >> 15)                   (uint16)
//   svb16    ceil(n/8) key bytes, one bit per value (bit i%8 of byte i/8): 0 = one data byte, 1 = two (little endian);
//            the data bytes follow the keys
#include "vbz_decode.hpp"

#include <cstring>

#include "../../include/dynamont_mi.h"

namespace dynvbz {

dynzstd::Zstd& zstd() {
  static dynzstd::Zstd z;
  return z;
}

bool svb16_decode(const uint8_t* buf, size_t bytes, uint32_t n, int16_t* out) {
  const size_t kb = ((size_t)n + 7) / 8;
  if (kb > bytes) return false;
  const uint8_t* keys = buf;
  const uint8_t* data = buf + kb;
  const uint8_t* end = buf + bytes;
  uint16_t prev = 0;
  uint32_t i = 0;
  // eight values per key byte; the bounds check once per group (16 data bytes at most)
  for (; i + 8 <= n && data + 16 <= end; i += 8) {
    const unsigned k = keys[i >> 3];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned two = (k >> b) & 1u;
      const uint16_t v = (uint16_t)(data[0] | (two ? (unsigned)data[1] << 8 : 0u));
      data += 1 + two;
      prev = (uint16_t)(prev + (uint16_t)((v >> 1) ^ (uint16_t)(0u - (v & 1u))));
      out[i + b] = (int16_t)prev;
    }
  }
  for (; i < n; ++i) {
    const unsigned two = (keys[i >> 3] >> (i & 7)) & 1u;
    if (data + 1 + two > end) return false;
    const uint16_t v = (uint16_t)(data[0] | (two ? (unsigned)data[1] << 8 : 0u));
    data += 1 + two;
    prev = (uint16_t)(prev + (uint16_t)((v >> 1) ^ (uint16_t)(0u - (v & 1u))));
    out[i] = (int16_t)prev;
  }
  return true;
}

bool decode_chunk(const void* blob, size_t blob_bytes, uint32_t samples, int16_t* out, std::vector<uint8_t>& tmp, std::string& err) {
  dynzstd::Zstd& z = zstd();
  if (!z.load(err)) return false;  // (one acquire load once the library is in)
  if (samples == 0) return true;
  unsigned long long n = z.getFrameContentSize(blob, blob_bytes);
  const size_t worst = ((size_t)samples + 7) / 8 + 2 * (size_t)samples;
  if (n >= (1ull << 62)) n = worst;  // unknown: bound by the worst case of svb16
  if (n > worst) {
    err = "VBZ: chunk decompresses to more bytes than its sample count allows";
    return false;
  }
  if (tmp.size() < (size_t)n + 16) tmp.resize((size_t)n + 16);
  const size_t rc = z.decompress(tmp.data(), (size_t)n, blob, blob_bytes);
  if (z.isError(rc)) {
    err = std::string("VBZ: zstd: ") + z.getErrorName(rc);
    return false;
  }
  std::memset(tmp.data() + rc, 0, 16);  // the grouped loop reads up to 16 bytes ahead of its bounds check
  if (!svb16_decode(tmp.data(), rc, samples, out)) {
    err = "VBZ: truncated svb16 stream";
    return false;
  }
  return true;
}

}  // namespace dynvbz

extern "C" int dyn_vbz_decode(const void* blob, uint64_t blob_bytes, uint32_t samples, int16_t* out, char* err, uint64_t errcap) {
  if (!blob || (!out && samples)) return DYN_ERR_INVALID_ARGUMENT;
  std::vector<uint8_t> tmp;
  std::string e;
  if (!dynvbz::decode_chunk(blob, (size_t)blob_bytes, samples, out, tmp, e)) {
    if (err && errcap) std::snprintf(err, (size_t)errcap, "%s", e.c_str());
    return DYN_ERR_RUNTIME;
  }
  return DYN_OK;
}
