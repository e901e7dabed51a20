// model_format.cpp -- the text of a k-mer model file, natively.
//
// dynamont-train writes a model file after every batch (src/dynamont/segmentation/utils.py:136-152,
// train.py:221-224): header "kmer\tlevel_mean\tlevel_stdv", then f"{kmer}\t{mean}\t{stdev}\n" per k-mer -- 262 144
// rows for a 9-mer model, 0.25 s of Python string formatting per batch next to 25 ms of kernels. dyn_format_model
// produces the same bytes: a float64 prints as Python's repr() does (shortest digits that round-trip; fixed notation
// while the decimal point lies within 16 digits before / 4 zeros after it, else d.ddde+XX).
#include "../../include/dynamont_mi.h"

#include <charconv>
#include <cmath>
#include <cstring>

namespace {

// repr(float): returns the number of characters written (at most 25)
int py_repr(double x, char* out) {
  if (std::isnan(x)) {
    std::memcpy(out, "nan", 3);
    return 3;
  }
  if (std::isinf(x)) {
    const int n = x < 0 ? 4 : 3;
    std::memcpy(out, x < 0 ? "-inf" : "inf", n);
    return n;
  }
  char* p = out;
  if (std::signbit(x)) {
    *p++ = '-';
    x = -x;
  }
  if (x == 0.0) {
    std::memcpy(p, "0.0", 3);
    return (int)(p - out) + 3;
  }
  char sci[32];  // d[.ddd]e[+-]XX, shortest round-trip digits
  const auto res = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);
  char digits[24];
  int nd = 0, e = 0;
  const char* q = sci;
  for (; q < res.ptr && *q != 'e'; ++q)
    if (*q != '.') digits[nd++] = *q;
  ++q;  // past 'e'
  const bool eneg = *q == '-';
  ++q;
  for (; q < res.ptr; ++q) e = e * 10 + (*q - '0');
  if (eneg) e = -e;
  const int decpt = e + 1;  // value = 0.DIGITS x 10^decpt
  if (decpt > 16 || decpt < -3) {  // float_repr_style 'short', format code 'r' (Python/pystrtod.c)
    *p++ = digits[0];
    if (nd > 1) {
      *p++ = '.';
      std::memcpy(p, digits + 1, nd - 1);
      p += nd - 1;
    }
    *p++ = 'e';
    *p++ = e < 0 ? '-' : '+';
    int a = e < 0 ? -e : e;
    char t[4];
    int nt = 0;
    do {
      t[nt++] = (char)('0' + a % 10);
      a /= 10;
    } while (a);
    if (nt < 2) t[nt++] = '0';
    while (nt) *p++ = t[--nt];
  } else if (decpt <= 0) {
    *p++ = '0';
    *p++ = '.';
    for (int i = 0; i < -decpt; ++i) *p++ = '0';
    std::memcpy(p, digits, nd);
    p += nd;
  } else if (decpt >= nd) {
    std::memcpy(p, digits, nd);
    p += nd;
    for (int i = nd; i < decpt; ++i) *p++ = '0';
    *p++ = '.';
    *p++ = '0';
  } else {
    std::memcpy(p, digits, decpt);
    p += decpt;
    *p++ = '.';
    std::memcpy(p, digits + decpt, nd - decpt);
    p += nd - decpt;
  }
  return (int)(p - out);
}

}  // namespace

extern "C" uint64_t dyn_format_model(const char* kmers, int k, const double* mean, const double* stdev, uint64_t n,
                                     char* out, uint64_t cap) {
  static const char header[] = "kmer\tlevel_mean\tlevel_stdv\n";
  const uint64_t hdr = sizeof header - 1;
  const uint64_t worst = hdr + n * ((uint64_t)k + 2 * 25 + 3);
  if (!out || cap < worst) return worst;  // (the caller sizes the buffer with a first call)
  char* p = out;
  std::memcpy(p, header, hdr);
  p += hdr;
  for (uint64_t i = 0; i < n; ++i) {
    std::memcpy(p, kmers + i * (uint64_t)k, (size_t)k);
    p += k;
    *p++ = '\t';
    p += py_repr(mean[i], p);
    *p++ = '\t';
    p += py_repr(stdev[i], p);
    *p++ = '\n';
  }
  return (uint64_t)(p - out);
}
