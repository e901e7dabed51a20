// zstd_dl.hpp -- the handful of libzstd entry points the library uses, bound with dlopen: the ROCm image ships
// libzstd.so.1 (1.4.8) without its header, and nothing but the CSV sink (one-frame compression) and the POD5 signal
// decoder (VBZ chunks) needs it.
#pragma once

#include <dlfcn.h>

#include <atomic>
#include <cstddef>
#include <mutex>
#include <string>

namespace dynzstd {

struct Zstd {
  void* lib = nullptr;
  void* (*createCCtx)() = nullptr;
  size_t (*freeCCtx)(void*) = nullptr;
  size_t (*compressBegin)(void*, int) = nullptr;
  size_t (*compressContinue)(void*, void*, size_t, const void*, size_t) = nullptr;
  size_t (*compressEnd)(void*, void*, size_t, const void*, size_t) = nullptr;
  void (*invalidateRepCodes)(void*) = nullptr;
  size_t (*compressBound)(size_t) = nullptr;
  size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
  unsigned long long (*getFrameContentSize)(const void*, size_t) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  const char* (*getErrorName)(size_t) = nullptr;

  // Thread-safe and idempotent: the first VBZ batch of a process decodes on every helper thread at once. `ready` is
  // published (release) only after EVERY entry point has been resolved; a caller that sees it set (acquire) may use
  // them all, anybody else queues behind the mutex. A failed load leaves the object untouched for the next attempt.
  std::atomic<bool> ready{false};
  std::mutex load_m;

  bool load(std::string& err) {
    if (ready.load(std::memory_order_acquire)) return true;
    std::lock_guard<std::mutex> lk(load_m);
    if (ready.load(std::memory_order_relaxed)) return true;
    void* h = nullptr;
    for (const char* name : {"libzstd.so.1", "libzstd.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) {
      err = "libzstd.so.1 not found";
      return false;
    }
#define DYN_Z(field, name)                                        \
  field = reinterpret_cast<decltype(field)>(dlsym(h, name));      \
  if (!field) {                                                   \
    err = std::string("libzstd lacks ") + name;                   \
    dlclose(h);                                                   \
    return false;                                                 \
  }
    DYN_Z(createCCtx, "ZSTD_createCCtx");
    DYN_Z(freeCCtx, "ZSTD_freeCCtx");
    DYN_Z(compressBegin, "ZSTD_compressBegin");
    DYN_Z(compressContinue, "ZSTD_compressContinue");
    DYN_Z(compressEnd, "ZSTD_compressEnd");
    DYN_Z(invalidateRepCodes, "ZSTD_invalidateRepCodes");
    DYN_Z(compressBound, "ZSTD_compressBound");
    DYN_Z(decompress, "ZSTD_decompress");
    DYN_Z(getFrameContentSize, "ZSTD_getFrameContentSize");
    DYN_Z(isError, "ZSTD_isError");
    DYN_Z(getErrorName, "ZSTD_getErrorName");
#undef DYN_Z
    lib = h;
    ready.store(true, std::memory_order_release);
    return true;
  }
};

}  // namespace dynzstd
