// engine_internal.hpp -- what the translation units of the host engine share beyond engine.hpp (dynamont_mi.cpp: the C ABI,
// buffers, classic launches; session.cpp: the resident read queue's host side).
#pragma once

#include "engine.hpp"

#include <cstdint>
#include <string>
#include <vector>

// HIP failure inside a function that returns a dyn status: the text goes to the handle (under err_mu: the pipeline's back
// thread reports failures without the handle's lock), the status is returned
#define HIP_TRY(a, expr)                                                                  \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      {                                                                                   \
        std::lock_guard<std::mutex> _elk((a)->err_mu);                                    \
        (a)->last_error = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr; \
      }                                                                                   \
      return _e == hipErrorOutOfMemory ? DYN_ERR_OUT_OF_MEMORY : DYN_ERR_DEVICE;          \
    }                                                                                     \
  } while (0)

namespace dyneng {

// bytes of lattice pools that destroyed handles have parked on `device` (they are this process's to use)
size_t parked_bytes(int device);
// give back what destroyed handles have parked on `device`
void free_parked(int device);
// a destroyed handle's pool array stays with the process for the next handle on that device
void park_pool_buffer(int device, int kind, DevBuf& b);
// the three arrays of a lattice pool together (parked buffers taken over where they fit; out of memory: everything the handle
// holds of the pool is released and the three are allocated again at their own sizes)
hipError_t ensure_pool(int device, DevBuf& ws, size_t ws_bytes, DevBuf& lpe, size_t lpe_bytes, DevBuf& bits, size_t bits_bytes, double headroom);
// 0 = the read carries no structural tie; else the forward rows strict mode "ties" runs bit for bit (0xffffffff = all)
uint32_t tie_rows(const dynhost::PoreModel& m, const int32_t* km, uint64_t kc, uint64_t S);
// queue order of a page-starved launch (host replay of the launch; `order` comes in longest first)
void plan_queue(std::vector<uint32_t>& order, const std::vector<uint32_t>& need, const std::vector<uint64_t>& rows, size_t n_slots, uint64_t pool_pages);
// low-discrepancy order of a paged session's ticket (`order` comes in longest first)
void spread_order(std::vector<uint32_t>& order, int tail_div);
constexpr int SESSION_TAIL_DIV = 8;

}  // namespace dyneng
