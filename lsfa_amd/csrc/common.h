// Shared host-side plumbing for liblsfa_hip.so: error reporting, argument checks,
// per-op HIP-event timing.  gfx950 only; no portability layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "lsfa_hip.h"

namespace lsfa {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// RAII bracket: records start/stop events on `stream` around an entry point's
// launches when lsfa_prof_enable(1) is active; otherwise free.
class ProfScope {
 public:
  ProfScope(int op, hipStream_t stream);
  ~ProfScope();
 private:
  hipStream_t stream_;
  int slot_;
};

// Function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) are per DEVICE: run the setter once
// per device the call site is used on.  Thread-safe; a setter that races with itself is idempotent.
class PerDeviceOnce {
 public:
  template <class F> void run(F&& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    const uint64_t bit = 1ULL << (d & 63);
    if (done_.load(std::memory_order_acquire) & bit) return;
    f();
    done_.fetch_or(bit, std::memory_order_release);
  }
 private:
  std::atomic<uint64_t> done_{0};
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace lsfa

#define LSFA_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::lsfa::set_error(__VA_ARGS__);      \
      return LSFA_EINVAL;                  \
    }                                      \
  } while (0)

#define LSFA_LAUNCH_CHECK(what)                                   \
  do {                                                            \
    hipError_t e__ = hipGetLastError();                           \
    if (e__ != hipSuccess) return ::lsfa::hip_fail(e__, what);    \
  } while (0)

// ---- device helpers shared by the kernels ------------------------------------------
// Correctly rounded float exp via the fp64 unit (full-rate on CDNA4): the oracle's
// definition of exp shared with the CPU checker ("Floating-point contract").
__device__ __forceinline__ float expf_cr(float x) { return (float)exp((double)x); }
