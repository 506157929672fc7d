// Error reporting + live per-op timing for liblsfa_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace lsfa {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

// ---- profiling ------------------------------------------------------------------
struct EventPair { hipEvent_t a, b; int op; };
static std::mutex g_mu;
static unsigned g_prof = 0;  // bit i set: op i is timed
static std::vector<EventPair> g_live;
static std::vector<EventPair> g_free;

ProfScope::ProfScope(int op, hipStream_t stream) : stream_(stream), slot_(-1) {
  if (!((g_prof >> op) & 1u)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  EventPair p;
  if (!g_free.empty()) { p = g_free.back(); g_free.pop_back(); }
  else {
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
  }
  p.op = op;
  (void)hipEventRecord(p.a, stream_);
  g_live.push_back(p);
  slot_ = (int)g_live.size() - 1;
}

ProfScope::~ProfScope() {
  if (slot_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot_ < (int)g_live.size()) (void)hipEventRecord(g_live[slot_].b, stream_);
}

}  // namespace lsfa

using namespace lsfa;

extern "C" const char* lsfa_last_error(void) { return g_err; }
extern "C" int lsfa_abi_version(void) { return 1; }

// A stream of our own.  PyTorch hands out streams from a pool of 32 per priority and wraps around, so two
// `torch.cuda.Stream()` objects may be ONE hipStream_t; its BLAS workspace is keyed by (handle, stream), so
// graphs captured on such twins bake the same split-K / stream-K scratch in and corrupt (or dead-lock: the
// library's kernels spin on flags in that scratch) each other when they replay concurrently.
extern "C" int lsfa_stream_create(void** stream_out, int high_priority) {
  LSFA_REQUIRE(stream_out, "lsfa_stream_create: NULL argument");
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = least, hi = greatest priority (numerically lower)
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo);
  if (e != hipSuccess) return hip_fail(e, "lsfa_stream_create");
  *stream_out = (void*)s;
  return LSFA_OK;
}

// Up to four device-to-device copies of 4-byte elements as ONE launch: a frame's image, motion vectors and residual go
// into the static buffers a captured graph reads from; three separate copy nodes are three ~5 us latencies per frame.
namespace {
constexpr int kCopyJobsMax = 32;
struct CopyJobs { void* dst[kCopyJobsMax]; const void* src[kCopyJobsMax]; long end[kCopyJobsMax]; long elems[kCopyJobsMax]; };     // end = running total of 4-element groups
__global__ __launch_bounds__(256) void copy_many_kernel(CopyJobs j, int njobs) {
  // the job table through LDS: indexing the by-value argument with a run-time job number would send it to scratch
  __shared__ long s_end[kCopyJobsMax], s_elems[kCopyJobsMax];
  __shared__ uintptr_t s_dst[kCopyJobsMax], s_src[kCopyJobsMax];
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < kCopyJobsMax; ++k) {
      s_end[k] = j.end[k]; s_elems[k] = j.elems[k];
      s_dst[k] = reinterpret_cast<uintptr_t>(j.dst[k]); s_src[k] = reinterpret_cast<uintptr_t>(j.src[k]);
    }
  }
  __syncthreads();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= s_end[njobs - 1]) return;
  int k = 0;
  while (i >= s_end[k]) ++k;
  const long o = (i - (k ? s_end[k - 1] : 0)) * 4;            // first element of this thread's group
  uint32_t* d = reinterpret_cast<uint32_t*>(s_dst[k]) + o;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(s_src[k]) + o;
  const bool vec = o + 4 <= s_elems[k] && ((s_dst[k] | s_src[k]) & 15) == 0;
  if (s_src[k] == 0) {      // r5: a job without a source zero-fills (amax slots, status words, the padding channels of a Concat map)
    if (vec) *reinterpret_cast<uint4*>(d) = make_uint4(0u, 0u, 0u, 0u);
    else for (long e = 0; e < 4 && o + e < s_elems[k]; ++e) d[e] = 0u;
  } else if (vec) {
    *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
  } else {
    for (long e = 0; e < 4 && o + e < s_elems[k]; ++e) d[e] = s[e];
  }
}
}  // namespace

extern "C" int lsfa_copy_many(int njobs, void* const* dst, const void* const* src, const long* elems4, void* stream) {
  LSFA_REQUIRE(njobs >= 1 && njobs <= kCopyJobsMax && dst && src && elems4, "lsfa_copy_many: 1..32 jobs");
  CopyJobs j;
  long total = 0;
  for (int k = 0; k < kCopyJobsMax; ++k) {
    if (k < njobs) {
      LSFA_REQUIRE(dst[k] && elems4[k] >= 0, "lsfa_copy_many: NULL destination or negative size");
      total += (elems4[k] + 3) / 4;
    }
    j.dst[k] = k < njobs ? dst[k] : nullptr;
    j.src[k] = k < njobs ? src[k] : nullptr;
    j.elems[k] = k < njobs ? elems4[k] : 0;
    j.end[k] = total;
  }
  if (total == 0) return LSFA_OK;
  hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, j, njobs);
  LSFA_LAUNCH_CHECK("lsfa_copy_many");
  return LSFA_OK;
}

// r6: a device table of pointers (the per-image bases lsfa_avgpool_nchw_tbl / lsfa_stem_conv7x7s2_tbl read) rewritten from host values by one
// tiny launch: the values travel as kernel arguments, so the call is captured-graph friendly on the CALLER's side only (a replayed graph reads the
// table; the table itself is rewritten before each replay, in stream order)
namespace {
constexpr int kPtrTableMax = 64;
struct PtrJobs { const void* p[kPtrTableMax]; };
__global__ void ptr_table_kernel(const void** __restrict__ table, PtrJobs j, int n) {
  const int i = threadIdx.x;
  if (i < n) table[i] = j.p[i];
}
}  // namespace

extern "C" int lsfa_ptr_table_set(void** table_dev, int n, const void* const* ptrs_host, void* stream) {
  LSFA_REQUIRE(table_dev && ptrs_host && n >= 1, "lsfa_ptr_table_set: NULL argument or empty table");
  for (int i0 = 0; i0 < n; i0 += kPtrTableMax) {
    PtrJobs j;
    const int m = n - i0 < kPtrTableMax ? n - i0 : kPtrTableMax;
    for (int k = 0; k < kPtrTableMax; ++k) j.p[k] = k < m ? ptrs_host[i0 + k] : nullptr;
    hipLaunchKernelGGL(ptr_table_kernel, dim3(1), dim3(kPtrTableMax), 0, (hipStream_t)stream, (const void**)table_dev + i0, j, m);
  }
  LSFA_LAUNCH_CHECK("lsfa_ptr_table_set");
  return LSFA_OK;
}

extern "C" int lsfa_stream_destroy(void* stream) {
  if (!stream) return LSFA_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "lsfa_stream_destroy");
  return LSFA_OK;
}

extern "C" int lsfa_prof_enable(int mask) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof = (unsigned)mask;
  return LSFA_OK;
}

extern "C" int lsfa_prof_read(double* ms_host, int* launches_host) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < LSFA_OP_COUNT; ++i) { ms_host[i] = 0.0; launches_host[i] = 0; }
  for (auto& p : g_live) {
    hipError_t e = hipEventSynchronize(p.b);
    if (e != hipSuccess) return hip_fail(e, "lsfa_prof_read: hipEventSynchronize");
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, p.a, p.b);
    if (e != hipSuccess) return hip_fail(e, "lsfa_prof_read: hipEventElapsedTime");
    ms_host[p.op] += ms;
    launches_host[p.op] += 1;
    g_free.push_back(p);
  }
  g_live.clear();
  return LSFA_OK;
}

extern "C" const char* lsfa_op_name(int op) {
  static const char* names[LSFA_OP_COUNT] = {"psroi_pool", "rfcn_head", "warp_bilinear", "aggregate",
                                             "proposal", "nms", "det_postprocess", "deform_im2col",
                                             "scale_shift_relu", "conv_nhwc", "stem", "flownet_small"};
  return (op >= 0 && op < LSFA_OP_COUNT) ? names[op] : "?";
}
