// NMS on score-sorted boxes: device-resident entry point and the reference's
// host-pointer `_nms` (lib/nms/gpu_nms.hpp:14-15).  Kernels: nms_kernels.h.
#include <mutex>
#include <vector>

#include "nms_kernels.h"

using namespace lsfa;

extern "C" size_t lsfa_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  const size_t col_blocks = (size_t)ceil_div(n, 64);
  // mask (n, col_blocks) + transposed diagonal words (n)
  return align_up((size_t)n * col_blocks * sizeof(uint64_t), 256) + align_up((size_t)n * sizeof(uint64_t), 256);
}

extern "C" int lsfa_nms_sorted(const float* boxes, int n, int box_dim, float thresh, int* keep, int* num_keep,
                               void* ws, size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(n >= 0 && box_dim >= 4, "lsfa_nms_sorted: bad shape n=%d box_dim=%d", n, box_dim);
  LSFA_REQUIRE(keep && num_keep, "lsfa_nms_sorted: keep/num_keep must be non-NULL");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int), s);
    if (e != hipSuccess) return hip_fail(e, "lsfa_nms_sorted: hipMemsetAsync");
    return LSFA_OK;
  }
  LSFA_REQUIRE(boxes && ws, "lsfa_nms_sorted: boxes/ws must be non-NULL");
  const int col_blocks = ceil_div(n, 64);
  if (col_blocks > kSweepMaxBlocks) {
    set_error("lsfa_nms_sorted: n=%d exceeds the supported %d boxes", n, kSweepMaxBlocks * 64);
    return LSFA_ENOTSUP;
  }
  if (ws_bytes < lsfa_nms_workspace_bytes(n)) {
    set_error("lsfa_nms_sorted: workspace %zu < %zu bytes", ws_bytes, lsfa_nms_workspace_bytes(n));
    return LSFA_EWORKSPACE;
  }
  uint64_t* mask = (uint64_t*)ws;
  uint64_t* diagT = (uint64_t*)((unsigned char*)ws + align_up((size_t)n * col_blocks * sizeof(uint64_t), 256));
  ProfScope prof(LSFA_OP_NMS, s);
  hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(ceil_div(nms_tile_count(col_blocks), 4), 1, 1), dim3(256), 0, s, boxes,
                     (long)n * box_dim, box_dim, (const int*)nullptr, n, make_iou_test(thresh), mask, diagT, col_blocks, 0);
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), 0, s, (const uint64_t*)mask, (const uint64_t*)diagT, n,
                     col_blocks, n, keep, num_keep, (const float4*)nullptr, (const uint32_t*)nullptr, (float*)nullptr,
                     (float*)nullptr);
  LSFA_LAUNCH_CHECK("lsfa_nms_sorted");
  return LSFA_OK;
}

extern "C" int lsfa_nms_sorted_f64(const double* boxes, int n, int box_dim, double thresh, int* keep, int* num_keep,
                                   void* ws, size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(n >= 0 && box_dim >= 4, "lsfa_nms_sorted_f64: bad shape n=%d box_dim=%d", n, box_dim);
  LSFA_REQUIRE(keep && num_keep, "lsfa_nms_sorted_f64: keep/num_keep must be non-NULL");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int), s);
    if (e != hipSuccess) return hip_fail(e, "lsfa_nms_sorted_f64: hipMemsetAsync");
    return LSFA_OK;
  }
  LSFA_REQUIRE(boxes && ws, "lsfa_nms_sorted_f64: boxes/ws must be non-NULL");
  const int col_blocks = ceil_div(n, 64);
  if (col_blocks > kSweepMaxBlocks) {
    set_error("lsfa_nms_sorted_f64: n=%d exceeds the supported %d boxes", n, kSweepMaxBlocks * 64);
    return LSFA_ENOTSUP;
  }
  if (ws_bytes < lsfa_nms_workspace_bytes(n)) {
    set_error("lsfa_nms_sorted_f64: workspace %zu < %zu bytes", ws_bytes, lsfa_nms_workspace_bytes(n));
    return LSFA_EWORKSPACE;
  }
  uint64_t* mask = (uint64_t*)ws;
  uint64_t* diagT = (uint64_t*)((unsigned char*)ws + align_up((size_t)n * col_blocks * sizeof(uint64_t), 256));
  ProfScope prof(LSFA_OP_NMS, s);
  hipLaunchKernelGGL(nms_mask_f64_kernel, dim3(ceil_div(nms_tile_count(col_blocks), 4)), dim3(256), 0, s, boxes, box_dim, n,
                     make_iou_test64(thresh), mask, diagT, col_blocks);
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), 0, s, (const uint64_t*)mask, (const uint64_t*)diagT, n,
                     col_blocks, n, keep, num_keep, (const float4*)nullptr, (const uint32_t*)nullptr, (float*)nullptr,
                     (float*)nullptr);
  LSFA_LAUNCH_CHECK("lsfa_nms_sorted_f64");
  return LSFA_OK;
}

namespace {
// `_nms` keeps the reference's host-pointer, synchronous contract, but not its three cudaMalloc/cudaFree
// per call (nms_kernel.cu:113-146): one scratch block per device, grown on demand, reused under a lock.
struct HostNmsScratch { void* p = nullptr; size_t cap = 0; };
std::mutex g_host_nms_mu;
HostNmsScratch g_host_nms[64];
}  // namespace

// lib/nms/nms_kernel.cu:97-150: host pointers in and out, synchronous, device selected by id.
// The reference prints CUDA errors and carries on (CUDA_CHECK :18-25); here an error leaves
// *num_out = 0 and the message in lsfa_last_error().  If the caller was on another device it stays
// switched to `device_id`, like after the reference's cudaSetDevice (:106-111).
extern "C" void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                     float nms_overlap_thresh, int device_id) {
  *num_out = 0;
  if (boxes_num <= 0) return;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) { set_error("_nms: hipGetDevice failed"); return; }
  if (cur != device_id && hipSetDevice(device_id) != hipSuccess) { set_error("_nms: hipSetDevice(%d) failed", device_id); return; }
  if (device_id < 0 || device_id >= 64) { set_error("_nms: device_id %d out of range", device_id); return; }
  const size_t ws_bytes = lsfa_nms_workspace_bytes(boxes_num);
  const size_t bbytes = align_up((size_t)boxes_num * boxes_dim * sizeof(float), 256);
  const size_t kbytes = align_up(sizeof(int) * ((size_t)boxes_num + 1), 256);
  std::lock_guard<std::mutex> lk(g_host_nms_mu);
  HostNmsScratch& sc = g_host_nms[device_id];
  hipError_t e = hipSuccess;
  if (sc.cap < bbytes + ws_bytes + kbytes) {
    if (sc.p) (void)hipFree(sc.p);
    sc.p = nullptr; sc.cap = 0;
    const size_t want = (bbytes + ws_bytes + kbytes) * 2;      // headroom: the next, slightly larger call reuses it
    e = hipMalloc(&sc.p, want);
    if (e == hipSuccess) sc.cap = want;
  }
  if (e == hipSuccess) {
    float* boxes_dev = (float*)sc.p;
    void* ws = (unsigned char*)sc.p + bbytes;
    int* keep_dev = (int*)((unsigned char*)sc.p + bbytes + ws_bytes);
    e = hipMemcpy(boxes_dev, boxes_host, (size_t)boxes_num * boxes_dim * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      int rc = lsfa_nms_sorted(boxes_dev, boxes_num, boxes_dim, nms_overlap_thresh, keep_dev, keep_dev + boxes_num, ws,
                               ws_bytes, nullptr);
      if (rc == LSFA_OK) {
        std::vector<int> host((size_t)boxes_num + 1);
        e = hipMemcpy(host.data(), keep_dev, sizeof(int) * host.size(), hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
          const int k = host[boxes_num];
          for (int i = 0; i < k; ++i) keep_out[i] = host[i];
          *num_out = k;
        }
      }
    }
  }
  if (e != hipSuccess) hip_fail(e, "_nms");
}
