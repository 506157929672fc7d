// Compressed-domain motion vectors on the device: accumulation of per-frame macroblock vectors back
// to the key frame, the accumulated motion-vector field and the residual image
// (external/data_loader_py2/coviar_data_loader.c:71-177, create_and_load_mv_residual, accumulate = 1).
//
// The reference walks the decoder's block list serially and lets later blocks overwrite earlier ones.
// Every write reads the PREVIOUS frame's map (accu_src_old), so a frame is one gather once it is known,
// per pixel, WHICH block wrote it last:
//   mv_owner_kernel    one thread per (block, pixel of the block): atomicMax of the block index into an
//                      owner map (-1 = untouched) — "last writer wins" without the serial walk;
//   mv_gather_kernel   one thread per pixel: new[p] = owner >= 0 ? old[p - dst + src] : old[p].  The
//                      output is complete, so old/new ping-pong and the reference's per-frame memcpy
//                      (:123-125) disappears.
// Pure int32 index work, HBM/L2-bound, bit-exact with orc_coviar_* by construction.  Maps are (H, W, 2)
// row-major (the reference's are x-major; only its index arithmetic differs).
#include "common.h"

using namespace lsfa;

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void mv_fill_kernel(int* __restrict__ p, int n, int v) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ __launch_bounds__(kThreads) void mv_identity_kernel(int* __restrict__ accu, int width, int height) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= width * height) return;
  const int y = i / width, x = i - y * width;
  reinterpret_cast<int2*>(accu)[i] = make_int2(x, y);
}

// mvs (n, 7) int32 = {source, w, h, src_x, src_y, dst_x, dst_y}; grid (n, ceil(max_area / 256))
__global__ __launch_bounds__(kThreads) void mv_owner_kernel(const int* __restrict__ mvs, int n, int width, int height,
                                                            int* __restrict__ owner) {
  const int i = blockIdx.x;
  const int* mv = mvs + (size_t)i * 7;
  const int w = mv[1], h = mv[2], src_x = mv[3], src_y = mv[4], dst_x = mv[5], dst_y = mv[6];
  if (dst_x - src_x == 0 && dst_y - src_y == 0) return;
  // the reference's loops: x_start in [(-1 * w) / 2, w / 2), C division truncating toward zero
  const int x_lo = (-1 * w) / 2, x_hi = w / 2, y_lo = (-1 * h) / 2, y_hi = h / 2;
  const int bw = x_hi - x_lo, bh = y_hi - y_lo;
  if (bw <= 0 || bh <= 0) return;
  const int t = blockIdx.y * kThreads + threadIdx.x;
  if (t >= bw * bh) return;
  const int xs = x_lo + t % bw, ys = y_lo + t / bw;
  const int pdx = dst_x + xs, pdy = dst_y + ys, psx = src_x + xs, psy = src_y + ys;
  if (pdy >= 0 && pdy < height && pdx >= 0 && pdx < width && psy >= 0 && psy < height && psx >= 0 && psx < width)
    atomicMax(&owner[pdy * width + pdx], i);
}

__global__ __launch_bounds__(kThreads) void mv_gather_kernel(const int* __restrict__ mvs, const int* __restrict__ owner,
                                                             const int* __restrict__ accu_old, int* __restrict__ accu_new,
                                                             int width, int height) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= width * height) return;
  const int o = owner[p];
  int src = p;
  if (o >= 0) {
    const int* mv = mvs + (size_t)o * 7;
    const int y = p / width, x = p - y * width;
    src = (y - mv[6] + mv[4]) * width + (x - mv[5] + mv[3]);
  }
  reinterpret_cast<int2*>(accu_new)[p] = reinterpret_cast<const int2*>(accu_old)[src];
}

__global__ __launch_bounds__(kThreads) void mv_field_kernel(const int* __restrict__ accu, int width, int height,
                                                            int* __restrict__ mv) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= width * height) return;
  const int y = p / width, x = p - y * width;
  const int2 a = reinterpret_cast<const int2*>(accu)[p];
  reinterpret_cast<int2*>(mv)[p] = make_int2(x - a.x, y - a.y);
}

__global__ __launch_bounds__(kThreads) void mv_residual_kernel(const unsigned char* __restrict__ cur,
                                                               const unsigned char* __restrict__ ref,
                                                               const int* __restrict__ accu, int width, int height,
                                                               int* __restrict__ res) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= width * height) return;
  const int2 a = reinterpret_cast<const int2*>(accu)[p];
  const size_t s = ((size_t)a.y * width + a.x) * 3, o = (size_t)p * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) res[o + c] = (int)cur[o + c] - (int)ref[s + c];
}

}  // namespace

extern "C" size_t lsfa_mv_workspace_bytes(int width, int height) {
  if (width <= 0 || height <= 0) return 0;
  return align_up((size_t)width * height * sizeof(int), 256);
}

extern "C" int lsfa_mv_identity(int* accu, int width, int height, void* stream) {
  LSFA_REQUIRE(accu && width > 0 && height > 0, "lsfa_mv_identity: bad argument");
  hipLaunchKernelGGL(mv_identity_kernel, dim3(ceil_div(width * height, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     accu, width, height);
  LSFA_LAUNCH_CHECK("lsfa_mv_identity");
  return LSFA_OK;
}

extern "C" int lsfa_mv_accumulate(const int* mvs, int n_mvs, int max_block_area, const int* accu_old, int* accu_new,
                                  int width, int height, void* ws, size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(accu_old && accu_new && accu_old != accu_new, "lsfa_mv_accumulate: accu_old/accu_new must be two buffers");
  LSFA_REQUIRE(width > 0 && height > 0 && n_mvs >= 0 && max_block_area >= 0, "lsfa_mv_accumulate: bad shape");
  LSFA_REQUIRE(n_mvs == 0 || mvs, "lsfa_mv_accumulate: mvs is NULL");
  LSFA_REQUIRE((long)width * height < (1L << 30), "lsfa_mv_accumulate: frame too large");
  if (ws_bytes < lsfa_mv_workspace_bytes(width, height) || !ws) {
    set_error("lsfa_mv_accumulate: workspace %zu < %zu bytes", ws_bytes, lsfa_mv_workspace_bytes(width, height));
    return LSFA_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  int* owner = (int*)ws;
  const int px = width * height;
  hipLaunchKernelGGL(mv_fill_kernel, dim3(ceil_div(px, kThreads)), dim3(kThreads), 0, s, owner, px, -1);
  if (n_mvs > 0 && max_block_area > 0) {
    hipLaunchKernelGGL(mv_owner_kernel, dim3(n_mvs, ceil_div(max_block_area, kThreads)), dim3(kThreads), 0, s, mvs, n_mvs,
                       width, height, owner);
  }
  hipLaunchKernelGGL(mv_gather_kernel, dim3(ceil_div(px, kThreads)), dim3(kThreads), 0, s, mvs, (const int*)owner, accu_old,
                     accu_new, width, height);
  LSFA_LAUNCH_CHECK("lsfa_mv_accumulate");
  return LSFA_OK;
}

extern "C" int lsfa_mv_field(const int* accu, int width, int height, int* mv, void* stream) {
  LSFA_REQUIRE(accu && mv && width > 0 && height > 0, "lsfa_mv_field: bad argument");
  hipLaunchKernelGGL(mv_field_kernel, dim3(ceil_div(width * height, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, accu,
                     width, height, mv);
  LSFA_LAUNCH_CHECK("lsfa_mv_field");
  return LSFA_OK;
}

extern "C" int lsfa_mv_residual(const unsigned char* bgr_cur, const unsigned char* bgr_ref, const int* accu, int width,
                                int height, int* res, void* stream) {
  LSFA_REQUIRE(bgr_cur && bgr_ref && accu && res && width > 0 && height > 0, "lsfa_mv_residual: bad argument");
  hipLaunchKernelGGL(mv_residual_kernel, dim3(ceil_div(width * height, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     bgr_cur, bgr_ref, accu, width, height, res);
  LSFA_LAUNCH_CHECK("lsfa_mv_residual");
  return LSFA_OK;
}
