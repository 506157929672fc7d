// A/B harness for the warp kernel variants (tools/lab/warp_lab.py builds and drives it).  Not part of
// liblsfa_hip.so.  variant: 0 = round-1 kernel; 1/2/3 = current kernel with 8/4/16 channels per wave;
// 4/5 = 8/4 channels per wave with tap sharing between the two pixels of a lane.
#include "warp_kernels.h"
#include "warp_r1_kernel.h"

extern "C" int lab_warp(int variant, const float* feat, int feat_n, const float* flow, int N, int C, int H, int W,
                        const float* mul, const float* add, const float* res, int res_c, const float* res_w,
                        const float* res_b, float* out, void* stream) {
  using namespace lsfa;
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  if (HW % 2 != 0) return -1;
  const warp::Args a = {feat, feat_n, flow, N, C, H, W, mul, add, res, res_c, res_w, res_b, out};
  switch (variant) {
    case 0: {
      const int gx = ceil_div(HW, 256 * 2);
      dim3 grid(gx, ceil_div(C, 8), N);
      if (mul && !add && !res)
        hipLaunchKernelGGL((warp_r1::warp_kernel<2, true, false, false>), grid, dim3(256), 0, s, feat, feat_n, flow, C, H, W,
                           mul, add, res, res_c, res_w, res_b, out, 8);
      else if (!mul && add && res)
        hipLaunchKernelGGL((warp_r1::warp_kernel<2, false, true, true>), grid, dim3(256), 0, s, feat, feat_n, flow, C, H, W,
                           mul, add, res, res_c, res_w, res_b, out, 8);
      else if (!mul && !add && !res)
        hipLaunchKernelGGL((warp_r1::warp_kernel<2, false, false, false>), grid, dim3(256), 0, s, feat, feat_n, flow, C, H, W,
                           mul, add, res, res_c, res_w, res_b, out, 8);
      else return -2;
      break;
    }
    case 1: warp::launch<2, 8, false>(s, a); break;
    case 2: warp::launch<2, 4, false>(s, a); break;
    case 3: warp::launch<2, 16, false>(s, a); break;
    case 4: warp::launch<2, 8, true>(s, a); break;
    case 5: warp::launch<2, 4, true>(s, a); break;
    default: return -3;
  }
  return (int)hipGetLastError();
}
