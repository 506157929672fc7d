// A/B harness for warp kernel variants (tools/lab/warp_lab.py builds and drives it).  Not part of liblsfa_hip.so.
#include "warp_r1_kernel.h"
#include "warp_variants.h"
#include "warp_lds.h"
#include "warp_r2_attempt.h"

extern "C" int lab_warp(int variant, const float* feat, int feat_n, const float* flow, int N, int C, int H, int W,
                        const float* mul, const float* add, const float* res, int res_c, const float* res_w,
                        const float* res_b, float* out, void* stream) {
  using lsfa::ceil_div;
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  if (HW % 2 != 0 || (res && res_c != 3) || C % 16 != 0) return -1;
  const warp_r2::Args a = {feat, feat_n, flow, N, C, H, W, mul, add, res, res_c, res_w, res_b, out};
  const warp_lab::Args g = {feat, feat_n, flow, N, C, H, W, mul, add, res, res_w, res_b, out};
  int rc = 0;
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
      else return -2;
      break;
    }
    case 1: warp_r2::launch<2, 8, false, 1>(s, a); break;     // round-2 rewrite, __launch_bounds__(256, 1) as first shipped
    case 15: warp_r2::launch<2, 8, false, 0>(s, a); break;    // the same with plain __launch_bounds__(256)
    case 16: warp_r2::launch<2, 8, true, 0>(s, a); break;     // + taps shared between a lane's two pixels
    //                         RUN U  FLAT   HOIST  DWORD  THREADS
    case 2: rc = warp_lab::launch<8, 4, false, false, false, 256>(s, g); break;       // = r1's structure
    case 3: rc = warp_lab::launch<8, 4, true, false, false, 256>(s, g); break;        // + wave-granular items
    case 4: rc = warp_lab::launch<8, 8, true, false, false, 256>(s, g); break;        // + one batch of 8
    case 5: rc = warp_lab::launch<8, 2, true, false, false, 256>(s, g); break;        // batches of 2
    case 6: rc = warp_lab::launch<8, 8, true, true, false, 256>(s, g); break;         // one batch, operands hoisted
    case 7: rc = warp_lab::launch<8, 8, false, false, false, 256>(s, g); break;       // tiled grid, one batch
    case 8: rc = warp_lab::launch<4, 4, true, false, false, 256>(s, g); break;        // 4 ch / wave
    case 9: rc = warp_lab::launch<16, 4, true, false, false, 256>(s, g); break;       // 16 ch / wave
    case 10: rc = warp_lab::launch<8, 4, true, false, true, 256>(s, g); break;        // taps as 4-byte loads
    case 11: rc = warp_lab::launch<8, 4, true, false, false, 64>(s, g); break;        // one wave per workgroup
    case 12: rc = warp_lab::launch<8, 4, true, false, false, 1024>(s, g); break;      // 16 waves per workgroup
    case 13: rc = warp_lab::launch<16, 8, true, false, false, 256>(s, g); break;      // 16 ch / wave, batches of 8
    case 14: rc = warp_lab::launch<4, 2, true, false, false, 256>(s, g); break;       // 4 ch / wave, batches of 2
    //                        THREADS NPAIR NDMA STAGES          channels per workgroup
    case 17: rc = warp_lds::launch<640, 2, 1, 3>(s, g, 8); break;
    case 18: rc = warp_lds::launch<640, 2, 1, 3>(s, g, 16); break;
    case 19: rc = warp_lds::launch<640, 2, 1, 4>(s, g, 16); break;
    case 20: rc = warp_lds::launch<320, 4, 2, 3>(s, g, 16); break;
    case 21: rc = warp_lds::launch<320, 4, 2, 4>(s, g, 16); break;
    case 22: rc = warp_lds::launch<256, 5, 3, 3>(s, g, 16); break;
    case 23: rc = warp_lds::launch<640, 2, 1, 3>(s, g, 4); break;
    case 24: rc = warp_lds::launch<640, 2, 1, 3>(s, g, 32); break;
    case 25: rc = warp_lds::launch<640, 2, 1, 3>(s, g, 2); break;
    case 26: rc = warp_lds::launch<640, 2, 1, 4>(s, g, 8); break;
    case 27: rc = warp_lds::launch<256, 5, 3, 3>(s, g, 8); break;
    case 28: rc = warp_lds::launch<256, 5, 3, 3>(s, g, 4); break;
    case 29: rc = warp_lds::launch<1024, 2, 1, 3>(s, g, 8); break;
    case 30: rc = warp_lds::launch<1024, 2, 1, 3>(s, g, 4); break;
    case 31: rc = warp_lds::launch<512, 3, 2, 3>(s, g, 8); break;
    case 32: rc = warp_lds::launch<512, 3, 2, 3>(s, g, 4); break;
    case 99: {     // not a warp: the same three streams (two in, one out) moved with 16-byte accesses
      const size_t n4 = (size_t)N * C * HW / 4;
      const float* other = mul ? mul : add;
      if (mul) hipLaunchKernelGGL(warp_lds::stream_floor_kernel<true>, dim3(256 * 16), dim3(256), 0, s, (const float4*)feat, (const float4*)other, (float4*)out, n4);
      else hipLaunchKernelGGL(warp_lds::stream_floor_kernel<false>, dim3(256 * 16), dim3(256), 0, s, (const float4*)feat, (const float4*)other, (float4*)out, n4);
      break;
    }
    default: return -3;
  }
  if (rc) return rc;
  return (int)hipGetLastError();
}
