// conv_split_kernel built with -DLSFA_CS_STAMPS: where one chunk's cycles go (tools/lab/conv_split_lab.py --stamps).
#include "conv_split_kernel.h"

using namespace lsfa::convsplit;

extern "C" int conv_split_lab_run(const float* x, const void* wfrag, float* part, int N, int H, int W, int Cin, int Cout, int k,
                                  int slices, long long* host_stamps8) {
  const int chunk_total = k * k * (Cin / 32);
  Args a = {};
  a.x = x; a.wfrag = (const uint4*)wfrag; a.y = part; a.part = part;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.kh = a.kw = k; a.stride = 1; a.pad_h = a.pad_w = k / 2; a.dil = 1;
  a.Ho = H; a.Wo = W; a.chunks_per_slice = (chunk_total + slices - 1) / slices; a.lda = Cin; a.ldy = Cout;
  a.out_H = H; a.out_W = W; a.out_sy = a.out_sx = 1;
  const int P = N * H * W;
  const int nx = (P + kWgPix - 1) / kWgPix, ny = Cout / kWgCh;
  const int tiles = nx * ny * slices;
  hipLaunchKernelGGL(conv_split_kernel, dim3(8 * ((tiles + 7) / 8)), dim3(kThreads), 0, 0, a, nx, ny, slices);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
#ifdef LSFA_CS_STAMPS
  if (hipMemcpyFromSymbol(host_stamps8, HIP_SYMBOL(g_cs_stamps), sizeof(long long) * 8) != hipSuccess) return 3;
#endif
  return 0;
}
