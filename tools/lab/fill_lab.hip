// Fill-rate lab (r5): how fast can ONE CU take bytes in, by LDS-DMA (global_load_lds_dwordx4) or by register loads
// (global_load_dwordx4), from L2-resident and from Infinity-Cache-resident sources, as a function of waves per CU and of
// loads in flight per wave?  The ring kernel's chunk time (0.45-0.9 us per 32 KB at 128 x 128 tiles) equals 35-70 GB/s per CU;
// this says whether that is a port limit (then only more FLOPs per byte help) or a latency x in-flight product (then a deeper ring helps).
//   hipcc --offload-arch=gfx950 -O3 tools/lab/fill_lab.hip -o tools/lab/_build/fill_lab && tools/lab/_build/fill_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each wave: `iters` rounds of DEPTH loads of 1 KB (16 B per lane); wave w of workgroup b reads piece (i * DEPTH + d) of its own
// stream: src + ((b * waves + w) * stride_wave + (i * DEPTH + d) * 1024) % span   (span = bytes the whole grid cycles over)
template <int DEPTH, bool DMA>
__global__ __launch_bounds__(1024) void fill_kernel(const char* __restrict__ src, size_t span, size_t wg_stride, size_t wg_span, int iters, float* sink) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  // wg_span != 0: the workgroup cycles over its own wg_span bytes at blockIdx.x * wg_stride (L1-resident when small); else over the whole span
  const size_t base = wg_span ? ((size_t)blockIdx.x * wg_stride) % span : 0;
  if (wg_span) span = wg_span;
  src += base;
  size_t off = ((wg_span ? 0 : (size_t)blockIdx.x * wg_stride) + (size_t)wave * 1024) % span;
  const size_t step = (size_t)waves * 1024;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint4* dst = lds + (size_t)wave * DEPTH * 64;
  for (int i = 0; i < iters; ++i) {
    if (DMA) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(src + off) + lane, dst + d * 64, 16, 0, 0);
        off += step;
        if (off >= span) off -= span;
      }
      wait_vmcnt<DEPTH / 2>();      // half the window stays in flight across rounds
    } else {
      uint4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        v[d] = (reinterpret_cast<const uint4*>(src + off))[lane];
        off += step;
        if (off >= span) off -= span;
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc.x ^= v[d].x;
    }
  }
  wait_vmcnt<0>();
  __syncthreads();
  if (sink && lane == 0 && blockIdx.x == 0x7fffffff) sink[0] = (float)(acc.x + lds[lane].x);
}

template <int DEPTH, bool DMA>
float run(const char* src, size_t span, size_t wg_stride, size_t wg_span, int waves, int iters, int grid, hipStream_t s) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t lds = (size_t)waves * DEPTH * 1024;
  auto k = fill_kernel<DEPTH, DMA>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(k, dim3(grid), dim3(waves * 64), lds, s, src, span, wg_stride, wg_span, iters, (float*)nullptr);
  CK(hipEventRecord(a, s));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(waves * 64), lds, s, src, span, wg_stride, wg_span, iters, (float*)nullptr);
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / 3;
}

int main() {
  const size_t big = 1ull << 30;
  char* src;
  CK(hipMalloc(&src, big));
  CK(hipMemset(src, 1, big));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  struct Src { const char* name; size_t span; size_t wg_stride; size_t wg_span; } srcs[] = {
      {"L1: every workgroup re-reads the same 16 KB", 16u << 10, 0, 0},
      {"L1: every workgroup re-reads its OWN 16 KB (256 KB apart)", 64u << 20, 256u << 10, 16u << 10},
      {"L1 overflow: every workgroup re-reads its OWN 64 KB (256 KB apart)", 64u << 20, 256u << 10, 64u << 10},
      {"L2: every workgroup reads the same 1 MB", 1u << 20, 0, 0},
      {"L2/MALL: 8 MB shared, workgroups 32 KB apart", 8u << 20, 32u << 10, 0},
      {"MALL: 128 MB, workgroups 512 KB apart", 128u << 20, 512u << 10, 0},
      {"HBM: 1 GB, workgroups 4 MB apart", big, 4u << 20, 0},
  };
  const int grid = 256;
  printf("# one workgroup per CU (grid 256); GB/s per CU and TB/s chip-wide; loads of 1 KB per wave-instruction\n");
  for (auto& S : srcs) {
    printf("\n## %s\n", S.name);
    for (int dma = 1; dma >= 0; --dma) {
      for (int waves : {4, 8, 16}) {
        printf("   %s waves %2d:", dma ? "lds-dma " : "register", waves);
        for (int depth : {4, 8, 16}) {
          if ((size_t)waves * depth * 1024 > 160 * 1024) { printf("   d%-2d    -   ", depth); continue; }
          const int iters = 4096 / depth;      // 4 MB per wave
          float ms;
          if (dma) ms = depth == 4 ? run<4, true>(src, S.span, S.wg_stride, S.wg_span, waves, iters, grid, s) : depth == 8 ? run<8, true>(src, S.span, S.wg_stride, S.wg_span, waves, iters, grid, s) : run<16, true>(src, S.span, S.wg_stride, S.wg_span, waves, iters, grid, s);
          else ms = depth == 4 ? run<4, false>(src, S.span, S.wg_stride, S.wg_span, waves, iters, grid, s) : depth == 8 ? run<8, false>(src, S.span, S.wg_stride, S.wg_span, waves, iters, grid, s) : run<16, false>(src, S.span, S.wg_stride, S.wg_span, waves, iters, grid, s);
          const double bytes_cu = (double)waves * iters * depth * 1024;
          printf("   d%-2d %5.1f GB/s/CU (%4.1f TB/s)", depth, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * grid / (ms * 1e-3) / 1e12);
        }
        printf("\n");
      }
    }
  }
  return 0;
}
