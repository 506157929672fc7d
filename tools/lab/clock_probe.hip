// Shader clock while other kernels run: one wave spins for `spin_ref` ticks of the constant 100 MHz counter (s_memrealtime) and
// reports how many shader cycles (s_memtime) passed meanwhile.  tools/lab/clock_probe.py runs it beside the convolution kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_probe_kernel(long long spin_ref, long long* out) {
  const long long r0 = wall_clock64(), c0 = clock64();
  long long r1 = r0;
  while (r1 - r0 < spin_ref) r1 = wall_clock64();
  const long long c1 = clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

extern "C" int clock_probe(long long spin_ref, long long* out_dev, void* stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, spin_ref, out_dev);
  return (int)hipGetLastError();
}
