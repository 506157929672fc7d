// Lab: what a dependency between two phases of work costs on gfx950, two ways:
//   (a) a kernel boundary inside a hipGraph (N dependent launches of a kernel that touches one cache line per workgroup),
//   (b) a grid barrier inside ONE persistent kernel (atomic arrive + spin, agent-scope release / acquire so that data written
//       before the barrier by any XCD is visible after it to every XCD).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/lab/grid_barrier_lab.hip -o /tmp/gbl && /tmp/gbl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void phase_kernel(float* data, int phase) {
    // each workgroup reads what its neighbour wrote in the previous phase and writes its own line
    const int g = blockIdx.x, n = gridDim.x;
    if (threadIdx.x == 0) data[g * 32] = data[((g + 1) % n) * 32 + 0] * 0.5f + (float)phase;
}

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned spin_limit) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit) { ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ void persistent_kernel(float* data, unsigned* counter, int phases, int* bad) {
    const int g = blockIdx.x, n = gridDim.x;
    for (int p = 0; p < phases; ++p) {
        if (threadIdx.x == 0) {
            float v = __hip_atomic_load(&data[((g + 1) % n) * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            data[g * 32] = v * 0.5f + (float)p;
        }
        if (!grid_barrier(counter, (unsigned)(n * (p + 1)), 1u << 22)) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }
    }
}

// the same with the data movement of a real phase: every workgroup writes `bytes` and reads the neighbour's
__global__ void persistent_payload_kernel(float4* buf, unsigned* counter, int phases, int per_wg_vec, int* bad) {
    const int g = blockIdx.x, n = gridDim.x;
    float4 acc = {0, 0, 0, 0};
    for (int p = 0; p < phases; ++p) {
        const float4* src = buf + (size_t)(((g + 1 + p) % n)) * per_wg_vec + (size_t)(p & 1) * n * per_wg_vec;
        float4* dst = buf + (size_t)g * per_wg_vec + (size_t)((p + 1) & 1) * n * per_wg_vec;
        for (int i = threadIdx.x; i < per_wg_vec; i += blockDim.x) {
            float4 v = src[i];
            acc.x += v.x; v.x = acc.x * 0.5f + 1.0f;
            dst[i] = v;
        }
        if (!grid_barrier(counter, (unsigned)(n * (p + 1)), 1u << 22)) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }
    }
}
__global__ void payload_kernel(float4* buf, int p, int per_wg_vec) {
    const int g = blockIdx.x, n = gridDim.x;
    const float4* src = buf + (size_t)(((g + 1 + p) % n)) * per_wg_vec + (size_t)(p & 1) * n * per_wg_vec;
    float4* dst = buf + (size_t)g * per_wg_vec + (size_t)((p + 1) & 1) * n * per_wg_vec;
    for (int i = threadIdx.x; i < per_wg_vec; i += blockDim.x) {
        float4 v = src[i];
        v.x = v.x * 0.5f + 1.0f;
        dst[i] = v;
    }
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int phases = 64;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int wgs : {64, 256, 512}) {
        float* data; unsigned* counter; int* bad;
        CK(hipMalloc(&data, 1024 * 32 * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4));
        CK(hipMemset(data, 0, 1024 * 32 * 4)); CK(hipMemset(bad, 0, 4));
        // (a) graph of dependent launches
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int p = 0; p < phases; ++p) phase_kernel<<<wgs, 256, 0, s>>>(data, p);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("wgs %4d  kernel boundary in a graph: %7.2f us per phase\n", wgs, ms * 1e3 / (10 * phases));
        // (b) one persistent kernel with grid barriers
        float best = 1e9;
        for (int i = 0; i < 12; ++i) {
            CK(hipMemsetAsync(counter, 0, 4, s));
            CK(hipEventRecord(a, s));
            persistent_kernel<<<wgs, 256, 0, s>>>(data, counter, phases, bad);
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, a, b));
            if (i >= 2 && ms < best) best = ms;
        }
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("wgs %4d  grid barrier in one kernel:     %7.2f us per phase (whole kernel %.1f us, timed-out workgroups %d)\n", wgs, best * 1e3 / phases, best * 1e3, hb);
        // payload variants: 32 KB and 128 KB per workgroup per phase
        for (int kb : {32, 128}) {
            const int per = kb * 1024 / 16;
            float4* buf; CK(hipMalloc(&buf, (size_t)2 * wgs * per * 16)); CK(hipMemset(buf, 0, (size_t)2 * wgs * per * 16));
            hipGraph_t g2; hipGraphExec_t ge2;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int p = 0; p < phases; ++p) payload_kernel<<<wgs, 256, 0, s>>>(buf, p, per);
            CK(hipStreamEndCapture(s, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge2, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(a, s));
            for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge2, s));
            CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, a, b));
            printf("wgs %4d  %3d KB/wg  kernel boundary: %7.2f us per phase\n", wgs, kb, ms * 1e3 / (10 * phases));
            best = 1e9;
            for (int i = 0; i < 8; ++i) {
                CK(hipMemsetAsync(counter, 0, 4, s));
                CK(hipEventRecord(a, s));
                persistent_payload_kernel<<<wgs, 256, 0, s>>>(buf, counter, phases, per, bad);
                CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
                CK(hipEventElapsedTime(&ms, a, b));
                if (i >= 2 && ms < best) best = ms;
            }
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("wgs %4d  %3d KB/wg  grid barrier:    %7.2f us per phase (timed-out %d)\n", wgs, kb, best * 1e3 / phases, hb);
            CK(hipFree(buf)); CK(hipGraphExecDestroy(ge2)); CK(hipGraphDestroy(g2));
        }
        CK(hipFree(data)); CK(hipFree(counter)); CK(hipFree(bad)); CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
