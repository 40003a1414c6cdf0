// Experiment: what a grid-wide barrier costs inside one persistent launch on MI355X (release / acquire at agent scope
// around an atomic counter), against the ~8.6 us a dependent launch costs.  extern "C" gbar_run(wgs, threads, rounds,
// touch_kb) -> microseconds per barrier; touch_kb: every workgroup writes that many KB before each barrier and reads a
// neighbour's after it (so that the fences have dirty lines to write back).
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: this workgroup's stores reach the memory side
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

__global__ void k_gbar(unsigned *ctr, float *buf, int rounds, int touch_floats, float *sink)
{
    const int wg = blockIdx.x, nwg = gridDim.x;
    float acc = 0.0f;
    for (int r = 0; r < rounds; r++) {
        float *mine = buf + (size_t)wg * touch_floats;
        for (int i = threadIdx.x; i < touch_floats; i += blockDim.x) mine[i] = (float)(r + i);
        grid_barrier(ctr, (unsigned)(r + 1) * nwg);
        const float *other = buf + (size_t)((wg + 37) % nwg) * touch_floats;
        for (int i = threadIdx.x; i < touch_floats; i += blockDim.x)
            acc += __builtin_nontemporal_load(other + i);
    }
    if (acc == -1.0f) *sink = acc;
}

extern "C" double gbar_run(int wgs, int threads, int rounds, int touch_kb)
{
    unsigned *ctr; float *buf, *sink;
    const int tf = touch_kb * 256;
    hipMalloc((void **)&ctr, 4); hipMalloc((void **)&sink, 4);
    hipMalloc((void **)&buf, (size_t)wgs * (tf ? tf : 1) * 4);
    hipMemset(ctr, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gbar, dim3(wgs), dim3(threads), 0, 0, ctr, buf, 10, tf, sink);   // warm
    hipDeviceSynchronize();
    hipMemset(ctr, 0, 4);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_gbar, dim3(wgs), dim3(threads), 0, 0, ctr, buf, rounds, tf, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipFree(ctr); hipFree(buf); hipFree(sink);
    return 1e3 * ms / rounds;
}

__global__ void k_small(float *buf, int touch_floats, int r)
{
    float *mine = buf + (size_t)blockIdx.x * touch_floats;
    for (int i = threadIdx.x; i < touch_floats; i += blockDim.x) mine[i] = mine[i] + (float)r;
}
// the same work as dependent launches
extern "C" double launch_run(int wgs, int threads, int rounds, int touch_kb)
{
    float *buf;
    const int tf = touch_kb * 256;
    hipMalloc((void **)&buf, (size_t)wgs * (tf ? tf : 1) * 4);
    hipMemset(buf, 0, (size_t)wgs * (tf ? tf : 1) * 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_small, dim3(wgs), dim3(threads), 0, s, buf, tf, r);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_small, dim3(wgs), dim3(threads), 0, s, buf, tf, r);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipFree(buf); hipStreamDestroy(s);
    return 1e3 * ms / rounds;
}
