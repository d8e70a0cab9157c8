// A spinning kernel: `n` single-wavefront workgroups of `vgprs`-class register footprint busy-wait for `us` microseconds.
// Used to measure how library GEMMs react to a few CUs being held by another stream (scripts/gemm_vs_hog.py).
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC scripts/microbench/hog.hip -o scripts/microbench/libhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_hog(long long cycles, float *sink)
{
    const long long t0 = wall_clock64();
    float a = threadIdx.x;
    while (wall_clock64() - t0 < cycles) a = a * 1.0001f + 0.5f;
    if (a == 12345.678f) sink[0] = a;
}
extern "C" int hog_launch(void *stream, int n_wg, int threads, double us, float *sink)
{
    // wall_clock64 ticks at 100 MHz on gfx9
    hipLaunchKernelGGL(k_hog, dim3(n_wg), dim3(threads), 0, (hipStream_t)stream, (long long)(us * 100.0), sink);
    return (int)hipGetLastError();
}
