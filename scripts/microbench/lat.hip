// Dependent-issue latency of the ops on the rgb chain's critical path, one lone wavefront (gfx950).
// build+run: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/lat lat.hip && /tmp/lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#define N 200000
#define REP 16
template <int OP> __global__ void k(double *out, double a, double b, float fa, int nw)
{
    double x = a + threadIdx.x * 1e-9, y = b;
    float f = fa;
    uint32_t u = threadIdx.x + 3;
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) x = fma(x, a, b);
            if (OP == 1) x = x + b;
            if (OP == 2) x = x * a;
            if (OP == 3) { f = (float)((double)f); asm volatile("" : "+v"(f)); f = (float)((double)f * 1.0000001); }   // cvt, mul, cvt
            if (OP == 4) { u = (uint32_t)((double)u * 0.999999); }      // cvt_f64_u32, mul_f64, cvt_u32_f64
            if (OP == 5) x = __builtin_amdgcn_rcp(x);
            if (OP == 6) f = f * fa;
            if (OP == 7) { u = (uint32_t)((float)u * 1.0000001f); }     // cvt_f32_u32, mul_f32, cvt_u32_f32
            if (OP == 8) u = (uint32_t)__builtin_amdgcn_mov_dpp((int)u, 0x55, 0xf, 0xf, false) + 1u;
            if (OP == 9) x = x / y + 1.0;                               // full divide + add
            if (OP == 10) { float g = (float)x; x = (double)g + b; }   // w chain: cvt_f32_f64, cvt_f64_f32, add
            if (OP == 11) x = trunc(x * a) + b;
            if (OP == 12) x = (double)(float)x;                         // cvt pair only
            if (OP == 13) x = (double)(uint32_t)x;                      // cvt_u32_f64 + cvt_f64_u32
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + f + u + y;
}
template <int OP> void run(const char *name, int nops, int nblk = 1)
{
    double *d; hipMalloc(&d, 8 * 64 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(64), 0, 0, d, 1.0000001, 1e-7, 1.0000001f, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(64), 0, 0, d, 1.0000001, 1e-7, 1.0000001f, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f ns/iter  (%d ops) -> %6.2f ns/op\n", name, ms * 1e6 / ((double)N * REP), nops, ms * 1e6 / ((double)N * REP * nops));
    hipFree(d);
}
int main()
{
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk);
    run<0>("fma_f64", 1); run<1>("add_f64", 1); run<2>("mul_f64", 1);
    run<3>("cvt_f64_f32,cvt_f32_f64,cvt,mul_f64,cvt", 5); run<12>("cvt_f32_f64,cvt_f64_f32", 2); run<13>("cvt_u32_f64,cvt_f64_u32", 2);
    run<4>("cvt_f64_u32,mul_f64,cvt_u32_f64", 3); run<5>("rcp_f64", 1); run<6>("mul_f32", 1);
    run<7>("cvt_f32_u32,mul_f32,cvt_u32_f32", 3); run<8>("dpp mov + add", 2); run<9>("div_f64 + add", 1);
    run<10>("w chain (cvt,cvt,add)", 3); run<11>("mul,trunc,add f64", 3);
    return 0;
}
