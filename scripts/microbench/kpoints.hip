// Ablation microbenchmark of k_points (fast geometry): which part of the per-point work costs the time?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I bsc-nav_amd/csrc scripts/microbench/kpoints.hip -o scripts/microbench/kpoints.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <limits.h>
#include "geometry_dev.h"
struct PointRec { uint32_t alo, ahi, rgbv; };
#define TPB 256
// ABL bits: 1 no rgb gather, 2 no occ read/atomic, 4 no rec store, 8 T from SGPR (readfirstlane), 16 no alpha (exp),
//           32 no cell/patf stores, 64 no geometry at all (stream only)
template <int ABL>
__global__ __launch_bounds__(TPB) void kp(GeomConst gc, const float *__restrict__ depth, const uint8_t *__restrict__ rgb,
                                          int rgb_ch, const double *__restrict__ transforms, int64_t P, float inv_w,
                                          int32_t *occ, int32_t *__restrict__ p_cell, uint32_t *__restrict__ p_patf,
                                          PointRec *__restrict__ p_rec)
{
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= P) return;
    const int32_t N = gc.H * gc.W;
    const uint32_t j0 = blockIdx.x * (uint32_t)TPB;
    int f = (int)(j0 / (uint32_t)N);
    int32_t i = (int32_t)(j0 - (uint32_t)f * (uint32_t)N) + (int32_t)threadIdx.x;
    if (i >= N) { i -= N; ++f; }
    if (ABL & 8) f = __builtin_amdgcn_readfirstlane(f);
    const float z = depth[(int64_t)f * N + i];
    int32_t y = (int32_t)((float)i * inv_w);
    int32_t x = i - y * gc.W;
    if (x < 0) { --y; x += gc.W; } else if (x >= gc.W) { ++y; x -= gc.W; }
    GeomFastOut o;
    if (ABL & 64) { o.cell = (int)(z * 1000.f) & 0xffff; o.sx = x; o.sy = y; o.patch = 3; o.alpha = z; o.r2 = z; }
    else geom_point_fast(gc, x, y, z, transforms + 16 * f, o, !(ABL & 16));
    if (!(ABL & 32)) p_cell[j] = o.cell;
    if (o.cell < 0) return;
    uint32_t rgbv = 0x010203;
    if (!(ABL & 1)) {
        const uint8_t *pv = rgb + ((int64_t)f * N + (int64_t)o.sy * gc.W + o.sx) * rgb_ch;
        rgbv = (uint32_t)pv[0] | ((uint32_t)pv[1] << 8) | ((uint32_t)pv[2] << 16);
    }
    if (!(ABL & 32)) p_patf[j] = ((uint32_t)f << 16) | o.patch;
    if (ABL & 128) {
        ((uint4 *)p_rec)[j] = make_uint4((uint32_t)__double2loint(o.alpha), (uint32_t)__double2hiint(o.alpha), rgbv, 0u);
    } else if (ABL & 256) {
        ((double *)p_rec)[j] = o.alpha;
        ((uint32_t *)p_rec)[2 * P + j] = rgbv;
    } else if (!(ABL & 4)) {
        PointRec rec;
        rec.alo = (uint32_t)__double2loint(o.alpha); rec.ahi = (uint32_t)__double2hiint(o.alpha); rec.rgbv = rgbv;
        p_rec[j] = rec;
    } else if (rgbv == 0x7fffffff) p_cell[j] = 1;
    if (!(ABL & 2)) { if (occ[o.cell] < 0) atomicMin(&occ[o.cell], INT_MIN + (int32_t)j); }
}

template <int ABL>
static float run(const char *name, GeomConst gc, float *depth, uint8_t *rgb, double *T, int64_t P, int32_t *occ,
                 int32_t *cell, uint32_t *patf, PointRec *rec)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const dim3 grid((unsigned)((P + TPB - 1) / TPB));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kp<ABL>, grid, dim3(TPB), 0, 0, gc, depth, rgb, 4, T, P, 1.0f / gc.W, occ, cell, patf, rec);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kp<ABL>, grid, dim3(TPB), 0, 0, gc, depth, rgb, 4, T, P, 1.0f / gc.W, occ, cell, patf, rec);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %7.3f ms / launch   (%.1f Gpts/s)\n", name, ms / 5, P / (ms / 5) * 1e-6);
    return ms / 5;
}

int main(int argc, char **argv)
{
    const int F = argc > 1 ? atoi(argv[1]) : 384, H = 480, W = 640, gs = 256, g = 14;
    const int64_t N = (int64_t)H * W, P = F * N;
    GeomConst gc{};
    const double fx = W / 2.0;
    double K[9] = {fx, 0, W / 2.0, 0, fx, H / 2.0, 0, 0, 1}, Ki[9] = {1 / fx, 0, -(W / 2.0) / fx, 0, 1 / fx, -(H / 2.0) / fx, 0, 0, 1};
    double Kp[9] = {g / 2.0, 0, g / 2.0, 0, g / 2.0, g / 2.0, 0, 0, 1};
    memcpy(gc.K, K, sizeof K); memcpy(gc.Kinv, Ki, sizeof Ki); memcpy(gc.Kp, Kp, sizeof Kp);
    gc.cs = 0.1; gc.half_gs = gs / 2.0; gc.min_depth = 0.1; gc.max_depth = 10; gc.H = H; gc.W = W; gc.gs = gs;
    gc.min_h = -128; gc.max_h = 128; gc.nh = 256; gc.g = g; gc.fast = 1; gc.rcs = 1.0 / 0.1;
    std::vector<uint8_t> tx(W), ty(H);
    for (int i = 0; i < W; ++i) { long t = (long)(Kp[0] * (Ki[0] * (i + 0.5) + Ki[2]) + Kp[2] - 0.5); tx[i] = (t >= 0 && t < g) ? t : 255; }
    for (int i = 0; i < H; ++i) { double u = Kp[4] * (Ki[4] * (i + 0.5) + Ki[5]) + Kp[5] - 0.5; long t = (long)u; ty[i] = (u > -1 && t >= 0 && t < g) ? t : 255; }
    uint8_t *dtx, *dty; hipMalloc(&dtx, W); hipMalloc(&dty, H);
    hipMemcpy(dtx, tx.data(), W, hipMemcpyHostToDevice); hipMemcpy(dty, ty.data(), H, hipMemcpyHostToDevice);
    gc.pat_x = dtx; gc.pat_y = dty;
    // smooth depth: a wall 3 m ahead with centimetre noise; identity-ish transforms
    std::vector<float> hd(N);
    const int vary = argc > 2 ? atoi(argv[2]) : 0;   // 0: flat wall, 1: slanted planes (cells vary across the image and per frame)
    for (int64_t i = 0; i < N; ++i) {
        const int xx = (int)(i % W), yy = (int)(i / W);
        hd[i] = (vary ? 2.0f + 2.0f * xx / W + 0.7f * yy / H : 3.0f) + 0.01f * ((i * 2654435761u >> 8) % 100) / 100.f;
    }
    float *depth; uint8_t *rgb; double *T; int32_t *occ, *cell; uint32_t *patf; PointRec *rec;
    hipMalloc(&depth, P * 4); hipMalloc(&rgb, P * 4); hipMalloc(&T, F * 16 * 8); hipMalloc(&occ, (size_t)gs * gs * 256 * 4);
    hipMalloc(&cell, P * 4); hipMalloc(&patf, P * 4); hipMalloc(&rec, P * 16);
    for (int f = 0; f < F; ++f) {
        if (vary) for (int64_t i = 0; i < N; i += 7) hd[i] += 0.003f;
        hipMemcpy(depth + f * N, hd.data(), N * 4, hipMemcpyHostToDevice);
    }
    hipMemset(rgb, 7, P * 4);
    std::vector<double> hT(F * 16, 0.0);
    for (int f = 0; f < F; ++f) { double *t = &hT[f * 16]; t[2] = 1; t[4] = -1; t[9] = -1; t[15] = 1; t[3] = 0.01 * f; t[11] = 1.5; if (vary) { t[7] = 0.02 * (f % 50); } }
    hipMemcpy(T, hT.data(), F * 16 * 8, hipMemcpyHostToDevice);
    hipMemset(occ, 0, (size_t)gs * gs * 256 * 4);     // all cells "occupied": no atomics, like the steady state
#define RUN(A, name) run<A>(name, gc, depth, rgb, T, P, occ, cell, patf, rec)
    RUN(0, "full");
    RUN(8, "T through readfirstlane (scalar loads)");
    RUN(8 | 128, "scalar T + 16-byte records (dwordx4)");
    RUN(8 | 256, "scalar T + SoA records (8 B alpha, 4 B rgb)");
    RUN(1, "no rgb gather");
    RUN(2, "no occ read");
    RUN(4, "no rec store");
    RUN(16, "no exp");
    RUN(32, "no cell/patf stores");
    RUN(8 | 16, "scalar T + no exp");
    RUN(1 | 2 | 4 | 32, "geometry only (no gathers, no stores)");
    RUN(64, "no geometry (loads + stores only)");
    RUN(64 | 1 | 2, "stream only: depth in, 20 B out");
    return 0;
}
