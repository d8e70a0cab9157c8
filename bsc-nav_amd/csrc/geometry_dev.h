// geometry_dev.h — fp64 point geometry of one depth sample (device code, gfx950).
//
// Follows utils.py:153-214 and memory_2.py:864-875 of the reference.  The reference evaluates the
// 3x3 / 4x4 products with NumPy `@` (OpenBLAS); those results are reproduced bit-for-bit by an
// ascending-k chain of IEEE fused multiply-adds starting from 0, written here with
// explicit __fma_rn.  The translation unit is compiled with -ffp-contract=off so that no other
// multiply-add pair is fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GeomConst {
    double K[9], Kinv[9], Kp[9];
    double cs, half_gs, min_depth, max_depth;
    int32_t H, W, gs, min_h, max_h, nh, g;
};

struct GeomOut {
    double pc[3], pg[3];
    double r2, alpha;
    int32_t vox[3];   // row, col, h (before the -min_h shift)
    int32_t pix[2];   // recovered source pixel x, y
    int32_t pat[2];   // patch x, y
    uint32_t flags;   // bit0 depth-valid, bit1 grid in-range, bit2 patch in-range
};

__device__ __forceinline__ double dot3_fma(const double *r, double a, double b, double c)
{
    double acc = 0.0;
    acc = __fma_rn(r[0], a, acc);
    acc = __fma_rn(r[1], b, acc);
    acc = __fma_rn(r[2], c, acc);
    return acc;
}

__device__ __forceinline__ double dot4_fma(const double *r, double a, double b, double c, double d)
{
    double acc = 0.0;
    acc = __fma_rn(r[0], a, acc);
    acc = __fma_rn(r[1], b, acc);
    acc = __fma_rn(r[2], c, acc);
    acc = __fma_rn(r[3], d, acc);
    return acc;
}

// i: row-major pixel index inside the frame; z: its depth; T: 4x4 pc_transform (row-major)
__device__ __forceinline__ void geom_point(const GeomConst &c, int32_t i, float zf, const double *T, GeomOut &o,
                                           bool want_alpha)
{
    const int y = i / c.W, x = i - y * c.W;
    const double px = (double)x + 0.5, py = (double)y + 0.5;   // utils.py:167-168
    const double z = (double)zf;
    // utils.py:172-173  pc = Kinv @ p2d ; pc = pc * z
    const double p0 = __dmul_rn(dot3_fma(c.Kinv + 0, px, py, 1.0), z);
    const double p1 = __dmul_rn(dot3_fma(c.Kinv + 3, px, py, 1.0), z);
    const double p2 = __dmul_rn(dot3_fma(c.Kinv + 6, px, py, 1.0), z);
    o.pc[0] = p0; o.pc[1] = p1; o.pc[2] = p2;
    o.flags = 0;
    if (!((p2 > c.min_depth) && (p2 < c.max_depth))) return;   // utils.py:175-177 (strict)
    o.flags = 1;
    // utils.py:189-199 transform_pc
    const double g0 = dot4_fma(T + 0, p0, p1, p2, 1.0);
    const double g1 = dot4_fma(T + 4, p0, p1, p2, 1.0);
    const double g2 = dot4_fma(T + 8, p0, p1, p2, 1.0);
    o.pg[0] = g0; o.pg[1] = g1; o.pg[2] = g2;
    // utils.py:201-205 base_pos2grid_id_3d: int() truncates toward zero, twice
    const int32_t row = (int32_t)(c.half_gs - (double)(int32_t)__ddiv_rn(g0, c.cs));
    const int32_t col = (int32_t)(c.half_gs - (double)(int32_t)__ddiv_rn(g1, c.cs));
    const int32_t h = (int32_t)__ddiv_rn(g2, c.cs);
    o.vox[0] = row; o.vox[1] = col; o.vox[2] = h;
    // memory_2.py:755-756
    const bool in_range = !(col >= c.gs || row >= c.gs || h >= c.max_h || col < 0 || row < 0 || h < c.min_h);
    if (in_range) o.flags |= 2;
    // utils.py:208-214 project_point (calib_mat), then (patch intrinsics)
    {
        const double q0 = dot3_fma(c.K + 0, p0, p1, p2);
        const double q1 = dot3_fma(c.K + 3, p0, p1, p2);
        const double q2 = dot3_fma(c.K + 6, p0, p1, p2);
        o.pix[0] = (int32_t)__dsub_rn(__ddiv_rn(q0, q2), 0.5);
        o.pix[1] = (int32_t)__dsub_rn(__ddiv_rn(q1, q2), 0.5);
    }
    {
        const double q0 = dot3_fma(c.Kp + 0, p0, p1, p2);
        const double q1 = dot3_fma(c.Kp + 3, p0, p1, p2);
        const double q2 = dot3_fma(c.Kp + 6, p0, p1, p2);
        o.pat[0] = (int32_t)__dsub_rn(__ddiv_rn(q0, q2), 0.5);
        o.pat[1] = (int32_t)__dsub_rn(__ddiv_rn(q1, q2), 0.5);
    }
    if (!(o.pat[0] < 0 || o.pat[1] < 0 || o.pat[0] >= c.g || o.pat[1] >= c.g)) o.flags |= 4;   // memory_2.py:878
    // memory_2.py:873-875  r2 = (x^2 + y^2) + z^2 ; alpha = exp(-r2 / 1.2)
    o.r2 = __dadd_rn(__dadd_rn(__dmul_rn(p0, p0), __dmul_rn(p1, p1)), __dmul_rn(p2, p2));
    o.alpha = want_alpha ? exp(__ddiv_rn(-o.r2, 2 * 0.6)) : 0.0;
}
