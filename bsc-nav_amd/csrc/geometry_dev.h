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

typedef unsigned long long u64_t;

struct GeomConst {
    double K[9], Kinv[9], Kp[9];
    double cs, half_gs, min_depth, max_depth;
    int32_t H, W, gs, min_h, max_h, nh, g;
    // fast path (geom_point_fast), valid when `fast` != 0: the three intrinsic matrices have the pinhole structure
    // [[a,0,b],[0,c,d],[0,0,1]] exactly, and no pixel centre maps onto a patch boundary (patch tables are exact)
    int32_t fast;
    double rcs;               // RN(1 / cs)
    const uint8_t *pat_x;     // (W) patch column of pixel column x, 255 = outside [0, g)
    const uint8_t *pat_y;     // (H) patch row of pixel row y
    // bsc_exp (below): constants as kernel arguments (scalar registers: an f64 instruction reads them directly; as literals every
    // use cost two register moves) and the table of 2^(j/64) as (hi, lo) pairs
    double exp_il, exp_l1, exp_l2, exp_c2, exp_c3, exp_c4, exp_c5;
    const double2 *exp_tab;
    // 8-byte point records (rec8_pack / rec8_alpha below; every-pixel builds with the fast geometry): depth as an offset from
    // zbase = the bits of a float at or below min_depth, and exact division of a point index by N = H * W and of a pixel index by W
    // (multiply by div_*_m, shift by 32 + div_*_s: exact for every 32-bit dividend)
    uint32_t zbase;
    int32_t rec_lb;           // log2 of the points per k_points workgroup (the record carries the point's index inside its block)
    int32_t div_n_s, div_w_s;
    u64_t div_n_m, div_w_m;
    // round 6: two shortcuts of the fast path, each with the form it replaces kept behind its flag
    int32_t proj_id;          // K (Kinv p2d) is p2d up to rounding for every pixel centre and min_depth >= 0 (bsc_create checks): the
                              // source pixel is x or x - 1 and the choice is made without the division (geom_point_fast_t)
    int32_t gs_even;          // gs / 2 is an integer: row = gs / 2 - trunc(x / cs) in integer arithmetic
};

// exp(x) for the point weights alpha = exp(-r^2 / 1.2) (memory_2.py:873-875), x <= 0.  Table-driven (Tang): k = rint(x 64 / ln 2),
// r = x - k ln2/64 in two steps (|r| <= ln2/128: the degree-5 polynomial for exp(r) - 1 is good to 2^-56), exp(x) = 2^(k >> 6)
// T[k & 63] (1 + p) with T as (hi, lo): 14 f64 instructions against ~25 (+ ~18 register moves for its literals) of the device
// library's exp, which this replaces on every device-alpha path.  Error <= 1 ulp (tests/test_gpu_edges.py: against NumPy's exp
// over the whole depth range) — the same class as the library routine: the dense modes' rgb bytes differ from a host-alpha build
// in < 0.1 % of the bytes either way; the reference-exact mode takes alpha from the host (bit-exact).  tab: 64 x (hi, lo), in LDS
// for k_points.  Out-of-range lanes (their results are discarded): the table index stays in range, nothing traps; -inf (an
// infinite depth, an r^2 beyond the f64 range) gives 0 like exp() — bsc_geometry reports alpha for flagged points too.
__device__ __forceinline__ double bsc_exp(double x, const GeomConst &c, const double2 *tab)
{
    x = x < -800.0 ? -800.0 : x;            // exp(-inf) = 0 like exp() (the scaling below underflows to 0 from about -745); NaN stays NaN
    const double kf = __builtin_rint(__dmul_rn(x, c.exp_il));
    const int k = (int)kf;
    double r = __fma_rn(-kf, c.exp_l1, x);
    r = __fma_rn(-kf, c.exp_l2, r);
    double q = __fma_rn(r, c.exp_c5, c.exp_c4);
    q = __fma_rn(r, q, c.exp_c3);
    q = __fma_rn(r, q, c.exp_c2);
    q = __fma_rn(r, q, 1.0);
    const double p = __dmul_rn(r, q);                       // exp(r) - 1
    const double2 t = tab[k & 63];
    return ldexp(__dadd_rn(t.x, __fma_rn(t.x, p, t.y)), k >> 6);
}

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
                                           bool want_alpha, const double2 *exp_tab)
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
    o.alpha = want_alpha ? bsc_exp(__ddiv_rn(-o.r2, 2 * 0.6), c, exp_tab) : 0.0;
}


// ---- fast path --------------------------------------------------------------------------------------------------
// Same results as geom_point, bit for bit, for what the ingest needs (voxel cell, source pixel, patch, r2, alpha), at a
// third of the instructions.  What is used:
//  * pinhole structure: an ascending-k fma chain over a row [a, 0, b] collapses to RN(RN(a*u) + b*w) — the zero term
//    adds exactly 0 to a non-zero partial sum; the last row [0, 0, 1] returns its third operand.  Hence p2 == z and
//    the denominators of both projections are p2.
//  * x / cs with the constant cs: q = x * RN(1/cs); r = fma(-cs, q, x); fma(r, RN(1/cs), q) is the correctly rounded
//    quotient (Markstein: y the correctly rounded reciprocal, q within an ulp, exact residual) — checked against `/`
//    on 4.8e8 operands incl. the neighbours of every integer multiple of cs.
//  * a / p2 for the projections: the hardware's own f64 division sequence (v_rcp_f64, two Newton steps, multiply,
//    residual fma, final fma — what `/` compiles to inside the normal range) with the reciprocal shared by the quotients.
//  * the patch index depends on the pixel only: (Kp (Kinv p2d z)) / z is z-free up to a few ulps and bsc_create verified
//    that no pixel centre of this configuration lies within 1e-9 of a patch boundary, so it comes from two tables.
struct GeomFastOut {
    int32_t cell;      // (row * gs + col) * nh + (h - min_h), or -1: depth invalid / outside the grid / outside the patches
    int32_t sx, sy;    // source pixel of rgb_v (memory_2.py:869-870), wrapped like a negative NumPy index, clamped
    uint32_t patch;    // py * g + px
    double r2, alpha;
};

__device__ __forceinline__ double div_by_const(double x, double c, double rc)
{
    const double q = __dmul_rn(x, rc);
    return __fma_rn(__fma_rn(-c, q, x), rc, q);
}

// tx, ty: the patch column / row of the pixel (c.pat_x[x], c.pat_y[y]; 255 = outside), fetched by the caller — k_points reads
// them for all its rounds ahead of the geometry so that no round waits for a table lookup
__device__ __forceinline__ void geom_point_fast_t(const GeomConst &c, int32_t x, int32_t y, float zf, const double *T,
                                                  GeomFastOut &o, bool want_alpha, uint32_t tx, uint32_t ty, const double2 *exp_tab)
{
    // Straight-line on purpose: every lane evaluates the whole chain and the checks only gate the result.  With early returns the
    // wavefront still issued every instruction (some lane is always valid) and paid ~40 register moves per point on top for the
    // defaults and joins of the divergent branches.  Lanes that fail a check compute on whatever their depth gives; conversions
    // saturate, nothing traps, and their outputs are discarded.
    const double px = (double)x + 0.5, py = (double)y + 0.5;
    const double z = (double)zf;
    bool ok = (z > c.min_depth) && (z < c.max_depth);           // (false for NaN)
    ok = ok && tx != 255u && ty != 255u;                        // memory_2.py:878 patch range
    o.patch = ty * (uint32_t)c.g + tx;
    const double p0 = __dmul_rn(__dadd_rn(__dmul_rn(c.Kinv[0], px), c.Kinv[2]), z);
    const double p1 = __dmul_rn(__dadd_rn(__dmul_rn(c.Kinv[4], py), c.Kinv[5]), z);
    const double g0 = dot4_fma(T + 0, p0, p1, z, 1.0);
    const double g1 = dot4_fma(T + 4, p0, p1, z, 1.0);
    const double g2 = dot4_fma(T + 8, p0, p1, z, 1.0);
    const int32_t tr = (int32_t)div_by_const(g0, c.cs, c.rcs), tc = (int32_t)div_by_const(g1, c.cs, c.rcs);
    int32_t row, col;
    if (c.gs_even) {
        // int(gs/2 - float(int(q))) with an integral gs/2: the difference of two integers below 2^32 is exact in f64, so the
        // second truncation does nothing (a saturated first conversion lands outside the grid either way)
        row = (int32_t)((uint32_t)(c.gs >> 1) - (uint32_t)tr);
        col = (int32_t)((uint32_t)(c.gs >> 1) - (uint32_t)tc);
    } else {
        row = (int32_t)(c.half_gs - (double)tr);
        col = (int32_t)(c.half_gs - (double)tc);
    }
    const int32_t h = (int32_t)div_by_const(g2, c.cs, c.rcs);
    ok = ok && !(col >= c.gs || row >= c.gs || h >= c.max_h || col < 0 || row < 0 || h < c.min_h);
    // project_point(calib_mat): q0 / p2 - 0.5, q1 / p2 - 0.5 (mathematically integers: knife edge, evaluated exactly)
    const double q0 = __fma_rn(c.K[2], z, __dmul_rn(c.K[0], p0));
    const double q1 = __fma_rn(c.K[5], z, __dmul_rn(c.K[4], p1));
    int sx, sy;
    if (c.proj_id) {
        // u = RN(q0 / z) is px = x + 0.5 up to a few ulps (K Kinv = 1), and int(u - 0.5) — the subtraction is exact — is x, or
        // x - 1 when u < px (x >= 1; at x = 0 both signs of u - 0.5 truncate to 0).  px has a handful of significant bits: it is the
        // even neighbour of the tie, so u < px  <=>  q0 / z < px - h with h half the spacing of the doubles below px (px = x + 0.5
        // with x >= 1 is no power of two)  <=>  q0 - px z < -h z for z > 0.  px z is exact (<= 25 + 24 bits), q0 lies within a few
        // ulps of it, so the fma returns the difference exactly; h z is a scaling by a power of two.  No reciprocal, no Newton
        // steps, no quotients: 10 f64 instructions for both coordinates instead of 19.
        const double rx = __fma_rn(-px, z, q0), ry = __fma_rn(-py, z, q1);
        const double hx = __hiloint2double((__double2hiint(px) & 0x7ff00000) - (53 << 20), 0);
        const double hy = __hiloint2double((__double2hiint(py) & 0x7ff00000) - (53 << 20), 0);
        sx = x - ((x >= 1 && rx < -__dmul_rn(hx, z)) ? 1 : 0);
        sy = y - ((y >= 1 && ry < -__dmul_rn(hy, z)) ? 1 : 0);
    } else {
        double rz = __builtin_amdgcn_rcp(z);
        double e = __fma_rn(-z, rz, 1.0); rz = __fma_rn(rz, e, rz);
        e = __fma_rn(-z, rz, 1.0); rz = __fma_rn(rz, e, rz);
        const double u0 = __dmul_rn(q0, rz), u1 = __dmul_rn(q1, rz);
        const double u = __fma_rn(__fma_rn(-z, u0, q0), rz, u0);
        const double v = __fma_rn(__fma_rn(-z, u1, q1), rz, u1);
        sx = (int32_t)__dsub_rn(u, 0.5); sy = (int32_t)__dsub_rn(v, 0.5);
        sx += sx < 0 ? c.W : 0;
        sy += sy < 0 ? c.H : 0;
    }
    o.sx = min(max(sx, 0), c.W - 1);
    o.sy = min(max(sy, 0), c.H - 1);
    o.r2 = __dadd_rn(__dadd_rn(__dmul_rn(p0, p0), __dmul_rn(p1, p1)), __dmul_rn(z, z));
    o.alpha = want_alpha ? bsc_exp(div_by_const(-o.r2, 1.2, 1.0 / 1.2), c, exp_tab) : 0.0;
    o.cell = ok ? (row * c.gs + col) * c.nh + (h - c.min_h) : -1;
}

__device__ __forceinline__ void geom_point_fast(const GeomConst &c, int32_t x, int32_t y, float zf, const double *T,
                                                GeomFastOut &o, bool want_alpha)
{
    geom_point_fast_t(c, x, y, zf, T, o, want_alpha, c.pat_x[x], c.pat_y[y], c.exp_tab);
}

// ---- 8-byte point records ---------------------------------------------------------------------------------------------------
// What the rgb chain needs of a point is its colour and its weight alpha = exp(-r^2 / 1.2) (memory_2.py:870-875).  alpha is a
// function of the pixel and the depth alone (r^2 is taken in the camera frame), and a valid depth lies in (min_depth, max_depth):
// float_as_uint(z) - float_as_uint(min_depth) needs 26 bits for (0.1, 10) — sign and most exponent bits are constant.  So a record
// is   rgb (24 bits) | index of the point inside its k_points block (lb bits) | depth offset (40 - lb bits)   = 8 bytes instead of
// the 12 of {alpha f64, rgb}: the chain recomputes alpha with the very instructions k_points used (same bsc_exp, same operand
// order: bit-identical), from the record and the record's position (which names the block).
__device__ __forceinline__ uint32_t div_magic(uint32_t v, u64_t m, int s)
{
    // floor(v / d) for the divisor the (m, s) pair was made for: m = floor(2^(32+s) / d) + 1 has 33 bits, v * m < 2^65 is taken as
    // (v * low 32 bits of m) + (v << 32) when bit 32 of m is set — the sum is below 2^64 for v < 2^31, which every caller guarantees
    // (point indices are below max_points < 2^31)
    const u64_t lo = (u64_t)v * (uint32_t)m + ((m >> 32) ? ((u64_t)v << 32) : 0ull);
    return (uint32_t)(lo >> (32 + s));
}

__device__ __forceinline__ void rec8_pack(const GeomConst &c, uint32_t rgbv, uint32_t p_local, float z, uint32_t &lo, uint32_t &hi)
{
    lo = (rgbv & 0xffffffu) | (p_local << 24);
    hi = (p_local >> 8) | ((__float_as_uint(z) - c.zbase) << (c.rec_lb - 8));
}

// alpha of the record (lo, hi) that sits at position `pos` of the call's record array
__device__ __forceinline__ double rec8_alpha(const GeomConst &c, uint32_t lo, uint32_t hi, uint32_t pos, const double2 *exp_tab)
{
    const int lb = c.rec_lb;
    const uint32_t p_local = (lo >> 24) | ((hi & ((1u << (lb - 8)) - 1u)) << 8);
    const double z = (double)__uint_as_float((hi >> (lb - 8)) + c.zbase);
    const uint32_t j = (pos & ~((1u << lb) - 1u)) | p_local;
    const uint32_t f = div_magic(j, c.div_n_m, c.div_n_s);
    const uint32_t i = j - f * (uint32_t)(c.H * c.W);
    const uint32_t y = div_magic(i, c.div_w_m, c.div_w_s);
    const uint32_t x = i - y * (uint32_t)c.W;
    const double px = (double)(int32_t)x + 0.5, py = (double)(int32_t)y + 0.5;
    const double p0 = __dmul_rn(__dadd_rn(__dmul_rn(c.Kinv[0], px), c.Kinv[2]), z);
    const double p1 = __dmul_rn(__dadd_rn(__dmul_rn(c.Kinv[4], py), c.Kinv[5]), z);
    const double r2 = __dadd_rn(__dadd_rn(__dmul_rn(p0, p0), __dmul_rn(p1, p1)), __dmul_rn(z, z));
    return bsc_exp(div_by_const(-r2, 1.2, 1.0 / 1.2), c, exp_tab);      // the expression of geom_point_fast_t, operand for operand
}

