// encoder_gemm.hip — the encoder's dense layers at the reference's precision on the 16-bit matrix cores.
//
// The reference runs DINOv2 in f32 (memory_2.py:43, 738-739).  gfx950 has no TF32 and its f32 MFMA peaks at 157 TFLOP/s, so an f32
// GEMM of the vendor library runs ~115 TFLOP/s; the 16-bit MFMA peaks at 2.5 PFLOP/s.  An f32 value is, to 22 significant bits,
// the sum of two fp16 pieces  h = fp16(x), l = fp16(x - h)  (the difference is exact in f32; both roundings to nearest), so
//     x w = xh wh + xh wl + xl wh + O(2^-22 |x w|)
// and the three piece products are exact in the matrix core's f32 accumulator: three v_mfma_f32_32x32x16_f16 per f32 product
// term at 16x the f32 MFMA rate.  (Three bf16 pieces need six products for the same accuracy; with three products they leave
// 2^-16 — k_cosine_bf16x3 of localize.hip is the six-product form.  fp16 pieces have a 5-bit exponent: the weights are scaled by
// a power of two per matrix so that their largest element sits near 2^3, the activation operand by a per-layer power of two, and
// the scales leave through the epilogue exactly; what remains of the narrow range is that the l piece of an element 2^10 below the
// operand's typical magnitude goes subnormal, i.e. is kept to an ABSOLUTE 2^-25 — far below the 2^-22 relative error of the
// typical term.  |x a_scale| must stay below 65504.)
//
//   C[m][n] = epilogue( out_scale * sum_k A[m][k] W[n][k] + bias[n] )        A (M,K) f32 row-major, W (N,K) as nn.Linear holds it
//
//   workgroup  256 rows of A x 256 rows of W, 8 wavefronts; wavefront w owns the 32 rows [32 w, 32 w + 32) of the tile
//   A operand  straight from global memory in fragment layout — lane (i = lane & 31, g = lane >> 5) loads the 16 floats
//              [32 c + 16 g, + 16) of its row (one 128-byte line per row and K chunk over the two lane groups) and splits
//              them in registers (v_cvt_pk_f16_f32);
//   W operand  pieces precomputed once per weight matrix (bsc_enc_split_weights: planes h, l of (N,K) fp16), staged per 32-wide
//              K chunk through LDS (80-byte row pitch: conflict-free 16-byte reads), double-buffered;
//   MFMA       the weights are the FIRST operand (D' = W X^T), so a lane of the accumulator tile holds ONE row of C and four
//              consecutive columns per register quad;
//   epilogue   bias, bias + GELU(tanh), bias + residual (C may alias the residual) — nothing elementwise is left between GEMMs.
//              Each wavefront turns its 32 x 32 tiles round in 4 KB of LDS of its own and moves them as whole 128-byte lines
//              (f32 rows or piece chunks), the residual rows requested three tiles ahead.
// Workgroup ids map so that the workgroups of an XCD (id mod 8) walk the column tiles of ONE row tile back to back: the 786 KB
// A tile is fetched into that XCD's L2 once.
#include "bsc_internal.h"

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define GS_KC 32                 // K chunk
#define GS_PITCH 40              // fp16 elements per staged weight row (80 bytes)
#define GS_TPB 256
#ifndef GS_RESID_DEPTH
#define GS_RESID_DEPTH 1             // tiles of residual rows in flight in the epilogue (incl. the one in use).  Round 5 sweep at 768 frames
                                     // (proj / fc2, us): depth 3 736 / 2466, 2 719 / 2405, 1 679 / 2328 — the registers of the prefetched rows
                                     // (16 per tile in flight) are worth more to the main loop than the prefetch is to the epilogue
#endif
#ifndef GS_STATS_RESID_DEPTH
#define GS_STATS_RESID_DEPTH 1       // the same in the epilogue that also takes the row statistics
#endif
#ifndef BSC_GEMM_NARROW_TILE
#define BSC_GEMM_NARROW_TILE 1           // tile variant for N <= 1024 (see bsc_enc_gemm_split)
#endif

enum { GS_EPI_BIAS = 0, GS_EPI_GELU = 1, GS_EPI_RESID = 2, GS_EPI_GELU_ERF = 3 };
#define GS_LN_REC 20             // floats per LayerNorm statistics record of a residual-stream row (see gemm_split_tile)

#ifdef BSC_GEMM_PROFILE        // per-workgroup phase stamps (100 MHz wall clock) + the CU it ran on: -DBSC_GEMM_PROFILE, BSC_GEMM_PROFILE_DUMP=1
#define GS_PROF_MAX 8192
__device__ uint64_t g_gemm_prof[GS_PROF_MAX][6];
#define GS_T(k)                                                                                                                   \
    do {                                                                                                                          \
        if (threadIdx.x == 0 && prof_idx < GS_PROF_MAX) g_gemm_prof[prof_idx][k] = wall_clock64();                                \
    } while (0)
#else
#define GS_T(k)
#endif

__device__ __forceinline__ uint32_t pack_f16_rne(float lo, float hi)        // v_cvt_pk_f16_f32
{
    const f32x2_t v = {lo, hi};
    const half2_t r = __builtin_convertvector(v, half2_t);
    return *(const uint32_t *)&r;
}

// orders a wavefront's own LDS accesses (waits for its LDS operations only; no workgroup barrier)
__device__ __forceinline__ void gs_wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

// (a, b) -> packed fp16 pieces h, l of both
__device__ __forceinline__ void split2(float a, float b, uint32_t &h, uint32_t &l)
{
    h = pack_f16_rne(a, b);
    const half2_t hv = *(const half2_t *)&h;
    const f32x2_t hb = __builtin_convertvector(hv, f32x2_t);
    l = pack_f16_rne(a - hb[0], b - hb[1]);
}

__device__ __forceinline__ void split2v(f32x2_t v, uint32_t &h, uint32_t &l)
{
    const half2_t hv = __builtin_convertvector(v, half2_t);
    h = *(const uint32_t *)&hv;
    const half2_t lv = __builtin_convertvector(v - __builtin_convertvector(hv, f32x2_t), half2_t);
    l = *(const uint32_t *)&lv;
}

// W (N,K) f32 -> planes h, l of (n_pad, K) fp16 pieces of scale * W (rows N..n_pad-1 zero)
__global__ __launch_bounds__(GS_TPB) void k_split_weights(const float *__restrict__ W, int64_t n_el, int64_t n_pad_el, float scale,
                                                          uint16_t *__restrict__ out)
{
    const int64_t i = ((int64_t)blockIdx.x * GS_TPB + threadIdx.x) * 2;
    if (i >= n_pad_el) return;
    uint32_t h = 0, l = 0;
    if (i < n_el) split2(W[i] * scale, W[i + 1] * scale, h, l);
    *(uint32_t *)(out + i) = h;
    *(uint32_t *)(out + n_pad_el + i) = l;
}

__device__ __forceinline__ float gelu_tanh(float x)
{
    // torch.nn.functional.gelu(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))), with
    // 1 + tanh(u) = 2 - 2 / (1 + e^(2u)) on the hardware's exp2 / rcp (each within an ulp; the sum is taken against 1)
    const float kBeta = 0.7978845608028654f * 0.044715f, kAlpha = 0.7978845608028654f;
    const float u = fmaf(kBeta, x * x * x, kAlpha * x);
    const float e = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);          // e^(2u)
    return x * (1.0f - __builtin_amdgcn_rcpf(1.0f + e));
}

// The exact form, torch.nn.GELU() as the reference's DINOv2 uses it (memory_2.py:43,738: hub model, approximate='none'):
// 0.5 x (1 + erf(x / sqrt 2)).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, the resolution of f32 itself) on the
// hardware's exp2 / rcp: with q = (a1 t + .. + a5 t^5) e^(-z^2), t = 1 / (1 + p z), z = |x| / sqrt 2 (q = erfc z),
// 1 + erf = 2 - q for x >= 0 and q for x < 0 — no cancellation on either side.  The tanh form differs from it by up to 5e-4.
__device__ __forceinline__ float gelu_erf(float x)
{
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float q = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    return x * (x >= 0.f ? fmaf(-0.5f, q, 1.0f) : 0.5f * q);
}
template <int EPI> __device__ __forceinline__ float gelu_of(float x) { return EPI == GS_EPI_GELU_ERF ? gelu_erf(x) : gelu_tanh(x); }

// ---- activation pieces --------------------------------------------------------------------------------------------------------
// Layout P32 of an (M,K) activation matrix as fp16 pieces: row m = K/32 chunks of 64 halfs, chunk c = [h of k = 32c..32c+31 |
// l of the same] — one 128-byte line per row and K chunk holds both pieces, as the f32 row did.
__device__ __forceinline__ int64_t p32_off(int64_t m, int K, int k) { return m * 2 * K + (int64_t)(k >> 5) * 64 + (k & 31); }

// LayerNorm of f32 rows straight into pieces (the input of the qkv / fc1 GEMMs): one wavefront per row, two passes over
// registers (mean, then centred variance — as torch.nn.functional.layer_norm does in f32)
template <int VPL>      // float4 per lane: width = 256 * VPL
__global__ __launch_bounds__(GS_TPB) void k_layernorm_split(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, int64_t rows, float eps, float a_scale,
                                                            uint16_t *__restrict__ out)
{
    constexpr int Wd = 256 * VPL;
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * GS_TPB + threadIdx.x) >> 6;
    if (row >= rows) return;
    const float4 *xr = (const float4 *)(x + row * Wd);
    float4 v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) { v[j] = xr[lane + 64 * j]; sum += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.0f / Wd);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        sq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / Wd) + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int k = 4 * (lane + 64 * j);
        const float4 gm = ((const float4 *)gamma)[lane + 64 * j], bt = ((const float4 *)beta)[lane + 64 * j];
        const float y0 = (v[j].x * rstd * gm.x + bt.x) * a_scale, y1 = (v[j].y * rstd * gm.y + bt.y) * a_scale;
        const float y2 = (v[j].z * rstd * gm.z + bt.z) * a_scale, y3 = (v[j].w * rstd * gm.w + bt.w) * a_scale;
        uint32_t h0, l0, h1, l1;
        split2(y0, y1, h0, l0);
        split2(y2, y3, h1, l1);
        uint16_t *o = out + p32_off(row, Wd, k);
        *(uint2 *)o = make_uint2(h0, h1);
        *(uint2 *)(o + 32) = make_uint2(l0, l1);
    }
}

// Token assembly of the f32 forward + the first LayerNorm, one pass (one wavefront per token row): row (b, 0) = cls + pos[0], rows
// (b, 1 .. R) = the register tokens (no position term), rows (b, 1 + R + j) = patch[b][j] + pos[1 + j] -> the residual stream u
// (f32) and LayerNorm(u) as the pieces the first qkv GEMM reads.
template <int VPL>
__global__ __launch_bounds__(GS_TPB) void k_embed_layernorm_split(const float *__restrict__ patch, const float *__restrict__ cls,
                                                                  const float *__restrict__ reg, const float *__restrict__ pos,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                  int64_t rows, int T, int R, float eps, float *__restrict__ u,
                                                                  uint16_t *__restrict__ out, float *__restrict__ stats,
                                                                  float *__restrict__ mu)
{
    constexpr int Wd = 256 * VPL;
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * GS_TPB + threadIdx.x) >> 6;
    if (row >= rows) return;
    const int64_t b = row / T;
    const int t = (int)(row - b * T);
    const int np = T - 1 - R;
    const float4 *src = (const float4 *)(t == 0 ? cls : t <= R ? reg + (int64_t)(t - 1) * Wd : patch + (b * np + (t - 1 - R)) * Wd);
    const float4 *pp = (t >= 1 && t <= R) ? nullptr : (const float4 *)(pos + (int64_t)(t == 0 ? 0 : t - R) * Wd);
    float4 v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        v[j] = src[lane + 64 * j];
        if (pp) { const float4 q = pp[lane + 64 * j]; v[j].x += q.x; v[j].y += q.y; v[j].z += q.z; v[j].w += q.w; }
        ((float4 *)(u + row * Wd))[lane + 64 * j] = v[j];
        sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.0f / Wd);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        sq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (stats) {        // the row's statistics record for a GEMM that folds the LayerNorm into its operand load (GS_LN_REC): exact two-pass
        if (lane < GS_LN_REC) stats[row * GS_LN_REC + lane] = lane == 0 ? mean : lane == 3 ? sq : 0.f;
        if (lane == 0) mu[row] = mean;
    }
    if (!out) return;
    const float rstd = 1.0f / sqrtf(sq * (1.0f / Wd) + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int k = 4 * (lane + 64 * j);
        const float4 gm = ((const float4 *)gamma)[lane + 64 * j], bt = ((const float4 *)beta)[lane + 64 * j];
        uint32_t h0, l0, h1, l1;
        split2(v[j].x * rstd * gm.x + bt.x, v[j].y * rstd * gm.y + bt.y, h0, l0);
        split2(v[j].z * rstd * gm.z + bt.z, v[j].w * rstd * gm.w + bt.w, h1, l1);
        uint16_t *o = out + p32_off(row, Wd, k);
        *(uint2 *)o = make_uint2(h0, h1);
        *(uint2 *)(o + 32) = make_uint2(l0, l1);
    }
}

// The final LayerNorm of the f32 forward over the PATCH rows only (cls and register rows skipped), written as the token tensor
// (B, g*g, width) f32 the memory path reads (memory_2.py:739 x_norm_patchtokens)
template <int VPL>
__global__ __launch_bounds__(GS_TPB) void k_final_layernorm_f32(const float *__restrict__ u, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, int64_t out_rows, int T, int skip,
                                                                float eps, float *__restrict__ out)
{
    constexpr int Wd = 256 * VPL;
    const int lane = threadIdx.x & 63;
    const int64_t orow = ((int64_t)blockIdx.x * GS_TPB + threadIdx.x) >> 6;
    if (orow >= out_rows) return;
    const int np = T - skip;
    const int64_t b = orow / np;
    const int64_t row = b * T + skip + (orow - b * np);
    const float4 *xr = (const float4 *)(u + row * Wd);
    float4 v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) { v[j] = xr[lane + 64 * j]; sum += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.0f / Wd);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        sq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / Wd) + eps);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const float4 gm = ((const float4 *)gamma)[lane + 64 * j], bt = ((const float4 *)beta)[lane + 64 * j];
        ((float4 *)(out + orow * Wd))[lane + 64 * j] =
            make_float4(v[j].x * rstd * gm.x + bt.x, v[j].y * rstd * gm.y + bt.y, v[j].z * rstd * gm.z + bt.z, v[j].w * rstd * gm.w + bt.w);
    }
}

// f32 rows -> pieces (attention output, patch matrix)
__global__ __launch_bounds__(GS_TPB) void k_split_rows(const float *__restrict__ x, int64_t M, int K, float a_scale,
                                                       uint16_t *__restrict__ out)
{
    const int64_t e = ((int64_t)blockIdx.x * GS_TPB + threadIdx.x) * 4;
    if (e >= M * K) return;
    const int64_t m = e / K;
    const int k = (int)(e - m * K);
    const float4 v = *(const float4 *)(x + e);
    uint32_t h0, l0, h1, l1;
    split2(v.x * a_scale, v.y * a_scale, h0, l0);
    split2(v.z * a_scale, v.w * a_scale, h1, l1);
    uint16_t *o = out + p32_off(m, K, k);
    *(uint2 *)o = make_uint2(h0, h1);
    *(uint2 *)(o + 32) = make_uint2(l0, l1);
}

// Workgroup tile (WR * MR * 32) rows x (WC * NT * 32) columns; wavefront (wr, wc) owns MR row fragments x NT column tiles:
// every weight fragment read from LDS feeds MR * 3 MFMAs, every activation fragment NT * 3.
// APIECES: A comes as P32 pieces (made by the producing kernel) — otherwise as f32 rows, split in registers.
// CPIECES: C leaves as P32 pieces scaled by c_scale (the hidden tensor of the MLP: read by the next GEMM only).
// One tile of a persistent workgroup.  primed: the first two A chunks (xf, xg) and the first weight chunk (LDS buffer `par`) were
// requested / staged by the tile before, under its last two K chunks; has_next: do the same for the tile (next_tm, next_n0) — of
// the same shape — so that a workgroup's tiles form one continuous stream of chunks and only its first tile pays the latency of
// a prologue.  par: parity of the LDS buffer that holds chunk 0 (advances by the chunk count per tile).
template <int MR> struct gs_xf { typedef uint32_t type __attribute__((ext_vector_type(16 * MR))); };
// AMODE: how the activation operand arrives — GS_A_F32 f32 rows (split in registers), GS_A_PIECES P32 pieces, GS_A_LN f32 rows of
// the residual stream that are LayerNorm'd while they are split: y = (x - mean) * rstd per row (gamma is folded into the weight
// columns and beta into the bias by the host, once per matrix), mean / rstd from the row's statistics record (below).
// STATS (residual epilogue): the tile's output rows leave their sums for the NEXT LayerNorm in the statistics records.
//
// LayerNorm statistics record of a row of the residual stream (GS_LN_REC floats): [0] shift s, [1] unused, [2 + 2 p] = sum (x - s),
// [3 + 2 p] = sum (x - s)^2 over the columns [128 p, 128 p + 128) — one slot per 128-column strip, written by the wavefront whose
// tile holds the strip (no atomics); unused slots zero.  mean = s + A / W, var = B / W - (A / W)^2 with A, B the slot sums: the
// shift is the row's previous mean (ln_mu, written by the GEMM that consumed the previous record), so the one-pass variance is
// taken about a point within a fraction of a standard deviation of the new mean — as accurate as the two-pass form.
enum { GS_A_F32 = 0, GS_A_PIECES = 1, GS_A_LN = 2 };
template <int MR, int NT, int WR, int WC, int EPI, int AMODE, bool CPIECES, bool STATS>
__device__ __forceinline__ void gemm_split_tile(uint16_t *Ws, char *epi_lds, const float *bias_lds, const void *__restrict__ Av, int64_t M, int K,
                                                const uint16_t *__restrict__ Wp, int64_t w_plane, int N,
                                                const float *R, void *Cv,
                                                float a_scale, float out_scale, float c_scale, int64_t tm, int n0, bool primed,
                                                bool has_next, int64_t next_tm, int next_n0, int &par, typename gs_xf<MR>::type &xf,
                                                typename gs_xf<MR>::type &xg, int prof_idx, float *ln_stats, float *ln_mu, float ln_eps,
                                                int k_full)
{
    // k_full: the operands' row length (A rows, weight rows); K: the stretch of it this launch contracts (split-K slices of the
    // few-rows path: the kernel wrapper has moved the operand pointers to the slice's first column; otherwise K == k_full)
    constexpr bool APIECES = AMODE == GS_A_PIECES;
    constexpr int WROWS = WC * NT * 32;                                       // Ws: [2][2][WROWS][GS_PITCH]
    constexpr int TROWS = WR * MR * 32;
    constexpr int BUF = 2 * WROWS * GS_PITCH;
    constexpr int NTHR = 64 * WR * WC;
    constexpr int NLD = 2 * WROWS * 4 / NTHR;                                 // 16-byte pieces of a weight chunk per thread
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WC, wc = w - wr * WC;
    const int i = lane & 31, g = lane >> 5;
#ifdef BSC_GEMM_PROFILE
    if (threadIdx.x == 0 && prof_idx < GS_PROF_MAX) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_gemm_prof[prof_idx][0] = ((uint64_t)xcc << 32) | hw;
    }
    GS_T(1);
#endif
    const int64_t row0 = tm * TROWS + wr * (MR * 32);
    // per-lane row base: f32 rows (16 floats of the chunk per lane) or P32 pieces (8 halfs of either piece per sub-step)
    auto row_base = [&](int64_t first_row, int mr) {
        const int64_t row = first_row + mr * 32 + i;
        const int64_t rc = row < M ? row : M - 1;                           // clamped: padded rows are not stored
        // lane group g contracts k = 16 g + 8 sstep + (0..7) of the chunk in sub-step sstep — the same split of the 32 on both operands
        return APIECES ? (const char *)((const uint16_t *)Av + rc * 2 * k_full + g * 16) : (const char *)((const float *)Av + rc * k_full + g * 16);
    };
    const char *arow[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) arow[mr] = row_base(row0, mr);
    // LayerNorm constants of the lane's rows (one record per row; the tile with n0 == 0 leaves the means for the next producer)
    float ln_m[MR], ln_rs[MR];
    if constexpr (AMODE == GS_A_LN) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int64_t row = row0 + mr * 32 + i;
            const f32x4_t *rec = (const f32x4_t *)(ln_stats + (row < M ? row : M - 1) * GS_LN_REC);
            const f32x4_t r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
            const float sa = ((r0[2] + r1[0]) + (r1[2] + r2[0])) + ((r2[2] + r3[0]) + (r3[2] + r4[0]));
            const float sb = ((r0[3] + r1[1]) + (r1[3] + r2[1])) + ((r2[3] + r3[1]) + (r3[3] + r4[1]));
            const float inv_w = 1.0f / (float)K, da = sa * inv_w;
            ln_m[mr] = r0[0] + da;
            ln_rs[mr] = 1.0f / sqrtf(fmaxf(sb * inv_w - da * da, 0.f) + ln_eps);
            if (n0 == 0 && wc == 0 && g == 0 && row < M) ln_mu[row] = ln_m[mr];
        }
    }
    const int nchunks = K / GS_KC;
    has_next = has_next && nchunks >= 2;
    f32x16 acc[MR][NT];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][t][r] = 0.f;
    // prefetch registers as first-class vector values (an array that is live across the chunk loop would be left in scratch):
    // xf = the A chunk about to be used, xg = the one after it (two chunks of load latency hidden)
    typedef typename gs_xf<MR>::type xf_t;
    typedef uint32_t qr_t __attribute__((ext_vector_type(4 * NLD)));
    qr_t qr;
    auto load_a = [&](xf_t &xf, const char *const (&ar)[MR], int c) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                // f32: floats [4 v4, 4 v4 + 4) of the lane's 16; pieces: v4 = 2 piece + sub-step -> halfs [16 g + 8 sstep, + 8) of the piece
                const uint4 v = APIECES ? *(const uint4 *)(ar[mr] + (int64_t)c * 128 + (v4 >> 1) * 64 + (v4 & 1) * 16)
                                        : *(const uint4 *)(ar[mr] + (int64_t)c * 128 + 16 * v4);
                xf[16 * mr + 4 * v4] = v.x; xf[16 * mr + 4 * v4 + 1] = v.y; xf[16 * mr + 4 * v4 + 2] = v.z; xf[16 * mr + 4 * v4 + 3] = v.w;
            }
    };
    // weight chunk: thread tid moves the 16-byte part tid & 3 of row (tid >> 2) + PR jj of piece plane p, j = p PPP + jj — ONE
    // per-thread pointer and LDS offset; everything else in the addresses is uniform
    constexpr int PR = NTHR / 4, PPP = WROWS / PR;
    static_assert(WROWS % PR == 0 && NLD == 2 * PPP, "weight staging plan");
    // (buffer loads: a 32-bit per-thread offset + a scalar offset.  With 64-bit pointers the compiler kept one loop-invariant
    // pointer per j in registers — and, in the persistent form of this kernel, in scratch.)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Wp, 0, (int)(4 * w_plane), 0x00020000);
    const int wthr = ((tid >> 2) * k_full + (tid & 3) * 8) * 2;
    const int lthr = (tid >> 2) * GS_PITCH + (tid & 3) * 8;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto load_w = [&](qr_t &q, int nbase, int c) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int uni = (int)(((int64_t)(j / PPP) * w_plane + (int64_t)(nbase + (j % PPP) * PR) * k_full + c * GS_KC) * 2);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wthr, uni, 0);
            q[4 * j] = v[0]; q[4 * j + 1] = v[1]; q[4 * j + 2] = v[2]; q[4 * j + 3] = v[3];
        }
    };
    auto store_w = [&](qr_t &q, int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            *(uint4 *)&Ws[buf * BUF + lthr + ((j / PPP) * WROWS + (j % PPP) * PR) * GS_PITCH] = make_uint4(q[4 * j], q[4 * j + 1], q[4 * j + 2], q[4 * j + 3]);
    };
    // DEEP (the 32-row few-rows tile): a chunk is only 6 MFMAs per wavefront — nothing to hide a weight chunk's round trip behind —,
    // so the weight chunks travel TWO iterations ahead in two register sets that alternate (chunk c + 2 is requested in iteration
    // c, chunk c + 1 — requested an iteration earlier — is stored to LDS at its end): the loop turns on LDS + barrier time instead of
    // one L2 round trip per chunk.  Such tiles do not chain (every tile has its prologue).
    constexpr bool DEEP = MR == 1 && NT == 1 && WR == 1;
    qr_t qr2;
    if (!primed) {
        load_a(xf, arow, 0);
        load_w(qr, n0, 0);
        if (nchunks > 1) load_a(xg, arow, 1);
        if (DEEP && nchunks > 1) load_w(qr2, n0, 1);
        store_w(qr, par);
        __syncthreads();
    }
    GS_T(2);
    auto chunk_body = [&](const int c, qr_t &q_load, qr_t &q_store) __attribute__((always_inline)) {
        const int buf = (c + par) & 1;
        // this chunk's rows as fp16 pieces (two sub-steps of 8 halfs), then the next chunk's loads go in flight
        uint32_t ah[MR][2][4], al[MR][2][4];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            if (APIECES) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ah[mr][0][e] = xf[16 * mr + e]; ah[mr][1][e] = xf[16 * mr + 4 + e];
                    al[mr][0][e] = xf[16 * mr + 8 + e]; al[mr][1][e] = xf[16 * mr + 12 + e];
                }
            } else {
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    // pairs through the packed f32 instructions (v_pk_add_f32 / v_pk_mul_f32): half the vector issue slots
                    f32x2_t p0 = {__uint_as_float(xf[16 * mr + 4 * v4]), __uint_as_float(xf[16 * mr + 4 * v4 + 1])};
                    f32x2_t p1 = {__uint_as_float(xf[16 * mr + 4 * v4 + 2]), __uint_as_float(xf[16 * mr + 4 * v4 + 3])};
                    if constexpr (AMODE == GS_A_LN) {
                        const f32x2_t mm = {ln_m[mr], ln_m[mr]}, rr = {ln_rs[mr], ln_rs[mr]};
                        p0 = (p0 - mm) * rr;
                        p1 = (p1 - mm) * rr;
                    } else {
                        const f32x2_t ss = {a_scale, a_scale};
                        p0 *= ss;
                        p1 *= ss;
                    }
                    split2v(p0, ah[mr][v4 >> 1][2 * (v4 & 1)], al[mr][v4 >> 1][2 * (v4 & 1)]);
                    split2v(p1, ah[mr][v4 >> 1][2 * (v4 & 1) + 1], al[mr][v4 >> 1][2 * (v4 & 1) + 1]);
                }
            }
        }
        xf = xg;
        // ONE load site per operand: the last two chunks request the next tile's first two (a second load site under its own
        // branch made the compiler merge the two results with register copies — and wait for the loads right there)
        const bool a_next = c + 2 >= nchunks, w_next = c + 1 >= nchunks;
        if (DEEP) {
            // unconditional requests (the last two iterations repeat the last chunk): behind a branch the compiler cannot count on the
            // younger loads being in flight and makes the LDS stores below wait for everything
            load_a(xg, arow, a_next ? nchunks - 1 : c + 2);
        } else if (!a_next || has_next) {
            const char *ap[MR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) ap[mr] = a_next ? row_base(next_tm * TROWS + wr * (MR * 32), mr) : arow[mr];
            load_a(xg, ap, a_next ? c + 2 - nchunks : c + 2);
        }
        if (DEEP) load_w(q_load, n0, a_next ? nchunks - 1 : c + 2);
        else if (!w_next || has_next) load_w(q_load, w_next ? next_n0 : n0, w_next ? 0 : c + 1);
        const uint16_t *wb = &Ws[buf * BUF + (wc * NT * 32 + i) * GS_PITCH + g * 16];
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
            // one piece of the weights at a time; smallest terms first; consecutive MFMAs go to different accumulators
            half8_t wf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *(const half8_t *)(wb + (1 * WROWS + t * 32) * GS_PITCH + sstep * 8);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[mr][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t], *(const half8_t *)ah[mr][sstep], acc[mr][t], 0, 0, 0);     // h l
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *(const half8_t *)(wb + (0 * WROWS + t * 32) * GS_PITCH + sstep * 8);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[mr][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t], *(const half8_t *)al[mr][sstep], acc[mr][t], 0, 0, 0);     // l h
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[mr][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t], *(const half8_t *)ah[mr][sstep], acc[mr][t], 0, 0, 0);     // h h
        }
        if (c + 1 < nchunks || (!DEEP && has_next)) store_w(q_store, buf ^ 1);
        __syncthreads();
    };
    if (DEEP) {
        // (two iterations per trip, no branch between them: a conditional second half made the compiler merge the register sets with
        //  copies at the join — and wait for the loads just issued.  An odd chunk count — the 19 chunks of a 608-column patch matrix —
        //  runs its last chunk after the loop.)
        const int even = nchunks & ~1;
        for (int c = 0; c < even; c += 2) {
            chunk_body(c, qr, qr2);
            chunk_body(c + 1, qr2, qr);
        }
        if (nchunks & 1) chunk_body(nchunks - 1, qr, qr2);
    } else {
        for (int c = 0; c < nchunks; ++c) chunk_body(c, qr, qr);
    }
    par = (par + nchunks) & 1;
    GS_T(3);
    // The weights are the MFMA's FIRST operand (D' = W X^T): lane (i, g) of accumulator tile (mr, t) holds ROW row0 + 32 mr + i of C
    // and, in registers 4 q .. 4 q + 3, the four consecutive columns 32 t + 8 q + 4 g + (0..3) of the wavefront's strip.  A 32 x 32
    // tile is 128 bytes per row either way (32 floats, or the P32 chunk [32 h | 32 l]): the wavefront turns it round in its own
    // 4 KB of LDS (the weight buffers are free after the loop's last barrier; no workgroup barrier here) and moves it as WHOLE
    // 128-byte lines — 8 lanes per row, 8 rows per 16-byte instruction.  How this epilogue got here (scripts/gemm_phase_profile.sh):
    // one 4-byte access per element in accumulator order was 15 us per tile for piece output (VALU: ~25 instructions per element)
    // and 65 us for the residual form (R may alias C, so every load stayed behind the store before it: a memory round trip per
    // element) against 75 us of main loop at K = 768; 16-byte quads straight from the registers were 9 and 34 us (32 lines
    // touched per instruction: the L1's line rate).  The residual rows are requested RD tiles ahead of their use.
    const bool full = row0 + MR * 32 <= M && n0 + WROWS <= N;               // (uniform) no bounds checks inside the tile
    float *C = (float *)Cv;
    uint16_t *Cp = (uint16_t *)Cv;
    constexpr int EP = 136;                                                 // staged row pitch, bytes: conflict-free 8-byte writes
    constexpr int WLB = 32 * EP;                                            // per wavefront: one tile
    constexpr int RD = STATS ? GS_STATS_RESID_DEPTH : GS_RESID_DEPTH;
    char *wl = epi_lds + w * WLB;                                           // beyond the weight buffers: the next tile's chunk 0 may sit there
    const float *bs = bias_lds + n0 + wc * NT * 32;                         // the bias values of the wavefront's column strip
    const int rl = lane >> 3, seg = lane & 7;                               // line phase: row rl + 8 k of the tile, 16-byte segment seg
    auto epilogue = [&](const bool chk) __attribute__((always_inline)) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int64_t mbase = row0 + mr * 32;
            if (CPIECES || (N & 3) == 0) {
                f32x4_t rq[RD][4];
                auto line_ok = [&](int t, int k) {
                    const int n = n0 + (wc * NT + t) * 32 + (CPIECES ? 0 : 4 * seg);
                    return !chk || (mbase + rl + 8 * k < M && n < N);
                };
                auto load_r = [&](int slot, int t) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                        rq[slot][k] = line_ok(t, k) ? *(const f32x4_t *)(R + (mbase + rl + 8 * k) * N + n0 + (wc * NT + t) * 32 + 4 * seg) : z;
                    }
                };
                if (EPI == GS_EPI_RESID) {
#pragma unroll
                    for (int t = 0; t < RD - 1 && t < NT; ++t) load_r(t, t);
                }
                // statistics of the finished rows for the next LayerNorm: shifted sums per 128-column strip (the line phase below
                // holds row rl + 8 k of the tile, four columns per lane: eight lanes per row)
                float st_s[4], st_a[4], st_b[4];
                if constexpr (STATS) {
                    static_assert(NT % 4 == 0 && EPI == GS_EPI_RESID && !CPIECES, "statistics ride on the f32 residual epilogue");
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int64_t m = mbase + rl + 8 * k;
                        st_s[k] = ln_mu[m < M ? m : M - 1];
                        st_a[k] = 0.f; st_b[k] = 0.f;
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (EPI == GS_EPI_RESID && t + RD - 1 < NT) load_r((t + RD - 1) % RD, t + RD - 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4_t b4 = *(const f32x4_t *)&bs[32 * t + 8 * q + 4 * g];
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = fmaf(acc[mr][t][4 * q + e], out_scale, b4[e]);
                            if (EPI == GS_EPI_GELU || EPI == GS_EPI_GELU_ERF) v[e] = gelu_of<EPI>(v[e]);
                        }
                        if (CPIECES) {
                            uint32_t h0, l0, h1, l1;
                            split2(v[0] * c_scale, v[1] * c_scale, h0, l0);
                            split2(v[2] * c_scale, v[3] * c_scale, h1, l1);
                            *(uint2 *)(wl + i * EP + 16 * q + 8 * g) = make_uint2(h0, h1);
                            *(uint2 *)(wl + i * EP + 64 + 16 * q + 8 * g) = make_uint2(l0, l1);
                        } else {
                            *(float2 *)(wl + i * EP + 32 * q + 16 * g) = make_float2(v[0], v[1]);
                            *(float2 *)(wl + i * EP + 32 * q + 16 * g + 8) = make_float2(v[2], v[3]);
                        }
                    }
                    gs_wave_lds_order();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint2 lo = *(const uint2 *)(wl + (rl + 8 * k) * EP + 16 * seg);
                        const uint2 hi = *(const uint2 *)(wl + (rl + 8 * k) * EP + 16 * seg + 8);
                        if (!line_ok(t, k)) continue;
                        const int64_t m = mbase + rl + 8 * k;
                        if (CPIECES) {
                            *(uint4 *)(Cp + m * 2 * N + (int64_t)((n0 >> 5) + wc * NT + t) * 64 + 8 * seg) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        } else {
                            f32x4_t o = {__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(hi.x), __uint_as_float(hi.y)};
                            if (EPI == GS_EPI_RESID) o += rq[t % RD][k];
                            *(f32x4_t *)(C + m * N + n0 + (wc * NT + t) * 32 + 4 * seg) = o;
                            if constexpr (STATS) {
                                const float d0 = o[0] - st_s[k], d1 = o[1] - st_s[k], d2 = o[2] - st_s[k], d3 = o[3] - st_s[k];
                                st_a[k] += (d0 + d1) + (d2 + d3);
                                st_b[k] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                            }
                        }
                    }
                    gs_wave_lds_order();
                    if constexpr (STATS) {
                        if ((t & 3) == 3) {                                 // a 128-column strip is complete: eight lanes per row -> one
                            const int slot = (n0 + (wc * NT + t - 3) * 32) >> 7;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float a = st_a[k], b = st_b[k];
                                a += __shfl_xor(a, 1); b += __shfl_xor(b, 1);
                                a += __shfl_xor(a, 2); b += __shfl_xor(b, 2);
                                a += __shfl_xor(a, 4); b += __shfl_xor(b, 4);
                                const int64_t m = mbase + rl + 8 * k;
                                if (seg == 0 && m < M && n0 + (wc * NT + t) * 32 < N) {
                                    *(float2 *)(ln_stats + m * GS_LN_REC + 2 + 2 * slot) = make_float2(a, b);
                                    if (slot == 0) ln_stats[m * GS_LN_REC] = st_s[k];
                                }
                                st_a[k] = 0.f; st_b[k] = 0.f;
                            }
                        }
                    }
                }
            } else {                    // f32 output with N not a multiple of 4 (no encoder shape): element by element
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t m = mbase + i;
                        const int nl = 32 * t + 8 * (r >> 2) + 4 * g + (r & 3), n = n0 + wc * NT * 32 + nl;
                        if (m >= M || n >= N) continue;
                        float v = fmaf(acc[mr][t][r], out_scale, bs[nl]);
                        if (EPI == GS_EPI_GELU || EPI == GS_EPI_GELU_ERF) v = gelu_of<EPI>(v);
                        if (EPI == GS_EPI_RESID) v += R[m * N + n];
                        C[m * N + n] = v;
                    }
            }
        }
    };
    if (full) epilogue(false);
    else epilogue(true);
#ifdef BSC_GEMM_PROFILE
    GS_T(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GS_T(5);
#endif
}

// Persistent workgroups, one per CU: workgroup p serves XCD p mod 8 (the hardware deals consecutive workgroup ids round the XCDs) and
// walks that XCD's tile list q = p / 8, p / 8 + per_round, ...  XCD k takes the row tiles k, k + 8, ... and all column tiles of one
// before the next (the 786 KB A tile is fetched into that XCD's L2 once).  q_full: the first q_full list entries of every XCD
// are whole tiles; the rest — a last, partly filled round (N = 768: 888 tiles on 256 CUs are 3.47 rounds) — are split into two
// half-width tiles each, so that round costs half a tile's time instead of a whole one.
template <int MR, int NT, int WR, int WC, int EPI, int AMODE, bool CPIECES, bool STATS>
__global__ __launch_bounds__(64 * WR * WC) void k_gemm_split(const void *__restrict__ Av, int64_t M, int K,
                                                             const uint16_t *__restrict__ Wp, int64_t w_plane, int N,
                                                             const float *__restrict__ bias, const float *R, void *Cv,
                                                             float a_scale, float out_scale, float c_scale, int n_tiles_n, int n_tiles_m,
                                                             int64_t q_full, int64_t q_virtual, int per_round, int epi_off,
                                                             float *ln_stats, float *ln_mu, float ln_eps, int k_full, int64_t c_slice)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t Ws[];
    // split-K (few rows): slice blockIdx.y contracts columns [y K, y K + K) of the k_full-long operand rows into its own f32 partial
    // result (c_slice elements apart); k_splitk_finish adds the partials in slice order and applies the epilogue
    if (blockIdx.y) {
        const int64_t k0 = (int64_t)blockIdx.y * K;
        Av = AMODE == GS_A_PIECES ? (const void *)((const uint16_t *)Av + (k0 >> 5) * 64) : (const void *)((const float *)Av + k0);
        Wp += k0;
        Cv = (void *)((float *)Cv + (int64_t)blockIdx.y * c_slice);
    }
    constexpr bool HALF_OK = NT % 2 == 0 && (WC * (NT / 2) * 32) % (16 * WR * WC) == 0;       // the half tile's weight staging plan exists
    constexpr int NH = HALF_OK ? NT / 2 : NT;
    const int xcd = (int)(blockIdx.x & 7);
    typename gs_xf<MR>::type xf, xg;
    int par = 0;
    bool primed = false;
    struct Tile { int64_t tm; int n0; bool half, valid; };
    auto decode = [&](int64_t qv) {
        Tile t;
        t.half = HALF_OK && qv >= q_full;
        const int64_t qt = t.half ? q_full + ((qv - q_full) >> 1) : qv;
        t.tm = (qt / n_tiles_n) * 8 + xcd;
        t.n0 = (int)(qt % n_tiles_n) * (WC * NT * 32) + (t.half ? (int)((qv - q_full) & 1) * (WC * NH * 32) : 0);
        t.valid = qv < q_virtual && t.tm < n_tiles_m;
        return t;
    };
    // the whole (padded) bias row, once per workgroup, behind the epilogue's tile blocks
    char *epi_lds = (char *)Ws + epi_off;
    float *bias_lds = (float *)(epi_lds + (WR * WC) * (32 * 136));
    for (int e = threadIdx.x; e < n_tiles_n * (WC * NT * 32); e += 64 * WR * WC) bias_lds[e] = (bias && e < N) ? bias[e] : 0.f;
    __syncthreads();
    for (int64_t qv = blockIdx.x >> 3; qv < q_virtual; qv += per_round) {
        const Tile cur = decode(qv), nxt = decode(qv + per_round);
        if (!cur.valid) { primed = false; continue; }
        const bool has_next = nxt.valid && nxt.half == cur.half && !(MR == 1 && NT == 1 && WR == 1);      // (the few-rows tile does not chain)
        const int prof_idx = (int)(qv * 8 + xcd);
        if (!cur.half)
            gemm_split_tile<MR, NT, WR, WC, EPI, AMODE, CPIECES, STATS>(Ws, epi_lds, bias_lds, Av, M, K, Wp, w_plane, N, R, Cv, a_scale,
                                                                        out_scale, c_scale, cur.tm, cur.n0, primed, has_next, nxt.tm, nxt.n0,
                                                                        par, xf, xg, prof_idx, ln_stats, ln_mu, ln_eps, k_full);
        else
            gemm_split_tile<MR, NH, WR, WC, EPI, AMODE, CPIECES, STATS>(Ws, epi_lds, bias_lds, Av, M, K, Wp, w_plane, N, R, Cv, a_scale,
                                                                        out_scale, c_scale, cur.tm, cur.n0, primed, has_next, nxt.tm, nxt.n0,
                                                                        par, xf, xg, prof_idx, ln_stats, ln_mu, ln_eps, k_full);
        primed = has_next && K / GS_KC >= 2;
    }
}

// ---- attention at f32 accuracy on fp16 pieces ------------------------------------------------------------------------------------
// softmax(Q K^T / 8) V per (image, head) with every matrix product as three piece products (v_mfma_f32_16x16x32_f16), f32
// softmax.  Layout of the work as in k_attention (encoder_ops.hip): persistent workgroups walk the (image, head) items with the
// next item's K / V loads in flight; K and V^T of the head live in LDS — as h and l planes —; a wavefront takes strips of 16
// queries whose fragments come straight from global memory.  qkv: P32 pieces of the (B T, 3 H 64) matrix the qkv GEMM wrote;
// out: P32 pieces of out_scale * attention output (B T, H 64), what the projection GEMM reads.  P is scaled by 2^8 before it is
// split (probabilities of 1/T would push their l piece into fp16's subnormals), the factor leaves with the row sum.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int N> struct gs_u32vec { typedef uint32_t type __attribute__((ext_vector_type(N))); };

template <int NLD, int NTHR>
__device__ __forceinline__ void atts_issue_loads(typename gs_u32vec<4 * NLD>::type &k8, typename gs_u32vec<4 * NLD>::type &v8,
                                                 const uint16_t *__restrict__ qkv, int item, int T, int H, int tid)
{
    const int64_t tok_stride = (int64_t)2 * 3 * H * 64;             // halfs per token row of the piece matrix
    const int b = item / H, h = item % H;
    const uint16_t *Kp = qkv + (int64_t)b * T * tok_stride + (int64_t)2 * (H * 64 + h * 64);
    const uint16_t *Vp = qkv + (int64_t)b * T * tok_stride + (int64_t)2 * (2 * H * 64 + h * 64);
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int idx = tid + NTHR * r;
        const int t = idx >> 4, pc = idx & 15;                      // 16 sixteen-byte pieces per token: [h 0-31 | l 0-31 | h 32-63 | l 32-63]
        const int tc = t < T ? t : T - 1;
        const uint4 k = *(const uint4 *)(Kp + (int64_t)tc * tok_stride + pc * 8);
        const uint4 v = *(const uint4 *)(Vp + (int64_t)tc * tok_stride + pc * 8);
        k8[4 * r] = k.x; k8[4 * r + 1] = k.y; k8[4 * r + 2] = k.z; k8[4 * r + 3] = k.w;
        v8[4 * r] = v.x; v8[4 * r + 1] = v.y; v8[4 * r + 2] = v.z; v8[4 * r + 3] = v.w;
    }
}

// PF: the next item's K / V pieces are loaded into registers before the strips of the current one (needs the register budget of
// one wavefront per SIMD); otherwise an item's loads are waited for on the spot
template <int NT, int NW, bool PF>
__global__ __launch_bounds__(64 * NW) void k_attention_split(const uint16_t *__restrict__ qkv, int T, int H, int items,
                                                             uint16_t *__restrict__ out, float out_scale, int *work)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t att_lds[];
    __shared__ int s_ticket;
    constexpr int NTHR = 64 * NW;
    constexpr int TP = NT * 16;
    constexpr int KP = 64 + 8;                              // K row pitch (halfs)
    constexpr int VP = 4 * (((TP / 4 - 1) | 7) + 1) + 8;    // V^T row pitch: granules of 4 keys (xor-swizzled) + pad
    constexpr int NLD = (TP * 16 + NTHR - 1) / NTHR;        // 16-byte pieces of K (and of V) per thread
    constexpr int NSTRIP = (NT + NW - 1) / NW;              // strips of 16 queries per wavefront
    uint16_t *sK = att_lds;                                 // [2 pieces][TP * KP]
    uint16_t *sVt = att_lds + 2 * TP * KP;                  // [2 pieces][64 * VP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int64_t tok_stride = (int64_t)2 * 3 * H * 64;
    const float c = 0.125f * 1.44269504088896340736f;               // 1/sqrt(64) * log2(e)
    const int nstrip = (T + 15) >> 4;
    typename gs_u32vec<4 * NLD>::type k8, v8;

    int item = blockIdx.x;
    if (work) {
        if (tid == 0) s_ticket = atomicAdd(&work[0], 1);
        __syncthreads();
        item = s_ticket;
    }
    if (PF) atts_issue_loads<NLD, NTHR>(k8, v8, qkv, item < items ? item : items - 1, T, H, tid);
    int nxt = item;
    for (; item < items; item = nxt) {
        const int b = item / H, h = item % H;
        const uint16_t *Qp = qkv + (int64_t)b * T * tok_stride + (int64_t)2 * h * 64;
        if (!PF) atts_issue_loads<NLD, NTHR>(k8, v8, qkv, item, T, H, tid);
        __syncthreads();                                            // the previous item's strips are done with LDS
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int idx = tid + NTHR * r;
            const int t = idx >> 4, pc = idx & 15;
            if (idx < TP * 16) {
                const bool in = t < T;
                const int piece = (pc >> 2) & 1, ch = (pc >> 3) * 4 + (pc & 3);     // ch: which 8 of the 64 head dims
                const uint32_t keep = in ? 0xffffffffu : 0u;        // padded keys: zero rows (a vector select would go through scratch)
                *(uint4 *)&sK[piece * TP * KP + t * KP + ch * 8] =
                    make_uint4(k8[4 * r] & keep, k8[4 * r + 1] & keep, k8[4 * r + 2] & keep, k8[4 * r + 3] & keep);
                const uint32_t vv[4] = {v8[4 * r] & keep, v8[4 * r + 1] & keep, v8[4 * r + 2] & keep, v8[4 * r + 3] & keep};
                // V^T[d][t] lives at granule (t >> 2) ^ (d >> 3) of row d (bank spread for the 16-bit scatter and the reads)
                const int col = 4 * ((t >> 2) ^ ch) + (t & 3);
                uint16_t *vt = sVt + piece * 64 * VP;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vt[(ch * 8 + 2 * e) * VP + col] = (uint16_t)(vv[e] & 0xffffu);
                    vt[(ch * 8 + 2 * e + 1) * VP + col] = (uint16_t)(vv[e] >> 16);
                }
            }
        }
        if (work && tid == 0) s_ticket = atomicAdd(&work[0], 1);    // everyone has read the previous ticket (barrier above)
        // query fragments of the wavefront's first strip: lane (n, g) = query q0 + n, dims [32 kk + 8 g, + 8) of either piece
        u32x4_t qh[2], ql[2];
        {
            const int q = wave * 16 + n;
            const uint16_t *qr = Qp + (int64_t)(q < T ? q : T - 1) * tok_stride + g * 8;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) { qh[kk] = *(const u32x4_t *)(qr + kk * 64); ql[kk] = *(const u32x4_t *)(qr + kk * 64 + 32); }
        }
        __syncthreads();
        nxt = work ? s_ticket : item + (int)gridDim.x;
        if (PF) atts_issue_loads<NLD, NTHR>(k8, v8, qkv, nxt < items ? nxt : items - 1, T, H, tid);      // in flight during the strips below
#pragma unroll 1
        for (int si = 0; si < NSTRIP; ++si) {
            const int strip = wave + NW * si;
            if (strip >= nstrip) break;
            const int q0 = strip * 16;
            half8_t bqh[2], bql[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) { bqh[kk] = __builtin_bit_cast(half8_t, qh[kk]); bql[kk] = __builtin_bit_cast(half8_t, ql[kk]); }
            if (si + 1 < NSTRIP) {                                  // the next strip's query fragments
                const int q = (strip + NW) * 16 + n;
                const uint16_t *qr = Qp + (int64_t)(q < T ? q : T - 1) * tok_stride + g * 8;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) { qh[kk] = *(const u32x4_t *)(qr + kk * 64); ql[kk] = *(const u32x4_t *)(qr + kk * 64 + 32); }
            }
            f32x4_t acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const half8_t ah = *(const half8_t *)&sK[(t * 16 + n) * KP + kk * 32 + g * 8];
                    const half8_t al = *(const half8_t *)&sK[TP * KP + (t * 16 + n) * KP + kk * 32 + g * 8];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bql[kk], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bqh[kk], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bqh[kk], acc[t], 0, 0, 0);
                }
                // keep the fragment loads of later key tiles behind this tile's MFMAs: left alone, the scheduler hoists all
                // 4 NT of them (16 NT registers) to the top of the strip
                if (t & 1) __builtin_amdgcn_sched_barrier(0);
            }
            // padded keys leave the softmax with -inf (the lane's key limit is made opaque per strip: see k_attention)
            int lim = T - g * 4;
            asm volatile("" : "+v"(lim));
            if (T > 16 * (NT - 2)) {
#pragma unroll
                for (int t = NT - 2; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (t * 16 + i >= lim) acc[t][i] = -INFINITY;
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (t * 16 + i >= lim) acc[t][i] = -INFINITY;
            }
            f32x4_t mv = acc[0];
#pragma unroll
            for (int t = 1; t < NT; ++t) mv = __builtin_elementwise_max(mv, acc[t]);
            float m = fmaxf(fmaxf(mv[0], mv[1]), fmaxf(mv[2], mv[3]));
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            // p = 2^8 exp((s - m) / 8): the 2^8 rides in the exponent
            const f32x4_t cv = {c, c, c, c}, nmc = {8.f - m * c, 8.f - m * c, 8.f - m * c, 8.f - m * c};
            f32x4_t sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4_t e = __builtin_elementwise_fma(acc[t], cv, nmc);
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
                acc[t] = e;
                sv += e;
            }
            float sum = (sv[0] + sv[1]) + (sv[2] + sv[3]);
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            f32x4_t o[4];
            const uint16_t *v0p[4], *v1p[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                const int d = dt * 16 + n;
                v0p[dt] = &sVt[d * VP + 4 * (g ^ (d >> 3))];
                v1p[dt] = &sVt[d * VP + 4 * ((4 + g) ^ (d >> 3))];
            }
#pragma unroll
            for (int ks = 0; ks < NT / 2; ++ks) {
                uint32_t ph0, ph1, ph2, ph3, pl0, pl1, pl2, pl3;
                split2(acc[2 * ks][0], acc[2 * ks][1], ph0, pl0);
                split2(acc[2 * ks][2], acc[2 * ks][3], ph1, pl1);
                split2(acc[2 * ks + 1][0], acc[2 * ks + 1][1], ph2, pl2);
                split2(acc[2 * ks + 1][2], acc[2 * ks + 1][3], ph3, pl3);
                const half8_t ah = __builtin_bit_cast(half8_t, (u32x4_t){ph0, ph1, ph2, ph3});
                const half8_t al = __builtin_bit_cast(half8_t, (u32x4_t){pl0, pl1, pl2, pl3});
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const uint2 h0 = *(const uint2 *)(v0p[dt] + 32 * ks), h1 = *(const uint2 *)(v1p[dt] + 32 * ks);
                    const uint2 l0 = *(const uint2 *)(v0p[dt] + 64 * VP + 32 * ks), l1 = *(const uint2 *)(v1p[dt] + 64 * VP + 32 * ks);
                    const half8_t vh = __builtin_bit_cast(half8_t, (u32x4_t){h0.x, h0.y, h1.x, h1.y});
                    const half8_t vl = __builtin_bit_cast(half8_t, (u32x4_t){l0.x, l0.y, l1.x, l1.y});
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ah, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, al, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ah, o[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // V^T is the FIRST operand: o[dt][i] = 2^8 sum O^T[d = 16 dt + 4 g + i][query q0 + n] — the lane that holds a query's row sum
            // (x 2^8 as well) holds its outputs, four consecutive features per accumulator: 8-byte stores of the h and the l pieces,
            // no shuffles (with the queries along the registers this epilogue was 32 two-byte stores and 4 shuffles per lane and strip)
            const float inv = out_scale / sum;
            const int q = q0 + n;
            if (q < T) {
                uint16_t *dst = out + ((int64_t)b * T + q) * 2 * H * 64 + (int64_t)2 * h * 64 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    uint32_t ph0, pl0, ph1, pl1;
                    split2(o[dt][0] * inv, o[dt][1] * inv, ph0, pl0);
                    split2(o[dt][2] * inv, o[dt][3] * inv, ph1, pl1);
                    uint16_t *d2 = dst + (dt >> 1) * 64 + (dt & 1) * 16;          // chunk 2 h + (dt >> 1): [h 32 | l 32]
                    *(uint2 *)d2 = make_uint2(ph0, ph1);
                    *(uint2 *)(d2 + 32) = make_uint2(pl0, pl1);
                }
            }
        }
    }
    // the last workgroup to leave re-arms the counters for the next launch on this stream
    if (work && tid == 0) {
        __threadfence();
        if (atomicAdd(&work[1], 1) == (int)gridDim.x - 1) {
            work[0] = 0;
            work[1] = 0;
            __threadfence();
        }
    }
}

// The epilogue of a split-K GEMM (few rows): out = epilogue(sum over the S partial results, in slice order, + bias) — the same
// three epilogues and two output forms as k_gemm_split; four consecutive columns per thread (N % 4 == 0).
template <int EPI, bool CPIECES>
__global__ __launch_bounds__(GS_TPB) void k_splitk_finish(const float *__restrict__ part, int S, int64_t c_slice, int64_t M, int N,
                                                          const float *__restrict__ bias, const float *R, void *Cv, float c_scale)
{
    const int64_t e = ((int64_t)blockIdx.x * GS_TPB + threadIdx.x) * 4;
    if (e >= M * N) return;
    const int64_t m = e / N;
    const int n = (int)(e - m * N);
    f32x4_t v = *(const f32x4_t *)(part + e);
    for (int sl = 1; sl < S; ++sl) v += *(const f32x4_t *)(part + sl * c_slice + e);
    if (bias) v += *(const f32x4_t *)(bias + n);
    if (EPI == GS_EPI_GELU || EPI == GS_EPI_GELU_ERF) { v[0] = gelu_of<EPI>(v[0]); v[1] = gelu_of<EPI>(v[1]); v[2] = gelu_of<EPI>(v[2]); v[3] = gelu_of<EPI>(v[3]); }
    if (EPI == GS_EPI_RESID) v += *(const f32x4_t *)(R + e);
    if (CPIECES) {
        uint32_t h0, l0, h1, l1;
        split2(v[0] * c_scale, v[1] * c_scale, h0, l0);
        split2(v[2] * c_scale, v[3] * c_scale, h1, l1);
        uint16_t *o = (uint16_t *)Cv + p32_off(m, N, n);
        *(uint2 *)o = make_uint2(h0, h1);
        *(uint2 *)(o + 32) = make_uint2(l0, l1);
    } else {
        *(f32x4_t *)((float *)Cv + e) = v;
    }
}

// current device ordinal and its CU count (cached per ordinal; kernel attributes and the persistent grids are per device)
static bsc_status gs_device(int *dev, int *n_cu)
{
    static int cus[64] = {0};
    BSC_HIP(hipGetDevice(dev));
    int &c = cus[*dev & 63];
    if (!c) BSC_HIP(hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, *dev));
    *n_cu = c;
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_attention_split(const void *qkv_pieces_dev, int32_t B, int32_t T, int32_t heads, int32_t head_dim,
                                              void *out_pieces_dev, float out_scale, int32_t *work2_dev, void *hip_stream)
{
    if (!qkv_pieces_dev || !out_pieces_dev || B < 1 || T < 1 || heads < 1) return BSC_E_INVALID;
    if (head_dim != 64 || T > 288) {
        bsc_set_error("bsc_enc_attention_split: head_dim 64 and T <= 288 only (got %d, %d)", head_dim, T);
        return BSC_E_INVALID;
    }
    hipStream_t s = (hipStream_t)hip_stream;
    int dev = 0, n_cu = 0;              // one workgroup per CU of the CURRENT device (attributes are per device)
    BSC_TRY(gs_device(&dev, &n_cu));
    const int64_t items = (int64_t)B * heads;
    const dim3 grid((unsigned)(items < n_cu ? items : n_cu));
#define BSC_ATT_LAUNCH(NTV, NWV, PFV)                                                                                                   \
    do {                                                                                                                             \
        constexpr int TPv = NTV * 16, VPv = 4 * (((TPv / 4 - 1) | 7) + 1) + 8;                                                       \
        const size_t lds = (size_t)2 * (TPv * 72 + 64 * VPv) * sizeof(uint16_t);                                                     \
        static uint64_t attr_set = 0;       /* bit per device ordinal */                                                              \
        if (!(attr_set >> (dev & 63) & 1)) {                                                                                         \
            BSC_HIP(hipFuncSetAttribute((const void *)k_attention_split<NTV, NWV, PFV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set |= 1ull << (dev & 63);                                                                                          \
        }                                                                                                                            \
        hipLaunchKernelGGL((k_attention_split<NTV, NWV, PFV>), grid, dim3(64 * NWV), lds, s, (const uint16_t *)qkv_pieces_dev, T, heads,  \
                           (int)items, (uint16_t *)out_pieces_dev, out_scale, (int *)work2_dev);                                     \
    } while (0)
    // 4 wavefronts, one per SIMD, with the 512-register budget: the next item's K / V pieces wait in registers during the strips
    static const int att_mode = getenv("BSC_ATT_SPLIT_MODE") ? atoi(getenv("BSC_ATT_SPLIT_MODE")) : 0;
    if (T <= 224) { if (att_mode == 1) BSC_ATT_LAUNCH(14, 4, true); else BSC_ATT_LAUNCH(14, 7, false); }
    else { if (att_mode == 1) BSC_ATT_LAUNCH(18, 4, true); else BSC_ATT_LAUNCH(18, 8, false); }
#undef BSC_ATT_LAUNCH
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_split_weights(const float *w_dev, int32_t N, int32_t K, float scale, void *pieces_dev, void *hip_stream)
{
    if (!w_dev || !pieces_dev || N <= 0 || K <= 0 || (K & 1)) { bsc_set_error("bsc_enc_split_weights: invalid argument"); return BSC_E_INVALID; }
    const int64_t n_pad = ((int64_t)N + 255) / 256 * 256;
    const int64_t n_el = (int64_t)N * K, n_pad_el = n_pad * K;
    hipLaunchKernelGGL(k_split_weights, dim3((unsigned)((n_pad_el / 2 + GS_TPB - 1) / GS_TPB)), dim3(GS_TPB), 0, (hipStream_t)hip_stream,
                       w_dev, n_el, n_pad_el, scale, (uint16_t *)pieces_dev);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_gemm_split_ws(const void *a_dev, int64_t M, int32_t K, const void *pieces_dev, int32_t N,
                                            const float *bias_dev, const float *resid_dev, void *c_dev, float a_scale, float out_scale,
                                            int32_t epilogue, int32_t a_mode, float c_pieces_scale, float *ln_stats_dev,
                                            float *ln_mu_dev, float ln_eps, void *ws_dev, int64_t ws_bytes, void *hip_stream)
{
    if (!a_dev || !pieces_dev || !c_dev || M <= 0 || N <= 0 || K <= 0 || (K % GS_KC) || epilogue < 0 || epilogue > 3 ||
        a_mode < 0 || a_mode > 2 || (epilogue == GS_EPI_RESID && !resid_dev) || (c_pieces_scale != 0.f && (N % 32))) {
        bsc_set_error("bsc_enc_gemm_split: invalid argument (K must be a multiple of 32; piece output needs N %% 32 == 0)");
        return BSC_E_INVALID;
    }
    if (epilogue == GS_EPI_RESID && c_pieces_scale != 0.f) { bsc_set_error("bsc_enc_gemm_split: the residual epilogue writes f32"); return BSC_E_INVALID; }
    const bool ln = a_mode == GS_A_LN, stats = epilogue == GS_EPI_RESID && ln_stats_dev != nullptr;
    if ((ln || stats) && (!ln_stats_dev || !ln_mu_dev)) {
        bsc_set_error("bsc_enc_gemm_split_ln: the LayerNorm modes need both ln_stats_dev and ln_mu_dev");
        return BSC_E_INVALID;
    }
    if (ln && (K % 128 || K > 1024 || epilogue == GS_EPI_RESID || c_pieces_scale == 0.f)) {
        // (an f32-output form was measured and dropped: fc1 2 497 -> 2 687 us, its epilogue holds the tile twice)
        bsc_set_error("bsc_enc_gemm_split_ln: a_mode 2 reads rows of width K = 128 .. 1024 (multiple of 128) and writes pieces (epilogue 0 / 1)");
        return BSC_E_INVALID;
    }
    if (stats && (N % 128 || N > 1024 || a_mode == GS_A_LN)) {
        bsc_set_error("bsc_enc_gemm_split_ln: row statistics ride on the residual epilogue of a GEMM with N = 128 .. 1024 (multiple of 128)");
        return BSC_E_INVALID;
    }
    // tile shape: 256 x 256 (8 wavefronts x 32 rows x 256 columns), or — for the narrow outputs (N <= 1024: 888 tiles of 256 x 256 on
    // 256 CUs is 3.47 rounds, 13 % of the last one idle) — a smaller tile that balances better; BSC_GEMM_TILE = 1 / 3 / 4 forces one
    static const int tile_env = getenv("BSC_GEMM_TILE") ? atoi(getenv("BSC_GEMM_TILE")) : 0;
    // Few rows (a frame or a handful per call: M = 197 .. ~3 500): 256-row tiles leave N / 256 = 3 .. 12 workgroups on 256 CUs and a
    // forward of ONE frame took 6 ms (3x PyTorch's f32 GEMMs).  Tile 6 = 32 rows x 128 columns, four wavefronts side by side on the
    // columns (each 32 x 32): 60 KB of LDS, two or three workgroups per CU cover each other's chunk latency (a 32-row tile has 6
    // MFMAs per chunk to hide a weight chunk's round trip behind).  While even those tiles do not fill the chip, K is split over
    // grid.y: every slice writes an f32 partial result, k_splitk_finish adds them in slice order (deterministic) and applies the
    // epilogue.  Per flop the small tile moves 8x the weight bytes through LDS, so it is taken only while the big tiles would not fill
    // the chip once.  (LayerNorm-in-the-load and the statistics epilogue exist for the big tile only: callers with few rows use the
    // LayerNorm pass.)
    int dev = 0, n_cu = 0;
    BSC_TRY(gs_device(&dev, &n_cu));
    const int64_t big_tiles = ((M + 255) / 256) * ((N + 255) / 256);
    const bool few_rows = !(ln || stats) && !tile_env && big_tiles <= n_cu && M <= 8192;
    // (32-row tiles re-read the weights once per 32 rows: from ~500 rows on the launch is bound by that L2 traffic — 4.6 TB/s at
    //  1 576 rows — and 128 x 128 tiles, 4 wavefronts x 32 rows x 128 columns, take over, with the same split-K)
    const int tile = (ln || stats) ? 1 : few_rows ? (M <= 512 ? 6 : 3) : tile_env ? tile_env : (N <= 1024 ? BSC_GEMM_NARROW_TILE : 1);
    const int TROWS = tile == 6 ? 32 : tile == 3 ? 128 : 256, TCOLS = tile == 1 ? 256 : 128, NTHR = (tile == 3 || tile == 6) ? 256 : 512;
    const int64_t n_pad = ((int64_t)N + 255) / 256 * 256;
    const int n_tiles_n = (int)(n_pad / TCOLS);
    const int64_t n_tiles_m = (M + TROWS - 1) / TROWS;
    const int64_t groups = (n_tiles_m + 7) / 8;                    // row tiles per XCD
    // a last round that fills at most half of the CUs runs as half-width tiles (tile 1 only; BSC_GEMM_TAIL=0: whole tiles)
    static const int tail_env = getenv("BSC_GEMM_TAIL") ? atoi(getenv("BSC_GEMM_TAIL")) : 1;
    // persistent workgroups: one per CU; the 60 KB few-rows tile two per CU
    const int64_t q_all = groups * n_tiles_n, per_round = ((tile == 6 || tile == 3) ? 2 : 1) * (n_cu / 8 > 0 ? n_cu / 8 : 1);
    const int64_t q_rem = q_all % per_round;
    const int64_t q_full = (tile == 1 && tail_env && q_rem > 0 && 2 * q_rem <= per_round && q_all > per_round) ? q_all - q_rem : q_all;
    const int64_t q_virtual = q_full + 2 * (q_all - q_full);
    const int64_t n_wg = (q_virtual < per_round ? q_virtual : per_round) * 8;     // persistent: one workgroup per CU
    const size_t lds_loop = (size_t)2 * 2 * TCOLS * GS_PITCH * sizeof(uint16_t);
    const size_t lds_epi = (size_t)(NTHR / 64) * (32 * 136);                      // the epilogue's per-wavefront tile blocks
    const size_t lds = lds_loop + lds_epi + (size_t)n_pad * sizeof(float);        // + the bias row
    if (lds > 160 * 1024) { bsc_set_error("bsc_enc_gemm_split: N = %d does not fit the kernel's LDS plan (bias row)", N); return BSC_E_INVALID; }
    hipStream_t s = (hipStream_t)hip_stream;
    const bool ap = a_mode == GS_A_PIECES;
    // split-K of the few-rows tile: the largest slice count that keeps the launch within two workgroups per CU, slices of whole
    // chunks, at least two chunks each
    int S = 1;
    if (few_rows && (N % 4) == 0 && !getenv("BSC_GEMM_NO_SPLITK")) {
        static const int cand[] = {24, 16, 12, 8, 6, 4, 3, 2};
        for (int c : cand)
            if (K % (GS_KC * c) == 0 && K / c >= 2 * GS_KC && n_tiles_m * n_tiles_n * c <= 2 * (int64_t)n_cu) { S = c; break; }
    }
    // f32 partial results live in the caller's workspace (stream-ordered, capture-safe: the library allocates nothing here); a
    // workspace that is absent or too small means fewer slices
    while (S > 1 && (!ws_dev || (int64_t)S * M * N * (int64_t)sizeof(float) > ws_bytes)) {
        int nxt_s = 1;
        static const int cand2[] = {16, 12, 8, 6, 4, 3, 2};
        for (int c : cand2) if (c < S && K % (GS_KC * c) == 0 && K / c >= 2 * GS_KC) { nxt_s = c; break; }
        S = nxt_s;
    }
    float *part = S > 1 ? (float *)ws_dev : nullptr;
    const int k_len = K / S;
    const float *bias_l = S > 1 ? nullptr : bias_dev, *resid_l = S > 1 ? nullptr : resid_dev;
    void *c_l = S > 1 ? (void *)part : c_dev;
    const float cps_l = S > 1 ? 0.f : c_pieces_scale;
    const int epilogue_l = S > 1 ? (int)GS_EPI_BIAS : epilogue;
    const bool cp = cps_l != 0.f;
#define BSC_GEMM_LAUNCH2(MRV, NTV, WRV, WCV, EPIV, AMV, CPV, STV)                                                                    \
    do {                                                                                                                             \
        static uint64_t attr_set = 0;       /* bit per device ordinal */                                                              \
        if (!(attr_set >> (dev & 63) & 1)) {                                                                                         \
            BSC_HIP(hipFuncSetAttribute((const void *)k_gemm_split<MRV, NTV, WRV, WCV, EPIV, AMV, CPV, STV>,                         \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                      \
            attr_set |= 1ull << (dev & 63);                                                                                          \
        }                                                                                                                            \
        hipLaunchKernelGGL((k_gemm_split<MRV, NTV, WRV, WCV, EPIV, AMV, CPV, STV>), dim3((unsigned)n_wg, (unsigned)S), dim3(NTHR),   \
                           lds, s, a_dev, M, k_len, (const uint16_t *)pieces_dev, n_pad * K, N, bias_l, resid_l, c_l, a_scale,       \
                           out_scale, cps_l, n_tiles_n, (int)n_tiles_m, q_full, q_virtual, (int)per_round, (int)lds_loop,             \
                           ln_stats_dev, ln_mu_dev, ln_eps, K, (int64_t)M * N);                                                       \
    } while (0)
#define BSC_GEMM_LAUNCH(EPIV, AMV, CPV)                                                                                              \
    do {                                                                                                                             \
        if (tile == 6) BSC_GEMM_LAUNCH2(1, 1, 1, 4, EPIV, AMV, CPV, false);                                                          \
        else if (tile == 3) BSC_GEMM_LAUNCH2(1, 4, 4, 1, EPIV, AMV, CPV, false);                                                     \
        else if (tile == 4) BSC_GEMM_LAUNCH2(1, 4, 8, 1, EPIV, AMV, CPV, false);                                                     \
        else BSC_GEMM_LAUNCH2(1, 8, 8, 1, EPIV, AMV, CPV, false);                                                                    \
    } while (0)
    if (ln) {                           // LayerNorm folded into the operand load: qkv (bias) and fc1 (bias + GELU), piece output
        if (epilogue == GS_EPI_GELU) BSC_GEMM_LAUNCH2(1, 8, 8, 1, GS_EPI_GELU, GS_A_LN, true, false);
        else if (epilogue == GS_EPI_GELU_ERF) BSC_GEMM_LAUNCH2(1, 8, 8, 1, GS_EPI_GELU_ERF, GS_A_LN, true, false);
        else BSC_GEMM_LAUNCH2(1, 8, 8, 1, GS_EPI_BIAS, GS_A_LN, true, false);
    } else if (stats) {                 // residual epilogue that leaves the row statistics for the next LayerNorm
        if (ap) BSC_GEMM_LAUNCH2(1, 8, 8, 1, GS_EPI_RESID, GS_A_PIECES, false, true);
        else BSC_GEMM_LAUNCH2(1, 8, 8, 1, GS_EPI_RESID, GS_A_F32, false, true);
    } else if (epilogue_l == GS_EPI_GELU) {
        if (ap && cp) BSC_GEMM_LAUNCH(GS_EPI_GELU, GS_A_PIECES, true);
        else if (ap) BSC_GEMM_LAUNCH(GS_EPI_GELU, GS_A_PIECES, false);
        else if (cp) BSC_GEMM_LAUNCH(GS_EPI_GELU, GS_A_F32, true);
        else BSC_GEMM_LAUNCH(GS_EPI_GELU, GS_A_F32, false);
    } else if (epilogue_l == GS_EPI_GELU_ERF) {
        if (ap && cp) BSC_GEMM_LAUNCH(GS_EPI_GELU_ERF, GS_A_PIECES, true);
        else if (ap) BSC_GEMM_LAUNCH(GS_EPI_GELU_ERF, GS_A_PIECES, false);
        else if (cp) BSC_GEMM_LAUNCH(GS_EPI_GELU_ERF, GS_A_F32, true);
        else BSC_GEMM_LAUNCH(GS_EPI_GELU_ERF, GS_A_F32, false);
    } else if (epilogue_l == GS_EPI_RESID) {
        if (cp) { bsc_set_error("bsc_enc_gemm_split: the residual epilogue writes f32"); return BSC_E_INVALID; }
        if (ap) BSC_GEMM_LAUNCH(GS_EPI_RESID, GS_A_PIECES, false);
        else BSC_GEMM_LAUNCH(GS_EPI_RESID, GS_A_F32, false);
    } else {
        if (ap && cp) BSC_GEMM_LAUNCH(GS_EPI_BIAS, GS_A_PIECES, true);
        else if (ap) BSC_GEMM_LAUNCH(GS_EPI_BIAS, GS_A_PIECES, false);
        else if (cp) BSC_GEMM_LAUNCH(GS_EPI_BIAS, GS_A_F32, true);
        else BSC_GEMM_LAUNCH(GS_EPI_BIAS, GS_A_F32, false);
    }
#undef BSC_GEMM_LAUNCH
#undef BSC_GEMM_LAUNCH2
    if (S > 1) {
        const dim3 fgrid((unsigned)(((int64_t)M * N / 4 + GS_TPB - 1) / GS_TPB));
        const bool cpo = c_pieces_scale != 0.f;
#define BSC_FIN(EPIV, CPV)                                                                                                           \
    hipLaunchKernelGGL((k_splitk_finish<EPIV, CPV>), fgrid, dim3(GS_TPB), 0, s, (const float *)part, S, (int64_t)M * N, M, N, bias_dev, \
                       resid_dev, c_dev, c_pieces_scale)
        if (epilogue == GS_EPI_GELU) { if (cpo) BSC_FIN(GS_EPI_GELU, true); else BSC_FIN(GS_EPI_GELU, false); }
        else if (epilogue == GS_EPI_GELU_ERF) { if (cpo) BSC_FIN(GS_EPI_GELU_ERF, true); else BSC_FIN(GS_EPI_GELU_ERF, false); }
        else if (epilogue == GS_EPI_RESID) BSC_FIN(GS_EPI_RESID, false);
        else { if (cpo) BSC_FIN(GS_EPI_BIAS, true); else BSC_FIN(GS_EPI_BIAS, false); }
#undef BSC_FIN
    }
    BSC_HIP(hipGetLastError());
#ifdef BSC_GEMM_PROFILE
    if (getenv("BSC_GEMM_PROFILE_DUMP")) {
        static uint64_t host[GS_PROF_MAX][6];
        BSC_HIP(hipStreamSynchronize(s));
        BSC_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_prof), sizeof(host)));
        const int64_t n = q_virtual * 8 < GS_PROF_MAX ? q_virtual * 8 : GS_PROF_MAX;
        for (int64_t w = 0; w < n; ++w)
            fprintf(stderr, "GP %lld %llx %llu %llu %llu %llu %llu\n", (long long)w, (unsigned long long)host[w][0], (unsigned long long)host[w][1],
                    (unsigned long long)host[w][2], (unsigned long long)host[w][3], (unsigned long long)host[w][4], (unsigned long long)host[w][5]);
    }
#endif
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_gemm_split_ln(const void *a_dev, int64_t M, int32_t K, const void *pieces_dev, int32_t N,
                                            const float *bias_dev, const float *resid_dev, void *c_dev, float a_scale, float out_scale,
                                            int32_t epilogue, int32_t a_mode, float c_pieces_scale, float *ln_stats_dev,
                                            float *ln_mu_dev, float ln_eps, void *hip_stream)
{
    return bsc_enc_gemm_split_ws(a_dev, M, K, pieces_dev, N, bias_dev, resid_dev, c_dev, a_scale, out_scale, epilogue, a_mode,
                                 c_pieces_scale, ln_stats_dev, ln_mu_dev, ln_eps, nullptr, 0, hip_stream);
}

extern "C" bsc_status bsc_enc_gemm_split(const void *a_dev, int64_t M, int32_t K, const void *pieces_dev, int32_t N,
                                         const float *bias_dev, const float *resid_dev, void *c_dev, float a_scale, float out_scale,
                                         int32_t epilogue, int32_t a_pieces, float c_pieces_scale, void *hip_stream)
{
    return bsc_enc_gemm_split_ln(a_dev, M, K, pieces_dev, N, bias_dev, resid_dev, c_dev, a_scale, out_scale, epilogue, a_pieces ? 1 : 0,
                                 c_pieces_scale, nullptr, nullptr, 0.f, hip_stream);
}

extern "C" bsc_status bsc_enc_layernorm_split(const float *x_dev, const float *gamma_dev, const float *beta_dev, int64_t rows,
                                              int32_t width, float eps, float a_scale, void *pieces_dev, void *hip_stream)
{
    if (!x_dev || !gamma_dev || !beta_dev || !pieces_dev || rows <= 0 || (width != 256 && width != 512 && width != 768 && width != 1024)) {
        bsc_set_error("bsc_enc_layernorm_split: width must be 256, 512, 768 or 1024");
        return BSC_E_INVALID;
    }
    const dim3 grid((unsigned)((rows * 64 + GS_TPB - 1) / GS_TPB)), block(GS_TPB);
    hipStream_t s = (hipStream_t)hip_stream;
    uint16_t *out = (uint16_t *)pieces_dev;
    switch (width / 256) {
    case 1: hipLaunchKernelGGL(k_layernorm_split<1>, grid, block, 0, s, x_dev, gamma_dev, beta_dev, rows, eps, a_scale, out); break;
    case 2: hipLaunchKernelGGL(k_layernorm_split<2>, grid, block, 0, s, x_dev, gamma_dev, beta_dev, rows, eps, a_scale, out); break;
    case 3: hipLaunchKernelGGL(k_layernorm_split<3>, grid, block, 0, s, x_dev, gamma_dev, beta_dev, rows, eps, a_scale, out); break;
    default: hipLaunchKernelGGL(k_layernorm_split<4>, grid, block, 0, s, x_dev, gamma_dev, beta_dev, rows, eps, a_scale, out); break;
    }
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_split_rows(const float *x_dev, int64_t M, int32_t K, float a_scale, void *pieces_dev, void *hip_stream)
{
    if (!x_dev || !pieces_dev || M <= 0 || K <= 0 || (K % 32)) { bsc_set_error("bsc_enc_split_rows: K must be a multiple of 32"); return BSC_E_INVALID; }
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)((M * K / 4 + GS_TPB - 1) / GS_TPB)), dim3(GS_TPB), 0, (hipStream_t)hip_stream, x_dev, M,
                       K, a_scale, (uint16_t *)pieces_dev);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_embed_layernorm_f32(const float *patch_dev, const float *cls_dev, const float *reg_dev, const float *pos_dev,
                                                  const float *gamma_dev, const float *beta_dev, int32_t B, int32_t T, int32_t registers,
                                                  int32_t width, float eps, float *u_dev, void *pieces_dev, float *ln_stats_dev,
                                                  float *ln_mu_dev, void *hip_stream)
{
    if (!patch_dev || !cls_dev || !pos_dev || !u_dev || (!pieces_dev && !ln_stats_dev) || (pieces_dev && (!gamma_dev || !beta_dev)) ||
        (ln_stats_dev && !ln_mu_dev) || B < 1 || T < 2 || registers < 0 ||
        registers > T - 2 || (registers > 0 && !reg_dev) || (width != 256 && width != 512 && width != 768 && width != 1024)) {
        bsc_set_error("bsc_enc_embed_layernorm_f32: invalid argument (width must be 256, 512, 768 or 1024)");
        return BSC_E_INVALID;
    }
    const int64_t rows = (int64_t)B * T;
    const dim3 grid((unsigned)((rows * 64 + GS_TPB - 1) / GS_TPB)), block(GS_TPB);
    hipStream_t s = (hipStream_t)hip_stream;
    uint16_t *out = (uint16_t *)pieces_dev;
#define BSC_EMB(V) hipLaunchKernelGGL(k_embed_layernorm_split<V>, grid, block, 0, s, patch_dev, cls_dev, reg_dev, pos_dev, gamma_dev, beta_dev, \
                                      rows, T, registers, eps, u_dev, out, ln_stats_dev, ln_mu_dev)
    switch (width / 256) {
    case 1: BSC_EMB(1); break;
    case 2: BSC_EMB(2); break;
    case 3: BSC_EMB(3); break;
    default: BSC_EMB(4); break;
    }
#undef BSC_EMB
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_final_layernorm_f32(const float *u_dev, const float *gamma_dev, const float *beta_dev, int32_t B, int32_t T,
                                                  int32_t skip, int32_t width, float eps, float *out_dev, void *hip_stream)
{
    if (!u_dev || !gamma_dev || !beta_dev || !out_dev || B < 1 || skip < 0 || skip >= T ||
        (width != 256 && width != 512 && width != 768 && width != 1024)) {
        bsc_set_error("bsc_enc_final_layernorm_f32: invalid argument (width must be 256, 512, 768 or 1024)");
        return BSC_E_INVALID;
    }
    const int64_t rows = (int64_t)B * (T - skip);
    const dim3 grid((unsigned)((rows * 64 + GS_TPB - 1) / GS_TPB)), block(GS_TPB);
    hipStream_t s = (hipStream_t)hip_stream;
#define BSC_FIN(V) hipLaunchKernelGGL(k_final_layernorm_f32<V>, grid, block, 0, s, u_dev, gamma_dev, beta_dev, rows, T, skip, eps, out_dev)
    switch (width / 256) {
    case 1: BSC_FIN(1); break;
    case 2: BSC_FIN(2); break;
    case 3: BSC_FIN(3); break;
    default: BSC_FIN(4); break;
    }
#undef BSC_FIN
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}
