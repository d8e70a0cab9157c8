// localize.hip — voxel_localized (memory_2.py:563-671): query pooling, cosine scan over every stored token,
// per-voxel max, stable top-K in HDF5 group-name order.
//
//   k_pool_query   Gaussian centre-weighted pooling of (B,T,D) query tokens -> (D)          (:591-608)
//   k_cosine       one wavefront per token row: dot(q^, x) and |x|^2 with 16-byte loads and
//                  wave-shuffle reductions; the (rows, D) matrix is streamed from HBM exactly once
//                  for all Q queries of the call (HBM-bound)                  (:656)
//   k_candidates   per voxel: region / floor filters, max over its <= cache_size tokens, 64-bit
//                  rank key (similarity descending, then name order)                         (:624-663)
//   k_block_topk   rounds of per-1024 bitonic selections shrink the candidates to the K smallest rank keys =
//                  the reference's stable-sort top-K (device-wide radix sort only for K > 512)  (:665-667)
#include "bsc_internal.h"

#include <math.h>

#define TPB 256

// ---- HDF5 link-name order of "grid_{r}_{c}_{h}" --------------------------------------------------
// Bytewise string order: digits sort before '_' (0x5f), end-of-string before digits.  Each number is
// written as 6 left-aligned base-11 symbols; non-final fields pad with 10 (after every digit), the final
// field shifts digits to 1..10 and pads with 0.
#define NAME_DIGITS 6
__host__ __device__ static inline u64 name_field(int32_t v, bool last)
{
    int dig[12];
    int n = 0;
    if (v == 0) dig[n++] = 0;
    while (v > 0) { dig[n++] = v % 10; v /= 10; }
    u64 k = 0;
    for (int i = 0; i < NAME_DIGITS; ++i) {
        int sym = (i < n) ? dig[n - 1 - i] + (last ? 1 : 0) : (last ? 0 : 10);
        k = k * 11 + (u64)sym;
    }
    return k;
}
__host__ __device__ static inline u64 name_key(int32_t r, int32_t c, int32_t h)
{
    const u64 B = 1771561ull;   // 11^6
    return (name_field(r, false) * B + name_field(c, false)) * B + name_field(h, true);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// memory_2.py:591-608.  One thread per feature d; weights recomputed in f32 like the torch expression.
__global__ __launch_bounds__(TPB) void k_pool_query(const float *__restrict__ tokens, int B, int T, int D,
                                                    float *__restrict__ out)
{
    const int d = blockIdx.x * TPB + threadIdx.x;
    if (d >= D) return;
    const int g = (int)sqrtf((float)T);
    const float center = (float)((g - 1) / 2.0);
    const float sigma = (float)((g / 2.0) * (g / 2.0));
    float wsum = 0.f;
    for (int t = 0; t < T; ++t) {
        const float xs = (float)(t % g) - center, ys = (float)(t / g) - center;
        wsum += expf(-(xs * xs + ys * ys) / (2 * sigma));
    }
    float total = 0.f;
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) {
            const float xs = (float)(t % g) - center, ys = (float)(t / g) - center;
            const float w = expf(-(xs * xs + ys * ys) / (2 * sigma)) / wsum;
            s += tokens[((int64_t)b * T + t) * D + d] * w;
        }
        total += s;
    }
    out[d] = total / (float)B;
}

// q (Q,D) -> q / max(|q|, 1e-8)   (torch cosine_similarity eps clamp)
__global__ __launch_bounds__(64) void k_normalize_q(const float *__restrict__ q, int D, float *__restrict__ qn)
{
    const int lane = threadIdx.x;
    const float *src = q + (int64_t)blockIdx.x * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += src[k] * src[k];
    s = wave_sum(s);
    const float nrm = fmaxf(sqrtf(s), 1e-8f);
    for (int k = lane; k < D; k += 64) qn[(int64_t)blockIdx.x * D + k] = src[k] / nrm;
}

// Wave-wide sums of QT per-lane values at once (QT = 2, 4, 8): every butterfly step over a lane bit also halves the number
// of values a lane carries (the half of the wavefront with the bit set keeps the upper values), so QT sums cost
// QT - 1 + log2(64 / QT) exchanges instead of 6 QT.  The pairs added at each distance are those of wave_sum, so every
// sum is bit-identical to wave_sum of that value.  Returns the sum of value `q` in the lanes whose bits 5.. select q:
// q = lane >> (6 - log2 QT); all lanes of that group hold it.
template <int QT>
__device__ __forceinline__ float multi_wave_sum(float (&v)[QT], int lane)
{
    int mask = 32;
#pragma unroll
    for (int keep = QT >> 1; keep >= 1; keep >>= 1) {
        const bool hi = (lane & mask) != 0;
#pragma unroll
        for (int i = 0; i < keep; ++i) {
            const float mine = hi ? v[i + keep] : v[i], theirs = hi ? v[i] : v[i + keep];
            v[i] = mine + __shfl_xor(theirs, mask);
        }
        mask >>= 1;
    }
    float r = v[0];
    for (; mask > 0; mask >>= 1) r += __shfl_xor(r, mask);
    return r;
}

// sims[qi * n_rows + row] = dot(q^[qi], x[row]) / max(|x[row]|, 1e-8).
// One wavefront per row (grid-stride), NV float4 per lane; the row is loaded once and reused for QT queries.
template <int NV, int QT>
__global__ __launch_bounds__(TPB) void k_cosine(const float *__restrict__ rows, int64_t n_rows, int D,
                                                const float *__restrict__ qn, int q0, float *__restrict__ sims,
                                                int64_t sims_stride)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    const int D4 = D >> 2;
    float4 qv[QT][NV];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi)
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int v = lane + 64 * t;
            qv[qi][t] = (v < D4) ? ((const float4 *)(qn + (int64_t)(q0 + qi) * D))[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    for (int64_t r = wave; r < n_rows; r += nwaves) {
        const float4 *src = (const float4 *)(rows + r * D);
        float4 xv[NV];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int v = lane + 64 * t;
            xv[t] = (v < D4) ? src[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float n2 = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) n2 += xv[t].x * xv[t].x + xv[t].y * xv[t].y + xv[t].z * xv[t].z + xv[t].w * xv[t].w;
        n2 = wave_sum(n2);
        const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-8f);
        float dsum[QT];
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < NV; ++t)
                d += xv[t].x * qv[qi][t].x + xv[t].y * qv[qi][t].y + xv[t].z * qv[qi][t].z + xv[t].w * qv[qi][t].w;
            dsum[qi] = d;
        }
        if (QT == 1) {
            const float d = wave_sum(dsum[0]);
            if (lane == 0) sims[(int64_t)q0 * sims_stride + r] = d * inv;
        } else {
            // the wavefront's QT groups of 64 / QT lanes end up with one query's sum each; their first lanes store
            const float d = multi_wave_sum<QT>(dsum, lane);
            if ((lane & (64 / QT - 1)) == 0) sims[(int64_t)(q0 + lane / (64 / QT)) * sims_stride + r] = d * inv;
        }
    }
}

// ---- batched queries on the matrix cores -----------------------------------------------------------------------
// S^T tile (32 queries x 32 rows) = Qn (32 x D) . X^T (D x 32) with v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate,
// bit-for-bit an fmaf chain (no TF32 on gfx950), 64 cycles per instruction.  One workgroup = 4 waves = 128 rows;
// each wave owns 32 rows and NT 32-query tiles (16 accumulator registers each).  K is walked in chunks of 32 staged
// through LDS (row stride 33 floats: conflict-free ds_read_b32 for the 32-lane operand groups), double-buffered
// with register prefetch of the next chunk.  The squared row norms fall out of the B operands the lanes already hold.
// Queries sit on the M axis so that the accumulator columns are rows of X: stores are 128-byte row runs per query.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF_KC 32
#ifndef BSC_MFMA_MIN_Q
#define BSC_MFMA_MIN_Q 5     // up to 4 queries: one wavefront per row on the vector ALUs (k_cosine, 5.6 TB/s); from 5 on the
                             // VALU dot products no longer hide behind the row stream (8 queries: 3.6 TB/s) and a zero-padded
                             // 32-query MFMA tile is faster (4.9 TB/s for 5..32 queries over 2^20 x 768)
#endif
#define MF_LD 33
template <int NT>
__global__ __launch_bounds__(TPB) void k_cosine_mfma(const float *__restrict__ X, int64_t n_rows, int D,
                                                     const float *__restrict__ qn, int q0, int q_valid,
                                                     float *__restrict__ sims, int64_t sims_stride)
{
    __shared__ float Xs[2][128 * MF_LD];
    __shared__ float Qs[2][NT * 32 * MF_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int nchunks = D / MF_KC;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float nrm = 0.f;
    float4 xr[4], qr[NT];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + TPB * i, r = idx >> 3, c4 = idx & 7;
            const int64_t gr = row0 + r;
            xr[i] = gr < n_rows ? *(const float4 *)(X + gr * D + c * MF_KC + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int idx = tid + TPB * i, r = idx >> 3, c4 = idx & 7;
            qr[i] = *(const float4 *)(qn + (int64_t)(q0 + r) * D + c * MF_KC + c4 * 4);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + TPB * i, r = idx >> 3, c4 = idx & 7;
            float *d = &Xs[buf][r * MF_LD + c4 * 4];
            d[0] = xr[i].x; d[1] = xr[i].y; d[2] = xr[i].z; d[3] = xr[i].w;
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int idx = tid + TPB * i, r = idx >> 3, c4 = idx & 7;
            float *d = &Qs[buf][r * MF_LD + c4 * 4];
            d[0] = qr[i].x; d[1] = qr[i].y; d[2] = qr[i].z; d[3] = qr[i].w;
        }
    };
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
        const float *xb = &Xs[buf][(w * 32 + (lane & 31)) * MF_LD + (lane >> 5)];
        const float *qb = &Qs[buf][(lane & 31) * MF_LD + (lane >> 5)];
#pragma unroll 4
        for (int kk = 0; kk < MF_KC; kk += 2) {
            const float b = xb[kk];
            nrm = fmaf(b, b, nrm);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float a = qb[t * 32 * MF_LD + kk];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    nrm += __shfl_xor(nrm, 32);
    const float inv = 1.0f / fmaxf(sqrtf(nrm), 1e-8f);
    const int64_t row = row0 + w * 32 + (lane & 31);
    if (row < n_rows) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (q < q_valid) sims[(int64_t)q * sims_stride + row] = acc[t][r] * inv;
            }
    }
}

// ---- batched queries on the bf16 matrix cores at f32 accuracy -----------------------------------------------------------
// gfx950 has no TF32; its f32 MFMA peaks at 157 TFLOP/s, its bf16 MFMA at 2.5 PFLOP/s.  An f32 value is the exact sum of three
// bf16 pieces (8 significand bits each: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), the differences are exact), so a
// product x q is the sum of nine piece products; the six of weight >= 2^-16 (hh, hm, mh, hl, mm, lh) leave a truncation of
// 2^-24 per product — the rounding an f32 multiply has anyway — and accumulate in the matrix core's f32 accumulators like the
// f32 instruction does.  Six bf16 MFMAs at 16x the rate replace one f32 MFMA: 0.375x the matrix time, and the scan moves from
// MFMA-bound (4.1 ms for 256 queries over 2^20 x 768) towards its HBM time.
//   A operand (M axis): 32 queries per tile, NT tiles — the three pieces of the normalised queries come precomputed
//                       (k_split_q) and are staged per 32-wide K chunk through LDS (80-byte row pitch), double-buffered;
//   B operand (N axis): the wavefront's 32 rows straight from global memory in fragment layout — lane (n, g) loads the 16
//                       floats [32 c + 16 g, + 16) of row n (one 128-byte line per row and chunk over the two lane groups),
//                       splits them in registers; sub-step s of a chunk contracts floats [8 s, 8 s + 8) of every lane;
//   v_mfma_f32_32x32x16_bf16: 12 per tile and chunk, interleaved over the NT independent accumulators.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#define BX_KC 32                 // K chunk
#define BX_PITCH 40              // bf16 elements per staged query row (80 bytes: conflict-free 16-byte reads)

__device__ __forceinline__ uint32_t pack_bf16_rne(float lo, float hi)      // v_cvt_pk_bf16_f32
{
    const f32x2_t v = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return *(const uint32_t *)&r;
}

// (a, b) -> packed bf16 pieces h, m, l of both
__device__ __forceinline__ void split3(float a, float b, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = pack_bf16_rne(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pack_bf16_rne(ra, rb);
    l = pack_bf16_rne(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}

// qn (Q, D) f32 -> qp (3, Q, D) bf16 pieces
__global__ __launch_bounds__(TPB) void k_split_q(const float *__restrict__ qn, int64_t n, uint16_t *__restrict__ qp)
{
    const int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t h, m, l;
    split3(qn[i], qn[i + 1], h, m, l);
    *(uint32_t *)(qp + i) = h;
    *(uint32_t *)(qp + n + i) = m;
    *(uint32_t *)(qp + 2 * n + i) = l;
}

// WV wavefronts per workgroup, 32 rows each: WV = 8 puts two wavefronts on every SIMD (256 registers each) that share one staged
// query chunk — one covers the other's LDS / global / barrier waits
// SIX: all six products of weight >= 2^-16 (hh, hm, mh, hl, mm, lh: 2^-24 per product, what an f32 multiply rounds away); otherwise
// the three of weight >= 2^-8 (hh, hm, mh): 2^-16 per product term, ~4e-7 on the cosine of unit vectors after the 1/sqrt(D)
// averaging of D independent terms — inside the 2e-6 the tests hold the scan to and 2 500x inside the north star's 1e-3 —
// at half the matrix work and without the l plane of the queries.
template <int NT, int WV, bool SIX>
__global__ __launch_bounds__(64 * WV) void k_cosine_bf16x3(const float *__restrict__ X, int64_t n_rows, int D,
                                                       const uint16_t *__restrict__ qp, int64_t q_plane, int q0, int q_valid,
                                                       float *__restrict__ sims, int64_t sims_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t Qs[];           // [2][3][NT * 32][BX_PITCH]
    constexpr int QROWS = NT * 32;
    constexpr int BUF = 3 * QROWS * BX_PITCH;
    constexpr int NTHR = 64 * WV;
    constexpr int NLD = 3 * QROWS * 4 / NTHR;                                 // 16-byte pieces of a query chunk per thread
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 31, g = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * (32 * WV);
    const int64_t row = row0 + w * 32 + n;
    const int64_t rowc = row < n_rows ? row : n_rows - 1;                  // clamped: results of padded rows are not stored
    const float *xrow = X + rowc * D + g * 16;
    const int nchunks = D / BX_KC;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float nrm = 0.f;
    // prefetch registers as first-class vector values: an ARRAY that is live across the chunk loop is left in scratch memory
    // by the compiler (12 scratch stores + 12 loads per chunk and lane: 4.5 ms instead of 1.x)
    typedef float xf_t __attribute__((ext_vector_type(16)));
    typedef uint32_t qr_t __attribute__((ext_vector_type(4 * NLD)));
    xf_t xf;
    qr_t qr;
    auto load_x = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = *(const float4 *)(xrow + c * BX_KC + 4 * i);
            xf[4 * i] = v.x; xf[4 * i + 1] = v.y; xf[4 * i + 2] = v.z; xf[4 * i + 3] = v.w;
        }
    };
    auto load_q = [&](int c) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + NTHR * j, p = i / (QROWS * 4), rem = i - p * (QROWS * 4), q = rem >> 2, part = rem & 3;
            const uint4 v = *(const uint4 *)(qp + (int64_t)p * q_plane + (int64_t)(q0 + q) * D + c * BX_KC + part * 8);
            qr[4 * j] = v.x; qr[4 * j + 1] = v.y; qr[4 * j + 2] = v.z; qr[4 * j + 3] = v.w;
        }
    };
    auto store_q = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + NTHR * j, p = i / (QROWS * 4), rem = i - p * (QROWS * 4), q = rem >> 2, part = rem & 3;
            *(uint4 *)&Qs[buf * BUF + (p * QROWS + q) * BX_PITCH + part * 8] = make_uint4(qr[4 * j], qr[4 * j + 1], qr[4 * j + 2], qr[4 * j + 3]);
        }
    };
    load_x(0);
    load_q(0);
    store_q(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        // this chunk's rows -> bf16 pieces (two sub-steps of 8 floats), then the next chunk's loads go in flight
        uint32_t bh[2][4], bm[2][4], bl[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = make_float4(xf[4 * i], xf[4 * i + 1], xf[4 * i + 2], xf[4 * i + 3]);
            nrm = fmaf(v.x, v.x, nrm); nrm = fmaf(v.y, v.y, nrm); nrm = fmaf(v.z, v.z, nrm); nrm = fmaf(v.w, v.w, nrm);
            split3(v.x, v.y, bh[i >> 1][2 * (i & 1)], bm[i >> 1][2 * (i & 1)], bl[i >> 1][2 * (i & 1)]);
            split3(v.z, v.w, bh[i >> 1][2 * (i & 1) + 1], bm[i >> 1][2 * (i & 1) + 1], bl[i >> 1][2 * (i & 1) + 1]);
        }
        if (c + 1 < nchunks) { load_x(c + 1); load_q(c + 1); }
        const uint16_t *qb = &Qs[buf * BUF + n * BX_PITCH + g * 16];
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
            const bf16x8_t xh = *(const bf16x8_t *)bh[sstep], xm = *(const bf16x8_t *)bm[sstep], xl = *(const bf16x8_t *)bl[sstep];
            // one piece of the queries at a time (NT fragments live instead of 3 NT): l is used once, m twice, h three times;
            // smallest terms first; consecutive MFMAs go to different accumulators
            bf16x8_t af[NT];
            if (SIX) {
#pragma unroll
                for (int t = 0; t < NT; ++t) af[t] = *(const bf16x8_t *)(qb + (2 * QROWS + t * 32) * BX_PITCH + sstep * 8);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], xh, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) af[t] = *(const bf16x8_t *)(qb + (1 * QROWS + t * 32) * BX_PITCH + sstep * 8);
            if (SIX) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], xm, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], xh, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) af[t] = *(const bf16x8_t *)(qb + (0 * QROWS + t * 32) * BX_PITCH + sstep * 8);
            if (SIX) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], xl, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], xm, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], xh, acc[t], 0, 0, 0);
        }
        if (c + 1 < nchunks) store_q(buf ^ 1);
        __syncthreads();
    }
    nrm += __shfl_xor(nrm, 32);
    const float inv = 1.0f / fmaxf(sqrtf(nrm), 1e-8f);
    if (row < n_rows) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (q < q_valid) sims[(int64_t)q * sims_stride + row] = acc[t][r] * inv;
            }
    }
}

// ---- the same scan on fp16 pieces: three products instead of six -------------------------------------------------------------
// An f32 value is, to 22 significant bits, the sum of TWO fp16 pieces (11 significand bits each): x q = xh qh + xh ql + xl qh +
// O(2^-22 |x q|) — three v_mfma_f32_32x32x16_f16 per product instead of the six bf16 ones above, the same f32 accumulators
// (encoder_gemm.hip runs the encoder's dense layers this way).  What fp16 lacks is range (5 exponent bits), so both operands are
// brought to a fixed magnitude by exact power-of-two scales that leave through the result:
//   queries  unit vectors (k_normalize_q) x 2^11: |element| <= 2048;
//   rows     x s_r with s_r = the power of two that puts the row's NORM in [2^11, 2^12): every element below 4096, the typical
//            one (norm / sqrt(D)) near 2^7; an element 2^10 below the typical one keeps its l piece to an absolute 2^-25 of the
//            scaled row — far below the 2^-22 relative error of the typical term.
// s_r and 1 / (norm s_r 2^11) per row come from k_row_scale, one pass over the rows that is redone only after the rows changed
// (x->row_scale_dirty: ingest, flush, imports, merges, reset) — a loaded memory that is queried many times pays it once.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#define FX_QSCALE 2048.0f

__device__ __forceinline__ void split2h(f32x2_t v, uint32_t &h, uint32_t &l)      // 2 x v_cvt_pk_f16_f32 around a packed subtract
{
    const half2_t hv = __builtin_convertvector(v, half2_t);
    h = *(const uint32_t *)&hv;
    const half2_t lv = __builtin_convertvector(v - __builtin_convertvector(hv, f32x2_t), half2_t);
    l = *(const uint32_t *)&lv;
}

// one wavefront per row: rs[row] = (s_r, 1 / (max(norm, 1e-8) s_r 2^11))
__global__ __launch_bounds__(TPB) void k_row_scale(const float *__restrict__ X, int64_t n_rows, int D, float2 *__restrict__ rs)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    if (row >= n_rows) return;
    const float4 *xr = (const float4 *)(X + row * D);
    float a = 0.f;
    for (int k = lane; k < D / 4; k += 64) {
        const float4 v = xr[k];
        a = fmaf(v.x, v.x, a); a = fmaf(v.y, v.y, a); a = fmaf(v.z, v.z, a); a = fmaf(v.w, v.w, a);
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) rs[row] = bsc_row_scale_of(a);
}

// qn (Q, D) f32 unit rows -> qp (2, Q, D) fp16 pieces of 2^11 qn
__global__ __launch_bounds__(TPB) void k_split_q_f16(const float *__restrict__ qn, int64_t n, uint16_t *__restrict__ qp)
{
    const int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t h, l;
    const f32x2_t v = {qn[i] * FX_QSCALE, qn[i + 1] * FX_QSCALE};
    split2h(v, h, l);
    *(uint32_t *)(qp + i) = h;
    *(uint32_t *)(qp + n + i) = l;
}

// layout of the work as k_cosine_bf16x3: WV wavefronts x 32 rows, NT tiles of 32 queries, query pieces staged per 32-wide K chunk
// through LDS (two planes), rows straight from global memory in fragment layout and split in registers
template <int NT, int WV>
__global__ __launch_bounds__(64 * WV) void k_cosine_f16x2(const float *__restrict__ X, int64_t n_rows, int D,
                                                           const uint16_t *__restrict__ qp, int64_t q_plane, int q0, int q_valid,
                                                           const float2 *__restrict__ rs, float *__restrict__ sims, int64_t sims_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t Qs[];           // [2][2][NT * 32][BX_PITCH]
    constexpr int QROWS = NT * 32;
    constexpr int BUF = 2 * QROWS * BX_PITCH;
    constexpr int NTHR = 64 * WV;
    constexpr int NLD = 2 * QROWS * 4 / NTHR;                                 // 16-byte pieces of a query chunk per thread
    static_assert(2 * QROWS * 4 % NTHR == 0, "query staging plan");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 31, g = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * (32 * WV);
    const int64_t row = row0 + w * 32 + n;
    const int64_t rowc = row < n_rows ? row : n_rows - 1;                  // clamped: results of padded rows are not stored
    const float *xrow = X + rowc * D + g * 16;
    const float2 rsc = rs[rowc];
    const f32x2_t sc2 = {rsc.x, rsc.x};
    const int nchunks = D / BX_KC;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    typedef float xf_t __attribute__((ext_vector_type(16)));
    typedef uint32_t qr_t __attribute__((ext_vector_type(4 * NLD)));
    xf_t xf;
    qr_t qr;
    auto load_x = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = *(const float4 *)(xrow + c * BX_KC + 4 * i);
            xf[4 * i] = v.x; xf[4 * i + 1] = v.y; xf[4 * i + 2] = v.z; xf[4 * i + 3] = v.w;
        }
    };
    auto load_q = [&](int c) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + NTHR * j, p = i / (QROWS * 4), rem = i - p * (QROWS * 4), q = rem >> 2, part = rem & 3;
            const uint4 v = *(const uint4 *)(qp + (int64_t)p * q_plane + (int64_t)(q0 + q) * D + c * BX_KC + part * 8);
            qr[4 * j] = v.x; qr[4 * j + 1] = v.y; qr[4 * j + 2] = v.z; qr[4 * j + 3] = v.w;
        }
    };
    auto store_q = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + NTHR * j, p = i / (QROWS * 4), rem = i - p * (QROWS * 4), q = rem >> 2, part = rem & 3;
            *(uint4 *)&Qs[buf * BUF + (p * QROWS + q) * BX_PITCH + part * 8] = make_uint4(qr[4 * j], qr[4 * j + 1], qr[4 * j + 2], qr[4 * j + 3]);
        }
    };
    load_x(0);
    load_q(0);
    store_q(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        // this chunk's rows -> fp16 pieces of s_r x (two sub-steps of 8 floats), then the next chunk's loads go in flight
        uint32_t bh[2][4], bl[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2_t p0 = (f32x2_t){xf[4 * i], xf[4 * i + 1]} * sc2, p1 = (f32x2_t){xf[4 * i + 2], xf[4 * i + 3]} * sc2;
            split2h(p0, bh[i >> 1][2 * (i & 1)], bl[i >> 1][2 * (i & 1)]);
            split2h(p1, bh[i >> 1][2 * (i & 1) + 1], bl[i >> 1][2 * (i & 1) + 1]);
        }
        if (c + 1 < nchunks) { load_x(c + 1); load_q(c + 1); }
        const uint16_t *qb = &Qs[buf * BUF + n * BX_PITCH + g * 16];
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
            const half8_t xh = *(const half8_t *)bh[sstep], xl = *(const half8_t *)bl[sstep];
            // one piece of the queries at a time; smallest terms first; consecutive MFMAs go to different accumulators
            half8_t af[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) af[t] = *(const half8_t *)(qb + (1 * QROWS + t * 32) * BX_PITCH + sstep * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], xh, acc[t], 0, 0, 0);      // ql xh
#pragma unroll
            for (int t = 0; t < NT; ++t) af[t] = *(const half8_t *)(qb + (0 * QROWS + t * 32) * BX_PITCH + sstep * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], xl, acc[t], 0, 0, 0);      // qh xl
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t], xh, acc[t], 0, 0, 0);      // qh xh
        }
        if (c + 1 < nchunks) store_q(buf ^ 1);
        __syncthreads();
    }
    if (row < n_rows) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (q < q_valid) sims[(int64_t)q * sims_stride + row] = acc[t][r] * rsc.y;
            }
    }
}

__device__ __forceinline__ uint32_t float_desc_key(float f)
{
    uint32_t u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-orderable
    return ~u;                                          // descending similarity == ascending key
}

// candidate c in [0, n_cand): c < max_id is voxel id c, c == max_id is the grid_0_0_0 group (entry vcap)
__global__ __launch_bounds__(TPB) void k_name_keys(int n_cand, int max_id, int vcap, const int32_t *__restrict__ rgb_pos,
                                                   const int32_t *__restrict__ cnt, u64 *__restrict__ keys,
                                                   uint32_t *__restrict__ vals)
{
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c >= n_cand) return;
    const int e = (c == max_id) ? vcap : c;
    u64 k = ~0ull;
    if (cnt[e] > 0) k = (c == max_id) ? name_key(0, 0, 0) : name_key(rgb_pos[3 * e], rgb_pos[3 * e + 1], rgb_pos[3 * e + 2]);
    keys[c] = k;
    vals[c] = (uint32_t)c;
}

__global__ __launch_bounds__(TPB) void k_name_rank(int n_cand, const uint32_t *__restrict__ sorted_vals,
                                                   uint32_t *__restrict__ rank)
{
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n_cand) return;
    rank[sorted_vals[i]] = (uint32_t)i;
}

struct CandArgs {
    int n_cand, max_id, vcap, cache_size, exact, use_radius, c0, c1, c2, floor_lo, floor_hi;
    double radius2;
    const int32_t *rgb_pos, *cnt, *store_rows;
    const uint32_t *name_rank;
};

// similarity part of the rank key of candidate c for one query (0xffffffff: filtered out / empty):
// region / floor filters, max over the voxel's tokens
__device__ __forceinline__ uint32_t cand_simkey(const CandArgs &a, int c, const float *__restrict__ sims)
{
    if (c >= a.n_cand) return 0xffffffffu;
    const int e = (c == a.max_id) ? a.vcap : c;
    const int m = a.cnt[e];
    if (m <= 0) return 0xffffffffu;
    if (a.use_radius || a.floor_lo <= a.floor_hi) {
        int r = 0, cc = 0, h = 0;
        if (c != a.max_id) { r = a.rgb_pos[3 * e]; cc = a.rgb_pos[3 * e + 1]; h = a.rgb_pos[3 * e + 2]; }
        if (a.use_radius) {   // memory_2.py:624-629 (integer squared distance compared with radius**2)
            const double dx = r - a.c0, dy = cc - a.c1, dz = h - a.c2;
            if (!((dx * dx + dy * dy + dz * dz) <= a.radius2)) return 0xffffffffu;
        }
        if (a.floor_lo <= a.floor_hi && !((a.floor_lo <= h) && (h <= a.floor_hi))) return 0xffffffffu;   // :633-640
    }
    float best = -INFINITY;
    if (a.exact) {
        for (int k = 0; k < m; ++k) best = fmaxf(best, sims[a.store_rows[(int64_t)e * a.cache_size + k]]);   // :661
    } else {
        best = sims[e];
    }
    const uint32_t sk = float_desc_key(best);
    return sk == 0xffffffffu ? 0xfffffffeu : sk;     // keep the all-ones pattern for "no candidate"
}

// full rank key: similarity descending, ties in HDF5 name order
__device__ __forceinline__ u64 cand_key(const CandArgs &a, int c, const float *__restrict__ sims)
{
    const uint32_t sk = cand_simkey(a, c, sims);
    if (sk == 0xffffffffu) return ~0ull;
    return ((u64)sk << 32) | (u64)a.name_rank[c];
}

__global__ __launch_bounds__(TPB) void k_candidates(CandArgs a, const float *__restrict__ sims, u64 *__restrict__ keys,
                                                    uint32_t *__restrict__ vals)
{
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c >= a.n_cand) return;
    keys[c] = cand_key(a, c, sims);
    vals[c] = (uint32_t)c;
}

__global__ __launch_bounds__(TPB) void k_gather_topk(int K, int n_cand, int max_id, int vcap,
                                                     const u64 *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     int64_t in_stride, const int32_t *__restrict__ rgb_pos,
                                                     int32_t *__restrict__ out_pos, float *__restrict__ out_sim)
{
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= K) return;
    keys += (int64_t)blockIdx.y * in_stride;
    vals += (int64_t)blockIdx.y * in_stride;
    out_pos += (int64_t)blockIdx.y * K * 3;
    out_sim += (int64_t)blockIdx.y * K;
    int32_t r = -1, c = -1, h = -1;
    float s = -INFINITY;
    if (i < n_cand && keys[i] != ~0ull) {
        const uint32_t cand = vals[i];
        if ((int)cand != max_id) { r = rgb_pos[3 * cand]; c = rgb_pos[3 * cand + 1]; h = rgb_pos[3 * cand + 2]; }
        else { r = 0; c = 0; h = 0; }
        uint32_t u = ~(uint32_t)(keys[i] >> 32);
        u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
        s = __uint_as_float(u);
    }
    out_pos[3 * i] = r; out_pos[3 * i + 1] = c; out_pos[3 * i + 2] = h;
    out_sim[i] = s;
}

// K smallest rank keys of each 1024-element slice: bitonic sort in LDS (keys are unique: similarity key | name rank).
// Rounds of this kernel shrink n candidates to K without a device-wide sort: n -> ceil(n/1024)*K -> ... -> K.
#define TK_N 1024
#define SEL_CNT_PAD 32          // int32 slots between the survivor counters of consecutive queries (one 128-byte line each)
// Ascending bitonic sort of 1024 (key, value) pairs held 4 per thread (element e = 256 r + tid in slot r).  A compare-exchange
// distance j >= 256 pairs two slots of one thread, j < 64 two lanes of one wavefront (shuffles), only j = 64 and 128 cross
// wavefronts and go through LDS: 7 of the 55 steps, one barrier each (two buffers alternate), instead of a barrier after every
// step of an all-LDS network (round 3: ~40 us per block sort, most of it barriers).
__device__ __forceinline__ void kv_cx(u64 &k, uint32_t &v, const u64 pk, const uint32_t pv, const bool take_min)
{
    const bool swap = take_min ? (pk < k) : (pk > k);
    k = swap ? pk : k;
    v = swap ? pv : v;
}

// value of lane (lane ^ j), j < 64, without the LDS crossbar: DPP quad permutes / row mirrors below 16 (xor 4 = reverse of 8, then
// of each quad; xor 8 = reverse of 16, then of each 8), v_permlane16_swap / v_permlane32_swap above.  As ds_bpermute a 1024-key sort
// issued 2 160 wavefront-wide permutes at ~32 cycles of the CU's LDS pipe each: 33 us per sort and CU, whatever else ran beside it.
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int j, int lane)
{
    if (j == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);           // quad_perm [1,0,3,2]
    if (j == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);           // quad_perm [2,3,0,1]
    if (j == 4) {
        const int t = __builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);                     // row_half_mirror
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x1B, 0xf, 0xf, true);                        // quad_perm [3,2,1,0]
    }
    if (j == 8) {
        const int t = __builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true);                     // row_mirror
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x141, 0xf, 0xf, true);                       // row_half_mirror
    }
    if (j == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);                           // j == 32
    return (lane & 32) ? r[0] : r[1];
}

__device__ __forceinline__ void bitonic_1024_regs(u64 (&key)[4], uint32_t (&val)[4], u64 (*bk)[TK_N], uint32_t (*bv)[TK_N])
{
    const int tid = threadIdx.x;
    int buf = 0;
#pragma unroll
    for (int k = 2; k <= TK_N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 256) {                             // partner slot r ^ (j / 256) of the same thread
                const int dr = j >> 8;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r & dr) continue;
                    const int e = r * 256 + tid;
                    const bool up = (e & k) == 0;       // the lower element of the pair keeps the minimum when the run ascends
                    const bool sw = (key[r] > key[r | dr]) == up;
                    const u64 ka = key[r], kb = key[r | dr];
                    const uint32_t va = val[r], vb = val[r | dr];
                    key[r] = sw ? kb : ka; key[r | dr] = sw ? ka : kb;
                    val[r] = sw ? vb : va; val[r | dr] = sw ? va : vb;
                }
            } else if (j < 64) {                        // partner lane tid ^ j of the same wavefront
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = r * 256 + tid;
                    const u64 pk = (u64)lane_xor((uint32_t)key[r], j, tid) | ((u64)lane_xor((uint32_t)(key[r] >> 32), j, tid) << 32);
                    const uint32_t pv = lane_xor(val[r], j, tid);
                    kv_cx(key[r], val[r], pk, pv, ((e & j) == 0) == ((e & k) == 0));
                }
            } else {                                    // j = 64, 128: another wavefront, through LDS
#pragma unroll
                for (int r = 0; r < 4; ++r) { bk[buf][r * 256 + tid] = key[r]; bv[buf][r * 256 + tid] = val[r]; }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = r * 256 + tid;
                    kv_cx(key[r], val[r], bk[buf][e ^ j], bv[buf][e ^ j], ((e & j) == 0) == ((e & k) == 0));
                }
                buf ^= 1;                               // the next cross-wavefront step writes the other buffer: one barrier per step
            }
        }
    }
}

// first round, fused with the candidate scan: block (b, q) ranks candidates [1024 b, 1024 b + 1024) of query q
__global__ __launch_bounds__(TPB) void k_cand_topk(CandArgs a, const float *__restrict__ sims, int64_t sims_stride, int K,
                                                   u64 *__restrict__ out_keys, uint32_t *__restrict__ out_vals,
                                                   int64_t out_stride, int nb, int nq, int bstride)
{
    __shared__ u64 bk[2][TK_N];
    __shared__ uint32_t bv[2][TK_N];
    for (int wi = blockIdx.x; wi < nb * nq; wi += gridDim.x) {      // persistent: (block, query) items
        const int q = wi / nb, bx = wi - q * nb;
        const float *qs = sims + (int64_t)q * sims_stride;
        u64 key[4];
        uint32_t val[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = bx * bstride * TK_N + r * 256 + (int)threadIdx.x;     // bstride > 1: a sample of blocks spread over the whole map
            key[r] = cand_key(a, c, qs);
            val[r] = (uint32_t)c;
        }
        __syncthreads();                                            // the previous item's partner reads are done
        bitonic_1024_regs(key, val, bk, bv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = r * 256 + (int)threadIdx.x;
            if (e < K) {
                out_keys[(int64_t)q * out_stride + (int64_t)bx * K + e] = key[r];
                out_vals[(int64_t)q * out_stride + (int64_t)bx * K + e] = val[r];
            }
        }
    }
}

__global__ __launch_bounds__(TPB) void k_block_topk(const u64 *__restrict__ in_keys, const uint32_t *__restrict__ in_vals,
                                                    int64_t n, const int32_t *__restrict__ n_per_q, int64_t in_stride, int K,
                                                    u64 *__restrict__ out_keys, uint32_t *__restrict__ out_vals,
                                                    int64_t out_stride, int nb, int nq)
{
    __shared__ u64 bk[2][TK_N];
    __shared__ uint32_t bv[2][TK_N];
    for (int wi = blockIdx.x; wi < nb * nq; wi += gridDim.x) {      // persistent: (block, query) items
        // the query is the FAST index: the non-empty slices of the survivor lists (the first ~7 of every query's 32) are then the
        // first items and spread over all CUs — with the slice as the fast index they sat in every fourth workgroup of an XCD,
        // i.e. on a quarter of its CUs, 16 sorts deep (first survivor round 258 us)
        const int bx = wi / nq, q = wi - bx * nq;
        const int64_t nn = n_per_q ? min((int64_t)n_per_q[q * SEL_CNT_PAD], n) : n;
        const int64_t base = (int64_t)bx * TK_N;
        if (base >= nn) {
            // nothing in this slice (survivor lists are sized for the worst case: ~6 400 of 32 768 slots hold a candidate at K = 100,
            // so 25 of a query's 32 first-round slices are empty): no sort, K empty winners
            for (int e = (int)threadIdx.x; e < K; e += TPB) {
                out_keys[(int64_t)q * out_stride + (int64_t)bx * K + e] = ~0ull;
                out_vals[(int64_t)q * out_stride + (int64_t)bx * K + e] = 0u;
            }
            continue;
        }
        u64 key[4];
        uint32_t val[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t g = base + r * 256 + (int)threadIdx.x;
            key[r] = g < nn ? in_keys[(int64_t)q * in_stride + g] : ~0ull;
            val[r] = g < nn ? in_vals[(int64_t)q * in_stride + g] : 0u;
        }
        __syncthreads();
        bitonic_1024_regs(key, val, bk, bv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = r * 256 + (int)threadIdx.x;
            if (e < K) {
                out_keys[(int64_t)q * out_stride + (int64_t)bx * K + e] = key[r];
                out_vals[(int64_t)q * out_stride + (int64_t)bx * K + e] = val[r];
            }
        }
    }
}

// (persistent 1-D grid: 16 k workgroups walking the (query, chunk) items measured 2x faster than one short-lived
//  workgroup per item, see profiles/README.md)
// threshold filter: the K-th best key of a SAMPLE of the candidates bounds the K-th best of all of them from above,
// so only candidates at least that good can be in the answer; they are appended (unordered) to a short survivor list
#define FILT_PER_BLOCK 4096
#define FILT_G (FILT_PER_BLOCK / (4 * TPB))
// bit c of the map: candidate c (a dense row) holds points.  The filter reads a query's similarities once and, per 16 of them, two
// bytes of this map instead of 64 bytes of counts.
__global__ __launch_bounds__(TPB) void k_valid_bits(const int32_t *__restrict__ cnt, int n, uint32_t *__restrict__ bits)
{
    const int c = blockIdx.x * TPB + threadIdx.x;
    const u64 b = __ballot(c < n && cnt[c < n ? c : 0] > 0);
    if ((threadIdx.x & 63) == 0) *(u64 *)(bits + 2 * (c >> 6)) = b;
}
template <bool FAST>      // FAST: dense map, no region / floor filter — candidate c is row c, validity from the bitmap (zero past max_id)
__global__ __launch_bounds__(TPB) void k_cand_filter(CandArgs a, const float *__restrict__ sims, int64_t sims_stride,
                                                     const u64 *__restrict__ thr_keys, int K, int cap,
                                                     u64 *__restrict__ out_keys, uint32_t *__restrict__ out_vals,
                                                     int32_t *__restrict__ counts, int nbf, int nq, const uint32_t *__restrict__ valid)
{
  for (int wi = blockIdx.x; wi < nbf * nq; wi += gridDim.x) {
    const int q = wi / nbf, bx = wi - q * nbf;
    const u64 thr = thr_keys[q];
    const uint32_t thr_hi = (uint32_t)(thr >> 32);
    const float *qs = sims + (int64_t)q * sims_stride;
    const int lane = threadIdx.x & 63;
    uint32_t sk[FILT_G][4];
    // phase 1: every load of the block's 4096 candidates is in flight before anything is consumed
#pragma unroll
    for (int g = 0; g < FILT_G; ++g) {
        const int c0 = bx * FILT_PER_BLOCK + g * 4 * TPB + threadIdx.x * 4;
        if constexpr (FAST) {
            // (a group of four that reaches past max_id — the map's tail, max_id rounded up to 4 — reads nothing: its bits count as zero;
            //  inside the last word the bits past max_id ARE zero)
            const bool in = c0 < a.max_id;
            const float4 sv = in ? *(const float4 *)(qs + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const uint32_t vb = in ? valid[c0 >> 5] >> (c0 & 31) : 0u;  // c0 is a multiple of 4: its four bits sit in one word
            sk[g][0] = (vb & 1u) ? float_desc_key(sv.x) : 0xffffffffu;
            sk[g][1] = (vb & 2u) ? float_desc_key(sv.y) : 0xffffffffu;
            sk[g][2] = (vb & 4u) ? float_desc_key(sv.z) : 0xffffffffu;
            sk[g][3] = (vb & 8u) ? float_desc_key(sv.w) : 0xffffffffu;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) sk[g][k] = cand_simkey(a, c0 + k, qs);
        }
    }
    // phase 2: which of the lane's 16 candidates survive (the name rank breaks ties with the threshold's similarity bits)
    uint32_t keepbits = 0u;
#pragma unroll
    for (int g = 0; g < FILT_G; ++g) {
        const int c0 = bx * FILT_PER_BLOCK + g * 4 * TPB + threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool keep = sk[g][k] != 0xffffffffu && sk[g][k] <= thr_hi;
            if (keep && sk[g][k] == thr_hi) keep = (((u64)sk[g][k] << 32) | (u64)a.name_rank[c0 + k]) <= thr;
            keepbits |= keep ? (1u << (g * 4 + k)) : 0u;
        }
    }
    // phase 3: ONE atomic per wavefront with survivors (the counters of the queries sit a cache line apart: with one counter
    // per survivor on 16 shared lines this kernel spent 1.1 ms of its 1.2 ms queueing at the L2 atomic units)
    const int mine = __popc(keepbits);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const int total = __shfl(incl, 63);
    // ONE atomic per work item (4096 candidates): the four wavefronts' totals meet in LDS.  Atomics on one address are served one
    // at a time by the L2 (~0.4 us each with the returned value): with an atomic per wavefront a query's counter took 1024 of them in
    // a row — that, not the 1.07 GB it reads, was this kernel's 0.46 ms
    __shared__ int s_tot[TPB / 64], s_base;
    const int wv = threadIdx.x >> 6;
    __syncthreads();                                    // the previous item's readers of s_tot / s_base are done
    if (lane == 0) s_tot[wv] = total;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) { const int t = s_tot[w]; before += w < wv ? t : 0; all += t; }
    if (all == 0) continue;                             // (uniform for the workgroup)
    if (threadIdx.x == 0) s_base = atomicAdd(&counts[q * SEL_CNT_PAD], all);
    __syncthreads();
    int base = s_base + before + incl - mine;
#pragma unroll
    for (int g = 0; g < FILT_G; ++g) {
        const int c0 = bx * FILT_PER_BLOCK + g * 4 * TPB + threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (keepbits & (1u << (g * 4 + k))) {
                if (base < cap) {
                    out_keys[(int64_t)q * cap + base] = ((u64)sk[g][k] << 32) | (u64)a.name_rank[c0 + k];
                    out_vals[(int64_t)q * cap + base] = (uint32_t)(c0 + k);
                }
                ++base;
            }
        }
    }
  }
}

// thresholds of the sample selection: K-th key of every query's winners -> thr[q]; also clears the survivor counters
__global__ void k_sel_thresholds(const u64 *__restrict__ win_keys, int64_t stride, int K, int nq, u64 *__restrict__ thr,
                                 int32_t *__restrict__ counts)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    thr[q] = win_keys[(int64_t)q * stride + (K - 1)];
    counts[q * SEL_CNT_PAD] = 0;
}

// the filter is exact unless a survivor list overflowed or the sample held fewer than K valid candidates
__global__ void k_sel_check(const u64 *__restrict__ thr, const int32_t *__restrict__ counts, int nq, int cap, int32_t *flag)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    if (counts[q * SEL_CNT_PAD] > cap || thr[q] == ~0ull) *flag = 1;
#ifdef BSC_SEL_DEBUG
    if (q < 4 || q == nq - 1) printf("query %d: %d survivors (cap %d)\n", q, counts[q * SEL_CNT_PAD], cap);
#endif
}

static bsc_status grow_dev(void **p, int64_t *cap, int64_t need_bytes)
{
    if (*cap >= need_bytes) return BSC_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    hipError_t e = hipMalloc(p, (size_t)need_bytes);
    if (e != hipSuccess) { bsc_set_error("bsc_localize scratch (%lld bytes): %s", (long long)need_bytes, hipGetErrorString(e)); return BSC_E_HIP; }
    *cap = need_bytes;
    return BSC_OK;
}

// K <= 512, all queries at once, no device-wide sort:
//   small candidate sets: rounds of 1024-key bitonic selections (round 1 fused with the candidate scan);
//   large sets: selection over a sample of 16 blocks spread over the map gives a per-query threshold, one streaming filter
//   pass keeps the few candidates that beat it, rounds over the survivors finish.  *overflow is set when the host must fall
//   back.  Sample size: nbs x nq bitonic selections for the sample + (K n_cand / (1024 nbs)) / 1024 x nq for the survivors —
//   at 2^20 candidates and K = 100, 16 blocks (~6400 survivors per query) cost a third of 64 blocks (~1600 survivors).
#define SEL_SAMPLE_BLOCKS 16
#define SEL_SURVIVOR_CAP 32768
static bsc_status bitonic_rounds(bsc_ctx *x, int nq, int64_t n, const int32_t *n_per_q, int K, int64_t stride, int *cur)
{
    bool first = true;
    while (n > K || first) {
        const int64_t nb = (n + TK_N - 1) / TK_N;
        const int64_t items = nb * nq;
        hipLaunchKernelGGL(k_block_topk, dim3((unsigned)(items < 4096 ? items : 4096)), dim3(TPB), 0, x->stream,
                           x->l_sel_key[*cur], x->l_sel_val[*cur], n, first ? n_per_q : (const int32_t *)nullptr, stride, K,
                           x->l_sel_key[*cur ^ 1], x->l_sel_val[*cur ^ 1], stride, (int)nb, nq);
        *cur ^= 1;
        n = nb * K;
        first = false;
        if (nb == 1) break;
    }
    return BSC_OK;
}

static bsc_status select_topk_batched(bsc_ctx *x, const CandArgs &ca, int nq, int64_t sims_stride, int K, bool allow_filter,
                                      u64 **win_keys, uint32_t **win_vals, int64_t *win_stride, bool *filtered)
{
    const int64_t nb1 = ((int64_t)ca.n_cand + TK_N - 1) / TK_N;
    // the K-th best of a sample of nbs blocks lets about K nb1 / nbs candidates through the filter: enough sample blocks that this
    // stays at a quarter of the survivor lists (K = 100 over 2^20 candidates: 16 blocks, 6 400 survivors; K = 512: 64 blocks;
    // 3 M candidates, K = 100: 36 blocks) — with the fixed 16 of round 3 every K > ~180 overflowed the lists on every call and
    // paid the sample, the filter AND the unfiltered fallback
    int64_t nbs_want = (4 * (int64_t)K * nb1 + SEL_SURVIVOR_CAP - 1) / SEL_SURVIVOR_CAP;
    if (nbs_want < SEL_SAMPLE_BLOCKS) nbs_want = SEL_SAMPLE_BLOCKS;
    const bool use_filter = allow_filter && nb1 > 4 * nbs_want;
    const int64_t nbs = use_filter ? nbs_want : nb1;
    int64_t stride = nbs * K;
    if (use_filter && stride < SEL_SURVIVOR_CAP) stride = SEL_SURVIVOR_CAP;
    BSC_TRY(grow_dev((void **)&x->l_sel_key[0], &x->l_sel_cap[0], sizeof(u64) * stride * nq));
    BSC_TRY(grow_dev((void **)&x->l_sel_key[1], &x->l_sel_cap[1], sizeof(u64) * stride * nq));
    BSC_TRY(grow_dev((void **)&x->l_sel_val[0], &x->l_sel_cap[2], sizeof(uint32_t) * stride * nq));
    BSC_TRY(grow_dev((void **)&x->l_sel_val[1], &x->l_sel_cap[3], sizeof(uint32_t) * stride * nq));
    BSC_TRY(grow_dev((void **)&x->l_sel_thr, &x->l_sel_cap[4], sizeof(u64) * (int64_t)(nq + 1)));
    BSC_TRY(grow_dev((void **)&x->l_sel_cnt, &x->l_sel_cap[5], sizeof(int32_t) * nq * SEL_CNT_PAD));
    int cur = 0;
    // round 1 over the sample (or over everything), fused with the candidate keys
    hipLaunchKernelGGL(k_cand_topk, dim3((unsigned)(nbs * nq < 4096 ? nbs * nq : 4096)), dim3(TPB), 0, x->stream, ca, x->l_sims,
                       sims_stride, K, x->l_sel_key[0], x->l_sel_val[0], stride, (int)nbs, nq, use_filter ? (int)(nb1 / nbs) : 1);
    if (nbs > 1) BSC_TRY(bitonic_rounds(x, nq, nbs * K, nullptr, K, stride, &cur));
    *filtered = false;
    if (use_filter) {
        // per-query threshold = K-th key of the sample winners (kept aside: the selection buffers are reused)
        hipLaunchKernelGGL(k_sel_thresholds, dim3((nq + 255) / 256), dim3(256), 0, x->stream, x->l_sel_key[cur], stride, K, nq,
                           x->l_sel_thr, x->l_sel_cnt);
        cur = 0;
        const unsigned nbf = (unsigned)(((int64_t)ca.n_cand + FILT_PER_BLOCK - 1) / FILT_PER_BLOCK);
        if (!ca.exact) {
            // every wavefront of the ceil(max_id / TPB) workgroups stores one 64-bit word: TPB / 32 uint32 per workgroup.  Grown with
            // slack (a map that gains a few voxels between two queries keeps its bitmap allocation)
            const int64_t words = (TPB / 32) * (((int64_t)ca.max_id + TPB - 1) / TPB) + 2;
            if ((int64_t)sizeof(uint32_t) * words > x->l_sel_cap[6])
                BSC_TRY(grow_dev((void **)&x->l_valid, &x->l_sel_cap[6], sizeof(uint32_t) * (words + words / 4 + 1024)));
            if (ca.max_id > 0)
                hipLaunchKernelGGL(k_valid_bits, dim3((unsigned)((ca.max_id + TPB - 1) / TPB)), dim3(TPB), 0, x->stream, ca.cnt, ca.max_id, x->l_valid);
        }
        // dense maps without region / floor filter: candidate c is row c, similarities stream as 16-byte loads
        const bool fast = !ca.exact && !ca.use_radius && !(ca.floor_lo <= ca.floor_hi) && (sims_stride % 4 == 0);
        const dim3 fgrid(nbf * (unsigned)nq < 16384u ? nbf * (unsigned)nq : 16384u);
        if (fast)
            hipLaunchKernelGGL(k_cand_filter<true>, fgrid, dim3(TPB), 0, x->stream, ca, x->l_sims, sims_stride, x->l_sel_thr, K,
                               SEL_SURVIVOR_CAP, x->l_sel_key[0], x->l_sel_val[0], x->l_sel_cnt, (int)nbf, nq, (const uint32_t *)x->l_valid);
        else
            hipLaunchKernelGGL(k_cand_filter<false>, fgrid, dim3(TPB), 0, x->stream, ca, x->l_sims, sims_stride, x->l_sel_thr, K,
                               SEL_SURVIVOR_CAP, x->l_sel_key[0], x->l_sel_val[0], x->l_sel_cnt, (int)nbf, nq, (const uint32_t *)x->l_valid);
        // survivors of query q sit at [q * CAP, q * CAP + count); the rounds use stride CAP for them
        BSC_TRY(bitonic_rounds(x, nq, SEL_SURVIVOR_CAP, x->l_sel_cnt, K, SEL_SURVIVOR_CAP, &cur));
        stride = SEL_SURVIVOR_CAP;
        *filtered = true;
    }
    BSC_HIP(hipGetLastError());
    *win_keys = x->l_sel_key[cur];
    *win_vals = x->l_sel_val[cur];
    *win_stride = stride;
    return BSC_OK;
}

bsc_status pool_query_impl(bsc_ctx *x, const float *tokens, int32_t B, int32_t T, int32_t D, float *out)
{
    hipLaunchKernelGGL(k_pool_query, dim3((D + TPB - 1) / TPB), dim3(TPB), 0, x->stream, tokens, B, T, D, out);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// similarity rows of consecutive queries must not sit a power of two apart: a 2^22-byte stride puts the rows that
// are read concurrently (8+ queries in flight) on the same HBM channel / bank set and the candidate passes drop to
// 0.4 TB/s.  Rows are padded to a multiple of 64 floats plus an odd number of 256-byte units.
int64_t sims_row_stride(int64_t n_rows) { return ((n_rows + 63) & ~(int64_t)63) + 64 * 33; }

template <int QT>
static void launch_cosine(bsc_ctx *x, const float *rows, int64_t n_rows, int q0)
{
    const int D = x->c.token_dim;
    const int nv = (D / 4 + 63) / 64;
    int64_t blocks = (n_rows * 64 + TPB - 1) / TPB;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(TPB);
#define LC(NV) hipLaunchKernelGGL((k_cosine<NV, QT>), grid, block, 0, x->stream, rows, n_rows, D, x->l_q, q0, x->l_sims, sims_row_stride(n_rows))
    if (nv <= 1) LC(1);
    else if (nv == 2) LC(2);
    else if (nv == 3) LC(3);
    else if (nv == 4) LC(4);
    else LC(8);
#undef LC
}

// What the first query after a change of the stored voxels / rows would have to do first — the name ranks (HDF5 iteration order)
// and, for the batched fp16-piece scan, every row's operand scale and inverse norm — done by the one that changed them.  Called at
// the end of the import entry points (round 6: a loaded memory's first batch of 256 queries took 3.1 ms instead of 2.2; an ingest
// leaves both current on its own).  A failure here is not an error of the import: the flags stay set and the query does the work.
void localize_prepare(bsc_ctx *x)
{
    if (read_scalars(x) != BSC_OK) return;
    hipStream_t s = x->stream;
    const bool exact = x->c.mode == BSC_MODE_EXACT;
    const int max_id = (int)x->hscal[DS_MAX_ID], vcap = x->c.voxel_capacity, n_cand = max_id + 1, D = x->c.token_dim;
    const int64_t n_rows = exact ? x->hscal[DS_POOL_N] : max_id;
    const float *rows = exact ? x->pool : x->acc;
    const int32_t *cnt = exact ? x->store_cnt : x->acnt;
    const dim3 block(TPB), cgrid((n_cand + TPB - 1) / TPB);
    if (x->names_dirty && max_id > 0) {
        hipLaunchKernelGGL(k_name_keys, cgrid, block, 0, s, n_cand, max_id, vcap, x->rgb_pos, cnt, x->l_key_a, x->l_val_a);
        if (prim_sort_pairs(x, x->l_key_a, x->l_key_b, x->l_val_a, x->l_val_b, (size_t)n_cand, 0, 64) != BSC_OK) return;
        hipLaunchKernelGGL(k_name_rank, cgrid, block, 0, s, n_cand, x->l_val_b, x->l_name_rank);
        x->names_dirty = false;
    }
    if (x->row_scale_dirty && n_rows > 0 && D % MF_KC == 0) {
        if (x->l_rscale_cap < (int64_t)sizeof(float2) * n_rows &&
            grow_dev((void **)&x->l_rscale, &x->l_rscale_cap, sizeof(float2) * (n_rows + n_rows / 8 + 1024)) != BSC_OK) return;
        hipLaunchKernelGGL(k_row_scale, dim3((unsigned)((n_rows * 64 + TPB - 1) / TPB)), block, 0, s, rows, n_rows, D, x->l_rscale);
        x->row_scale_dirty = false;
    }
    (void)hipGetLastError();
}

bsc_status localize_impl(bsc_ctx *x, const float *q_dev, int32_t nq, int32_t K, double radius, const int32_t *curr,
                         int32_t floor_lo, int32_t floor_hi, int32_t *out_pos, float *out_sim, int32_t *out_count)
{
    hipStream_t s = x->stream;
    const int D = x->c.token_dim;
    const bool exact = x->c.mode == BSC_MODE_EXACT;
    BSC_TRY(read_scalars(x));
    const int max_id = (int)x->hscal[DS_MAX_ID];
    const int vcap = x->c.voxel_capacity;
    const int n_cand = max_id + 1;
    const int64_t n_rows = exact ? x->hscal[DS_POOL_N] : max_id;
    const float *rows = exact ? x->pool : x->acc;
    const int32_t *cnt = exact ? x->store_cnt : x->acnt;
    const dim3 block(TPB), cgrid((n_cand + TPB - 1) / TPB);
    if (K > 4096 || K < 1 || nq < 1 || nq > 1024) {
        bsc_set_error("bsc_localize: K=%d (1..4096), n_queries=%d (1..1024)", K, nq);
        return BSC_E_INVALID;
    }
    // name ranks (HDF5 iteration order) are rebuilt only when the set of stored voxels changed
    if (x->names_dirty) {
        hipLaunchKernelGGL(k_name_keys, cgrid, block, 0, s, n_cand, max_id, vcap, x->rgb_pos, cnt, x->l_key_a, x->l_val_a);
        BSC_TRY(prim_sort_pairs(x, x->l_key_a, x->l_key_b, x->l_val_a, x->l_val_b, (size_t)n_cand, 0, 64));
        hipLaunchKernelGGL(k_name_rank, cgrid, block, 0, s, n_cand, x->l_val_b, x->l_name_rank);
        x->names_dirty = false;
    }
    {   // the MFMA path reads whole query tiles: zero-pad up to the next multiple of 256 (l_q holds 1024 rows)
        const int padded = ((nq + 255) / 256) * 256;
        BSC_HIP(hipMemsetAsync(x->l_q, 0, sizeof(float) * (size_t)(padded > 1024 ? 1024 : padded) * D, s));
    }
    hipLaunchKernelGGL(k_normalize_q, dim3(nq), dim3(64), 0, s, q_dev, D, x->l_q);
    // dense acnt has no slot for the grid_0_0_0 group: k_candidates reads cnt[vcap]; acnt is allocated vcap+1
    int done = 0;
    stat_begin(x, 1);
    const int64_t sstride = sims_row_stride(n_rows);
    int passes = 0;                          // times the row matrix is streamed
    if (nq >= BSC_MFMA_MIN_Q && D % MF_KC == 0 && n_rows > 0) {
        // batched queries on the matrix cores, 32-query tiles (l_q is zero-padded to a multiple of 256 rows): more than 32
        // 64 queries -> bf16 pieces at f32 accuracy (k_cosine_bf16x3); up to 64 -> the f32 MFMA, HBM-bound at that size anyway
        const dim3 mgrid((unsigned)((n_rows + 127) / 128));
        static const bool f32_only = getenv("BSC_COSINE_F32") != nullptr;
        // six piece products (f32 accuracy, the default) or BSC_COSINE_PIECES=3 (hh, hm, mh): 0.7x the scan time, scores within
        // ~4e-6 instead of 3e-7 — enough for the north star's 1e-3, not for the 2e-6 the fp64 parity tests ask; read per call
        const char *pcs = getenv("BSC_COSINE_PIECES");
        const bool six = !(pcs && atoi(pcs) == 3);
        static const bool wv8 = getenv("BSC_COSINE_WV4") == nullptr;                     // A/B: 8 (default) or 4 wavefronts per workgroup                 // A/B: the round-2 f32 MFMA scan throughout
        const int padded = ((nq + 255) / 256) * 256 > 1024 ? 1024 : ((nq + 255) / 256) * 256;
        const bool bf16_pieces = getenv("BSC_COSINE_BF16") != nullptr;                 // A/B (read per call): the round-3/4 six-product bf16 scan
        if (!f32_only && nq > 64 && !bf16_pieces) {
            // fp16 pieces, three products (round 5): per-row scales / inverse norms cached until the rows change
            const int64_t nel = (int64_t)padded * D;
            if (x->row_scale_dirty || !x->l_rscale || x->l_rscale_cap < (int64_t)sizeof(float2) * n_rows) {
                if (x->l_rscale_cap < (int64_t)sizeof(float2) * n_rows)
                    BSC_TRY(grow_dev((void **)&x->l_rscale, &x->l_rscale_cap, sizeof(float2) * (n_rows + n_rows / 8 + 1024)));
                hipLaunchKernelGGL(k_row_scale, dim3((unsigned)((n_rows * 64 + TPB - 1) / TPB)), block, 0, s, rows, n_rows, D, x->l_rscale);
                x->row_scale_dirty = false;
            }
            hipLaunchKernelGGL(k_split_q_f16, dim3((unsigned)((nel / 2 + TPB - 1) / TPB)), block, 0, s, x->l_q, nel, x->l_qp);
            while (done < nq) {
                const int left = nq - done;
                ++passes;
#define FX_LAUNCH(NTV, ADV)                                                                                                         \
    do {                                                                                                                            \
        const size_t lds = (size_t)2 * 2 * (NTV * 32) * BX_PITCH * sizeof(uint16_t);                                               \
        (void)hipFuncSetAttribute((const void *)k_cosine_f16x2<NTV, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
        hipLaunchKernelGGL((k_cosine_f16x2<NTV, 8>), dim3((unsigned)((n_rows + 255) / 256)), dim3(512), lds, s, rows, n_rows, D,    \
                           (const uint16_t *)x->l_qp, nel, done, nq, (const float2 *)x->l_rscale, x->l_sims, sstride);              \
        done += ADV;                                                                                                                \
    } while (0)
                if (left > 128) FX_LAUNCH(8, 256);
                else if (left > 64) FX_LAUNCH(4, 128);
                else { --passes; break; }                    // the remainder (<= 64 queries) goes to the f32 MFMA below
#undef FX_LAUNCH
            }
        } else if (!f32_only && nq > 64) {         // measured over 2^20 x 768: 33..64 queries 1.17-1.28 ms against 1.09 ms on the f32 MFMA (HBM-bound either way)
            const int64_t nel = (int64_t)padded * D;
            hipLaunchKernelGGL(k_split_q, dim3((unsigned)((nel / 2 + TPB - 1) / TPB)), block, 0, s, x->l_q, nel, x->l_qp);
            // k_split_q wrote planes nel apart; the scan indexes them with the same stride
            while (done < nq) {
                const int left = nq - done;
                ++passes;
#define BX_LAUNCH2(NTV, WVV, SIXV, GRID, BLOCK)                                                                                     \
    do {                                                                                                                            \
        (void)hipFuncSetAttribute((const void *)k_cosine_bf16x3<NTV, WVV, SIXV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_cosine_bf16x3<NTV, WVV, SIXV>), GRID, BLOCK, lds, s, rows, n_rows, D, (const uint16_t *)x->l_qp, nel,  \
                           done, nq, x->l_sims, sstride);                                                                           \
    } while (0)
#define BX_LAUNCH(NTV, ADV)                                                                                                        \
    do {                                                                                                                            \
        const size_t lds = (size_t)2 * 3 * (NTV * 32) * BX_PITCH * sizeof(uint16_t);                                               \
        const dim3 g8((unsigned)((n_rows + 255) / 256)), b8(512);                                                                   \
        if (wv8 && six) BX_LAUNCH2(NTV, 8, true, g8, b8);                                                                           \
        else if (wv8) BX_LAUNCH2(NTV, 8, false, g8, b8);                                                                            \
        else if (six) BX_LAUNCH2(NTV, 4, true, mgrid, block);                                                                       \
        else BX_LAUNCH2(NTV, 4, false, mgrid, block);                                                                               \
        done += ADV;                                                                                                                \
    } while (0)
                if (left > 128) BX_LAUNCH(8, 256);
                else if (left > 64) BX_LAUNCH(4, 128);
                else { --passes; break; }                    // the remainder (<= 64 queries) goes to the f32 MFMA below
#undef BX_LAUNCH
#undef BX_LAUNCH2
            }
        }
        while (done < nq) {
            const int left = nq - done;
            ++passes;
            if (left > 128) { hipLaunchKernelGGL((k_cosine_mfma<8>), mgrid, block, 0, s, rows, n_rows, D, x->l_q, done, nq, x->l_sims, sstride); done += 256; }
            else if (left > 64) { hipLaunchKernelGGL((k_cosine_mfma<4>), mgrid, block, 0, s, rows, n_rows, D, x->l_q, done, nq, x->l_sims, sstride); done += 128; }
            else if (left > 32) { hipLaunchKernelGGL((k_cosine_mfma<2>), mgrid, block, 0, s, rows, n_rows, D, x->l_q, done, nq, x->l_sims, sstride); done += 64; }
            else { hipLaunchKernelGGL((k_cosine_mfma<1>), mgrid, block, 0, s, rows, n_rows, D, x->l_q, done, nq, x->l_sims, sstride); done += 32; }
        }
        done = nq;
    }
    while (done < nq && n_rows > 0) {      // the row matrix is streamed once per group of up to 8 queries
        const int left = nq - done;
        ++passes;
        if (left >= 8) { launch_cosine<8>(x, rows, n_rows, done); done += 8; }
        else if (left >= 4) { launch_cosine<4>(x, rows, n_rows, done); done += 4; }
        else if (left >= 2) { launch_cosine<2>(x, rows, n_rows, done); done += 2; }
        else { launch_cosine<1>(x, rows, n_rows, done); done += 1; }
    }
    stat_end(x, 1, (double)n_rows * D * 4.0 * passes + (double)nq * n_rows * 4.0);
    CandArgs ca;
    ca.n_cand = n_cand; ca.max_id = max_id; ca.vcap = vcap; ca.cache_size = x->c.cache_size; ca.exact = exact ? 1 : 0;
    ca.use_radius = radius >= 0 ? 1 : 0;
    ca.c0 = curr ? curr[0] : 0; ca.c1 = curr ? curr[1] : 0; ca.c2 = curr ? curr[2] : 0;
    ca.floor_lo = floor_lo; ca.floor_hi = floor_hi; ca.radius2 = radius * radius;
    ca.rgb_pos = x->rgb_pos; ca.cnt = cnt; ca.store_rows = x->store_rows; ca.name_rank = x->l_name_rank;
    bool filtered = false;
    if (K <= TK_N / 2) {
        static const int sel_min_q = getenv("BSC_SEL_MIN_Q") ? atoi(getenv("BSC_SEL_MIN_Q")) : 5;     // sample + filter from this many queries on (Q = 1: 0.79 vs 0.87 ms without / with; Q = 8: 1.08 vs 0.98; Q = 12: 1.20 vs 1.00)
        for (int attempt = 0; attempt < 2; ++attempt) {
            u64 *wk; uint32_t *wv; int64_t ws;
            BSC_TRY(select_topk_batched(x, ca, nq, sstride, K, attempt == 0 && nq >= sel_min_q, &wk, &wv, &ws, &filtered));
            hipLaunchKernelGGL(k_gather_topk, dim3((K + TPB - 1) / TPB, (unsigned)nq), block, 0, s, K, /*entries*/ K, max_id,
                               vcap, wk, wv, ws, x->rgb_pos, x->l_out_pos, x->l_out_sim);
            if (!filtered) break;
            int32_t *flag = (int32_t *)(x->l_sel_thr + nq);      // spare slot behind the thresholds
            BSC_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), s));
            hipLaunchKernelGGL(k_sel_check, dim3((nq + 255) / 256), dim3(256), 0, s, x->l_sel_thr, x->l_sel_cnt, nq,
                               SEL_SURVIVOR_CAP, flag);
            int32_t redo = 0;
            BSC_HIP(hipMemcpyAsync(&redo, flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            BSC_HIP(hipStreamSynchronize(s));
            if (!redo) break;
        }
    } else {                                   // large K: device-wide sort per query
        for (int qi = 0; qi < nq; ++qi) {
            hipLaunchKernelGGL(k_candidates, cgrid, block, 0, s, ca, x->l_sims + (int64_t)qi * sstride, x->l_key_a, x->l_val_a);
            BSC_TRY(prim_sort_pairs(x, x->l_key_a, x->l_key_b, x->l_val_a, x->l_val_b, (size_t)n_cand, 0, 64));
            hipLaunchKernelGGL(k_gather_topk, dim3((K + TPB - 1) / TPB), block, 0, s, K, n_cand, max_id, vcap, x->l_key_b,
                               x->l_val_b, (int64_t)0, x->rgb_pos, x->l_out_pos + (int64_t)qi * K * 3,
                               x->l_out_sim + (int64_t)qi * K);
        }
    }
    BSC_HIP(hipGetLastError());
    BSC_HIP(hipMemcpyAsync(out_pos, x->l_out_pos, sizeof(int32_t) * (size_t)nq * K * 3, hipMemcpyDeviceToHost, s));
    BSC_HIP(hipMemcpyAsync(out_sim, x->l_out_sim, sizeof(float) * (size_t)nq * K, hipMemcpyDeviceToHost, s));
    BSC_HIP(hipStreamSynchronize(s));
    for (int qi = 0; qi < nq; ++qi) {
        int n = 0;
        while (n < K && out_pos[((int64_t)qi * K + n) * 3] >= 0) ++n;
        out_count[qi] = n;
        x->last_counts[qi] = n;
    }
    x->last_nq = nq;
    x->last_K = K;
    return BSC_OK;
}
