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

// sims[qi * n_rows + row] = dot(q^[qi], x[row]) / max(|x[row]|, 1e-8).
// One wavefront per row (grid-stride), NV float4 per lane; the row is loaded once and reused for QT queries.
template <int NV, int QT>
__global__ __launch_bounds__(TPB) void k_cosine(const float *__restrict__ rows, int64_t n_rows, int D,
                                                const float *__restrict__ qn, int q0, float *__restrict__ sims)
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
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            float dsum = 0.f;
#pragma unroll
            for (int t = 0; t < NV; ++t)
                dsum += xv[t].x * qv[qi][t].x + xv[t].y * qv[qi][t].y + xv[t].z * qv[qi][t].z + xv[t].w * qv[qi][t].w;
            dsum = wave_sum(dsum);
            if (lane == 0) sims[(int64_t)(q0 + qi) * n_rows + r] = dsum * inv;
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

__global__ __launch_bounds__(TPB) void k_candidates(int n_cand, int max_id, int vcap, const int32_t *__restrict__ rgb_pos,
                                                    const int32_t *__restrict__ cnt, const int32_t *__restrict__ store_rows,
                                                    int cache_size, int exact, const float *__restrict__ sims,
                                                    const uint32_t *__restrict__ name_rank, double radius2, int use_radius,
                                                    int c0, int c1, int c2, int floor_lo, int floor_hi,
                                                    u64 *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int c = blockIdx.x * TPB + threadIdx.x;
    if (c >= n_cand) return;
    const int e = (c == max_id) ? vcap : c;
    u64 key = ~0ull;
    const int m = cnt[e];
    if (m > 0) {
        int r = 0, cc = 0, h = 0;
        if (c != max_id) { r = rgb_pos[3 * e]; cc = rgb_pos[3 * e + 1]; h = rgb_pos[3 * e + 2]; }
        bool ok = true;
        if (use_radius) {   // memory_2.py:624-629 (integer squared distance compared with radius**2)
            const double dx = r - c0, dy = cc - c1, dz = h - c2;
            ok = (dx * dx + dy * dy + dz * dz) <= radius2;
        }
        if (floor_lo <= floor_hi) ok = ok && (floor_lo <= h) && (h <= floor_hi);   // :633-640
        if (ok) {
            float best = -INFINITY;
            if (exact) {
                for (int k = 0; k < m; ++k) best = fmaxf(best, sims[store_rows[(int64_t)e * cache_size + k]]);   // :661
            } else {
                best = sims[e];
            }
            key = ((u64)float_desc_key(best) << 32) | (u64)name_rank[c];
        }
    }
    keys[c] = key;
    vals[c] = (uint32_t)c;
}

__global__ __launch_bounds__(TPB) void k_gather_topk(int K, int n_cand, int max_id, int vcap,
                                                     const u64 *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     const int32_t *__restrict__ rgb_pos, int32_t *__restrict__ out_pos,
                                                     float *__restrict__ out_sim)
{
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= K) return;
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
__global__ __launch_bounds__(TPB) void k_block_topk(const u64 *__restrict__ in_keys, const uint32_t *__restrict__ in_vals,
                                                    int64_t n, int K, u64 *__restrict__ out_keys,
                                                    uint32_t *__restrict__ out_vals)
{
    __shared__ u64 sk[TK_N];
    __shared__ uint32_t sv[TK_N];
    const int64_t base = (int64_t)blockIdx.x * TK_N;
    for (int i = threadIdx.x; i < TK_N; i += TPB) {
        const int64_t g = base + i;
        sk[i] = g < n ? in_keys[g] : ~0ull;
        sv[i] = g < n ? in_vals[g] : 0u;
    }
    __syncthreads();
    for (int k = 2; k <= TK_N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < TK_N / 2; t += TPB) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));     // lower index of the pair
                const int l = i | j;
                const bool up = (i & k) == 0;
                const u64 a = sk[i], b = sk[l];
                if ((a > b) == up) {
                    sk[i] = b; sk[l] = a;
                    const uint32_t va = sv[i]; sv[i] = sv[l]; sv[l] = va;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < K; i += TPB) {
        out_keys[(int64_t)blockIdx.x * K + i] = sk[i];
        out_vals[(int64_t)blockIdx.x * K + i] = sv[i];
    }
}

// after the rounds the K winners sit sorted in (keys, vals)[0..K)
static bsc_status select_topk(bsc_ctx *x, int64_t n, int K)
{
    u64 *ka = x->l_key_a, *kb = x->l_key_b;
    uint32_t *va = x->l_val_a, *vb = x->l_val_b;
    if (K > TK_N / 2) {                     // large K: device-wide sort
        BSC_TRY(prim_sort_pairs(x, ka, kb, va, vb, (size_t)n, 0, 64));
        return BSC_OK;
    }
    for (;;) {
        const int64_t nb = (n + TK_N - 1) / TK_N;
        hipLaunchKernelGGL(k_block_topk, dim3((unsigned)nb), dim3(TPB), 0, x->stream, ka, va, n, K, kb, vb);
        n = nb * K;
        if (nb == 1) break;
        u64 *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    if (kb != x->l_key_b) {                 // winners must end in the *_b buffers
        BSC_HIP(hipMemcpyAsync(x->l_key_b, kb, sizeof(u64) * K, hipMemcpyDeviceToDevice, x->stream));
        BSC_HIP(hipMemcpyAsync(x->l_val_b, vb, sizeof(uint32_t) * K, hipMemcpyDeviceToDevice, x->stream));
    }
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

bsc_status pool_query_impl(bsc_ctx *x, const float *tokens, int32_t B, int32_t T, int32_t D, float *out)
{
    hipLaunchKernelGGL(k_pool_query, dim3((D + TPB - 1) / TPB), dim3(TPB), 0, x->stream, tokens, B, T, D, out);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

template <int QT>
static void launch_cosine(bsc_ctx *x, const float *rows, int64_t n_rows, int q0)
{
    const int D = x->c.token_dim;
    const int nv = (D / 4 + 63) / 64;
    int64_t blocks = (n_rows * 64 + TPB - 1) / TPB;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(TPB);
#define LC(NV) hipLaunchKernelGGL((k_cosine<NV, QT>), grid, block, 0, x->stream, rows, n_rows, D, x->l_q, q0, x->l_sims)
    if (nv <= 1) LC(1);
    else if (nv == 2) LC(2);
    else if (nv == 3) LC(3);
    else if (nv == 4) LC(4);
    else LC(8);
#undef LC
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
    hipLaunchKernelGGL(k_normalize_q, dim3(nq), dim3(64), 0, s, q_dev, D, x->l_q);
    // dense acnt has no slot for the grid_0_0_0 group: k_candidates reads cnt[vcap]; acnt is allocated vcap+1
    int done = 0;
    stat_begin(x, 1);
    while (done < nq && n_rows > 0) {      // the row matrix is streamed once per group of up to 8 queries
        const int left = nq - done;
        if (left >= 8) { launch_cosine<8>(x, rows, n_rows, done); done += 8; }
        else if (left >= 4) { launch_cosine<4>(x, rows, n_rows, done); done += 4; }
        else if (left >= 2) { launch_cosine<2>(x, rows, n_rows, done); done += 2; }
        else { launch_cosine<1>(x, rows, n_rows, done); done += 1; }
    }
    stat_end(x, 1, (double)n_rows * D * 4.0 * ((nq + 7) / 8) + (double)nq * n_rows * 4.0);
    const double r2 = radius * radius;
    for (int qi = 0; qi < nq; ++qi) {
        hipLaunchKernelGGL(k_candidates, cgrid, block, 0, s, n_cand, max_id, vcap, x->rgb_pos, cnt, x->store_rows,
                           x->c.cache_size, exact ? 1 : 0, x->l_sims + (int64_t)qi * n_rows, x->l_name_rank, r2,
                           radius >= 0 ? 1 : 0, curr ? curr[0] : 0, curr ? curr[1] : 0, curr ? curr[2] : 0, floor_lo,
                           floor_hi, x->l_key_a, x->l_val_a);
        BSC_TRY(select_topk(x, n_cand, K));
        hipLaunchKernelGGL(k_gather_topk, dim3((K + TPB - 1) / TPB), block, 0, s, K, n_cand, max_id, vcap, x->l_key_b,
                           x->l_val_b, x->rgb_pos, x->l_out_pos + (int64_t)qi * K * 3, x->l_out_sim + (int64_t)qi * K);
    }
    BSC_HIP(hipGetLastError());
    BSC_HIP(hipMemcpyAsync(out_pos, x->l_out_pos, sizeof(int32_t) * (size_t)nq * K * 3, hipMemcpyDeviceToHost, s));
    BSC_HIP(hipMemcpyAsync(out_sim, x->l_out_sim, sizeof(float) * (size_t)nq * K, hipMemcpyDeviceToHost, s));
    BSC_HIP(hipStreamSynchronize(s));
    for (int qi = 0; qi < nq; ++qi) {
        int n = 0;
        while (n < K && out_pos[((int64_t)qi * K + n) * 3] >= 0) ++n;
        out_count[qi] = n;
    }
    return BSC_OK;
}
