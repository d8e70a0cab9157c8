// dense.hip — dense per-voxel feature reduce (mean / max modes): the HBM-bound heart of the ingest.
//
// Every pixel of a ViT patch carries the same token, so the unit of feature work is not the point but the
// unique (voxel, frame, patch) PAIR with its multiplicity:
//
//   k_keys_pairs   one workgroup per tile of 1024 points (32x32 pixel tiles when every pixel is ingested,
//                  1024 consecutive points otherwise).  Aggregates the tile's (voxel, frame, patch) codes in an LDS hash table
//                  (ds_cmpst_b64 insert + ds_add count); the distinct pairs are appended to a global list.
//                  A 10 cm voxel 2 m away covers ~16x16 pixels, so a tile collapses 1024 points to a
//                  handful of pairs.
//   radix sort     of the pair list by (voxel, frame, patch) -> each voxel's pairs are contiguous and equal
//                  codes coming from neighbouring tiles are adjacent.
//   k_pair_heads   voxel segments of the sorted pair list.
//   k_dense_reduce one wavefront per voxel: merges equal codes (integer multiplicities, so the result does
//                  not depend on the order tiles were appended in), accumulates multiplicity x token row with
//                  16-byte loads (lanes stride D, 1 KiB per wave-instruction, token tile L2 / MALL resident)
//                  and performs exactly ONE read-modify-write of the voxel's (D,) f32 accumulator row.
//
// Algorithmic HBM bytes of k_dense_reduce per call: (2U - U_new) * D*4 + 8U + F*g^2*D*4 + 12 * n_pairs.
#include "bsc_internal.h"

#include <math.h>

#define TPB 256
#define PT_TILE 1024
#define PT_HS 2048

__device__ __forceinline__ u64 mix64(u64 x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
    return x;
}

template <bool PAIRS>
__global__ __launch_bounds__(TPB) void k_keys_pairs(int64_t P, int tiled2d, int H, int W, int tx_n, int ty_n,
                                                    const int32_t *__restrict__ p_cell, const int32_t *__restrict__ occ,
                                                    const uint32_t *__restrict__ p_patf, u64 *__restrict__ pstage_key,
                                                    uint32_t *__restrict__ pstage_cnt, int32_t *__restrict__ tile_cnt,
                                                    int pb, int cb)
{
    __shared__ u64 hkey[PAIRS ? PT_HS : 1];
    __shared__ uint32_t hcnt[PAIRS ? PT_HS : 1];
    __shared__ int nloc;
    const int tid = threadIdx.x;
    if (PAIRS) {
        for (int s = tid; s < PT_HS; s += TPB) { hkey[s] = ~0ull; hcnt[s] = 0u; }
        if (tid == 0) nloc = 0;
        __syncthreads();
    }
    const int64_t tile = blockIdx.x;
    int64_t jbase = tile * PT_TILE;
    int x0 = 0, y0 = 0;
    if (tiled2d) {
        const int per_frame = tx_n * ty_n;
        const int64_t f = tile / per_frame;
        const int r = (int)(tile - f * per_frame);
        y0 = (r / tx_n) * 32;
        x0 = (r % tx_n) * 32;
        jbase = f * (int64_t)H * W;
    }
    // all loads of the thread's four points are issued before the first dependent use (clamped addresses, no branches)
    int32_t cell[PT_TILE / TPB];
    uint32_t patf[PT_TILE / TPB];
    bool in[PT_TILE / TPB];
#pragma unroll
    for (int r = 0; r < PT_TILE / TPB; ++r) {
        const int l = tid + r * TPB;
        int64_t j;
        if (tiled2d) {
            const int y = y0 + (l >> 5), x = x0 + (l & 31);
            in[r] = x < W && y < H;
            j = jbase + (int64_t)y * W + x;
        } else {
            j = jbase + l;
            in[r] = j < P;
        }
        if (!in[r]) j = jbase;
        cell[r] = p_cell[j];
        patf[r] = p_patf[j];
    }
    int32_t vid[PT_TILE / TPB];
#pragma unroll
    for (int r = 0; r < PT_TILE / TPB; ++r) vid[r] = occ[cell[r] > 0 ? cell[r] : 0];
    const int lane = tid & 63;
#pragma unroll
    for (int r = 0; r < PT_TILE / TPB; ++r) {
        u64 code = ~0ull;
        if (in[r] && cell[r] >= 0 && vid[r] >= 0)         // frame << 16 | patch  ->  voxel << cb | frame << pb | patch
            code = ((u64)(uint32_t)vid[r] << cb) | ((u64)(patf[r] >> 16) << pb) | (u64)(patf[r] & 0xffffu);
        if (PAIRS) {
            // neighbouring pixels share (voxel, frame, patch): only the first lane of every stretch of equal codes
            // inserts, with the stretch's length, instead of 64 conflicting LDS atomics
            const u64 prev = __shfl_up(code, 1);
            const bool edge = lane == 0 || code != prev;
            const u64 em = __ballot(edge);
            const u64 above = lane == 63 ? 0ull : (em & (~0ull << (lane + 1)));
            const int end = above ? (__ffsll((long long)above) - 1) : 64;
            if (edge && code != ~0ull) {
                uint32_t h = (uint32_t)mix64(code) & (PT_HS - 1);
                for (;;) {
                    const u64 old = atomicCAS(&hkey[h], ~0ull, code);
                    if (old == ~0ull || old == code) { atomicAdd(&hcnt[h], (uint32_t)(end - lane)); break; }
                    h = (h + 1) & (PT_HS - 1);
                }
            }
        }
    }
    if (PAIRS) {
        // the tile's distinct pairs go to its private staging slice (no global same-address atomics);
        // k_pair_compact packs the slices after an exclusive scan of the per-tile counts
        __syncthreads();
        for (int s = tid; s < PT_HS; s += TPB) {
            const u64 code = hkey[s];
            if (code != ~0ull) {
                const int li = atomicAdd(&nloc, 1);
                pstage_key[tile * PT_TILE + li] = code;
                pstage_cnt[tile * PT_TILE + li] = hcnt[s];
            }
        }
        __syncthreads();
        if (tid == 0) tile_cnt[tile] = nloc;
    }
}

__global__ __launch_bounds__(TPB) void k_pair_compact(int64_t n_tiles, const int32_t *__restrict__ tile_cnt,
                                                      const int32_t *__restrict__ tile_off,
                                                      const u64 *__restrict__ pstage_key,
                                                      const uint32_t *__restrict__ pstage_cnt, u64 *__restrict__ pair_key,
                                                      uint32_t *__restrict__ pair_cnt, int64_t pair_cap, int64_t *dscal)
{
    const int64_t tile = blockIdx.x;
    const int n = tile_cnt[tile];
    const int64_t off = tile_off[tile];
    for (int i = threadIdx.x; i < n; i += TPB) {
        if (off + i < pair_cap) {
            pair_key[off + i] = pstage_key[tile * PT_TILE + i];
            pair_cnt[off + i] = pstage_cnt[tile * PT_TILE + i];
        }
    }
    if (tile == n_tiles - 1 && threadIdx.x == 0) dscal[DS_B_NPAIR] = off + n;
}

// ---- deterministic, ORDERED compaction of segment heads: per-block counts, exclusive scan, ranked write ------
#define HB 1024   // elements per block
template <typename K>
__device__ __forceinline__ bool is_head(const K *__restrict__ keys, int64_t i, int64_t n, int shift, K invalid)
{
    if (i >= n) return false;
    const K k = keys[i];
    if (k == invalid) return false;
    return i == 0 || (keys[i - 1] >> shift) != (k >> shift);
}

template <typename K>
__global__ __launch_bounds__(TPB) void k_head_count(const K *__restrict__ keys, int64_t n, int shift, K invalid,
                                                    int32_t *__restrict__ blk_cnt)
{
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int local = 0;
    for (int t = 0; t < HB / TPB; ++t)
        local += is_head(keys, (int64_t)blockIdx.x * HB + t * TPB + threadIdx.x, n, shift, invalid) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = cnt;
}

template <typename K>
__global__ __launch_bounds__(TPB) void k_head_write(const K *__restrict__ keys, int64_t n, int shift, K invalid,
                                                    const int32_t *__restrict__ blk_off, int32_t *__restrict__ out,
                                                    int64_t *count_dev)
{
    // ranks inside the block follow the element order (round, then wave, then lane), so `out` is ascending
    __shared__ int wcnt[HB / TPB][TPB / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int32_t base = blk_off[blockIdx.x];
    u64 bal[HB / TPB];
#pragma unroll
    for (int t = 0; t < HB / TPB; ++t) {
        bal[t] = __ballot(is_head(keys, (int64_t)blockIdx.x * HB + t * TPB + threadIdx.x, n, shift, invalid));
        if (lane == 0) wcnt[t][wid] = __popcll(bal[t]);
    }
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int t = 0; t < HB / TPB; ++t) {
        int before = run;
        for (int w = 0; w < TPB / 64; ++w) {
            if (w < wid) before += wcnt[t][w];
            run += wcnt[t][w];
        }
        if (bal[t] & (1ull << lane))
            out[base + before + __popcll(bal[t] & ((1ull << lane) - 1ull))] = (int32_t)((int64_t)blockIdx.x * HB + t * TPB + threadIdx.x);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count_dev = (int64_t)base + run;
}

template <typename K>
static bsc_status compact_heads(bsc_ctx *x, const K *keys, int64_t n, int shift, K invalid, int32_t *out,
                                int64_t *count_dev)
{
    const int64_t nb = (n + HB - 1) / HB;
    if (nb > x->nblk_cap) { bsc_set_error("compact_heads: block table too small"); return BSC_E_CAPACITY; }
    hipLaunchKernelGGL((k_head_count<K>), dim3((unsigned)nb), dim3(TPB), 0, x->stream, keys, n, shift, invalid, x->blk_cnt);
    BSC_TRY(prim_exclusive_sum_i32(x, x->blk_cnt, x->blk_off, (size_t)nb));
    hipLaunchKernelGGL((k_head_write<K>), dim3((unsigned)nb), dim3(TPB), 0, x->stream, keys, n, shift, invalid, x->blk_off,
                       out, count_dev);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

bsc_status compact_heads_u32(bsc_ctx *x, const uint32_t *keys, int64_t n, int32_t *out, int64_t *count_dev)
{
    return compact_heads<uint32_t>(x, keys, n, 0, 0xffffffffu, out, count_dev);
}

bsc_status compact_heads_u64(bsc_ctx *x, const u64 *keys, int64_t n, int shift, int32_t *out, int64_t *count_dev)
{
    return compact_heads<u64>(x, keys, n, shift, ~0ull, out, count_dev);
}

// ---- locality-aware order of the voxel segments ---------------------------------------------------------------------
// Token rows are re-read once per (voxel, frame, patch) pair; with 128 frames per call the token tile (77 MB) is far
// larger than an XCD's 4 MB L2, and in voxel-id order the concurrently running wavefronts touch all of it (PMC: 4.5x the
// algorithmic bytes fetched).  Spatially adjacent voxels see the same patches in every frame, so the segments are
// walked in Morton order of their voxel coordinates, one contiguous eighth of that order per XCD (workgroup b runs on
// XCD b % 8 — a placement used for speed only): each XCD's working set of token rows then fits its L2.
__device__ __forceinline__ uint32_t spread3(uint32_t v)      // 10 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(TPB) void k_seg_morton(int64_t n_bound, const int64_t *dscal, const int32_t *__restrict__ seg_start,
                                                    const u64 *__restrict__ pkey, int cb, const int32_t *__restrict__ rgb_pos,
                                                    uint32_t *__restrict__ okey, uint32_t *__restrict__ oval)
{
    const int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (s >= n_bound) return;
    uint32_t key = 0xffffffffu, val = 0;
    if (s < dscal[DS_B_NPSEG]) {
        val = (uint32_t)seg_start[s];
        const uint32_t vid = (uint32_t)(pkey[val] >> cb);
        const uint32_t r = (uint32_t)rgb_pos[3 * (int64_t)vid], c = (uint32_t)rgb_pos[3 * (int64_t)vid + 1],
                       h = (uint32_t)rgb_pos[3 * (int64_t)vid + 2];
        // coarse cells of 4 voxels keep 10 bits per axis up to a 4096-cell grid; ties inside a cell are harmless
        key = (spread3(r >> 2) << 2) | (spread3(c >> 2) << 1) | spread3(h >> 2);
        key &= 0x3fffffffu;
    }
    okey[s] = key;
    oval[s] = val;
}

// four runs at a time: all token-row loads (4 x NV x 16 B per lane) are in flight before the first FMA, so a voxel
// with many (frame, patch) pairs pays the L2 / Infinity-Cache latency once per four rows instead of once per row
template <int NV, int MODE, typename TOK>
__device__ __forceinline__ void apply_runs4(float4 (&a)[NV], const uint32_t (&code)[4], const uint32_t (&cnt)[4],
                                            const TOK *__restrict__ tokens, int g2, int D, int D4, int lane, int pb)
{
    float4 xv[4][NV];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const TOK *row = tokens + ((int64_t)(code[r] >> pb) * g2 + (code[r] & ((1u << pb) - 1u))) * D;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int v = lane + 64 * t;
            xv[r][t] = (v < D4) ? load_tok4(row, v) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mult = (float)cnt[r];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (MODE == BSC_MODE_MAX) {     // padding runs repeat run 0: max is idempotent
                a[t].x = fmaxf(a[t].x, xv[r][t].x); a[t].y = fmaxf(a[t].y, xv[r][t].y);
                a[t].z = fmaxf(a[t].z, xv[r][t].z); a[t].w = fmaxf(a[t].w, xv[r][t].w);
            } else {                        // padding runs have multiplicity 0
                a[t].x = fmaf(mult, xv[r][t].x, a[t].x); a[t].y = fmaf(mult, xv[r][t].y, a[t].y);
                a[t].z = fmaf(mult, xv[r][t].z, a[t].z); a[t].w = fmaf(mult, xv[r][t].w, a[t].w);
            }
        }
    }
}

template <int NV, int MODE, typename TOK>
__global__ __launch_bounds__(TPB) void k_dense_reduce(const u64 *__restrict__ pkey, const uint32_t *__restrict__ pcnt,
                                                      int64_t n_pairs, const uint32_t *__restrict__ seg_start,
                                                      const int64_t *dscal, const TOK *__restrict__ tokens, int g2,
                                                      int D, float *__restrict__ acc, int32_t *__restrict__ acnt, int pb,
                                                      int cb)
{
    const u64 cmask = (1ull << cb) - 1ull;
    const int lane = threadIdx.x & 63;
    const int64_t nseg = dscal[DS_B_NPSEG];
    const int64_t max_id_prev = dscal[DS_MAX_ID_PREV];
    const int D4 = D >> 2;
    // XCD x (workgroups b with b % 8 == x) walks the super-chunks x, x+8, x+16, ... of the Morton-ordered segment
    // list (64 super-chunks: spatial locality inside each, heavy regions spread over all XCDs)
    const int xcd = blockIdx.x & 7;
    const int64_t w_local = (int64_t)(blockIdx.x >> 3) * (TPB / 64) + (threadIdx.x >> 6);
    const int64_t w_per_xcd = (int64_t)(gridDim.x >> 3) * (TPB / 64);
    const int64_t chunk = (nseg + 63) / 64;
    for (int64_t i = w_local; i < 8 * chunk; i += w_per_xcd) {
        const int64_t s = (xcd + 8 * (i / chunk)) * chunk + (i % chunk);
        if (s >= nseg) continue;
        const int64_t i0 = seg_start[s];
        const uint32_t vid = (uint32_t)(pkey[i0] >> cb);
        const bool is_new = (int64_t)vid >= max_id_prev;
        float4 *dst = (float4 *)(acc + (int64_t)vid * D);
        float4 a[NV];
#pragma unroll
        for (int t = 0; t < NV; ++t)
            a[t] = (MODE == BSC_MODE_MAX) ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t total = 0;
        uint32_t pend_code = 0, pend_cnt = 0;               // run still open (may continue in the next 64 pairs)
        for (int64_t base = i0;; base += 64) {
            const int64_t k = base + lane;
            const u64 key = (k < n_pairs) ? pkey[k] : ~0ull;
            const bool inseg = (k < n_pairs) && ((uint32_t)(key >> cb) == vid);
            const uint32_t code = inseg ? (uint32_t)(key & cmask) : 0xffffffffu;
            const uint32_t cnt = inseg ? pcnt[k] : 0u;
            const int n = __popcll(__ballot(inseg));
            uint32_t ps = cnt;                               // inclusive prefix sum of the multiplicities
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(ps, o);
                if (lane >= o) ps += t;
            }
            const uint32_t prev = __shfl_up(code, 1);
            const bool head = inseg && (lane == 0 || code != prev);
            u64 hm = __ballot(head);
            // multiplicity of the run that starts at this lane (equal codes are adjacent after the sort)
            const u64 above = (lane == 63) ? 0ull : (hm & (~0ull << (lane + 1)));
            int e = above ? (__ffsll((long long)above) - 1) : n;
            e = e < 1 ? 1 : e;
            const uint32_t upto = __shfl(ps, e - 1);
            uint32_t before = __shfl_up(ps, 1);
            if (lane == 0) before = 0;
            uint32_t run_cnt = head ? (upto - before) : 0u;
            if (pend_cnt) {
                const uint32_t c0 = __shfl(code, 0);
                if (n > 0 && c0 == pend_code) {
                    if (lane == 0) run_cnt += pend_cnt;     // the open run continues in this chunk
                } else {
                    const uint32_t cc[4] = {pend_code, pend_code, pend_code, pend_code};
                    const uint32_t rc[4] = {pend_cnt, 0u, 0u, 0u};
                    apply_runs4<NV, MODE, TOK>(a, cc, rc, tokens, g2, D, D4, lane, pb);
                }
                pend_cnt = 0;
            }
            if (n == 64) {                                   // the last run of a full chunk may continue
                const int lh = 63 - __clzll((long long)hm);
                pend_code = __shfl(code, lh);
                pend_cnt = __shfl(run_cnt, lh);
                hm &= ~(1ull << lh);
            }
            while (hm) {
                uint32_t cc[4], rc[4];
                int b = __ffsll((long long)hm) - 1;
                hm &= hm - 1;
                cc[0] = __shfl(code, b);
                rc[0] = __shfl(run_cnt, b);
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    if (hm) {
                        b = __ffsll((long long)hm) - 1;
                        hm &= hm - 1;
                        cc[r] = __shfl(code, b);
                        rc[r] = __shfl(run_cnt, b);
                    } else {
                        cc[r] = cc[0];
                        rc[r] = 0u;
                    }
                }
                apply_runs4<NV, MODE, TOK>(a, cc, rc, tokens, g2, D, D4, lane, pb);
            }
            if (n > 0) total += __shfl(ps, n - 1);
            if (n < 64) break;
        }
        if (pend_cnt) {
            const uint32_t cc[4] = {pend_code, pend_code, pend_code, pend_code};
            const uint32_t rc[4] = {pend_cnt, 0u, 0u, 0u};
            apply_runs4<NV, MODE, TOK>(a, cc, rc, tokens, g2, D, D4, lane, pb);
        }
        // one read-modify-write of the accumulator row (read here, not before the walk: 12 registers fewer per lane
        // during the walk = one more resident wavefront per SIMD, and the other wavefronts cover this latency)
        float4 old[NV];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int v = lane + 64 * t;
            old[t] = (!is_new && v < D4) ? dst[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int32_t old_cnt = (!is_new && lane == 0) ? acnt[vid] : 0;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int v = lane + 64 * t;
            if (v < D4) {
                float4 o = a[t];
                if (!is_new) {
                    if (MODE == BSC_MODE_MAX) {
                        o.x = fmaxf(o.x, old[t].x); o.y = fmaxf(o.y, old[t].y); o.z = fmaxf(o.z, old[t].z); o.w = fmaxf(o.w, old[t].w);
                    } else {
                        o.x += old[t].x; o.y += old[t].y; o.z += old[t].z; o.w += old[t].w;
                    }
                }
                dst[v] = o;
            }
        }
        if (lane == 0) acnt[vid] = old_cnt + (int32_t)total;
    }
}

__global__ void k_dense_counters(int64_t *dscal)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dscal[DS_RMW_TOTAL] += dscal[DS_B_NPSEG];
        dscal[DS_PAIR_TOTAL] += dscal[DS_B_NPAIR];
    }
}

static inline int code_patch_bits(const bsc_ctx *x) { return ceil_log2_u64((uint64_t)x->g2); }
static inline int code_bits(const bsc_ctx *x, int n_frames) { return code_patch_bits(x) + ceil_log2_u64((uint64_t)n_frames); }

template <int MODE, typename TOK>
static void launch_dense(bsc_ctx *x, int64_t n_pairs, const TOK *tokens, int pb, int cb)
{
    const int D = x->c.token_dim;
    const int nv = (D / 4 + 63) / 64;
    const dim3 grid(256 * 8), block(TPB);
#define LD(NV)                                                                                                          \
    hipLaunchKernelGGL((k_dense_reduce<NV, MODE, TOK>), grid, block, 0, x->stream, x->pair_key_b, x->pair_cnt_b, n_pairs,    \
                       (const uint32_t *)x->pseg_start, x->dscal, tokens, x->g2, D, x->acc, x->acnt, pb, cb)
    if (nv <= 1) LD(1);
    else if (nv == 2) LD(2);
    else if (nv == 3) LD(3);
    else if (nv == 4) LD(4);
    else LD(8);
#undef LD
}

// dense modes: the per-tile (voxel, frame, patch) pairs of the batch
bsc_status launch_keys_pairs(bsc_ctx *x, int64_t P, int n_frames, bool all_pixels)
{
    const bool pairs = x->c.mode != BSC_MODE_EXACT;
    if (!pairs) return BSC_OK;
    const int H = x->c.height, W = x->c.width;
    const int tx_n = (W + 31) / 32, ty_n = (H + 31) / 32;
    const int64_t tiles = all_pixels ? (int64_t)n_frames * tx_n * ty_n : (P + PT_TILE - 1) / PT_TILE;
    if (pairs && tiles > x->max_tiles) { bsc_set_error("launch_keys_pairs: %lld tiles > %lld", (long long)tiles, (long long)x->max_tiles); return BSC_E_CAPACITY; }
    const int pb = code_patch_bits(x), cb = code_bits(x, n_frames);
    const dim3 grid((unsigned)tiles), block(TPB);
    if (pairs) {
        hipLaunchKernelGGL((k_keys_pairs<true>), grid, block, 0, x->stream, P, all_pixels ? 1 : 0, H, W, tx_n, ty_n,
                           x->p_cell, x->occ, x->p_patf, x->pstage_key, x->pstage_cnt, x->tile_cnt,
                           pb, cb);
        BSC_TRY(prim_exclusive_sum_i32(x, x->tile_cnt, x->tile_off, (size_t)tiles));
        hipLaunchKernelGGL(k_pair_compact, grid, block, 0, x->stream, tiles, x->tile_cnt, x->tile_off, x->pstage_key,
                           x->pstage_cnt, x->pair_key_a, x->pair_cnt_a, x->pair_cap, x->dscal);
    }
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

bsc_status dense_reduce_batch(bsc_ctx *x, const void *tokens, int token_dtype, int n_frames)
{
    hipStream_t s = x->stream;
    const int64_t n_pairs = x->hscal[DS_B_NPAIR];   // read back by ingest_batch after the front end
    if (n_pairs > x->pair_cap) {
        bsc_set_error("pair list overflow (%lld > %lld)", (long long)n_pairs, (long long)x->pair_cap);
        return BSC_E_CAPACITY;
    }
    if (n_pairs == 0) return BSC_OK;
    const int pb = code_patch_bits(x), cb = code_bits(x, n_frames);
    const int vid_bits = ceil_log2_u64((uint64_t)x->hscal[DS_MAX_ID] + 1);
    BSC_TRY(prim_sort_pairs_onesweep(x, x->pair_key_a, x->pair_key_b, x->pair_cnt_a, x->pair_cnt_b, (size_t)n_pairs, 0,
                                     cb + vid_bits));
    BSC_TRY(compact_heads_u64(x, x->pair_key_b, n_pairs, cb, x->pseg_start, x->dscal + DS_B_NPSEG));
    // segments in Morton order of their voxels (the number of segments is only known on the device; it is bounded by
    // the voxel count read back earlier, slots beyond it carry 0xffffffff keys and sort last)
    const int64_t n_bound = n_pairs < x->hscal[DS_MAX_ID] ? n_pairs : x->hscal[DS_MAX_ID];
    hipLaunchKernelGGL(k_seg_morton, dim3((unsigned)((n_bound + TPB - 1) / TPB)), dim3(TPB), 0, s, n_bound, x->dscal, x->pseg_start,
                       x->pair_key_b, cb, x->rgb_pos, x->skey_a, x->sval_a);
    BSC_TRY(prim_sort_pairs_u32(x, x->skey_a, x->pair_cnt_a, x->sval_a, (uint32_t *)x->pseg_start, (size_t)n_bound, 0, 30));
    stat_begin(x, 0);
    if (token_dtype == BSC_TOK_BF16) {
        if (x->c.mode == BSC_MODE_MEAN) launch_dense<BSC_MODE_MEAN>(x, n_pairs, (const bf16_t *)tokens, pb, cb);
        else launch_dense<BSC_MODE_MAX>(x, n_pairs, (const bf16_t *)tokens, pb, cb);
    } else {
        if (x->c.mode == BSC_MODE_MEAN) launch_dense<BSC_MODE_MEAN>(x, n_pairs, (const float *)tokens, pb, cb);
        else launch_dense<BSC_MODE_MAX>(x, n_pairs, (const float *)tokens, pb, cb);
    }
    stat_end(x, 0, 0.0);   // bytes are derived from the device counters (voxel rows, new rows, pairs)
    hipLaunchKernelGGL(k_dense_counters, dim3(1), dim3(64), 0, s, x->dscal);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}
