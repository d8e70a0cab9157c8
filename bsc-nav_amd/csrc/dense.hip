// dense.hip — dense per-voxel feature reduce (mean / max modes): the HBM-bound heart of the ingest.
//
// Every pixel of a ViT patch carries the same token, so the unit of feature work is not the point but the
// unique (voxel, frame, patch) PAIR with its multiplicity:
//
//   k_keys_pairs   one workgroup per tile of 1024 points (32x32 pixel tiles when every pixel is ingested,
//                  1024 consecutive points otherwise).  Aggregates the tile's (cell, frame, patch) codes in an LDS hash
//                  table (ds_cmpst_b64 insert + ds_add count); the distinct pairs are appended to a global list.
//                  A 10 cm voxel 2 m away covers ~16x16 pixels, so a tile collapses 1024 points to a
//                  handful of pairs.  The voxel is named by the MORTON code of its cell (row, col, h), not by its id:
//                  the pairs do not wait for the ids, and the sorted list walks the voxels in a spatially coherent order.
//   radix sort     of the pair list by (cell code, frame, patch) -> each voxel's pairs are contiguous and equal
//                  codes coming from neighbouring tiles are adjacent.
//   compact_heads  voxel segments of the sorted pair list.
//   k_dense_reduce one wavefront per voxel: merges equal codes (integer multiplicities, so the result does
//                  not depend on the order tiles were appended in), accumulates multiplicity x token row with
//                  16-byte loads (lanes stride D, 1 KiB per wave-instruction, token tile L2 / MALL resident)
//                  and performs exactly ONE read-modify-write of the voxel's (D,) f32 accumulator row.
//
// Algorithmic HBM bytes of k_dense_reduce per call: (2U - U_new) * D*4 + 8U + F*g^2*D*4 + 12 * n_pairs.
#include "bsc_internal.h"

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

#include <math.h>

#define TPB 256
#define PT_TILE 1024
#define PT_HS 2048

__device__ __forceinline__ u64 mix64(u64 x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
    return x;
}

// ---- cell <-> sort code -------------------------------------------------------------------------------------------
// Morton interleave of (row, col, h), `ab` bits per axis (ab = 0: the linear cell index itself, for grids whose Morton
// code would not fit beside the frame / patch bits).  Spatially adjacent voxels see the same patches in every frame, so
// walking the voxel segments in this order keeps the working set of token rows of concurrently running wavefronts small.
struct CellCode {
    int ab;        // bits per axis (0 = linear)
    int gs, nh;
};
__device__ __forceinline__ u64 spread3(u64 v)       // 21 bits -> every third bit
{
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__device__ __forceinline__ u64 compact3(u64 v)
{
    v &= 0x1249249249249249ull;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ull;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00full;
    v = (v ^ (v >> 8)) & 0x1f0000ff0000ffull;
    v = (v ^ (v >> 16)) & 0x1f00000000ffffull;
    v = (v ^ (v >> 32)) & 0x1fffffull;
    return v;
}
__device__ __forceinline__ u64 cell_to_code(const CellCode &cc, int32_t cell)
{
    if (cc.ab == 0) return (u64)(uint32_t)cell;
    const int32_t h = cell % cc.nh, rc = cell / cc.nh;
    return (spread3((u64)(rc / cc.gs)) << 2) | (spread3((u64)(rc % cc.gs)) << 1) | spread3((u64)h);
}
__device__ __forceinline__ int32_t code_to_cell(const CellCode &cc, u64 code)
{
    if (cc.ab == 0) return (int32_t)code;
    const int32_t row = (int32_t)compact3(code >> 2), col = (int32_t)compact3(code >> 1), h = (int32_t)compact3(code);
    return (row * cc.gs + col) * cc.nh + h;
}

// p_patf == nullptr: every pixel is a point (tiled2d) and the patch is a function of the pixel (pat_x / pat_y tables of
// the fast geometry); otherwise frame << 16 | patch per point.
__global__ __launch_bounds__(TPB) void k_keys_pairs(int64_t P, int tiled2d, int H, int W, int tx_n, int ty_n, CellCode cc,
                                                    const int32_t *__restrict__ p_cell, const uint32_t *__restrict__ p_patf,
                                                    const uint8_t *__restrict__ pat_x, const uint8_t *__restrict__ pat_y, int g,
                                                    u64 *__restrict__ pstage_key, uint32_t *__restrict__ pstage_cnt,
                                                    int32_t *__restrict__ tile_cnt, int pb, int cb)
{
    __shared__ u64 hkey[PT_HS];
    __shared__ uint32_t hcnt[PT_HS];
    __shared__ int nloc;
    const int tid = threadIdx.x;
    for (int s = tid; s < PT_HS; s += TPB) { hkey[s] = ~0ull; hcnt[s] = 0u; }
    if (tid == 0) nloc = 0;
    __syncthreads();
    const int64_t tile = blockIdx.x;
    int64_t jbase = tile * PT_TILE;
    int x0 = 0, y0 = 0;
    uint32_t frame = 0;
    if (tiled2d) {
        const int per_frame = tx_n * ty_n;
        const int64_t f = tile / per_frame;
        const int r = (int)(tile - f * per_frame);
        y0 = (r / tx_n) * 32;
        x0 = (r % tx_n) * 32;
        jbase = f * (int64_t)H * W;
        frame = (uint32_t)f;
    }
    // all loads of the thread's four points are issued before the first dependent use (clamped addresses, no branches)
    int32_t cell[PT_TILE / TPB];
    uint32_t patf[PT_TILE / TPB];
#pragma unroll
    for (int r = 0; r < PT_TILE / TPB; ++r) {
        const int l = tid + r * TPB;
        int64_t j;
        bool in;
        int x = 0, y = 0;
        if (tiled2d) {
            y = y0 + (l >> 5); x = x0 + (l & 31);
            in = x < W && y < H;
            j = jbase + (int64_t)y * W + x;
        } else {
            j = jbase + l;
            in = j < P;
        }
        if (!in) j = jbase;
        const int32_t c = p_cell[j];
        cell[r] = in ? c : -1;
        if (p_patf) patf[r] = p_patf[j];
        else patf[r] = (frame << 16) | ((uint32_t)pat_y[in ? y : 0] * (uint32_t)g + (uint32_t)pat_x[in ? x : 0]);
    }
    const int lane = tid & 63;
#pragma unroll
    for (int r = 0; r < PT_TILE / TPB; ++r) {
        // neighbouring pixels share (cell, frame, patch): only the first lane of every stretch of equal pairs
        // inserts, with the stretch's length, instead of 64 conflicting LDS atomics
        const int32_t pc = __shfl_up(cell[r], 1);
        const uint32_t pp = __shfl_up(patf[r], 1);
        const bool edge = lane == 0 || cell[r] != pc || patf[r] != pp;
        const u64 em = __ballot(edge);
        const u64 above = lane == 63 ? 0ull : (em & (~0ull << (lane + 1)));
        const int end = above ? (__ffsll((long long)above) - 1) : 64;
        if (edge && cell[r] >= 0) {        // frame << 16 | patch  ->  cell code << cb | frame << pb | patch
            const u64 code = (cell_to_code(cc, cell[r]) << cb) | ((u64)(patf[r] >> 16) << pb) | (u64)(patf[r] & 0xffffu);
            uint32_t h = (uint32_t)mix64(code) & (PT_HS - 1);
            for (;;) {
                const u64 old = atomicCAS(&hkey[h], ~0ull, code);
                if (old == ~0ull || old == code) { atomicAdd(&hcnt[h], (uint32_t)(end - lane)); break; }
                h = (h + 1) & (PT_HS - 1);
            }
        }
    }
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

// ---- patch-aligned tiles (every pixel ingested, fast geometry) -----------------------------------------------------
// One workgroup per (frame, ViT patch): all pixels of the tile carry the same token row, so the tile's pairs are simply
// its distinct cells with their point counts — a (voxel, frame, patch) pair can only come from ONE tile.  Consequences:
// no duplicate pairs to merge, a 32-bit hash key (the cell), and after a STABLE sort by any function of the cell the pairs
// of a voxel are in (frame, patch) order whatever order the tile listed them in — the reduce is deterministic without
// sorting the frame / patch bits at all.
// Pair record: token row (frame * g^2 + patch) << 32 | count; sort key: the cell code (Morton).
#define PP_HS 4096
#define PP_R 13                 // rounds of TPB pixels: tiles of up to 3328 pixels (640x480 at 14x14 patches: 46 x 46, and 68 x 46 in
                                // patch column 0, which int() widens to u in (-1, 1))
#define PP_HEADS 256            // stretch heads a wavefront lists before it hashes them
#define PP_CELL_BITS 26         // a head is cell | (points - 1) << 26: the path is taken for grids of up to 2^26 cells (capi.hip)
// LDS accesses of ONE wavefront execute in program order; this keeps the compiler from reordering them (a fence over the LDS address
// space alone: over all memory it is lowered to s_waitcnt vmcnt(0))
__device__ __forceinline__ void pp_wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}
// Persistent workgroups walk the tiles: the 4096-slot table (sized for one cell per pixel) is cleared ONCE; every insert that
// claims an empty slot also lists it, so a tile's pairs are emitted — and its slots cleared again — by walking that short list
// (a few dozen entries for a surface at room distance) instead of scanning and re-initialising 4096 slots per tile.
// Instructions per pixel (round 5: ~1000 -> ~600 per wavefront and tile):
//   * the pixel of round r + 1 follows from the pixel of round r (TPB pixels further along the rows of the tile) with an add and a
//     compare instead of a division per load; rounds past the tile's end are skipped;
//   * neighbouring pixels share the cell: only the first lane of a stretch inserts, with the stretch's length — and the inserts are
//     not made round by round by the few lanes that head a stretch (13 rounds x ~40 instructions, mostly idle lanes): every
//     wavefront LISTS its stretch heads in LDS (ballot + rank, one store per round) and hashes the list densely, one head per
//     lane, once per tile (or whenever the list is nearly full).
// None of that moved the kernel's time by more than 5 % (phase stamps, -DBSC_PAIRS_PROFILE: ~10.5 k clocks per tile and workgroup in
// every variant, the wait just moves to whichever instruction touches memory next; a prefetch of the next tile's cells into LDS
// changed nothing either): the tile's time is its share of the memory system, which the call's side stream is loading at the same
// time — what counts is the bytes.  A tile row is 46 pixels = 184 bytes at an arbitrary offset: 2-3 cache lines, shared with the
// tiles left and right of it.  Tiles are therefore dealt out so that the workgroups of one XCD (blockIdx % 8: the round-robin
// dispatch) walk CONSECUTIVE tiles at the same time and the shared lines are fetched once into that XCD's L2.
__global__ __launch_bounds__(TPB) void k_patch_pairs(int W, int64_t N, int g, int64_t n_tiles, const int32_t *__restrict__ pt_rect,
                                                     const int32_t *__restrict__ pt_off, CellCode cc,
                                                     const int32_t *__restrict__ p_cell, u64 *__restrict__ stage_rec,
                                                     uint32_t *__restrict__ stage_blk, int32_t *__restrict__ tile_cnt)
{
    __shared__ uint32_t hkey[PP_HS];
    __shared__ uint32_t hcnt2[PP_HS / 2];           // points per slot, 16 bits each (a tile has < 2^16 pixels): slot h in half h & 1 of word h >> 1
    __shared__ uint16_t hlist[PP_R * TPB];          // slots claimed by the tile in flight (<= one per pixel)
    __shared__ uint32_t s_head[TPB / 64][PP_HEADS];
    __shared__ int nloc;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g2 = g * g;
    for (int s = tid; s < PP_HS; s += TPB) hkey[s] = 0xffffffffu;
    for (int s = tid; s < PP_HS / 2; s += TPB) hcnt2[s] = 0u;
    if (tid == 0) nloc = 0;
    __syncthreads();
#ifdef BSC_PAIRS_PROFILE
    long long tph[4] = {0, 0, 0, 0}, tq = clock64();
    int n_prof = 0;
#define PP_T(k) { const long long now_ = clock64(); tph[k] += now_ - tq; tq = now_; }
#else
#define PP_T(k)
#endif
    // the wavefront's listed heads -> the table: one head per lane
    auto hash_heads = [&](int cnt) {
        pp_wave_lds_order();
        for (int i = lane; i < cnt; i += 64) {
            const uint32_t hd = s_head[wv][i];
            const uint32_t key = hd & ((1u << PP_CELL_BITS) - 1u), len = (hd >> PP_CELL_BITS) + 1u;
            uint32_t h = (key * 2654435761u) >> 20;            // 12 bits
            for (;;) {
                const uint32_t old = atomicCAS(&hkey[h], 0xffffffffu, key);
                if (old == 0xffffffffu) hlist[atomicAdd(&nloc, 1)] = (uint16_t)h;      // first claim of the slot: list it
                if (old == 0xffffffffu || old == key) { atomicAdd(&hcnt2[h >> 1], len << (16 * (h & 1u))); break; }
                h = (h + 1) & (PP_HS - 1);
            }
        }
        pp_wave_lds_order();
    };
    // chunk c = `per` consecutive tiles; XCD x takes the chunks x, x + 8, ...; its workgroup `slot` the slot-th tile of each
    const bool by_xcd = (gridDim.x & 7u) == 0u;
    const int64_t per = by_xcd ? gridDim.x >> 3 : gridDim.x;
    const int64_t slot = by_xcd ? blockIdx.x >> 3 : blockIdx.x, chunk0 = by_xcd ? blockIdx.x & 7u : 0, chunk_step = by_xcd ? 8 : 1;
    for (int64_t chunk = chunk0; chunk * per < n_tiles; chunk += chunk_step) {
        const int64_t tile = chunk * per + slot;
        if (tile >= n_tiles) break;
        const int64_t f = tile / g2;
        const int p = (int)(tile - f * g2);
        const int x0 = pt_rect[4 * p], w = pt_rect[4 * p + 1], y0 = pt_rect[4 * p + 2], n = pt_rect[4 * p + 3];    // n = w * h pixels
        if (n == 0) {
            if (tid == 0) tile_cnt[tile] = 0;
            continue;
        }
        const int dq = TPB / w, dr = TPB - dq * w;              // TPB pixels further = dq rows + dr columns (uniform)
        const int d_off = dq * W + dr, d_wrap = W - w;
        int py = (int)((float)tid * (1.0f / (float)w));
        int px = tid - py * w;
        if (px < 0) { --py; px += w; } else if (px >= w) { ++py; px -= w; }
        int off = py * W + px;
        const int32_t *tbase = p_cell + (f * N + (int64_t)y0 * W + x0);
        int32_t cell[PP_R];
#pragma unroll
        for (int r = 0; r < PP_R; ++r) {                // all loads in flight before the first use (clamped addresses)
            cell[r] = -1;
            if (r * TPB < n) {
                cell[r] = tbase[tid + r * TPB < n ? off : 0];
                px += dr; off += d_off;
                if (px >= w) { px -= w; off += d_wrap; }
            }
        }
        PP_T(0)
        int cnt = 0;                                    // heads in the wavefront's list (uniform)
#pragma unroll
        for (int r = 0; r < PP_R; ++r) {
            if (r * TPB < n) {
                if (cnt > PP_HEADS - 64) { hash_heads(cnt); cnt = 0; }
                const int32_t c = tid + r * TPB < n ? cell[r] : -1;
                const int32_t pc = __builtin_amdgcn_update_dpp(0, c, 0x138, 0xf, 0xf, false);      // wave_shr:1
                const bool edge = lane == 0 || c != pc;
                const u64 em = __ballot(edge);
                const u64 above = lane == 63 ? 0ull : (em & (~0ull << (lane + 1)));
                const int end = above ? (__ffsll((long long)above) - 1) : 64;
                const bool ins = edge && c >= 0;
                const u64 im = __ballot(ins);
                if (ins)
                    s_head[wv][cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u))] =
                        (uint32_t)c | ((uint32_t)(end - lane - 1) << PP_CELL_BITS);
                cnt += __popcll(im);
            }
        }
        PP_T(1)
        hash_heads(cnt);
        PP_T(2)
        // the tile's distinct cells go to its private staging slice; k_patch_compact packs the slices after a scan of the counts
        __syncthreads();
        const int nl = nloc;
        const int64_t base = f * N + pt_off[p];              // the tile's staging slice: as many slots as it has pixels
        const u64 row = (u64)(f * g2 + p) << 32;
        for (int li = tid; li < nl; li += TPB) {
            const uint32_t h = hlist[li];
            const uint32_t key = hkey[h];
            stage_rec[base + li] = row | (u64)((hcnt2[h >> 1] >> (16 * (h & 1u))) & 0xffffu);
            stage_blk[base + li] = (uint32_t)cell_to_code(cc, (int32_t)key);
            hkey[h] = 0xffffffffu;                           // the table is clean again for the workgroup's next tile
            atomicAnd(&hcnt2[h >> 1], (h & 1u) ? 0x0000ffffu : 0xffff0000u);
        }
        __syncthreads();
        if (tid == 0) { tile_cnt[tile] = nl; nloc = 0; }
        __syncthreads();
        PP_T(3)
#ifdef BSC_PAIRS_PROFILE
        ++n_prof;
#endif
    }
#ifdef BSC_PAIRS_PROFILE
    if ((blockIdx.x == 100 || blockIdx.x == 611) && (tid == 0 || tid == 192))
        printf("k_patch_pairs wg %d wave %d: %d tiles; issue %lld  arrive + list %lld  hash %lld  emit %lld (clocks per tile)\n",
               (int)blockIdx.x, tid >> 6, n_prof, tph[0] / n_prof, tph[1] / n_prof, tph[2] / n_prof, tph[3] / n_prof);
#endif
#undef PP_T
}

__global__ __launch_bounds__(TPB) void k_patch_compact(int64_t n_tiles, int g2, int64_t N, const int32_t *__restrict__ pt_off,
                                                       const int32_t *__restrict__ tile_cnt, const int32_t *__restrict__ tile_off,
                                                       const u64 *__restrict__ stage_rec, const uint32_t *__restrict__ stage_blk,
                                                       u64 *__restrict__ pair_rec, uint32_t *__restrict__ pair_blk,
                                                       uint32_t *__restrict__ pair_idx, int64_t pair_cap, int64_t *dscal)
{
    const int64_t tile = blockIdx.x;
    const int n = tile_cnt[tile];
    const int64_t off = tile_off[tile];
    const int64_t f = tile / g2;
    const int64_t base = f * N + pt_off[tile - f * g2];
    for (int i = threadIdx.x; i < n; i += TPB) {
        if (off + i < pair_cap) {
            pair_rec[off + i] = stage_rec[base + i];
            pair_blk[off + i] = stage_blk[base + i];
            pair_idx[off + i] = (uint32_t)(off + i);
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
    // own count / offset tables: blk_cnt / blk_off hold the run counts of the batch until k_runs has consumed them
    hipLaunchKernelGGL((k_head_count<K>), dim3((unsigned)nb), dim3(TPB), 0, x->stream, keys, n, shift, invalid, x->hb_cnt);
    BSC_TRY(prim_exclusive_sum_i32(x, x->hb_cnt, x->hb_off, (size_t)nb));
    hipLaunchKernelGGL((k_head_write<K>), dim3((unsigned)nb), dim3(TPB), 0, x->stream, keys, n, shift, invalid, x->hb_off,
                       out, count_dev);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

static bsc_status compact_heads_u32(bsc_ctx *x, const uint32_t *keys, int64_t n, int32_t *out, int64_t *count_dev)
{
    return compact_heads<uint32_t>(x, keys, n, 0, 0xffffffffu, out, count_dev);
}

bsc_status compact_heads_u64(bsc_ctx *x, const u64 *keys, int64_t n, int shift, int32_t *out, int64_t *count_dev)
{
    return compact_heads<u64>(x, keys, n, shift, ~0ull, out, count_dev);
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
                                                      int cb, CellCode cc, const int32_t *__restrict__ occ)
{
    const u64 cmask = (1ull << cb) - 1ull;
    const int lane = threadIdx.x & 63;
    const int64_t nseg = dscal[DS_B_NPSEG];
    const int64_t max_id_prev = dscal[DS_MAX_ID_PREV];
    const int D4 = D >> 2;
    // The segment list is in Morton order of the voxel cells.  XCD x (workgroups b with b % 8 == x) walks the
    // super-chunks x, x+8, x+16, ... of it (64 super-chunks: spatial locality inside each, so an XCD's working set of
    // token rows fits its L2; heavy regions spread over all XCDs)
    const int xcd = blockIdx.x & 7;
    const int64_t w_local = (int64_t)(blockIdx.x >> 3) * (TPB / 64) + (threadIdx.x >> 6);
    const int64_t w_per_xcd = (int64_t)(gridDim.x >> 3) * (TPB / 64);
    const int64_t chunk = (nseg + 63) / 64;
    for (int64_t i = w_local; i < 8 * chunk; i += w_per_xcd) {
        const int64_t s = (xcd + 8 * (i / chunk)) * chunk + (i % chunk);
        if (s >= nseg) continue;
        const int64_t i0 = seg_start[s];
        const u64 ccode = pkey[i0] >> cb;                    // cell code of the segment
        const uint32_t vid = (uint32_t)occ[code_to_cell(cc, ccode)];
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
            const bool inseg = (k < n_pairs) && ((key >> cb) == ccode);
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

// ---- per-voxel reduce over the pairs of the patch-aligned tiles --------------------------------------------------------
// The pairs are unique per (voxel, frame, patch) and, sorted by the cell code, contiguous per voxel in (frame, patch)
// order: one wavefront per voxel walks its pairs 64 at a time, keeps four token rows (NV x 16 B per lane each) in flight
// and finishes with ONE read-modify-write of the voxel's accumulator row.  No merging, no atomics.
// float4 slot of accumulator t in a voxel row.  Plain: 4 columns at 4 * (lane + 64 t).  PAIRED (bf16 rows, sum): accumulators
// 2u and 2u+1 are the two halves of 8 consecutive columns at 8 * (lane + 64 u), which one 16-byte load of a bf16 row
// delivers — the kernel is bound by the number of row-load instructions it issues, and this halves them (an odd last
// accumulator keeps the plain slot).
template <int NV, bool PAIRED> __device__ __forceinline__ int acc_slot(int t, int lane)
{
    if (!PAIRED || ((NV & 1) && t == NV - 1)) return lane + 64 * t;
    return 2 * (lane + 64 * (t >> 1)) + (t & 1);
}

// token rows in flight per wavefront: 8 row gathers outstanding cover the L2-miss latency of the one-voxel-per-point regime (a 58 MB
// token tile against 4 MB of L2 per XCD: 45 GB per call past the L2 by PMC) — reduce 11.2 -> 8.9 ms there, neutral where rows hit
#ifndef BSC_REDUCE_RF
#define BSC_REDUCE_RF 8
#endif
template <int NV, int MODE, typename TOK, bool PAIRED = false>
__global__ __launch_bounds__(TPB) void k_dense_reduce_voxels(const uint32_t *__restrict__ code_sorted,
                                                             const uint32_t *__restrict__ idx_sorted,
                                                             const u64 *__restrict__ pair_rec, int64_t n_pairs,
                                                             const uint32_t *__restrict__ seg_start, const int64_t *dscal,
                                                             const TOK *__restrict__ tokens, int D,
                                                             float *__restrict__ acc_g, int32_t *__restrict__ acnt,
                                                             CellCode cc, const int32_t *__restrict__ occ,
                                                             uint32_t row_lo, uint32_t row_hi, int later_pass,
                                                             float2 *__restrict__ rscale)
{
    // rscale (may be null): per finished row the operand scale and inverse norm the batched localize scan reads (bsc_row_scale_of) —
    // the wavefront holds the row in registers here, so the scan never needs its own pass over the map after an ingest (3.8 ms for
    // 2^20 x 1024 rows, paid by the first query batch after every ingest until round 5).
    // [row_lo, row_hi): the token rows (frame * g^2 + patch) this launch reduces.  A call whose token tile is larger than the
    // 256 MB MALL is reduced in passes over slices of its frames (dense_reduce_batch): a voxel's pairs are in row order, so every
    // pass takes a contiguous stretch of its segment; passes after the first add to what the earlier ones stored.
    constexpr int RF = BSC_REDUCE_RF;          // token rows in flight per wavefront
    const int lane = threadIdx.x & 63;
    const int64_t nseg = dscal[DS_B_NPSEG];
    const int64_t max_id_prev = dscal[DS_MAX_ID_PREV];
    const int D4 = D >> 2;
    // the segment list is in Morton order of the cells: XCD x walks the super-chunks x, x+8, ... (spatial locality inside
    // each, so an XCD's working set of token rows fits its L2; heavy regions spread over all XCDs)
    const int xcd = blockIdx.x & 7;
    const int64_t w_local = (int64_t)(blockIdx.x >> 3) * (TPB / 64) + (threadIdx.x >> 6);
    const int64_t w_per_xcd = (int64_t)(gridDim.x >> 3) * (TPB / 64);
    const int64_t chunk = (nseg + 63) / 64;
#ifdef BSC_REDUCE_PROFILE
    const long long t_begin = clock64();
    int n_vox = 0, n_rows_p = 0, n_max = 0;
#endif
    for (int64_t it = w_local; it < 8 * chunk; it += w_per_xcd) {
        const int64_t s = (xcd + 8 * (it / chunk)) * chunk + (it % chunk);
        if (s >= nseg) continue;
        const int64_t i0 = seg_start[s];
        const uint32_t code = code_sorted[i0];
        const int64_t vid = occ[code_to_cell(cc, (u64)code)];
#ifdef BSC_REDUCE_PROFILE
        {
            const int64_t i1 = s + 1 < nseg ? (int64_t)seg_start[s + 1] : n_pairs;
            ++n_vox; n_rows_p += (int)(i1 - i0); n_max = max(n_max, (int)(i1 - i0));
        }
#endif
        float4 a[NV];
#pragma unroll
        for (int t = 0; t < NV; ++t)
            a[t] = (MODE == BSC_MODE_MAX) ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t total = 0;
        for (int64_t base = i0;; base += 64) {
            const int64_t k = base + lane;
            const bool in = k < n_pairs && code_sorted[k < n_pairs ? k : 0] == code;
            const u64 rec = in ? pair_rec[idx_sorted[k]] : 0ull;
            const int n_in = __popcll(__ballot(in));
            const uint32_t row_l = (uint32_t)(rec >> 32), cnt_l = (uint32_t)rec & 0xffffffu;
            const bool in_r = in && row_l >= row_lo && row_l < row_hi;
            const int j_lo = __popcll(__ballot(in && row_l < row_lo));
            const int n = j_lo + __popcll(__ballot(in_r));                // the lanes [j_lo, n) hold this pass's pairs of the chunk
            const bool past = __ballot(in && row_l >= row_hi) != 0ull;     // the rest of the segment belongs to later passes
            for (int j = j_lo; j < n; j += RF) {
                if constexpr (sizeof(TOK) == 2 && MODE != BSC_MODE_MAX) {
                    // bf16 rows, sum: v_dot2c_f32_bf16 with (m, 0) / (0, m) as the second operand adds m x one element of the
                    // pair to the f32 accumulator — widening and multiply-add in one instruction (the row loads are 8 bytes
                    // per lane; a separate shift / mask per element made this path slower than f32 rows).  m is exact in bf16
                    // up to 256; a larger multiplicity (a pair holds at most a tile's ~3300 points) goes byte by byte, each part exact.
                    uint2 xr[RF][NV];
                    uint32_t mi[RF];
#pragma unroll
                    for (int q = 0; q < RF; ++q) {
                        const int jj = j + q < n ? j + q : j;
                        const TOK *row = tokens + (int64_t)__builtin_amdgcn_readlane((int)row_l, jj) * D;
                        mi[q] = j + q < n ? (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, jj) : 0u;
#pragma unroll
                        for (int t = 0; t < NV; ++t) {
                            if (PAIRED && !((NV & 1) && t == NV - 1)) {
                                if (t & 1) continue;                            // loaded with its even partner
                                const int v8 = lane + 64 * (t >> 1);            // 8 columns at 8 * v8
                                const uint4 raw = (2 * v8 + 1 < D4) ? ((const uint4 *)row)[v8] : make_uint4(0u, 0u, 0u, 0u);
                                xr[q][t] = make_uint2(raw.x, raw.y);
                                xr[q][t + 1] = make_uint2(raw.z, raw.w);
                            } else {
                                const int v = lane + 64 * t;
                                xr[q][t] = (v < D4) ? ((const uint2 *)row)[v] : make_uint2(0u, 0u);
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < RF; ++q) {
                        // one exact bf16 factor per byte of the multiplicity (wave-uniform; one pass in the usual case)
                        for (uint32_t rem = mi[q], sh = 0; rem; rem >>= 8, sh += 8) {
                            const uint32_t part = (rem & 255u) << sh;
                            if (!part) continue;
                            const uint32_t lo = __float_as_uint((float)part) >> 16, hi = lo << 16;
                            const bf16x2_t blo = __builtin_bit_cast(bf16x2_t, lo), bhi = __builtin_bit_cast(bf16x2_t, hi);
#pragma unroll
                            for (int t = 0; t < NV; ++t) {
                                const bf16x2_t e01 = __builtin_bit_cast(bf16x2_t, xr[q][t].x), e23 = __builtin_bit_cast(bf16x2_t, xr[q][t].y);
                                a[t].x = __builtin_amdgcn_fdot2_f32_bf16(e01, blo, a[t].x, false);
                                a[t].y = __builtin_amdgcn_fdot2_f32_bf16(e01, bhi, a[t].y, false);
                                a[t].z = __builtin_amdgcn_fdot2_f32_bf16(e23, blo, a[t].z, false);
                                a[t].w = __builtin_amdgcn_fdot2_f32_bf16(e23, bhi, a[t].w, false);
                            }
                        }
                    }
                    continue;
                }
                float4 xv[RF][NV];
                float m[RF];
#pragma unroll
                for (int q = 0; q < RF; ++q) {
                    const int jj = j + q < n ? j + q : j;                  // padding repeats pair j with multiplicity 0
                    const TOK *row = tokens + (int64_t)__builtin_amdgcn_readlane((int)row_l, jj) * D;
                    m[q] = j + q < n ? (float)(uint32_t)__builtin_amdgcn_readlane((int)cnt_l, jj) : 0.f;
#pragma unroll
                    for (int t = 0; t < NV; ++t) {
                        const int v = lane + 64 * t;
                        xv[q][t] = (v < D4) ? load_tok4(row, v) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int q = 0; q < RF; ++q)
#pragma unroll
                    for (int t = 0; t < NV; ++t) {
                        if (MODE == BSC_MODE_MAX) {         // padding repeats a real row: max is idempotent
                            a[t].x = fmaxf(a[t].x, xv[q][t].x); a[t].y = fmaxf(a[t].y, xv[q][t].y);
                            a[t].z = fmaxf(a[t].z, xv[q][t].z); a[t].w = fmaxf(a[t].w, xv[q][t].w);
                        } else {
                            a[t].x = fmaf(m[q], xv[q][t].x, a[t].x); a[t].y = fmaf(m[q], xv[q][t].y, a[t].y);
                            a[t].z = fmaf(m[q], xv[q][t].z, a[t].z); a[t].w = fmaf(m[q], xv[q][t].w, a[t].w);
                        }
                    }
            }
            uint32_t cs = in_r ? cnt_l : 0u;
            for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o);
            total += cs;
            if (n_in < 64 || past) break;
        }
        const bool is_new = !later_pass && vid >= max_id_prev;
        if (total == 0 && !is_new) continue;                               // nothing of this voxel in this pass
        float4 *dst = (float4 *)(acc_g + vid * D);
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int v = acc_slot<NV, PAIRED>(t, lane);
            if (v < D4) {
                float4 o = a[t];
                if (!is_new) {
                    const float4 old = dst[v];
                    if (MODE == BSC_MODE_MAX) { o.x = fmaxf(o.x, old.x); o.y = fmaxf(o.y, old.y); o.z = fmaxf(o.z, old.z); o.w = fmaxf(o.w, old.w); }
                    else { o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                }
                dst[v] = o;
                sq = fmaf(o.x, o.x, sq); sq = fmaf(o.y, o.y, sq); sq = fmaf(o.z, o.z, sq); sq = fmaf(o.w, o.w, sq);
            }
        }
        if (rscale) {           // the same sum as k_row_scale forms (lane l: the columns 4 (l + 64 t) .., then the butterfly)
            for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            if (lane == 0) rscale[vid] = bsc_row_scale_of(sq);
        }
        if (lane == 0) acnt[vid] = (is_new ? 0 : acnt[vid]) + (int32_t)total;
    }
#ifdef BSC_REDUCE_PROFILE
    if ((blockIdx.x % 128) == 5 && threadIdx.x == 0)
        printf("reduce block %d wave 0: %lld clocks, %d voxels, %d pairs, longest %d\n", (int)blockIdx.x, clock64() - t_begin, n_vox, n_rows_p, n_max);
#endif
}

// ---- column-sliced per-voxel reduce: the regime where the token rows do not stay in the L2 ---------------------------------
// With one voxel per handful of points (no spatial coherence between a frame's pixels) the voxels running at one time on an XCD
// reference more distinct token rows than its 4 MB of L2 hold (the 1024 voxels of 32 CUs x 32 wavefronts: ~6 k rows of 1.5 KB), and
// two thirds of the row gathers go out to the Infinity Cache / HBM.  Here every XCD reduces only ONE column slice of the rows — 384 B
// of a 1536 B bf16 row (a whole number of 128 B lines) — for every voxel of its share: the same voxels in flight now keep a quarter
// of the bytes live in each L2 and the reuse between neighbouring voxels (a patch's frustum crosses them all) is served from it.
// A wavefront holds RPI = 64 / (slice bytes / 16) rows per load instruction (lane group `sub` takes the pairs sub, sub + RPI, ...),
// RF instructions in flight; the groups' partial sums are combined in lane-group order at the end: a fixed order, so the result
// does not depend on the schedule.  The pair records arrive gathered into sorted order (k_pairs_gather) — read once per slice.
__global__ __launch_bounds__(TPB) void k_pairs_gather(const uint32_t *__restrict__ idx_sorted, const u64 *__restrict__ pair_rec,
                                                      int64_t n, u64 *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) out[i] = pair_rec[idx_sorted[i]];
}

// voxel id of every segment of the sorted pair list
__global__ __launch_bounds__(TPB) void k_seg_vid(const uint32_t *__restrict__ code_sorted, const uint32_t *__restrict__ seg_start,
                                                 const int64_t *dscal, CellCode cc, const int32_t *__restrict__ occ,
                                                 int32_t *__restrict__ seg_vid)
{
    const int64_t nseg = dscal[DS_B_NPSEG];
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nseg; i += (int64_t)gridDim.x * TPB)
        seg_vid[i] = occ[code_to_cell(cc, (u64)code_sorted[seg_start[i]])];
}

// What bounds this kernel is instruction issue — one vector and one scalar instruction per cycle and CU — and, per visit of a
// voxel, a chain of dependent memory round trips (segment -> pair records -> token rows -> old accumulator).  So:
//  * one row slice per load instruction, 8 bytes per lane (a 384-byte slice = 48 lanes): a row's byte offset and multiplicity are
//    wave-uniform — computed once per 64-pair chunk in the lanes, fetched with v_readlane at a CONSTANT lane index (the chunk is
//    walked by fully unrolled batches of RF pairs) and used as the scalar offset of a buffer_load / the scalar operand of
//    v_dot2c_f32_bf16: per row slice two or three v_readlane, one load, one dot2 per element, and no address arithmetic at all;
//  * lanes past the end of a segment repeat its last row with multiplicity 0 (max: idempotent), so batches need no tail handling;
//  * the segment bounds and voxel id of the visit after the next arrive by scalar loads, the pair records of the next visit are
//    fetched under the row gathers of the current one, the old accumulator slice is the initial value of the sums;
//  * bf16 multiplicities go byte by byte (each part exact in bf16), the upper bytes only for a chunk that has them.
template <int MODE, typename TOK, int RF = 8, int WPE = 8>
__global__ __launch_bounds__(TPB, WPE) void k_dense_reduce_sliced(const u64 *__restrict__ rec_sorted, int64_t n_pairs,
                                                                  const uint32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_vid,
                                                                  const int64_t *dscal, const TOK *__restrict__ tokens, int D,
                                                                  float *__restrict__ acc_g, int32_t *__restrict__ acnt, int nslice,
                                                                  uint32_t tok_bytes)
{
    constexpr int E = 8 / (int)sizeof(TOK);                 // columns per 8-byte load
    constexpr bool DOT2 = sizeof(TOK) == 2 && MODE != BSC_MODE_MAX;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nseg = (int)dscal[DS_B_NPSEG];
    const int max_id_prev = (int)dscal[DS_MAX_ID_PREV];
    const int SC = D / nslice;                              // columns of a slice; lanes 0 .. SC / E - 1 hold them
    const bool active = lane * E < SC;
    const int xcd = blockIdx.x & 7, slice = xcd % nslice, part = xcd / nslice, nparts = 8 / nslice;
    const int col = slice * SC + (active ? lane : 0) * E;
    const uint32_t colb = (uint32_t)col * (uint32_t)sizeof(TOK);
    const uint32_t rowb = (uint32_t)D * (uint32_t)sizeof(TOK);
    // the token tile as a raw buffer: a gather is buffer_load_dwordx2 with the lane's column offset in a VGPR and the row's byte
    // offset in an SGPR
    const __amdgpu_buffer_rsrc_t tok_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tokens, 0, tok_bytes, 0x00020000);
    const int w_local = (int)(blockIdx.x >> 3) * (TPB / 64) + wave;
    const int w_per_xcd = (int)(gridDim.x >> 3) * (TPB / 64);
    const int chunk = (nseg + 63) / 64;                     // the XCD part walks the super-chunks part, part + nparts, ...
    const int n_super = 64 / nparts;
    // cursor of the segment fetch: visit number = cq * chunk + cr (advanced without divisions)
    int cq = w_local / (chunk > 0 ? chunk : 1), cr = w_local - cq * chunk;
    const int step_q = w_per_xcd / (chunk > 0 ? chunk : 1), step_r = w_per_xcd - step_q * chunk;

    // the next visit of this wavefront: first pair, pair count, voxel id (wave-uniform: scalar loads); n = 0 past the end
#define SL_META(i0_, n_, vid_)                                                                             \
    do {                                                                                                   \
        i0_ = 0; n_ = 0; vid_ = -1;                                                                        \
        if (cq < n_super) {                                                                                \
            const int sg_ = (part + nparts * cq) * chunk + cr;                                             \
            if (sg_ < nseg) {                                                                              \
                i0_ = (int)seg_start[sg_];                                                                 \
                n_ = (sg_ + 1 < nseg ? (int)seg_start[sg_ + 1] : (int)n_pairs) - i0_;                      \
                vid_ = seg_vid[sg_];                                                                       \
            }                                                                                              \
        }                                                                                                  \
        cq += step_q; cr += step_r;                                                                        \
        if (cr >= chunk) { cr -= chunk; ++cq; }                                                            \
    } while (0)
    // per chunk of up to 64 pair records: the lane's row offset and multiplicity operands
#define SL_CHUNK(nn_)                                                                                      \
    do {                                                                                                   \
        const uint32_t row_ = (uint32_t)(rec >> 32);                                                       \
        cnt_l = (int)((uint32_t)rec & 0xffffffu);                                                          \
        const uint32_t last_ = (uint32_t)__builtin_amdgcn_readlane((int)row_, (nn_) - 1);                  \
        roff_l = (int)((lane < (nn_) ? row_ : last_) * rowb);                                              \
        if constexpr (DOT2) {                                                                              \
            m0_l = (int)(__float_as_uint((float)(cnt_l & 0xff)) >> 16);                                    \
            big = __ballot(cnt_l > 255) != 0ull;                                                           \
        } else m0_l = (int)__float_as_uint((float)cnt_l);                                                  \
    } while (0)
#define SL_ISSUE(B_)                                                                                       \
    _Pragma("unroll") for (int q = 0; q < RF; ++q) {                                                       \
        const uint32_t off_ = (uint32_t)__builtin_amdgcn_readlane(roff_l, (B_) * RF + q);                  \
        x[q] = __builtin_amdgcn_raw_buffer_load_b64(tok_rsrc, colb, off_, 0);                              \
    }
#define SL_DOT2(B_, mb_)                                                                                   \
    _Pragma("unroll") for (int q = 0; q < RF; ++q) {                                                       \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_readlane(mb_, (B_) * RF + q), hi_ = lo_ << 16;     \
        const bf16x2_t blo_ = __builtin_bit_cast(bf16x2_t, lo_), bhi_ = __builtin_bit_cast(bf16x2_t, hi_); \
        const uint32_t w0_ = x[q].x, w1_ = x[q].y;      /* (bit_cast of a vector ELEMENT reads element 0 whatever the index) */ \
        const bf16x2_t e01_ = __builtin_bit_cast(bf16x2_t, w0_), e23_ = __builtin_bit_cast(bf16x2_t, w1_); \
        a[0] = __builtin_amdgcn_fdot2_f32_bf16(e01_, blo_, a[0], false);                                   \
        a[1 % E] = __builtin_amdgcn_fdot2_f32_bf16(e01_, bhi_, a[1 % E], false);                           \
        a[2 % E] = __builtin_amdgcn_fdot2_f32_bf16(e23_, blo_, a[2 % E], false);                           \
        a[3 % E] = __builtin_amdgcn_fdot2_f32_bf16(e23_, bhi_, a[3 % E], false);                           \
    }
#define SL_ACCUM(B_)                                                                                       \
    if constexpr (DOT2) {                                                                                  \
        SL_DOT2(B_, m0_l)                                                                                  \
        if (big) {                                                                                         \
            const int m1_l = (int)(__float_as_uint((float)(cnt_l & 0xff00)) >> 16);                        \
            const int m2_l = (int)(__float_as_uint((float)(cnt_l & 0xff0000)) >> 16);                      \
            SL_DOT2(B_, m1_l)                                                                              \
            SL_DOT2(B_, m2_l)                                                                              \
        }                                                                                                  \
    } else {                                                                                               \
        _Pragma("unroll") for (int q = 0; q < RF; ++q) {                                                   \
            const float m_ = __uint_as_float((uint32_t)__builtin_amdgcn_readlane(m0_l, (B_) * RF + q));    \
            float v_[4];                                                                                   \
            if constexpr (sizeof(TOK) == 2) {                                                              \
                v_[0] = __uint_as_float(x[q].x << 16); v_[1] = __uint_as_float(x[q].x & 0xffff0000u);      \
                v_[2] = __uint_as_float(x[q].y << 16); v_[3] = __uint_as_float(x[q].y & 0xffff0000u);      \
            } else {                                                                                       \
                v_[0] = __uint_as_float(x[q].x); v_[1] = __uint_as_float(x[q].y); v_[2] = 0.f; v_[3] = 0.f; \
            }                                                                                              \
            _Pragma("unroll") for (int e = 0; e < E; ++e)                                                  \
                a[e] = (MODE == BSC_MODE_MAX) ? fmaxf(a[e], v_[e]) : fmaf(m_, v_[e], a[e]);                \
        }                                                                                                  \
    }

    int i0_1, vid_1, i0_2, vid_2, n_1, n_2;
    SL_META(i0_1, n_1, vid_1);
    SL_META(i0_2, n_2, vid_2);
    u64 r1 = lane < n_1 ? rec_sorted[i0_1 + lane] : 0ull;
    while (n_1 > 0 || n_2 > 0 || cq < n_super) {
        const int i0 = i0_1, vid = vid_1, n = n_1;
        u64 rec = r1;
        i0_1 = i0_2; n_1 = n_2; vid_1 = vid_2;
        if (n <= 0) {                                       // a hole of the walk: keep the pipeline moving
            r1 = lane < n_1 ? rec_sorted[i0_1 + lane] : 0ull;
            SL_META(i0_2, n_2, vid_2);
            continue;
        }
        const bool is_new = vid >= max_id_prev;
        float *dst = acc_g + (int64_t)vid * D + col;
        float a[E];
#pragma unroll
        for (int e = 0; e < E; ++e) a[e] = (MODE == BSC_MODE_MAX) ? -INFINITY : 0.f;
        if (!is_new && active) {
            if constexpr (E == 4) { const float4 o = *(const float4 *)dst; a[0] = o.x; a[1 % E] = o.y; a[2 % E] = o.z; a[3 % E] = o.w; }
            else { const float2 o = *(const float2 *)dst; a[0] = o.x; a[1 % E] = o.y; }
        }
        u32x2_t x[RF];
        int roff_l, cnt_l, m0_l;
        bool big = false;
        int nn = n < 64 ? n : 64;
        SL_CHUNK(nn);
        uint32_t cs = (uint32_t)cnt_l;              // lanes past the segment hold 0
        SL_ISSUE(0)
        // under the first row gathers: the next visit's pair records, the segment of the visit after that
        r1 = lane < n_1 ? rec_sorted[i0_1 + lane] : 0ull;
        SL_META(i0_2, n_2, vid_2);
        SL_ACCUM(0)
#pragma unroll
        for (int B = 1; B < 64 / RF; ++B) {
            if (nn > B * RF) {
                SL_ISSUE(B)
                SL_ACCUM(B)
            }
        }
        for (int base = 64; base < n; base += 64) {
            rec = base + lane < n ? rec_sorted[i0 + base + lane] : 0ull;
            nn = n - base < 64 ? n - base : 64;
            SL_CHUNK(nn);
            cs += (uint32_t)cnt_l;
#pragma unroll
            for (int B = 0; B < 64 / RF; ++B) {
                if (nn > B * RF) {
                    SL_ISSUE(B)
                    SL_ACCUM(B)
                }
            }
        }
        if (active) {
            if constexpr (E == 4) *(float4 *)dst = make_float4(a[0], a[1 % E], a[2 % E], a[3 % E]);
            else *(float2 *)dst = make_float2(a[0], a[1 % E]);
        }
        if (slice == 0) {
            for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o);
            if (lane == 0) acnt[vid] = (is_new ? 0 : acnt[vid]) + (int32_t)cs;
        }
    }
#undef SL_META
#undef SL_ISSUE
#undef SL_ACCUM
#undef SL_DOT2
#undef SL_CHUNK
}

// column slices of the sliced reduce: the fewest of 1 / 2 / 4 / 8 for which a slice is a whole number of 128-byte lines and one
// load instruction of 8 bytes per lane covers it (<= 512 bytes); 0: no such split (the per-voxel kernel stays)
static int reduce_slices(int D, int tok_bytes)
{
    const int64_t rowb = (int64_t)D * tok_bytes;
    for (int ns = 1; ns <= 8; ns <<= 1)
        if (rowb % (128 * ns) == 0 && rowb / ns <= 512) return ns;
    return 0;
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

static CellCode make_cell_code(const bsc_ctx *x, int cb)
{
    CellCode cc;
    cc.gs = x->c.grid_size; cc.nh = x->nh;
    const int a = ceil_log2_u64((uint64_t)x->c.grid_size), b = ceil_log2_u64((uint64_t)x->nh);
    cc.ab = a > b ? a : b;
    if (cc.ab < 1) cc.ab = 1;
    if (cc.ab > 21 || 3 * cc.ab + cb > 62) cc.ab = 0;          // linear cell index (< 2^31) instead
    return cc;
}
static inline int cell_code_bits(const CellCode &cc) { return cc.ab ? 3 * cc.ab : 31; }
// the patch-tile path keeps the code in 32 bits
static CellCode make_cell_code32(const bsc_ctx *x)
{
    CellCode cc = make_cell_code(x, 0);
    if (cc.ab > 10) cc.ab = 0;
    return cc;
}
template <int MODE, typename TOK>
static void launch_dense(bsc_ctx *x, int64_t n_pairs, const TOK *tokens, int pb, int cb)
{
    const CellCode cc = make_cell_code(x, cb);
    const int D = x->c.token_dim;
    const int nv = (D / 4 + 63) / 64;
    const dim3 grid(256 * 8), block(TPB);
#define LD(NV)                                                                                                          \
    hipLaunchKernelGGL((k_dense_reduce<NV, MODE, TOK>), grid, block, 0, x->stream, x->pair_key_b, x->pair_cnt_b, n_pairs,    \
                       (const uint32_t *)x->pseg_start, x->dscal, tokens, x->g2, D, x->acc, x->acnt, pb, cb, cc, x->occ)
    if (nv <= 1) LD(1);
    else if (nv == 2) LD(2);
    else if (nv == 3) LD(3);
    else if (nv == 4) LD(4);
    else LD(8);
#undef LD
}

// dense modes: the per-tile (cell, frame, patch) pairs of the batch
bsc_status launch_keys_pairs(bsc_ctx *x, int64_t P, int n_frames, bool all_pixels, const uint32_t *p_patf)
{
    if (x->c.mode == BSC_MODE_EXACT) return BSC_OK;
    const int H = x->c.height, W = x->c.width;
    x->pair_path = 0;
    if (all_pixels && p_patf == nullptr && x->patch_tiles) {
        // patch-aligned tiles: pairs unique by construction, sort key = block of 2^bb Morton-adjacent cells
        x->pair_path = 1;
        const int g2 = x->g2;
        const int64_t tiles = (int64_t)n_frames * g2, N = (int64_t)H * W;
        if (tiles > x->max_tiles) { bsc_set_error("launch_keys_pairs: %lld tiles > %lld", (long long)tiles, (long long)x->max_tiles); return BSC_E_CAPACITY; }
        const dim3 grid((unsigned)tiles), block(TPB);
        const dim3 pgrid((unsigned)(tiles < 256 * 4 ? tiles : 256 * 4));      // persistent: 4 workgroups per CU (41 KB of LDS each)
        uint32_t *pair_idx = (uint32_t *)x->pair_key_b;
        hipLaunchKernelGGL(k_patch_pairs, pgrid, block, 0, x->stream, W, N, x->c.patch_grid, tiles, x->pt_rect, x->pt_off,
                           make_cell_code32(x), x->p_cell, x->pstage_key, x->pstage_cnt, x->tile_cnt);
        BSC_TRY(prim_exclusive_sum_i32(x, x->tile_cnt, x->tile_off, (size_t)tiles));
        hipLaunchKernelGGL(k_patch_compact, grid, block, 0, x->stream, tiles, g2, N, x->pt_off, x->tile_cnt, x->tile_off,
                           x->pstage_key, x->pstage_cnt, x->pair_key_a, x->pair_cnt_a, pair_idx, x->pair_cap, x->dscal);
        BSC_HIP(hipGetLastError());
        return BSC_OK;
    }
    const int tx_n = (W + 31) / 32, ty_n = (H + 31) / 32;
    const int64_t tiles = all_pixels ? (int64_t)n_frames * tx_n * ty_n : (P + PT_TILE - 1) / PT_TILE;
    if (tiles > x->max_tiles) { bsc_set_error("launch_keys_pairs: %lld tiles > %lld", (long long)tiles, (long long)x->max_tiles); return BSC_E_CAPACITY; }
    const int pb = code_patch_bits(x), cb = code_bits(x, n_frames);
    const dim3 grid((unsigned)tiles), block(TPB);
    hipLaunchKernelGGL(k_keys_pairs, grid, block, 0, x->stream, P, all_pixels ? 1 : 0, H, W, tx_n, ty_n, make_cell_code(x, cb),
                       x->p_cell, p_patf, x->pat_x, x->pat_y, x->c.patch_grid, x->pstage_key, x->pstage_cnt, x->tile_cnt, pb, cb);
    BSC_TRY(prim_exclusive_sum_i32(x, x->tile_cnt, x->tile_off, (size_t)tiles));
    hipLaunchKernelGGL(k_pair_compact, grid, block, 0, x->stream, tiles, x->tile_cnt, x->tile_off, x->pstage_key,
                       x->pstage_cnt, x->pair_key_a, x->pair_cnt_a, x->pair_cap, x->dscal);
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
    if (x->pair_path == 1) {
        // pairs of the patch-aligned tiles: sort by the cell code alone (stable: a voxel's pairs stay in frame / patch
        // order), segment heads, one wavefront per voxel
        const int D = x->c.token_dim, nv = (D / 4 + 63) / 64;
        const CellCode cc = make_cell_code32(x);
        uint32_t *pair_idx = (uint32_t *)x->pair_key_b, *idx_sorted = pair_idx + x->pair_cap;
        stat_begin(x, BSC_STAT_PAIRSORT);
        if (x->radix_intree)
            BSC_TRY(radix_sort_pairs_u32(x, &x->rx_main, s, x->pair_cnt_a, x->pair_cnt_b, pair_idx, idx_sorted, (size_t)n_pairs, 0, cell_code_bits(cc)));
        else
            BSC_TRY(prim_sort_pairs_u32_onesweep(x, x->pair_cnt_a, x->pair_cnt_b, pair_idx, idx_sorted, (size_t)n_pairs, 0, cell_code_bits(cc)));
        BSC_TRY(compact_heads_u32(x, x->pair_cnt_b, n_pairs, x->pseg_start, x->dscal + DS_B_NPSEG));
        stat_end(x, BSC_STAT_PAIRSORT, 0.0);
        BSC_HIP(hipEventRecord(x->ev_psort, s));        // the rgb chain of this call starts behind the pair sort (launch_pending_chain)
        x->ev_psort_valid = true;
        stat_begin(x, BSC_STAT_DENSE);
        const dim3 grid(256 * 8), block(TPB);
        {
            // BSC_SLICED_MIN_PAIRS=<n>: pair lists of n or more take the column-sliced kernel.  Off by default: on the one-voxel-per-
            // point workload it halves the bytes fetched past the L2 (38 -> 19 GB per call) and still runs 9.1 ms against 8.5 ms for
            // the per-voxel kernel — four visits per voxel, each bound by instruction issue (profiles/r04_iid_reduce.txt)
            const int64_t sliced_min = getenv("BSC_SLICED_MIN_PAIRS") ? atoll(getenv("BSC_SLICED_MIN_PAIRS")) : INT64_MAX;
            const int ns = reduce_slices(D, token_dtype == BSC_TOK_BF16 ? 2 : 4);
            const int64_t tile_bytes = (int64_t)n_frames * x->g2 * D * (token_dtype == BSC_TOK_BF16 ? 2 : 4);     // < 4 GB: 32-bit row offsets
            if (ns && n_pairs >= sliced_min && tile_bytes < ((int64_t)1 << 32)) {
                u64 *rec_sorted = x->pstage_key;           // free since k_patch_compact
                int32_t *seg_vid = (int32_t *)x->pair_cnt_a;    // the unsorted codes: free since the sort
                hipLaunchKernelGGL(k_pairs_gather, dim3(256 * 8), block, 0, s, idx_sorted, x->pair_key_a, n_pairs, rec_sorted);
                hipLaunchKernelGGL(k_seg_vid, dim3(256 * 4), block, 0, s, x->pair_cnt_b, (const uint32_t *)x->pseg_start, x->dscal, cc,
                                   x->occ, seg_vid);
#define LS(MODEV, TOKT)                                                                                                         \
    hipLaunchKernelGGL((k_dense_reduce_sliced<MODEV, TOKT>), grid, block, 0, s, rec_sorted, n_pairs,                            \
                       (const uint32_t *)x->pseg_start, seg_vid, x->dscal, (const TOKT *)tokens, D, x->acc, x->acnt, ns, \
                       (uint32_t)tile_bytes)
                if (token_dtype == BSC_TOK_BF16) { if (x->c.mode == BSC_MODE_MEAN) LS(BSC_MODE_MEAN, bf16_t); else LS(BSC_MODE_MAX, bf16_t); }
                else { if (x->c.mode == BSC_MODE_MEAN) LS(BSC_MODE_MEAN, float); else LS(BSC_MODE_MAX, float); }
#undef LS
                stat_end(x, BSC_STAT_DENSE, 0.0);
                hipLaunchKernelGGL(k_dense_counters, dim3(1), dim3(64), 0, s, x->dscal);
                BSC_HIP(hipGetLastError());
                return BSC_OK;
            }
        }
        // A call whose token tile is larger than the 256 MB MALL is reduced in passes over slices of its frames, each slice's token
        // rows at most BSC_REDUCE_PASS_BYTES (default 256 MB; 0: always one pass): 768 frames of 14 x 14 x 768 f32 rows are 462 MB ->
        // two passes.  Measured (round 5, one box, alternating runs): the kernel itself 1.73 -> 1.67 ms (it is bound by the L2 -> CU rate
        // of its row gathers, 17 GB per call, not by where the rows come from), but the isolated call + sync 6.09-6.17 -> 5.83-5.85 ms
        // and the call inside the pipeline 5.59 -> 5.51: the call as a whole is bound by its HBM traffic (profiles/README.md) and
        // the rows of a slice are gathered from the MALL instead of from HBM.  Smaller slices lose: every pass visits every voxel
        // again (160 MB: kernel 1.86 ms; 115 MB: 2.05; 58 MB: 2.31).  A voxel's sum is then formed slice by slice (f32 order differs
        // from the one-pass sum in the last bits; max is exact): tests/test_gpu_edges.py.
        const int64_t pass_b = getenv("BSC_REDUCE_PASS_BYTES") ? atoll(getenv("BSC_REDUCE_PASS_BYTES")) : ((int64_t)256 << 20);   // read per call
        const int64_t tile_b = (int64_t)n_frames * x->g2 * D * (token_dtype == BSC_TOK_BF16 ? 2 : 4);
        int n_pass = pass_b > 0 ? (int)((tile_b + pass_b - 1) / pass_b) : 1;
        n_pass = n_pass < 1 ? 1 : (n_pass > 8 ? 8 : n_pass);
        if (n_pass > n_frames) n_pass = n_frames;
        const int frames_per_pass = (n_frames + n_pass - 1) / n_pass;
        uint32_t row_lo = 0, row_hi = 0xffffffffu;
        int later_pass = 0;
        // the rows' scale / inverse norm for the batched scan, while they are in registers (every voxel of the map was created by a
        // point that left a pair: every row below max_id has been written by this kernel, unless an import / merge / reset wrote it)
        float2 *rscale = nullptr;
        if (x->l_rscale && x->l_rscale_cap >= (int64_t)sizeof(float2) * ((int64_t)x->c.voxel_capacity + 1) && !getenv("BSC_NO_RSCALE_IN_REDUCE")) {
            rscale = x->l_rscale;
            x->rscale_from_reduce = true;
        }
#define LVP(NVV, MODEV, TOKT, PAIR)                                                                                            \
    hipLaunchKernelGGL((k_dense_reduce_voxels<NVV, MODEV, TOKT, PAIR>), grid, block, 0, s, x->pair_cnt_b, idx_sorted, x->pair_key_a,  \
                       n_pairs, (const uint32_t *)x->pseg_start, x->dscal, (const TOKT *)tokens, D, x->acc, x->acnt, cc, x->occ,     \
                       row_lo, row_hi, later_pass, rscale)
#define LV(NVV, MODEV, TOKT) LVP(NVV, MODEV, TOKT, false)
#define LVM(NVV)                                                                                   \
    do {                                                                                           \
        if (token_dtype == BSC_TOK_BF16) {                                                         \
            if (x->c.mode == BSC_MODE_MEAN) { if (D % 8 == 0 && !getenv("BSC_REDUCE_PLAIN")) LVP(NVV, BSC_MODE_MEAN, bf16_t, true); else LV(NVV, BSC_MODE_MEAN, bf16_t); } \
            else LV(NVV, BSC_MODE_MAX, bf16_t); \
        } else {                                                                                   \
            if (x->c.mode == BSC_MODE_MEAN) LV(NVV, BSC_MODE_MEAN, float); else LV(NVV, BSC_MODE_MAX, float);   \
        }                                                                                          \
    } while (0)
        for (int ps = 0; ps < n_pass; ++ps) {
            row_lo = (uint32_t)((int64_t)ps * frames_per_pass * x->g2);
            row_hi = ps + 1 == n_pass ? 0xffffffffu : (uint32_t)((int64_t)(ps + 1) * frames_per_pass * x->g2);
            later_pass = ps > 0;
            if (nv <= 1) LVM(1); else if (nv == 2) LVM(2); else if (nv == 3) LVM(3); else if (nv == 4) LVM(4); else LVM(8);
        }
#undef LVM
#undef LVP
#undef LV
        stat_end(x, BSC_STAT_DENSE, 0.0);
        hipLaunchKernelGGL(k_dense_counters, dim3(1), dim3(64), 0, s, x->dscal);
        BSC_HIP(hipGetLastError());
        return BSC_OK;
    }
    const int pb = code_patch_bits(x), cb = code_bits(x, n_frames);
    stat_begin(x, BSC_STAT_PAIRSORT);
    BSC_TRY(prim_sort_pairs_onesweep(x, x->pair_key_a, x->pair_key_b, x->pair_cnt_a, x->pair_cnt_b, (size_t)n_pairs, 0,
                                     cb + cell_code_bits(make_cell_code(x, cb))));
    // voxel segments of the sorted list, already in Morton order of the cells
    BSC_TRY(compact_heads_u64(x, x->pair_key_b, n_pairs, cb, x->pseg_start, x->dscal + DS_B_NPSEG));
    stat_end(x, BSC_STAT_PAIRSORT, 0.0);
    stat_begin(x, BSC_STAT_DENSE);
    if (token_dtype == BSC_TOK_BF16) {
        if (x->c.mode == BSC_MODE_MEAN) launch_dense<BSC_MODE_MEAN>(x, n_pairs, (const bf16_t *)tokens, pb, cb);
        else launch_dense<BSC_MODE_MAX>(x, n_pairs, (const bf16_t *)tokens, pb, cb);
    } else {
        if (x->c.mode == BSC_MODE_MEAN) launch_dense<BSC_MODE_MEAN>(x, n_pairs, (const float *)tokens, pb, cb);
        else launch_dense<BSC_MODE_MAX>(x, n_pairs, (const float *)tokens, pb, cb);
    }
    stat_end(x, BSC_STAT_DENSE, 0.0);   // bytes are derived from the device counters (voxel rows, new rows, pairs)
    hipLaunchKernelGGL(k_dense_counters, dim3(1), dim3(64), 0, s, x->dscal);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}
