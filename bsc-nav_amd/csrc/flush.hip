// flush.hip — update_memory_dist_base (memory_2.py:326-358) for the device-resident token cache.
//
// Reference semantics: for EVERY cache row i in order (unused rows are zero vectors at [0,0,0]):
//   group missing -> create with this token; fewer than cache_size tokens -> append;
//   otherwise overwrite slot random.choice(range(cache_size)) — one Python RNG draw per such row.
// Parallel form: rows are radix-sorted by (store entry, row) so each voxel's rows are contiguous and in
// order; rank inside the segment + tokens already stored decide append vs. "hit"; hit rows get their
// index in the draw stream by an exclusive scan in ROW order; the host then produces exactly n_hits
// draws (keeping Python's RNG stream aligned) and the last hit per (voxel, slot) wins.
#include "bsc_internal.h"

#define TPB 256

__global__ __launch_bounds__(TPB) void k_flush_keys(int n_rows, const int32_t *__restrict__ cache_pos,
                                                    const int32_t *__restrict__ occ, int gs, int nh, int vcap,
                                                    u64 *__restrict__ keys)
{
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n_rows) return;
    const int32_t r = cache_pos[3 * i], c = cache_pos[3 * i + 1], h = cache_pos[3 * i + 2];
    const int64_t cell = ((int64_t)r * gs + c) * nh + h;
    // entry vcap is the group "grid_0_0_0": zero rows land there even when no point ever did
    int64_t e = (cell == 0) ? vcap : occ[cell];
    if (e < 0) e = vcap;   // cannot happen for rows written by k_append; keeps the key well-formed
    keys[i] = ((u64)e << 20) | (u64)i;
}

__global__ __launch_bounds__(TPB) void k_flush_heads(int n_rows, const u64 *__restrict__ keys,
                                                     int32_t *__restrict__ headflag)
{
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n_rows) return;
    const bool head = (i == 0) || ((keys[i] >> 20) != (keys[i - 1] >> 20));
    headflag[i] = head ? i : 0;
}

__global__ __launch_bounds__(TPB) void k_flush_plan(int n_rows, const u64 *__restrict__ keys,
                                                    const int32_t *__restrict__ headpos,
                                                    const int32_t *__restrict__ store_cnt, int32_t *__restrict__ store_rows,
                                                    int cache_size, int64_t *dscal, int64_t token_cap,
                                                    int32_t *__restrict__ rowdst, int32_t *__restrict__ hit,
                                                    int32_t *__restrict__ rowseg, int32_t *__restrict__ rowe)
{
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n_rows) return;
    const u64 key = keys[i];
    const int32_t e = (int32_t)(key >> 20);
    const int32_t r = (int32_t)(key & 0xfffffull);
    const int32_t hp = headpos[i];
    const int32_t rank = i - hp;
    const int32_t c0 = store_cnt[e];
    rowseg[r] = hp;
    rowe[r] = e;
    if (c0 + rank < cache_size) {       // :335-349 create / append
        const int64_t prow = (int64_t)atomicAdd((u64 *)&dscal[DS_POOL_N], 1ull);
        if (prow >= token_cap) {
            dscal[DS_ERROR] = 2;
            rowdst[r] = -1;
        } else {
            store_rows[(int64_t)e * cache_size + c0 + rank] = (int32_t)prow;
            rowdst[r] = (int32_t)prow;
        }
        hit[r] = 0;
    } else {                            // :351-354 replacement, slot decided by the host draw
        rowdst[r] = -1;
        hit[r] = 1;
    }
}

__global__ __launch_bounds__(TPB) void k_flush_counts(int n_rows, const u64 *__restrict__ keys,
                                                      const int32_t *__restrict__ headpos, int32_t *store_cnt,
                                                      int cache_size)
{
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n_rows) return;
    const bool tail = (i == n_rows - 1) || ((keys[i] >> 20) != (keys[i + 1] >> 20));
    if (!tail) return;
    const int32_t e = (int32_t)(keys[i] >> 20);
    const int32_t m = i - headpos[i] + 1;
    const int32_t c = store_cnt[e] + m;
    store_cnt[e] = c < cache_size ? c : cache_size;
}

// one wavefront per cache row with a destination: copy token + distance into the pool
__global__ __launch_bounds__(TPB) void k_flush_copy(int n_rows, const int32_t *__restrict__ rowdst,
                                                    const float *__restrict__ cache_f, const float *__restrict__ cache_d,
                                                    int D, float *__restrict__ pool, float *__restrict__ pool_d)
{
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * TPB + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    const int32_t dstrow = rowdst[r];
    if (dstrow < 0) return;
    const float4 *src = (const float4 *)(cache_f + (int64_t)r * D);
    float4 *dst = (float4 *)(pool + (int64_t)dstrow * D);
    for (int v = lane; v < (D >> 2); v += 64) dst[v] = src[v];
    if (lane == 0) pool_d[dstrow] = cache_d[r];
}

__global__ void k_flush_nhits(int n_rows, const int32_t *hit, const int32_t *hidx, int64_t *dscal)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) dscal[DS_N_HITS] = (int64_t)hidx[n_rows - 1] + hit[n_rows - 1];
}

__global__ __launch_bounds__(TPB) void k_flush_vote(int n_rows, const int32_t *__restrict__ hit,
                                                    const int32_t *__restrict__ hidx, const uint32_t *__restrict__ draws,
                                                    const int32_t *__restrict__ rowseg, int cache_size, int32_t *win,
                                                    int32_t *__restrict__ rowdst)
{
    const int r = blockIdx.x * TPB + threadIdx.x;
    if (r >= n_rows) return;
    if (!hit[r]) {
        rowdst[r] = -1;   // already copied by the append pass; must not race with a winner of the same slot
        return;
    }
    const uint32_t k = draws[hidx[r]];
    atomicMax(&win[(int64_t)rowseg[r] * cache_size + k], r);   // the LAST row drawing slot k keeps it
}

__global__ __launch_bounds__(TPB) void k_flush_winners(int n_rows, const int32_t *__restrict__ hit,
                                                       const int32_t *__restrict__ hidx,
                                                       const uint32_t *__restrict__ draws,
                                                       const int32_t *__restrict__ rowseg,
                                                       const int32_t *__restrict__ rowe,
                                                       const int32_t *__restrict__ store_rows, int cache_size,
                                                       const int32_t *__restrict__ win, int32_t *__restrict__ rowdst)
{
    const int r = blockIdx.x * TPB + threadIdx.x;
    if (r >= n_rows || !hit[r]) return;
    const uint32_t k = draws[hidx[r]];
    if (win[(int64_t)rowseg[r] * cache_size + k] == r) rowdst[r] = store_rows[(int64_t)rowe[r] * cache_size + k];
}

// token pool of at least `need` rows: new allocation of max(2 x capacity, need), rows in use copied, old pool freed
bsc_status grow_token_pool(bsc_ctx *x, int64_t need)
{
    if (need <= x->c.token_capacity) return BSC_OK;
    const int64_t D = x->c.token_dim;
    int64_t cap = 2 * x->c.token_capacity;
    if (cap < need) cap = need;
    float *pool = nullptr, *pool_d = nullptr;
    hipError_t e = hipMalloc((void **)&pool, sizeof(float) * (size_t)cap * D);
    if (e == hipSuccess) e = hipMalloc((void **)&pool_d, sizeof(float) * (size_t)cap);
    if (e != hipSuccess) {
        if (pool) (void)hipFree(pool);
        bsc_set_error("token store: cannot grow the pool from %lld to %lld rows of %lld floats (%s)", (long long)x->c.token_capacity,
                      (long long)cap, (long long)D, hipGetErrorString(e));
        return BSC_E_CAPACITY;
    }
    BSC_TRY(sync_all(x));                        // nothing in flight reads the old pool
    if (x->pool_n_host > 0) {
        BSC_HIP(hipMemcpy(pool, x->pool, sizeof(float) * (size_t)x->pool_n_host * D, hipMemcpyDeviceToDevice));
        BSC_HIP(hipMemcpy(pool_d, x->pool_d, sizeof(float) * (size_t)x->pool_n_host, hipMemcpyDeviceToDevice));
    }
    (void)hipFree(x->pool);
    (void)hipFree(x->pool_d);
    x->pool = pool;
    x->pool_d = pool_d;
    x->c.token_capacity = cap;
    return BSC_OK;
}

bsc_status flush_cache(bsc_ctx *x, bsc_draw_fn draw, void *user)
{
    const int n = x->c.iter_size, D = x->c.token_dim, cs = x->c.cache_size;
    hipStream_t s = x->stream;
    const dim3 block(TPB), grid((n + TPB - 1) / TPB), wgrid((unsigned)(((int64_t)n * 64 + TPB - 1) / TPB));
    const int ebits = ceil_log2_u64((uint64_t)x->c.voxel_capacity + 2);
    // A flush appends at most one pool row per cache row.  The reference's HDF5 store is unbounded (memory_2.py:330-354), so
    // the pool GROWS here when this flush could overflow it — before anything is changed: a flush triggered from the middle
    // of an ingest call (memory_2.py:880-881) must never fail half way through the frame.
    BSC_TRY(grow_token_pool(x, x->pool_n_host + n));
    hipLaunchKernelGGL(k_flush_keys, grid, block, 0, s, n, x->cache_pos, x->occ, x->c.grid_size, x->nh,
                       x->c.voxel_capacity, x->f_keys_a);
    BSC_TRY(prim_sort_keys(x, x->f_keys_a, x->f_keys_b, (size_t)n, 0, 20 + ebits));
    hipLaunchKernelGGL(k_flush_heads, grid, block, 0, s, n, x->f_keys_b, x->f_hit /*tmp*/);
    BSC_TRY(prim_inclusive_max_i32(x, x->f_hit, x->f_headpos, (size_t)n));
    hipLaunchKernelGGL(k_flush_plan, grid, block, 0, s, n, x->f_keys_b, x->f_headpos, x->store_cnt, x->store_rows, cs,
                       x->dscal, x->c.token_capacity, x->f_rowdst, x->f_hit, x->f_rowseg, x->f_rowe);
    hipLaunchKernelGGL(k_flush_counts, grid, block, 0, s, n, x->f_keys_b, x->f_headpos, x->store_cnt, cs);
    hipLaunchKernelGGL(k_flush_copy, wgrid, block, 0, s, n, x->f_rowdst, x->cache_f, x->cache_d, D, x->pool, x->pool_d);
    BSC_TRY(prim_exclusive_sum_i32(x, x->f_hit, x->f_hidx, (size_t)n));
    hipLaunchKernelGGL(k_flush_nhits, dim3(1), dim3(64), 0, s, n, x->f_hit, x->f_hidx, x->dscal);
    BSC_HIP(hipGetLastError());
    BSC_TRY(read_scalars(x));
    if (x->hscal[DS_ERROR]) {
        bsc_set_error("capacity exceeded during flush (code %lld; token_capacity=%lld)", (long long)x->hscal[DS_ERROR],
                      (long long)x->c.token_capacity);
        return BSC_E_CAPACITY;
    }
    const int64_t n_hits = x->hscal[DS_N_HITS];
    if (n_hits > 0) {
        if (!draw) {
            bsc_set_error("flush met %lld full voxels but no draw callback was given", (long long)n_hits);
            return BSC_E_INVALID;
        }
        uint32_t *h = (uint32_t *)malloc(sizeof(uint32_t) * n_hits);
        draw(user, (uint32_t)n_hits, h);   // memory_2.py:352, in row order
        for (int64_t i = 0; i < n_hits; ++i)
            if (h[i] >= (uint32_t)cs) {
                free(h);
                bsc_set_error("draw callback returned %u >= cache_size", h[i]);
                return BSC_E_INVALID;
            }
        hipError_t e = hipMemcpyAsync(x->f_draws, h, sizeof(uint32_t) * n_hits, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        free(h);
        BSC_HIP(e);
        BSC_HIP(hipMemsetAsync(x->f_win, 0xff, sizeof(int32_t) * (size_t)n * cs, s));
        hipLaunchKernelGGL(k_flush_vote, grid, block, 0, s, n, x->f_hit, x->f_hidx, x->f_draws, x->f_rowseg, cs, x->f_win,
                           x->f_rowdst);
        // rowdst is -1 for every hit row; winners get their slot's pool row, then the same copy kernel runs
        hipLaunchKernelGGL(k_flush_winners, grid, block, 0, s, n, x->f_hit, x->f_hidx, x->f_draws, x->f_rowseg, x->f_rowe,
                           x->store_rows, cs, x->f_win, x->f_rowdst);
        hipLaunchKernelGGL(k_flush_copy, wgrid, block, 0, s, n, x->f_rowdst, x->cache_f, x->cache_d, D, x->pool,
                           x->pool_d);
    }
    // _reinit_cache (memory_2.py:724-729)
    BSC_HIP(hipMemsetAsync(x->cache_f, 0, sizeof(float) * (size_t)n * D, s));
    BSC_HIP(hipMemsetAsync(x->cache_pos, 0, sizeof(int32_t) * (size_t)n * 3, s));
    BSC_HIP(hipMemsetAsync(x->cache_d, 0, sizeof(float) * (size_t)n, s));
    BSC_HIP(hipGetLastError());
    x->iter_id = 0;
    x->n_flush++;
    x->names_dirty = true; x->row_scale_dirty = true;
    x->pool_n_host = x->hscal[DS_POOL_N];
    return BSC_OK;
}
