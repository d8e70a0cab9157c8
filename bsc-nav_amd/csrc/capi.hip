// capi.hip — the extern "C" surface of libbscnav.so (include/bscnav.h): context lifetime, HBM state layout,
// import/export of the reference's on-disk arrays, and dispatch into the kernel pipelines.
#include "bsc_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <thread>
#include <vector>

#define BSC_H2D_NBUF 4                          // pinned staging buffers of h2d_pipelined (imports from pageable host memory)
static const size_t BSC_H2D_CHUNK = (size_t)(getenv("BSC_H2D_CHUNK_MB") ? atoi(getenv("BSC_H2D_CHUNK_MB")) : 64) << 20;

#define TPB 256

static thread_local char g_err[512] = "";

void bsc_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *bsc_last_error(void) { return g_err; }
extern "C" const char *bsc_version(void) { return "bscnav 0.1 (gfx950)"; }

bsc_status sync_all(bsc_ctx *x)
{
    BSC_TRY(launch_pending_chain(x));          // the rgb chain of the last ingest is deferred until someone needs it
    BSC_HIP(hipStreamSynchronize(x->stream));
    BSC_HIP(hipStreamSynchronize(x->side));
    return BSC_OK;
}

bsc_status read_scalars(bsc_ctx *x)
{
    BSC_HIP(hipMemcpyAsync(x->hscal, x->dscal, sizeof(int64_t) * DS_COUNT, hipMemcpyDeviceToHost, x->stream));
    BSC_HIP(hipStreamSynchronize(x->stream));
    return BSC_OK;
}

template <typename T>
__global__ void k_fill(T *p, int64_t n, T v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

template <typename T>
static void fill(bsc_ctx *x, T *p, int64_t n, T v)
{
    if (n <= 0) return;
    int64_t blocks = (n + TPB - 1) / TPB;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((k_fill<T>), dim3((unsigned)blocks), dim3(TPB), 0, x->stream, p, n, v);
}

#define ALLOC(ptr, count)                                                               \
    do {                                                                                \
        size_t _b = sizeof(*(ptr)) * (size_t)(count);                                   \
        hipError_t _e = hipMalloc((void **)&(ptr), _b ? _b : 16);                       \
        if (_e != hipSuccess) {                                                         \
            bsc_set_error("hipMalloc(%zu bytes) for %s: %s", _b, #ptr, hipGetErrorString(_e)); \
            bsc_destroy(x);                                                             \
            return BSC_E_HIP;                                                           \
        }                                                                               \
    } while (0)

static bsc_status reset_state(bsc_ctx *x)
{
    hipStream_t s = x->stream;
    x->chain_pending = false;                  // the state it would update is being cleared
    x->last_chain_set = -1;
    if (x->side) BSC_HIP(hipStreamSynchronize(x->side));
    x->ev_done_valid[0] = x->ev_done_valid[1] = false;
    x->ev_runs_valid = false;
    x->last_order_set = -1;
    const int64_t gs2 = (int64_t)x->c.grid_size * x->c.grid_size;
    const int64_t vcap = x->c.voxel_capacity;
    fill<int32_t>(x, x->occ, x->ncell, -1);                                     // memory_2.py:717
    BSC_HIP(hipMemsetAsync(x->rgb_pos, 0, sizeof(int32_t) * 3 * (vcap + 1), s));
    BSC_HIP(hipMemsetAsync(x->rgb, 0, 3 * vcap, s));
    BSC_HIP(hipMemsetAsync(x->weight, 0, sizeof(float) * vcap, s));
    BSC_HIP(hipMemsetAsync(x->hmap, 0, sizeof(u64) * gs2, s));                  // 0 == -inf (memory_2.py:99)
    BSC_HIP(hipMemsetAsync(x->cv_map, 0, 3 * gs2, s));
    BSC_HIP(hipMemsetAsync(x->dscal, 0, sizeof(int64_t) * DS_COUNT, s));
    if (x->c.mode == BSC_MODE_EXACT) {
        BSC_HIP(hipMemsetAsync(x->cache_f, 0, sizeof(float) * (size_t)x->c.iter_size * x->c.token_dim, s));
        BSC_HIP(hipMemsetAsync(x->cache_pos, 0, sizeof(int32_t) * 3 * (size_t)x->c.iter_size, s));
        BSC_HIP(hipMemsetAsync(x->cache_d, 0, sizeof(float) * (size_t)x->c.iter_size, s));
        BSC_HIP(hipMemsetAsync(x->store_cnt, 0, sizeof(int32_t) * (vcap + 1), s));
    } else {
        BSC_HIP(hipMemsetAsync(x->acnt, 0, sizeof(int32_t) * (vcap + 1), s));
    }
    BSC_HIP(hipGetLastError());
    x->iter_id = 0;
    x->n_flush = 0;
    x->pool_n_host = 0;
    x->order_base = 0;
    x->names_dirty = true;
    // an empty dense map has no row whose scale could be stale: the reduce keeps the cache current from the first ingest on
    x->row_scale_dirty = !(x->c.mode != BSC_MODE_EXACT && x->l_rscale && x->l_rscale_cap >= (int64_t)sizeof(float2) * (vcap + 1));
    x->log_n = 0;
    x->log_stale = false;
    return BSC_OK;
}

// Fast-geometry preconditions (geometry_dev.h): pinhole structure of K, Kinv, Kpatch, and per-pixel patch tables.
// The patch index int(u - 0.5), u = (Kp (Kinv p2d z))[0] / z (utils.py:208-214 with get_sim_cam_mat, memory_2.py:871),
// is evaluated per pixel column / row in extended precision; the fp64 chain of the reference differs from it by a few
// ulps (< 1e-13), so the table is exact when no value lies within 1e-9 of an integer — otherwise (a pixel centre on a
// patch boundary) the generic per-point chain stays in charge.
static bool pinhole(const double *m)
{
    return m[1] == 0.0 && m[3] == 0.0 && m[6] == 0.0 && m[7] == 0.0 && m[8] == 1.0 && m[0] != 0.0 && m[4] != 0.0;
}

static bool patch_table(int n, double kinv_a, double kinv_b, double kp_a, double kp_b, int g, uint8_t *out)
{
    for (int i = 0; i < n; ++i) {
        const long double a = (long double)kinv_a * ((long double)i + 0.5L) + (long double)kinv_b;
        const long double u = (long double)kp_a * a + (long double)kp_b - 0.5L;
        if (fabsl(u - rintl(u)) < 1e-9L) return false;
        const long t = (long)truncl(u);                    // int() truncates toward zero (utils.py:212-213)
        out[i] = (t >= 0 && t < g) ? (uint8_t)t : (uint8_t)255;
    }
    return true;
}

// K (Kinv (i + 0.5)) == i + 0.5 up to rounding for every pixel column / row: the source pixel of a point is then its own pixel or
// the one before it (geom_point_fast_t decides which without the division)
static bool proj_identity(int n, double kinv_a, double kinv_b, double k_a, double k_b)
{
    for (int i = 0; i < n; ++i) {
        const long double p = (long double)i + 0.5L;
        const long double u = (long double)k_a * ((long double)kinv_a * p + (long double)kinv_b) + (long double)k_b;
        if (!(fabsl(u - p) < 1e-9L * p)) return false;
    }
    return true;
}

extern "C" bsc_status bsc_create(const bsc_config *cfg, int32_t device, void *hip_stream, bsc_ctx **out)
{
    if (!cfg || !out) { bsc_set_error("bsc_create: null argument"); return BSC_E_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        bsc_set_error("bsc_create: no HIP device visible (libbscnav has no CPU path)");
        return BSC_E_HIP;
    }
    if (device < 0 || device >= ndev) { bsc_set_error("bsc_create: device %d of %d", device, ndev); return BSC_E_INVALID; }
    const bsc_config &c = *cfg;
    const int64_t nh = (int64_t)c.max_h - c.min_h;
    const int64_t ncell = (int64_t)c.grid_size * c.grid_size * nh;
    if (c.height <= 0 || c.width <= 0 || c.grid_size <= 0 || nh <= 0 || c.patch_grid <= 0 || c.patch_grid > 255 ||
        c.token_dim <= 0 || (c.token_dim & 3) || c.token_dim > 2048 || c.cache_size <= 0 || c.cache_size > 64 ||
        c.iter_size <= 0 || c.iter_size > (1 << 20) || c.voxel_capacity <= 0 || c.max_points <= 0 ||
        c.mode < 0 || c.mode > 2 || ncell >= (1ll << 31) || !(c.cell_size > 0)) {
        bsc_set_error("bsc_create: invalid configuration (need token_dim %% 4 == 0 <= 2048, patch_grid <= 255, "
                      "iter_size <= 2^20, gs*gs*(max_h-min_h) < 2^31)");
        return BSC_E_INVALID;
    }
    BSC_HIP(hipSetDevice(device));
    bsc_ctx *x = (bsc_ctx *)calloc(1, sizeof(bsc_ctx));
    x->c = c;
    x->device = device;
    x->stream = (hipStream_t)hip_stream;
    x->nh = (int)nh;
    x->ncell = ncell;
    x->g2 = c.patch_grid * c.patch_grid;
    x->max_frames = 65535;
    const int64_t vcap = c.voxel_capacity, gs2 = (int64_t)c.grid_size * c.grid_size;
    const int64_t D = c.token_dim;
    const int64_t np = c.max_points > c.iter_size ? c.max_points : c.iter_size;
    ALLOC(x->occ, ncell);
    ALLOC(x->rgb_pos, 3 * (vcap + 1));
    ALLOC(x->rgb, 3 * vcap);
    ALLOC(x->weight, vcap);
    ALLOC(x->hmap, gs2);
    ALLOC(x->cv_map, 3 * gs2);
    ALLOC(x->dscal, DS_COUNT);
    BSC_HIP(hipHostMalloc((void **)&x->hscal, sizeof(int64_t) * (DS_COUNT + 1)));      // + a slot for the pair count read on its own
    if (getenv("BSC_NO_MAILBOX") == nullptr &&
        hipHostMalloc((void **)&x->mail, sizeof(int64_t) * (DS_COUNT + 1), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        memset(x->mail, 0, sizeof(int64_t) * (DS_COUNT + 1));
        if (hipHostGetDevicePointer((void **)&x->mail_dev, x->mail, 0) != hipSuccess) { hipHostFree(x->mail); x->mail = nullptr; }
    } else {
        x->mail = nullptr;
        (void)hipGetLastError();
    }
    ALLOC(x->exp_tab, 128);
    {
        double tab[128];
        for (int j = 0; j < 64; ++j) {
            const long double v = exp2l((long double)j / 64.0L);
            tab[2 * j] = (double)v;
            tab[2 * j + 1] = (double)(v - (long double)tab[2 * j]);
        }
        if (hipMemcpy(x->exp_tab, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) { bsc_set_error("bsc_create: exp table upload failed"); bsc_destroy(x); return BSC_E_HIP; }
    }
    ALLOC(x->pat_x, c.width);
    ALLOC(x->pat_y, c.height);
    {
        uint8_t *tx = (uint8_t *)malloc(c.width), *ty = (uint8_t *)malloc(c.height);
        x->long_chain = getenv("BSC_QUAD_CHAIN_ONLY") == nullptr;
        {
            // 8-byte point records: a valid depth z (min_depth < z < max_depth, utils.py:175-177) as float_as_uint(z) - zbase
            float zlo = (float)c.min_depth, zhi = (float)c.max_depth;
            if ((double)zlo > c.min_depth) zlo = nextafterf(zlo, 0.f);
            if ((double)zhi < c.max_depth) zhi = nextafterf(zhi, INFINITY);
            uint32_t blo = 0, bhi = 0;
            memcpy(&blo, &zlo, 4); memcpy(&bhi, &zhi, 4);
            x->rec8_zbase = blo;
            x->rec8_ok = getenv("BSC_REC12") == nullptr && c.min_depth > 0 && zlo > 0.f && c.max_depth > c.min_depth && isfinite(zhi) &&
                         bhi - blo < (1u << 28) && (int64_t)c.height * c.width < (1ll << 31);
        }
        x->geom_fast = getenv("BSC_GENERIC_GEOMETRY") == nullptr && c.patch_grid < 255 && pinhole(c.K) && pinhole(c.Kinv) &&
                       pinhole(c.Kpatch) && c.width < (1 << 24) / c.height &&
                       patch_table(c.width, c.Kinv[0], c.Kinv[2], c.Kpatch[0], c.Kpatch[2], c.patch_grid, tx) &&
                       patch_table(c.height, c.Kinv[4], c.Kinv[5], c.Kpatch[4], c.Kpatch[5], c.patch_grid, ty);
        hipError_t e = hipSuccess;
        x->proj_id = getenv("BSC_PROJ_DIVIDE") == nullptr && c.min_depth >= 0.0 &&
                     proj_identity(c.width, c.Kinv[0], c.Kinv[2], c.K[0], c.K[2]) &&
                     proj_identity(c.height, c.Kinv[4], c.Kinv[5], c.K[4], c.K[5]);
        x->pat_all_in = x->geom_fast;           // every pixel centre inside the patch grid (the usual case: K_patch is K scaled)
        for (int px = 0; px < c.width && x->pat_all_in; ++px) x->pat_all_in = tx[px] != 255;
        for (int py = 0; py < c.height && x->pat_all_in; ++py) x->pat_all_in = ty[py] != 255;
        if (x->geom_fast) {
            e = hipMemcpy(x->pat_x, tx, c.width, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(x->pat_y, ty, c.height, hipMemcpyHostToDevice);
        }
        // pixel rectangle of every patch (the tables are monotone: a patch is a contiguous range of columns x rows)
        const int g = c.patch_grid, g2 = g * g;
        ALLOC(x->pt_rect, 4 * g2);
        ALLOC(x->pt_off, g2 + 1);
        x->patch_tiles = x->geom_fast && c.mode != BSC_MODE_EXACT;
        if (x->patch_tiles) {
            int32_t *rect = (int32_t *)calloc(4 * g2, sizeof(int32_t)), *off = (int32_t *)calloc(g2 + 1, sizeof(int32_t));
            int32_t *xs = (int32_t *)calloc(2 * g, sizeof(int32_t)), *ys = (int32_t *)calloc(2 * g, sizeof(int32_t));
            for (int k = 0; k < g; ++k) { xs[2 * k] = ys[2 * k] = -1; }
            for (int i = 0; i < c.width; ++i) if (tx[i] != 255) { if (xs[2 * tx[i]] < 0) xs[2 * tx[i]] = i; xs[2 * tx[i] + 1] = i + 1; }
            for (int i = 0; i < c.height; ++i) if (ty[i] != 255) { if (ys[2 * ty[i]] < 0) ys[2 * ty[i]] = i; ys[2 * ty[i] + 1] = i + 1; }
            int64_t total = 0;
            for (int py = 0; py < g; ++py)
                for (int px = 0; px < g; ++px) {
                    const int p = py * g + px;
                    const int w = xs[2 * px] < 0 ? 0 : xs[2 * px + 1] - xs[2 * px], h = ys[2 * py] < 0 ? 0 : ys[2 * py + 1] - ys[2 * py];
                    rect[4 * p] = w ? xs[2 * px] : 0; rect[4 * p + 1] = w; rect[4 * p + 2] = h ? ys[2 * py] : 0; rect[4 * p + 3] = w * h;
                    off[p] = (int32_t)total;
                    total += (int64_t)w * h;
                    if (w * h > 3328) x->patch_tiles = false;   // PP_R * TPB of dense.hip
                }
            off[g2] = (int32_t)total;
            if (total > (int64_t)c.width * c.height) x->patch_tiles = false;
            if ((int64_t)c.grid_size * c.grid_size * x->nh > (1ll << 26)) x->patch_tiles = false;   // PP_CELL_BITS of dense.hip
            if (x->patch_tiles) {
                e = hipMemcpy(x->pt_rect, rect, sizeof(int32_t) * 4 * g2, hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipMemcpy(x->pt_off, off, sizeof(int32_t) * (g2 + 1), hipMemcpyHostToDevice);
            }
            free(rect); free(off); free(xs); free(ys);
        }
        if (getenv("BSC_DEBUG"))
            fprintf(stderr, "bsc_create: %dx%d g=%d fast geometry %d, patch-aligned pair tiles %d\n", c.width, c.height,
                    c.patch_grid, (int)x->geom_fast, (int)x->patch_tiles);
        free(tx); free(ty);
        if (e != hipSuccess) { bsc_set_error("bsc_create: patch tables: %s", hipGetErrorString(e)); bsc_destroy(x); return BSC_E_HIP; }
    }
    if (c.mode == BSC_MODE_EXACT) {
        if (c.token_capacity <= 0) { bsc_set_error("bsc_create: token_capacity"); bsc_destroy(x); return BSC_E_INVALID; }
        ALLOC(x->cache_f, (int64_t)c.iter_size * D);
        ALLOC(x->cache_pos, 3 * (int64_t)c.iter_size);
        ALLOC(x->cache_d, c.iter_size);
        ALLOC(x->pool, c.token_capacity * D);
        ALLOC(x->pool_d, c.token_capacity);
        ALLOC(x->store_rows, (vcap + 1) * c.cache_size);
        ALLOC(x->store_cnt, vcap + 1);
        ALLOC(x->f_rowdst, c.iter_size); ALLOC(x->f_hit, c.iter_size); ALLOC(x->f_hidx, c.iter_size);
        ALLOC(x->f_rowseg, c.iter_size); ALLOC(x->f_rowe, c.iter_size); ALLOC(x->f_headpos, c.iter_size);
        ALLOC(x->f_win, (int64_t)c.iter_size * c.cache_size);
        ALLOC(x->f_draws, c.iter_size);
        ALLOC(x->f_keys_a, c.iter_size); ALLOC(x->f_keys_b, c.iter_size);
    } else {
        ALLOC(x->acc, vcap * D);
        ALLOC(x->acnt, vcap + 1);
        ALLOC(x->l_rscale, vcap + 1);           // row scales of the batched scan, kept current by the dense reduce (8 B per voxel)
        x->l_rscale_cap = (int64_t)sizeof(float2) * (vcap + 1);
    }
    ALLOC(x->p_cell, np); ALLOC(x->p_patf, np); ALLOC(x->p_r2f, np);
    ALLOC(x->skey_a, np); ALLOC(x->sval_a, np);
    ALLOC(x->stage_cell, np + 4096); ALLOC(x->stage_pos, np + 4096);
    {
        const int rpw = getenv("BSC_GROUP_RPW") ? atoi(getenv("BSC_GROUP_RPW")) : 8;
        x->group_rpw = rpw == 16 ? 16 : (rpw == 8 ? 8 : 4);
    }
    ALLOC(x->new_cells, np); ALLOC(x->run_val_b, np); ALLOC(x->run_scan, np); ALLOC(x->seg_k0, np); ALLOC(x->seg_vid, np);
    x->nblk_cap = np / 1024 + 16;
    ALLOC(x->blk_cnt, x->nblk_cap); ALLOC(x->blk_off, x->nblk_cap);
    ALLOC(x->blk_pass, x->nblk_cap); ALLOC(x->blk_pass_off, x->nblk_cap);
    ALLOC(x->hb_cnt, x->nblk_cap); ALLOC(x->hb_off, x->nblk_cap);
    ALLOC(x->pass_list, np);
    for (int k = 0; k < 2; ++k) {
        ALLOC(x->p_rec_s[k], np); ALLOC(x->skey_b_s[k], np); ALLOC(x->sval_b_s[k], np);
        ALLOC(x->run_bits_s[k], np / 64 + 2); ALLOC(x->ck_run_s[k], np / 64 + 2); ALLOC(x->ck_start_s[k], np / 64 + 2); ALLOC(x->run_val_s[k], np);
        ALLOC(x->seg_info_s[k], (np < vcap ? np : vcap) + 1); ALLOC(x->seg_last_s[k], (np < vcap ? np : vcap) + 1); ALLOC(x->bscal_s[k], 8);
        BSC_HIP(hipEventCreateWithFlags(&x->ev_ready[k], hipEventDisableTiming));
        BSC_HIP(hipEventCreateWithFlags(&x->ev_done[k], hipEventDisableTiming));
    }
    {
        int lo = 0, hi = 0;   // rgb chain: latency-bound, give it the highest priority the device offers
        BSC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        BSC_HIP(hipStreamCreateWithPriority(&x->side, hipStreamNonBlocking, hi));
        BSC_HIP(hipStreamCreateWithPriority(&x->side2, hipStreamNonBlocking, hi));
        BSC_HIP(hipEventCreateWithFlags(&x->ev_chain0, hipEventDisableTiming));
        BSC_HIP(hipEventCreateWithFlags(&x->ev_mid, hipEventDisableTiming));
    }
    if (c.mode != BSC_MODE_EXACT) {
        x->pair_cap = np;
        ALLOC(x->pair_key_a, np); ALLOC(x->pair_key_b, np); ALLOC(x->pair_cnt_a, np); ALLOC(x->pair_cnt_b, np);
        ALLOC(x->pseg_start, np);
        const int64_t npix = (int64_t)c.height * c.width;
        const int64_t fmax = (np + npix - 1) / npix;
        const int64_t t2d = fmax * ((c.width + 31) / 32) * ((c.height + 31) / 32), t1d = (np + 1023) / 1024;
        x->max_tiles = (t2d > t1d ? t2d : t1d) + 1;
        const int64_t pt_tiles = fmax * c.patch_grid * c.patch_grid + 1;        // patch-aligned tiles: g^2 per frame
        if (pt_tiles > x->max_tiles) x->max_tiles = pt_tiles;
        const int64_t stage_n = x->max_tiles * 1024 > fmax * npix ? x->max_tiles * 1024 : fmax * npix;
        ALLOC(x->pstage_key, stage_n); ALLOC(x->pstage_cnt, stage_n);
        ALLOC(x->tile_cnt, x->max_tiles); ALLOC(x->tile_off, x->max_tiles);
    }
    ALLOC(x->d_transforms, (int64_t)x->max_frames * 16);
    ALLOC(x->d_offsets, x->max_frames + 1);
    ALLOC(x->l_key_a, vcap + 1); ALLOC(x->l_key_b, vcap + 1);
    ALLOC(x->l_val_a, vcap + 1); ALLOC(x->l_val_b, vcap + 1);
    ALLOC(x->l_name_rank, vcap + 1);
    ALLOC(x->l_q, 1024 * D);
    ALLOC(x->l_qp, 3 * 1024 * D);
    int64_t prim_items = (np > vcap + 1) ? np : vcap + 1;
    if (x->max_tiles > prim_items) prim_items = x->max_tiles;
    x->prim_tmp_bytes = prim_workspace_bytes((size_t)prim_items);
    hipError_t e = hipMalloc(&x->prim_tmp, x->prim_tmp_bytes);
    if (e != hipSuccess) { bsc_set_error("hipMalloc prim workspace: %s", hipGetErrorString(e)); bsc_destroy(x); return BSC_E_HIP; }
    e = hipMalloc(&x->prim_tmp_side, x->prim_tmp_bytes);
    if (e != hipSuccess) { bsc_set_error("hipMalloc prim workspace: %s", hipGetErrorString(e)); bsc_destroy(x); return BSC_E_HIP; }
    x->order_on_side = getenv("BSC_ORDER_MAIN") == nullptr;
    x->radix_intree = getenv("BSC_SORT_ROCPRIM") == nullptr;
    if (radix_ws_create(&x->rx_main, (size_t)prim_items) != BSC_OK || radix_ws_create(&x->rx_side, (size_t)prim_items) != BSC_OK) { bsc_destroy(x); return BSC_E_HIP; }
    BSC_HIP(hipEventCreateWithFlags(&x->ev_ids, hipEventDisableTiming));
    BSC_HIP(hipEventCreateWithFlags(&x->ev_runs, hipEventDisableTiming));
    BSC_HIP(hipEventCreateWithFlags(&x->ev_tot, hipEventDisableTiming));
    BSC_HIP(hipEventCreateWithFlags(&x->ev_psort, hipEventDisableTiming));
    BSC_HIP(hipStreamCreateWithFlags(&x->copy, hipStreamNonBlocking));
    for (int w = 0; w < BSC_STAT_SLOTS; ++w)
        for (int i = 0; i < 2 * BSC_EV_RING; ++i) BSC_HIP(hipEventCreate(&x->ev[w][i]));
    x->timing = true;
    bsc_status st = reset_state(x);
    if (st != BSC_OK) { bsc_destroy(x); return st; }
    BSC_HIP(hipStreamSynchronize(x->stream));
    *out = x;
    return BSC_OK;
}

extern "C" void bsc_destroy(bsc_ctx *x)
{
    if (!x) return;
    hipSetDevice(x->device);
    if (x->side) hipStreamSynchronize(x->side);
    hipStreamSynchronize(x->stream);
    void *ptrs[] = {x->exp_tab, x->pat_x, x->pat_y, x->pt_rect, x->pt_off, x->occ, x->rgb_pos, x->rgb, x->weight, x->hmap, x->cv_map, x->dscal, x->cache_f, x->cache_pos,
                    x->cache_d, x->pool, x->pool_d, x->store_rows, x->store_cnt, x->acc, x->acnt, x->p_cell, x->p_patf,
                    x->p_rec_s[0], x->p_rec_s[1], x->p_r2f, x->new_cells, x->run_scan, x->seg_k0, x->seg_vid, x->blk_pass, x->blk_pass_off, x->hb_cnt, x->hb_off,
                    x->skey_a, x->sval_a, x->skey_b_s[0], x->skey_b_s[1], x->sval_b_s[0], x->sval_b_s[1], x->blk_cnt, x->blk_off,
                    x->pstage_key, x->pstage_cnt, x->tile_cnt, x->tile_off, x->pass_list, x->seg_info_s[0], x->seg_info_s[1], x->run_val_b,
                    x->seg_last_s[0], x->seg_last_s[1], x->bscal_s[0], x->bscal_s[1], x->f_keys_a, x->f_keys_b, x->pair_key_a, x->pair_key_b, x->pair_cnt_a, x->pair_cnt_b, x->pseg_start,
                    x->d_transforms, x->d_offsets, x->f_rowdst, x->f_hit, x->f_hidx, x->f_rowseg, x->f_rowe,
                    x->f_headpos, x->f_win, x->f_draws, x->l_sims, x->l_key_a, x->l_key_b, x->l_val_a, x->l_val_b,
                    x->l_name_rank, x->l_q, x->l_qp, x->l_out_pos, x->l_out_sim, x->l_sel_key[0], x->l_sel_key[1], x->l_sel_val[0],
                    x->l_sel_val[1], x->l_sel_thr, x->l_sel_cnt, x->l_valid, x->l_rscale, x->prim_tmp, x->prim_tmp_side, x->fr_mask, x->fr_in, x->fr_parent, x->fr_size,
                    x->fr_ord, x->fr_roots, x->fr_labels, x->fr_first, x->fr_sizes, x->fr_scal, x->fr_sumx, x->fr_sumy,
                    x->fr_centers, x->fr_gains, x->log_cell, x->log_rec, x->stage_cell, x->stage_pos,
                    x->run_bits_s[0], x->run_bits_s[1], x->ck_run_s[0], x->ck_run_s[1], x->ck_start_s[0], x->ck_start_s[1], x->run_val_s[0], x->run_val_s[1]};
    for (void *p : ptrs)
        if (p) hipFree(p);
    for (int b = 0; b < BSC_H2D_NBUF; ++b) {
        if (x->h2d_pin[b]) { hipHostFree(x->h2d_pin[b]); hipEventDestroy(x->h2d_ev[b]); }
    }
    radix_ws_destroy(&x->rx_main);
    radix_ws_destroy(&x->rx_side);
    if (x->hscal) hipHostFree(x->hscal);
    if (x->mail) hipHostFree(x->mail);
    if (x->side) hipStreamDestroy(x->side);
    if (x->side2) { hipStreamSynchronize(x->side2); hipStreamDestroy(x->side2); }
    if (x->ev_chain0) hipEventDestroy(x->ev_chain0);
    if (x->ev_mid) hipEventDestroy(x->ev_mid);
    if (x->ev_ids) hipEventDestroy(x->ev_ids);
    if (x->ev_runs) hipEventDestroy(x->ev_runs);
    if (x->ev_tot) hipEventDestroy(x->ev_tot);
    if (x->ev_psort) hipEventDestroy(x->ev_psort);
    if (x->copy) hipStreamDestroy(x->copy);
    for (int k = 0; k < 2; ++k) {
        if (x->ev_ready[k]) hipEventDestroy(x->ev_ready[k]);
        if (x->ev_done[k]) hipEventDestroy(x->ev_done[k]);
    }
    for (int w = 0; w < BSC_STAT_SLOTS; ++w)
        for (int i = 0; i < 2 * BSC_EV_RING; ++i)
            if (x->ev[w][i]) hipEventDestroy(x->ev[w][i]);
    free(x);
}

extern "C" bsc_status bsc_reset(bsc_ctx *x)
{
    if (!x) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(reset_state(x));
    BSC_HIP(hipStreamSynchronize(x->stream));
    return BSC_OK;
}

extern "C" bsc_status bsc_sync(bsc_ctx *x)
{
    if (!x) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    return sync_all(x);
}

extern "C" bsc_status bsc_stream_wait_chain(bsc_ctx *x, void *hip_stream)
{
    if (!x) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    // the most recently LAUNCHED chain (a deferred one that has not been launched yet is not waited for)
    const int last = x->last_chain_set;
    if (last >= 0 && x->ev_done_valid[last]) BSC_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, x->ev_done[last], 0));
    return BSC_OK;
}

extern "C" bsc_status bsc_ingest_typed(bsc_ctx *x, int32_t n_frames, const float *depth_dev, const uint8_t *rgb_dev,
                                       int32_t rgb_channels, const void *tokens_dev, int32_t token_dtype,
                                       const double *transforms_host, const int32_t *sample_idx_dev,
                                       const int64_t *offsets_host, const double *alpha_dev, bsc_draw_fn draw, void *user)
{
    if (!x || !depth_dev || !rgb_dev || !tokens_dev || !transforms_host || n_frames < 1 || n_frames > x->max_frames ||
        rgb_channels < 3 || (sample_idx_dev && !offsets_host) || (token_dtype != BSC_TOK_F32 && token_dtype != BSC_TOK_BF16)) {
        bsc_set_error("bsc_ingest: invalid argument");
        return BSC_E_INVALID;
    }
    BSC_HIP(hipSetDevice(x->device));
    BSC_HIP(hipMemcpyAsync(x->d_transforms, transforms_host, sizeof(double) * 16 * n_frames, hipMemcpyHostToDevice,
                           x->stream));
    x->names_dirty = true;
    const bool scales_were_current = !x->row_scale_dirty;
    x->row_scale_dirty = true;
    x->rscale_from_reduce = false;
    bsc_status st = ingest_batch(x, n_frames, depth_dev, rgb_dev, rgb_channels, tokens_dev, token_dtype, sample_idx_dev,
                                 offsets_host, alpha_dev, draw, user);
    // the per-voxel dense reduce leaves the scale / inverse norm of every row it wrote: the cache stays current across ingests
    if (st == BSC_OK && scales_were_current && x->rscale_from_reduce) x->row_scale_dirty = false;
    return st;
}

extern "C" bsc_status bsc_ingest(bsc_ctx *x, int32_t n_frames, const float *depth_dev, const uint8_t *rgb_dev,
                                 int32_t rgb_channels, const float *tokens_dev, const double *transforms_host,
                                 const int32_t *sample_idx_dev, const int64_t *offsets_host, const double *alpha_dev,
                                 bsc_draw_fn draw, void *user)
{
    return bsc_ingest_typed(x, n_frames, depth_dev, rgb_dev, rgb_channels, tokens_dev, BSC_TOK_F32, transforms_host,
                            sample_idx_dev, offsets_host, alpha_dev, draw, user);
}

extern "C" bsc_status bsc_flush(bsc_ctx *x, bsc_draw_fn draw, void *user)
{
    if (!x) return BSC_E_INVALID;
    if (x->c.mode != BSC_MODE_EXACT) { bsc_set_error("bsc_flush: only the exact mode has a token cache"); return BSC_E_STATE; }
    BSC_HIP(hipSetDevice(x->device));
    return flush_cache(x, draw, user);
}

__global__ __launch_bounds__(TPB) void k_store_totals(int n, const int32_t *__restrict__ cnt, int64_t *out2)
{
    __shared__ long long sv[TPB], st[TPB];
    long long v = 0, t = 0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const int c = cnt[i];
        v += c > 0;
        t += c;
    }
    sv[threadIdx.x] = v; st[threadIdx.x] = t;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { sv[threadIdx.x] += sv[threadIdx.x + o]; st[threadIdx.x] += st[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd((u64 *)&out2[0], (u64)sv[0]);
        atomicAdd((u64 *)&out2[1], (u64)st[0]);
    }
}

extern "C" bsc_status bsc_counters(bsc_ctx *x, int64_t *out)
{
    if (!x || !out) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    const int32_t *cnt = x->c.mode == BSC_MODE_EXACT ? x->store_cnt : x->acnt;
    BSC_HIP(hipMemsetAsync(x->dscal + DS_TMP0, 0, sizeof(int64_t) * 2, x->stream));
    hipLaunchKernelGGL(k_store_totals, dim3(512), dim3(TPB), 0, x->stream, x->c.voxel_capacity + 1, cnt, x->dscal + DS_TMP0);
    BSC_TRY(read_scalars(x));
    out[0] = x->hscal[DS_MAX_ID];
    out[1] = x->iter_id;
    out[2] = x->hscal[DS_TMP0];
    out[3] = x->hscal[DS_TMP1];
    out[4] = x->n_flush;
    out[5] = x->hscal[DS_NPASS_TOTAL];
    out[6] = x->hscal[DS_NSEEN_TOTAL];
    out[7] = x->hscal[DS_RMW_TOTAL];
    out[8] = x->hscal[DS_PAIR_TOTAL];
    out[9] = x->hscal[DS_B_NPAIR];
    if (x->hscal[DS_ERROR]) {
        bsc_set_error("device capacity error flag %lld (1 voxel_capacity, 2 token_capacity)", (long long)x->hscal[DS_ERROR]);
        return BSC_E_CAPACITY;
    }
    return BSC_OK;
}

extern "C" bsc_status bsc_geometry(bsc_ctx *x, const float *depth_dev, const double *transform_host,
                                   const int32_t *sample_idx_dev, int64_t P, uint8_t *flags_host, double *pc_host,
                                   double *pg_host, int32_t *vox_host, int32_t *pix_host, int32_t *pat_host,
                                   double *r2_host, double *alpha_host)
{
    if (!x || !depth_dev || !transform_host || P < 1) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    hipStream_t s = x->stream;
    BSC_HIP(hipMemcpyAsync(x->d_transforms, transform_host, sizeof(double) * 16, hipMemcpyHostToDevice, s));
    uint8_t *flags; double *pc, *pg, *r2, *al; int32_t *vox, *pix, *pat;
    BSC_HIP(hipMalloc((void **)&flags, P)); BSC_HIP(hipMalloc((void **)&pc, 24 * P)); BSC_HIP(hipMalloc((void **)&pg, 24 * P));
    BSC_HIP(hipMalloc((void **)&vox, 12 * P)); BSC_HIP(hipMalloc((void **)&pix, 8 * P)); BSC_HIP(hipMalloc((void **)&pat, 8 * P));
    BSC_HIP(hipMalloc((void **)&r2, 8 * P)); BSC_HIP(hipMalloc((void **)&al, 8 * P));
    BSC_HIP(hipMemsetAsync(pg, 0, 24 * P, s)); BSC_HIP(hipMemsetAsync(vox, 0, 12 * P, s));
    BSC_HIP(hipMemsetAsync(pix, 0, 8 * P, s)); BSC_HIP(hipMemsetAsync(pat, 0, 8 * P, s));
    BSC_HIP(hipMemsetAsync(r2, 0, 8 * P, s)); BSC_HIP(hipMemsetAsync(al, 0, 8 * P, s));
    bsc_status st = launch_geometry_debug(x, depth_dev, sample_idx_dev, P, flags, pc, pg, vox, pix, pat, r2, al);
    if (st == BSC_OK) {
        hipStreamSynchronize(s);
        if (flags_host) hipMemcpy(flags_host, flags, P, hipMemcpyDeviceToHost);
        if (pc_host) hipMemcpy(pc_host, pc, 24 * P, hipMemcpyDeviceToHost);
        if (pg_host) hipMemcpy(pg_host, pg, 24 * P, hipMemcpyDeviceToHost);
        if (vox_host) hipMemcpy(vox_host, vox, 12 * P, hipMemcpyDeviceToHost);
        if (pix_host) hipMemcpy(pix_host, pix, 8 * P, hipMemcpyDeviceToHost);
        if (pat_host) hipMemcpy(pat_host, pat, 8 * P, hipMemcpyDeviceToHost);
        if (r2_host) hipMemcpy(r2_host, r2, 8 * P, hipMemcpyDeviceToHost);
        if (alpha_host) hipMemcpy(alpha_host, al, 8 * P, hipMemcpyDeviceToHost);
    }
    hipFree(flags); hipFree(pc); hipFree(pg); hipFree(vox); hipFree(pix); hipFree(pat); hipFree(r2); hipFree(al);
    return st;
}

// parity entry of the in-tree radix sort (radix.hip): device arrays in, device arrays out, on the context's stream
extern "C" bsc_status bsc_sort_pairs_u32(bsc_ctx *x, const uint32_t *keys_dev, const uint32_t *vals_dev, int64_t n, int32_t begin_bit,
                                         int32_t end_bit, uint32_t *keys_out_dev, uint32_t *vals_out_dev)
{
    if (!x || n < 0 || (n > 0 && (!keys_dev || !vals_dev || !keys_out_dev || !vals_out_dev)) || begin_bit < 0 || end_bit > 32 || begin_bit > end_bit)
        return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    return radix_sort_pairs_u32(x, &x->rx_main, x->stream, keys_dev, keys_out_dev, vals_dev, vals_out_dev, (size_t)n, begin_bit, end_bit);
}

// ---- exports ---------------------------------------------------------------------------------------
extern "C" bsc_status bsc_export_rgb(bsc_ctx *x, int32_t *pos, uint8_t *rgb, float *weight)
{
    if (!x) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));
    BSC_TRY(read_scalars(x));
    const int64_t n = x->hscal[DS_MAX_ID];
    if (n == 0) return BSC_OK;
    if (pos) BSC_HIP(hipMemcpy(pos, x->rgb_pos, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost));
    if (rgb) BSC_HIP(hipMemcpy(rgb, x->rgb, 3 * n, hipMemcpyDeviceToHost));
    if (weight) BSC_HIP(hipMemcpy(weight, x->weight, sizeof(float) * n, hipMemcpyDeviceToHost));
    return BSC_OK;
}

extern "C" bsc_status bsc_export_occupied(bsc_ctx *x, int32_t *occ)
{
    if (!x || !occ) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_HIP(hipStreamSynchronize(x->stream));
    BSC_HIP(hipMemcpy(occ, x->occ, sizeof(int32_t) * x->ncell, hipMemcpyDeviceToHost));
    return BSC_OK;
}

extern "C" bsc_status bsc_export_heightmap(bsc_ctx *x, double *max_height, uint8_t *cv_map)
{
    if (!x) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));
    const int64_t gs2 = (int64_t)x->c.grid_size * x->c.grid_size;
    if (max_height) {
        u64 *h = (u64 *)malloc(sizeof(u64) * gs2);
        hipError_t e = hipMemcpy(h, x->hmap, sizeof(u64) * gs2, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (int64_t i = 0; i < gs2; ++i) max_height[i] = h[i] ? (double)((int64_t)(h[i] >> 40) - 1) : -INFINITY;
        free(h);
        BSC_HIP(e);
    }
    if (cv_map) BSC_HIP(hipMemcpy(cv_map, x->cv_map, 3 * gs2, hipMemcpyDeviceToHost));
    return BSC_OK;
}

extern "C" bsc_status bsc_import_cv_map(bsc_ctx *x, const uint8_t *cv_map)
{
    if (!x || !cv_map) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));
    const int64_t gs2 = (int64_t)x->c.grid_size * x->c.grid_size;
    BSC_HIP(hipMemcpy(x->cv_map, cv_map, 3 * gs2, hipMemcpyHostToDevice));
    BSC_HIP(hipMemset(x->hmap, 0, sizeof(u64) * gs2));
    return BSC_OK;
}

extern "C" bsc_status bsc_frontier_mask(bsc_ctx *x, const uint8_t *navigable_host, uint8_t *mask_host)
{
    if (!x) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));                   // the top-down map is written on the side stream
    return frontier_mask_impl(x, navigable_host, mask_host);
}

extern "C" bsc_status bsc_frontier_clusters(bsc_ctx *x, const uint8_t *frontier_host, int32_t min_cluster_size,
                                            int32_t ig_radius, int32_t max_clusters, int32_t *n_clusters_host,
                                            int32_t *labels_host, int32_t *first_host, int32_t *sizes_host,
                                            double *centers_host, double *gains_host, int32_t *best_host)
{
    if (!x || !n_clusters_host || max_clusters < 0 || ig_radius < 0) {
        bsc_set_error("bsc_frontier_clusters: invalid argument");
        return BSC_E_INVALID;
    }
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));
    return frontier_clusters_impl(x, frontier_host, min_cluster_size, ig_radius, max_clusters, n_clusters_host, labels_host,
                                  first_host, sizes_host, centers_host, gains_host, best_host);
}

extern "C" bsc_status bsc_export_cache(bsc_ctx *x, float *feat, int32_t *pos, float *dis)
{
    if (!x || x->c.mode != BSC_MODE_EXACT) return BSC_E_STATE;
    BSC_HIP(hipSetDevice(x->device));
    BSC_HIP(hipStreamSynchronize(x->stream));
    const int64_t n = x->iter_id;
    if (n == 0) return BSC_OK;
    if (feat) BSC_HIP(hipMemcpy(feat, x->cache_f, sizeof(float) * n * x->c.token_dim, hipMemcpyDeviceToHost));
    if (pos) BSC_HIP(hipMemcpy(pos, x->cache_pos, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost));
    if (dis) BSC_HIP(hipMemcpy(dis, x->cache_d, sizeof(float) * n, hipMemcpyDeviceToHost));
    return BSC_OK;
}

// name key on the host (same encoding as localize.hip)
static u64 host_name_field(int32_t v, bool last)
{
    int dig[12], n = 0;
    if (v == 0) dig[n++] = 0;
    while (v > 0) { dig[n++] = v % 10; v /= 10; }
    u64 k = 0;
    for (int i = 0; i < 6; ++i) {
        int sym = (i < n) ? dig[n - 1 - i] + (last ? 1 : 0) : (last ? 0 : 10);
        k = k * 11 + (u64)sym;
    }
    return k;
}
static u64 host_name_key(int32_t r, int32_t c, int32_t h)
{
    const u64 B = 1771561ull;
    return (host_name_field(r, false) * B + host_name_field(c, false)) * B + host_name_field(h, true);
}
struct name_ent { u64 key; int32_t e; };
static int cmp_name_ent(const void *a, const void *b)
{
    const u64 x = ((const name_ent *)a)->key, y = ((const name_ent *)b)->key;
    return x < y ? -1 : (x > y ? 1 : 0);
}

// gather pool rows into name-ordered contiguous output: one wavefront per output token
__global__ __launch_bounds__(TPB) void k_gather_rows(int64_t n, const int32_t *__restrict__ src_rows,
                                                     const float *__restrict__ pool, const float *__restrict__ pool_d,
                                                     int D, float *__restrict__ out, float *__restrict__ out_d)
{
    const int lane = threadIdx.x & 63;
    const int64_t t = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    if (t >= n) return;
    const int64_t r = src_rows[t];
    const float4 *src = (const float4 *)(pool + r * D);
    float4 *dst = (float4 *)(out + t * D);
    for (int v = lane; v < (D >> 2); v += 64) dst[v] = src[v];
    if (lane == 0) out_d[t] = pool_d[r];
}

extern "C" bsc_status bsc_export_store(bsc_ctx *x, int32_t *pos_host, int32_t *cnt_host, float *feats_host,
                                       float *dists_host)
{
    if (!x || x->c.mode != BSC_MODE_EXACT) { bsc_set_error("bsc_export_store: exact mode only"); return BSC_E_STATE; }
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(read_scalars(x));
    const int64_t max_id = x->hscal[DS_MAX_ID], vcap = x->c.voxel_capacity, cs = x->c.cache_size, D = x->c.token_dim;
    // host copies of the small tables, name order on the host, feature gather on the device
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (vcap + 1));
    int32_t *pos = (int32_t *)malloc(sizeof(int32_t) * 3 * (max_id + 1));
    int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * (vcap + 1) * cs);
    hipError_t e = hipMemcpy(cnt, x->store_cnt, sizeof(int32_t) * (vcap + 1), hipMemcpyDeviceToHost);
    if (e == hipSuccess && max_id) e = hipMemcpy(pos, x->rgb_pos, sizeof(int32_t) * 3 * max_id, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(rows, x->store_rows, sizeof(int32_t) * (vcap + 1) * cs, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(cnt); free(pos); free(rows); BSC_HIP(e); }
    name_ent *ents = (name_ent *)malloc(sizeof(name_ent) * (max_id + 1));
    int64_t nv = 0, nt = 0;
    for (int64_t c = 0; c <= max_id; ++c) {
        const int64_t en = (c == max_id) ? vcap : c;
        if (cnt[en] <= 0) continue;
        ents[nv].e = (int32_t)en;
        ents[nv].key = (c == max_id) ? host_name_key(0, 0, 0) : host_name_key(pos[3 * c], pos[3 * c + 1], pos[3 * c + 2]);
        ++nv;
        nt += cnt[en];
    }
    qsort(ents, nv, sizeof(name_ent), cmp_name_ent);
    int32_t *src = (int32_t *)malloc(sizeof(int32_t) * (nt ? nt : 1));
    int64_t t = 0;
    for (int64_t i = 0; i < nv; ++i) {
        const int32_t en = ents[i].e;
        if (pos_host) {
            if (en == vcap) { pos_host[3 * i] = pos_host[3 * i + 1] = pos_host[3 * i + 2] = 0; }
            else memcpy(pos_host + 3 * i, pos + 3 * (int64_t)en, sizeof(int32_t) * 3);
        }
        if (cnt_host) cnt_host[i] = cnt[en];
        for (int k = 0; k < cnt[en]; ++k) src[t++] = rows[(int64_t)en * cs + k];
    }
    bsc_status st = BSC_OK;
    if (nt > 0 && (feats_host || dists_host)) {
        int32_t *d_src = nullptr; float *d_out = nullptr, *d_outd = nullptr;
        e = hipMalloc((void **)&d_src, sizeof(int32_t) * nt);
        if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof(float) * nt * D);
        if (e == hipSuccess) e = hipMalloc((void **)&d_outd, sizeof(float) * nt);
        if (e == hipSuccess) e = hipMemcpyAsync(d_src, src, sizeof(int32_t) * nt, hipMemcpyHostToDevice, x->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((nt * 64 + TPB - 1) / TPB)), dim3(TPB), 0, x->stream, nt, d_src,
                               x->pool, x->pool_d, (int)D, d_out, d_outd);
            e = hipStreamSynchronize(x->stream);
        }
        if (e == hipSuccess && feats_host) e = hipMemcpy(feats_host, d_out, sizeof(float) * nt * D, hipMemcpyDeviceToHost);
        if (e == hipSuccess && dists_host) e = hipMemcpy(dists_host, d_outd, sizeof(float) * nt, hipMemcpyDeviceToHost);
        if (d_src) hipFree(d_src);
        if (d_out) hipFree(d_out);
        if (d_outd) hipFree(d_outd);
        if (e != hipSuccess) { bsc_set_error("bsc_export_store: %s", hipGetErrorString(e)); st = BSC_E_HIP; }
    }
    free(cnt); free(pos); free(rows); free(ents); free(src);
    return st;
}

extern "C" bsc_status bsc_export_dense(bsc_ctx *x, float *acc_host, int32_t *cnt_host)
{
    if (!x || x->c.mode == BSC_MODE_EXACT) { bsc_set_error("bsc_export_dense: dense modes only"); return BSC_E_STATE; }
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(read_scalars(x));
    const int64_t n = x->hscal[DS_MAX_ID];
    if (n == 0) return BSC_OK;
    if (acc_host) BSC_HIP(hipMemcpy(acc_host, x->acc, sizeof(float) * n * x->c.token_dim, hipMemcpyDeviceToHost));
    if (cnt_host) BSC_HIP(hipMemcpy(cnt_host, x->acnt, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return BSC_OK;
}

// ---- imports (load_memory, memory_2.py:189-200) ---------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_import_occ(int64_t n, const int32_t *__restrict__ pos, int gs, int nh, int32_t *occ,
                                                    int64_t *dscal)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int32_t r = pos[3 * i], c = pos[3 * i + 1], h = pos[3 * i + 2];
    if (r < 0 || c < 0 || h < 0 || r >= gs || c >= gs || h >= nh) { dscal[DS_ERROR] = 3; return; }
    occ[((int64_t)r * gs + c) * nh + h] = (int32_t)i;
}

extern "C" bsc_status bsc_import_rgb(bsc_ctx *x, int64_t max_id, const int32_t *pos, const uint8_t *rgb, const float *weight)
{
    if (!x || max_id < 0 || max_id > x->c.voxel_capacity) { bsc_set_error("bsc_import_rgb: max_id vs capacity"); return BSC_E_CAPACITY; }
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(reset_state(x));
    if (x->log_cap && max_id > 0) x->log_stale = true;     // a replayed point log would rebuild colours that ignore the imported state
    if (max_id == 0) return BSC_OK;
    hipStream_t s = x->stream;
    BSC_HIP(hipMemcpyAsync(x->rgb_pos, pos, sizeof(int32_t) * 3 * max_id, hipMemcpyHostToDevice, s));
    BSC_HIP(hipMemcpyAsync(x->rgb, rgb, 3 * max_id, hipMemcpyHostToDevice, s));
    BSC_HIP(hipMemcpyAsync(x->weight, weight, sizeof(float) * max_id, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_import_occ, dim3((unsigned)((max_id + TPB - 1) / TPB)), dim3(TPB), 0, s, max_id, x->rgb_pos,
                       x->c.grid_size, x->nh, x->occ, x->dscal);
    int64_t m[2] = {max_id, max_id};
    BSC_HIP(hipMemcpyAsync(x->dscal + DS_MAX_ID, m, sizeof(int64_t) * 2, hipMemcpyHostToDevice, s));
    BSC_HIP(hipStreamSynchronize(s));
    BSC_TRY(read_scalars(x));
    if (x->hscal[DS_ERROR]) { bsc_set_error("bsc_import_rgb: position outside the grid"); return BSC_E_INVALID; }
    return BSC_OK;
}

// ---- host -> device at link speed from PAGEABLE memory ---------------------------------------------------------------------------
// load_memory hands over NumPy arrays (memory_2.py:189-200: np.load of a memory directory; a 512^3 scene's token store is 20+ GB).
// hipMemcpy from pageable memory stages through one driver thread: 4.6 GB/s on this box against the 57 GB/s a pinned copy reaches.
// Here a few host threads copy slices of a chunk into one of four pinned buffers while the DMA engine drains the previous ones
// (hipMemcpyAsync + an event per buffer): the copy runs at what the host's memory and the link deliver together.
static bsc_status h2d_pipelined(bsc_ctx *x, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return BSC_OK;
    if (bytes < 4 * BSC_H2D_CHUNK) {            // small: not worth the threads
        BSC_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
        return BSC_OK;
    }
    if (!x->h2d_pin[0]) {
        for (int b = 0; b < BSC_H2D_NBUF; ++b) {
            BSC_HIP(hipHostMalloc(&x->h2d_pin[b], BSC_H2D_CHUNK));
            BSC_HIP(hipEventCreateWithFlags(&x->h2d_ev[b], hipEventDisableTiming));
        }
    }
    int nthr = getenv("BSC_H2D_THREADS") ? atoi(getenv("BSC_H2D_THREADS")) : 8;
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && nthr > hw) nthr = hw;
    nthr = nthr < 1 ? 1 : (nthr > 32 ? 32 : nthr);
    const size_t n_chunks = (bytes + BSC_H2D_CHUNK - 1) / BSC_H2D_CHUNK;
    std::atomic<int> arrive{0};
    std::atomic<int> failed{0};
    // sense-reversing barrier over the nthr workers (spin + yield: the phases between barriers are a few milliseconds)
    auto barrier = [&](int &phase) {
        const int target = (++phase) * nthr;
        arrive.fetch_add(1, std::memory_order_acq_rel);
        while (arrive.load(std::memory_order_acquire) < target) std::this_thread::yield();
    };
    auto worker = [&](int t) {
        (void)hipSetDevice(x->device);
        int phase = 0;
        for (size_t c = 0; c < n_chunks; ++c) {
            const int b = (int)(c % BSC_H2D_NBUF);
            const size_t off = c * BSC_H2D_CHUNK, n = bytes - off < BSC_H2D_CHUNK ? bytes - off : BSC_H2D_CHUNK;
            if (t == 0 && c >= BSC_H2D_NBUF && hipEventSynchronize(x->h2d_ev[b]) != hipSuccess) failed.store(1);
            barrier(phase);                                 // buffer b is free again
            const size_t per = ((n + nthr - 1) / nthr + 63) & ~(size_t)63, lo = per * t < n ? per * t : n, hi = lo + per < n ? lo + per : n;
            if (hi > lo) memcpy((char *)x->h2d_pin[b] + lo, (const char *)src_host + off + lo, hi - lo);
            barrier(phase);                                 // chunk c is staged
            if (t == 0) {
                if (hipMemcpyAsync((char *)dst_dev + off, x->h2d_pin[b], n, hipMemcpyHostToDevice, x->copy) != hipSuccess ||
                    hipEventRecord(x->h2d_ev[b], x->copy) != hipSuccess)
                    failed.store(1);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthr; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto &th : pool) th.join();
    BSC_HIP(hipStreamSynchronize(x->copy));
    if (failed.load()) { bsc_set_error("h2d_pipelined: a staged copy failed"); return BSC_E_HIP; }
    return BSC_OK;
}

extern "C" bsc_status bsc_import_store(bsc_ctx *x, int64_t nv, int64_t nt, const int32_t *pos, const int32_t *cnt,
                                       const float *feats, const float *dists)
{
    if (!x || x->c.mode != BSC_MODE_EXACT) { bsc_set_error("bsc_import_store: exact mode only"); return BSC_E_STATE; }
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(grow_token_pool(x, nt));
    const int64_t vcap = x->c.voxel_capacity, cs = x->c.cache_size, D = x->c.token_dim;
    // voxel ids come from the occupied map imported before (bsc_import_rgb)
    int32_t *occ = (int32_t *)malloc(sizeof(int32_t) * x->ncell);
    hipError_t e = hipMemcpy(occ, x->occ, sizeof(int32_t) * x->ncell, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(occ); BSC_HIP(e); }
    int32_t *h_cnt = (int32_t *)calloc(vcap + 1, sizeof(int32_t));
    int32_t *h_rows = (int32_t *)calloc((vcap + 1) * cs, sizeof(int32_t));
    int64_t t = 0;
    bsc_status st = BSC_OK;
    for (int64_t i = 0; i < nv && st == BSC_OK; ++i) {
        const int32_t r = pos[3 * i], c = pos[3 * i + 1], h = pos[3 * i + 2];
        int64_t en = -1;
        if (r == 0 && c == 0 && h == 0) en = vcap;
        else if (r >= 0 && c >= 0 && h >= 0 && r < x->c.grid_size && c < x->c.grid_size && h < x->nh)
            en = occ[((int64_t)r * x->c.grid_size + c) * x->nh + h];
        if (en < 0 || cnt[i] > cs) {
            bsc_set_error("bsc_import_store: group grid_%d_%d_%d has no voxel id (or more than cache_size tokens)", r, c, h);
            st = BSC_E_INVALID;
            break;
        }
        h_cnt[en] = cnt[i];
        for (int k = 0; k < cnt[i]; ++k) h_rows[en * cs + k] = (int32_t)(t++);
    }
    if (st == BSC_OK && t != nt) { bsc_set_error("bsc_import_store: token count mismatch"); st = BSC_E_INVALID; }
    if (st == BSC_OK) {
        e = hipMemcpy(x->store_cnt, h_cnt, sizeof(int32_t) * (vcap + 1), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(x->store_rows, h_rows, sizeof(int32_t) * (vcap + 1) * cs, hipMemcpyHostToDevice);
        if (e == hipSuccess && nt) {
            // the token rows: the large array (sum M x D x 4 bytes), staged through pinned buffers at link speed
            e = hipStreamSynchronize(x->stream);
            if (e == hipSuccess) st = h2d_pipelined(x, x->pool, feats, sizeof(float) * (size_t)nt * D);
            if (e == hipSuccess && st == BSC_OK) st = h2d_pipelined(x, x->pool_d, dists, sizeof(float) * (size_t)nt);
        }
        if (e == hipSuccess) e = hipMemcpy(x->dscal + DS_POOL_N, &nt, sizeof(int64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { bsc_set_error("bsc_import_store: %s", hipGetErrorString(e)); st = BSC_E_HIP; }
    }
    free(occ); free(h_cnt); free(h_rows);
    x->names_dirty = true; x->row_scale_dirty = true;
    if (st == BSC_OK) { x->pool_n_host = nt; localize_prepare(x); }
    return st;
}

extern "C" bsc_status bsc_import_dense(bsc_ctx *x, int64_t max_id, const float *acc, const int32_t *cnt)
{
    if (!x || x->c.mode == BSC_MODE_EXACT) { bsc_set_error("bsc_import_dense: dense modes only"); return BSC_E_STATE; }
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(read_scalars(x));
    if (max_id != x->hscal[DS_MAX_ID]) { bsc_set_error("bsc_import_dense: call bsc_import_rgb first (max_id mismatch)"); return BSC_E_INVALID; }
    if (max_id == 0) return BSC_OK;
    BSC_HIP(hipStreamSynchronize(x->stream));
    BSC_TRY(h2d_pipelined(x, x->acc, acc, sizeof(float) * (size_t)max_id * x->c.token_dim));
    BSC_HIP(hipMemcpy(x->acnt, cnt, sizeof(int32_t) * max_id, hipMemcpyHostToDevice));
    x->names_dirty = true; x->row_scale_dirty = true;
    localize_prepare(x);
    return BSC_OK;
}


// ---- point log + colour replay (exact rgb / weights across ranks) ------------------------------------------------------
extern "C" bsc_status bsc_point_log_enable(bsc_ctx *x, int64_t capacity)
{
    if (!x || capacity < 0) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));
    if (x->log_cell) { (void)hipFree(x->log_cell); x->log_cell = nullptr; }
    if (x->log_rec) { (void)hipFree(x->log_rec); x->log_rec = nullptr; }
    x->log_cap = 0;
    x->log_n = 0;
    x->log_stale = false;
    if (capacity == 0) return BSC_OK;
    BSC_HIP(hipMalloc((void **)&x->log_cell, sizeof(int32_t) * (size_t)capacity));
    BSC_HIP(hipMalloc((void **)&x->log_rec, sizeof(PointRec) * (size_t)capacity));
    x->log_cap = capacity;
    return BSC_OK;
}

extern "C" bsc_status bsc_point_log_read(bsc_ctx *x, int32_t *cells_out_dev, uint32_t *records_out_dev, int64_t capacity,
                                         int64_t *n_points)
{
    if (!x || !n_points || capacity < 0) return BSC_E_INVALID;
    if (!x->log_cap) { bsc_set_error("bsc_point_log_read: the log is not enabled"); return BSC_E_STATE; }
    if (x->log_stale) {
        bsc_set_error("bsc_point_log_read: colour state was imported / replaced after the log was enabled (a map that was loaded "
                      "or merged and then extended): the log does not describe it; bsc_reset or bsc_point_log_enable starts a new one");
        return BSC_E_STATE;
    }
    BSC_HIP(hipSetDevice(x->device));
    *n_points = x->log_n;
    const int64_t n = x->log_n < capacity ? x->log_n : capacity;
    if (n > 0 && cells_out_dev)
        BSC_HIP(hipMemcpyAsync(cells_out_dev, x->log_cell, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, x->stream));
    if (n > 0 && records_out_dev)
        BSC_HIP(hipMemcpyAsync(records_out_dev, x->log_rec, sizeof(PointRec) * (size_t)n, hipMemcpyDeviceToDevice, x->stream));
    BSC_HIP(hipStreamSynchronize(x->stream));
    return BSC_OK;
}

// one thread per voxel: its records [lower_bound(v), lower_bound(v + 1)) replayed from the empty state (memory_2.py:888-899)
__global__ __launch_bounds__(256) void k_replay_colour(int64_t n, const int32_t *__restrict__ vox, const PointRec *__restrict__ rec,
                                                       int64_t n_vox, uint8_t *__restrict__ rgb, float *__restrict__ weight)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n_vox) return;
    int64_t lo = 0, hi = n;                       // first record with vox >= v
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (vox[mid] < (int32_t)v) lo = mid + 1; else hi = mid; }
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    float w = 0.f;
    bool first = true;
    for (int64_t i = lo; i < n && vox[i] == (int32_t)v; ++i) {
        const PointRec r = rec[i];
        const double a = __hiloint2double((int)r.ahi, (int)r.alo);
        const uint32_t r0 = r.rgbv & 0xffu, r1 = (r.rgbv >> 8) & 0xffu, r2 = (r.rgbv >> 16) & 0xffu;
        if (first) {                              // :890-894 a new id takes the colour of its first point, weight f32(0 + alpha)
            c0 = r0; c1 = r1; c2 = r2;
            w = (float)((double)0.f + a);
            first = false;
            continue;
        }
        const double den = (double)w + a;         // :896 (f32 + f64); the product u8 * f32 is rounded in f32, the rest f64
        c0 = (uint32_t)(((double)((float)c0 * w) + (double)r0 * a) / den);
        c1 = (uint32_t)(((double)((float)c1 * w) + (double)r1 * a) / den);
        c2 = (uint32_t)(((double)((float)c2 * w) + (double)r2 * a) / den);
        w = (float)den;                           // :899
    }
    rgb[3 * v] = (uint8_t)c0; rgb[3 * v + 1] = (uint8_t)c1; rgb[3 * v + 2] = (uint8_t)c2;
    weight[v] = w;
}

extern "C" bsc_status bsc_replay_colour(int64_t n_records, const int32_t *vox_sorted_dev, const uint32_t *records_dev, int64_t n_vox,
                                        uint8_t *rgb_dev, float *weight_dev, void *hip_stream)
{
    if (n_records < 0 || n_vox < 0 || (n_records && (!vox_sorted_dev || !records_dev)) || (n_vox && (!rgb_dev || !weight_dev)))
        return BSC_E_INVALID;
    if (n_vox == 0) return BSC_OK;
    hipLaunchKernelGGL(k_replay_colour, dim3((unsigned)((n_vox + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, n_records,
                       vox_sorted_dev, (const PointRec *)records_dev, n_vox, rgb_dev, weight_dev);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// ---- query ----------------------------------------------------------------------------------------------
extern "C" bsc_status bsc_pool_query(bsc_ctx *x, const float *tokens_dev, int32_t B, int32_t T, int32_t D, float *out_dev)
{
    if (!x || !tokens_dev || !out_dev || B < 1 || T < 1 || D < 1) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    return pool_query_impl(x, tokens_dev, B, T, D, out_dev);
}

extern "C" bsc_status bsc_localize(bsc_ctx *x, const float *q_dev, int32_t nq, int32_t K, double radius,
                                   const int32_t *curr_host, int32_t floor_lo, int32_t floor_hi, int32_t *out_pos_host,
                                   float *out_sim_host, int32_t *out_count_host)
{
    if (!x || !q_dev || !out_pos_host || !out_sim_host || !out_count_host) return BSC_E_INVALID;
    if (radius >= 0 && !curr_host) { bsc_set_error("bsc_localize: region filter needs curr"); return BSC_E_INVALID; }
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(read_scalars(x));
    const int64_t n_rows = x->c.mode == BSC_MODE_EXACT ? x->hscal[DS_POOL_N] : x->hscal[DS_MAX_ID];
    const int64_t need = (int64_t)nq * sims_row_stride(n_rows > 0 ? n_rows : 1);
    // scratch grows on demand (similarities for every query x row, top-K staging)
    struct grow { static bsc_status run(void **p, int64_t *cap, int64_t need_bytes) {
        if (*cap >= need_bytes) return BSC_OK;
        if (*p) hipFree(*p);
        *p = nullptr; *cap = 0;
        hipError_t e = hipMalloc(p, (size_t)need_bytes);
        if (e != hipSuccess) { bsc_set_error("bsc_localize scratch: %s", hipGetErrorString(e)); return BSC_E_HIP; }
        *cap = need_bytes;
        return BSC_OK;
    } };
    BSC_TRY(grow::run((void **)&x->l_sims, &x->l_sims_cap, need * 4));
    BSC_TRY(grow::run((void **)&x->l_out_pos, &x->l_out_pos_cap, (int64_t)nq * K * 12));
    BSC_TRY(grow::run((void **)&x->l_out_sim, &x->l_out_sim_cap, (int64_t)nq * K * 4));
    return localize_impl(x, q_dev, nq, K, radius, curr_host, floor_lo, floor_hi, out_pos_host, out_sim_host, out_count_host);
}

// ---- multi-GPU merge helpers (dense modes) ---------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_dense_gather(int64_t n, const int32_t *__restrict__ keys, int gs, int nh,
                                                      const int32_t *__restrict__ occ, const float *__restrict__ acc,
                                                      const int32_t *__restrict__ acnt, int D, int mode,
                                                      float *__restrict__ out, int32_t *__restrict__ out_cnt)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    if (i >= n) return;
    const int32_t r = keys[3 * i], c = keys[3 * i + 1], h = keys[3 * i + 2];
    int32_t vid = -1;
    if (r >= 0 && c >= 0 && h >= 0 && r < gs && c < gs && h < nh) vid = occ[((int64_t)r * gs + c) * nh + h];
    const int32_t m = vid >= 0 ? acnt[vid] : 0;
    float4 *dst = (float4 *)(out + i * D);
    const float fillv = (mode == BSC_MODE_MAX) ? -INFINITY : 0.f;
    for (int v = lane; v < (D >> 2); v += 64)
        dst[v] = (m > 0) ? ((const float4 *)(acc + (int64_t)vid * D))[v] : make_float4(fillv, fillv, fillv, fillv);
    if (lane == 0) out_cnt[i] = m;
}

extern "C" bsc_status bsc_dense_gather(bsc_ctx *x, int64_t n, const int32_t *keys_dev, float *acc_dev, int32_t *cnt_dev)
{
    if (!x || x->c.mode == BSC_MODE_EXACT) { bsc_set_error("bsc_dense_gather: dense modes only"); return BSC_E_STATE; }
    if (n <= 0) return BSC_OK;
    BSC_HIP(hipSetDevice(x->device));
    hipLaunchKernelGGL(k_dense_gather, dim3((unsigned)((n * 64 + TPB - 1) / TPB)), dim3(TPB), 0, x->stream, n, keys_dev,
                       x->c.grid_size, x->nh, x->occ, x->acc, x->acnt, x->c.token_dim, x->c.mode, acc_dev, cnt_dev);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// rgb / weight of the voxels `keys` (absent voxels: weight 0) — the rank-local colour state that travels with a merge
__global__ __launch_bounds__(TPB) void k_rgb_gather(int64_t n, const int32_t *__restrict__ keys, int gs, int nh,
                                                    const int32_t *__restrict__ occ, const uint8_t *__restrict__ rgb,
                                                    const float *__restrict__ weight, uint8_t *__restrict__ out_rgb,
                                                    float *__restrict__ out_w)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int32_t r = keys[3 * i], c = keys[3 * i + 1], h = keys[3 * i + 2];
    int32_t vid = -1;
    if (r >= 0 && c >= 0 && h >= 0 && r < gs && c < gs && h < nh) vid = occ[((int64_t)r * gs + c) * nh + h];
    out_w[i] = vid >= 0 ? weight[vid] : 0.f;
    for (int k = 0; k < 3; ++k) out_rgb[3 * i + k] = vid >= 0 ? rgb[3 * (int64_t)vid + k] : (uint8_t)0;
}

extern "C" bsc_status bsc_dense_gather_rgb(bsc_ctx *x, int64_t n, const int32_t *keys_dev, uint8_t *rgb_dev, float *weight_dev)
{
    if (!x || !keys_dev || !rgb_dev || !weight_dev) { bsc_set_error("bsc_dense_gather_rgb: null argument"); return BSC_E_INVALID; }
    if (n <= 0) return BSC_OK;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));                              // the rgb chain of the last ingest writes rgb / weight on the side stream
    hipLaunchKernelGGL(k_rgb_gather, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, x->stream, n, keys_dev, x->c.grid_size,
                       x->nh, x->occ, x->rgb, x->weight, rgb_dev, weight_dev);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// The map becomes exactly the n voxels handed in, ids 0..n-1 in the order given: keys, feature rows, counts AND the
// colour state (rgb, weight) are replaced together, so every per-id array stays aligned (memory_2.py:888-899 state).
// rgb_dev / weight_dev may be null: colours and weights are then zeroed (a map that was never given colours).
extern "C" bsc_status bsc_dense_replace_full(bsc_ctx *x, int64_t n, const int32_t *keys_dev, const float *acc_dev,
                                             const int32_t *cnt_dev, const uint8_t *rgb_dev, const float *weight_dev)
{
    if (!x || x->c.mode == BSC_MODE_EXACT) { bsc_set_error("bsc_dense_replace: dense modes only"); return BSC_E_STATE; }
    if (x->log_cap && n > 0) x->log_stale = true;          // a replayed point log would rebuild colours that ignore this state
    if (n < 0 || (n > 0 && (!keys_dev || !acc_dev || !cnt_dev)) || ((rgb_dev == nullptr) != (weight_dev == nullptr))) {
        bsc_set_error("bsc_dense_replace: invalid argument");
        return BSC_E_INVALID;
    }
    if (n > x->c.voxel_capacity) { bsc_set_error("bsc_dense_replace: %lld voxels > capacity", (long long)n); return BSC_E_CAPACITY; }
    BSC_HIP(hipSetDevice(x->device));
    hipStream_t s = x->stream;
    BSC_TRY(sync_all(x));
    const int64_t vcap = x->c.voxel_capacity;
    fill<int32_t>(x, x->occ, x->ncell, -1);
    BSC_HIP(hipMemsetAsync(x->acnt, 0, sizeof(int32_t) * (vcap + 1), s));
    BSC_HIP(hipMemsetAsync(x->rgb, 0, 3 * vcap, s));
    BSC_HIP(hipMemsetAsync(x->weight, 0, sizeof(float) * vcap, s));
    BSC_HIP(hipMemsetAsync(x->dscal + DS_ERROR, 0, sizeof(int64_t), s));
    if (n > 0) {
        BSC_HIP(hipMemcpyAsync(x->rgb_pos, keys_dev, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToDevice, s));
        BSC_HIP(hipMemcpyAsync(x->acc, acc_dev, sizeof(float) * n * x->c.token_dim, hipMemcpyDeviceToDevice, s));
        BSC_HIP(hipMemcpyAsync(x->acnt, cnt_dev, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, s));
        if (rgb_dev) {
            BSC_HIP(hipMemcpyAsync(x->rgb, rgb_dev, 3 * n, hipMemcpyDeviceToDevice, s));
            BSC_HIP(hipMemcpyAsync(x->weight, weight_dev, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(k_import_occ, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, n, x->rgb_pos, x->c.grid_size,
                           x->nh, x->occ, x->dscal);
    }
    int64_t m[2] = {n, n};
    BSC_HIP(hipMemcpyAsync(x->dscal + DS_MAX_ID, m, sizeof(int64_t) * 2, hipMemcpyHostToDevice, s));
    BSC_TRY(read_scalars(x));
    x->names_dirty = true; x->row_scale_dirty = true;
    if (x->hscal[DS_ERROR]) {
        BSC_HIP(hipMemsetAsync(x->dscal + DS_ERROR, 0, sizeof(int64_t), s));
        bsc_set_error("bsc_dense_replace: a key lies outside the grid");
        return BSC_E_INVALID;
    }
    localize_prepare(x);        // the merged / replaced memory is about to be queried: name ranks and row scales now
    return BSC_OK;
}

extern "C" bsc_status bsc_dense_replace(bsc_ctx *x, int64_t n, const int32_t *keys_dev, const float *acc_dev,
                                        const int32_t *cnt_dev)
{
    return bsc_dense_replace_full(x, n, keys_dev, acc_dev, cnt_dev, nullptr, nullptr);
}

// top-down map from host arrays (merge of per-rank maps; memory_2.py:901-903 state): max_height (gs,gs) f64 with -inf
// for empty cells, cv_map (gs,gs,3).  Imported cells carry tie order 0, so any later point at the same height wins
// them, as `h >= max_height` does in the reference.
extern "C" bsc_status bsc_import_heightmap(bsc_ctx *x, const double *max_height, const uint8_t *cv_map)
{
    if (!x || !max_height || !cv_map) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));
    const int64_t gs2 = (int64_t)x->c.grid_size * x->c.grid_size;
    u64 *h = (u64 *)malloc(sizeof(u64) * gs2);
    for (int64_t i = 0; i < gs2; ++i) h[i] = isfinite(max_height[i]) ? ((u64)((int64_t)max_height[i] + 1) << 40) : 0ull;
    hipError_t e = hipMemcpy(x->hmap, h, sizeof(u64) * gs2, hipMemcpyHostToDevice);
    free(h);
    BSC_HIP(e);
    BSC_HIP(hipMemcpy(x->cv_map, cv_map, 3 * gs2, hipMemcpyHostToDevice));
    return BSC_OK;
}

extern "C" bsc_status bsc_keys_dev(bsc_ctx *x, const int32_t **keys_dev, int64_t *max_id)
{
    if (!x || !keys_dev || !max_id) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(read_scalars(x));
    *keys_dev = x->rgb_pos;
    *max_id = x->hscal[DS_MAX_ID];
    return BSC_OK;
}

extern "C" bsc_status bsc_kernel_stats(bsc_ctx *x, int32_t which, int32_t reset, double *out)
{
    if (!x || !out || which < 0 || which >= BSC_STAT_SLOTS) return BSC_E_INVALID;
    BSC_HIP(hipSetDevice(x->device));
    BSC_TRY(sync_all(x));
    const int n = x->ev_n[which], m = n < BSC_EV_RING ? n : BSC_EV_RING;
    double total = 0.0;
    for (int i = n - m; i < n; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, x->ev[which][2 * (i % BSC_EV_RING)], x->ev[which][2 * (i % BSC_EV_RING) + 1]) == hipSuccess)
            total += ms;
    }
    out[0] = total;                 // ms summed over the last m launches
    out[1] = (double)m;             // launches covered
    out[2] = x->stat_bytes[which];  // algorithmic bytes accumulated since the last reset (localize only)
    out[3] = (double)n;             // launches since the last reset
    if (reset) { x->ev_n[which] = 0; x->stat_bytes[which] = 0.0; }
    return BSC_OK;
}
