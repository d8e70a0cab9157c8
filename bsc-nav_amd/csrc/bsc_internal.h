// bsc_internal.h — shared declarations of libbscnav.so (gfx950 only; no CPU path).
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bscnav.h"

typedef unsigned long long u64;
// per-point payload of the rgb chain: one 12-byte gather per point
struct PointRec {
    uint32_t alo, ahi; // exp(-r^2 / 1.2) as a double, low / high word   (memory_2.py:875)
    uint32_t rgbv;     // rgb[py, px] packed r | g << 8 | b << 16   (memory_2.py:870)
};
#define BSC_EV_RING 512
// timed stages (bsc_kernel_stats `which`)
enum {
    BSC_STAT_DENSE = 0,    // k_dense_reduce
    BSC_STAT_COSINE = 1,   // cosine scan of bsc_localize
    BSC_STAT_POINTS = 2,   // k_points
    BSC_STAT_PAIRS = 3,    // k_keys_pairs + tile scan + k_pair_compact
    BSC_STAT_ORDER = 4,    // new-voxel ids, k_runs, run sort, scan, k_expand, segment order (everything the rgb chain needs)
    BSC_STAT_PAIRSORT = 5, // pair sort + segment heads
    BSC_STAT_INGEST = 6,   // main-stream work of one bsc_ingest call, first to last kernel
    BSC_STAT_CHAIN = 7,    // k_chain + k_hwin (side stream)
    BSC_STAT_SLOTS = 8
};

// device scalar block indices (int64 each)
enum {
    DS_MAX_ID = 0,      // number of voxel ids assigned so far (memory_2.py max_id)
    DS_MAX_ID_PREV = 1, // max_id before the batch in flight (voxels >= this are new in the batch)
    DS_NPASS_TOTAL = 2, // points that passed all checks since creation
    DS_NSEEN_TOTAL = 3,
    DS_POOL_N = 4,      // rows used in the token pool (exact mode)
    DS_ERROR = 5,       // sticky device-side error flag (capacity)
    DS_B_NPASS = 6,     // batch: passing points
    DS_B_NFIRST = 7,    // batch: first-touch points (= new voxels)
    DS_N_HITS = 8,      // flush: rows that met a full voxel
    DS_B_NSEG = 9,      // batch: voxel segments in the sorted point list
    DS_RMW_TOTAL = 10,  // dense: voxel rows read-modify-written since creation
    DS_B_NPAIR = 11,    // batch: unique (voxel, frame, patch) pairs emitted by the tiles
    DS_TMP0 = 12,       // bsc_counters scratch (store voxels / tokens)
    DS_TMP1 = 13,
    DS_B_NPSEG = 14,    // batch: voxel segments of the sorted pair list
    DS_PAIR_TOTAL = 15, // dense: pairs reduced since creation
    DS_B_NRUN = 16,     // batch: runs of consecutive points that fall into the same cell
    DS_B_NNEW = 17,     // batch: cells listed as new by their first claimer (== new voxels; before the capacity clip)
    DS_COUNT = 18
};

// workspace of the in-tree radix sort (radix.hip), one per stream that sorts
struct RadixWs {
    u64 *status;               // per tile and digit: epoch << 32 | state << 30 | count
    uint32_t *partial;         // histogram partials of the workgroups of k_radix_hist
    uint32_t *gpartial;        // ... added up per group of 32 workgroups
    uint32_t *goff;            // (4, 256) exclusive digit offsets of the passes
    uint32_t *tickets;         // [0] k_radix_hist's last-group ticket, [1 + p] the tile ticket of pass p, [8 + g] the ticket of group g
    uint32_t *tmp_k, *tmp_v;   // intermediate passes
    size_t max_items;
    uint32_t epoch;            // host counter: one value per (sort, pass)
};

struct bsc_ctx {
    bsc_config c;
    int device;
    hipStream_t stream;
    int nh;
    int64_t ncell;
    int g2;
    // ---- persistent state in HBM ----
    int32_t *occ;      // (gs,gs,nh) voxel id or -1 (memory_2.py occupied_ids)
    int32_t *rgb_pos;  // (vcap+1,3)
    uint8_t *rgb;      // (vcap,3)
    float *weight;     // (vcap)
    u64 *hmap;         // (gs,gs) packed (h+1)<<40 | order  (max_height + tie order)
    uint8_t *cv_map;   // (gs,gs,3)
    // fast geometry (geometry_dev.h geom_point_fast): pinhole intrinsics + per-pixel patch tables, verified at creation
    bool geom_fast;
    bool proj_id;              // K Kinv p2d == p2d up to rounding and min_depth >= 0: source pixel without the division (GeomConst.proj_id)
    bool pat_all_in;           // no pixel centre outside the patch grid: the every-pixel dense build needs no patch-table look-ups
    bool long_chain;           // segments of >= 64 points go to the wavefront-per-voxel chain (BSC_QUAD_CHAIN_ONLY unsets)
    // 8-byte point records for the every-pixel dense build (geometry_dev.h rec8_*): possible when float_as_uint over the valid
    // depth range (min_depth, max_depth) spans fewer than 2^28 values (BSC_REC12=1 keeps the 12-byte {alpha, rgb} records)
    bool rec8_ok;
    uint32_t rec8_zbase;       // bits of the largest float <= min_depth
    bool rec8_s[2];            // record format of the call in scratch set k (its chain is launched later)
    uint8_t *pat_x, *pat_y;   // (W), (H): patch column / row of a pixel column / row, 255 = outside the patch grid
    double *exp_tab;          // 64 x (hi, lo) of 2^(j/64): the table of bsc_exp (geometry_dev.h)
    // patch-aligned pair tiles (dense.hip): pixel rectangle {x0, width, y0, pixels} of every patch and the start of its
    // staging slice inside a frame; valid when every patch covers at most 3328 pixels
    bool patch_tiles;
    int32_t *pt_rect, *pt_off;
    int pair_path;            // how the pairs of the batch in flight were built: 0 generic tiles, 1 patch tiles
    int64_t *dscal;    // DS_COUNT device scalars
    int64_t *hscal;    // pinned host mirror for readbacks
    int64_t *mail, *mail_dev;  // DS_COUNT scalars + a sequence number, in coherent pinned host memory: written by k_block_totals, polled by
    int64_t mail_seq;          // the host (ingest_batch) — no event / copy / synchronize between the front end and the order stage
    // exact mode
    float *cache_f;     // (iter_size,D)
    int32_t *cache_pos; // (iter_size,3)
    float *cache_d;     // (iter_size)
    int64_t iter_id;    // host mirror of the cache fill
    float *pool;        // (token_capacity,D)
    float *pool_d;      // (token_capacity)
    int32_t *store_rows; // (vcap+1,cache_size) pool rows; entry vcap is the grid_0_0_0 group
    int32_t *store_cnt;  // (vcap+1)
    int64_t n_flush;
    int64_t pool_n_host;   // host mirror of DS_POOL_N (token pool rows in use), exact after every flush / import / reset
    // dense modes
    float *acc;   // (vcap,D)
    int32_t *acnt; // (vcap)
    // point log for the exact colour merge across ranks (bsc_point_log_*): cells + records of every ingested point
    int32_t *log_cell;
    PointRec *log_rec;
    int64_t log_cap, log_n;
    bool log_stale;            // the colour state was imported / replaced while the log was on: the log no longer describes it
    // ---- per-batch scratch (max_points) ----
    int32_t *p_cell;
    uint32_t *p_patf;
    PointRec *p_rec_s[2];    // double-buffered: read by the rgb chain on the side stream
    float *p_r2f;
    // Points are ordered per voxel through their RUNS (maximal stretches of consecutive points in one cell; a 10 cm
    // voxel a few metres away covers ~16 pixels of an image row): the runs are sorted by voxel id (stable radix sort
    // keeps the order j inside a voxel) and expanded back into the per-voxel point order the rgb chain walks.
    int32_t *new_cells;                 // cells claimed for the first time in this batch (one entry per new voxel)
    int32_t *blk_cnt, *blk_off;         // per 1024-point block: runs (count / exclusive prefix); also head compaction
    int32_t *blk_pass, *blk_pass_off;   // per block: passing points
    int32_t *hb_cnt, *hb_off;           // segment-head compaction (dense.hip compact_heads): per-block counts / offsets
    int64_t nblk_cap;
    uint32_t *stage_cell, *stage_pos;   // k_points: the runs of every block of points (cell, first position), slice b * GB
    int group_rpw;                      // rounds of 64 points per wavefront in k_points: 4 (1024-point blocks) or 8 (2048; BSC_GROUP_RPW)
    uint32_t *skey_a, *sval_a;          // run sort input: key = voxel id | (length - 1) << id bits, value = first point
    uint32_t *skey_b_s[2];              // sorted run keys
    uint32_t *run_val_b;                // sorted run values
    int64_t *run_scan;                  // per 1024-run block: sum, then exclusive prefix, of (length | segment head << 32)
    int32_t *seg_k0, *seg_vid;          // voxel segments of the point order: first position, voxel id
    uint32_t *sval_b_s[2];              // point order: j of the k-th point, voxel by voxel — written only where the quad chain reads it
                                        // (short segments, the first points of new voxels: k_expand_short)
    // the point order by RUNS (read by k_chain_long): position k belongs to the run whose start is the last set bit at or below k
    u64 *run_bits_s[2];                 // bit k: a run starts at position k
    uint32_t *ck_run_s[2], *ck_start_s[2];   // per 64 positions: the sorted run that covers position 64 w, and where it starts
    uint32_t *run_val_s[2];             // sorted run values (first record of the run), per scratch set
    int4 *seg_info_s[2];                // per voxel segment: {first k, end k, voxel id, rank in the length-class order}
    u64 *f_keys_a, *f_keys_b;  // flush-private sort buffers (iter_size)
    int32_t *pass_list;
    int32_t *seg_last_s[2];
    int64_t *bscal_s[2];       // per-set scalars: [0] voxel segments of the batch, [1] max_id before the batch
    int cur_set;
    hipStream_t side;          // rgb chain + top-down map run here, overlapped with the dense reduce / next encoder
    hipStream_t side2;         // the one-wavefront segments of the long chain, beside the hot tiles on `side`
    hipEvent_t ev_chain0, ev_mid;
    hipEvent_t ev_ready[2], ev_done[2];
    bool ev_done_valid[2];
    bool chain_pending;        // the last call's rgb chain / top-down map kernels are still to be launched (deferred)
    int chain_set;
    int last_chain_set;        // scratch set of the most recently launched chain (-1: none since the last reset)
    int64_t chain_order_base;
    int64_t chain_points;      // points of the batch whose chain is pending
    u64 *pair_key_a, *pair_key_b;   // dense: voxel id << cb | frame << pb | patch
    u64 *pstage_key;                // per-tile staging of the LDS-aggregated pairs
    uint32_t *pstage_cnt;
    int32_t *tile_cnt, *tile_off;
    int64_t max_tiles;
    uint32_t *pair_cnt_a, *pair_cnt_b;
    int32_t *pseg_start;
    int64_t pair_cap;
    double *d_transforms; // (max_frames,16)
    int64_t *d_offsets;   // (max_frames+1)
    int max_frames;
    // flush scratch (iter_size)
    int32_t *f_rowdst, *f_hit, *f_hidx, *f_rowseg, *f_rowe, *f_headpos, *f_win;
    uint32_t *f_draws;
    int64_t draws_cap;
    // localize scratch
    float *l_sims;       // per token row / voxel row
    int64_t l_sims_cap, l_out_pos_cap, l_out_sim_cap;   // bytes
    u64 *l_key_a, *l_key_b;
    uint32_t *l_val_a, *l_val_b;
    uint32_t *l_name_rank;
    float *l_q;          // normalised queries
    uint16_t *l_qp;      // their three bf16 pieces (3, 1024, D) for the bf16x3 scan
    u64 *l_sel_key[2];   // batched top-K selection rounds (grown on demand)
    uint32_t *l_sel_val[2];
    u64 *l_sel_thr;      // per-query threshold keys of the sample selection
    int32_t *l_sel_cnt;  // per-query survivor counts
    uint32_t *l_valid;   // one bit per dense row: the row holds points (what the filter needs of acnt)
    int64_t l_sel_cap[7];
    int32_t *l_out_pos;
    float *l_out_sim;
    bool names_dirty;
    // per-row operand scale + inverse norm of the rows the batched fp16-piece scan reads (dense rows or token-pool rows), rebuilt —
    // like the name ranks — only after the rows changed (ingest, flush, imports, merges, reset)
    float2 *l_rscale;
    int64_t l_rscale_cap;
    bool row_scale_dirty;
    bool rscale_from_reduce;        // the last ingest's dense reduce wrote the scale / inverse norm of every row it finished
                                    // (k_dense_reduce_voxels): the scan's k_row_scale pass is then only needed after imports
    int last_nq, last_K;            // shape of the last bsc_localize call (its top-K stays resident for clustering)
    int32_t last_counts[1024];
    // frontier helpers (allocated on first use, gs*gs each)
    uint8_t *fr_mask, *fr_in;
    int32_t *fr_parent, *fr_size, *fr_ord, *fr_roots, *fr_labels, *fr_first, *fr_sizes, *fr_scal;
    unsigned long long *fr_sumx, *fr_sumy;
    double *fr_centers, *fr_gains;
    // primitives workspace
    void *prim_tmp;
    size_t prim_tmp_bytes;
    void *prim_tmp_side;       // second rocPRIM workspace: the order pipeline sorts on the side stream beside the pair sort
    RadixWs rx_main, rx_side;  // in-tree radix sort: the pair sort (main stream) and the run sort (side stream) may run at the same time
    bool radix_intree;         // BSC_SORT_ROCPRIM=1: both sorts through rocPRIM as until round 5
    bool order_on_side;        // per-voxel point order (k_runs .. k_seg_order) on the side stream (BSC_ORDER_MAIN=1 keeps it on the main stream)
    hipEvent_t ev_ids, ev_runs;   // voxel ids assigned; side: k_runs has read the call's cells / block offsets
    hipEvent_t ev_tot;            // main: k_totals done (the call's run / new-voxel counts exist)
    hipEvent_t ev_psort;          // main: the pair sort of the last call is done (the rgb chain starts behind it, launch_pending_chain)
    bool ev_psort_valid;
    hipStream_t copy;             // early readback of those counts while the main stream goes on with the pair tiles
    void *h2d_pin[4];             // pinned staging of the pageable-host imports (capi.hip h2d_pipelined), allocated on first use
    hipEvent_t h2d_ev[4];
    bool ev_runs_valid;
    int last_order_set;        // scratch set of the last order stage enqueued on the side stream (-1: none): its ev_ready marks it complete
    // bookkeeping
    int64_t order_base; // global point counter (top-down map tie order)
    // HIP-event rings around the stages of the path (BSC_STAT_*), recorded on the stream the stage is launched on
    hipEvent_t ev[BSC_STAT_SLOTS][2 * BSC_EV_RING];
    int ev_n[BSC_STAT_SLOTS];
    double stat_bytes[BSC_STAT_SLOTS];
    bool timing;
};

void bsc_set_error(const char *fmt, ...);
#define BSC_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            bsc_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return BSC_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)
#define BSC_TRY(expr)                 \
    do {                              \
        bsc_status _s = (expr);       \
        if (_s != BSC_OK) return _s;  \
    } while (0)

// ---- primitives (prims.hip; rocPRIM device-wide sort and scans) ----
size_t prim_workspace_bytes(size_t max_items);
bsc_status prim_sort_keys(bsc_ctx *x, const u64 *in, u64 *out, size_t n, int begin_bit, int end_bit);
bsc_status prim_sort_pairs(bsc_ctx *x, const u64 *kin, u64 *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                           int begin_bit, int end_bit);
bsc_status prim_sort_pairs_onesweep(bsc_ctx *x, const u64 *kin, u64 *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                                    int begin_bit, int end_bit);
bsc_status prim_sort_pairs_u32(bsc_ctx *x, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                               size_t n, int begin_bit, int end_bit);
bsc_status prim_sort_pairs_u32_onesweep(bsc_ctx *x, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                                        size_t n, int begin_bit, int end_bit);
// in-tree onesweep radix sort of (u32 key, u32 value) pairs on the key bits [b0, b1): stable, input preserved (radix.hip)
bsc_status radix_ws_create(RadixWs *ws, size_t max_items);
void radix_ws_destroy(RadixWs *ws);
bsc_status radix_sort_pairs_u32(bsc_ctx *x, RadixWs *ws, hipStream_t st, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                                uint32_t *vout, size_t n, int b0, int b1);
bsc_status prim_exclusive_sum_i64(bsc_ctx *x, const int64_t *in, int64_t *out, size_t n);
bsc_status prim_exclusive_sum_i32(bsc_ctx *x, const int32_t *in, int32_t *out, size_t n);
bsc_status prim_inclusive_max_i32(bsc_ctx *x, const int32_t *in, int32_t *out, size_t n);

// ---- kernels launchers ----
bsc_status launch_geometry_debug(bsc_ctx *x, const float *depth, const int32_t *idx, int64_t P, uint8_t *flags,
                                 double *pc, double *pg, int32_t *vox, int32_t *pix, int32_t *pat, double *r2,
                                 double *alpha);
// token rows come as f32 (the reference's DINOv2 output) or as bf16 (what a bf16 encoder emits; widened exactly)
typedef uint16_t bf16_t;
__device__ __forceinline__ float4 load_tok4(const float *row, int v) { return ((const float4 *)row)[v]; }
__device__ __forceinline__ float4 load_tok4(const bf16_t *row, int v)
{
    const uint2 raw = ((const uint2 *)row)[v];
    return make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                       __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u));
}

// (power-of-two operand scale, 1 / (norm scale 2^11)) of a token row from the sum of its squares: what the fp16-piece batched scan
// (localize.hip k_cosine_f16x2) reads per row; computed by k_row_scale (one pass over the rows) and, row by row, by the dense reduce
__device__ __forceinline__ float2 bsc_row_scale_of(float sumsq)
{
    const float nrm = sqrtf(sumsq);
    int e = 0;
    if (nrm > 0.f && nrm < INFINITY) (void)frexpf(nrm, &e);          // nrm = m 2^e, m in [0.5, 1)
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float sc = (nrm > 0.f && nrm < INFINITY) ? ldexpf(1.0f, 12 - e) : 1.0f;     // norm * sc in [2^11, 2^12)
    // the result leaves as acc * (1 / (sc 2^11)) / max(norm, 1e-8): the two powers of two are exact factors
    return make_float2(sc, (1.0f / fmaxf(nrm, 1e-8f)) / sc * (1.0f / 2048.0f));
}

bsc_status ingest_batch(bsc_ctx *x, int32_t n_frames, const float *depth, const uint8_t *rgb, int32_t rgb_ch,
                        const void *tokens, int token_dtype, const int32_t *idx, const int64_t *offsets_host,
                        const double *alpha, bsc_draw_fn draw, void *user);
bsc_status flush_cache(bsc_ctx *x, bsc_draw_fn draw, void *user);
bsc_status grow_token_pool(bsc_ctx *x, int64_t need_rows);   // exact mode: the token store is unbounded like the reference's
bsc_status launch_pending_chain(bsc_ctx *x);
bsc_status launch_keys_pairs(bsc_ctx *x, int64_t P, int n_frames, bool all_pixels, const uint32_t *p_patf);
bsc_status frontier_mask_impl(bsc_ctx *x, const uint8_t *navigable_host, uint8_t *mask_host);
bsc_status frontier_clusters_impl(bsc_ctx *x, const uint8_t *frontier_host, int32_t min_cluster_size, int32_t ig_radius,
                                  int32_t max_clusters, int32_t *n_clusters_host, int32_t *labels_host,
                                  int32_t *first_host, int32_t *sizes_host, double *centers_host, double *gains_host,
                                  int32_t *best_host);
bsc_status dense_reduce_batch(bsc_ctx *x, const void *tokens, int token_dtype, int n_frames);
// list of segment starts of a sorted key array (segments = runs of equal key >> shift; keys == invalid are skipped);
// the number of segments is written to *count_dev.  Deterministic: per-block counts + exclusive scan.
bsc_status compact_heads_u64(bsc_ctx *x, const u64 *keys, int64_t n, int shift, int32_t *out, int64_t *count_dev);
bsc_status localize_impl(bsc_ctx *x, const float *q_dev, int32_t nq, int32_t K, double radius, const int32_t *curr,
                         int32_t floor_lo, int32_t floor_hi, int32_t *out_pos, float *out_sim, int32_t *out_count);
int64_t sims_row_stride(int64_t n_rows);
bsc_status pool_query_impl(bsc_ctx *x, const float *tokens, int32_t B, int32_t T, int32_t D, float *out);
bsc_status read_scalars(bsc_ctx *x); // dscal -> hscal (synchronises the main stream)
void localize_prepare(bsc_ctx *x);   // name ranks + row scales of the batched scan, eagerly (localize.hip; the imports call it)
bsc_status sync_all(bsc_ctx *x);     // main + side stream
// record the start / stop event of launch number ev_n[which] (ring; older launches are overwritten)
static inline void stat_begin(bsc_ctx *x, int which, hipStream_t s = nullptr)
{
    if (x->timing) (void)hipEventRecord(x->ev[which][2 * (x->ev_n[which] % BSC_EV_RING)], s ? s : x->stream);
}
static inline void stat_end(bsc_ctx *x, int which, double bytes, hipStream_t s = nullptr)
{
    if (x->timing) {
        (void)hipEventRecord(x->ev[which][2 * (x->ev_n[which] % BSC_EV_RING) + 1], s ? s : x->stream);
        x->ev_n[which]++;
        x->stat_bytes[which] += bytes;
    }
}

static inline int ceil_log2_u64(uint64_t v)
{
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}
