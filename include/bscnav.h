/*
 * bscnav.h — C-ABI of libbscnav.so, the MI355X (gfx950) implementation of BSC-Nav's
 * structured-spatial-memory construction and query path.
 *
 * The reference (Heathcliff-saku/BSC-Nav) has no native code and no FFI: its boundary for
 * this path is the Python class VoxelTokenMemory (memory_2.py:38).  The entry points below
 * are what a ctypes binding inside that class binds instead of the per-point Python loops;
 * each one cites the reference lines it replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns a bsc_status
 *     (0 = ok, negative = error) and bsc_last_error() returns a thread-local message.
 *   - "dev" pointers are device (HBM) addresses owned by the caller (e.g. torch tensors);
 *     "host" pointers are ordinary host memory.  The library owns only the state inside
 *     bsc_ctx and copies in/out through the export/import calls.
 *   - one bsc_ctx per GPU; a ctx is not thread-safe; calls are ordered on the stream given
 *     at creation (pass the caller's hipStream_t, or NULL for the default stream).
 *   - there is NO CPU fallback: bsc_create fails when no gfx950 device is present.
 */
#ifndef BSCNAV_H
#define BSCNAV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t bsc_status;
#define BSC_OK 0
#define BSC_E_INVALID (-1)   /* bad argument / configuration                       */
#define BSC_E_CAPACITY (-2)  /* voxel or token capacity exceeded (memory_2.py:715 overflows silently) */
#define BSC_E_HIP (-3)       /* HIP runtime error                                   */
#define BSC_E_STATE (-4)     /* call not valid in the ctx's feature mode            */

/* feature modes */
#define BSC_MODE_EXACT 0     /* reference semantics: token cache + <=cache_size tokens per voxel */
#define BSC_MODE_MEAN 1      /* dense reduce: per-voxel sum + count (north-star mode)             */
#define BSC_MODE_MAX 2       /* dense reduce: per-voxel element-wise max                          */

typedef struct bsc_config {
    int32_t height, width;       /* frame H, W (args.py:27-28)                                  */
    int32_t grid_size;           /* gs (args.py:58)                                             */
    int32_t min_h, max_h;        /* int(floor_height/cs), int(map_height/cs) (memory_2.py:122-123) */
    int32_t patch_grid;          /* g, tokens per side (memory_2.py:80-83)                      */
    int32_t token_dim;           /* D (memory_2.py:107)                                         */
    int32_t iter_size;           /* token-cache rows (memory_2.py:109)                          */
    int32_t cache_size;          /* tokens per voxel (memory_2.py:111)                          */
    int32_t mode;                /* BSC_MODE_*                                                  */
    int32_t voxel_capacity;      /* rows of grid_rgb/weight/... (reference: gs*gs, memory_2.py:715) */
    int32_t max_points;          /* largest number of points one bsc_ingest call may carry      */
    int64_t token_capacity;      /* rows of the per-voxel token pool (exact mode)               */
    double cell_size;            /* cs (args.py:57)                                             */
    double min_depth, max_depth; /* strict bounds (utils.py:175-177)                            */
    double K[9];                 /* calib_mat, row-major (utils.py:181-186)                     */
    double Kinv[9];              /* np.linalg.inv(calib_mat) (utils.py:164)                     */
    double Kpatch[9];            /* patch-grid intrinsics (utils.py:144-150)                    */
} bsc_config;

typedef struct bsc_ctx bsc_ctx;

/* Replacement-index source for full voxels: fill out[0..n) with draws in [0, cache_size)
 * in the order the reference would call random.choice(range(cache_size)) (memory_2.py:352). */
typedef void (*bsc_draw_fn)(void *user, uint32_t n, uint32_t *out);

const char *bsc_last_error(void);
const char *bsc_version(void);

/* VoxelTokenMemory.__init__/_init_cache (memory_2.py:39,708-722): allocate all state in HBM. */
bsc_status bsc_create(const bsc_config *cfg, int32_t device, void *hip_stream, bsc_ctx **out);
void bsc_destroy(bsc_ctx *ctx);
/* _init_cache again (load_memory, memory_2.py:172-184) */
bsc_status bsc_reset(bsc_ctx *ctx);

/* obs2voxeltoken's per-point loop (memory_2.py:859-903) for a batch of n_frames frames.
 *   depth_dev   (n_frames,H,W) f32          rgb_dev (n_frames,H,W,rgb_channels) u8
 *   tokens_dev  (n_frames,g,g,D) f32        transforms_host (n_frames,16) f64 = pc_transform (memory_2.py:860)
 *   sample_idx_dev  int32 pixel indices, frame f owns [offsets_host[f], offsets_host[f+1]) in the
 *                   reference's shuffled order (memory_2.py:747-749); NULL = every pixel, row-major
 *   alpha_dev   optional f64 per point (same indexing as sample_idx): host-computed
 *               exp(-r2/1.2) (memory_2.py:873-875); NULL = computed on the device
 * Frame order and point order inside the call define the sequential semantics (first-touch ids,
 * rgb running mean, top-down map ties, token-cache order).  In exact mode draw() is called when a
 * token-cache flush meets full voxels. */
bsc_status bsc_ingest(bsc_ctx *ctx, int32_t n_frames, const float *depth_dev, const uint8_t *rgb_dev,
                      int32_t rgb_channels, const float *tokens_dev, const double *transforms_host,
                      const int32_t *sample_idx_dev, const int64_t *offsets_host, const double *alpha_dev,
                      bsc_draw_fn draw, void *user);

/* bsc_ingest for an encoder that emits half-width tokens: tokens_dev is (n_frames,g,g,D) of token_dtype.
 * BSC_TOK_BF16 rows are widened exactly to f32 on load, so the result equals bsc_ingest on the widened tokens
 * (memory_2.py:883 stores whatever _get_patch_token returned); accumulators, caches and the store stay f32. */
#define BSC_TOK_F32 0
#define BSC_TOK_BF16 1
bsc_status bsc_ingest_typed(bsc_ctx *ctx, int32_t n_frames, const float *depth_dev, const uint8_t *rgb_dev,
                            int32_t rgb_channels, const void *tokens_dev, int32_t token_dtype,
                            const double *transforms_host, const int32_t *sample_idx_dev, const int64_t *offsets_host,
                            const double *alpha_dev, bsc_draw_fn draw, void *user);

/* update_memory_dist_base (memory_2.py:326-358): all iter_size rows incl. the zero rows. */
bsc_status bsc_flush(bsc_ctx *ctx, bsc_draw_fn draw, void *user);

/* counters (host sync): out[0]=max_id out[1]=iter_id out[2]=store voxels out[3]=store tokens
 * out[4]=flushes out[5]=points passed so far out[6]=points seen so far out[7]=voxel-row RMWs (dense)
 * out[8]=(voxel,frame,patch) pairs reduced so far (dense) out[9]=pairs of the last call */
bsc_status bsc_counters(bsc_ctx *ctx, int64_t *out10_host);

/* geometry only (utils.py:153-214, memory_2.py:864-875) for one frame, outputs to host; NULL skips.
 * vox is row,col,h before the -min_h shift; flags bit0 depth-valid, bit1 in-range, bit2 patch-in-range */
bsc_status bsc_geometry(bsc_ctx *ctx, const float *depth_dev, const double *transform_host,
                        const int32_t *sample_idx_dev, int64_t n_points, uint8_t *flags_host, double *pc_host,
                        double *pg_host, int32_t *vox_host, int32_t *pix_host, int32_t *pat_host,
                        double *r2_host, double *alpha_host);

/* the stable radix sort of (u32 key, u32 value) pairs on the key bits [begin_bit, end_bit) that orders a voxel's points and
 * tokens inside bsc_ingest (memory_2.py:888-903: colour and mean of a voxel are defined by the order of its points) — exposed for
 * the parity tests; device arrays, input preserved, at most max(max_points, voxel_capacity + 1) items, on the context's stream */
bsc_status bsc_sort_pairs_u32(bsc_ctx *ctx, const uint32_t *keys_dev, const uint32_t *vals_dev, int64_t n, int32_t begin_bit,
                              int32_t end_bit, uint32_t *keys_out_dev, uint32_t *vals_out_dev);

/* on-disk layout exchange (memory_2.py:1136-1145 save, :189-200 load); host buffers sized from bsc_counters */
bsc_status bsc_export_rgb(bsc_ctx *ctx, int32_t *pos_host, uint8_t *rgb_host, float *weight_host);
bsc_status bsc_export_occupied(bsc_ctx *ctx, int32_t *occ_host /* (gs,gs,max_h-min_h) */);
bsc_status bsc_export_heightmap(bsc_ctx *ctx, double *max_height_host, uint8_t *cv_map_host);
bsc_status bsc_export_cache(bsc_ctx *ctx, float *feat_host, int32_t *pos_host, float *dis_host);
/* feature store in HDF5 name order: pos (V,3), cnt (V), feats (T,D), dists (T)  (feat.h5df, memory_2.py:330-354) */
bsc_status bsc_export_store(bsc_ctx *ctx, int32_t *pos_host, int32_t *cnt_host, float *feats_host, float *dists_host);
/* dense modes: accumulator rows in voxel-id order: acc (max_id,D) sum-or-max, cnt (max_id) */
bsc_status bsc_export_dense(bsc_ctx *ctx, float *acc_host, int32_t *cnt_host);
bsc_status bsc_import_rgb(bsc_ctx *ctx, int64_t max_id, const int32_t *pos_host, const uint8_t *rgb_host,
                          const float *weight_host);
bsc_status bsc_import_store(bsc_ctx *ctx, int64_t n_voxels, int64_t n_tokens, const int32_t *pos_host,
                            const int32_t *cnt_host, const float *feats_host, const float *dists_host);
bsc_status bsc_import_dense(bsc_ctx *ctx, int64_t max_id, const float *acc_host, const int32_t *cnt_host);

/* voxel_localized, query pooling (memory_2.py:591-608): tokens_dev (B,T,D) -> out_dev (D) */
bsc_status bsc_pool_query(bsc_ctx *ctx, const float *tokens_dev, int32_t B, int32_t T, int32_t D, float *out_dev);

/* voxel_localized scan (memory_2.py:623-671) for n_queries pooled queries q_dev (Q,D):
 * cosine vs every stored token, per-voxel max, stable top-K in HDF5 name order.
 * radius<0 disables the sphere filter (:624-629); floor_lo>floor_hi disables the floor filter (:633-640).
 * out_pos_host (Q,K,3) i32, out_sim_host (Q,K) f32, out_count_host (Q) = rows actually written.
 * Arithmetic: f32 dot products and norms (F.cosine_similarity, :655) — on the vector ALUs up to 4 queries, on the f32 matrix
 * cores up to 64, and beyond that on the bf16 matrix cores with every f32 operand split exactly into three bf16 pieces (the six
 * piece products of weight >= 2^-16, f32 accumulation: truncation 2^-24 per product, scores within 3e-6 of an fp64 scan). */
bsc_status bsc_localize(bsc_ctx *ctx, const float *q_dev, int32_t n_queries, int32_t K, double radius,
                        const int32_t *curr_host, int32_t floor_lo, int32_t floor_hi, int32_t *out_pos_host,
                        float *out_sim_host, int32_t *out_count_host);

/* GESObjectNavRobot.weighted_cluster_centers (BSCAgent.py:479-497), the consumer of voxel_localized's output:
 * DBSCAN(eps, min_samples) over K top-ranked positions (scikit-learn semantics), similarity-weighted centres,
 * clusters ordered by mean similarity (stable, descending).  pos_host (K,3) / sim_host (K) NULL = cluster the first K
 * results of query `query_index` of the last bsc_localize call, which are still resident in HBM.
 * centers_host (K,3) f64 (first n_clusters rows valid), labels_host (K) (-1 noise), sizes_host (K).  K <= 1024. */
bsc_status bsc_cluster_centers(bsc_ctx *ctx, int32_t query_index, int32_t K, const int32_t *pos_host,
                               const float *sim_host, double eps, int32_t min_samples, double *centers_host,
                               int32_t *labels_host, int32_t *sizes_host, int32_t *n_clusters_host);

/* FrontierExplorer helpers (memory_2.py:1147-1311) on the resident top-down colour map (cv_map, kept by bsc_ingest).
 *   bsc_frontier_mask     is_unknown / is_known / build_navigable_mask / find_frontiers (:1165-1207):
 *                         mask_host[x*gs+y] bit0 = known (cv_map[x,y].sum() != 0), bit1 = frontier cell (known, navigable,
 *                         an in-bounds 4-neighbour unknown).  navigable_host (gs,gs) u8 comes from the simulator's
 *                         pathfinder (NULL = every cell navigable); the reference's navigable_mask works as well.
 *   bsc_frontier_clusters cluster_frontiers + compute_cluster_center + compute_information_gain +
 *                         select_best_cluster_center_by_ig (:1209-1311): 4-connected clusters of the frontier cells
 *                         (frontier_host (gs,gs) nonzero = frontier; NULL = the cells found by the last bsc_frontier_mask)
 *                         with >= min_cluster_size cells, in the reference's order (by first cell, row-major);
 *                         centers_host (n,2) f64 mean cell; gains_host (n) = unknown cells in the clipped (2r+1)^2 window
 *                         around the centre rounded half-to-even; *best_host = first cluster with the largest gain > 0
 *                         (-1: none); labels_host (gs,gs) cluster ordinal or -1 (optional); first_host (n,2) first cell.
 *                         *n_clusters_host = clusters found; at most max_clusters rows are written.
 *   bsc_import_cv_map     set the colour map (gs,gs,3) u8, e.g. a saved exploration state (heights restart). */
bsc_status bsc_frontier_mask(bsc_ctx *ctx, const uint8_t *navigable_host, uint8_t *mask_host);
bsc_status bsc_frontier_clusters(bsc_ctx *ctx, const uint8_t *frontier_host, int32_t min_cluster_size, int32_t ig_radius,
                                 int32_t max_clusters, int32_t *n_clusters_host, int32_t *labels_host, int32_t *first_host,
                                 int32_t *sizes_host, double *centers_host, double *gains_host, int32_t *best_host);
bsc_status bsc_import_cv_map(bsc_ctx *ctx, const uint8_t *cv_map_host);

/* multi-GPU merge helpers (dense modes).  The library never calls RCCL: the host
 * moves the buffers with torch.distributed and hands them back.
 *   bsc_dense_gather : rows of the local map for the given voxel keys -> acc_dev (n,D), cnt_dev (n);
 *                      keys this rank never touched give zeros (mean) / -inf (max) and count 0
 *   bsc_dense_gather_rgb : colour state of the same keys -> rgb_dev (n,3) u8, weight_dev (n) f32 (memory_2.py:888-899
 *                      grid_rgb / weight; weight 0 for keys this rank never touched)
 *   bsc_dense_replace_full: the map becomes exactly the n voxels handed in, ids 0..n-1 in the given order — keys
 *                      (grid_rgb_pos), feature rows, counts, rgb and weights together, occupied_ids rebuilt, so every
 *                      per-id array of the memory directory (memory_2.py:1136-1145) stays aligned.  rgb_dev and
 *                      weight_dev may both be NULL: colours and weights are zeroed.
 *   bsc_dense_replace: bsc_dense_replace_full without colours
 *   bsc_import_heightmap: top-down map state (memory_2.py:98-100,901-903) from host arrays: max_height (gs,gs) f64
 *                      (-inf = empty) and cv_map (gs,gs,3) u8 — the merged map of several ranks */
bsc_status bsc_dense_gather(bsc_ctx *ctx, int64_t n, const int32_t *keys_dev /* (n,3) */, float *acc_dev, int32_t *cnt_dev);
bsc_status bsc_dense_gather_rgb(bsc_ctx *ctx, int64_t n, const int32_t *keys_dev, uint8_t *rgb_dev, float *weight_dev);
bsc_status bsc_dense_replace(bsc_ctx *ctx, int64_t n, const int32_t *keys_dev, const float *acc_dev, const int32_t *cnt_dev);
bsc_status bsc_dense_replace_full(bsc_ctx *ctx, int64_t n, const int32_t *keys_dev, const float *acc_dev,
                                  const int32_t *cnt_dev, const uint8_t *rgb_dev, const float *weight_dev);
bsc_status bsc_import_heightmap(bsc_ctx *ctx, const double *max_height_host, const uint8_t *cv_map_host);
/* Exact colour across ranks (SURVEY.md §8e "rgb/weights exact via replay"; memory_2.py:888-899 defines the state).
 * The running colour mean truncates at every step, so per-rank results cannot be combined; what can be exchanged is the
 * points themselves.  With the log enabled every bsc_ingest call appends, for each point of the call in order j, its cell
 * ((row*gs + col)*(max_h-min_h) + h, or < 0: no voxel) and its 12-byte record {alpha f64 as lo/hi words, rgb packed
 * r | g<<8 | b<<16} — 16 bytes per point, meant for the sub-sampled modes (depth_sample_rate >= 50: a few thousand points per
 * frame).  The host sends each record to the rank that owns its voxel (torch.distributed all-to-all), the owner lists the
 * records voxel by voxel in global order (rank, then local order) and replays the chain:
 *   bsc_point_log_enable  allocate room for `capacity` points (0 disables and frees); cleared by bsc_reset
 *   bsc_point_log_read    *n_points = points logged so far; when the output pointers are given, the first min(n, capacity)
 *                         entries are copied to the caller's device buffers: cells (n) i32, records (n,3) u32.  An ingest
 *                         call that would overflow the log fails with BSC_E_CAPACITY before anything is changed
 *   bsc_replay_colour     stateless: n records sorted by voxel (vox_sorted ascending, values in [0, n_vox)), each voxel's
 *                         records in global point order -> rgb (n_vox,3) u8, weight (n_vox) f32 exactly as the sequential
 *                         loop leaves them (first point: colour copied, weight f32(0 + alpha); then
 *                         c = u8(trunc((f32(c*w) + r*alpha) / (w + alpha))), w = f32(w + alpha)); voxels without records: 0 */
bsc_status bsc_point_log_enable(bsc_ctx *ctx, int64_t capacity);
bsc_status bsc_point_log_read(bsc_ctx *ctx, int32_t *cells_out_dev, uint32_t *records_out_dev, int64_t capacity,
                              int64_t *n_points);
bsc_status bsc_replay_colour(int64_t n_records, const int32_t *vox_sorted_dev, const uint32_t *records_dev, int64_t n_vox,
                             uint8_t *rgb_dev, float *weight_dev, void *hip_stream);
/* device views for the host-side collective: voxel keys (max_id,3) i32 */
bsc_status bsc_keys_dev(bsc_ctx *ctx, const int32_t **keys_dev, int64_t *max_id);

/* Host helper (no device work): `idx = arange(n); np.random.shuffle(idx); idx[::rate]` of _backproject_depth
 * (memory_2.py:747-749), bit for bit, on a copy of NumPy's global MT19937 state (np.random.get_state()[1:3] =
 * key624 / pos, advanced in place so that np.random.set_state() leaves the stream where NumPy's own shuffle would).
 * scratch_n: n int32 of workspace; out: ceil(n / rate) int32. */
bsc_status bsc_host_shuffled_sample(uint32_t *key624, int32_t *pos, int64_t n, int32_t rate, int32_t *scratch_n,
                                    int32_t *out);

/* Host helper: n draws of Python's `random.choice(range(n_choices))` (the replacement index of a full voxel,
 * memory_2.py:352) on a copy of the `random` module's MT19937 state (random.getstate()[1]: 624 key words, then pos),
 * advanced in place — what a bsc_draw_fn that must stay on Python's stream can call instead of looping in Python. */
bsc_status bsc_host_choice_draws(uint32_t *key624, int32_t *pos, uint32_t n_choices, uint32_t n, uint32_t *out);

/* Encoder helper (stateless, bf16): s = x + delta ; y = LayerNorm(s)*gamma + beta, one pass over the (rows,width)
 * token matrix.  delta/xout may be NULL (plain LayerNorm).  Fuses the residual add and the LayerNorm that sit
 * between the library GEMMs of the ViT patch-feature provider (memory_2.py:738).  width % 256 == 0, <= 2048. */
bsc_status bsc_enc_add_layernorm(const void *x_dev, const void *delta_dev, const void *gamma_dev, const void *beta_dev,
                                 void *xout_dev, void *y_dev, int64_t rows, int32_t width, float eps, void *hip_stream);

/* Encoder helpers for the two ends of the transformer stack (bf16, width % 256 == 0, <= 2048):
 *   bsc_enc_embed_layernorm  token assembly + first LayerNorm: row 0 = cls + pos[0], rows 1..registers = register tokens,
 *                            the others = patch embedding (B, T-1-registers, width) + pos[1..] -> xout (B,T,width) and
 *                            y = LayerNorm(xout)   (replaces cat / add / LayerNorm passes)
 *   bsc_enc_final_layernorm  last residual add + final LayerNorm of the patch rows only (the first `skip` rows of every
 *                            image are dropped), written as the (B, T-skip, width) token tensor bsc_ingest reads:
 *                            f32 (out_f32 != 0: the bf16 result widened) or bf16 */
bsc_status bsc_enc_embed_layernorm(const void *patch_dev, const void *cls_dev, const void *reg_dev, const void *pos_dev,
                                   const void *gamma_dev, const void *beta_dev, void *xout_dev, void *y_dev, int32_t B,
                                   int32_t T, int32_t registers, int32_t width, float eps, void *hip_stream);
bsc_status bsc_enc_final_layernorm(const void *x_dev, const void *delta_dev, const void *gamma_dev, const void *beta_dev,
                                   void *out_dev, int32_t out_f32, int32_t B, int32_t T, int32_t skip, int32_t width,
                                   float eps, void *hip_stream);

/* LayerNorm of a bias-lagged residual stream (bf16 rows u, f32 vector bias_sum of `width`): y = LayerNorm(u + bias_sum).
 * The stream is the one the projection / fc2 GEMMs accumulate into with beta = 1 and no bias (encoder.py), bias_sum the sum
 * of the biases of the residual updates so far; same place in the reference's path as bsc_enc_add_layernorm
 * (memory_2.py:739, the ViT behind forward_features).  skip > 0 normalises rows [skip, T) of every image only and writes
 * them densely (final LayerNorm of the patch rows); out_f32 selects f32 output (bf16-rounded values widened). */
bsc_status bsc_enc_bias_layernorm(const void *u_dev, const void *bias_sum_f32_dev, const void *gamma_dev,
                                  const void *beta_dev, void *y_dev, int32_t out_f32, int32_t B, int32_t T, int32_t skip,
                                  int32_t width, float eps, void *hip_stream);

/* Encoder helper: softmax(Q K^T / sqrt(d)) V for the short sequences of the ViT provider, one workgroup per (image, head)
 * with K and V of the head resident in LDS.  qkv_dev (B,T,3,heads,head_dim) bf16 as the fused qkv GEMM writes it,
 * out_dev (B,T,heads*head_dim) bf16.  head_dim == 64, T <= 288 (ViT-B/16: 197, ViT-L/14 + 4 registers: 261). */
bsc_status bsc_enc_attention(const void *qkv_dev, int32_t B, int32_t T, int32_t heads, int32_t head_dim, void *out_dev,
                             void *hip_stream);
/* Same, with the (image, head) items handed to the persistent workgroups by a ticket counter instead of a static stride:
 * work2_dev = two int32 in device memory, zero before the first launch (the kernel re-arms them when it finishes;
 * consecutive launches on one stream may share them).  Robust when some CUs are busy with another stream's kernels. */
bsc_status bsc_enc_attention_dyn(const void *qkv_dev, int32_t B, int32_t T, int32_t heads, int32_t head_dim, void *out_dev,
                                 int32_t *work2_dev, void *hip_stream);

/* The encoder's dense layers at the REFERENCE'S precision (memory_2.py:43,738-739: DINOv2 runs f32) on the fp16 matrix cores:
 * every f32 operand as two fp16 pieces h = fp16(x), l = fp16(x - h) (22 significant bits), products hh + hl + lh accumulated
 * in f32 (encoder_gemm.hip).  Replaces the f32 nn.Linear calls of the reference's ViT forward.
 *   bsc_enc_split_weights  W (N,K) f32 as nn.Linear holds it -> pieces_dev: 2 planes of (ceil(N/256)*256, K) fp16 of scale * W
 *                          (scale: a power of two that puts the largest |w| near 8; once per weight matrix)
 *   bsc_enc_gemm_split     C (M,N) = epilogue(out_scale * (a_scale A) (scale W)^T + bias): epilogue 0 bias, 1 bias + GELU(tanh),
 *                          2 bias + residual (c_dev may alias resid_dev); out_scale = 1 / (a_scale * scale); K % 32 == 0;
 *                          |a_scale * A| < 65504 (fp16 range).  a_pieces != 0: a_dev holds the activation pieces already (layout
 *                          below; a_scale was applied by their producer) — otherwise f32 rows, split in registers.
 *                          c_pieces_scale != 0 (epilogues 0, 1): c_dev receives pieces of c_pieces_scale * C instead of f32.
 *   piece layout of an (M,K) activation matrix: fp16, row m = K/32 chunks of 64, chunk c = [h of k in 32c..32c+31 | l of the same]
 *   bsc_enc_layernorm_split  LayerNorm of f32 rows (width 256..1024) written as pieces of a_scale * y
 *   bsc_enc_split_rows       f32 rows -> pieces of a_scale * x
 *   bsc_enc_attention_split  softmax(Q K^T / sqrt(64)) V per (image, head) at f32 accuracy (three piece products per matrix
 *                            product, f32 softmax): qkv_pieces_dev = pieces of the (B T, 3 heads 64) output of the qkv GEMM
 *                            (c_pieces_scale 1), out = pieces of out_scale * attention (B T, heads 64), the projection GEMM's
 *                            operand; head_dim 64, T <= 288; work2_dev as bsc_enc_attention_dyn (NULL: static schedule) */
bsc_status bsc_enc_split_weights(const float *w_dev, int32_t N, int32_t K, float scale, void *pieces_dev, void *hip_stream);
bsc_status bsc_enc_gemm_split(const void *a_dev, int64_t M, int32_t K, const void *pieces_dev, int32_t N, const float *bias_dev,
                              const float *resid_dev, void *c_dev, float a_scale, float out_scale, int32_t epilogue,
                              int32_t a_pieces, float c_pieces_scale, void *hip_stream);
bsc_status bsc_enc_layernorm_split(const float *x_dev, const float *gamma_dev, const float *beta_dev, int64_t rows, int32_t width,
                                   float eps, float a_scale, void *pieces_dev, void *hip_stream);
bsc_status bsc_enc_split_rows(const float *x_dev, int64_t M, int32_t K, float a_scale, void *pieces_dev, void *hip_stream);
bsc_status bsc_enc_attention_split(const void *qkv_pieces_dev, int32_t B, int32_t T, int32_t heads, int32_t head_dim,
                                   void *out_pieces_dev, float out_scale, int32_t *work2_dev, void *hip_stream);
/* The same GEMM with the LayerNorms of the pre-LN block (memory_2.py:738-739 runs them inside DINOv2's forward) folded in — no
 * LayerNorm pass over the residual stream at all:
 *   a_mode 0 f32 rows / 1 pieces (as a_pieces above) / 2 f32 rows of the RESIDUAL STREAM, normalised ((x - mean) * rstd) while they
 *          are split for the matrix cores; the LayerNorm's gamma is expected folded into the weight columns and W beta into the
 *          bias by the caller (once per matrix); mean / rstd come from ln_stats_dev, the row means are left in ln_mu_dev;
 *          K = LayerNorm width (multiple of 128, <= 1024), piece output (epilogues 0, 1)
 *   epilogue 2 with ln_stats_dev != NULL: the finished rows of the residual stream leave their statistics in ln_stats_dev for
 *          the next a_mode-2 GEMM (shift = ln_mu_dev[row], the previous mean); N = LayerNorm width, a_mode 1
 *   ln_stats_dev (M, 20) f32, one record per row: [0] shift s, [2 + 2p] sum (x - s), [3 + 2p] sum (x - s)^2 over the columns
 *          [128 p, 128 p + 128); ln_mu_dev (M) f32.  bsc_enc_embed_layernorm_f32 writes the first records. */
bsc_status bsc_enc_gemm_split_ln(const void *a_dev, int64_t M, int32_t K, const void *pieces_dev, int32_t N, const float *bias_dev,
                                 const float *resid_dev, void *c_dev, float a_scale, float out_scale, int32_t epilogue,
                                 int32_t a_mode, float c_pieces_scale, float *ln_stats_dev, float *ln_mu_dev, float ln_eps,
                                 void *hip_stream);
/* The same with a workspace for the FEW-ROWS case (a frame or a handful per call, M <= ~3 500 rows — the reference's online use:
 * one DINOv2 forward per simulator step, memory_2.py:732-742): the GEMM then runs on 32- or 128-row tiles and, while those do not
 * fill the chip, splits K over the grid — every slice writes an f32 partial result into ws_dev, a second kernel adds the partials
 * in slice order (deterministic) and applies the epilogue.  ws_bytes >= 34 MB is always enough; NULL / too small: fewer slices.
 * bsc_enc_gemm_split_ln is this call without a workspace. */
bsc_status bsc_enc_gemm_split_ws(const void *a_dev, int64_t M, int32_t K, const void *pieces_dev, int32_t N, const float *bias_dev,
                                 const float *resid_dev, void *c_dev, float a_scale, float out_scale, int32_t epilogue,
                                 int32_t a_mode, float c_pieces_scale, float *ln_stats_dev, float *ln_mu_dev, float ln_eps,
                                 void *ws_dev, int64_t ws_bytes, void *hip_stream);
/* Token assembly of the f32 forward, one pass: row (b, 0) = cls + pos[0], rows (b, 1..registers) = the register tokens, rows
 * (b, 1 + registers + j) = patch[b][j] + pos[1 + j] -> u_dev (B T, width) f32; pieces_dev != NULL: LayerNorm(u) (gamma, beta) as
 * operand pieces; ln_stats_dev != NULL: the rows' statistics records + means (exact two-pass) for bsc_enc_gemm_split_ln.
 * bsc_enc_final_layernorm_f32: the last LayerNorm over the patch rows only (the first `skip` rows of every image dropped) ->
 * out_dev (B, T - skip, width) f32 = x_norm_patchtokens (memory_2.py:739).  width 256 / 512 / 768 / 1024. */
bsc_status bsc_enc_embed_layernorm_f32(const float *patch_dev, const float *cls_dev, const float *reg_dev, const float *pos_dev,
                                       const float *gamma_dev, const float *beta_dev, int32_t B, int32_t T, int32_t registers,
                                       int32_t width, float eps, float *u_dev, void *pieces_dev, float *ln_stats_dev,
                                       float *ln_mu_dev, void *hip_stream);
bsc_status bsc_enc_final_layernorm_f32(const float *u_dev, const float *gamma_dev, const float *beta_dev, int32_t B, int32_t T,
                                       int32_t skip, int32_t width, float eps, float *out_dev, void *hip_stream);

/* Encoder helper: u8 frames (B,H,W,C>=3) -> /255 -> antialiased bilinear resize to (S,S) -> (x-mean)/std ->
 * bf16 patch matrix (B, (S/patch)^2, 3*patch*patch), ready for the patch-embedding GEMM
 * (memory_2.py:733-736 and transform_, :71-74, fused into one pass). */
bsc_status bsc_enc_preprocess_patches(const void *rgb_dev, int32_t B, int32_t H, int32_t W, int32_t C, int32_t S,
                                      int32_t patch, void *out_dev, const float *mean3_host, const float *std3_host,
                                      void *hip_stream);
/* the same pass with a choice of output: out_mode 0 bf16 (as above), 1 f32, 2 fp16 pieces for the split-operand patch-embedding
 * GEMM — the reference-precision encoder starts from these; piece rows are padded with zeros to K' = 3 patch^2 rounded up to a
 * multiple of 32 (patch 14: 588 -> 608): out_dev is (B, (S/patch)^2, 2 K') fp16 */
bsc_status bsc_enc_preprocess_patches_typed(const void *rgb_dev, int32_t B, int32_t H, int32_t W, int32_t C, int32_t S,
                                            int32_t patch, void *out_dev, int32_t out_mode, const float *mean3_host,
                                            const float *std3_host, void *hip_stream);

/* HIP-event timing of the stages of the path, recorded around every launch on the stream the stage runs on.
 * which: 0 dense feature reduce (k_dense_reduce), 1 cosine scan of bsc_localize, 2 k_points (geometry + claims),
 *        3 (cell, frame, patch) pair aggregation, 4 voxel ids + per-voxel point order (runs, sort, expansion),
 *        5 pair sort + segment heads, 6 the main-stream work of one whole bsc_ingest call, 7 rgb chain + top-down map
 *        (library side stream, overlaps the call's tail and the next call).
 * out[0]=ms summed over the covered launches, out[1]=launches covered (ring of 512), out[2]=algorithmic
 * bytes accumulated (localize; for ingest derive them from bsc_counters), out[3]=launches since reset. */
bsc_status bsc_kernel_stats(bsc_ctx *ctx, int32_t which, int32_t reset, double *out4_host);

/* Completes everything bsc_ingest has started or deferred: the rgb running mean and the top-down map of the last call
 * (memory_2.py:888-903) run on a library-owned side stream and are launched lazily — by the next bsc_ingest, by any
 * export / merge / reset, or here — so that their sequential tail overlaps the next call's front end rather than the
 * caller's encoder.  Returns when both library streams are idle. */
bsc_status bsc_sync(bsc_ctx *ctx);
/* GPU-side: `hip_stream` waits for the rgb chain kernels launched so far (not for a still-deferred one) — for callers that
 * schedule their own work (an encoder) around the library's side stream without blocking the host. */
bsc_status bsc_stream_wait_chain(bsc_ctx *ctx, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
