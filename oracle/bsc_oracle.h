/*
 * bsc_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the BSC-Nav spatial-memory hot path
 * (reference: memory_2.py / utils.py of Heathcliff-saku/BSC-Nav).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (bsc-nav_amd/) never links, imports or calls it.
 *
 * Parity pin: every function below is checked against golden vectors that
 * were produced by executing the reference's own Python on seeded inputs
 * (tests/golden/gen_golden.py -> tests/golden/ npz files; tests/test_oracle_golden.py).
 *
 * Numerics recipe (SURVEY.md §7 "hard parts"): NumPy's `@` on the 3x3 / 4x4
 * products of the reference is reproduced bit-for-bit by
 *     acc = 0; for k ascending: acc = fma(A[i][k], B[k][j], acc)
 * in IEEE double; everything else is plain double mul/add/div with C-style
 * truncation for int().
 */
#ifndef BSC_ORACLE_H
#define BSC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_config {
    int32_t height, width;      /* frame H, W                                   */
    int32_t grid_size;          /* gs  (args.py:58)                             */
    int32_t min_h, max_h;       /* int(floor_height/cs), int(map_height/cs)     */
    int32_t patch_grid;         /* g: tokens per side of the ViT patch grid     */
    int32_t token_dim;          /* D                                            */
    int32_t iter_size;          /* token-cache rows, 50000 (memory_2.py:109)    */
    int32_t cache_size;         /* tokens per voxel, 10 (memory_2.py:111)       */
    int32_t mode;               /* 0 exact token-cache, 1 dense mean, 2 dense max */
    double cell_size, min_depth, max_depth;
    double K[9], Kinv[9], Kpatch[9];   /* row-major; computed by the host with NumPy */
} orc_config;

typedef struct orc_mem orc_mem;

/* draws one replacement index in [0, n) — memory_2.py:352 random.choice(range(n)) */
typedef uint32_t (*orc_draw_fn)(void *user, uint32_t n);

/* ---- stateless geometry (utils.py:153-214, memory_2.py:864-875) ------------------ */
/* For P pixel indices (row-major i = y*W + x) of one frame.  Outputs may be NULL. */
void orc_geometry(const orc_config *cfg, const float *depth, const int32_t *idx, int64_t P,
                  const double *T /*4x4 pc_transform*/,
                  uint8_t *valid,      /* depth mask  min<z<max (utils.py:175-177)          */
                  double *pc,          /* (P,3) camera-frame point                           */
                  double *pg,          /* (P,3) map-frame point                              */
                  int32_t *vox,        /* (P,3) row,col,h BEFORE the -minh shift             */
                  uint8_t *in_range,   /* !_out_of_range (memory_2.py:755)                   */
                  int32_t *pix,        /* (P,2) x,y recovered source pixel                   */
                  int32_t *pat,        /* (P,2) x,y patch coordinates                        */
                  double *r2, double *alpha);

/* ---- stateful memory -------------------------------------------------------------- */
orc_mem *orc_create(const orc_config *cfg, int64_t voxel_capacity);
void orc_destroy(orc_mem *m);

/* One frame through obs2voxeltoken's per-point loop (memory_2.py:863-903).
 * idx: sampled pixel indices in reference (shuffled) order, already strided by the
 * sample rate but NOT yet depth-masked (memory_2.py:747-752 does the mask after).
 * idx == NULL means all H*W pixels in row-major order.
 * alpha_override: optional per-point alpha (same indexing as idx) replacing exp().
 * tokens: (g,g,D) f32 of this frame.  rgb: (H,W,rgb_stride) u8.
 * In exact mode in-loop flushes call draw().  Returns number of points that passed. */
int64_t orc_ingest_frame(orc_mem *m, const float *depth, const uint8_t *rgb, int32_t rgb_stride,
                         const int32_t *idx, int64_t P, const double *T, const float *tokens,
                         const double *alpha_override, orc_draw_fn draw, void *user);

/* update_memory_dist_base (memory_2.py:326-358): ALL iter_size rows, zero rows included */
void orc_flush(orc_mem *m, orc_draw_fn draw, void *user);

/* counters: out[0]=max_id out[1]=iter_id out[2]=#store voxels out[3]=#store tokens out[4]=#flushes */
void orc_counters(const orc_mem *m, int64_t *out);

void orc_export_rgb(const orc_mem *m, int32_t *pos /*(max_id,3)*/, uint8_t *rgb /*(max_id,3)*/, float *weight);
void orc_export_occupied(const orc_mem *m, int32_t *occ /*(gs,gs,nh)*/);
void orc_export_heightmap(const orc_mem *m, double *max_height /*(gs,gs)*/, uint8_t *cv_map /*(gs,gs,3)*/);
void orc_export_cache(const orc_mem *m, float *feat, int32_t *pos, float *dis); /* first iter_id rows */
/* feature store in HDF5 name order: pos (V,3), cnt (V), feats (T,D), dists (T) */
void orc_export_store(const orc_mem *m, int32_t *pos, int32_t *cnt, float *feats, float *dists);
/* dense modes: per voxel id (first-touch order) sum-or-max (V,D) and count (V) */
void orc_export_dense(const orc_mem *m, float *acc, int32_t *cnt);

/* query pooling (memory_2.py:591-608): tokens (B,T,D) -> out (D) */
void orc_pool_query(const float *tokens, int32_t B, int32_t T, int32_t D, float *out);

/* voxel_localized scan (memory_2.py:623-671).  radius<0: no region filter; floor_lo>floor_hi: no
 * floor filter.  Returns number of results written (<=K). */
int32_t orc_localize(const orc_mem *m, const float *q, int32_t K, double radius, const int32_t *curr,
                     int32_t floor_lo, int32_t floor_hi, int32_t *out_pos /*(K,3)*/, float *out_sim);

/* GESObjectNavRobot.weighted_cluster_centers (BSCAgent.py:479-497): DBSCAN(eps, min_samples) over the top-K voxel
 * positions (scikit-learn's algorithm restated: brute-force eps-neighbourhoods incl. the point itself, clusters
 * grown depth-first from unlabelled core points in index order, border points keep the first cluster that reaches
 * them), similarity-weighted centres, clusters ordered by mean similarity (stable, descending).
 * centers (K,3) f64, labels (K) i32 (-1 noise), sizes (K) i32; returns the number of clusters. */
int32_t orc_cluster_centers(const int32_t *pos, const double *sim, int32_t K, double eps, int32_t min_samples,
                            double *centers, int32_t *labels, int32_t *sizes);

/* FrontierExplorer helpers (memory_2.py:1165-1207): mask[x*gs+y] bit0 = known (cv_map[x,y].sum() != 0, :1165),
 * bit1 = frontier: known, navigable[x,y] != 0 (NULL: every cell) and at least one in-bounds 4-neighbour unknown. */
void orc_frontier_mask(const uint8_t *cv_map /*(gs,gs,3)*/, const uint8_t *navigable /*(gs,gs) or NULL*/, int32_t gs,
                       uint8_t *mask /*(gs,gs)*/);
/* cluster_frontiers + compute_cluster_center + compute_information_gain + select_best_cluster_center_by_ig
 * (memory_2.py:1209-1311): 4-connected clusters of the cells with frontier[x*gs+y] != 0, grown breadth-first from
 * the cells in row-major order, clusters smaller than min_cluster_size dropped; centre = mean cell; gain = unknown
 * cells of cv_map inside the in-bounds part of the (2r+1)^2 window around the centre rounded half-to-even (Python
 * round); best = first cluster with the strictly largest gain > 0 (-1: none).  labels (gs,gs): cluster ordinal or -1;
 * first (n,2): the cluster's first cell; returns the number of clusters kept (at most max_clusters are written). */
int32_t orc_frontier_clusters(const uint8_t *cv_map, const uint8_t *frontier, int32_t gs, int32_t min_cluster_size,
                              int32_t ig_radius, int32_t max_clusters, int32_t *labels, int32_t *first, int32_t *sizes,
                              double *centers /*(n,2)*/, double *gains, int32_t *best);

/* name-order key of "grid_r_c_h" (bytewise string order) */
uint64_t orc_name_key(int32_t r, int32_t c, int32_t h);

#ifdef __cplusplus
}
#endif
#endif
