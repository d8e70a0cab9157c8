/*
 * bsc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see bsc_oracle.h).
 *
 * Sequential plain-C restatement of the reference hot path.  Each function
 * cites the reference lines it follows (paths relative to the reference
 * checkout).  Build: make -C oracle   (gcc -O2 -ffp-contract=off).
 */
#include "bsc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small helpers                                                             */
/* ------------------------------------------------------------------------- */
static inline double dot3_fma(const double *row, double a, double b, double c)
{
    /* NumPy/OpenBLAS 3x3 @ 3xN product: acc=0; acc=fma(A[i][k],B[k][j],acc), k ascending */
    double acc = 0.0;
    acc = fma(row[0], a, acc);
    acc = fma(row[1], b, acc);
    acc = fma(row[2], c, acc);
    return acc;
}

static inline double dot4_fma(const double *row, double a, double b, double c, double d)
{
    double acc = 0.0;
    acc = fma(row[0], a, acc);
    acc = fma(row[1], b, acc);
    acc = fma(row[2], c, acc);
    acc = fma(row[3], d, acc);
    return acc;
}

typedef struct {
    int valid, in_range;
    double pc[3], pg[3];
    int32_t vox[3];     /* row, col, h (before -minh) */
    int32_t pix[2], pat[2];
    double r2, alpha;
} geom_t;

static void geom_point(const orc_config *c, const float *depth, int32_t i, const double *T, geom_t *o)
{
    const int W = c->width;
    const int y = i / W, x = i - y * W;
    /* utils.py:166-172  pixel centres, homogeneous; pc = Kinv @ p2d ; pc *= z */
    const double px = (double)x + 0.5, py = (double)y + 0.5;
    const double z = (double)depth[i];
    double p0 = dot3_fma(c->Kinv + 0, px, py, 1.0) * z;
    double p1 = dot3_fma(c->Kinv + 3, px, py, 1.0) * z;
    double p2 = dot3_fma(c->Kinv + 6, px, py, 1.0) * z;
    o->pc[0] = p0; o->pc[1] = p1; o->pc[2] = p2;
    /* utils.py:175-177  strict depth mask on pc[2] */
    o->valid = (p2 > c->min_depth) && (p2 < c->max_depth);
    o->in_range = 0;
    if (!o->valid) return;
    /* utils.py:189-199 transform_pc: (T @ [pc;1])[:3] */
    double g0 = dot4_fma(T + 0, p0, p1, p2, 1.0);
    double g1 = dot4_fma(T + 4, p0, p1, p2, 1.0);
    double g2 = dot4_fma(T + 8, p0, p1, p2, 1.0);
    o->pg[0] = g0; o->pg[1] = g1; o->pg[2] = g2;
    /* utils.py:201-205 base_pos2grid_id_3d: double truncation toward zero */
    const double half = (double)c->grid_size / 2.0;
    int32_t row = (int32_t)(half - (double)(int32_t)(g0 / c->cell_size));
    int32_t col = (int32_t)(half - (double)(int32_t)(g1 / c->cell_size));
    int32_t h = (int32_t)(g2 / c->cell_size);
    o->vox[0] = row; o->vox[1] = col; o->vox[2] = h;
    /* memory_2.py:755-756 _out_of_range */
    o->in_range = !(col >= c->grid_size || row >= c->grid_size || h >= c->max_h || col < 0 || row < 0 || h < c->min_h);
    /* utils.py:208-214 project_point with calib_mat, then with the patch intrinsics */
    {
        double q0 = dot3_fma(c->K + 0, p0, p1, p2);
        double q1 = dot3_fma(c->K + 3, p0, p1, p2);
        double q2 = dot3_fma(c->K + 6, p0, p1, p2);
        o->pix[0] = (int32_t)(q0 / q2 - 0.5);
        o->pix[1] = (int32_t)(q1 / q2 - 0.5);
    }
    {
        double q0 = dot3_fma(c->Kpatch + 0, p0, p1, p2);
        double q1 = dot3_fma(c->Kpatch + 3, p0, p1, p2);
        double q2 = dot3_fma(c->Kpatch + 6, p0, p1, p2);
        o->pat[0] = (int32_t)(q0 / q2 - 0.5);
        o->pat[1] = (int32_t)(q1 / q2 - 0.5);
    }
    /* memory_2.py:873-875 */
    o->r2 = (p0 * p0 + p1 * p1) + p2 * p2;
    o->alpha = exp(-o->r2 / (2 * 0.6));
}

void orc_geometry(const orc_config *cfg, const float *depth, const int32_t *idx, int64_t P, const double *T,
                  uint8_t *valid, double *pc, double *pg, int32_t *vox, uint8_t *in_range, int32_t *pix,
                  int32_t *pat, double *r2, double *alpha)
{
    for (int64_t j = 0; j < P; ++j) {
        geom_t g;
        memset(&g, 0, sizeof g);
        geom_point(cfg, depth, idx ? idx[j] : (int32_t)j, T, &g);
        if (valid) valid[j] = (uint8_t)g.valid;
        if (pc) memcpy(pc + 3 * j, g.pc, sizeof g.pc);
        if (pg) memcpy(pg + 3 * j, g.pg, sizeof g.pg);
        if (vox) memcpy(vox + 3 * j, g.vox, sizeof g.vox);
        if (in_range) in_range[j] = (uint8_t)g.in_range;
        if (pix) memcpy(pix + 2 * j, g.pix, sizeof g.pix);
        if (pat) memcpy(pat + 2 * j, g.pat, sizeof g.pat);
        if (r2) r2[j] = g.r2;
        if (alpha) alpha[j] = g.alpha;
    }
}

/* ------------------------------------------------------------------------- */
/* name order of HDF5 group "grid_{r}_{c}_{h}"                                */
/* ------------------------------------------------------------------------- */
#define NAME_DIGITS 6
static uint64_t enc_field(int32_t v, int last)
{
    /* decimal digits left-aligned in NAME_DIGITS symbols, base 11.
     * non-final fields are followed by '_' (0x5f) which sorts AFTER every digit -> pad symbol 10, digits 0..9
     * the final field is followed by end-of-string which sorts BEFORE every digit -> pad 0, digits 1..10 */
    char buf[16];
    int n = 0;
    if (v == 0) buf[n++] = 0;
    while (v > 0) { buf[n++] = (char)(v % 10); v /= 10; }
    uint64_t k = 0;
    for (int i = 0; i < NAME_DIGITS; ++i) {
        int sym;
        if (i < n) sym = buf[n - 1 - i] + (last ? 1 : 0);
        else sym = last ? 0 : 10;
        k = k * 11 + (uint64_t)sym;
    }
    return k;
}

uint64_t orc_name_key(int32_t r, int32_t c, int32_t h)
{
    const uint64_t B = 1771561ull; /* 11^6 */
    return (enc_field(r, 0) * B + enc_field(c, 0)) * B + enc_field(h, 1);
}

/* ------------------------------------------------------------------------- */
/* memory object                                                             */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t pos[3];
    int32_t cnt;
    int64_t *rows; /* cache_size pool row ids */
} store_ent;

struct orc_mem {
    orc_config c;
    int32_t nh;
    int64_t vcap;
    /* rgb voxel state (memory_2.py:708-722) */
    int32_t *occ;
    int32_t *rgb_pos;
    uint8_t *rgb;
    float *weight;
    int64_t max_id;
    uint8_t *cv_map;
    double *max_height;
    /* token cache */
    float *cf;
    int32_t *cp;
    float *cd;
    int64_t iter_id;
    /* feature store (stands in for feat.h5df) */
    store_ent *ents;
    int64_t n_ents, cap_ents;
    int64_t *htab;
    int64_t hcap;
    float *pool;
    float *pool_d;
    int64_t n_rows, cap_rows;
    int64_t n_flush;
    /* dense modes */
    float *acc;
    int32_t *acnt;
};

static uint64_t pack_key(int32_t r, int32_t c, int32_t h)
{
    return ((uint64_t)(uint32_t)r << 42) | ((uint64_t)(uint32_t)c << 21) | (uint64_t)(uint32_t)h;
}
static uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

orc_mem *orc_create(const orc_config *cfg, int64_t voxel_capacity)
{
    orc_mem *m = (orc_mem *)calloc(1, sizeof *m);
    m->c = *cfg;
    m->nh = cfg->max_h - cfg->min_h;
    m->vcap = voxel_capacity;
    const int64_t gs = cfg->grid_size, ncell = gs * gs * m->nh;
    m->occ = (int32_t *)malloc(sizeof(int32_t) * ncell);
    for (int64_t i = 0; i < ncell; ++i) m->occ[i] = -1;
    m->rgb_pos = (int32_t *)calloc(voxel_capacity * 3, sizeof(int32_t));
    m->rgb = (uint8_t *)calloc(voxel_capacity * 3, 1);
    m->weight = (float *)calloc(voxel_capacity, sizeof(float));
    m->cv_map = (uint8_t *)calloc(gs * gs * 3, 1);
    m->max_height = (double *)malloc(sizeof(double) * gs * gs);
    for (int64_t i = 0; i < gs * gs; ++i) m->max_height[i] = -INFINITY;
    if (cfg->mode == 0) {
        m->cf = (float *)calloc((size_t)cfg->iter_size * cfg->token_dim, sizeof(float));
        m->cp = (int32_t *)calloc((size_t)cfg->iter_size * 3, sizeof(int32_t));
        m->cd = (float *)calloc((size_t)cfg->iter_size, sizeof(float));
        m->hcap = 1 << 12;
        m->htab = (int64_t *)malloc(sizeof(int64_t) * m->hcap);
        for (int64_t i = 0; i < m->hcap; ++i) m->htab[i] = -1;
    } else {
        m->acc = (float *)calloc((size_t)voxel_capacity * cfg->token_dim, sizeof(float));
        m->acnt = (int32_t *)calloc((size_t)voxel_capacity, sizeof(int32_t));
    }
    return m;
}

void orc_destroy(orc_mem *m)
{
    if (!m) return;
    for (int64_t i = 0; i < m->n_ents; ++i) free(m->ents[i].rows);
    free(m->occ); free(m->rgb_pos); free(m->rgb); free(m->weight); free(m->cv_map); free(m->max_height);
    free(m->cf); free(m->cp); free(m->cd); free(m->ents); free(m->htab); free(m->pool); free(m->pool_d);
    free(m->acc); free(m->acnt);
    free(m);
}

static int64_t store_find(const orc_mem *m, const int32_t *pos)
{
    uint64_t k = pack_key(pos[0], pos[1], pos[2]);
    uint64_t h = mix64(k) & (uint64_t)(m->hcap - 1);
    for (;;) {
        int64_t e = m->htab[h];
        if (e < 0) return -1;
        if (m->ents[e].pos[0] == pos[0] && m->ents[e].pos[1] == pos[1] && m->ents[e].pos[2] == pos[2]) return e;
        h = (h + 1) & (uint64_t)(m->hcap - 1);
    }
}

static void store_rehash(orc_mem *m)
{
    free(m->htab);
    m->hcap *= 2;
    m->htab = (int64_t *)malloc(sizeof(int64_t) * m->hcap);
    for (int64_t i = 0; i < m->hcap; ++i) m->htab[i] = -1;
    for (int64_t e = 0; e < m->n_ents; ++e) {
        uint64_t h = mix64(pack_key(m->ents[e].pos[0], m->ents[e].pos[1], m->ents[e].pos[2])) & (uint64_t)(m->hcap - 1);
        while (m->htab[h] >= 0) h = (h + 1) & (uint64_t)(m->hcap - 1);
        m->htab[h] = e;
    }
}

static int64_t store_create(orc_mem *m, const int32_t *pos)
{
    if (m->n_ents == m->cap_ents) {
        m->cap_ents = m->cap_ents ? m->cap_ents * 2 : 1024;
        m->ents = (store_ent *)realloc(m->ents, sizeof(store_ent) * m->cap_ents);
    }
    if ((m->n_ents + 1) * 2 > m->hcap) store_rehash(m);
    int64_t e = m->n_ents++;
    memcpy(m->ents[e].pos, pos, sizeof(int32_t) * 3);
    m->ents[e].cnt = 0;
    m->ents[e].rows = (int64_t *)malloc(sizeof(int64_t) * m->c.cache_size);
    uint64_t h = mix64(pack_key(pos[0], pos[1], pos[2])) & (uint64_t)(m->hcap - 1);
    while (m->htab[h] >= 0) h = (h + 1) & (uint64_t)(m->hcap - 1);
    m->htab[h] = e;
    return e;
}

static int64_t pool_new_row(orc_mem *m)
{
    if (m->n_rows == m->cap_rows) {
        m->cap_rows = m->cap_rows ? m->cap_rows * 2 : 4096;
        m->pool = (float *)realloc(m->pool, sizeof(float) * m->cap_rows * m->c.token_dim);
        m->pool_d = (float *)realloc(m->pool_d, sizeof(float) * m->cap_rows);
    }
    return m->n_rows++;
}

/* memory_2.py:326-358 update_memory_dist_base — loops over ALL iter_size cache rows (line 331),
 * unused rows are zero vectors at position [0,0,0]; then _reinit_cache (line 358). */
void orc_flush(orc_mem *m, orc_draw_fn draw, void *user)
{
    const int D = m->c.token_dim;
    for (int64_t i = 0; i < m->c.iter_size; ++i) {
        const int32_t *pos = m->cp + 3 * i;
        int64_t e = store_find(m, pos);
        int64_t row;
        if (e < 0) {                                       /* :335-338 create group with one token */
            e = store_create(m, pos);
            row = pool_new_row(m);
            m->ents[e].rows[m->ents[e].cnt++] = row;
        } else if (m->ents[e].cnt < m->c.cache_size) {     /* :345-349 append */
            row = pool_new_row(m);
            m->ents[e].rows[m->ents[e].cnt++] = row;
        } else {                                           /* :351-354 random replacement */
            uint32_t k = draw(user, (uint32_t)m->ents[e].cnt);
            row = m->ents[e].rows[k];
        }
        memcpy(m->pool + row * D, m->cf + i * D, sizeof(float) * D);
        m->pool_d[row] = m->cd[i];
    }
    memset(m->cf, 0, sizeof(float) * (size_t)m->c.iter_size * D);
    memset(m->cp, 0, sizeof(int32_t) * (size_t)m->c.iter_size * 3);
    memset(m->cd, 0, sizeof(float) * (size_t)m->c.iter_size);
    m->iter_id = 0;
    m->n_flush++;
}

/* memory_2.py:863-903 */
int64_t orc_ingest_frame(orc_mem *m, const float *depth, const uint8_t *rgb, int32_t rgb_stride, const int32_t *idx,
                         int64_t P, const double *T, const float *tokens, const double *alpha_override,
                         orc_draw_fn draw, void *user)
{
    const orc_config *c = &m->c;
    const int D = c->token_dim, g = c->patch_grid, W = c->width, H = c->height;
    const int64_t gs = c->grid_size;
    int64_t passed = 0;
    for (int64_t j = 0; j < P; ++j) {
        geom_t q;
        geom_point(c, depth, idx ? idx[j] : (int32_t)j, T, &q);
        if (!q.valid) continue;            /* memory_2.py:750-752: masked before the loop */
        if (!q.in_range) continue;         /* :865-866 */
        const int32_t row = q.vox[0], col = q.vox[1], h = q.vox[2] - c->min_h; /* :867 */
        /* :869-870 rgb_v = rgb[py, px] — NumPy negative indices wrap */
        int32_t sx = q.pix[0], sy = q.pix[1];
        if (sx < 0) sx += W;
        if (sy < 0) sy += H;
        const uint8_t *rgb_v = rgb + ((int64_t)sy * W + sx) * rgb_stride;
        const int32_t px = q.pat[0], py = q.pat[1];
        const double alpha = alpha_override ? alpha_override[j] : q.alpha;
        if (px < 0 || py < 0 || px >= g || py >= g) continue;   /* :878 */
        ++passed;
        const float *tok = tokens + ((int64_t)py * g + px) * D;
        if (c->mode == 0) {
            if (m->iter_id >= c->iter_size) {                    /* :880-881 flush; this token is dropped */
                orc_flush(m, draw, user);
            } else {                                             /* :882-886 */
                memcpy(m->cf + m->iter_id * D, tok, sizeof(float) * D);
                m->cp[3 * m->iter_id + 0] = row;
                m->cp[3 * m->iter_id + 1] = col;
                m->cp[3 * m->iter_id + 2] = h;
                m->cd[m->iter_id] = (float)q.r2;
                m->iter_id++;
            }
        }
        /* :888-899 rgb voxel: first-touch id or running weighted mean with truncating store */
        const int64_t cell = ((int64_t)row * gs + col) * m->nh + h;
        int32_t id = m->occ[cell];
        if (id == -1) {
            if (m->max_id >= m->vcap) abort();
            id = (int32_t)m->max_id;
            m->occ[cell] = id;
            m->rgb[3 * id + 0] = rgb_v[0]; m->rgb[3 * id + 1] = rgb_v[1]; m->rgb[3 * id + 2] = rgb_v[2];
            m->weight[id] = (float)((double)m->weight[id] + alpha);
            m->rgb_pos[3 * id + 0] = row; m->rgb_pos[3 * id + 1] = col; m->rgb_pos[3 * id + 2] = h;
            m->max_id++;
        } else {
            const float w = m->weight[id];
            const double den = (double)w + alpha;
            for (int k = 0; k < 3; ++k) {
                float a = (float)m->rgb[3 * id + k] * w;          /* u8 * f32 -> f32 */
                double b = (double)rgb_v[k] * alpha;              /* u8 * f64 -> f64 */
                double v = ((double)a + b) / den;
                m->rgb[3 * id + k] = (uint8_t)v;                  /* truncating store */
            }
            m->weight[id] = (float)den;
        }
        if (c->mode == 1) {
            float *a = m->acc + (int64_t)id * D;
            for (int k = 0; k < D; ++k) a[k] += tok[k];
            m->acnt[id]++;
        } else if (c->mode == 2) {
            float *a = m->acc + (int64_t)id * D;
            if (m->acnt[id] == 0) memcpy(a, tok, sizeof(float) * D);
            else for (int k = 0; k < D; ++k) a[k] = tok[k] > a[k] ? tok[k] : a[k];
            m->acnt[id]++;
        }
        /* :901-903 top-down map */
        if ((double)h >= m->max_height[row * gs + col]) {
            m->max_height[row * gs + col] = (double)h;
            memcpy(m->cv_map + (row * gs + col) * 3, rgb_v, 3);
        }
    }
    return passed;
}

void orc_counters(const orc_mem *m, int64_t *out)
{
    out[0] = m->max_id; out[1] = m->iter_id; out[2] = m->n_ents; out[3] = 0; out[4] = m->n_flush;
    for (int64_t e = 0; e < m->n_ents; ++e) out[3] += m->ents[e].cnt;
}

void orc_export_rgb(const orc_mem *m, int32_t *pos, uint8_t *rgb, float *weight)
{
    memcpy(pos, m->rgb_pos, sizeof(int32_t) * 3 * m->max_id);
    memcpy(rgb, m->rgb, 3 * m->max_id);
    memcpy(weight, m->weight, sizeof(float) * m->max_id);
}
void orc_export_occupied(const orc_mem *m, int32_t *occ)
{
    memcpy(occ, m->occ, sizeof(int32_t) * (int64_t)m->c.grid_size * m->c.grid_size * m->nh);
}
void orc_export_heightmap(const orc_mem *m, double *max_height, uint8_t *cv_map)
{
    const int64_t n = (int64_t)m->c.grid_size * m->c.grid_size;
    memcpy(max_height, m->max_height, sizeof(double) * n);
    memcpy(cv_map, m->cv_map, 3 * n);
}
void orc_export_cache(const orc_mem *m, float *feat, int32_t *pos, float *dis)
{
    memcpy(feat, m->cf, sizeof(float) * m->iter_id * m->c.token_dim);
    memcpy(pos, m->cp, sizeof(int32_t) * 3 * m->iter_id);
    memcpy(dis, m->cd, sizeof(float) * m->iter_id);
}

typedef struct { uint64_t key; int64_t e; } name_ord;
static int cmp_name(const void *a, const void *b)
{
    uint64_t x = ((const name_ord *)a)->key, y = ((const name_ord *)b)->key;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static name_ord *name_order(const orc_mem *m)
{
    name_ord *o = (name_ord *)malloc(sizeof(name_ord) * (m->n_ents ? m->n_ents : 1));
    for (int64_t e = 0; e < m->n_ents; ++e) {
        o[e].key = orc_name_key(m->ents[e].pos[0], m->ents[e].pos[1], m->ents[e].pos[2]);
        o[e].e = e;
    }
    qsort(o, m->n_ents, sizeof(name_ord), cmp_name);
    return o;
}

void orc_export_store(const orc_mem *m, int32_t *pos, int32_t *cnt, float *feats, float *dists)
{
    const int D = m->c.token_dim;
    name_ord *o = name_order(m);
    int64_t t = 0;
    for (int64_t i = 0; i < m->n_ents; ++i) {
        const store_ent *e = &m->ents[o[i].e];
        memcpy(pos + 3 * i, e->pos, sizeof(int32_t) * 3);
        cnt[i] = e->cnt;
        for (int k = 0; k < e->cnt; ++k, ++t) {
            memcpy(feats + t * D, m->pool + e->rows[k] * D, sizeof(float) * D);
            dists[t] = m->pool_d[e->rows[k]];
        }
    }
    free(o);
}

void orc_export_dense(const orc_mem *m, float *acc, int32_t *cnt)
{
    memcpy(acc, m->acc, sizeof(float) * m->max_id * m->c.token_dim);
    memcpy(cnt, m->acnt, sizeof(int32_t) * m->max_id);
}

/* memory_2.py:591-608: Gaussian centre-weighted pooling, mean over the batch */
void orc_pool_query(const float *tokens, int32_t B, int32_t T, int32_t D, float *out)
{
    const int g = (int)sqrt((double)T);
    float *w = (float *)malloc(sizeof(float) * T);
    const float center = (float)((g - 1) / 2.0);
    const float sigma = (float)((g / 2.0) * (g / 2.0));
    float wsum = 0.f;
    for (int t = 0; t < T; ++t) {
        float xs = (float)(t % g), ys = (float)(t / g);
        float d2 = (xs - center) * (xs - center) + (ys - center) * (ys - center);
        w[t] = expf(-d2 / (2 * sigma));
        wsum += w[t];
    }
    for (int t = 0; t < T; ++t) w[t] = w[t] / wsum;
    for (int d = 0; d < D; ++d) out[d] = 0.f;
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d) {
            float s = 0.f;
            for (int t = 0; t < T; ++t) s += tokens[((int64_t)b * T + t) * D + d] * w[t];
            out[d] += s;
        }
    for (int d = 0; d < D; ++d) out[d] /= (float)B;
    free(w);
}

/* torch F.cosine_similarity(x1 (1,D), x2 (M,D), dim=1, eps=1e-8): sum((x1/max(|x1|,eps)) * (x2/max(|x2|,eps))) */
static float cos_row(const float *qn, const float *x, int D)
{
    double n2 = 0.0;
    for (int k = 0; k < D; ++k) n2 += (double)x[k] * (double)x[k];
    float nx = (float)sqrt(n2);
    if (nx < 1e-8f) nx = 1e-8f;
    double s = 0.0;
    for (int k = 0; k < D; ++k) s += (double)(qn[k] * (x[k] / nx));
    return (float)s;
}

typedef struct { float sim; int64_t ord; int32_t pos[3]; } cand_t;
static int cmp_cand(const void *a, const void *b)
{
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->sim > y->sim) return -1;
    if (x->sim < y->sim) return 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord ? 1 : 0); /* list.sort is stable (memory_2.py:665) */
}

int32_t orc_localize(const orc_mem *m, const float *q, int32_t K, double radius, const int32_t *curr, int32_t floor_lo,
                     int32_t floor_hi, int32_t *out_pos, float *out_sim)
{
    const int D = m->c.token_dim;
    float *qn = (float *)malloc(sizeof(float) * D);
    double n2 = 0.0;
    for (int k = 0; k < D; ++k) n2 += (double)q[k] * (double)q[k];
    float nq = (float)sqrt(n2);
    if (nq < 1e-8f) nq = 1e-8f;
    for (int k = 0; k < D; ++k) qn[k] = q[k] / nq;
    int64_t nv = (m->c.mode == 0) ? m->n_ents : m->max_id;
    cand_t *cands = (cand_t *)malloc(sizeof(cand_t) * (nv ? nv : 1));
    int64_t nc = 0;
    name_ord *o = NULL;
    if (m->c.mode == 0) o = name_order(m);
    else {
        o = (name_ord *)malloc(sizeof(name_ord) * (nv ? nv : 1));
        for (int64_t e = 0; e < nv; ++e) {
            o[e].key = orc_name_key(m->rgb_pos[3 * e], m->rgb_pos[3 * e + 1], m->rgb_pos[3 * e + 2]);
            o[e].e = e;
        }
        qsort(o, nv, sizeof(name_ord), cmp_name);
    }
    for (int64_t i = 0; i < nv; ++i) {
        const int64_t e = o[i].e;
        const int32_t *pos = (m->c.mode == 0) ? m->ents[e].pos : (m->rgb_pos + 3 * e);
        if (radius >= 0) {                                       /* memory_2.py:624-629 */
            double dx = pos[0] - curr[0], dy = pos[1] - curr[1], dz = pos[2] - curr[2];
            if (dx * dx + dy * dy + dz * dz > radius * radius) continue;
        }
        if (floor_lo <= floor_hi) {                              /* :633-640 */
            if (!(floor_lo <= pos[2] && pos[2] <= floor_hi)) continue;
        }
        float best = -INFINITY;
        if (m->c.mode == 0) {
            for (int k = 0; k < m->ents[e].cnt; ++k) {           /* :656-662 per-voxel max */
                float s = cos_row(qn, m->pool + m->ents[e].rows[k] * D, D);
                if (s > best) best = s;
            }
        } else {
            best = cos_row(qn, m->acc + e * D, D);
        }
        cands[nc].sim = best; cands[nc].ord = i;
        memcpy(cands[nc].pos, pos, sizeof(int32_t) * 3);
        ++nc;
    }
    qsort(cands, nc, sizeof(cand_t), cmp_cand);
    int32_t n = (int32_t)(nc < K ? nc : K);
    for (int32_t i = 0; i < n; ++i) {
        memcpy(out_pos + 3 * i, cands[i].pos, sizeof(int32_t) * 3);
        out_sim[i] = cands[i].sim;
    }
    free(cands); free(o); free(qn);
    return n;
}

/* ------------------------------------------------------------------------- */
/* BSCAgent.py:479-497 weighted_cluster_centers                               */
/* ------------------------------------------------------------------------- */
int32_t orc_cluster_centers(const int32_t *pos, const double *sim, int32_t K, double eps, int32_t min_samples,
                            double *centers, int32_t *labels, int32_t *sizes)
{
    uint8_t *nb = (uint8_t *)calloc((size_t)K * K, 1);
    uint8_t *core = (uint8_t *)calloc(K, 1);
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)K * K + 16);
    for (int i = 0; i < K; ++i) {
        int cnt = 0;
        for (int j = 0; j < K; ++j) {
            double dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
            if (sqrt(dx * dx + dy * dy + dz * dz) <= eps) { nb[(size_t)i * K + j] = 1; ++cnt; }
        }
        core[i] = cnt >= min_samples;           /* the point itself counts (sklearn DBSCAN) */
        labels[i] = -1;
    }
    int32_t label_num = 0;
    for (int s0 = 0; s0 < K; ++s0) {            /* sklearn.cluster._dbscan_inner.dbscan_inner */
        if (labels[s0] != -1 || !core[s0]) continue;
        int64_t sp = 0;
        int i = s0;
        for (;;) {
            if (labels[i] == -1) {
                labels[i] = label_num;
                if (core[i])
                    for (int j = 0; j < K; ++j)
                        if (nb[(size_t)i * K + j] && labels[j] == -1) stack[sp++] = j;
            }
            if (sp == 0) break;
            i = stack[--sp];
        }
        ++label_num;
    }
    /* BSCAgent.py:484-491: per cluster np.average(points, weights=sim), np.mean(sim), size */
    double *avg = (double *)malloc(sizeof(double) * (label_num ? label_num : 1));
    double *ctr = (double *)malloc(sizeof(double) * 3 * (label_num ? label_num : 1));
    int32_t *sz = (int32_t *)malloc(sizeof(int32_t) * (label_num ? label_num : 1));
    for (int l = 0; l < label_num; ++l) {
        double sw = 0.0, sx = 0.0, sy = 0.0, szz = 0.0;
        int n = 0;
        for (int i = 0; i < K; ++i)
            if (labels[i] == l) {
                sw += sim[i];
                sx += pos[3 * i] * sim[i]; sy += pos[3 * i + 1] * sim[i]; szz += pos[3 * i + 2] * sim[i];
                ++n;
            }
        avg[l] = sw / n; sz[l] = n;
        ctr[3 * l] = sx / sw; ctr[3 * l + 1] = sy / sw; ctr[3 * l + 2] = szz / sw;
    }
    /* :493 stable sort by mean similarity, descending */
    for (int l = 0; l < label_num; ++l) {
        int rank = 0;
        for (int m = 0; m < label_num; ++m)
            if (avg[m] > avg[l] || (avg[m] == avg[l] && m < l)) ++rank;
        memcpy(centers + 3 * rank, ctr + 3 * l, sizeof(double) * 3);
        sizes[rank] = sz[l];
    }
    free(nb); free(core); free(stack); free(avg); free(ctr); free(sz);
    return label_num;
}

/* ---- FrontierExplorer helpers (memory_2.py:1165-1311) -------------------------------------------------------- */
static int cell_unknown(const uint8_t *cv, int32_t gs, int32_t x, int32_t y)
{
    const uint8_t *p = cv + 3 * ((int64_t)x * gs + y);
    return (int)p[0] + (int)p[1] + (int)p[2] == 0;            /* :1165 cv_map[x, y].sum() == 0 */
}

void orc_frontier_mask(const uint8_t *cv_map, const uint8_t *navigable, int32_t gs, uint8_t *mask)
{
    static const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};   /* :1191 */
    for (int32_t x = 0; x < gs; ++x)
        for (int32_t y = 0; y < gs; ++y) {
            const int64_t i = (int64_t)x * gs + y;
            const int known = !cell_unknown(cv_map, gs, x, y);
            uint8_t m = known ? 1 : 0;
            if (known && (!navigable || navigable[i])) {
                for (int d = 0; d < 4; ++d) {
                    const int32_t nx = x + dx[d], ny = y + dy[d];
                    if (nx >= 0 && nx < gs && ny >= 0 && ny < gs && cell_unknown(cv_map, gs, nx, ny)) { m |= 2; break; }
                }
            }
            mask[i] = m;
        }
}

int32_t orc_frontier_clusters(const uint8_t *cv_map, const uint8_t *frontier, int32_t gs, int32_t min_cluster_size,
                              int32_t ig_radius, int32_t max_clusters, int32_t *labels, int32_t *first, int32_t *sizes,
                              double *centers, double *gains, int32_t *best)
{
    static const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
    const int64_t n = (int64_t)gs * gs;
    uint8_t *visited = (uint8_t *)calloc((size_t)n, 1);
    int64_t *queue = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int32_t kept = 0;
    double best_ig = 0.0;
    *best = -1;
    for (int64_t i = 0; i < n; ++i) labels[i] = -1;
    for (int64_t f = 0; f < n; ++f) {                         /* :1222 frontiers are listed row-major */
        if (!frontier[f] || visited[f]) continue;
        int64_t qh = 0, qt = 0, sx = 0, sy = 0;
        queue[qt++] = f;
        visited[f] = 1;
        while (qh < qt) {                                     /* :1230-1238 breadth-first, 4-neighbourhood */
            const int64_t c = queue[qh++];
            const int32_t cx = (int32_t)(c / gs), cy = (int32_t)(c % gs);
            sx += cx; sy += cy;
            for (int d = 0; d < 4; ++d) {
                const int32_t nx = cx + dx[d], ny = cy + dy[d];
                if (nx < 0 || nx >= gs || ny < 0 || ny >= gs) continue;
                const int64_t ni = (int64_t)nx * gs + ny;
                if (frontier[ni] && !visited[ni]) { visited[ni] = 1; queue[qt++] = ni; }
            }
        }
        if (qt < min_cluster_size) continue;                  /* :1245 */
        const double cx = (double)sx / (double)qt, cy = (double)sy / (double)qt;   /* :1256-1257 */
        const int32_t rx = (int32_t)rint(cx), ry = (int32_t)rint(cy);             /* :1265 Python round: half to even */
        int32_t unknown = 0;
        for (int32_t ddx = -ig_radius; ddx <= ig_radius; ++ddx)
            for (int32_t ddy = -ig_radius; ddy <= ig_radius; ++ddy) {
                const int32_t nx = rx + ddx, ny = ry + ddy;
                if (nx < 0 || nx >= gs || ny < 0 || ny >= gs) continue;
                if (cell_unknown(cv_map, gs, nx, ny)) ++unknown;
            }
        if (kept < max_clusters) {
            for (int64_t k = 0; k < qt; ++k) labels[queue[k]] = kept;
            first[2 * kept] = (int32_t)(f / gs); first[2 * kept + 1] = (int32_t)(f % gs);
            sizes[kept] = (int32_t)qt;
            centers[2 * kept] = cx; centers[2 * kept + 1] = cy;
            gains[kept] = (double)unknown;
            if ((double)unknown > best_ig) { best_ig = (double)unknown; *best = kept; }   /* :1301 strict */
        }
        ++kept;
    }
    free(visited);
    free(queue);
    return kept;
}
