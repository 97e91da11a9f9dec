/*
 * oracle/sampling_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the integer half of the dense contrastive loss of
 * RViMLab/ECCV2022-multi-scale-and-cross-scale-contrastive-segmentation:
 * label down-sampling, per-(image,class) histogram, pair selection, the
 * views-per-class rule, and the permutation-based anchor pick.  Every function
 * cites the reference lines it restates (paths relative to /root/reference).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object.  The product path (HIP kernels behind
 * include/dcl_hip.h) never links or calls it.
 *
 * Third-party arithmetic restated here: torch.randperm on the CPU default
 * generator (PyTorch 2.10.0: at::mt19937 seeded with the low 32 bits of the
 * manual seed; randperm_cpu = forward Fisher-Yates, one 32-bit draw per step,
 * z = draw % (n - i), n - 1 draws per call).  tests/test_oracle.py pins it
 * against torch.randperm itself and against tests/golden/G8.
 *
 * Build (done by __graft_entry__.build() and tests/conftest.py):
 *   gcc -O2 -shared -fPIC -o oracle/_build/libsampling_oracle.so oracle/sampling_oracle.c
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* MT19937 (Matsumoto & Nishimura 1998), the engine behind at::mt19937 */
/* ------------------------------------------------------------------ */
typedef struct {
    uint32_t mt[624];
    int idx;
} orc_mt19937;

void orc_mt_seed(orc_mt19937 *g, uint64_t seed)
{
    g->mt[0] = (uint32_t)(seed & 0xffffffffu);
    for (int j = 1; j < 624; ++j)
        g->mt[j] = 1812433253u * (g->mt[j - 1] ^ (g->mt[j - 1] >> 30)) + (uint32_t)j;
    g->idx = 624;
}

static void orc_mt_twist(orc_mt19937 *g)
{
    uint32_t *mt = g->mt;
    for (int k = 0; k < 624; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
}

uint32_t orc_mt_next(orc_mt19937 *g)
{
    if (g->idx >= 624)
        orc_mt_twist(g);
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* torch.randperm(n) on the CPU generator (losses/DenseContrastiveLossV2.py:121). */
void orc_randperm(orc_mt19937 *g, int64_t n, int64_t *out)
{
    for (int64_t i = 0; i < n; ++i)
        out[i] = i;
    for (int64_t i = 0; i + 1 < n; ++i) {
        int64_t z = (int64_t)(orc_mt_next(g) % (uint64_t)(n - i));
        int64_t t = out[i];
        out[i] = out[z + i];
        out[z + i] = t;
    }
}

/* --------------------------------------------------------------------- */
/* F.interpolate(label.float(), (h//s, w//s), mode='nearest').long()      */
/* losses/DenseContrastiveLossV2.py:194-206.  ATen's legacy 'nearest'      */
/* source index: identity if sizes match, >>1 if out == 2*in, otherwise    */
/* min(floorf(dst * (float)in/out), in-1) in single precision.             */
/* --------------------------------------------------------------------- */
static int64_t orc_nearest_src(int64_t dst, int64_t in_size, int64_t out_size)
{
    if (out_size == in_size)
        return dst;
    if (out_size == 2 * in_size)
        return dst >> 1;
    float scale = (float)in_size / (float)out_size;
    int64_t src = (int64_t)floorf((float)dst * scale);
    return src < in_size - 1 ? src : in_size - 1;
}

void orc_downsample_labels(const int64_t *label, int64_t n, int64_t H, int64_t W,
                           int64_t scale, int64_t *out /* [n, H/scale, W/scale] */)
{
    int64_t h = H / scale, w = W / scale;
    for (int64_t b = 0; b < n; ++b)
        for (int64_t i = 0; i < h; ++i) {
            int64_t si = orc_nearest_src(i, H, h);
            for (int64_t j = 0; j < w; ++j) {
                int64_t sj = orc_nearest_src(j, W, w);
                out[(b * h + i) * w + j] = label[(b * H + si) * W + sj];
            }
        }
}

/* compare = lbl.unsqueeze(-1) == arange(K); cls_counts = compare.sum(1)  (:100-103) */
void orc_class_counts(const int64_t *lbl, int64_t n, int64_t hw, int64_t K,
                      int64_t *counts /* [n, K] */)
{
    memset(counts, 0, sizeof(int64_t) * (size_t)(n * K));
    for (int64_t b = 0; b < n; ++b)
        for (int64_t p = 0; p < hw; ++p) {
            int64_t c = lbl[b * hw + p];
            if (c >= 0 && c < K) /* ids outside [0,K) match no arange entry */
                counts[b * K + c] += 1;
        }
}

/* present_inds = torch.where(cls_counts[:, :-1] >= min_views)  (:106-107):
 * row-major (image asc, class asc); the LAST class column is always dropped.
 * Returns T. */
int64_t orc_select_pairs(const int64_t *counts, int64_t n, int64_t K, int64_t min_views,
                         int64_t *pair_b, int64_t *pair_k, int64_t *pair_cnt)
{
    int64_t T = 0;
    for (int64_t b = 0; b < n; ++b)
        for (int64_t k = 0; k + 1 < K; ++k)
            if (counts[b * K + k] >= min_views) {
                pair_b[T] = b;
                pair_k[T] = k;
                pair_cnt[T] = counts[b * K + k];
                ++T;
            }
    return T;
}

/* _select_views_per_class (:64-84) applied to min_views = min over pairs (:110).
 * *log_flag is set when either cap binds (self.log_this_step = True). */
int64_t orc_select_views(const int64_t *pair_cnt, int64_t T, int64_t max_views_per_class,
                         int64_t max_features_total, int *log_flag)
{
    int64_t m = pair_cnt[0];
    for (int64_t t = 1; t < T; ++t)
        if (pair_cnt[t] < m)
            m = pair_cnt[t];
    int64_t V;
    *log_flag = 0;
    if (max_views_per_class == 1) {
        V = m;
    } else {
        V = m < max_views_per_class ? m : max_views_per_class;
        if (V == max_views_per_class)
            *log_flag = 1;
    }
    if (V * T > max_features_total) {
        V = max_features_total / T;
        *log_flag = 1;
    }
    return V;
}

/* Sampling loop (:117-124): for each pair in order, ascending pixel list of the
 * class (nonzero), randperm(len) from the running generator, keep the first V.
 * pix is [T, V]. */
void orc_sample_pixels(orc_mt19937 *g, const int64_t *lbl, int64_t hw,
                       const int64_t *pair_b, const int64_t *pair_k, int64_t T, int64_t V,
                       int64_t *pix)
{
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)hw);
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)hw);
    for (int64_t t = 0; t < T; ++t) {
        const int64_t *row = lbl + pair_b[t] * hw;
        int64_t m = 0;
        for (int64_t p = 0; p < hw; ++p)
            if (row[p] == pair_k[t])
                pos[m++] = p;
        orc_randperm(g, m, perm);
        for (int64_t v = 0; v < V; ++v)
            pix[t * V + v] = pos[perm[v]];
    }
    free(pos);
    free(perm);
}

/* Same loop but with caller-provided permutations (concatenated, pair t's
 * permutation starts at perm_off[t]); used when the draws come from
 * torch.randperm itself. */
void orc_sample_pixels_with_perms(const int64_t *lbl, int64_t hw, const int64_t *pair_b,
                                  const int64_t *pair_k, int64_t T, int64_t V,
                                  const int64_t *perms, const int64_t *perm_off, int64_t *pix)
{
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)hw);
    for (int64_t t = 0; t < T; ++t) {
        const int64_t *row = lbl + pair_b[t] * hw;
        int64_t m = 0;
        for (int64_t p = 0; p < hw; ++p)
            if (row[p] == pair_k[t])
                pos[m++] = p;
        for (int64_t v = 0; v < V; ++v)
            pix[t * V + v] = pos[perms[perm_off[t] + v]];
    }
    free(pos);
}

size_t orc_mt_sizeof(void) { return sizeof(orc_mt19937); }
