/*
 * Plain-C restatement of the two index/byte-exact pieces of the Snuffy hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.
 * Only tests/ may load it (through ctypes).  It is an independent second opinion for the torch oracle
 * (oracle/snuffy_oracle.py) and for the HIP kernels:
 *
 *   orc_topk_desc_stable   top-Lambda selection         reference snuffy.py:128-130  (torch.sort descending, slice)
 *                          tie rule: descending score, ascending index; NaN first (torch.sort semantics)
 *   orc_k_split            k1 / k2 arithmetic           reference snuffy.py:124,129,137-140 (python float math == C double)
 *   orc_sparse_attention   attention()                  reference snuffy.py:160-168, accumulated in double
 *
 * PARITY PIN: tests/test_oracle_c.py checks these against the golden vectors captured from the reference
 * (tests/golden/f2_selection.npz, f1_*.npz) and against the torch oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float v;
    int64_t i;
} pair_t;

static int cmp_desc(const void* a, const void* b) {
    const pair_t* x = (const pair_t*)a;
    const pair_t* y = (const pair_t*)b;
    int xn = isnan(x->v), yn = isnan(y->v);
    if (xn || yn) {
        if (xn && !yn) return -1;
        if (!xn && yn) return 1;
    } else {
        if (x->v > y->v) return -1;
        if (x->v < y->v) return 1;
    }
    return (x->i > y->i) - (x->i < y->i); /* ties (incl. -0.0 == +0.0): ascending index */
}

int orc_topk_desc_stable(const float* c, int64_t n, int64_t stride, int k, int64_t* idx) {
    if (!c || !idx || n < 1 || k < 0 || k > n) return -1;
    pair_t* p = (pair_t*)malloc((size_t)n * sizeof(pair_t));
    if (!p) return -2;
    for (int64_t i = 0; i < n; ++i) {
        p[i].v = c[i * stride];
        p[i].i = i;
    }
    qsort(p, (size_t)n, sizeof(pair_t), cmp_desc);
    for (int j = 0; j < k; ++j) idx[j] = p[j].i;
    free(p);
    return 0;
}

void orc_k_split(int big_lambda, double random_patch_share, int64_t n, int64_t* k1, int64_t* k2) {
    double top_share = 1.0 - random_patch_share;
    int64_t ceil_top = (int64_t)ceil((double)big_lambda * top_share);
    *k1 = ceil_top < n ? ceil_top : n;
    int64_t r = (int64_t)((double)big_lambda * random_patch_share); /* int() truncates toward zero */
    int64_t room = n - ceil_top > 0 ? n - ceil_top : 0;
    *k2 = r < room ? r : room;
}

/* q, v [n, h*dk]; kp [k, h*dk]; out [k, h*dk]; attn [h, n, k] (nullable). */
int orc_sparse_attention(const float* q, const float* kp, const float* v, int64_t n, int k, int h, int dk, float* out,
                         float* attn) {
    if (!q || !kp || !v || !out || n < 1 || k < 1 || h < 1 || dk < 1) return -1;
    const int d = h * dk;
    const double scale = 1.0 / sqrt((double)dk);
    double* acc = (double*)calloc((size_t)k * d, sizeof(double));
    double* s = (double*)malloc((size_t)k * sizeof(double));
    if (!acc || !s) return -2;
    for (int a = 0; a < h; ++a) {
        for (int64_t i = 0; i < n; ++i) {
            const float* qi = q + i * d + a * dk;
            double m = -INFINITY;
            for (int j = 0; j < k; ++j) {
                const float* kj = kp + (int64_t)j * d + a * dk;
                double t = 0.0;
                for (int e = 0; e < dk; ++e) t += (double)qi[e] * (double)kj[e];
                s[j] = t * scale;
                if (s[j] > m) m = s[j];
            }
            double l = 0.0;
            for (int j = 0; j < k; ++j) {
                s[j] = exp(s[j] - m);
                l += s[j];
            }
            const float* vi = v + i * d + a * dk;
            for (int j = 0; j < k; ++j) {
                double p = s[j] / l;
                if (attn) attn[((int64_t)a * n + i) * k + j] = (float)p;
                double* o = acc + (int64_t)j * d + a * dk;
                for (int e = 0; e < dk; ++e) o[e] += p * (double)vi[e];
            }
        }
    }
    for (int64_t t = 0; t < (int64_t)k * d; ++t) out[t] = (float)acc[t];
    free(acc);
    free(s);
    return 0;
}
