/*
 * oracle/chain.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Scalar CPU restatement of the score arithmetic of the hot path, bit-for-bit
 * the arithmetic the gfx950 kernels perform (include/mf_numerics.h): a k-ordered
 * fmaf chain per (row, column), then the reference's logit formula.
 *
 * Follows: xfmr_rec/losses.py:9-12 (squared_distance), :181-183 / :204-206 /
 * :234-236 / :334-336 (logits = -D * sign(target) * sigma), and for retrieval
 * xfmr_rec/data/lightning.py:237-259 (cosine score of a unit-norm query against
 * unit-norm item rows == their dot product; exclusion prefilter; top-k desc).
 * The reference's own retrieval is LanceDB ANN (absent here, approximate), so the
 * exact brute-force order below is OUR spec; it is pinned against fp64 matmul + stable
 * sort and an independent restatement of the key order in tests/test_oracle_pins.py.
 *
 * Built by oracle/Makefile into oracle/_build/liborc.so; only tests/, smoke()
 * and bench.py's cpu_baseline leg may load it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mf_numerics.h"

/* out[i] = chain ||x_i||^2 */
void orc_sqnorm(const float* x, int64_t n, int d, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = mf_dot_chain(x + i * d, x + i * d, d);
}

/* out[i][j] = chain dot(u_i, v_j) */
void orc_scores(const float* u, const float* v, int64_t B, int64_t N, int d, float* out) {
    for (int64_t i = 0; i < B; ++i)
        for (int64_t j = 0; j < N; ++j) out[i * N + j] = mf_dot_chain(u + i * d, v + j * d, d);
}

/* full logits matrix; logq may be NULL */
void orc_logits(const float* u, const float* v, const float* target, const float* logq,
                int64_t B, int64_t N, int d, float sigma, float* out) {
    float* nu = (float*)malloc(sizeof(float) * (size_t)B);
    float* nv = (float*)malloc(sizeof(float) * (size_t)N);
    orc_sqnorm(u, B, d, nu);
    orc_sqnorm(v, N, d, nv);
    for (int64_t i = 0; i < B; ++i) {
        float sgn = mf_sign(target[i]);
        for (int64_t j = 0; j < N; ++j) {
            float dot = mf_dot_chain(u + i * d, v + j * d, d);
            out[i * N + j] = mf_logit(nu[i], nv[j], dot, sgn, sigma, logq ? logq[j] : 0.0f);
        }
    }
    free(nu);
    free(nv);
}

/* mining keys for one logits matrix (see mf_key_mining); masked columns get 0 */
void orc_mining_keys(const float* logits, const uint8_t* neg_mask, int64_t B, int64_t N,
                     uint64_t* keys) {
    for (int64_t i = 0; i < B; ++i) {
        float lii = logits[i * N + i];
        for (int64_t j = 0; j < N; ++j)
            keys[i * N + j] =
                neg_mask[i * N + j] ? mf_key_mining(logits[i * N + j] - lii, (unsigned)j) : 0ull;
    }
}

static int cmp_u64_desc(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x < y) - (x > y);
}

/*
 * Exact brute-force top-k of chain scores with a per-query exclusion list
 * (CSR: excl_off[Q+1], excl_idx[...] item row indices, any order).
 * Order: score desc, item index asc.  Rows with fewer than k candidates are
 * padded with index -1 / score -inf.
 */
void orc_topk(const float* q, const float* items, int64_t Q, int64_t N, int d, int k,
              const int64_t* excl_off, const int64_t* excl_idx, float* out_scores,
              int64_t* out_idx) {
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
    uint8_t* ex = (uint8_t*)malloc((size_t)N);
    for (int64_t r = 0; r < Q; ++r) {
        memset(ex, 0, (size_t)N);
        if (excl_off)
            for (int64_t e = excl_off[r]; e < excl_off[r + 1]; ++e)
                if (excl_idx[e] >= 0 && excl_idx[e] < N) ex[excl_idx[e]] = 1;
        int64_t m = 0;
        for (int64_t j = 0; j < N; ++j) {
            if (ex[j]) continue;
            keys[m++] = mf_key_retrieval(mf_dot_chain(q + r * d, items + j * d, d), (unsigned)j);
        }
        qsort(keys, (size_t)m, sizeof(uint64_t), cmp_u64_desc);
        for (int t = 0; t < k; ++t) {
            if (t < m) {
                out_scores[r * k + t] = mf_key_retrieval_score(keys[t]);
                out_idx[r * k + t] = (int64_t)mf_key_retrieval_col(keys[t]);
            } else {
                out_scores[r * k + t] = -INFINITY;
                out_idx[r * k + t] = -1;
            }
        }
    }
    free(keys);
    free(ex);
}
