/*
 * mf_numerics.h -- the canonical fp32 arithmetic of the hot path, shared by the
 * HIP kernels (device) and by the CPU oracle (oracle/chain.c, host, gcc).
 *
 * Why a shared header: the parity contract of this repo is
 *   * top-k indices bit-exact,
 *   * mined negative masks bit-exact,
 *   * logits / loss within 1e-4 (fp32, sigma = 1),
 * and the first two only hold if the score of a (row, column) pair is the same
 * 32 bits on both sides.  gfx950's `v_mfma_f32_32x32x2_f32` is, per output
 * element, a k-ordered chain of single-rounded fmaf() (MI355X_MICROARCH.md,
 * "Matrix cores"), so the dot product is *defined* here as that chain and every
 * non-MFMA use (diagonal, sparse mined pairs, CPU oracle) calls the same inline.
 *
 * Reference semantics being restated (yxtay/matrix-factorization-torch):
 *   xfmr_rec/losses.py:9-12   squared_distance = cdist(u, v) ** 2 / 2
 *   xfmr_rec/losses.py:181-183 logits = -squared_distance * sign(target) * sigma
 * cdist's own summation order is BLAS-defined (not specified), so the reference
 * pins values only to ~5e-7; the order below is ours.
 *
 * k order ("group-of-8 interleave"): a lane of the MFMA tile holds a float4 of
 * its row at k = 8g + 4h .. 8g + 4h + 3 (h = lane >> 5), and MFMA step t of group
 * g multiplies element t of both halves, first h = 0 then h = 1:
 *     8g+0, 8g+4, 8g+1, 8g+5, 8g+2, 8g+6, 8g+3, 8g+7,  g = 0 .. d/8 - 1.
 * d must be a multiple of 8 (the host wrappers zero-pad otherwise; fmaf(0,0,acc)
 * is exact, so padding never changes a value).
 */
#ifndef MF_NUMERICS_H
#define MF_NUMERICS_H

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define MF_HD __host__ __device__ __forceinline__
#define MF_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define MF_FMAXF(a, b) __builtin_fmaxf((a), (b))
#else
#include <math.h>
#define MF_HD static inline
#define MF_FMAF(a, b, c) fmaf((a), (b), (c))
#define MF_FMAXF(a, b) fmaxf((a), (b))
#endif

/* canonical k-ordered fmaf chain == one output element of the fp32 MFMA tile */
MF_HD float mf_dot_chain(const float* a, const float* b, int d) {
    float acc = 0.0f;
    for (int g = 0; g < d; g += 8) {
        for (int t = 0; t < 4; ++t) {
            acc = MF_FMAF(a[g + t], b[g + t], acc);
            acc = MF_FMAF(a[g + 4 + t], b[g + 4 + t], acc);
        }
    }
    return acc;
}

/* half squared distance from the three chain products; clamp mirrors cdist's
 * clamp_min(0) before its sqrt (losses.py:12 via ATen _euclidean_dist). */
MF_HD float mf_half_sqdist(float nu, float nv, float dot) {
    float sq = MF_FMAXF(MF_FMAF(-2.0f, dot, nu + nv), 0.0f);
    return 0.5f * sq;
}

/* logits[i][j] = -D_ij * sign(target_i) * sigma (losses.py:181-183) [- log q_j: our optional logQ
 * correction, no reference counterpart; SURVEY 0.3], evaluated as two fused multiply-adds on the
 * chain products:
 *     L = fma(sigma*s, dot, fma(-0.5*sigma*s, nu + nv, -logq))
 * (D = 0.5 (nu + nv - 2 dot) expanded).  Three VALU instructions per element instead of eight:
 * on gfx950 every VALU instruction in an MFMA loop costs matrix throughput.  cdist's clamp of
 * tiny negative squared distances is not reproduced here (|effect| <= 1e-7 * sigma); the
 * AlignmentLoss value uses mf_half_sqdist, which keeps it. */
MF_HD float mf_logit(float nu, float nv, float dot, float sgn, float sigma, float logq) {
    const float as = sigma * sgn;
    const float hs = -0.5f * as;
    return MF_FMAF(as, dot, MF_FMAF(hs, nu + nv, -logq));
}

MF_HD float mf_sign(float t) { return (t > 0.0f) ? 1.0f : ((t < 0.0f) ? -1.0f : 0.0f); }

/* order-preserving map float -> uint32 (larger float => larger uint). */
MF_HD unsigned mf_orderable(float x) {
    union { float f; unsigned u; } c;
    c.f = x;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
MF_HD float mf_unorderable(unsigned k) {
    union { float f; unsigned u; } c;
    c.u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return c.f;
}

/*
 * 64-bit selection keys: "larger key == better", unique per column, so every
 * top-k in this repo is a max-k over uint64 and ties cannot exist.
 *
 * retrieval (our spec; SURVEY 8c: stable (score desc, item index asc)):
 *     key = orderable(score) << 32 | ~col
 * semi-hard mining (losses.py:134-162): with Dm = L_ij - L_ii,
 *     semi-hard (Dm < 0) before hard (Dm >= 0); inside semi-hard larger Dm
 *     first, inside hard smaller Dm first; lowest column wins exact ties.
 *     The reference sorts by where(Dm<0, Dm - min_j Dm, -Dm); subtracting the
 *     row constant is monotone, so this order is a refinement of the
 *     reference's (it only decides what torch.topk leaves unspecified).
 *     key = cls << 62 | orderable(cls==2 ? Dm : -Dm) << 30 | (0x3FFFFFFF - col)
 */
/* the keys are built from two 32-bit halves so that the hot loops never touch 64-bit ALU ops */
MF_HD unsigned mf_key_retrieval_hi(float score) { return mf_orderable(score); }
MF_HD unsigned mf_key_retrieval_lo(unsigned col) { return ~col; }
MF_HD unsigned long long mf_key_retrieval(float score, unsigned col) {
    return ((unsigned long long)mf_key_retrieval_hi(score) << 32) | (unsigned long long)mf_key_retrieval_lo(col);
}
MF_HD unsigned mf_key_retrieval_col(unsigned long long key) { return ~(unsigned)(key & 0xFFFFFFFFull); }
MF_HD float mf_key_retrieval_score(unsigned long long key) { return mf_unorderable((unsigned)(key >> 32)); }

MF_HD unsigned mf_key_mining_hi(float dm) {
    const unsigned cls = (dm < 0.0f) ? 2u : 1u;
    const unsigned ord = mf_orderable((dm < 0.0f) ? dm : (0.0f - dm));   /* 0 - dm: -0 and +0 give the same key */
    return (cls << 30) | (ord >> 2);
}
MF_HD unsigned mf_key_mining_lo(float dm, unsigned col) {
    const unsigned ord = mf_orderable((dm < 0.0f) ? dm : (0.0f - dm));
    return (ord << 30) | (0x3FFFFFFFu - col);
}
MF_HD unsigned long long mf_key_mining(float dm, unsigned col) {
    return ((unsigned long long)mf_key_mining_hi(dm) << 32) | (unsigned long long)mf_key_mining_lo(dm, col);
}
MF_HD unsigned mf_key_mining_col(unsigned long long key) {
    return 0x3FFFFFFFu - (unsigned)(key & 0x3FFFFFFFull);
}

/*
 * Hash / bloom embeddings (BASELINE config 5; our spec, no reference counterpart): an id owns
 * num_hashes rows of a table of num_buckets rows and its embedding is their sum.  Hash j of id x is
 * the SplitMix64 output function applied to x + seed + (j + 1) * golden-gamma, reduced modulo
 * num_buckets (for x = seed = 0, j = 0 the mixed word is SplitMix64's first output
 * 0xE220A8397B1DCDAF -- the known-answer that pins the constants, tests/test_host_cpu.py).
 */
MF_HD unsigned long long mf_splitmix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
MF_HD long long mf_hash_bucket(long long id, int j, unsigned long long seed, long long num_buckets) {
    const unsigned long long z = (unsigned long long)id + seed + (unsigned long long)(j + 1) * 0x9E3779B97F4A7C15ull;
    return (long long)(mf_splitmix64(z) % (unsigned long long)num_buckets);
}

#endif /* MF_NUMERICS_H */
