// mf_loss_math.h -- per-row arithmetic of the seven losses (statistics, loss values, backward coefficients), shared by the
// batch kernels of mf_loss.hip and the one-workgroup step of mf_step_small.hip: the same expressions, compiled once each way
// with -ffp-contract=off, give the same bits.
// Reference restated: alignment/contrastive/infonce/mine xfmr_rec/losses.py:164-246, pairwise :324-359; SURVEY.md Appendix A.
#pragma once

#include <cfloat>

#include "mf_common.h"

#ifdef __HIPCC__

static constexpr int NSTAT = 8;  // cnt, A(contrastive), mx, se, H, Hc, Lg, Ls
enum { ST_CNT = 0, ST_A = 1, ST_MX = 2, ST_SE = 3, ST_H = 4, ST_HC = 5, ST_LG = 6, ST_LS = 7 };
enum { NEED_CONTR = 1, NEED_LSE = 2, NEED_HINGE = 4, NEED_LOGI = 8 };
enum { G_EXP = 0, G_STEP = 1, G_SIGM = 2 };
static constexpr int KSEL_MAX = 64;
static constexpr int HITS_PLANES = 2;      // planes of the users' hit bit-vectors (= HITS_MAX_SPLIT, mask build)


struct RowStats {
    float cnt, A, mx, se, H, Hc, Lg, Ls;
};

__device__ __forceinline__ void stats_init(RowStats& s) {
    s.cnt = s.A = s.se = s.H = s.Hc = s.Lg = s.Ls = 0.f;
    s.mx = -FLT_MAX;
}
__device__ __forceinline__ void lse_merge(float& mx, float& se, float mx2, float se2) {
    const float m = fmaxf(mx, mx2);
    se = se * __expf(mx - m) + se2 * __expf(mx2 - m);
    mx = m;
}
// softplus(x) and sigmoid(x) from one exp:  e = exp(-|x|)
__device__ __forceinline__ void softplus_sigmoid(float x, float& sp, float& sg) {
    const float e = __expf(-fabsf(x));
    const float r = 1.f / (1.f + e);
    sp = fmaxf(x, 0.f) + __logf(1.f + e);
    sg = x >= 0.f ? r : e * r;
}
// accumulate one valid logit into the per-row statistics (mx handled by caller)
__device__ __forceinline__ void stats_add(RowStats& s, int need, float L, float sm, float lii, float margin) {
    s.cnt += 1.f;
    if (need & NEED_CONTR) s.A += fmaxf(L + sm, 0.f);
    if (need & (NEED_HINGE | NEED_LOGI)) {
        const float x = (L - lii) + margin;
        if (need & NEED_HINGE) {
            s.H += fmaxf(x, 0.f);
            s.Hc += x > 0.f ? 1.f : 0.f;
        }
        if (need & NEED_LOGI) {
            float sp, sg;
            softplus_sigmoid(x, sp, sg);
            s.Lg += sp;
            s.Ls += sg;
        }
    }
}

// the seven per-row loss values from a row's final statistics acc[NSTAT] (losses.py:164-246, :324-359)
__device__ __forceinline__ void row_losses(const float (&acc)[NSTAT], float t, float l, float dii, float sigma, float (&o)[MF_NUM_KINDS]) {
    const float w = fabsf(t);
    const float cnt = acc[ST_CNT], mx = acc[ST_MX], se = acc[ST_SE];
    const float den = cnt + 1e-10f;
    const float align = dii * t * sigma;                          // losses.py:164-170
    const float contr = (acc[ST_A] / den) * w;                    // losses.py:172-193
    const float m2 = fmaxf(mx, l);
    const float se2 = se * __expf(mx - m2) + __expf(l - m2);
    const float lse_all = m2 + __logf(se2);                       // diagonal forced in, :213-215
    const float lse_neg = cnt > 0.f ? mx + __logf(se) : -INFINITY;  // :242 (no valid negative: -inf)
    o[MF_ALIGNMENT] = align;
    o[MF_CONTRASTIVE] = contr;
    o[MF_ALIGNMENT_CONTRASTIVE] = align + contr;
    o[MF_INFONCE] = (lse_all - l) * w;
    o[MF_MINE] = (-l + lse_neg) * w;
    o[MF_PAIRWISE_HINGE] = (acc[ST_H] / den) * w;
    o[MF_PAIRWISE_LOGISTIC] = (acc[ST_LG] / den) * w;
}

// Per-row coefficients of dloss/dL for loss `kind`, WITHOUT the upstream gradient (the backward kernels scale
// by grad_out[0]):  rowc[0] = a, rowc[1] = b, rowc[2] = coefG, rowc[3] = gdiag  with, for element (i, j),
//   G'_ij = coefG_i * g((L_ij - a_i) + b_i)   (valid negatives),  G'_ii = gdiag_i,
//   du_i = sum_j G'_ij (v_j - u_i),  dv_j = sum_i G'_ij (u_i - v_j).
__device__ __forceinline__ void rowc_row(int kind, float t, float s, float l, float cnt, float mx, float se, float hc,
                                         float ls, float sigma, float margin, float& a, float& b, float& cg, float& gd) {
    a = b = cg = gd = 0.f;
    const float base = sigma * s * fabsf(t);
    const float den = cnt + 1e-10f;
    switch (kind) {
        case MF_ALIGNMENT: gd = -base; break;
        case MF_CONTRASTIVE: b = s * margin; cg = base / den; break;
        case MF_ALIGNMENT_CONTRASTIVE: b = s * margin; cg = base / den; gd = -base; break;
        // softmax weights as exp((L - max) - log(sum)): `a` is a logit (the subtraction is exact for the logits that matter),
        // `b` is small.  With a = lse -- one fp32 number at the logits' magnitude, 2000 at sigma = 1000 -- every weight of the
        // row carried lse's rounding (6e-5 relative at 2000) and the diagonal's 1 - p_ii did not match the sum of the others:
        // the components of du along v_j ~ v_i lost their leading digits (round 4; torch's log_softmax backward has the form below)
        case MF_INFONCE: {
            const float m2 = fmaxf(mx, l);
            const float neg = se * __expf(mx - m2);                  // sum over the valid negatives, relative to the row maximum
            const float se2 = neg + __expf(l - m2);
            a = m2; b = -__logf(se2); cg = base; gd = -base * (neg / se2);      // 1 - p_ii = (sum of the others) / (sum of all)
        } break;
        case MF_MINE:
            if (cnt > 0.f) { a = mx; b = -__logf(se); cg = base; }
            gd = -base;
            break;
        case MF_PAIRWISE_HINGE: a = l; b = margin; cg = base / den; gd = -cg * hc; break;
        case MF_PAIRWISE_LOGISTIC: a = l; b = margin; cg = base / den; gd = -cg * ls; break;
    }
}


__host__ __device__ static inline int gmode_of(int kind) {
    switch (kind) {
        case MF_INFONCE: case MF_MINE: return G_EXP;
        case MF_PAIRWISE_LOGISTIC: return G_SIGM;
        default: return G_STEP;
    }
}

__device__ __forceinline__ float g_of(int gmode, float x) {
    if (gmode == G_EXP) return __expf(x);
    if (gmode == G_STEP) return x > 0.f ? 1.f : 0.f;
    const float e = __expf(-fabsf(x));
    const float r = 1.f / (1.f + e);
    return x >= 0.f ? r : e * r;
}


// The mined dV accumulator is a 64-bit fixed-point integer per feature (integer addition commutes: the sum is the same
// bits in whatever order the terms arrive).  Its unit is 2^-E with E chosen PER BATCH so that the sum can never wrap:
// every term is g (u_e - v_e) with |g| <= gmax = max_i max(|cg_i|, |gd_i|) (the row coefficients; exp / step / sigmoid
// factors are <= 1) and |u_e - v_e| <= xmax = sqrt(max |u|^2) + sqrt(max |v|^2); a column receives at most B + 1 terms.
// With 2^ce > gmax xmax and 2^nb >= B + 2:  E = min(40, 61 - nb - ce), terms clamped to +-2^ce  =>  |sum| < 2^61.
// E = 40 (the unit every round before used) whenever gmax xmax (B + 2) < 2^21 -- sigma = 1, ratings <= 5, B = 8192 --
// so those batches keep their bits; sigma = 1000 takes E = 34 at B = 8192 instead of wrapping.  The maxima
// are exact whatever the reduction order, so the multi-kernel path and the one-launch step pick the same E.
struct DvFix {
    float scale;        // 2^E
    float clamp;        // 2^ce (FLT_MAX when that overflows)
    float inv;          // 2^-E
};
__host__ __device__ static inline DvFix dv_fix_of(float gmax, float nu2max, float nv2max, long long B) {
    const float p = gmax * (sqrtf(nu2max) + sqrtf(nv2max));
    int ce = 128;
    if (p < __builtin_inff()) {                 // (false for NaN too)
        ce = -126;
        if (p > 0.f) { (void)frexpf(p, &ce); }  // p = m 2^ce, m in [0.5, 1)
    }
    int nb = 1;
    while ((1ll << nb) < B + 2) ++nb;
    int E = 61 - nb - ce;
    E = E > 40 ? 40 : (E < -120 ? -120 : E);
    DvFix f;
    f.scale = ldexpf(1.f, E);
    f.inv = ldexpf(1.f, -E);
    f.clamp = ce >= 128 ? 3.4028234663852886e38f : ldexpf(1.f, ce);
    return f;
}
__device__ __forceinline__ unsigned dv_mag(float x) { return __builtin_bit_cast(unsigned, x) & 0x7FFFFFFFu; }
__device__ __forceinline__ long long dv_fix_term(float x, float scale, float clamp) {
    return (long long)__builtin_rintf(fminf(fmaxf(x, -clamp), clamp) * scale);
}


#endif  // __HIPCC__
