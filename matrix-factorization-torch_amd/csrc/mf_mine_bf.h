// mf_mine_bf.h -- the mined losses' candidate search through a SPLIT-bf16 prefilter (gfx950).
//
// Semi-hard mining (xfmr_rec/losses.py:134-162) needs, per user, the k best columns of the B x N matrix of
// Dm = L_ij - L_ii by the mining order (mf_numerics.h), bit-exact.  select_kernel (mf_select.h) streams that matrix on the
// fp32 matrix cores -- 1/16 of the bf16 rate -- although all but a few dozen columns per user are nowhere near the cut.
// Here the matrix is streamed ONCE on the bf16 cores with every fp32 operand split in two bf16 numbers,
//
//     x = xh + xl + r,  xh = bf16(x),  xl = bf16(x - xh),  |r| <= 2^-18 |x|
//     x y ~ xh yh + xl yh + xh yl            (three products, each exact in fp32; what is dropped: <= 3.03 2^-18 |x||y|)
//
// and the O(1) terms of Dm ride in one extra k-step of the hi x hi product (16 more k slots):
//
//     Dm_ij = as_i (u_i . v_j)  -  s_i w_j  +  lqn_j  +  rho_i          as = sigma s_i, w_j = sigma |v_j|^2 / 2 (= -hs nv_j),
//     item side  [ v_j | w1 w2 w3 | q1 q2 q3 | 1 1 1 | 0.. ]            lqn_j = -logq_j, rho_i = hs |u_i|^2 - L_ii - mid_i
//     user side  [ as u_i | -s -s -s | 1 1 1 | r1 r2 r3 | 0.. ]         (w, lqn, rho: three bf16 pieces each -- 24 bits)
//
// so the accumulator of an element IS Dm_ij - mid_i, and "may this column be among the k best" is ONE compare per element:
// |acc| <= half_i, where [mid - half, mid + half] is the interval of Dm the user's bound admits (MiningPolicy::make_thr of the
// seeding pass's bound) widened by a RIGOROUS bound eps_i of |acc + mid - Dm as select_kernel would compute it| (below).
// The few columns that pass (tens per user) are rescored with the canonical fp32 chain and keyed exactly as
// MiningPolicy::key does; the row lists that come out feed mined_rows_kernel unchanged.  A user whose lists overflow (all
// scores equal, a zero target, non-finite inputs) is answered by its rescoring wave walking the whole row exactly.
//
// Error bound (mine_bound_kernel).  With P = |as| |u_i| max|v|, W = sigma max|v|^2 / 2, Q = max |lqn|, and
// S = 1.01 P + W + Q + |rho| >= the sum of the absolute values of everything one accumulator adds up (nt = 3 d + 16 terms):
//   * the matrix core's fp32 accumulation, any order, any grouping: <= nt 2^-23 S (twice the round-to-nearest bound);
//   * the dropped products and the rounding of as u:                 <= 3.1 2^-18 P;
//   * w's rounding, the three-piece splits, rho's rounding:          <= 2^-23 (W + Q + |rho|);
//   * the exact side's own rounding (chain, two fma, one subtraction), M = P + (|hs| |u|^2 + W) + Q + |L_ii|:
//                                                                    <= d 2^-23 P + 2^-21 M;
// eps = 1.01 x their sum (the norms are themselves fp32 chains).
#pragma once

#include <cstdlib>

#include "mf_common.h"
#include "mf_select.h"
#include "mf_stream.h"

typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));

static constexpr int MBF_WAVES = 8;                     // waves per scan workgroup, two per SIMD
static constexpr int MBF_XT = 2;                        // user tiles (32 users) a wave keeps in registers
static constexpr int MBF_BLOCK = 2;                     // item tiles between two meeting points of the workgroup
static constexpr int MBF_CAPL = 16;                     // entries a rescoring lane takes: a user's list of one chunk holds lpc x MBF_CAPL (lpc = 2, or 4 for num_negatives > 16)
static constexpr int MBF_HCAP = 2048;                   // hits a scan wave buffers in LDS per chunk (64 users; ~80 are usual at num_negatives = 4, ~1000 at 16)
static constexpr int MBF_ROWS_WG = 32 * MBF_XT * MBF_WAVES;      // 512 users per workgroup
static constexpr int MBF_MAXSLOTS = 16;                 // pairs of words the item kernel's blocks spread their maxima over ...
static constexpr int MBF_MAXSTRIDE = 32;                // ... one cache line apart (in words)
static constexpr int MBF_SPILL = 128;                   // entries of a user's spill list: hits beyond a lane's 16 (more: the fp32 search answers the batch)
static constexpr int MBF_MAXLISTS = 64;                 // lanes of the rescoring wave that take list entries (lpc per chunk, at most 16 chunks)

struct MineBfPlan {
    bool ok;                // the shape CAN be served by this path (its arrays exist in the workspace)
    bool pays;              // ... and is served by default: where it is faster than the fp32 search (measured, tools/lab/mined_shapes.py)
    int NT;                 // 32-row item tiles
    int64_t Xq, Nq;         // users padded to a workgroup's 512, items to a tile's 32
    int gy;                 // user blocks
    int nchunk, tpc;        // item chunks (grid.x) and tiles per chunk
    int lpc, nlists;        // rescoring lanes per chunk (a chunk's list holds 16 lpc columns), lpc x nchunk lanes in all
    int rowb;               // bytes of an item's row in the plane: (2 d + 16) bf16
    int blk;                // columns per bit of a representative's copy bitmap (64 bits span the batch; a power of two >= 256)
};
static inline MineBfPlan mine_bf_plan(int64_t B, int64_t N, int d, int k) {
    MineBfPlan p{};
    p.NT = (int)((N + 31) / 32);
    p.Nq = (int64_t)p.NT * 32;
    p.Xq = (B + MBF_ROWS_WG - 1) / MBF_ROWS_WG * MBF_ROWS_WG;
    p.gy = (int)(p.Xq / MBF_ROWS_WG);
    p.rowb = (2 * d + 16) * 2;
    p.blk = 256;
    while ((int64_t)p.blk * 64 < p.Nq) p.blk *= 2;
    int want = (256 + p.gy - 1) / p.gy;                  // one workgroup per CU
    if (want > 16) want = 16;
    if (want < 1) want = 1;
    p.tpc = (p.NT + want - 1) / want;
    p.tpc = (p.tpc + MBF_BLOCK - 1) / MBF_BLOCK * MBF_BLOCK;
    p.nchunk = (p.NT + p.tpc - 1) / p.tpc;
    p.lpc = k > 16 ? 4 : 2;                              // (~4 k columns per user pass at k = 32, most of them among the batch's first half)
    p.nlists = p.lpc * p.nchunk;
    // (the seeding pass of mf_select_plan exists from 64 tiles on; below that, and for short batches, select_kernel is fast anyway)
    p.ok = (d == 64 || d == 128) && B >= 256 && N >= 2048 && k >= 1 && k <= 32 && (int64_t)p.tpc * 32 * p.rowb <= (int64_t)MF_SRD_MAX_BYTES;
    // B = 8192, N = 16,384, d = 128 (us per step, prefilter / fp32 search; profiles/r04_mined_shapes.log): k = 4: 389 / 677,
    // k = 8: 456 / 737, k = 16: 614 / 855 (2048 hits buffered per wave and chunk; 512 overflowed there), k = 32: 1043 / 1142 (four
    // rescoring lanes per chunk and a quarter-sample seed: ~150 columns per user are rescored); at B = 2048 there are too few user
    // blocks to fill the chip (196 / 187)
    p.pays = p.ok && B >= 4096;
    return p;
}

// run-time switch: 1 = the prefilter where it pays (default), 2 = wherever it can serve (tests and the stress compare the two
// searches on small shapes), 0 = the fp32 search everywhere; MF_MINE_BF in the environment sets the initial value
static int g_mine_bf_mode = -1;
static inline int mine_bf_mode() {
    if (g_mine_bf_mode < 0) {
        const char* e = getenv("MF_MINE_BF");
        g_mine_bf_mode = (e && e[0] == '0') ? 0 : (e && e[0] == '2') ? 2 : 1;
    }
    return g_mine_bf_mode;
}
static inline bool mine_bf_use(const MineBfPlan& p) { return p.ok && (mine_bf_mode() == 2 || (mine_bf_mode() == 1 && p.pays)); }

#ifdef __HIPCC__

__device__ __forceinline__ unsigned short mbf_round(float x) {
    const __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float mbf_float(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
// x -> three bf16 pieces whose sum is x to 2^-26 |x| (the subtractions are exact)
__device__ __forceinline__ void mbf_split3(float x, unsigned short (&o)[3]) {
    o[0] = mbf_round(x);
    const float r1 = x - mbf_float(o[0]);
    o[1] = mbf_round(r1);
    const float r2 = r1 - mbf_float(o[1]);
    o[2] = mbf_round(r2);
}

// ------------------------------------------------------- user fragments ----
// as u (as = sigma sign(target)) split in two bf16 numbers, in MFMA operand order (hi steps, then lo steps); one thread per
// (user, 8 elements); users past B: zeros
struct MineUserFrags {
    const float *u, *sgn;
    int64_t B, Xq;
    float sigma;
    mbf16x8* ufrag;          // [Xq / 32][2 KS][64]
};
template <int D>
__device__ __forceinline__ void mine_user_frags(const MineUserFrags& p, int64_t t) {
    constexpr int LPR = D / 8, KS = D / 16;
    const int64_t x = t / LPR;
    const int j = (int)(t % LPR);
    if (x >= p.Xq) return;
    const bool real = x < p.B;
    const float as = p.sigma * (real ? p.sgn[x] : 0.f);
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    u16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
    if (real) {
        const f32x4 a = reinterpret_cast<const f32x4*>(p.u + x * D)[2 * j], b = reinterpret_cast<const f32x4*>(p.u + x * D)[2 * j + 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float y = as * (i < 4 ? a[i] : b[i - 4]);
            hi[i] = mbf_round(y);
            lo[i] = mbf_round(y - mbf_float(hi[i]));
        }
    }
    const int step = j >> 1, h = j & 1;
    mbf16x8* f = p.ufrag + ((x >> 5) * (2 * KS)) * 64 + (x & 31) + 32 * h;
    f[step * 64] = __builtin_bit_cast(mbf16x8, hi);
    f[(KS + step) * 64] = __builtin_bit_cast(mbf16x8, lo);
}

// ------------------------------------------------------------- item plane ----
// One d/8-lane group per item: [hi d | lo d | aug 16] bf16, plus the maxima the error bound needs.
//
// Duplicate columns.  A batch drawn by popularity repeats its popular items hundreds of times (Zipf(1) over 62,423 items, 8192
// draws: ~700 copies of the first), and copies tie EXACTLY: whenever one is near a user's cut all are, and no list is long
// enough.  But copies need no search: with rep(j) = the first column whose row AND logQ are bit for bit column j's (found through
// the mask builder's colfirst -- first column with j's item id -- and verified on the values themselves), the copies of a column
// rank directly behind it, in column order.  So only representatives are scanned (a copy's plane row is made unreachable like a
// row past N), and the rescoring wave lists the copies of the few winners that have any: copybits[rep] has bit b set when a
// copy lies in columns [b blk, (b + 1) blk) -- the wave reads rep[] only there (mine_copies).
template <int D>
__global__ __launch_bounds__(256) void mine_items_kernel(const float* __restrict__ v, const float* __restrict__ nv,
                                                         const float* __restrict__ lqn, const int32_t* __restrict__ colfirst,
                                                         int64_t N, int64_t Nq, float sigma, unsigned short* __restrict__ plane,
                                                         unsigned* __restrict__ maxima, int32_t* __restrict__ rep,
                                                         unsigned long long* __restrict__ copybits, int32_t* __restrict__ lastcopy, int blk_shift, int abl,
                                                         int item_blocks, MineUserFrags uf) {
    if ((int)blockIdx.x >= item_blocks) {                     // (the same launch converts the users' rows: blocks behind the items')
        mine_user_frags<D>(uf, (int64_t)(blockIdx.x - item_blocks) * 256 + threadIdx.x);
        return;
    }
    constexpr int LPR = D / 8, RW = 2 * D + 16;
    const int lane = mf_lane();
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = t / LPR;
    const int c = (int)(t % LPR);
    const bool live = r < Nq;                                 // (whole groups: Nq LPR is a multiple of 64)
    const bool real = live && r < N;
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // everything that depends on r alone is asked for at once: the row, its first column by item id, its norm and logQ
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    int64_t cf = r;
    float nvr = 0.f, lqr = 0.f;
    if (real) {
        a = reinterpret_cast<const f32x4*>(v + r * D)[2 * c];
        b = reinterpret_cast<const f32x4*>(v + r * D)[2 * c + 1];
        if (colfirst) cf = (int64_t)colfirst[r];
        if (c == 0) { nvr = nv[r]; lqr = lqn[r]; }
    }
    u16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
    bool same = false;
    int64_t f = r;
    if (real) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = i < 4 ? a[i] : b[i - 4];
            hi[i] = mbf_round(x);
            lo[i] = mbf_round(x - mbf_float(hi[i]));
        }
        if (cf >= 0 && cf < r) {                              // an earlier column carries the same item id: the same row?
            const u32x4 fa = reinterpret_cast<const u32x4*>(v + cf * D)[2 * c], fb = reinterpret_cast<const u32x4*>(v + cf * D)[2 * c + 1];
            const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
            same = ua[0] == fa[0] && ua[1] == fa[1] && ua[2] == fa[2] && ua[3] == fa[3] && ub[0] == fb[0] && ub[1] == fb[1] && ub[2] == fb[2] && ub[3] == fb[3];
            // (... and the same logQ: a table looked up by item id gives copies the same value, a per-column tensor need not)
            if (c == 0) same = same && __builtin_bit_cast(unsigned, lqn[cf]) == __builtin_bit_cast(unsigned, lqr);
            f = cf;
        }
    }
    const unsigned long long gm = (LPR >= 64 ? ~0ull : ((1ull << LPR) - 1ull)) << (lane & ~(LPR - 1));
    const bool dup = (__ballot(same) & gm) == gm;             // every chunk of the row equals the first copy's
    float mx_nv = 0.f, mx_q = 0.f;
    if (live) {
        unsigned short* row = plane + r * RW;
        *reinterpret_cast<u16x8*>(row + 8 * c) = hi;
        *reinterpret_cast<u16x8*>(row + D + 8 * c) = lo;
        if (c == 0) {
            u16x8 a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
            if (real && !dup) {
                const float w = (0.5f * sigma) * nvr;
                unsigned short ws[3], qs[3];
                mbf_split3(w, ws);
                mbf_split3(lqr, qs);
                const unsigned short one = 0x3F80;
                a0 = u16x8{ws[0], ws[1], ws[2], qs[0], qs[1], qs[2], one, one};
                a1[0] = one;
                mx_nv = nvr;
                mx_q = fabsf(lqr);
            } else {
                a0[3] = 0x7E00;      // rows past N and copies: 2^125 in the first logQ slot -- their accumulators never pass |acc| <= half
            }
            *reinterpret_cast<u16x8*>(row + 2 * D) = a0;
            *reinterpret_cast<u16x8*>(row + 2 * D + 8) = a1;
            rep[r] = (int32_t)(dup ? f : r);
#ifdef MF_BF3_LAB
            if (dup && !(abl & 32)) {
#else
            if (dup) {
#endif
                // (fire and forget: a popular item's copies queue up at its representative's two words either way -- they all run at once)
                atomicOr(copybits + f, 1ull << (unsigned)(r >> blk_shift));
                atomicMax(lastcopy + f, (int32_t)r);
            }
        }
    }
    // the block's maxima through LDS, then ONE pair of atomics per block, dealt over MBF_MAXSLOTS pairs of words a cache line
    // apart (atomics on one line queue up at its L2 channel: 4096 waves on one pair of words were 95 of this kernel's 100 us;
    // non-negative floats order like their bits, NaN above everything: it reaches the bound)
    __shared__ unsigned blk_max[2][4];
    unsigned bn = __builtin_bit_cast(unsigned, mx_nv), bq = __builtin_bit_cast(unsigned, mx_q);
    bn = mf_wave_max_u32(bn);
    bq = mf_wave_max_u32(bq);
    if (lane == 0) { blk_max[0][threadIdx.x >> 6] = bn; blk_max[1][threadIdx.x >> 6] = bq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int slot = (int)(blockIdx.x & (MBF_MAXSLOTS - 1)) * MBF_MAXSTRIDE;
        atomicMax(maxima + slot, max(max(blk_max[0][0], blk_max[0][1]), max(blk_max[0][2], blk_max[0][3])));
        atomicMax(maxima + slot + 1, max(max(blk_max[1][0], blk_max[1][1]), max(blk_max[1][2], blk_max[1][3])));
    }
}

// --------------------------------------------- bound, interval, row constants ----
// One wave per user, behind the fp32 seeding pass: its starting bound from its seed keys exactly as select_bound_kernel
// (mf_select.h) finds it -- every lane reduces its share of the seeds to its best, the k-th largest of the 64 lane maxima by rank
// counting -- and, in the same launch, what the scan needs of it: the interval of Dm the bound admits
// (Policy::make_thr), its centre folded into rho, the error bound eps of the header, half = half width + eps.
struct MineBound {
    const unsigned long long* seeds;
    int seeds_per_row, k;
    const float *nu, *lii, *sgn;
    const unsigned* maxima;
    int64_t B, Xq;
    float sigma;
    unsigned* gtau;          // [Bp]: the rank bound (the fp32 fallback search starts from it)
    f32x4* rowk;             // [Xq]: {half, rho, s, eps}
    int32_t* rowflag;        // [Xq]: 1 = a zero target: the rescoring wave walks the row (Dm does not depend on the embeddings)
    int32_t* gate;           // set: this batch is not for the prefilter (the fp32 search runs behind it)
    unsigned long long* dbg; // lab: [3] += no bound, [4] += non-finite, [5] += zero targets (NULL: off)
};
template <int D, class Policy>
__global__ __launch_bounds__(256) void mine_bound_kernel(MineBound p) {
    __shared__ unsigned lane_best[4][64];
    const int lane = mf_lane(), wv = threadIdx.x >> 6;
    const int64_t x = (int64_t)blockIdx.x * 4 + wv;           // (the grid covers Xq: a multiple of 4)
    const bool real = x < p.B;
    unsigned best = 0u;
    if (real) {
        const unsigned long long* src = p.seeds + x * (int64_t)p.seeds_per_row;
        for (int i = lane; i < p.seeds_per_row; i += 64) best = max(best, (unsigned)(src[i] >> 32));
    }
    lane_best[wv][lane] = best;
    __syncthreads();
    int above = 0;                    // lanes whose maximum beats mine (ties broken by lane: a strict order)
    for (int q = 0; q < 64; ++q) {
        const unsigned o = lane_best[wv][q];
        above += (o > best || (o == best && q < lane)) ? 1 : 0;
    }
    const bool is_kth = real && above == p.k - 1 && best != 0u;
    if (is_kth) p.gtau[x] = best;
    const unsigned long long kb = __ballot(is_kth);
    const unsigned bound = kb ? (unsigned)__shfl((int)best, __builtin_ctzll(kb), 64) : 0u;     // (0: fewer than k seeds in sight)
    // the maxima the item launch left (16 pairs of words a cache line apart; non-negative floats order like their bits, NaN above all)
    const int mi = lane & (MBF_MAXSLOTS - 1);
    const float nvmax_f = __builtin_bit_cast(float, mf_wave_max_u32(p.maxima[MBF_MAXSTRIDE * mi]));
    const float qmax_f = __builtin_bit_cast(float, mf_wave_max_u32(p.maxima[MBF_MAXSTRIDE * mi + 1]));
    if (lane != 0) return;
    float half = -1.f, rho = 0.f, eps = 0.f, s = 0.f;
    int flag = 0;
    if (real) {
        s = p.sgn[x];
        const float as = p.sigma * s;
        const double nvmax = (double)nvmax_f, Q = (double)qmax_f;
        const double nu = (double)p.nu[x], li = (double)p.lii[x], sg = (double)p.sigma, aas = fabs((double)as);
        const double P = aas * sqrt(nu * nvmax), W = 0.5 * fabs(sg) * nvmax, Hn = 0.5 * aas * nu;
        const double M = P + Hn + W + Q + fabs(li);
        const typename Policy::Thr th = Policy::make_thr(bound);
        const bool none = th.lo > th.hi;                                   // (thr_none: cannot happen for a real row; nothing passes)
        double lo_ = (double)th.lo, hi_ = (double)th.hi;
        const double big = 1.01 * M + 1e-30;
        if (!(lo_ > -big)) lo_ = -big;                                     // an open end: no Dm of this row lies beyond +-M
        if (!(hi_ < big)) hi_ = big;
        const double mid = 0.5 * (lo_ + hi_);
        const double hs = -0.5 * (double)as;
        const double rd = hs * nu - li - mid;
        rho = (float)rd;
        const double ar = fabs((double)rho);
        const double S = 1.01 * P + W + Q + ar;
        const double nt = 3.0 * D + 16.0;
        const double e = 1.01 * (nt * 0x1p-23 * S + 3.1 * 0x1p-18 * P + 0x1p-23 * (W + Q + ar) + D * 0x1p-23 * P + 0x1p-21 * M);
        eps = (float)e;
        double hf = (0.5 * (hi_ - lo_) + e) * (1.0 + 0x1p-20) + 1e-37;
        half = (float)hf;
        if ((double)half < hf) half = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, half) + 1u);     // (positive, finite: the next float up)
        if (none) half = -1.f;
        if ((bound >> 30) == 0u) { half = -1.f; *p.gate = 1; if (p.dbg) atomicAdd(p.dbg + 3, 1ull); }            // no bound (fewer than k seeds in sight): every column would pass
        if (!(hf < 1e37) || !(M < 1e37)) { half = -1.f; *p.gate = 1; if (p.dbg) atomicAdd(p.dbg + 4, 1ull); }        // non-finite inputs
        if (s == 0.f) { half = -1.f; flag = 1; if (p.dbg) atomicAdd(p.dbg + 5, 1ull); }                           // a zero target: Dm does not depend on the embeddings -- the exact walk is cheap
    }
    p.rowk[x] = f32x4{half, rho, s, eps};
    p.rowflag[x] = flag;
}

// ------------------------------------------------------------------ scan ----
struct MineScan {
    const unsigned short* plane;     // [Nq][2 d + 16]
    int64_t Nq;
    int NT, tpc;
    const mbf16x8* ufrag;
    const f32x4* rowk;
    int64_t Xq;
    uint32_t* plist;                 // [nchunk][Xq][lcap]: columns
    int lcap;                        // 16 lpc
    uint32_t* pcnt;                  // [nchunk][Xq]
    uint32_t* spill;                 // [Xq][MBF_SPILL]: a user's hits beyond its lists
    int32_t* spill_cnt;              // [Xq], zeroed by the host
    int32_t* gate;                   // set when a spill list overflows too
    unsigned long long* dbg;         // lab: [6] += entries spilled, [7] = max half (bits) of spilling rows
    int abl;                         // lab (MF_MBF_ABL, -DMF_BF3_LAB builds): 1 = hits are not stored, 2 = no compares, 4 = no MFMAs, 8 = no barriers, 64 = no LDS operand reads, 128 = no staging -- wrong results, timing only
};
template <int D>
struct MineLds {
    static constexpr int ROWB = 2 * D;                        // bytes of a bf16 row of one part
    static constexpr int SUB = 32 * ROWB;                     // hi (and lo) sub-tile
    static constexpr int TILEB = 2 * SUB + 1024;              // + the 16 augmented slots of the 32 rows
    static constexpr int PIECES = TILEB / 1024;               // 17 (d = 128), 9
    static constexpr int HP = SUB / 1024;                     // pieces of a sub-tile
    static constexpr int PPW = (PIECES + MBF_WAVES - 1) / MBF_WAVES;
    static constexpr int CPR = ROWB / 16;
    static __device__ __forceinline__ int swz(int row) { return CPR >= 16 ? (row & 15) : ((row >> 1) & 7); }
    static constexpr int NS = 2 * MBF_BLOCK;
    static constexpr int RING = NS * TILEB;
    static constexpr int DUMP0 = RING;
    static constexpr int HB0 = RING + 1024;                   // per wave: MBF_HCAP hit words, then 64 list cursors
    static constexpr int HBW = MBF_HCAP * 4 + 64 * 4;
    static constexpr int BYTES = HB0 + MBF_WAVES * HBW;
};

template <int D>
__global__ __launch_bounds__(64 * MBF_WAVES, 2) void mine_scan_kernel(MineScan p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = MineLds<D>;
    constexpr int KS = D / 16, NS = L::NS, RW = (2 * D + 16) * 2;
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
    const int64_t x0 = (int64_t)blockIdx.y * MBF_ROWS_WG + (int64_t)wave * (32 * MBF_XT);
    const int chunk = blockIdx.x;
    const int t0 = chunk * p.tpc, t1 = min(p.NT, t0 + p.tpc);
    const int nv = t1 - t0;

    const int64_t row0 = (int64_t)t0 * 32;
    mf_rsrc_t trsrc;
    unsigned toff[L::PPW];
    {
        int64_t bytes = (p.Nq - row0) * (int64_t)RW;
        bytes = bytes < 0 ? 0 : (bytes > (int64_t)MF_SRD_MAX_BYTES ? (int64_t)MF_SRD_MAX_BYTES : bytes);
#if defined(__HIP_DEVICE_COMPILE__)
        trsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.plane + row0 * (2 * D + 16)), 0, (int)(unsigned)bytes, 0x00020000);
#else
        (void)trsrc;
#endif
#pragma unroll
        for (int q = 0; q < L::PPW; ++q) {
            const int piece = wave + q * MBF_WAVES;
            unsigned o = MF_SRD_DEAD;
            if (piece < 2 * L::HP) {
                const int part = piece / L::HP;                               // 0 hi, 1 lo
                const int off = (piece - part * L::HP) * 1024 + lane * 16;
                const int row = off / L::ROWB;
                const int ch = ((off % L::ROWB) >> 4) ^ L::swz(row);
                o = (unsigned)(row * RW + part * L::ROWB + ch * 16);
            } else if (piece == 2 * L::HP) {
                o = (unsigned)((lane >> 1) * RW + 2 * L::ROWB + (lane & 1) * 16);
            }
            toff[q] = o;
            asm volatile("" : "+v"(toff[q]));
        }
    }
    auto stage = [&](int v) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef MF_BF3_LAB
        if (p.abl & 128) return;                             // (lab: nothing is staged)
#endif
        char* slot = smem + (v % NS) * L::TILEB;
        const int soff = v < nv ? (int)((int64_t)v * 32 * RW) : (int)MF_SRD_DEAD;
#pragma unroll
        for (int q = 0; q < L::PPW; ++q) {
            const int piece = wave + q * MBF_WAVES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(trsrc, (mf_lds_ptr)(piece < L::PIECES ? slot + piece * 1024 : smem + L::DUMP0), 16, (int)toff[q], soff, 0, 0);
        }
#else
        (void)v;
#endif
    };
#pragma unroll
    for (int j = 0; j < MBF_BLOCK; ++j) stage(j);

    mbf16x8 ubh[MBF_XT][KS], ubl[MBF_XT][KS], uaug[MBF_XT];
    float half[MBF_XT];
    // Hits go to a per-wave buffer in LDS ({lane, user tile} << 24 | column: N < 2^24), appended behind a wave-uniform cursor; the
    // wave deals them to its users' lists after the loop.  (Round 4, first version: one global store per hit from inside the loop --
    // stores share vmcnt with the tile DMA, so every meeting point also waited for the last store's acknowledgement: 117 us
    // with them, 79 without.)
    uint32_t* hbuf = reinterpret_cast<uint32_t*>(smem + L::HB0 + wave * L::HBW);
    int hn = 0;
#pragma unroll
    for (int xt = 0; xt < MBF_XT; ++xt) {
        const int64_t x = x0 + 32 * xt + c;
        const mbf16x8* xf = p.ufrag + ((x0 / 32 + xt) * (2 * KS)) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) { ubh[xt][s] = xf[s * 64]; ubl[xt][s] = xf[(KS + s) * 64]; }
        const f32x4 rk = p.rowk[x];
        half[xt] = rk[0];
        unsigned short r3[3];
        mbf_split3(rk[1], r3);
        const unsigned short one = 0x3F80, ms = mbf_round(-rk[2]);
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
        const u16x8 a = h == 0 ? u16x8{ms, ms, ms, one, one, one, r3[0], r3[1]} : u16x8{r3[2], 0, 0, 0, 0, 0, 0, 0};
        uaug[xt] = __builtin_bit_cast(mbf16x8, a);
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int v = 0; v < nv; ++v) {
        if ((v & (MBF_BLOCK - 1)) == 0) {
            mf_wait_vmcnt<0>();
#ifdef MF_BF3_LAB
            if (!(p.abl & 8))
#endif
            mf_block_barrier();
            if (v + MBF_BLOCK < nv) {
#pragma unroll
                for (int j = 0; j < MBF_BLOCK; ++j) stage(v + MBF_BLOCK + j);
            }
        }
        const char* slot = smem + (v % NS) * L::TILEB;
        const char* rowp = slot + c * L::ROWB;
        const int sw = L::swz(c);
        mbf16x8 ahi[KS], alo[KS];
        mbf16x8 aaug;
#ifdef MF_BF3_LAB
        if (p.abl & 64) {                                    // (lab: the operands stay what they were -- no LDS reads)
#pragma unroll
            for (int s = 0; s < KS; ++s) { ahi[s] = ubh[0][s]; alo[s] = ubl[0][s]; }
            aaug = uaug[0];
        } else
#endif
        {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ahi[s] = *reinterpret_cast<const mbf16x8*>(rowp + (((2 * s + h) ^ sw) << 4));
                alo[s] = *reinterpret_cast<const mbf16x8*>(rowp + L::SUB + (((2 * s + h) ^ sw) << 4));
            }
            aaug = *reinterpret_cast<const mbf16x8*>(slot + 2 * L::SUB + c * 32 + h * 16);
        }
        const unsigned col0 = (unsigned)(t0 + v) * 32u + 4u * (unsigned)h;
#pragma unroll
        for (int xt = 0; xt < MBF_XT; ++xt) {
            f32x16 acc = zero16;
#ifdef MF_BF3_LAB
            if (!(p.abl & 4))
#endif
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[s], ubh[xt][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[s], ubh[xt][s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[s], ubl[xt][s], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aaug, uaug[xt], acc, 0, 0, 0);
            // one compare per element into a wave mask; the masks are OR-ed on the scalar unit, and only a tile with a hit
            // (tens of columns per user in all) looks at them one by one
            unsigned long long m[16], any = 0ull;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                m[e] = __ballot(fabsf(acc[e]) <= half[xt]);
                any |= m[e];
            }
#ifdef MF_BF3_LAB
            if (p.abl & 2) { any = 0ull; asm volatile("" :: "v"(acc)); }
            if (p.abl & 1) any = 0ull;
#endif
            if (any) {
                // (the masks are laundered through the scalar file so that the skips stay SCALAR branches on "some lane hit":
                // left to itself the compiler re-derives each mask from its compare and emits two vector-side branches per element)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned long long gq = m[4 * q] | m[4 * q + 1] | m[4 * q + 2] | m[4 * q + 3];
                    asm volatile("" : "+s"(gq));
                    if (gq) {
#pragma unroll
                        for (int e = 4 * q; e < 4 * q + 4; ++e) {
                            unsigned long long me = m[e];
                            asm volatile("" : "+s"(me));
                            if (me) {
                                const int pos = hn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(me >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)me, 0u));
                                if (fabsf(acc[e]) <= half[xt] && pos < MBF_HCAP) {
                                    const unsigned colv = col0 + (unsigned)((e & 3) + 8 * (e >> 2));
                                    asm volatile("ds_write_b32 %0, %1" ::"v"(mf_lds_addr(hbuf + pos)), "v"(colv | ((unsigned)(lane | (xt << 6)) << 24)) : "memory");
                                }
                                hn += __popcll(me);
                            }
                        }
                    }
                }
            }
        }
    }
    // deal the buffered hits to the users' lists of this chunk: a user belongs to this wave alone, so the list cursors are LDS
    // counters (one returning LDS atomic per hit), and the lengths go out as plain stores -- nothing to clear between batches
    {
        uint32_t* cur = hbuf + MBF_HCAP;
        asm volatile("ds_write_b32 %0, %1" ::"v"(mf_lds_addr(cur + lane)), "v"(0u) : "memory");
        mf_wave_sync();
        if (hn > MBF_HCAP) { *p.gate = 1; hn = MBF_HCAP; }
        for (int i0 = 0; i0 < hn; i0 += 64) {
            if (i0 + lane < hn) {
                const uint32_t ent = hbuf[i0 + lane];
                const int who = (int)(ent >> 24);                                  // lane | user tile << 6
                const int ul = (who & 31) + 32 * (who >> 6);                       // the user among the wave's 64
                const uint32_t colv = ent & 0xFFFFFFu;
                const int at = (int)atomicAdd(cur + ul, 1u);
                const int64_t xr = x0 + ul;
                if (at < p.lcap) {
                    p.plist[((int64_t)chunk * p.Xq + xr) * p.lcap + at] = colv;
                } else {                                                           // (rare: the user's spill list, behind a global cursor)
                    const int sa = atomicAdd(p.spill_cnt + xr, 1);
                    if (sa < MBF_SPILL) p.spill[xr * MBF_SPILL + sa] = colv;
                    else *p.gate = 1;
                    if (p.dbg) { atomicAdd(p.dbg + 6, 1ull); atomicMax(p.dbg + 7, (unsigned long long)hn); }
                }
            }
        }
        mf_wave_sync();
        p.pcnt[(int64_t)chunk * p.Xq + x0 + lane] = min(cur[lane], (uint32_t)p.lcap);
    }
}

// --------------------------------------------------------------- rescore ----
struct MineRescore {
    const float *u, *v, *nu, *nv, *lii, *sgn, *lqn;
    const uint32_t* maskW;
    int64_t B, Bp, N, Xq;
    float sigma;
    int nlists, k;
    int lpc, keys_cap;               // lanes per chunk; capacity of keys[] (16 nlists + MBF_SPILL + 8)
    const uint32_t* plist;
    const uint32_t* pcnt;
    const int32_t* rowflag;
    const uint32_t* spill;
    const int32_t* spill_cnt;
    const int32_t* gate;
    const int32_t* rep;
    const unsigned long long* copybits;
    const int32_t* lastcopy;
    int blk;
    unsigned long long* cand;
    int32_t* cand_cnt;
    int rowcap;
    unsigned long long* dbg;         // lab: [0] += candidates rescored, [1] += rows, [2] += rows walked exactly (NULL: off)
};
template <int D>
struct MineRescoreGeom {
    static constexpr int CPR = D / 4;                   // 16-byte chunks of an fp32 row
    static constexpr int RPI = 64 / CPR;                // rows per DMA instruction
    static constexpr int RB = 8;                        // candidates per round (a user has ~20: three rounds; LDS is this kernel's occupancy limit -- 32 per round: 69 us)
    static constexpr int NI = RB / RPI;
    static constexpr int LMAX = MBF_MAXLISTS * MBF_CAPL + MBF_SPILL;
    static constexpr int keys_cap(int nlists) { return nlists * MBF_CAPL + MBF_SPILL + 8; }      // (+ the stand-in of a masked diagonal)
    static constexpr int bytes(int nlists) { return RB * D * 4 + keys_cap(nlists) * 8 + 2 * 64 * 8 + D * 4; }      // rows | keys | win, sorted | the user's row
};

// the exact key of (user, column) from the chain product -- MiningPolicy::key, word for word
__device__ __forceinline__ unsigned long long mine_exact_key(float nu, float nvj, float dot, float s, float sigma, float lqnj, float lii, unsigned col) {
    const float Lg = mf_logit(nu, nvj, dot, s, sigma, -lqnj);
    return mf_key_mining(Lg - lii, col);
}
__device__ __forceinline__ unsigned long long mine_shfl_u64(unsigned long long v, int src) {
    return ((unsigned long long)(unsigned)__shfl((int)(v >> 32), src, 64) << 32) | (unsigned long long)(unsigned)__shfl((int)(unsigned)v, src, 64);
}
// The copies of representative column f behind column `from`, in column order, by the whole wave: emit(col, i) for the i-th
// one that is a valid negative of user x (a copy shares its representative's item id, hence its hits: only the user's own
// diagonal can differ), i < want.  `bits`: copybits[f].  Returns min(number of valid copies behind `from`, want).
template <class Emit>
__device__ __forceinline__ int mine_copies(const MineRescore& p, int64_t x, unsigned f, unsigned from, unsigned long long bits, int want, Emit emit) {
    const int lane = mf_lane();
    const unsigned long long below = (1ull << lane) - 1ull;
    int got = 0;
    bits &= ~0ull << (from / (unsigned)p.blk);               // blocks from `from`'s on
    while (bits && got < want) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1;
        for (int64_t base = (int64_t)b * p.blk; base < (int64_t)(b + 1) * p.blk && base < p.N && got < want; base += 256) {
            const int64_t c0 = base + 4 * lane;
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            i32x4 rv = {-1, -1, -1, -1};
            if (c0 < (p.N + 31) / 32 * 32) rv = *reinterpret_cast<const i32x4*>(p.rep + c0);      // (rep[] covers the padded last tile)
            unsigned mm = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t col = c0 + q;
                if (col < p.N && col > (int64_t)from && (unsigned)rv[q] == f && col != x) mm |= 1u << q;
            }
            int pre = 0, tot = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long bal = __ballot((mm >> q) & 1u);
                pre += __popcll(bal & below);
                tot += __popcll(bal);
            }
            int i = got + pre;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((mm >> q) & 1u) {
                    if (i < want) emit((unsigned)(c0 + q), i);
                    ++i;
                }
            got = min(want, got + tot);
        }
    }
    return got;
}

// One wave (= one workgroup) per user (grid: the padded user count).  The wave also FINISHES its user -- Fin::run: the selected
// columns, their logits, the row's statistics (MinedRowFinish in mf_loss.hip; mined_rows_kernel's second half) --: the row lists
// of the fp32 search (cand / cand_cnt) are not written on this path at all.
template <int D, class Fin>
__global__ __launch_bounds__(64) void mine_rescore_kernel(MineRescore p, typename Fin::Params fp) {
    using G = MineRescoreGeom<D>;
    extern __shared__ __attribute__((aligned(1024))) char fsm[];
    const int lane = mf_lane();
    float* rows_lds = reinterpret_cast<float*>(fsm);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(fsm + G::RB * D * 4);      // columns first, their keys later
    unsigned long long* win = keys + p.keys_cap;
    unsigned long long* sorted = win + 64;
    float* xq = reinterpret_cast<float*>(sorted + 64);
    const int64_t x = blockIdx.x;
    // (the fp32 search behind this kernel fills the lists -- cand_cnt was cleared by the host -- and mined_rows_kernel finishes)
    if (__hip_atomic_load(p.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (x >= p.B) { Fin::run(fp, x, 0, sorted); return; }    // padding users: empty statistics
    const unsigned long long below = (1ull << lane) - 1ull;
    // round trip 1: the user's scalars and row, the lengths of its lists (lane l owns list l = 2 chunk + lane half)
    const float nu = p.nu[x], lii = p.lii[x], s = p.sgn[x];
    const int flag = p.rowflag[x];
    const int nsp = min(p.spill_cnt[x], MBF_SPILL);
    // (a user's list of chunk c is taken by lanes lpc c .. lpc c + lpc - 1, sixteen entries each)
    const int lch = lane / p.lpc, lsub = lane % p.lpc;
    const int64_t lbase = ((int64_t)lch * p.Xq + x) * p.lpc + lsub;
    int nl = (lane < p.nlists && !flag) ? (int)p.pcnt[(int64_t)lch * p.Xq + x] - MBF_CAPL * lsub : 0;
    nl = nl < 0 ? 0 : (nl > MBF_CAPL ? MBF_CAPL : nl);
    float xv[(D + 63) / 64];
#pragma unroll
    for (int j = 0; j < (D + 63) / 64; ++j) xv[j] = lane + 64 * j < D ? p.u[x * D + lane + 64 * j] : 0.f;
#pragma unroll
    for (int j = 0; j < (D + 63) / 64; ++j)
        if (lane + 64 * j < D) xq[lane + 64 * j] = xv[j];
    int n_out = 0;
    if (!flag) {
        // round trip 2: every list whole (a lane's 16 entries = 64 bytes), all loads in flight; 3: their mask words
        unsigned col[MBF_CAPL];
        {
            const uint4* src = reinterpret_cast<const uint4*>(p.plist + lbase * MBF_CAPL);
#pragma unroll
            for (int j = 0; j < MBF_CAPL / 4; ++j) {
                uint4 e = {0u, 0u, 0u, 0u};
                if (4 * j < nl) e = src[j];
                col[4 * j] = e.x; col[4 * j + 1] = e.y; col[4 * j + 2] = e.z; col[4 * j + 3] = e.w;
            }
        }
        // every listed column into keys[] (positions from the list lengths alone: no memory round trip in between)
        int n = 0;
#pragma unroll
        for (int j = 0; j < MBF_CAPL; ++j) {
            if (__any(j < nl)) {                             // (wave-uniform)
                const bool in = j < nl;
                const unsigned long long bal = __ballot(in);
                if (in) keys[n + __popcll(bal & below)] = (unsigned long long)col[j];
                n += __popcll(bal);
            }
        }
        {   // the spill list (almost always empty)
            for (int i0 = 0; i0 < nsp; i0 += 64) {
                const bool in = i0 + lane < nsp;
                const unsigned cs = in ? p.spill[x * MBF_SPILL + i0 + lane] : 0u;
                const unsigned long long bal = __ballot(in);
                if (in) keys[n + __popcll(bal & below)] = (unsigned long long)cs;
                n += __popcll(bal);
            }
        }
        mf_row_topk_sync<true>();
        if (p.dbg && lane == 0) { atomicAdd(p.dbg, (unsigned long long)n); atomicAdd(p.dbg + 1, 1ull); }
        // round trip 3, per round of 32 columns, everything in flight together: the fp32 rows (LDS-DMA), and for the lane's own
        // column its mask word, norm, logQ and copy bitmap.  A masked column (hit, diagonal, padding) gets key 0 = no key.
        const int pz = lane % G::CPR, sub = lane / G::CPR;
        const int sl = lane & (G::RB - 1);
        const float* row_l = rows_lds + sl * D;
        const int sw = sl & 15;
        unsigned long long kdiag = 0ull, cbdiag = 0ull;       // the user's own diagonal, if it is listed: its key and copy bitmap
        bool anycb = false;                                  // some valid column has copies (then the winners look theirs up again)
        for (int base = 0; base < n; base += G::RB) {
            const int nr = min(G::RB, n - base);
            const bool have = lane < nr;
            const unsigned cj = have ? (unsigned)keys[base + lane] : 0u;
            __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the columns have arrived
#pragma unroll
            for (int t = 0; t < G::NI; ++t) {
                if (t * G::RPI < nr) {                       // (wave-uniform)
                    unsigned rr = (unsigned)__builtin_amdgcn_readlane((int)cj, G::RPI * t);
#pragma unroll
                    for (int j = 1; j < G::RPI; ++j) {
                        const unsigned rj = (unsigned)__builtin_amdgcn_readlane((int)cj, G::RPI * t + j);
                        rr = sub == j ? rj : rr;
                    }
                    if ((int64_t)rr >= p.N) rr = 0u;         // (never listed; a guard for the address only)
                    const int ch = pz ^ ((G::RPI * t + sub) & 15);
                    const float* srcp = p.v + (int64_t)rr * D + 4 * ch;
                    __builtin_amdgcn_global_load_lds((mf_glb_ptr)srcp, (mf_lds_ptr)(rows_lds + t * 256), 16, 0, 0);
                }
            }
            const bool inr = have && (int64_t)cj < p.N;
            const float nvj = inr ? p.nv[cj] : 0.f, lqj = inr ? p.lqn[cj] : 0.f;
            const uint32_t mwj = inr ? p.maskW[(int64_t)(cj >> 5) * p.Bp + x] : ~0u;
            const unsigned long long cbj = inr ? p.copybits[cj] : 0ull;
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
            asm volatile("" ::: "memory");
            float acc = 0.f;
#pragma unroll 4
            for (int g = 0; g < D / 8; ++g) {                // k order of mf_dot_chain
                const f32x4 a = *reinterpret_cast<const f32x4*>(row_l + (((2 * g) ^ sw) << 2));
                const f32x4 b = *reinterpret_cast<const f32x4*>(row_l + (((2 * g + 1) ^ sw) << 2));
                const f32x4 xa = *reinterpret_cast<const f32x4*>(xq + 8 * g), xb = *reinterpret_cast<const f32x4*>(xq + 8 * g + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc = __builtin_fmaf(xa[t], a[t], acc);
                    acc = __builtin_fmaf(xb[t], b[t], acc);
                }
            }
            const unsigned long long kj = mine_exact_key(nu, nvj, acc, s, p.sigma, lqj, lii, cj);
            const bool valid = inr && !((mwj >> (cj & 31)) & 1u);
            if (have) keys[base + lane] = valid ? kj : 0ull;
            anycb = anycb || (valid && cbj != 0ull);
            if (inr && (int64_t)cj == x && !valid && cbj != 0ull) { kdiag = kj; cbdiag = cbj; }
            mf_row_topk_sync<true>();                        // the rows are consumed before the next round's DMA lands
        }
        // The user's own diagonal passed the scan (Dm = 0 is in every semi-hard interval).  If it is a representative with copies
        // and masked ONLY as the diagonal (a positive that is not on the user's list: its last copy, which shares the item id,
        // is not masked), the copies are valid negatives and the first of them stands in for it -- same row, same Dm.
        {
            const unsigned long long ob = __ballot(kdiag != 0ull);
            if (ob) {
                const int src_l = __builtin_ctzll(ob);
                const unsigned long long kd = mine_shfl_u64(kdiag, src_l), cbd = mine_shfl_u64(cbdiag, src_l);
                const unsigned lc = (unsigned)p.lastcopy[x];
                if (!((p.maskW[(int64_t)(lc >> 5) * p.Bp + x] >> (lc & 31)) & 1u))
                    n += mine_copies(p, x, (unsigned)x, (unsigned)x, cbd, 1, [&](unsigned cc, int) {
                        keys[n] = (kd & ~0x3FFFFFFFull) | (unsigned long long)(0x3FFFFFFFu - cc);
                        anycb = true;
                    });
                mf_row_topk_sync<true>();
            }
        }
        // the k best representatives; then, behind each winner that has copies, the copies a cut at k could still reach
        int m;
        if (n <= 64) m = mf_row_topk<1, true>(keys, n, p.k, win, sorted);
        else if (n <= 256) m = mf_row_topk<4, true>(keys, n, p.k, win, sorted);
        else m = mf_row_topk<G::LMAX / 64, true>(keys, n, p.k, win, sorted);
        unsigned long long kt_l = 0ull, bits_l = 0ull;
        unsigned fr_l = 0u;
        if (lane < m) kt_l = sorted[lane];
        if (__any(anycb)) {                                  // (LDS is the limit of this kernel's occupancy: no table of bitmaps, one more L2 round trip here)
            if (lane < m) {
                const unsigned cw = mf_key_mining_col(kt_l);
                fr_l = (unsigned)p.rep[cw];                  // (the column itself, or -- a stand-in -- its representative)
                bits_l = p.copybits[cw];                     // (both loads in flight; a stand-in asks again)
                if (fr_l != cw) bits_l = p.copybits[fr_l];
            }
        }
        unsigned long long todo = __ballot(bits_l != 0ull && lane < p.k - 1);      // a copy of the winner at position t sits at t + 1 or later
        if (todo) {
            // the pool: the winners and, behind each, the copies a cut at k can reach (<= k (k + 1) / 2 keys, in keys[]: the
            // candidates' keys are not needed any more); its k best, in order, are the user's selection
            mf_row_topk_sync<true>();
            if (lane < m) keys[lane] = kt_l;
            int n_pool = m;
            while (todo) {
                const int t = __builtin_ctzll(todo);
                todo &= todo - 1;
                const unsigned long long kt = mine_shfl_u64(kt_l, t), bits = mine_shfl_u64(bits_l, t);
                const unsigned fr = (unsigned)__shfl((int)fr_l, t, 64);
                const int at = n_pool;
                n_pool += mine_copies(p, x, fr, mf_key_mining_col(kt), bits, p.k - 1 - t, [&](unsigned cc, int i) {
                    keys[at + i] = (kt & ~0x3FFFFFFFull) | (unsigned long long)(0x3FFFFFFFu - cc);
                });
            }
            mf_row_topk_sync<true>();
            if (n_pool <= 64) m = mf_row_topk<1, true>(keys, n_pool, p.k, win, sorted);
            else m = mf_row_topk<G::LMAX / 64, true>(keys, n_pool, p.k, win, sorted);
        }
        n_out = m;
    } else {
        // a zero target: the whole row by the exact formulas (no product needed), 64 columns a round; the k best keys so far ride along in win[]
        if (p.dbg && lane == 0) atomicAdd(p.dbg + 2, 1ull);
        mf_row_topk_sync<true>();
        int carry = 0;
        for (int64_t base = 0; base < p.N; base += 64) {
            const int64_t cj = base + lane;
            unsigned long long v0 = 0ull;
            if (cj < p.N && !((p.maskW[(cj >> 5) * p.Bp + x] >> (cj & 31)) & 1u)) {
                v0 = mine_exact_key(nu, p.nv[cj], 0.f, s, p.sigma, p.lqn[cj], lii, (unsigned)cj);      // (s = 0: fma(0, dot, c) = c for every finite dot)
            }
            const unsigned long long v1 = lane < carry ? win[lane] : 0ull;
            mf_row_topk_sync<true>();
            const int have = __popcll(__ballot(v0 != 0ull)) + carry;
            unsigned long long tau = 1ull;
            if (have > p.k) {
                unsigned long long th = 0ull;
                for (int b = 63; b >= 0; --b) {
                    const unsigned long long cnd = th | (1ull << b);
                    const int cge = __popcll(__ballot(v0 >= cnd)) + __popcll(__ballot(v1 >= cnd));
                    if (cge >= p.k) th = cnd;
                }
                tau = th;
            }
            int pos = 0;
            {
                const bool w0 = v0 != 0ull && v0 >= tau;
                const unsigned long long m0 = __ballot(w0);
                if (w0) win[__popcll(m0 & below)] = v0;
                pos = __popcll(m0);
                const bool w1 = v1 != 0ull && v1 >= tau;
                const unsigned long long m1 = __ballot(w1);
                if (w1) win[pos + __popcll(m1 & below)] = v1;
                pos += __popcll(m1);
            }
            carry = pos;
            mf_row_topk_sync<true>();
        }
        unsigned long long mine_k = 0ull;
        if (lane < carry) mine_k = win[lane];
        mf_row_topk_sync<true>();
        if (lane < carry) {
            int rk = 0;
            for (int q = 0; q < carry; ++q) rk += win[q] > mine_k ? 1 : 0;
            sorted[rk] = mine_k;
        }
        mf_row_topk_sync<true>();
        n_out = carry;
    }
    Fin::run(fp, x, n_out, sorted);
}

#endif  // __HIPCC__
