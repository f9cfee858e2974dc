// mf_loss.hip -- the in-batch score matrix and the seven xfmr_rec/losses.py losses,
// forward and backward, without ever materialising the B x N logits or the
// B x N x P positive-mask temp of the reference (xfmr_rec/losses.py:108).
//
// Pipeline (all on the caller's stream, no allocation, no sync):
//   fwd : chain norms -> diagonal (L_ii) -> sort item ids -> hit bitmask
//         (B x N BITS, both orientations) ->
//           dense : score tiles + per-row statistics (split over item ranges)
//           mined : streaming per-row top-k of mining keys -> merge -> statistics
//         -> per-row losses -> fixed-order reduction to the 7 scalars
//   bwd : per-row coefficients of dloss/dL -> dense: dU pass + dV pass (each a
//         recompute-and-contract sweep over score tiles), or mined: sparse rows.
//
// Reference restated: negative_masks :92-110, semi_hard_mining :134-162,
// alignment/contrastive/infonce/mine :164-246, pairwise :324-359, reduction
// `(loss * target.abs()).sum()`.  Formulas: SURVEY.md Appendix A.
#include <cfloat>
#include <type_traits>

#include "mf_common.h"
#include "mf_loss_math.h"
#include "mf_mine_bf.h"
#include "mf_select.h"
#include "mf_stream.h"

static int need_flags(int kind_mask) {
    int f = 0;
    if (kind_mask & ((1 << MF_CONTRASTIVE) | (1 << MF_ALIGNMENT_CONTRASTIVE))) f |= NEED_CONTR;
    if (kind_mask & ((1 << MF_INFONCE) | (1 << MF_MINE))) f |= NEED_LSE;
    if (kind_mask & (1 << MF_PAIRWISE_HINGE)) f |= NEED_HINGE;
    if (kind_mask & (1 << MF_PAIRWISE_LOGISTIC)) f |= NEED_LOGI;
    return f;
}

// ------------------------------------------------------------------ workspace --
struct LossWs {
    int64_t B, N, Bp, Np;
    int BT, NT, d;
    bool mined;
    int nsplit_f, tps_f;          // dense fwd: item-range splits
    int nsplit_u, tps_u;          // dU pass: item-range splits
    int nsplit_v, tps_v;          // dV pass: user-range splits
    int T, CAP, nchunk, tpc;      // mining select geometry (nchunk = candidate sets per row)
    int NW;                       // waves per workgroup of the sweeps (mf_nw(d))
    SelectPlan plan;
    float *nu, *nv, *lii, *dii, *sgn, *logq;      // logq: zero-padded NEGATED copy (-logq_j)
    float* tgt;                                  // fp32 copy of the targets (the forward may receive int64)
    long long* gtab;
    int M;
    int32_t *colslot, *gfirst, *colfirst;
    uint32_t* bmap;               // "some column carries an id hashing here": 2^bmbits bits, the positives' prefilter
    int bmbits;
    uint32_t* ubits;
    uint32_t* maskW;
    float *part, *stats, *rowloss, *rowc, *dpart, *rpart, *dpart_v, *rpart_v, *stash, *gstash, *blockpart;
    unsigned* ticket;
    unsigned long long *cand, *priv, *seeds;
    int32_t *cand_cnt, *sel, *sel_cnt;
    unsigned* gtau;
    float* sel_L;
    MineBfPlan mbf;              // the split-bf16 candidate search (mf_mine_bf.h); its arrays exist whenever the shape is eligible
    unsigned short* mbf_plane;
    void* mbf_ufrag;
    void* mbf_rowk;
    int32_t* mbf_flag;
    unsigned* mbf_max;
    unsigned long long* mbf_copybits;
    int32_t *mbf_lastcopy, *mbf_rep, *mbf_spill_cnt, *mbf_gate;
    uint32_t* mbf_spill;
    uint32_t* mbf_plist;
    uint32_t* mbf_pcnt;
    long long* dvfix;            // mined backward: exact fixed-point accumulator of dv [N][d]
    float* dvsc;                 // its unit for this batch: {2^E, clamp, 2^-E} (dv_fix_of, mf_loss_math.h)
    size_t total;
};

#ifndef BWD_MIN_WG
#define BWD_MIN_WG 2      // A/B knob: workgroups per CU the backward sweeps are compiled for at d = 128 ...
#define BWD_WGS 512       // ... and their grid size
#endif

static void split_geometry(int x_tiles, int y_tiles, int* nsplit, int* tps, int target_blocks) {
    int want = (target_blocks + x_tiles - 1) / x_tiles;
    if (want < 1) want = 1;
    if (want > y_tiles) want = y_tiles;
    *tps = (y_tiles + want - 1) / want;
    *nsplit = (y_tiles + *tps - 1) / *tps;
}

static bool mining_on(int num_negatives, int64_t N) { return num_negatives > 0 && num_negatives < N; }

static LossWs loss_ws(void* base, int64_t B, int64_t N, int d, int P, int num_negatives) {
    LossWs w{};
    w.B = B; w.N = N; w.d = d;
    const int XB = 32 * mf_nw(d);                                 // X rows per workgroup of the sweeps
    w.Bp = (B + XB - 1) / XB * XB; w.Np = (N + XB - 1) / XB * XB;
    w.NW = mf_nw(d);
    w.BT = (int)(w.Bp / 32); w.NT = (int)(w.Np / 32);
    w.mined = mining_on(num_negatives, N);
    // enough workgroups for two waves per SIMD: 256 of eight waves, 512 of four
    const int wgs = 2048 / w.NW;
    split_geometry(w.BT / w.NW, w.NT, &w.nsplit_f, &w.tps_f, wgs);
    split_geometry(w.BT / w.NW, w.NT, &w.nsplit_u, &w.tps_u, d == 128 ? BWD_WGS : wgs);
    split_geometry(w.NT / w.NW, w.BT, &w.nsplit_v, &w.tps_v, d == 128 ? BWD_WGS : wgs);
    const int k = num_negatives;
    w.plan = mf_select_plan(B, N, d, k);
    w.T = w.plan.T; w.CAP = w.plan.CAP; w.nchunk = w.plan.nsets; w.tpc = w.plan.tpc;
    MfArena a(base);
    w.nu = a.take<float>(w.Bp); w.nv = a.take<float>(w.Np);
    w.lii = a.take<float>(w.Bp); w.dii = a.take<float>(w.Bp); w.sgn = a.take<float>(w.Bp);
    w.logq = a.take<float>(w.Np);
    w.tgt = a.take<float>(w.Bp);
    (void)P;
    w.M = 64;
    while (w.M < 2 * w.Np) w.M *= 2;
    w.gtab = a.take<long long>((size_t)w.M);
    w.gfirst = a.take<int32_t>((size_t)w.M);
    w.colslot = a.take<int32_t>((size_t)w.Np);
    w.colfirst = a.take<int32_t>((size_t)w.Np);
    w.bmbits = 13;                                                // >= 8 bits per column, 1 KiB .. 16 KiB (every block copies it into LDS)
    while (w.bmbits < 17 && (1ll << w.bmbits) < 8 * w.Np) ++w.bmbits;
    w.bmap = a.take<uint32_t>((size_t)1 << (w.bmbits - 5));
    w.ubits = a.take<uint32_t>((size_t)HITS_PLANES * w.Np * (w.Bp / 32));        // ubits [plane][user / 32][column]
    w.maskW = a.take<uint32_t>((size_t)w.NT * w.Bp);
    w.part = a.take<float>((size_t)w.nsplit_f * NSTAT * w.Bp);
    w.stats = a.take<float>((size_t)NSTAT * w.Bp);
    w.rowloss = a.take<float>((size_t)MF_NUM_KINDS * w.Bp);
    w.rowc = a.take<float>((size_t)4 * w.Bp);
    w.blockpart = a.take<float>((size_t)MF_NUM_KINDS * (w.Bp / 64 + 1));
    w.ticket = a.take<unsigned>(4);
    w.dvsc = a.take<float>(4);
    if (w.mined) {
        w.cand = a.take<unsigned long long>((size_t)w.Bp * w.plan.rowcap);
        w.priv = a.take<unsigned long long>((size_t)w.plan.nsets * w.Bp * w.plan.CAP);
        w.seeds = a.take<unsigned long long>((size_t)w.Bp * (w.plan.seeds_per_row > 0 ? w.plan.seeds_per_row : 1));
        w.sel = a.take<int32_t>((size_t)w.Bp * KSEL_MAX);
        w.sel_cnt = a.take<int32_t>(w.Bp);
        w.sel_L = a.take<float>((size_t)w.Bp * KSEL_MAX);
        w.dvfix = a.take<long long>((size_t)w.N * d);
        w.gtau = a.take<unsigned>((size_t)w.Bp);
        w.cand_cnt = a.take<int32_t>((size_t)w.Bp);      // right behind gtau: one memset clears both
        w.mbf_max = a.take<unsigned>(MBF_MAXSLOTS * MBF_MAXSTRIDE);  // ... and these maxima
        w.mbf = mine_bf_plan(B, N, d, k);
        if (w.mbf.ok) {
            w.mbf_copybits = a.take<unsigned long long>((size_t)w.mbf.Nq);        // (cleared with the maxima, and the next)
            w.mbf_lastcopy = a.take<int32_t>((size_t)w.mbf.Nq);
            w.mbf_spill_cnt = a.take<int32_t>((size_t)w.mbf.Xq);
            w.mbf_gate = a.take<int32_t>(4);                             // (the cleared range ends here)
            w.mbf_rep = a.take<int32_t>((size_t)w.mbf.Nq);
            w.mbf_spill = a.take<uint32_t>((size_t)w.mbf.Xq * MBF_SPILL);
            w.mbf_plane = a.take<unsigned short>((size_t)w.mbf.Nq * (2 * d + 16) + 64);
            w.mbf_ufrag = a.take<char>((size_t)w.mbf.Xq * d * 4);
            w.mbf_rowk = a.take<char>((size_t)w.mbf.Xq * 16);
            w.mbf_flag = a.take<int32_t>((size_t)w.mbf.Xq);
            w.mbf_plist = a.take<uint32_t>((size_t)w.mbf.nchunk * w.mbf.Xq * w.mbf.lpc * MBF_CAPL);
            w.mbf_pcnt = a.take<uint32_t>((size_t)w.mbf.nchunk * w.mbf.Xq);
        }
        w.dpart = nullptr;
    } else {
        // the two backward sweeps keep their split partials apart: ONE launch adds up both after the second sweep
        const size_t rows = (size_t)w.nsplit_u * w.Bp, rows_v = (size_t)w.nsplit_v * w.Np;
        w.dpart = a.take<float>(rows * d);
        w.rpart = a.take<float>(rows);
        w.dpart_v = a.take<float>(rows_v * d);
        w.rpart_v = a.take<float>(rows_v);
        w.stash = a.take<float>((size_t)w.Bp * w.Np);
        w.gstash = a.take<float>((size_t)w.Bp * w.Np);
    }
    w.total = a.used();
    return w;
}

extern "C" size_t mf_loss_ws_bytes(int64_t B, int64_t N, int d, int P, int num_negatives) {
    if (B <= 0 || N < B) return 0;
    return loss_ws(nullptr, B, N, d, P, num_negatives).total;
}

// ------------------------------------------------------------- per-call setup ---
// ONE launch for everything the sweeps need that is O(B + N): chain norms of both operands, the
// diagonal (L_ii, D_ii, sign), the zero-padded logQ copy, and the clearing of the hash table, the
// per-user bit rows and the reduction ticket.  (Each of these used to be its own ~5 us launch.)
struct PrepParams {
    const float *u, *v;
    const void* target;
    const float* logq;
    const int64_t* item_idx;      // logq_rows > 0: logq is a table looked up by these ids
    int64_t logq_rows;
    int target_i64;
    int64_t B, N, Bp, Np;
    int d;
    float sigma;
    float *nu, *nv, *lii, *dii, *sgn, *wlogq, *wtgt;
    uint4* gtab; int64_t gtab16;        // fills, in 16-byte units (0 = skip)
    uint4* gfirst; int64_t gfirst16;
    uint4* ubits; int64_t ubits16;
    uint4* bmap; int64_t bmap16;
    unsigned* ticket;
};

__device__ __forceinline__ void fill16(uint4* dst, int64_t n16, unsigned pat, int64_t tid, int64_t nthreads) {
    const uint4 v = {pat, pat, pat, pat};
    for (int64_t q = tid; q < n16; q += nthreads) dst[q] = v;
}

// 64-thread workgroups: the per-row chains are serial, so the rows are spread over as many CUs as possible.
// D = 0: no rows (clears only).  The rows come in batches of 32 floats per operand (16 loads in flight per thread).
template <int D>
__global__ __launch_bounds__(64) void prep_kernel(PrepParams p) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (D > 0 && i < p.Np) {
        const bool hv = i < p.N, hu = i < p.B;                  // B <= N: a user row always has its item row
        float nvv = 0.f, nuu = 0.f, dot = 0.f;
        if (hv) {
            const f32x4* pv = reinterpret_cast<const f32x4*>(p.v + i * D);
            const f32x4* pu = reinterpret_cast<const f32x4*>(p.u + (hu ? i : 0) * D);     // (no user: row 0 is read and dropped)
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            // batches of 16 loads; at most 4 batches (64 registers x 4) are in flight at once: fully unrolled, d = 256
            // asked for 8 batches' worth of registers and spilled 396 of them
            constexpr int PREP_UNROLL = D / 32 <= 4 ? (D / 32 > 0 ? D / 32 : 1) : 2;
#pragma unroll PREP_UNROLL
            for (int g0 = 0; g0 < D / 8; g0 += 4) {
                f32x4 va[8], ua[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { va[j] = pv[2 * g0 + j]; ua[j] = pu[2 * g0 + j]; }
#pragma unroll
                for (int g = 0; g < 4; ++g) {                   // k order of mf_dot_chain
                    const f32x4 a = va[2 * g], b = va[2 * g + 1];
                    const f32x4 xa = hu ? ua[2 * g] : zero4, xb = hu ? ua[2 * g + 1] : zero4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        nvv = __builtin_fmaf(a[t], a[t], nvv);  nvv = __builtin_fmaf(b[t], b[t], nvv);
                        nuu = __builtin_fmaf(xa[t], xa[t], nuu); nuu = __builtin_fmaf(xb[t], xb[t], nuu);
                        dot = __builtin_fmaf(xa[t], a[t], dot);  dot = __builtin_fmaf(xb[t], b[t], dot);
                    }
                }
            }
        }
        p.nv[i] = nvv;
        float lq = 0.f;
        if (p.logq && hv) {
            if (p.logq_rows > 0) {
                const int64_t id = p.item_idx[i];
                lq = (id >= 0 && id < p.logq_rows) ? p.logq[id] : 0.f;
            } else {
                lq = p.logq[i];
            }
        }
        p.wlogq[i] = -lq;                 // the sweeps read -logq (one fma operand, no negation in the loop)
        if (i < p.Bp) {
            float l = 0.f, dd = 0.f, sg = 0.f;
            float tg = 0.f;
            if (hu) {
                tg = p.target_i64 ? (float)static_cast<const int64_t*>(p.target)[i] : static_cast<const float*>(p.target)[i];
                sg = mf_sign(tg);
                dd = mf_half_sqdist(nuu, nvv, dot);
                l = mf_logit(nuu, nvv, dot, sg, p.sigma, lq);
            }
            p.nu[i] = nuu; p.lii[i] = l; p.dii[i] = dd; p.sgn[i] = sg; p.wtgt[i] = tg;
        }
    }
    if (i == 0) *p.ticket = 0u;
    const int64_t nthreads = (int64_t)gridDim.x * 64;
    fill16(p.gtab, p.gtab16, 0x80808080u, i, nthreads);
    fill16(p.gfirst, p.gfirst16, 0x7f7f7f7fu, i, nthreads);
    fill16(p.ubits, p.ubits16, 0u, i, nthreads);
    fill16(p.bmap, p.bmap16, 0u, i, nthreads);
}

// ------------------------------------------------------------------ hit masks --
// maskW [tj][i]  bit c : item column 32 tj + c is NOT a valid negative of user i
//
// The reference compares every (user, column, positive) triple (B x N x P bools,
// losses.py:108).  Here:
//   1. the batch's item ids go into ONE open-addressing table (M >= 2N slots); every column
//      learns the FIRST column that carries the same item id (duplicates are common: Zipf);
//   2. each user's positives (plus its own item: the accidental-hit term, losses.py:103) are
//      looked up there and set ONE bit, in the bit-vector over users of that first column:
//      ubitsT[first][user / 32] -- positives absent from the batch cost nothing more;
//   3. the 32 columns of a tile then each fetch the user bit-vector of THEIR first column (a
//      duplicate column simply reads another row: no per-duplicate probing) and a 32 x 32 bit
//      transpose across the half-wave turns (lane = column, bit = user) into the mask words
//      (lane = user, bit = column).  No atomics in the sweep, no B x N x P temp.
static constexpr long long HT_EMPTY = (long long)0x8080808080808080ull;   // memset(0x80)

__device__ __forceinline__ unsigned ht_hash(long long id, unsigned slots_mask) {
    return ((((unsigned)id * 2654435761u) ^ ((unsigned)((unsigned long long)id >> 32) * 40503u)) >> 5) & slots_mask;
}

// bit of an id in the batch-membership bitmap (2^bits bits): a second, independent mix of the id
__device__ __forceinline__ unsigned bm_hash(long long id, int bits) {
    return (unsigned)(((unsigned long long)id * 0x9E3779B97F4A7C15ull) >> (64 - bits));
}

// (64-thread workgroups: every thread is one chain of dependent memory operations; spread over all CUs)
// Round trips per column: the id; then the slot, its first-column word and the bitmap word TOGETHER (a Zipf batch repeats its
// popular items hundreds of times: looking before every atomic saves most of them -- a stale read only costs the atomic it
// would have saved; slots never change once taken, gfirst only decreases, bitmap bits are only set); then the claiming CAS
// when the slot was empty.  The two updates behind it return nothing and are not waited for.
__global__ __launch_bounds__(64) void gt_insert_kernel(const int64_t* __restrict__ item_idx, int64_t N, int M,
                                                        long long* __restrict__ gtab, int32_t* __restrict__ gfirst,
                                                        int32_t* __restrict__ colslot, uint32_t* __restrict__ bmap, int bmbits) {
    // Round 4: ONE lane per item and wave talks to the tables.  Every column of the batch runs at once, so "look before the
    // atomic" saves nothing on the popular items (all 700 copies of the first one see an empty slot), and atomics on one word
    // queue up at its L2 channel: 700 claims + 700 minima on two words were most of this kernel's 16.5 us.  The wave's 64
    // columns elect, per item, their lowest lane (a 128-slot LDS table: the key, then the minimum lane among the lanes that
    // find their own key there; two different items in one slot: the loser just acts for itself); the leader probes, claims,
    // publishes the slot; its followers -- later columns of the same item: their minimum is not needed -- take it from LDS.
    __shared__ unsigned long long e_key[128];
    __shared__ int e_lane[128];
    __shared__ unsigned e_hpos[128];
    const int lane = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * 64 + lane;
    const bool live = j < N;
    const long long key = live ? item_idx[j] : 0;
    const unsigned es = (unsigned)(((unsigned long long)key * 0x9E3779B97F4A7C15ull) >> 57);      // 7 bits
    e_lane[lane] = 64; e_lane[lane + 64] = 64;
    __syncthreads();
    if (live) e_key[es] = (unsigned long long)key;          // (any one of the slot's writers stays)
    __syncthreads();
    const bool mine_slot = live && e_key[es] == (unsigned long long)key;
    if (mine_slot) atomicMin(&e_lane[es], lane);
    __syncthreads();
    const bool leader = live && (!mine_slot || e_lane[es] == lane);
    unsigned hpos = 0u;
    if (leader) {
        unsigned long long* tab = reinterpret_cast<unsigned long long*>(gtab);
        const volatile unsigned long long* vtab = tab;
        hpos = ht_hash(key, M - 1);
        const unsigned b = bm_hash(key, bmbits);
        const uint32_t bit = 1u << (b & 31);
        unsigned long long old = vtab[hpos];                                   // three independent loads in flight
        int32_t first_seen = *(const volatile int32_t*)&gfirst[hpos];
        const uint32_t bm_seen = *(const volatile uint32_t*)&bmap[b >> 5];
        for (int probe = 0; probe < M; ++probe) {
            if (probe) {
                old = vtab[hpos];
                first_seen = *(const volatile int32_t*)&gfirst[hpos];
            }
            if (old == (unsigned long long)HT_EMPTY)
                old = atomicCAS(&tab[hpos], (unsigned long long)HT_EMPTY, (unsigned long long)key);
            if (old == (unsigned long long)HT_EMPTY || old == (unsigned long long)key) break;
            hpos = (hpos + 1) & (M - 1);
        }
        if (first_seen > (int32_t)j) atomicMin(&gfirst[hpos], (int32_t)j);     // cleared to 0x7f7f7f7f
        if (!(bm_seen & bit)) atomicOr(&bmap[b >> 5], bit);
        if (mine_slot) e_hpos[es] = hpos;
    }
    __syncthreads();
    if (live) colslot[j] = (int32_t)(leader ? hpos : e_hpos[es]);
}

// colfirst[j] = first column with column j's item (-1: padding column)
__device__ __forceinline__ void colfirst_body(int64_t j, const int32_t* __restrict__ colslot,
                                              const int32_t* __restrict__ gfirst, int64_t N, int64_t Np,
                                              int32_t* __restrict__ colfirst) {
    int32_t f = -1;
    if (j < N) f = gfirst[colslot[j]];
    if (j < Np) colfirst[j] = f;
}

// gtab lookup: first column of `key` in the batch, or -1
__device__ __forceinline__ int32_t ht_find(long long key, int M, const long long* __restrict__ gtab,
                                           const int32_t* __restrict__ gfirst) {
    unsigned hpos = ht_hash(key, M - 1);
    for (int probe = 0; probe < M; ++probe) {
        const long long sv = gtab[hpos];
        if (sv == key) return gfirst[hpos];
        if (sv == HT_EMPTY) return -1;        // this positive is not in the batch
        hpos = (hpos + 1) & (M - 1);
    }
    return -1;
}

// Where a user's positives come from: the reference's padded matrix pos_idx[B][P] (0-padded on the right,
// data/load.py:38-55; the zeros are looked up like any id: they match a column whose item id is 0, as upstream), or the
// batch producer's own CSR lists -- user_ids[i]'s items are pos_items[pos_off[u] .. pos_off[u + 1]) -- which never
// materialises a [B, P] tensor: a MovieLens-25M user can hold > 10^4 positives.
struct PosSrc {
    const int64_t* pos_idx; int P;
    const int64_t *user_ids, *pos_off, *pos_items; int64_t num_users;
};
__device__ __forceinline__ int pos_list(const PosSrc& s, int64_t i, const int64_t*& base) {
    if (s.pos_off) {
        const int64_t u = s.user_ids[i];
        if (u < 0 || u >= s.num_users) { base = s.pos_items; return 0; }
        const int64_t o = s.pos_off[u];
        base = s.pos_items + o;
        const int64_t len = s.pos_off[u + 1] - o;
        return (int)(len < 0 ? 0 : (len > 0x7FFFFFF ? 0x7FFFFFF : len));
    }
    base = s.pos_idx + i * s.P;
    return s.P;
}

// One launch for both consumers of the finished hash table.  Blocks [0, nb_col): colfirst.  The rest: `split` blocks per
// group of 32 users.  ubits is laid out [plane][group][column]: a block owns ONE ROW -- word `column` of its group, in its
// plane -- keeps that row in LDS (64 KiB at N = 16,384), sets a bit per hit with a fire-and-forget LDS atomic indexed by the
// hit's first column, and writes the whole row out with coalesced 16-byte stores: nothing needs clearing, no hash table of
// hits, no scattered 4-byte global stores (the [column][group] layout of round 2 scattered ~4 M of them per batch at the
// MovieLens-25M list profile, and per-hit global atomics were 5 x slower still: 967 us).
// The lists of the 32 users are walked as ONE flat range, cut in `split` equal pieces, one block and one plane each (a user
// with 30,000 positives -- and the group it sits in -- is spread over `split` x 1024 threads; list lengths are heavy-tailed:
// with one block per group the heaviest group set the kernel's time); the sweep ORs the planes.  Most positives are not in
// the batch at all: every id is first tested against the batch-membership bitmap (a copy in LDS; ~7 % false positives at
// 8 bits per column), and only the survivors probe the hash table in L2 -- HITS_MLP ids in flight per thread, first probes
// and first-column reads batched.  Batches wider than HITS_WIN columns are handled in windows (the lists are walked once
// per window).
static constexpr int HITS_THREADS = 1024, HITS_MLP = 8, HITS_MAX_SPLIT = HITS_PLANES, HITS_WIN = 32768;

__global__ __launch_bounds__(HITS_THREADS) void hits_kernel(const int64_t* __restrict__ item_idx, PosSrc src,
                                                            int64_t B, int64_t N, int64_t Bp, int64_t Np, int M, int nb_col, int split,
                                                            int win, const long long* __restrict__ gtab,
                                                            const int32_t* __restrict__ gfirst, const int32_t* __restrict__ colslot,
                                                            int32_t* __restrict__ colfirst, uint32_t* __restrict__ ubits,
                                                            const uint32_t* __restrict__ bmap, int bmbits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hl[];          // [win] the row, then the bitmap
    __shared__ int lstart[33];
    __shared__ const int64_t* lbase[32];
    if ((int)blockIdx.x < nb_col) {
        colfirst_body((int64_t)blockIdx.x * HITS_THREADS + threadIdx.x, colslot, gfirst, N, Np, colfirst);
        return;
    }
    const int grp = (blockIdx.x - nb_col) / split, sub = (blockIdx.x - nb_col) % split;
    const int64_t ngrp = Bp >> 5;
    uint32_t* out_row = ubits + ((int64_t)sub * ngrp + grp) * Np;
    uint32_t* row = hl;
    uint32_t* lbm = hl + win;
    const int bmw = 1 << (bmbits - 5);
    for (int e = threadIdx.x; e < bmw / 4; e += HITS_THREADS) reinterpret_cast<uint4*>(lbm)[e] = reinterpret_cast<const uint4*>(bmap)[e];
    if (threadIdx.x < 64) {                                                // wave 0: the 32 lists and their flat offsets
        const int ul = threadIdx.x & 31;
        const int64_t i = (int64_t)grp * 32 + ul;
        const int64_t* base = nullptr;
        int len = 0;
        if (i < B) len = pos_list(src, i, base);
        int incl = len;                                                    // inclusive scan over the 32 lanes of a half-wave
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (ul >= o) incl += up;
        }
        if (threadIdx.x < 32) {
            lstart[ul + 1] = incl;
            lbase[ul] = base;
            if (ul == 0) lstart[0] = 0;
        }
    }
    for (int64_t w0 = 0; w0 < Np; w0 += win) {
        const int nw = (int)min((int64_t)win, Np - w0);                    // (Np is a multiple of 128)
        for (int e = threadIdx.x; e < nw / 4; e += HITS_THREADS) reinterpret_cast<uint4*>(row)[e] = uint4{0u, 0u, 0u, 0u};
        __syncthreads();
        const int total = lstart[32];
        // flat positions [0, total): the lists; [total, total + 32): every user's own item (the accidental-hit term, losses.py:103)
        const int piece = (total + 32 + split - 1) / split;
        const int p0 = sub * piece, p1 = min(total + 32, p0 + piece);
        for (int t0 = p0 + threadIdx.x; t0 < p1; t0 += HITS_THREADS * HITS_MLP) {
            long long key[HITS_MLP];
            int ulv[HITS_MLP];
            bool live[HITS_MLP];
#pragma unroll
            for (int j = 0; j < HITS_MLP; ++j) {
                const int t = t0 + j * HITS_THREADS;
                live[j] = false; key[j] = 0; ulv[j] = 0;
                if (t < p1 && t < total) {
                    int lo = 0, hi = 32;                                   // the list holding flat position t: last start <= t
#pragma unroll
                    for (int sft = 0; sft < 5; ++sft) {
                        const int mid = (lo + hi) >> 1;
                        if (lstart[mid] <= t) lo = mid; else hi = mid;
                    }
                    ulv[j] = lo;
                    key[j] = lbase[lo][t - lstart[lo]];
                    live[j] = true;
                } else if (t < p1) {
                    ulv[j] = t - total;
                    const int64_t i = (int64_t)grp * 32 + ulv[j];
                    if (i < B) { key[j] = item_idx[i]; live[j] = true; }
                }
            }
            unsigned hpos[HITS_MLP];
            long long sv[HITS_MLP];
#pragma unroll
            for (int j = 0; j < HITS_MLP; ++j) {                           // prefilter, then the first probes, all in flight
                if (live[j]) {
                    const unsigned b = bm_hash(key[j], bmbits);
                    live[j] = (lbm[b >> 5] >> (b & 31)) & 1u;
                }
                hpos[j] = ht_hash(key[j], M - 1);
                sv[j] = live[j] ? gtab[hpos[j]] : HT_EMPTY;
            }
            int32_t f[HITS_MLP];
#pragma unroll
            for (int j = 0; j < HITS_MLP; ++j) {                           // the first columns of the direct hits, all in flight
                f[j] = -1;
                if (live[j] && sv[j] == key[j]) f[j] = gfirst[hpos[j]];
            }
#pragma unroll
            for (int j = 0; j < HITS_MLP; ++j) {
                if (!live[j] || sv[j] == HT_EMPTY) continue;
                if (sv[j] != key[j]) {                                     // collision: walk on (rare at load <= 1/2)
                    unsigned hp = (hpos[j] + 1) & (M - 1);
                    for (int probe = 1; probe < M; ++probe) {
                        const long long v = gtab[hp];
                        if (v == key[j]) { f[j] = gfirst[hp]; break; }
                        if (v == HT_EMPTY) break;
                        hp = (hp + 1) & (M - 1);
                    }
                }
                const int64_t fw = (int64_t)f[j] - w0;
                if (f[j] >= 0 && fw >= 0 && fw < nw) atomicOr(&row[fw], 1u << ulv[j]);   // (LDS, no return value)
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nw / 4; e += HITS_THREADS)
            reinterpret_cast<uint4*>(out_row + w0)[e] = reinterpret_cast<const uint4*>(row)[e];
        __syncthreads();
    }
}

// 32 x 32 bit transpose across the 32 lanes of a half-wave: lane i holds row i; five block-swap steps
template <int J, uint32_t M0>
__device__ __forceinline__ uint32_t bit_transpose_step(uint32_t w, int lane) {
    const uint32_t p = mf_xor_lane_u32<J>(w);
    return (lane & J) ? ((w & ~M0) | ((p & ~M0) >> J)) : ((w & M0) | ((p & M0) << J));
}
__device__ __forceinline__ uint32_t bit_transpose32(uint32_t w, int lane) {
    w = bit_transpose_step<16, 0x0000FFFFu>(w, lane);
    w = bit_transpose_step<8, 0x00FF00FFu>(w, lane);
    w = bit_transpose_step<4, 0x0F0F0F0Fu>(w, lane);
    w = bit_transpose_step<2, 0x33333333u>(w, lane);
    return bit_transpose_step<1, 0x55555555u>(w, lane);
}

// one half-wave per (column tile, group of 128 users): lane = column fetches word `first column` of the four 32-user groups
// (ubits [plane][group][column]: a coalesced 128-byte read per group and half-wave, duplicates of a column simply read
// another word; the planes of a split group are ORed), four transposes give the mask words of 4 x 32 users
__global__ __launch_bounds__(256) void mask_sweep_kernel(const int32_t* __restrict__ colfirst,
                                                         const uint32_t* __restrict__ ubits, int64_t B, int64_t Bp, int64_t Np,
                                                         int NT, uint32_t* __restrict__ maskW, int planes) {
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int ngrp = (int)(Bp >> 7);
    const int64_t unit = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + h;
    if (unit >= (int64_t)NT * ngrp) return;
    const int tj = (int)(unit / ngrp), g = (int)(unit % ngrp);
    const int32_t f = colfirst[tj * 32 + c];                 // -1: padding column, never a negative
    uint32_t in[4] = {~0u, ~0u, ~0u, ~0u};
    if (f >= 0) {
        const int64_t plane_words = (Bp >> 5) * Np;
#pragma unroll
        for (int q = 0; q < 4; ++q) in[q] = ubits[((int64_t)4 * g + q) * Np + f];
        for (int pl = 1; pl < planes; ++pl) {
#pragma unroll
            for (int q = 0; q < 4; ++q) in[q] |= ubits[pl * plane_words + ((int64_t)4 * g + q) * Np + f];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t word = bit_transpose32(in[q], c);
        const int64_t i = ((int64_t)4 * g + q) * 32 + c;
        maskW[(int64_t)tj * Bp + i] = i < B ? word : ~0u;    // padding users: no negatives at all
    }
}

// ---------------------------------------------------------- dense forward ------
struct FwdParams {
    const float *u, *v, *nu, *nv, *lii, *sgn, *logq;
    const uint32_t* maskW;
    float* part;
    float* stash;          // [Bp/32][NT] blocks of 32 x 32 logits (see BwdParams)
    int64_t B, N, Bp;
    int NT, tps, need;
    float sigma, margin;
    // side inputs of a tile as ONE buffer: byte offsets of maskW / nv / logq from aux_base, and the span covered
    const char* aux_base;
    uint32_t aux_mask, aux_nv, aux_lq, aux_bytes;
};

// Workgroup = 4 waves x 32 users; item tiles arrive through a 2-slot LDS ring filled by LDS-DMA, together with
// their mask words, norms and -logQ (a separate 4-deep ring of side-input slots).  The loop is software-pipelined
// inside each wave: the 64 MFMAs of tile t+1 are issued in the same basic block as the (branch-free) statistics of
// tile t, and the DMA pieces of tile t+2 -- into the slot tile t just left -- go out one per MFMA group, so the
// matrix pipe, the VALU and the memory instructions overlap without relying on a second wave.  The tile loop is
// unrolled by the two slots: every LDS address is lane base + immediate, and the two accumulator sets swap roles
// by name (no register copies at the loop edge).
template <int D>
struct FwdLds {
    using G = TileGeom<D>;
    // side inputs: one 1-KiB DMA per wave and tile -- lanes 0..7 the wave's 32 mask words, 8..15 the tile's item
    // norms, 16..23 its -logq (every wave keeps its own copy: no cross-wave dependency), the other lanes zeros
    static constexpr int AUX_NV = 128, AUX_LQ = 256, AUXW = 1024, AUXB = G::NW * AUXW;
    static constexpr int AUX0 = 2 * G::TILEB;           // 4 side-input slots after the 2 tile slots
    static constexpr int BYTES = AUX0 + 4 * AUXB;
    static constexpr int NG = D / 8;                    // MFMA groups per tile
    static constexpr int LASTG = G::PPW;                // group of a stage's last DMA (tile pieces, then the side inputs)
    // stash stores of a tile (issued behind logit slices 3, 7, 11, 15 of 32) that are YOUNGER than the stage's last
    // DMA: `s_waitcnt vmcnt(KEEP)` at the next tile's top then proves every DMA of the stage has landed
    // (a group issues its DMA before its slices, so a store of group LASTG itself is already younger)
    static constexpr int store_group(int s) { return s * NG / 32; }
    static constexpr int KEEP = (store_group(3) >= LASTG) + (store_group(7) >= LASTG) + (store_group(11) >= LASTG) + (store_group(15) >= LASTG);
    static_assert(LASTG < NG, "a stage's DMAs must fit the tile's MFMA groups");
};

template <int NEED>
__device__ __forceinline__ void stats_add_masked(RowStats& s, float Lm, float sm, float lii, float margin) {
    // Lm is -inf for masked elements: every term below is then exactly 0 (the count is kept separately)
    if (NEED & NEED_CONTR) s.A += fmaxf(Lm + sm, 0.f);
    if (NEED & (NEED_HINGE | NEED_LOGI)) {
        const float x = (Lm - lii) + margin;
        if (NEED & NEED_HINGE) {
            s.H += fmaxf(x, 0.f);
            s.Hc += x > 0.f ? 1.f : 0.f;
        }
        if (NEED & NEED_LOGI) {
            float sp, sg;
            softplus_sigmoid(x, sp, sg);
            s.Lg += sp;
            s.Ls += sg;
        }
    }
}

template <int D, int NEED>
__global__ __launch_bounds__(64 * mf_nw(D), mf_wg_per_cu(D)) void loss_fwd_dense_kernel(FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = TileGeom<D>;
    using L = FwdLds<D>;
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
    // grid = (item-range split, user block): consecutive workgroup ids -- dealt round-robin to the 8 XCDs --
    // differ in the SPLIT, so the workgroups of one XCD stream the same 1/nsplit of V through its L2
    const int64_t i0 = (int64_t)blockIdx.y * G::XB;
    const int64_t i = i0 + wave * 32 + c;
    const int t0 = blockIdx.x * p.tps, t1 = min(p.NT, t0 + p.tps);
    RowFrag<D> xf;
    mf_load_frag<D>(xf, p.u, i, i < p.B);
    const float nu_i = p.nu[i], s_i = p.sgn[i], lii = p.lii[i];
    const float sm = s_i * p.margin;
    RowStats st;
    stats_init(st);

    TileSrc<D> tsrc;
    mf_tile_src_init<D>(tsrc, p.v, p.N, (int64_t)t0 * 32);
    // side inputs: per-lane byte offset into the aux buffer, advanced by a per-lane stride from stage to stage
    // (stages are issued in tile order t0, t0 + 1, ...); lanes 24.. stay out of range and bring zeros
    mf_rsrc_t arsrc;
    uint32_t aoff, astep;
    {
        const int part = lane >> 3, l8 = lane & 7;
        aoff = part == 0 ? p.aux_mask + (uint32_t)(((int64_t)t0 * p.Bp + i0 + wave * 32) * 4) + l8 * 16
             : part == 1 ? p.aux_nv + (uint32_t)t0 * 128u + l8 * 16
             : part == 2 ? p.aux_lq + (uint32_t)t0 * 128u + l8 * 16 : MF_SRD_DEAD;
        astep = part == 0 ? (uint32_t)p.Bp * 4u : part <= 2 ? 128u : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.aux_base), 0, (int)p.aux_bytes, 0x00020000);
#else
        (void)arsrc; (void)aoff;
#endif
    }
    // one memory instruction of the stage of tile t (into tile slot `slot`): j < PPW a tile piece, j = PPW the side inputs
    auto stage_piece = [&](int t, int slot, int j, bool live) {
        if (j < G::PPW) {
            mf_stage_tile_piece<D>(smem + slot * G::TILEB, t * 32, j, tsrc, live);
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (mf_lds_ptr)(smem + L::AUX0 + ((t - t0) & 3) * L::AUXB + wave * L::AUXW), 16,
                                                     (int)aoff, live ? 0 : (int)MF_SRD_DEAD, 0, 0);
#endif
            aoff += astep;
        }
    };
    auto stage = [&](int t, int slot) {
#pragma unroll
        for (int j = 0; j <= L::LASTG; ++j) stage_piece(t, slot, j, true);
    };
    // The statistics of one tile, cut in 32 slices so they can be threaded between the MFMAs of
    // the next tile: slices 0..15 turn score e into logit e (and stash it), slices 16..31 fold
    // logit e into the running statistics.
    float Lg[16];
    float tmax = -FLT_MAX, nmx = -FLT_MAX, nmx2 = 0.f;
    uint32_t mw = 0u;
    f32x4 nv4 = {0.f, 0.f, 0.f, 0.f}, lq4 = {0.f, 0.f, 0.f, 0.f};
    auto slice = [&](int sidx, int te, const f32x16& acc) {
        const char* aux = smem + L::AUX0 + ((te - t0) & 3) * L::AUXB + wave * L::AUXW;
        if (sidx < 16) {
            const int e = sidx, q = e >> 2, r = e & 3;
            if (e == 0) {
                mw = reinterpret_cast<const uint32_t*>(aux)[c];
                tmax = -FLT_MAX;
                // valid negatives among this lane's 16 rows of the tile (rows (e&3) + 8 (e>>2) + 4 h)
                st.cnt += (float)(16 - __builtin_popcount(mw & (h ? 0xF0F0F0F0u : 0x0F0F0F0Fu)));
                mw >>= 4 * h;           // this lane's rows now sit at the compile-time bit (e&3) + 8 (e>>2)
            }
            if (r == 0) {
                nv4 = *reinterpret_cast<const f32x4*>(aux + L::AUX_NV + (8 * q + 4 * h) * 4);
                lq4 = *reinterpret_cast<const f32x4*>(aux + L::AUX_LQ + (8 * q + 4 * h) * 4);
            }
            // masked logit: -inf where the column is not a valid negative -> every statistic below and
            // every dloss/dL of the backward is exactly 0 there, with no further mask test
            const float Lraw = mf_logit(nu_i, nv4[r], acc[e], s_i, p.sigma, -lq4[r]);      // lq4 holds -logq
            Lg[e] = (mw & (1u << mf_acc_row(e, 0))) ? -INFINITY : Lraw;
            tmax = fmaxf(tmax, Lg[e]);
            if (r == 3) {   // stash 4 masked logits of the block for the backward sweeps
                float* blk = p.stash + ((int64_t)(i0 / 32 + wave) * p.NT + te) * 1024 + lane * 4;
                *reinterpret_cast<f32x4*>(blk + q * 256) = f32x4{Lg[e - 3], Lg[e - 2], Lg[e - 1], Lg[e]};
            }
        } else {
            const int e = sidx - 16;
            if (e == 0) {
                nmx = fmaxf(st.mx, tmax);
                if (NEED & NEED_LSE) st.se *= __expf(st.mx - nmx);
                st.mx = nmx;
                // exp(L - nmx) = exp2(fma(L, log2 e, -nmx log2 e)): one fma + one exp2 per element; the clamp
                // keeps the constant finite while nothing valid has been seen (nmx = -FLT_MAX, every L = -inf)
                nmx2 = -fmaxf(nmx, -1e30f) * 1.44269504088896341f;
            }
            stats_add_masked<NEED>(st, Lg[e], sm, lii, p.margin);
            if (NEED & NEED_LSE) st.se += __builtin_amdgcn_exp2f(__builtin_fmaf(Lg[e], 1.44269504088896341f, nmx2));
        }
    };
    // one iteration: wait for tile tj + 1 (slot NS), then its 64 MFMAs into `nxt`, interleaved with the slices of tile
    // tj on `cur` and -- MODE 1: always, 2: if it exists, 0: never -- the stage of tile tj + 2 into the slot tile tj has left
    auto iterate = [&](int tj, auto cs_tag, auto mode_tag, const f32x16& cur, f32x16& nxt) {
        constexpr int CS = decltype(cs_tag)::value, NS = CS ^ 1, MODE = decltype(mode_tag)::value;
        // every DMA of the stage of tile tj + 1 is older than the KEEP youngest stores
        if (tj == t0) mf_wait_vmcnt<0>(); else mf_wait_vmcnt<L::KEEP>();
        mf_block_barrier();
        const bool live = MODE == 1 || tj + 2 < t1;
        nxt = mf_tile_scores_interleaved<D, 32, (40 * 32 / D)>(
            smem + NS * G::TILEB, xf, [&](int sidx) { slice(sidx, tj, cur); },
            [&](int g) { if (MODE != 0 && g <= L::LASTG) stage_piece(tj + 2, CS, g, live); });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    if (t0 < t1) {
        stage(t0, 0);
        if (t0 + 1 < t1) stage(t0 + 1, 1);
        // tile t0: its stage's DMAs are the older ones
        if (t0 + 1 < t1) mf_wait_vmcnt<L::LASTG + 1>(); else mf_wait_vmcnt<0>();
        mf_block_barrier();
        f32x16 accA = mf_tile_scores_interleaved<D, 32, (40 * 32 / D)>(smem, xf, [](int) {}), accB;
        int tj = t0;
        for (; tj + 2 < t1; tj += 2) {      // two tiles per trip: slots and accumulators swap roles by name
            iterate(tj, I0{}, I1{}, accA, accB);
            iterate(tj + 1, I1{}, I2{}, accB, accA);
        }
        if (tj + 1 < t1) {                  // tiles tj (scores in accA) and tj + 1 remain
            iterate(tj, I0{}, I0{}, accA, accB);
#pragma unroll
            for (int sidx = 0; sidx < 32; ++sidx) slice(sidx, tj + 1, accB);
        } else {                            // tile tj remains
#pragma unroll
            for (int sidx = 0; sidx < 32; ++sidx) slice(sidx, tj, accA);
        }
        mf_wait_vmcnt<0>();                 // nothing of this workgroup may still be on its way into LDS when it ends
    }
    // the row's other half of the columns lives in lane ^ 32
    {
        const float mx2 = mf_shfl_xor32(st.mx), se2 = mf_shfl_xor32(st.se);
        lse_merge(st.mx, st.se, mx2, se2);
        st.cnt += mf_shfl_xor32(st.cnt); st.A += mf_shfl_xor32(st.A);
        st.H += mf_shfl_xor32(st.H); st.Hc += mf_shfl_xor32(st.Hc);
        st.Lg += mf_shfl_xor32(st.Lg); st.Ls += mf_shfl_xor32(st.Ls);
    }
    if (h == 0) {
        float* o = p.part + (int64_t)blockIdx.x * NSTAT * p.Bp + i;
        o[ST_CNT * p.Bp] = st.cnt; o[ST_A * p.Bp] = st.A; o[ST_MX * p.Bp] = st.mx; o[ST_SE * p.Bp] = st.se;
        o[ST_H * p.Bp] = st.H; o[ST_HC * p.Bp] = st.Hc; o[ST_LG * p.Bp] = st.Lg; o[ST_LS * p.Bp] = st.Ls;
    }
}

template <int D, int NEED>
static void launch_fwd_n(dim3 grid, const FwdParams& fp, hipStream_t s) {
    auto fn = loss_fwd_dense_kernel<D, NEED>;
    if (FwdLds<D>::BYTES > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, FwdLds<D>::BYTES);
    fn<<<grid, 64 * mf_nw(D), FwdLds<D>::BYTES, s>>>(fp);
}
template <int D>
static void launch_fwd(int need, dim3 grid, const FwdParams& fp, hipStream_t s) {
    switch (need) {   // single-loss calls get a specialised epilogue; anything else computes every statistic
        case NEED_CONTR: launch_fwd_n<D, NEED_CONTR>(grid, fp, s); break;
        case NEED_LSE: launch_fwd_n<D, NEED_LSE>(grid, fp, s); break;
        case NEED_HINGE: launch_fwd_n<D, NEED_HINGE>(grid, fp, s); break;
        case NEED_LOGI: launch_fwd_n<D, NEED_LOGI>(grid, fp, s); break;
        default: launch_fwd_n<D, 15>(grid, fp, s); break;
    }
}

// The tail of the forward in ONE launch: merge the item-range splits of the row statistics in split
// order (nsplit = 0: `stats` is already final), evaluate the seven per-row losses, and sum them over
// the batch in a fixed order: in-block tree, then the last workgroup to finish (ticket) adds the
// block sums in block order -- deterministic, no second launch.
__global__ __launch_bounds__(64) void finish_kernel(const float* __restrict__ part, int nsplit, int64_t B, int64_t Bp,
                                                     const float* __restrict__ target, const float* __restrict__ lii,
                                                     const float* __restrict__ dii, float sigma, int kind_mask,
                                                     float* __restrict__ stats, float* __restrict__ rowloss,
                                                     float* __restrict__ blockpart, unsigned* __restrict__ ticket,
                                                     float* __restrict__ out, int rowc_kind, float margin,
                                                     const float* __restrict__ sgn, float* __restrict__ rowc) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    float acc[NSTAT];
    if (i < Bp) {
        if (nsplit > 0) {
            for (int s = 0; s < NSTAT; ++s) acc[s] = part[(int64_t)s * Bp + i];
            for (int sp0 = 1; sp0 < nsplit; sp0 += 7) {             // seven splits' loads in flight, merged in order
                float q[7][NSTAT];
#pragma unroll
                for (int j = 0; j < 7; ++j)
#pragma unroll
                    for (int s = 0; s < NSTAT; ++s)
                        q[j][s] = sp0 + j < nsplit ? part[((int64_t)(sp0 + j) * NSTAT + s) * Bp + i] : 0.f;
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    if (sp0 + j >= nsplit) break;
                    lse_merge(acc[ST_MX], acc[ST_SE], q[j][ST_MX], q[j][ST_SE]);
                    acc[ST_CNT] += q[j][ST_CNT]; acc[ST_A] += q[j][ST_A]; acc[ST_H] += q[j][ST_H];
                    acc[ST_HC] += q[j][ST_HC]; acc[ST_LG] += q[j][ST_LG]; acc[ST_LS] += q[j][ST_LS];
                }
            }
            for (int s = 0; s < NSTAT; ++s) stats[(int64_t)s * Bp + i] = acc[s];
        } else {
            for (int s = 0; s < NSTAT; ++s) acc[s] = stats[(int64_t)s * Bp + i];
        }
    }
    float o[MF_NUM_KINDS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < B) row_losses(acc, target[i], lii[i], dii[i], sigma, o);
    if (rowc_kind >= 0 && i < Bp) {       // the backward's row coefficients of the one trained loss (MF_LOSS_ROWC)
        float a = 0.f, b = 0.f, cg = 0.f, gd = 0.f;
        if (i < B) rowc_row(rowc_kind, target[i], sgn[i], lii[i], acc[ST_CNT], acc[ST_MX], acc[ST_SE], acc[ST_HC], acc[ST_LS],
                            sigma, margin, a, b, cg, gd);
        rowc[i] = a; rowc[Bp + i] = b; rowc[2 * Bp + i] = cg; rowc[3 * Bp + i] = gd;
    }
    // one wave per workgroup: the block sum is a fixed xor tree over the lanes (no LDS, no barrier)
    const int lane = mf_lane();
    for (int k = 0; k < MF_NUM_KINDS; ++k) {
        if (i < Bp) rowloss[(int64_t)k * Bp + i] = o[k];
        float v = o[k];
        v = mf_wave_sum(v);
        if (lane == 0) blockpart[(int64_t)k * gridDim.x + blockIdx.x] = v;
    }
    // release only (the block sums are in L2 before the ticket moves): a full fence also invalidates the L1, ~1.7 us each;
    // the last workgroup reads the sums with L1-bypassing loads instead of an acquire fence
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    int last = 0;
    if (lane == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    last = __shfl(last, 0, 64);
    if (last) {
        // the last workgroup adds the block sums of every loss kind: lane-strided partial sums, then a fixed
        // xor tree -- the same order on every run
        float tot[MF_NUM_KINDS];
#pragma unroll
        for (int k = 0; k < MF_NUM_KINDS; ++k) tot[k] = 0.f;
        for (unsigned b = lane; b < gridDim.x; b += 64) {       // the seven kinds' loads of a round are in flight together
            float v[MF_NUM_KINDS];
#pragma unroll
            for (int k = 0; k < MF_NUM_KINDS; ++k)
                v[k] = __hip_atomic_load(blockpart + (int64_t)k * gridDim.x + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < MF_NUM_KINDS; ++k) tot[k] += v[k];
        }
#pragma unroll
        for (int k = 0; k < MF_NUM_KINDS; ++k) {
            float t = tot[k];
            t = mf_wave_sum(t);
            if (lane == 0) out[k] = ((kind_mask >> k) & 1) ? t : 0.f;   // every entry is written
        }
    }
}

__global__ __launch_bounds__(256) void mask_export_dense_kernel(const uint32_t* __restrict__ maskW, int64_t B,
                                                                int64_t N, int64_t Bp, int NT,
                                                                uint32_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= B * NT) return;
    const int64_t i = t / NT;
    const int w = (int)(t % NT);
    uint32_t bits = ~maskW[(int64_t)w * Bp + i];
    const int64_t rem = N - (int64_t)w * 32;
    if (rem < 32) bits &= (1u << rem) - 1u;
    out[i * NT + w] = bits;
}

// ------------------------------------------------------------- mined forward ---
struct MiningPolicy {
    struct Params {
        const float *nu, *nv, *lii, *sgn, *logq;
        const uint32_t* maskW;
        int64_t Bp, N;
        float sigma;
    };
    struct Row {
        float nu, sgn, lii;
    };
    struct Tile {
        uint32_t mw;
        f32x4 nv4[4], lq4[4];
    };
    static constexpr int AUX_DMA = 2;
    // The mining rank orders Dm = L_ij - L_ii: semi-hard (Dm < 0, larger first) above hard (Dm >= 0, smaller
    // first), so `rank >= bound` is an INTERVAL of Dm: [v, 0) for a semi-hard bound, (-inf, v'] for a hard one.
    // Dm is computed exactly as key() does; the interval is widened to the bound's 30 kept rank bits
    // (conservative), everything exact happens in key().
    struct Thr {
        float lo, hi;
    };
    static __device__ __forceinline__ Thr thr_all() { return Thr{-__builtin_inff(), __builtin_inff()}; }
    static __device__ __forceinline__ Thr thr_none() { return Thr{__builtin_inff(), -__builtin_inff()}; }
    static __device__ __forceinline__ Thr make_thr(unsigned bound) {
        const unsigned cls = bound >> 30;
        const float v = mf_unorderable((bound & 0x3FFFFFFFu) << 2);   // smallest value whose rank has these 30 bits
        if (cls == 2u) return Thr{v, 0.f};              // semi-hard bound: Dm in [v, 0]  (Dm = 0 itself fails key())
        if (cls == 1u) return Thr{-__builtin_inff(), 0.f - v};   // hard bound at -Dm >= v: every Dm <= -v
        return cls == 0u ? thr_all() : thr_none();
    }
    static __device__ __forceinline__ bool maybe(const Params& p, const Row& r, const Tile& t, float score, int e, int,
                                                 const Thr& th) {
        const float L = mf_logit(r.nu, t.nv4[e >> 2][e & 3], score, r.sgn, p.sigma, -t.lq4[e >> 2][e & 3]);   // lq4 holds -logq
        const float dm = L - r.lii;
        return dm >= th.lo && dm <= th.hi;
    }
    static __device__ __forceinline__ void stage_aux(const Params& p, char* aux, int wave, int t, int64_t x0, int W0) {
        mf_stage_small(aux + wave * 128, p.maskW + (int64_t)t * p.Bp + x0, 128);
        const float* src = (wave == 1 && p.logq) ? p.logq : p.nv;
        mf_stage_small(aux + W0 + wave * 128, src + (int64_t)t * 32, 128);    // W0: nv, W0 + 128: logq, rest: copies
    }
    static __device__ __forceinline__ Row row_init(const Params& p, int64_t x, bool) {
        return Row{p.nu[x], p.sgn[x], p.lii[x]};
    }
    static __device__ __forceinline__ Tile tile_init(const Params& p, const Row&, const char* aux, int wave, int c, int h, int W0) {
        Tile t;
        t.mw = reinterpret_cast<const uint32_t*>(aux + wave * 128)[c];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t.nv4[q] = *reinterpret_cast<const f32x4*>(aux + W0 + (8 * q + 4 * h) * 4);
            t.lq4[q] = p.logq ? *reinterpret_cast<const f32x4*>(aux + W0 + 128 + (8 * q + 4 * h) * 4)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return t;
    }
    static __device__ __forceinline__ bool key(const Params& p, const Row& r, const Tile& t, float score, int e, int h,
                                               unsigned y, unsigned& hi, unsigned& lo) {
        const float L = mf_logit(r.nu, t.nv4[e >> 2][e & 3], score, r.sgn, p.sigma, -t.lq4[e >> 2][e & 3]);
        const float dm = L - r.lii;
        hi = mf_key_mining_hi(dm);
        lo = mf_key_mining_lo(dm, y);
        return !((t.mw >> mf_acc_row(e, h)) & 1u);          // hit, diagonal or padding column
    }
};

// One wave per user: exact ordered top-k of the row's candidate list -> sel[i][0..cnt), then -- same wave, the
// user's row still in its registers -- the logits of the selected negatives (every item row is read by the whole
// wave: coalesced 16-byte lanes; the dot is a wave reduction here, not the chain: these logits feed the loss value
// and its gradient, tolerance 1e-4, not the bit-exact selection) and the row's statistics in selection order.
struct MinedRowParams {
    const unsigned long long* cand;
    const int32_t* cand_cnt;
    int rowcap, k;
    const float *u, *v, *nu, *nv, *lii, *sgn, *nlogq;    // nlogq: the workspace's -logq copy
    int64_t B, Bp;
    int d;
    float sigma, margin;
    int need;
    int32_t *sel, *sel_cnt;
    float *sel_L, *stats;
    const int32_t* gate;         // not NULL: the launch does nothing unless *gate is set (behind the prefilter, whose rescoring waves
                                 // finish their users themselves: MinedRowFinish)
};
// sorted[0 .. m) (LDS): user i's selected negatives in order -> sel, their logits, the row's statistics.  i >= B: only the
// (empty) statistics are written.
struct MinedRowFinish {
    using Params = MinedRowParams;
    static __device__ __forceinline__ void run(const MinedRowParams& p, int64_t i, int m, const unsigned long long* sorted) {
        const int lane = mf_lane();
        RowStats st;
        stats_init(st);
        if (i < p.B) {
            if (lane < m) p.sel[i * KSEL_MAX + lane] = (int32_t)mf_key_mining_col(sorted[lane]);
            if (lane == 0) p.sel_cnt[i] = m;
            const float s_i = p.sgn[i], l = p.lii[i], sm = s_i * p.margin, nu_i = p.nu[i];
            const int d4 = p.d / 4;                  // <= 64 chunks of 16 bytes (d <= 256): one per lane
            f32x4 ur = {0.f, 0.f, 0.f, 0.f};
            if (lane < d4) ur = reinterpret_cast<const f32x4*>(p.u + i * p.d)[lane];
            // (four rows, their norms and logQ in flight at a time: one at a time was a memory round trip per selected negative)
            for (int t0 = 0; t0 < m; t0 += 4) {
                f32x4 vr[4];
                float nvj[4], lqj[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t j = t0 + q < m ? (int64_t)mf_key_mining_col(sorted[t0 + q]) : 0;
                    vr[q] = (lane < d4 && t0 + q < m) ? reinterpret_cast<const f32x4*>(p.v + j * p.d)[lane] : f32x4{0.f, 0.f, 0.f, 0.f};
                    nvj[q] = p.nv[j];
                    lqj[q] = p.nlogq[j];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (t0 + q < m) {                            // (wave-uniform)
                        float part = ur[0] * vr[q][0] + ur[1] * vr[q][1] + ur[2] * vr[q][2] + ur[3] * vr[q][3];
                        part = mf_wave_sum(part);
                        const float L = mf_logit(nu_i, nvj[q], part, s_i, p.sigma, -lqj[q]);
                        if (lane == 0) p.sel_L[i * KSEL_MAX + t0 + q] = L;
                        stats_add(st, p.need, L, sm, l, p.margin);
                        if (p.need & NEED_LSE) lse_merge(st.mx, st.se, L, 1.f);
                    }
                }
            }
        }
        if (lane == 0) {
            float* o = p.stats + i;
            o[ST_CNT * p.Bp] = st.cnt; o[ST_A * p.Bp] = st.A; o[ST_MX * p.Bp] = st.mx; o[ST_SE * p.Bp] = st.se;
            o[ST_H * p.Bp] = st.H; o[ST_HC * p.Bp] = st.Hc; o[ST_LG * p.Bp] = st.Lg; o[ST_LS * p.Bp] = st.Ls;
        }
    }
};
__global__ __launch_bounds__(64) void mined_rows_kernel(MinedRowParams p) {
    __shared__ unsigned long long win[64], sorted[64];
    const int64_t i = blockIdx.x;
    if (p.gate && __hip_atomic_load(p.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    int m = 0;
    if (i < p.B) m = mf_row_topk<8>(p.cand + i * (int64_t)p.rowcap, p.cand_cnt[i], p.k, win, sorted);
    MinedRowFinish::run(p, i, m, sorted);
}

__global__ __launch_bounds__(256) void mask_export_mined_kernel(const int32_t* __restrict__ sel,
                                                                const int32_t* __restrict__ sel_cnt, int64_t B,
                                                                int NT, uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int n = sel_cnt[i];
    for (int t = 0; t < n; ++t) {
        const int j = sel[i * KSEL_MAX + t];
        out[i * NT + (j >> 5)] |= 1u << (j & 31);   // one thread owns the whole row
    }
}

// ------------------------------------------------------------------ backward ---
__global__ __launch_bounds__(256) void rowc_kernel(const float* __restrict__ stats, const float* __restrict__ target,
                                                   const float* __restrict__ lii, const float* __restrict__ sgn,
                                                   int64_t B, int64_t Bp, int kind, float sigma, float margin,
                                                   float* __restrict__ rowc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Bp) return;
    float a = 0.f, b = 0.f, cg = 0.f, gd = 0.f;
    if (i < B)
        rowc_row(kind, target[i], sgn[i], lii[i], stats[ST_CNT * Bp + i], stats[ST_MX * Bp + i], stats[ST_SE * Bp + i],
                 stats[ST_HC * Bp + i], stats[ST_LS * Bp + i], sigma, margin, a, b, cg, gd);
    rowc[0 * Bp + i] = a; rowc[1 * Bp + i] = b; rowc[2 * Bp + i] = cg; rowc[3 * Bp + i] = gd;
}

// The forward stashes the MASKED logits (-inf where the column is not a valid negative) of every
// 32 x 32 (user tile, item tile) block in HBM (B x N fp32: 0.5 GB at B = 8192 -- MI355X has
// 288 GB) so that the two backward sweeps only contract: they never recompute the score tile.
// Block (ti, tj) is 4 KiB: [q = e / 4][lane][e % 4] in the forward's accumulator layout
// (lane = user), i.e. four whole-KiB coalesced stores per tile and a linear LDS-DMA on the way
// back.  The dU sweep turns each block into G' = dloss/dL (one exp / step / sigmoid per element,
// diagonal patched) and writes it to a second stash, so the dV sweep, which runs after it, reads G'
// and has no per-element arithmetic at all: on gfx950 fp32 MFMA and VALU share the SIMD's FP32
// lanes (a VALU op costs ~2 cycles of MFMA throughput, a transcendental ~17 -- measured), so
// every VALU instruction removed from these loops is MFMA time won back.
// (Folding the split sum into these kernels -- the workgroup that reaches an X block's ticket last adds the
// partials -- was measured and dropped: 64..128 last workgroups reading 8 x 64 KiB each is far less parallel than
// the separate sum_parts launch; dU 0.287 -> 0.489 ms.)
struct BwdParams {
    const float *u, *v, *rowc;
    const float* grad_out;   // device scalar: the row coefficients are scaled by it here
    const float* stash;      // masked logits (forward); stays intact so that a backward can be repeated
    float* gstash;           // G' = dloss/dL blocks: written by the dU sweep, read by the dV sweep
    float* dpart;
    float* rpart;            // [split][Xp] partial row sums of G'
    int64_t B, N, Bp, Np;
    int NT, YT, tps;
};

template <int D, bool XU>
struct BwdLds {
    using G = TileGeom<D>;
    // Two schedules (see the kernel): SPREAD for long tiles (d >= 64) -- 2-slot tile ring, the wave's own
    // 4 KiB stash / G' block comes straight into registers (it is wave-private: routing it through LDS
    // only added 16 KiB of DMA writes and 16 KiB of reads per tile to an LDS port that the MFMA operand
    // reads need); otherwise whole stages two tiles ahead, stash blocks behind the tile in each slot.
    static constexpr bool SPREAD = D >= 64;
    static constexpr int LT = G::TILEB;                  // (not SPREAD) NW x 4 KiB stash blocks behind the tile
    static constexpr int SLOT = G::TILEB + (SPREAD ? 0 : G::NW * 4096);
    static constexpr int EXTRA = XU ? 0 : G::NW * 32 * 32 * 4;   // dV: per-wave 32 x 32 transpose scratch (XOR-swizzled)
    // the deepest ring that still puts two waves on every SIMD (two 4-wave workgroups per CU, or one of
    // eight waves); 2 slots cost a second barrier per tile, measured free
    static constexpr int NSLOT = SPREAD ? 2 : (((G::NW == 8 ? 1 : 2) * (3 * SLOT + EXTRA) <= 160 * 1024) ? 3 : 2);
    static constexpr int TR = NSLOT * SLOT;
    static constexpr int BYTES = TR + EXTRA;
    // d = 64 (config C2): the sweeps are HBM-bound on the 4 KiB stash blocks, not MFMA-bound (a block costs 4 KiB at every width, a tile's
    // MFMAs shrink with d).  There the dV sweep derives G' from the LOGIT stash itself -- the per-element map dU applies, with
    // the users' coefficients loaded per tile (lanes are users BEFORE the block is transposed) -- and dU no longer writes a
    // G' stash: dU moves 0.57 GB instead of 1.07 at C2's shape (round 4; same bits: the same map on the same logits).
    static constexpr bool RECOMP = D == 64;              // (d = 32 stages two tiles ahead: the coefficients would need a ring of their own)
    static constexpr int NCOEF = (RECOMP && !XU) ? 4 : 0;   // a, b, coefG, gdiag of the tile's users
    static constexpr int NDMA = G::PPW + 4 + NCOEF;      // memory instructions per wave per stage (4 = the stash block)
    static constexpr int NSTORE = (XU && !RECOMP) ? 4 : 0;  // G' stores per tile
};

// XU = true : lanes hold users, item tiles stream, result d loss / d u   (reads L, writes G' back)
// XU = false: lanes hold items, user tiles stream, result d loss / d v   (reads G')
template <int D, bool XU, int GMODE>
__global__ __launch_bounds__(64 * mf_nw(D), (D == 128 || D == 64) ? BWD_MIN_WG : 1) void loss_bwd_dense_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = BwdLds<D, XU>;
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
    const int64_t x0 = (int64_t)blockIdx.y * TileGeom<D>::XB + wave * 32;     // this wave's X tile (grid = (Y split, X block))
    const int64_t x = x0 + c;
    const int xt = (int)(x0 >> 5);
    const int64_t nX = XU ? p.B : p.N, nY = XU ? p.N : p.B;
    const int64_t Xp = XU ? p.Bp : p.Np;
    const float* Y = XU ? p.v : p.u;
    const int t0 = blockIdx.x * p.tps, t1 = min(p.YT, t0 + p.tps);
    constexpr bool RECOMP = L::RECOMP;
    float xa = 0.f, xb = 0.f, xc = 0.f, xd = 0.f;
    const float gout = p.grad_out[0];
    if (XU) {
        xa = p.rowc[x]; xb = p.rowc[p.Bp + x]; xc = gout * p.rowc[2 * p.Bp + x]; xd = gout * p.rowc[3 * p.Bp + x];
    }
    float xb2 = xb * 1.44269504088896341f;             // exp((L - a) + b) = exp2((L - a) log2e + b log2e): a is the row's largest logit
                                                       // (L - a: exact where it matters), b = -log(sum) is small -- rowc_row
    float cn[4] = {0.f, 0.f, 0.f, 0.f};                // dV, RECOMP: the coefficients of the NEXT tile's user (lane = user c of the tile)
    f32x16 dacc[D / 32];
#pragma unroll
    for (int mb = 0; mb < D / 32; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) dacc[mb][e] = 0.f;
    float rsum = 0.f;

    TileSrc<D> tsrc;
    mf_tile_src_init<D>(tsrc, Y, nY, (int64_t)t0 * 32);
    auto block_of = [&](int t) { return XU ? ((int64_t)xt * p.NT + t) : ((int64_t)t * p.NT + xt); };
    // piece j of the staging of tile t: 0..3 the wave's 4 KiB stash / G' block (into LDS, or -- SPREAD --
    // straight into the registers Gn), 4.. its share of the Y tile
    constexpr bool SPREAD = L::SPREAD;
    f32x4 Gn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) Gn[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto stage_piece = [&](int t, int slot_idx, int j) {
        char* slot = smem + slot_idx * L::SLOT;
        if (L::NCOEF && j >= L::NDMA - L::NCOEF) {
            cn[j - (L::NDMA - L::NCOEF)] = p.rowc[(int64_t)(j - (L::NDMA - L::NCOEF)) * p.Bp + (int64_t)t * 32 + c];
        } else if (j < 4) {
            const char* lsrc = reinterpret_cast<const char*>(((XU || RECOMP) ? p.stash : p.gstash) + block_of(t) * 1024) + lane * 16;
            if (SPREAD) Gn[j] = *reinterpret_cast<const f32x4*>(lsrc + j * 1024);
            else __builtin_amdgcn_global_load_lds((mf_glb_ptr)(lsrc + j * 1024), (mf_lds_ptr)(slot + L::LT + wave * 4096 + j * 1024), 16, 0, 0);
        } else {
            mf_stage_tile_piece<D>(slot, t * 32, j - 4, tsrc);
        }
    };
    auto stage = [&](int t, int slot_idx) {
#pragma unroll
        for (int j = 0; j < L::NDMA; ++j) stage_piece(t, slot_idx, j);
    };
    // Two schedules.  SPREAD (2-slot ring, d >= 64): the loads of tile ty+1 are issued ONE PER MFMA STEP
    // inside the contraction of tile ty, the tile pieces into the slot tile ty-1 left; no second barrier.
    // Otherwise (3 slots, short tiles): whole stages, two tiles ahead.
    if (t0 < t1) stage(t0, 0);
    if (!SPREAD && t0 + 1 < t1) stage(t0 + 1, 1);
    int cur = 0;
    for (int ty = t0; ty < t1; ++ty) {
        if (SPREAD) {
            mf_wait_vmcnt<0>();           // tile ty: issued one tile ago, spread over the previous contraction
        } else {
            // queue, oldest first: [DMA(ty)] [G stores(ty-2)] [DMA(ty+1)] [G stores(ty-1)]   (stores: dU only)
            if (ty + 1 >= t1) mf_wait_vmcnt<0>();
            else if (L::NSTORE == 0 || ty == t0) mf_wait_vmcnt<L::NDMA>();
            else mf_wait_vmcnt<L::NDMA + L::NSTORE>();
        }
        mf_block_barrier();
        if (!SPREAD && L::NSLOT == 3 && ty + 2 < t1) stage(ty + 2, cur >= 1 ? cur - 1 : 2);
        const char* slot = smem + cur * L::SLOT;
        const char* lt = slot + L::LT + wave * 4096;
        float Gv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t4 = SPREAD ? Gn[q] : *reinterpret_cast<const f32x4*>(lt + q * 1024 + lane * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t) Gv[4 * q + t] = t4[t];
        }
        if (!XU && RECOMP) {            // this tile's users' coefficients (asked for one tile ago)
            xa = cn[0]; xb = cn[1]; xc = gout * cn[2]; xd = gout * cn[3];
            xb2 = xb * 1.44269504088896341f;
        }
        if (XU || RECOMP) {
            // masked logits -> G' (masked entries are -inf: exp / step / sigmoid give exactly 0)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (GMODE == G_EXP) Gv[e] = xc * __builtin_amdgcn_exp2f(__builtin_fmaf(Gv[e] - xa, 1.44269504088896341f, xb2));
                else Gv[e] = xc * g_of(GMODE, (Gv[e] - xa) + xb);
            }
            if (ty == xt) {             // only the diagonal tile holds the user's own positive
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (mf_acc_row(e, h) == c) Gv[e] = xd;
            }
        }
        if (XU && !RECOMP) {
            float* blk = p.gstash + block_of(ty) * 1024 + lane * 4;     // hand G' to the dV sweep
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(blk + q * 256) = f32x4{Gv[4 * q], Gv[4 * q + 1], Gv[4 * q + 2], Gv[4 * q + 3]};
        }
        if (!XU) {
            // block layout is (lane = user, register = item row): transpose to (lane = item, register = user row)
            float* tr = reinterpret_cast<float*>(smem + L::TR) + wave * (32 * 32);
#pragma unroll
            for (int e = 0; e < 16; ++e) tr[mf_acc_row(e, h) * 32 + (c ^ mf_acc_row(e, h))] = Gv[e];
#pragma unroll
            for (int e = 0; e < 16; ++e) Gv[e] = tr[c * 32 + (mf_acc_row(e, h) ^ c)];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) rsum += Gv[e];
        // dX[m][x] += sum_y Y[y][m] * G[y][x]   (the G tile is already a B operand)
        // (the Y fragment of step t + 1 is read before the MFMAs of step t are issued)
        const bool more = ty + 1 < t1;
        float yv[D / 32];
        mf_lds_cols<D>(yv, slot, mf_acc_row(0, h), c);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float yn[D / 32];
            if (t + 1 < 16) mf_lds_cols<D>(yn, slot, mf_acc_row(t + 1, h), c);
            if (SPREAD && t < L::NDMA && more) stage_piece(ty + 1, cur ^ 1, t);
#pragma unroll
            for (int j = 0; j < D / 32; ++j)
                dacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv[j], Gv[t], dacc[j], 0, 0, 0);
            if (t + 1 < 16) {
#pragma unroll
                for (int j = 0; j < D / 32; ++j) yv[j] = yn[j];
            }
        }
        if (!SPREAD && L::NSLOT == 2) {            // the slot just read is the one tile ty+2 lands in
            mf_block_barrier();
            if (ty + 2 < t1) stage(ty + 2, cur);
        }
        cur = cur + 1 == L::NSLOT ? 0 : cur + 1;
    }
    mf_wait_vmcnt<0>();                 // nothing of this workgroup may still be on its way into LDS when it ends
    rsum += mf_shfl_xor32(rsum);
    // dX[x][m] = sum over splits of (dacc - rsum * X[x][m]): this sweep writes its partial dacc (register e of
    // block j is feature m = NB * row(e, h) + j: NB consecutive floats per register index) and its partial row sum;
    // sum_parts_kernel adds the splits in order and applies the X term once.
    if (x < nX) {
        constexpr int NB = D / 32;
        float* o = p.dpart + ((int64_t)blockIdx.x * Xp + x) * D;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m0 = NB * mf_acc_row(e, h);
            if constexpr (NB == 4) {
                *reinterpret_cast<f32x4*>(o + m0) = f32x4{dacc[0][e], dacc[1][e], dacc[2][e], dacc[3][e]};
            } else if constexpr (NB == 8) {
                *reinterpret_cast<f32x4*>(o + m0) = f32x4{dacc[0][e], dacc[1][e], dacc[2][e], dacc[3][e]};
                *reinterpret_cast<f32x4*>(o + m0 + 4) = f32x4{dacc[4][e], dacc[5][e], dacc[6][e], dacc[7][e]};
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) o[m0 + j] = dacc[j][e];
            }
        }
        if (h == 0) p.rpart[(int64_t)blockIdx.x * Xp + x] = rsum;
    }
}

// out[r] = sum over splits (in split order) of dpart[s][r]  -  (sum over splits of rpart[s][r]) * X[r],
// for the rows of dU (workgroups [0, nb_a)) and of dV (the rest) in one launch
struct SumJob {
    const float *dpart, *rpart, *X;
    int nsplit;
    int64_t rows, rows_p;
    float* out;
};
__global__ __launch_bounds__(256) void sum_parts_kernel(SumJob ja, SumJob jb, int d, int nb_a) {
    const bool first = (int)blockIdx.x < nb_a;
    const SumJob& j = first ? ja : jb;
    const int64_t t = (int64_t)(blockIdx.x - (first ? 0 : nb_a)) * 256 + threadIdx.x;   // one float4 each
    const int64_t per_row = d / 4;
    if (t >= j.rows * per_row) return;
    const int64_t r = t / per_row, cidx = t % per_row;
    f32x4 acc = reinterpret_cast<const f32x4*>(j.dpart + r * d)[cidx];
    float rs = j.rpart[r];
    for (int s = 1; s < j.nsplit; ++s) {
        acc += reinterpret_cast<const f32x4*>(j.dpart + ((int64_t)s * j.rows_p + r) * d)[cidx];
        rs += j.rpart[(int64_t)s * j.rows_p + r];
    }
    const f32x4 xr = reinterpret_cast<const f32x4*>(j.X + r * d)[cidx];
    reinterpret_cast<f32x4*>(j.out + r * d)[cidx] = acc - rs * xr;
}

// alignment-only backward: only the diagonal carries gradient
template <int D>
__global__ __launch_bounds__(256) void diag_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                       const float* __restrict__ rowc, const float* __restrict__ grad_out,
                                                       int64_t B, int64_t N, int64_t Bp,
                                                       float* __restrict__ du, float* __restrict__ dv) {
    constexpr int LPR = D / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = t / LPR;
    const int c = (int)(t % LPR);
    if (r >= N) return;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (r < B) {
        const float gd = grad_out[0] * rowc[3 * Bp + r];
        const f32x4 ui = reinterpret_cast<const f32x4*>(u + r * D)[c];
        const f32x4 vi = reinterpret_cast<const f32x4*>(v + r * D)[c];
        reinterpret_cast<f32x4*>(du + r * D)[c] = gd * (vi - ui);
        reinterpret_cast<f32x4*>(dv + r * D)[c] = gd * (ui - vi);
    } else {
        reinterpret_cast<f32x4*>(dv + r * D)[c] = z;
    }
}

// Mined backward.  du_i accumulates in registers, in selection order.  dv_j sums contributions of many users:
// they are added as 64-bit FIXED-POINT integers with integer atomics -- integer addition is associative, so the sum is
// the same bits whatever order the atomics land in (run-to-run deterministic, unlike the fp32 atomics this replaces),
// and it is rounded to fp32 once, by dv_fix_to_f32_kernel.  The unit 2^-E is chosen per batch by dv_scale_kernel so
// that the sum cannot wrap (dv_fix_of, mf_loss_math.h): 2^-40 for ordinary batches -- a contribution is then exact on the
// grid down to |x| = 2^-17 and off by <= 2^-41 below.  Lane c of a 32-lane row group owns the features c, c + 32, ..:
// one atomic instruction covers 256 contiguous bytes of a row.
__global__ __launch_bounds__(1024) void dv_scale_kernel(const float* __restrict__ rowc, const float* __restrict__ grad_out,
                                                        const float* __restrict__ nu, const float* __restrict__ nv, int64_t B, int64_t N,
                                                        int64_t Bp, float* __restrict__ dvsc, uint4* __restrict__ zero16, int64_t n16) {
    // (blocks 1 .. clear the fixed-point accumulator: one launch instead of two)
    if (blockIdx.x > 0) {
        const int64_t stride = (int64_t)(gridDim.x - 1) * 1024;
        for (int64_t q = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x; q < n16; q += stride) zero16[q] = uint4{0u, 0u, 0u, 0u};
        return;
    }
    // (maxima of magnitudes as unsigned bit patterns: exact, order-free, and the same rule as the one-launch step's)
    __shared__ unsigned red[3][16];
    const int tid = threadIdx.x, lane = mf_lane(), wave = mf_wave_id();
    const float go = grad_out[0];
    unsigned g = 0u, a = 0u, b = 0u;
    for (int64_t i = tid; i < B; i += 1024) {
        g = max(g, max(dv_mag(go * rowc[2 * Bp + i]), dv_mag(go * rowc[3 * Bp + i])));
        a = max(a, dv_mag(nu[i]));
    }
    for (int64_t j = tid; j < N; j += 1024) b = max(b, dv_mag(nv[j]));
    g = mf_wave_max_u32(g); a = mf_wave_max_u32(a); b = mf_wave_max_u32(b);
    if (lane == 0) { red[0][wave] = g; red[1][wave] = a; red[2][wave] = b; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) { g = max(g, red[0][w]); a = max(a, red[1][w]); b = max(b, red[2][w]); }
        const DvFix f = dv_fix_of(__builtin_bit_cast(float, g), __builtin_bit_cast(float, a), __builtin_bit_cast(float, b), (long long)B);
        dvsc[0] = f.scale; dvsc[1] = f.clamp; dvsc[2] = f.inv;
    }
}

// Round 4: a workgroup is 32 users (1024 threads), and its contributions to dV meet in LDS first -- 64-bit fixed point is
// associative, so ANY grouping gives the same bits.  The mined columns of a Zipf batch are few and hot (the first copies of
// the popular items are picked by thousands of users): 5.2 M global 64-bit atomics, thousands of them on the same words, were
// this kernel's 60 us.  Now a column is claimed in a small LDS table (key by LDS compare-and-swap; a column that loses its
// slot to another goes to global memory as before), the users add with LDS atomics, and the workgroup flushes every
// occupied slot with one global atomic per feature.
template <int D>
struct MinedBwdLds {
    static constexpr int SLOTS = D <= 128 ? 48 : 24;        // 48 KB of accumulators at d = 128 (d x 8 bytes per slot)
    static constexpr int BYTES = SLOTS * D * 8 + SLOTS * 4;
};
template <int D>
__global__ __launch_bounds__(1024) void mined_bwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                         const float* __restrict__ rowc, const int32_t* __restrict__ sel,
                                                         const int32_t* __restrict__ sel_cnt,
                                                         const float* __restrict__ sel_L, const float* __restrict__ grad_out,
                                                         int64_t B, int64_t Bp,
                                                         int gmode, float* __restrict__ du, long long* __restrict__ dvfix,
                                                         const float* __restrict__ dvsc) {
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    using L = MinedBwdLds<D>;
    unsigned long long* lacc = reinterpret_cast<unsigned long long*>(bsm);                  // [SLOTS][D]
    int* lkey = reinterpret_cast<int*>(bsm + L::SLOTS * D * 8);                             // [SLOTS]: the column, -1 = free
    constexpr int LPR = 32, NE = D / LPR;            // lanes per row, features per lane (c, c + 32, ...)
    for (int q = threadIdx.x; q < L::SLOTS * D; q += 1024) lacc[q] = 0ull;
    if (threadIdx.x < L::SLOTS) lkey[threadIdx.x] = -1;
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int64_t i = t / LPR;
    const int c = (int)(t % LPR);
    if (i < B) {
        const float go = grad_out[0];
        const float fix_scale = dvsc[0], fix_clamp = dvsc[1];
        const float a = rowc[i], b = rowc[Bp + i], cg = go * rowc[2 * Bp + i], gd = go * rowc[3 * Bp + i];
        float ui[NE], acc[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) { ui[e] = u[i * D + c + LPR * e]; acc[e] = 0.f; }
        const int n = sel_cnt[i];
        for (int s = -1; s < n; ++s) {
            int64_t j;
            float g;
            if (s < 0) { j = i; g = gd; }
            else { j = sel[i * KSEL_MAX + s]; g = cg * g_of(gmode, (sel_L[i * KSEL_MAX + s] - a) + b); }
            // the user's own column is its alone (one diagonal per column): straight to memory; a mined column through LDS
            int slot = -1;
            if (s >= 0) {
                const int want = (int)(((unsigned)j * 2654435761u) >> 16) % L::SLOTS;
                int got = 0;
                if (c == 0) {
                    const int old = atomicCAS(&lkey[want], -1, (int)j);
                    got = (old == -1 || old == (int)j) ? 1 : 0;
                }
                got = __shfl(got, threadIdx.x & 32, 64);     // (the group's lane 0: lane 0 or 32 of the wave)
                slot = got ? want : -1;
            }
            unsigned long long* o = reinterpret_cast<unsigned long long*>(dvfix) + j * D + c;
            unsigned long long* lo = lacc + (slot < 0 ? 0 : slot) * D + c;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float vj = v[j * D + c + LPR * e];
                acc[e] += g * (vj - ui[e]);
                const float dvj = g * (ui[e] - vj);
                const long long q = dv_fix_term(dvj, fix_scale, fix_clamp);
                if (slot >= 0) atomicAdd(lo + LPR * e, (unsigned long long)q);      // two's complement: the unsigned add IS the signed add
                else if (s >= 0) atomicAdd(o + LPR * e, (unsigned long long)q);
                // (s < 0, the user's own column: ONE such term per column -- dv_fix_to_f32_kernel adds it, no atomic at all)
            }
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) du[i * D + c + LPR * e] = acc[e];
    }
    __syncthreads();
    // flush: one global atomic per occupied slot and feature (zeros are skipped)
    for (int q = threadIdx.x; q < L::SLOTS * D; q += 1024) {
        const int slot = q / D, f = q % D;
        const int j = lkey[slot];
        const unsigned long long val = lacc[q];
        if (j >= 0 && val != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(dvfix) + (int64_t)j * D + f, val);
    }
}

// fixed point -> fp32, and the diagonal terms on the way: column j < B also receives gd_j (u_j - v_j) -- one term per column,
// the same integer the backward kernel used to add atomically
__global__ __launch_bounds__(256) void dv_fix_to_f32_kernel(const long long* __restrict__ dvfix, int64_t n, float* __restrict__ dv,
                                                            const float* __restrict__ dvsc, const float* __restrict__ u,
                                                            const float* __restrict__ v, const float* __restrict__ rowc,
                                                            const float* __restrict__ grad_out, int64_t B, int64_t Bp, int d) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    long long acc = dvfix[t];
    const int64_t j = t / d;
    if (j < B) {
        const float gd = grad_out[0] * rowc[3 * Bp + j];
        acc += dv_fix_term(gd * (u[t] - v[t]), dvsc[0], dvsc[1]);
    }
    dv[t] = (float)((double)acc * (double)dvsc[2]);
}

// ------------------------------------------------------------------ C ABI ------
// negative_masks of the reference (losses.py:92-110) into w.maskW
static void clear_mask_tables(const LossWs& w, hipStream_t s) {     // mf_loss_fwd does this inside prep_kernel
    (void)hipMemsetAsync(w.gtab, 0x80, (size_t)w.M * 8, s);
    (void)hipMemsetAsync(w.gfirst, 0x7f, (size_t)w.M * 4, s);
    (void)hipMemsetAsync(w.bmap, 0, (size_t)4 << (w.bmbits - 5), s);
}
static void prep_clears(PrepParams& pp, const LossWs& w) {
    pp.gtab = reinterpret_cast<uint4*>(w.gtab); pp.gtab16 = (int64_t)w.M * 8 / 16;
    pp.gfirst = reinterpret_cast<uint4*>(w.gfirst); pp.gfirst16 = (int64_t)w.M * 4 / 16;
    pp.ubits = nullptr; pp.ubits16 = 0;                  // (the hit rows are written whole: nothing to clear)
    pp.bmap = reinterpret_cast<uint4*>(w.bmap); pp.bmap16 = ((int64_t)4 << (w.bmbits - 5)) / 16;
}

// pieces a group's flat list range is cut into (= planes of ubits in use): CSR lists are heavy-tailed and unknown to the
// host -> two; padded lists by their width (P = 64: one piece)
static int hits_split(const PosSrc& src) {
    if (src.pos_off) return 2;
    const int per_group = 32 * (src.P + 1);
    int sp = (per_group + 16383) / 16384;
    return sp < 1 ? 1 : (sp > HITS_MAX_SPLIT ? HITS_MAX_SPLIT : sp);
}

// expects gtab / gfirst / bmap cleared (ubits needs no clearing: every word in use is written)
static void build_masks(const LossWs& w, const int64_t* item_idx, const PosSrc& src, int64_t B, int64_t N, hipStream_t s) {
    gt_insert_kernel<<<dim3((unsigned)((N + 63) / 64)), 64, 0, s>>>(item_idx, N, w.M, w.gtab, w.gfirst, w.colslot, w.bmap, w.bmbits);
    const int split = hits_split(src);
    const int nb_col = (int)((w.Np + HITS_THREADS - 1) / HITS_THREADS);
    const int nb_u = (int)(w.Bp / 32) * split;               // (padding groups too: their rows are read by the sweep)
    const int win = (int)(w.Np < HITS_WIN ? w.Np : HITS_WIN);
    const size_t lds = (size_t)win * 4 + ((size_t)4 << (w.bmbits - 5));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)hits_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HITS_WIN * 4 + (4 << 12));
        attr_set = true;
    }
    hits_kernel<<<dim3((unsigned)(nb_col + nb_u)), HITS_THREADS, lds, s>>>(item_idx, src, B, N, w.Bp, w.Np, w.M, nb_col, split, win, w.gtab,
                                                                           w.gfirst, w.colslot, w.colfirst, w.ubits, w.bmap, w.bmbits);
    const int64_t units = (int64_t)w.NT * (w.Bp >> 7);
    mask_sweep_kernel<<<dim3((unsigned)((units + 7) / 8)), 256, 0, s>>>(w.colfirst, w.ubits, B, w.Bp, w.Np, w.NT, w.maskW, split);
}

template <int D, bool XU>
static void launch_bwd(int gmode, dim3 grid, const BwdParams& bp, hipStream_t s);

template <int D, bool XU, int GMODE>
static void launch_bwd_g(dim3 grid, const BwdParams& bp, hipStream_t s) {
    auto fn = loss_bwd_dense_kernel<D, XU, GMODE>;
    if (BwdLds<D, XU>::BYTES > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, BwdLds<D, XU>::BYTES);
    fn<<<grid, 64 * mf_nw(D), BwdLds<D, XU>::BYTES, s>>>(bp);
}
template <int D, bool XU>
static void launch_bwd(int gmode, dim3 grid, const BwdParams& bp, hipStream_t s) {
    if (gmode == G_EXP) launch_bwd_g<D, XU, G_EXP>(grid, bp, s);
    else if (gmode == G_STEP) launch_bwd_g<D, XU, G_STEP>(grid, bp, s);
    else launch_bwd_g<D, XU, G_SIGM>(grid, bp, s);
}

static int check_loss_args(const char* what, int64_t B, int64_t N, int d, int P, int num_negatives,
                           const void* u, const void* v, const void* target, void* ws, size_t ws_bytes) {
    if (B <= 0 || N < B) return mf_set_error(MF_EINVAL, "%s: need 0 < B <= N (B=%lld N=%lld)", what, (long long)B, (long long)N);
    if (!mf_width_ok(d)) return mf_set_error(MF_EINVAL, "%s: embedding width %d not in {32,64,128,256}", what, d);
    if (P < 0 || !u || !v || !target || !ws) return mf_set_error(MF_EINVAL, "%s: bad argument", what);
    if (N >= (1 << 24)) return mf_set_error(MF_ENOTSUP, "%s: N >= 2^24", what);
    if (mining_on(num_negatives, N) && (num_negatives > KSEL_MAX || !mf_select_plan(B, N, d, num_negatives).ok))
        return mf_set_error(MF_ENOTSUP, "%s: mining with num_negatives = %d unsupported (max %d; 32 at d = 256)", what,
                            num_negatives, KSEL_MAX);
    if (ws_bytes < mf_loss_ws_bytes(B, N, d, P, num_negatives))
        return mf_set_error(MF_ENOSPC, "%s: workspace too small (%zu < %zu)", what, ws_bytes,
                            mf_loss_ws_bytes(B, N, d, P, num_negatives));
    return MF_OK;
}

// The hit masks depend on the batch's ids only: built ahead of the forward (on another stream, beside the
// tower gathers) they leave its critical path.  Fills ws's maskW; the forward is then called with
// item_idx = NULL ("masks are in ws").
static int loss_masks_impl(const char* what, int64_t B, int64_t N, int d, int num_negatives, const int64_t* item_idx, const PosSrc& src,
                           void* ws, size_t ws_bytes, mf_stream_t stream) {
    if (B <= 0 || N < B || !mf_width_ok(d) || !item_idx || !ws) return mf_set_error(MF_EINVAL, "%s: bad argument", what);
    if (N >= (1 << 24)) return mf_set_error(MF_ENOTSUP, "%s: N >= 2^24", what);
    if (ws_bytes < mf_loss_ws_bytes(B, N, d, 0, num_negatives)) return mf_set_error(MF_ENOSPC, "%s: workspace too small", what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    LossWs w = loss_ws(ws, B, N, d, 0, num_negatives);
    PrepParams pp{};
    prep_clears(pp, w);
    pp.ticket = w.ticket;
    const int64_t want = (pp.gtab16 + pp.gfirst16 + pp.bmap16 + 64 * 8 - 1) / (64 * 8);
    prep_kernel<0><<<dim3((unsigned)(want < 8192 ? (want > 0 ? want : 1) : 8192)), 64, 0, s>>>(pp);   // clears only
    build_masks(w, item_idx, src, B, N, s);
    return mf_check_launch(what);
}

// The candidate search of the mined losses through the split-bf16 prefilter: the fp32 seeding pass and its bound as before,
// then item plane / user fragments + intervals / ONE scan on the bf16 cores / exact rescoring into the row lists.
static unsigned long long* g_mine_dbg = nullptr;
#ifdef MF_BF3_LAB          // per-kernel spans in lab builds; the product records ONE span over the whole search (spans do not nest)
#define MBF_TIMED(name, s, ...) MF_TIMED(name, s, __VA_ARGS__)
#else
#define MBF_TIMED(name, s, ...) do { __VA_ARGS__; } while (0)
#endif
template <int D>
static void mine_bf_prepare(const LossWs& w, const float* u, const float* v, int64_t B, int64_t N, float sigma, hipStream_t s) {
    const MineBfPlan& m = w.mbf;
    int lab_abl = 0;
#ifdef MF_BF3_LAB
    if (const char* e = getenv("MF_MBF_ABL")) lab_abl = atoi(e);
#endif
    const int item_blocks = (int)((m.Nq * (D / 8) + 255) / 256), user_blocks = (int)((m.Xq * (D / 8) + 255) / 256);
    MineUserFrags uf{u, w.sgn, B, m.Xq, sigma, static_cast<mbf16x8*>(w.mbf_ufrag)};
    MBF_TIMED("mining_items", s, (mine_items_kernel<D><<<dim3((unsigned)(item_blocks + user_blocks)), 256, 0, s>>>(
        v, w.nv, w.logq, w.colfirst, N, m.Nq, sigma, w.mbf_plane, w.mbf_max, w.mbf_rep, w.mbf_copybits, w.mbf_lastcopy,
        __builtin_ctz((unsigned)m.blk), lab_abl, item_blocks, uf)));
}
template <int D>
static void mine_bf_launch(const LossWs& w, const MinedRowParams& fin, const float* u, const float* v, int64_t B, int64_t N, int k, float sigma, hipStream_t s) {
    const MineBfPlan& m = w.mbf;
    {
        MineBound mb{w.seeds, w.plan.seeds_per_row, k, w.nu, w.lii, w.sgn, w.mbf_max, B, m.Xq, sigma, w.gtau, static_cast<f32x4*>(w.mbf_rowk),
                     w.mbf_flag, w.mbf_gate, g_mine_dbg};
        MBF_TIMED("mining_bound", s, (mine_bound_kernel<D, MiningPolicy><<<dim3((unsigned)(m.Xq / 4)), 256, 0, s>>>(mb)));
    }
    {
        MineScan ms{w.mbf_plane, m.Nq, m.NT, m.tpc, static_cast<const mbf16x8*>(w.mbf_ufrag), static_cast<const f32x4*>(w.mbf_rowk), m.Xq,
                    w.mbf_plist, m.lpc * MBF_CAPL, w.mbf_pcnt, w.mbf_spill, w.mbf_spill_cnt, w.mbf_gate, g_mine_dbg, 0};
#ifdef MF_BF3_LAB
        if (const char* e = getenv("MF_MBF_ABL")) ms.abl = atoi(e);
#endif
        auto fn = mine_scan_kernel<D>;
        const int bytes = MineLds<D>::BYTES;
        static bool attr = false;                            // (per instantiation)
        if (!attr) { (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); attr = true; }
        MBF_TIMED("mining_scan", s, (fn<<<dim3((unsigned)m.nchunk, (unsigned)m.gy), 64 * MBF_WAVES, bytes, s>>>(ms)));
    }
    {
        MineRescore mr{u, v, w.nu, w.nv, w.lii, w.sgn, w.logq, w.maskW, B, w.Bp, N, m.Xq, sigma, m.nlists, k,
                       m.lpc, MineRescoreGeom<D>::keys_cap(m.nlists), w.mbf_plist, w.mbf_pcnt, w.mbf_flag, w.mbf_spill, w.mbf_spill_cnt, w.mbf_gate, w.mbf_rep, w.mbf_copybits, w.mbf_lastcopy, m.blk, w.cand, w.cand_cnt, w.plan.rowcap,
                       g_mine_dbg};
        auto fn = mine_rescore_kernel<D, MinedRowFinish>;
        const int bytes = MineRescoreGeom<D>::bytes(m.nlists);      // (at most 8 KB rows + 9.3 KB keys + 1.5 KB: below the 64 KB that need no attribute)
        MBF_TIMED("mining_rescore", s, (fn<<<dim3((unsigned)w.Bp), 64, bytes, s>>>(mr, fin)));
    }
}
static int mine_bf_run(const LossWs& w, const MiningPolicy::Params& mp, const SelectCommon& sc, const MinedRowParams& fin, const float* u, const float* v, int64_t B,
                       int64_t N, int d, int k, float sigma, hipStream_t s) {
#ifdef MF_BF3_LAB
    const bool whole = false;
#else
    const bool whole = mf_timing_on();
#endif
    if (whole) mf_timing_begin("mining_prefilter", s);
    // item plane + user fragments (one launch) -> fp32 seeding pass -> bound + intervals (one launch) -> scan -> rescoring
    if (d == 64) mine_bf_prepare<64>(w, u, v, B, N, sigma, s);
    else mine_bf_prepare<128>(w, u, v, B, N, sigma, s);
    MF_DISPATCH_D(d, { MBF_TIMED("mining_select", s, (mf_select_run<D, MiningPolicy>(w.plan, mp, sc, w.seeds, B, s, true, false, true))); });
    if (d == 64) mine_bf_launch<64>(w, fin, u, v, B, N, k, sigma, s);
    else mine_bf_launch<128>(w, fin, u, v, B, N, k, sigma, s);
    // behind it, gated on the device: a batch the prefilter gave up on (a spill list overflowed -- thousands of exact ties --,
    // a user without a bound, non-finite norms) is answered by the fp32 search; otherwise its workgroups leave at once
    SelectCommon scg = sc;
    scg.gate = w.mbf_gate;
    MF_DISPATCH_D(d, { mf_select_run<D, MiningPolicy>(w.plan, mp, scg, w.seeds, B, s, false, true); });
    if (whole) mf_timing_end("mining_prefilter", s);
    return MF_OK;
}
// tools/lab/mined_timeline.py: [0] candidates rescored, [1] users, [2] users walked exactly, since the last call
extern "C" int mf_probe_mining_prefilter(unsigned long long* out3, int enable) {
    static unsigned long long* buf = nullptr;
    if (!buf) {
        if (hipMalloc(reinterpret_cast<void**>(&buf), 64) != hipSuccess) return -1;
        (void)hipMemset(buf, 0, 64);
    }
    (void)hipDeviceSynchronize();
    if (out3) (void)hipMemcpy(out3, buf, 64, hipMemcpyDeviceToHost);      // (eight counters)
    (void)hipMemset(buf, 0, 64);
    g_mine_dbg = enable ? buf : nullptr;
    return 0;
}
// 1 = the prefilter where it pays (default), 2 = wherever it can serve (tests), 0 = select_kernel only
extern "C" void mf_set_mining_prefilter(int mode) { g_mine_bf_mode = mode <= 0 ? 0 : (mode >= 2 ? 2 : 1); }

static int pos_src_check(const char* what, const PosSrc& src) {
    if (src.pos_off) {
        if (!src.user_ids || !src.pos_items || src.num_users <= 0) return mf_set_error(MF_EINVAL, "%s: CSR positives need user_ids, pos_off, pos_items, num_users > 0", what);
    } else if (src.P < 0 || (src.P > 0 && !src.pos_idx)) {
        return mf_set_error(MF_EINVAL, "%s: bad positives", what);
    }
    return MF_OK;
}

extern "C" int mf_loss_masks(int64_t B, int64_t N, int d, int P, int num_negatives, const int64_t* item_idx,
                             const int64_t* pos_idx, void* ws, size_t ws_bytes, mf_stream_t stream) {
    const PosSrc src{pos_idx, P, nullptr, nullptr, nullptr, 0};
    if (int rc = pos_src_check("mf_loss_masks", src)) return rc;
    return loss_masks_impl("mf_loss_masks", B, N, d, num_negatives, item_idx, src, ws, ws_bytes, stream);
}

extern "C" int mf_loss_masks_csr(int64_t B, int64_t N, int d, int num_negatives, const int64_t* item_idx, const int64_t* user_ids,
                                 const int64_t* pos_off, const int64_t* pos_items, int64_t num_users, void* ws, size_t ws_bytes,
                                 mf_stream_t stream) {
    const PosSrc src{nullptr, 0, user_ids, pos_off, pos_items, num_users};
    if (!pos_off) return mf_set_error(MF_EINVAL, "mf_loss_masks_csr: pos_off is NULL");
    if (int rc = pos_src_check("mf_loss_masks_csr", src)) return rc;
    return loss_masks_impl("mf_loss_masks_csr", B, N, d, num_negatives, item_idx, src, ws, ws_bytes, stream);
}

static int loss_fwd_impl(const char* what, int64_t B, int64_t N, int d, const PosSrc& src, int num_negatives, float sigma, float margin,
                         int kind_mask, const float* u, const float* v, const void* target,
                         const int64_t* item_idx, const float* logq, int64_t logq_rows,
                         int flags, void* ws, size_t ws_bytes, float* out_losses, uint32_t* out_mask_bits,
                         mf_stream_t stream) {
    int rc = check_loss_args(what, B, N, d, 0, num_negatives, u, v, target, ws, ws_bytes);
    if (rc) return rc;
    const bool masks_ready = item_idx == nullptr || (flags & MF_LOSS_MASKS_READY);       // mf_loss_masks ran on this workspace
    if (!out_losses || !(kind_mask & 0x7F)) return mf_set_error(MF_EINVAL, "%s: bad argument", what);
    if (!masks_ready && (rc = pos_src_check(what, src))) return rc;
    if (logq && logq_rows > 0 && !item_idx) return mf_set_error(MF_EINVAL, "%s: a logQ table needs item_idx", what);
    int rowc_kind = -1;
    if (flags & MF_LOSS_ROWC) {
        if (__builtin_popcount(kind_mask & 0x7F) != 1) return mf_set_error(MF_EINVAL, "%s: MF_LOSS_ROWC needs exactly one loss kind", what);
        rowc_kind = __builtin_ctz(kind_mask & 0x7F);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    LossWs w = loss_ws(ws, B, N, d, 0, num_negatives);
    const int need = need_flags(kind_mask);
    const bool scores_needed = (kind_mask & ~(1 << MF_ALIGNMENT)) != 0 || out_mask_bits;

    {
        PrepParams pp{u, v, target, logq, item_idx, logq_rows, (flags & MF_LOSS_TARGET_I64) ? 1 : 0,
                      B, N, w.Bp, w.Np, d, sigma, w.nu, w.nv, w.lii, w.dii, w.sgn, w.logq, w.tgt,
                      nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, w.ticket};
        int nb = (int)((w.Np + 63) / 64);
        if (scores_needed && !masks_ready) prep_clears(pp, w);
        if (scores_needed && w.mined) {
            // the mined path's counters and bounds (gtau .. cand_cnt, and behind them the prefilter's maxima, copy bitmaps, spill
            // cursors and gate: one contiguous range of the workspace) are cleared here too -- it used to be a launch of its own
            pp.ubits = reinterpret_cast<uint4*>(w.gtau);
            pp.ubits16 = (int64_t)(((char*)(w.mbf.ok ? (void*)(w.mbf_gate + 4) : (void*)(w.mbf_max + MBF_MAXSLOTS * MBF_MAXSTRIDE)) - (char*)w.gtau) / 16);
        }
        {
            const int64_t want = (pp.gtab16 + pp.gfirst16 + pp.ubits16 + pp.bmap16 + 64 * 8 - 1) / (64 * 8);       // ~8 stores per thread
            if (want > nb) nb = (int)(want < 8192 ? want : 8192);
        }
        MF_DISPATCH_D(d, { prep_kernel<D><<<dim3((unsigned)nb), 64, 0, s>>>(pp); });
    }
    if (scores_needed && !masks_ready) build_masks(w, item_idx, src, B, N, s);
    // logq is read by whole float4s up to the padded width: prep_kernel keeps a zero-padded copy in ws
    // (all zeros when there is no logQ correction: L - 0 is exact, and the kernels stay branch-free)
    const float* logq_p = w.logq;
    int merge_splits = 0;
    if (scores_needed && !w.mined) {
        FwdParams fp{u, v, w.nu, w.nv, w.lii, w.sgn, logq_p, w.maskW, w.part, w.stash, B, N, w.Bp, w.NT, w.tps_f, need, sigma, margin};
        {   // the three side inputs live in the workspace: one descriptor over their span
            const char *pm = (const char*)w.maskW, *pn = (const char*)w.nv, *pl = (const char*)logq_p;
            const char* lo = pm < pn ? (pm < pl ? pm : pl) : (pn < pl ? pn : pl);
            const char *em = pm + (size_t)w.NT * w.Bp * 4, *en = pn + (size_t)w.NT * 128, *el = pl + (size_t)w.NT * 128;
            const char* hi = em > en ? (em > el ? em : el) : (en > el ? en : el);
            if ((size_t)(hi - lo) > MF_SRD_MAX_BYTES || (size_t)w.tps_f * 32 * d * 4 > MF_SRD_MAX_BYTES)
                return mf_set_error(MF_EINVAL, "mf_loss_fwd: batch x catalog too large for one sweep (mask words beyond 4 GiB)");
            fp.aux_base = lo; fp.aux_mask = (uint32_t)(pm - lo); fp.aux_nv = (uint32_t)(pn - lo); fp.aux_lq = (uint32_t)(pl - lo);
            fp.aux_bytes = (uint32_t)(hi - lo);
        }
        MF_DISPATCH_D(d, {
            dim3 grid((unsigned)w.nsplit_f, (unsigned)(w.BT / w.NW));
            MF_TIMED("loss_fwd_dense", s, (launch_fwd<D>(need, grid, fp, s)));
        });
        merge_splits = w.nsplit_f;
        if (out_mask_bits)
            mask_export_dense_kernel<<<dim3((unsigned)((B * ((N + 31) / 32) + 255) / 256)), 256, 0, s>>>(w.maskW, B, N, w.Bp, (int)((N + 31) / 32), out_mask_bits);
    } else if (scores_needed) {
        MiningPolicy::Params mp{w.nu, w.nv, w.lii, w.sgn, logq_p, w.maskW, w.Bp, N, sigma};
        SelectCommon sc{u, B, v, N, 0, (int)((N + 31) / 32), w.plan.tpc, w.Bp, num_negatives, w.plan.xw, w.gtau, w.priv, w.cand, w.cand_cnt, w.plan.rowcap};
        // (gtau, cand_cnt and the prefilter's cleared range: prep_kernel did it)
        MinedRowParams mr{w.cand, w.cand_cnt, w.plan.rowcap, num_negatives, u, v, w.nu, w.nv, w.lii, w.sgn, logq_p, B, w.Bp, d,
                          sigma, margin, need, w.sel, w.sel_cnt, w.sel_L, w.stats, nullptr};
        if (mine_bf_use(w.mbf) && w.plan.YTa > 0 && w.plan.rowcap >= num_negatives) {
            // (the rescoring waves finish their users themselves; mined_rows_kernel behind them only runs for a batch handed
            // to the fp32 search)
            if (int rc2 = mine_bf_run(w, mp, sc, mr, u, v, B, N, d, num_negatives, sigma, s)) return rc2;
            mr.gate = w.mbf_gate;
        } else {
            MF_DISPATCH_D(d, { MF_TIMED("mining_select", s, (mf_select_run<D, MiningPolicy>(w.plan, mp, sc, w.seeds, B, s))); });
        }
        mined_rows_kernel<<<dim3((unsigned)w.Bp), 64, 0, s>>>(mr);
        if (out_mask_bits) {
            (void)hipMemsetAsync(out_mask_bits, 0, (size_t)B * ((N + 31) / 32) * 4, s);
            mask_export_mined_kernel<<<dim3((unsigned)((B + 255) / 256)), 256, 0, s>>>(w.sel, w.sel_cnt, B, (int)((N + 31) / 32), out_mask_bits);
        }
    } else {
        mf_zero_async(w.stats, (size_t)NSTAT * w.Bp * 4, s);
    }
    finish_kernel<<<dim3((unsigned)((w.Bp + 63) / 64)), 64, 0, s>>>(w.part, merge_splits, B, w.Bp, w.tgt, w.lii, w.dii,
                                                                                       sigma, kind_mask, w.stats, w.rowloss,
                                                                                       w.blockpart, w.ticket, out_losses,
                                                                                       rowc_kind, margin, w.sgn, w.rowc);
    return mf_check_launch(what);
}

extern "C" int mf_loss_fwd(int64_t B, int64_t N, int d, int P, int num_negatives, float sigma, float margin,
                           int kind_mask, const float* u, const float* v, const void* target,
                           const int64_t* item_idx, const int64_t* pos_idx, const float* logq, int64_t logq_rows,
                           int flags, void* ws, size_t ws_bytes, float* out_losses, uint32_t* out_mask_bits,
                           mf_stream_t stream) {
    const PosSrc src{pos_idx, P, nullptr, nullptr, nullptr, 0};
    return loss_fwd_impl("mf_loss_fwd", B, N, d, src, num_negatives, sigma, margin, kind_mask, u, v, target, item_idx, logq, logq_rows,
                         flags, ws, ws_bytes, out_losses, out_mask_bits, stream);
}

extern "C" int mf_loss_fwd_csr(int64_t B, int64_t N, int d, int num_negatives, float sigma, float margin,
                               int kind_mask, const float* u, const float* v, const void* target,
                               const int64_t* item_idx, const int64_t* user_ids, const int64_t* pos_off, const int64_t* pos_items,
                               int64_t num_users, const float* logq, int64_t logq_rows,
                               int flags, void* ws, size_t ws_bytes, float* out_losses, uint32_t* out_mask_bits,
                               mf_stream_t stream) {
    if (!pos_off && !(item_idx == nullptr || (flags & MF_LOSS_MASKS_READY))) return mf_set_error(MF_EINVAL, "mf_loss_fwd_csr: pos_off is NULL");
    const PosSrc src{nullptr, 0, user_ids, pos_off, pos_items, num_users};
    return loss_fwd_impl("mf_loss_fwd_csr", B, N, d, src, num_negatives, sigma, margin, kind_mask, u, v, target, item_idx, logq, logq_rows,
                         flags, ws, ws_bytes, out_losses, out_mask_bits, stream);
}

extern "C" int mf_loss_bwd(int64_t B, int64_t N, int d, int P, int num_negatives, float sigma, float margin,
                           int kind, const float* u, const float* v, int flags, void* ws, size_t ws_bytes,
                           const float* grad_out, float* du, float* dv, mf_stream_t stream) {
    int rc = check_loss_args("mf_loss_bwd", B, N, d, P, num_negatives, u, v, ws, ws, ws_bytes);
    if (rc) return rc;
    if (kind < 0 || kind >= MF_NUM_KINDS || !grad_out || !du || !dv)
        return mf_set_error(MF_EINVAL, "mf_loss_bwd: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    LossWs w = loss_ws(ws, B, N, d, P, num_negatives);
    if (!(flags & MF_LOSS_ROWC))      // else finish_kernel of the forward already wrote them
        rowc_kernel<<<dim3((unsigned)((w.Bp + 255) / 256)), 256, 0, s>>>(w.stats, w.tgt, w.lii, w.sgn, B, w.Bp, kind, sigma, margin, w.rowc);
    const int gmode = gmode_of(kind);
    if (kind == MF_ALIGNMENT) {
        MF_DISPATCH_D(d, {
            const int64_t nthreads = N * (D / 4);
            diag_bwd_kernel<D><<<dim3((unsigned)((nthreads + 255) / 256)), 256, 0, s>>>(u, v, w.rowc, grad_out, B, N, w.Bp, du, dv);
        });
    } else if (w.mined) {
        {
            const int64_t n16 = N * d * 8 / 16;              // (d is a multiple of 32: whole 16-byte units)
            int64_t zb = (n16 + 1023) / 1024;
            if (zb > 1024) zb = 1024;
            dv_scale_kernel<<<dim3((unsigned)(1 + zb)), 1024, 0, s>>>(w.rowc, grad_out, w.nu, w.nv, B, N, w.Bp, w.dvsc, reinterpret_cast<uint4*>(w.dvfix), n16);
        }
        MF_DISPATCH_D(d, {
            const int64_t nthreads = B * 32;
            mined_bwd_kernel<D><<<dim3((unsigned)((nthreads + 1023) / 1024)), 1024, MinedBwdLds<D>::BYTES, s>>>(u, v, w.rowc, w.sel, w.sel_cnt, w.sel_L,
                                                                                                              grad_out, B, w.Bp, gmode, du, w.dvfix, w.dvsc);
        });
        dv_fix_to_f32_kernel<<<dim3((unsigned)((N * d + 255) / 256)), 256, 0, s>>>(w.dvfix, N * d, dv, w.dvsc, u, v, w.rowc, grad_out, B, w.Bp, d);
    } else {
        if ((size_t)w.tps_u * 32 * d * 4 > MF_SRD_MAX_BYTES || (size_t)w.tps_v * 32 * d * 4 > MF_SRD_MAX_BYTES)
            return mf_set_error(MF_ENOTSUP, "mf_loss_bwd: a sweep's share of the batch exceeds 4 GiB (buffer descriptor)");
        BwdParams bp{u, v, w.rowc, grad_out, w.stash, w.gstash, w.dpart, w.rpart, B, N, w.Bp, w.Np, w.NT, 0, 0};
        MF_DISPATCH_D(d, {
            bp.YT = w.NT; bp.tps = w.tps_u;
            MF_TIMED("loss_bwd_du", s, (launch_bwd<D, true>(gmode, dim3((unsigned)w.nsplit_u, (unsigned)(w.BT / w.NW)), bp, s)));
            bp.YT = w.BT; bp.tps = w.tps_v; bp.dpart = w.dpart_v; bp.rpart = w.rpart_v;
            MF_TIMED("loss_bwd_dv", s, (launch_bwd<D, false>(D == 64 ? gmode : G_EXP, dim3((unsigned)w.nsplit_v, (unsigned)(w.NT / w.NW)), bp, s)));
            const SumJob ja{w.dpart, w.rpart, u, w.nsplit_u, B, w.Bp, du}, jb{w.dpart_v, w.rpart_v, v, w.nsplit_v, N, w.Np, dv};
            const int nb_a = (int)((B * (D / 4) + 255) / 256), nb_b = (int)((N * (D / 4) + 255) / 256);
            sum_parts_kernel<<<dim3((unsigned)(nb_a + nb_b)), 256, 0, s>>>(ja, jb, D, nb_a);
        });
    }
    return mf_check_launch("mf_loss_bwd");
}

#ifdef MF_PROBE
extern "C" void mf_probe_mining_counters(unsigned long long* out16, int reset) {   // tools/mined_probe.py
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(mf_sel_dbg), 16 * 8);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mf_sel_dbg), z, 16 * 8);
    }
}
#endif

// ------------------------------------------------- public mask / mining helpers ----
// API parity with the reference's EmbeddingLoss methods on caller-provided tensors:
// negative_masks (losses.py:92-110), hard_mining (:112-132, never called upstream) and
// semi_hard_mining (:134-162) on a MATERIALISED logits matrix.  Not the hot path.
__global__ __launch_bounds__(256) void mask_export_bool_kernel(const uint32_t* __restrict__ maskW, int64_t B, int64_t N,
                                                               int64_t Bp, uint8_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= B * N) return;
    const int64_t i = t / N, j = t % N;
    out[t] = ((maskW[(j >> 5) * Bp + i] >> (j & 31)) & 1u) ? 0 : 1;
}

extern "C" size_t mf_negative_masks_ws_bytes(int64_t B, int64_t N, int P) { return mf_loss_ws_bytes(B, N, 32, P, 1); }

extern "C" int mf_negative_masks(int64_t B, int64_t N, int P, const int64_t* item_idx, const int64_t* pos_idx, void* ws,
                                 size_t ws_bytes, uint8_t* out_mask, mf_stream_t stream) {
    if (B <= 0 || N < B || P < 0 || !item_idx || !ws || !out_mask || (P > 0 && !pos_idx))
        return mf_set_error(MF_EINVAL, "mf_negative_masks: bad argument");
    if (N >= (1 << 24)) return mf_set_error(MF_ENOTSUP, "mf_negative_masks: N >= 2^24");
    if (ws_bytes < mf_negative_masks_ws_bytes(B, N, P)) return mf_set_error(MF_ENOSPC, "mf_negative_masks: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    LossWs w = loss_ws(ws, B, N, 32, P, 1);
    const PosSrc src{pos_idx, P, nullptr, nullptr, nullptr, 0};
    clear_mask_tables(w, s);
    build_masks(w, item_idx, src, B, N, s);
    mask_export_bool_kernel<<<dim3((unsigned)((B * N + 255) / 256)), 256, 0, s>>>(w.maskW, B, N, w.Bp, out_mask);
    return mf_check_launch("mf_negative_masks");
}

// one wave per row: k rounds, each extracting the best remaining valid key (exact, any k)
__global__ __launch_bounds__(64) void mine_logits_kernel(const float* __restrict__ logits, int64_t N, int k, int semi_hard,
                                                         uint8_t* __restrict__ mask) {
    const int64_t i = blockIdx.x;
    const int lane = mf_lane();
    const float* row = logits + i * N;
    uint8_t* mrow = mask + i * N;
    const float lii = row[i];
    unsigned long long prev = ~0ull;
    int taken = 0;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = 0ull;
        for (int64_t j = lane; j < N; j += 64) {
            if (!(mrow[j] & 1)) continue;
            const unsigned long long key = semi_hard ? mf_key_mining(row[j] - lii, (unsigned)j)
                                                     : (((unsigned long long)mf_orderable(row[j]) << 30) | (0x3FFFFFFFu - (unsigned)j));
            if (key < prev && key > best) best = key;
        }
        best = mf_wave_max_u64(best);
        if (best == 0ull) break;
        prev = best;
        ++taken;
        const unsigned col = 0x3FFFFFFFu - (unsigned)(best & 0x3FFFFFFFull);
        if (lane == 0) mrow[col] |= 2;
    }
    __syncthreads();
    (void)taken;
    for (int64_t j = lane; j < N; j += 64) mrow[j] = (mrow[j] & 2) ? 1 : 0;
}

extern "C" int mf_mine_logits(const float* logits, int64_t B, int64_t N, int k, int semi_hard, uint8_t* mask,
                              mf_stream_t stream) {
    if (!logits || !mask || B <= 0 || N < B) return mf_set_error(MF_EINVAL, "mf_mine_logits: bad argument");
    if (N >= (1ll << 30)) return mf_set_error(MF_ENOTSUP, "mf_mine_logits: N >= 2^30");
    if (k <= 0 || k >= N) return MF_OK;       // losses.py:115-119 / :137-141: mining disabled, mask unchanged
    mine_logits_kernel<<<dim3((unsigned)B), 64, 0, static_cast<hipStream_t>(stream)>>>(logits, N, k, semi_hard, mask);
    return mf_check_launch("mf_mine_logits");
}
