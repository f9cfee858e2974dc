// mf_topk_bf3.hip -- exact full-catalog top-k for MANY queries through a bf16 prefilter (gfx950).
//
// fp32 MFMA runs at 1/16 of the bf16 rate on CDNA4, and an exact top-k needs exact scores only for the few rows
// that can be among the k best.  So the catalog is scanned with bf16 operands (v_mfma_f32_32x32x16_bf16, fp32
// accumulation) and a RIGOROUS error bound decides which rows are rescored with the canonical fp32 fmaf chain:
//
//   index      yb = bf16(y) for every catalog row (round to nearest even), ymax = max row norm      (built once)
//   pass A     a(q, y) = xb_q . yb_y for every (query, row); per query only the maximum of every GROUP of 128
//              rows (by lane half: two values per group) is kept -- each is the score of a distinct row, so the
//              k-th largest of them, tau_q, is a lower bound of the k-th largest a(q, .)
//   bound      |a - s| <= eps_q for the exact chain score s (below), so k distinct rows have s >= tau_q - eps_q and
//              every row of the true top k has a >= tau_q - 2 eps_q =: thr_q
//   pass B     the same scan again; rows with a >= thr_q (a few dozen per query) go to lane-private lists
//   final      one wave per query rescoring its candidates with mf_dot_chain (bit for bit the fp32 MFMA element of
//              mf_topk) and selecting the k best 64-bit keys: the result is IDENTICAL to mf_topk's
//
// Error bound.  bf16 keeps 8 significant bits: |xb - x| <= 2^-8 |x| elementwise, so
// |xb yb - x y| <= (2^-7 + 2^-16) |x||y| per product and, by Cauchy-Schwarz, (2^-7 + 2^-16) |x|_2 |y|_2 for the sum;
// the products of bf16 pairs are exact in fp32 and the two fp32 accumulations (the MFMA's, d terms in some order,
// and the chain's) each stay within d 2^-24 sum |terms| (1 + O(d 2^-24)) of the real sum.  With 1 % slack for the
// fp32 evaluation of the norms:  eps_q = 1.01 (2^-7 + 2^-16 + d 2^-22) |x_q|_2 ymax.
//
// A query whose candidates do not fit (more than BF3_LIST rows above thr in one lane's share of a chunk, or more
// than BF3_CAND in all: a zero query, thousands of duplicate rows, a catalog whose best rows all sit in one group)
// is answered by the same wave scanning the whole catalog with the exact chain: slow, but exact.
// Two scans of a catalog HALF the size of the fp32 one, 1/16 of the matrix time each: DMA-bound, not MFMA-bound.
#include "mf_common.h"
#include "mf_select.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int BF3_NS = 4;            // ring slots (units of ST tiles); NS - 1 units in flight
static constexpr int BF3_GROUP = 4;         // tiles per maxima group (128 rows)
static constexpr int BF3_LIST = 16;         // rows per lane-private list (per chunk, query, lane half)
static constexpr int BF3_CAND = 1024;       // candidates rescored per query
static constexpr int BF3_MAXV = 32;         // group maxima per lane in the bound kernel (64 x 32 per query)

__device__ __forceinline__ unsigned short bf3_round(float x) {
    const __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
}

// ------------------------------------------------------------------- index ----
struct Bf3Index {
    unsigned short* plane;      // [N][d] bf16
    float* ymax2;               // max squared row norm (of the fp32 rows)
    size_t total;
};
static Bf3Index bf3_index(void* base, int64_t N, int d) {
    MfArena a(base);
    Bf3Index ix;
    ix.plane = a.take<unsigned short>((size_t)N * d + 64);
    ix.ymax2 = a.take<float>(4);
    ix.total = a.used();
    return ix;
}
extern "C" size_t mf_topk_bf3_index_bytes(int64_t N, int d) {
    if (N <= 0 || !mf_width_ok(d)) return 0;
    return bf3_index(nullptr, N, d).total;
}

// one d/4-lane group per row
template <int D>
__global__ __launch_bounds__(256) void bf3_build_kernel(const float* __restrict__ items, int64_t N, unsigned short* __restrict__ plane,
                                                        float* __restrict__ ymax2) {
    constexpr int LPR = D / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = t / LPR;
    const int c = (int)(t % LPR);
    float ss = 0.f;
    if (r < N) {
        const f32x4 v = reinterpret_cast<const f32x4*>(items + r * D)[c];
        ushort4 o;
        o.x = bf3_round(v[0]); o.y = bf3_round(v[1]); o.z = bf3_round(v[2]); o.w = bf3_round(v[3]);
        reinterpret_cast<ushort4*>(plane + r * D)[c] = o;
        ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int w = LPR / 2; w >= 1; w >>= 1) ss += __shfl_xor(ss, w, 64);
    // (non-negative floats order like their bit patterns)
    if (r < N && c == 0) atomicMax(reinterpret_cast<unsigned*>(ymax2), __builtin_bit_cast(unsigned, ss));
}

extern "C" int mf_topk_bf3_build(const float* items, int64_t N, int d, void* index, size_t index_bytes, mf_stream_t stream) {
    if (!items || !index || N <= 0 || !mf_width_ok(d)) return mf_set_error(MF_EINVAL, "mf_topk_bf3_build: bad argument");
    if (index_bytes < mf_topk_bf3_index_bytes(N, d)) return mf_set_error(MF_ENOSPC, "mf_topk_bf3_build: index buffer too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Bf3Index ix = bf3_index(index, N, d);
    (void)hipMemsetAsync(ix.ymax2, 0, 16, s);
    MF_DISPATCH_D(d, {
        const int64_t threads = N * (D / 4);
        bf3_build_kernel<D><<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>(items, N, ix.plane, ix.ymax2);
    });
    return mf_check_launch("mf_topk_bf3_build");
}

// -------------------------------------------------------------------- plan ----
struct Bf3Plan {
    int NT;             // 32-row tiles of the catalog
    int ST;             // tiles per ring unit
    int upc, nchunk;    // units per chunk (workgroup), chunks
    int gpc;            // maxima groups per chunk
    int nvals;          // group maxima per query: nchunk * gpc * 2
    int64_t Qp;         // queries padded to 128
    int gy;             // query blocks (128 queries)
};
static Bf3Plan bf3_plan(int64_t Q, int64_t N, int d) {
    Bf3Plan p{};
    p.NT = (int)((N + 31) / 32);
    p.ST = d >= 256 ? 1 : 2;                                   // 16 KiB per unit
    p.Qp = (Q + 127) / 128 * 128;
    p.gy = (int)(p.Qp / 128);
    const int units = (p.NT + p.ST - 1) / p.ST;
    int want = (512 + p.gy - 1) / p.gy;                        // two workgroups per CU
    if (want > 128) want = 128;
    if (want > units) want = units;
    if (want < 1) want = 1;
    p.upc = (units + want - 1) / want;
    // whole groups per chunk: a chunk is a multiple of BF3_GROUP tiles
    const int tiles_pc = (p.upc * p.ST + BF3_GROUP - 1) / BF3_GROUP * BF3_GROUP;
    p.upc = tiles_pc / p.ST;
    p.nchunk = (units + p.upc - 1) / p.upc;
    p.gpc = tiles_pc / BF3_GROUP;
    p.nvals = p.nchunk * p.gpc * 2;
    return p;
}

struct Bf3Ws {
    Bf3Plan plan;
    float* gmax;            // [Qp][nvals]
    float* thr;             // [Qp]
    uint32_t* lists;        // [nchunk][Qp][2][BF3_LIST] catalog rows
    int32_t* lcnt;          // [nchunk][Qp][2]
    uint32_t* exclW;        // [NT][Qp]
    size_t total;
};
static Bf3Ws bf3_ws(void* base, int64_t Q, int64_t N, int d) {
    Bf3Ws w{};
    w.plan = bf3_plan(Q, N, d);
    MfArena a(base);
    w.gmax = a.take<float>((size_t)w.plan.Qp * w.plan.nvals);
    w.thr = a.take<float>((size_t)w.plan.Qp);
    w.lists = a.take<uint32_t>((size_t)w.plan.nchunk * w.plan.Qp * 2 * BF3_LIST);
    w.lcnt = a.take<int32_t>((size_t)w.plan.nchunk * w.plan.Qp * 2);
    w.exclW = a.take<uint32_t>((size_t)w.plan.NT * w.plan.Qp);
    w.total = a.used();
    return w;
}
extern "C" size_t mf_topk_bf3_ws_bytes(int64_t Q, int64_t N, int d, int k) {
    if (Q <= 0 || N <= 0 || k <= 0 || !mf_width_ok(d)) return 0;
    return bf3_ws(nullptr, Q, N, d).total;
}

// -------------------------------------------------------------------- scan ----
struct Bf3Scan {
    const float* q;             // [Q][D] fp32 queries
    int64_t Q, Qp;
    const unsigned short* plane;
    int64_t N;
    int NT, upc, gpc, nvals;
    const uint32_t* exclW;      // [NT][Qp]
    float* gmax;                // pass A out
    const float* thr;           // pass B in
    uint32_t* lists;
    int32_t* lcnt;
};

template <int D, int ST, bool EXCL>
struct Bf3Lds {
    using G = TileGeom<D / 2>;                               // a bf16 row is as long as a fp32 row of half the width
    static constexpr int UNITB = ST * G::TILEB;
    static constexpr int AUXW = 1024;                        // per wave and unit: ST x 128 B of exclusion words (+ zero fill)
    static constexpr int AUX0 = BF3_NS * UNITB;
    static constexpr int BYTES = AUX0 + (EXCL ? BF3_NS * G::NW * AUXW : 0);
    static constexpr int SI = ST * G::PPW + (EXCL ? 1 : 0);  // memory instructions per wave and stage
};

// PASS 0: group maxima; PASS 1: candidate lists
template <int D, int ST, bool EXCL, int PASS>
__global__ __launch_bounds__(256, 2) void bf3_scan_kernel(Bf3Scan p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = Bf3Lds<D, ST, EXCL>;
    using G = typename L::G;
    constexpr int KS = D / 16;                               // MFMA steps per tile
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
    const int chunk = blockIdx.x;
    const int64_t x0 = ((int64_t)blockIdx.y * G::NW + wave) * 32;
    const int64_t x = x0 + c;
    const int u0 = chunk * p.upc;
    const int units = (p.NT + ST - 1) / ST;
    const int u1 = min(units, u0 + p.upc);

    // the query's bf16 fragments: step s covers k = 16 s + 8 h .. + 7
    bf16x8 xb[KS];
    {
        const bool ok = x < p.Q;
        const float* xr = p.q + (ok ? x : 0) * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            if (ok) {
                a = *reinterpret_cast<const f32x4*>(xr + 16 * s + 8 * h);
                b = *reinterpret_cast<const f32x4*>(xr + 16 * s + 8 * h + 4);
            }
            xb[s] = bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
        }
    }
    const float thr = (PASS == 1 && x < p.Q) ? p.thr[x] : __builtin_inff();

    TileSrc<D / 2> tsrc;
    mf_tile_src_init<D / 2>(tsrc, reinterpret_cast<const float*>(p.plane), p.N, (int64_t)u0 * ST * 32);
    mf_rsrc_t arsrc;
    uint32_t aoff = 0u, astep = 0u;
    if (EXCL) {
        // exclusion words of a unit: tile j of the unit by lanes 8 j .. 8 j + 7 (128 B = this wave's 32 queries)
        const int part = lane >> 3, l8 = lane & 7;
        aoff = part < ST ? (uint32_t)((((int64_t)(u0 * ST + part)) * p.Qp + x0) * 4) + l8 * 16 : MF_SRD_DEAD;
        astep = part < ST ? (uint32_t)(ST * p.Qp * 4) : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint64_t ab = (uint64_t)p.NT * (uint64_t)p.Qp * 4u;
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.exclW), 0, (int)(ab > MF_SRD_MAX_BYTES ? MF_SRD_MAX_BYTES : ab), 0x00020000);
#else
        (void)arsrc; (void)aoff;
#endif
    }
    auto stage = [&](int u, bool live) {                    // unit u into slot (u - u0) % NS
        char* slot = smem + ((u - u0) % BF3_NS) * L::UNITB;
#pragma unroll
        for (int st = 0; st < ST; ++st)
#pragma unroll
            for (int q = 0; q < G::PPW; ++q) mf_stage_tile_piece<D / 2>(slot + st * G::TILEB, (u * ST + st) * 32, q, tsrc, live);
        if (EXCL) {
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (mf_lds_ptr)(smem + L::AUX0 + (((u - u0) % BF3_NS) * G::NW + wave) * L::AUXW), 16,
                                                     (int)aoff, live ? 0 : (int)MF_SRD_DEAD, 0, 0);
#endif
            aoff += astep;
        }
    };

    float gm = -__builtin_inff();                            // running maximum of the current group (this lane's 16 rows per tile)
    int lc = 0;                                              // PASS 1: entries in this lane's list
    uint32_t* mylist = PASS == 1 ? p.lists + (((int64_t)chunk * p.Qp + x) * 2 + h) * BF3_LIST : nullptr;
    const int tail_tile = (p.N & 31) ? p.NT - 1 : -1;        // its rows past N score 0: never a maximum, never a candidate
    const int tail_rows = (int)(p.N & 31);

    if (u0 < u1) {
#pragma unroll
        for (int j = 0; j < BF3_NS - 1; ++j) stage(u0 + j, u0 + j < u1);
        for (int u = u0; u < u1; ++u) {
            // unit u is older than the NS - 2 stages issued after it (every stage issues SI instructions, live or not)
            mf_wait_vmcnt<(BF3_NS - 2) * L::SI>();
            mf_block_barrier();                              // ... for every wave; and unit u - 1's slot is free
            stage(u + BF3_NS - 1, u + BF3_NS - 1 < u1);
            const char* slot = smem + ((u - u0) % BF3_NS) * L::UNITB;
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                const int t = u * ST + st;
                if (t >= p.NT) break;
                const char* rowp = slot + st * G::TILEB + c * G::ROWB;
                const int sw = G::swz(c);
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(rowp + (((2 * s + h) ^ sw) << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[s], acc, 0, 0, 0);
                }
                uint32_t dead = 0u;                          // rows of this tile that do not count for this lane's query
                if (EXCL) dead = reinterpret_cast<const uint32_t*>(smem + L::AUX0 + (((u - u0) % BF3_NS) * G::NW + wave) * L::AUXW)[st * 32 + c];
                if (t == tail_tile) dead |= ~0u << tail_rows;
                if (PASS == 0) {
                    if (__any(dead != 0u)) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) gm = fmaxf(gm, ((dead >> mf_acc_row(e, h)) & 1u) ? -__builtin_inff() : acc[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; e += 2) gm = fmaxf(gm, fmaxf(acc[e], acc[e + 1]));
                    }
                    const int tl = t - u0 * ST;              // tile of the chunk
                    if ((tl % BF3_GROUP) == BF3_GROUP - 1 || t == p.NT - 1) {
                        if (x < p.Q) p.gmax[x * p.nvals + ((int64_t)chunk * p.gpc + tl / BF3_GROUP) * 2 + h] = gm;
                        gm = -__builtin_inff();
                    }
                } else {
                    float best = acc[0];
#pragma unroll
                    for (int e = 1; e < 16; ++e) best = fmaxf(best, acc[e]);
                    if (__any(best >= thr)) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            if (acc[e] >= thr && !((dead >> mf_acc_row(e, h)) & 1u)) {
                                if (lc < BF3_LIST) mylist[lc] = (uint32_t)t * 32u + (uint32_t)mf_acc_row(e, h);
                                ++lc;
                            }
                        }
                    }
                }
            }
        }
        mf_wait_vmcnt<0>();                                  // nothing of this workgroup may still be on its way into LDS when it ends
    }
    if (PASS == 0) {
        // groups this chunk never reached (short last chunk): no rows
        const int done = u0 < u1 ? (min(u1 * ST, p.NT) - u0 * ST + BF3_GROUP - 1) / BF3_GROUP : 0;
        if (x < p.Q)
            for (int g = done; g < p.gpc; ++g) p.gmax[x * p.nvals + ((int64_t)chunk * p.gpc + g) * 2 + h] = -__builtin_inff();
    } else if (x < p.Q) {
        p.lcnt[((int64_t)chunk * p.Qp + x) * 2 + h] = lc;    // > BF3_LIST: overflow
    }
}

// ------------------------------------------------------------------- bound ----
// one wave per query: thr = (k-th largest group maximum) - 2 eps
__global__ __launch_bounds__(64) void bf3_bound_kernel(const float* __restrict__ gmax, int nvals, int k, const float* __restrict__ q,
                                                       int d, const float* __restrict__ ymax2, float* __restrict__ thr) {
    const int64_t r = blockIdx.x;
    const int lane = mf_lane();
    unsigned v[BF3_MAXV];
#pragma unroll
    for (int j = 0; j < BF3_MAXV; ++j) {
        const int i = lane + 64 * j;
        // -inf (no row) ranks lowest among what can occur; 0 marks "no value"
        v[j] = i < nvals ? mf_orderable(gmax[r * nvals + i]) : 0u;
    }
    const unsigned ninf = mf_orderable(-__builtin_inff());
    unsigned th = 0u;
    for (int b = 31; b >= 0; --b) {                          // largest th with #{v >= th} >= k
        const unsigned cnd = th | (1u << b);
        int cge = 0;
#pragma unroll
        for (int j = 0; j < BF3_MAXV; ++j) cge += __popcll(__ballot(v[j] >= cnd));
        if (cge >= k) th = cnd;
    }
    float ss = 0.f;
    for (int i = lane; i < d; i += 64) ss = __builtin_fmaf(q[r * d + i], q[r * d + i], ss);
    for (int w = 32; w >= 1; w >>= 1) ss += __shfl_xor(ss, w, 64);
    if (lane == 0) {
        const float c = 1.01f * (0x1p-7f + 0x1p-16f + (float)d * 0x1p-22f);
        const float eps = c * sqrtf(ss) * sqrtf(ymax2[0]);
        // fewer than k rows in sight (or a NaN bound): everything is a candidate
        float t = (th <= ninf) ? -__builtin_inff() : mf_unorderable(th) - 2.f * eps;
        if (!(t == t)) t = -__builtin_inff();
        thr[r] = t;
    }
}

// ------------------------------------------------------------------- final ----
struct Bf3Final {
    const float* q;
    const float* items;
    int64_t N;
    int d, k, nchunk;
    int64_t Qp;
    const uint32_t* lists;
    const int32_t* lcnt;
    const uint32_t* exclW;      // NULL: nothing excluded
    int64_t idx_base;
    float* out_scores;
    int64_t* out_idx;
};

__global__ __launch_bounds__(64) void bf3_final_kernel(Bf3Final p) {
    __shared__ unsigned long long keys[BF3_CAND];
    __shared__ unsigned long long win[64], sorted[64];
    __shared__ float xq[256];
    __shared__ int s_n;
    const int64_t r = blockIdx.x;
    const int lane = mf_lane();
    for (int i = lane; i < p.d; i += 64) xq[i] = p.q[r * p.d + i];
    if (lane == 0) s_n = 0;
    __syncthreads();
    // gather the lane-private lists of every chunk: list j = (chunk, half)
    bool overflow = false;
    const int nl = p.nchunk * 2;
    for (int j0 = 0; j0 < nl; j0 += 64) {
        const int j = j0 + lane;
        int cnt = 0;
        const uint32_t* src = nullptr;
        if (j < nl) {
            const int64_t li = ((int64_t)(j >> 1) * p.Qp + r) * 2 + (j & 1);
            cnt = p.lcnt[li];
            src = p.lists + li * BF3_LIST;
        }
        if (cnt > BF3_LIST) overflow = true;
        const int take = min(cnt, BF3_LIST);
        int at = 0;
        if (take > 0) at = atomicAdd(&s_n, take);
        for (int e = 0; e < take; ++e)
            if (at + e < BF3_CAND) keys[at + e] = (unsigned long long)src[e];     // (row ids for now)
    }
    __syncthreads();
    int n = s_n;
    overflow = __any(overflow) || n > BF3_CAND;
    int m;
    if (!overflow) {
        // exact rescoring: one candidate per lane and round
        for (int i = lane; i < n; i += 64) {
            const unsigned row = (unsigned)keys[i];
            const float s = mf_dot_chain(xq, p.items + (int64_t)row * p.d, p.d);
            keys[i] = mf_key_retrieval(s, row);
        }
        __syncthreads();
        m = mf_row_topk<BF3_CAND / 64>(keys, n, p.k, win, sorted);
    } else {
        // the whole catalog by the exact chain, 64 rows a round; the winners so far ride along in win[]
        const unsigned long long below = (1ull << lane) - 1ull;
        int carry = 0;
        for (int64_t base = 0; base < p.N; base += 64) {
            const int64_t row = base + lane;
            unsigned long long v0 = 0ull;
            if (row < p.N) {
                const bool ex = p.exclW && ((p.exclW[(row >> 5) * p.Qp + r] >> (row & 31)) & 1u);
                if (!ex) v0 = mf_key_retrieval(mf_dot_chain(xq, p.items + row * p.d, p.d), (unsigned)row);
            }
            const unsigned long long v1 = lane < carry ? win[lane] : 0ull;
            __syncthreads();
            const int have = __popcll(__ballot(v0 != 0ull)) + carry;
            unsigned long long tau = 1ull;
            if (have > p.k) {
                unsigned long long th = 0ull;
                for (int b = 63; b >= 0; --b) {                // largest th with #{key >= th} >= k (keys are unique)
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
            __syncthreads();
        }
        m = carry;
        if (lane < m) {
            const unsigned long long mine = win[lane];
            int rk = 0;
            for (int qq = 0; qq < m; ++qq) rk += win[qq] > mine ? 1 : 0;
            sorted[rk] = mine;
        }
        __syncthreads();
    }
    if (lane < p.k) {
        if (lane < m) {
            p.out_scores[r * p.k + lane] = mf_key_retrieval_score(sorted[lane]);
            p.out_idx[r * p.k + lane] = p.idx_base + (int64_t)mf_key_retrieval_col(sorted[lane]);
        } else {
            p.out_scores[r * p.k + lane] = -INFINITY;
            p.out_idx[r * p.k + lane] = -1;
        }
    }
}

__global__ __launch_bounds__(256) void bf3_excl_scatter_kernel(const int64_t* __restrict__ excl_off, const int64_t* __restrict__ excl_idx,
                                                               int64_t idx_base, int64_t N, int64_t Qp, uint32_t* __restrict__ exclW) {
    const int64_t r = blockIdx.x;
    for (int64_t e = excl_off[r] + threadIdx.x; e < excl_off[r + 1]; e += 256) {
        const int64_t y = excl_idx[e] - idx_base;
        if (y >= 0 && y < N) atomicOr(&exclW[(y >> 5) * Qp + r], 1u << (y & 31));
    }
}

template <int D, int ST, bool EXCL, int PASS>
static void bf3_launch_scan(const Bf3Plan& pl, const Bf3Scan& sp, hipStream_t s) {
    auto fn = bf3_scan_kernel<D, ST, EXCL, PASS>;
    const int bytes = Bf3Lds<D, ST, EXCL>::BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr_set = true;
    }
    fn<<<dim3((unsigned)pl.nchunk, (unsigned)pl.gy), 256, bytes, s>>>(sp);
}
template <int D, int ST>
static void bf3_run(const Bf3Ws& w, Bf3Scan sp, bool excl, int k, const float* q, const float* ymax2, hipStream_t s) {
    if (excl) bf3_launch_scan<D, ST, true, 0>(w.plan, sp, s); else bf3_launch_scan<D, ST, false, 0>(w.plan, sp, s);
    bf3_bound_kernel<<<dim3((unsigned)sp.Q), 64, 0, s>>>(w.gmax, w.plan.nvals, k, q, D, ymax2, w.thr);
    if (excl) bf3_launch_scan<D, ST, true, 1>(w.plan, sp, s); else bf3_launch_scan<D, ST, false, 1>(w.plan, sp, s);
}

extern "C" int mf_topk_bf3(const float* q, int64_t Q, const float* items, const void* index, int64_t N, int d, int k,
                           const int64_t* excl_off, const int64_t* excl_idx, int64_t idx_base, void* ws, size_t ws_bytes,
                           float* out_scores, int64_t* out_idx, mf_stream_t stream) {
    if (!q || !items || !index || !out_scores || !out_idx || !ws || Q <= 0 || N <= 0)
        return mf_set_error(MF_EINVAL, "mf_topk_bf3: bad argument");
    if (k <= 0 || k > 64) return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: k = %d outside 1..64", k);
    if (d != 64 && d != 128 && d != 256) return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: embedding width %d not in {64,128,256}", d);
    if (N >= (1ll << 31) || idx_base < 0 || idx_base + N > (1ll << 32))
        return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: item indices must fit 32 bits");
    if ((excl_off == nullptr) != (excl_idx == nullptr)) return mf_set_error(MF_EINVAL, "mf_topk_bf3: excl_off/excl_idx mismatch");
    if (ws_bytes < mf_topk_bf3_ws_bytes(Q, N, d, k)) return mf_set_error(MF_ENOSPC, "mf_topk_bf3: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Bf3Ws w = bf3_ws(ws, Q, N, d);
    if (w.plan.nvals > 64 * BF3_MAXV) return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: catalog too long for the bound kernel");
    if ((uint64_t)w.plan.upc * w.plan.ST * 32 * d * 2 > MF_SRD_MAX_BYTES) return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: chunk beyond 4 GiB");
    Bf3Index ix = bf3_index(const_cast<void*>(index), N, d);
    const bool excl = excl_off != nullptr;
    if (excl) {
        (void)hipMemsetAsync(w.exclW, 0, (size_t)w.plan.NT * w.plan.Qp * 4, s);
        bf3_excl_scatter_kernel<<<dim3((unsigned)Q), 256, 0, s>>>(excl_off, excl_idx, idx_base, N, w.plan.Qp, w.exclW);
    }
    Bf3Scan sp{q, Q, w.plan.Qp, ix.plane, N, w.plan.NT, w.plan.upc, w.plan.gpc, w.plan.nvals, w.exclW, w.gmax, w.thr, w.lists, w.lcnt};
    MF_TIMED("topk_bf3", s, {
        if (d == 64) bf3_run<64, 2>(w, sp, excl, k, q, ix.ymax2, s);
        else if (d == 128) bf3_run<128, 2>(w, sp, excl, k, q, ix.ymax2, s);
        else bf3_run<256, 1>(w, sp, excl, k, q, ix.ymax2, s);
        Bf3Final fp{q, items, N, d, k, w.plan.nchunk, w.plan.Qp, w.lists, w.lcnt, excl ? w.exclW : nullptr, idx_base, out_scores, out_idx};
        bf3_final_kernel<<<dim3((unsigned)Q), 64, 0, s>>>(fp);
    });
    return mf_check_launch("mf_topk_bf3");
}
