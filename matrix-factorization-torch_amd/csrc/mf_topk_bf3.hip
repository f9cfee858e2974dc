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
// A query whose candidates do not fit (more than BF3_LCAP rows above thr in one lane's share of a chunk, or more
// than BF3_CAND in all: a zero query, thousands of duplicate rows, a catalog whose best rows all sit in one group)
// is answered by the same wave scanning the whole catalog with the exact chain: slow, but exact.
// Two scans of a catalog HALF the size of the fp32 one, 1/16 of the matrix time each: DMA-bound, not MFMA-bound.
#include <cstdlib>

#include "mf_common.h"
#include "mf_select.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int BF3_NS = 4;            // ring slots (units of ST tiles); NS - 1 units in flight
static constexpr int BF3_GROUP = 4;         // tiles per maxima group (128 rows)
static constexpr int BF3_LIST = 16;         // words per lane-private list (per chunk, query, lane half): the count, then <= 15 rows
static constexpr int BF3_LCAP = BF3_LIST - 1;
static constexpr int BF3_CAND = 1024;       // candidates rescored per query

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
    int XT;             // 32-query tiles per wave (1 or 2): a workgroup scans for 128 XT queries
    int64_t Qp;         // queries padded to 128 XT
    int gy;             // query blocks
};
static Bf3Plan bf3_plan(int64_t Q, int64_t N, int d) {
    Bf3Plan p{};
    p.NT = (int)((N + 31) / 32);
    p.ST = d >= 256 ? 1 : 2;                                   // 16 KiB per unit
    // every block of queries stages the whole catalog through LDS: twice the queries per workgroup, half the traffic
    p.XT = (Q > 256 && d <= 128) ? 2 : 1;
    p.Qp = (Q + 128 * p.XT - 1) / (128 * p.XT) * (128 * p.XT);
    p.gy = (int)(p.Qp / (128 * p.XT));
    const int units = (p.NT + p.ST - 1) / p.ST;
    int want = (512 + p.gy - 1) / p.gy;                        // two workgroups per CU
    if (want > 128) want = 128;                                // (bf3_final_kernel: at most 256 lists per query)
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
    bf16x8* xfrag;          // [Qp / 32][d / 16][64]: the queries as pass A rounded them, in MFMA operand order
    uint32_t* lists;        // [nchunk][Qp][2][BF3_LIST]: word 0 the number of rows found (> BF3_LCAP: overflow), then the rows
    uint32_t* exclW;        // [Qp][NT]: bit r of word t = row 32 t + r is excluded for the query
    size_t total;
};
static Bf3Ws bf3_ws(void* base, int64_t Q, int64_t N, int d) {
    Bf3Ws w{};
    w.plan = bf3_plan(Q, N, d);
    MfArena a(base);
    w.gmax = a.take<float>((size_t)w.plan.Qp * w.plan.nvals);
    w.thr = a.take<float>((size_t)w.plan.Qp);
    w.xfrag = a.take<bf16x8>((size_t)w.plan.Qp * d / 8);
    w.lists = a.take<uint32_t>((size_t)w.plan.nchunk * w.plan.Qp * 2 * BF3_LIST);
    w.exclW = a.take<uint32_t>((size_t)w.plan.NT * w.plan.Qp + 64);
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
    const uint32_t* exclW;      // [Qp][NT]
    float* gmax;                // pass A out
    const float* thr;           // pass B in
    bf16x8* xfrag;              // pass A out (chunk 0), pass B in
    uint32_t* lists;
    int abl;                    // lab knob (MF_BF3_ABL): 1 = no staging, 2 = no arithmetic -- wrong results, for timing only
};

template <int D, int ST, int XT, bool EXCL>
struct Bf3Lds {
    using G = TileGeom<D / 2>;                               // a bf16 row is as long as a fp32 row of half the width
    static constexpr int UNITB = ST * G::TILEB;
    static constexpr int AUXW = XT * 256;                    // per wave and unit: one exclusion word per lane and query tile ([xt][tile of the unit][query])
    static constexpr int AUX0 = BF3_NS * UNITB;
    static constexpr int BYTES = AUX0 + (EXCL ? BF3_NS * G::NW * AUXW : 0);
    static constexpr int SI = ST * G::PPW + (EXCL ? XT : 0); // memory instructions per wave and stage
};

#ifdef BF3_PROBE
// tools/lab/bf3_probe.py: cycles of wave 0 of every workgroup, summed: 0 total, 1 prologue, 2 wait + barrier, 3 stage issue,
// 4 LDS reads + MFMAs, 5 epilogue, 6 tail, 7 workgroups
static __device__ unsigned long long bf3_dbg[2][8];
#define BF3_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define BF3_ADD(i, x) dbg[i] += (x)
#else
#define BF3_T(v)
#define BF3_ADD(i, x)
#endif

// PASS 0: group maxima; PASS 1: candidate lists
template <int D, int ST, int XT, bool EXCL, int PASS>
__global__ __launch_bounds__(256, 2) void bf3_scan_kernel(Bf3Scan p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = Bf3Lds<D, ST, XT, EXCL>;
    using G = typename L::G;
    constexpr int KS = D / 16;                               // MFMA steps per tile
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
    const int chunk = blockIdx.x;
    const int64_t x0 = ((int64_t)blockIdx.y * G::NW + wave) * (32 * XT);     // this wave's XT query tiles
    const int u0 = chunk * p.upc;
    const int units = (p.NT + ST - 1) / ST;
    const int u1 = min(units, u0 + p.upc);
#ifdef BF3_PROBE
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 1};
#endif
    BF3_T(pt0);

    TileSrc<D / 2> tsrc;
    mf_tile_src_init<D / 2>(tsrc, reinterpret_cast<const float*>(p.plane), p.N, (int64_t)u0 * ST * 32);
    mf_rsrc_t arsrc;
    uint32_t aoff[XT], astep = 0u;
    if (EXCL) {
        // exclusion words of a unit: lane (c, j) fetches the word of query x0 + 32 xt + c for tile j of the unit (4-byte DMA)
        // the descriptor starts at the first query of this WORKGROUP (blockIdx only: provably uniform), so a lane's offset
        // stays below 128 XT queries x NT words whatever Q is (the host refuses catalogs beyond that: bf3_excl_fits)
        const int64_t xb0 = (int64_t)blockIdx.y * G::NW * (32 * XT);
#pragma unroll
        for (int xt = 0; xt < XT; ++xt)
            aoff[xt] = h < ST ? (uint32_t)((((int64_t)(x0 - xb0 + 32 * xt + c)) * p.NT + u0 * ST + h) * 4) : MF_SRD_DEAD;
        astep = h < ST ? (uint32_t)(ST * 4) : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint64_t ab = ((uint64_t)p.NT * (uint64_t)(p.Qp - xb0) + 64u) * 4u;
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.exclW + xb0 * p.NT), 0,
                                                  (int)(ab > MF_SRD_MAX_BYTES ? MF_SRD_MAX_BYTES : ab), 0x00020000);
#else
        (void)arsrc; (void)aoff;
#endif
    }
    // memory instruction j (0 .. SI-1) of the staging of unit u into slot (u - u0) % NS
    auto stage_piece = [&](int u, int j, bool live) {
        if (j < ST * G::PPW) {
            char* slot = smem + ((u - u0) % BF3_NS) * L::UNITB;
            const int st = j / G::PPW, q = j % G::PPW;
            mf_stage_tile_piece<D / 2>(slot + st * G::TILEB, (u * ST + st) * 32, q, tsrc, live);
        } else if (EXCL) {
            const int xt = j - ST * G::PPW;
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (mf_lds_ptr)(smem + L::AUX0 + (((u - u0) % BF3_NS) * G::NW + wave) * L::AUXW + xt * 256), 4,
                                                     (int)aoff[xt], live ? 0 : (int)MF_SRD_DEAD, 0, 0);
#endif
            aoff[xt] += astep;
        }
    };
    auto stage = [&](int u, bool live) {
#pragma unroll
        for (int j = 0; j < L::SI; ++j) stage_piece(u, j, live);
    };
    // the first units are on their way while the queries are fetched and rounded
    if (u0 < u1) {
#pragma unroll
        for (int j = 0; j < BF3_NS - 1; ++j) stage(u0 + j, u0 + j < u1 && !(p.abl & 1));
    }

    // the queries' bf16 fragments: step s covers k = 16 s + 8 h .. + 7.  Reading the fp32 rows here (one row per lane:
    // every load instruction touches 64 cache lines) costs ~6 us of address processing per workgroup, so it happens at
    // most once: pass A does it when nothing else ran before (no exclusion lists) and, in chunk 0, leaves the rounded
    // fragments in operand order; with exclusion lists bf3_excl_rows_kernel has written them already.  Everything else
    // reads the fragments back with coalesced 1 KiB loads
    bf16x8 xb[XT][KS];
    float thr[XT];
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) {
        const int64_t x = x0 + 32 * xt + c;
        const bool ok = x < p.Q;
        bf16x8* xf = p.xfrag + ((x0 / 32 + xt) * KS) * 64 + lane;
        if (PASS == 0 && !EXCL) {
            // (padding lanes read the last query: no branch around the loads, and nothing of theirs is ever stored)
            const float* xr = p.q + (ok ? x : p.Q - 1) * D;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * s + 8 * h);
                const f32x4 b = *reinterpret_cast<const f32x4*>(xr + 16 * s + 8 * h + 4);
                xb[xt][s] = bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
            }
            if (chunk == 0) {
#pragma unroll
                for (int s = 0; s < KS; ++s) xf[s * 64] = xb[xt][s];
            }
        } else {
            const bf16x8 zero8 = {};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 f = xf[s * 64];
                xb[xt][s] = ok ? f : zero8;         // (the fragments of padding queries were never written)
            }
        }
        thr[xt] = (PASS == 1 && ok) ? p.thr[x] : __builtin_inff();      // (bf3_bound_kernel: already nudged below the bound)
    }

    float gm[XT];                                            // running maximum of the current group (this lane's 16 rows per tile)
    int lc[XT];                                              // PASS 1: entries in this lane's lists
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) { gm[xt] = -__builtin_inff(); lc[xt] = 0; }
    const int tail_tile = (p.N & 31) ? p.NT - 1 : -1;        // its rows past N score 0: never a maximum, never a candidate
    const int tail_rows = (int)(p.N & 31);

    BF3_T(pt1);
    BF3_ADD(1, pt1 - pt0);
    if (u0 < u1) {
        for (int u = u0; u < u1; ++u) {
            BF3_T(pa);
            // unit u is older than the NS - 2 stages issued after it (every stage issues SI instructions, live or not)
            mf_wait_vmcnt<(BF3_NS - 2) * L::SI>();
            mf_block_barrier();                              // ... for every wave; and unit u - 1's slot is free
            BF3_T(pb);
            const bool live_n = u + BF3_NS - 1 < u1 && !(p.abl & 1);
            BF3_T(pc);
            BF3_ADD(2, pb - pa);
            BF3_ADD(3, pc - pb);
            if (p.abl & 2) { stage(u + BF3_NS - 1, live_n); continue; }
            const char* slot = smem + ((u - u0) % BF3_NS) * L::UNITB;
            // every catalog fragment of the unit first (one LDS latency per unit, not per step), then its MFMAs
            bf16x8 afr[ST][KS];
            {
                const char* rowp = slot + c * G::ROWB;
                const int sw = G::swz(c);
#pragma unroll
                for (int st = 0; st < ST; ++st)
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        afr[st][s] = *reinterpret_cast<const bf16x8*>(rowp + st * G::TILEB + (((2 * s + h) ^ sw) << 4));
            }
            // the staging of unit u + NS - 1 (into the slot unit u - 1 left) goes out one instruction per few MFMAs
            f32x16 accs[ST][XT];
            constexpr int NMF = ST * XT * KS, GAP = NMF / L::SI;
            int issued = 0;
#pragma unroll
            for (int st = 0; st < ST; ++st)
#pragma unroll
                for (int xt = 0; xt < XT; ++xt) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) accs[st][xt][e] = 0.f;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const int i = (st * XT + xt) * KS + s;
                        if (i % GAP == 0 && i / GAP < L::SI) { stage_piece(u + BF3_NS - 1, i / GAP, live_n); ++issued; }
                        accs[st][xt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[st][s], xb[xt][s], accs[st][xt], 0, 0, 0);
                    }
                }
            static_assert(GAP >= 1, "more staging instructions than MFMAs");
            (void)issued;
#ifdef BF3_PROBE
            asm volatile("s_nop 0" :: "v"(accs[ST - 1][XT - 1][15]));       // (the stamp waits for the last MFMA)
#endif
            BF3_T(pd);
            BF3_ADD(4, pd - pc);
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                const int t = u * ST + st;
                if (t >= p.NT) break;
                const f32x16 (&acc)[XT] = accs[st];
                const int tl = t - u0 * ST;                  // tile of the chunk
#pragma unroll
                for (int xt = 0; xt < XT; ++xt) {
                    const int64_t x = x0 + 32 * xt + c;
                    uint32_t dead = 0u;                      // rows of this tile that do not count for this lane's query
                    if (EXCL) dead = reinterpret_cast<const uint32_t*>(smem + L::AUX0 + (((u - u0) % BF3_NS) * G::NW + wave) * L::AUXW + xt * 256)[st * 32 + c];
                    if (PASS == 0) {
                        // (v_med3(a, b, +inf) = max(a, b) on finite scores, without the NaN-quieting moves fmaxf asks for)
                        if (EXCL && __any(dead != 0u)) {
#pragma unroll
                            for (int e = 0; e < 16; ++e)
                                gm[xt] = __builtin_amdgcn_fmed3f(gm[xt], ((dead >> mf_acc_row(e, h)) & 1u) ? -__builtin_inff() : acc[xt][e], __builtin_inff());
                        } else {
#pragma unroll
                            for (int e = 0; e < 16; ++e) gm[xt] = __builtin_amdgcn_fmed3f(gm[xt], acc[xt][e], __builtin_inff());
                        }
                        if ((tl % BF3_GROUP) == BF3_GROUP - 1 || t == p.NT - 1) {
                            // the zero rows past N in the last tile are not rows: its group tells the bound nothing
                            if (x < p.Q) p.gmax[x * p.nvals + ((int64_t)chunk * p.gpc + tl / BF3_GROUP) * 2 + h] = (t == tail_tile) ? -__builtin_inff() : gm[xt];
                            gm[xt] = -__builtin_inff();
                        }
                    } else {
                        // bit (15 - e) of hm: element e is above the bound.  Two instructions per element: the sign of
                        // thr - score is shifted in (v_alignbit); "score > thr" instead of ">=" is why thr is nudged down below
                        uint32_t hm = 0u;
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            hm = __builtin_amdgcn_alignbit(hm, __builtin_bit_cast(uint32_t, thr[xt] - acc[xt][e]), 31);
                        if (__any(hm != 0u)) {
                            if (t == tail_tile) dead |= ~0u << tail_rows;
                            uint32_t* mylist = p.lists + (((int64_t)chunk * p.Qp + x) * 2 + h) * BF3_LIST;
                            while (hm) {
                                const int e = 15 - __builtin_ctz(hm);
                                hm &= hm - 1;
                                const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                                if (!((dead >> rr) & 1u)) {
                                    if (lc[xt] < BF3_LCAP) mylist[1 + lc[xt]] = (uint32_t)t * 32u + (uint32_t)rr;
                                    ++lc[xt];
                                }
                            }
                        }
                    }
                }
            }
        }
        mf_wait_vmcnt<0>();                                  // nothing of this workgroup may still be on its way into LDS when it ends
    }
#ifdef BF3_PROBE
    {
        BF3_T(pe);
        dbg[0] = pe - pt0;
        dbg[6] = pe - pt1 - dbg[2] - dbg[3] - dbg[4];
        if (threadIdx.x == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&bf3_dbg[PASS][i], dbg[i]);
    }
#endif
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) {
        const int64_t x = x0 + 32 * xt + c;
        if (x >= p.Q) continue;
        if (PASS == 0) {
            // groups this chunk never reached (short last chunk): no rows
            const int done = u0 < u1 ? (min(u1 * ST, p.NT) - u0 * ST + BF3_GROUP - 1) / BF3_GROUP : 0;
            for (int g = done; g < p.gpc; ++g) p.gmax[x * p.nvals + ((int64_t)chunk * p.gpc + g) * 2 + h] = -__builtin_inff();
        } else {
            p.lists[(((int64_t)chunk * p.Qp + x) * 2 + h) * BF3_LIST] = (uint32_t)lc[xt];      // > BF3_LCAP: overflow
        }
    }
}

// ------------------------------------------------------------------- bound ----
// one wave per query: thr = (a lower bound of the k-th largest group maximum) - 2 eps.  Every lane reduces its
// share of the maxima to its two best; the k-th largest of those 128 values (all scores of distinct rows) is at
// most a few ranks below the exact k-th largest, at a tenth of the cost of searching all of them.
__global__ __launch_bounds__(64) void bf3_bound_kernel(const float* __restrict__ gmax, int nvals, int k, const float* __restrict__ q,
                                                       int d, const float* __restrict__ ymax2, float* __restrict__ thr) {
    const int64_t r = blockIdx.x;
    const int lane = mf_lane();
    const unsigned ninf = mf_orderable(-__builtin_inff());
    unsigned m1 = 0u, m2 = 0u;                               // 0: "no value" (ranks below -inf)
    for (int i0 = 0; i0 < nvals; i0 += 64 * 8) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + lane + 64 * j;
            f[j] = i < nvals ? gmax[r * nvals + i] : -__builtin_inff();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned v = (i0 + lane + 64 * j) < nvals ? mf_orderable(f[j]) : 0u;
            m2 = max(m2, min(m1, v));
            m1 = max(m1, v);
        }
    }
    unsigned th = 0u;
    for (int b = 31; b >= 0; --b) {                          // largest th with #{v >= th} >= k
        const unsigned cnd = th | (1u << b);
        const int cge = __popcll(__ballot(m1 >= cnd)) + __popcll(__ballot(m2 >= cnd));
        if (cge >= k) th = cnd;
    }
    float ss = 0.f;
    for (int i = lane; i < d; i += 64) ss = __builtin_fmaf(q[r * d + i], q[r * d + i], ss);
    for (int w = 32; w >= 1; w >>= 1) ss += __shfl_xor(ss, w, 64);
    if (lane == 0) {
        const float c = 1.01f * (0x1p-7f + 0x1p-16f + (float)d * 0x1p-22f);
        const float eps = c * sqrtf(ss) * sqrtf(ymax2[0]);
        // fewer than k rows in sight (or a NaN bound): everything is a candidate
        // the scan tests "score > thr": thr sits strictly below the bound (one part in 2^22, and past zero)
        float t = (th <= ninf) ? -__builtin_inff() : mf_unorderable(th) - 2.f * eps;
        t = t - fabsf(t) * 0x1p-22f - 1e-37f;
        if (!(t == t)) t = -__builtin_inff();
        thr[r] = t;
    }
}

// ------------------------------------------------------------------- final ----
struct Bf3Final {
    const float* q;
    const float* items;
    int64_t N;
    int d, k, nchunk, NT;
    int64_t Qp;
    const uint32_t* lists;
    const uint32_t* exclW;      // NULL: nothing excluded
    int64_t idx_base;
    float* out_scores;
    int64_t* out_idx;
};

// the canonical chain (mf_dot_chain's order) with up to 128 floats of the row in flight at a time
template <int D>
__device__ __forceinline__ float bf3_exact_dot(const float* xq, const float* __restrict__ row) {
    constexpr int B = D < 128 ? D : 128;
    float acc = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < D; g0 += B) {
        f32x4 y[B / 4];
#pragma unroll
        for (int j = 0; j < B / 4; ++j) y[j] = *reinterpret_cast<const f32x4*>(row + g0 + 4 * j);
#pragma unroll
        for (int g = 0; g < B; g += 8)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc = __builtin_fmaf(xq[g0 + g + t], y[g / 4][t], acc);
                acc = __builtin_fmaf(xq[g0 + g + 4 + t], y[g / 4 + 1][t], acc);
            }
    }
    return acc;
}

template <int D>
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
    // (nchunk <= 128: at most four lists per lane; their heads -- the count and the first three rows -- in one trip)
    uint4 head[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = jj * 64 + lane;
        head[jj] = uint4{0u, 0u, 0u, 0u};
        if (j < nl) head[jj] = *reinterpret_cast<const uint4*>(p.lists + (((int64_t)(j >> 1) * p.Qp + r) * 2 + (j & 1)) * BF3_LIST);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = jj * 64 + lane;
        const int cnt = (int)head[jj].x;
        if (cnt > BF3_LCAP) overflow = true;
        const int take = min(cnt, BF3_LCAP);
        int at = 0;
        if (take > 0) at = atomicAdd(&s_n, take);
        const uint32_t* src = p.lists + (((int64_t)(j >> 1) * p.Qp + r) * 2 + (j & 1)) * BF3_LIST;
        for (int e = 0; e < take; ++e) {
            const uint32_t row = e == 0 ? head[jj].y : e == 1 ? head[jj].z : e == 2 ? head[jj].w : src[1 + e];
            if (at + e < BF3_CAND) keys[at + e] = (unsigned long long)row;     // (row ids for now)
        }
    }
    __syncthreads();
    int n = s_n;
    overflow = __any(overflow) || n > BF3_CAND;
    int m;
    if (!overflow) {
        // exact rescoring: one candidate per lane and round
        for (int i = lane; i < n; i += 64) {
            const unsigned row = (unsigned)keys[i];
            const float s = bf3_exact_dot<D>(xq, p.items + (int64_t)row * D);
            keys[i] = mf_key_retrieval(s, row);
        }
        __syncthreads();
        if (n <= 64) m = mf_row_topk<1>(keys, n, p.k, win, sorted);
        else if (n <= 256) m = mf_row_topk<4>(keys, n, p.k, win, sorted);
        else m = mf_row_topk<BF3_CAND / 64>(keys, n, p.k, win, sorted);
    } else {
        // the whole catalog by the exact chain, 64 rows a round; the winners so far ride along in win[]
        const unsigned long long below = (1ull << lane) - 1ull;
        int carry = 0;
        for (int64_t base = 0; base < p.N; base += 64) {
            const int64_t row = base + lane;
            unsigned long long v0 = 0ull;
            if (row < p.N) {
                const bool ex = p.exclW && ((p.exclW[r * p.NT + (row >> 5)] >> (row & 31)) & 1u);
                if (!ex) v0 = mf_key_retrieval(bf3_exact_dot<D>(xq, p.items + row * D), (unsigned)row);
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

// one workgroup per query: its exclusion list -> its row of bit words, through an LDS window (no memset, no global atomics)
static constexpr int BF3_EXCL_WIN = 8192;       // words per window (262,144 catalog rows)
__global__ __launch_bounds__(256) void bf3_excl_rows_kernel(const int64_t* __restrict__ excl_off, const int64_t* __restrict__ excl_idx,
                                                            int64_t idx_base, int64_t N, int NT, uint32_t* __restrict__ exclW,
                                                            const float* __restrict__ q, int d, bf16x8* __restrict__ xfrag) {
    __shared__ uint32_t win[BF3_EXCL_WIN];
    const int64_t r = blockIdx.x;
    // this query's bf16 fragments in MFMA operand order (lane (c, h) of tile r / 32, step s: k = 16 s + 8 h .. + 7)
    if ((int)threadIdx.x < d / 8) {
        const int s = threadIdx.x >> 1, h = threadIdx.x & 1;
        const f32x4 a = *reinterpret_cast<const f32x4*>(q + r * d + 16 * s + 8 * h);
        const f32x4 b = *reinterpret_cast<const f32x4*>(q + r * d + 16 * s + 8 * h + 4);
        xfrag[((r >> 5) * (d / 16) + s) * 64 + h * 32 + (r & 31)] =
            bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    }
    const int64_t e0 = excl_off[r], e1 = excl_off[r + 1];
    for (int w0 = 0; w0 < NT; w0 += BF3_EXCL_WIN) {
        const int nw = min(BF3_EXCL_WIN, NT - w0);
        for (int i = threadIdx.x; i < nw; i += 256) win[i] = 0u;
        __syncthreads();
        for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
            const int64_t y = excl_idx[e] - idx_base;
            const int64_t wd = (y >> 5) - w0;
            if (y >= 0 && y < N && wd >= 0 && wd < nw) atomicOr(&win[wd], 1u << (y & 31));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nw; i += 256) exclW[r * NT + w0 + i] = win[i];
        __syncthreads();
    }
}

template <int D, int ST, int XT, bool EXCL, int PASS>
static void bf3_launch_scan(const Bf3Plan& pl, const Bf3Scan& sp, hipStream_t s) {
    auto fn = bf3_scan_kernel<D, ST, XT, EXCL, PASS>;
    const int bytes = Bf3Lds<D, ST, XT, EXCL>::BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr_set = true;
    }
    fn<<<dim3((unsigned)pl.nchunk, (unsigned)pl.gy), 256, bytes, s>>>(sp);
}
template <int D, int ST, int XT>
static void bf3_run_xt(const Bf3Ws& w, const Bf3Scan& sp, bool excl, int k, const float* q, const float* ymax2, hipStream_t s) {
    if (excl) bf3_launch_scan<D, ST, XT, true, 0>(w.plan, sp, s); else bf3_launch_scan<D, ST, XT, false, 0>(w.plan, sp, s);
    bf3_bound_kernel<<<dim3((unsigned)sp.Q), 64, 0, s>>>(w.gmax, w.plan.nvals, k, q, D, ymax2, w.thr);
    if (excl) bf3_launch_scan<D, ST, XT, true, 1>(w.plan, sp, s); else bf3_launch_scan<D, ST, XT, false, 1>(w.plan, sp, s);
}
template <int D, int ST>
static void bf3_run(const Bf3Ws& w, const Bf3Scan& sp, bool excl, int k, const float* q, const float* ymax2, hipStream_t s) {
    if (w.plan.XT == 2) {
        if constexpr (D <= 128) bf3_run_xt<D, ST, 2>(w, sp, excl, k, q, ymax2, s);
    } else {
        bf3_run_xt<D, ST, 1>(w, sp, excl, k, q, ymax2, s);
    }
}

#ifdef BF3_PROBE
extern "C" void mf_probe_bf3(unsigned long long* out16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(bf3_dbg), 16 * 8);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(bf3_dbg), z, 16 * 8);
    }
}
#endif

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
    // geometry limits first (host arithmetic only: nothing below this point may run for an unsupported shape)
    const Bf3Plan plan = bf3_plan(Q, N, d);
    if ((uint64_t)plan.upc * plan.ST * 32 * d * 2 > MF_SRD_MAX_BYTES) return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: chunk beyond 4 GiB");
    // exclusion words are staged through a 32-bit buffer descriptor based at a workgroup's first query: its 128 XT query
    // rows of NT words must fit (NT < ~4 M tiles: 134 M catalog rows per shard)
    if (excl_off && ((uint64_t)plan.NT * (uint64_t)(128 * plan.XT) + 64u) * 4u > MF_SRD_MAX_BYTES)
        return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: exclusion words of one query block beyond 4 GiB (use mf_topk)");
    if (ws_bytes < mf_topk_bf3_ws_bytes(Q, N, d, k)) return mf_set_error(MF_ENOSPC, "mf_topk_bf3: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Bf3Ws w = bf3_ws(ws, Q, N, d);
    Bf3Index ix = bf3_index(const_cast<void*>(index), N, d);
    const bool excl = excl_off != nullptr;
    if (excl) bf3_excl_rows_kernel<<<dim3((unsigned)Q), 256, 0, s>>>(excl_off, excl_idx, idx_base, N, w.plan.NT, w.exclW, q, d, w.xfrag);
    Bf3Scan sp{q, Q, w.plan.Qp, ix.plane, N, w.plan.NT, w.plan.upc, w.plan.gpc, w.plan.nvals, w.exclW, w.gmax, w.thr, w.xfrag, w.lists,
               getenv("MF_BF3_ABL") ? atoi(getenv("MF_BF3_ABL")) : 0};
    MF_TIMED("topk_bf3", s, {
        if (d == 64) bf3_run<64, 2>(w, sp, excl, k, q, ix.ymax2, s);
        else if (d == 128) bf3_run<128, 2>(w, sp, excl, k, q, ix.ymax2, s);
        else bf3_run<256, 1>(w, sp, excl, k, q, ix.ymax2, s);
        Bf3Final fp{q, items, N, d, k, w.plan.nchunk, w.plan.NT, w.plan.Qp, w.lists, excl ? w.exclW : nullptr, idx_base, out_scores, out_idx};
        if (d == 64) bf3_final_kernel<64><<<dim3((unsigned)Q), 64, 0, s>>>(fp);
        else if (d == 128) bf3_final_kernel<128><<<dim3((unsigned)Q), 64, 0, s>>>(fp);
        else bf3_final_kernel<256><<<dim3((unsigned)Q), 64, 0, s>>>(fp);
    });
    return mf_check_launch("mf_topk_bf3");
}
