// mf_topk_small.hip -- exact full-catalog top-k for a HANDFUL of queries (Q <= 32, typically 1), gfx950.
//
// The reference's retrieval surface is one query per call (ItemProcessor.search,
// xfmr_rec/data/lightning.py:237-259, reached from recommend, xfmr_rec/lightning.py:76-95).  For one query the
// scan is a matrix-VECTOR product: 4 N d bytes streamed once, 2 N d flop -- bandwidth-bound, and the MFMA tile
// engine of mf_topk.hip (32 queries per tile, several launches) is the wrong tool.  Here:
//
//   * the index keeps a second, BLOCKED copy of the catalog (built once, mf_topk_blocked_build):
//     [block of 64 rows][16-byte chunk j][row r] -- so that with lane = row a wave's load of chunk j of its 64
//     rows is ONE coalesced 1 KiB access straight into registers (no LDS round trip, all 32 .. 64 loads of a
//     block in flight at once);
//   * every lane runs the canonical k-ordered fmaf chain (mf_numerics.h: bit-for-bit the MFMA's element) of its
//     row against the queries, which are wave-uniform and come through the scalar cache as SGPR operands;
//   * launch 1 leaves, per query, every row's score (4 bytes: < 1 % of the catalog's traffic) and every 64-row
//     block's best key; it selects nothing -- a selection is a serial 32-step search, and almost no block can
//     hold one of the k best rows;
//   * launch 2 (one 1024-thread workgroup per query) picks the k best BLOCK maxima: only those blocks can hold
//     a row of the top k and the k-th of them bounds the k-th best key from below; their <= 64 k scores are
//     filtered against that bound and one wave selects, orders and writes the k winners.
// Exclusion lists (the `movie_id NOT IN (...)` prefilter, data/lightning.py:250-253) are matched per block
// against the query's id list -- a few hundred ids -- instead of scattering a bitmap first.
// Results are bit-identical to mf_topk (same chain, same 64-bit keys): tests/test_gpu_parity.py.
#include "mf_common.h"
#include "mf_select.h"

static constexpr int SQ_MAXQ = 32;       // queries per call
static constexpr int SQ_NW = 4;          // waves per workgroup (each takes one 64-row block per pass)

// ------------------------------------------------------------- blocked catalog ---
template <int D>
__global__ __launch_bounds__(256) void blocked_build_kernel(const float* __restrict__ items, int64_t N, int64_t nblocks,
                                                            float* __restrict__ out) {
    constexpr int CPR = D / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte chunk each: t = (b * CPR + j) * 64 + r
    if (t >= nblocks * CPR * 64) return;
    const int r = (int)(t & 63);
    const int j = (int)((t >> 6) % CPR);
    const int64_t b = (t >> 6) / CPR;
    const int64_t row = b * 64 + r;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (row < N) x = reinterpret_cast<const f32x4*>(items + row * D)[j];
    reinterpret_cast<f32x4*>(out)[t] = x;
}

extern "C" size_t mf_topk_blocked_bytes(int64_t N, int d) {
    if (N <= 0 || !mf_width_ok(d)) return 0;
    return (size_t)((N + 63) / 64) * 64 * d * 4;
}

extern "C" int mf_topk_blocked_build(const float* items, int64_t N, int d, float* out_blocked, mf_stream_t stream) {
    if (!items || !out_blocked || N <= 0) return mf_set_error(MF_EINVAL, "mf_topk_blocked_build: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nblocks = (N + 63) / 64;
    MF_DISPATCH_D(d, {
        const int64_t chunks = nblocks * (D / 4) * 64;
        blocked_build_kernel<D><<<dim3((unsigned)((chunks + 255) / 256)), 256, 0, s>>>(items, N, nblocks, out_blocked);
    });
    return mf_check_launch("mf_topk_blocked_build");
}

// ------------------------------------------------------------ wave selection ----
// Exact top-k of the 64 NV keys a wave holds in registers (0 = empty slot; keys are unique): the k-th largest
// is found by bit-wise threshold search (32 steps on the rank word; 32 more on the column word only when equal
// ranks straddle the cut), the winners go, unordered, to dst[0 .. m), m = min(k, #keys) is returned.
template <int NV>
__device__ __forceinline__ int sq_wave_topk(const unsigned long long (&v)[NV], int k, unsigned long long* dst) {
    const int lane = mf_lane();
    const unsigned long long below = (1ull << lane) - 1ull;
    int have = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) have += __popcll(__ballot(v[j] != 0ull));
    unsigned long long tau = 1ull;                     // keeps every real key
    if (have > k) {
        // bits on which all the keys' rank words agree need no step: an all-ones bit is always accepted, an all-zeros
        // bit never (the decision of a step does not depend on accepted lower all-ones bits: every key has them)
        unsigned all_or = 0u, all_and = ~0u;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const unsigned r = (unsigned)(v[j] >> 32);
            all_or |= r;
            all_and &= v[j] != 0ull ? r : ~0u;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            all_or |= (unsigned)__shfl_xor((int)all_or, m, 64);
            all_and &= (unsigned)__shfl_xor((int)all_and, m, 64);
        }
        unsigned vary = all_or ^ all_and;
        unsigned th = all_and;
        th = __builtin_amdgcn_readfirstlane(th);
        vary = __builtin_amdgcn_readfirstlane(vary);
        while (vary) {                                 // largest th with #{rank >= th} >= k, most significant open bit first
            const int b = 31 - __builtin_clz(vary);
            vary &= ~(1u << b);
            const unsigned cnd = th | (1u << b);
            int cge = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) cge += __popcll(__ballot((unsigned)(v[j] >> 32) >= cnd));
            if (cge >= k) th = cnd;
        }
        int above = 0, equal = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            above += __popcll(__ballot((unsigned)(v[j] >> 32) > th));
            equal += __popcll(__ballot(v[j] != 0ull && (unsigned)(v[j] >> 32) == th));
        }
        const int need = k - above;                    // >= 1 of the keys ranked exactly th
        unsigned tlo = 0u;
        if (equal > need) {
            for (int b = 31; b >= 0; --b) {
                const unsigned cnd = tlo | (1u << b);
                int cge = 0;
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    cge += __popcll(__ballot(v[j] != 0ull && (unsigned)(v[j] >> 32) == th && (unsigned)v[j] >= cnd));
                if (cge >= need) tlo = cnd;
            }
        }
        tau = ((unsigned long long)th << 32) | tlo;
    }
    int pos = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const bool wj = v[j] != 0ull && v[j] >= tau;
        const unsigned long long m = __ballot(wj);
        if (wj) dst[pos + __popcll(m & below)] = v[j];
        pos += __popcll(m);
    }
    return pos;
}

__device__ __forceinline__ unsigned long long sq_wave_or_u64(unsigned long long x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x |= mf_shfl_xor_u64(x, m);
    return x;
}

// ------------------------------------------------------------------ the scan ----
// Launch 1: every wave scores its 64-row block against the queries and leaves, per query, the 64 scores (4 bytes per
// row and query: under 1 % of the catalog's bytes) and its best key.  No selection here: a selection is a serial
// 32-step search, and almost none of the ~10^3 blocks can hold one of the k best rows.
struct SmallParams {
    const float* blocked;      // [nblocks][D/4][64][4]
    const float* q;            // [Q][D]
    int Q, k;
    int64_t N, nblocks;
    const int64_t *excl_off, *excl_idx;     // nullable
    int64_t idx_base;
    float* scores;             // [Q][nblocks * 64]: the score, or all-ones bits where the row holds no key (excluded, past N)
    unsigned long long* wmax;  // [Q][nblocks]: best key of every block (0: none)
};
static constexpr unsigned SQ_NOKEY = 0xFFFFFFFFu;

// QG queries at a time (compile-time: the accumulators are registers).  The rows of a block stay in registers
// across the query groups when they fit (d <= 128), else each group re-reads them (L2 / Infinity Cache).  The
// queries are wave-uniform: they are read by SCALAR loads (constant address space -> s_load_dwordx8 through the
// scalar cache) and enter the fma as SGPR operands -- no LDS, no vector registers (read by broadcast from LDS the
// compiler hoisted every fragment: 256 VGPRs, kilobytes of scratch per lane).
typedef const __attribute__((address_space(4))) float* mf_const_f32;

template <int D, int QG>
__global__ __launch_bounds__(64 * SQ_NW) void topk_small_scan_kernel(SmallParams p) {
    constexpr int CPR = D / 4;
    constexpr int HALF = CPR > 32 ? 32 : CPR;                  // chunks held in registers at once
    const int lane = mf_lane(), wave = mf_wave_id();
    const int Qp = (p.Q + QG - 1) / QG * QG;
    mf_const_f32 qc = (mf_const_f32)p.q;

    for (int64_t blk = (int64_t)blockIdx.x * SQ_NW + wave; blk < p.nblocks; blk += (int64_t)gridDim.x * SQ_NW) {
        const int64_t row0 = blk * 64;
        const int64_t row = row0 + lane;
        const f32x4* src = reinterpret_cast<const f32x4*>(p.blocked) + blk * (int64_t)CPR * 64 + lane;
        f32x4 rc[HALF];
        if constexpr (CPR <= 32) {
#pragma unroll
            for (int j = 0; j < CPR; ++j) rc[j] = src[(int64_t)j * 64];
        }
        for (int q0 = 0; q0 < Qp; q0 += QG) {
            float acc[QG];
#pragma unroll
            for (int qi = 0; qi < QG; ++qi) acc[qi] = 0.f;
#pragma unroll
            for (int h0 = 0; h0 < CPR; h0 += HALF) {
                if constexpr (CPR > 32) {
#pragma unroll
                    for (int j = 0; j < HALF; ++j) rc[j] = src[(int64_t)(h0 + j) * 64];
                }
#pragma unroll
                for (int g = 0; g < HALF / 2; ++g) {
                    const f32x4 a = rc[2 * g], b = rc[2 * g + 1];
#pragma unroll
                    for (int qi = 0; qi < QG; ++qi) {
                        const int qq = q0 + qi < p.Q ? q0 + qi : p.Q - 1;              // padding queries repeat the last one
                        mf_const_f32 qv = qc + (size_t)qq * D + 4 * (h0 + 2 * g);       // wave-uniform address
#pragma unroll
                        for (int t = 0; t < 4; ++t) {              // k order of mf_dot_chain
                            acc[qi] = __builtin_fmaf(a[t], qv[t], acc[qi]);
                            acc[qi] = __builtin_fmaf(b[t], qv[4 + t], acc[qi]);
                        }
                    }
                }
            }
#pragma unroll
            for (int qi = 0; qi < QG; ++qi) {
                const int q = q0 + qi;
                if (q >= p.Q) break;
                unsigned long long excl = 0ull;
                if (p.excl_off) {             // the query's exclusion list against this block's 64 rows
                    for (int64_t e = p.excl_off[q] + lane; e < p.excl_off[q + 1]; e += 64) {
                        const int64_t y = p.excl_idx[e] - p.idx_base - row0;
                        if (y >= 0 && y < 64) excl |= 1ull << y;
                    }
                    excl = sq_wave_or_u64(excl);
                }
                const bool ok = row < p.N && !((excl >> lane) & 1ull);
                const unsigned long long key = ok ? mf_key_retrieval(acc[qi], (unsigned)row) : 0ull;
                p.scores[(size_t)q * p.nblocks * 64 + row] = ok ? acc[qi] : __builtin_bit_cast(float, SQ_NOKEY);
                const unsigned long long best = mf_wave_max_u64(key);
                if (lane == 0) p.wmax[(size_t)q * p.nblocks + blk] = best;
            }
        }
    }
}

// Launch 1 for 2 .. 32 queries: the same hand-over (every row's score, every 64-row block's best key, per query) from the
// MATRIX core.  32 queries are exactly one MFMA tile side: v_mfma_f32_32x32x2_f32 over a 32-row tile of the blocked copy
// (lane (r, h) reads chunk 2g + h of row r: two coalesced 512-byte runs per load instruction) yields the canonical chain of
// every (row, query) pair -- 0.5 GFLOP at Q = 32, 3.3 us at the fp32 MFMA peak, under the 5.5 us the 32 MB take from HBM --
// so the scan costs the same for 2 queries as for 32, where the FMA-chain scan above pays per query (Q = 32: 99 us).
// Scores leave through a 32 x 32 LDS transpose (rows of 128 contiguous bytes per query instead of 4-byte scatters).
template <int D>
__global__ __launch_bounds__(64 * SQ_NW, (D <= 128 ? 2 : 1)) void topk_small_mfma_scan_kernel(SmallParams p) {      // (d = 256 spills at two)
    constexpr int CPR = D / 4;
    __shared__ float tr[SQ_NW][32][33];
    __shared__ unsigned long long exm[SQ_NW][32];
    // the exclusion lists of all queries, once per workgroup, as (row, query) per entry: a block then walks the ENTRIES (a few
    // thousand, from LDS) instead of every query's list from memory (32 dependent list walks per block: Q = 32 25 -> 59 us).  All of a thread's list loads are in flight at once (EX_CAP / 256 = 32: one
    // memory round trip; a loop over the queries would be one per query).  Measured and dropped: keeping only the entries of
    // the workgroup's own blocks (46 us: the filtering costs more than the walks it saves).
    constexpr int EX_CAP = 8192;
    __shared__ int ex_row[EX_CAP];
    __shared__ unsigned char ex_q[EX_CAP];
    __shared__ long long ex_off[SQ_MAXQ + 1];
    const int lane = mf_lane(), wave = mf_wave_id();
    const int c = lane & 31, h = lane >> 5;
    int64_t ex_total = 0;
    bool ex_lds = false;
    if (p.excl_off) {
        if ((int)threadIdx.x <= p.Q) ex_off[threadIdx.x] = p.excl_off[threadIdx.x];
        __syncthreads();
        const int64_t e0 = ex_off[0];
        ex_total = ex_off[p.Q] - e0;
        ex_lds = ex_total <= EX_CAP;                            // (else: the per-query walks below)
        if (ex_lds) {
            constexpr int EPT = EX_CAP / (64 * SQ_NW);
            int64_t idv[EPT];
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int64_t i = (int64_t)threadIdx.x + u * 64 * SQ_NW;
                idv[u] = i < ex_total ? p.excl_idx[e0 + i] : -1;
            }
            int qrun = 0;                                        // (a thread's entries ascend: its query pointer only moves forward)
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int64_t i = (int64_t)threadIdx.x + u * 64 * SQ_NW;
                if (i < ex_total) {
                    while (qrun + 1 < p.Q && ex_off[qrun + 1] <= e0 + i) ++qrun;      // the q with off[q] <= e0 + i < off[q + 1]
                    const int64_t y = idv[u] - p.idx_base;
                    ex_row[i] = (y >= 0 && y < p.N) ? (int)y : -1;
                    ex_q[i] = (unsigned char)qrun;
                }
            }
            __syncthreads();
        }
    }
    RowFrag<D> xq;                                               // B operand: query c (zeros beyond Q)
    mf_load_frag<D>(xq, p.q, c, c < p.Q);
    // a wave takes ONE 32-row tile: two waves per 64-row block, two blocks per workgroup, twice the waves of a block-per-wave
    // cut -- at N = 62,423 that is two waves per SIMD, one's loads under the other's MFMAs (scan 13.5 -> see the probe log)
    __shared__ unsigned long long bests[SQ_NW][32];
    const int tile = wave & 1;
    for (int64_t b0 = (int64_t)blockIdx.x * (SQ_NW / 2); b0 < p.nblocks; b0 += (int64_t)gridDim.x * (SQ_NW / 2)) {
        const int64_t blk = b0 + (wave >> 1);
        const bool active = blk < p.nblocks;                     // (wave-uniform; the workgroup's barriers are outside)
        const int64_t row0 = blk * 64;
        unsigned long long best = 0ull;
        const int ex_slot = ex_lds ? (wave >> 1) : wave;         // staged entries: one mask row per block of the workgroup
        if (p.excl_off && ex_lds) {
            // ONE walk over the staged entries by the whole workgroup, for both of its blocks (128 consecutive rows), eight LDS
            // reads in flight per thread -- not a walk per wave
            if (threadIdx.x < 64) exm[threadIdx.x >> 5][threadIdx.x & 31] = 0ull;
            __syncthreads();
            for (int i0 = threadIdx.x; i0 < (int)ex_total; i0 += 8 * 64 * SQ_NW) {
                int r8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) r8[u] = i0 + 64 * SQ_NW * u < (int)ex_total ? ex_row[i0 + 64 * SQ_NW * u] : -1;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t y = (int64_t)r8[u] - b0 * 64;
                    if (r8[u] >= 0 && y >= 0 && y < 128) atomicOr(&exm[y >> 6][ex_q[i0 + 64 * SQ_NW * u]], 1ull << (y & 63));
                }
            }
            __syncthreads();
        }
        if (active) {
        if (p.excl_off && ex_lds) {
        } else if (p.excl_off) {              // every query's exclusion list against this block's 64 rows (as in the scan above)
            for (int q = 0; q < p.Q; ++q) {
                unsigned long long excl = 0ull;
                for (int64_t e = p.excl_off[q] + lane; e < p.excl_off[q + 1]; e += 64) {
                    const int64_t y = p.excl_idx[e] - p.idx_base - row0;
                    if (y >= 0 && y < 64) excl |= 1ull << y;
                }
                excl = sq_wave_or_u64(excl);
                if (lane == 0) exm[wave][q] = excl;
            }
            mf_wave_sync();
        }
        const unsigned long long myex = (p.excl_off && c < p.Q) ? exm[ex_slot][c] : 0ull;
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(p.blocked) + blk * (int64_t)CPR * 64 + 32 * tile + c;
            f32x4 a[D / 8];
#pragma unroll
            for (int g = 0; g < D / 8; ++g) a[g] = src[(int64_t)(2 * g + h) * 64];
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int g = 0; g < D / 8; ++g)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][t], xq.v[g][t], acc, 0, 0, 0);
            // element e: row 32 tile + mf_acc_row(e, h) of the block, query c
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rb = 32 * tile + mf_acc_row(e, h);
                const int64_t row = row0 + rb;
                const bool ok = row < p.N && !((myex >> rb) & 1ull);
                const unsigned long long key = ok ? mf_key_retrieval(acc[e], (unsigned)row) : 0ull;
                best = key > best ? key : best;
                tr[wave][mf_acc_row(e, h)][c] = ok ? acc[e] : __builtin_bit_cast(float, SQ_NOKEY);
            }
            mf_wave_sync();
            // out: lane (r = c, q half = h) writes row r of queries h, h + 2, ...: 32 lanes = 128 contiguous bytes per query
            for (int q = h; q < p.Q; q += 2) p.scores[(size_t)q * p.nblocks * 64 + row0 + 32 * tile + c] = tr[wave][c][q];
            mf_wave_sync();
        }
        const unsigned long long other = mf_xor_lane_u64<32>(best);
        best = other > best ? other : best;
        }
        if (h == 0) bests[wave][c] = best;                       // the block's best key per query: the larger of its two tiles'
        __syncthreads();
        if (active && tile == 0 && h == 0 && c < p.Q) {
            const unsigned long long o = bests[wave + 1][c];
            p.wmax[(size_t)c * p.nblocks + blk] = o > best ? o : best;
        }
        __syncthreads();
    }
}

// Launch 2, one 256-thread workgroup per query.  (a) wave 0 reduces the block maxima to two per lane and takes the k-th
// largest of those 128 keys as the bound tau: a lower bound of the k-th best key (they are keys of distinct rows), at
// most a few ranks below the exact k-th largest maximum.  (b) only blocks whose maximum reaches tau can hold a row of
// the top k: their scores are turned back into keys and filtered against tau, 64 blocks a round.  (c) wave 0 selects
// the k best candidates (the winners so far riding along between rounds), orders them by rank counting and writes
// scores / rows.  One search where the first version ran four (three merge levels of exact block selection).
static constexpr int SQ_SEL_WAVES = 4;
static constexpr int SQ_ROUND_BLOCKS = 64;      // winning blocks per round: <= 4096 candidates in LDS

__global__ __launch_bounds__(64 * SQ_SEL_WAVES) void topk_small_select_kernel(const float* __restrict__ scores,
                                                                              const unsigned long long* __restrict__ wmax,
                                                                              int64_t nblocks, int k, int64_t idx_base,
                                                                              float* __restrict__ out_scores,
                                                                              int64_t* __restrict__ out_idx) {
    __shared__ unsigned long long top[64];       // the winners so far
    __shared__ unsigned long long two[64];
    __shared__ unsigned long long cand[SQ_ROUND_BLOCKS * 64];
    __shared__ int wlist[SQ_ROUND_BLOCKS];
    __shared__ int cand_n, wl_n, top_m;
    __shared__ unsigned long long tau_s;
    const int lane = mf_lane(), wave = mf_wave_id();
    const int64_t q = blockIdx.x;
    const unsigned long long* src = wmax + q * nblocks;
    if (threadIdx.x == 0) { wl_n = 0; cand_n = 0; top_m = 0; }
    __syncthreads();
    if (wave == 0) {
        // (a) two best maxima per lane; the maxima of 1024 blocks (16 per lane) are in flight at once, and stay in
        // registers for the list of winning blocks when the catalog has no more than that
        unsigned long long m1 = 0ull, m2 = 0ull;
        unsigned long long v[16];
        for (int64_t base = 0; base < nblocks; base += 64 * 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int64_t idx = base + lane + 64 * j;
                v[j] = idx < nblocks ? src[idx] : 0ull;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned long long lo = v[j] < m1 ? v[j] : m1;
                m2 = lo > m2 ? lo : m2;
                m1 = v[j] > m1 ? v[j] : m1;
            }
        }
        const unsigned long long v2[2] = {m1, m2};
        const int m = sq_wave_topk<2>(v2, k, two);
        mf_wave_sync();
        unsigned long long mn = lane < m ? two[lane] : ~0ull;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const unsigned long long o = mf_shfl_xor_u64(mn, s);
            mn = o < mn ? o : mn;
        }
        const unsigned long long tau0 = m >= k ? mn : 1ull;          // fewer than k maxima: every key passes
        if (lane == 0) tau_s = tau0;
        // the winning blocks (about k of them), by the same wave: no second trip to memory for a catalog of <= 1024 blocks
        int listed = 0;
        for (int64_t base = 0; base < nblocks; base += 64 * 16) {
            if (nblocks > 64 * 16) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int64_t idx = base + lane + 64 * j;
                    v[j] = idx < nblocks ? src[idx] : 0ull;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {                  // (one wave: the list's fill count is a scalar, no atomics)
                const bool hit = v[j] != 0ull && v[j] >= tau0;
                const unsigned long long bal = __ballot(hit);
                const int at = listed + __popcll(bal & ((1ull << lane) - 1ull));
                if (hit && at < SQ_ROUND_BLOCKS) wlist[at] = (int)(base + lane + 64 * j);
                listed += __popcll(bal);
            }
        }
        if (lane == 0) wl_n = listed;
    }
    __syncthreads();
    const unsigned long long tau = tau_s;
    // (b) + (c) for the W blocks in wlist: candidates into LDS, wave 0 merges them into the winners so far
    auto round = [&](int W) {
        // a wave's blocks eight at a time: their scores are asked for together (one memory round trip, not one per block)
        for (int t0 = wave; t0 < W; t0 += SQ_SEL_WAVES * 8) {
            float sc[8];
            int64_t rows[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = t0 + j * SQ_SEL_WAVES;
                rows[j] = (int64_t)wlist[t < W ? t : t0] * 64 + lane;
                sc[j] = scores[(size_t)q * nblocks * 64 + rows[j]];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (t0 + j * SQ_SEL_WAVES >= W) break;                  // (wave-uniform)
                const unsigned long long key = __builtin_bit_cast(unsigned, sc[j]) == SQ_NOKEY ? 0ull : mf_key_retrieval(sc[j], (unsigned)rows[j]);
                const bool keep = key != 0ull && key >= tau;
                const unsigned long long mk = __ballot(keep);
                int base = 0;
                if (lane == 0 && mk) base = atomicAdd(&cand_n, __popcll(mk));
                base = __builtin_amdgcn_readfirstlane(base);
                if (keep) cand[base + __popcll(mk & ((1ull << lane) - 1ull))] = key;
            }
        }
        __syncthreads();
        if (wave == 0) {
            const int C = cand_n;
            int car = top_m;
            if (C <= 64 && car == 0) {              // the usual case: one key per lane, one search
                const unsigned long long v1[1] = {lane < C ? cand[lane] : 0ull};
                mf_wave_sync();
                car = sq_wave_topk<1>(v1, k, top);
                mf_wave_sync();
            } else
            for (int base = 0; base < C; base += 64 * 4) {
                unsigned long long v[5];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = base + lane + 64 * j < C ? cand[base + lane + 64 * j] : 0ull;
                v[4] = lane < car ? top[lane] : 0ull;
                mf_wave_sync();
                car = sq_wave_topk<5>(v, k, top);
                mf_wave_sync();
            }
            if (lane == 0) { top_m = car; cand_n = 0; }
        }
        __syncthreads();
    };
    auto hit_of = [&](int64_t blk) { return blk < nblocks && src[blk] >= tau && src[blk] != 0ull; };
    const int Wt = wl_n;
    if (Wt <= SQ_ROUND_BLOCKS) {
        round(Wt);
    } else {
        // (maxima piled up in a few lanes' shares, or thousands of tied rows) SQ_ROUND_BLOCKS blocks a round, in block
        // order: a block's position among the hits by ballot prefix sums
        __shared__ int wtot[SQ_SEL_WAVES];
        for (int lo = 0; lo < Wt; lo += SQ_ROUND_BLOCKS) {
            int running = 0;
            for (int64_t base = 0; base < nblocks; base += 64 * SQ_SEL_WAVES) {
                const int64_t blk = base + threadIdx.x;
                const bool hit = hit_of(blk);
                const unsigned long long bal = __ballot(hit);
                if (lane == 0) wtot[wave] = __popcll(bal);
                __syncthreads();
                int off = running + __popcll(bal & ((1ull << lane) - 1ull)), tot = 0;
                for (int w = 0; w < SQ_SEL_WAVES; ++w) {
                    if (w < wave) off += wtot[w];
                    tot += wtot[w];
                }
                if (hit && off >= lo && off < lo + SQ_ROUND_BLOCKS) wlist[off - lo] = (int)blk;
                running += tot;
                __syncthreads();
            }
            round(min(SQ_ROUND_BLOCKS, Wt - lo));
        }
    }
    if (wave != 0) return;
    const int m = top_m;
    const unsigned long long mine = lane < m ? top[lane] : 0ull;
    int r = 0;
    for (int t = 0; t < m; ++t) r += top[t] > mine ? 1 : 0;
    if (lane < k) {
        if (lane < m) {
            out_scores[q * k + r] = mf_key_retrieval_score(mine);
            out_idx[q * k + r] = idx_base + (int64_t)mf_key_retrieval_col(mine);
        } else {                                  // fewer than k rows were eligible: -inf / -1 tail
            out_scores[q * k + lane] = -INFINITY;
            out_idx[q * k + lane] = -1;
        }
    }
}

static int small_grid(int64_t nblocks) {
    const int64_t nwg = (nblocks + SQ_NW - 1) / SQ_NW;
    return (int)(nwg < 2048 ? nwg : 2048);       // <= 8 workgroups per CU; beyond that the waves loop
}

struct SmallWs {
    float* scores;
    unsigned long long* wmax;
    size_t total;
};
static SmallWs small_ws(void* base, int64_t Q, int64_t nblocks) {
    MfArena a(base);
    SmallWs w;
    w.scores = a.take<float>((size_t)Q * nblocks * 64);
    w.wmax = a.take<unsigned long long>((size_t)Q * nblocks);
    w.total = a.used();
    return w;
}

extern "C" size_t mf_topk_small_ws_bytes(int64_t Q, int64_t N, int d, int k) {
    if (Q <= 0 || Q > SQ_MAXQ || N <= 0 || k <= 0 || k > 64 || !mf_width_ok(d)) return 0;
    return small_ws(nullptr, Q, (N + 63) / 64).total;
}

template <int D, int QG>
static void launch_small(const SmallParams& sp, int grid, hipStream_t s) {
    topk_small_scan_kernel<D, QG><<<dim3((unsigned)grid), 64 * SQ_NW, 0, s>>>(sp);
}

extern "C" int mf_topk_small(const float* q, int64_t Q, const float* blocked, int64_t N, int d, int k,
                             const int64_t* excl_off, const int64_t* excl_idx, int64_t idx_base, void* ws,
                             size_t ws_bytes, float* out_scores, int64_t* out_idx, mf_stream_t stream) {
    if (!q || !blocked || !out_scores || !out_idx || !ws || Q <= 0 || N <= 0)
        return mf_set_error(MF_EINVAL, "mf_topk_small: bad argument");
    if (Q > SQ_MAXQ) return mf_set_error(MF_ENOTSUP, "mf_topk_small: Q = %lld > %d (use mf_topk)", (long long)Q, SQ_MAXQ);
    if (k <= 0 || k > 64) return mf_set_error(MF_ENOTSUP, "mf_topk_small: k = %d outside 1..64", k);
    if (!mf_width_ok(d)) return mf_set_error(MF_EINVAL, "mf_topk_small: embedding width %d not in {32,64,128,256}", d);
    if (N >= (1ll << 31) || idx_base < 0 || idx_base + N > (1ll << 32))
        return mf_set_error(MF_ENOTSUP, "mf_topk_small: item indices must fit 32 bits");
    if ((excl_off == nullptr) != (excl_idx == nullptr)) return mf_set_error(MF_EINVAL, "mf_topk_small: excl_off/excl_idx mismatch");
    if (ws_bytes < mf_topk_small_ws_bytes(Q, N, d, k)) return mf_set_error(MF_ENOSPC, "mf_topk_small: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t nblocks = (N + 63) / 64;
    const int grid = small_grid(nblocks);
    SmallWs w = small_ws(ws, Q, nblocks);
    SmallParams sp{blocked, q, (int)Q, k, N, nblocks, excl_off, excl_idx, idx_base, w.scores, w.wmax};
    MF_DISPATCH_D(d, {
        MF_TIMED("topk_small", s, {
            if (Q == 1) launch_small<D, 1>(sp, grid, s);
            else {                                               // 2 .. 32 queries: one MFMA tile side; a wave per 32-row tile
                const int64_t nwg2 = (nblocks + SQ_NW / 2 - 1) / (SQ_NW / 2);
                topk_small_mfma_scan_kernel<D><<<dim3((unsigned)(nwg2 < 4096 ? nwg2 : 4096)), 64 * SQ_NW, 0, s>>>(sp);
            }
            topk_small_select_kernel<<<dim3((unsigned)Q), 64 * SQ_SEL_WAVES, 0, s>>>(w.scores, w.wmax, nblocks, k, idx_base, out_scores, out_idx);
        });
    });
    return mf_check_launch("mf_topk_small");
}
