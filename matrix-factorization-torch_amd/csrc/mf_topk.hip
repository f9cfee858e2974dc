// mf_topk.hip -- exact brute-force full-catalog top-k (retrieval), gfx950.
//
// Replaces ItemProcessor.search (xfmr_rec/data/lightning.py:237-259, LanceDB cosine
// ANN + prefilter) by an exact scan: score tiles from the fp32 MFMA engine, streaming
// per-query selection (mf_select.h), exclusion lists scattered once into a
// bitmask (one coalesced word per query and tile), then an exact ordered merge per query.
// The same merge kernel joins the all-gathered partial top-k of a row-sharded
// catalog (mf_topk_merge).
#include <cmath>

#include "mf_common.h"
#include <vector>

#include "mf_select.h"

// Exclusion prefilter as a bitmask: exclW[tile][query] holds the 32 "excluded" bits of
// that query for that item tile (the lists are sparse -- ~10^2 of 10^4..10^8 rows --
// so the words are scattered once per call and read with one coalesced load per tile).
struct RetrievalPolicy {
    struct Params {
        const uint32_t* exclW;     // [NT][Qp], all zero when nothing is excluded
        int64_t Qp, nY;
    };
    struct Row {};
    struct Tile {
        uint32_t ew;
    };
    static constexpr int AUX_DMA = 1;
    // the rank is mf_orderable(score): monotone in the raw score, so the bound is one float (NaN and signed
    // zeros pass: conservative)
    struct Thr {
        float f;
    };
    static __device__ __forceinline__ Thr thr_all() { return Thr{__builtin_bit_cast(float, 0xFFFFFFFFu)}; }   // NaN: nothing is below it
    static __device__ __forceinline__ Thr thr_none() { return Thr{__builtin_inff()}; }
    static __device__ __forceinline__ Thr make_thr(unsigned bound) { return Thr{mf_unorderable(bound)}; }
    static __device__ __forceinline__ bool maybe(const Params&, const Row&, const Tile&, float score, int, int, const Thr& t) {
        return !(score < t.f);
    }
    static __device__ __forceinline__ void stage_aux(const Params& p, char* aux, int wave, int t, int64_t x0, int) {
        mf_stage_small(aux + wave * 128, p.exclW + (int64_t)t * p.Qp + x0, 128);
    }
    static __device__ __forceinline__ Row row_init(const Params&, int64_t, bool) { return Row{}; }
    static __device__ __forceinline__ Tile tile_init(const Params&, const Row&, const char* aux, int wave, int c, int, int) {
        return Tile{reinterpret_cast<const uint32_t*>(aux + wave * 128)[c]};
    }
    static __device__ __forceinline__ bool key(const Params& p, const Row&, const Tile& t, float score, int e, int h,
                                               unsigned y, unsigned& hi, unsigned& lo) {
        hi = mf_key_retrieval_hi(score);
        lo = mf_key_retrieval_lo(y);
        return !((t.ew >> mf_acc_row(e, h)) & 1u) && y < (unsigned)p.nY;
    }
};

__global__ __launch_bounds__(256) void excl_scatter_kernel(const int64_t* __restrict__ excl_off,
                                                           const int64_t* __restrict__ excl_idx, int64_t idx_base,
                                                           int64_t N, int64_t Qp, uint32_t* __restrict__ exclW) {
    const int64_t r = blockIdx.x;
    for (int64_t e = excl_off[r] + threadIdx.x; e < excl_off[r + 1]; e += 256) {
        const int64_t y = excl_idx[e] - idx_base;
        if (y >= 0 && y < N) atomicOr(&exclW[(y >> 5) * Qp + r], 1u << (y & 31));
    }
}

struct TopkWs {
    SelectPlan plan;
    int NT;
    unsigned long long* cand;
    unsigned long long* priv;
    unsigned long long* seeds;
    int32_t* cand_cnt;
    uint32_t* exclW;
    unsigned* gtau;      // gtau and cand_cnt sit right behind exclW: one memset clears all three
    size_t total;
};

static TopkWs topk_ws(void* base, int64_t Q, int64_t N, int d, int k) {
    TopkWs w{};
    w.plan = mf_select_plan(Q, N, d, k);
    w.NT = (int)((N + 31) / 32);
    MfArena a(base);
    w.cand = a.take<unsigned long long>((size_t)w.plan.Xp * w.plan.rowcap);
    w.priv = a.take<unsigned long long>((size_t)w.plan.nsets * w.plan.Xp * w.plan.CAP);
    w.seeds = a.take<unsigned long long>((size_t)w.plan.Xp * (w.plan.seeds_per_row > 0 ? w.plan.seeds_per_row : 1));
    w.exclW = a.take<uint32_t>((size_t)w.NT * w.plan.Xp);
    w.gtau = a.take<unsigned>((size_t)w.plan.Xp);
    w.cand_cnt = a.take<int32_t>((size_t)w.plan.Xp);
    w.total = a.used();
    return w;
}

extern "C" size_t mf_topk_ws_bytes(int64_t Q, int64_t N, int d, int k) {
    if (Q <= 0 || N <= 0 || k <= 0 || !mf_width_ok(d)) return 0;
    return topk_ws(nullptr, Q, N, d, k).total;
}

// host-only geometry query (tests, sizing): how mf_topk would cut a catalog for Q queries
extern "C" int mf_topk_chunks(int64_t Q, int64_t N, int d, int k, int64_t* rows_per_chunk) {
    if (Q <= 0 || N <= 0 || k <= 0 || !mf_width_ok(d)) return 0;
    const SelectPlan pl = mf_select_plan(Q, N, d, k);
    if (!pl.ok) return 0;
    if (rows_per_chunk) *rows_per_chunk = (int64_t)pl.tpc * 32;
    return pl.nchunk;
}

// one wave per query: the row's candidate list -> ordered top-k
__global__ __launch_bounds__(64) void topk_merge_cand_kernel(const unsigned long long* __restrict__ cand,
                                                             const int32_t* __restrict__ cand_cnt, int rowcap, int k,
                                                             int64_t idx_base, float* __restrict__ out_scores,
                                                             int64_t* __restrict__ out_idx) {
    __shared__ unsigned long long win[64], sorted[64];
    const int64_t r = blockIdx.x;
    const int m = mf_row_topk<8>(cand + r * (int64_t)rowcap, cand_cnt[r], k, win, sorted);
    const int t = mf_lane();
    if (t < k) {
        if (t < m) {
            out_scores[r * k + t] = mf_key_retrieval_score(sorted[t]);
            out_idx[r * k + t] = idx_base + (int64_t)mf_key_retrieval_col(sorted[t]);
        } else {
            out_scores[r * k + t] = -INFINITY;
            out_idx[r * k + t] = -1;
        }
    }
}

// merge of G already-ordered partial results with GLOBAL indices (< 2^32)
__global__ __launch_bounds__(64) void topk_merge_parts_kernel(const float* __restrict__ ps, const int64_t* __restrict__ pi,
                                                              int G, int64_t Q, int k, float* __restrict__ out_scores,
                                                              int64_t* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    const int64_t r = blockIdx.x;
    const int lane = mf_lane();
    const int total = G * k;
    for (int t = lane; t < total; t += 64) {
        const int g = t / k, e = t % k;
        const int64_t id = pi[((int64_t)g * Q + r) * k + e];
        s_keys[t] = id >= 0 ? mf_key_retrieval(ps[((int64_t)g * Q + r) * k + e], (unsigned)id) : 0ull;
    }
    __syncthreads();
    mf_wave_select(s_keys, total, k, [&](int t, unsigned long long key) {
        if (key != 0ull) {
            out_scores[r * k + t] = mf_key_retrieval_score(key);
            out_idx[r * k + t] = (int64_t)mf_key_retrieval_col(key);
        } else {
            out_scores[r * k + t] = -INFINITY;
            out_idx[r * k + t] = -1;
        }
    });
}

// Sharded retrieval, one exchange instead of two: a rank's partial result (scores, LOCAL rows) becomes one int64 per entry
// -- score bits << 32 | global row (local * stride + offset; < 2^32), all ones for "none" -- in a single launch (the row
// mapping alone took four elementwise launches), travels as ONE all-to-all, and is merged straight from that form.
__global__ __launch_bounds__(256) void topk_pack_kernel(const float* __restrict__ s, const int64_t* __restrict__ rows, int64_t n,
                                                        int64_t stride, int64_t offset, int64_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int64_t r = rows[t];
    out[t] = r >= 0 ? (int64_t)(((unsigned long long)__builtin_bit_cast(unsigned, s[t]) << 32) | (unsigned long long)(unsigned)(r * stride + offset))
                    : -1ll;
}

__global__ __launch_bounds__(64) void topk_merge_packed_kernel(const int64_t* __restrict__ packed, int G, int64_t Q, int k,
                                                               float* __restrict__ out_scores, int64_t* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
    const int64_t r = blockIdx.x;
    const int lane = mf_lane();
    const int total = G * k;
    for (int t = lane; t < total; t += 64) {
        const int g = t / k, e = t % k;
        const int64_t v = packed[((int64_t)g * Q + r) * k + e];
        s_keys[t] = v != -1ll ? mf_key_retrieval(__builtin_bit_cast(float, (unsigned)((unsigned long long)v >> 32)), (unsigned)v) : 0ull;
    }
    __syncthreads();
    mf_wave_select(s_keys, total, k, [&](int t, unsigned long long key) {
        if (key != 0ull) {
            out_scores[r * k + t] = mf_key_retrieval_score(key);
            out_idx[r * k + t] = (int64_t)mf_key_retrieval_col(key);
        } else {
            out_scores[r * k + t] = -INFINITY;
            out_idx[r * k + t] = -1;
        }
    });
}

extern "C" int mf_topk_pack(const float* scores, const int64_t* rows, int64_t n, int64_t stride, int64_t offset, int64_t* out_packed,
                            mf_stream_t stream) {
    if (!scores || !rows || !out_packed || n < 0 || stride <= 0 || offset < 0) return mf_set_error(MF_EINVAL, "mf_topk_pack: bad argument");
    if (n == 0) return MF_OK;
    topk_pack_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, static_cast<hipStream_t>(stream)>>>(scores, rows, n, stride, offset, out_packed);
    return mf_check_launch("mf_topk_pack");
}

extern "C" int mf_topk_merge_packed(const int64_t* packed, int G, int64_t Q, int k, float* out_scores, int64_t* out_idx, mf_stream_t stream) {
    if (!packed || !out_scores || !out_idx || G <= 0 || Q <= 0 || k <= 0) return mf_set_error(MF_EINVAL, "mf_topk_merge_packed: bad argument");
    if ((size_t)G * k * 8 > 64 * 1024) return mf_set_error(MF_ENOTSUP, "mf_topk_merge_packed: G * k too large");
    topk_merge_packed_kernel<<<dim3((unsigned)Q), 64, (size_t)G * k * 8, static_cast<hipStream_t>(stream)>>>(packed, G, Q, k, out_scores, out_idx);
    return mf_check_launch("mf_topk_merge_packed");
}

#ifdef MF_PROBE
static TopkWs g_probe_ws;
extern "C" long long mf_probe_topk_cand() {       // tools/topk_probe.py: candidate keys kept by the last mf_topk
    (void)hipDeviceSynchronize();
    const size_t n = (size_t)g_probe_ws.plan.Xp;
    std::vector<int32_t> h(n);
    (void)hipMemcpy(h.data(), g_probe_ws.cand_cnt, n * 4, hipMemcpyDeviceToHost);
    long long t = 0;
    for (int32_t c : h) t += c;
    return t;
}
extern "C" void mf_probe_sel_counters(unsigned long long* out16, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(mf_sel_dbg), 16 * 8);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mf_sel_dbg), z, 16 * 8);
    }
}
#endif

extern "C" int mf_topk(const float* q, int64_t Q, const float* items, int64_t N, int d, int k,
                       const int64_t* excl_off, const int64_t* excl_idx, int64_t idx_base, void* ws,
                       size_t ws_bytes, float* out_scores, int64_t* out_idx, mf_stream_t stream) {
    if (!q || !items || !out_scores || !out_idx || !ws || Q <= 0 || N <= 0)
        return mf_set_error(MF_EINVAL, "mf_topk: bad argument");
    if (k <= 0 || k > 64) return mf_set_error(MF_ENOTSUP, "mf_topk: k = %d outside 1..64", k);
    if (!mf_width_ok(d)) return mf_set_error(MF_EINVAL, "mf_topk: embedding width %d not in {32,64,128,256}", d);
    if (N >= (1ll << 31) || idx_base < 0 || idx_base + N > (1ll << 32))
        return mf_set_error(MF_ENOTSUP, "mf_topk: item indices must fit 32 bits");
    if ((excl_off == nullptr) != (excl_idx == nullptr)) return mf_set_error(MF_EINVAL, "mf_topk: excl_off/excl_idx mismatch");
    if (ws_bytes < mf_topk_ws_bytes(Q, N, d, k)) return mf_set_error(MF_ENOSPC, "mf_topk: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    TopkWs w = topk_ws(ws, Q, N, d, k);
    if (!w.plan.ok) return mf_set_error(MF_ENOTSUP, "mf_topk: k = %d too large at d = %d", k, d);
#ifdef MF_PROBE
    g_probe_ws = w;
#endif
    (void)hipMemsetAsync(w.exclW, 0, (size_t)((char*)(w.cand_cnt + w.plan.Xp) - (char*)w.exclW), s);
    if (excl_off) excl_scatter_kernel<<<dim3((unsigned)Q), 256, 0, s>>>(excl_off, excl_idx, idx_base, N, w.plan.Xp, w.exclW);
    RetrievalPolicy::Params rp{w.exclW, w.plan.Xp, N};
    SelectCommon sc{q, Q, items, N, 0, w.NT, w.plan.tpc, w.plan.Xp, k, w.plan.xw, w.gtau, w.priv, w.cand, w.cand_cnt, w.plan.rowcap};
    MF_DISPATCH_D(d, { MF_TIMED("topk_select", s, (mf_select_run<D, RetrievalPolicy>(w.plan, rp, sc, w.seeds, Q, s))); });
    topk_merge_cand_kernel<<<dim3((unsigned)Q), 64, 0, s>>>(w.cand, w.cand_cnt, w.plan.rowcap, k, idx_base, out_scores, out_idx);
    return mf_check_launch("mf_topk");
}

extern "C" int mf_topk_merge(const float* part_scores, const int64_t* part_idx, int G, int64_t Q, int k,
                             float* out_scores, int64_t* out_idx, mf_stream_t stream) {
    if (!part_scores || !part_idx || !out_scores || !out_idx || G <= 0 || Q <= 0 || k <= 0)
        return mf_set_error(MF_EINVAL, "mf_topk_merge: bad argument");
    if ((size_t)G * k * 8 > 64 * 1024) return mf_set_error(MF_ENOTSUP, "mf_topk_merge: G * k too large");
    topk_merge_parts_kernel<<<dim3((unsigned)Q), 64, (size_t)G * k * 8, static_cast<hipStream_t>(stream)>>>(
        part_scores, part_idx, G, Q, k, out_scores, out_idx);
    return mf_check_launch("mf_topk_merge");
}

// ------------------------------------------------------------ retrieval metrics ----
// one wave per query; lane t holds the item retrieved at rank t
__global__ __launch_bounds__(64) void retrieval_metrics_kernel(const int64_t* __restrict__ topk_idx, int k,
                                                               const int64_t* __restrict__ tgt_off,
                                                               const int64_t* __restrict__ tgt_idx,
                                                               const float* __restrict__ tgt_rel, float* __restrict__ out) {
    const int64_t q = blockIdx.x;
    const int lane = mf_lane();
    const int64_t e0 = tgt_off[q], e1 = tgt_off[q + 1];
    const int64_t item = lane < k ? topk_idx[q * k + lane] : -1;
    float rel = 0.f;                                   // rating of the item at this rank (0: not a target)
    // a target the search missed ranks right below the retrieved items (list order): it only enters the
    // top k when fewer than k items were retrieved, exactly as in the reference's union of both sets
    int slot = __popcll(__ballot(item >= 0));
    for (int64_t e = e0; e < e1; ++e) {
        const bool mine = item >= 0 && tgt_idx[e] == item;
        if (mine) rel = tgt_rel[e];
        if (!__any(mine)) {
            if (lane == slot && lane < k) rel = tgt_rel[e];
            ++slot;
        }
    }
    // ideal DCG: the targets' ratings in descending order (rank by counting), best k of them
    float idcg = 0.f;
    int npos = 0;
    for (int64_t e = e0 + lane; e < e1; e += 64) {
        const float r = tgt_rel[e];
        npos += r > 0.f ? 1 : 0;
        int rank = 0;
        for (int64_t f = e0; f < e1; ++f) {
            const float o = tgt_rel[f];
            rank += (o > r || (o == r && f < e)) ? 1 : 0;
        }
        if (rank < k) idcg += r / log2f((float)rank + 2.f);
    }
    float dcg = rel / log2f((float)lane + 2.f);
    dcg = mf_wave_sum(dcg);
    idcg = mf_wave_sum(idcg);
    npos = mf_wave_sum_int(npos);
    const unsigned long long hit = __ballot(rel > 0.f);
    const int hits = __popcll(hit);
    float ap = 0.f;                                    // precision at every relevant rank
    if (rel > 0.f) ap = (float)__popcll(hit & ((2ull << lane) - 1ull)) / (float)(lane + 1);
    ap = mf_wave_sum(ap);
    if (lane == 0) {
        float* o = out + q * 6;
        const bool any = npos > 0;
        o[0] = (any && idcg > 0.f) ? dcg / idcg : 0.f;
        o[1] = any ? (float)hits / (float)npos : 0.f;
        o[2] = any ? (float)hits / (float)k : 0.f;
        o[3] = hits > 0 ? ap / (float)hits : 0.f;
        o[4] = hits > 0 ? 1.f : 0.f;
        o[5] = hits > 0 ? 1.f / (float)(__builtin_ctzll(hit) + 1) : 0.f;
    }
}

extern "C" int mf_retrieval_metrics(const int64_t* topk_idx, int64_t Q, int k, const int64_t* tgt_off,
                                    const int64_t* tgt_idx, const float* tgt_rel, float* out, mf_stream_t stream) {
    if (!topk_idx || !tgt_off || !tgt_idx || !tgt_rel || !out || Q <= 0) return mf_set_error(MF_EINVAL, "mf_retrieval_metrics: bad argument");
    if (k <= 0 || k > 64) return mf_set_error(MF_ENOTSUP, "mf_retrieval_metrics: k = %d outside 1..64", k);
    retrieval_metrics_kernel<<<dim3((unsigned)Q), 64, 0, static_cast<hipStream_t>(stream)>>>(topk_idx, k, tgt_off, tgt_idx, tgt_rel, out);
    return mf_check_launch("mf_retrieval_metrics");
}
