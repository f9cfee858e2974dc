// mf_select.h -- streaming per-row top-k over MFMA score tiles (gfx950).
//
// One wavefront owns 32 "X" rows (queries / users: one per lane pair) and streams
// 32-row "Y" tiles (catalog items / batch items) through the fp32 MFMA engine of
// mf_common.h.  Nothing of the 32 x N score slab is ever written to HBM: every
// element becomes a unique 64-bit key (mf_numerics.h, larger == better) and only
// keys that can still be among the row's best k survive:
//
//   1. each lane keeps, in registers, the T = ceil(k/2) best 32-bit ranks
//      (key >> 32) it has accepted (static insertion network).  tau_row = min over
//      the row's two lanes of their T-th best is a lower bound of the row's k-th
//      best, because >= 2T >= k accepted keys are >= it.  Keys ranked below tau_row
//      are dropped by one compare.
//   2. a surviving key is appended to the lane's PRIVATE LDS segment (one ds_write,
//      the fill count lives in a register: no atomics, no cross-lane traffic).  The
//      T-list takes only the best accepted rank of each tile (one branch-free insertion
//      per tile; per-element insertion would run whenever "some lane of 64" needs it).
//      A segment that could overflow on the next tile is filtered in place by its
//      owner against the current tau_row (no sort needed).
//   3. pathological inputs (e.g. all scores equal) defeat 1-2; then the wave selects
//      that row's k best keys exactly and installs a full 64-bit floor.
//
// The surviving candidates of every (row, column-chunk) go to HBM (<= 2 CAPL keys
// each, appended to the row's list); the exact ordered top-k is produced by a merge kernel (one wave per row).
#pragma once

#include "mf_common.h"

// T = per-lane register list length (>= ceil(k / 2)); CAPL = per-lane LDS segment capacity: as
// large as the 160 KiB of LDS allow next to the tile ring (a segment is re-filtered only when it
// could overflow on the next tile, so a roomy segment means almost never).
static inline int mf_select_T(int k) { return k <= 4 ? 2 : k <= 8 ? 4 : k <= 16 ? 8 : k <= 20 ? 10 : k <= 24 ? 12 : k <= 32 ? 16 : 32; }
static inline int mf_select_nslot(int d) { return d == 256 ? 2 : 3; }
static inline int mf_select_capl(int d) {
    const int ring = mf_select_nslot(d) * (32 * d * 4) + 4 * 1536;
    int capl = (160 * 1024 - ring - 1024) / (4 * 64 * 8) - 1;
    capl = capl > 64 ? 64 : capl;
    return capl & ~1;        // even capacity -> odd segment stride: lanes of a half hit distinct banks
}

#ifdef __HIPCC__

#include "mf_stream.h"

struct SelectCommon {
    const float* X;      // [nX, D] rows kept on the lanes
    int64_t nX;
    const float* Y;      // [nY, D] rows streamed
    int64_t nY;
    int YT;              // number of 32-row Y tiles
    int tiles_per_chunk;
    int64_t Xp;          // nX padded to 128
    int k;
    int xw;              // X tiles per workgroup (1, 2 or 4); the 4 / xw waves that share an X tile
                         // deal the chunk's Y tiles round-robin and emit separate candidate sets
    int capl;            // per-lane LDS segment capacity
    unsigned* gtau;      // [Xp], zeroed by the host: best known lower bound (rank) of every X row's k-th best key,
                         // shared by all chunks of the launch (max-published, so always a valid bound)
    unsigned long long* cand;   // [Xp][rowcap]: every X row's surviving keys of ALL chunks, contiguous
    int32_t* cand_cnt;          // [Xp], zeroed by the host: fill count of the row's list (atomic cursor)
    int rowcap;                 // nsets * 2 capl
};

#ifdef MF_PROBE
// tools/topk_probe.py: wave-summed cycle / event counters of the last launches
// 0 total cycles, 1 settle cycles, 2 warm-loop cycles, 3 filter+slow-path cycles, 4 accepted keys,
// 5 slices whose body ran (any lane passed the prefilter), 6 filter calls, 7 slow-path rows, 8 waves
static __device__ unsigned long long mf_sel_dbg[16];
#define MF_PROBE_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define MF_PROBE_ADD(i, x) dbg[i] += (x)
#else
#define MF_PROBE_T(v)
#define MF_PROBE_ADD(i, x)
#endif

template <int T>
__device__ __forceinline__ void mf_tlist_insert(unsigned (&tl)[T], unsigned rank) {
    tl[T - 1] = rank;
#pragma unroll
    for (int i = T - 1; i > 0; --i) {
        const unsigned hi = max(tl[i - 1], tl[i]);
        const unsigned lo = min(tl[i - 1], tl[i]);
        tl[i - 1] = hi;
        tl[i] = lo;
    }
}

__device__ __forceinline__ void mf_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int D>
struct SelectLds {
    using G = TileGeom<D>;
    static constexpr int AUXB = 1536;                    // [4 x 128 B per-wave words][nv 128][logq 128][pad][4 x 128 B gtau]
    static constexpr int GT0 = 1024;                     // per-wave copies of gtau[x0 .. x0 + 31]
    static constexpr int NSLOT = D == 256 ? 2 : 3;       // d = 256: 2-deep tile ring, 2 barriers per tile
    static constexpr int AUX0 = NSLOT * G::TILEB;        // 4 side-input slots after the tile slots
    static constexpr int RING = AUX0 + 4 * AUXB;
    static __host__ __device__ int seg(int capl) { return 64 * (capl + 1) * 8; }   // one wave's 64 lane-private segments
    static __host__ __device__ int bytes(int capl) { return RING + 4 * seg(capl) + 4 * 32 * 8; }
};

// Policy interface:
//   struct Params;  struct Row;  struct Tile;
//   static constexpr int AUX_DMA;                   DMA instructions per wave per stage for side inputs
//   static void stage_aux(P, aux, wave, t, x0)      issue them (side inputs of Y tile t for X rows x0..x0+31)
//   static Row  row_init(P, x, valid)
//   static Tile tile_init(P, row, aux, wave, c, h)  read the staged side inputs
//   static bool key(P, row, tile, score, e, h, y, hi&, lo&)   false = never a candidate; (hi, lo) = key halves
//   static constexpr bool PREFILTER                 true: hi is mf_orderable(score), so `score < bound` may skip key()
template <int D, int T, class Policy>
__global__ __launch_bounds__(256) void select_kernel(typename Policy::Params pp, SelectCommon sc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = TileGeom<D>;
    using L = SelectLds<D>;
    constexpr int NWAIT = G::PPW + Policy::AUX_DMA + 1;
    const int CAPL = sc.capl;

    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
    const int nsub = 4 / sc.xw;
    const int sub = wave / sc.xw;
    // grid = (Y chunk, X block): the workgroups of one XCD (ids 8 apart) share chunks -> the catalog
    // slice they stream stays in that XCD's L2
    const int64_t x0 = ((int64_t)blockIdx.y * sc.xw + (wave % sc.xw)) * 32;
    const int64_t x = x0 + c;
    const int chunk = blockIdx.x;
    const int t0 = chunk * sc.tiles_per_chunk;
    const int t1 = min(sc.YT, t0 + sc.tiles_per_chunk);

    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem + L::RING + wave * L::seg(CAPL));
    unsigned long long* floor64 = reinterpret_cast<unsigned long long*>(smem + L::RING + 4 * L::seg(CAPL)) + wave * 32;
#define MF_BUF(l, i) buf[(l) * (CAPL + 1) + (i)]

    RowFrag<D> xf;
    mf_load_frag<D>(xf, sc.X, x, x < sc.nX);
    typename Policy::Row row = Policy::row_init(pp, x, x < sc.nX);

#ifdef MF_PROBE
    unsigned long long dbg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MF_PROBE_T(pt_begin);
    unsigned tl[T];
#pragma unroll
    for (int i = 0; i < T; ++i) tl[i] = 0u;
    unsigned tau_row = 0u;
    unsigned long long fl = 0ull;
    int cnt = 0;
    if (lane < 32) floor64[lane] = 0ull;

    auto stage = [&](int t) {
        const int kk = t - t0;
        mf_stage_tile<D>(smem + (kk % L::NSLOT) * G::TILEB, sc.Y, (int64_t)t * 32, sc.nY);
        Policy::stage_aux(pp, smem + L::AUX0 + (kk & 3) * L::AUXB, wave, t, x0);
        mf_stage_small<17>(smem + L::AUX0 + (kk & 3) * L::AUXB + L::GT0 + wave * 128, sc.gtau + x0, 128);
    };
    auto mine = [&](int t) { return (t - t0) % nsub == sub; };   // else another wave of this X tile takes tile t

    // Software pipeline inside the wave (as in the loss forward): the MFMAs of tile t+1 are issued
    // in the same basic block as the per-element key work of tile t (16 slices).
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typename Policy::Tile tile;
    unsigned tmaxr = 0u;                         // best rank this lane accepted in the current tile
    const bool row_ok = x < sc.nX;               // padding rows (all-zero X) never collect candidates
    unsigned y0 = 0u;                            // first Y row of the current tile (rows < 2^32)
    // 32-bit-only fast path: rank (= high key word) against the row bound; the exact 64-bit floor of
    // the degenerate path is checked by halves as well
    float thr_f = __builtin_bit_cast(float, 0xFFFFFFFFu);   // NaN: everything passes until a bound exists
    unsigned pub = 0u;                                       // last bound this lane published
    auto slice = [&](int e) {
        // one compare per element on the raw score where the policy's rank is monotone in it
        // (conservative: NaN and signed zeros pass), everything exact happens behind the branch
#ifdef MF_ABL_NOSLICE
        return;
#endif
        if (Policy::PREFILTER && (acc[e] < thr_f)) return;
#ifdef MF_PROBE
        if (lane == (int)__builtin_ctzll(__ballot(1))) dbg[5] += 1;
#endif
        unsigned hi, lo;
        const bool ok = Policy::key(pp, row, tile, acc[e], e, h, y0 + (unsigned)mf_acc_row(e, h), hi, lo);
        const unsigned fhi = (unsigned)(fl >> 32), flo = (unsigned)fl;
        const bool above_floor = hi > fhi || (hi == fhi && lo >= flo);
        if (ok && row_ok && hi >= tau_row && above_floor) {
            mf_lds_store_b64(buf + lane * (CAPL + 1) + cnt, lo, hi);     // asm store: must not drain the DMA queue
            ++cnt;
            MF_PROBE_ADD(4, 1);
            tmaxr = max(tmaxr, hi);
        }
    };
    // In-place filter of the lane's own segment against the current bound, 8 independent LDS reads
    // per round (writes only go to positions already read).
    auto filter_segment = [&]() {
        int w = 0;
        for (int t0f = 0; __any(t0f < cnt); t0f += 8) {
            unsigned long long kk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) kk[j] = (t0f + j < cnt) ? MF_BUF(lane, t0f + j) : 0ull;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kk[j] != 0ull && (unsigned)(kk[j] >> 32) >= tau_row && kk[j] >= fl) MF_BUF(lane, w++) = kk[j];
        }
        cnt = w;
    };
    // after the slices of a tile: refresh the bound, keep the segments from overflowing.
    // Only the tile's BEST accepted rank enters the lane's T-list: its entries are then T distinct
    // accepted elements (one per tile), so tl[T-1] stays a valid -- and, with the row's best spread
    // over many tiles, nearly tight -- lower bound, at the cost of one branch-free insertion per tile.
    // While the lists fill (the first T tiles of the chunk) every accepted key is inserted instead.
    auto settle = [&](int cnt0, bool warm, const char* aux) {
        MF_PROBE_T(ps0);
        if (warm) {
            for (int i = cnt0; __any(i < cnt); ++i) {
                if (i < cnt) {
                    const unsigned r = (unsigned)(MF_BUF(lane, i) >> 32);
                    if (r > tl[T - 1]) mf_tlist_insert<T>(tl, r);
                }
            }
        } else if (tmaxr > tl[T - 1]) {
            mf_tlist_insert<T>(tl, tmaxr);
        }
        tmaxr = 0u;
        MF_PROBE_T(ps1);
        MF_PROBE_ADD(2, ps1 - ps0);
        {
            const unsigned own = tl[T - 1];
            const unsigned mine_ = min(own, mf_shfl_xor32u(own));
            // every chunk's bound is a valid bound of the row: take the best one published so far
            // (the copy staged with this tile; a stale value only prunes less) and publish ours
#ifdef MF_ABL_NOGTAU
            const unsigned g = 0u * reinterpret_cast<const unsigned*>(aux + L::GT0 + wave * 128)[c] + 0xFFFFFFFFu * 0u;
            pub = 0xFFFFFFFFu;
#else
            const unsigned g = reinterpret_cast<const unsigned*>(aux + L::GT0 + wave * 128)[c];
#endif
            if (h == 0 && mine_ > g && mine_ > pub) {
                mf_global_umax(sc.gtau + x, mine_);
                pub = mine_;
            }
            tau_row = max(mine_, g);
        }
        MF_PROBE_T(ps2);
        if (__any(cnt > CAPL - 16)) {
            MF_PROBE_ADD(6, lane == 0 ? 1 : 0);
            filter_segment();   // drop, in place, what has fallen below the row's current bound
            // rare: a row still too full -> exact selection of its k best keys (wave-local)
            const unsigned long long ovb = __ballot(cnt > CAPL - 16);
            unsigned rows = (unsigned)(ovb | (ovb >> 32));
            while (rows) {
                const int r = __builtin_ctz(rows);
                rows &= rows - 1;
                MF_PROBE_ADD(7, lane == 0 ? 1 : 0);
                const int n0 = __shfl(cnt, r, 64), n1 = __shfl(cnt, r + 32, 64);
                const int m = n0 + n1;                       // <= 2 CAPL <= 160
                mf_wave_sync();
                unsigned long long ev[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int t = lane + 64 * q;
                    ev[q] = t < m ? (t < n0 ? MF_BUF(r, t) : MF_BUF(r + 32, t - n0)) : 0ull;
                }
                mf_wave_sync();
                unsigned long long kth = 0ull;
                const int keep = min(sc.k, m);
                for (int t = 0; t < keep; ++t) {
                    unsigned long long loc = ev[0] > ev[1] ? ev[0] : ev[1];
                    loc = loc > ev[2] ? loc : ev[2];
                    const unsigned long long best = mf_wave_max_u64(loc);
                    if (ev[0] == best) ev[0] = 0ull;
                    else if (ev[1] == best) ev[1] = 0ull;
                    else if (ev[2] == best) ev[2] = 0ull;
                    if (lane == 0) MF_BUF(r + 32 * (t & 1), t >> 1) = best;
                    kth = best;
                }
                if (lane == r) cnt = (keep + 1) >> 1;
                if (lane == r + 32) cnt = keep >> 1;
                if (lane == 0) floor64[r] = (m >= sc.k) ? kth : 0ull;
                mf_wave_sync();
            }
            fl = floor64[c];
        }
        {
            const unsigned t = max(tau_row, (unsigned)(fl >> 32));
            thr_f = row_ok ? mf_unorderable(t) : __builtin_inff();
            MF_PROBE_T(ps3);
            MF_PROBE_ADD(3, ps3 - ps2);
            MF_PROBE_ADD(1, ps3 - ps0);
#ifdef MF_ABL_NOPASS
            thr_f = __builtin_inff();
#endif
        }
    };

    if (t0 < t1) {
        stage(t0);
        if (t0 + 1 < t1) stage(t0 + 1);
        if (t0 + 1 < t1) mf_wait_vmcnt<NWAIT>(); else mf_wait_vmcnt<0>();
        mf_block_barrier();
        if (L::NSLOT == 3 && t0 + 2 < t1) stage(t0 + 2);
        if (mine(t0)) acc = mf_tile_scores_interleaved<D, 16, (48 * 32 / D)>(smem, xf, [](int) {});
        for (int ty = t0; ty < t1; ++ty) {
            const bool cur = mine(ty), nxt = ty + 1 < t1 && mine(ty + 1);
            if (L::NSLOT == 2) {                      // tile ty's slot is free now: tile ty+2 lands there
                mf_block_barrier();
                if (ty + 2 < t1) stage(ty + 2);
            }
            if (ty + 1 < t1) {
                if (ty + 2 < t1) mf_wait_vmcnt<NWAIT>(); else mf_wait_vmcnt<0>();
                mf_block_barrier();
                if (L::NSLOT == 3 && ty + 3 < t1) stage(ty + 3);
            }
            const int cnt0 = cnt;
            if (cur) {
                tile = Policy::tile_init(pp, row, smem + L::AUX0 + ((ty - t0) & 3) * L::AUXB, wave, c, h);
                y0 = (unsigned)ty * 32u;
            }
            const char* next_tile = smem + ((ty + 1 - t0) % L::NSLOT) * G::TILEB;
            if (cur && nxt) {
                const f32x16 acc_n = mf_tile_scores_interleaved<D, 16, (48 * 32 / D)>(next_tile, xf, slice);
                acc = acc_n;
            } else if (cur) {
#pragma unroll
                for (int e = 0; e < 16; ++e) slice(e);
            } else if (nxt) {
                acc = mf_tile_scores_interleaved<D, 16, (48 * 32 / D)>(next_tile, xf, [](int) {});
            }
            if (cur) settle(cnt0, (ty - t0) < (T + 1) * nsub, smem + L::AUX0 + ((ty - t0) & 3) * L::AUXB);
        }
    }

    // final filter with the final bound, then ship every row's survivors
    filter_segment();
    // a row's two lanes reserve one contiguous piece of the row's list (order of the pieces varies from
    // run to run; the exact selection that follows does not depend on it: keys are unique)
    const int n_other = __shfl_xor(cnt, 32, 64);
    int base = 0;
    if (h == 0 && cnt + n_other > 0) base = atomicAdd(&sc.cand_cnt[x], cnt + n_other);
    base = __shfl(base, c, 64);
    unsigned long long* dst = sc.cand + x * (int64_t)sc.rowcap + base + (h ? n_other : 0);
    for (int t = 0; t < cnt; ++t) dst[t] = MF_BUF(lane, t);
#ifdef MF_PROBE
    {
        MF_PROBE_T(pt_end);
        dbg[0] = pt_end - pt_begin;
        dbg[8] = 1;
        for (int i = 0; i < 9; ++i) {
            unsigned long long v = (i == 4 || i == 5) ? dbg[i] : (lane == 0 ? dbg[i] : 0ull);
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) atomicAdd(&mf_sel_dbg[i], v);
        }
    }
#endif
#undef MF_BUF
}

// geometry shared by the two users of select_kernel
struct SelectPlan {
    int T, CAPL, CAP, xw, nsub, gx, nchunk, tpc, nsets;
    int64_t Xp;
    bool ok;             // false: k too large for the LDS budget at this width
};
static inline SelectPlan mf_select_plan(int64_t nX, int64_t nY, int d, int k) {
    SelectPlan s;
    s.T = mf_select_T(k);
    s.CAPL = mf_select_capl(d);
    s.ok = s.CAPL >= s.T + 16;
    s.CAP = 2 * s.CAPL;
    s.Xp = (nX + 127) / 128 * 128;
    const int xt = (int)((nX + 31) / 32);
    s.xw = xt >= 4 ? 4 : xt >= 2 ? 2 : 1;
    s.nsub = 4 / s.xw;
    s.gx = (xt + s.xw - 1) / s.xw;
    const int YT = (int)((nY + 31) / 32);
    const int max_sets = (64 * 1024) / (s.CAP * 8);          // merge kernel stages all sets of a row in <= 64 KiB LDS
    int want = (256 + s.gx - 1) / s.gx;                      // ~ one workgroup per CU
    if (want * s.nsub > max_sets) want = max_sets / s.nsub;
    if (want > YT) want = YT;
    if (want < 1) want = 1;
    s.tpc = (YT + want - 1) / want;
    s.nchunk = (YT + s.tpc - 1) / s.tpc;
    s.nsets = s.nchunk * s.nsub;
    return s;
}

// Exact ordered selection of the k largest keys staged in LDS `s[0..total)`, by one
// wave; emit(t, key) is called by lane 0 for t = 0..k-1 (key == 0: no more keys).
template <class Emit>
__device__ __forceinline__ void mf_wave_select(unsigned long long* s, int total, int k, Emit emit) {
    const int lane = mf_lane();
    for (int t = 0; t < k; ++t) {
        unsigned long long best = 0ull;
        int bestq = -1;
        for (int q = lane; q < total; q += 64) {
            const unsigned long long kk = s[q];
            if (kk > best) {
                best = kk;
                bestq = q;
            }
        }
        const unsigned long long w = mf_wave_max_u64(best);
        if (w != 0ull && best == w && bestq >= 0) s[bestq] = 0ull;
        if (lane == 0) emit(t, w);
        __syncthreads();
    }
}

#endif  // __HIPCC__
