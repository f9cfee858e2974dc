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
//      best, because >= 2T >= k accepted keys are >= it; the best bound any chunk of the
//      launch has found for the row is shared through HBM (gtau, max-published).  Keys
//      ranked below the bound are dropped by one compare.
//   2. a surviving key is appended to the lane's PRIVATE candidate list in HBM with one
//      fire-and-forget 8-byte store (the fill count lives in a register: no atomics, no
//      cross-lane traffic, and no LDS -- the workgroup's LDS is the tile ring alone, so two
//      workgroups share a CU and one's candidate handling overlaps the other's MFMAs).
//      While the T-lists fill (the chunk's first tiles) every accepted rank is inserted;
//      afterwards only the tile's best one (one branch-free insertion per tile).  A list that
//      could overflow on the next tile is filtered in place by its owner against the current
//      bound (read back through L2).
//   3. pathological inputs (e.g. all scores equal) defeat 1-2; then the wave selects
//      that row's k best keys exactly and installs a full 64-bit floor.
//
// At the end of its chunk a lane filters its list once more and appends the survivors to the
// row's list (atomic cursor); the exact ordered top-k is produced by a merge kernel (one wave
// per row).
#pragma once

#include <type_traits>

#include "mf_common.h"

// T = per-lane register list length (>= ceil(k / 2)); CAPH = per-lane private list capacity (HBM)
static inline int mf_select_T(int k) { return k <= 4 ? 2 : k <= 8 ? 4 : k <= 16 ? 8 : k <= 20 ? 10 : k <= 24 ? 12 : k <= 32 ? 16 : 32; }
static constexpr int MF_SELECT_CAPH = 64;

#ifdef __HIPCC__

#include "mf_stream.h"

struct SelectCommon {
    const float* X;      // [nX, D] rows kept on the lanes
    int64_t nX;
    const float* Y;      // [nY, D] rows streamed
    int64_t nY;
    int t_begin, t_end;  // this launch streams the 32-row Y tiles [t_begin, t_end)
    int tiles_per_chunk;
    int64_t Xp;          // nX padded to 128
    int k;
    int xw;              // X tiles per workgroup (1, 2, 4 or 8); the NW / xw waves that share an X tile
                         // deal the chunk's Y tiles round-robin and keep separate candidate lists
    unsigned* gtau;      // [Xp], zeroed by the host: best known lower bound (rank) of every X row's k-th best key,
                         // shared by all chunks of the launch (max-published, so always a valid bound)
    unsigned long long* priv;   // [nsets][Xp][2 lanes][CAPH]: lane-private lists of the running chunk
    unsigned long long* cand;   // [Xp][rowcap]: every X row's surviving keys of ALL chunks, contiguous
    int32_t* cand_cnt;          // [Xp], zeroed by the host: fill count of the row's list (atomic cursor)
    int rowcap;                 // nsets * 2 CAPH
    const int32_t* gate;        // not NULL: the launch does nothing unless *gate is set (mf_mine_bf.h: the fallback behind the prefilter)
    int tstride;                // seeding pass: virtual tile t of [t_begin, t_end) is real tile t * tstride (0 or 1: contiguous)
};

#ifdef MF_PROBE
// tools/topk_probe.py: wave-summed cycle / event counters of the last launches
// 0 total cycles, 1 settle cycles, 2 (unused), 3 compact+slow-path cycles, 4 accepted keys,
// 5 slices whose body ran (any lane passed the prefilter), 6 compact calls, 7 slow-path rows, 8 waves
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

// candidate lists live in HBM: appended with fire-and-forget stores (asm: the compiler must not tie a
// wait to them), read back -- rarely -- past the L1 (the wave wrote these lines itself, through to L2)
__device__ __forceinline__ void mf_cand_store(unsigned long long* p, unsigned lo, unsigned hi) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long mf_cand_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int D>
struct SelectLds {
    using G = TileGeom<D>;
    static constexpr int W0 = G::NW * 128;               // [NW x 128 B per-wave words][W0: nv, logq, copies][GT0: NW x 128 B gtau]
    static constexpr int GT0 = 2 * W0;                   // per-wave copies of gtau[x0 .. x0 + 31]
    static constexpr int AUXB = 3 * W0;
    static constexpr int NSLOT = D == 256 ? 2 : 3;       // d = 256: 2-deep tile ring, 2 barriers per tile
    static constexpr int AUX0 = NSLOT * G::TILEB;        // 4 side-input slots after the tile slots
    static constexpr int RING = AUX0 + 4 * AUXB;
    static constexpr int BYTES = RING + G::NW * 32 * 8;  // + the exact 64-bit floors of the degenerate path
};

// Policy interface:
//   struct Params;  struct Row;  struct Tile;
//   static constexpr int AUX_DMA;                   DMA instructions per wave per stage for side inputs
//   static void stage_aux(P, aux, wave, t, x0, W0)  issue them (side inputs of Y tile t for X rows x0..x0+31; W0 = NW x 128)
//   static Row  row_init(P, x, valid)
//   static Tile tile_init(P, row, aux, wave, c, h, W0)  read the staged side inputs
//   static bool key(P, row, tile, score, e, h, y, hi&, lo&)   false = never a candidate; (hi, lo) = key halves
//   struct Thr;  static Thr thr_all(), thr_none(), make_thr(unsigned rank_bound)
//   static bool maybe(P, row, tile, score, e, h, thr)   cheap, CONSERVATIVE test of `rank >= bound` on the raw score:
//                                                    false only if key() could not pass the exact test
// (T = 32 -- k in 33..64 -- needs more than 256 registers: compiled for one workgroup per CU instead of spilling)
template <int D, int T, class Policy>
__global__ __launch_bounds__(64 * mf_nw(D), T >= 32 ? 1 : mf_wg_per_cu(D)) void select_kernel(typename Policy::Params pp, SelectCommon sc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = TileGeom<D>;
    using L = SelectLds<D>;
    constexpr int NWAIT = G::PPW + Policy::AUX_DMA + 1;
    constexpr int CAPH = MF_SELECT_CAPH;
    if (sc.gate && __hip_atomic_load(sc.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;      // (uniform: every wave of the grid leaves)

    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
    const int nsub = G::NW / sc.xw;
    const int sub = wave / sc.xw;
    // grid = (Y chunk, X block): the workgroups of one XCD (ids 8 apart) share chunks -> the catalog
    // slice they stream stays in that XCD's L2
    const int64_t x0 = ((int64_t)blockIdx.y * sc.xw + (wave % sc.xw)) * 32;
    const int64_t x = x0 + c;
    const int chunk = blockIdx.x;
    const int t0 = sc.t_begin + chunk * sc.tiles_per_chunk;
    const int t1 = min(sc.t_end, t0 + sc.tiles_per_chunk);
    const int64_t set = (int64_t)chunk * nsub + sub;

    unsigned long long* floor64 = reinterpret_cast<unsigned long long*>(smem + L::RING) + wave * 32;
    auto list_of = [&](int r, int hh) {      // private list of X row x0 + r, lane half hh
        return sc.priv + ((set * sc.Xp + x0 + r) * 2 + hh) * CAPH;
    };
    unsigned long long* mine = list_of(c, h);

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

    TileSrc<D> tsrc;
    mf_tile_src_init<D>(tsrc, sc.Y, sc.nY, (int64_t)t0 * 32);
    auto stage = [&](int t) {
        const int kk = t - t0;
        mf_stage_tile<D>(smem + (kk % L::NSLOT) * G::TILEB, t * 32, tsrc);
        Policy::stage_aux(pp, smem + L::AUX0 + (kk & 3) * L::AUXB, wave, t, x0, L::W0);
        mf_stage_small<17>(smem + L::AUX0 + (kk & 3) * L::AUXB + L::GT0 + wave * 128, sc.gtau + x0, 128);
    };
    auto mine_tile = [&](int t) { return (t - t0) % nsub == sub; };   // else another wave of this X tile takes tile t

    // Software pipeline inside the wave (as in the loss forward): the MFMAs of tile t+1 are issued
    // in the same basic block as the per-element key work of tile t (16 slices).
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typename Policy::Tile tile;
    unsigned tmaxr = 0u;                         // best rank this lane accepted in the current tile
    const bool row_ok = x < sc.nX;               // padding rows (all-zero X) never collect candidates
    unsigned y0 = 0u;                            // first Y row of the current tile (rows < 2^32)
#ifdef MF_ABL_NOPASS
    typename Policy::Thr thr = Policy::thr_none();
#else
    typename Policy::Thr thr = Policy::thr_all();            // everything passes until a bound exists
#endif
    unsigned pub = 0u;                                       // last bound this lane published
    // 32-bit-only fast path: rank (= high key word) against the row bound; the exact 64-bit floor of
    // the degenerate path is checked by halves as well.  WARM: the T-lists are still filling.
    auto slice_t = [&](int e, auto warm_tag) {
        constexpr bool WARM = decltype(warm_tag)::value;
        // a cheap conservative test on the raw score first (retrieval: one compare), everything exact
        // happens behind the branch
        if (!Policy::maybe(pp, row, tile, acc[e], e, h, thr)) return;
#ifdef MF_PROBE
        if (lane == (int)__builtin_ctzll(__ballot(1))) dbg[5] += 1;
#endif
        unsigned hi, lo;
        const bool ok = Policy::key(pp, row, tile, acc[e], e, h, y0 + (unsigned)mf_acc_row(e, h), hi, lo);
        const unsigned fhi = (unsigned)(fl >> 32), flo = (unsigned)fl;
        const bool above_floor = hi > fhi || (hi == fhi && lo >= flo);
        if (ok && row_ok && hi >= tau_row && above_floor) {
            mf_cand_store(mine + cnt, lo, hi);
            ++cnt;
            MF_PROBE_ADD(4, 1);
            if (WARM) {
                if (hi > tl[T - 1]) mf_tlist_insert<T>(tl, hi);
            } else {
                tmaxr = max(tmaxr, hi);
            }
        }
    };
    auto slice_warm = [&](int e) { slice_t(e, std::true_type{}); };
    auto slice_hot = [&](int e) { slice_t(e, std::false_type{}); };

    // In-place filter of the lane's own list against the current bound, 8 independent loads per round
    // (a store only goes to a position already read).  Rare: drains the wave's memory queue first.
    auto compact = [&]() {
        mf_wait_vmcnt<0>();
        int w = 0;
        for (int t0f = 0; __any(t0f < cnt); t0f += 8) {
            unsigned long long kk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) kk[j] = (t0f + j < cnt) ? mf_cand_load(mine + t0f + j) : 0ull;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kk[j] != 0ull && (unsigned)(kk[j] >> 32) >= tau_row && kk[j] >= fl) {
                    mf_cand_store(mine + w, (unsigned)kk[j], (unsigned)(kk[j] >> 32));
                    ++w;
                }
        }
        cnt = w;
        mf_wait_vmcnt<0>();
    };
    // after the slices of a tile: refresh the bound, keep the lists from overflowing.
    // Steady state: only the tile's BEST accepted rank enters the lane's T-list -- its entries are then
    // T distinct accepted elements, so tl[T-1] stays a valid and, with the row's best spread over many
    // tiles, nearly tight lower bound, at the cost of one branch-free insertion per tile.
    auto settle = [&](const char* aux) {
        MF_PROBE_T(ps0);
        if (tmaxr > tl[T - 1]) mf_tlist_insert<T>(tl, tmaxr);
        tmaxr = 0u;
        {
            const unsigned own = tl[T - 1];
            const unsigned mine_ = min(own, mf_shfl_xor32u(own));
            // every chunk's bound is a valid bound of the row: take the best one published so far
            // (the copy staged with this tile; a stale value only prunes less) and publish ours
            const unsigned g = reinterpret_cast<const unsigned*>(aux + L::GT0 + wave * 128)[c];
            if (h == 0 && mine_ > g && mine_ > pub) {
                mf_global_umax(sc.gtau + x, mine_);
                pub = mine_;
            }
            tau_row = max(mine_, g);
        }
        MF_PROBE_T(ps2);
        if (__any(cnt > CAPH - 16)) {
            MF_PROBE_ADD(6, lane == 0 ? 1 : 0);
            compact();          // drop, in place, what has fallen below the row's current bound
            // rare: a row still too full -> exact selection of its k best keys (wave-local)
            const unsigned long long ovb = __ballot(cnt > CAPH - 16);
            unsigned rows = (unsigned)(ovb | (ovb >> 32));
            while (rows) {
                const int r = __builtin_ctz(rows);
                rows &= rows - 1;
                MF_PROBE_ADD(7, lane == 0 ? 1 : 0);
                const int n0 = __shfl(cnt, r, 64), n1 = __shfl(cnt, r + 32, 64);
                const int m = n0 + n1;                       // <= 2 CAPH = 128
                const unsigned long long* l0 = list_of(r, 0);
                const unsigned long long* l1 = list_of(r, 1);
                unsigned long long ev[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int t = lane + 64 * q;
                    ev[q] = t < m ? (t < n0 ? mf_cand_load(l0 + t) : mf_cand_load(l1 + (t - n0))) : 0ull;
                }
                unsigned long long kth = 0ull;
                const int keep = min(sc.k, m);
                for (int t = 0; t < keep; ++t) {
                    const unsigned long long loc = ev[0] > ev[1] ? ev[0] : ev[1];
                    const unsigned long long best = mf_wave_max_u64(loc);
                    if (ev[0] == best) ev[0] = 0ull;
                    else if (ev[1] == best) ev[1] = 0ull;
                    if (lane == 0) mf_cand_store(list_of(r, t & 1) + (t >> 1), (unsigned)best, (unsigned)(best >> 32));
                    kth = best;
                }
                if (lane == r) cnt = (keep + 1) >> 1;
                if (lane == r + 32) cnt = keep >> 1;
                if (lane == 0) floor64[r] = (m >= sc.k) ? kth : 0ull;
                mf_wait_vmcnt<0>();
                mf_wave_sync();
            }
            mf_wave_sync();
            fl = floor64[c];
        }
        {
            const unsigned t = max(tau_row, (unsigned)(fl >> 32));
            thr = row_ok ? Policy::make_thr(t) : Policy::thr_none();
            MF_PROBE_T(ps3);
            MF_PROBE_ADD(3, ps3 - ps2);
            MF_PROBE_ADD(1, ps3 - ps0);
#ifdef MF_ABL_NOPASS
            thr = Policy::thr_none();
#endif
        }
    };

    if (t0 < t1) {
        stage(t0);
        if (t0 + 1 < t1) stage(t0 + 1);
        if (t0 + 1 < t1) mf_wait_vmcnt<NWAIT>(); else mf_wait_vmcnt<0>();
        mf_block_barrier();
        if (L::NSLOT == 3 && t0 + 2 < t1) stage(t0 + 2);
        if (mine_tile(t0)) acc = mf_tile_scores_interleaved<D, 16, (48 * 32 / D)>(smem, xf, [](int) {});
        for (int ty = t0; ty < t1; ++ty) {
            const bool cur = mine_tile(ty), nxt = ty + 1 < t1 && mine_tile(ty + 1);
            if (L::NSLOT == 2) {                      // tile ty's slot is free now: tile ty+2 lands there
                mf_block_barrier();
                if (ty + 2 < t1) stage(ty + 2);
            }
            if (ty + 1 < t1) {
                if (ty + 2 < t1) mf_wait_vmcnt<NWAIT>(); else mf_wait_vmcnt<0>();
                mf_block_barrier();
                if (L::NSLOT == 3 && ty + 3 < t1) stage(ty + 3);
            }
            const char* aux = smem + L::AUX0 + ((ty - t0) & 3) * L::AUXB;
            if (cur) {
                tile = Policy::tile_init(pp, row, aux, wave, c, h, L::W0);
                y0 = (unsigned)ty * 32u;
            }
            const bool warm = (ty - t0) < 2 * nsub;   // this wave's first two tiles: the T-lists fill
            const char* next_tile = smem + ((ty + 1 - t0) % L::NSLOT) * G::TILEB;
            if (cur && nxt) {
                f32x16 acc_n;
                if (warm) acc_n = mf_tile_scores_interleaved<D, 16, (48 * 32 / D)>(next_tile, xf, slice_warm);
                else acc_n = mf_tile_scores_interleaved<D, 16, (48 * 32 / D)>(next_tile, xf, slice_hot);
                acc = acc_n;
            } else if (cur) {
                if (warm) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) slice_warm(e);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) slice_hot(e);
                }
            } else if (nxt) {
                acc = mf_tile_scores_interleaved<D, 16, (48 * 32 / D)>(next_tile, xf, [](int) {});
            }
            if (cur) settle(aux);
        }
    }

    // final filter with the final bound, then append every row's survivors to the row's list: a row's
    // two lanes reserve one contiguous piece (the order of the pieces varies from run to run; the exact
    // selection that follows does not depend on it: keys are unique)
    compact();
    const int n_other = __shfl_xor(cnt, 32, 64);
    int base = 0;
    if (h == 0 && cnt + n_other > 0) base = atomicAdd(&sc.cand_cnt[x], cnt + n_other);
    base = __shfl(base, c, 64);
    unsigned long long* dst = sc.cand + x * (int64_t)sc.rowcap + base + (h ? n_other : 0);
    for (int t8 = 0; __any(t8 < cnt); t8 += 8) {
        unsigned long long kk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kk[j] = (t8 + j < cnt) ? mf_cand_load(mine + t8 + j) : 0ull;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (t8 + j < cnt) dst[t8 + j] = kk[j];
    }
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
}

// ------------------------------------------------------------ seeding pass -------
// A streaming selection that starts with no bound accepts ~T ln(n/T) keys per lane before its
// threshold is tight, and on fp32 MFMA every accepted key costs matrix time.  So a launch first runs
// this branch-free pass over a slice of Y (1/8 of the tiles): per lane, the best rank of each tile
// (16 max operations) feeds a T-list (one insertion network per tile) -- nothing is stored, nothing
// branches, the VALU work threads between the MFMAs.  The T-list entries of all lanes and chunks of a
// row are distinct elements, so the k-th largest of them (select_bound_kernel) is a valid lower bound
// of the row's k-th best key; the main pass then starts from it and accepts only a few keys per row.
template <int D, int T, class Policy>
__global__ __launch_bounds__(64 * mf_nw(D), T >= 32 ? 1 : mf_wg_per_cu(D)) void select_seed_kernel(typename Policy::Params pp, SelectCommon sc,
                                                                            unsigned long long* __restrict__ seeds,
                                                                            int seeds_per_row) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = TileGeom<D>;
    using L = SelectLds<D>;
    constexpr int NWAIT = G::PPW + Policy::AUX_DMA;
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
    const int nsub = G::NW / sc.xw;
    const int sub = wave / sc.xw;
    const int64_t x0 = ((int64_t)blockIdx.y * sc.xw + (wave % sc.xw)) * 32;
    const int64_t x = x0 + c;
    const int chunk = blockIdx.x;
    const int t0 = sc.t_begin + chunk * sc.tiles_per_chunk;
    const int t1 = min(sc.t_end, t0 + sc.tiles_per_chunk);
    const int set = chunk * nsub + sub;

    RowFrag<D> xf;
    mf_load_frag<D>(xf, sc.X, x, x < sc.nX);
    typename Policy::Row row = Policy::row_init(pp, x, x < sc.nX);
    const bool row_ok = x < sc.nX;
    unsigned tl[T];
#pragma unroll
    for (int i = 0; i < T; ++i) tl[i] = 0u;

    // A STRIDED sample (round 4): the first eighth of a batch's columns are the positives of its first users -- popular items --
    // while the uniform negatives sit at the end: a bound from the head alone missed the cut of ~4 % of the users by a wide margin.
    const int ts = sc.tstride > 1 ? sc.tstride : 1;
    TileSrc<D> tsrc;
    mf_tile_src_init<D>(tsrc, sc.Y, sc.nY, (int64_t)t0 * ts * 32);
    auto stage = [&](int t) {
        const int kk = t - t0;
        mf_stage_tile<D>(smem + (kk % L::NSLOT) * G::TILEB, t * ts * 32, tsrc);
        Policy::stage_aux(pp, smem + L::AUX0 + (kk & 3) * L::AUXB, wave, t * ts, x0, L::W0);
    };
    auto mine_tile = [&](int t) { return (t - t0) % nsub == sub; };
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typename Policy::Tile tile;
    unsigned tmaxr = 0u, y0 = 0u;
    auto slice = [&](int e) {
        unsigned hi, lo;
        const bool ok = Policy::key(pp, row, tile, acc[e], e, h, y0 + (unsigned)mf_acc_row(e, h), hi, lo);
        tmaxr = max(tmaxr, (ok && row_ok) ? hi : 0u);
    };
    auto settle = [&]() {
        tl[T - 1] = max(tl[T - 1], tmaxr);        // branch-free insertion: a rank below the list changes nothing
#pragma unroll
        for (int i = T - 1; i > 0; --i) {
            const unsigned a = max(tl[i - 1], tl[i]), b = min(tl[i - 1], tl[i]);
            tl[i - 1] = a;
            tl[i] = b;
        }
        tmaxr = 0u;
    };
    if (t0 < t1) {
        stage(t0);
        if (t0 + 1 < t1) stage(t0 + 1);
        if (t0 + 1 < t1) mf_wait_vmcnt<NWAIT>(); else mf_wait_vmcnt<0>();
        mf_block_barrier();
        if (L::NSLOT == 3 && t0 + 2 < t1) stage(t0 + 2);
        if (mine_tile(t0)) acc = mf_tile_scores_interleaved<D, 16, (48 * 32 / D)>(smem, xf, [](int) {});
        for (int ty = t0; ty < t1; ++ty) {
            const bool cur = mine_tile(ty), nxt = ty + 1 < t1 && mine_tile(ty + 1);
            if (L::NSLOT == 2) {
                mf_block_barrier();
                if (ty + 2 < t1) stage(ty + 2);
            }
            if (ty + 1 < t1) {
                if (ty + 2 < t1) mf_wait_vmcnt<NWAIT>(); else mf_wait_vmcnt<0>();
                mf_block_barrier();
                if (L::NSLOT == 3 && ty + 3 < t1) stage(ty + 3);
            }
            if (cur) {
                tile = Policy::tile_init(pp, row, smem + L::AUX0 + ((ty - t0) & 3) * L::AUXB, wave, c, h, L::W0);
                y0 = (unsigned)(ty * ts) * 32u;
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
            if (cur) settle();
        }
    }
    // keys (rank, unique tag) so that the row selection can treat them like any key list; 0 = empty
    unsigned long long* out = seeds + x * (int64_t)seeds_per_row + ((int64_t)set * 2 + h) * T;
#pragma unroll
    for (int i = 0; i < T; ++i)
        out[i] = tl[i] ? (((unsigned long long)tl[i] << 32) | (unsigned)((set * 2 + h) * T + i + 1)) : 0ull;
}

// ------------------------------------------------------------ row lists -> top-k ----
// Exact ordered top-k of a row's key list in HBM by ONE wave, without sorting the list: the k-th
// largest key is found by bit-wise threshold search over the keys held in registers (32 steps on the
// rank word; 32 more on the column word only if equal ranks straddle the cut), the <= k winners are
// compacted and ordered by rank counting.  Lists longer than 64 KPL keys go through in batches, the
// winners so far riding along.  Returns m = min(n, k); sorted[0..m) (LDS) is in descending key order.
// WAVE_LOCAL: the calling wave is one of several in its workgroup, each on its OWN win / sorted arrays -- a workgroup barrier
// would be wrong (the waves run different trip counts); the LDS executes one wave's operations in order, so ordering the
// compiler is enough.
template <bool WAVE_LOCAL>
__device__ __forceinline__ void mf_row_topk_sync() {
    if (WAVE_LOCAL) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    else mf_row_topk_sync<WAVE_LOCAL>();
}
template <int KPL, bool WAVE_LOCAL = false>
__device__ __forceinline__ int mf_row_topk(const unsigned long long* __restrict__ src, int n, int k,
                                           unsigned long long* win, unsigned long long* sorted) {
    const int lane = mf_lane();
    const unsigned long long below = (1ull << lane) - 1ull;
    int carry = 0;
    for (int base = 0; base < n; base += 64 * KPL) {
        unsigned long long v[KPL + 1];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int idx = base + lane + 64 * j;
            v[j] = idx < n ? src[idx] : 0ull;
        }
        v[KPL] = lane < carry ? win[lane] : 0ull;
        mf_row_topk_sync<WAVE_LOCAL>();                                   // win[] was read by everyone before it is rewritten
        const int have = min(n - base, 64 * KPL) + carry;
        unsigned long long tau = 1ull;                     // keeps every real key
        if (have > k) {
            unsigned th = 0u;
            for (int b = 31; b >= 0; --b) {                // largest th with #{rank >= th} >= k
                const unsigned cnd = th | (1u << b);
                int cge = 0;
#pragma unroll
                for (int j = 0; j <= KPL; ++j) cge += __popcll(__ballot((unsigned)(v[j] >> 32) >= cnd));
                if (cge >= k) th = cnd;
            }
            int above = 0, equal = 0;
#pragma unroll
            for (int j = 0; j <= KPL; ++j) {
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
                    for (int j = 0; j <= KPL; ++j)
                        cge += __popcll(__ballot(v[j] != 0ull && (unsigned)(v[j] >> 32) == th && (unsigned)v[j] >= cnd));
                    if (cge >= need) tlo = cnd;
                }
            }
            tau = ((unsigned long long)th << 32) | tlo;
        }
        int pos = 0;
#pragma unroll
        for (int j = 0; j <= KPL; ++j) {
            const bool wj = v[j] != 0ull && v[j] >= tau;
            const unsigned long long m = __ballot(wj);
            if (wj) win[pos + __popcll(m & below)] = v[j];
            pos += __popcll(m);
        }
        carry = pos;                                       // == min(#real keys, k)
        mf_row_topk_sync<WAVE_LOCAL>();
    }
    const int m = carry;
    if (lane < m) {
        const unsigned long long mine = win[lane];
        int r = 0;
        for (int q = 0; q < m; ++q) r += win[q] > mine ? 1 : 0;
        sorted[r] = mine;
    }
    mf_row_topk_sync<WAVE_LOCAL>();
    return m;
}

// Between the seeding pass and the main pass: a row's starting bound from its seed keys.  Any k
// distinct seeds give a valid bound (their minimum); the cheap choice here: every lane reduces its
// share of the row's seeds to its best one, and the k-th largest of the 64 lane maxima (rank counting
// over LDS) is taken -- within a few ranks of the exact k-th best seed, at a twentieth of its cost.
template <int DUMMY>   // (a template so that both translation units may include the definition)
__global__ __launch_bounds__(256) void select_bound_kernel(const unsigned long long* __restrict__ seeds, int seeds_per_row, int k,
                                                           int64_t nX, unsigned* __restrict__ gtau) {
    __shared__ unsigned lane_best[4][64];
    const int lane = mf_lane(), wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    unsigned best = 0u;
    if (r < nX) {
        const unsigned long long* src = seeds + r * (int64_t)seeds_per_row;
        for (int i = lane; i < seeds_per_row; i += 64) best = max(best, (unsigned)(src[i] >> 32));
    }
    lane_best[wv][lane] = best;
    __syncthreads();
    int above = 0;                    // lanes whose maximum beats mine (ties broken by lane: a strict order)
    for (int q = 0; q < 64; ++q) {
        const unsigned o = lane_best[wv][q];
        above += (o > best || (o == best && q < lane)) ? 1 : 0;
    }
    if (r < nX && above == k - 1 && best != 0u) gtau[r] = best;
}

// geometry shared by the two users of select_kernel
struct SelectPlan {
    int T, CAP, xw, nsub, gx, YT;
    int YTa, tpcA, nchunkA;      // seeding pass: YTa sampled tiles (every seed_div-th) in nchunkA chunks (YTa = 0: none)
    int seed_div;
    int seeds_per_row;           // nchunkA * nsub * 2 T seed keys per row
    int tpc, nchunk;             // main pass: all YT tiles
    int nsets;                   // lane-private list sets of the main pass
    int rowcap;                  // capacity of a row's list
    int64_t Xp;
    bool ok;                     // false: k too large for the per-lane lists
};
static inline SelectPlan mf_select_plan(int64_t nX, int64_t nY, int d, int k) {
    SelectPlan s{};
    s.T = mf_select_T(k);
    s.CAP = 2 * MF_SELECT_CAPH;                               // a row's two lanes
    s.ok = MF_SELECT_CAPH >= s.T + 16 && k <= MF_SELECT_CAPH;
    const int nw = mf_nw(d);
    s.Xp = (nX + 32 * nw - 1) / (32 * nw) * (32 * nw);
    const int xt = (int)((nX + 31) / 32);
    s.xw = (xt >= 8 && nw == 8) ? 8 : xt >= 4 ? 4 : xt >= 2 ? 2 : 1;
    s.nsub = nw / s.xw;
    s.gx = (xt + s.xw - 1) / s.xw;
    s.YT = (int)((nY + 31) / 32);
    const int wgs = 2048 / nw;                               // two waves per SIMD: 256 workgroups of eight waves
    int want = (wgs + s.gx - 1) / s.gx;
    if (want > 128) want = 128;
    // a seeding pass over 1/8 of Y pays when Y is long enough to profit from its bound
#ifndef MF_SEED_DIV
#define MF_SEED_DIV 8      // A/B knob: the seeding pass scans 1 / MF_SEED_DIV of Y
#endif
    // (a quarter instead of an eighth of Y for k > 16: the bound is the k-th best of the sample, ~ the (k x seed_div)-th best of the
    // row -- at k = 32 an eighth let ~8 k = 256 columns per row through)
    s.seed_div = k > 16 ? MF_SEED_DIV / 2 : MF_SEED_DIV;
    s.YTa = (s.YT >= 64) ? s.YT / s.seed_div : 0;
    if (s.YTa > 0) {
        int na = want < s.YTa / 4 ? want : s.YTa / 4;        // >= 4 tiles per chunk
        if (na < 1) na = 1;
        s.tpcA = (s.YTa + na - 1) / na;
        s.nchunkA = (s.YTa + s.tpcA - 1) / s.tpcA;
    }
    if (want > s.YT) want = s.YT;
    if (want < 1) want = 1;
    // a chunk is staged through ONE 32-bit buffer descriptor based at its first row (mf_stream.h): its tiles must fit
    // MF_SRD_MAX_BYTES, or rows past that would arrive as zeros -- unmasked, since they are below nY
    {
        const int64_t tile_bytes = 32ll * d * 4, max_tpc = (int64_t)0xFFF00000ll / tile_bytes;
        const int64_t min_chunks = (s.YT + max_tpc - 1) / max_tpc;
        if (min_chunks > want) want = (int)(min_chunks < s.YT ? min_chunks : s.YT);
        if (min_chunks > 128) s.ok = false;          // (a catalog beyond 512 GiB: sharded long before)
    }
    s.tpc = (s.YT + want - 1) / want;
    s.nchunk = (s.YT + s.tpc - 1) / s.tpc;
    s.nsets = s.nchunk * s.nsub;
    s.rowcap = s.nsets * s.CAP;
    s.seeds_per_row = s.nchunkA * s.nsub * 2 * s.T;
    return s;
}

template <int D, int T, class Policy>
static void mf_select_launch_t(const typename Policy::Params& pp, const SelectCommon& sc, int nchunk, int gx, hipStream_t s) {
    auto fn = select_kernel<D, T, Policy>;
    const int bytes = SelectLds<D>::BYTES;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    fn<<<dim3((unsigned)nchunk, (unsigned)gx), 64 * mf_nw(D), bytes, s>>>(pp, sc);
}
template <int D, class Policy>
static void mf_select_launch(int T, const typename Policy::Params& pp, const SelectCommon& sc, int nchunk, int gx, hipStream_t s) {
    switch (T) {
        case 2: mf_select_launch_t<D, 2, Policy>(pp, sc, nchunk, gx, s); break;
        case 4: mf_select_launch_t<D, 4, Policy>(pp, sc, nchunk, gx, s); break;
        case 8: mf_select_launch_t<D, 8, Policy>(pp, sc, nchunk, gx, s); break;
        case 10: mf_select_launch_t<D, 10, Policy>(pp, sc, nchunk, gx, s); break;
        case 12: mf_select_launch_t<D, 12, Policy>(pp, sc, nchunk, gx, s); break;
        case 16: mf_select_launch_t<D, 16, Policy>(pp, sc, nchunk, gx, s); break;
        default: mf_select_launch_t<D, 32, Policy>(pp, sc, nchunk, gx, s); break;
    }
}
template <int D, int T, class Policy>
static void mf_select_seed_t(const typename Policy::Params& pp, const SelectCommon& sc, unsigned long long* seeds,
                             int seeds_per_row, int nchunk, int gx, hipStream_t s) {
    auto fn = select_seed_kernel<D, T, Policy>;
    const int bytes = SelectLds<D>::BYTES;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    fn<<<dim3((unsigned)nchunk, (unsigned)gx), 64 * mf_nw(D), bytes, s>>>(pp, sc, seeds, seeds_per_row);
}
// seeding pass + bound + main pass; `sc` arrives with everything but the tile range filled in, gtau and
// cand_cnt zeroed; `seeds`: [Xp][plan.seeds_per_row] scratch
template <int D, class Policy>
static void mf_select_run(const SelectPlan& plan, const typename Policy::Params& pp, SelectCommon sc,
                          unsigned long long* seeds, int64_t nX, hipStream_t s, bool seed_only = false, bool main_only = false,
                          bool no_bound = false) {
    if (plan.YTa > 0 && !main_only) {
        sc.t_begin = 0; sc.t_end = plan.YTa; sc.tiles_per_chunk = plan.tpcA;
        sc.tstride = plan.seed_div;                     // tile t of the sample is tile seed_div t (YTa = YT / seed_div)
        switch (plan.T) {
            case 2: mf_select_seed_t<D, 2, Policy>(pp, sc, seeds, plan.seeds_per_row, plan.nchunkA, plan.gx, s); break;
            case 4: mf_select_seed_t<D, 4, Policy>(pp, sc, seeds, plan.seeds_per_row, plan.nchunkA, plan.gx, s); break;
            case 8: mf_select_seed_t<D, 8, Policy>(pp, sc, seeds, plan.seeds_per_row, plan.nchunkA, plan.gx, s); break;
            case 10: mf_select_seed_t<D, 10, Policy>(pp, sc, seeds, plan.seeds_per_row, plan.nchunkA, plan.gx, s); break;
            case 12: mf_select_seed_t<D, 12, Policy>(pp, sc, seeds, plan.seeds_per_row, plan.nchunkA, plan.gx, s); break;
            case 16: mf_select_seed_t<D, 16, Policy>(pp, sc, seeds, plan.seeds_per_row, plan.nchunkA, plan.gx, s); break;
            default: mf_select_seed_t<D, 32, Policy>(pp, sc, seeds, plan.seeds_per_row, plan.nchunkA, plan.gx, s); break;
        }
        if (!no_bound)       // (mf_mine_bf.h finds the bound in a launch of its own that also prepares the scan's intervals)
            select_bound_kernel<0><<<dim3((unsigned)((nX + 3) / 4)), 256, 0, s>>>(seeds, plan.seeds_per_row, sc.k, nX, sc.gtau);
    }
    if (seed_only) return;          // (mf_mine_bf.h: the bound is all the prefilter wants from here)
    sc.tstride = 1;
    sc.t_begin = 0; sc.t_end = plan.YT; sc.tiles_per_chunk = plan.tpc;
    mf_select_launch<D, Policy>(plan.T, pp, sc, plan.nchunk, plan.gx, s);
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
