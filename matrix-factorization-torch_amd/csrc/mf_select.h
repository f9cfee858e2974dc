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
//      T-list is refreshed from the few new entries once per tile, outside the
//      per-element branch (rare per lane, but "some lane of 64" is the common case).
//      A segment that could overflow on the next tile is filtered in place by its
//      owner against the current tau_row (no sort needed).
//   3. pathological inputs (e.g. all scores equal) defeat 1-2; then the wave selects
//      that row's k best keys exactly and installs a full 64-bit floor.
//
// The surviving candidates of every (row, column-chunk) go to HBM (<= 2 CAPL keys
// each); the exact ordered top-k is produced by a merge kernel (one wave per row).
#pragma once

#include "mf_common.h"

// (T, CAPL) by k: CAPL >= 2 T + 16 so that a filtered segment always has room for a tile
static inline void mf_select_geometry(int k, int* T, int* CAPL) {
    *T = k <= 4 ? 2 : k <= 8 ? 4 : k <= 16 ? 8 : k <= 24 ? 12 : k <= 32 ? 16 : 32;
    *CAPL = *T <= 12 ? 40 : *T == 16 ? 48 : 80;
}

#ifdef __HIPCC__

struct SelectCommon {
    const float* X;      // [nX, D] rows kept on the lanes
    int64_t nX;
    const float* Y;      // [nY, D] rows streamed
    int64_t nY;
    int YT;              // number of 32-row Y tiles
    int tiles_per_chunk;
    int64_t Xp;          // nX padded to 32
    int k;
    unsigned long long* cand;   // [nchunk][Xp][2 CAPL]
    int32_t* cand_cnt;          // [nchunk][Xp]
};

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

// Policy interface:
//   struct Params;                         kernel-argument block
//   struct Row;                            per-lane state of X row x
//   struct Tile;                           per-lane state of the current Y tile
//   static Row  row_init(P, x, valid)
//   static Tile tile_init(P, row, y0, x)
//   static u64  key(P, row, tile, score, e, h, y)   0 = never a candidate
template <int D, int T, int CAPL, class Policy>
__global__ __launch_bounds__(64) void select_kernel(typename Policy::Params pp, SelectCommon sc) {
    __shared__ unsigned long long buf[64][CAPL + 1];   // +1: the 32 lanes of a half hit distinct banks
    __shared__ unsigned long long floor64[32];

    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int64_t x = (int64_t)blockIdx.x * 32 + c;
    const int chunk = blockIdx.y;
    const int t0 = chunk * sc.tiles_per_chunk;
    const int t1 = min(sc.YT, t0 + sc.tiles_per_chunk);

    RowFrag<D> xf;
    mf_load_frag<D>(xf, sc.X, x, x < sc.nX);
    typename Policy::Row row = Policy::row_init(pp, x, x < sc.nX);

    unsigned tl[T];
#pragma unroll
    for (int i = 0; i < T; ++i) tl[i] = 0u;
    unsigned tau_row = 0u;
    unsigned long long fl = 0ull;
    int cnt = 0;
    if (lane < 32) floor64[lane] = 0ull;
    __syncthreads();

    for (int ty = t0; ty < t1; ++ty) {
        const int64_t y0 = (int64_t)ty * 32;
        RowFrag<D> yf;
        mf_load_frag<D>(yf, sc.Y, y0 + c, y0 + c < sc.nY);
        const f32x16 acc = mf_tile_scores<D>(yf, xf);
        typename Policy::Tile tile = Policy::tile_init(pp, row, y0, x);
        const int cnt0 = cnt;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t y = y0 + mf_acc_row(e, h);
            const unsigned long long key = Policy::key(pp, row, tile, acc[e], e, h, y);
            if (key != 0ull && (unsigned)(key >> 32) >= tau_row && key >= fl) {
                buf[lane][cnt] = key;
                ++cnt;
            }
        }
        // refresh the T-list from this tile's accepted keys (few per lane)
        for (int i = cnt0; __any(i < cnt); ++i) {
            if (i < cnt) {
                const unsigned r = (unsigned)(buf[lane][i] >> 32);
                if (r > tl[T - 1]) mf_tlist_insert<T>(tl, r);
            }
        }
        {
            const unsigned own = tl[T - 1];
            tau_row = min(own, mf_shfl_xor32u(own));
        }
        if (__any(cnt > CAPL - 16)) {
            {   // drop, in place, what has fallen below the row's current bound
                int w = 0;
                for (int t = 0; t < cnt; ++t) {
                    const unsigned long long kk = buf[lane][t];
                    if ((unsigned)(kk >> 32) >= tau_row && kk >= fl) buf[lane][w++] = kk;
                }
                cnt = w;
            }
            // rare: a row still too full -> exact selection of its k best keys
            const unsigned long long ovb = __ballot(cnt > CAPL - 16);
            unsigned rows = (unsigned)(ovb | (ovb >> 32));
            while (rows) {
                const int r = __builtin_ctz(rows);
                rows &= rows - 1;
                const int n0 = __shfl(cnt, r, 64), n1 = __shfl(cnt, r + 32, 64);
                const int m = n0 + n1;                       // <= 2 CAPL <= 160
                unsigned long long ev[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int t = lane + 64 * q;
                    ev[q] = t < m ? (t < n0 ? buf[r][t] : buf[r + 32][t - n0]) : 0ull;
                }
                __syncthreads();
                unsigned long long kth = 0ull;
                const int keep = min(sc.k, m);
                for (int t = 0; t < keep; ++t) {
                    unsigned long long loc = ev[0] > ev[1] ? ev[0] : ev[1];
                    loc = loc > ev[2] ? loc : ev[2];
                    const unsigned long long best = mf_wave_max_u64(loc);
                    if (ev[0] == best) ev[0] = 0ull;
                    else if (ev[1] == best) ev[1] = 0ull;
                    else if (ev[2] == best) ev[2] = 0ull;
                    if (lane == 0) buf[r + 32 * (t & 1)][t >> 1] = best;
                    kth = best;
                }
                if (lane == r) cnt = (keep + 1) >> 1;
                if (lane == r + 32) cnt = keep >> 1;
                if (lane == 0) floor64[r] = (m >= sc.k) ? kth : 0ull;
                __syncthreads();
            }
            fl = floor64[c];
        }
    }

    // final filter with the final bound, then ship every row's survivors
    {
        int w = 0;
        for (int t = 0; t < cnt; ++t) {
            const unsigned long long kk = buf[lane][t];
            if ((unsigned)(kk >> 32) >= tau_row && kk >= fl) buf[lane][w++] = kk;
        }
        cnt = w;
    }
    const int n_other = __shfl_xor(cnt, 32, 64);
    unsigned long long* dst = sc.cand + ((int64_t)chunk * sc.Xp + x) * (2 * CAPL) + (h ? n_other : 0);
    for (int t = 0; t < cnt; ++t) dst[t] = buf[lane][t];
    if (h == 0) sc.cand_cnt[(int64_t)chunk * sc.Xp + x] = cnt + n_other;
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
