// mf_select.h -- streaming per-row top-k over MFMA score tiles (gfx950).
//
// One wavefront owns 32 "X" rows (queries / users: one per lane pair) and streams
// 32-row "Y" tiles (catalog items / batch items) through the fp32 MFMA engine of
// mf_common.h.  Nothing of the 32 x N score slab is ever written to HBM: every
// element becomes a unique 64-bit key (mf_numerics.h, larger == better) and only
// keys that can still be among the row's best k survive, in three filters:
//
//   1. each lane keeps the T = ceil(k/2) best 32-bit ranks (key >> 32) it has seen
//      in registers (static insertion network).  tau_row = min over the row's two
//      lanes of their T-th best is a lower bound of the row's k-th best, because
//      >= 2T >= k accepted keys are >= it.  Keys ranked below tau_row are dropped.
//   2. survivors are appended to a per-row LDS buffer (one LDS atomic + one
//      ds_write); when a buffer could overflow on the next tile, its owner lane
//      drops what has fallen below the current tau_row (no sort needed).
//   3. pathological inputs (e.g. all scores equal) defeat 1-2; then the wave sorts
//      that row's buffer exactly and installs a full 64-bit floor.
//
// The surviving candidates of every (row, column-chunk) go to HBM (<= CAP keys
// each); the exact ordered top-k is produced by select_merge (one wave per row),
// which is also the multi-GPU merge.
#pragma once

#include "mf_common.h"

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
    unsigned long long* cand;   // [nchunk][Xp][CAP]
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
//   static bool excluded(P, row, y)        expensive test, evaluated for survivors only
template <int D, int T, int CAP, class Policy>
__global__ __launch_bounds__(64) void select_kernel(typename Policy::Params pp, SelectCommon sc) {
    __shared__ unsigned long long buf[32][CAP + 1];   // +1: lanes of a half hit distinct banks
    __shared__ int cnt[32];
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
    if (lane < 32) {
        cnt[lane] = 0;
        floor64[lane] = 0ull;
    }
    __syncthreads();

    for (int ty = t0; ty < t1; ++ty) {
        const int64_t y0 = (int64_t)ty * 32;
        RowFrag<D> yf;
        mf_load_frag<D>(yf, sc.Y, y0 + c, y0 + c < sc.nY);
        const f32x16 acc = mf_tile_scores<D>(yf, xf);
        typename Policy::Tile tile = Policy::tile_init(pp, row, y0, x);
        const unsigned long long fl = floor64[c];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t y = y0 + mf_acc_row(e, h);
            const unsigned long long key = Policy::key(pp, row, tile, acc[e], e, h, y);
            const unsigned rank = (unsigned)(key >> 32);
            if (key != 0ull && rank >= tau_row && key >= fl) {
                if (!Policy::excluded(pp, row, y)) {
                    if (rank > tl[T - 1]) mf_tlist_insert<T>(tl, rank);
                    const int pos = atomicAdd(&cnt[c], 1);
                    buf[c][pos] = key;
                }
            }
        }
        __syncthreads();
        {
            const unsigned own = tl[T - 1];
            tau_row = min(own, mf_shfl_xor32u(own));
        }
        if (__any(cnt[c] > CAP - 32)) {
            // filter 2: owner lane drops stale keys of its row, in place
            if (lane < 32) {
                const int n0 = cnt[lane];
                const unsigned long long f2 = floor64[lane];
                int w = 0;
                for (int t = 0; t < n0; ++t) {
                    const unsigned long long kk = buf[lane][t];
                    if ((unsigned)(kk >> 32) >= tau_row && kk >= f2) buf[lane][w++] = kk;
                }
                cnt[lane] = w;
            }
            __syncthreads();
            // filter 3 (rare): exact selection of the k best keys of an overfull row
            unsigned long long over = __ballot(lane < 32 && cnt[c] > CAP - 32);
            while (over) {
                const int r = __builtin_ctzll(over);
                over &= over - 1;
                const int m = cnt[r];
                unsigned long long e0 = lane < m ? buf[r][lane] : 0ull;
                unsigned long long e1 = (CAP > 64 && lane + 64 < m) ? buf[r][lane + 64] : 0ull;
                __syncthreads();
                unsigned long long kth = 0ull;
                const int keep = min(sc.k, m);
                for (int t = 0; t < keep; ++t) {
                    const unsigned long long best = mf_wave_max_u64(e0 > e1 ? e0 : e1);
                    if (e0 == best) e0 = 0ull;
                    else if (e1 == best) e1 = 0ull;
                    if (lane == 0) buf[r][t] = best;
                    kth = best;
                }
                if (lane == 0) {
                    cnt[r] = keep;
                    floor64[r] = (m >= sc.k) ? kth : 0ull;
                }
                __syncthreads();
            }
        }
    }

    // final filter with the final tau_row, then ship every row's survivors
    if (lane < 32) {
        const int n0 = cnt[lane];
        const unsigned long long f2 = floor64[lane];
        int w = 0;
        for (int t = 0; t < n0; ++t) {
            const unsigned long long kk = buf[lane][t];
            if ((unsigned)(kk >> 32) >= tau_row && kk >= f2) buf[lane][w++] = kk;
        }
        cnt[lane] = w;
    }
    __syncthreads();
    const int64_t x0 = (int64_t)blockIdx.x * 32;
    for (int r = 0; r < 32; ++r) {
        const int n = cnt[r];
        unsigned long long* dst = sc.cand + ((int64_t)chunk * sc.Xp + x0 + r) * CAP;
        for (int t = lane; t < n; t += 64) dst[t] = buf[r][t];
        if (lane == 0) sc.cand_cnt[(int64_t)chunk * sc.Xp + x0 + r] = n;
    }
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
