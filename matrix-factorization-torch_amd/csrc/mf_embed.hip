// mf_embed.hip -- embedding-table rows: gather (tower forward), chain norms, the
// stable id sort and the sparse SGD / row-wise Adam updates.  All HBM-bound:
// every row moves as 16-byte lane accesses, d/4 consecutive lanes per row, so one
// wave-instruction covers whole 128..1024-byte rows (coalesced).
//
// Reference interfaces replaced: xfmr_rec/lightning.py:60-74 (tower forward),
// :238-239 (optimiser).  Neither has a table-based implementation upstream; the
// spec is oracle/embed.py.
#include "mf_common.h"

// ------------------------------------------------------------------ gather ----
template <int D>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table,
                                                          int64_t n_rows,
                                                          const int64_t* __restrict__ idx, int64_t n,
                                                          int normalize, float* __restrict__ out,
                                                          float* __restrict__ out_inv) {
    constexpr int LPR = D / 4;        // lanes per row
    constexpr int RPW = 64 / LPR;     // rows per wave
    const int lane = mf_lane();
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t r = wave * RPW + lane / LPR;
    const int c = lane % LPR;
    const bool valid = r < n;
    int64_t row = valid ? idx[r] : 0;
    const bool in_range = row >= 0 && row < n_rows;
    row = in_range ? row : 0;
    f32x4 x = reinterpret_cast<const f32x4*>(table + row * D)[c];
    if (!in_range) x = f32x4{0.f, 0.f, 0.f, 0.f};
    float inv = 1.f;
    if (normalize) {
        float ss = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        ss = mf_group_sum(ss, LPR);
        inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        x = x * inv;
    }
    if (valid) {
        reinterpret_cast<f32x4*>(out + r * D)[c] = x;
        if (out_inv && c == 0) out_inv[r] = inv;
    }
}

extern "C" int mf_gather_rows(const float* table, int64_t n_rows, int d, const int64_t* idx,
                              int64_t n, int normalize, float* out, float* out_inv_norm,
                              mf_stream_t stream) {
    if (!table || !idx || !out || n < 0 || n_rows <= 0) return mf_set_error(MF_EINVAL, "mf_gather_rows: bad argument");
    if (n == 0) return MF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    MF_DISPATCH_D(d, {
        constexpr int RPB = (64 / (D / 4)) * 4;  // rows per 256-thread block
        dim3 grid((unsigned)((n + RPB - 1) / RPB));
        MF_TIMED("gather_rows", s, gather_rows_kernel<D><<<grid, 256, 0, s>>>(table, n_rows, idx, n, normalize, out, out_inv_norm));
    });
    return mf_check_launch("mf_gather_rows");
}

// ------------------------------------------------------- hash / bloom towers ----
template <int D>
__global__ __launch_bounds__(256) void gather_hashed_kernel(const float* __restrict__ table, int64_t num_buckets,
                                                            const int64_t* __restrict__ idx, int64_t n, int num_hashes,
                                                            unsigned long long seed, int normalize,
                                                            float* __restrict__ out, float* __restrict__ out_inv) {
    constexpr int LPR = D / 4;
    constexpr int RPW = 64 / LPR;
    const int lane = mf_lane();
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t r = wave * RPW + lane / LPR;
    const int c = lane % LPR;
    const bool valid = r < n;
    const long long id = valid ? idx[r] : 0;
    f32x4 rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // all rows in flight, added in hash order
        rows[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < num_hashes) rows[j] = reinterpret_cast<const f32x4*>(table + mf_hash_bucket(id, j, seed, num_buckets) * D)[c];
    }
    f32x4 x = rows[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (j < num_hashes) x += rows[j];
    float inv = 1.f;
    if (normalize) {
        float ss = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        ss = mf_group_sum(ss, LPR);
        inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        x = x * inv;
    }
    if (valid) {
        reinterpret_cast<f32x4*>(out + r * D)[c] = x;
        if (out_inv && c == 0) out_inv[r] = inv;
    }
}

extern "C" int mf_gather_hashed(const float* table, int64_t num_buckets, int d, const int64_t* idx, int64_t n,
                                int num_hashes, uint64_t seed, int normalize, float* out, float* out_inv_norm,
                                mf_stream_t stream) {
    if (!table || !idx || !out || n < 0 || num_buckets <= 0) return mf_set_error(MF_EINVAL, "mf_gather_hashed: bad argument");
    if (num_hashes < 1 || num_hashes > 4) return mf_set_error(MF_ENOTSUP, "mf_gather_hashed: num_hashes = %d outside 1..4", num_hashes);
    if (n == 0) return MF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    MF_DISPATCH_D(d, {
        constexpr int RPB = (64 / (D / 4)) * 4;
        dim3 grid((unsigned)((n + RPB - 1) / RPB));
        MF_TIMED("gather_rows", s, gather_hashed_kernel<D><<<grid, 256, 0, s>>>(table, num_buckets, idx, n, num_hashes, seed, normalize, out, out_inv_norm));
    });
    return mf_check_launch("mf_gather_hashed");
}

__global__ __launch_bounds__(256) void hash_buckets_kernel(const int64_t* __restrict__ idx, int64_t n, int num_hashes,
                                                           unsigned long long seed, int64_t num_buckets,
                                                           int64_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * num_hashes) return;
    out[t] = mf_hash_bucket(idx[t / num_hashes], (int)(t % num_hashes), seed, num_buckets);
}

extern "C" int mf_hash_buckets(const int64_t* idx, int64_t n, int num_hashes, uint64_t seed, int64_t num_buckets,
                               int64_t* out_buckets, mf_stream_t stream) {
    if (!idx || !out_buckets || n < 0 || num_buckets <= 0) return mf_set_error(MF_EINVAL, "mf_hash_buckets: bad argument");
    if (num_hashes < 1 || num_hashes > 4) return mf_set_error(MF_ENOTSUP, "mf_hash_buckets: num_hashes = %d outside 1..4", num_hashes);
    if (n == 0) return MF_OK;
    hash_buckets_kernel<<<dim3((unsigned)((n * num_hashes + 255) / 256)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        idx, n, num_hashes, seed, num_buckets, out_buckets);
    return mf_check_launch("mf_hash_buckets");
}

template <int D>
__global__ __launch_bounds__(256) void normalize_backward_kernel(const float* __restrict__ unit, const float* __restrict__ inv_norm,
                                                                 const float* __restrict__ grad, int64_t n,
                                                                 float* __restrict__ graw) {
    constexpr int LPR = D / 4;
    constexpr int RPW = 64 / LPR;
    const int lane = mf_lane();
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t r = wave * RPW + lane / LPR;
    const int c = lane % LPR;
    const int64_t rr = r < n ? r : 0;
    const f32x4 u = reinterpret_cast<const f32x4*>(unit + rr * D)[c];
    const f32x4 g = reinterpret_cast<const f32x4*>(grad + rr * D)[c];
    const float pr = mf_group_sum(g[0] * u[0] + g[1] * u[1] + g[2] * u[2] + g[3] * u[3], LPR);
    if (r < n) reinterpret_cast<f32x4*>(graw + r * D)[c] = (g - u * pr) * inv_norm[r];
}

extern "C" int mf_normalize_backward(const float* out_unit, const float* inv_norm, const float* grad, int64_t n, int d,
                                     float* grad_raw, mf_stream_t stream) {
    if (!out_unit || !inv_norm || !grad || !grad_raw || n < 0) return mf_set_error(MF_EINVAL, "mf_normalize_backward: bad argument");
    if (n == 0) return MF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    MF_DISPATCH_D(d, {
        constexpr int RPB = (64 / (D / 4)) * 4;
        normalize_backward_kernel<D><<<dim3((unsigned)((n + RPB - 1) / RPB)), 256, 0, s>>>(out_unit, inv_norm, grad, n, grad_raw);
    });
    return mf_check_launch("mf_normalize_backward");
}

// ------------------------------------------------------------- chain norms ----
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int64_t n, int d,
                                                         float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(x + r * d);
    float acc = 0.f;
    for (int g = 0; g < d / 8; ++g) {   // same k order as mf_dot_chain
        f32x4 a = p[2 * g], b = p[2 * g + 1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc = __builtin_fmaf(a[t], a[t], acc);
            acc = __builtin_fmaf(b[t], b[t], acc);
        }
    }
    out[r] = acc;
}

extern "C" int mf_row_sqnorm(const float* x, int64_t n, int d, float* out, mf_stream_t stream) {
    if (!x || !out || n < 0 || d <= 0 || d % 8) return mf_set_error(MF_EINVAL, "mf_row_sqnorm: bad argument");
    if (n == 0) return MF_OK;
    row_sqnorm_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, static_cast<hipStream_t>(stream)>>>(x, n, d, out);
    return mf_check_launch("mf_row_sqnorm");
}

// ------------------------------------------------------- raw MFMA score tiles --
template <int D>
__global__ __launch_bounds__(64) void scores_kernel(const float* __restrict__ u, int64_t B,
                                                    const float* __restrict__ v, int64_t N,
                                                    float* __restrict__ out) {
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + c;
    const int64_t j0 = (int64_t)blockIdx.y * 32;
    RowFrag<D> xf, yf;
    mf_load_frag<D>(xf, u, i, i < B);
    mf_load_frag<D>(yf, v, j0 + c, j0 + c < N);
    f32x16 acc = mf_tile_scores<D>(yf, xf);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int64_t j = j0 + mf_acc_row(e, h);
        if (i < B && j < N) out[i * N + j] = acc[e];
    }
}

extern "C" int mf_scores(const float* u, int64_t B, const float* v, int64_t N, int d, float* out,
                         mf_stream_t stream) {
    if (!u || !v || !out || B <= 0 || N <= 0) return mf_set_error(MF_EINVAL, "mf_scores: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    MF_DISPATCH_D(d, {
        dim3 grid((unsigned)((B + 31) / 32), (unsigned)((N + 31) / 32));
        scores_kernel<D><<<grid, 64, 0, s>>>(u, B, v, N, out);
    });
    return mf_check_launch("mf_scores");
}

// ---------------------------------------------------------------- id sort ----
// Stable rank sort: rank(r) = #{r' : (key[r'], r') < (key[r], r)}, an O(n^2)
// all-pairs count spread over the whole chip.  n is a batch (<= a few 10^4 ids)
// and the count is integer-exact and order-free, so the result is deterministic;
// it costs 1/(6 d) of the score contraction of the same batch.
static constexpr int SORT_TILE = 512;

__global__ __launch_bounds__(256) void rank_count_kernel(const int64_t* __restrict__ keys, int n,
                                                         int32_t* __restrict__ rank) {
    __shared__ unsigned long long tile[SORT_TILE];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int base = blockIdx.y * SORT_TILE;
    for (int t = threadIdx.x; t < SORT_TILE; t += 256) {
        const int q = base + t;
        tile[t] = q < n ? (((unsigned long long)keys[q] << 24) | (unsigned)q) : ~0ull;
    }
    __syncthreads();
    if (r >= n) return;
    const unsigned long long mine = ((unsigned long long)keys[r] << 24) | (unsigned)r;
    int cnt = 0;
#pragma unroll 8
    for (int t = 0; t < SORT_TILE; ++t) cnt += tile[t] < mine ? 1 : 0;
    if (cnt) atomicAdd(&rank[r], cnt);
}

__global__ __launch_bounds__(256) void rank_scatter_kernel(const int64_t* __restrict__ keys, int n,
                                                           const int32_t* __restrict__ rank,
                                                           int32_t* __restrict__ perm,
                                                           int64_t* __restrict__ sorted_keys) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int p = rank[r];
    perm[p] = r;
    if (sorted_keys) sorted_keys[p] = keys[r];
}

// Packed variant used by the sparse updates: when bits(id_limit + 1) + bits(n) <= 32 the pair
// (id, position) is ONE 32-bit word -- four keys per ds_read_b128, two VALU instructions per
// comparison -- and out-of-range ids (skipped by the update anyway) share the bucket id_limit.
// Every (row block, tile stripe) writes its own partial count: no atomics, no memset.
static constexpr int SORT_TILE32 = 1024;
static constexpr int SORT_MAX_STRIPES = 32;

__device__ __forceinline__ unsigned sort_pack(long long id, unsigned q, long long id_limit, int posbits) {
    const unsigned b = (id >= 0 && id < id_limit) ? (unsigned)id : (unsigned)id_limit;
    return (b << posbits) | q;
}

__global__ __launch_bounds__(256) void rank_count32_kernel(const int64_t* __restrict__ keys, int n, long long id_limit,
                                                           int posbits, int32_t* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) unsigned tile[SORT_TILE32];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const unsigned mine = r < n ? sort_pack(keys[r], (unsigned)r, id_limit, posbits) : 0u;
    int cnt = 0;
    for (int base = blockIdx.y * SORT_TILE32; base < n; base += gridDim.y * SORT_TILE32) {
        __syncthreads();
        for (int t = threadIdx.x; t < SORT_TILE32; t += 256) {
            const int q = base + t;
            tile[t] = q < n ? sort_pack(keys[q], (unsigned)q, id_limit, posbits) : 0xFFFFFFFFu;
        }
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < SORT_TILE32; t += 4) {
            const uint4 k = *reinterpret_cast<const uint4*>(&tile[t]);   // same address in every lane: LDS broadcast
            cnt += (k.x < mine ? 1 : 0) + (k.y < mine ? 1 : 0) + (k.z < mine ? 1 : 0) + (k.w < mine ? 1 : 0);
        }
    }
    if (r < n) part[(int64_t)blockIdx.y * n + r] = cnt;
}

__global__ __launch_bounds__(256) void rank_scatter32_kernel(const int64_t* __restrict__ keys, int n, int stripes,
                                                             const int32_t* __restrict__ part,
                                                             int32_t* __restrict__ perm,
                                                             int64_t* __restrict__ sorted_keys) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    int p = 0;
    for (int y = 0; y < stripes; ++y) p += part[(int64_t)y * n + r];
    perm[p] = r;
    sorted_keys[p] = keys[r];
}

static int sort_stripes(int64_t n) {
    const int ny = (int)((n + SORT_TILE32 - 1) / SORT_TILE32);
    return ny < SORT_MAX_STRIPES ? ny : SORT_MAX_STRIPES;
}
// posbits if the packed sort applies to (n keys, ids below id_limit), else -1
static int sort_packed_posbits(int64_t n, int64_t id_limit) {
    if (n <= 0 || n > (1 << 20) || id_limit <= 0) return -1;
    int posbits = 0;
    while ((1ll << posbits) < n) ++posbits;
    return (((unsigned long long)(id_limit + 1)) << posbits) <= (1ull << 32) ? posbits : -1;
}

extern "C" size_t mf_sort_ws_bytes(int64_t n) { return mf_align_up((size_t)(n > 0 ? n : 1) * 4, 256); }

extern "C" int mf_sort_keys(const int64_t* keys, int64_t n, int32_t* perm, int64_t* sorted_keys,
                            void* ws, size_t ws_bytes, mf_stream_t stream) {
    if (!keys || !perm || !ws || n < 0) return mf_set_error(MF_EINVAL, "mf_sort_keys: bad argument");
    if (n >= (1 << 24)) return mf_set_error(MF_ENOTSUP, "mf_sort_keys: n = %lld >= 2^24", (long long)n);
    if (ws_bytes < mf_sort_ws_bytes(n)) return mf_set_error(MF_ENOSPC, "mf_sort_keys: workspace too small");
    if (n == 0) return MF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int32_t* rank = static_cast<int32_t*>(ws);
    if (hipMemsetAsync(rank, 0, (size_t)n * 4, s) != hipSuccess) return mf_set_error(MF_ELAUNCH, "mf_sort_keys: memset failed");
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)((n + SORT_TILE - 1) / SORT_TILE));
    rank_count_kernel<<<grid, 256, 0, s>>>(keys, (int)n, rank);
    rank_scatter_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, s>>>(keys, (int)n, rank, perm, sorted_keys);
    return mf_check_launch("mf_sort_keys");
}

// ---- stable grouping by a SMALL key (the owner rank of a routed id: 0 <= key < nkeys <= 64) ---------------------------
// The sharded step's exchange plan orders a batch's ids by owner.  mf_sort_keys does that with an all-pairs rank count:
// 6e8 comparisons at n = 24,576, 100 us of every CU on the plan stream beside the sweeps.  A counting sort needs one pass:
// ONE workgroup keeps the keys as bytes in LDS (n <= 32,768), counts them per key, and places chunk after chunk of 1024
// positions -- inside a wave by ballots per distinct key, across waves by a 16 x 64 count table.  ~15 us on one CU, and the
// group bounds (what the caller used to get from a searchsorted) come with it.
static constexpr int GROUP_MAX_N = 32768, GROUP_MAX_KEYS = 64;
__global__ __launch_bounds__(1024) void group_keys_kernel(const int64_t* __restrict__ keys, int n, int nk, int32_t* __restrict__ perm,
                                                          int64_t* __restrict__ sorted_keys, int64_t* __restrict__ bounds) {
    __shared__ unsigned char kb[GROUP_MAX_N];
    __shared__ int tot[GROUP_MAX_KEYS], start[GROUP_MAX_KEYS + 1], wcnt[16][GROUP_MAX_KEYS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i0 = tid; i0 < n; i0 += 8 * 1024) {                    // eight loads in flight per thread (one at a time: a memory round trip each)
        long long k8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k8[u] = i0 + u * 1024 < n ? keys[i0 + u * 1024] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long k = k8[u] < 0 ? 0 : (k8[u] >= nk ? nk - 1 : k8[u]);      // (precondition: 0 <= key < nkeys)
            if (i0 + u * 1024 < n) kb[i0 + u * 1024] = (unsigned char)k;
        }
    }
    if (tid < GROUP_MAX_KEYS) tot[tid] = 0;
    wcnt[wave][lane] = 0;
    __syncthreads();
    // this lane's rank among the lanes of its wave with the same key, and their number (one ballot per distinct key)
    auto split = [&](bool valid, int k, int& rank, int& count) {
        unsigned long long left = __ballot(valid);
        rank = 0; count = 0;
        while (left) {
            const int lead = __builtin_ctzll(left);
            const int kk = __builtin_amdgcn_readlane(k, lead);
            const unsigned long long m = __ballot(valid && k == kk);
            if (valid && k == kk) { rank = __popcll(m & below); count = __popcll(m); }
            left &= ~m;
        }
    };
    const int nch = (n + 1023) / 1024;
    for (int c = 0; c < nch; ++c) {
        const int i = c * 1024 + tid;
        const bool valid = i < n;
        const int k = valid ? kb[i] : 0;
        int rank, count;
        split(valid, k, rank, count);
        if (valid && rank == 0) atomicAdd(&tot[k], count);
    }
    __syncthreads();
    if (tid == 0) {
        int at = 0;
        for (int k = 0; k < nk; ++k) { start[k] = at; bounds[k] = at; at += tot[k]; }
        start[nk] = at;
        bounds[nk] = at;
    }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int i = c * 1024 + tid;
        const bool valid = i < n;
        const int k = valid ? kb[i] : 0;
        int rank, count;
        split(valid, k, rank, count);
        if (valid && rank == 0) wcnt[wave][k] = count;
        __syncthreads();
        if (valid) {
            int pos = start[k] + rank;
            for (int w = 0; w < wave; ++w) pos += wcnt[w][k];
            perm[pos] = i;
            if (sorted_keys) sorted_keys[pos] = k;
        }
        __syncthreads();
        if (tid < nk) {
            int sum = 0;
            for (int w = 0; w < 16; ++w) sum += wcnt[w][tid];
            start[tid] += sum;
        }
        __syncthreads();
        wcnt[wave][lane] = 0;
        __syncthreads();
    }
}

extern "C" int mf_group_keys(const int64_t* keys, int64_t n, int nkeys, int32_t* perm, int64_t* sorted_keys, int64_t* bounds,
                             mf_stream_t stream) {
    if (!keys || !perm || !bounds || n < 0 || nkeys <= 0) return mf_set_error(MF_EINVAL, "mf_group_keys: bad argument");
    if (n > GROUP_MAX_N || nkeys > GROUP_MAX_KEYS)
        return mf_set_error(MF_ENOTSUP, "mf_group_keys: n = %lld > %d or nkeys = %d > %d (use mf_sort_keys)", (long long)n, GROUP_MAX_N, nkeys,
                            GROUP_MAX_KEYS);
    group_keys_kernel<<<dim3(1), 1024, 0, static_cast<hipStream_t>(stream)>>>(keys, (int)n, nkeys, perm, sorted_keys, bounds);
    return mf_check_launch("mf_group_keys");
}

#include "mf_update.h"


template <int D, bool ADAM, int PHASE>
__global__ __launch_bounds__(256) void update_rows_kernel(float* __restrict__ table,
                                                          float* __restrict__ exp_avg,
                                                          float* __restrict__ exp_avg_sq,
                                                          int64_t n_rows,
                                                          const int32_t* __restrict__ perm,
                                                          const int64_t* __restrict__ skeys, int64_t n,
                                                          const float* __restrict__ grad,
                                                          float* __restrict__ partial,
                                                          int normalized, AdamHyper hp) {
    constexpr int LPR = D / 4;
    constexpr int RPW = 64 / LPR;
    const int lane = mf_lane();
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t p = wave * RPW + lane / LPR;
    const int c = lane % LPR;
    int64_t row = 0;
    bool head = false, owner = false;
    if (p < n) {
        row = skeys[p];
        const bool in_range = row >= 0 && row < n_rows;
        head = in_range && (p == 0 || skeys[p - 1] != row);
        owner = in_range && (head || (p % RUN_CHUNK) == 0);
    }
    const int64_t chunk_end = (p / RUN_CHUNK + 1) * RUN_CHUNK;         // next chunk boundary after p
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bool apply = false;
    if (PHASE == 1) {
        if (owner) {
            int64_t e = p + 1;                                          // end of this chunk inside the run
            while (e < n && e < chunk_end && skeys[e] == row) ++e;
            int64_t q = p;
            for (; q + 4 <= e; q += 4) {                                // four rows in flight, added in order
                f32x4 g[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = reinterpret_cast<const f32x4*>(grad + (int64_t)perm[q + j] * D)[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += g[j];
            }
            for (; q < e; ++q) acc += reinterpret_cast<const f32x4*>(grad + (int64_t)perm[q] * D)[c];
            const bool whole = head && (e >= n || skeys[e] != row);     // the run is this one chunk
            if (whole) apply = true;
            else reinterpret_cast<f32x4*>(partial + p * D)[c] = acc;
        }
    } else {
        const bool multi = head && chunk_end < n && skeys[chunk_end] == row;   // run continues into the next chunk
        if (multi) {
            acc = reinterpret_cast<const f32x4*>(partial + p * D)[c];
            for (int64_t q = chunk_end; q < n && skeys[q] == row; q += RUN_CHUNK)
                acc += reinterpret_cast<const f32x4*>(partial + q * D)[c];
            apply = true;
        }
    }
    apply_row_update<D, ADAM>(apply, row, c, acc, table, exp_avg, exp_avg_sq, normalized, hp);
}


struct UpdateWs {
    int32_t* perm;
    int64_t* skeys;
    void* sort_ws;
    float* partial;
    unsigned long long *gk0, *gk1;
    size_t total;
};
static UpdateWs update_ws(void* ws, int64_t n, int d) {
    MfArena a(ws);
    UpdateWs w;
    w.perm = a.take<int32_t>((size_t)n);
    w.skeys = a.take<int64_t>((size_t)n);
    const size_t packed = (size_t)n * 4 * sort_stripes(n);       // partial counts of the packed sort
    w.sort_ws = a.take<char>(packed > mf_sort_ws_bytes(n) ? packed : mf_sort_ws_bytes(n));
    w.partial = a.take<float>((size_t)n * d);
    w.gk0 = a.take<unsigned long long>((size_t)n);
    w.gk1 = a.take<unsigned long long>((size_t)n);
    w.total = a.used();
    return w;
}

extern "C" size_t mf_update_ws_bytes(int64_t n, int d) { return update_ws(nullptr, n > 0 ? n : 1, d).total; }

template <bool ADAM>
static int update_common(float* table, float* m, float* v, int64_t n_rows, int d, const int64_t* idx,
                         int64_t n, const float* grad, int normalized, AdamHyper hp, void* ws,
                         size_t ws_bytes, mf_stream_t stream, const char* what) {
    if (!table || !idx || !grad || !ws || n < 0 || n_rows <= 0 || (ADAM && (!m || !v)))
        return mf_set_error(MF_EINVAL, "%s: bad argument", what);
    if (n_rows >= (1ll << 39)) return mf_set_error(MF_ENOTSUP, "%s: table too large", what);
    if (!mf_width_ok(d)) return mf_set_error(MF_EINVAL, "%s: embedding width %d not in {32,64,128,256}", what, d);
    if (ws_bytes < mf_update_ws_bytes(n, d)) return mf_set_error(MF_ENOSPC, "%s: workspace too small", what);
    if (n == 0) return MF_OK;
    UpdateWs w = update_ws(ws, n, d);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n <= FUSED_MAX_N) {
        const int bits = fused_bucket_bits(n);      // buckets (= workgroups): ~32 ids each, at most two per CU
        FusedUpdateParams fp{table, m, v, (long long)n_rows, reinterpret_cast<const long long*>(idx), (int)n, bits, grad,
                             w.partial, w.gk0, w.gk1, normalized, hp};
        MF_DISPATCH_D(d, {
            auto fn = update_fused_kernel<D, ADAM>;
            static bool attr_set = false;           // (per instantiation)
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_CAP * 8);
                attr_set = true;
            }
            MF_TIMED("update_rows", s, (fn<<<dim3(1u << bits), FUSED_THREADS, FUSED_CAP * 8, s>>>(fp)));
        });
        return mf_check_launch(what);
    }
    const int posbits = sort_packed_posbits(n, n_rows);
    if (posbits >= 0) {
        const int stripes = sort_stripes(n);
        int32_t* part = reinterpret_cast<int32_t*>(w.sort_ws);
        rank_count32_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)stripes), 256, 0, s>>>(idx, (int)n, n_rows, posbits, part);
        rank_scatter32_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, s>>>(idx, (int)n, stripes, part, w.perm, w.skeys);
    } else {
        int rc = mf_sort_keys(idx, n, w.perm, w.skeys, w.sort_ws, mf_sort_ws_bytes(n), stream);
        if (rc) return rc;
    }
    MF_DISPATCH_D(d, {
        constexpr int RPB = (64 / (D / 4)) * 4;
        dim3 grid((unsigned)((n + RPB - 1) / RPB));
        MF_TIMED("update_rows", s, {
            (update_rows_kernel<D, ADAM, 1><<<grid, 256, 0, s>>>(table, m, v, n_rows, w.perm, w.skeys, n, grad, w.partial, normalized, hp));
            (update_rows_kernel<D, ADAM, 2><<<grid, 256, 0, s>>>(table, m, v, n_rows, w.perm, w.skeys, n, grad, w.partial, normalized, hp));
        });
    });
    return mf_check_launch(what);
}

extern "C" int mf_update_sgd(float* table, int64_t n_rows, int d, const int64_t* idx, int64_t n,
                             const float* grad, int normalized, float lr, float weight_decay,
                             void* ws, size_t ws_bytes, mf_stream_t stream) {
    AdamHyper hp{lr, 0.f, 0.f, 0.f, weight_decay, 1, nullptr, 0.0, 0.0};
    return update_common<false>(table, nullptr, nullptr, n_rows, d, idx, n, grad, normalized, hp, ws,
                                ws_bytes, stream, "mf_update_sgd");
}

extern "C" int mf_update_adam(float* table, float* exp_avg, float* exp_avg_sq, int64_t n_rows, int d,
                              const int64_t* idx, int64_t n, const float* grad, int normalized,
                              int64_t step, const int64_t* step_dev, float lr, float beta1, float beta2, float eps,
                              float weight_decay, void* ws, size_t ws_bytes, mf_stream_t stream) {
    if (!step_dev && step < 1) return mf_set_error(MF_EINVAL, "mf_update_adam: step must be >= 1");
    AdamHyper hp{lr, beta1, beta2, eps, weight_decay, (long long)step, reinterpret_cast<const long long*>(step_dev),
                 log((double)beta1), log((double)beta2)};
    return update_common<true>(table, exp_avg, exp_avg_sq, n_rows, d, idx, n, grad, normalized, hp, ws,
                               ws_bytes, stream, "mf_update_adam");
}

template <bool ADAM>
static int update_pair_common(int d, float* ta, float* ma, float* va, int64_t rows_a, const int64_t* idx_a, int64_t n_a, const float* grad_a,
                              int norm_a, void* ws_a, size_t ws_a_bytes, float* tb, float* mb, float* vb, int64_t rows_b,
                              const int64_t* idx_b, int64_t n_b, const float* grad_b, int norm_b, void* ws_b, size_t ws_b_bytes,
                              AdamHyper hp, mf_stream_t stream) {
    if (!ta || !tb || !idx_a || !idx_b || !grad_a || !grad_b || !ws_a || !ws_b || rows_a <= 0 || rows_b <= 0 ||
        (ADAM && (!ma || !va || !mb || !vb)))
        return mf_set_error(MF_EINVAL, "mf_update_pair: bad argument");
    if (!mf_width_ok(d)) return mf_set_error(MF_EINVAL, "mf_update_pair: embedding width %d not in {32,64,128,256}", d);
    if (rows_a >= (1ll << 39) || rows_b >= (1ll << 39)) return mf_set_error(MF_ENOTSUP, "mf_update_pair: table too large");
    if (n_a <= 0 || n_b <= 0 || n_a > FUSED_MAX_N || n_b > FUSED_MAX_N)
        return mf_set_error(MF_ENOTSUP, "mf_update_pair: list lengths outside 1..%d (update the tables one by one)", FUSED_MAX_N);
    if (ws_a_bytes < mf_update_ws_bytes(n_a, d) || ws_b_bytes < mf_update_ws_bytes(n_b, d))
        return mf_set_error(MF_ENOSPC, "mf_update_pair: workspace too small");
    UpdateWs wa = update_ws(ws_a, n_a, d), wb = update_ws(ws_b, n_b, d);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bits_a = fused_bucket_bits(n_a), bits_b = fused_bucket_bits(n_b);
    FusedUpdateParams fa{ta, ma, va, (long long)rows_a, reinterpret_cast<const long long*>(idx_a), (int)n_a, bits_a, grad_a,
                         wa.partial, wa.gk0, wa.gk1, norm_a, hp};
    FusedUpdateParams fb{tb, mb, vb, (long long)rows_b, reinterpret_cast<const long long*>(idx_b), (int)n_b, bits_b, grad_b,
                         wb.partial, wb.gk0, wb.gk1, norm_b, hp};
    MF_DISPATCH_D(d, {
        auto fn = update_fused_pair_kernel<D, ADAM>;
        static bool attr_set = false;           // (per instantiation)
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_CAP * 8);
            attr_set = true;
        }
        MF_TIMED("update_rows", s, (fn<<<dim3((1u << bits_a) + (1u << bits_b)), FUSED_THREADS, FUSED_CAP * 8, s>>>(fa, fb, 1 << bits_a)));
    });
    return mf_check_launch("mf_update_pair");
}

extern "C" int mf_update_pair(int adam, int d, float* table_a, float* exp_avg_a, float* exp_avg_sq_a, int64_t n_rows_a, const int64_t* idx_a,
                              int64_t n_a, const float* grad_a, int normalized_a, void* ws_a, size_t ws_a_bytes, float* table_b,
                              float* exp_avg_b, float* exp_avg_sq_b, int64_t n_rows_b, const int64_t* idx_b, int64_t n_b, const float* grad_b,
                              int normalized_b, void* ws_b, size_t ws_b_bytes, int64_t step, const int64_t* step_dev, float lr, float beta1,
                              float beta2, float eps, float weight_decay, mf_stream_t stream) {
    if (adam) {
        if (!step_dev && step < 1) return mf_set_error(MF_EINVAL, "mf_update_pair: step must be >= 1");
        AdamHyper hp{lr, beta1, beta2, eps, weight_decay, (long long)step, reinterpret_cast<const long long*>(step_dev),
                     log((double)beta1), log((double)beta2)};
        return update_pair_common<true>(d, table_a, exp_avg_a, exp_avg_sq_a, n_rows_a, idx_a, n_a, grad_a, normalized_a, ws_a, ws_a_bytes, table_b,
                                        exp_avg_b, exp_avg_sq_b, n_rows_b, idx_b, n_b, grad_b, normalized_b, ws_b, ws_b_bytes, hp, stream);
    }
    AdamHyper hp{lr, 0.f, 0.f, 0.f, weight_decay, 1, nullptr, 0.0, 0.0};
    return update_pair_common<false>(d, table_a, nullptr, nullptr, n_rows_a, idx_a, n_a, grad_a, normalized_a, ws_a, ws_a_bytes, table_b, nullptr,
                                     nullptr, n_rows_b, idx_b, n_b, grad_b, normalized_b, ws_b, ws_b_bytes, hp, stream);
}

// ---- lab: the DPP / permlane lane exchanges of mf_common.h against the `__shfl_xor` they replace ----------------------
// (tests/test_gpu_parity.py: every butterfly sum, the 64-bit wave maximum and every single exchange, bit for bit)
__global__ __launch_bounds__(64) void lane_ops_probe_kernel(const uint32_t* __restrict__ in, unsigned* __restrict__ bad) {
    const int lane = threadIdx.x;
    const uint32_t a = in[(size_t)blockIdx.x * 128 + lane], b = in[(size_t)blockIdx.x * 128 + 64 + lane];
    const float x = __builtin_bit_cast(float, a);
    auto ref_sum = [&](int width) {
        float r = x;
        for (int m = width >> 1; m >= 1; m >>= 1) r += __shfl_xor(r, m, 64);
        return __builtin_bit_cast(uint32_t, r);
    };
    auto note = [&](int op, bool wrong) { if (wrong) atomicAdd(bad + op, 1u); };      // bad[16]: mismatching lanes per operation
    note(0, ref_sum(64) != __builtin_bit_cast(uint32_t, mf_butterfly_sum<64>(x)));
    note(1, ref_sum(32) != __builtin_bit_cast(uint32_t, mf_butterfly_sum<32>(x)));
    note(2, ref_sum(16) != __builtin_bit_cast(uint32_t, mf_butterfly_sum<16>(x)));
    note(3, ref_sum(8) != __builtin_bit_cast(uint32_t, mf_butterfly_sum<8>(x)));
    note(4, ref_sum(4) != __builtin_bit_cast(uint32_t, mf_butterfly_sum<4>(x)));
    note(5, ref_sum(2) != __builtin_bit_cast(uint32_t, mf_butterfly_sum<2>(x)));
    note(6, (uint32_t)__shfl_xor((int)a, 1, 64) != mf_xor_lane_u32<1>(a));
    note(7, (uint32_t)__shfl_xor((int)a, 2, 64) != mf_xor_lane_u32<2>(a));
    note(8, (uint32_t)__shfl_xor((int)a, 4, 64) != mf_xor_lane_u32<4>(a));
    note(9, (uint32_t)__shfl_xor((int)a, 8, 64) != mf_xor_lane_u32<8>(a));
    note(10, (uint32_t)__shfl_xor((int)a, 16, 64) != mf_xor_lane_u32<16>(a));
    note(11, (uint32_t)__shfl_xor((int)a, 32, 64) != mf_xor_lane_u32<32>(a));
    {
        const unsigned long long k = ((unsigned long long)a << 32) | b;
        unsigned long long r = k;
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(r >> 32), m, 64) << 32) |
                                         (uint32_t)__shfl_xor((int)(uint32_t)r, m, 64);
            r = o > r ? o : r;
        }
        note(12, r != mf_wave_max_u64(k));
        int s = (int)b;
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        note(13, s != mf_wave_sum_int((int)b));
    }
}

extern "C" int mf_probe_lane_ops(const uint32_t* in, int64_t waves, unsigned* mismatches, mf_stream_t stream) {
    if (!in || !mismatches || waves <= 0 || waves > (1 << 20)) return mf_set_error(MF_EINVAL, "mf_probe_lane_ops: bad argument");
    lane_ops_probe_kernel<<<dim3((unsigned)waves), 64, 0, static_cast<hipStream_t>(stream)>>>(in, mismatches);
    return mf_check_launch("mf_probe_lane_ops");
}
