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

// ---------------------------------------------------------- sparse updates ----
struct AdamHyper {
    float lr, beta1, beta2, eps, wd;
    long long step;              // global step (1-based) ...
    const long long* step_dev;   // ... or, when non-NULL, where to read it on the device (hipGraph replays: a captured
                                 // launch freezes its by-value arguments, a device counter keeps counting)
    double ln_beta1, ln_beta2;   // log(beta), evaluated once on the host
};
// bias corrections 1 - beta^step = -expm1(step * ln beta), evaluated on the device in both modes (so an eager step and a
// replayed one agree bit for bit); double precision, but one expm1 each instead of a pow
__device__ __forceinline__ void adam_bias(const AdamHyper& hp, float& bc1, float& bc2) {
    const double st = (double)(hp.step_dev ? *hp.step_dev : hp.step);
    bc1 = (float)(-expm1(st * hp.ln_beta1));
    bc2 = (float)(-expm1(st * hp.ln_beta2));
}

// One d/4-lane group per sorted position.  Runs of equal ids are summed in two
// deterministic levels so that a very popular row (Zipf: hundreds of duplicates in
// one batch) does not serialise on one group.  A CHUNK starts at a run's first position
// and at every position that is a multiple of 32; its owner sums the chunk's <= 32 gradient
// rows in sorted (= batch) order.  A run that is one chunk is applied at once; otherwise the
// chunk sums are parked and the run's first position adds them up, in order, in a second
// launch.  Chunk boundaries depend on sorted positions only: deterministic.
// The arithmetic of one row update on values already in registers (lane c of the row's D/4-lane group holds
// floats 4c .. 4c+3 of the row, its moments and its summed gradient).
template <int D, bool ADAM>
__device__ __forceinline__ void row_update_math(f32x4& w, f32x4& m, f32x4& v, f32x4 acc, int normalized,
                                                const AdamHyper& hp, float bc1, float bc2) {
    constexpr int LPR = D / 4;
    f32x4 g = acc;
    if (normalized) {   // grad is w.r.t. w / max(||w||, 1e-12): apply the Jacobian
        float ss = mf_group_sum(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3], LPR);
        const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        const f32x4 uh = w * inv;
        float pr = mf_group_sum(acc[0] * uh[0] + acc[1] * uh[1] + acc[2] * uh[2] + acc[3] * uh[3], LPR);
        g = (acc - uh * pr) * inv;
    }
    if (!ADAM) {
        w = w - hp.lr * (g + hp.wd * w);
    } else {
        w = w * (1.f - hp.lr * hp.wd);
        m = m * hp.beta1 + (1.f - hp.beta1) * g;
        v = v * hp.beta2 + (1.f - hp.beta2) * g * g;
        f32x4 den;
#pragma unroll
        for (int t = 0; t < 4; ++t) den[t] = sqrtf(v[t] / bc2) + hp.eps;
        w = w - (hp.lr / bc1) * m / den;
    }
}

template <int D, bool ADAM>
__device__ __forceinline__ void apply_row_update(bool active, int64_t row, int c, f32x4 acc,
                                                 float* __restrict__ table, float* __restrict__ exp_avg,
                                                 float* __restrict__ exp_avg_sq, int normalized,
                                                 const AdamHyper& hp) {
    // a group (the D/4 lanes of one row) is active or not as a whole: idle groups leave before touching memory
    // (they used to read row 0 -- half a million lanes on one 512-byte line)
    if (!active) return;
    f32x4 w = reinterpret_cast<const f32x4*>(table + row * D)[c];
    f32x4 m = {0.f, 0.f, 0.f, 0.f}, v = m;
    float bc1 = 1.f, bc2 = 1.f;
    if (ADAM) {
        m = reinterpret_cast<const f32x4*>(exp_avg + row * D)[c];
        v = reinterpret_cast<const f32x4*>(exp_avg_sq + row * D)[c];
        adam_bias(hp, bc1, bc2);
    }
    row_update_math<D, ADAM>(w, m, v, acc, normalized, hp, bc1, bc2);
    if (ADAM) {
        reinterpret_cast<f32x4*>(exp_avg + row * D)[c] = m;
        reinterpret_cast<f32x4*>(exp_avg_sq + row * D)[c] = v;
    }
    reinterpret_cast<f32x4*>(table + row * D)[c] = w;
}

static constexpr int RUN_CHUNK = 32;

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


// ------------------------------------------------- one-launch sparse update ----
// Batch-sized id lists (n <= FUSED_MAX_N): grouping equal ids and applying the rows in ONE launch, one
// workgroup per hash bucket of the id space.  Every workgroup scans all n ids (n x 8 bytes from L2), keeps
// those of its bucket as 64-bit keys (id << 24 | batch position) in LDS, sorts them there (all-pairs rank for a
// short list, bitonic for a long one: the cost does not depend on how many duplicates a popular row has), and
// then works through its runs of equal ids like the two-phase kernel above -- chunks of <= 32 gradient rows
// summed in batch order, a run that is one chunk applied at once, longer runs through parked chunk sums --
// except that the two phases sit on either side of a workgroup barrier instead of a launch boundary.  All
// occurrences of an id are in one bucket, so no two workgroups touch the same table row.  Nothing depends on
// arrival order: the result is a function of the batch alone.  A bucket that does not fit the LDS list
// (> FUSED_CAP ids in one of up to 512 buckets: one row holding > 12 % of a 65,536-id batch) is sorted in global
// memory by the same workgroup (slow, O(m^2 / 1024), but correct).
static constexpr int FUSED_MAX_N = 65536;
static constexpr int FUSED_CAP = 8192;          // keys per bucket in LDS (64 KiB)
static constexpr int FUSED_RANK_MAX = 512;      // all-pairs rank sort up to here (the sorted copy goes to the list's upper half)
static constexpr int FUSED_THREADS = 1024;
static constexpr int FUSED_MAX_BITS = 9;
static constexpr int FUSED_SCAN_UNROLL = 8;
#ifndef FUSED_NF_ADAM
#define FUSED_NF_ADAM 12
#endif
static constexpr unsigned long long FUSED_PAD = ~0ull;
static constexpr unsigned FUSED_POS_MASK = 0xFFFFFFu;

__device__ __forceinline__ unsigned fused_bucket(long long id, int bucket_bits) {
    return bucket_bits ? (((unsigned)id * 0x9E3779B1u) >> (32 - bucket_bits)) : 0u;
}

struct FusedUpdateParams {
    float *table, *exp_avg, *exp_avg_sq;
    long long n_rows;
    const long long* idx;
    int n, bucket_bits;
    const float* grad;
    float* partial;                     // [n][d] parked chunk sums (by batch position of the chunk's first row)
    unsigned long long *gk0, *gk1;      // [n] each: the global-memory lists of an overflowing bucket
    int normalized;
    AdamHyper hp;
};

// append this thread's key (if `mine`) to the list; the order of the list is irrelevant (it is sorted next)
template <class List>
__device__ __forceinline__ void fused_append(bool mine, unsigned long long key, int* counter, List list, int cap) {
    const unsigned long long bal = __ballot(mine);
    if (bal) {
        const int lane = mf_lane();
        int at = 0;
        if (lane == 0) at = atomicAdd(counter, __popcll(bal));
        at = __shfl(at, 0) + __popcll(bal & ((1ull << lane) - 1ull));
        if (mine && at < cap) list[at] = key;
    }
}

// The runs of equal ids of a sorted key list K[0 .. m): sorted position k goes to wave (k + k / 32) % NWAVE --
// consecutive positions AND consecutive chunk starts (multiples of 32: the chunks of one popular row) land in
// different waves -- GPW owners at a time per wave, one per D/4-lane group.
// pass 0: chunk sums (a run that is one chunk is applied);  pass 1: runs of several chunks
template <int D, bool ADAM, class Keys>
__device__ __forceinline__ void fused_runs(Keys K, int m, const FusedUpdateParams& p, float bc1, float bc2) {
    constexpr int LPR = D / 4, GPW = 64 / LPR, NWAVE = FUSED_THREADS / 64;
    // gradient rows in flight per lane group.  1024 threads leave 128 registers per lane; with Adam the row and its two
    // moments are in flight beside the gradients (12 registers): sixteen rows spilled 20 registers, twelve fit
    constexpr int NF = ADAM ? FUSED_NF_ADAM : 16;
    const int lane = mf_lane(), wave = threadIdx.x >> 6;
    const int grp = lane / LPR, c = lane % LPR;
    for (int pass = 0; pass < 2; ++pass) {
        for (int k0 = 0; k0 < m; k0 += FUSED_THREADS) {
            auto pos_of = [&](int ln) { return k0 + ln * NWAVE + ((wave - (k0 >> 5) - (ln >> 1)) & (NWAVE - 1)); };
            static_assert(NWAVE == 16 && RUN_CHUNK == 32, "position <-> (wave, lane) map");
            const int k = pos_of(lane);
            bool todo = false;
            if (k < m) {
                const unsigned long long row = K[k] >> 24;
                const bool head = k == 0 || (K[k - 1] >> 24) != row;
                const int chunk_end = (k / RUN_CHUNK + 1) * RUN_CHUNK;
                if (pass == 0) todo = head || (k % RUN_CHUNK) == 0;
                else todo = head && chunk_end < m && (K[chunk_end] >> 24) == row;
            }
            unsigned long long work = __ballot(todo);
            while (work) {
                int src = -1;
#pragma unroll
                for (int g = 0; g < GPW; ++g) {
                    if (work) {
                        const int b = __ffsll((long long)work) - 1;
                        work &= work - 1;
                        if (g == grp) src = b;
                    }
                }
                if (src < 0) continue;                                  // (a whole group: the row's lanes stay together)
                const int kk = pos_of(src);
                const unsigned long long key = K[kk];
                const unsigned long long row = key >> 24;
                const bool head = kk == 0 || (K[kk - 1] >> 24) != row;
                const int chunk_end = min(m, (kk / RUN_CHUNK + 1) * RUN_CHUNK);
                // the row and its moments are asked for before the gradient rows: one memory round trip, not two
                const bool will_apply = pass == 1 || head;
                f32x4 w = {0.f, 0.f, 0.f, 0.f}, mm = w, vv = w;
                if (will_apply) {
                    w = reinterpret_cast<const f32x4*>(p.table + (int64_t)row * D)[c];
                    if (ADAM) {
                        mm = reinterpret_cast<const f32x4*>(p.exp_avg + (int64_t)row * D)[c];
                        vv = reinterpret_cast<const f32x4*>(p.exp_avg_sq + (int64_t)row * D)[c];
                    }
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                bool apply = false;
                float* park = p.partial + (int64_t)((unsigned)key & FUSED_POS_MASK) * D;
                if (pass == 0) {
                    int e = kk + 1;
                    while (e < chunk_end && (K[e] >> 24) == row) ++e;
                    int q = kk;
                    for (; q + NF <= e; q += NF) {                      // NF rows in flight, added in order
                        f32x4 g16[NF];
#pragma unroll
                        for (int j = 0; j < NF; ++j)
                            g16[j] = reinterpret_cast<const f32x4*>(p.grad + (int64_t)((unsigned)K[q + j] & FUSED_POS_MASK) * D)[c];
#pragma unroll
                        for (int j = 0; j < NF; ++j) acc += g16[j];
                    }
                    for (; q + 4 <= e; q += 4) {
                        f32x4 g4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            g4[j] = reinterpret_cast<const f32x4*>(p.grad + (int64_t)((unsigned)K[q + j] & FUSED_POS_MASK) * D)[c];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc += g4[j];
                    }
                    for (; q < e; ++q) acc += reinterpret_cast<const f32x4*>(p.grad + (int64_t)((unsigned)K[q] & FUSED_POS_MASK) * D)[c];
                    apply = head && (e >= m || (K[e] >> 24) != row);    // the run is this one chunk
                    if (!apply) reinterpret_cast<f32x4*>(park)[c] = acc;
                } else {
                    acc = reinterpret_cast<const f32x4*>(park)[c];
                    int q = chunk_end;
                    while (q < m && (K[q] >> 24) == row) {              // the run's further chunks: NF parked sums in flight
                        f32x4 g16[NF];
                        int cnt = 0;
#pragma unroll
                        for (int j = 0; j < NF; ++j) {
                            const int qq = q + j * RUN_CHUNK;
                            const bool in_run = qq < m && (K[qq < m ? qq : q] >> 24) == row;
                            g16[j] = reinterpret_cast<const f32x4*>(p.partial + (int64_t)((unsigned)K[in_run ? qq : q] & FUSED_POS_MASK) * D)[c];
                            cnt += in_run ? 1 : 0;      // (a run is contiguous: the chunks in it are the first cnt)
                        }
#pragma unroll
                        for (int j = 0; j < NF; ++j)
                            if (j < cnt) acc += g16[j];
                        q += NF * RUN_CHUNK;
                        if (cnt < NF) break;
                    }
                    apply = true;
                }
                if (apply) {
                    row_update_math<D, ADAM>(w, mm, vv, acc, p.normalized, p.hp, bc1, bc2);
                    if (ADAM) {
                        reinterpret_cast<f32x4*>(p.exp_avg + (int64_t)row * D)[c] = mm;
                        reinterpret_cast<f32x4*>(p.exp_avg_sq + (int64_t)row * D)[c] = vv;
                    }
                    reinterpret_cast<f32x4*>(p.table + (int64_t)row * D)[c] = w;
                }
            }
        }
        __syncthreads();                // parked chunk sums are visible to the workgroup's other waves
    }
}

template <int D, bool ADAM>
__global__ __launch_bounds__(FUSED_THREADS) void update_fused_kernel(FusedUpdateParams p) {
    extern __shared__ __attribute__((aligned(16))) char fused_smem[];
    unsigned long long* lk = reinterpret_cast<unsigned long long*>(fused_smem);
    __shared__ int s_count, s_lower;
    __shared__ float s_bias[2];
    const int tid = threadIdx.x, lane = mf_lane();
    const unsigned myb = blockIdx.x;
    if (tid == 0) { s_count = 0; s_lower = 0; }
    __syncthreads();
    // ---- scan: collect this bucket's (id, position) keys; FUSED_SCAN_UNROLL loads in flight per thread
    for (int base = 0; base < p.n; base += FUSED_THREADS * FUSED_SCAN_UNROLL) {
        long long id[FUSED_SCAN_UNROLL];
#pragma unroll
        for (int u = 0; u < FUSED_SCAN_UNROLL; ++u) {
            const int q = base + u * FUSED_THREADS + tid;
            id[u] = q < p.n ? p.idx[q] : -1;
        }
        if (ADAM && base == 0 && tid == FUSED_THREADS - 64) {
            // one thread evaluates the two double-precision powers for the workgroup, under the latency of the loads
            float b1, b2;
            adam_bias(p.hp, b1, b2);
            s_bias[0] = b1; s_bias[1] = b2;
        }
#pragma unroll
        for (int u = 0; u < FUSED_SCAN_UNROLL; ++u) {
            const int q = base + u * FUSED_THREADS + tid;
            const bool mine = id[u] >= 0 && id[u] < p.n_rows && fused_bucket(id[u], p.bucket_bits) == myb;
            fused_append(mine, ((unsigned long long)id[u] << 24) | (unsigned)q, &s_count, lk, FUSED_CAP);
        }
    }
    __syncthreads();
    const int m = s_count;
    const float bc1 = ADAM ? s_bias[0] : 1.f, bc2 = ADAM ? s_bias[1] : 1.f;
    __syncthreads();                    // (s_count is reused below)
    if (m == 0) return;
    if (m <= FUSED_RANK_MAX) {
        // short list: every key counts the keys below it (LDS broadcast reads) and drops into its place
        unsigned long long* sorted = lk + FUSED_CAP / 2;
        if (tid < m) {
            const unsigned long long mine = lk[tid];
            int rank = 0;
            for (int j = 0; j < m; ++j) rank += lk[j] < mine ? 1 : 0;
            sorted[rank] = mine;
        }
        __syncthreads();
        fused_runs<D, ADAM>(sorted, m, p, bc1, bc2);
    } else if (m <= FUSED_CAP) {
        int P = 1024;
        while (P < m) P <<= 1;
        for (int t = m + tid; t < P; t += FUSED_THREADS) lk[t] = FUSED_PAD;
        __syncthreads();
        // compare-exchange t touches elements 2 (t & ~(j-1)) | (t & (j-1)) and that | j: for j <= 64 the 64 consecutive
        // t of a wave stay inside the wave's own 128 elements, so those stages need no workgroup barrier -- LDS
        // executes one wave's instructions in order; the fence only keeps the compiler from moving them
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += FUSED_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const unsigned long long a = lk[i], b = lk[l];
                    if ((a > b) == ((i & k) == 0)) { lk[i] = b; lk[l] = a; }
                }
                // a barrier after a cross-wave stage, and after a phase's last stage when the next phase opens cross-wave
                if (j > 64 || (j == 1 && k >= 128)) __syncthreads();
                else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
        }
        fused_runs<D, ADAM>(lk, m, p, bc1, bc2);
    } else {
        // the bucket does not fit: its segment of the global lists starts after every key of a lower bucket
        int low = 0;
        for (int q = tid; q < p.n; q += FUSED_THREADS) {
            const long long id = p.idx[q];
            low += (id >= 0 && id < p.n_rows && fused_bucket(id, p.bucket_bits) < myb) ? 1 : 0;
        }
        for (int w = 32; w >= 1; w >>= 1) low += __shfl_xor(low, w);
        if (lane == 0 && low) atomicAdd(&s_lower, low);
        if (tid == 0) s_count = 0;
        __syncthreads();
        unsigned long long* g0 = p.gk0 + s_lower;
        unsigned long long* g1 = p.gk1 + s_lower;
        for (int base = 0; base < p.n; base += FUSED_THREADS) {
            const int q = base + tid;
            const long long id = q < p.n ? p.idx[q] : -1;
            const bool mine = id >= 0 && id < p.n_rows && fused_bucket(id, p.bucket_bits) == myb;
            fused_append(mine, ((unsigned long long)id << 24) | (unsigned)q, &s_count, g0, p.n);
        }
        __syncthreads();                // (workgroup-scope release / acquire: the waves of a workgroup share the CU's L1)
        for (int i = tid; i < m; i += FUSED_THREADS) {
            const unsigned long long mine = g0[i];
            int rank = 0;
            for (int j = 0; j < m; ++j) rank += g0[j] < mine ? 1 : 0;
            g1[rank] = mine;
        }
        __syncthreads();
        fused_runs<D, ADAM>(static_cast<const unsigned long long*>(g1), m, p, bc1, bc2);
    }
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
        int bits = 0;                   // buckets (= workgroups): ~32 ids each, at most two per CU
        while (bits < FUSED_MAX_BITS && (32ll << bits) < n) ++bits;
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
