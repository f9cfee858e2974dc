// mf_update.h -- the arithmetic and the one-workgroup body of the sparse row updates, shared by mf_embed.hip (one workgroup
// per id bucket: update_fused_kernel) and mf_step_small.hip (the whole reference-default step in one workgroup, which runs
// the same body bucket after bucket: bit-identical tables by construction).
#pragma once

#include "mf_common.h"

#ifdef __HIPCC__

static constexpr int RUN_CHUNK = 32;

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
#ifndef MF_FUSED_RANK_MAX
#define MF_FUSED_RANK_MAX 512
#endif
#ifndef MF_FUSED_MAX_BITS
#define MF_FUSED_MAX_BITS 8     // 256 buckets = one workgroup per CU.  Every workgroup scans ALL n ids (L2): with 512 buckets that scan was
#endif                          // 64 MB of L2 reads per 16,384-id update -- 41.0 -> 35.7 us (Zipf ids), 25.2 -> 19.6 (uniform); 128: 38.7 / 25.4

static constexpr int FUSED_RANK_MAX = MF_FUSED_RANK_MAX; // all-pairs rank sort up to here (the sorted copy goes to the list's upper half)
static constexpr int FUSED_THREADS = 1024;
static constexpr int FUSED_MAX_BITS = MF_FUSED_MAX_BITS;
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
// `params(k)`: the table / gradient / hyper-parameter set of sorted position k, `params.row(key)`: the row a key addresses
// (FusedOneTable for update_fused_kernel; the one-launch step lays the lists of its two tables end to end and updates
// both in one pass: FusedTwoTables)
template <int D, bool ADAM, int NFLIGHT, class Keys, class Params>
__device__ __forceinline__ void fused_runs_of(Keys K, int m, Params params, float bc1, float bc2) {
    constexpr int LPR = D / 4, GPW = 64 / LPR, NWAVE = FUSED_THREADS / 64;
    // gradient rows in flight per lane group.  1024 threads leave 128 registers per lane; with Adam the row and its two
    // moments are in flight beside the gradients (12 registers): sixteen rows spilled 20 registers, twelve fit
    constexpr int NF = NFLIGHT;      // (rows are added in order whatever NF is: the sums do not depend on it)
    const int lane = mf_lane(), wave = threadIdx.x >> 6;
    const int grp = lane / LPR, c = lane % LPR;
    int parked = 0;                     // this thread left a chunk sum for the second pass
    for (int pass = 0; pass < 2; ++pass) {
        for (int k0 = 0; k0 < m; k0 += FUSED_THREADS) {
            auto pos_of = [&](int ln) { return k0 + ln * NWAVE + ((wave - (k0 >> 5) - (ln >> 1)) & (NWAVE - 1)); };
            static_assert(NWAVE == 16 && RUN_CHUNK == 32, "position <-> (wave, lane) map");
            const int k = pos_of(lane);
            bool todo = false;
            if (k < m && K[k] != FUSED_PAD) {                           // (padding: fused_update_small's bucket boundaries)
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
                const FusedUpdateParams& p = params(kk);
                const unsigned long long key = K[kk];
                const unsigned long long row = key >> 24;              // what a run is a run of (fused_update_small: the table's bit on top)
                const int64_t ri = params.row(key);                    // the row to address
                const bool head = kk == 0 || (K[kk - 1] >> 24) != row;
                const int chunk_end = min(m, (kk / RUN_CHUNK + 1) * RUN_CHUNK);
                // the row and its moments are asked for before the gradient rows: one memory round trip, not two
                const bool will_apply = pass == 1 || head;
                f32x4 w = {0.f, 0.f, 0.f, 0.f}, mm = w, vv = w;
                if (will_apply) {
                    w = reinterpret_cast<const f32x4*>(p.table + ri * D)[c];
                    if (ADAM) {
                        mm = reinterpret_cast<const f32x4*>(p.exp_avg + ri * D)[c];
                        vv = reinterpret_cast<const f32x4*>(p.exp_avg_sq + ri * D)[c];
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
                    if (!apply) { reinterpret_cast<f32x4*>(park)[c] = acc; parked = 1; }
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
                        reinterpret_cast<f32x4*>(p.exp_avg + ri * D)[c] = mm;
                        reinterpret_cast<f32x4*>(p.exp_avg_sq + ri * D)[c] = vv;
                    }
                    reinterpret_cast<f32x4*>(p.table + ri * D)[c] = w;
                }
            }
        }
        // parked chunk sums are visible to the workgroup's other waves -- and when nobody parked one (no run longer than a
        // chunk: the usual batch) there is no second pass; nor a barrier behind it (every caller begins with one)
        if (pass == 0 && !__syncthreads_or(parked)) return;
    }
}

struct FusedOneTable {
    const FusedUpdateParams& p;
    __device__ __forceinline__ const FusedUpdateParams& operator()(int) const { return p; }
    __device__ __forceinline__ int64_t row(unsigned long long key) const { return (int64_t)(key >> 24); }
};
// two lists end to end: positions below `split` are table A's; B's keys carry bit 60 (bit 36 of the row field), so that a run
// never continues into an equal id of the other table
static constexpr unsigned long long FUSED_ROW_MASK = (1ull << 36) - 1ull;
struct FusedTwoTables {
    const FusedUpdateParams &pa, &pb;
    int split;
    __device__ __forceinline__ const FusedUpdateParams& operator()(int k) const { return k < split ? pa : pb; }
    __device__ __forceinline__ int64_t row(unsigned long long key) const { return (int64_t)((key >> 24) & FUSED_ROW_MASK); }
};
template <int D, bool ADAM, int NFLIGHT, class Keys>
__device__ __forceinline__ void fused_runs(Keys K, int m, const FusedUpdateParams& p, float bc1, float bc2) {
    fused_runs_of<D, ADAM, NFLIGHT>(K, m, FusedOneTable{p}, bc1, bc2);
}

// The work of ONE bucket `myb` by a workgroup of FUSED_THREADS threads; `lk`: FUSED_CAP keys of LDS.
template <int D, bool ADAM, int NFLIGHT = (ADAM ? FUSED_NF_ADAM : 16)>
__device__ __forceinline__ void fused_update_body(const FusedUpdateParams& p, const unsigned myb, unsigned long long* lk) {
    __shared__ int s_count, s_lower;
    __shared__ float s_bias[2];
    const int tid = threadIdx.x, lane = mf_lane();
    __syncthreads();                    // (a caller running several buckets: the previous one is done with the shared state)
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
        if (tid < m) {                  // (m <= FUSED_THREADS)
            const unsigned long long mine = lk[tid];
            int rank = 0;
            for (int j0 = 0; j0 < m; j0 += 8) {                      // (eight broadcast reads in flight)
                unsigned long long o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = lk[j0 + u < m ? j0 + u : tid];
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += o[u] < mine ? 1 : 0;
            }
            sorted[rank] = mine;
        }
        __syncthreads();
        fused_runs<D, ADAM, NFLIGHT>(sorted, m, p, bc1, bc2);
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
        fused_runs<D, ADAM, NFLIGHT>(lk, m, p, bc1, bc2);
    } else {
        // the bucket does not fit: its segment of the global lists starts after every key of a lower bucket
        int low = 0;
        for (int q = tid; q < p.n; q += FUSED_THREADS) {
            const long long id = p.idx[q];
            low += (id >= 0 && id < p.n_rows && fused_bucket(id, p.bucket_bits) < myb) ? 1 : 0;
        }
        low = mf_wave_sum_int(low);
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
        fused_runs<D, ADAM, NFLIGHT>(static_cast<const unsigned long long*>(g1), m, p, bc1, bc2);
    }
}

// EVERY bucket of TWO short lists (n <= FUSED_SMALL_N each, bucket_bits <= 3, ids < 2^36; threads [0, n) scan list A,
// threads [512, 512 + n) list B) by one workgroup in ONE pass -- the one-launch step (mf_step_small.hip), where a bucket
// after a bucket, a table after a table, would pay the memory latency of a row update once each.  The result is the one
// update_fused_kernel's workgroups produce: the sorted lists of the buckets are laid end to end, each starting at a
// multiple of RUN_CHUNK (the gap filled with FUSED_PAD), so that every run is cut into the same chunks, summed in the same
// order, as in its own bucket's list.  `bias`: the two Adam bias corrections, evaluated by the caller.
static constexpr int FUSED_SMALL_N = 256;
template <int D, bool ADAM, int NFLIGHT>
__device__ __forceinline__ void fused_update_small(const FusedUpdateParams& pa, const FusedUpdateParams& pb, const float* bias,
                                                   unsigned long long* lk) {
    __shared__ int s_n[2], s_cnt[16], s_pad[17];
    const int tid = threadIdx.x;
    const float bc1 = ADAM ? bias[0] : 1.f, bc2 = ADAM ? bias[1] : 1.f;      // (read first: `bias` may lie inside `lk`)
    __syncthreads();
    if (tid < 16) s_cnt[tid] = 0;
    if (tid < 2) s_n[tid] = 0;
    __syncthreads();
    unsigned long long* raw[2] = {lk, lk + FUSED_SMALL_N};
    {
        const int which = tid >> 9, t = tid & 511;                   // (wave-uniform)
        const FusedUpdateParams& p = which ? pb : pa;
        const long long id = t < p.n ? p.idx[t] : -1;
        const bool mine = id >= 0 && id < p.n_rows;
        const unsigned b = mine ? fused_bucket(id, p.bucket_bits) : 0u;
        fused_append(mine, ((unsigned long long)b << 61) | ((unsigned long long)which << 60) | ((unsigned long long)id << 24) | (unsigned)t, &s_n[which], raw[which],
                     FUSED_SMALL_N);
        if (mine) atomicAdd(&s_cnt[8 * which + b], 1);
    }
    __syncthreads();
    const int ma = s_n[0], mb = s_n[1];
    if (tid == 0) {
        int at = 0;                                                  // where bucket b's segment starts: a multiple of RUN_CHUNK
        for (int b = 0; b < 16; ++b) {
            s_pad[b] = at;
            at += (s_cnt[b] + RUN_CHUNK - 1) / RUN_CHUNK * RUN_CHUNK;
        }
        s_pad[16] = at;
    }
    unsigned long long* sorted = lk + FUSED_CAP / 2;
    int* rk = reinterpret_cast<int*>(lk + 2 * FUSED_SMALL_N);         // [2][FUSED_SMALL_N] ranks inside the segments
    for (int t = tid; t < 2 * FUSED_SMALL_N + 16 * RUN_CHUNK; t += FUSED_THREADS) sorted[t] = FUSED_PAD;
    for (int t = tid; t < 2 * FUSED_SMALL_N; t += FUSED_THREADS) rk[t] = 0;
    __syncthreads();
    if (ma + mb == 0) return;
    {
        // ranks inside the bucket segments: a key's m comparisons are dealt to 512 / (m rounded up to 64) threads (m <= 64:
        // one batch of 8 broadcast reads each), partial counts meet in LDS
        const int which = tid >> 9, t5 = tid & 511;
        const int m = which ? mb : ma;
        const int mw = ((m + 63) & ~63) > 0 ? ((m + 63) & ~63) : 64, parts = 512 / mw;      // (mw in {64, 128, 192, 256}: parts 8, 4, 2, 2)
        const int t = t5 % mw, part = t5 / mw;
        if (t < m && part < parts) {
            const unsigned long long* list = raw[which];
            const unsigned long long mine = list[t];
            const unsigned b = (unsigned)(mine >> 61);
            const int per = (m + parts - 1) / parts, j_lo = part * per, j_hi = min(m, j_lo + per);
            int below = 0;
            for (int j0 = j_lo; j0 < j_hi; j0 += 8) {
                unsigned long long o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = list[j0 + u < j_hi ? j0 + u : t];      // (own key: neither below nor counted)
#pragma unroll
                for (int u = 0; u < 8; ++u) below += ((o[u] >> 61) == b && o[u] < mine) ? 1 : 0;
            }
            if (below) atomicAdd(&rk[which * FUSED_SMALL_N + t], below);
        }
    }
    __syncthreads();
    {
        const int which = tid >> 9, t = tid & 511;
        const int m = which ? mb : ma;
        if (t < m) {
            const unsigned long long mine = raw[which][t];
            sorted[s_pad[8 * which + (unsigned)(mine >> 61)] + rk[which * FUSED_SMALL_N + t]] = mine & ((1ull << 61) - 1ull);
        }
    }
    __syncthreads();
    const int split = s_pad[8], total = s_pad[16];
    fused_runs_of<D, ADAM, NFLIGHT>(sorted, total, FusedTwoTables{pa, pb, split}, bc1, bc2);
}

template <int D, bool ADAM>
__global__ __launch_bounds__(FUSED_THREADS) void update_fused_kernel(FusedUpdateParams p) {
    extern __shared__ __attribute__((aligned(16))) char fused_smem[];
    fused_update_body<D, ADAM>(p, blockIdx.x, reinterpret_cast<unsigned long long*>(fused_smem));
}

// BOTH tables of a step in one launch: workgroups [0, nb_a) take table A's buckets, the rest table B's.  The two updates are
// independent; side by side the long pole of one (a popular item's run of 700 duplicates: one workgroup busy for 35 us) no
// longer delays the other's launch, and one launch's fixed cost goes (round 4: 13.8 + 35.7 us in sequence -> one launch).
template <int D, bool ADAM>
__global__ __launch_bounds__(FUSED_THREADS) void update_fused_pair_kernel(FusedUpdateParams pa, FusedUpdateParams pb, int nb_a) {
    extern __shared__ __attribute__((aligned(16))) char fused_smem[];
    // (two calls, not one call on a selected parameter set: the selection moved the set out of scalar registers and spilled)
    if ((int)blockIdx.x < nb_a) fused_update_body<D, ADAM>(pa, blockIdx.x, reinterpret_cast<unsigned long long*>(fused_smem));
    else fused_update_body<D, ADAM>(pb, blockIdx.x - nb_a, reinterpret_cast<unsigned long long*>(fused_smem));
}

// buckets (= workgroups) of a list of n ids: ~32 ids each, at most 2^FUSED_MAX_BITS
__host__ __device__ static inline int fused_bucket_bits(int64_t n) {
    int bits = 0;
    while (bits < FUSED_MAX_BITS && (32ll << bits) < n) ++bits;
    return bits;
}

#endif  // __HIPCC__
