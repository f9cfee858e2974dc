// mf_data.hip -- the batch producer of the training path, on the device.
//
// Reference: every step's batch is assembled on the host, example by example
// (xfmr_rec/data/lightning.py:311-363: InteractionProcessor.process / collate /
// get_processed_data -- shuffled (user, item, rating) examples, the user's positive set
// `pos_idx` padded with 0 (data/load.py:38-55), one uniformly cycled negative item per
// example).  Here the interaction list and the per-user positive lists stay in HBM and one
// launch writes the whole batch in the layout the loss path reads (InteractionBatchType,
// data/lightning.py:72-76): ids only, no host round trip, no ragged tensors.
//
// Order and randomness are counter-based, so a batch is a pure function of (seed, position):
//   * example p of the stream is interaction perm_e(p mod n) of epoch e = p / n, where perm_e is a
//     4-round Feistel permutation of [0, n) (cycle-walking; round function = SplitMix64 of
//     (half, round, epoch, seed)): every epoch visits every interaction exactly once, reshuffled;
//   * its negative is item 1 + SplitMix64(seed, p) mod (num_items - 1): uniform over the real
//     items (row 0 is padding), like the reference's independently shuffled item cycle.
// Spec: oracle/data.py (exact integer arithmetic); no reference arithmetic to pin (parity unpinned).
#include "mf_common.h"

__host__ __device__ __forceinline__ unsigned long long mf_feistel_perm(unsigned long long x, unsigned long long n, int half_bits,
                                                                       unsigned long long epoch, unsigned long long seed) {
    const unsigned long long mask = (1ull << half_bits) - 1ull;
    do {                                               // cycle-walking: the 2^(2 half_bits) permutation restricted to [0, n)
        unsigned long long l = x >> half_bits, r = x & mask;
        for (int round = 0; round < 4; ++round) {
            const unsigned long long f = mf_splitmix64(r + ((unsigned long long)round << 56) + epoch * 0x9E3779B97F4A7C15ull + seed) & mask;
            const unsigned long long nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << half_bits) | r;
    } while (x >= n);
    return x;
}

__global__ __launch_bounds__(256) void sample_batch_kernel(const int64_t* __restrict__ pair_user, const int64_t* __restrict__ pair_item,
                                                           const float* __restrict__ pair_target, int64_t n_pairs, int half_bits,
                                                           const int64_t* __restrict__ pos_off, const int64_t* __restrict__ pos_items,
                                                           int64_t num_items, unsigned long long seed, int64_t start, int64_t B, int P,
                                                           int64_t* __restrict__ out_user, int64_t* __restrict__ out_item,
                                                           float* __restrict__ out_target, int64_t* __restrict__ out_pos) {
    // one wave per example: lane 0 resolves the example, all lanes copy its positive list (coalesced)
    const int lane = mf_lane();
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;
    const unsigned long long p = (unsigned long long)(start + r);
    const unsigned long long epoch = p / (unsigned long long)n_pairs;
    const unsigned long long e = mf_feistel_perm(p % (unsigned long long)n_pairs, (unsigned long long)n_pairs, half_bits, epoch, seed);
    const int64_t user = pair_user[e];
    if (lane == 0) {
        out_user[r] = user;
        out_item[r] = pair_item[e];
        out_item[B + r] = 1 + (int64_t)(mf_splitmix64(p * 0xD1342543DE82EF95ull + seed + 0x632BE59BD9B4E019ull) % (unsigned long long)(num_items - 1));
        out_target[r] = pair_target[e];
    }
    const int64_t o0 = pos_off[user], len = pos_off[user + 1] - o0;
    for (int t = lane; t < P; t += 64) out_pos[r * P + t] = t < len ? pos_items[o0 + t] : 0;   // 0-padded on the right
}

extern "C" int mf_sample_batch(const int64_t* pair_user, const int64_t* pair_item, const float* pair_target, int64_t n_pairs,
                               const int64_t* pos_off, const int64_t* pos_items, int64_t num_items, uint64_t seed,
                               int64_t start, int64_t B, int P, int64_t* out_user, int64_t* out_item, float* out_target,
                               int64_t* out_pos, mf_stream_t stream) {
    if (!pair_user || !pair_item || !pair_target || !pos_off || !pos_items || !out_user || !out_item || !out_target ||
        (P > 0 && !out_pos) || n_pairs <= 0 || num_items < 2 || start < 0 || B <= 0 || P < 0)
        return mf_set_error(MF_EINVAL, "mf_sample_batch: bad argument");
    int bits = 1;
    while ((1ll << bits) < n_pairs) ++bits;
    const int half_bits = (bits + 1) / 2;
    sample_batch_kernel<<<dim3((unsigned)((B + 3) / 4)), 256, 0, static_cast<hipStream_t>(stream)>>>(
        pair_user, pair_item, pair_target, n_pairs, half_bits, pos_off, pos_items, num_items, seed, start, B, P, out_user, out_item,
        out_target, out_pos);
    return mf_check_launch("mf_sample_batch");
}
