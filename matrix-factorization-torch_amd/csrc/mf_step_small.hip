// mf_step_small.hip -- the reference's DEFAULT training step in ONE launch (gfx950).
//
// The reference trains with 32 pairs per step (BATCH_SIZE, xfmr_rec/params.py:18), PairwiseHingeLoss and 4 mined negatives
// (xfmr_rec/lightning.py:38-39), 32 features.  At that size the multi-kernel path -- 11+ launches of a few microseconds
// each -- is bound by launch latency, not by work: ~0.5 MFLOP per step.  Here ONE workgroup of 1024 threads runs the whole
// step on B <= 128 pairs, and what bounds it is one CU's instruction issue and the chain of dependent memory round trips:
//   ids -> [table rows + logq + a word of every Adam-moment line (L2 warming) in flight; meanwhile: id table in LDS, hit
//   masks from the padded positives asked for at the very top] -> rows normalised into LDS -> chain norms, one chain per
//   thread -> u.v on the matrix core (32 x 32 x 2 fp32 MFMA == the k-ordered fmaf chain) -> mining keys, k rounds of a
//   DPP wave maximum, the mined logits 64/(D/4) at a time -> row statistics, the seven losses, batch sums -> dU (a lane
//   group per user) and dV (a lane group per COLUMN walking the users that mined it: the batch kernel's fixed-point sums
//   without atomics) in LDS -> both tables' sparse SGD / row-Adam updates in one pass (fused_update_small, mf_update.h).
// Per-row scalars, selections, chain products and gradients stay in LDS when the batch fits (the default shapes do);
// larger ones fall back to the caller's workspace (L2).
//
// Bit-identical to the multi-kernel path (tests/test_gpu_module.py: torch.equal on both tables after several steps): every
// floating-point expression is the one the batch kernels evaluate, in the same order -- the gather's butterfly, the
// k-ordered fmaf chains, mined_rows_kernel's wave-reduced logits, finish_kernel's block sums, mined_bwd_kernel's
// per-feature sums and fixed-point dV, and for the update the very same run logic (fused_runs_of) over the same chunks.
// Integer work (masks, the exact top-k of unique 64-bit keys) is free to take the short way.
//
// Scope: mining on (0 < num_negatives <= 64, num_negatives < N); the in-batch dense path (num_negatives = 0) stays with the
// MFMA sweeps.  s_memrealtime stamps at the phase boundaries go to the workspace's first 128 bytes (tools/lab/small_step_probe.py).
#include "mf_common.h"
#include "mf_loss_math.h"
#include "mf_update.h"

static constexpr int SS_MAXB = 128, SS_MAXN = 256, SS_THREADS = FUSED_THREADS, SS_HT = 512;
// dynamic LDS: [0, FUSED_CAP * 8) the update body's key list (before the updates: the id table and the hit masks), then the
// normalised rows of the batch when they fit (rows padded by 16 bytes: lanes reading the same 16 bytes of 16 different rows
// hit 64 different banks) -- chains that read them from L2 instead cost a memory round trip per 8 features
static constexpr int SS_ROWS0 = FUSED_CAP * 8, SS_ROWS_BYTES = 88 * 1024;
static constexpr int SS_LONG = 512;      // positives of a user walked by its own half-wave; the rest of a longer list by the workgroup
static_assert(SS_THREADS == 1024, "the update body runs with FUSED_THREADS threads");

struct StepSmallWs {
    float *u, *v, *nu, *nv, *lii, *dii, *sgn, *tgt, *nlogq, *stats, *rowc, *sel_L, *du, *dv, *partial, *blockpart;
    int32_t *sel, *sel_cnt;
    float* dots;                     // [SS_MAXB][SS_MAXN + 8] chain products of a batch too large for LDS
    unsigned long long *gk0, *gk1;
    unsigned long long* stamps;      // [16] at offset 0: s_memrealtime (100 MHz) at the phase boundaries of the last step (tools/lab/small_step_probe.py)
    size_t total;
};
static StepSmallWs step_small_ws(void* base, int d) {
    MfArena a(base);
    StepSmallWs w;
    const size_t Bp = SS_MAXB, Np = SS_MAXN;
    w.stamps = a.take<unsigned long long>(16);
    w.u = a.take<float>(Bp * d); w.v = a.take<float>(Np * d);
    w.nu = a.take<float>(Bp); w.nv = a.take<float>(Np);
    w.lii = a.take<float>(Bp); w.dii = a.take<float>(Bp); w.sgn = a.take<float>(Bp); w.tgt = a.take<float>(Bp);
    w.nlogq = a.take<float>(Np);
    w.stats = a.take<float>((size_t)NSTAT * Bp); w.rowc = a.take<float>(4 * Bp);
    w.sel_L = a.take<float>(Bp * KSEL_MAX); w.sel = a.take<int32_t>(Bp * KSEL_MAX); w.sel_cnt = a.take<int32_t>(Bp);
    w.du = a.take<float>(Bp * d); w.dv = a.take<float>(Np * d);
    w.dots = a.take<float>(Bp * (Np + 8));
    w.partial = a.take<float>((Np + Bp) * d);
    w.blockpart = a.take<float>((size_t)MF_NUM_KINDS * 4);
    w.gk0 = a.take<unsigned long long>(Np); w.gk1 = a.take<unsigned long long>(Np);
    w.total = a.used();
    return w;
}
extern "C" size_t mf_step_small_ws_bytes(int d) { return mf_width_ok(d) ? step_small_ws(nullptr, d).total : 0; }

struct StepSmallParams {
    float *ut, *um, *uv; long long n_users;
    float *it, *im, *iv; long long n_items;
    const int64_t *user_ids, *item_ids;
    const void* target; int target_i64;
    const int64_t* pos_idx; int P;
    const int64_t *pos_off, *pos_items; long long pos_users;
    int B, N, kind, kind_mask, k, normalize;
    float sigma, margin;
    const float* logq; long long logq_rows;
    AdamHyper hp;
    StepSmallWs w;
    float* out;
};

__device__ __forceinline__ unsigned long long ss_shfl_or_u64(unsigned long long x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x |= mf_shfl_xor_u64(x, m);
    return x;
}

// a workgroup barrier for LDS traffic only: global loads stay in flight across it (__syncthreads waits for them)
__device__ __forceinline__ void ss_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int D, bool ADAM>
__global__ __launch_bounds__(SS_THREADS) void step_small_kernel(StepSmallParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];                 // FUSED_CAP * 8 bytes (the update body's key list)
    const int tid = threadIdx.x, lane = mf_lane(), wave = tid >> 6;
    const int B = p.B, N = p.N;
    constexpr int Bp = SS_MAXB;
    const StepSmallWs& w = p.w;
    int stamp_i = 0;
    auto stamp = [&]() {
        if (tid == 0) w.stamps[stamp_i] = __builtin_amdgcn_s_memrealtime();
        ++stamp_i;
    };
    stamp();
    // LDS before the updates: the batch's item ids -> column sets (a 512-slot table), and the users' hit masks
    long long* hkey = reinterpret_cast<long long*>(smem);                       // [SS_HT]
    unsigned long long* hset = reinterpret_cast<unsigned long long*>(smem + SS_HT * 8);          // [SS_HT][4]: columns carrying the id
    unsigned long long* hitm = hset + SS_HT * 4;                                                 // [SS_MAXB][4]: bit j = column j is NOT a valid negative
    // ... and the per-row scalars of the step (norms, diagonal, targets, statistics, backward coefficients), so that no phase
    // waits on a round trip to L2 for a value another wave of this workgroup has just produced
    float* fl = reinterpret_cast<float*>(hitm + SS_MAXB * 4);
    float *nu_l = fl, *lii_l = fl + 128, *dii_l = fl + 256, *sgn_l = fl + 384, *tgt_l = fl + 512, *nv_l = fl + 640, *nlq_l = fl + 896,
          *stats_l = fl + 1152, *rowc_l = fl + 2176, *bpart_l = fl + 2688;
    float* bias_l = fl + 2704;                                                  // [2] Adam's bias corrections
    unsigned* dvred_l = reinterpret_cast<unsigned*>(fl + 2708);                 // [4] magnitudes behind the dV unit: |cg|, |gd| per 64-row block; max |u|^2; max |v|^2
    int32_t* cnt_l = reinterpret_cast<int32_t*>(fl + 2720);                     // [SS_MAXB] selected columns per user
    long long* itemid_l = reinterpret_cast<long long*>(cnt_l + SS_MAXB);        // [SS_MAXN] the columns' item ids
    const long long** lists_l = reinterpret_cast<const long long**>(itemid_l + SS_MAXN);     // [SS_MAXB] the users' positive lists ...
    int32_t* lens_l = reinterpret_cast<int32_t*>(lists_l + SS_MAXB);            // [SS_MAXB] ... and their lengths
    int32_t* any_long = lens_l + SS_MAXB;                                       // [1] some user's list is longer than SS_LONG (+ 3 words of padding)
    int32_t* sel_l = any_long + 4;                                              // [B][k] selected columns, then [B][k] their logits
    const bool sel_in_lds = (size_t)B * p.k * 8 <= (size_t)(SS_ROWS0 - ((char*)sel_l - smem));
    int32_t* selp = sel_in_lds ? sel_l : w.sel;
    float* sel_Lp = sel_in_lds ? reinterpret_cast<float*>(sel_l + (size_t)B * p.k) : w.sel_L;
    const int sels = sel_in_lds ? p.k : KSEL_MAX;                               // entries per user
    constexpr int ROWF = D + 4;
    const bool in_lds = (size_t)(B + N) * ROWF * 4 <= (size_t)SS_ROWS_BYTES;
    float* rows_lds = reinterpret_cast<float*>(smem + SS_ROWS0);
    // row r of the batch (users 0 .. B-1, then the N columns): from LDS when the batch fits, else from the workspace copy
    const float* ubase = in_lds ? rows_lds : w.u;
    const float* vbase = in_lds ? rows_lds + (size_t)B * ROWF : w.v;
    const int rstride = in_lds ? ROWF : D;
    // the chain products of every (user, column): behind the rows when that fits too
    const int dstride = ((N + 31) & ~31) + 8;                 // (+8: the two half-waves of a tile store land in different banks)
    const bool dots_in_lds = in_lds && (size_t)(B + N) * ROWF * 4 + (size_t)B * dstride * 4 <= (size_t)SS_ROWS_BYTES;
    float* dots = dots_in_lds ? rows_lds + (size_t)(B + N) * ROWF : w.dots;
    // the gradients of the gathered rows (du, dv): over the chain products, which are dead by then
    const bool grads_in_lds = in_lds && (size_t)(B + N) * ROWF * 4 + (size_t)(B + N) * D * 4 <= (size_t)SS_ROWS_BYTES;
    float* dup = grads_in_lds ? rows_lds + (size_t)(B + N) * ROWF : w.du;
    float* dvp = grads_in_lds ? dup + (size_t)B * D : w.dv;

    // ---- 1-3. the front of the step is a chain of dependent memory round trips (ids -> table rows; ids -> logq; positives ->
    //           masks), each ~1-2 us for one workgroup: they are issued as early as their addresses are known and the work
    //           that needs only LDS (the id table, the masks) runs under the table rows' latency.  Barriers in between are
    //           LDS-only (ss_lds_barrier: no vmcnt wait), loads return in order.
    constexpr int LPR = D / 4, RPW = 64 / LPR, RPI = (SS_THREADS / 64) * RPW, PF = 4;      // rows per pass; passes in flight
    const int gc = lane % LPR, rsub = wave * RPW + lane / LPR;
    long long rid[PF];
#pragma unroll
    for (int it = 0; it < PF; ++it) {
        const int r = it * RPI + rsub;
        rid[it] = r < B ? p.user_ids[r] : (r < B + N ? p.item_ids[r - B] : -1ll);
    }
    const long long my_item = tid < N ? p.item_ids[tid] : 0ll;                           // column tid's id
    float tg_reg = 0.f;
    if (tid < B) tg_reg = p.target_i64 ? (float)static_cast<const int64_t*>(p.target)[tid] : static_cast<const float*>(p.target)[tid];
    // the first 64 padded positives of the first 32 users (a half-wave per user): their addresses need no id
    constexpr long long NO_KEY = (long long)0x8080808080808080ull;                       // (never an id: the table's empty slot)
    long long pre[2] = {NO_KEY, NO_KEY};
    {
        const int i = 2 * wave + (lane >> 5), l32 = lane & 31;
        if (!p.pos_off && i < B) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                if (l32 + 32 * sl < p.P) pre[sl] = p.pos_idx[(size_t)i * p.P + l32 + 32 * sl];
        }
    }
    if (tid == 0) *any_long = 0;
    for (int e = tid; e < SS_HT; e += SS_THREADS) {
        hkey[e] = NO_KEY;
        hset[4 * e] = hset[4 * e + 1] = hset[4 * e + 2] = hset[4 * e + 3] = 0ull;
    }
    ss_lds_barrier();
    auto slot_of = [](long long key) {
        return ((((unsigned)key * 2654435761u) ^ ((unsigned)((unsigned long long)key >> 32) * 40503u)) >> 5) & (SS_HT - 1);
    };
    if (tid < N) {                                               // column tid joins the set of its item id
        unsigned h = slot_of(my_item);
        for (;;) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&hkey[h]), (unsigned long long)NO_KEY, (unsigned long long)my_item);
            if (old == (unsigned long long)NO_KEY || old == (unsigned long long)my_item) break;
            h = (h + 1) & (SS_HT - 1);
        }
        atomicOr(&hset[4 * h + (tid >> 6)], 1ull << (tid & 63));
        itemid_l[tid] = my_item;
    }
    // the table rows (and logq) are asked for now ...
    auto load_row = [&](int r, long long row) {
        const bool is_u = r < B;
        const bool in_range = row >= 0 && row < (is_u ? p.n_users : p.n_items);
        f32x4 x = reinterpret_cast<const f32x4*>((is_u ? p.ut : p.it) + (in_range ? row : 0) * D)[gc];
        if (!in_range) x = f32x4{0.f, 0.f, 0.f, 0.f};
        return x;
    };
    auto keep_row = [&](int r, f32x4 x) {                        // gather_rows_kernel: D/4 lanes per row, the same butterfly
        if (p.normalize) {
            float ss = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
            ss = mf_group_sum(ss, LPR);
            const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
            x = x * inv;
        }
        if (r < B + N) {
            if (in_lds) reinterpret_cast<f32x4*>(rows_lds + (size_t)r * ROWF)[gc] = x;
            else reinterpret_cast<f32x4*>((r < B ? w.u + (size_t)r * D : w.v + (size_t)(r - B) * D))[gc] = x;
        }
    };
    f32x4 xrow[PF];
#pragma unroll
    for (int it = 0; it < PF; ++it) xrow[it] = load_row(it * RPI + rsub, rid[it]);
    // ... and one word of every 128-byte line of the rows' Adam moments, which the updates at the far end of the step would
    // otherwise fetch cold (HBM + a TLB miss, on the critical path): this brings them into the XCD's L2
    unsigned warm = 0u;
    if (ADAM) {
        constexpr int LINES = D * 4 / 128;                       // (2 * LINES <= D / 4 lanes of the row's group)
#pragma unroll
        for (int it = 0; it < PF; ++it) {
            const int r = it * RPI + rsub;
            const long long row = rid[it];
            if (gc < 2 * LINES && r < B + N && row >= 0 && row < (r < B ? p.n_users : p.n_items)) {
                const float* base = r < B ? ((gc & 1) ? p.uv : p.um) : ((gc & 1) ? p.iv : p.im);
                warm ^= *(reinterpret_cast<const unsigned*>(base + row * D) + 32 * (gc >> 1));
            }
        }
    }
    if (ADAM && tid == SS_THREADS - 1) {                         // two double-precision expm1 by one lane: under the rows' latency, once
        float b1, b2;
        adam_bias(p.hp, b1, b2);
        bias_l[0] = b1; bias_l[1] = b2;
    }
    float lq_reg = 0.f;
    if (p.logq && tid < N) {
        if (p.logq_rows > 0) lq_reg = (my_item >= 0 && my_item < p.logq_rows) ? p.logq[my_item] : 0.f;
        else lq_reg = p.logq[tid];
    }
    ss_lds_barrier();                                          // the id table is complete
    stamp();

    // ... and the hit masks (negative_masks, losses.py:92-110) are made while they travel: a half-wave per user ORs the column
    // sets of its positives and of its own item
    for (int i0 = 0; i0 < B; i0 += 2 * (SS_THREADS / 64)) {
        const int i = i0 + 2 * wave + (lane >> 5), l32 = lane & 31;
        const int64_t* list = nullptr;
        int len = -1;                                                // (no user: nothing to look up, not even an own item)
        if (i < B) {
            len = 0;
            if (p.pos_off) {
                const long long uidx = p.user_ids[i];
                if (uidx >= 0 && uidx < p.pos_users) {
                    const long long o = p.pos_off[uidx];
                    list = p.pos_items + o;
                    const long long ln = p.pos_off[uidx + 1] - o;
                    len = (int)(ln < 0 ? 0 : (ln > 0x7FFFFFF ? 0x7FFFFFF : ln));
                }
            } else if (p.P > 0) {
                list = p.pos_idx + (size_t)i * p.P;
                len = p.P;
            }
        }
        // a half-wave walks the first SS_LONG entries of its user's list; what lies beyond (a heavy user: tens of thousands
        // of positives) is left to the whole workgroup below -- one half-wave would need a memory round trip per 256 of them
        if (l32 == 0 && i < B) {
            lists_l[i] = reinterpret_cast<const long long*>(list);
            lens_l[i] = len;
            if (len > SS_LONG) *any_long = 1;
        }
        if (len > SS_LONG) len = SS_LONG;
        unsigned long long m4[4] = {0ull, 0ull, 0ull, 0ull};
        auto look_up = [&](long long key) {
            unsigned h = slot_of(key);
            for (;;) {
                const long long sv = hkey[h];
                if (sv == key) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) m4[q] |= hset[4 * h + q];
                    break;
                }
                if (sv == NO_KEY) break;
                h = (h + 1) & (SS_HT - 1);
            }
        };
        int t = l32;
        if (i0 == 0 && !p.pos_off) {                                 // (wave-uniform) the entries asked for at the top
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
                if (t < len) { look_up(pre[sl]); t += 32; }
        }
        for (; t < len; t += 32 * 8) {                               // eight list loads in flight per lane (a heavy user's 30,000
            long long k8[8];                                         // positives: a memory round trip per 256 of them, not per 32)
#pragma unroll
            for (int u = 0; u < 8; ++u) k8[u] = t + 32 * u < len ? list[t + 32 * u] : NO_KEY;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k8[u] != NO_KEY) look_up(k8[u]);
        }
        if (l32 == 0 && len >= 0) look_up(itemid_l[i]);              // the user's own item
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned long long m = m4[q];
#pragma unroll
            for (int sh = 16; sh >= 1; sh >>= 1) m |= mf_shfl_xor_u64(m, sh);      // (within the half-wave)
            // padding columns (>= N) are never negatives
            const int lo = 64 * q;
            if (N < lo + 64) m |= N <= lo ? ~0ull : (~0ull << (N - lo));
            if (l32 == 0 && i < B) hitm[4 * i + q] = m;
        }
    }
    stamp();

    // the rows have landed: tower forward (L2 normalisation), into LDS
#pragma unroll
    for (int it = 0; it < PF; ++it) keep_row(it * RPI + rsub, xrow[it]);
    for (int r0 = PF * RPI; r0 < B + N; r0 += RPI) {
        const int r = r0 + rsub;
        const long long row = r < B ? p.user_ids[r] : (r < B + N ? p.item_ids[r - B] : -1ll);
        keep_row(r, load_row(r, row));
    }
    __syncthreads();
    if (*any_long) {                                             // (workgroup-uniform; rare: nothing but this test otherwise)
        // the long lists' remainders, by all 1024 threads, eight loads in flight each (a 30,000-entry list: four round trips)
        for (int i = 0; i < B; ++i) {
            const int len = lens_l[i];
            if (len <= SS_LONG) continue;                            // (workgroup-uniform)
            const long long* list = lists_l[i];
            unsigned long long m4[4] = {0ull, 0ull, 0ull, 0ull};
            for (int t = SS_LONG + tid; t < len; t += SS_THREADS * 8) {
                long long k8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) k8[u] = t + SS_THREADS * u < len ? list[t + SS_THREADS * u] : NO_KEY;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k8[u] == NO_KEY) continue;
                    unsigned h = slot_of(k8[u]);
                    for (;;) {
                        const long long sv = hkey[h];
                        if (sv == k8[u]) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) m4[q] |= hset[4 * h + q];
                            break;
                        }
                        if (sv == NO_KEY) break;
                        h = (h + 1) & (SS_HT - 1);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned long long m = m4[q];
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) m |= mf_shfl_xor_u64(m, sh);
                if (lane == 0 && m) atomicOr(&hitm[4 * i + q], m);
            }
        }
        ss_lds_barrier();
    }
    stamp();

    // ---- chain norms of both operands, the diagonal, -logq (prep_kernel).  One chain per thread -- N of v.v, B of u.u, B of
    //      u_i.v_i -- instead of three per column: a chain is a sequence of dependent LDS round trips, and the default batch
    //      would leave them all to one wave
    {
        float* dot_l = stats_l;                                  // [SS_MAXB] (the statistics are written later)
        const int which = tid < N ? 0 : (tid < N + B ? 1 : 2), i = tid < N ? tid : (tid < N + B ? tid - N : tid - N - B);
        if (tid < N + 2 * B) {
            const f32x4* pa = reinterpret_cast<const f32x4*>((which == 0 ? vbase : ubase) + (size_t)i * rstride);
            const f32x4* pb = reinterpret_cast<const f32x4*>((which == 1 ? ubase : vbase) + (size_t)i * rstride);
            float acc = 0.f;
#pragma unroll 4
            for (int g = 0; g < D / 8; ++g) {                    // k order of mf_dot_chain
                const f32x4 a0 = pa[2 * g], a1 = pa[2 * g + 1], b0 = pb[2 * g], b1 = pb[2 * g + 1];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc = __builtin_fmaf(a0[t], b0[t], acc);
                    acc = __builtin_fmaf(a1[t], b1[t], acc);
                }
            }
            (which == 0 ? nv_l : (which == 1 ? nu_l : dot_l))[i] = acc;
        }
        if (tid >= N && tid < SS_MAXN) nv_l[tid] = 0.f;
        if (tid < SS_MAXN) nlq_l[tid] = tid < N ? -lq_reg : 0.f;
        ss_lds_barrier();
        if (tid < Bp) {
            float l = 0.f, dd = 0.f, sg = 0.f, tg = 0.f, nuu = 0.f;
            if (tid < B) {
                nuu = nu_l[tid];
                const float nvv = nv_l[tid], dot = dot_l[tid];
                tg = tg_reg;
                sg = mf_sign(tg);
                dd = mf_half_sqdist(nuu, nvv, dot);
                l = mf_logit(nuu, nvv, dot, sg, p.sigma, lq_reg);
            }
            nu_l[tid] = nuu; lii_l[tid] = l; dii_l[tid] = dd; sgn_l[tid] = sg; tgt_l[tid] = tg;
        }
    }
    __syncthreads();
    stamp();

    // ---- 4. logits of every (user, column) -> mining keys -> the k best, in order; then (mined_rows_kernel) the selected
    //         negatives' logits by the wave-reduced dot and the row statistics in selection order.  One wave per user.
    const int need = ((p.kind_mask & ((1 << MF_CONTRASTIVE) | (1 << MF_ALIGNMENT_CONTRASTIVE))) ? NEED_CONTR : 0) |
                     ((p.kind_mask & ((1 << MF_INFONCE) | (1 << MF_MINE))) ? NEED_LSE : 0) |
                     ((p.kind_mask & (1 << MF_PAIRWISE_HINGE)) ? NEED_HINGE : 0) | ((p.kind_mask & (1 << MF_PAIRWISE_LOGISTIC)) ? NEED_LOGI : 0);
    {
        // 32 x 32 tiles of u . v on the matrix core: v_mfma_f32_32x32x2_f32 with a lane pair holding elements 8g + 4h .. + 3
        // of a row IS the k order of mf_dot_chain (mf_common.h: mf_tile_scores), and one CU's LDS feeds it 32 x less data
        // than it would feed one FMA chain per lane
        const int TU = (B + 31) / 32, TN = (N + 31) / 32;
        const int r = lane & 31, h = lane >> 5;
        for (int tt = wave; tt < TU * TN; tt += SS_THREADS / 64) {
            const int tu = tt / TN, tn = tt % TN;
            const int iu = 32 * tu + r, jv = 32 * tn + r;
            const f32x4* pa = reinterpret_cast<const f32x4*>(ubase + (size_t)(iu < B ? iu : 0) * rstride + 4 * h);
            const f32x4* pb = reinterpret_cast<const f32x4*>(vbase + (size_t)(jv < N ? jv : 0) * rstride + 4 * h);
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 4
            for (int g = 0; g < D / 8; ++g) {
                const f32x4 a = pa[2 * g], b = pb[2 * g];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * tu + mf_acc_row(e, h);
                if (row < B) dots[(size_t)row * dstride + 32 * tn + r] = acc[e];      // (columns >= N: inside the row's padding, never read)
            }
        }
    }
    __syncthreads();
    for (int i = wave; i < B; i += SS_THREADS / 64) {
        RowStats st;
        stats_init(st);
        {
            const float nu_i = nu_l[i], s_i = sgn_l[i], l = lii_l[i];
            unsigned long long keyv[SS_MAXN / 64];
#pragma unroll
            for (int q = 0; q < SS_MAXN / 64; ++q) {
                const int j = lane + 64 * q;
                keyv[q] = 0ull;
                if (64 * q >= N) continue;                       // (wave-uniform)
                if (j < N && !((hitm[4 * i + q] >> lane) & 1ull)) {
                    const float L = mf_logit(nu_i, nv_l[j], dots[(size_t)i * dstride + j], s_i, p.sigma, -nlq_l[j]);
                    keyv[q] = mf_key_mining(L - l, (unsigned)j);
                }
            }
            int m = 0, mysel = 0;
            for (int t = 0; t < p.k; ++t) {                      // unique keys: k rounds of "largest remaining"
                unsigned long long loc = 0ull;
#pragma unroll
                for (int q = 0; q < SS_MAXN / 64; ++q)
                    if (64 * q < N) loc = keyv[q] > loc ? keyv[q] : loc;
                const unsigned long long best = mf_wave_max_u64(loc);
                if (best == 0ull) break;
#pragma unroll
                for (int q = 0; q < SS_MAXN / 64; ++q)
                    if (64 * q < N && keyv[q] == best) keyv[q] = 0ull;
                if (lane == t) mysel = (int)mf_key_mining_col(best);        // (k <= 64: lane t keeps the t-th selected column)
                ++m;
            }
            if (lane < m) selp[i * sels + lane] = mysel;
            if (lane == 0) cnt_l[i] = m;
            const float sm = s_i * p.margin;
            // the selected negatives' logits (mined_rows_kernel: D/4 lanes x 4 elements, then the wave's butterfly with the other
            // lanes at zero): 64 / (D/4) of them at a time, a lane group each.  A group's butterfly is the wave's with the
            // levels that only add those zeros (x + 0) made explicit -- the same bits.
            constexpr int d4 = D / 4, G = 64 / d4;
            const int grp = lane / d4, c = lane % d4;
            const f32x4 ur = reinterpret_cast<const f32x4*>(ubase + (size_t)i * rstride)[c];
            float myL = 0.f;                                     // lane t: the logit of selection t
            for (int t0 = 0; t0 < m; t0 += G) {
                const int t = t0 + grp;
                const int j = __shfl(mysel, t < m ? t : 0, 64);
                const f32x4 vr = reinterpret_cast<const f32x4*>(vbase + (size_t)j * rstride)[c];
                float part = ur[0] * vr[0] + ur[1] * vr[1] + ur[2] * vr[2] + ur[3] * vr[3];
                if (d4 < 64) part = part + 0.f;
                part = mf_butterfly_sum<d4>(part);
                const float L = mf_logit(nu_i, nv_l[j], part, s_i, p.sigma, -nlq_l[j]);
                if (c == 0 && t < m) sel_Lp[i * sels + t] = L;
                const float got = __shfl(L, ((lane - t0) & (G - 1)) * d4, 64);
                if (lane >= t0 && lane < t0 + G) myL = got;
            }
            for (int t = 0; t < m; ++t) {                        // the statistics, in selection order
                const float L = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myL), t));
                stats_add(st, need, L, sm, l, p.margin);
                if (need & NEED_LSE) lse_merge(st.mx, st.se, L, 1.f);
            }
        }
        if (lane == 0) {
            float* o = stats_l + i;
            o[ST_CNT * Bp] = st.cnt; o[ST_A * Bp] = st.A; o[ST_MX * Bp] = st.mx; o[ST_SE * Bp] = st.se;
            o[ST_H * Bp] = st.H; o[ST_HC * Bp] = st.Hc; o[ST_LG * Bp] = st.Lg; o[ST_LS * Bp] = st.Ls;
        }
    }
    __syncthreads();
    stamp();

    // ---- 5. the seven row losses, the trained loss's backward coefficients, the batch sums (finish_kernel: 64-row blocks, xor trees)
    if (wave < Bp / 64) {
        const int i = wave * 64 + lane;
        float acc[NSTAT];
        for (int s = 0; s < NSTAT; ++s) acc[s] = stats_l[s * Bp + i];
        float o[MF_NUM_KINDS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (i < B) row_losses(acc, tgt_l[i], lii_l[i], dii_l[i], p.sigma, o);
        {
            float a = 0.f, b = 0.f, cg = 0.f, gd = 0.f;
            if (i < B) rowc_row(p.kind, tgt_l[i], sgn_l[i], lii_l[i], acc[ST_CNT], acc[ST_MX], acc[ST_SE], acc[ST_HC], acc[ST_LS],
                                p.sigma, p.margin, a, b, cg, gd);
            rowc_l[i] = a; rowc_l[Bp + i] = b; rowc_l[2 * Bp + i] = cg; rowc_l[3 * Bp + i] = gd;
            const unsigned gm = mf_wave_max_u32(i < B ? max(dv_mag(cg), dv_mag(gd)) : 0u);       // (dv_scale_kernel's maxima)
            if (lane == 0) dvred_l[wave] = gm;
        }
        for (int k = 0; k < MF_NUM_KINDS; ++k) {
            float v = o[k];
            v = mf_wave_sum(v);
            if (lane == 0) bpart_l[k * (Bp / 64) + wave] = v;
        }
    }
    else if (wave == 2 || wave == 3) {                          // (Bp / 64 <= 2) the squared-norm maxima: users by wave 2, columns by wave 3
        unsigned m = 0u;
        if (wave == 2) { for (int i = lane; i < B; i += 64) m = max(m, dv_mag(nu_l[i])); }
        else { for (int j = lane; j < N; j += 64) m = max(m, dv_mag(nv_l[j])); }
        m = mf_wave_max_u32(m);
        if (lane == 0) dvred_l[wave] = m;
    }
    __syncthreads();
    if (wave == 0) {
        for (int k = 0; k < MF_NUM_KINDS; ++k) {
            float t = lane < Bp / 64 ? bpart_l[k * (Bp / 64) + lane] : 0.f;
            t = mf_wave_sum(t);
            if (lane == 0) p.out[k] = ((p.kind_mask >> k) & 1) ? t : 0.f;
        }
    }
    stamp();

    // ---- 6. backward of the trained loss (mined_bwd_kernel's arithmetic).  du: a lane group per user walks its selected columns in
    //         selection order.  dv: the batch kernel adds fixed-point terms with integer atomics, a sum that does not depend on
    //         the order -- here a lane group per COLUMN walks the users that mined it (a bit set per column) and adds the same
    //         integers in registers: no atomics, no zeroing, no second pass
    {
        unsigned long long* colm = reinterpret_cast<unsigned long long*>(smem);      // [N][2] (the id table is dead)
        for (int e = tid; e < 2 * N; e += SS_THREADS) colm[e] = 0ull;
        __syncthreads();
        const int gmode = gmode_of(p.kind);
        const DvFix fix = dv_fix_of(__builtin_bit_cast(float, Bp > 64 ? max(dvred_l[0], dvred_l[1]) : dvred_l[0]),
                                    __builtin_bit_cast(float, dvred_l[2]), __builtin_bit_cast(float, dvred_l[3]), (long long)B);
        for (int e = tid; e < B * p.k; e += SS_THREADS) {
            const int i = e / p.k, sl = e % p.k;
            if (sl < cnt_l[i]) {
                const float a = rowc_l[i], b = rowc_l[Bp + i], cg = rowc_l[2 * Bp + i];      // (upstream gradient of loss.backward(): one)
                const float g = cg * g_of(gmode, (sel_Lp[i * sels + sl] - a) + b);
                sel_Lp[i * sels + sl] = g;                                                   // the logit is not needed again
                atomicOr(&colm[2 * selp[i * sels + sl] + (i >> 6)], 1ull << (i & 63));
            }
        }
        __syncthreads();
        // D/4 lanes per row, four consecutive features each (one 16-byte LDS read per operand row; every feature's sum runs
        // over the selections in the same order whichever lane holds it)
        constexpr int LPR = D / 4;
        for (int t0 = 0; t0 < (B + N) * LPR; t0 += SS_THREADS) {
            const int t = t0 + tid;
            const int r = t / LPR, c = t % LPR;
            if (r < B) {
                const int i = r;
                const f32x4 ui = reinterpret_cast<const f32x4*>(ubase + (size_t)i * rstride)[c];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int n = cnt_l[i];
                for (int sl = -1; sl < n; ++sl) {
                    const int j = sl < 0 ? i : selp[i * sels + sl];
                    const float g = sl < 0 ? rowc_l[3 * Bp + i] : sel_Lp[i * sels + sl];
                    const f32x4 vj = reinterpret_cast<const f32x4*>(vbase + (size_t)j * rstride)[c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += g * (vj[e] - ui[e]);
                }
                reinterpret_cast<f32x4*>(dup + (size_t)i * D)[c] = acc;
            } else if (r < B + N) {
                const int j = r - B;
                const f32x4 vj = reinterpret_cast<const f32x4*>(vbase + (size_t)j * rstride)[c];
                long long sum[4] = {0ll, 0ll, 0ll, 0ll};
                auto add = [&](int i, float g) {
                    const f32x4 ui = reinterpret_cast<const f32x4*>(ubase + (size_t)i * rstride)[c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dvj = g * (ui[e] - vj[e]);
                        sum[e] += dv_fix_term(dvj, fix.scale, fix.clamp);
                    }
                };
                if (j < B) add(j, rowc_l[3 * Bp + j]);
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    unsigned long long m = colm[2 * j + h];
                    while (m) {
                        const int i = 64 * h + __builtin_ctzll(m);
                        m &= m - 1;
                        int sl = 0;
                        while (selp[i * sels + sl] != j) ++sl;                                // (there: the bit says so)
                        add(i, sel_Lp[i * sels + sl]);
                    }
                }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (float)((double)sum[e] * (double)fix.inv);
                reinterpret_cast<f32x4*>(dvp + (size_t)j * D)[c] = o;
            }
        }
    }
    __syncthreads();
    stamp();

    // ---- 7. the sparse updates: the one-launch update's own run logic (fused_runs_of), every bucket of both tables in one pass
    {
        unsigned long long* lk = reinterpret_cast<unsigned long long*>(smem);
        FusedUpdateParams fi{p.it, p.im, p.iv, p.n_items, reinterpret_cast<const long long*>(p.item_ids), N, fused_bucket_bits(N), dvp,
                             w.partial, w.gk0, w.gk1, p.normalize, p.hp};
        FusedUpdateParams fu{p.ut, p.um, p.uv, p.n_users, reinterpret_cast<const long long*>(p.user_ids), B, fused_bucket_bits(B), dup,
                             w.partial + (size_t)SS_MAXN * D, w.gk0, w.gk1, p.normalize, p.hp};
        fused_update_small<D, ADAM, 4>(fi, fu, bias_l, lk);
    }
    stamp();
    if (warm == 0x7FC5A5A5u) w.stamps[15] = warm;               // (keeps the warming loads; practically never true, harmless when it is)
}

extern "C" int mf_step_small(float* user_table, float* user_m, float* user_v, int64_t num_users, float* item_table, float* item_m,
                             float* item_v, int64_t num_items, int d, int normalize, const int64_t* user_ids, const int64_t* item_ids,
                             const void* target, int target_i64, const int64_t* pos_idx, int P, const int64_t* pos_off,
                             const int64_t* pos_items, int64_t pos_users, int64_t B, int64_t N, int kind, int kind_mask,
                             int num_negatives, float sigma, float margin, const float* logq, int64_t logq_rows, int adam, int64_t step,
                             const int64_t* step_dev, float lr, float beta1, float beta2, float eps, float weight_decay, void* ws,
                             size_t ws_bytes, float* out_losses, mf_stream_t stream) {
    if (!user_table || !item_table || !user_ids || !item_ids || !target || !ws || !out_losses || num_users <= 0 || num_items <= 0)
        return mf_set_error(MF_EINVAL, "mf_step_small: bad argument");
    if (!mf_width_ok(d)) return mf_set_error(MF_EINVAL, "mf_step_small: embedding width %d not in {32,64,128,256}", d);
    if (B <= 0 || N < B || B > SS_MAXB || N > SS_MAXN)
        return mf_set_error(MF_ENOTSUP, "mf_step_small: needs 0 < B <= %d and B <= N <= %d (B=%lld N=%lld)", SS_MAXB, SS_MAXN, (long long)B, (long long)N);
    if (!(num_negatives > 0 && num_negatives < N && num_negatives <= KSEL_MAX))
        return mf_set_error(MF_ENOTSUP, "mf_step_small: the one-launch step covers mined losses (0 < num_negatives <= %d < N)", KSEL_MAX);
    if (kind < 0 || kind >= MF_NUM_KINDS || !((kind_mask >> kind) & 1)) return mf_set_error(MF_EINVAL, "mf_step_small: kind / kind_mask");
    if (adam && (!user_m || !user_v || !item_m || !item_v || (!step_dev && step < 1))) return mf_set_error(MF_EINVAL, "mf_step_small: Adam state / step");
    if (pos_off ? (!pos_items || pos_users <= 0) : (P < 0 || (P > 0 && !pos_idx))) return mf_set_error(MF_EINVAL, "mf_step_small: bad positives");
    if (logq && logq_rows <= 0 && false) return MF_EINVAL;
    if (num_users >= (1ll << 36) || num_items >= (1ll << 36)) return mf_set_error(MF_ENOTSUP, "mf_step_small: tables of < 2^36 rows");
    static_assert(SS_MAXN <= FUSED_SMALL_N && SS_MAXB <= FUSED_SMALL_N, "fused_update_small's list");
    if (ws_bytes < mf_step_small_ws_bytes(d)) return mf_set_error(MF_ENOSPC, "mf_step_small: workspace too small");
    StepSmallParams sp{};
    sp.ut = user_table; sp.um = user_m; sp.uv = user_v; sp.n_users = num_users;
    sp.it = item_table; sp.im = item_m; sp.iv = item_v; sp.n_items = num_items;
    sp.user_ids = user_ids; sp.item_ids = item_ids; sp.target = target; sp.target_i64 = target_i64;
    sp.pos_idx = pos_idx; sp.P = P; sp.pos_off = pos_off; sp.pos_items = pos_items; sp.pos_users = pos_users;
    sp.B = (int)B; sp.N = (int)N; sp.kind = kind; sp.kind_mask = kind_mask; sp.k = num_negatives; sp.normalize = normalize;
    sp.sigma = sigma; sp.margin = margin; sp.logq = logq; sp.logq_rows = logq_rows;
    sp.hp = adam ? AdamHyper{lr, beta1, beta2, eps, weight_decay, (long long)step, reinterpret_cast<const long long*>(step_dev),
                             log((double)beta1), log((double)beta2)}
                 : AdamHyper{lr, 0.f, 0.f, 0.f, weight_decay, 1, nullptr, 0.0, 0.0};
    sp.w = step_small_ws(ws, d);
    sp.out = out_losses;
    hipStream_t s = static_cast<hipStream_t>(stream);
    MF_DISPATCH_D(d, {
        if (adam) {
            auto fn = step_small_kernel<D, true>;
            static bool set = false;
            if (!set) { (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, SS_ROWS0 + SS_ROWS_BYTES); set = true; }
            fn<<<dim3(1), SS_THREADS, SS_ROWS0 + SS_ROWS_BYTES, s>>>(sp);
        } else {
            auto fn = step_small_kernel<D, false>;
            static bool set = false;
            if (!set) { (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, SS_ROWS0 + SS_ROWS_BYTES); set = true; }
            fn<<<dim3(1), SS_THREADS, SS_ROWS0 + SS_ROWS_BYTES, s>>>(sp);
        }
    });
    return mf_check_launch("mf_step_small");
}
