// mf_topk_bf3.hip -- exact full-catalog top-k for MANY queries through a bf16 prefilter (gfx950).
//
// fp32 MFMA runs at 1/16 of the bf16 rate on CDNA4, and an exact top-k needs exact scores only for the few rows
// that can be among the k best.  So the catalog is scanned with bf16 operands (v_mfma_f32_32x32x16_bf16, fp32
// accumulation) and a RIGOROUS error bound decides which rows are rescored with the canonical fp32 fmaf chain:
//
//   index      yb = bf16(y) for every catalog row (round to nearest even), ymax = max row norm      (built once)
//   prep       per query tile: the exclusion lists as bit words laid out [4-tile block][query][4 words] (what a scan wave
//              wants for its 32 queries and one block is 512 contiguous bytes), the bf16 fragments in MFMA operand order, counters
//   seed       a(q, y) = xb_q . yb_y for a SAMPLE of the catalog (one 4-tile block in every `bs`: half of a long
//              catalog, all of a short one); per query only the maximum of every block (by lane half: two values per
//              block) is kept -- each is the score of a distinct, non-excluded row, so the k-th largest of them,
//              tau_q, is a lower bound of the k-th largest a(q, .) over the WHOLE catalog
//   bound      |a - s| <= eps_q for the exact chain score s (below), so k distinct rows have s >= tau_q - eps_q and
//              every row of the true top k has a >= tau_q - 2 eps_q =: thr_q
//   scan       every (query, row) once; rows with a >= thr_q (a few dozen to a few hundred per query) are appended to
//              the query's candidate list
//   final      one wave per query rescoring its candidates with mf_dot_chain (bit for bit the fp32 MFMA element of
//              mf_topk) and selecting the k best 64-bit keys: the result is IDENTICAL to mf_topk's
//
// Geometry of the two scans (round 3; round 2 staged the catalog once per 256-query block and pass -- 8 x 16 MB at
// Q = 1024 -- and spent more time in workgroup prologues than in MFMAs).  A workgroup is four waves, ONE per SIMD, with the
// whole 512-register file each: a wave keeps XT = 8 query tiles (256 queries) as MFMA B operands, so a workgroup covers
// 1024 queries and every catalog tile is staged ONCE for all of them; per tile a wave reads the 8 catalog fragments from LDS
// and runs 8 x 8 MFMAs, the (short) epilogue of query tile xt threaded between the MFMAs of query tile xt + 1.  Exclusion
// words come straight into registers, four tiles per 16-byte load (a query's row of words is contiguous).
//
// Error bound.  bf16 keeps 8 significant bits: |xb - x| <= 2^-8 |x| elementwise, so
// |xb yb - x y| <= (2^-7 + 2^-16) |x||y| per product and, by Cauchy-Schwarz, (2^-7 + 2^-16) |x|_2 |y|_2 for the sum;
// the products of bf16 pairs are exact in fp32 and the two fp32 accumulations (the MFMA's, d terms in some order,
// and the chain's) each stay within d 2^-24 sum |terms| (1 + O(d 2^-24)) of the real sum.  With 1 % slack for the
// fp32 evaluation of the norms:  eps_q = 1.01 (2^-7 + 2^-16 + d 2^-22) |x_q|_2 ymax.
//
// A query whose candidates do not fit (more than BF3_LCAP hits in one lane's share of a chunk, or more than BF3_CAND
// in all: a zero query, thousands of duplicate rows) is answered by the final wave scanning the whole catalog with the
// exact chain: slow, but exact.
#include <cstdlib>

#include "mf_common.h"
#include "mf_select.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int BF3_BLOCK = 4;         // tiles per block: the unit of sampling, of maxima groups and of exclusion-word loads
static constexpr int BF3_SLOTS = 2;         // hits a lane keeps per query tile and chunk, in registers (more: the query's overflow list)
static constexpr int BF3_OVF = 256;         // entries of a query's overflow list
static constexpr int BF3_CAND = 512;        // candidates rescored per query (more: the exact scan of the whole catalog)
static constexpr int BF3_WAVES = 8;         // waves per scan workgroup

__device__ __forceinline__ unsigned short bf3_round(float x) {
    const __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
}

// ------------------------------------------------------------------- index ----
struct Bf3Index {
    unsigned short* plane;      // [N][d] bf16
    float* ymax2;               // max squared row norm (of the fp32 rows)
    size_t total;
};
static Bf3Index bf3_index(void* base, int64_t N, int d) {
    MfArena a(base);
    Bf3Index ix;
    ix.plane = a.take<unsigned short>((size_t)N * d + 64);
    ix.ymax2 = a.take<float>(4);
    ix.total = a.used();
    return ix;
}
extern "C" size_t mf_topk_bf3_index_bytes(int64_t N, int d) {
    if (N <= 0 || !mf_width_ok(d)) return 0;
    return bf3_index(nullptr, N, d).total;
}

// one d/4-lane group per row
template <int D>
__global__ __launch_bounds__(256) void bf3_build_kernel(const float* __restrict__ items, int64_t N, unsigned short* __restrict__ plane,
                                                        float* __restrict__ ymax2) {
    constexpr int LPR = D / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = t / LPR;
    const int c = (int)(t % LPR);
    float ss = 0.f;
    if (r < N) {
        const f32x4 v = reinterpret_cast<const f32x4*>(items + r * D)[c];
        ushort4 o;
        o.x = bf3_round(v[0]); o.y = bf3_round(v[1]); o.z = bf3_round(v[2]); o.w = bf3_round(v[3]);
        reinterpret_cast<ushort4*>(plane + r * D)[c] = o;
        ss = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = mf_butterfly_sum<LPR>(ss);
    // (non-negative floats order like their bit patterns)
    if (r < N && c == 0) atomicMax(reinterpret_cast<unsigned*>(ymax2), __builtin_bit_cast(unsigned, ss));
}

extern "C" int mf_topk_bf3_build(const float* items, int64_t N, int d, void* index, size_t index_bytes, mf_stream_t stream) {
    if (!items || !index || N <= 0 || !mf_width_ok(d)) return mf_set_error(MF_EINVAL, "mf_topk_bf3_build: bad argument");
    if (index_bytes < mf_topk_bf3_index_bytes(N, d)) return mf_set_error(MF_ENOSPC, "mf_topk_bf3_build: index buffer too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Bf3Index ix = bf3_index(index, N, d);
    (void)hipMemsetAsync(ix.ymax2, 0, 16, s);
    MF_DISPATCH_D(d, {
        const int64_t threads = N * (D / 4);
        bf3_build_kernel<D><<<dim3((unsigned)((threads + 255) / 256)), 256, 0, s>>>(items, N, ix.plane, ix.ymax2);
    });
    return mf_check_launch("mf_topk_bf3_build");
}

// -------------------------------------------------------------------- plan ----
struct Bf3Plan {
    int NT, NTp;        // 32-row tiles of the catalog; row stride of the exclusion words (a multiple of 4, + 4)
    int nblk;           // 4-tile blocks
    int XT, xw;         // query tiles per wave; distinct query-tile sets among a workgroup's eight waves (1, 2, 4 or 8)
    int64_t Qp;         // queries padded to a workgroup's 32 * XT * xw
    int gy;             // query blocks
    int bs;             // the seed scan takes one block in every bs
    int nvb_seed, bpc_seed, nwg_seed;   // seed scan: virtual blocks, blocks per workgroup, workgroups
    int bpc, nwg;       // full scan
    int nvals;          // seed maxima per query: nvb_seed * 2 * (BF3_WAVES / xw)
};
static Bf3Plan bf3_plan(int64_t Q, int64_t N, int d) {
    Bf3Plan p{};
    p.NT = (int)((N + 31) / 32);
    p.nblk = (p.NT + BF3_BLOCK - 1) / BF3_BLOCK;
    p.NTp = p.nblk * BF3_BLOCK + 4;
    const int qt = (int)((Q + 31) / 32);
    // eight waves per workgroup, two per SIMD (one's epilogue beside the other's MFMAs), XT query tiles each: 1024 queries
    // per workgroup at d <= 128 (512 at d = 256: its fragments are twice as long), one workgroup per CU
    const int xt_big = d >= 256 ? 2 : 4;
    if (qt <= BF3_WAVES) { p.XT = 1; p.xw = qt <= 1 ? 1 : qt <= 2 ? 2 : qt <= 4 ? 4 : 8; }
    else { p.XT = xt_big; p.xw = BF3_WAVES; }
    const int per_wg = 32 * p.XT * p.xw;
    p.Qp = (Q + per_wg - 1) / per_wg * per_wg;
    p.gy = (int)(p.Qp / per_wg);
    // long catalogs: the bound comes from a quarter of the rows; short ones are sampled whole (few maxima otherwise)
    p.bs = p.nblk >= 256 ? 2 : 1;          // (measured at Q = 1024, N = 62,423: every 4th / 2nd / every block -> 76.5 / 70.4 / 75.9 us, 151 / 78 / 40 candidates per query)
#ifdef MF_BF3_LAB
    if (const char* e = getenv("MF_BF3_BS")) { const int v = atoi(e); if (v >= 1 && p.nblk >= 64 * v) p.bs = v; }   // lab knob (make EXTRA=-DMF_BF3_LAB)
#endif
    p.nvb_seed = (p.nblk + p.bs - 1) / p.bs;
    // (measured and dropped in round 4: XT = 2 at d <= 128 with two workgroups per CU -- the 128-register budget of four waves
    // per SIMD spills 26..49 registers of the scan)
    auto cut = [&](int blocks, int* bpc, int* nwg) {
        int want = (256 + p.gy - 1) / p.gy;                 // one workgroup per CU
        if (want > blocks) want = blocks;
        if (want < 1) want = 1;
        *bpc = (blocks + want - 1) / want;
        *nwg = (blocks + *bpc - 1) / *bpc;
    };
    cut(p.nvb_seed, &p.bpc_seed, &p.nwg_seed);
    cut(p.nblk, &p.bpc, &p.nwg);
    p.nvals = p.nvb_seed * 2 * (BF3_WAVES / p.xw);               // per block: two lane halves x the waves that share a query-tile set
    return p;
}

struct Bf3Ws {
    Bf3Plan plan;
    float* gmax;            // [Qp][nvals]
    float* thr;             // [Qp]
    bf16x8* xfrag;          // [Qp / 32][d / 16][64]: the queries rounded to bf16, in MFMA operand order
    uint32_t* cand;         // [nwg x waves per set][Qp][2 lane halves][4]: {row, row, score - thr, score - thr} of the half's two slots (row 0xFFFFFFFF: none)
    float* eps;             // [Qp]: the queries' error bounds (written by the bound kernel, read by the final one)
    uint32_t* ovf_list;     // [Qp][BF3_OVF]: hits beyond a lane's slots
    int32_t* ovf_cnt;       // [Qp]: their number
    int32_t* ovf;           // [Qp]: the overflow list overflowed too (thousands of duplicate rows, a zero query): exact path
    uint32_t* exclW;        // [nblk + 1][Qp][4]: bit r of word (t & 3) of block t / 4 = row 32 t + r is excluded for the query
    size_t total;
};
static Bf3Ws bf3_ws(void* base, int64_t Q, int64_t N, int d) {
    Bf3Ws w{};
    w.plan = bf3_plan(Q, N, d);
    MfArena a(base);
    w.gmax = a.take<float>((size_t)w.plan.Qp * w.plan.nvals);
    w.thr = a.take<float>((size_t)w.plan.Qp);
    w.xfrag = a.take<bf16x8>((size_t)w.plan.Qp * d / 8);
    w.cand = a.take<uint32_t>((size_t)w.plan.nwg * (BF3_WAVES / w.plan.xw) * w.plan.Qp * 2 * 2 * BF3_SLOTS);
    w.eps = a.take<float>((size_t)w.plan.Qp);
    w.ovf_list = a.take<uint32_t>((size_t)w.plan.Qp * BF3_OVF);
    w.ovf_cnt = a.take<int32_t>((size_t)w.plan.Qp);
    w.ovf = a.take<int32_t>((size_t)w.plan.Qp);
    w.exclW = a.take<uint32_t>((size_t)(w.plan.nblk + 1) * w.plan.Qp * 4 + 64);
    w.total = a.used();
    return w;
}
extern "C" size_t mf_topk_bf3_ws_bytes(int64_t Q, int64_t N, int d, int k) {
    if (Q <= 0 || N <= 0 || k <= 0 || !mf_width_ok(d)) return 0;
    return bf3_ws(nullptr, Q, N, d).total;
}

// -------------------------------------------------------------------- scan ----
struct Bf3Scan {
    int64_t Q, Qp;
    const unsigned short* plane;
    int64_t N;
    int NT, NTp, nblk;
    int bs, bpc, nvb;           // this launch: block stride of the sample, virtual blocks per workgroup, virtual blocks in all
    int xw;                     // distinct query-tile sets among the four waves
    int nvals;
    const uint32_t* exclW;      // [nblk + 1][Qp][4]
    float* gmax;                // PASS 0 out: [Qp][nvals]
    const float* thr;           // PASS 1 in
    const bf16x8* xfrag;
    uint32_t* cand;
    uint32_t* ovf_list;
    int32_t* ovf_cnt;
    int32_t* ovf;
    int abl;                    // lab knob (MF_BF3_ABL): 1 = no arithmetic, 2 = hits are not parked, 4 = no query fragments -- wrong results, for timing only
    unsigned long long* stamps; // lab: [pass][workgroup][4] s_memrealtime at entry / loop start / loop end / exit (NULL: off)
};

// The scans stage tiles with all eight waves: the tile geometry of mf_stream.h is written for four, so a tile's pieces are
// dealt to the waves explicitly here (piece j of a tile = 1 KiB of its LDS image: wave j % 8 stages it).
template <int D, int XT>
struct Bf3Lds {
    static constexpr int ROWB = D * 2;                        // bytes of a bf16 row
    static constexpr int TILEB = 32 * ROWB;
    static constexpr int PIECES = TILEB / 1024;               // 4 (d = 64), 8, 16
    static constexpr int PPW = (PIECES + BF3_WAVES - 1) / BF3_WAVES;    // DMA instructions per wave and tile (1, 1, 2)
    static constexpr int CPR = ROWB / 16;                     // 16-byte chunks per row
    static __device__ __forceinline__ int swz(int row) { return CPR >= 16 ? (row & 15) : ((row >> 1) & 7); }
    static constexpr int NS = 2 * BF3_BLOCK;                  // ring slots (one 32-row tile each): the block being scored and the block landing behind it
    static constexpr int RING = NS * TILEB;
    static constexpr int EX0 = RING;
    static constexpr int EXW = XT * 512;                      // exclusion words of one block: per query tile 32 queries x 4 tiles
    static constexpr int EXS = 2;                             // blocks in rotation per wave, like the tiles
    static constexpr int EXB = BF3_WAVES * EXS * EXW;
    static constexpr int DUMP0 = EX0 + EXB;                   // where the (zero) pieces of waves without a share of a short tile land
    static constexpr int BYTES = DUMP0 + 1024;
};

// PASS 0: block maxima of the sample; PASS 1: candidate lists of the whole catalog
template <int D, int XT, bool EXCL, int PASS>
__global__ __launch_bounds__(64 * BF3_WAVES, 2) void bf3_scan_kernel(Bf3Scan p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = Bf3Lds<D, XT>;
    constexpr int KS = D / 16;                               // MFMA steps per tile
    constexpr int NS = L::NS;
    const int lane = mf_lane(), c = lane & 31, h = lane >> 5;
    const int wave = mf_wave_id();
#ifdef MF_BF3_LAB
    unsigned long long* st = p.stamps ? p.stamps + ((size_t)PASS * 4096 + blockIdx.y * gridDim.x + blockIdx.x) * 4 : nullptr;
    if (st && threadIdx.x == 0) st[0] = __builtin_amdgcn_s_memrealtime();
#endif
    const int nsub = BF3_WAVES / p.xw, set = wave % p.xw, sub = wave / p.xw;   // waves of one set deal the chunk's tiles round-robin
    const int64_t xb0 = (int64_t)blockIdx.y * p.xw * XT * 32;               // first query of the workgroup
    const int64_t x0 = xb0 + (int64_t)set * XT * 32;                        // this wave's XT query tiles
    const int vb0 = blockIdx.x * p.bpc, vb1 = min(p.nvb, vb0 + p.bpc);      // virtual blocks of this workgroup
    // virtual tile v (0 .. nv) of the chunk is real tile (vb0 + v / 4) * bs * 4 + v % 4
    const int nv = (vb1 - vb0) * BF3_BLOCK;
    auto real_tile = [&](int v) { return (vb0 + (v >> 2)) * p.bs * BF3_BLOCK + (v & 3); };

    // tile source: a raw buffer descriptor over the rows from the chunk's first (hardware range check: rows past N arrive as
    // zeros), per-lane source offsets of this wave's pieces computed once (piece = 1 KiB of the tile's LDS image; lane l
    // writes 16 bytes at + 16 l and reads them from row r, chunk ch ^ swz(r) of the source tile)
    const int64_t row0 = (int64_t)real_tile(0) * 32;
    mf_rsrc_t trsrc;
    unsigned toff[L::PPW];
    {
        int64_t bytes = (p.N - row0) * (int64_t)L::ROWB;
        bytes = bytes < 0 ? 0 : (bytes > (int64_t)MF_SRD_MAX_BYTES ? (int64_t)MF_SRD_MAX_BYTES : bytes);
#if defined(__HIP_DEVICE_COMPILE__)
        trsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.plane + row0 * D), 0, (int)(unsigned)bytes, 0x00020000);
#else
        (void)trsrc;
#endif
#pragma unroll
        for (int q = 0; q < L::PPW; ++q) {
            const int piece = wave + q * BF3_WAVES;
            const int off = piece * 1024 + lane * 16;
            const int row = off / L::ROWB;
            const int ch = ((off % L::ROWB) >> 4) ^ L::swz(row);
            toff[q] = piece < L::PIECES ? (unsigned)(row * L::ROWB + ch * 16) : MF_SRD_DEAD;
            asm volatile("" : "+v"(toff[q]));
        }
    }
    // exclusion words: lane c (lower half-wave) fetches the 16 bytes = 4 tiles of its query in the block's row of the
    // [block][query][4] array -- a wave's 32 queries are 512 contiguous bytes.  The descriptor starts at the workgroup's first
    // block and first query (blockIdx only: provably uniform); the block comes in as the scalar offset
    mf_rsrc_t arsrc;
    uint32_t aoff = MF_SRD_DEAD;
    if (EXCL) {
        if (lane < 32) aoff = (uint32_t)((x0 - xb0 + c) * 16);
#if defined(__HIP_DEVICE_COMPILE__)
        const int64_t blk0 = (int64_t)vb0 * p.bs;
        const int64_t left = ((int64_t)(p.nblk + 1) - blk0) * p.Qp - xb0;      // 16-byte entries from the base to the end of the array
        const uint64_t ab = left > 0 ? (uint64_t)left * 16u : 0u;
        arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.exclW + (blk0 * p.Qp + xb0) * 4), 0,
                                                  (int)(ab > MF_SRD_MAX_BYTES ? MF_SRD_MAX_BYTES : ab), 0x00020000);
#else
        (void)arsrc;
#endif
    }
    (void)aoff;
    char* exl = smem + L::EX0 + wave * L::EXS * L::EXW;
    auto stage = [&](int v) {                                // (a tile past the chunk or the catalog: a dead scalar offset, zeros arrive)
        const int t = real_tile(v < nv ? v : 0);
        const bool live = v < nv && t < p.NT;
#if defined(__HIP_DEVICE_COMPILE__)
        char* slot = smem + (v % NS) * L::TILEB;
        const int soff = live ? (int)(((int64_t)t * 32 - row0) * L::ROWB) : (int)MF_SRD_DEAD;
#pragma unroll
        for (int q = 0; q < L::PPW; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(trsrc, (mf_lds_ptr)((wave + q * BF3_WAVES) < L::PIECES ? slot + (wave + q * BF3_WAVES) * 1024 : smem + L::DUMP0),
                                                     16, (int)toff[q], soff, 0, 0);
        if (EXCL && (v & 3) == 0) {
            char* dst = exl + ((v >> 2) % L::EXS) * L::EXW;
            const int boff = live ? (int)((int64_t)(v >> 2) * p.bs * p.Qp * 16) : (int)MF_SRD_DEAD;      // blocks past the workgroup's first
#pragma unroll
            for (int xt = 0; xt < XT; ++xt)
                if (lane < 32)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (mf_lds_ptr)(dst + xt * 512), 16, (int)(aoff + (uint32_t)(xt * 512)), boff, 0, 0);
        }
#else
        (void)live;
#endif
    };
    // Round 4: ONE meeting point per BLOCK of four tiles, not per tile.  The block's four tiles (and its exclusion words) were
    // staged while the previous block was scored; at the block's start every wave waits for its own DMA pieces, all meet, the
    // next block is staged into the slots just vacated, and the four tiles are scored WITHOUT a barrier between them: the waves
    // drift apart instead of arriving at the LDS port and the matrix pipe in lockstep eight times as often (per-tile barriers:
    // scan loop 19.8 us, seed 8.1; none at all, in the lab: 16.1 / 6.6).
#pragma unroll
    for (int j = 0; j < BF3_BLOCK; ++j) stage(j);

    // the queries' bf16 fragments (written once, in operand order, by the prep launch): coalesced 1 KiB loads
    bf16x8 xb[XT][KS];
    float thr[XT];
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) {
        const int64_t x = x0 + 32 * xt + c;
        const bf16x8* xf = p.xfrag + ((x0 / 32 + xt) * KS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) xb[xt][s] = (p.abl & 4) ? bf16x8{} : xf[s * 64];          // (padding queries: zero fragments, written by prep)
        thr[xt] = (PASS == 1 && x < p.Q) ? p.thr[x] : __builtin_inff();      // (bf3_bound_kernel: already nudged below the bound)
    }
    uint32_t slot_[XT][BF3_SLOTS];                            // PASS 1: this lane's hits per query tile, newest first (0xFFFFFFFF: free)
    float sval_[XT][BF3_SLOTS];                               // ... and their approximate scores minus thr (what the accumulator holds)
#pragma unroll
    for (int xt = 0; xt < XT; ++xt)
#pragma unroll
        for (int j = 0; j < BF3_SLOTS; ++j) { slot_[xt][j] = 0xFFFFFFFFu; sval_[xt][j] = 0.f; }
    // PASS 0: the running maximum of the lane's share of the current block -- as INTEGER bit patterns (v_max3_i32: two
    // elements per instruction, no canonicalising moves): for floats of either sign the integer maximum is the float
    // maximum whenever that is >= 0, and SOME element's value otherwise -- the bound only needs scores of distinct rows --
    // A lane's 16 rows of a tile that hold a row which must not count for its query (excluded, or past N) are left out of the
    // maxima whole (~4 % of the shares at 150 exclusions per query: the bound does not notice), instead of a mask select per element.
    int gmi[XT];
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) gmi[xt] = (int)0x80000000;
    const int tail_tile = (p.N & 31) ? p.NT - 1 : -1;        // its rows past N score 0: they are not rows
    const int tail_rows = (int)(p.N & 31);
    // Round 4: the epilogues were the scans' limit -- every vector instruction beside the MFMAs costs ~2.8 cycles of matrix
    // time (two waves per SIMD share its issue), and a 32 x 32 x 128 block is only 8 MFMAs = 256 cycles.  PASS 0 went from
    // ~60 instructions per block (mask select + v_med3 per element, accumulator zeroing) to ~14 (8 v_max3_i32 + the share's
    // mask test; its accumulators start from the inline constant 0): seed loop 9.5 -> 7.8 us.
    typedef int i32x16_t __attribute__((ext_vector_type(16)));
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#ifdef MF_BF3_LAB
    if (st && threadIdx.x == 0) { asm volatile("" : : "v"(xb[0][0]) : "memory"); st[1] = __builtin_amdgcn_s_memrealtime(); }
    if (st && threadIdx.x == 0 && p.stamps[(size_t)(2 * 4096 * 4 + 4096 * 8)] == 1ull) st[3] = __builtin_amdgcn_s_memtime();       // (... and at loop start, in slots 3 / 0)
#endif
    for (int v = 0; v < nv; ++v) {
        if ((v & (BF3_BLOCK - 1)) == 0) {
            mf_wait_vmcnt<0>();                              // this wave's pieces of block v / 4 have landed ...
            mf_block_barrier();                              // ... every wave's have; and everybody is done with block v / 4 - 1
            if (v + BF3_BLOCK < nv) {
#pragma unroll
                for (int j = 0; j < BF3_BLOCK; ++j) stage(v + BF3_BLOCK + j);
            }
        }
        const int t = real_tile(v);
        const bool mine = (v % nsub) == sub && t < p.NT && !(p.abl & 1);
        if (mine) {
            const char* slot = smem + (v % NS) * L::TILEB;
            constexpr int KH = D <= 128 ? KS : KS / 2;               // fragments held at a time (d = 256: half a row, read per query tile)
            bf16x8 afr[KH];
            const char* rowp = slot + c * L::ROWB;
            const int sw = L::swz(c);
            if constexpr (D <= 128) {
#pragma unroll
                for (int s = 0; s < KS; ++s) afr[s] = *reinterpret_cast<const bf16x8*>(rowp + (((2 * s + h) ^ sw) << 4));
            }
            const uint32_t tdead = t == tail_tile ? (~0u << tail_rows) : 0u;
            const unsigned row0t = (unsigned)t * 32u + 4u * (unsigned)h;     // element e of this lane is row row0t + (e & 3) + 8 (e >> 2)
            // epilogue of one query tile's 32 x 32 block
            auto epi = [&](int xt, const f32x16& acc) {
                if (PASS == 0) {
                    uint32_t dead = tdead;
                    if (EXCL) dead |= reinterpret_cast<const uint32_t*>(exl + ((v >> 2) % L::EXS) * L::EXW + xt * 512 + c * 16)[v & 3];
                    // (the whole vector is bit-cast, THEN indexed: a bit-cast of an extracted element is folded to element 0 by
                    // this compiler, the same family as the permlane swap fold in mf_common.h)
                    const i32x16_t ai = __builtin_bit_cast(i32x16_t, acc);
                    int g = (int)0x80000000;
#pragma unroll
                    for (int e = 0; e < 16; e += 2) g = max(max(g, ai[e]), ai[e + 1]);
                    // the lane's 16 rows of this tile (4 h + {0..3} + 8 j) hold a row that must not count: the tile's share is left out
                    g = ((dead >> (4 * h)) & 0x0F0F0F0Fu) != 0u ? (int)0x80000000 : g;
                    gmi[xt] = max(gmi[xt], g);
                } else {
                    // bit e: element e reaches the bound.  The accumulators START at -thr (the matrix core does the subtraction: a
                    // lane's 16 elements are one query's), so ONE instruction per element shifts the sign of score - thr in
                    // (v_alignbit), last element first; sign clear = hit.  (The extra rounding of the accumulation from -thr is
                    // inside eps' allowance for the accumulations, bf3_bound_kernel.)  Measured against it in round 4: 16 compares
                    // into wave masks + scalar "any" + a branch per element -- fewer vector instructions, 23 us instead of 17.
                    uint32_t hm = 0u;
                    typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
                    const u32x16_t ab = __builtin_bit_cast(u32x16_t, acc);
#pragma unroll
                    for (int e = 15; e >= 0; --e) hm = __builtin_amdgcn_alignbit(hm, ab[e], 31);
                    const uint32_t d4 = tdead >> (4 * h);            // rows past N (the last tile): they score 0, they are not rows
                    const uint32_t m = (d4 & 0xFu) | ((d4 >> 4) & 0xF0u) | ((d4 >> 8) & 0xF00u) | ((d4 >> 12) & 0xF000u);
                    hm = ~hm & 0xFFFFu & ~m;
                    // Hits are parked HERE, while the accumulator still holds the block: a slot keeps the row and its score - thr,
                    // so that the final kernel can cut the candidates a second time, against the k-th best approximate score of
                    // the WHOLE catalog (the seed's bound saw half of it): ~78 -> ~40 rows whose fp32 copies are gathered.
                    if (__any(hm != 0u) && !(p.abl & 2)) {
                        uint32_t w = hm;
                        uint32_t xword = 0u;
                        if (EXCL) xword = reinterpret_cast<const uint32_t*>(exl + ((v >> 2) % L::EXS) * L::EXW + xt * 512 + c * 16)[v & 3];
                        while (w) {                              // (1.3 hits per block of 1024 elements: rarely a second trip)
                            const int e = __builtin_ctz(w);
                            w &= w - 1;
                            const unsigned row = row0t + (unsigned)((e & 3) + 8 * (e >> 2));
                            // a row on the query's exclusion list is dropped here (the block's words are in LDS): the final kernel's
                            // second cut then needs no exclusion words -- one dependent memory round trip less in its chain
                            if (EXCL && ((xword >> (row & 31u)) & 1u)) continue;
                            // element e of the accumulator, e differing from lane to lane: a select tree over the 16 registers
                            // (v_cndmask by hand: the compiler rewrites a select between two elements of a vector into a dynamic vector
                            // index and answers it through scratch memory.  The accumulator was last written long before -- the sixteen
                            // v_alignbit above have read it -- so no MFMA hazard is in reach of these instructions)
                            const unsigned long long m8 = __ballot((e & 8) != 0), m4 = __ballot((e & 4) != 0), m2 = __ballot((e & 2) != 0),
                                                     m1 = __ballot((e & 1) != 0);
                            auto pick = [](float lo, float hi, unsigned long long msk) {
                                float o;
                                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(o) : "v"(lo), "v"(hi), "s"(msk));
                                return o;
                            };
                            float t8[8], t4[4], t2[2];
#pragma unroll
                            for (int i = 0; i < 8; ++i) t8[i] = pick(acc[i], acc[8 + i], m8);
#pragma unroll
                            for (int i = 0; i < 4; ++i) t4[i] = pick(t8[i], t8[4 + i], m4);
#pragma unroll
                            for (int i = 0; i < 2; ++i) t2[i] = pick(t4[i], t4[2 + i], m2);
                            const float val = pick(t2[0], t2[1], m1);
                            if (slot_[xt][BF3_SLOTS - 1] == 0xFFFFFFFFu) {      // a free slot left
#pragma unroll
                                for (int j = BF3_SLOTS - 1; j > 0; --j) { slot_[xt][j] = slot_[xt][j - 1]; sval_[xt][j] = sval_[xt][j - 1]; }
                                slot_[xt][0] = row;
                                sval_[xt][0] = val;
                            } else {                              // beyond the slots: the query's overflow list (an atomic: rare)
                                const int64_t x = x0 + 32 * xt + c;
                                const int at = atomicAdd(&p.ovf_cnt[x], 1);
                                if (at < BF3_OVF) p.ovf_list[x * BF3_OVF + at] = row;
                                else p.ovf[x] = 1;
                            }
                        }
                    }
                }
            };
            if constexpr (D <= 128 && PASS == 1 && XT > 1) {
                // the full scan at XT = 4 sits at the 256-register line: ONE accumulator set (its 16-instruction epilogue is short,
                // and the SIMD's other wave has matrix work for the gap)
#pragma unroll
                for (int xt = 0; xt < XT; ++xt) {
                    f32x16 acc;
                    float nt = -thr[xt];
                    asm volatile("" : "+v"(nt));       // (per tile: a loop-invariant 16-register splat per query tile would be hoisted and kept)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = nt;
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[s], xb[xt][s], acc, 0, 0, 0);
                    epi(xt, acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (D <= 128) {
                // two accumulator sets: the epilogue of query tile xt - 1 is threaded between the MFMAs of query tile xt
                f32x16 accA, accB;
#pragma unroll
                for (int xt = 0; xt < XT; ++xt) {
                    f32x16& cur = (xt & 1) ? accB : accA;
                    const f32x16& prev = (xt & 1) ? accA : accB;
                    if (PASS == 1) {
                        float nt = -thr[xt];
                        asm volatile("" : "+v"(nt));
#pragma unroll
                        for (int e = 0; e < 16; ++e) cur[e] = nt;
#pragma unroll
                        for (int s = 0; s < KS; ++s) cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[s], xb[xt][s], cur, 0, 0, 0);
                    } else {
                        cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[0], xb[xt][0], zero16, 0, 0, 0);
#pragma unroll
                        for (int s = 1; s < KS; ++s) cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[s], xb[xt][s], cur, 0, 0, 0);
                    }
                    if (xt > 0) {
                        epi(xt - 1, prev);
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, (PASS == 1 ? 32 : 16) / KS, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                epi(XT - 1, ((XT - 1) & 1) ? accB : accA);
            } else {
                // d = 256: the fragments alone take half the registers -- one accumulator set, the partner wave of the SIMD
                // covers the epilogues
#pragma unroll
                for (int xt = 0; xt < XT; ++xt) {
                    f32x16 acc;
                    float nt = PASS == 1 ? -thr[xt] : 0.f;
                    if (PASS == 1) asm volatile("" : "+v"(nt));
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = nt;
#pragma unroll
                    for (int s0 = 0; s0 < KS; s0 += KH) {
#pragma unroll
                        for (int s = 0; s < KH; ++s) afr[s] = *reinterpret_cast<const bf16x8*>(rowp + (((2 * (s0 + s) + h) ^ sw) << 4));
#pragma unroll
                        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[s], xb[xt][s0 + s], acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);        // (the next half's fragments are not read ahead: registers)
                    }
                    epi(xt, acc);
                }
            }
        }
        if (PASS == 0 && (v & 3) == 3) {
            // one block done: the maximum per (query, lane half) over THIS wave's tiles of the block (the waves of a set deal
            // the tiles among them: each writes its own value -- scores of distinct rows either way); -inf: no row in sight
            const int vb = vb0 + (v >> 2);
#pragma unroll
            for (int xt = 0; xt < XT; ++xt) {
                const int64_t x = x0 + 32 * xt + c;
                const float gmf = gmi[xt] == (int)0x80000000 ? -__builtin_inff() : __builtin_bit_cast(float, gmi[xt]);
                if (x < p.Q) p.gmax[x * p.nvals + ((int64_t)vb * 2 + h) * nsub + sub] = gmf;
                gmi[xt] = (int)0x80000000;
            }
        }
    }
    mf_wait_vmcnt<0>();                                      // nothing of this workgroup may still be on its way into LDS when it ends
#ifdef MF_BF3_LAB
    if (st && threadIdx.x == 0) st[2] = __builtin_amdgcn_s_memrealtime();
    if (st && threadIdx.x == 0 && p.stamps[(size_t)(2 * 4096 * 4 + 4096 * 8)] == 1ull) st[0] = __builtin_amdgcn_s_memtime();       // (clock probe: shader cycles at loop end ...)
#endif
    if (PASS == 1) {
        // this lane's slots of every query tile: 16 bytes each, a wave's 32 queries x 2 halves contiguous (no counters, no atomics)
#pragma unroll
        for (int xt = 0; xt < XT; ++xt) {
            const int64_t x = x0 + 32 * xt + c;
            *reinterpret_cast<uint4*>(p.cand + ((((int64_t)blockIdx.x * nsub + sub) * p.Qp + x) * 2 + h) * (2 * BF3_SLOTS)) =
                uint4{slot_[xt][0], slot_[xt][1], __builtin_bit_cast(uint32_t, sval_[xt][0]), __builtin_bit_cast(uint32_t, sval_[xt][1])};
        }
    }
#ifdef MF_BF3_LAB
    if (st && threadIdx.x == 0 && p.stamps[(size_t)(2 * 4096 * 4 + 4096 * 8)] != 1ull) { mf_wait_vmcnt<0>(); st[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// ------------------------------------------------------------------- bound ----
// one wave per query: thr = (a lower bound of the k-th largest group maximum) - 2 eps.  Every lane reduces its
// share of the maxima to its two best; the k-th largest of those 128 values (all scores of distinct rows) is at
// most a few ranks below the exact k-th largest, at a tenth of the cost of searching all of them.
__global__ __launch_bounds__(64) void bf3_bound_kernel(const float* __restrict__ gmax, int nvals, int k, const float* __restrict__ q,
                                                       int d, const float* __restrict__ ymax2, float* __restrict__ thr,
                                                       float* __restrict__ eps_out) {
    const int64_t r = blockIdx.x;
    const int lane = mf_lane();
    const unsigned ninf = mf_orderable(-__builtin_inff());
    float ss = 0.f;                                          // (the query's row and the largest norm are asked for with the maxima: one round trip)
    for (int i = lane; i < d; i += 64) ss = __builtin_fmaf(q[r * d + i], q[r * d + i], ss);
    const float ym2 = ymax2[0];
    unsigned m1 = 0u, m2 = 0u;                               // 0: "no value" (ranks below -inf)
    for (int i0 = 0; i0 < nvals; i0 += 64 * 8) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + lane + 64 * j;
            f[j] = i < nvals ? gmax[r * nvals + i] : -__builtin_inff();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned v = (i0 + lane + 64 * j) < nvals ? mf_orderable(f[j]) : 0u;
            m2 = max(m2, min(m1, v));
            m1 = max(m1, v);
        }
    }
    unsigned th = 0u;
    for (int b = 31; b >= 0; --b) {                          // largest th with #{v >= th} >= k
        const unsigned cnd = th | (1u << b);
        const int cge = __popcll(__ballot(m1 >= cnd)) + __popcll(__ballot(m2 >= cnd));
        if (cge >= k) th = cnd;
    }
    ss = mf_wave_sum(ss);
    if (lane == 0) {
        const float c = 1.01f * (0x1p-7f + 0x1p-16f + (float)d * 0x1p-22f);
        const float eps = c * sqrtf(ss) * sqrtf(ym2);
        // fewer than k rows in sight (or a NaN bound): everything is a candidate
        // the scan tests "score > thr": thr sits strictly below the bound (one part in 2^22, and past zero)
        float t = (th <= ninf) ? -__builtin_inff() : mf_unorderable(th) - 2.f * eps;
        t = t - fabsf(t) * 0x1p-22f - 1e-37f;
        if (!(t == t)) t = -__builtin_inff();
        thr[r] = t;
        eps_out[r] = eps;
    }
}

// ------------------------------------------------------------------- final ----
struct Bf3Final {
    int64_t Q;
    const float* q;
    const float* items;
    int64_t N;
    int d, k, NT, NTp;
    int64_t Qp;
    int nlists;                 // candidate blocks per query: scan workgroups x waves sharing a query tile
    const uint32_t* cand;       // [nlists][Qp][2][2 BF3_SLOTS]: rows, then their scores - thr
    const float* eps;           // [Qp]
    const uint32_t* ovf_list;
    const int32_t* ovf_cnt;
    const int32_t* ovf;
    const uint32_t* exclW;      // NULL: nothing excluded
    int64_t idx_base;
    float* out_scores;
    int64_t* out_idx;
    int abl;                    // lab knob (MF_BF3_ABL): 8 = no rescoring, 16 = no selection -- wrong results, for timing only
    unsigned long long* dbg;    // lab: [0] += candidates gathered, [1] += queries, [2] += candidates rescored (NULL: off)
    unsigned long long* stamps; // lab: [query][8] s_memrealtime at the phases of the final kernel (NULL: off)
};

// the canonical chain (mf_dot_chain's order) with up to 128 floats of the row in flight at a time
template <int D>
__device__ __forceinline__ float bf3_exact_dot(const float* xq, const float* __restrict__ row) {
    constexpr int B = D < 128 ? D : 128;
    float acc = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < D; g0 += B) {
        f32x4 y[B / 4];
#pragma unroll
        for (int j = 0; j < B / 4; ++j) y[j] = *reinterpret_cast<const f32x4*>(row + g0 + 4 * j);
#pragma unroll
        for (int g = 0; g < B; g += 8)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc = __builtin_fmaf(xq[g0 + g + t], y[g / 4][t], acc);
                acc = __builtin_fmaf(xq[g0 + g + 4 + t], y[g / 4 + 1][t], acc);
            }
    }
    return acc;
}

// The second cut (one wave, its own list in LDS: {orderable(score - thr) << 32 | row}, n <= 64 VPL entries).  Every row whose
// approximate score a reaches thr is in the list (that is what the full scan guarantees; rows on the exclusion list were
// dropped by it), so the k-th largest a in the list, tau', is the k-th largest a of the whole admissible catalog: k distinct
// rows have exact scores >= tau' - eps, and a row of the exact top k has a >= tau' - 2 eps.  The seed's bound saw half of the
// catalog; this one sees all of it -- the list shrinks from ~78 to ~40 rows before any fp32 row is fetched.  Rows of unknown
// score (overflow list; thr = -inf) are kept and not counted.  Leaves the survivors' ROWS at the list's front; returns their
// number.
template <int VPL>
__device__ __forceinline__ int bf3_second_cut(unsigned long long* keys, int n, int k, float eps) {
    constexpr unsigned ORD_INF = 0xFF800000u;
    const int lane = mf_lane();
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long kk[VPL];
    unsigned ov[VPL];                                        // orderable scores that count (0: unknown / none)
    int known = 0;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = lane + 64 * j;
        kk[j] = i < n ? keys[i] : 0ull;
        const unsigned o = (unsigned)(kk[j] >> 32);
        ov[j] = (kk[j] != 0ull && o != ORD_INF) ? o : 0u;
        known += __popcll(__ballot(ov[j] != 0u));
    }
    unsigned cut = 0u;                                       // (fewer than k known scores: nothing can be cut)
    if (known >= k) {
        unsigned th = 0u;
        for (int b = 31; b >= 0; --b) {                      // largest th with #{ov >= th} >= k
            const unsigned cnd = th | (1u << b);
            int cge = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) cge += __popcll(__ballot(ov[j] >= cnd));
            if (cge >= k) th = cnd;
        }
        float t = mf_unorderable(th) - 2.f * eps;
        t = t - fabsf(t) * 0x1p-22f - 1e-37f;                // strictly below the bound
        cut = (t == t) ? mf_orderable(t) : 0u;
    }
    mf_row_topk_sync<true>();                                // (every entry is in a register by now)
    int n2 = 0;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const bool keep = kk[j] != 0ull && (unsigned)(kk[j] >> 32) >= cut;
        const unsigned long long bal = __ballot(keep);
        if (keep) keys[n2 + __popcll(bal & below)] = kk[j] & 0xFFFFFFFFull;      // (row ids for now)
        n2 += __popcll(bal);
    }
    mf_row_topk_sync<true>();
    return n2;
}

// One wave per query, FOUR queries (waves) per workgroup.  Round 4:
//  * the scan's slot blocks are laid out [block][query][32 bytes]: one query's share is 244 pieces on 244 cache lines, each
//    line shared with three neighbours -- a wave per query fetched 32 MB of lines for 8 MB of slots (5.7 of its 13.6 us).  The
//    four waves of a workgroup take four NEIGHBOURING queries and read their blocks together: 128 contiguous bytes per
//    block, every byte used once, the entries dealt to the queries' LDS lists with LDS atomics;
//  * a second cut (below) halves the list before any fp32 row is fetched;
//  * the survivors' fp32 rows are GATHERED BY LDS-DMA -- `global_load_lds_dwordx4` takes a 64-bit address per lane, so one
//    instruction brings 64 / (D / 4) whole rows (1 KiB) and all instructions of a round are in flight together: ONE memory
//    round trip per round of 64 candidates (round 3 went through registers, a chunk of 64 floats at a time).  The 16-byte
//    chunks of candidate s land XOR-swizzled by s & 15 (on the SOURCE address: the DMA's destination is lane-linear), so the
//    16 lanes of a ds_read_b128 phase -- 16 candidates, the same chunk -- hit 64 different banks.  Then every lane runs the
//    canonical chain over ITS candidate's row from LDS.
// After the gather the waves never meet again: every later synchronisation is wave-local.
static constexpr int BF3_FQ = 4;                        // queries (waves) per workgroup of the final kernel
template <int D>
struct Bf3FinalGeom {
    static constexpr int CPR = D / 4;                   // 16-byte chunks of an fp32 row: 16, 32, 64
    static constexpr int RPI = 64 / CPR;                // rows per DMA instruction: 4, 2, 1
    static constexpr int RB = D <= 128 ? 64 : 32;       // candidates per round (32 KiB of rows at d >= 128, 16 at d = 64)
    static constexpr int NI = RB / RPI;                 // DMA instructions of a full round
};
template <int D>
__global__ __launch_bounds__(64 * BF3_FQ) void bf3_final_kernel(Bf3Final p) {
    using G = Bf3FinalGeom<D>;
    extern __shared__ __attribute__((aligned(1024))) char fsm[];
    const int lane = mf_lane(), wave = mf_wave_id();
    // per wave: rows [RB][D] floats | keys [BF3_CAND] | win [64] | sorted [64] | xq [D]; behind them the four list lengths
    constexpr int PER_WAVE = G::RB * D * 4 + BF3_CAND * 8 + 64 * 8 * 2 + D * 4;
    char* mine_ = fsm + wave * PER_WAVE;
    float* rows_lds = reinterpret_cast<float*>(mine_);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(mine_ + G::RB * D * 4);
    unsigned long long* win = keys + BF3_CAND;
    unsigned long long* sorted = win + 64;
    float* xq = reinterpret_cast<float*>(sorted + 64);
    int* lcnt = reinterpret_cast<int*>(fsm + BF3_FQ * PER_WAVE);          // [BF3_FQ] entries appended to the queries' lists
    auto keys_of = [&](int qq) { return reinterpret_cast<unsigned long long*>(fsm + qq * PER_WAVE + G::RB * D * 4); };
    const int64_t q0 = (int64_t)blockIdx.x * BF3_FQ;
    const int64_t r = q0 + wave;
    const bool real = r < p.Q;
    const unsigned long long below = (1ull << lane) - 1ull;
    int n = 0;
#ifdef MF_BF3_LAB
    unsigned long long* st = (p.stamps && real && r < 4096) ? p.stamps + (size_t)2 * 4096 * 4 + r * 8 : nullptr;
#define BF3_STAMP(i) do { if (st && lane == 0) st[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BF3_STAMP(i) do { } while (0)
#endif
    BF3_STAMP(0);
    // (the wave's scalars are asked for before the slot records: they travel under the same round trip)
    const float* xq_g = p.q + (real ? r : 0) * D;
    const int novf = real ? p.ovf_cnt[r] : 0;
    const float eps = real ? p.eps[r] : 0.f;
    const int ovf_g = real ? p.ovf[r] : 0;
    float xv[(D + 63) / 64];
#pragma unroll
    for (int j = 0; j < (D + 63) / 64; ++j) xv[j] = (real && lane + 64 * j < D) ? xq_g[lane + 64 * j] : 0.f;
    if (threadIdx.x < BF3_FQ) lcnt[threadIdx.x] = 0;
    __syncthreads();
    static_assert(BF3_SLOTS == 2 && BF3_FQ == 4, "a block's share of the workgroup = 4 queries x 2 lane halves x 16 bytes = 128 bytes");
    constexpr unsigned ORD_INF = 0xFF800000u;                // mf_orderable(+inf): "score unknown" (overflow-list entries)
    // the workgroup's records: block j, query q0 + qq, lane half hh -> 16 bytes {row, row, score, score}; record index
    // 8 j + 2 qq + hh is also its position in memory (Qp is a multiple of 32: the four queries never straddle a row end)
    {
        const int nrec = p.nlists * 8;
        constexpr int PF = 8;                                  // records in flight per thread (244 blocks: one trip)
        for (int i0 = 0; i0 < nrec; i0 += 64 * BF3_FQ * PF) {
            uint4 v[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int idx = i0 + j * 64 * BF3_FQ + (int)threadIdx.x;
                v[j] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u};
                if (idx < nrec) v[j] = *reinterpret_cast<const uint4*>(p.cand + (((int64_t)(idx >> 3) * p.Qp + q0) * 2) * (2 * BF3_SLOTS) + (idx & 7) * 4);
            }
            // (a thread's records all belong to ONE of the four queries -- record index mod 8 is thread index mod 8 --: it counts its
            // occupied slots, reserves them with ONE LDS atomic and writes them; sixteen conditional atomics in a row, each waited
            // for, were a microsecond of dependent LDS round trips)
            const int qq = ((int)threadIdx.x >> 1) & 3;
            int mine_n = 0;
#pragma unroll
            for (int j = 0; j < PF; ++j) mine_n += (v[j].x != 0xFFFFFFFFu ? 1 : 0) + (v[j].y != 0xFFFFFFFFu ? 1 : 0);
            int at = mine_n ? atomicAdd(&lcnt[qq], mine_n) : 0;
            unsigned long long* kq = keys_of(qq);
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const uint32_t rw[2] = {v[j].x, v[j].y}, vl[2] = {v[j].z, v[j].w};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (rw[t] != 0xFFFFFFFFu) {
                        // (orderable score - thr above, the row below; a non-finite score -- thr = -inf: fewer than k rows were
                        // in sight of the seed -- counts as unknown)
                        const float a = __builtin_bit_cast(float, vl[t]);
                        const unsigned oa = (a - a == 0.f) ? mf_orderable(a) : ORD_INF;
                        if (at < BF3_CAND) kq[at] = ((unsigned long long)oa << 32) | (unsigned long long)rw[t];
                        ++at;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < (D + 63) / 64; ++j)
        if (lane + 64 * j < D) xq[lane + 64 * j] = xv[j];
    bool overflow = real && (ovf_g != 0 || novf > BF3_OVF);
    __syncthreads();                                         // the four lists are complete; from here on every wave is on its own
    if (!real) return;
    n = lcnt[wave];
    for (int i0 = 0; i0 < min(novf, BF3_OVF); i0 += 64) {
        const bool ok = i0 + lane < novf;
        const uint32_t row = ok ? p.ovf_list[r * BF3_OVF + i0 + lane] : 0u;
        const unsigned long long bal = __ballot(ok);
        const int at = n + __popcll(bal & below);
        if (ok && at < BF3_CAND) keys[at] = ((unsigned long long)ORD_INF << 32) | (unsigned long long)row;
        n += __popcll(bal);
    }
    overflow = overflow || n > BF3_CAND;
    if (p.dbg && lane == 0) { atomicAdd(p.dbg, (unsigned long long)n); atomicAdd(p.dbg + 1, 1ull); }
    mf_row_topk_sync<true>();
    BF3_STAMP(1);
    if (!overflow && n > 0) {
        int n2;
        if (n <= 64) n2 = bf3_second_cut<1>(keys, n, p.k, eps);
        else if (n <= 128) n2 = bf3_second_cut<2>(keys, n, p.k, eps);
        else n2 = bf3_second_cut<BF3_CAND / 64>(keys, n, p.k, eps);
        if (p.dbg && lane == 0) atomicAdd(p.dbg + 2, (unsigned long long)n2);
        n = n2;
    }
    BF3_STAMP(2);
    int m = 0;
    if (!overflow) {
        // (candidates on the query's exclusion list were dropped with the second cut: the full scan does not look at exclusions)
        const int pz = lane % G::CPR, sub = lane / G::CPR;   // this lane's piece of a DMA instruction: chunk position, row among RPI
        const int sl = lane & (G::RB - 1);                   // the candidate slot this lane scores (d = 256: the lower half-wave)
        const float* row_l = rows_lds + sl * D;
        const int sw = sl & 15;
#ifdef MF_BF3_LAB
        const int rounds_cap = (p.abl & 8) ? 0 : (1 << 30);
#else
        constexpr int rounds_cap = 1 << 30;
#endif
        for (int base = 0; base < n && base < rounds_cap; base += G::RB) {
            const int nr = min(G::RB, n - base);
            const bool have = lane < nr;
            const unsigned row = have ? (unsigned)keys[base + lane] : 0u;          // (LDS reads BEFORE the first DMA is issued)
            __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the row ids have arrived
#pragma unroll
            for (int t = 0; t < G::NI; ++t) {
                if (t * G::RPI < nr) {                       // (wave-uniform)
                    // the candidate this lane fetches a piece of: slot RPI t + sub (compile-time lanes: v_readlane, no LDS traffic
                    // between the DMAs -- the compiler drains vmcnt before any LDS access it cannot tell apart from their target)
                    unsigned rr = (unsigned)__builtin_amdgcn_readlane((int)row, G::RPI * t);
#pragma unroll
                    for (int j = 1; j < G::RPI; ++j) {
                        const unsigned rj = (unsigned)__builtin_amdgcn_readlane((int)row, G::RPI * t + j);
                        rr = sub == j ? rj : rr;
                    }
                    const int ch = pz ^ ((G::RPI * t + sub) & 15);
                    const float* src = p.items + (int64_t)rr * D + 4 * ch;
                    __builtin_amdgcn_global_load_lds((mf_glb_ptr)src, (mf_lds_ptr)(rows_lds + t * 256), 16, 0, 0);
                }
            }
            // (a REAL s_waitcnt, not inline asm: the compiler's own counter tracking must see it, or it assumes the loads above
            // still pending at the loop's back edge and drains vmcnt in front of every DMA of the next round)
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
            asm volatile("" ::: "memory");
            if (base == 0) BF3_STAMP(3);
            float acc = 0.f;
#pragma unroll 4
            for (int g = 0; g < D / 8; ++g) {                // k order of mf_dot_chain
                const f32x4 a = *reinterpret_cast<const f32x4*>(row_l + (((2 * g) ^ sw) << 2));
                const f32x4 b = *reinterpret_cast<const f32x4*>(row_l + (((2 * g + 1) ^ sw) << 2));
                const f32x4 xa = *reinterpret_cast<const f32x4*>(xq + 8 * g), xb = *reinterpret_cast<const f32x4*>(xq + 8 * g + 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc = __builtin_fmaf(xa[t], a[t], acc);
                    acc = __builtin_fmaf(xb[t], b[t], acc);
                }
            }
            if (have) keys[base + lane] = mf_key_retrieval(acc, row);
        }
        mf_row_topk_sync<true>();
        BF3_STAMP(4);
#ifdef MF_BF3_LAB
        if (p.abl & 16) n = 0;
#endif
        if (n <= 64) m = mf_row_topk<1, true>(keys, n, p.k, win, sorted);
        else if (n <= 256) m = mf_row_topk<4, true>(keys, n, p.k, win, sorted);
        else m = mf_row_topk<BF3_CAND / 64, true>(keys, n, p.k, win, sorted);
    } else {
        // the whole catalog by the exact chain, 64 rows a round; the winners so far ride along in win[]
        int carry = 0;
        for (int64_t base = 0; base < p.N; base += 64) {
            const int64_t row = base + lane;
            unsigned long long v0 = 0ull;
            if (row < p.N) {
                const bool ex = p.exclW && ((p.exclW[((row >> 7) * p.Qp + r) * 4 + ((row >> 5) & 3)] >> (row & 31)) & 1u);
                if (!ex) v0 = mf_key_retrieval(bf3_exact_dot<D>(xq, p.items + row * D), (unsigned)row);
            }
            const unsigned long long v1 = lane < carry ? win[lane] : 0ull;
            mf_row_topk_sync<true>();
            const int have = __popcll(__ballot(v0 != 0ull)) + carry;
            unsigned long long tau = 1ull;
            if (have > p.k) {
                unsigned long long th = 0ull;
                for (int b = 63; b >= 0; --b) {                // largest th with #{key >= th} >= k (keys are unique)
                    const unsigned long long cnd = th | (1ull << b);
                    const int cge = __popcll(__ballot(v0 >= cnd)) + __popcll(__ballot(v1 >= cnd));
                    if (cge >= p.k) th = cnd;
                }
                tau = th;
            }
            int pos = 0;
            {
                const bool w0 = v0 != 0ull && v0 >= tau;
                const unsigned long long m0 = __ballot(w0);
                if (w0) win[__popcll(m0 & below)] = v0;
                pos = __popcll(m0);
                const bool w1 = v1 != 0ull && v1 >= tau;
                const unsigned long long m1 = __ballot(w1);
                if (w1) win[pos + __popcll(m1 & below)] = v1;
                pos += __popcll(m1);
            }
            carry = pos;
            mf_row_topk_sync<true>();
        }
        m = carry;
        if (lane < m) {
            const unsigned long long mine = win[lane];
            int rk = 0;
            for (int qq = 0; qq < m; ++qq) rk += win[qq] > mine ? 1 : 0;
            sorted[rk] = mine;
        }
        mf_row_topk_sync<true>();
    }
    BF3_STAMP(5);
    if (lane < p.k) {
        if (lane < m) {
            p.out_scores[r * p.k + lane] = mf_key_retrieval_score(sorted[lane]);
            p.out_idx[r * p.k + lane] = p.idx_base + (int64_t)mf_key_retrieval_col(sorted[lane]);
        } else {
            p.out_scores[r * p.k + lane] = -INFINITY;
            p.out_idx[r * p.k + lane] = -1;
        }
    }
}

// Prep, one workgroup per FOUR queries: their bf16 fragments in MFMA operand order (zeros for padding queries), their
// candidate counters, and -- with exclusion lists -- their bit words through an LDS window (no memset, no global atomics),
// written out as [4-tile block][query][4 words]: a scan wave's 32 queries x one block are 512 contiguous bytes (round 3 kept a
// row of words per query: that DMA touched 32 cache lines for 512 bytes).  The four lists are one contiguous range of the CSR
// array: all 256 threads stride over it with their loads in flight together; a workgroup writes 64-byte pieces, its
// neighbours the rest of the line.
static constexpr int BF3_PREP_Q = 4;            // queries per workgroup
static constexpr int BF3_PREP_W = 2048;         // words per query in a window (65,536 catalog rows; longer catalogs: more windows)
__global__ __launch_bounds__(256) void bf3_prep_kernel(const int64_t* __restrict__ excl_off, const int64_t* __restrict__ excl_idx,
                                                       int64_t idx_base, int64_t Q, int64_t Qp, int64_t N, int nblk,
                                                       uint32_t* __restrict__ exclW, const float* __restrict__ q, int d,
                                                       bf16x8* __restrict__ xfrag, int32_t* __restrict__ ovf_cnt, int32_t* __restrict__ ovf) {
    __shared__ __attribute__((aligned(16))) uint32_t win[BF3_PREP_Q][BF3_PREP_W];
    const int64_t q0 = (int64_t)blockIdx.x * BF3_PREP_Q;
    const int tid = threadIdx.x;
    // entry (query j, step s, lane half h): k = 16 s + 8 h .. + 7 of query q0 + j, at its place in the 32-query tile's block
    for (int e = tid; e < BF3_PREP_Q * (d / 8); e += 256) {
        const int j = e & (BF3_PREP_Q - 1), sh = e / BF3_PREP_Q, s = sh >> 1, h = sh & 1;
        const int64_t r = q0 + j;
        bf16x8 f = {};
        if (r < Q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(q + r * d + 16 * s + 8 * h);
            const f32x4 b = *reinterpret_cast<const f32x4*>(q + r * d + 16 * s + 8 * h + 4);
            f = bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
        }
        xfrag[((r >> 5) * (d / 16) + s) * 64 + h * 32 + (r & 31)] = f;
    }
    if (tid < BF3_PREP_Q) { ovf_cnt[q0 + tid] = 0; ovf[q0 + tid] = 0; }
    if (!exclW) return;
    int64_t off[BF3_PREP_Q + 1];
#pragma unroll
    for (int j = 0; j <= BF3_PREP_Q; ++j) off[j] = excl_off[min(q0 + j, Q)];      // (padding queries: empty lists)
    const int nwords = (nblk + 1) * BF3_BLOCK;
    for (int w0 = 0; w0 < nwords; w0 += BF3_PREP_W) {
        const int nw = min(BF3_PREP_W, nwords - w0);
        for (int i = tid; i < BF3_PREP_Q * BF3_PREP_W / 4; i += 256) reinterpret_cast<uint4*>(&win[0][0])[i] = uint4{0u, 0u, 0u, 0u};
        __syncthreads();
        for (int64_t eb = off[0]; eb < off[BF3_PREP_Q]; eb += 256 * 4) {
            int64_t y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t e = eb + j * 256 + tid;
                y[j] = e < off[BF3_PREP_Q] ? excl_idx[e] - idx_base : -1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t e = eb + j * 256 + tid;
                const int64_t wd = (y[j] >> 5) - w0;
                if (y[j] >= 0 && y[j] < N && wd >= 0 && wd < nw) {
                    int qq = 0;                              // the list the entry belongs to (empty lists share a boundary: the last wins)
#pragma unroll
                    for (int t = 1; t < BF3_PREP_Q; ++t) qq += off[t] <= e ? 1 : 0;
                    atomicOr(&win[qq][wd], 1u << (y[j] & 31));
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < BF3_PREP_Q * (nw / 4); i += 256) {
            const int qq = i & (BF3_PREP_Q - 1), bb = i / BF3_PREP_Q;
            *reinterpret_cast<uint4*>(exclW + (((int64_t)(w0 / 4) + bb) * Qp + q0 + qq) * 4) = *reinterpret_cast<const uint4*>(&win[qq][4 * bb]);
        }
        __syncthreads();
    }
}

template <int D, int XT, bool EXCL, int PASS>
static void bf3_launch_scan(int nwg, int gy, const Bf3Scan& sp, hipStream_t s) {
    auto fn = bf3_scan_kernel<D, XT, EXCL, PASS>;
    const int bytes = Bf3Lds<D, XT>::BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr_set = true;
    }
    fn<<<dim3((unsigned)nwg, (unsigned)gy), 64 * BF3_WAVES, bytes, s>>>(sp);
}
template <int D, int XT>
static void bf3_run_xt(const Bf3Ws& w, Bf3Scan sp, bool excl, int k, const float* q, const float* ymax2, hipStream_t s) {
    const Bf3Plan& pl = w.plan;
    sp.bs = pl.bs; sp.bpc = pl.bpc_seed; sp.nvb = pl.nvb_seed;
    if (excl) bf3_launch_scan<D, XT, true, 0>(pl.nwg_seed, pl.gy, sp, s); else bf3_launch_scan<D, XT, false, 0>(pl.nwg_seed, pl.gy, sp, s);
    bf3_bound_kernel<<<dim3((unsigned)sp.Q), 64, 0, s>>>(w.gmax, pl.nvals, k, q, D, ymax2, w.thr, w.eps);
    sp.bs = 1; sp.bpc = pl.bpc; sp.nvb = pl.nblk;
    // (the full scan stages the exclusion words too: excluded rows are dropped where they are found)
    if (excl) bf3_launch_scan<D, XT, true, 1>(pl.nwg, pl.gy, sp, s); else bf3_launch_scan<D, XT, false, 1>(pl.nwg, pl.gy, sp, s);
}
template <int D>
static void bf3_run(const Bf3Ws& w, const Bf3Scan& sp, bool excl, int k, const float* q, const float* ymax2, hipStream_t s) {
    if (w.plan.XT == 1) bf3_run_xt<D, 1>(w, sp, excl, k, q, ymax2, s);
    else bf3_run_xt<D, (D >= 256 ? 2 : 4)>(w, sp, excl, k, q, ymax2, s);
}

// lab: device counters of the final kernel (candidates gathered, queries); tools/lab/bf3_probe.py
static unsigned long long* g_bf3_dbg = nullptr;
static unsigned long long* g_bf3_stamps = nullptr;       // lab: [2 passes][4096 workgroups][4], then [4096 queries][8] of the final kernel
// lab: out = host buffer of (2 x 4096 x 4 + 4096 x 8) stamps of the LAST search (100 MHz ticks); enable = 1 allocates and switches them on
extern "C" int mf_probe_bf3_stamps(unsigned long long* out, int enable) {
    const size_t bytes = ((size_t)2 * 4096 * 4 + (size_t)4096 * 8 + 8) * 8;       // (+ a mode word: 1 = clock probe)
    if (g_bf3_stamps && enable >= 1) { const unsigned long long mode = enable == 2 ? 1ull : 0ull; (void)hipMemcpy(g_bf3_stamps + (bytes / 8 - 8), &mode, 8, hipMemcpyHostToDevice); }
    if (!g_bf3_stamps && enable) {
        if (hipMalloc(reinterpret_cast<void**>(&g_bf3_stamps), bytes) != hipSuccess) return -1;
        (void)hipMemset(g_bf3_stamps, 0, bytes);
    }
    (void)hipDeviceSynchronize();
    if (out && g_bf3_stamps) (void)hipMemcpy(out, g_bf3_stamps, bytes, hipMemcpyDeviceToHost);
    return 0;
}
extern "C" int mf_probe_bf3_candidates(unsigned long long* out2, int enable) {
    static unsigned long long* buf = nullptr;
    if (!buf) {
        if (hipMalloc(reinterpret_cast<void**>(&buf), 32) != hipSuccess) return -1;
        (void)hipMemset(buf, 0, 32);
    }
    (void)hipDeviceSynchronize();
    if (out2) (void)hipMemcpy(out2, buf, 24, hipMemcpyDeviceToHost);      // (three counters: callers pass room for three)
    (void)hipMemset(buf, 0, 32);
    g_bf3_dbg = enable ? buf : nullptr;          // counting costs two contended atomics per query: off while timing
    return 0;
}

template <int D>
static void bf3_launch_final(unsigned grid, const Bf3Final& fp, hipStream_t s) {
    using G = Bf3FinalGeom<D>;
    constexpr int bytes = BF3_FQ * (G::RB * D * 4 + BF3_CAND * 8 + 64 * 8 * 2 + D * 4) + 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bf3_final_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr_set = true;
    }
    bf3_final_kernel<D><<<dim3(grid), 64 * BF3_FQ, bytes, s>>>(fp);
}
#define MF_DISPATCH_BF3_FINAL(DD) if (d == DD) bf3_launch_final<DD>(fgrid, fp, s);

extern "C" int mf_topk_bf3(const float* q, int64_t Q, const float* items, const void* index, int64_t N, int d, int k,
                           const int64_t* excl_off, const int64_t* excl_idx, int64_t idx_base, void* ws, size_t ws_bytes,
                           float* out_scores, int64_t* out_idx, mf_stream_t stream) {
    if (!q || !items || !index || !out_scores || !out_idx || !ws || Q <= 0 || N <= 0)
        return mf_set_error(MF_EINVAL, "mf_topk_bf3: bad argument");
    if (k <= 0 || k > 64) return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: k = %d outside 1..64", k);
    if (d != 64 && d != 128 && d != 256) return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: embedding width %d not in {64,128,256}", d);
    if (N >= (1ll << 28) || idx_base < 0 || idx_base + N > (1ll << 32))
        return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: item indices must fit 32 bits (and a shard 2^28 rows)");
    if ((excl_off == nullptr) != (excl_idx == nullptr)) return mf_set_error(MF_EINVAL, "mf_topk_bf3: excl_off/excl_idx mismatch");
    // geometry limits first (host arithmetic only: nothing below this point may run for an unsupported shape)
    const Bf3Plan plan = bf3_plan(Q, N, d);
    const uint64_t span_tiles = (uint64_t)plan.bpc * BF3_BLOCK, span_seed = (uint64_t)plan.bpc_seed * plan.bs * BF3_BLOCK;
    if ((span_tiles > span_seed ? span_tiles : span_seed) * 32 * d * 2 > MF_SRD_MAX_BYTES) return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: chunk beyond 4 GiB");
    // exclusion words are staged through a 32-bit buffer descriptor based at a workgroup's first block and first query: the
    // blocks of its chunk (bpc_seed x bs rows of Qp 16-byte entries) plus one workgroup's queries must fit
    if (excl_off && ((uint64_t)(plan.bpc_seed * plan.bs > plan.bpc ? plan.bpc_seed * plan.bs : plan.bpc) * (uint64_t)plan.Qp +
                     (uint64_t)(32 * plan.XT * plan.xw)) * 16u + 64u > MF_SRD_MAX_BYTES)
        return mf_set_error(MF_ENOTSUP, "mf_topk_bf3: exclusion words of one chunk beyond 4 GiB (use mf_topk)");
    if (ws_bytes < mf_topk_bf3_ws_bytes(Q, N, d, k)) return mf_set_error(MF_ENOSPC, "mf_topk_bf3: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Bf3Ws w = bf3_ws(ws, Q, N, d);
    Bf3Index ix = bf3_index(const_cast<void*>(index), N, d);
    const bool excl = excl_off != nullptr;
    MF_TIMED("topk_bf3", s, {
        bf3_prep_kernel<<<dim3((unsigned)(w.plan.Qp / BF3_PREP_Q)), 256, 0, s>>>(excl_off, excl_idx, idx_base, Q, w.plan.Qp, N, w.plan.nblk,
                                                                                 excl ? w.exclW : nullptr, q, d, w.xfrag, w.ovf_cnt, w.ovf);
        int abl = 0;
#ifdef MF_BF3_LAB
        if (const char* e = getenv("MF_BF3_ABL")) abl = atoi(e);        // lab build only (make EXTRA=-DMF_BF3_LAB): wrong results, for timing
#endif
        Bf3Scan sp{Q, w.plan.Qp, ix.plane, N, w.plan.NT, w.plan.NTp, w.plan.nblk, 1, 1, 1, w.plan.xw, w.plan.nvals, w.exclW, w.gmax,
                   w.thr, w.xfrag, w.cand, w.ovf_list, w.ovf_cnt, w.ovf, abl, g_bf3_stamps};
        if (d == 64) bf3_run<64>(w, sp, excl, k, q, ix.ymax2, s);
        else if (d == 128) bf3_run<128>(w, sp, excl, k, q, ix.ymax2, s);
        else bf3_run<256>(w, sp, excl, k, q, ix.ymax2, s);
        Bf3Final fp{Q, q, items, N, d, k, w.plan.NT, w.plan.NTp, w.plan.Qp, w.plan.nwg * (BF3_WAVES / w.plan.xw), w.cand, w.eps, w.ovf_list, w.ovf_cnt, w.ovf,
                    excl ? w.exclW : nullptr, idx_base, out_scores, out_idx, sp.abl, g_bf3_dbg, g_bf3_stamps};
        const unsigned fgrid = (unsigned)((Q + BF3_FQ - 1) / BF3_FQ);
        MF_DISPATCH_BF3_FINAL(64) else MF_DISPATCH_BF3_FINAL(128) else MF_DISPATCH_BF3_FINAL(256)
    });
    return mf_check_launch("mf_topk_bf3");
}
