// sweep_lab.hip -- skeleton laboratory for the fp32-MFMA streaming sweeps (gfx950).
//
// Standalone (no torch): hipcc --offload-arch=gfx950 -O3 -o sweep_lab sweep_lab.hip ; ./sweep_lab
//
// What it answers: where the ~33 % of the matrix pipe's time go in the skeleton
// that loss_fwd_dense / loss_bwd_dense / select_kernel share -- {X rows in registers, Y tiles through an
// LDS ring filled by LDS-DMA, one barrier per tile, ds_read_b128 A-fragments, v_mfma_f32_32x32x2_f32} --
// and which geometry removes them.  One kernel template, knobs:
//   XT    X tiles (32 rows) per wave                         1, 2, 4
//   WGPC  workgroups (4 waves) per CU the kernel is built for 1, 2
//   YR    Y rows per ring slot / barrier                      32, 64
//   F_NODMA    tiles staged once, the loop issues no DMA      (wrong results: prices the DMA)
//   F_NOBAR    no s_barrier in the loop                        (wrong results: prices the barrier)
//   F_NOLDS    A fragments read once, reused                   (wrong results: prices the LDS reads)
//   F_MIDBAR   wait + barrier + DMA issue in the MIDDLE of a tile's MFMAs (correct results)
//   F_STAMP    s_memtime stamps: wait / barrier / DMA issue / MFMA share of a tile (diagnostic build)
//   F_BF3      the contraction as THREE v_mfma_f32_32x32x16_bf16 per 16 k (hi.hi + lo.hi + hi.lo of a split-bf16 pair:
//              24 MFMAs of 32 cycles per tile instead of 64 of 64; same LDS bytes; operand bits are whatever the fp32
//              data holds -- timing only, no check).  NV / NE then count per bf16 MFMA: 11 / 1 give the forward's
//              ~260 VALU + 16 exp per tile.  Prices what a split-bf16 training sweep could reach
// Output per variant: median ms over interleaved rounds, TFLOP/s, fraction of 157.3, the in-kernel clock
// (s_memtime / s_memrealtime), and for the correct variants a check against a host reference.
//
// acc is NOT cleared between tiles: out[x][row32] = sum over the split's tiles of Y[32 t + row] . X[x]
// (a plain chain over k and tiles), which the host reproduces exactly in double on a sample.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

static constexpr int D = 128, ROWB = D * 4, NW = 4;
enum { F_NODMA = 1, F_NOBAR = 2, F_NOLDS = 4, F_MIDBAR = 8, F_STAMP = 16, F_EPI = 32, F_BF3 = 64 };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// one wave's share (piece q of PPW) of the DMA of Y rows [y0, y0 + YR) into lds_tile ([YR][128] floats, 16-byte
// chunks XOR-swizzled by (row & 15) on the SOURCE address)
template <int YR>
__device__ __forceinline__ unsigned piece_lane_off(int q) {      // source byte offset of this lane inside the tile, piece q
    constexpr int PPW = YR * ROWB / NW / 1024;
    const int lane = lane_id(), wave = wave_id();
    const int off = (wave * PPW + q) * 1024 + lane * 16;
    const int row = off / ROWB;
    const int ch = ((off % ROWB) >> 4) ^ (row & 15);
    return (unsigned)(row * ROWB + ch * 16);
}
template <int YR>
struct LaneOffs {
    unsigned v[YR * ROWB / NW / 1024];
};
template <int YR>
__device__ __forceinline__ void stage_piece(char* lds_tile, const float* __restrict__ Y, int64_t y0, int q, const LaneOffs<YR>& lo) {
    constexpr int PPW = YR * ROWB / NW / 1024;
    const int pb = (wave_id() * PPW + q) * 1024;
    const char* tb = reinterpret_cast<const char*>(Y) + y0 * ROWB;          // wave-uniform
    __builtin_amdgcn_global_load_lds((glb_ptr)(tb + lo.v[q]), (lds_ptr)(lds_tile + pb), 16, 0, 0);
}
template <int YR>
__device__ __forceinline__ void stage_tile(char* lds_tile, const float* __restrict__ Y, int64_t y0, const LaneOffs<YR>& lo) {
#pragma unroll
    for (int q = 0; q < YR * ROWB / NW / 1024; ++q) stage_piece<YR>(lds_tile, Y, y0, q, lo);
}

template <int XT>
struct XRegs {
    f32x4 v[XT][D / 8];
};

// 64 * XT MFMAs over one 32-row sub-tile; `mid(g)` is called after group g (hook for the mid-tile barrier)
// NV: dummy v_fma_f32 per MFMA, NE: dummy v_exp_f32 per 4 MFMAs (pricing the epilogue's VALU work)
template <int NV, int NE>
__device__ __forceinline__ void dummy_valu(float (&d)[8], int t) {
#pragma unroll
    for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(d[(k + 2 * t) & 7]) : "v"(d[(k + 2 * t + 4) & 7]));
    if (NE > 0 && t == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(d[t & 7]));
}
template <int XT, int FLAGS, int NV, int NE, class Mid>
__device__ __forceinline__ void contract(const char* lds_sub, const XRegs<XT>& x, f32x16 (&acc)[XT], float (&dv)[8], Mid&& mid) {
    constexpr int NG = D / 8;
    const int lane = lane_id(), r = lane & 31, h = lane >> 5;
    const char* rowp = lds_sub + r * ROWB;
    const int sw = r & 15;
    f32x4 a_next = *reinterpret_cast<const f32x4*>(rowp + ((h ^ sw) << 4));
    f32x4 a_next2 = *reinterpret_cast<const f32x4*>(rowp + (((2 + h) ^ sw) << 4));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const f32x4 a = a_next;
        if (!(FLAGS & F_NOLDS)) {
            a_next = a_next2;
            if (g + 2 < NG) a_next2 = *reinterpret_cast<const f32x4*>(rowp + (((2 * (g + 2) + h) ^ sw) << 4));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < XT; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], x.v[j][g][t], acc[j], 0, 0, 0);
                dummy_valu<NV, NE>(dv, t);
            }
        mid(g);
    }
}

template <int XT, int FLAGS, int NV, int NE, class Mid>
__device__ __forceinline__ void contract_bf3(const char* lds_sub, const XRegs<XT>& x, f32x16 (&acc)[XT], float (&dv)[8], Mid&& mid) {
    constexpr int KS = D / 16;
    const int lane = lane_id(), r = lane & 31, h = lane >> 5;
    const char* rowp = lds_sub + r * ROWB;
    const int sw = r & 15;
    bf16x8 ah = *reinterpret_cast<const bf16x8*>(rowp + ((h ^ sw) << 4));
    bf16x8 al = *reinterpret_cast<const bf16x8*>(rowp + (((16 + h) ^ sw) << 4));
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 a_h = ah, a_l = al;
        if (!(FLAGS & F_NOLDS) && s + 1 < KS) {
            ah = *reinterpret_cast<const bf16x8*>(rowp + (((2 * (s + 1) + h) ^ sw) << 4));
            al = *reinterpret_cast<const bf16x8*>(rowp + (((16 + 2 * (s + 1) + h) ^ sw) << 4));
        }
#pragma unroll
        for (int j = 0; j < XT; ++j) {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, x.v[j][2 * s]), xl = __builtin_bit_cast(bf16x8, x.v[j][2 * s + 1]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, xh, acc[j], 0, 0, 0);
            dummy_valu<NV, NE>(dv, 1);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, xh, acc[j], 0, 0, 0);
            dummy_valu<NV, NE>(dv, 1);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, xl, acc[j], 0, 0, 0);
            dummy_valu<NV, 0>(dv, 2);
        }
        mid(2 * s);
        mid(2 * s + 1);
    }
}

template <int XT, int FLAGS, int NV, int NE, class Mid>
__device__ __forceinline__ void contract_sel(const char* lds_sub, const XRegs<XT>& x, f32x16 (&acc)[XT], float (&dv)[8], Mid&& mid) {
    if constexpr ((FLAGS & F_BF3) != 0) contract_bf3<XT, FLAGS, NV, NE>(lds_sub, x, acc, dv, mid);
    else contract<XT, FLAGS, NV, NE>(lds_sub, x, acc, dv, mid);
}

struct LabParams {
    const float *X, *Y;
    float* out;                 // [nsplit][nX][32]
    float* scratch;             // stash-like traffic target (B x N floats)
    unsigned long long* dbg;    // [WGs][8]
    int64_t nX, nY;
    int tps;                    // YR-row tiles per split
};

template <int XT, int WGPC, int YR, int FLAGS, int ST, int LD, int NV, int NE>
__global__ __launch_bounds__(256, WGPC) void lab_kernel(LabParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILEB = YR * ROWB;
    constexpr int NSLOT = 3;
    constexpr int PPW = TILEB / NW / 1024;
    constexpr int NSUB = YR / 32;
    const int lane = lane_id(), c = lane & 31, h = lane >> 5, wave = wave_id();
    const int64_t x0 = ((int64_t)blockIdx.y * NW + wave) * (32 * XT);
    const int t0 = blockIdx.x * p.tps, t1 = t0 + p.tps;
    XRegs<XT> x;
#pragma unroll
    for (int j = 0; j < XT; ++j) {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.X + (x0 + 32 * j + c) * D + 4 * h);
#pragma unroll
        for (int g = 0; g < D / 8; ++g) x.v[j][g] = src[2 * g];
    }
    f32x16 acc[XT];
#pragma unroll
    for (int j = 0; j < XT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    unsigned long long c_wait = 0, c_bar = 0, c_dma = 0, c_mfma = 0, c_tot0 = 0, r0 = 0;
    if (FLAGS & F_STAMP) { c_tot0 = stamp(); r0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }

    LaneOffs<YR> lo;
#pragma unroll
    for (int q = 0; q < PPW; ++q) { lo.v[q] = piece_lane_off<YR>(q); asm volatile("" : "+v"(lo.v[q])); }
    auto slot = [&](int t) { return smem + ((t - t0) % NSLOT) * TILEB; };
    float dv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dv[k] = 0.5f + 0.01f * (float)(lane + k);
    // stash-like traffic: ST stores / LD loads of 1 KiB per X tile and 32-row Y sub-tile, spread over the MFMA groups
    f32x4 ldv[XT][LD > 0 ? LD : 1];
    float* sbase = p.scratch + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * (int64_t)p.tps * (YR / 32) * XT * 1024;
    auto vmem_hook = [&](int ty, int s, int g) {
        if ((ST > 0 || LD > 0) && (g & 3) == 1) {
            const int q = g >> 2;
            float* blk = sbase + ((int64_t)((ty - t0) * (YR / 32) + s) * XT) * 1024 + lane * 4;
#pragma unroll
            for (int j = 0; j < XT; ++j) {
                if (q < ST) *reinterpret_cast<f32x4*>(blk + j * 1024 + q * 256) = x.v[j][q];
                if (q < LD) ldv[j][q] = *reinterpret_cast<const f32x4*>(blk + j * 1024 + q * 256 + (ST > 0 ? 0 : 0));
            }
        }
        if (LD > 0 && g == 15) {
#pragma unroll
            for (int j = 0; j < XT; ++j)
#pragma unroll
                for (int q = 0; q < LD; ++q) asm volatile("" ::"v"(ldv[j][q]));
        }
    };
    // prologue: two tiles in flight
    stage_tile<YR>(slot(t0), p.Y, (int64_t)t0 * YR, lo);
    if (t0 + 1 < t1) stage_tile<YR>(slot(t0 + 1), p.Y, (int64_t)(t0 + 1) * YR, lo);
    if (FLAGS & F_NODMA) {
        if (t0 + 2 < t1) stage_tile<YR>(slot(t0 + 2), p.Y, (int64_t)(t0 + 2) * YR, lo);
        wait_vmcnt<0>();
        block_barrier();
    }
    if (FLAGS & F_MIDBAR) {
        // tile t0 must be complete before the loop; inside the loop the wait/barrier for tile t+1 sits mid-tile
        if (t0 + 1 < t1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
        block_barrier();
    }
    for (int ty = t0; ty < t1; ++ty) {
        if (!(FLAGS & (F_NODMA | F_MIDBAR))) {
            unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if (FLAGS & F_STAMP) { __builtin_amdgcn_sched_barrier(0); s0 = stamp(); __builtin_amdgcn_sched_barrier(0); }
            if (ty + 1 < t1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
            if (FLAGS & F_STAMP) { __builtin_amdgcn_sched_barrier(0); s1 = stamp(); __builtin_amdgcn_sched_barrier(0); }
            if (!(FLAGS & F_NOBAR)) block_barrier();
            if (FLAGS & F_STAMP) { __builtin_amdgcn_sched_barrier(0); s2 = stamp(); __builtin_amdgcn_sched_barrier(0); }
            if (ty + 2 < t1) stage_tile<YR>(slot(ty + 2), p.Y, (int64_t)(ty + 2) * YR, lo);
            if (FLAGS & F_STAMP) {
                __builtin_amdgcn_sched_barrier(0); s3 = stamp(); __builtin_amdgcn_sched_barrier(0);
                c_wait += s1 - s0; c_bar += s2 - s1; c_dma += s3 - s2;
            }
        } else if (!(FLAGS & F_MIDBAR) && !(FLAGS & F_NOBAR)) {
            block_barrier();
        }
        const char* tile = slot(ty);
        unsigned long long m0 = 0;
        if (FLAGS & F_STAMP) { __builtin_amdgcn_sched_barrier(0); m0 = stamp(); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            if ((FLAGS & F_MIDBAR) && s == NSUB / 2) {
                contract_sel<XT, FLAGS, NV, NE>(tile + s * 32 * ROWB, x, acc, dv, [&](int g) {
                    vmem_hook(ty, s, g);
                    if (g == (NSUB == 1 ? 7 : 0)) {
                        // every wave is past the middle of tile ty => all of them finished tile ty-1: its slot is free
                        if (ty + 1 < t1) {
                            wait_vmcnt<0>();   // tile ty+1: issued one tile ago (tile ty+2 goes out behind this barrier)
                            block_barrier();
                            if (ty + 2 < t1) stage_tile<YR>(slot(ty + 2), p.Y, (int64_t)(ty + 2) * YR, lo);
                        }
                    }
                });
            } else {
                contract_sel<XT, FLAGS, NV, NE>(tile + s * 32 * ROWB, x, acc, dv, [&](int g) { vmem_hook(ty, s, g); });
            }
        }
        if (FLAGS & F_STAMP) { __builtin_amdgcn_sched_barrier(0); c_mfma += stamp() - m0; __builtin_amdgcn_sched_barrier(0); }
    }
    if (FLAGS & F_STAMP) {
        const unsigned long long c1 = stamp();
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (lane == 0) {
            unsigned long long* o = p.dbg + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
            o[0] = c1 - c_tot0; o[1] = c_wait; o[2] = c_bar; o[3] = c_dma; o[4] = c_mfma; o[5] = r1 - r0;
        }
    }
#pragma unroll
    for (int j = 0; j < XT; ++j) {
        float* o = p.out + ((int64_t)blockIdx.x * p.nX + x0 + 32 * j + c) * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[(e & 3) + 8 * (e >> 2) + 4 * h] = acc[j][e];
    }
}

struct Variant {
    std::string name;
    int xt, wgpc, yr, flags;
    void (*launch)(const LabParams&, dim3, int, hipStream_t);
    int lds;
    bool correct;
};

template <int XT, int WGPC, int YR, int FLAGS, int ST, int LD, int NV, int NE>
static void launch_v(const LabParams& p, dim3 grid, int lds, hipStream_t s) {
    auto fn = lab_kernel<XT, WGPC, YR, FLAGS, ST, LD, NV, NE>;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); attr = true; }
    fn<<<grid, 256, lds, s>>>(p);
}
#define V(XT, WGPC, YR, FLAGS, NAME) VE(XT, WGPC, YR, FLAGS, 0, 0, 0, 0, NAME)
#define VE(XT, WGPC, YR, FLAGS, ST, LD, NV, NE, NAME) \
    Variant { NAME, XT, WGPC, YR, FLAGS, launch_v<XT, WGPC, YR, FLAGS, ST, LD, NV, NE>, ((WGPC) == 1 && 3 * YR * ROWB < 84 * 1024) ? 84 * 1024 : 3 * YR * ROWB, !((FLAGS) & (F_NODMA | F_NOBAR | F_NOLDS | F_BF3)) }

int main(int argc, char** argv) {
    const int64_t nX = 8192, nY = 16384;
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    std::vector<float> hX(nX * D), hY(nY * D);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; };
    for (auto& v : hX) v = rnd() * 0.1f;
    for (auto& v : hY) v = rnd() * 0.1f;
    float *dX, *dY, *dOut, *dScr;
    unsigned long long* dDbg;
    CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dY, hY.size() * 4));
    CK(hipMalloc(&dOut, (size_t)64 * nX * 32 * 4)); CK(hipMalloc(&dScr, (size_t)nX * nY * 4)); CK(hipMemset(dScr, 0, (size_t)nX * nY * 4)); CK(hipMalloc(&dDbg, (size_t)4096 * 4 * 8 * 8));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dY, hY.data(), hY.size() * 4, hipMemcpyHostToDevice));

    std::vector<Variant> vs = {
        VE(1, 2, 32, 0, 0, 0, 0, 0, "xt1 wg2 yr32 | base"),
        VE(1, 2, 32, 0, 4, 0, 0, 0, "xt1 wg2 yr32 | st4"),
        VE(1, 2, 32, 0, 0, 4, 0, 0, "xt1 wg2 yr32 | ld4"),
        VE(1, 2, 32, 0, 4, 4, 0, 0, "xt1 wg2 yr32 | ld4 st4"),
        VE(1, 2, 32, 0, 0, 0, 2, 0, "xt1 wg2 yr32 | valu2"),
        VE(1, 2, 32, 0, 0, 0, 4, 0, "xt1 wg2 yr32 | valu4"),
        VE(1, 2, 32, 0, 0, 0, 0, 1, "xt1 wg2 yr32 | exp1"),
        VE(1, 2, 32, 0, 0, 0, 4, 1, "xt1 wg2 yr32 | valu4 exp1"),
        VE(1, 2, 32, 0, 4, 0, 4, 1, "xt1 wg2 yr32 | fwd-like: st4 valu4 exp1"),
        VE(1, 2, 32, 0, 4, 4, 2, 1, "xt1 wg2 yr32 | dU-like: ld4 st4 valu2 exp1"),
        VE(2, 2, 32, 0, 0, 0, 0, 0, "xt2 wg2 yr32 | base"),
        VE(2, 2, 32, 0, 4, 0, 0, 0, "xt2 wg2 yr32 | st4"),
        VE(2, 2, 32, 0, 0, 4, 0, 0, "xt2 wg2 yr32 | ld4"),
        VE(2, 2, 32, 0, 4, 4, 0, 0, "xt2 wg2 yr32 | ld4 st4"),
        VE(2, 2, 32, 0, 0, 0, 2, 0, "xt2 wg2 yr32 | valu2"),
        VE(2, 2, 32, 0, 0, 0, 4, 0, "xt2 wg2 yr32 | valu4"),
        VE(2, 2, 32, 0, 0, 0, 0, 1, "xt2 wg2 yr32 | exp1"),
        VE(2, 2, 32, 0, 0, 0, 4, 1, "xt2 wg2 yr32 | valu4 exp1"),
        VE(2, 2, 32, 0, 4, 0, 4, 1, "xt2 wg2 yr32 | fwd-like: st4 valu4 exp1"),
        VE(2, 2, 32, 0, 4, 4, 2, 1, "xt2 wg2 yr32 | dU-like: ld4 st4 valu2 exp1"),
        VE(2, 1, 64, F_MIDBAR, 0, 0, 0, 0, "xt2 wg1 yr64 midbar | base"),
        VE(2, 1, 64, F_MIDBAR, 4, 0, 0, 0, "xt2 wg1 yr64 midbar | st4"),
        VE(2, 1, 64, F_MIDBAR, 0, 4, 0, 0, "xt2 wg1 yr64 midbar | ld4"),
        VE(2, 1, 64, F_MIDBAR, 4, 4, 0, 0, "xt2 wg1 yr64 midbar | ld4 st4"),
        VE(2, 1, 64, F_MIDBAR, 0, 0, 2, 0, "xt2 wg1 yr64 midbar | valu2"),
        VE(2, 1, 64, F_MIDBAR, 0, 0, 4, 0, "xt2 wg1 yr64 midbar | valu4"),
        VE(2, 1, 64, F_MIDBAR, 0, 0, 0, 1, "xt2 wg1 yr64 midbar | exp1"),
        VE(2, 1, 64, F_MIDBAR, 0, 0, 4, 1, "xt2 wg1 yr64 midbar | valu4 exp1"),
        VE(2, 1, 64, F_MIDBAR, 4, 0, 4, 1, "xt2 wg1 yr64 midbar | fwd-like: st4 valu4 exp1"),
        VE(2, 1, 64, F_MIDBAR, 4, 4, 2, 1, "xt2 wg1 yr64 midbar | dU-like: ld4 st4 valu2 exp1"),
        V(1, 2, 32, F_NODMA | F_NOBAR | F_NOLDS, "xt1 wg2 yr32 bare mfma"),
        VE(1, 2, 32, F_BF3, 0, 0, 0, 0, "bf3 xt1 wg2 yr32 | base"),
        VE(1, 2, 32, F_BF3, 4, 0, 0, 0, "bf3 xt1 wg2 yr32 | st4"),
        VE(1, 2, 32, F_BF3, 0, 0, 11, 1, "bf3 xt1 wg2 yr32 | valu264 exp16 per tile"),
        VE(1, 2, 32, F_BF3, 4, 0, 11, 1, "bf3 xt1 wg2 yr32 | fwd-like: st4 valu264 exp16"),
        VE(1, 2, 32, F_BF3, 4, 4, 6, 1, "bf3 xt1 wg2 yr32 | dU-like: ld4 st4 valu144 exp16"),
        VE(2, 2, 32, F_BF3, 0, 0, 0, 0, "bf3 xt2 wg2 yr32 | base"),
        VE(2, 2, 32, F_BF3, 4, 0, 11, 1, "bf3 xt2 wg2 yr32 | fwd-like: st4 valu264 exp16"),
        VE(2, 2, 32, F_BF3, 4, 4, 6, 1, "bf3 xt2 wg2 yr32 | dU-like: ld4 st4 valu144 exp16"),
        V(1, 2, 32, F_BF3 | F_NODMA | F_NOBAR | F_NOLDS, "bf3 xt1 wg2 yr32 bare mfma"),
        V(4, 1, 64, 0, "xt4 wg1 yr64"),
    };
    const char* only = argc > 2 ? argv[2] : nullptr;

    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = 2.0 * nX * nY * D;
    std::vector<std::vector<float>> times(vs.size());
    std::vector<LabParams> ps(vs.size());
    std::vector<dim3> grids(vs.size());
    for (size_t i = 0; i < vs.size(); ++i) {
        const Variant& v = vs[i];
        const int xblocks = (int)(nX / (32 * v.xt * NW));
        int nsplit = 256 * v.wgpc / xblocks;
        const int ytiles = (int)(nY / v.yr);
        if (nsplit > ytiles) nsplit = ytiles;
        ps[i] = LabParams{dX, dY, dOut, dScr, dDbg, nX, nY, ytiles / nsplit};
        grids[i] = dim3(nsplit, xblocks);
    }
    const int reps = 10;
    for (int r = 0; r < rounds + 1; ++r) {
        for (size_t i = 0; i < vs.size(); ++i) {
            if (only && !strstr(vs[i].name.c_str(), only)) continue;
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < reps; ++k) vs[i].launch(ps[i], grids[i], vs[i].lds, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) times[i].push_back(ms / reps);
        }
    }
    printf("%-56s %9s %8s %6s  %s\n", "variant", "ms", "TF/s", "frac", "notes");
    for (size_t i = 0; i < vs.size(); ++i) {
        if (times[i].empty()) continue;
        std::sort(times[i].begin(), times[i].end());
        const double ms = times[i][times[i].size() / 2], mn = times[i][0];
        std::string notes;
        // correctness / stamps: one more launch
        CK(hipMemsetAsync(dOut, 0, (size_t)grids[i].x * nX * 32 * 4, s));
        vs[i].launch(ps[i], grids[i], vs[i].lds, s);
        CK(hipStreamSynchronize(s));
        if (vs[i].correct) {
            std::vector<float> ho((size_t)grids[i].x * nX * 32);
            CK(hipMemcpy(ho.data(), dOut, ho.size() * 4, hipMemcpyDeviceToHost));
            double maxerr = 0;
            const int rows_per_split = ps[i].tps * vs[i].yr;
            for (int sidx = 0; sidx < 16; ++sidx) {
                const int64_t xr = (sidx * 523 + 17) % nX;
                const int sp = sidx % grids[i].x;
                for (int row = 0; row < 32; row += 5) {
                    double ref = 0;
                    for (int t = 0; t < rows_per_split / 32; ++t) {
                        const float* y = &hY[((int64_t)sp * rows_per_split + 32 * t + row) * D];
                        const float* xx = &hX[xr * D];
                        for (int k = 0; k < D; ++k) ref += (double)y[k] * xx[k];
                    }
                    const double got = ho[((int64_t)sp * nX + xr) * 32 + row];
                    maxerr = std::max(maxerr, std::fabs(got - ref));
                }
            }
            char b[64];
            snprintf(b, sizeof b, "maxerr %.2e%s", maxerr, maxerr < 1e-4 ? "" : " WRONG");
            notes += b;
        }
        if (vs[i].flags & F_STAMP) {
            const int nwv = grids[i].x * grids[i].y * NW;
            std::vector<unsigned long long> hd((size_t)nwv * 8);
            CK(hipMemcpy(hd.data(), dDbg, hd.size() * 8, hipMemcpyDeviceToHost));
            double tot = 0, w = 0, b = 0, dm = 0, mf = 0, clk = 0;
            for (int q = 0; q < nwv; ++q) {
                tot += hd[q * 8 + 0]; w += hd[q * 8 + 1]; b += hd[q * 8 + 2]; dm += hd[q * 8 + 3]; mf += hd[q * 8 + 4];
                clk += (double)hd[q * 8 + 0] / (double)hd[q * 8 + 5] * 100.0;
            }
            const double tiles = (double)nwv * ps[i].tps;
            char bb[256];
            snprintf(bb, sizeof bb, " | per wave-tile cycles: total %.0f = wait %.0f + barrier %.0f + dma-issue %.0f + mfma-block %.0f (ideal %d); clock %.0f MHz",
                     tot / tiles, w / tiles, b / tiles, dm / tiles, mf / tiles, 64 * 64 * vs[i].xt * (vs[i].yr / 32), clk / nwv);
            notes += bb;
        }
        printf("%-56s %9.4f %8.1f %6.3f  min %.4f  %s\n", vs[i].name.c_str(), ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3, mn, notes.c_str());
    }
    return 0;
}
