// mf_common.h -- device-side building blocks shared by the gfx950 kernels.
//
// The score tile engine: one wavefront (64 lanes) owns a 32 x 32 tile of
//     C[y][x] = sum_k Y[y][k] * X[x][k]
// computed with v_mfma_f32_32x32x2_f32 (exact fp32, one k-ordered fmaf chain per
// element == mf_dot_chain of include/mf_numerics.h).  X rows live "on the lane"
// (column x = lane & 31), Y rows land in the 16 accumulator registers:
//     row(e, h) = (e & 3) + 8 * (e >> 2) + 4 * h,   h = lane >> 5.
// Per-X-row reductions (softmax statistics, per-row top-k) are therefore
// lane-local plus one exchange between lanes l and l + 32.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mf_hip.h"
#include "mf_numerics.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF_WAVE 64

// ---- host-side error plumbing (definitions in mf_api.hip) -------------------
int mf_set_error(int code, const char* fmt, ...);
int mf_check_launch(const char* what);
// optional HIP-event spans around the dominant kernels (mf_api.hip)
bool mf_timing_on();
void mf_timing_begin(const char* name, hipStream_t s);
void mf_timing_end(const char* name, hipStream_t s);
#define MF_TIMED(name, stream, ...)                        \
    do {                                                   \
        const bool mf_t_ = mf_timing_on();                 \
        if (mf_t_) mf_timing_begin(name, stream);          \
        __VA_ARGS__;                                       \
        if (mf_t_) mf_timing_end(name, stream);            \
    } while (0)
static inline bool mf_width_ok(int d) { return d == 32 || d == 64 || d == 128 || d == 256; }
static inline size_t mf_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t mf_pad32(int64_t x) { return (x + 31) / 32 * 32; }

#define MF_DISPATCH_D(d, ...)                                   \
    switch (d) {                                                \
        case 32: { constexpr int D = 32; __VA_ARGS__; } break;  \
        case 64: { constexpr int D = 64; __VA_ARGS__; } break;  \
        case 128: { constexpr int D = 128; __VA_ARGS__; } break;\
        case 256: { constexpr int D = 256; __VA_ARGS__; } break;\
        default: return mf_set_error(MF_EINVAL, "embedding width %d not in {32,64,128,256}", d); \
    }

// ---- bump allocator over the caller's workspace ------------------------------
struct MfArena {
    char* base;
    size_t off;
    explicit MfArena(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <typename T>
    T* take(size_t count) {
        off = mf_align_up(off, 256);
        T* p = reinterpret_cast<T*>(base ? base + off : nullptr);
        off += count * sizeof(T);
        return p;
    }
    size_t used() const { return mf_align_up(off, 256); }
};

#ifdef __HIPCC__

__device__ __forceinline__ int mf_lane() { return threadIdx.x & 63; }

// Fragment of one row per lane pair: lane (r = lane & 31, h = lane >> 5) holds
// elements 8g + 4h .. 8g + 4h + 3 of row r for every group g.
template <int D>
struct RowFrag {
    f32x4 v[D / 8];
};

template <int D>
__device__ __forceinline__ void mf_load_frag(RowFrag<D>& f, const float* __restrict__ base,
                                             int64_t row, bool valid) {
    // rows past the end are read from row 0 and zeroed (never an out-of-bounds load)
    const int h = mf_lane() >> 5;
    const f32x4* p = reinterpret_cast<const f32x4*>(base + (valid ? row : 0) * D + 4 * h);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < D / 8; ++g) {
        f32x4 x = p[2 * g];
        f.v[g] = valid ? x : zero;
    }
}

// 32x32 score tile: rows = the Y fragment's rows, columns = the X fragment's rows.
template <int D>
__device__ __forceinline__ f32x16 mf_tile_scores(const RowFrag<D>& y, const RowFrag<D>& x) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int g = 0; g < D / 8; ++g) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y.v[g][t], x.v[g][t], acc, 0, 0, 0);
    }
    return acc;
}

// accumulator register e of lane half h  ->  tile row
__device__ __forceinline__ constexpr int mf_acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// ---- lane exchanges without the LDS crossbar ---------------------------------------------------------------------
// `__shfl_xor` is a ds_bpermute_b32: an LDS-pipe round trip (~100+ cycles, and contended by every other wave's LDS work)
// per step of a reduction.  The value of lane (l ^ M) comes as well from the VALU's own cross-lane paths: DPP within a
// row of 16 (quad_perm for M = 1 and 2, two bank-masked row shifts for M = 4, row_ror:8 for M = 8) and gfx950's
// v_permlane16_swap / v_permlane32_swap between rows.  The same value moves, so a butterfly built on these has the
// result of the `__shfl_xor` one bit for bit.  Like every cross-lane operation: call it with the whole wave active.
template <int CTRL, int BANKS = 0xF>
__device__ __forceinline__ unsigned mf_dpp_u32(unsigned old, unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, 0xF, BANKS, false);
}
// {rows of the lower half twice, rows of the upper half twice} (M = 32; M = 16: of each half's lower / upper row).  The second
// operand goes through an empty asm: given the SAME value twice, this compiler (ROCm 7.2 clang) folds the two results into
// one register (`v_add_f32 v1, v7, v7` for r[0] + r[1]) -- they are different lanes' values
template <int M>
__device__ __forceinline__ auto mf_swap_halves(unsigned x) {
    unsigned y = x;
    asm volatile("" : "+v"(y));
    if constexpr (M == 16) return __builtin_amdgcn_permlane16_swap(x, y, false, false);
    else return __builtin_amdgcn_permlane32_swap(x, y, false, false);
}
template <int M>
__device__ __forceinline__ unsigned mf_xor_lane_u32(unsigned x) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "a single lane bit");
    if constexpr (M == 1) return mf_dpp_u32<0xB1>(x, x);                                   // quad_perm:[1,0,3,2]
    else if constexpr (M == 2) return mf_dpp_u32<0x4E>(x, x);                              // quad_perm:[2,3,0,1]
    else if constexpr (M == 4) return mf_dpp_u32<0x114, 0xA>(mf_dpp_u32<0x104, 0x5>(x, x), x);   // row_shl:4 into lanes with bit 2 clear, row_shr:4 into the others
    else if constexpr (M == 8) return mf_dpp_u32<0x128>(x, x);                             // row_ror:8
    else {
        const auto r = mf_swap_halves<M>(x);                 // the partner's value is the one that is not mine
        return (threadIdx.x & M) ? r[0] : r[1];
    }
}
template <int M>
__device__ __forceinline__ float mf_xor_lane(float x) { return __builtin_bit_cast(float, mf_xor_lane_u32<M>(__builtin_bit_cast(unsigned, x))); }
template <int M>
__device__ __forceinline__ unsigned long long mf_xor_lane_u64(unsigned long long x) {
    return ((unsigned long long)mf_xor_lane_u32<M>((unsigned)(x >> 32)) << 32) | mf_xor_lane_u32<M>((unsigned)x);
}
// x + (lane ^ M's x): for the two swaps the sum of the two halves is symmetric, no select needed
template <int M>
__device__ __forceinline__ float mf_xor_add(float x) {
    if constexpr (M >= 16) {
        const auto r = mf_swap_halves<M>(__builtin_bit_cast(unsigned, x));
        unsigned a = r[0], b = r[1];
        asm volatile("" : "+v"(a), "+v"(b));       // (see mf_swap_halves: without it the sum comes out as r[0] + r[0])
        return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    } else {
        return x + mf_xor_lane<M>(x);
    }
}
// the butterfly 32, 16, .. 1 of `for (m = 32; m; m >>= 1) x += __shfl_xor(x, m)`, from step WIDTH / 2 down
template <int WIDTH>
__device__ __forceinline__ float mf_butterfly_sum(float x) {
    if constexpr (WIDTH >= 64) x = mf_xor_add<32>(x);
    if constexpr (WIDTH >= 32) x = mf_xor_add<16>(x);
    if constexpr (WIDTH >= 16) x = mf_xor_add<8>(x);
    if constexpr (WIDTH >= 8) x = mf_xor_add<4>(x);
    if constexpr (WIDTH >= 4) x = mf_xor_add<2>(x);
    if constexpr (WIDTH >= 2) x = mf_xor_add<1>(x);
    return x;
}
__device__ __forceinline__ float mf_wave_sum(float x) { return mf_butterfly_sum<64>(x); }
__device__ __forceinline__ int mf_wave_sum_int(int x) {
    x += (int)mf_xor_lane_u32<32>((unsigned)x); x += (int)mf_xor_lane_u32<16>((unsigned)x); x += (int)mf_xor_lane_u32<8>((unsigned)x);
    x += (int)mf_xor_lane_u32<4>((unsigned)x); x += (int)mf_xor_lane_u32<2>((unsigned)x); x += (int)mf_xor_lane_u32<1>((unsigned)x);
    return x;
}

__device__ __forceinline__ float mf_shfl_xor32(float x) { return mf_xor_lane<32>(x); }
__device__ __forceinline__ unsigned mf_shfl_xor32u(unsigned x) { return mf_xor_lane_u32<32>(x); }

__device__ __forceinline__ unsigned long long mf_shfl_xor_u64(unsigned long long x, int m) {
    switch (m) {                                   // (a constant at every call site)
        case 1: return mf_xor_lane_u64<1>(x);
        case 2: return mf_xor_lane_u64<2>(x);
        case 4: return mf_xor_lane_u64<4>(x);
        case 8: return mf_xor_lane_u64<8>(x);
        case 16: return mf_xor_lane_u64<16>(x);
        default: return mf_xor_lane_u64<32>(x);
    }
}
// the wave's maximum, in every lane: max is exact and order-free, so the butterfly's partner may as well be any lane that
// makes the classes merge -- here the cheapest ones
__device__ __forceinline__ unsigned mf_wave_max_u32(unsigned x) {
    {
        const auto r = mf_swap_halves<32>(x);
        x = r[0] > r[1] ? r[0] : r[1];
    }
    {
        const auto r = mf_swap_halves<16>(x);
        x = r[0] > r[1] ? r[0] : r[1];
    }
    unsigned o;
    o = mf_xor_lane_u32<8>(x); x = o > x ? o : x;
    o = mf_dpp_u32<0x124>(x, x); x = o > x ? o : x;            // row_ror:4 (classes are closed under ^8 by now: the same merge as ^4)
    o = mf_xor_lane_u32<2>(x); x = o > x ? o : x;
    o = mf_xor_lane_u32<1>(x); x = o > x ? o : x;
    return x;
}
// 64-bit keys: the largest high word, then the largest low word among the lanes that hold it -- two 32-bit reductions
// instead of one of 64-bit compare-and-selects over two exchanged words
__device__ __forceinline__ unsigned long long mf_wave_max_u64(unsigned long long x) {
    const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
    const unsigned mh = mf_wave_max_u32(hi);
    const unsigned long long holders = __ballot(hi == mh);
    unsigned ml;
    if (__popcll(holders) == 1)                                 // (wave-uniform) the usual case: one lane holds the largest high word
        ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(holders));
    else
        ml = mf_wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}
__device__ __forceinline__ float mf_group_sum(float x, int width) {  // width: power of two <= 64 (a constant at every call site)
    switch (width) {
        case 64: return mf_butterfly_sum<64>(x);
        case 32: return mf_butterfly_sum<32>(x);
        case 16: return mf_butterfly_sum<16>(x);
        case 8: return mf_butterfly_sum<8>(x);
        case 4: return mf_butterfly_sum<4>(x);
        case 2: return mf_butterfly_sum<2>(x);
        default: return x;
    }
}

// A zero fill as a KERNEL (16-byte aligned pointer, bytes a multiple of 4): inside a captured hipGraph a memset node
// costs ~100 us per replay on this stack, a kernel node its own few microseconds.
template <int DUMMY>
__global__ __launch_bounds__(256) void mf_zero_kernel(uint32_t* __restrict__ p, int64_t n4) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, n16 = n4 >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t q = t; q < n16; q += stride) reinterpret_cast<uint4*>(p)[q] = uint4{0u, 0u, 0u, 0u};
    if (t < (n4 & 3)) p[n16 * 4 + t] = 0u;
}
static inline void mf_zero_async(void* p, size_t bytes, hipStream_t s) {
    const int64_t n4 = (int64_t)(bytes / 4);
    if (n4 <= 0) return;
    int64_t blocks = (n4 / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    mf_zero_kernel<0><<<dim3((unsigned)blocks), 256, 0, s>>>(static_cast<uint32_t*>(p), n4);
}

#endif  // __HIPCC__
