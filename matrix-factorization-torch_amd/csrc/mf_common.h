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

__device__ __forceinline__ float mf_shfl_xor32(float x) { return __shfl_xor(x, 32, 64); }
__device__ __forceinline__ unsigned mf_shfl_xor32u(unsigned x) { return (unsigned)__shfl_xor((int)x, 32, 64); }

__device__ __forceinline__ unsigned long long mf_shfl_xor_u64(unsigned long long x, int m) {
    unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(x & 0xFFFFFFFFull), m, 64);
    unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(x >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long mf_wave_max_u64(unsigned long long x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned long long o = mf_shfl_xor_u64(x, m);
        x = o > x ? o : x;
    }
    return x;
}
__device__ __forceinline__ float mf_group_sum(float x, int width) {  // width: power of two <= 64
    for (int m = width >> 1; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
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
