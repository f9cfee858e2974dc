// mf_stream.h -- LDS-staged tile streaming for the score engines (gfx950).
//
// A workgroup of 8 wavefronts (4 at d = 256), each owning 32 "X" rows in registers, walks
// a range of 32-row "Y" tiles.  Every tile is brought from HBM/L2 into LDS ONCE per
// workgroup by LDS-DMA (global_load_lds_dwordx4: whole 1 KiB pieces, full cache
// lines, no VGPR staging) into a 3-deep ring, two tiles ahead of the MFMAs, with a
// counted s_waitcnt vmcnt(N) and ONE raw s_barrier per tile:
//
//     wait(tile t landed)  ->  barrier  ->  issue DMA of tile t+2  ->  compute tile t
//
// (the barrier also proves every wave has finished tile t-1, whose slot tile t+2
// overwrites).  The small per-tile side inputs (mask words, norms, logQ, per-row
// coefficients) travel the same way, so the loop contains no VGPR-destination global
// load that would make hipcc drain the DMA queue (cdna_hip_programming.md 5, "Three
// .s-level traps" (b)).
//
// LDS image of a tile: [32 rows][D floats], linear for the DMA, with the 16-byte
// chunks of a row XOR-swizzled on the SOURCE address (rule 21) so that
//   * the MFMA A-fragment read (lane = row, ds_read_b128 of chunk 2g+h) and
//   * the transposed read of the backward (lane = column, ds_read_b32 along a row)
// are both bank-conflict-free.
#pragma once

#include "mf_common.h"

#ifdef __HIPCC__

typedef __attribute__((address_space(3))) void* mf_lds_ptr;
typedef const __attribute__((address_space(1))) void* mf_glb_ptr;

#ifndef MF_ABL_DMADIV
#define MF_ABL_DMADIV 1   // A/B knob (tools/): stage only 1/DMADIV of every tile -- wrong results, measures the DMA's cost
#endif

// Waves per workgroup (each owns 32 X rows; all share one tile ring).  Everything below is written for
// 4 or 8.  Measured at d = 128 (B = 8192): eight waves on ONE ring -- every Y tile DMA'd into LDS once
// per CU instead of twice, same two waves per SIMD -- against two independent 4-wave workgroups per
// CU: forward -2 %, dU +1 %, dV +8 %, selection -2 %: what the halved DMA saves, the barrier across
// eight waves takes back.  Hence four.
static constexpr int mf_nw(int d) { return (void)d, 4; }
// workgroups per CU the sweeps are compiled for (two waves per SIMD either way)
static constexpr int mf_wg_per_cu(int d) { return (mf_nw(d) == 8 || d == 256) ? 1 : 2; }

template <int D>
struct TileGeom {
    static constexpr int NW = mf_nw(D);           // waves per workgroup
    static constexpr int XB = 32 * NW;            // X rows per workgroup
    static constexpr int ROWB = D * 4;            // bytes per row
    static constexpr int TILEB = 32 * ROWB;       // bytes per tile
    static constexpr int CPR = D / 4;             // 16-byte chunks per row
    static constexpr int WAVEB = TILEB / NW;      // bytes of a tile each wave stages
    static constexpr int PIECEB = WAVEB < 1024 ? WAVEB : 1024;   // bytes per DMA instruction (16 B per active lane)
    static constexpr int PPW = WAVEB / PIECEB / MF_ABL_DMADIV;   // DMA instructions per wave and tile
    static __device__ __forceinline__ int swz(int row) { return CPR >= 16 ? (row & 15) : ((row >> 1) & 7); }
};

__device__ __forceinline__ unsigned mf_lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ int mf_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Tile source of a workgroup: a raw buffer descriptor over the rows [row0, nY) of Y plus the per-lane source
// offsets of the DMA pieces, both computed ONCE per kernel.  Piece q of this wave covers bytes
// [(wave PPW + q) PIECEB, +PIECEB) of the tile's LDS image; lane l writes 16 bytes at + 16 l and reads them from
// row r, chunk ch ^ swz(r) of the source tile.  Inside the tile loop a piece is then ONE
//     buffer_load_dwordx4 v_off, s[rsrc], s_tile_offset offen lds
// -- no per-tile VALU address arithmetic (measured in tools/lab/sweep_lab.hip: the generic per-piece index math
// the compiler otherwise emits costs ~25 VALU instructions per piece, 3 % of the matrix pipe's time) and no
// ragged-tile branch: the hardware range check covers voffset + soffset at dword granularity (measured:
// tools/lab/srd_bounds.hip), so rows past nY arrive as ZEROS (the kernels mask them anyway).  The descriptor
// starts at the workgroup's first row and covers at most MF_SRD_MAX_BYTES: every HOST plan must keep a workgroup's
// share of Y (tiles per chunk x 32 x row bytes) within that -- mf_select_plan, bf3_plan's check and the loss sweeps'
// checks do; a row beyond it would arrive as zeros while still counting as a real row.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t mf_rsrc_t;
#else
struct mf_rsrc_t { int w[4]; };     // host pass of hipcc: the target type does not exist there (never executed)
#endif
template <int D>
struct TileSrc {
    mf_rsrc_t rsrc;
    unsigned off[TileGeom<D>::PPW > 0 ? TileGeom<D>::PPW : 1];
    int wave_pb;        // LDS byte offset of this wave's first piece inside a tile image (wave-uniform)
    int row0;
};
// A descriptor never covers more than MF_SRD_MAX_BYTES, so the scalar offset MF_SRD_DEAD (+ any lane offset of
// a tile image) is out of range without wrapping 32 bits: a load issued with it fetches nothing and writes zeros.
constexpr unsigned MF_SRD_MAX_BYTES = 0xFFF00000u, MF_SRD_DEAD = 0xFFF00000u;
template <int D>
__device__ __forceinline__ void mf_tile_src_init(TileSrc<D>& ts, const float* __restrict__ Y, int64_t nY, int64_t row0) {
    using G = TileGeom<D>;
    const int lane = mf_lane();
    const int wave = mf_wave_id();
    // row0 / nY derive from kernel arguments and blockIdx only: the descriptor is provably wave-uniform (no waterfall loop)
    int64_t bytes = (nY - row0) * (int64_t)G::ROWB;
    bytes = bytes < 0 ? 0 : (bytes > (int64_t)MF_SRD_MAX_BYTES ? (int64_t)MF_SRD_MAX_BYTES : bytes);
#if defined(__HIP_DEVICE_COMPILE__)
    ts.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Y + row0 * D), 0, (int)(unsigned)bytes, 0x00020000);
#endif
    ts.row0 = (int)row0;
    ts.wave_pb = __builtin_amdgcn_readfirstlane(wave * G::PPW * G::PIECEB);
#pragma unroll
    for (int q = 0; q < G::PPW; ++q) {
        const int off = (wave * G::PPW + q) * G::PIECEB + ((lane * 16 < G::PIECEB) ? lane * 16 : 0);
        const int row = off / G::ROWB;
        const int ch = ((off % G::ROWB) >> 4) ^ G::swz(row);
        ts.off[q] = (unsigned)(row * G::ROWB + ch * 16);
        asm volatile("" : "+v"(ts.off[q]));          // keep it in a register: never rematerialised inside the loop
    }
}

// DMA piece q (0 .. PPW-1) of this wave's share of the tile of rows y0 .. y0+31 into `lds_tile`
template <int D>
__device__ __forceinline__ void mf_stage_tile_piece(char* lds_tile, int y0, int q, const TileSrc<D>& ts, bool live = true) {
    using G = TileGeom<D>;
    const bool active = mf_lane() * 16 < G::PIECEB;                            // (fewer than 64 lanes per piece only for short tiles)
#ifdef MF_ABL_SAMETILE      // A/B knob: always stage the first tile (cache-resident) -- wrong results, measures the memory side
    y0 = ts.row0;
#endif
    const int soff = live ? (y0 - ts.row0) * G::ROWB : (int)MF_SRD_DEAD;      // (a scalar select: no branch)
#if defined(__HIP_DEVICE_COMPILE__)
    if (active)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ts.rsrc, (mf_lds_ptr)(lds_tile + ts.wave_pb + q * G::PIECEB), 16, (int)ts.off[q], soff, 0, 0);
#else
    (void)active; (void)soff; (void)lds_tile;
#endif
}
template <int D>
__device__ __forceinline__ void mf_stage_tile(char* lds_tile, int y0, const TileSrc<D>& ts) {
#pragma unroll
    for (int q = 0; q < TileGeom<D>::PPW; ++q) mf_stage_tile_piece<D>(lds_tile, y0, q, ts);
}

// DMA `nbytes` (multiple of 16, <= 1024) from src to lds_dst by the calling wave.
// CPOL: cache-policy bits of the load (0 = default; 17 = sc0 sc1: read at L2, past a possibly stale L1 line)
template <int CPOL = 0>
__device__ __forceinline__ void mf_stage_small(char* lds_dst, const void* src, int nbytes) {
    // `src` is wave-uniform (SGPR pair) -- the lane part is the 32-bit offset 16 * lane: no 64-bit VALU address
    static_assert(CPOL == 0 || CPOL == 17, "cache policy: default or sc0 sc1");
    const int lane = mf_lane();
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned dst = mf_lds_addr(lds_dst);
    unsigned keep;
    if (lane * 16 < nbytes) {
        if (CPOL == 0)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc0 sc1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    }
}
// fire-and-forget device-scope max (no return value, so no wait is ever attached to it); sc1: performed
// past the XCD's own L2, so that workgroups on the other XCDs see it (the L2s are not coherent)
__device__ __forceinline__ void mf_global_umax(unsigned* p, unsigned v) {
    asm volatile("global_atomic_umax %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// LDS stores that must not drain the DMA queue: while a global_load_lds is in flight, hipcc puts
// `s_waitcnt vmcnt(0)` in front of EVERY compiler-visible LDS store (it cannot prove the store does
// not alias the DMA destination).  These go through inline asm instead; the "memory" clobber keeps
// their order against the surrounding C++ accesses, and the LDS itself executes a wave's
// operations in order.
__device__ __forceinline__ void mf_lds_store_b64(void* p, unsigned lo, unsigned hi) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    asm volatile("ds_write_b64 %0, %1" ::"v"(mf_lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void mf_lds_store_b32(void* p, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(mf_lds_addr(p)), "v"(v) : "memory");
}

template <int N>
__device__ __forceinline__ void mf_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void mf_block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the previous tile have returned
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// MFMA A-fragment of the staged tile: lane (row r = lane & 31, half h) gets chunks 2g + h.
template <int D>
__device__ __forceinline__ void mf_lds_frag(RowFrag<D>& f, const char* lds_tile) {
    using G = TileGeom<D>;
    const int lane = mf_lane(), r = lane & 31, h = lane >> 5;
    const char* rowp = lds_tile + r * G::ROWB;
    const int sw = G::swz(r);
#pragma unroll
    for (int g = 0; g < D / 8; ++g)
        f.v[g] = *reinterpret_cast<const f32x4*>(rowp + (((2 * g + h) ^ sw) << 4));
}

// Score tile of the staged Y tile with VALU work of the PREVIOUS tile threaded between the MFMA
// groups: a wave issues in order, and a dependent fp32 MFMA occupies the matrix pipe for 64
// cycles, so ~14 VALU issue slots per MFMA are free if -- and only if -- the VALU instructions
// sit between the MFMAs in program order.  `slice(s)`, s = 0 .. NSLICE-1, is that work, cut in
// NSLICE pieces; VPM = VALU/SALU instructions to place behind each MFMA.  sched_barrier /
// sched_group_barrier pin the interleave (hipcc otherwise clusters the MFMAs).
template <int D, int NSLICE, int VPM, class Slice, class Hook>
__device__ __forceinline__ f32x16 mf_tile_scores_interleaved(const char* lds_tile, const RowFrag<D>& x, Slice&& slice, Hook&& hook) {
    using G = TileGeom<D>;
    constexpr int NG = D / 8;
    static_assert(NSLICE % NG == 0 || NG % NSLICE == 0, "slices and MFMA groups must nest");
    const int lane = mf_lane(), r = lane & 31, h = lane >> 5;
    const char* rowp = lds_tile + r * G::ROWB;
    const int sw = G::swz(r);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // fragments are read two groups (8 MFMAs) ahead of their use
    f32x4 a_next = *reinterpret_cast<const f32x4*>(rowp + ((h ^ sw) << 4));
    f32x4 a_next2 = *reinterpret_cast<const f32x4*>(rowp + (((NG > 1 ? 2 : 0) + h) ^ sw) * 16);
    int s_done = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const f32x4 a = a_next;
        a_next = a_next2;
        if (g + 2 < NG) a_next2 = *reinterpret_cast<const f32x4*>(rowp + (((2 * (g + 2) + h) ^ sw) << 4));
        hook(g);                  // (memory instructions of the software pipeline: one per MFMA group)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], x.v[g][t], acc, 0, 0, 0);
        const int s_end = (g + 1) * NSLICE / NG;
#pragma unroll
        for (int sidx = g * NSLICE / NG; sidx < s_end; ++sidx) slice(sidx);
        s_done = s_end;
        // inside this group: one MFMA, then a share of the group's VALU/SALU/LDS work, four times
        // (a burst of VALU after four back-to-back MFMAs would only overlap the last of them)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, VPM, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    (void)s_done;
    return acc;
}
template <int D, int NSLICE, int VPM, class Slice>
__device__ __forceinline__ f32x16 mf_tile_scores_interleaved(const char* lds_tile, const RowFrag<D>& x, Slice&& slice) {
    return mf_tile_scores_interleaved<D, NSLICE, VPM>(lds_tile, x, slice, [](int) {});
}

// Transposed operand of the backward: lane c gets the D/32 consecutive floats
// [NB c, NB c + NB) of `row` (NB = D / 32) in one wide LDS read; v[j] is the A operand
// of accumulator block j, whose MFMA row index r then stands for feature NB r + j.
template <int D>
__device__ __forceinline__ void mf_lds_cols(float (&v)[D / 32], const char* lds_tile, int row, int c) {
    using G = TileGeom<D>;
    const char* rowp = lds_tile + row * G::ROWB;
    const int sw = G::swz(row);
    if constexpr (D == 128) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(rowp + ((c ^ sw) << 4));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else if constexpr (D == 256) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(rowp + (((2 * c) ^ sw) << 4));
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(rowp + (((2 * c + 1) ^ sw) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = t0[j]; v[4 + j] = t1[j]; }
    } else if constexpr (D == 64) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 t = *reinterpret_cast<const f32x2*>(rowp + ((((c >> 1) ^ sw) << 4) | ((c & 1) << 3)));
        v[0] = t[0]; v[1] = t[1];
    } else {
        v[0] = *reinterpret_cast<const float*>(rowp + ((((c >> 2) ^ sw) << 4) | ((c & 3) << 2)));
    }
}

#endif  // __HIPCC__
