// srd_bounds.hip -- does the hardware range check of a raw buffer_load ... lds include the SGPR offset? (gfx950)
// A 64 KiB allocation filled with 1.0f is read through a descriptor of num_records = 1000 bytes; each of 64 lanes
// asks for 16 bytes at voffset = 16 * lane (+ soffset).  Lanes whose bytes lie past num_records must read zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* Y, int nrec, int soff, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)Y, 0, nrec, 0x00020000);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256; i += 64) ((float*)smem)[i] = -7.f;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)smem, 16, lane * 16, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((float*)smem)[i];
}
int main() {
    float *d, *o, h[256];
    hipMalloc(&d, 65536); hipMalloc(&o, 1024);
    float* ones = new float[16384];
    for (int i = 0; i < 16384; ++i) ones[i] = 1.0f;
    hipMemcpy(d, ones, 65536, hipMemcpyHostToDevice);
    const int cases[][2] = {{1000, 0}, {1000, 512}, {1000, 992}, {1000, 1008}, {2048, 1024}, {2048, 1040}};
    for (auto& c : cases) {
        k<<<1, 64, 1024>>>(d, c[0], c[1], o);
        hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
        int ones_n = 0, zeros = 0, other = 0;
        for (int i = 0; i < 256; ++i) { if (h[i] == 1.0f) ++ones_n; else if (h[i] == 0.0f) ++zeros; else ++other; }
        const int expect_inc = (c[0] - c[1]) > 0 ? ((c[0] - c[1]) / 16) * 4 : 0;           // if soffset is range-checked
        const int expect_exc = c[0] / 16 * 4 > 256 ? 256 : c[0] / 16 * 4;                    // if it is not
        printf("num_records %4d soffset %4d: %3d floats read, %3d zeros, %d untouched | expected %d if soffset is range-checked, %d if not\n",
               c[0], c[1], ones_n, zeros, other, expect_inc > 256 ? 256 : expect_inc, expect_exc);
    }
    return 0;
}
