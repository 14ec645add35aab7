// FETCH_SIZE calibration for the two slab-DMA patterns of the layer-1 kernel (MI355X_MICROARCH.md: "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Both kernels sweep `npix` pixels x 2 lines of a
// 64-channel tensor ONCE by LDS-DMA (16 B per lane, eight lanes per pixel-line, source-side slot swizzle as in
// conv3x3s1_slab3_kernel):
//   k128: 128-byte lines, pixel stride 256 B, all lanes active          -> npix * 256 bytes needed
//   k96 :  96-byte lines, pixel stride 192 B, the lanes of slots 5, 7 idle -> npix * 192 bytes needed
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace=0 ...` and divide each kernel's counter by the bytes it needed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
static __device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
template <bool L96>
__global__ __launch_bounds__(256) void k(const char* in, long long npix, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PIX = 384;                                       // pixels per workgroup and line (48 KB of LDS image)
    const long long p0 = (long long)blockIdx.x * PIX;
    if (p0 + PIX > npix) return;                               // (never true: the grid holds whole workgroups)
    const int pixstride = L96 ? 192 : 256;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(in + p0 * pixstride), 0, PIX * pixstride, 0x00020000);
    const int i = wave * 64 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
    const bool act = !L96 || (sl != 5 && sl != 7);
    const int voff = Pl * pixstride + (L96 ? (sl == 6 ? 5 : sl) : sl) * 16;
    unsigned acc = 0;
    for (int c = 0; c < 2; ++c) {
        for (int i0 = wave * 64, r = 0; i0 < PIX * 8; i0 += 256, ++r)
            if (act) dma16_buf(rs, smem + (size_t)i0 * 16, voff, c * (L96 ? 96 : 128) + r * 32 * pixstride);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *(const unsigned*)(smem + tid * 16);
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// The tail's residual pattern (conv_dev.h resid_tile_dma_buf96): a wave fetches 32-pixel tiles of ONE 96-byte line - four DMA
// instructions per tile, eight lanes per pixel (slots 5 / 7 idle), its partner wave the same pixels' other line at about the same
// time - 256 pixels per workgroup (the layer-1 kernels' tile), every pixel-line once.
__global__ __launch_bounds__(256) void kres(const char* in, long long npix, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long p0 = (long long)blockIdx.x * 256;
    if (p0 + 256 > npix) return;
    const int pixstride = 192;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(in + p0 * pixstride), 0, 256 * pixstride, 0x00020000);
    const int half = wave >> 1, line = wave & 1;               // waves (0,1) = pixels 0-127, (2,3) = 128-255; wave parity = line
    unsigned acc = 0;
    for (int t = 0; t < 4; ++t) {                              // four 32-pixel tiles per wave
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pp = 8 * j + (lane >> 3), q = half * 128 + t * 32 + pp;
            const int sl = (lane & 7) ^ ((pp >> 1) & 7);
            if (sl != 5 && sl != 7) dma16_buf(rs, smem + wave * 4096 + j * 1024, q * pixstride + (sl == 6 ? 5 : sl) * 16, line * 96);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *(const unsigned*)(smem + wave * 4096 + lane * 16);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// r05: 96-byte lines are LINE-PLANAR (all pixels' line 0, then line 1): the slab pattern reads contiguous 96-byte pieces of one plane
__global__ __launch_bounds__(256) void kplanar(const char* in, long long npix, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PIX = 384;
    const long long p0 = (long long)blockIdx.x * PIX;
    if (p0 + PIX > npix) return;
    const int i = wave * 64 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
    const bool act = sl != 5 && sl != 7;
    const int voff = Pl * 96 + (sl == 6 ? 5 : sl) * 16;
    unsigned acc = 0;
    for (int c = 0; c < 2; ++c) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)c * npix * 96 + p0 * 96), 0, PIX * 96, 0x00020000);
        for (int i0 = wave * 64, r = 0; i0 < PIX * 8; i0 += 256, ++r)
            if (act) dma16_buf(rs, smem + (size_t)i0 * 16, voff, r * 32 * 96);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *(const unsigned*)(smem + tid * 16);
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv) {
    const long long npix = (argc > 1 ? atoll(argv[1]) : 4096) * 4096ll;      // default 16 Mi pixels: 4 GiB / 3 GiB
    char* in; unsigned* sink;
    if (hipMalloc(&in, npix * 256) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(in, 1, npix * 256);
    const int grid = (int)(npix / 384);                         // whole workgroups only: nothing is read past the allocation
    hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 384 * 128, 0, in, npix, sink);
    hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 384 * 128, 0, in, npix, sink);
    const int gres = (int)(npix / 256);
    hipLaunchKernelGGL(kres, dim3(gres), dim3(256), 4 * 4096, 0, in, npix, sink);
    hipLaunchKernelGGL(kplanar, dim3(grid), dim3(256), 384 * 128, 0, in, npix, sink);
    hipDeviceSynchronize();
    printf("kplanar (96-byte lines, line-planar, slab pattern) needs %lld bytes\n", (long long)grid * 384 * 192);
    printf("needed bytes: k<false> (128-byte lines) %lld, k<true> (96-byte lines) %lld, kres (96-byte lines, residual tile pattern) %lld\n",
           (long long)grid * 384 * 256, (long long)grid * 384 * 192, (long long)gres * 256 * 192);
    return 0;
}
