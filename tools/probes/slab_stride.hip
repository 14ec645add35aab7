// Does the pixel-record stride of a slab load matter?  (r05 design probe for layers 2-4.)  A workgroup (256 threads, 45 KB of LDS + 32 KB
// unused, two per CU like conv3x3s1_wide_kernel) fetches the slab of one 32-channel line - 350 pixels x 128 bytes - for each of NC lines
// of its tile by LDS-DMA and waits.  Interleaved: line c of pixel p at p * (NC * 128) + c * 128 (the padded-flat tensors: 128 of every
// 512 / 1024 / 2048 bytes); planar: at c * plane + p * 128 (one contiguous 44.8 KB range per load).  Prints GB/s of slab bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
static __device__ __forceinline__ void dma16(const void* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds), "v"(g) : "memory");
}
__global__ __launch_bounds__(256, 2) void probe(const char* base, int planar, int NC, size_t plane_bytes, int px_per_tile, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const long long p0 = (long long)blockIdx.x * px_per_tile;
    constexpr int NPIX = 350;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    for (int c = 0; c < NC; ++c) {
        for (int i0 = wave * 64; i0 < NPIX * 8; i0 += 256) {
            const int i = i0 + lane, Pl = i >> 3, s = i & 7;
            const char* src = planar ? base + (size_t)c * plane_bytes + (size_t)(p0 + Pl) * 128 + s * 16
                                     : base + (size_t)(p0 + Pl) * (NC * 128) + c * 128 + s * 16;
            if (Pl < NPIX) dma16(src, lds0 + i0 * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __syncthreads();
    }
    if (tid == 0 && smem[17] == 123) sink[0] = 1.f;
}
int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 8000;
    char* d; float* sink;
    const size_t pixels = (size_t)tiles * 288 + 1024;
    hipMalloc(&d, pixels * 2048);
    hipMemset(d, 1, pixels * 2048);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int NC : {4, 8, 16}) {
        const int ntile = tiles * 4 / NC;                           // same bytes per run
        for (int planar = 0; planar < 2; ++planar) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe, dim3(ntile), dim3(256), 45056 + 32768, 0, d, planar, NC, pixels * 128, 288, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double bytes = (double)ntile * NC * 350 * 128;
            printf("NC %2d (C = %4d) %s: %.3f ms  %.0f GB/s of slab bytes (%.2f GB)\n", NC, NC * 32, planar ? "planar     " : "interleaved", best, bytes / best / 1e6, bytes / 1e9);
        }
    }
    return 0;
}
