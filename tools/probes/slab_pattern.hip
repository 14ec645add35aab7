// How fast can a CU fill LDS slabs by LDS-DMA, depending on how the 96-byte lines lie in memory?  (r05 design probe for the layer-1 kernel.)
// Each workgroup = 256 threads, 53 248 bytes of LDS (three workgroups per CU, like conv3x3s1_rows_kernel), one tile of 4 x 64 output pixels:
// for each of the two 32-channel lines it fetches the slab (391 pixels x six 16-byte pieces) and waits (vmcnt(0) + barrier) - the
// load phases of the real kernel without its arithmetic.  Layouts:
//   0  interleaved: pixel record = [line 0 (96 B)][line 1 (96 B)], i.e. 96 of every 192 bytes per load (the r03-r05 tensors)
//   1  planar: all pixels' line 0, then all pixels' line 1 - a slab is ONE contiguous 37.5 KB range
//   2  interleaved 128-byte lines (pixel record 256 B, 128 of every 256 bytes per load, eight pieces)
// `spin`: s_sleep units between the loads (stands in for the multiply phases).  Prints GB/s of useful bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
static __device__ __forceinline__ void dma16(const void* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds), "v"(g) : "memory");
}
template <int LAYOUT>
__global__ __launch_bounds__(256, 3) void probe(const char* base, size_t plane_bytes, int tiles_per_img, int spin, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int t = blockIdx.x;
    const int img = t / tiles_per_img, row4 = t % tiles_per_img;
    const long long p0 = (long long)img * 4225 + (long long)row4 * 4 * 65 + 66 - 66;        // slab start pixel (PF-like pitch 65)
    constexpr int PIECES = LAYOUT == 2 ? 8 : 6, NPIX = 391;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    for (int c = 0; c < 2; ++c) {
        for (int i0 = wave * 64; i0 < NPIX * 8; i0 += 256) {
            const int i = i0 + lane, Pl = i >> 3, s = i & 7;
            const bool on = LAYOUT == 2 ? true : (s != 5 && s != 7);
            const int piece = LAYOUT == 2 ? s : (s == 6 ? 5 : s);
            const char* src;
            if (LAYOUT == 0) src = base + (size_t)(p0 + Pl) * 192 + c * 96 + piece * 16;
            else if (LAYOUT == 1) src = base + (size_t)c * plane_bytes + (size_t)(p0 + Pl) * 96 + piece * 16;
            else src = base + (size_t)(p0 + Pl) * 256 + c * 128 + piece * 16;
            if (on && Pl < NPIX) dma16(src, lds0 + i0 * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
    }
    if (tid == 0 && smem[17] == 123) sink[0] = 1.f;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2000;                   // images of 64 x 64 pixels (16 tiles each)
    const size_t pixels = (size_t)n * 4225 + 1024;
    char* d; float* sink;
    hipMalloc(&d, pixels * 256);
    hipMemset(d, 1, pixels * 256);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 8, 16}) {
        for (int layout = 0; layout < 3; ++layout) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                const size_t lds = 53248;
                if (layout == 0) hipLaunchKernelGGL(probe<0>, dim3(n * 16), dim3(256), lds, 0, d, pixels * 96, 16, spin, sink);
                else if (layout == 1) hipLaunchKernelGGL(probe<1>, dim3(n * 16), dim3(256), lds, 0, d, pixels * 96, 16, spin, sink);
                else hipLaunchKernelGGL(probe<2>, dim3(n * 16), dim3(256), lds, 0, d, pixels * 96, 16, spin, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double bytes = (double)n * 16 * 2 * 391 * (layout == 2 ? 128 : 96);
            printf("spin %2d  layout %d (%s): %.3f ms  %.0f GB/s of slab bytes (%.2f GB)\n", spin, layout,
                   layout == 0 ? "interleaved 96 of 192" : layout == 1 ? "planar 96" : "interleaved 128 of 256", best, bytes / best / 1e6, bytes / 1e9);
        }
    }
    return 0;
}
