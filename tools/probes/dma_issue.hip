// How long does ONE wave need to issue an LDS-DMA instruction (buffer_load_dwordx4 ... lds, 1 KiB per instruction)?  r05: the band-marching
// experiment measured ~480 cycles per instruction from a single producer wave.  Variants: the compiler builtin with a fixed LDS destination,
// the asm helper of conv_dev.h (m0 saved, set, restored around every transfer), asm without the restore, asm with m0 written once.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/dma_issue tools/probes/dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(64) void probe(const char* src, int iters, int stride, unsigned long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(1024))) char lds[32768];
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 30, 0x00020000);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int voff = lane * 16 + it * stride;
        const unsigned dst = base + (it & 15) * 1024;
        if (MODE == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (it & 15) * 1024), 16, voff, 0, 0, 0);
        } else if (MODE == 1) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rs), "s"(0) : "memory");
        } else if (MODE == 2) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rs), "s"(0) : "memory");
        } else {
            if (it == 0) asm volatile("s_mov_b32 m0, %0" ::"s"(base) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(0) : "memory");
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) { out[0] = t1 - t0; out[1] = t2 - t0; }
    sink[lane] = *(unsigned*)(lds + lane * 4);
}

template <int MODE>
static void run(const char* name, const char* src, int stride) {
    unsigned long long* o; unsigned* s;
    (void)hipMalloc(&o, 16); (void)hipMalloc(&s, 256);
    const int iters = 2048;
    unsigned long long best[2] = {~0ull, ~0ull};
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, src, iters, stride, o, s);
        unsigned long long c[2]; (void)hipMemcpy(c, o, 16, hipMemcpyDeviceToHost);
        if (c[0] < best[0]) { best[0] = c[0]; best[1] = c[1]; }
    }
    printf("%-64s stride %6d B: issue %7.1f cycles / instruction, incl. landing %7.1f\n", name, stride, best[0] / (double)iters, best[1] / (double)iters);
    (void)hipFree(o); (void)hipFree(s);
}

int main() {
    char* src; (void)hipMalloc(&src, 64 << 20); (void)hipMemset(src, 1, 64 << 20);
    for (int stride : {0, 1024, 16384}) {
        run<0>("builtin raw_ptr_buffer_load_lds (compiler-managed m0)", src, stride);
        run<1>("asm: m0 saved, set, restored around every transfer (conv_dev.h)", src, stride);
        run<2>("asm: m0 set per transfer, not restored", src, stride);
        run<3>("asm: m0 written once, same LDS destination", src, stride);
    }
    return 0;
}
