// Which per-lane address patterns does ds_read_b128 / ds_write_b128 serve without bank conflicts on gfx950?  (r05: layout of the
// 64-byte pixel records of the fused decoder tail, csrc/tail.hip.)  One wave, a dependent-free stream of LDS accesses at a fixed
// per-lane address, s_memtime around it: cycles per access (8 = full rate: 1024 bytes at 128 B / clk).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/lds_pattern tools/probes/lds_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(64) void probe(const int* addr, int write, int iters, unsigned long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(256))) char lds[65536];
    const int lane = threadIdx.x;
    for (int i = lane * 16; i < 65536; i += 1024) *(u32x4*)(lds + i) = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)addr[lane];
    u32x4 acc = {0u, 0u, 0u, 0u};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x4 v0, v1, v2, v3;
        if (write) {
            asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(a), "v"(acc) : "memory");
        } else {
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4\n\tds_read_b128 %3, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a) : "memory");
            acc += v0 + v1 + v2 + v3;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) *out = t1 - t0;
    sink[lane] = acc[0] + acc[1] + acc[2] + acc[3];
}

static double run(const std::vector<int>& addr, int write) {
    int* d; unsigned long long* o; unsigned* s;
    hipMalloc(&d, 256); hipMalloc(&o, 8); hipMalloc(&s, 256);
    hipMemcpy(d, addr.data(), 256, hipMemcpyHostToDevice);
    const int iters = 4096;
    unsigned long long best = ~0ull;
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, write, iters, o, s);
        unsigned long long c; hipMemcpy(&c, o, 8, hipMemcpyDeviceToHost);
        if (c < best) best = c;
    }
    hipFree(d); hipFree(o); hipFree(s);
    return (double)best / (iters * 4.0);
}

int main() {
    struct P { const char* name; int (*f)(int lane); };
    static const P pats[] = {
        {"128 B pitch, slot h ^ ((p >> 1) & 7)             (slab image; lane = pixel)", [](int l) { int p = l & 31, h = l >> 5; return p * 128 + ((h ^ ((p >> 1) & 7)) << 4); }},
        {"128 B pitch, unswizzled slot h", [](int l) { int p = l & 31, h = l >> 5; return p * 128 + (h << 4); }},
        {"64 B pitch, slot h ^ ((p >> 2) & 3)                (tail M ring, r05 first cut)", [](int l) { int p = l & 31, h = l >> 5; return p * 64 + ((h ^ ((p >> 2) & 3)) << 4); }},
        {"64 B pitch, same, pixels p + 1", [](int l) { int p = (l & 31) + 1, h = l >> 5; return p * 64 + ((h ^ ((p >> 2) & 3)) << 4); }},
        {"64 B pitch, slot h ^ key, key = 2 (p>>2 & 1) | (p>>3 & 1)", [](int l) { int p = l & 31, h = l >> 5; return p * 64 + ((h ^ ((((p >> 2) & 1) << 1) | ((p >> 3) & 1))) << 4); }},
        {"64 B pitch, unswizzled slot h", [](int l) { int p = l & 31, h = l >> 5; return p * 64 + (h << 4); }},
        {"64 B pitch, slot 2h (hi/lo 32 B apart) ^ ((p >> 2) & 3)", [](int l) { int p = l & 31, h = l >> 5; return p * 64 + (((2 * h) ^ ((p >> 2) & 3)) << 4); }},
        {"64 B pitch, stride-2 pixels (conv1 epilogue writes), slot h ^ ((p >> 2) & 3)", [](int l) { int p = 2 * (l & 31) + 1, h = l >> 5; return p * 64 + ((h ^ ((p >> 2) & 3)) << 4); }},
        {"64 B pitch, stride-2 pixels, slot h ^ ((p >> 3) & 3)", [](int l) { int p = 2 * (l & 31) + 1, h = l >> 5; return p * 64 + ((h ^ ((p >> 3) & 3)) << 4); }},
        {"linear 16 B per lane", [](int l) { return l * 16; }},
        {"all lanes one address (broadcast)", [](int l) { return 0; }},
    };
    for (const P& p : pats) {
        std::vector<int> a(64);
        for (int l = 0; l < 64; ++l) a[l] = p.f(l);
        printf("%-90s read %6.2f  write %6.2f cycles / access\n", p.name, run(a, 0), run(a, 1));
    }
    return 0;
}
