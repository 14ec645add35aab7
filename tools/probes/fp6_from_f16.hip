// Probe for a 96-byte activation line (r03 design note, DESIGN.md "what comes next"): the hi6 plane of the mx line format is the fp6
// (e2m3) image of the line's fp16 plane, so it need not be stored if the consumer can rebuild it bit for bit.  Question: does
// v_cvt_scalef32_pk32_fp6_f16 (32 halves -> 32 fp6 fields, one instruction) give exactly the fields that the producers' path
// gives today (fp16 -> f32 -> v_cvt_scalef32_2xpk16_fp6_f32), for EVERY fp16 bit pattern and every scale the format can choose?
// Element order under test: pk32 field e = input element e; 2xpk16 field 2i = a[i], 2i+1 = b[i].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(32))) _Float16 f16x32;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(6))) unsigned u32x6;

__global__ void k(float scale, unsigned* mism, unsigned* nanlike) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;      // 2048 threads x 32 values = all 65536 bit patterns
    f16x32 h;
    f32x16 a, b;
    for (int i = 0; i < 32; ++i) {
        const unsigned short bits = (unsigned short)(t * 32 + i);
        h[i] = __builtin_bit_cast(_Float16, bits);
    }
    for (int i = 0; i < 16; ++i) { a[i] = (float)h[2 * i]; b[i] = (float)h[2 * i + 1]; }
    u32x6 r16, r32;
    asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(r16) : "v"(h), "v"(scale));
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r32) : "v"(a), "v"(b), "v"(scale));
    for (int e = 0; e < 32; ++e) {
        const int bit = 6 * e, d = bit >> 5, s = bit & 31;
        unsigned long long w16 = r16[d] | ((unsigned long long)(d + 1 < 6 ? r16[d + 1] : 0u) << 32);
        unsigned long long w32 = r32[d] | ((unsigned long long)(d + 1 < 6 ? r32[d + 1] : 0u) << 32);
        const unsigned f16f = (unsigned)(w16 >> s) & 63u, f32f = (unsigned)(w32 >> s) & 63u;
        const unsigned short bits = (unsigned short)(t * 32 + e);
        const bool special = (bits & 0x7c00) == 0x7c00;       // inf / NaN inputs: never stored (the producers clamp to 65504)
        if (f16f != f32f) atomicAdd(special ? nanlike : mism, 1u);
    }
}

int main() {
    unsigned *d, h[2];
    if (hipMalloc(&d, 8) != hipSuccess) return 1;
    unsigned total = 0, total_special = 0;
    for (int s = 127 - 30; s <= 127 + 16; ++s) {              // E8M0 scale bytes the format picks for fp16 magnitudes (2^-24 .. 65504)
        const float scale = ldexpf(1.f, s - 127);
        hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, scale, d, d + 1);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        if (h[0] || h[1]) printf("scale 2^%d: %u finite mismatches, %u on inf/NaN inputs\n", s - 127, h[0], h[1]);
        total += h[0]; total_special += h[1];
    }
    printf("f16 -> fp6 (pk32) vs f16 -> f32 -> fp6 (2xpk16), 65536 bit patterns x 47 scales: %u mismatching fields on finite inputs, %u on inf/NaN\n",
           total, total_special);
    return 0;
}
