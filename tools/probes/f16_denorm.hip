// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL operands (r05 design question for the fp16 hi + fp16 lo operand pair:
// the lo plane of every value below 0.125 is an fp16 subnormal)?  One wave: A = the subnormal 2^-20 (code 0x0010) in every
// element, B = 1.0 -> each output is 16 * 2^-20 = 2^-16 if subnormals are read, 0 if they are flushed.  Also: what
// v_cvt_pk_f16_f32 (RNE) returns for 3 * 2^-22 (exact subnormal 0x000C expected) and fp16 -> fp32 of a subnormal.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(float* out, float v) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = __builtin_bit_cast(_Float16, (unsigned short)0x0010);
        b[i] = (_Float16)1.0f;
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    f32x16 acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);
    const f16x2 h = __builtin_convertvector(f32x2{v, v * 0.5f}, f16x2);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        out[1] = acc2[0];
        out[2] = (float)__builtin_bit_cast(unsigned short, h[0]);
        out[3] = (float)h[0];
        out[4] = (float)__builtin_bit_cast(unsigned short, h[1]);
    }
}
int main() {
    float* d;
    hipMalloc(&d, 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 3.0f / 4194304.0f);
    float h[5];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mfma f16, A subnormal 2^-20 x B 1.0, K = 16: %.9g (expected 2^-16 = %.9g; 0 = flushed)\n", h[0], 1.0 / 65536);
    printf("mfma f16, A 1.0 x B subnormal:               %.9g\n", h[1]);
    printf("v_cvt_pk_f16_f32(3 * 2^-22) = 0x%04x (expected 0x000c), back to f32 %.9g (expected %.9g); (1.5 * 2^-22) -> 0x%04x (expected 0x0006)\n",
           (unsigned)h[2], h[3], 3.0 / 4194304, (unsigned)h[4]);
    return 0;
}
