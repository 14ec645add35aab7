// Register-only MFMA rate probe (design study): per iteration either
//   mode 0: 6 x v_mfma_f32_32x32x16_bf16              (today's hi/lo scheme per tile step)
//   mode 1: 2 x v_mfma_f32_32x32x16_f16 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands
//   mode 2: same with fp6 operands,  mode 3: 3 x bf16 (half the work, reference point)
// Operands are random bit patterns kept in registers; one wave per SIMD x 4 waves per block x 1024 blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const int* seed, float* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    i32x8 ra, rb;
    for (int i = 0; i < 8; ++i) { ra[i] = seed[(t * 8 + i) & 65535]; rb[i] = seed[(t * 8 + i + 4096) & 65535]; }
    bf16x8 a0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(ra, ra, 0, 1, 2, 3));
    bf16x8 a1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(ra, ra, 4, 5, 6, 7));
    bf16x8 b0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(rb, rb, 0, 1, 2, 3));
    bf16x8 b1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(rb, rb, 4, 5, 6, 7));
    // keep magnitudes sane: clear the top exponent bits of every 16-bit element
    for (int i = 0; i < 8; ++i) {
        unsigned short u = __builtin_bit_cast(unsigned short, a0[i]); u &= 0xBFFF; a0[i] = __builtin_bit_cast(__bf16, u);
        u = __builtin_bit_cast(unsigned short, a1[i]); u &= 0xBFFF; a1[i] = __builtin_bit_cast(__bf16, u);
        u = __builtin_bit_cast(unsigned short, b0[i]); u &= 0xBFFF; b0[i] = __builtin_bit_cast(__bf16, u);
        u = __builtin_bit_cast(unsigned short, b1[i]); u &= 0xBFFF; b1[i] = __builtin_bit_cast(__bf16, u);
    }
    f16x8 ha0 = __builtin_bit_cast(f16x8, a0), hb0 = __builtin_bit_cast(f16x8, b0);
    f16x8 ha1 = __builtin_bit_cast(f16x8, a1), hb1 = __builtin_bit_cast(f16x8, b1);
    f32x16 acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc0, 0, 0, 0);
            if (MODE == 0) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc1, 0, 0, 0);
            }
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha0, hb0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha1, hb1, acc1, 0, 0, 0);
            // fmt codes: 2 = fp6 (e2m3), 4 = fp4 (e2m1); scales 127 = 2^0
            if (MODE == 1) acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ra, rb, acc0, 4, 4, 0, 127, 0, 127);
            else acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ra, rb, acc0, 2, 2, 0, 127, 0, 127);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[t] = s;
}

int main() {
    const int nblk = 1024, iters = 20000;
    int* seed; float* out;
    hipMalloc(&seed, 65536 * 4); hipMalloc(&out, nblk * 256 * 4);
    int* h = (int*)malloc(65536 * 4);
    srand(1); for (int i = 0; i < 65536; ++i) h[i] = (rand() << 16) ^ rand();
    hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"6x bf16 32x32x16", "2x f16 + 1x MX-fp4 32x32x64", "2x f16 + 1x MX-fp6 32x32x64", "3x bf16 32x32x16"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipEventRecord(e0);
            if (mode == 0) probe<0><<<nblk, 256>>>(seed, out, iters);
            if (mode == 1) probe<1><<<nblk, 256>>>(seed, out, iters);
            if (mode == 2) probe<2><<<nblk, 256>>>(seed, out, iters);
            if (mode == 3) probe<3><<<nblk, 256>>>(seed, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-32s %8.3f ms per %d iterations x %d waves  (%.1f ns/iteration/wave-slot)\n", names[mode], ms, iters, nblk * 4,
                   ms * 1e6 / iters / (nblk * 4 / 1024.0));
        }
    return 0;
}
