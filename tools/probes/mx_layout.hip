// Layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1) operands and E8M0 scales, and for
// the f32 -> fp4 conversion builtin.  Hypothesis under test: lane l holds, for row/col (l & 31), the 32
// K-elements of K-half (l >> 5): element e in dword e/8, nibble e%8 (low nibble first); scale byte 0 of the
// scale operand is that lane's block scale (2^(s-127)); D uses the standard 32x32 accumulator map.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
static const float FP4V[16] = {0, .5f, 1, 1.5f, 2, 3, 4, 6, -0.f, -.5f, -1, -1.5f, -2, -3, -4, -6};

__global__ void k_mfma(const unsigned* a, const unsigned* b, const int* sa, const int* sb, float* d) {
    const int l = threadIdx.x;
    i32x8 va = {0}, vb = {0};
    for (int i = 0; i < 4; ++i) { va[i] = a[l * 4 + i]; vb[i] = b[l * 4 + i]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 4, 4, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = acc[r];
}

__global__ void k_cvt(const float* in, float scale, unsigned* out) {
    // two floats -> one byte (two fp4) in byte 0 of the destination dword
    const int l = threadIdx.x;
    unsigned v = 0;
    v = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(v, in[2 * l], in[2 * l + 1], scale, 0);
    out[l] = v;
}

int main() {
    unsigned ha[64 * 4], hb[64 * 4]; int hsa[64], hsb[64];
    srand(3);
    for (int i = 0; i < 256; ++i) { ha[i] = ((unsigned)rand() << 16) ^ rand(); hb[i] = ((unsigned)rand() << 16) ^ rand(); }
    for (int i = 0; i < 64; ++i) { hsa[i] = 124 + rand() % 7; hsb[i] = 125 + rand() % 5; }
    unsigned *da, *db; int *dsa, *dsb; float* dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(da, db, dsa, dsb, dd);
    float hd[64 * 16];
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    for (int order = 0; order < 2; ++order) {          // 0: low nibble = even element, 1: high nibble = even element
        double maxdiff = 0, maxref = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int h = 0; h < 2; ++h) {
                    const int la = row + 32 * h, lb = col + 32 * h;
                    double s = 0;
                    for (int e = 0; e < 32; ++e) {
                        int sh = 4 * (e & 7); if (order) sh ^= 4;
                        const unsigned na = (ha[la * 4 + e / 8] >> sh) & 15, nb = (hb[lb * 4 + e / 8] >> sh) & 15;
                        s += (double)FP4V[na] * FP4V[nb];
                    }
                    ref += s * ldexp(1.0, hsa[la] - 127) * ldexp(1.0, hsb[lb] - 127);
                }
                maxdiff = fmax(maxdiff, fabs(ref - hd[l * 16 + r])); maxref = fmax(maxref, fabs(ref));
            }
        printf("nibble order %d: max |device - host| = %g (max |ref| %g)\n", order, maxdiff, maxref);
    }
    // conversion builtin
    float hin[128]; const float vals[16] = {0.3f, 1.2f, -2.6f, 5.0f, 7.0f, 0.1f, 0.74f, 0.76f, 1.25f, 1.75f, -0.25f, 2.5f, 3.5f, -6.5f, 0.0f, 100.f};
    for (int i = 0; i < 128; ++i) hin[i] = vals[i % 16];
    float* din; unsigned* dout; hipMalloc(&din, sizeof hin); hipMalloc(&dout, 64 * 4);
    hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 2.0f, 0.5f}) {
        k_cvt<<<1, 64>>>(din, scale, dout);
        unsigned ho[64]; hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
        printf("cvt scale %.1f:", scale);
        for (int i = 0; i < 8; ++i) printf("  (%.2f,%.2f)->%02x=(%.1f,%.1f)", hin[2 * i], hin[2 * i + 1], ho[i] & 255, FP4V[ho[i] & 15], FP4V[(ho[i] >> 4) & 15]);
        printf("\n");
    }
    return 0;
}
