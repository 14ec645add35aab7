// Layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands and E8M0 scales, for the f32 -> fp6
// conversion v_cvt_scalef32_2xpk16_fp6_f32 and its inverse v_cvt_scalef32_pk32_f32_fp6 (r03 design study for the
// fp16 + MX-fp6 line format).  Hypotheses under test:
//   MFMA : lane l holds, for row / col (l & 31), the 32 K-elements of K-half (l >> 5); element e occupies bits
//          6e .. 6e+5 of the lane's 192-bit operand (registers 0-5 of the 8-register operand); byte 0 of the scale
//          operand is the lane's block scale 2^(s-127); D uses the standard 32x32 accumulator map.
//   cvt  : 2xpk16(a, b, scale) -> element order printed (interleaved a0 b0 a1 b1 ... or concatenated), RNE, saturating,
//          value = x / scale.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(32))) float f32x32;
typedef __attribute__((ext_vector_type(6))) unsigned u32x6;

static float fp6v(unsigned c) {                      // e2m3: sign(1) exp(2) man(3), bias 1
    const int e = (c >> 3) & 3, m = c & 7;
    const float v = e ? ldexpf(1.f + m / 8.f, e - 1) : m / 8.f;
    return (c & 32) ? -v : v;
}
static unsigned field(const unsigned* w, int e) {    // 6-bit field e of a 192-bit little-endian bit string
    const int bit = 6 * e, d = bit >> 5, s = bit & 31;
    unsigned long long two = w[d] | ((unsigned long long)(d + 1 < 6 ? w[d + 1] : 0u) << 32);
    return (unsigned)(two >> s) & 63u;
}

__global__ void k_mfma(const unsigned* a, const unsigned* b, const int* sa, const int* sb, float* d) {
    const int l = threadIdx.x;
    i32x8 va = {0}, vb = {0};
    for (int i = 0; i < 6; ++i) { va[i] = a[l * 6 + i]; vb[i] = b[l * 6 + i]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc, 2, 2, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = acc[r];
}

__global__ void k_cvt(const float* in, float scale, unsigned* out, float* back) {
    const int l = threadIdx.x;
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[l * 32 + i]; b[i] = in[l * 32 + 16 + i]; }
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; ++i) out[l * 6 + i] = r[i];
    const f32x32 d = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(r, scale);
    for (int i = 0; i < 32; ++i) back[l * 32 + i] = d[i];
}

int main() {
    unsigned ha[64 * 6], hb[64 * 6]; int hsa[64], hsb[64];
    srand(3);
    for (int i = 0; i < 64 * 6; ++i) { ha[i] = ((unsigned)rand() << 16) ^ rand(); hb[i] = ((unsigned)rand() << 16) ^ rand(); }
    for (int i = 0; i < 64; ++i) { hsa[i] = 124 + rand() % 7; hsb[i] = 125 + rand() % 5; }
    unsigned *da, *db; int *dsa, *dsb; float* dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(da, db, dsa, dsb, dd);
    float hd[64 * 16];
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    double maxdiff = 0, maxref = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            double ref = 0;
            for (int h = 0; h < 2; ++h) {
                const int la = row + 32 * h, lb = col + 32 * h;
                double s = 0;
                for (int e = 0; e < 32; ++e) s += (double)fp6v(field(ha + la * 6, e)) * fp6v(field(hb + lb * 6, e));
                ref += s * ldexp(1.0, hsa[la] - 127) * ldexp(1.0, hsb[lb] - 127);
            }
            maxdiff = fmax(maxdiff, fabs(ref - hd[l * 16 + r])); maxref = fmax(maxref, fabs(ref));
        }
    printf("fp6 mfma, element e at bits 6e..6e+5: max |device - host| = %g (max |ref| %g)\n", maxdiff, maxref);

    // conversion: distinct values per position so the order is visible
    float hin[64 * 32];
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) hin[l * 32 + i] = (l == 0) ? (i < 16 ? 0.125f * (i + 1) : 2.0f + 0.25f * (i - 16)) : (float)((rand() % 2001) - 1000) / 120.f;
    float* din; unsigned* dout; float* dback;
    hipMalloc(&din, sizeof hin); hipMalloc(&dout, 64 * 6 * 4); hipMalloc(&dback, 64 * 32 * 4);
    hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 2.0f, 0.5f}) {
        k_cvt<<<1, 64>>>(din, scale, dout, dback);
        unsigned ho[64 * 6]; float hbk[64 * 32];
        hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(hbk, dback, sizeof hbk, hipMemcpyDeviceToHost);
        if (scale == 1.0f) {
            printf("cvt lane 0 (a = 0.125..2.0 step 0.125, b = 2.0..5.75 step 0.25), fields in bit order:\n ");
            for (int e = 0; e < 32; ++e) printf(" %g", fp6v(field(ho, e)));
            printf("\n decode (pk32_f32_fp6) element order:\n ");
            for (int e = 0; e < 32; ++e) printf(" %g", hbk[e]);
            printf("\n");
        }
        // find which order the conversion uses: interleaved (field 2i = a[i], 2i+1 = b[i]) or concatenated
        for (int order = 0; order < 2; ++order) {
            double worst = 0; int sat = 0;
            for (int l = 1; l < 64; ++l)
                for (int i = 0; i < 32; ++i) {
                    const int e = order == 0 ? (i < 16 ? 2 * i : 2 * (i - 16) + 1) : i;
                    const float q = fp6v(field(ho + l * 6, e)) * scale, x = hin[l * 32 + i];
                    const float xc = fminf(fmaxf(x, -7.5f * scale), 7.5f * scale);
                    sat += xc != x;
                    // RNE to the fp6 grid: the error never exceeds half the local step (0.0625 .. 0.25) * scale
                    worst = fmax(worst, fabs(q - xc) / scale);
                }
            printf("cvt scale %.1f order %s: worst |q - clamp(x)| / scale = %.4f (saturated inputs %d)\n", scale, order ? "concat" : "interleaved", worst, sat);
        }
        double wb = 0;
        for (int l = 1; l < 64; ++l)
            for (int e = 0; e < 32; ++e) wb = fmax(wb, fabs(hbk[l * 32 + e] - fp6v(field(ho + l * 6, e)) * scale));
        printf("   decode element e == field e * scale: max diff %g\n", wb);
    }
    // ties: RNE check at scale 1 on exact midpoints
    {
        float t[64 * 32] = {0};
        const float mids[8] = {0.0625f, 0.1875f, 1.0625f, 1.1875f, 2.125f, 2.375f, 4.25f, 4.75f};
        for (int i = 0; i < 8; ++i) t[i] = mids[i];
        t[8] = 7.6f; t[9] = 7.74f; t[10] = 7.76f; t[11] = 100.f; t[12] = -100.f; t[13] = 1e-9f;
        hipMemcpy(din, t, sizeof t, hipMemcpyHostToDevice);
        k_cvt<<<1, 64>>>(din, 1.0f, dout, dback);
        float hbk[64 * 32]; hipMemcpy(hbk, dback, sizeof hbk, hipMemcpyDeviceToHost);
        printf("ties / saturation (decode order unknown: all 32 printed):\n ");
        for (int e = 0; e < 32; ++e) printf(" %g", hbk[e]);
        printf("\n");
    }
    return 0;
}
