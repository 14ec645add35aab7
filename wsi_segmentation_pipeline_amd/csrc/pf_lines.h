// Per-pixel line codecs of the padded-flat activation formats (common.h): one 128-byte line <-> its fp32 channel values.
// Shared by the layout converters / heads (heads.hip) and the U-Net decoder glue (unet.hip).
#pragma once
#include "common.h"

// nearest fp4 (e2m1) code, round-to-nearest-even, saturating (same rule as v_cvt_scalef32_pk_fp4_f32 and the host prepack)
static __device__ __forceinline__ unsigned fp4_encode_dev(float y) {
    const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    const float a = fabsf(y);
    int best = 7;
#pragma unroll
    for (int i = 6; i >= 0; --i) {
        const float mid = 0.5f * (mag[i] + mag[i + 1]);
        if (a < mid || (a == mid && (i & 1) == 0)) best = i;
    }
    return (unsigned)best | ((__float_as_uint(y) >> 31) ? 8u : 0u);
}

// mode-3 line codec for one pixel line (32 channels): v[32] -> 128 bytes, and back
static __device__ void mx_line_encode(const float* v, char* line) {
    float hi[32], lo[32], mh = 0.f, ml = 0.f;
    for (int c = 0; c < 32; ++c) {                        // c = line position (mx_line_pos order)
        const float t = v[mx_line_chan(c)];
        hi[c] = (float)(_Float16)t;
        lo[c] = t - hi[c];
        mh = fmaxf(mh, fabsf(hi[c]));
        ml = fmaxf(ml, fabsf(lo[c]));
    }
    const int sh = mx4_scale_byte(mh), sl = mx4_scale_byte(ml);
    const float ih = sh ? 1.0f / mx4_scale_value(sh) : 0.f, il = sl ? 1.0f / mx4_scale_value(sl) : 0.f;
    for (int c = 0; c < 32; ++c) ((_Float16*)line)[c] = (_Float16)hi[c];
    for (int b = 0; b < 16; ++b) {
        line[64 + b] = (char)(fp4_encode_dev(lo[2 * b] * il) | (fp4_encode_dev(lo[2 * b + 1] * il) << 4));
        line[80 + b] = (char)(fp4_encode_dev(hi[2 * b] * ih) | (fp4_encode_dev(hi[2 * b + 1] * ih) << 4));
    }
    for (int b = 0; b < 16; b += 4) {                     // scales replicated over their 16-byte slots
        *(unsigned*)(line + 96 + b) = (unsigned)sl;
        *(unsigned*)(line + 112 + b) = (unsigned)sh;
    }
}
static __device__ __forceinline__ float mx_line_decode(const char* line, int chan) {  // x = hi + lo4 * 2^(scale_lo-127)
    const int c = mx_line_pos(chan);
    const unsigned sl = *(const unsigned*)(line + 96) & 255u;
    const unsigned nib = ((unsigned)(unsigned char)line[64 + (c >> 1)] >> (4 * (c & 1))) & 15u;
    return (float)((const _Float16*)line)[c] + (sl ? fp4_value(nib) * mx4_scale_value((int)sl) : 0.f);
}


// any precision mode: line = 32 channels (planes 2, 3) or 64 (planes 1); v[] in channel order
template <int PLANES>
static __device__ __forceinline__ void pf_line_encode(const float* v, char* line) {
    if constexpr (PLANES == 3) {
        mx_line_encode(v, line);
    } else if constexpr (PLANES == 2) {
        for (int c = 0; c < 32; ++c) {
            __bf16 hi, lo;
            split_bf16(v[c], hi, lo);
            ((__bf16*)line)[c] = hi;
            ((__bf16*)(line + 64))[c] = lo;
        }
    } else {
        for (int c = 0; c < 64; ++c) ((__bf16*)line)[c] = (__bf16)v[c];
    }
}
template <int PLANES>
static __device__ __forceinline__ float pf_line_decode(const char* line, int chan) {
    if constexpr (PLANES == 3) return mx_line_decode(line, chan);
    else if constexpr (PLANES == 2) return (float)((const __bf16*)line)[chan] + (float)((const __bf16*)(line + 64))[chan];
    else return (float)((const __bf16*)line)[chan];
}
