// Per-pixel line codecs of the padded-flat activation formats (common.h): one 128-byte line <-> its fp32 channel values.
// Shared by the layout converters / heads (heads.hip) and the U-Net decoder glue (unet.hip).
#pragma once
#include "common.h"

// mode-3 line codec for one pixel line (32 channels): v[32] (channel order) -> 128 bytes, and back.  Scalar reference form
// of what the conv epilogues do with v_cvt_scalef32_2xpk16_fp6_f32 (same rounding, same scales, same layout).
static __device__ void mx_line_encode(const float* v, char* line) {
    float hi[32], lo[32], mh = 0.f, ml = 0.f;
    for (int c = 0; c < 32; ++c) {                        // c = fp16 line position (mx_line_pos order)
        const float t = fminf(fmaxf(v[mx_line_chan(c)], -65504.f), 65504.f);
        hi[c] = (float)(_Float16)t;
        lo[c] = t - hi[c];
        mh = fmaxf(mh, fabsf(hi[c]));
        ml = fmaxf(ml, fabsf(lo[c]));
    }
    const int sh = mx6_scale_byte(mh), sl = mx6_scale_byte(ml);
    const float ih = sh ? 1.0f / mx_scale_value(sh) : 0.f, il = sl ? 1.0f / mx_scale_value(sl) : 0.f;
    for (int c = 0; c < 32; ++c) ((_Float16*)line)[c] = (_Float16)hi[c];
    unsigned pl[6] = {0u, 0u, 0u, 0u, 0u, 0u}, ph[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        mx6_set_field(pl, mx6_field_of_pos(c), fp6_encode(lo[c] * il));
        mx6_set_field(ph, mx6_field_of_pos(c), fp6_encode(hi[c] * ih));
    }
    unsigned* w = (unsigned*)line;
    for (int d = 0; d < 4; ++d) { w[16 + d] = pl[d]; w[20 + d] = ph[d]; }
    w[24] = pl[4]; w[25] = pl[5]; w[26] = (unsigned)sl; w[27] = 0u;
    w[28] = ph[4]; w[29] = ph[5]; w[30] = (unsigned)sh; w[31] = 0u;
}
// the lo6 plane of a line as six dwords + its scale (0 for an all-zero block)
static __device__ __forceinline__ float mx_line_lo_plane(const char* line, unsigned (&pl)[6]) {
    const unsigned* w = (const unsigned*)line;
#pragma unroll
    for (int d = 0; d < 4; ++d) pl[d] = w[16 + d];
    pl[4] = w[24];
    pl[5] = w[25];
    const unsigned sl = w[26] & 255u;
    return sl ? mx_scale_value((int)sl) : 0.f;
}
static __device__ __forceinline__ float mx_line_decode(const char* line, int chan) {  // x = hi + lo6 * 2^(scale_lo-127)
    const int c = mx_line_pos(chan);
    unsigned pl[6];
    const float sc = mx_line_lo_plane(line, pl);
    return (float)((const _Float16*)line)[c] + fp6_value(mx6_get_field(pl, mx6_field_of_pos(c))) * sc;
}
// all 32 values of a line in fp16 POSITION order (position p holds channel mx_line_chan(p)): 16-byte loads + one
// v_cvt_scalef32_pk32_f32_fp6
static __device__ __forceinline__ void mx_line_decode_all(const char* line, float (&a)[32]) {
    const uint4* q = (const uint4*)line;
    const uint4 h0 = q[0], h1 = q[1], h2 = q[2], h3 = q[3], p0 = q[4], p1 = q[6];
    const unsigned sl = p1.z & 255u;
    const f32x32 d = mx6_unpack32(u32x6{p0.x, p0.y, p0.z, p0.w, p1.x, p1.y}, sl ? mx_scale_value((int)sl) : 0.f);
    const uint4 hh[4] = {h0, h1, h2, h3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f16x8 v = __builtin_bit_cast(f16x8, hh[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[8 * j + i] = (float)v[i] + d[2 * ((8 * j + i) & 15) + ((8 * j + i) >> 4)];
    }
}

// any precision mode: line = 32 channels (planes 2, 3) or 64 (planes 1); v[] in channel order
template <int PLANES>
static __device__ __forceinline__ void pf_line_encode(const float* v, char* line) {
    if constexpr (PLANES == 3) {
        mx_line_encode(v, line);
    } else if constexpr (PLANES == 2) {
        for (int c = 0; c < 32; ++c) {                         // fp16 pair (common.h split_f16)
            _Float16 hi, lo;
            split_f16(v[c], hi, lo);
            ((_Float16*)line)[c] = hi;
            ((_Float16*)(line + 64))[c] = lo;
        }
    } else {
        for (int c = 0; c < 64; ++c) ((__bf16*)line)[c] = (__bf16)v[c];
    }
}
template <int PLANES>
static __device__ __forceinline__ float pf_line_decode(const char* line, int chan) {
    if constexpr (PLANES == 3) return mx_line_decode(line, chan);
    else if constexpr (PLANES == 2) return (float)((const _Float16*)line)[chan] + (float)((const _Float16*)(line + 64))[chan];
    else return (float)((const __bf16*)line)[chan];
}
