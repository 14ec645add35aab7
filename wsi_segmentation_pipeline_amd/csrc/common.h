// Shared device/host definitions for the gfx950 WSI inference kernels.
//
// Activation layout ("padded-flat", PF) used by every conv kernel:
//   pixel index  q(n,y,x) = G + n*S + y*P + x      P = W+1, S = (H+1)*P, G = P+1
//   one shared zero column per row (x == W), one shared zero row per image (y == H) and G zero
//   guard pixels in front: a 3x3 tap (dy,dx) of pixel q is simply pixel q + dy*P + dx, never
//   out of bounds and zero where the reference zero-pads (resnets_shift.py:19-22, padding=1).
//   Kernels never store to pad positions, so a buffer zeroed once per plan stays valid.
//   per pixel: C*PLANES bf16, organised in 128-byte "lines":
//     PLANES=2 (fp16 pair, parity mode; r01-r04: bf16 pair): line l = channels 32l..32l+31 as [hi x32][lo x32]
//     PLANES=1 (single bf16, speed mode)  : line l = channels 64l..64l+63
//   A line is 8 slots of 16 bytes = 4 MFMA K-fragments f (slots 2f, 2f+1 by lane half h):
//     PLANES=2: f = plane*2 + ks (ks = 16-wide k-step inside the 32-channel chunk)
//     PLANES=1: f = ks           (4 k-steps inside the 64-channel chunk)
// Weights are packed host-side in exactly the per-lane MFMA operand order (1 KiB per fragment),
// so a wave's weight load is one fully coalesced global_load_dwordx4 and never touches LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define WSI_OK 0
#define WSI_EINVAL (-22)
#define WSI_ENOMEM (-12)
#define WSI_EFAULT (-14)

// Division of a non-negative int below 2^31 by a launch constant: q = mul_hi(i, m) >> s, exact for every such i (m = floor(2^(31+l)
// / d) + 1 with l = ceil(log2 d): the classic round-up magic, whose error d - (2^(31+l) mod d) <= 2^l keeps 31-bit dividends
// exact); two instructions instead of the ~20 of a runtime division.  r02: the tile prologues of the dense conv kernels spent
// 480-660 vector instructions per wave on `pixel index -> (n, y, x)`.
struct FastDiv { unsigned m; int s; };
static inline FastDiv fastdiv_make(int d) {
    FastDiv f;
    if (d <= 1) { f.m = 0u; f.s = -1; return f; }
    int l = 0;
    while ((1ll << l) < d) ++l;
    f.m = (unsigned)(((1ull << (31 + l)) / (unsigned long long)d) + 1ull);
    f.s = l - 1;
    return f;
}
#ifdef __HIPCC__
static inline __device__ int fd_div(int i, const FastDiv& f) { return f.s < 0 ? i : (int)(__umulhi((unsigned)i, f.m) >> f.s); }
#endif

struct PFGeom {          // padded-flat geometry of one activation tensor
    int N, H, W, C;
    int P, S, G;         // pitch, image stride, front guard (pixels)
    int NS;              // N*S
    FastDiv dHW, dW, dS, dP;   // filled by pf_geom_fd on the host (kernels that use them take their geometry from the launcher)
};

static inline __host__ __device__ PFGeom pf_geom(int n, int h, int w, int c) {
    PFGeom g;
    g.N = n; g.H = h; g.W = w; g.C = c;
    g.P = w + 1; g.S = (h + 1) * (w + 1); g.G = w + 2; g.NS = n * g.S;
    g.dHW.m = g.dW.m = g.dS.m = g.dP.m = 0u; g.dHW.s = g.dW.s = g.dS.s = g.dP.s = 0;
    return g;
}
static inline PFGeom pf_geom_fd(int n, int h, int w, int c) {          // host: geometry + the fast-division constants
    PFGeom g = pf_geom(n, h, w, c);
    g.dHW = fastdiv_make(h * w); g.dW = fastdiv_make(w); g.dS = fastdiv_make(g.S); g.dP = fastdiv_make(g.P);
    return g;
}
// pixel index i (n, y, x raster order over real pixels) -> PF position
static inline __device__ int pf_pos_of_index(const PFGeom& g, int i) {
    const int n = fd_div(i, g.dHW), rem = i - n * (g.H * g.W);
    const int y = fd_div(rem, g.dW), x = rem - y * g.W;
    return g.G + n * g.S + y * g.P + x;
}

#define PF_TILE_ROUND 512        // M tiles never exceed this many pixels

static inline __host__ __device__ long long pf_alloc_pixels(int n, int h, int w) {
    long long P = w + 1, S = (long long)(h + 1) * P, G = P + 1;
    long long body = ((long long)n * S + PF_TILE_ROUND - 1) / PF_TILE_ROUND * PF_TILE_ROUND;
    return 2 * G + body + PF_TILE_ROUND;
}

// q -> is it a real pixel (not a pad / guard position)?
static inline __device__ bool pf_is_pixel(const PFGeom& g, int q) {
    int r = q - g.G;
    if (r < 0 || r >= g.NS) return false;
    int x = r % g.P;
    int y = (r / g.P) % (g.H + 1);
    return x != g.W && y != g.H;
}

// Precision / storage modes ("planes" in the C ABI):
//   1  bf16 single             line = 64 channels x 2 B                              (speed mode)
//   2  fp16 hi + fp16 lo       line = 32 channels: [hi x32][lo x32]                  (3 MFMA passes; PairElem below)
//   3  fp16 hi + MX-fp6 cross  line = 32 channels: [hi fp16 x32 (64 B) | lo6 fields 0-20 (16 B) | hi6 fields 0-20 (16 B) |
//                              lo6 rest (8 B) scale_lo (4 B) pad | hi6 rest (8 B) scale_hi (4 B) pad]
//      x = hi + lo6*2^(scale_lo-127); the conv is  Wh*Xh (two fp16 MFMAs)  +  [Wh6*Xl6 | Wl6*Xh6] as the two K halves of ONE
//      v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands, one E8M0 block scale per 32 channels: 3 instructions per
//      (tile, line, tap) instead of the 6 of mode 2.  r01-r02 shipped fp4 (e2m1) cross terms in the same 128 bytes; fp6 runs
//      at the same MFMA rate (tools/probes/mx_rate.hip) with 5x finer operands (tests/studies/sim_mx_margin.py,
//      tools/probes/mx6_layout.hip)
template <int PLANES> struct PFmt {
    static constexpr int BPC = PLANES == 1 ? 2 : 4;          // bytes per channel in a pixel record
    static constexpr int CPL = PLANES == 1 ? 64 : 32;        // channels per 128-byte line
};
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

struct ConvArgs {
    const void* in;        // PF activations, PLANES planes
    void* out;             // PF activations
    const void* resid;     // PF activations shaped like out, or nullptr
    const void* wpk;       // fragment-packed weights [ntile][line][tap][f][lane][8] bf16
    const float* bias;     // folded BN bias, Cout floats
    PFGeom gi, go;         // input / output geometry
    int stride;            // 1 or 2
    int ksize;             // 3 or 1
    int relu;              // 0 / 1: ReLU after bias (+ residual)
    int flags;             // CONV_* below
    // fused 1x1 stride-2 downsample branch of the stride-2 kernel (all null when unused)
    void* out2;            // PF activations shaped like out
    const void* wpk2;      // 1x1 weights [ntile][line][1][f][lane][8]
    const float* bias2;
    // Phase-split tensors (the input layout of the stride-2 "wide" kernel): the four phase images (y&1, x&1) of an
    // (N,H,W,C) tensor, each a PF tensor of geometry (N,H/2,W/2,C), concatenated `*_split_pixels` pixels apart.
    // 0 = ordinary PF.  out_split_pixels: the epilogue writes `out` phase-split; in_split_pixels: `in` is phase-split.
    long long out_split_pixels, in_split_pixels;
    int mtiles = 0;                      // set by launchers whose grid is rounded up: number of real pixel tiles
    // Extra K segment of a stride-1 3x3 conv (mode 3, wide kernel): the 1x1 stride-2 downsample branch of a strided
    // BasicBlock (resnets_shift.py:173-177) computed INSIDE the block's second conv - `in2` = phase 00 of the block input
    // (a PF tensor of the OUTPUT's pixel geometry with in2_c channels), wpk2 / bias2 its packed 1x1 weights and folded BN bias:
    // out = relu(conv3x3(in) + bias + conv1x1(in2) + bias2), no downsample tensor, no residual read.  null = none.
    const void* in2 = nullptr;
    int in2_c = 0;
    // Fused `F.interpolate(x, scale_factor=2, mode='nearest')` + `torch.cat([x, skip], 1)` of the U-Net decoder blocks
    // (segmentation_models_pytorch DecoderBlock, called at /root/reference/utils/eval.py:199-200) as the conv's INPUT (slab3
    // kernel, stride-1 3x3): the first up_c of the gi.C input channels are read from `in_up`, a PF tensor of HALF the map size
    // (geometry gup: pixel (y, x) of the conv input = pixel (y >> 1, x >> 1) of in_up), the remaining gi.C - up_c from `in`, the
    // skip tensor at full size with gi.C - up_c channels (null when up_c == gi.C).  null = ordinary input.
    const void* in_up = nullptr;
    int up_c = 0;
    PFGeom gup = {};
    // 96-byte-line tensors (CONV_IN96 / OUT96 / RESID96) are LINE-PLANAR since r05: all pixels' line 0 (96 bytes each, pixel stride 96),
    // then all pixels' line 1, ... - plane96 = bytes from one 32-channel line plane to the next (in, out and resid of a layer-1 conv share
    // one geometry and one plan).  r03-r04 interleaved the lines per pixel (96 of every 192 bytes per slab load); tools/probes/slab_pattern
    // measured that pattern at 3.8 TB/s against 7.1-7.3 TB/s for contiguous 96-byte lines (profiles/r05_slab_pattern_probe.txt).
    long long plane96 = 0;
};

// ConvArgs.flags.  Product flags first; the CONV_ABL_* / study ones only act in builds with -DWSI_STUDY (bottleneck
// studies: tools/tune_conv.py), where some of them produce wrong or missing outputs by design.
enum : int {
    CONV_XCD_ORDER = 1,        // the channel blocks of one pixel tile get workgroup ids 8 apart (same XCD / L2)
    CONV_XCD_RANGES = 2,       // every XCD walks a contiguous range of pixel tiles (neighbours share halo rows in its L2)
    CONV_RESID_DIRECT = 4,     // A/B: residual read straight from memory instead of LDS-DMA staging
    // 96-byte activation lines (mode 3; the trunk's 64-channel layer 1, which is HBM-bound): the line in memory is
    // [fp16 plane 64 B][lo6 dwords 0-3][lo6 dwords 4-5, scale_lo, scale_hi] - the hi6 plane, which is the fp6 image of the fp16
    // plane, is not stored; a consumer rebuilds it in LDS after its slab has landed (conv_dev.h mx96_rebuild_hi6), where lines
    // keep their 128-byte pitch.  The tensor is line-planar (ConvArgs.plane96): pixel stride 96 bytes inside a line plane.
    CONV_IN96 = 16, CONV_OUT96 = 32, CONV_RESID96 = 64,
    CONV_ABL_NO_STORE = 1 << 8, CONV_ABL_DISPATCH_ONLY = 1 << 9, CONV_ABL_NO_MAINLOOP = 1 << 10, CONV_NONTEMPORAL = 1 << 11,
    CONV_WCOPIES_SHIFT = 16,   // bits 16-19: back-to-back copies of the packed weights minus one
};
#ifdef WSI_STUDY
#define CONV_STUDY(a, bits) ((a).flags & (bits))
#else
#define CONV_STUDY(a, bits) 0
#endif

// byte offset of output position q (a real pixel of geometry g) in an ordinary or phase-split tensor
static inline __device__ size_t pf_out_offset(const PFGeom& g, long long split_pixels, int q, size_t pixstride) {
    if (!split_pixels) return (size_t)q * pixstride;
    int r = q - g.G;
    const int n = fd_div(r, g.dS);
    r -= n * g.S;
    const int y = fd_div(r, g.dP), x = r - y * g.P;
    const int P2 = g.W / 2 + 1, S2 = (g.H / 2 + 1) * P2;
    const long long q2 = (long long)(P2 + 1) + (long long)n * S2 + (y >> 1) * P2 + (x >> 1);
    return (size_t)(((y & 1) * 2 + (x & 1)) * split_pixels + q2) * pixstride;
}

static inline __device__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 16-bit element type of the operand planes of a precision mode and its MFMA.  Mode 2 (the split-precision "parity" mode) is an
// fp16 PAIR since r05: x ~ hi + lo with hi = fp16(x), lo = fp16(x - hi) - 22 significand bits where both are normal, an absolute
// error of 2^-25 where lo is an fp16 subnormal (|x| < 2^-3; v_mfma_f32_32x32x16_f16 reads subnormals, tools/probes/f16_denorm.hip)
// - against the 16 bits of the bf16 pair of r01-r04 (2^-18 relative), at the same three MFMA passes.  On the dense per-pixel path
// the bf16 pair sat AT the 1e-3 logit contract (1.01e-3 at |logit| 16, r04); the CPU model of both pairs
// (tests/studies/sim_seg_precision.py) puts the fp16 pair 10-20x lower.  Fragments travel as opaque 16-byte `bf16x8` values in
// every mode; only the MFMA and the epilogue conversions know the element type.
template <int PLANES> struct PairElem { typedef __bf16 T; };
template <> struct PairElem<2> { typedef _Float16 T; };
template <int PLANES>
static inline __device__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    if constexpr (PLANES == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// mode 3 helpers ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(6))) unsigned u32x6;
typedef __attribute__((ext_vector_type(16))) unsigned u32x16;
typedef __attribute__((ext_vector_type(32))) float f32x32;
// E8M0 exponent byte s of a block with largest magnitude amax: the smallest s with amax / 2^(s-127) <= 7.75, i.e. the
// largest element lands in fp6's top binade [4, 7.5] (values in (7.5, 7.75] saturate to 7.5: an error below the top
// binade's own rounding error of 0.25).  Integer arithmetic on the float's bits: amax = m * 2^e with m in [1, 2), and
// 7.75 = 1.9375 * 2^2, so s - 127 = e - 2 (+ 1 if m > 1.9375).  amax == 0 -> 0 (the block decodes to zeros).
static inline __host__ __device__ int mx6_scale_byte(float amax) {
    union { float f; unsigned u; } c;
    c.f = amax;
    const unsigned u = c.u & 0x7fffffffu;
    if (u == 0u) return 0;
    const int e = (int)(u >> 23) - 2 + ((u & 0x7fffffu) > 0x780000u ? 1 : 0);
    return e < 1 ? 1 : (e > 254 ? 254 : e);
}
static inline __host__ __device__ float mx_scale_value(int s) {
    union { float f; unsigned u; } c;
    c.u = (unsigned)s << 23;                                  // 2^(s-127), s in [1,254]
    return c.f;
}
// Mode-3 channel order inside a 32-channel line.  The 32x32 MFMA accumulator leaves lane (pixel, h) with the
// 16 couts 8g + 4h + i (register r = 4g + i).  fp16 plane: line position 16h + r, so every lane's share is 32
// contiguous bytes and the epilogue moves 16-byte pieces.  fp6 planes: field 2r + h - what
// v_cvt_scalef32_2xpk16_fp6_f32 produces from (values of the h=0 lane, values of the h=1 lane), which interleaves its
// two sources.  K order inside a line is free as long as weights and pixels agree: prepack uses the same maps.
static inline __host__ __device__ int mx_line_pos(int c) { return 16 * ((c >> 2) & 1) + 4 * (c >> 3) + (c & 3); }
static inline __host__ __device__ int mx_line_chan(int p) { return 8 * ((p & 15) >> 2) + 4 * (p >> 4) + (p & 3); }
static inline __host__ __device__ int mx6_field_of_pos(int p) { return 2 * (p & 15) + (p >> 4); }      // fp16 position -> fp6 field
static inline __host__ __device__ int mx6_field_chan(int f) { return mx_line_chan(16 * (f & 1) + (f >> 1)); }
// byte offsets inside a 128-byte line: plane q (0 = lo6, 1 = hi6) keeps fields 0..20 (+ 2 bits of 21) in slot 4 + q and
// its last 64 bits in slot 6 + q, followed by the plane's scale dword
#define MX6_PLANE_LO(q) (64 + 16 * (q))
#define MX6_PLANE_HI(q) (96 + 16 * (q))
#define MX6_SCALE(q) (104 + 16 * (q))

// fp6 (e2m3) value of a 6-bit code / nearest code of y (round-to-nearest-even, saturating at 7.5: the rule of
// v_cvt_scalef32_2xpk16_fp6_f32, tools/probes/mx6_layout.hip)
static inline __host__ __device__ float fp6_value(unsigned c) {
    const int e = (c >> 3) & 3, m = c & 7;
    const float v = e ? (float)(8 + m) * (e == 1 ? 0.125f : e == 2 ? 0.25f : 0.5f) : (float)m * 0.125f;
    return (c & 32) ? -v : v;
}
static inline __host__ __device__ unsigned fp6_encode(float y) {
    union { float f; unsigned u; } c;
    c.f = y;
    const unsigned sign = (c.u >> 31) ? 32u : 0u;
    c.u &= 0x7fffffffu;
    const float a = c.f;
    if (!(a < 7.5f)) return sign | 31u;                       // saturate (also NaN)
    // step of the grid at a: 0.125 below 2, 0.25 in [2, 4), 0.5 in [4, 7.5]
    const float step = a < 2.f ? 0.125f : a < 4.f ? 0.25f : 0.5f;
    const float t = a / step;                                 // exact (power-of-two step)
    float r = (float)(int)t;
    const float frac = t - r;
    if (frac > 0.5f || (frac == 0.5f && ((int)r & 1))) r += 1.f;
    const float q = r * step;                                 // on the grid, possibly the first value of the next binade
    unsigned code;
    if (q < 1.f) code = (unsigned)(q * 8.f);                  // subnormals m / 8
    else if (q < 2.f) code = 8u + (unsigned)((q - 1.f) * 8.f);
    else if (q < 4.f) code = 16u + (unsigned)((q - 2.f) * 4.f);
    else code = 24u + (unsigned)((q - 4.f) * 2.f);
    return sign | code;
}
// 6-bit field f of a plane stored as six little-endian dwords
static inline __host__ __device__ unsigned mx6_get_field(const unsigned* w, int f) {
    const int bit = 6 * f, d = bit >> 5, s = bit & 31;
    unsigned v = w[d] >> s;
    if (s > 26) v |= w[d + 1] << (32 - s);
    return v & 63u;
}
static inline __host__ __device__ void mx6_set_field(unsigned* w, int f, unsigned code) {
    const int bit = 6 * f, d = bit >> 5, s = bit & 31;
    w[d] |= code << s;
    if (s > 26) w[d + 1] |= code >> (32 - s);
}

#ifdef __HIPCC__
// 2 x 16 floats -> 32 fp6 fields (RNE(v / scale), saturating): field 2i = a[i], field 2i + 1 = b[i].  Inline asm with an
// early-clobber result: hipcc (ROCm 7.2) lets the builtin's 6-register result overlap the tail of its second source,
// which the instruction has not finished reading when it writes (tools/probes/mx6_layout.hip: the last two fields of b
// came out wrong).
static __device__ __forceinline__ u32x6 mx6_pack32(f32x16 a, f32x16 b, float scale) {
    u32x6 r;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(scale));
    return r;
}
// 32 fp6 fields -> 32 floats (field * scale), same early-clobber caution
static __device__ __forceinline__ f32x32 mx6_unpack32(u32x6 q, float scale) {
    f32x32 r;
    asm volatile("v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2" : "=&v"(r) : "v"(q), "v"(scale));
    return r;
}
// Exchange between the two lanes of a pixel (l, l ^ 32): lanes 32-63 of a[i] <-> lanes 0-31 of b[i].  After it a lane of
// the lower half holds (its own a, the partner's a), a lane of the upper half (the partner's b, its own b).
// v_permlane32_swap_b32 by inline asm: at -O1 and above hipcc (ROCm 7.2) folds BOTH results of
// __builtin_amdgcn_permlane32_swap into the first.  The leading s_nop covers the two wait states between a VALU write
// of an operand and the swap (hipcc pads nothing inside asm).
static __device__ __forceinline__ void swap32_halves(f32x16& a, f32x16& b) {
#pragma unroll
    for (int k = 0; k < 16; k += 8) {
        float a0 = a[k], a1 = a[k + 1], a2 = a[k + 2], a3 = a[k + 3], a4 = a[k + 4], a5 = a[k + 5], a6 = a[k + 6], a7 = a[k + 7];
        float b0 = b[k], b1 = b[k + 1], b2 = b[k + 2], b3 = b[k + 3], b4 = b[k + 4], b5 = b[k + 5], b6 = b[k + 6], b7 = b[k + 7];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\t"
                     "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
                     "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15\n\ts_nop 1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                       "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        a[k] = a0; a[k + 1] = a1; a[k + 2] = a2; a[k + 3] = a3; a[k + 4] = a4; a[k + 5] = a5; a[k + 6] = a6; a[k + 7] = a7;
        b[k] = b0; b[k + 1] = b1; b[k + 2] = b2; b[k + 3] = b3; b[k + 4] = b4; b[k + 5] = b5; b[k + 6] = b6; b[k + 7] = b7;
    }
}
// max over the two lanes of a pixel (l, l ^ 32) of two values at once, by one v_permlane32_swap each (see swap32_halves)
static __device__ __forceinline__ void pair_max2(float& x, float& y) {
    float x2 = x, y2 = y;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1" : "+v"(x), "+v"(x2), "+v"(y), "+v"(y2));
    x = fmaxf(x, x2);
    y = fmaxf(y, y2);
}

#endif

// fp32 -> (hi, lo) fp16 pair of mode 2: hi + lo == clamp(x, +-65504) to 2^-22 relative / 2^-25 absolute (see PairElem)
static inline __device__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}
static inline __device__ void split_f16x4(const float* v, f16x4& hi, f16x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        _Float16 a, b;
        split_f16(v[i], a, b);
        hi[i] = a;
        lo[i] = b;
    }
}
// Mode-2 packed weights carry one power-of-two scale per output channel (capi.hip wsi_prepack_conv: every channel's largest
// |weight| is moved into [2^13, 2^14), so the lo plane of every weight within 2^-16 of it is a NORMAL fp16 and weights of any
// magnitude fit the format): the inverse scales, Cout floats, follow the fragment blocks of the pack; the epilogue multiplies
// the accumulator by them (exact) before it adds the bias.
static inline __device__ const float* conv_wscale_inv(const void* wpk, int cout, int cin, int ksize) {
    return (const float*)((const char*)wpk + (size_t)(cout / 32) * (cin / 32) * ksize * ksize * 4096);
}

struct StemArgs {
    // mode 0: f32 NCHW input; mode 1: u8 HWC slide + per-tile origins + LUT
    int mode;
    const float* in_f32;       // [N][3][H][W]
    const uint8_t* slide;      // [SH][pitch] bytes, 3 bytes per pixel
    long long slide_pitch;     // bytes per slide row
    int SH, SW;
    const int* origins;        // [N][2] (x, y) tile corner in slide pixels
    const float* lut;          // [3][256] normalised value of each u8 code
    const void* wpk;           // [nt 2][s 14][plane][lane 64][8] bf16
    const float* bias;         // 64
    float* out;                // [N][H/2][W/2][64] f32 (post ReLU)
    int N, H, W;               // patch size
    // integer path of the fused kernel (mode 1, split precision): pixels as i8 (x - 128) + an inside byte, weights in balanced
    // base-256 digits with the normalisation and BN folded in (capi.hip wsi_prepack_stem_u8, stem.hip stem_pool_kernel<.., DIG>)
    const void* wpk_u8;        // [nt 2][kh 7][digit][lane 64][16] i8 + float scale[64], or null: LUT path
    const float* bias_u8;      // 64
    // unfused conv kernel (stem_conv7x7_kernel): write the post-ReLU map as PF lines of `out_planes` instead of f32 NHWC into `out`
    // (the U-Net's half-resolution skip x0; r04: was f32 scratch + a re-encode pass)
    void* out_pf = nullptr;
    int out_planes = 0;
};
