// Shared device/host definitions for the gfx950 WSI inference kernels.
//
// Activation layout ("padded-flat", PF) used by every conv kernel:
//   pixel index  q(n,y,x) = G + n*S + y*P + x      P = W+1, S = (H+1)*P, G = P+1
//   one shared zero column per row (x == W), one shared zero row per image (y == H) and G zero
//   guard pixels in front: a 3x3 tap (dy,dx) of pixel q is simply pixel q + dy*P + dx, never
//   out of bounds and zero where the reference zero-pads (resnets_shift.py:19-22, padding=1).
//   Kernels never store to pad positions, so a buffer zeroed once per plan stays valid.
//   per pixel: C*PLANES bf16, organised in 128-byte "lines":
//     PLANES=2 (bf16x2 split, parity mode): line l = channels 32l..32l+31 as [hi x32][lo x32]
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
//   2  bf16 hi + bf16 lo       line = 32 channels: [hi x32][lo x32]                  (3 MFMA passes)
//   3  fp16 hi + MX-fp4 cross  line = 32 channels: [hi fp16 x32 | lo4 16 B | hi4 16 B | scale_lo dword.. | scale_hi dword..]
//      x = hi + lo4*2^(scale_lo-127); the conv is  Wh*Xh (two fp16 MFMAs)  +  [Wh4*Xl4 | Wl4*Xh4] as the two
//      K halves of ONE v_mfma_scale_f32_32x32x64_f8f6f4 (block scale per 32 channels): 3 instructions per
//      (tile, line, tap) instead of 6 (tools/sim_mx_numerics.py, tools/probes/mx_*.hip)
template <int PLANES> struct PFmt {
    static constexpr int BPC = PLANES == 1 ? 2 : 4;          // bytes per channel in a pixel record
    static constexpr int CPL = PLANES == 1 ? 64 : 32;        // channels per 128-byte line
};
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
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
};

// ConvArgs.flags.  Product flags first; the CONV_ABL_* / study ones only act in builds with -DWSI_STUDY (bottleneck
// studies: tools/tune_conv.py), where some of them produce wrong or missing outputs by design.
enum : int {
    CONV_XCD_ORDER = 1,        // the channel blocks of one pixel tile get workgroup ids 8 apart (same XCD / L2)
    CONV_XCD_RANGES = 2,       // every XCD walks a contiguous range of pixel tiles (neighbours share halo rows in its L2)
    CONV_RESID_DIRECT = 4,     // A/B: residual read straight from memory instead of LDS-DMA staging
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

// mode 3 helpers ---------------------------------------------------------------------------
// E8M0 exponent byte s such that amax / 2^(s-127) <= 6 (fp4 e2m1 max); amax == 0 -> 0
static inline __host__ __device__ int mx4_scale_byte(float amax) {
    if (!(amax > 0.f)) return 0;
    const float t = amax * (1.0f / 6.0f);
    union { float f; unsigned u; } c;
    c.f = t;
    int e = (int)((c.u >> 23) & 255) + ((c.u & 0x7fffffu) ? 1 : 0);   // ceil(log2 t) + 127
    return e < 1 ? 1 : (e > 254 ? 254 : e);
}
static inline __host__ __device__ float mx4_scale_value(int s) {
    union { float f; unsigned u; } c;
    c.u = (unsigned)s << 23;                                  // 2^(s-127), s in [1,254]
    return c.f;
}
// Mode-3 channel order inside a 32-channel line.  The 32x32 MFMA accumulator leaves lane (pixel, h) with the
// 16 couts 8g + 4h + i (register r = 4g + i): storing them at line positions 16h + r makes every lane's share of
// the line contiguous (32 B of fp16, 8 B of each fp4 plane), so the epilogue and the residual read move 16-byte
// pieces.  K order inside a line is free as long as weights and pixels agree: prepack uses the same map.
static inline __host__ __device__ int mx_line_pos(int c) { return 16 * ((c >> 2) & 1) + 4 * (c >> 3) + (c & 3); }
static inline __host__ __device__ int mx_line_chan(int p) { return 8 * ((p & 15) >> 2) + 4 * (p >> 4) + (p & 3); }

// 8 floats -> one dword of fp4 (RNE(v / scale), saturating); the byte selector must be a literal
static __device__ __forceinline__ unsigned mx4_pack8(const float* v, float scale) {
    unsigned q = 0u;
    q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[0], v[1], scale, 0);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[2], v[3], scale, 1);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[4], v[5], scale, 2);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(q, v[6], v[7], scale, 3);
    return q;
}
// one dword of fp4 -> 8 floats (value * scale)
static __device__ __forceinline__ void mx4_unpack8(unsigned q, float scale, float* v) {
    const f32x2 a = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, scale, 0), b = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, scale, 1);
    const f32x2 c = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, scale, 2), d = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(q, scale, 3);
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1]; v[4] = c[0]; v[5] = c[1]; v[6] = d[0]; v[7] = d[1];
}

// decode one fp4 (e2m1) nibble
static inline __host__ __device__ float fp4_value(unsigned n) {
    const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    const float v = mag[n & 7];
    return (n & 8) ? -v : v;
}

// fp32 -> (hi, lo) bf16 pair with hi + lo == x to ~2^-17 relative
static inline __device__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
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
};
