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

struct PFGeom {          // padded-flat geometry of one activation tensor
    int N, H, W, C;
    int P, S, G;         // pitch, image stride, front guard (pixels)
    int NS;              // N*S
};

static inline __host__ __device__ PFGeom pf_geom(int n, int h, int w, int c) {
    PFGeom g;
    g.N = n; g.H = h; g.W = w; g.C = c;
    g.P = w + 1; g.S = (h + 1) * (w + 1); g.G = w + 2; g.NS = n * g.S;
    return g;
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

struct ConvArgs {
    const void* in;        // PF activations, PLANES planes
    void* out;             // PF activations
    const void* resid;     // PF activations shaped like out, or nullptr
    const void* wpk;       // fragment-packed weights [ntile][line][tap][f][lane][8] bf16
    const float* bias;     // folded BN bias, Cout floats
    PFGeom gi, go;         // input / output geometry
    int stride;            // 1 or 2
    int ksize;             // 3 or 1
    int relu;
    // fused 1x1 stride-2 downsample branch of the stride-2 kernel (all null when unused)
    void* out2;            // PF activations shaped like out
    const void* wpk2;      // 1x1 weights [ntile][line][1][f][lane][8]
    const float* bias2;
};

static inline __device__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
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
};
