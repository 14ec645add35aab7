// Stem: (tile fetch +) colour normalisation + conv 7x7 stride 2 pad 3 (3->64) + folded BN + ReLU on
// MFMA, then maxpool 3x3 stride 2 pad 1 into the padded-flat layout.
// Replaces /root/reference/resnets_shift.py:196-199 (conv1, bn1, relu, maxpool) together with the
// eval transform /root/reference/utils/preprocessing.py:209-212 (ToTensor + Normalize) and the tile
// read /root/reference/utils/dataset.py:174-178 when the input is a u8 slide.
//
// K ordering for the MFMA: k = kh*32 + kw*4 + c with kw padded 7->8 and c padded 3->4 (zero
// weights), so K = 224 = 14 k-steps of 16 and a lane's 8 consecutive k are two neighbouring input
// pixels (4 channels each) = one aligned 16-byte LDS read of the [row][col][4ch] bf16 image.
#include "common.h"

constexpr int STEM_TR = 8;                       // conv-output rows per tile
constexpr int STEM_TC = 32;                      // conv-output cols per tile
constexpr int STEM_LR = 2 * STEM_TR + 5;         // 21 input rows
constexpr int STEM_LC = 70;                      // 2*32 + 6 input cols
constexpr int STEM_PLANE_BYTES = STEM_LR * STEM_LC * 8;

template <int PLANES>
__global__ __launch_bounds__(256, 2) void stem_conv7x7_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                  // weights: 28*PLANES KiB
    char* xl = smem + 28 * PLANES * 1024;             // input image planes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int Hc = a.H / 2, Wc = a.W / 2;
    const int tiles_x = (Wc + STEM_TC - 1) / STEM_TC, tiles_y = Hc / STEM_TR;
    const int total = a.N * tiles_x * tiles_y;

    for (int i = tid; i < 28 * PLANES * 64; i += 256) ((uint4*)wl)[i] = ((const uint4*)a.wpk)[i];

    float bias[2][16];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[nt][r] = a.bias[nt * 32 + 8 * (r >> 2) + 4 * h + (r & 3)];

    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y);
        const int trem = tile - n * tiles_x * tiles_y;
        const int oy0 = (trem / tiles_x) * STEM_TR, ox0 = (trem % tiles_x) * STEM_TC;
        __syncthreads();                              // previous tile's readers done (and weights landed)
        // ---- stage the normalised input patch as bf16 (hi, lo) [row][col][4] ------------------
        int tx = 0, ty = 0;
        if (a.mode == 1) { tx = a.origins[2 * n]; ty = a.origins[2 * n + 1]; }
        for (int i = tid; i < STEM_LR * STEM_LC; i += 256) {
            const int r = i / STEM_LC, cc = i - r * STEM_LC;
            const int iy = 2 * oy0 - 3 + r, ix = 2 * ox0 - 3 + cc;
            float v[3] = {0.f, 0.f, 0.f};
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                if (a.mode == 0) {
                    const size_t base = ((size_t)n * 3 * a.H + iy) * a.W + ix;
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = a.in_f32[base + (size_t)c * a.H * a.W];
                } else {
                    const int sx = tx + ix, sy = ty + iy;
                    uint8_t px[3] = {0, 0, 0};       // outside the slide OpenSlide pads with black
                    if (sx >= 0 && sx < a.SW && sy >= 0 && sy < a.SH) {
                        const uint8_t* p = a.slide + (size_t)sy * a.slide_pitch + (size_t)sx * 3;
                        px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[c] = a.lut[c * 256 + px[c]];
                }
            }
            typedef typename PairElem<PLANES>::T E;       // PLANES 2: fp16 pair (common.h), 1: bf16
            typedef __attribute__((ext_vector_type(4))) E Ex4;
            Ex4 hi, lo;
#pragma unroll
            for (int c = 0; c < 3; ++c) { hi[c] = (E)v[c]; lo[c] = (E)(v[c] - (float)hi[c]); }
            hi[3] = (E)0.f; lo[3] = (E)0.f;
            *(Ex4*)(xl + (size_t)i * 8) = hi;
            if constexpr (PLANES == 2) *(Ex4*)(xl + STEM_PLANE_BYTES + (size_t)i * 8) = lo;
        }
        __syncthreads();

        f32x16 acc[2][2];                             // [nt][mt]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;

#pragma unroll 2
        for (int s = 0; s < 14; ++s) {
            const int kh = s >> 1, kw0 = (s & 1) * 4 + 2 * h;
            bf16x8 wf[2][PLANES], xf[2][PLANES];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int p = 0; p < PLANES; ++p)
                    wf[nt][p] = *(const bf16x8*)(wl + ((size_t)((nt * 14 + s) * PLANES + p) * 64 + lane) * 16);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = 2 * (2 * wave + mt) + kh, col = 2 * l31 + kw0;
#pragma unroll
                for (int p = 0; p < PLANES; ++p)
                    xf[mt][p] = *(const bf16x8*)(xl + p * STEM_PLANE_BYTES + (size_t)(row * STEM_LC + col) * 8);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if constexpr (PLANES == 2) {
                        acc[nt][mt] = mfma16<PLANES>(wf[nt][1], xf[mt][0], acc[nt][mt]);
                        acc[nt][mt] = mfma16<PLANES>(wf[nt][0], xf[mt][1], acc[nt][mt]);
                    }
                    acc[nt][mt] = mfma16<PLANES>(wf[nt][0], xf[mt][0], acc[nt][mt]);
                }
        }
        // ---- epilogue: bias + ReLU -> PF lines (a.out_pf) or f32 NHWC --------------------------
        const int ox = ox0 + l31;
        if (a.out_pf) {
            // Lane (column, h) holds channels nt * 32 + 8g + 4h + i of its pixel: the conv kernels' accumulator layout, so the line
            // encodes are theirs (conv_dev.h conv_epilogue_q / conv_epilogue_mx; pf_lines.h is the scalar reference form)
            const PFGeom go = pf_geom(a.N, Hc, Wc, 64);
            const int bpc = a.out_planes == 1 ? 2 : 4;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int oy = oy0 + 2 * wave + mt;
                char* o = (char*)a.out_pf + (size_t)(go.G + n * go.S + oy * go.P + (ox < Wc ? ox : 0)) * (64 * bpc);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[nt][mt][r] + bias[nt][r], 0.f);
                    if (a.out_planes == 3) {
                        f32x16 hi, lo;
                        f16x8 hv[2];
                        float mh = 0.f, ml = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const float v0 = fminf(v[r], 65504.f), v1 = fminf(v[r + 1], 65504.f);
                            const f16x2 hh = __builtin_convertvector(f32x2{v0, v1}, f16x2);
                            hv[r >> 3][r & 7] = hh[0];
                            hv[r >> 3][(r & 7) + 1] = hh[1];
                            hi[r] = (float)hh[0];
                            hi[r + 1] = (float)hh[1];
                            lo[r] = v0 - hi[r];
                            lo[r + 1] = v1 - hi[r + 1];
                            mh = fmaxf(mh, fmaxf(hi[r], hi[r + 1]));
                            ml = fmaxf(ml, fmaxf(fabsf(lo[r]), fabsf(lo[r + 1])));
                        }
                        pair_max2(mh, ml);                                      // (every lane takes part: the stores below are masked, not these)
                        const int sh = mx6_scale_byte(mh), sl = mx6_scale_byte(ml);
                        const int sb = h ? sh : sl;
                        swap32_halves(lo, hi);
                        const u32x6 q = mx6_pack32(lo, hi, sb ? mx_scale_value(sb) : 1.f);
                        if (ox < Wc) {
                            char* ol = o + nt * 128;
                            *(f16x8*)(ol + 32 * h) = hv[0];
                            *(f16x8*)(ol + 32 * h + 16) = hv[1];
                            *(u32x4*)(ol + MX6_PLANE_LO(0) + 16 * h) = u32x4{q[0], q[1], q[2], q[3]};
                            *(u32x4*)(ol + MX6_PLANE_HI(0) + 16 * h) = u32x4{q[4], q[5], (unsigned)sb, 0u};
                        }
                    } else if (ox < Wc) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (a.out_planes == 2) {                             // fp16 pair (common.h split_f16)
                                f16x4 hi, lo;
                                split_f16x4(v + 4 * g, hi, lo);
                                *(f16x4*)(o + nt * 128 + (8 * g + 4 * h) * 2) = hi;
                                *(f16x4*)(o + nt * 128 + 64 + (8 * g + 4 * h) * 2) = lo;
                            } else {
                                bf16x4 hi;
#pragma unroll
                                for (int i = 0; i < 4; ++i) hi[i] = (__bf16)v[4 * g + i];
                                *(bf16x4*)(o + (nt * 32 + 8 * g + 4 * h) * 2) = hi;
                            }
                        }
                    }
                }
            }
        } else if (ox < Wc) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int oy = oy0 + 2 * wave + mt;
                float* o = a.out + (((size_t)n * Hc + oy) * Wc + ox) * 64;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[nt][mt][4 * g + i] + bias[nt][4 * g + i], 0.f);
                        *(f32x4*)(o + nt * 32 + 8 * g + 4 * h) = v;
                    }
            }
        }
    }
}

// maxpool 3x3 s2 p1 over f32 NHWC [N][Hc][Wc][64] -> PF activations (Hc/2 x Wc/2 x 64).
// One thread = one output pixel x 4 channels.
template <int PLANES>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* in, void* out, int N, int Hc, int Wc, PFGeom go) {
    const long long total = (long long)N * go.H * go.W * 16;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i & 15);
        long long p = i >> 4;
        const int px = (int)(p % go.W); p /= go.W;
        const int py = (int)(p % go.H);
        const int n = (int)(p / go.H);
        f32x4 m = {0.f, 0.f, 0.f, 0.f};                  // inputs are post-ReLU (>= 0): 0 == -inf padding
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = 2 * py + dy;
            if (y < 0 || y >= Hc) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = 2 * px + dx;
                if (x < 0 || x >= Wc) continue;
                const f32x4 v = *(const f32x4*)(in + (((size_t)n * Hc + y) * Wc + x) * 64 + c4 * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        }
        const int q = go.G + n * go.S + py * go.P + px;
        const int c = c4 * 4;
        char* o = (char*)out + (size_t)q * (64 * PLANES * 2);
        if constexpr (PLANES == 2) {                                              // fp16 pair (common.h split_f16)
            f16x4 hi, lo;
            const float mv[4] = {m[0], m[1], m[2], m[3]};
            split_f16x4(mv, hi, lo);
            *(f16x4*)(o + (c >> 5) * 128 + (c & 31) * 2) = hi;
            *(f16x4*)(o + (c >> 5) * 128 + 64 + (c & 31) * 2) = lo;
        } else {
            bf16x4 hi;
#pragma unroll
            for (int k = 0; k < 4; ++k) hi[k] = (__bf16)m[k];
            *(bf16x4*)(o + c * 2) = hi;
        }
    }
}

// --------------------------------------------------------------------------------------------
// Fused stem: tile read + LUT normalise + conv7x7 s2 + BN + ReLU + maxpool 3x3 s2 -> PF planes.
// The 4 MB/patch fp32 conv output never exists.  Work split:
//   workgroup = 2 waves = one strip of 15 pooled columns (31 conv columns: ONE MFMA pixel tile,
//   lane = conv column) x one segment of pooled rows; wave w owns output channels 32w..32w+31 and
//   keeps that slice of the packed weights in registers (14 k-steps x planes).
//   Per pooled row the wave computes the two new conv rows (2py, 2py+1); the vertical max with the
//   carried row 2py-1 is register-local, the horizontal 3-max is two lane shuffles, results sit on
//   odd lanes.  The normalised input lives in a 16-row LDS ring: 4 new rows are staged per step,
//   one barrier per step.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
struct StemPoolArgs {
    StemArgs s;              // s.out unused
    void* out_pf;            // PF (H/4, W/4, 64)
    int rows_per_seg;        // pooled rows per workgroup
    int out96;               // OUT == 3: 96-byte output lines (common.h CONV_OUT96), line-planar:
    long long plane96;       // bytes from the 32-channel line plane of wave 0's channels to wave 1's (ConvArgs.plane96)
    void* x0_pf;             // X0 kernels (U-Net): the conv output BEFORE the pool, relu(bn1(conv1(x))), as a PF tensor (H/2, W/2, 64), planes 2
};

constexpr int SP_COLS = 70;                  // input columns per strip (2*31 + 8: the widest lane's 16-byte read)
constexpr int SP_RING = 16;                  // ring rows
constexpr int SP_PLANE = SP_RING * SP_COLS * 8;

// PLANES = arithmetic of the stem itself (1: bf16, 2: fp16 hi/lo pair, 3-pass; r01-r04: bf16 pair); OUT = output line format (1, 2 or 3).
// U8X = DIG > 0 (u8 slide input, the product path): INTEGER arithmetic on v_mfma_i32_32x32x32_i8.  The pixel operand is the
// exact byte quadruple (R - 128, G - 128, B - 128, inside ? 127 : 0) - 4 bytes per pixel in the LDS ring, one 16-byte
// fragment = the four pixels (kw 4h .. 4h+3) of one kernel row, so K = 7 rows x 32 = 7 MFMA steps instead of 14.  The
// weights carry the normalisation and BN: w' = w bn_scale / (255 std[c]) per colour channel and, on the fourth ("inside")
// channel, kappa / 127 with kappa = sum_c w'_c (128 - 255 mean[c]) - so zero padding (all four bytes 0) contributes exactly
// nothing and a black out-of-slide pixel (0, 0, 0 -> -128, inside) exactly what the reference computes.  Each output
// channel's weights are fixed-point numbers in DIG = 3 balanced base-256 digits (24 bits, the precision of the fp32 weights
// themselves), one i8 MFMA pass per digit into its own exact i32 accumulator, recombined in the epilogue:
// (a2 2^16 + a1 2^8 + a0) scale[cout] + bn_shift[cout].  42 MFMAs of 32 cycles per pooled row and wave instead of the 56 of
// the fp16 hi/lo form it replaces (r02: 2.52 -> 2.08 ms per 2000 patches; DIG = 2 measured the same 2.08 ms - the kernel is
// then bound by its VALU epilogue - and 1.1-1.6x the logit error on the margin families, so it is not used).
typedef __attribute__((ext_vector_type(16))) int i32x16;
constexpr int STEM_I8_SCALE_OFFSET = 2 * 7 * 3 * 1024;           // float scale[64] behind the digit planes (capi.hip: wsi_prepack_stem_u8)
// NSTRIP (integer path only): strips per workgroup.  1 = the form above, weights in registers (236 VGPRs, two waves per SIMD).
// 2 = a 256-thread workgroup of two strips whose four waves share ONE copy of the digit planes in LDS (42 KB) and read each
// weight fragment right before its MFMA: ~150 VGPRs, three waves per SIMD.  r02 counters of the 1-strip form: VALU active 68 %
// + matrix pipe 32 % of the SIMD time with the waves waiting 25 % of their lifetime - more waves, not fewer instructions,
// is what the kernel lacks.
constexpr int SP_RING_I8 = SP_RING * SP_COLS * 4;                       // ring bytes of the integer path (4 B per pixel)
// X0 (r05, the U-Net's half-resolution skip): the kernel also stores every conv value it computes - the smp encoder's first feature map
// x0 = relu(bn1(conv1(x))), which the pool consumes and r02-r04 recomputed with the unfused fp16-pair stem kernel (325 us per 128 tiles
// of 256 x 256 beside this kernel's 227).  A strip's lanes 1..30 (odd layout) own conv columns 2 px0 .. 2 px0 + 29 - lane 0 is the
// left neighbour's last column - and the carry-only first step of a segment owns nothing, so every conv pixel is stored exactly once.
template <int PLANES, int OUT, int DIG = 0, int NSTRIP = 1, bool X0 = false>
__global__ __launch_bounds__(128 * NSTRIP, NSTRIP == 1 ? 2 : 3) void stem_pool_kernel(StemPoolArgs A) {
    constexpr bool U8X = DIG > 0, WLDS = NSTRIP > 1;
    static_assert(!WLDS || U8X, "shared weights: integer path only");
    static_assert(!X0 || (U8X && OUT == 2), "x0 output: integer path, fp16-pair lines");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const StemArgs& a = A.s;
    const int tid = threadIdx.x & 127, lane = tid & 63;                 // tid: thread within the strip's wave pair
    const int wave_g = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wave = wave_g & 1;                                        // = output-channel tile
    const int sl = wave_g >> 1;                                         // strip within the workgroup
    char* const smem = smem_all + (WLDS ? sl * SP_RING_I8 : 0);         // this strip's ring
    const int l31 = lane & 31, h = lane >> 5;
    const int Hc = a.H / 2, Wc = a.W / 2, Hp = a.H / 4, Wp = a.W / 4;
    // Strips: normally 15 pooled columns on the ODD lanes 1..29 (lane = conv column 2*px0 - 1 + lane: the column left of
    // the first pooled one is lane 0).  Pooled maps up to 16 wide (64x64 patches) fit ONE strip in the EVEN layout: lane =
    // conv column, pooled column p on lane 2p, whose left neighbour for p = 0 is the padding (lane 0 keeps its own value).
    const bool even = Wp <= 16;
    const int nstrips = even ? 1 : (Wp + 14) / 15, nsegs = (Hp + A.rows_per_seg - 1) / A.rows_per_seg;
    const long long nstrips_total = (long long)a.N * nsegs * nstrips;
    const bool strip_ok = (long long)blockIdx.x * NSTRIP + sl < nstrips_total;     // (a last workgroup may hold an idle strip)
    int b = strip_ok ? (int)(blockIdx.x * NSTRIP + sl) : (int)(nstrips_total - 1);
    const int strip = b % nstrips; b /= nstrips;
    const int seg = b % nsegs;
    const int n = b / nsegs;
    const int px0 = strip * 15, c0 = even ? 0 : 2 * px0 - 1;            // first pooled col / first conv col (lane 0)
    const int py0 = seg * A.rows_per_seg;
    const int py1 = min(py0 + A.rows_per_seg, Hp);
    const int ix0 = 2 * c0 - 3;                                         // input column of LDS column 0
    int tx = 0, ty = 0;
    if (a.mode == 1) { tx = a.origins[2 * n]; ty = a.origins[2 * n + 1]; }

    // this wave's weights: [nt][ks][plane][lane][8] -> registers
    constexpr int NKS = U8X ? 1 : 14, NDG = U8X ? DIG : 1;
    bf16x8 wreg[NKS][PLANES];
    i32x4 wq[U8X && !WLDS ? 7 : 1][NDG];
    float bias[U8X ? 1 : 16];
    // integer path: per-channel scale and shift live in LDS behind the (4-byte) ring and are read four at a time in the
    // epilogue - 32 fewer live registers across the MFMA loop
    float* const sb_lds = (float*)(smem_all + NSTRIP * SP_RING_I8);     // [scale 64][shift 64]
    char* const wl = smem_all + NSTRIP * SP_RING_I8 + 512;        // WLDS: the digit planes [nt 2][kh 7][digit][lane][16 B]
    if constexpr (U8X) {
        if constexpr (WLDS) {
            for (int i = threadIdx.x; i < 2 * 7 * DIG * 64; i += 128 * NSTRIP) ((uint4*)wl)[i] = ((const uint4*)a.wpk_u8)[i];
        } else {
#pragma unroll
            for (int ks = 0; ks < 7; ++ks)
#pragma unroll
                for (int d = 0; d < DIG; ++d) wq[ks][d] = *((const i32x4*)a.wpk_u8 + ((size_t)(wave * 7 + ks) * DIG + d) * 64 + lane);
        }
        if (threadIdx.x < 64) {
            sb_lds[threadIdx.x] = ((const float*)((const char*)a.wpk_u8 + STEM_I8_SCALE_OFFSET))[threadIdx.x];
            sb_lds[64 + threadIdx.x] = a.bias_u8[threadIdx.x];
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 14; ++ks)
#pragma unroll
            for (int p = 0; p < PLANES; ++p)
                wreg[ks][p] = *((const bf16x8*)a.wpk + ((size_t)(wave * 14 + ks) * PLANES + p) * 64 + lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = a.bias[wave * 32 + 8 * (r >> 2) + 4 * h + (r & 3)];
    }
    const size_t slide_bytes = (size_t)a.SH * (size_t)a.slide_pitch;

    auto stage_rows = [&](int row_lo, int nrows) {                      // input rows [row_lo, row_lo+nrows) -> ring
        for (int i = tid; i < nrows * SP_COLS; i += 128) {
            const int r = i / SP_COLS, cc = i - r * SP_COLS;
            const int iy = row_lo + r, ix = ix0 + cc;
            const int slot = (iy + 64) & (SP_RING - 1);
            const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            if constexpr (U8X) {
                unsigned q = 0u;                                        // zero padding: contributes exactly nothing
                if (inside) {
                    const int sx = tx + ix, sy = ty + iy;
                    unsigned rgb = 0u;                                  // outside the slide OpenSlide pads with black
                    if (sx >= 0 && sx < a.SW && sy >= 0 && sy < a.SH) {
                        const size_t off = (size_t)sy * a.slide_pitch + (size_t)sx * 3;
                        const uint8_t* pp = a.slide + off;
                        if (off + 4 <= slide_bytes) __builtin_memcpy(&rgb, pp, 4);   // one (unaligned) dword: R, G, B, next R
                        else rgb = pp[0] | (pp[1] << 8) | (pp[2] << 16);             // last pixel of the buffer
                    }
                    q = ((rgb & 0xffffffu) ^ 0x808080u) | 0x7f000000u;  // (R, G, B) - 128 as i8, inside = 127
                }
                *(unsigned*)(smem + (size_t)(slot * SP_COLS + cc) * 4) = q;
            } else {
                float v[3] = {0.f, 0.f, 0.f};
                if (inside) {
                    if (a.mode == 0) {
                        const size_t base = ((size_t)n * 3 * a.H + iy) * a.W + ix;
#pragma unroll
                        for (int c = 0; c < 3; ++c) v[c] = a.in_f32[base + (size_t)c * a.H * a.W];
                    } else {
                        const int sx = tx + ix, sy = ty + iy;
                        uint8_t px[3] = {0, 0, 0};
                        if (sx >= 0 && sx < a.SW && sy >= 0 && sy < a.SH) {
                            const uint8_t* pp = a.slide + (size_t)sy * a.slide_pitch + (size_t)sx * 3;
                            px[0] = pp[0]; px[1] = pp[1]; px[2] = pp[2];
                        }
#pragma unroll
                        for (int c = 0; c < 3; ++c) v[c] = a.lut[c * 256 + px[c]];
                    }
                }
                typedef typename PairElem<PLANES>::T E;   // PLANES 2: fp16 pair (common.h), 1: bf16
                typedef __attribute__((ext_vector_type(4))) E Ex4;
                Ex4 hi, lo;
#pragma unroll
                for (int c = 0; c < 3; ++c) { hi[c] = (E)v[c]; lo[c] = (E)(v[c] - (float)hi[c]); }
                hi[3] = (E)0.f; lo[3] = (E)0.f;
                *(Ex4*)(smem + (size_t)(slot * SP_COLS + cc) * 8) = hi;
                if constexpr (PLANES == 2) *(Ex4*)(smem + SP_PLANE + (size_t)(slot * SP_COLS + cc) * 8) = lo;
            }
        }
    };

    // U8X: the 4 rows of the NEXT step are fetched into registers before this step's MFMAs (their latency hides
    // behind them) and converted / written to the ring after the MFMAs, just before the step's barrier.
    constexpr int NPF = (4 * SP_COLS + 127) / 128;                      // pixels per thread per step
    unsigned pf_rgb[NPF];
    auto fetch_rows = [&](int row_lo) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * 128;
            const int r = i / SP_COLS, cc = i - r * SP_COLS;
            const int iy = row_lo + r, ix = ix0 + cc;
            unsigned rgb = 0u;
            if (i < 4 * SP_COLS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                const int sx = tx + ix, sy = ty + iy;
                if (sx >= 0 && sx < a.SW && sy >= 0 && sy < a.SH) {
                    const size_t off = (size_t)sy * a.slide_pitch + (size_t)sx * 3;
                    const uint8_t* pp = a.slide + off;
                    if (off + 4 <= slide_bytes) __builtin_memcpy(&rgb, pp, 4);
                    else rgb = pp[0] | (pp[1] << 8) | (pp[2] << 16);
                }
            }
            pf_rgb[k] = rgb;
        }
    };
    auto commit_rows = [&](int row_lo) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * 128;
            if (i >= 4 * SP_COLS) continue;
            const int r = i / SP_COLS, cc = i - r * SP_COLS;
            const int iy = row_lo + r, ix = ix0 + cc;
            const int slot = (iy + 64) & (SP_RING - 1);
            unsigned q = 0u;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) q = ((pf_rgb[k] & 0xffffffu) ^ 0x808080u) | 0x7f000000u;
            *(unsigned*)(smem + (size_t)(slot * SP_COLS + cc) * 4) = q;
        }
    };

    // the first step is py0-1: it only produces the carried conv row 2*py0-1 (all zero for py0 == 0)
    stage_rows(4 * (py0 - 1) - 3, 9);
    __syncthreads();
    float carry[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) carry[r] = 0.f;
    const bool col_ok = (c0 + l31) >= 0 && (c0 + l31) < Wc && (even || l31 < 31);
    const float col_ub = col_ok ? (OUT == 3 ? 65504.f : 3.4028234e38f) : 0.f;
    const int lc = (even || l31 < 31) ? l31 : 30;                       // odd layout: lane 31 is idle, keep its reads inside the row
    const bool o96 = OUT == 3 && A.out96;
    const size_t pixstride = o96 ? (size_t)96 : (size_t)64 * PFmt<OUT>::BPC;
    PFGeom go = pf_geom(a.N, Hp, Wp, 64);

    // every wave of the workgroup makes the same number of trips (one barrier each); strips with fewer rows idle at the end
    for (int it = 0; it <= A.rows_per_seg; ++it) {
        const int py = py0 - 1 + it;
        if (!strip_ok || py >= py1) { __syncthreads(); continue; }
        if constexpr (U8X) {
            if (py + 1 < py1) fetch_rows(4 * (py + 1) + 2);            // rows the NEXT step adds: loads in flight
        } else {
            if (py + 1 < py1) stage_rows(4 * (py + 1) + 2, 4);         // rows the NEXT step adds (not read by this step)
        }
        f32x16 acc[2];
        i32x16 aq[U8X ? 2 : 1][NDG];
        if constexpr (U8X) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int d = 0; d < DIG; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) aq[mt][d][r] = 0;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {                            // one kernel row per step: 8 pixels x 4 bytes = K 32
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int slot = (4 * py + 2 * mt - 3 + ks + 64) & (SP_RING - 1);
                    const char* xp = smem + (size_t)(slot * SP_COLS + 2 * lc + 4 * h) * 4;      // 8-byte aligned
                    const uint2 x01 = *(const uint2*)xp, x23 = *(const uint2*)(xp + 8);
                    const i32x4 x = {(int)x01.x, (int)x01.y, (int)x23.x, (int)x23.y};
#pragma unroll
                    for (int d = 0; d < DIG; ++d) {
                        i32x4 wv;
                        if constexpr (WLDS) wv = *(const i32x4*)(wl + ((wave * 7 + ks) * DIG + d) * 1024 + lane * 16);
                        else wv = wq[ks][d];
                        aq[mt][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wv, x, aq[mt][d], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            const int kh = ks >> 1, kw0 = (ks & 1) * 4 + 2 * h;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int slot = (4 * py + 2 * mt - 3 + kh + 64) & (SP_RING - 1);
                const char* xp = smem + (size_t)(slot * SP_COLS + 2 * lc + kw0) * 8;
                const bf16x8 x0 = *(const bf16x8*)xp;
                if constexpr (PLANES == 2) {
                    const bf16x8 x1 = *(const bf16x8*)(xp + SP_PLANE);
                    acc[mt] = mfma16<PLANES>(wreg[ks][1], x0, acc[mt]);
                    acc[mt] = mfma16<PLANES>(wreg[ks][0], x1, acc[mt]);
                }
                acc[mt] = mfma16<PLANES>(wreg[ks][0], x0, acc[mt]);
            }
        }
        }
        // bias + ReLU + validity mask, vertical max with the carried row, keep row 2py+1 as the next carry
        float v[16];
        float xv[2][4];                                                 // X0: four channels of the two conv rows
        unsigned xq[2][2][4];                                           //     [row][group parity][hi0 hi1 lo0 lo1] packed pairs
        (void)xv; (void)xq;
        if constexpr (U8X) {
            // Integer path: the carried row and the vertical max live in the RAW domain (the exact digit recombination, before the
            // channel's scale and shift).  scale > 0 (wsi_prepack_stem_u8), so x -> max(fma(x, scale, shift), 0) is monotone,
            // rounding included, and commutes with the maximum: one FMA + ReLU per pooled value instead of one per conv value,
            // bit-identical.  Conv rows 2py, 2py+1 are inside the map for every py >= 0; above the image (py = -1, the carry-only
            // step of the first segment) the carry is the pool's padding, -FLT_MAX here.
            auto raw = [&](int mt, int r) {                             // exact recombination of the digit planes
                float c = (float)(DIG >= 2 ? aq[mt][DIG >= 2 ? 1 : 0][r] * 256 + aq[mt][0][r] : aq[mt][0][r]);
                if constexpr (DIG == 3) c = __builtin_fmaf((float)aq[mt][DIG - 1][r], 65536.0f, c);
                return c;
            };
            const bool x0_store = X0 && py >= py0 && col_ok && (even || (l31 >= 1 && l31 <= 30));
            const PFGeom gx = pf_geom(a.N, Hc, Wc, 64);
            char* const x0p = X0 ? (char*)A.x0_pf + ((size_t)(gx.G + n * gx.S + 2 * py * gx.P + (x0_store ? c0 + l31 : 0)) * 256 + wave * 128 + 16 * h) : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float c0 = raw(0, r), c1 = raw(1, r);
                if constexpr (X0) {                                     // conv rows 2py, 2py + 1 themselves: scale, shift, ReLU, fp16 pair
                    xv[0][r & 3] = __builtin_amdgcn_fmed3f(__builtin_fmaf(c0, sb_lds[wave * 32 + 8 * (r >> 2) + 4 * h + (r & 3)],
                                                                          sb_lds[64 + wave * 32 + 8 * (r >> 2) + 4 * h + (r & 3)]), 0.f, 65504.f);
                    xv[1][r & 3] = __builtin_amdgcn_fmed3f(__builtin_fmaf(c1, sb_lds[wave * 32 + 8 * (r >> 2) + 4 * h + (r & 3)],
                                                                          sb_lds[64 + wave * 32 + 8 * (r >> 2) + 4 * h + (r & 3)]), 0.f, 65504.f);
                    if ((r & 3) == 3) {
                        const int g = r >> 2;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            f16x4 hi, lo;
                            split_f16x4(xv[mt], hi, lo);
                            const u32x2 hq = __builtin_bit_cast(u32x2, hi), lq = __builtin_bit_cast(u32x2, lo);
                            xq[mt][g & 1][0] = hq[0]; xq[mt][g & 1][1] = hq[1]; xq[mt][g & 1][2] = lq[0]; xq[mt][g & 1][3] = lq[1];
                        }
                        if (g & 1) {
                            // channels 8g' + 4h + i of the group pair (g - 1, g): lanes h = 0 give their second group for the partner's first, so
                            // h = 0 holds eight consecutive channels 16 (g >> 1) + 0..7 and h = 1 the next eight: 16-byte stores
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) {
                                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
                                             "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
                                             : "+v"(xq[mt][0][0]), "+v"(xq[mt][0][1]), "+v"(xq[mt][0][2]), "+v"(xq[mt][0][3]),
                                               "+v"(xq[mt][1][0]), "+v"(xq[mt][1][1]), "+v"(xq[mt][1][2]), "+v"(xq[mt][1][3]));
                                if (x0_store) {
                                    char* o = x0p + (size_t)mt * gx.P * 256 + (g >> 1) * 32;
                                    *(u32x4*)o = u32x4{xq[mt][0][0], xq[mt][0][1], xq[mt][1][0], xq[mt][1][1]};
                                    *(u32x4*)(o + 64) = u32x4{xq[mt][0][2], xq[mt][0][3], xq[mt][1][2], xq[mt][1][3]};
                                }
                            }
                        }
                    }
                }
                const float m = fmaxf(fmaxf(carry[r], c0), c1);
                carry[r] = c1;
                const int ch = wave * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                // ReLU, the column mask (columns outside the map are the pool's padding: 0) and - mode 3 - the fp16-range clamp of
                // the line encode in ONE v_med3_f32: clamp(x, 0, ub) with ub = 0 on masked columns (r03: v_max + v_cndmask here and
                // a v_min per value before the encode; min commutes with the horizontal max below, so the bits are the same)
                v[r] = __builtin_amdgcn_fmed3f(__builtin_fmaf(m, sb_lds[ch], sb_lds[64 + ch]), 0.f, col_ub);   // explicit FMA (-ffp-contract=off)
            }
            if (py < 0) {                                               // (uniform; first step of the first segment only)
#pragma unroll
                for (int r = 0; r < 16; ++r) carry[r] = -3.4028234e38f;
            }
        } else {
            const bool r0_ok = (2 * py) >= 0 && (2 * py) < Hc, r1_ok = (2 * py + 1) >= 0 && (2 * py + 1) < Hc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a0 = (col_ok && r0_ok) ? fmaxf(acc[0][r] + bias[r], 0.f) : 0.f;
                const float a1 = (col_ok && r1_ok) ? fmaxf(acc[1][r] + bias[r], 0.f) : 0.f;
                v[r] = fmaxf(fmaxf(carry[r], a0), a1);
                carry[r] = a1;
            }
        }
        if (py >= py0) {
            // horizontal 3-max: pooled column px0+j sits on odd lane 2j+1 of each 32-lane half
#pragma unroll
            for (int r = 0; r < 16; ++r) {                               // neighbours by DPP wave shifts (VALU, no LDS crossbar)
                // (bound_ctrl: a lane without a source reads 0 - the neutral element, every value here is >= 0 after the ReLU - so
                // no copy of the old value is needed and the shifts fold into the max instructions as DPP operands)
                const int vi = __builtin_bit_cast(int, v[r]);
                float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x138, 0xf, 0xf, true));          // lane i <- i-1
                if (even && l31 == 0) up = 0.f;                          // even layout: column -1 is padding (lane 32 must not see lane 31)
                const float dn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x130, 0xf, 0xf, true));    // lane i <- i+1
                v[r] = fmaxf(v[r], fmaxf(up, dn));
            }
            const int j = even ? (l31 >> 1) : ((l31 - 1) >> 1), px = px0 + j;
            const bool store = (even ? !(l31 & 1) : ((l31 & 1) && l31 <= 29)) && px < Wp;
            char* o = (char*)A.out_pf + (size_t)(go.G + n * go.S + py * go.P + (store ? px : 0)) * pixstride;
            if constexpr (OUT == 3) {
                // fp16 hi + MX-fp6 (lo6, hi6) with one scale per plane and 32-channel line; the line's channels sit in lanes l, l^32
                // (same encode as conv_epilogue_mx: conv_dev.h)
                f32x16 hi, lo;
                f16x8 hv[2];
                float mh = 0.f, ml = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float v0 = U8X ? v[r] : fminf(v[r], 65504.f), v1 = U8X ? v[r + 1] : fminf(v[r + 1], 65504.f);   // (integer path: clamped above)
                    const f16x2 hh = __builtin_convertvector(f32x2{v0, v1}, f16x2);
                    hv[r >> 3][r & 7] = hh[0];
                    hv[r >> 3][(r & 7) + 1] = hh[1];
                    hi[r] = (float)hh[0];
                    hi[r + 1] = (float)hh[1];
                    lo[r] = v0 - hi[r];
                    lo[r + 1] = v1 - hi[r + 1];
                    mh = fmaxf(mh, fmaxf(hi[r], hi[r + 1]));                                // (v >= 0 after the ReLU)
                    ml = fmaxf(ml, fmaxf(fabsf(lo[r]), fabsf(lo[r + 1])));
                }
                pair_max2(mh, ml);
                const int sh = mx6_scale_byte(mh), sl = mx6_scale_byte(ml);
                const int sb = h ? sh : sl;
                swap32_halves(lo, hi);
                const u32x6 q = mx6_pack32(lo, hi, sb ? mx_scale_value(sb) : 1.f);
                if (store) {                                                        // line order: common.h mx_line_pos / mx6_field_of_pos
                    char* ol = o + (o96 ? (size_t)wave * (size_t)A.plane96 : (size_t)wave * 128);
                    *(f16x8*)(ol + 32 * h) = hv[0];
                    *(f16x8*)(ol + 32 * h + 16) = hv[1];
                    if (!o96) {
                        *(u32x4*)(ol + MX6_PLANE_LO(0) + 16 * h) = u32x4{q[0], q[1], q[2], q[3]};
                        *(u32x4*)(ol + MX6_PLANE_HI(0) + 16 * h) = u32x4{q[4], q[5], (unsigned)sb, 0u};
                    } else if (h == 0) {                                            // lo6 plane + both scales (conv_dev.h conv_epilogue_mx)
                        *(u32x4*)(ol + 64) = u32x4{q[0], q[1], q[2], q[3]};
                        *(u32x4*)(ol + 80) = u32x4{q[4], q[5], (unsigned)sl, (unsigned)sh};
                    }
                }
            } else if (store) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = wave * 32 + 8 * g + 4 * h;
                    if constexpr (OUT == 2) {                                        // fp16 pair (common.h split_f16)
                        f16x4 hi, lo;
                        split_f16x4(v + 4 * g, hi, lo);
                        *(f16x4*)(o + wave * 128 + (8 * g + 4 * h) * 2) = hi;
                        *(f16x4*)(o + wave * 128 + 64 + (8 * g + 4 * h) * 2) = lo;
                    } else {
                        bf16x4 hi;
#pragma unroll
                        for (int i = 0; i < 4; ++i) hi[i] = (__bf16)v[4 * g + i];
                        *(bf16x4*)(o + c * 2) = hi;
                    }
                }
            }
        }
        if constexpr (U8X) {
            if (py + 1 < py1) commit_rows(4 * (py + 1) + 2);           // not read by this step: safe before the barrier
        }
        __syncthreads();                                                // next rows staged, this step's reads done
    }
}

int g_stem_shared_weights = 1;                        // A/B: wsi_stem_set_mode(fused = 3) selects the one-strip form (weights in registers)
int wsi_stem_pool_dispatch(const StemArgs& a, void* out_pf, int planes, int rows_per_seg, hipStream_t st, int out96, long long plane96, void* x0_pf) {
    if (a.H % 4 || a.W % 4 || a.N <= 0 || planes < 1 || planes > 3 || rows_per_seg <= 0) return WSI_EINVAL;
    StemPoolArgs A;
    A.s = a; A.out_pf = out_pf; A.rows_per_seg = rows_per_seg; A.out96 = out96; A.plane96 = plane96; A.x0_pf = x0_pf;
    if (out96 && plane96 <= 0) return WSI_EINVAL;
    const int Hp = a.H / 4, Wp = a.W / 4;
    if (rows_per_seg > Hp) rows_per_seg = Hp;        // (a 64 x 64 crop has 16 pooled rows: the kernel makes rows_per_seg + 1 barrier trips whatever the map holds)
    A.rows_per_seg = rows_per_seg;
    const long long grid = (long long)a.N * (Wp <= 16 ? 1 : (Wp + 14) / 15) * ((Hp + rows_per_seg - 1) / rows_per_seg);   // strips, see kernel
    if (grid > 0x7fffffffLL) return WSI_EINVAL;
    const bool u8x = a.mode == 1 && a.wpk_u8 && a.bias_u8 && planes >= 2;
    const size_t lds = (size_t)(planes == 1 || u8x ? 1 : 2) * SP_PLANE;
    if (u8x && g_stem_shared_weights) {               // integer stem, two strips per workgroup, digit planes shared in LDS
        const size_t lds2 = 2 * SP_RING_I8 + 512 + 2 * 7 * 3 * 1024;
        const int grid2 = (int)((grid + 1) / 2);
        if (x0_pf && planes != 2) return WSI_EINVAL;
        if (planes == 3) hipLaunchKernelGGL((stem_pool_kernel<2, 3, 3, 2>), dim3(grid2), dim3(256), lds2, st, A);
        else if (x0_pf) hipLaunchKernelGGL((stem_pool_kernel<2, 2, 3, 2, true>), dim3(grid2), dim3(256), lds2, st, A);
        else hipLaunchKernelGGL((stem_pool_kernel<2, 2, 3, 2>), dim3(grid2), dim3(256), lds2, st, A);
    } else if (x0_pf)
        return WSI_EINVAL;                            // the x0 output exists in the shared-weights integer form only
    else if (u8x && planes == 3)                      // integer stem, 24-bit weights in both modes (r02: the 16-bit form, DIG 2, costs mx margin)
        hipLaunchKernelGGL((stem_pool_kernel<2, 3, 3>), dim3((int)grid), dim3(128), lds, st, A);
    else if (u8x)
        hipLaunchKernelGGL((stem_pool_kernel<2, 2, 3>), dim3((int)grid), dim3(128), lds, st, A);
    else if (planes == 3)                             // f32 input: fp16 hi/lo arithmetic, mode-3 output lines
        hipLaunchKernelGGL((stem_pool_kernel<2, 3>), dim3((int)grid), dim3(128), lds, st, A);
    else if (planes == 2)
        hipLaunchKernelGGL((stem_pool_kernel<2, 2>), dim3((int)grid), dim3(128), lds, st, A);
    else
        hipLaunchKernelGGL((stem_pool_kernel<1, 1>), dim3((int)grid), dim3(128), lds, st, A);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_stem_dispatch(const StemArgs& a, int planes, hipStream_t st) {
    if (a.H % 16 || a.W % 2 || a.N <= 0 || (planes != 1 && planes != 2)) return WSI_EINVAL;
    const int Hc = a.H / 2, Wc = a.W / 2;
    const int total = a.N * ((Wc + STEM_TC - 1) / STEM_TC) * (Hc / STEM_TR);
    const int grid = total < 2048 ? total : 2048;
    const size_t lds = (size_t)28 * planes * 1024 + (size_t)planes * STEM_PLANE_BYTES;
    if (planes == 2) {
        auto k = stem_conv7x7_kernel<2>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return WSI_EINVAL;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, a);
    } else {
        hipLaunchKernelGGL(stem_conv7x7_kernel<1>, dim3(grid), dim3(256), lds, st, a);
    }
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_maxpool_dispatch(const float* in, void* out, int N, int Hc, int Wc, int planes, hipStream_t st) {
    if (Hc % 2 || Wc % 2) return WSI_EINVAL;
    PFGeom go = pf_geom(N, Hc / 2, Wc / 2, 64);
    const long long total = (long long)N * go.H * go.W * 16;
    long long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (planes == 2)
        hipLaunchKernelGGL(maxpool3x3s2_kernel<2>, dim3((int)grid), dim3(256), 0, st, in, out, N, Hc, Wc, go);
    else
        hipLaunchKernelGGL(maxpool3x3s2_kernel<1>, dim3((int)grid), dim3(256), 0, st, in, out, N, Hc, Wc, go);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}
