// extern "C" surface of libwsi_hip.so (include/wsi_hip.h): argument checking, host-side weight
// prepack, and the trunk launch sequence.  No device allocation, no synchronisation, no exceptions.
#include "common.h"
#include <mutex>
#include <unordered_map>
#include <vector>
#include <cmath>
#include "../../include/wsi_hip.h"
#include <math.h>
#include <string.h>

int wsi_conv_dispatch(const ConvArgs& a, int planes, int cfg, hipStream_t st);
int wsi_s2_dispatch(const ConvArgs& a, int planes, hipStream_t st);
int wsi_stem_dispatch(const StemArgs& a, int planes, hipStream_t st);
int wsi_maxpool_dispatch(const float* in, void* out, int N, int Hc, int Wc, int planes, hipStream_t st);
int wsi_stem_pool_dispatch(const StemArgs& a, void* out_pf, int planes, int rows_per_seg, hipStream_t st, int out96 = 0, long long plane96 = 0, void* x0_pf = nullptr);
int wsi_avgpool_fc_dispatch(const void* in, const PFGeom& g, const float* w, const float* b, int K, float* feat,
                            float* logits, int planes, hipStream_t st);
int wsi_linear_dispatch(const float* x, const float* w, const float* bias, float* y, int B, int K, int J, int relu,
                        hipStream_t st);
int wsi_pf_pack_dispatch(const float* in, void* out, const PFGeom& g, int planes, hipStream_t st);
int wsi_pf_unpack_dispatch(const void* in, float* out, const PFGeom& g, int planes, hipStream_t st);
int wsi_tile_gather_dispatch(const uint8_t* slide, long long pitch, int SH, int SW, const int* origins, const float* lut,
                             float* out, int N, int ph, int pw, hipStream_t st);
int wsi_stitch_add_dispatch(const float* logits, const int* txy, int T, int C, int dy, int dx, double* pred, int MH, int MW,
                            hipStream_t st);
int wsi_stitch_add_dense_dispatch(const float* tiles, const int* txy, int T, int C, int ph, int pw, double* pred, int MH,
                                  int MW, hipStream_t st);
int wsi_paint_dispatch(const long long* idx, const int* region_of, long long n, const uint8_t* cls, int* winner, long long* label,
                       long long npix, hipStream_t st);
int wsi_hsv_mask_dispatch(const uint8_t* rgb, long long npix, int stride, double thresh, uint8_t* mask, hipStream_t st);
size_t wsi_cc_scratch_bytes(int H, int W);
int wsi_cc_dispatch(const uint8_t* mask, int H, int W, int* labels_out, int* count_out, void* scratch, hipStream_t st, int conn4 = 0);
int wsi_lab_mask_dispatch(const uint8_t* rgb, long long npix, int stride, double mu_percent, uint8_t* mask, void* scratch, hipStream_t st);
size_t wsi_fill_holes_scratch_bytes_impl(int H, int W);
int wsi_fill_holes_dispatch(const uint8_t* mask, int H, int W, uint8_t* out, void* scratch, hipStream_t st);
int wsi_kmeans_dispatch(const int* pts, int n, double* centres, int k, int iters, int* labels, void* scratch, hipStream_t st);
int wsi_kmeans_seed_farthest_dispatch(const int* pts, int n, int k, double* centres, void* scratch, hipStream_t st);
int wsi_exponent_span_dispatch(const float* v, long long n, int* out2, hipStream_t st);
int wsi_softmax_dispatch(const double* pred, int C, long long HW, const double* thresh, double* probs, uint8_t* classes,
                         const uint8_t* mask, int heat_mode, uint8_t* heat, hipStream_t st);

int wsi_upsample_concat_dispatch(const void* x, const void* skip, void* out, int n, int h, int w, int cx, int cs, int planes, hipStream_t st);
int wsi_nhwc_to_pf_dispatch(const float* in, void* out, int n, int h, int w, int c, int planes, hipStream_t st);
int wsi_unet_tail_dispatch(const void* x4, const void* blob, int n, int h, int w, int classes, float* logits, hipStream_t st);
int wsi_unet_head_dispatch(const void* in, int n, int h, int w, int c_pf, const float* wt, const float* b, int cin, int k, float* out,
                           int planes, hipStream_t st);
int wsi_resize_nearest_dispatch(const float* src, long long planes_n, int hs, int ws, float* dst, int hd, int wd, hipStream_t st);
int wsi_resize_dispatch(const double* src, int C, int Hs, int Ws, double* dst, int Hd, int Wd, hipStream_t st);
int wsi_argmax_dispatch(const double* pred, int C, long long HW, uint8_t* classes, hipStream_t st);
int wsi_threshold_dispatch(const uint8_t* src, long long n, int lo, uint8_t* dst, hipStream_t st);
int wsi_morph_dispatch(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int H, int W, int k, int op, hipStream_t st);
int wsi_bwperim_dispatch(const uint8_t* src, uint8_t* dst, int H, int W, hipStream_t st);
size_t wsi_hull_ws_bytes(int H);
int wsi_hull_dispatch(const uint8_t* src, uint8_t* dst, int H, int W, void* ws, hipStream_t st);
int wsi_hull_polygon_dispatch(void* ws, int H, double* out_xy, int cap, hipStream_t st);
int wsi_iou_counts_dispatch(const uint8_t* a, const uint8_t* b, long long n, unsigned long long* out, hipStream_t st);
int wsi_score_counts_dispatch(const uint8_t* p, const uint8_t* gt, const uint8_t* mask, long long n, unsigned long long* out, hipStream_t st);
int wsi_esp_dispatch(const double* pts, int n, int num, double* out, double* scratch, hipStream_t st);

// ------------------------------------------------------------------------------------ host helpers
static inline uint16_t f2bf(float f) {            // round-to-nearest-even, same as the device cast
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline void split_host(float x, uint16_t& hi, uint16_t& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}
static void bn_fold(const float* g, const float* b, const float* m, const float* v, float eps, int co, double& scale,
                    double& shift) {
    if (!g) { scale = 1.0; shift = 0.0; return; }
    scale = (double)g[co] / sqrt((double)v[co] + (double)eps);
    shift = (double)b[co] - (double)m[co] * scale;
}
static inline float f16_round(float x) { return (float)(_Float16)x; }
static inline uint16_t f16_bits(float x) {
    const _Float16 hf = (_Float16)x;
    uint16_t u;
    memcpy(&u, &hf, 2);
    return u;
}

static inline void split_host_f16(float x, uint16_t& hi, uint16_t& lo) {   // the fp16 pair of mode 2 (common.h split_f16)
    x = fminf(fmaxf(x, -65504.f), 65504.f);
    const float h = f16_round(x);
    hi = f16_bits(h);
    lo = f16_bits(x - h);
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

long long wsi_tile_grid_candidates_impl(int iw, int ih, int ph, int pw, int sh, int sw);
size_t wsi_slic_scratch_bytes_impl(int H, int W, int K);
int wsi_slic_dispatch(const uint8_t* rgb, int H, int W, const double* fw, int radius, double* segs, int K, int step_y, int step_x,
                      double step, double compactness, int iters, int* labels, void* scratch, hipStream_t st);
size_t wsi_tile_grid_scratch_bytes_impl(long long n);
int wsi_tile_grid_dispatch(int iw, int ih, int ph, int pw, int sh, int sw, const uint8_t* mask, int MH, int MW, double m, double thresh,
                           int* out_xy, int* count_out, void* scratch, hipStream_t st);
// ingest.hip (C++ linkage)
struct wsi_ring;
struct wsi_resample_plan;
int wsi_ring_create_impl(wsi_ring** out, int slots, size_t slot_bytes);
void* wsi_ring_host_slot_impl(wsi_ring* r, int slot);
int wsi_ring_wait_slot_impl(wsi_ring* r, int slot);
int wsi_ring_submit_impl(wsi_ring* r, int slot, int rows, int width, int channels, long long src_pitch, uint8_t* dst, long long dst_pitch);
int wsi_ring_fence_impl(wsi_ring* r, hipStream_t compute);
int wsi_ring_acquire_impl(wsi_ring* r, hipStream_t compute);
int wsi_ring_device_impl(const wsi_ring* r);
int wsi_ring_drain_impl(wsi_ring* r);
void wsi_ring_destroy_impl(wsi_ring* r);
int wsi_resample_plan_create_impl(wsi_resample_plan** out, int in_h, int in_w, int out_h, int out_w);
void wsi_resample_plan_destroy_impl(wsi_resample_plan* p);
size_t wsi_resample_scratch_bytes_impl(const wsi_resample_plan* p, int n);
int wsi_resample_tiles_impl(const wsi_resample_plan* p, const uint8_t* slide, long long pitch, int SH, int SW, const int* origins, int N,
                            uint8_t* out, void* scratch, hipStream_t st);

extern "C" {

int wsi_hip_abi_version(void) { return WSI_HIP_ABI_VERSION; }

size_t wsi_pf_bytes(int n, int h, int w, int c, int planes) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || planes < 1 || planes > 3) return 0;
    return (size_t)pf_alloc_pixels(n, h, w) * (size_t)c * (planes == 1 ? 2 : 4);
}
long long wsi_pf_pixel_index(int n, int y, int x, int h, int w) {
    return (long long)(w + 2) + (long long)n * (h + 1) * (w + 1) + (long long)y * (w + 1) + x;
}

size_t wsi_prepack_conv_bytes(int cout, int cin, int k, int planes) {
    if (planes < 1 || planes > 3 || cout % 32 || cin % (planes == 1 ? 64 : 32) || (k != 1 && k != 3)) return 0;     // (whole 128-byte lines)
    // [cout/32][lines][k*k][4 frags][64 lanes][16 bytes]; lines = cin/64 (planes 1) or cin/32 (planes 2, 3)
    // planes 2: + cout floats, the inverse per-channel weight scales (common.h conv_wscale_inv)
    return (size_t)(cout / 32) * (planes == 1 ? cin / 64 : cin / 32) * k * k * 4 * 64 * 16 + (planes == 2 ? (size_t)cout * sizeof(float) : 0);
}

int wsi_prepack_conv(const float* w, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                     const float* bn_var, float eps, int cout, int cin, int k, int planes, void* wpk_out,
                     float* bias_out) {
    if (!w || !wpk_out || !bias_out || wsi_prepack_conv_bytes(cout, cin, k, planes) == 0) return WSI_EINVAL;
    uint16_t* o = (uint16_t*)wpk_out;
    const int NL = planes == 1 ? cin / 64 : cin / 32, NT = k * k;
    for (int co = 0; co < cout; ++co) {
        double sc, sh;
        bn_fold(bn_weight, bn_bias, bn_mean, bn_var, eps, co, sc, sh);
        bias_out[co] = (float)sh;
    }
    if (planes == 3) {
        // per (cout, line, tap): fp16 hi of the 32 channels; hi6 / lo6 = MX-fp6 (e2m3) of hi / (w - hi) with one E8M0
        // scale per block.  frag 0/1: fp16 k-steps (K position = activation line position, common.h mx_line_chan);
        // frag 2: dwords 0-3 of the lane's fp6 plane - lanes h=0 carry Wh6, h=1 carry Wl6 (the two K halves of the MX
        // instruction pair with Xl6 / Xh6), K position = activation field order (mx6_field_chan); frag 3: {dwords 4-5 of
        // the plane, the plane's block scale byte, 0}.
        memset(wpk_out, 0, wsi_prepack_conv_bytes(cout, cin, k, planes));
        for (int nt = 0; nt < cout / 32; ++nt)
            for (int l = 0; l < NL; ++l)
                for (int t = 0; t < NT; ++t) {
                    uint8_t* base = (uint8_t*)wpk_out + (((size_t)nt * NL + l) * NT + t) * 4096;
                    for (int r = 0; r < 32; ++r) {
                        const int co = nt * 32 + r;
                        double sc, sh;
                        bn_fold(bn_weight, bn_bias, bn_mean, bn_var, eps, co, sc, sh);
                        float hi[32], lo[32], mh = 0.f, ml = 0.f;                     // indexed by CHANNEL inside the line
                        for (int ci = 0; ci < 32; ++ci) {
                            const float wf = (float)((double)w[(((size_t)co * cin + 32 * l + ci) * k + t / k) * k + t % k] * sc);
                            hi[ci] = f16_round(wf);
                            lo[ci] = wf - hi[ci];
                            mh = fmaxf(mh, fabsf(hi[ci]));
                            ml = fmaxf(ml, fabsf(lo[ci]));
                        }
                        const int sh_b = mx6_scale_byte(mh), sl_b = mx6_scale_byte(ml);
                        const float ih = sh_b ? 1.0f / mx_scale_value(sh_b) : 0.f, il = sl_b ? 1.0f / mx_scale_value(sl_b) : 0.f;
                        for (int h = 0; h < 2; ++h) {
                            const int lane = r + 32 * h;
                            uint16_t* f0 = (uint16_t*)(base + 0 * 1024 + lane * 16);
                            uint16_t* f1 = (uint16_t*)(base + 1 * 1024 + lane * 16);
                            for (int j = 0; j < 8; ++j) {
                                f0[j] = f16_bits(hi[mx_line_chan(8 * h + j)]);
                                f1[j] = f16_bits(hi[mx_line_chan(16 + 8 * h + j)]);
                            }
                            unsigned pl[6] = {0u, 0u, 0u, 0u, 0u, 0u};
                            for (int f = 0; f < 32; ++f) {
                                const int ci = mx6_field_chan(f);
                                mx6_set_field(pl, f, h == 0 ? fp6_encode(hi[ci] * ih) : fp6_encode(lo[ci] * il));
                            }
                            uint32_t* f2 = (uint32_t*)(base + 2 * 1024 + lane * 16);
                            uint32_t* f3 = (uint32_t*)(base + 3 * 1024 + lane * 16);
                            for (int d = 0; d < 4; ++d) f2[d] = pl[d];
                            f3[0] = pl[4];
                            f3[1] = pl[5];
                            f3[2] = (uint32_t)(h == 0 ? sh_b : sl_b);
                        }
                    }
                }
        return WSI_OK;
    }
    // planes 2 (fp16 pair, common.h PairElem): every output channel's folded weights are multiplied by a power of two that puts
    // the channel's largest magnitude into [2^13, 2^14) - exact, and any weight within 2^-16 of the largest then has a NORMAL fp16
    // lo part (22 significand bits), whatever the magnitude of the trained weights; the inverse scales follow the fragment
    // blocks and the conv epilogues multiply the accumulators by them (common.h conv_wscale_inv)
    std::vector<float> wmul(cout, 1.0f);
    if (planes == 2) {
        float* inv = (float*)((char*)wpk_out + (size_t)(cout / 32) * NL * NT * 4096);
        for (int co = 0; co < cout; ++co) {
            double sc, sh;
            bn_fold(bn_weight, bn_bias, bn_mean, bn_var, eps, co, sc, sh);
            float amax = 0.f;
            for (size_t i = 0; i < (size_t)cin * k * k; ++i) amax = fmaxf(amax, fabsf((float)((double)w[(size_t)co * cin * k * k + i] * sc)));
            int e = 0;
            if (amax > 0.f && std::isfinite(amax)) {
                frexpf(amax, &e);                                                  // amax = m * 2^e, m in [0.5, 1)
                e = 14 - e;                                                        // amax * 2^e in [2^13, 2^14)
                e = e > 100 ? 100 : (e < -100 ? -100 : e);
            }
            wmul[co] = ldexpf(1.0f, e);
            inv[co] = ldexpf(1.0f, -e);
        }
    }
    for (int nt = 0; nt < cout / 32; ++nt)
        for (int l = 0; l < NL; ++l)
            for (int t = 0; t < NT; ++t)
                for (int f = 0; f < 4; ++f) {
                    uint16_t* frag = o + ((((size_t)nt * NL + l) * NT + t) * 4 + f) * 512;
                    const int plane = planes == 2 ? (f >> 1) : 0;
                    const int cbase = planes == 2 ? 32 * l + 16 * (f & 1) : 64 * l + 16 * f;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = nt * 32 + (lane & 31);
                        double sc, sh;
                        bn_fold(bn_weight, bn_bias, bn_mean, bn_var, eps, co, sc, sh);
                        for (int j = 0; j < 8; ++j) {
                            const int ci = cbase + 8 * (lane >> 5) + j;
                            const float wf = (float)((double)w[(((size_t)co * cin + ci) * k + t / k) * k + t % k] * sc);
                            uint16_t hi, lo;
                            if (planes == 2) split_host_f16(wf * wmul[co], hi, lo);
                            else split_host(wf, hi, lo);
                            frag[lane * 8 + j] = plane ? lo : hi;
                        }
                    }
                }
    return WSI_OK;
}

// Weights of the fused decoder tail (tail.hip): the last decoder block's two 3x3 convs (BN folded) and the 1x1 head, parity mode.
//   conv1 [2 py][6 taps = 2 low rows x 3 low columns][4 fragments] x 1 KiB: A rows 0-15 = output channels at px = 0, rows 16-31 at px = 1;
//         the weight of low-resolution offset (oy, ox) = the float64 SUM of the 3x3 taps whose upsampled source falls on it
//         (py = 0: dy 0 -> oy -1, dy 1, 2 -> oy 0; py = 1: dy 0, 1 -> oy 0, dy 2 -> oy +1; columns alike), rounded to fp32
//   conv2 [12 taps = 4 input rows x 3 columns][hi, lo] x 1 KiB: A rows 0-15 = output row 2k - 1 (dy = input row), rows 16-31 = row 2k
//         (dy = input row - 1), K = the 16 channels
//   then fp32: 1 / scale of conv1 [16], bias1 [16], 1 / scale of conv2 [16], bias2 [16], head_w [4][16] (zero padded), head_b [4]
// Every output channel's weights carry a power-of-two scale that puts its largest magnitude into [2^13, 2^14) (as wsi_prepack_conv).
size_t wsi_unet_tail_prepack_bytes(void) { return (size_t)(2 * 6 * 4 + 12 * 2) * 1024 + 132 * sizeof(float); }

int wsi_unet_tail_prepack(const float* w1, const float* bn1_weight, const float* bn1_bias, const float* bn1_mean, const float* bn1_var,
                          const float* w2, const float* bn2_weight, const float* bn2_bias, const float* bn2_mean, const float* bn2_var,
                          float eps, const float* head_w, const float* head_b, int cin, int cmid, int classes, void* out) {
    if (!w1 || !w2 || !head_w || !out || cin != 32 || cmid < 1 || cmid > 16 || classes < 1 || classes > 4) return WSI_EINVAL;
    memset(out, 0, wsi_unet_tail_prepack_bytes());
    uint16_t* o1 = (uint16_t*)out;
    uint16_t* o2 = (uint16_t*)((char*)out + 2 * 6 * 4 * 1024);
    float* fl = (float*)((char*)out + (2 * 6 * 4 + 12 * 2) * 1024);
    auto scale_of = [](float amax, float& mul, float& inv) {
        int e = 0;
        if (amax > 0.f && std::isfinite(amax)) {
            frexpf(amax, &e);
            e = 14 - e;
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
        }
        mul = ldexpf(1.0f, e);
        inv = ldexpf(1.0f, -e);
    };
    // conv1: combined (polyphase) weights wc[py][a][oxi][px][c][ci]
    static const int lo_set[2][2][3] = {{{1, 0, 0}, {0, 1, 1}}, {{1, 1, 0}, {0, 0, 1}}};     // [parity][first / second low offset][d] -> d contributes
    std::vector<float> wc((size_t)2 * 2 * 3 * 2 * 16 * 32, 0.f);
    auto WC = [&](int py, int a_, int oxi, int px, int c, int ci) -> float& { return wc[(((((size_t)py * 2 + a_) * 3 + oxi) * 2 + px) * 16 + c) * 32 + ci]; };
    for (int c = 0; c < cmid; ++c) {
        double sc, sh;
        bn_fold(bn1_weight, bn1_bias, bn1_mean, bn1_var, eps, c, sc, sh);
        fl[16 + c] = (float)sh;
        float amax = 0.f;
        for (int py = 0; py < 2; ++py)
            for (int a_ = 0; a_ < 2; ++a_)
                for (int px = 0; px < 2; ++px)
                    for (int b_ = 0; b_ < 2; ++b_) {
                        const int oxi = px + b_;                                          // px = 0: offsets -1, 0; px = 1: offsets 0, +1
                        for (int ci = 0; ci < 32; ++ci) {
                            double sum = 0.0;
                            for (int dy = 0; dy < 3; ++dy)
                                for (int dx = 0; dx < 3; ++dx)
                                    if (lo_set[py][a_][dy] && lo_set[px][b_][dx])
                                        sum += (double)(float)((double)w1[(((size_t)c * cin + ci) * 3 + dy) * 3 + dx] * sc);
                            const float v = (float)sum;
                            WC(py, a_, oxi, px, c, ci) = v;
                            amax = fmaxf(amax, fabsf(v));
                        }
                    }
        float mul, inv;
        scale_of(amax, mul, inv);
        fl[c] = inv;
        for (int py = 0; py < 2; ++py)
            for (int a_ = 0; a_ < 2; ++a_)
                for (int oxi = 0; oxi < 3; ++oxi)
                    for (int px = 0; px < 2; ++px)
                        for (int ci = 0; ci < 32; ++ci) WC(py, a_, oxi, px, c, ci) *= mul;
    }
    for (int py = 0; py < 2; ++py)
        for (int t = 0; t < 6; ++t)
            for (int f = 0; f < 4; ++f) {
                uint16_t* frag = o1 + (size_t)((py * 6 + t) * 4 + f) * 512;
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = lane & 31, px = row >> 4, c = row & 15;
                    for (int j = 0; j < 8; ++j) {
                        const int ci = 16 * (f & 1) + 8 * (lane >> 5) + j;
                        uint16_t hi, lo;
                        split_host_f16(WC(py, t / 3, t % 3, px, c, ci), hi, lo);
                        frag[lane * 8 + j] = (f >> 1) ? lo : hi;
                    }
                }
            }
    // conv2
    std::vector<float> w2s((size_t)16 * 16 * 9, 0.f);
    for (int c = 0; c < cmid; ++c) {
        double sc, sh;
        bn_fold(bn2_weight, bn2_bias, bn2_mean, bn2_var, eps, c, sc, sh);
        fl[48 + c] = (float)sh;
        float amax = 0.f;
        for (int ci = 0; ci < cmid; ++ci)
            for (int t = 0; t < 9; ++t) {
                const float v = (float)((double)w2[((size_t)c * cmid + ci) * 9 + t] * sc);
                w2s[((size_t)c * 16 + ci) * 9 + t] = v;
                amax = fmaxf(amax, fabsf(v));
            }
        float mul, inv;
        scale_of(amax, mul, inv);
        fl[32 + c] = inv;
        for (int ci = 0; ci < 16; ++ci)
            for (int t = 0; t < 9; ++t) w2s[((size_t)c * 16 + ci) * 9 + t] *= mul;
    }
    for (int t = 0; t < 12; ++t)
        for (int p = 0; p < 2; ++p) {
            uint16_t* frag = o2 + (size_t)(t * 2 + p) * 512;
            for (int lane = 0; lane < 64; ++lane) {
                const int row = lane & 31, rs = row >> 4, c = row & 15, dy = t / 3 - rs, dx = t % 3;
                for (int j = 0; j < 8; ++j) {
                    const int ci = 8 * (lane >> 5) + j;
                    uint16_t hi, lo;
                    split_host_f16((dy >= 0 && dy <= 2) ? w2s[((size_t)c * 16 + ci) * 9 + dy * 3 + dx] : 0.f, hi, lo);
                    frag[lane * 8 + j] = p ? lo : hi;
                }
            }
        }
    for (int k = 0; k < classes; ++k) {
        for (int c = 0; c < cmid; ++c) fl[64 + k * 16 + c] = head_w[(size_t)k * cmid + c];
        fl[128 + k] = head_b ? head_b[k] : 0.f;
    }
    return WSI_OK;
}

size_t wsi_prepack_stem_bytes(int planes) {
    if (planes == 3) planes = 2;                       // mode 3 keeps the stem's own arithmetic in the split pair (fp16 hi/lo)
    return (planes < 1 || planes > 2) ? 0 : (size_t)2 * 14 * planes * 64 * 8 * 2;
}

int wsi_prepack_stem(const float* w, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                     const float* bn_var, float eps, int planes, void* wpk_out, float* bias_out) {
    if (planes == 3) planes = 2;
    if (!w || !wpk_out || !bias_out || planes < 1 || planes > 2) return WSI_EINVAL;
    uint16_t* o = (uint16_t*)wpk_out;
    for (int co = 0; co < 64; ++co) {
        double sc, sh;
        bn_fold(bn_weight, bn_bias, bn_mean, bn_var, eps, co, sc, sh);
        bias_out[co] = (float)sh;
    }
    for (int nt = 0; nt < 2; ++nt)
        for (int s = 0; s < 14; ++s)
            for (int p = 0; p < planes; ++p) {
                uint16_t* frag = o + ((size_t)(nt * 14 + s) * planes + p) * 512;
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = nt * 32 + (lane & 31), h = lane >> 5;
                    double sc, sh;
                    bn_fold(bn_weight, bn_bias, bn_mean, bn_var, eps, co, sc, sh);
                    for (int j = 0; j < 8; ++j) {
                        const int kh = s >> 1, kw = (s & 1) * 4 + 2 * h + (j >> 2), c = j & 3;
                        float wf = 0.f;
                        if (kw < 7 && c < 3) wf = (float)((double)w[(((size_t)co * 3 + c) * 7 + kh) * 7 + kw] * sc);
                        uint16_t hi, lo;
                        if (planes == 2) split_host_f16(wf, hi, lo);     // the stem's own arithmetic in split precision: fp16 pair (r05)
                        else split_host(wf, hi, lo);
                        frag[lane * 8 + j] = p ? lo : hi;
                    }
                }
            }
    return WSI_OK;
}

// Stem weights for the integer (u8 slide) path, stem.hip stem_pool_kernel<.., DIG>: per output channel the folded weights
//   w'(c, kh, kw) = W bn_scale / (255 std[c])            on the colour bytes (x - 128)
//   k'(kh, kw)    = sum_c w'(c, kh, kw) (128 - 255 mean[c]) / 127      on the "inside" byte (127 inside the tile, 0 in the padding)
// so that  sum w' (x - 128) + sum_inside 127 k' + bn_shift == conv(W, (x/255 - mean)/std) bn_scale + bn_shift  exactly,
// written as fixed-point numbers q * scale[co] with q in DIG balanced base-256 digits (each an i8 in [-128, 127];
// |q| <= 127 * 256^(DIG-1)), DIG = 3 (24 bits) in both split-precision modes.
// Layout: [nt 2][kh 7][digit DIG][lane 64][16 B: k = 16 h + j -> kw = 4 h + (j >> 2), byte j & 3], then float scale[64]
// at byte 2 * 7 * 3 * 1024 (inside the wsi_prepack_stem_bytes(2) buffer the callers allocate); bias_out = bn_shift.
int wsi_prepack_stem_u8(const float* w, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                        const float* bn_var, float eps, const float mean[3], const float std_[3], int planes,
                        void* wpk_out, float* bias_out) {
    if (!w || !wpk_out || !bias_out || !mean || !std_ || planes < 2 || planes > 3) return WSI_EINVAL;
    const int DIG = 3;                                                        // stem.hip launches stem_pool_kernel<.., 3> in both modes
    int8_t* o = (int8_t*)wpk_out;
    float* scale_out = (float*)((char*)wpk_out + 2 * 7 * 3 * 1024);
    memset(wpk_out, 0, (size_t)2 * 7 * 3 * 1024 + 64 * sizeof(float));
    const double qmax = DIG == 3 ? 127.0 * 65536.0 : 127.0 * 256.0;
    for (int co = 0; co < 64; ++co) {
        double sc, sh;
        bn_fold(bn_weight, bn_bias, bn_mean, bn_var, eps, co, sc, sh);
        bias_out[co] = (float)sh;
        double val[7][8][4];                                                    // [kh][kw (7 -> 8)][colour bytes 0-2, inside byte 3]
        double amax = 0.0;
        for (int kh = 0; kh < 7; ++kh)
            for (int kw = 0; kw < 8; ++kw) {
                double kap = 0.0;
                for (int c = 0; c < 3; ++c) {
                    const double wd = kw < 7 ? (double)w[(((size_t)co * 3 + c) * 7 + kh) * 7 + kw] * sc / (255.0 * (double)std_[c]) : 0.0;
                    val[kh][kw][c] = wd;
                    kap += wd * (128.0 - 255.0 * (double)mean[c]);
                    amax = fmax(amax, fabs(wd));
                }
                val[kh][kw][3] = kap / 127.0;
                amax = fmax(amax, fabs(val[kh][kw][3]));
            }
        const double scale = amax > 0.0 ? amax / qmax : 1.0;
        scale_out[co] = (float)scale;
        const double fscale = (double)scale_out[co];                            // quantise against the fp32 scale the kernel multiplies by
        const int nt = co >> 5, l31 = co & 31;
        for (int kh = 0; kh < 7; ++kh)
            for (int kw = 0; kw < 8; ++kw)
                for (int c = 0; c < 4; ++c) {
                    long long q = llround(val[kh][kw][c] / fscale);
                    const int h = kw >> 2, j = (kw & 3) * 4 + c, lane = h * 32 + l31;
                    for (int d = 0; d < DIG; ++d) {
                        const long long dig = ((q + 128) & 255) - 128;           // balanced digit (two's-complement safe: & on negatives is modular)
                        q = (q - dig) / 256;
                        o[(((size_t)(nt * 7 + kh) * DIG + d) * 64 + lane) * 16 + j] = (int8_t)dig;
                    }
                }
    }
    return WSI_OK;
}

int wsi_normalize_u8_lut(const float mean[3], const float std_[3], float* lut_out) {
    if (!mean || !std_ || !lut_out) return WSI_EINVAL;
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float t = (float)v / 255.0f;     // ToTensor: fp32 division
            volatile float d = t - mean[c];           // Normalize: sub, then div, each rounded to fp32
            lut_out[c * 256 + v] = d / std_[c];
        }
    return WSI_OK;
}

// ------------------------------------------------------------------------------------ single ops
static int g_stem_fused = 1, g_stem_rows = 64;       // fused stem+maxpool kernel; pooled rows per workgroup (r05 sweep: 16 / 32 / 64 -> 6.41 / 6.16 / 6.08 ms per 6 162 tiles)
extern int g_stem_shared_weights;                     // stem.hip
static int g_stem_u8x = 1;                            // exact-u8 arithmetic when the caller supplies its weights (A/B: fused = 2 disables)

int wsi_stem_set_mode(int fused, int rows_per_seg) {
    if (rows_per_seg <= 0) return WSI_EINVAL;
    g_stem_fused = fused ? 1 : 0; g_stem_rows = rows_per_seg; g_stem_u8x = fused != 2;
    g_stem_shared_weights = fused != 3;                // 3: integer stem in its one-strip form (weights in registers), A/B
    return WSI_OK;
}

// out96 (trunk, mode 3): the pooled map is written in 96-byte lines (common.h CONV_OUT96) for a layer-1 kernel that reads them
static int stem_run(const float* in_f32, const uint8_t* slide, long long slide_pitch_bytes,
                    int slide_h, int slide_w, const int* tile_xy, const float* lut,
                    const void* stem_wpk, const float* stem_bias, const void* stem_wpk_u8,
                    const float* stem_bias_u8, const float* norm_mean_std, int n, int h, int w,
                    float* scratch, void* out_pf, int planes, void* stream, int out96, long long plane96 = 0, void* x0_pf = nullptr) {
    if (!stem_wpk || !stem_bias || !scratch || !out_pf || n <= 0 || h % 16 || w % 4) return WSI_EINVAL;
    if (!in_f32 && (!slide || !tile_xy || !lut)) return WSI_EINVAL;
    StemArgs a;
    a.mode = in_f32 ? 0 : 1;
    a.in_f32 = in_f32; a.slide = slide; a.slide_pitch = slide_pitch_bytes; a.SH = slide_h; a.SW = slide_w;
    a.origins = tile_xy; a.lut = lut; a.wpk = stem_wpk; a.bias = stem_bias; a.out = scratch;
    a.N = n; a.H = h; a.W = w;
    a.wpk_u8 = nullptr; a.bias_u8 = nullptr;
    // integer stem for u8 slide input (the transform is inside the packed weights; norm_mean_std is kept in the signature
    // for ABI stability and as the caller's statement of which transform those weights carry)
    if (stem_wpk_u8 && stem_bias_u8 && norm_mean_std && !in_f32 && g_stem_u8x) { a.wpk_u8 = stem_wpk_u8; a.bias_u8 = stem_bias_u8; }
    if (out96 && planes != 3) return WSI_EINVAL;
    if (x0_pf && !(a.wpk_u8 && g_stem_fused && planes == 2)) return WSI_EINVAL;       // (the x0 output: integer fused stem, fp16-pair lines)
    if (g_stem_fused || planes == 3) return wsi_stem_pool_dispatch(a, out_pf, planes, g_stem_rows, (hipStream_t)stream, out96, plane96, x0_pf);
    int rc = wsi_stem_dispatch(a, planes, (hipStream_t)stream);
    if (rc) return rc;
    return wsi_maxpool_dispatch(scratch, out_pf, n, h / 2, w / 2, planes, (hipStream_t)stream);
}

int wsi_stem_conv7x7_bn_relu_maxpool(const float* in_f32, const uint8_t* slide, long long slide_pitch_bytes,
                                     int slide_h, int slide_w, const int* tile_xy, const float* lut,
                                     const void* stem_wpk, const float* stem_bias, const void* stem_wpk_u8,
                                     const float* stem_bias_u8, const float* norm_mean_std, int n, int h, int w,
                                     float* scratch, void* out_pf, int planes, void* stream) {
    return stem_run(in_f32, slide, slide_pitch_bytes, slide_h, slide_w, tile_xy, lut, stem_wpk, stem_bias, stem_wpk_u8, stem_bias_u8,
                    norm_mean_std, n, h, w, scratch, out_pf, planes, stream, 0);
}

static int g_s2_slab = 1;                             // stride-2 convs: phase-slab kernel (1) or per-tap gather kernel (0)
static int g_s2_split = 1;                            // trunk: phase-split stage outputs + wide stride-2 kernel (A/B: wsi_conv_set_mode +128 off)
static int g_ds_fold = 1;                             // trunk, mode 3: the strided blocks' 1x1 downsample runs inside their second conv (A/B: +2048 off)

#ifdef WSI_STUDY
static void* g_study_debug = nullptr;                 // study builds: device buffer handed to stamped kernels through ConvArgs.out2
extern "C" int wsi_study_set_debug(void* dev_buf) { g_study_debug = dev_buf; return WSI_OK; }
#endif

static int conv_common(const void* in_pf, void* out_pf, const void* resid_pf, const void* wpk, const float* bias, int n,
                       int h_in, int w_in, int cin, int cout, int stride, int ksize, int relu, int planes, void* stream,
                       int cfg = -1, int split_out = 0, long long split_pixels = 0, const void* in2 = nullptr, int in2_c = 0,
                       const void* wpk2 = nullptr, const float* bias2 = nullptr, int line_flags = 0, const void* in_up = nullptr, int up_c = 0,
                       long long plane96 = 0) {
    if ((!in_pf && !(in_up && up_c == cin)) || !out_pf || !wpk || !bias || in_pf == out_pf || in_up == out_pf || n <= 0) return WSI_EINVAL;
    // 96-byte lines (CONV_IN96 / OUT96 / RESID96): mode 3, stride-1 3x3, 64 channels in and out (the slab3 kernel), no phase split
    if (line_flags && (planes != 3 || stride != 1 || ksize != 3 || cin != 64 || cout != 64 || (split_out && (line_flags & CONV_OUT96)) ||
                       (line_flags & ~(CONV_IN96 | CONV_OUT96 | CONV_RESID96)) || ((line_flags & CONV_RESID96) && !resid_pf)))
        return WSI_EINVAL;
    if ((stride != 1 && stride != 2) || h_in % stride || w_in % stride) return WSI_EINVAL;
    ConvArgs a;
    a.in = in_pf; a.out = out_pf; a.resid = resid_pf; a.wpk = wpk; a.bias = bias;
    a.gi = pf_geom_fd(n, h_in, w_in, cin);
    a.go = pf_geom_fd(n, h_in / stride, w_in / stride, cout);
    a.stride = stride; a.ksize = ksize; a.relu = relu & 1; a.flags = 0;
    a.plane96 = line_flags ? (plane96 > 0 ? plane96 : (long long)pf_alloc_pixels(n, h_in, w_in) * 96) : 0;   // line-planar 96-byte tensors (common.h)
#ifdef WSI_STUDY
    // study builds accept the r01 ablation masks of tools/tune_conv.py in `relu` (2 no stores, 64 dispatch only, 128 no main
    // loop, 256 non-temporal, bits 10-13 weight copies, 512 / 16384 XCD orders, 65536 residual read directly)
    if (relu & 2) a.flags |= CONV_ABL_NO_STORE;
    if (relu & 64) a.flags |= CONV_ABL_DISPATCH_ONLY;
    if (relu & 128) a.flags |= CONV_ABL_NO_MAINLOOP;
    if (relu & 256) a.flags |= CONV_NONTEMPORAL;
    if (relu & 512) a.flags |= CONV_XCD_ORDER;
    if (relu & 16384) a.flags |= CONV_XCD_RANGES;
    if (relu & 65536) a.flags |= CONV_RESID_DIRECT;
    a.flags |= ((relu >> 10) & 15) << CONV_WCOPIES_SHIFT;
#else
    if (relu & ~1) return WSI_EINVAL;
#endif
    a.flags |= line_flags;
    a.out2 = nullptr; a.wpk2 = nullptr; a.bias2 = nullptr;
    if (in_up) {                                       // fused nearest x2 upsample + concat input (common.h ConvArgs.in_up): stride-1 3x3, slab3 kernels
        if (stride != 1 || ksize != 3 || h_in % 2 || w_in % 2 || up_c <= 0 || up_c > cin || resid_pf || in2 || line_flags || split_out) return WSI_EINVAL;
        a.in_up = in_up; a.up_c = up_c; a.gup = pf_geom_fd(n, h_in / 2, w_in / 2, up_c);
    }
    if (in2) {                                         // extra K segment (common.h ConvArgs.in2): mode 3, stride-1 3x3, wide kernel only
        if (planes != 3 || stride != 1 || ksize != 3 || resid_pf || !wpk2 || !bias2 || in2_c <= 0 || in2_c % 32 || cout % 128 || cfg >= 0) return WSI_EINVAL;
        a.in2 = in2; a.in2_c = in2_c; a.wpk2 = wpk2; a.bias2 = bias2;
        cfg = 60;
    }
#ifdef WSI_STUDY
    if (cfg == 75 || cfg == 76) a.out2 = g_study_debug;
#endif
    a.in_split_pixels = 0;
    // distance between the four phase images: the caller's (a workspace planned for more images) or the tight one
    a.out_split_pixels = split_out ? (split_pixels ? split_pixels : pf_alloc_pixels(n, h_in / 2, w_in / 2)) : 0;
    if (split_out && (stride != 1 || ksize != 3 || h_in % 2 || w_in % 2 || planes < 2)) return WSI_EINVAL;
    if (ksize == 3 && stride == 2 && cout % 128 == 0 && cfg != 0 && g_s2_slab) {
        const int rc = wsi_s2_dispatch(a, planes, (hipStream_t)stream);
        if (rc != WSI_EINVAL) return rc;               // EINVAL: shape outside the slab kernel's range -> gather kernel
    }
    return wsi_conv_dispatch(a, planes, cfg, (hipStream_t)stream);
}

int wsi_conv3x3_bn_act(const void* in_pf, void* out_pf, const void* resid_pf, const void* wpk, const float* bias,
                       int n, int h_in, int w_in, int cin, int cout, int stride, int relu, int planes,
                       void* stream) {
    return conv_common(in_pf, out_pf, resid_pf, wpk, bias, n, h_in, w_in, cin, cout, stride, 3, relu, planes, stream);
}

size_t wsi_pf_split_bytes(int n, int h, int w, int c, int planes) {
    if (h % 2 || w % 2) return 0;
    return 4 * wsi_pf_bytes(n, h / 2, w / 2, c, planes);
}

int wsi_conv3x3_bn_act_split(const void* in_pf, void* out_split, const void* resid_pf, const void* wpk, const float* bias,
                             int n, int h, int w, int cin, int cout, int relu, int planes, void* stream) {
    return conv_common(in_pf, out_split, resid_pf, wpk, bias, n, h, w, cin, cout, 1, 3, relu, planes, stream, -1, 1);
}

static int s2_split_common(const void* in_split, void* out_conv_pf, void* out_ds_pf, const void* wpk3,
                           const float* bias3, const void* wpk1, const float* bias1, int n, int h_in, int w_in,
                           int cin, int cout, int planes, void* stream, long long split_pixels);

int wsi_conv3x3s2_ds_fused_split(const void* in_split, void* out_conv_pf, void* out_ds_pf, const void* wpk3,
                                 const float* bias3, const void* wpk1, const float* bias1, int n, int h_in, int w_in,
                                 int cin, int cout, int planes, void* stream) {
    return s2_split_common(in_split, out_conv_pf, out_ds_pf, wpk3, bias3, wpk1, bias1, n, h_in, w_in, cin, cout, planes, stream, 0);
}

static int s2_split_common(const void* in_split, void* out_conv_pf, void* out_ds_pf, const void* wpk3,
                           const float* bias3, const void* wpk1, const float* bias1, int n, int h_in, int w_in,
                           int cin, int cout, int planes, void* stream, long long split_pixels) {
    // out_ds_pf == null: the 3x3 conv alone (the trunk then computes the downsample inside the block's second conv)
    if (!in_split || !out_conv_pf || !wpk3 || !bias3 || (out_ds_pf && (!wpk1 || !bias1)) || n <= 0 || h_in % 2 || w_in % 2)
        return WSI_EINVAL;
    if (in_split == out_conv_pf || in_split == out_ds_pf || out_conv_pf == out_ds_pf) return WSI_EINVAL;
    ConvArgs a;
    a.in = in_split; a.out = out_conv_pf; a.resid = nullptr; a.wpk = wpk3; a.bias = bias3;
    a.gi = pf_geom_fd(n, h_in, w_in, cin);
    a.go = pf_geom_fd(n, h_in / 2, w_in / 2, cout);
    a.stride = 2; a.ksize = 3; a.relu = 1; a.flags = 0;
    a.out2 = out_ds_pf; a.wpk2 = out_ds_pf ? wpk1 : nullptr; a.bias2 = out_ds_pf ? bias1 : nullptr;
    a.out_split_pixels = 0;
    a.in_split_pixels = split_pixels ? split_pixels : pf_alloc_pixels(n, h_in / 2, w_in / 2);
    return wsi_s2_dispatch(a, planes, (hipStream_t)stream);      // EINVAL outside the wide kernel's range (output maps wider than 33)
}

int wsi_conv3x3_bn_act_cfg(const void* in_pf, void* out_pf, const void* resid_pf, const void* wpk, const float* bias,
                           int n, int h_in, int w_in, int cin, int cout, int stride, int relu, int planes, int cfg,
                           void* stream) {
    return conv_common(in_pf, out_pf, resid_pf, wpk, bias, n, h_in, w_in, cin, cout, stride, 3, relu, planes, stream, cfg);
}

int wsi_conv3x3_up_concat_bn_act(const void* up_pf, const void* skip_pf, void* out_pf, const void* wpk, const float* bias, int n, int h, int w,
                                 int c_up, int c_skip, int cout, int relu, int planes, void* stream) {
    if (!up_pf || c_up <= 0 || c_skip < 0 || (c_skip > 0 && !skip_pf)) return WSI_EINVAL;
    return conv_common(c_skip ? skip_pf : nullptr, out_pf, nullptr, wpk, bias, n, h, w, c_up + c_skip, cout, 1, 3, relu, planes, stream, -1, 0, 0,
                       nullptr, 0, nullptr, nullptr, 0, up_pf, c_up);
}

int wsi_conv3x3s2_ds_fused(const void* in_pf, void* out_conv_pf, void* out_ds_pf, const void* wpk3, const float* bias3,
                           const void* wpk1, const float* bias1, int n, int h_in, int w_in, int cin, int cout, int planes,
                           void* stream) {
    if (!in_pf || !out_conv_pf || !out_ds_pf || !wpk3 || !bias3 || !wpk1 || !bias1 || n <= 0 || h_in % 2 || w_in % 2)
        return WSI_EINVAL;
    if (in_pf == out_conv_pf || in_pf == out_ds_pf || out_conv_pf == out_ds_pf) return WSI_EINVAL;
    ConvArgs a;
    a.in = in_pf; a.out = out_conv_pf; a.resid = nullptr; a.wpk = wpk3; a.bias = bias3;
    a.gi = pf_geom_fd(n, h_in, w_in, cin);
    a.go = pf_geom_fd(n, h_in / 2, w_in / 2, cout);
    a.stride = 2; a.ksize = 3; a.relu = 1; a.flags = 0;
    a.out2 = out_ds_pf; a.wpk2 = wpk1; a.bias2 = bias1;
    a.in_split_pixels = 0; a.out_split_pixels = 0;
    int rc = wsi_s2_dispatch(a, planes, (hipStream_t)stream);
    if (rc == WSI_EINVAL) {                             // e.g. maps wider than 33: two per-tap gather launches
        rc = conv_common(in_pf, out_conv_pf, nullptr, wpk3, bias3, n, h_in, w_in, cin, cout, 2, 3, 1, planes, stream, 0);
        if (!rc) rc = conv_common(in_pf, out_ds_pf, nullptr, wpk1, bias1, n, h_in, w_in, cin, cout, 2, 1, 0, planes, stream, 0);
    }
    return rc;
}

extern int g_s2_small_tiles, g_xcd_order, g_wide_min_c, g_s2_ablate, g_xcd_ranges, g_slab_pair;
extern int g_l1_lines96, g_s2_nt4, g_l1_rows, g_wide_d8, g_l1p;
extern int g_unet_fuse_up;
extern int g_unet_tail;
extern int g_unet_tail_form;
extern int g_unet_x0_fused;
int wsi_conv_set_mode(int s2_slab) {
    g_s2_split = (s2_slab & 128) ? 0 : 1;
    g_ds_fold = (s2_slab & 2048) ? 0 : 1;
    g_slab_pair = (s2_slab & 4096) ? 0 : 1;
    g_l1_rows = (s2_slab & 1024) ? 0 : 1;
    g_unet_fuse_up = (s2_slab & 65536) ? 0 : 1;
    g_unet_tail = (s2_slab & 2097152) ? 0 : 1;
    g_unet_tail_form = (s2_slab & 4194304) ? 1 : 2;
    g_unet_x0_fused = (s2_slab & 8388608) ? 0 : 1;
    g_l1_lines96 = (s2_slab & 16384) ? 0 : 1;
    g_s2_nt4 = (s2_slab & 32768) ? 0 : 1;
    g_wide_d8 = (s2_slab & 131072) ? 0 : 1;
    g_l1p = (s2_slab & 1048576) ? 1 : 0;                  // A/B: persistent producer-fed layer-1 kernel (conv.hip conv3x3s1_l1p_kernel, r05)              // A/B: 8-pixel slab rows of the wide kernel on 8 x 8 maps (r05)
    g_xcd_ranges = (s2_slab & 256) ? 0 : (s2_slab & 512) ? 1 : 2;          // +256: off, +512: 64-channel layer only
    g_s2_ablate = (s2_slab & 64) ? 1 : 0;                 // bottleneck study only: stride-2 kernel without weight loads (wrong results)
    g_xcd_order = (s2_slab & 8) ? 1 : 0;
    g_wide_min_c = (s2_slab & 16) ? 256 : (s2_slab & 32) ? (1 << 30) : 128;      // +16: wide kernel from 256 channels, +32: never
    s2_slab &= 7;
    g_s2_slab = s2_slab ? 1 : 0; g_s2_small_tiles = s2_slab != 3;
    return WSI_OK;
}

int wsi_conv1x1_bn(const void* in_pf, void* out_pf, const void* wpk, const float* bias, int n, int h_in, int w_in,
                   int cin, int cout, int stride, int planes, void* stream) {
    return conv_common(in_pf, out_pf, nullptr, wpk, bias, n, h_in, w_in, cin, cout, stride, 1, 0, planes, stream);
}

int wsi_avgpool_fc(const void* in_pf, int n, int h, int w, int c, const float* fc_w, const float* fc_b, int k,
                   float* feat_out, float* logits_out, int planes, void* stream) {
    if (!in_pf || n <= 0 || (logits_out && (!fc_w || !fc_b || k <= 0))) return WSI_EINVAL;
    return wsi_avgpool_fc_dispatch(in_pf, pf_geom(n, h, w, c), fc_w, fc_b, k, feat_out, logits_out, planes,
                                   (hipStream_t)stream);
}

int wsi_linear(const float* x, const float* w, const float* bias, float* y, int b, int k, int j, int relu,
               void* stream) {
    if (!x || !w || !y) return WSI_EINVAL;
    return wsi_linear_dispatch(x, w, bias, y, b, k, j, relu, (hipStream_t)stream);
}

int wsi_pf_pack(const float* in_nchw, void* out_pf, int n, int c, int h, int w, int planes, void* stream) {
    if (!in_nchw || !out_pf || n <= 0 || planes < 1 || planes > 3 || c % (planes == 1 ? 64 : 32)) return WSI_EINVAL;
    return wsi_pf_pack_dispatch(in_nchw, out_pf, pf_geom(n, h, w, c), planes, (hipStream_t)stream);
}

int wsi_pf_unpack(const void* in_pf, float* out_nchw, int n, int c, int h, int w, int planes, void* stream) {
    if (!in_pf || !out_nchw || n <= 0 || planes < 1 || planes > 3 || c % (planes == 1 ? 64 : 32)) return WSI_EINVAL;
    return wsi_pf_unpack_dispatch(in_pf, out_nchw, pf_geom(n, h, w, c), planes, (hipStream_t)stream);
}

int wsi_tile_gather(const uint8_t* slide, long long slide_pitch_bytes, int slide_h, int slide_w, const int* tile_xy,
                    const float* lut, float* out_nchw, int n, int ph, int pw, void* stream) {
    if (!slide || !tile_xy || !lut || !out_nchw) return WSI_EINVAL;
    return wsi_tile_gather_dispatch(slide, slide_pitch_bytes, slide_h, slide_w, tile_xy, lut, out_nchw, n, ph, pw,
                                    (hipStream_t)stream);
}

int wsi_stitch_add(const float* tile_logits, const int* map_xy, int t, int c, int dy, int dx, double* pred, int map_h,
                   int map_w, void* stream) {
    if (!tile_logits || !map_xy || !pred || map_h <= 0 || map_w <= 0) return WSI_EINVAL;
    return wsi_stitch_add_dispatch(tile_logits, map_xy, t, c, dy, dx, pred, map_h, map_w, (hipStream_t)stream);
}

int wsi_stitch_add_dense(const float* tile_pred, const int* map_xy, int t, int c, int ph, int pw, double* pred, int map_h,
                         int map_w, void* stream) {
    if (!tile_pred || !map_xy || !pred || map_h <= 0 || map_w <= 0) return WSI_EINVAL;
    return wsi_stitch_add_dense_dispatch(tile_pred, map_xy, t, c, ph, pw, pred, map_h, map_w, (hipStream_t)stream);
}

int wsi_softmax_threshold_argmax(const double* pred, int c, long long hw, const double* class_thresh, double* probs,
                                 uint8_t* classes, const uint8_t* mask, int heat_mode, uint8_t* heat, void* stream) {
    if (!pred || !class_thresh) return WSI_EINVAL;
    return wsi_softmax_dispatch(pred, c, hw, class_thresh, probs, classes, mask, heat_mode, heat, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ ingestion ring + input resize
int wsi_ring_create(wsi_ring** out, int slots, size_t slot_bytes) { return wsi_ring_create_impl(out, slots, slot_bytes); }
void* wsi_ring_host_slot(wsi_ring* r, int slot) { return wsi_ring_host_slot_impl(r, slot); }
int wsi_ring_wait_slot(wsi_ring* r, int slot) { return wsi_ring_wait_slot_impl(r, slot); }
int wsi_ring_submit(wsi_ring* r, int slot, int rows, int width, int channels, long long src_pitch, uint8_t* level_rows, long long level_pitch) {
    return wsi_ring_submit_impl(r, slot, rows, width, channels, src_pitch, level_rows, level_pitch);
}
int wsi_ring_fence(wsi_ring* r, void* compute_stream) { return wsi_ring_fence_impl(r, (hipStream_t)compute_stream); }
int wsi_ring_acquire(wsi_ring* r, void* compute_stream) { return wsi_ring_acquire_impl(r, (hipStream_t)compute_stream); }
int wsi_ring_device(const wsi_ring* r) { return wsi_ring_device_impl(r); }
int wsi_ring_drain(wsi_ring* r) { return wsi_ring_drain_impl(r); }
void wsi_ring_destroy(wsi_ring* r) { wsi_ring_destroy_impl(r); }
int wsi_resample_plan_create(wsi_resample_plan** out, int in_h, int in_w, int out_h, int out_w) {
    return wsi_resample_plan_create_impl(out, in_h, in_w, out_h, out_w);
}
void wsi_resample_plan_destroy(wsi_resample_plan* p) { wsi_resample_plan_destroy_impl(p); }
size_t wsi_resample_scratch_bytes(const wsi_resample_plan* p, int n) { return wsi_resample_scratch_bytes_impl(p, n); }
int wsi_resample_tiles(const wsi_resample_plan* p, const uint8_t* slide, long long pitch, int sh, int sw, const int* tile_xy, int n,
                       uint8_t* out, void* scratch, void* stream) {
    return wsi_resample_tiles_impl(p, slide, pitch, sh, sw, tile_xy, n, out, scratch, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ region proposals
int wsi_find_nuclei_hsv(const uint8_t* rgb, long long npix, int pixel_stride, double mu_percent, uint8_t* mask_out, void* stream) {
    if (!rgb || !mask_out) return WSI_EINVAL;
    return wsi_hsv_mask_dispatch(rgb, npix, pixel_stride, mu_percent, mask_out, (hipStream_t)stream);
}
size_t wsi_connected_components_scratch_bytes(int h, int w) { return (h <= 0 || w <= 0) ? 0 : wsi_cc_scratch_bytes(h, w); }
int wsi_connected_components(const uint8_t* mask, int h, int w, int* labels_out, int* count_out, void* scratch, void* stream) {
    if (!mask || !labels_out || !scratch) return WSI_EINVAL;
    return wsi_cc_dispatch(mask, h, w, labels_out, count_out, scratch, (hipStream_t)stream);
}
int wsi_find_nuclei_lab(const uint8_t* rgb, long long npix, int pixel_stride, double mu_percent, uint8_t* mask_out, void* scratch, void* stream) {
    return wsi_lab_mask_dispatch(rgb, npix, pixel_stride, mu_percent, mask_out, scratch, (hipStream_t)stream);
}
size_t wsi_fill_holes_scratch_bytes(int h, int w) { return (h <= 0 || w <= 0) ? 0 : wsi_fill_holes_scratch_bytes_impl(h, w); }
int wsi_fill_holes(const uint8_t* mask, int h, int w, uint8_t* out, void* scratch, void* stream) {
    return wsi_fill_holes_dispatch(mask, h, w, out, scratch, (hipStream_t)stream);
}
size_t wsi_slic_scratch_bytes(int h, int w, int k) { return wsi_slic_scratch_bytes_impl(h, w, k); }
int wsi_slic(const uint8_t* rgb, int h, int w, const double* gauss_weights, int radius, double* segments, int k, int step_y, int step_x,
             double step, double compactness, int iters, int* labels_out, void* scratch, void* stream) {
    return wsi_slic_dispatch(rgb, h, w, gauss_weights, radius, segments, k, step_y, step_x, step, compactness, iters, labels_out, scratch,
                             (hipStream_t)stream);
}
int wsi_kmeans_points(const int* points_xy, int n, double* centres_xy, int k, int max_iters, int* labels_out, void* scratch, void* stream) {
    if (!points_xy || !centres_xy || !labels_out || !scratch) return WSI_EINVAL;
    return wsi_kmeans_dispatch(points_xy, n, centres_xy, k, max_iters, labels_out, scratch, (hipStream_t)stream);
}

int wsi_kmeans_seed_farthest(const int* points_xy, int n, int k, double* centres_xy_out, void* scratch, void* stream) {
    if (!points_xy || !centres_xy_out || !scratch) return WSI_EINVAL;
    return wsi_kmeans_seed_farthest_dispatch(points_xy, n, k, centres_xy_out, scratch, (hipStream_t)stream);
}

long long wsi_tile_grid_candidates(int iw, int ih, int ph, int pw, int sh, int sw) { return wsi_tile_grid_candidates_impl(iw, ih, ph, pw, sh, sw); }
size_t wsi_tile_grid_scratch_bytes(long long candidates) { return candidates < 0 ? 0 : wsi_tile_grid_scratch_bytes_impl(candidates); }
int wsi_tile_grid(int iw, int ih, int ph, int pw, int sh, int sw, const uint8_t* mask, int mask_h, int mask_w, double m, double thresh,
                  int* tile_xy_out, int* count_out, void* scratch, void* stream) {
    if (!tile_xy_out || !count_out || !scratch) return WSI_EINVAL;
    return wsi_tile_grid_dispatch(iw, ih, ph, pw, sh, sw, mask, mask_h, mask_w, m, thresh, tile_xy_out, count_out, scratch, (hipStream_t)stream);
}

int wsi_exponent_span(const float* values, long long n, int* out2, void* stream) {
    if (!values || !out2) return WSI_EINVAL;
    return wsi_exponent_span_dispatch(values, n, out2, (hipStream_t)stream);
}

int wsi_paint_regions(const long long* pixel_idx, const int* region_of, long long n, const uint8_t* region_class, int* winner_scratch,
                      long long* label, long long npix, void* stream) {
    if (!region_class || !winner_scratch || !label || (n && (!pixel_idx || !region_of))) return WSI_EINVAL;
    return wsi_paint_dispatch(pixel_idx, region_of, n, region_class, winner_scratch, label, npix, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ tumour-bed post-process
int wsi_resize_bilinear_f64(const double* src, int c, int hs, int ws, double* dst, int hd, int wd, void* stream) {
    if (!src || !dst || src == dst) return WSI_EINVAL;
    return wsi_resize_dispatch(src, c, hs, ws, dst, hd, wd, (hipStream_t)stream);
}
int wsi_argmax_classes(const double* pred, int c, long long hw, uint8_t* classes, void* stream) {
    if (!pred || !classes) return WSI_EINVAL;
    return wsi_argmax_dispatch(pred, c, hw, classes, (hipStream_t)stream);
}
int wsi_morph_rect(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int h, int w, int k, int op, void* stream) {
    if (!src || !dst || !tmp) return WSI_EINVAL;
    return wsi_morph_dispatch(src, dst, tmp, h, w, k, op, (hipStream_t)stream);
}
int wsi_bwperim(const uint8_t* src, uint8_t* dst, int h, int w, void* stream) {
    if (!src || !dst) return WSI_EINVAL;
    return wsi_bwperim_dispatch(src, dst, h, w, (hipStream_t)stream);
}
size_t wsi_tumor_bed_workspace_bytes(int h, int w) {
    if (h <= 0 || w <= 0) return 0;
    return 3 * align_up((size_t)h * w, 256) + align_up(wsi_hull_ws_bytes(h), 256);
}
int wsi_convex_hull_image(const uint8_t* src, uint8_t* dst, int h, int w, void* workspace, void* stream) {
    if (!src || !dst || !workspace || src == dst) return WSI_EINVAL;
    return wsi_hull_dispatch(src, dst, h, w, (char*)workspace + 3 * align_up((size_t)h * w, 256), (hipStream_t)stream);
}
int wsi_tumor_bed(const uint8_t* codes, int h, int w, int min_code, int open_k, int dilate_k, uint8_t* opened_out,
                  uint8_t* tb_pred_out, uint8_t* outline_out, void* workspace, void* stream) {
    if (!codes || !tb_pred_out || !outline_out || !workspace || h <= 0 || w <= 0 || open_k <= 0 || dilate_k <= 0 ||
        tb_pred_out == outline_out)
        return WSI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t plane = align_up((size_t)h * w, 256);
    uint8_t *A = (uint8_t*)workspace, *B = A + plane, *Cc = B + plane;
    void* hws = Cc + plane;
    int rc = wsi_threshold_dispatch(codes, (long long)h * w, min_code, A, st);
    if (!rc) rc = wsi_morph_dispatch(A, B, Cc, h, w, open_k, 2, st);                 // MORPH_OPEN
    if (!rc && opened_out && hipMemcpyAsync(opened_out, B, (size_t)h * w, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = WSI_EFAULT;
    if (!rc) rc = wsi_hull_dispatch(B, tb_pred_out, h, w, hws, st);                  // chull
    if (!rc) rc = wsi_bwperim_dispatch(tb_pred_out, A, h, w, st);                    // bwperim
    if (!rc) rc = wsi_morph_dispatch(A, outline_out, Cc, h, w, dilate_k, 1, st);     // dilate
    return rc;
}
int wsi_hull_polygon(void* workspace, int h, int w, double* out_xy, int cap, int* count_out, void* stream) {
    if (!workspace || !out_xy || !count_out || h <= 0 || w <= 0 || cap <= 0) return WSI_EINVAL;
    char* hws = (char*)workspace + 3 * align_up((size_t)h * w, 256);
    int rc = wsi_hull_polygon_dispatch(hws, h, out_xy, cap, (hipStream_t)stream);
    if (rc) return rc;
    const int* counts = (const int*)hws + 2 * (2 * (size_t)h + 1);
    return hipMemcpyAsync(count_out, counts + 2, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? WSI_OK : WSI_EFAULT;
}
int wsi_mask_iou_counts(const uint8_t* a, const uint8_t* b, long long n, unsigned long long* out2, void* stream) {
    if (!a || !b || !out2) return WSI_EINVAL;
    return wsi_iou_counts_dispatch(a, b, n, out2, (hipStream_t)stream);
}
int wsi_score_counts(const uint8_t* p, const uint8_t* gt, const uint8_t* mask, long long n, unsigned long long* out6, void* stream) {
    if (!p || !gt || !out6) return WSI_EINVAL;
    return wsi_score_counts_dispatch(p, gt, mask, n, out6, (hipStream_t)stream);
}
int wsi_esp(const double* pts_xy, int n, int num_pts, double* out_xy, double* scratch, void* stream) {
    if (!pts_xy || !out_xy || !scratch) return WSI_EINVAL;
    return wsi_esp_dispatch(pts_xy, n, num_pts, out_xy, scratch, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ profiler
// Optional HIP-event timing of every conv launch made by wsi_trunk_forward, on the stream the
// kernels run on (bench.py's roofline leg).  Off by default; never active inside graph capture.
#define WSI_PROF_MAX 16384
static struct {
    int enabled, count, cap;
    hipEvent_t ev[2 * WSI_PROF_MAX];
    int kind[WSI_PROF_MAX];
    double flops[WSI_PROF_MAX];
    int created;
} g_prof;

int wsi_prof_begin(int max_records) {
    if (max_records <= 0 || max_records > WSI_PROF_MAX) return WSI_EINVAL;
    for (; g_prof.created < 2 * max_records; ++g_prof.created)
        if (hipEventCreate(&g_prof.ev[g_prof.created]) != hipSuccess) return WSI_ENOMEM;
    g_prof.cap = max_records; g_prof.count = 0; g_prof.enabled = 1;
    return WSI_OK;
}

int wsi_prof_end(float* ms_out, int* kind_out, double* flops_out, int cap) {
    g_prof.enabled = 0;
    int n = g_prof.count < cap ? g_prof.count : cap;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return WSI_EFAULT;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return WSI_EFAULT;
        ms_out[i] = ms; kind_out[i] = g_prof.kind[i]; flops_out[i] = g_prof.flops[i];
    }
    g_prof.count = 0;
    return n;
}

static inline int prof_open(hipStream_t st, int kind, double flops) {
    if (!g_prof.enabled || g_prof.count >= g_prof.cap) return -1;
    const int i = g_prof.count++;
    g_prof.kind[i] = kind; g_prof.flops[i] = flops;
    (void)hipEventRecord(g_prof.ev[2 * i], st);
    return i;
}
static inline void prof_close(hipStream_t st, int i) { if (i >= 0) (void)hipEventRecord(g_prof.ev[2 * i + 1], st); }

// ------------------------------------------------------------------------------------ trunk
static int g_chunk_stem = 0, g_chunk_l1 = 0;   // sub-batch sizes (images); 0 = whole batch (measured r01: no gain)

// Layer-1 tensors of a full mode-3 trunk run live in 96-byte lines (common.h CONV_IN96): the pad positions of a PF buffer sit at
// other BYTES than in the 128-byte layout, and pads are only ever zero because nobody writes them - so a workspace remembers
// which layout its three stage-0 buffers last held, and a run in the other layout zero-fills them first (taps and the U-Net
// encoder keep the 128-byte layout; a workspace that only ever runs one kind of call never pays).  -1 = all zero (after
// wsi_trunk_workspace_init), otherwise 2 * planes + (1 if stage 0 holds 96-byte lines); an unknown workspace counts as dirty.
int g_l1_lines96 = 1;                           // A/B: wsi_conv_set_mode +16384 disables
static std::mutex g_ws_mutex;
struct WsTag { int layout; size_t bytes; };                  // bytes: what wsi_trunk_workspace_init planned (0 = never initialised here)
static std::unordered_map<const void*, WsTag> g_ws_layout;
// returns 1 if the stage-0 buffers must be zero-filled first, 2 if everything must, -1 if the current plan (`need` bytes) exceeds
// what the workspace was initialised for (r04 advisor finding: a workspace sized for one planes value - 2 bytes per channel at
// planes 1 - and then run with another would be zero-filled and written past its end; the API carries no size, the tag does)
static int ws_layout_switch(const void* ws, int want, size_t need) {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    auto it = g_ws_layout.find(ws);
    const int have = it == g_ws_layout.end() ? -2 : it->second.layout;
    const size_t bytes = it == g_ws_layout.end() ? 0 : it->second.bytes;
    if (bytes && need > bytes) return -1;
    g_ws_layout[ws] = WsTag{want, bytes};
    if (have == want || have == -1) return 0;
    return (have >= 0 && have / 2 != want / 2) ? 2 : 1;      // 2: the workspace last ran another planes value - every pad may be dirty
}
// A workspace that is freed must be forgotten: a later allocation at the same address would inherit its layout tag (r03 advisor
// finding) and the map would grow without bound.  Unknown pointers are fine (nothing to forget).
int wsi_trunk_workspace_release(void* workspace) {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    g_ws_layout.erase(workspace);
    return WSI_OK;
}

int wsi_trunk_set_chunks(int stem_chunk, int layer1_chunk) {
    if (stem_chunk < 0 || layer1_chunk < 0) return WSI_EINVAL;
    if (stem_chunk && layer1_chunk && layer1_chunk % stem_chunk) return WSI_EINVAL;
    g_chunk_stem = stem_chunk; g_chunk_l1 = layer1_chunk;
    return WSI_OK;
}
struct TrunkPlan {
    size_t stem_scratch;          // byte offsets into the workspace
    size_t buf[4][4];             // [stage][0..2]: rotating PF buffers; [stage][3]: phase-split output of the stage (stages 0-2)
    size_t total;
    int sh[4], sw[4], sc[4];
};

static int trunk_plan(int n, int h, int w, int planes, TrunkPlan& p) {
    if (n <= 0 || h <= 0 || w <= 0 || h % 32 || w % 32 || planes < 1 || planes > 3) return WSI_EINVAL;
    size_t off = 0;
    p.stem_scratch = off;
    off += align_up((size_t)n * (h / 2) * (w / 2) * 64 * sizeof(float), 256);
    for (int s = 0; s < 4; ++s) {
        p.sh[s] = h >> (2 + s); p.sw[s] = w >> (2 + s); p.sc[s] = 64 << s;
        for (int b = 0; b < 3; ++b) {
            p.buf[s][b] = off;
            off += align_up(wsi_pf_bytes(n, p.sh[s], p.sw[s], p.sc[s], planes), 256);
        }
        p.buf[s][3] = off;                             // never holds anything but the phase-split form: its pads stay zero
        if (s < 3 && planes >= 2) off += align_up(wsi_pf_split_bytes(n, p.sh[s], p.sw[s], p.sc[s], planes), 256);
    }
    p.total = off;
    return WSI_OK;
}

size_t wsi_trunk_workspace_bytes(int n, int h, int w, int planes) {
    TrunkPlan p;
    return trunk_plan(n, h, w, planes, p) ? 0 : p.total;
}

int wsi_trunk_workspace_init(void* workspace, int n, int h, int w, int planes, void* stream) {
    TrunkPlan p;
    if (!workspace || trunk_plan(n, h, w, planes, p)) return WSI_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g_ws_mutex);
        g_ws_layout[workspace] = WsTag{-1, p.total};
    }
    return hipMemsetAsync((char*)workspace + p.buf[0][0], 0, p.total - p.buf[0][0], (hipStream_t)stream) == hipSuccess
               ? WSI_OK
               : WSI_EFAULT;
}

// Runs stem + residual stages; stops after stage `stop_after` (0 = pool, 1..8 = blocks, >= 8 all).
// Returns the workspace offset / geometry of the last tensor produced.
// `p` is the plan of the workspace, made for `cap` >= n images: buffer offsets and the distance between phase images
// come from the plan, so one workspace serves every batch size up to cap (image i sits at the same place whatever n is;
// what images >= n still hold from an earlier, larger batch is never read: the zero row / column that close image
// n-1 belong to its own block).
static int trunk_run(const wsi_trunk_weights* wt, const float* in_f32, const uint8_t* slide, long long pitch, int slide_h,
                     int slide_w, const int* tile_xy, const float* lut, int n, int cap, int h, int w, void* workspace,
                     int stop_after, hipStream_t st, const TrunkPlan& p, size_t& last_off, int& last_stage,
                     bool allow_split = true, size_t* stage_off = nullptr, char* x0_out = nullptr) {
    char* ws = (char*)workspace;
    const int planes = wt->planes;
    int rc = WSI_OK;
    // kind: 1 = 3x3 stride-1 of layers 2-4 (wide kernel), 5 = 3x3 stride-1 of the 64-channel layer 1 (slab3 kernel),
    // 2 = 3x3 stride-2 (+ fused downsample), 3 = 1x1 downsample, 4 = stem+maxpool;
    // flops = 2*M*N*K over real output pixels (padding taps counted)
#define PROF_CONV(kind, NN, HO, WO, CI, CO, KK, call)                                                \
    do {                                                                                            \
        const int pi_ = prof_open(st, kind, 2.0 * (NN) * (HO) * (WO) * (double)(CO) * (CI) * (KK));  \
        rc = (call);                                                                                \
        prof_close(st, pi_);                                                                        \
        if (rc) return rc;                                                                          \
    } while (0)
    const size_t bpc = planes == 1 ? PFmt<1>::BPC : PFmt<2>::BPC;     // bytes per channel: 2 (speed) or 4 (parity, mx)
    // ---- stem + maxpool + layer1 run in sub-batches so that the 4 MB/patch fp32 stem scratch and
    //      the 1 MB/patch layer-1 tensors stay resident in the 256 MiB Infinity Cache; the deeper
    //      (small-map) stages run on the whole batch to fill the chip.
    const int cs = g_chunk_stem > 0 ? g_chunk_stem : n, c1 = g_chunk_l1 > 0 ? g_chunk_l1 : n;
    const int H1 = p.sh[0], W1 = p.sw[0];
    const int do_l1 = stop_after != 0;
    // stage s writes its output phase-split when the next stage's entry can read it with the wide stride-2 kernel:
    // full runs only (taps unpack ordinary PF), split precision, next output maps <= 33 wide, whole-batch stages
    auto can_split = [&](int s) { return allow_split && g_s2_split && g_s2_slab && stop_after >= 8 && planes >= 2 && s < 3 && p.sw[s + 1] <= 33; };
    const bool split0 = can_split(0);                  // (a layer-1 sub-batch writes its images' slice of each phase image)
    // r03: a full mode-3 run keeps stem output and layer-1 tensors in 96-byte lines (layer 1 is HBM-bound: 25 % fewer bytes);
    // the last layer-1 conv writes the ordinary (or phase-split) 128-byte form every other kernel reads
    // (only with the phase-split hand-over to layer 2: an ordinary 128-byte output would land in a buffer that held 96-byte lines)
    const bool l96 = g_l1_lines96 && planes == 3 && split0;
    // the tag is recorded for EVERY planes value (r03 advisor finding: a planes 1 / 2 run used to leave a stale '96-byte lines' tag,
    // and a later mx run on the same workspace then skipped the zero-fill): tag = 2 * planes + (96-byte lines)
    if (const int dirty = ws_layout_switch(workspace, 2 * planes + (l96 ? 1 : 0), p.total)) {
        if (dirty < 0) return WSI_EINVAL;              // planned for a smaller batch / another planes value than this call needs
        const size_t nbytes = dirty == 2 ? p.total - p.buf[0][0] : p.buf[0][3] - p.buf[0][0];   // the three rotating stage-0 buffers (or everything)
        if (hipMemsetAsync(ws + p.buf[0][0], 0, nbytes, st) != hipSuccess) return WSI_EFAULT;
    }
    // byte offset of image n0 inside a PF buffer of stage s
    // (96-byte lines are line-planar: an image's offset inside every line plane; the planes lie plane96 bytes apart, a distance fixed by
    //  the plan's capacity, so sub-batches and smaller batches address the same places)
    auto img_off = [&](int s, int n0) { return (size_t)n0 * (p.sh[s] + 1) * (p.sw[s] + 1) * (s == 0 && l96 ? (size_t)96 : (size_t)p.sc[s] * bpc); };
    const long long plane96 = l96 ? (long long)pf_alloc_pixels(cap, p.sh[0], p.sw[0]) * 96 : 0;
    // ... and inside one phase image of stage 0's phase-split output (a PF tensor of stage 1's map size, 64 channels)
    auto split_off = [&](int n0) { return (size_t)n0 * (p.sh[1] + 1) * (p.sw[1] + 1) * p.sc[0] * bpc; };

    int l1_out = 0;                                    // buffer index holding layer1's output
    for (int n1 = 0; n1 < n; n1 += c1) {
        const int nn1 = n - n1 < c1 ? n - n1 : c1;
        for (int n0 = n1; n0 < n1 + nn1; n0 += cs) {
            const int nn = n1 + nn1 - n0 < cs ? n1 + nn1 - n0 : cs;
            const int pi_ = prof_open(st, 4, 2.0 * nn * (h / 2) * (w / 2) * 64.0 * 147.0);
            rc = stem_run(in_f32 ? in_f32 + (size_t)n0 * 3 * h * w : nullptr, slide, pitch, slide_h,
                          slide_w, tile_xy ? tile_xy + 2 * n0 : nullptr, lut, wt->stem_w, wt->stem_b,
                          wt->stem_w_u8, wt->stem_b_u8, wt->norm,
                          nn, h, w, (float*)(ws + p.stem_scratch), ws + p.buf[0][0] + img_off(0, n0),
                          planes, st, l96 ? 1 : 0, plane96,
                          x0_out ? x0_out + (size_t)n0 * (h / 2 + 1) * (w / 2 + 1) * 64 * bpc : nullptr);     // (U-Net: the conv map before the pool)
            prof_close(st, pi_);
            if (rc) return rc;
        }
        if (!do_l1) continue;
        int cur = 0;
        for (int b = 0; b < 2 && (stop_after >= 8 || b < stop_after); ++b) {
            const int m = (cur + 1) % 3, o = (cur + 2) % 3;
            char *x = ws + p.buf[0][cur] + img_off(0, n1), *mid = ws + p.buf[0][m] + img_off(0, n1),
                 *out = ws + p.buf[0][o] + img_off(0, n1);
            const int f_in = l96 ? CONV_IN96 : 0, f_res = l96 ? CONV_RESID96 : 0;
            PROF_CONV(5, nn1, H1, W1, 64, 64, 9, conv_common(x, mid, nullptr, wt->conv_w[2 * b], wt->conv_b[2 * b], nn1,
                                                              H1, W1, 64, 64, 1, 3, 1, planes, st, -1, 0, 0, nullptr, 0, nullptr, nullptr,
                                                              f_in | (l96 ? CONV_OUT96 : 0), nullptr, 0, plane96));
            if (b == 1 && split0) {                    // layer1's output feeds only the stride-2 entry of layer2
                PROF_CONV(5, nn1, H1, W1, 64, 64, 9, conv_common(mid, ws + p.buf[0][3] + split_off(n1), x, wt->conv_w[3], wt->conv_b[3], nn1, H1, W1, 64,
                                                                  64, 1, 3, 1, planes, st, -1, 1, pf_alloc_pixels(cap, H1 / 2, W1 / 2), nullptr, 0, nullptr,
                                                                  nullptr, f_in | f_res, nullptr, 0, plane96));
            } else {                                   // (the stage's last conv writes 128-byte lines: layer 2, taps and skips read those)
                PROF_CONV(5, nn1, H1, W1, 64, 64, 9, conv_common(mid, out, x, wt->conv_w[2 * b + 1], wt->conv_b[2 * b + 1],
                                                                  nn1, H1, W1, 64, 64, 1, 3, 1, planes, st, -1, 0, 0, nullptr, 0, nullptr, nullptr,
                                                                  f_in | f_res | (l96 && b == 0 ? CONV_OUT96 : 0), nullptr, 0, plane96));
            }
            cur = o;
        }
        l1_out = cur;
    }
    last_off = p.buf[0][l1_out]; last_stage = 0;
    if (stage_off) stage_off[0] = p.buf[0][l1_out];           // (ordinary PF only when allow_split is off)
    if (stop_after >= 0 && stop_after <= 2) return WSI_OK;

    int cur = l1_out;
    const void* x = split0 ? ws + p.buf[0][3] : ws + p.buf[0][cur];
    bool x_split = split0;
    int block = 2;
    for (int s = 1; s < 4; ++s) {
        const int H = p.sh[s], W = p.sw[s], C = p.sc[s];
        for (int b = 0; b < 2; ++b) {
            const int wi = s * 4 + b * 2;
            void *mid, *out;
            const void* resid;
            // r03, mode 3: the 1x1 downsample of a strided block is computed INSIDE the block's second conv as an extra K segment
            // over phase 00 of the block input (ConvArgs.in2): the stride-2 kernel drops its second accumulator set and half its
            // tile epilogues, the downsample tensor is neither written nor read back as a residual
            const bool fold = b == 0 && x_split && planes == 3 && g_ds_fold && g_s2_slab && C % 128 == 0;
            const void* fold_in2 = fold ? x : nullptr;
            if (b == 0) {                              // strided block with 1x1 downsample branch
                mid = ws + p.buf[s][1];
                void* ds = fold ? nullptr : ws + p.buf[s][2];
                out = ws + p.buf[s][0];
                if (g_s2_slab) {
                    const int pi_ = prof_open(st, 2, 2.0 * n * H * W * (double)C * (C / 2) * (fold ? 9 : 10));
                    rc = x_split ? s2_split_common(x, mid, ds, wt->conv_w[wi], wt->conv_b[wi], wt->down_w[s - 1],
                                                   wt->down_b[s - 1], n, 2 * H, 2 * W, C / 2, C, planes, st, pf_alloc_pixels(cap, H, W))
                                 : wsi_conv3x3s2_ds_fused(x, mid, ds, wt->conv_w[wi], wt->conv_b[wi], wt->down_w[s - 1], wt->down_b[s - 1], n,
                                                          2 * H, 2 * W, C / 2, C, planes, st);
                    prof_close(st, pi_);
                    if (rc) return rc;
                } else {
                    PROF_CONV(2, n, H, W, C / 2, C, 9, wsi_conv3x3_bn_act(x, mid, nullptr, wt->conv_w[wi], wt->conv_b[wi], n, 2 * H,
                                                                     2 * W, C / 2, C, 2, 1, planes, st));
    PROF_CONV(3, n, H, W, C / 2, C, 1, wsi_conv1x1_bn(x, ds, wt->down_w[s - 1], wt->down_b[s - 1], n, 2 * H, 2 * W,
                                                                 C / 2, C, 2, planes, st));
                }
                resid = ds;
                cur = 0;
                last_off = p.buf[s][0];
            } else {
                const int m = (cur + 1) % 3, o = (cur + 2) % 3;
                mid = ws + p.buf[s][m];
                out = ws + p.buf[s][o];
                PROF_CONV(1, n, H, W, C, C, 9, wsi_conv3x3_bn_act(x, mid, nullptr, wt->conv_w[wi], wt->conv_b[wi], n, H, W, C, C,
                                                                 1, 1, planes, st));
                resid = x;
                cur = o;
                last_off = p.buf[s][o];
            }
            if (fold_in2) {                            // second conv of a strided block with the downsample folded in (never the stage's last conv)
                const int pi_ = prof_open(st, 1, 2.0 * n * H * W * (double)C * (C * 9 + C / 2));
                rc = conv_common(mid, out, nullptr, wt->conv_w[wi + 1], wt->conv_b[wi + 1], n, H, W, C, C, 1, 3, 1, planes, st, -1, 0, 0,
                                 fold_in2, C / 2, wt->down_w[s - 1], wt->down_b[s - 1]);
                prof_close(st, pi_);
                if (rc) return rc;
                x_split = false;
            } else if (b == 1 && can_split(s)) {       // the stage's output feeds only the next stage's stride-2 entry
                out = ws + p.buf[s][3];
                PROF_CONV(1, n, H, W, C, C, 9, conv_common(mid, out, resid, wt->conv_w[wi + 1], wt->conv_b[wi + 1], n, H, W, C, C, 1, 3, 1,
                                                            planes, st, -1, 1, pf_alloc_pixels(cap, H / 2, W / 2)));
                x_split = true;
            } else {
                PROF_CONV(1, n, H, W, C, C, 9, wsi_conv3x3_bn_act(mid, out, resid, wt->conv_w[wi + 1], wt->conv_b[wi + 1], n, H, W, C,
                                                                 C, 1, 1, planes, st));
                x_split = false;
            }
            x = out;
            if (b == 1 && stage_off) stage_off[s] = (size_t)((char*)out - ws);
            ++block;
            last_stage = s;
            if (block == stop_after) return WSI_OK;
        }
    }
    return WSI_OK;
}

int wsi_trunk_forward(const wsi_trunk_weights* wt, const float* in_f32, const uint8_t* slide,
                      long long slide_pitch_bytes, int slide_h, int slide_w, const int* tile_xy, const float* lut,
                      int n, int h, int w, void* workspace, int workspace_n, float* feat_out, float* logits_out,
                      float* fmap_out, void* stream) {
    TrunkPlan p;
    const int cap = workspace_n > 0 ? workspace_n : n;
    if (!wt || !workspace || n <= 0 || cap < n || trunk_plan(cap, h, w, wt->planes, p)) return WSI_EINVAL;
    if (logits_out && (!wt->head_w || !wt->head_b || wt->head_k <= 0)) return WSI_EINVAL;
    size_t off; int stage;
    int rc = trunk_run(wt, in_f32, slide, slide_pitch_bytes, slide_h, slide_w, tile_xy, lut, n, cap, h, w, workspace, 8,
                       (hipStream_t)stream, p, off, stage);
    if (rc) return rc;
    const char* last = (const char*)workspace + off;
    if (feat_out || logits_out) {
        rc = wsi_avgpool_fc(last, n, p.sh[3], p.sw[3], 512, wt->head_w, wt->head_b, wt->head_k, feat_out, logits_out,
                            wt->planes, stream);
        if (rc) return rc;
    }
    if (fmap_out) rc = wsi_pf_unpack(last, fmap_out, n, 512, p.sh[3], p.sw[3], wt->planes, stream);
    return rc;
}

int wsi_trunk_forward_tap(const wsi_trunk_weights* wt, const float* in_f32, const uint8_t* slide,
                          long long slide_pitch_bytes, int slide_h, int slide_w, const int* tile_xy, const float* lut,
                          int n, int h, int w, void* workspace, int workspace_n, int stop_after, float* tap_out_nchw,
                          void* stream) {
    TrunkPlan p;
    const int cap = workspace_n > 0 ? workspace_n : n;
    if (!wt || !workspace || !tap_out_nchw || stop_after < 0 || stop_after > 8 || n <= 0 || cap < n ||
        trunk_plan(cap, h, w, wt->planes, p))
        return WSI_EINVAL;
    size_t off; int stage;
    int rc = trunk_run(wt, in_f32, slide, slide_pitch_bytes, slide_h, slide_w, tile_xy, lut, n, cap, h, w, workspace,
                       stop_after, (hipStream_t)stream, p, off, stage);
    if (rc) return rc;
    return wsi_pf_unpack((const char*)workspace + off, tap_out_nchw, n, p.sc[stage], p.sh[stage], p.sw[stage], wt->planes,
                         stream);
}


// ------------------------------------------------------------------------------------ U-Net (dense 'seg' path)
// smp-style decoder on the ResNet-18 trunk: five blocks of [nearest x2 upsample, concat skip, 2 x (3x3 conv + BN + ReLU)]
// at channels 256/128/64/32/16 (stored padded to whole 128-byte lines - 32 channels in the split-precision modes, 64 in
// speed mode; the padding channels carry zero weights), 1x1 head.
static const int kUnetSkipC[5] = {256, 128, 64, 64, 0};      // encoder maps x3, x2, x1, x0 (and none for the last block)
int g_unet_fuse_up = 1;                                  // A/B: wsi_conv_set_mode +65536 off
int g_unet_tail = 1;                                     // A/B: wsi_conv_set_mode +2097152 off
int g_unet_x0_fused = 1;                                 // A/B: wsi_conv_set_mode +8388608 off
struct UnetPlan {
    size_t x0, cat[5], mid[5], out[5], total;
    int r_h[5], r_w[5], cx[5];                               // resolution of block L; channels of its upsampled input
};
static int unet_plan(const wsi_unet_decoder_weights* dw, int n, int h, int w, int planes, UnetPlan& u) {
    if (!dw || n <= 0 || h % 32 || w % 32 || planes < 1 || planes > 3) return WSI_EINVAL;
    size_t off = 0;
    u.x0 = off; off += align_up(wsi_pf_bytes(n, h / 2, w / 2, 64, planes), 256);
    int cprev = 512;
    for (int L = 0; L < 5; ++L) {
        u.r_h[L] = (h / 16) << L; u.r_w[L] = (w / 16) << L; u.cx[L] = cprev;
        const int cin = cprev + kUnetSkipC[L], cout = dw->cout[2 * L];
        if (dw->cin[2 * L] != cin || dw->cin[2 * L + 1] != cout || dw->cout[2 * L + 1] != cout || cout % (planes == 1 ? 64 : 32) || cout <= 0) return WSI_EINVAL;
        u.cat[L] = off; off += align_up(wsi_pf_bytes(n, u.r_h[L], u.r_w[L], cin, planes), 256);
        u.mid[L] = off; off += align_up(wsi_pf_bytes(n, u.r_h[L], u.r_w[L], cout, planes), 256);
        u.out[L] = off; off += align_up(wsi_pf_bytes(n, u.r_h[L], u.r_w[L], cout, planes), 256);
        cprev = cout;
    }
    if (dw->head_cin <= 0 || dw->head_cin > cprev || dw->classes <= 0) return WSI_EINVAL;
    u.total = off;
    return WSI_OK;
}

size_t wsi_unet_workspace_bytes(const wsi_unet_decoder_weights* dw, int n, int h, int w, int planes) {
    UnetPlan u;
    const size_t t = wsi_trunk_workspace_bytes(n, h, w, planes);
    return (!t || unet_plan(dw, n, h, w, planes, u)) ? 0 : align_up(t, 256) + u.total;
}

int wsi_unet_workspace_init(const wsi_unet_decoder_weights* dw, void* workspace, int n, int h, int w, int planes, void* stream) {
    UnetPlan u;
    if (!workspace || unet_plan(dw, n, h, w, planes, u)) return WSI_EINVAL;
    int rc = wsi_trunk_workspace_init(workspace, n, h, w, planes, stream);
    if (rc) return rc;
    char* base = (char*)workspace + align_up(wsi_trunk_workspace_bytes(n, h, w, planes), 256);
    return hipMemsetAsync(base, 0, u.total, (hipStream_t)stream) == hipSuccess ? WSI_OK : WSI_EFAULT;
}

// the decoder on five PF encoder maps (x4 deepest ... x0 = stem output at half resolution), `dec` = decoder part of the workspace
static int unet_decoder_run(const wsi_unet_decoder_weights* dw, const UnetPlan& u, const void* const enc[5], int n, int planes, char* dec,
                            float* logits_out, hipStream_t st) {
    const void* x = enc[0];
    int rc = WSI_OK;
    // wsi_prof kinds of the decoder (bench.py --workload seg): 6 = decoder 3x3 conv (algorithmic FLOPs over REAL channels are the
    // caller's business: the record carries 2 * N * H * W * cin_stored * cout_stored * 9), 7 = upsample + concat glue, 8 = 1x1 head
    // r05: parity mode runs the last block and the head as ONE kernel (tail.hip) when the caller prepacked its weights
    // (wsi_unet_tail_prepack -> dw->tail_w) and the map is at most 256 wide; A/B: wsi_conv_set_mode +2097152 off
    const bool tail = planes == 2 && dw->tail_w && g_unet_tail && u.cx[4] == 32 && kUnetSkipC[4] == 0 && dw->classes <= 4 &&
                      u.r_w[3] % 32 == 0 && u.r_w[3] <= 128 &&
                      (size_t)pf_alloc_pixels(n, u.r_h[3], u.r_w[3]) * 128 <= (size_t)0x7fffffff;      // (32-bit buffer offsets into x4: ~1000 tiles of 256 x 256)
    for (int L = 0; L < (tail ? 4 : 5) && !rc; ++L) {
        const int H = u.r_h[L], W = u.r_w[L], cin = dw->cin[2 * L], cout = dw->cout[2 * L];
        // r04: the block's first conv reads the low-resolution tensor and the skip directly (ConvArgs.in_up: nearest x2 upsample +
        // concat as source addresses of its slab DMA) where the shape's kernel is the slab3 kernel; otherwise (EINVAL) the
        // upsample_concat pass writes the concatenated tensor first, as in r02-r03
        int pi = prof_open(st, 6, 2.0 * n * H * W * (double)cin * cout * 9);
        rc = g_unet_fuse_up ? wsi_conv3x3_up_concat_bn_act(x, L < 4 ? enc[L + 1] : nullptr, dec + u.mid[L], dw->conv_w[2 * L], dw->conv_b[2 * L], n, H, W,
                                                           u.cx[L], kUnetSkipC[L], cout, 1, planes, st)
                            : WSI_EINVAL;
        prof_close(st, pi);
        if (rc == WSI_EINVAL) {
            if (pi >= 0) g_prof.kind[pi] = 9;             // (a refused launch: its empty record is not a decoder conv)
            pi = prof_open(st, 7, 0.0);
            rc = wsi_upsample_concat_dispatch(x, L < 4 ? enc[L + 1] : nullptr, dec + u.cat[L], n, H / 2, W / 2, u.cx[L], kUnetSkipC[L], planes, st);
            prof_close(st, pi);
            pi = prof_open(st, 6, 2.0 * n * H * W * (double)cin * cout * 9);
            if (!rc) rc = conv_common(dec + u.cat[L], dec + u.mid[L], nullptr, dw->conv_w[2 * L], dw->conv_b[2 * L], n, H, W, cin, cout, 1, 3, 1, planes, st);
            prof_close(st, pi);
        }
        pi = prof_open(st, 6, 2.0 * n * H * W * (double)cout * cout * 9);
        if (!rc) rc = conv_common(dec + u.mid[L], dec + u.out[L], nullptr, dw->conv_w[2 * L + 1], dw->conv_b[2 * L + 1], n, H, W, cout, cout, 1, 3, 1, planes, st);
        prof_close(st, pi);
        x = dec + u.out[L];
    }
    if (tail) {
        const int pt = prof_open(st, 10, 2.0 * n * u.r_h[4] * u.r_w[4] * (9.0 * (32.0 * 16.0 + 16.0 * 16.0) + 16.0 * dw->classes));    // kind 10: the reference formulation's FLOPs over REAL channels
        if (!rc) rc = wsi_unet_tail_dispatch(x, dw->tail_w, n, u.r_h[3], u.r_w[3], dw->classes, logits_out, st);
        prof_close(st, pt);
        return rc;
    }
    const int pi = prof_open(st, 8, 2.0 * n * u.r_h[4] * u.r_w[4] * (double)dw->head_cin * dw->classes);
    if (!rc) rc = wsi_unet_head_dispatch(x, n, u.r_h[4], u.r_w[4], dw->cout[9], dw->head_w, dw->head_b, dw->head_cin, dw->classes, logits_out, planes, st);
    prof_close(st, pi);
    return rc;
}

int wsi_unet_forward(const wsi_trunk_weights* wt, const wsi_unet_decoder_weights* dw, const float* in_f32, const uint8_t* slide,
                     long long slide_pitch_bytes, int slide_h, int slide_w, const int* tile_xy, const float* lut, int n, int h,
                     int w, void* workspace, int workspace_n, float* logits_out, float* enc_out[5], void* stream) {
    TrunkPlan p;
    UnetPlan u;
    const int cap = workspace_n > 0 ? workspace_n : n;
    if (!wt || !dw || !workspace || n <= 0 || cap < n || (!logits_out && !enc_out) || h % 32 || w % 32) return WSI_EINVAL;
    if (trunk_plan(cap, h, w, wt->planes, p) || unet_plan(dw, cap, h, w, wt->planes, u)) return WSI_EINVAL;
    if (!in_f32 && (!slide || !tile_xy || !lut)) return WSI_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    char* dec = ws + align_up(p.total, 256);
    const int planes = wt->planes;
    // encoder: the trunk with every stage output kept as an ordinary PF tensor (no phase-split hand-over) ...
    size_t off, stage_off[4];
    int stage;
    // r05: on the product path (u8 slide, parity mode) the fused stem + pool kernel stores x0 = relu(bn1(conv1(x))) itself - the conv
    // values it pools anyway, exact integer arithmetic - instead of a second, unfused stem conv (A/B: wsi_conv_set_mode +8388608 off)
    const bool x0_fused = g_unet_x0_fused && !in_f32 && planes == 2 && wt->stem_w_u8 && wt->stem_b_u8 && wt->norm && g_stem_u8x && g_stem_fused &&
                          g_stem_shared_weights;
    int rc = trunk_run(wt, in_f32, slide, slide_pitch_bytes, slide_h, slide_w, tile_xy, lut, n, cap, h, w, workspace, 8, st, p, off, stage,
                       false, stage_off, x0_fused ? dec + u.x0 : nullptr);
    if (rc) return rc;
    // ... plus x0 = relu(bn1(conv1(x))) before the max pool, which the fused stem kernel never writes: the unfused stem
    // conv (bf16 hi/lo arithmetic) into the fp32 scratch, then PF lines
    StemArgs a;
    a.mode = in_f32 ? 0 : 1;
    a.in_f32 = in_f32; a.slide = slide; a.slide_pitch = slide_pitch_bytes; a.SH = slide_h; a.SW = slide_w;
    a.origins = tile_xy; a.lut = lut; a.wpk = wt->stem_w; a.bias = wt->stem_b; a.out = (float*)(ws + p.stem_scratch);
    a.N = n; a.H = h; a.W = w; a.wpk_u8 = nullptr; a.bias_u8 = nullptr;
    const int pi = prof_open(st, 7, 0.0);                    // (glue: the unfused stem conv for the half-resolution skip x0)
    if (x0_fused) {
    } else if (g_unet_fuse_up) {                                    // r04: the conv kernel writes PF lines itself (was: f32 scratch + nhwc_to_pf pass)
        a.out_pf = dec + u.x0; a.out_planes = planes;
        rc = wsi_stem_dispatch(a, planes == 1 ? 1 : 2, st);
    } else {
        rc = wsi_stem_dispatch(a, planes == 1 ? 1 : 2, st);
        if (!rc) rc = wsi_nhwc_to_pf_dispatch(a.out, dec + u.x0, n, h / 2, w / 2, 64, planes, st);
    }
    prof_close(st, pi);
    if (rc) return rc;
    const void* enc[5] = {ws + stage_off[3], ws + stage_off[2], ws + stage_off[1], ws + stage_off[0], dec + u.x0};
    if (enc_out) {                                           // the `model.encoder(x)` surface: five fp32 NCHW maps, deepest first
        const int ec[5] = {512, 256, 128, 64, 64};
        for (int i = 0; i < 5 && !rc; ++i)
            if (enc_out[i]) rc = wsi_pf_unpack(enc[i], enc_out[i], n, ec[i], i < 4 ? h >> (5 - i) : h / 2, i < 4 ? w >> (5 - i) : w / 2, planes, stream);
        if (rc) return rc;
    }
    return logits_out ? unet_decoder_run(dw, u, enc, n, planes, dec, logits_out, st) : WSI_OK;
}

int wsi_resize_nearest_f32(const float* src, long long planes_n, int hs, int ws, float* dst, int hd, int wd, void* stream) {
    if (!src || !dst || src == dst) return WSI_EINVAL;
    return wsi_resize_nearest_dispatch(src, planes_n, hs, ws, dst, hd, wd, (hipStream_t)stream);
}

// `model.decoder(encoding)` with caller-held fp32 NCHW maps (deepest first): pack, then the same decoder launches
int wsi_unet_decoder(const wsi_unet_decoder_weights* dw, const float* const enc_nchw[5], int n, int h, int w, int planes, void* workspace,
                     int workspace_n, float* logits_out, void* stream) {
    TrunkPlan p;
    UnetPlan u;
    const int cap = workspace_n > 0 ? workspace_n : n;
    if (!dw || !enc_nchw || !workspace || !logits_out || n <= 0 || cap < n) return WSI_EINVAL;
    if (trunk_plan(cap, h, w, planes, p) || unet_plan(dw, cap, h, w, planes, u)) return WSI_EINVAL;
    char* ws = (char*)workspace;
    char* dec = ws + align_up(p.total, 256);
    // encoder maps are packed into the trunk part of the workspace (stage buffers 0 of stages 3..0) and x0
    const int ec[5] = {512, 256, 128, 64, 64};
    const void* enc[5];
    int rc = WSI_OK;
    for (int i = 0; i < 5 && !rc; ++i) {
        if (!enc_nchw[i]) return WSI_EINVAL;
        char* dst = i < 4 ? ws + p.buf[3 - i][0] : dec + u.x0;
        rc = wsi_pf_pack(enc_nchw[i], dst, n, ec[i], i < 4 ? h >> (5 - i) : h / 2, i < 4 ? w >> (5 - i) : w / 2, planes, stream);
        enc[i] = dst;
    }
    return rc ? rc : unet_decoder_run(dw, u, enc, n, planes, dec, logits_out, (hipStream_t)stream);
}

}  // extern "C"
