// Slide-side kernels of the sliding-window driver.
//   tile_gather   : u8 slide + tile corners -> normalised f32 NCHW batch
//                   (/root/reference/utils/dataset.py:171-185 + utils/preprocessing.py:209-212)
//   stitch_add    : pred[:, ty:ty+dy, tx:tx+dx] += tile_logits  in float64
//                   (/root/reference/utils/eval.py:213-215; also :58-60 with per-pixel tiles)
//   softmax_threshold_argmax : /root/reference/utils/preprocessing.py:156-172 (+ heat map of
//                   /root/reference/utils/eval.py:219-228)
#include "common.h"

__global__ __launch_bounds__(256) void tile_gather_kernel(const uint8_t* slide, long long pitch, int SH, int SW,
                                                          const int* origins, const float* lut, float* out, int N,
                                                          int ph, int pw) {
    const long long total = (long long)N * ph * pw;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % pw);
        long long p = i / pw;
        const int y = (int)(p % ph);
        const int n = (int)(p / ph);
        const int sx = origins[2 * n] + x, sy = origins[2 * n + 1] + y;
        uint8_t px[3] = {0, 0, 0};
        if (sx >= 0 && sx < SW && sy >= 0 && sy < SH) {
            const uint8_t* s = slide + (size_t)sy * pitch + (size_t)sx * 3;
            px[0] = s[0]; px[1] = s[1]; px[2] = s[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(((size_t)n * 3 + c) * ph + y) * pw + x] = lut[c * 256 + px[c]];
    }
}

// One thread per (tile, footprint pixel); every class of that pixel is added by the same thread.
// Sums of <= a few dozen fp32 logits are exact in float64, so the atomic order cannot change
// the result (the reference's shuffled DataLoader order does not either).
__global__ __launch_bounds__(256) void stitch_add_kernel(const float* logits, const int* txy, int T, int C, int dy, int dx,
                                                         double* pred, int MH, int MW) {
    const long long total = (long long)T * dy * dx;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int fx = (int)(i % dx);
        long long p = i / dx;
        const int fy = (int)(p % dy);
        const int t = (int)(p / dy);
        const int x = txy[2 * t] + fx, y = txy[2 * t + 1] + fy;
        if (x < 0 || x >= MW || y < 0 || y >= MH) continue;       // numpy slice clipping
        for (int c = 0; c < C; ++c)
            atomicAdd(pred + ((size_t)c * MH + y) * MW + x, (double)logits[(size_t)t * C + c]);
    }
}

// Dense form of the same accumulate (reference utils/eval.py:58-60, predict_wsis): every tile carries a
// (C, ph, pw) block.  One thread per (tile, pixel); numpy slice clipping at the map border.
__global__ __launch_bounds__(256) void stitch_add_dense_kernel(const float* tiles, const int* txy, int T, int C, int ph, int pw,
                                                               double* pred, int MH, int MW) {
    const long long total = (long long)T * ph * pw;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int fx = (int)(i % pw);
        long long p = i / pw;
        const int fy = (int)(p % ph);
        const int t = (int)(p / ph);
        const int x = txy[2 * t] + fx, y = txy[2 * t + 1] + fy;
        if (x < 0 || x >= MW || y < 0 || y >= MH) continue;
        for (int c = 0; c < C; ++c)
            atomicAdd(pred + ((size_t)c * MH + y) * MW + x, (double)tiles[(((size_t)t * C + c) * ph + fy) * pw + fx]);
    }
}

// Guard of the float64 stitch: sums of fp32 addends are EXACT in float64 (hence independent of the atomics' order, and of
// the reference's shuffled DataLoader order) while  exponent span of the addends + log2(addends per pixel) <= 29 bits.
// out2[0] = smallest, out2[1] = largest biased exponent among the nonzero finite values (255 / 0 when there are none).
__global__ __launch_bounds__(256) void exponent_span_kernel(const float* v, long long n, int* out2) {
    int lo = 255, hi = 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned e = (__float_as_uint(v[i]) >> 23) & 255u, m = __float_as_uint(v[i]) & 0x7fffffu;
        if (e == 255u || (e == 0u && m == 0u)) continue;               // inf / nan / zero
        lo = min(lo, (int)e); hi = max(hi, (int)e);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out2[0], lo); atomicMax(&out2[1], hi); }
}

#define WSI_MAX_CLASSES 16
__global__ __launch_bounds__(256) void softmax_threshold_argmax_kernel(const double* pred, int C, long long HW,
                                                                       const double* thresh, double* probs,
                                                                       uint8_t* classes, const uint8_t* mask,
                                                                       int heat_mode, uint8_t* heat) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        double v[WSI_MAX_CLASSES];
        double mx = pred[i];
        for (int c = 0; c < C; ++c) { v[c] = pred[(size_t)c * HW + i]; mx = v[c] > mx ? v[c] : mx; }
        double s = 0.0;
        for (int c = 0; c < C; ++c) { v[c] = exp(v[c] - mx); s += v[c]; }
        int best = 0;
        double bv = -1.0;
        for (int c = 0; c < C; ++c) {
            double p = v[c] / s;
            if (p < thresh[c]) p = 0.0;
            v[c] = p;
            if (probs) probs[(size_t)c * HW + i] = p;
            if (p > bv) { bv = p; best = c; }                      // first maximum, like np.argmax
        }
        if (classes) classes[i] = (uint8_t)best;
        if (heat) {
            double hv = heat_mode == 0 ? v[1] : v[2] + v[3];
            hv *= (double)(mask ? mask[i] : 1);
            heat[i] = (uint8_t)(int)(255.0 * hv);                  // np.uint8() truncation
        }
    }
}

// Region paint (/root/reference/scannet.py:154-155, slic.py:98-99: `pred_mask[metadata[tile_id]['foreground_indices']] = cls`
// in region order, so where candidate regions overlap the LAST one wins).  Entry e = flat pixel index idx[e] of region
// region_of[e] (regions numbered in paint order).  Pass 1 keeps, per pixel, the highest region number that covers it
// (atomicMax: order-independent), pass 2 writes that region's class: deterministic whatever the overlap.
__global__ __launch_bounds__(256) void paint_winner_kernel(const long long* idx, const int* region_of, long long n, int* winner, long long npix) {
    for (long long e = blockIdx.x * 256LL + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long p = idx[e];
        if (p >= 0 && p < npix) atomicMax(&winner[p], region_of[e] + 1);
    }
}
__global__ __launch_bounds__(256) void paint_write_kernel(const int* winner, const uint8_t* cls, long long npix, long long* label) {
    for (long long p = blockIdx.x * 256LL + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const int wn = winner[p];
        if (wn > 0) label[p] = cls[wn - 1];
    }
}

static int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

int wsi_tile_gather_dispatch(const uint8_t* slide, long long pitch, int SH, int SW, const int* origins, const float* lut,
                             float* out, int N, int ph, int pw, hipStream_t st) {
    if (N <= 0 || ph <= 0 || pw <= 0) return WSI_EINVAL;
    hipLaunchKernelGGL(tile_gather_kernel, dim3(grid_for((long long)N * ph * pw)), dim3(256), 0, st, slide, pitch, SH, SW,
                       origins, lut, out, N, ph, pw);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_stitch_add_dispatch(const float* logits, const int* txy, int T, int C, int dy, int dx, double* pred, int MH, int MW,
                            hipStream_t st) {
    if (T <= 0 || C <= 0 || dy <= 0 || dx <= 0) return WSI_EINVAL;
    hipLaunchKernelGGL(stitch_add_kernel, dim3(grid_for((long long)T * dy * dx)), dim3(256), 0, st, logits, txy, T, C, dy, dx,
                       pred, MH, MW);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_stitch_add_dense_dispatch(const float* tiles, const int* txy, int T, int C, int ph, int pw, double* pred, int MH,
                                  int MW, hipStream_t st) {
    if (T <= 0 || C <= 0 || ph <= 0 || pw <= 0) return WSI_EINVAL;
    hipLaunchKernelGGL(stitch_add_dense_kernel, dim3(grid_for((long long)T * ph * pw)), dim3(256), 0, st, tiles, txy, T, C, ph, pw,
                       pred, MH, MW);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_softmax_dispatch(const double* pred, int C, long long HW, const double* thresh, double* probs, uint8_t* classes,
                         const uint8_t* mask, int heat_mode, uint8_t* heat, hipStream_t st) {
    if (C <= 0 || C > WSI_MAX_CLASSES || HW <= 0) return WSI_EINVAL;
    if (heat && ((heat_mode == 0 && C < 2) || (heat_mode == 1 && C < 4))) return WSI_EINVAL;
    hipLaunchKernelGGL(softmax_threshold_argmax_kernel, dim3(grid_for(HW)), dim3(256), 0, st, pred, C, HW, thresh, probs,
                       classes, mask, heat_mode, heat);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_paint_dispatch(const long long* idx, const int* region_of, long long n, const uint8_t* cls, int* winner, long long* label,
                       long long npix, hipStream_t st) {
    if (n < 0 || npix <= 0) return WSI_EINVAL;
    if (hipMemsetAsync(winner, 0, (size_t)npix * sizeof(int), st) != hipSuccess) return WSI_EFAULT;
    if (n) hipLaunchKernelGGL(paint_winner_kernel, dim3(grid_for(n)), dim3(256), 0, st, idx, region_of, n, winner, npix);
    hipLaunchKernelGGL(paint_write_kernel, dim3(grid_for(npix)), dim3(256), 0, st, (const int*)winner, cls, npix, label);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_exponent_span_dispatch(const float* v, long long n, int* out2, hipStream_t st) {
    if (n <= 0) return WSI_EINVAL;
    // {255, 0} without a host buffer (a pageable H2D copy would block the host on the stream): zero both ints, then one 0xff byte
    if (hipMemsetAsync(out2, 0, 2 * sizeof(int), st) != hipSuccess || hipMemsetAsync(out2, 0xff, 1, st) != hipSuccess) return WSI_EFAULT;
    hipLaunchKernelGGL(exponent_span_kernel, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, v, n, out2);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}
