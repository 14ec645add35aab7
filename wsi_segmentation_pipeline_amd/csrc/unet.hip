// Glue kernels of the U-Net decoder: what sits between the 3x3 conv + BN + ReLU launches of the
// segmentation_models_pytorch `UnetDecoder` that the reference drives in predict_wsis / predict_tumorbed(mode='seg')
// (/root/reference/utils/eval.py:51,196-200; /root/reference/eval_tumorbed.py:21-28; the package itself is third-party,
// absent and un-pinned: the architecture is restated from its published 0.0.x source - parity unpinned).
//   upsample_concat : F.interpolate(x, scale_factor=2, mode='nearest') + torch.cat([x, skip], dim=1), written straight in
//                     the padded-flat line format the conv kernels read (whole 128-byte lines are copied: no arithmetic)
//   nhwc_to_pf      : the stem's pre-pool activation (fp32 NHWC scratch of the unfused stem kernel) -> PF lines: the
//                     64-channel, half-resolution skip connection x0 that the fused stem + maxpool kernel never materialises
//   unet_head       : final_conv (1x1, <= 64 input channels, K classes) from PF lines straight to fp32 NCHW logits
// All three are plain HBM-bound copies / dot products.
#include "pf_lines.h"

// out (N, 2h, 2w, Cx + Cs) <- x (N, h, w, Cx) upsampled x2 | skip (N, 2h, 2w, Cs); one 16-byte piece per thread
__global__ __launch_bounds__(256) void upsample_concat_kernel(const char* x, const char* skip, char* out, PFGeom gx, PFGeom go,
                                                              int px_bytes, int ps_bytes) {
    const int po_bytes = px_bytes + ps_bytes;                            // bytes per pixel record
    const int pieces = po_bytes >> 4;
    const long long total = (long long)go.N * go.H * go.W * pieces;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % pieces);
        long long p = i / pieces;
        const int xx = (int)(p % go.W); p /= go.W;
        const int yy = (int)(p % go.H);
        const int n = (int)(p / go.H);
        const size_t qo = (size_t)go.G + (size_t)n * go.S + (size_t)yy * go.P + xx;
        uint4 v;
        if (k * 16 < px_bytes) {
            const size_t qx = (size_t)gx.G + (size_t)n * gx.S + (size_t)(yy >> 1) * gx.P + (xx >> 1);
            v = *(const uint4*)(x + qx * px_bytes + (size_t)k * 16);
        } else {
            v = *(const uint4*)(skip + qo * ps_bytes + (size_t)k * 16 - px_bytes);
        }
        *(uint4*)(out + qo * po_bytes + (size_t)k * 16) = v;
    }
}

// in: fp32 [N][H][W][C] (NHWC) -> PF (N, H, W, C); one line per thread
template <int PLANES>
__global__ __launch_bounds__(256) void nhwc_to_pf_kernel(const float* in, char* out, PFGeom g) {
    constexpr int CPL = PFmt<PLANES>::CPL;
    const int lines = g.C / CPL;
    const long long total = (long long)g.N * g.H * g.W * lines;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int l = (int)(i % lines);
        long long p = i / lines;
        const int x = (int)(p % g.W); p /= g.W;
        const int y = (int)(p % g.H);
        const int n = (int)(p / g.H);
        float v[CPL];
        const float* src = in + (((size_t)n * g.H + y) * g.W + x) * g.C + (size_t)l * CPL;
#pragma unroll
        for (int c = 0; c < CPL; c += 4) {
            const float4 f = *(const float4*)(src + c);
            v[c] = f.x; v[c + 1] = f.y; v[c + 2] = f.z; v[c + 3] = f.w;
        }
        pf_line_encode<PLANES>(v, out + ((size_t)g.G + (size_t)n * g.S + (size_t)y * g.P + x) * ((size_t)g.C * PFmt<PLANES>::BPC) + (size_t)l * 128);
    }
}

// logits[n][k][y][x] = b[k] + sum_c w[k][c] * act(n, y, x, c), c < cin <= 64 (channels beyond cin are padding)
// One thread per pixel.  The pixel's 128-byte lines are fetched with 16-byte loads and decoded whole (line positions, not
// channels: the weights are permuted into line order once per workgroup, in LDS, zero for the padding channels); only the
// lines that hold real channels are read (smp's final conv has 16 inputs: one line of the two).  r02: the byte-wise
// per-channel form this replaces (64 values in scratch memory) took 2.68 ms per 128 tiles of 256x256 = 21 % of the seg path;
// this one < 0.4 ms (seg path 10.5 k -> 13.4 k tiles/s in mx).
template <int PLANES>
static __device__ __forceinline__ void head_line_decode(const char* line, float (&a)[PFmt<PLANES>::CPL]) {
    const uint4* q = (const uint4*)line;
    if constexpr (PLANES == 3) {
        mx_line_decode_all(line, a);
    } else if constexpr (PLANES == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x8 hi = __builtin_bit_cast(f16x8, q[j]), lo = __builtin_bit_cast(f16x8, q[4 + j]);   // fp16 pair
#pragma unroll
            for (int i = 0; i < 8; ++i) a[8 * j + i] = (float)hi[i] + (float)lo[i];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, q[j]);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[8 * j + i] = (float)v[i];
        }
    }
}
template <int PLANES, int KMAX>
__global__ __launch_bounds__(256) void unet_head_kernel(const char* in, PFGeom g, const float* w, const float* b, int cin, int K,
                                                        float* out) {
    constexpr int CPL = PFmt<PLANES>::CPL;
    __shared__ float wl[KMAX][64];                            // weights in LINE order: wl[k][line * CPL + position]
    for (int i = threadIdx.x; i < KMAX * 64; i += 256) {
        const int k = i >> 6, lp = i & 63, line = lp / CPL, pos = lp % CPL;
        const int chan = line * CPL + (PLANES == 3 ? mx_line_chan(pos) : pos);
        wl[k][lp] = (k < K && chan < cin) ? w[k * cin + chan] : 0.f;
    }
    __syncthreads();
    const int nlines = (cin + CPL - 1) / CPL;
    const long long total = (long long)g.N * g.H * g.W;
    const size_t pixstride = (size_t)g.C * PFmt<PLANES>::BPC;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % g.W);
        long long p = i / g.W;
        const int y = (int)(p % g.H);
        const int n = (int)(p / g.H);
        const char* px = in + ((size_t)g.G + (size_t)n * g.S + (size_t)y * g.P + x) * pixstride;
        float s[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s[k] = (b && k < K) ? b[k] : 0.f;
        for (int l = 0; l < nlines; ++l) {
            float a[CPL];
            head_line_decode<PLANES>(px + (size_t)l * 128, a);
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int c = 0; c < CPL; ++c) s[k] = fmaf(wl[k][l * CPL + c], a[c], s[k]);
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) out[(((size_t)n * K + k) * g.H + y) * g.W + x] = s[k];
    }
}

// F.interpolate(x, size=(Hd, Wd)) in its default mode 'nearest' (utils/eval.py:202-206): src index = floor(dst * in / out)
// evaluated like PyTorch does, in fp32: scale = (float)in / out, idx = min((int)floorf(dst * scale), in - 1)
__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* src, long long planes_n, int Hs, int Ws, float* dst, int Hd, int Wd) {
    const long long total = planes_n * Hd * Wd;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Wd);
        long long p = i / Wd;
        const int y = (int)(p % Hd);
        const long long pl = p / Hd;
        const int ys = min((int)floorf((float)y * sy), Hs - 1), xs = min((int)floorf((float)x * sx), Ws - 1);
        dst[i] = src[(pl * Hs + ys) * Ws + xs];
    }
}

static int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 32768 ? 32768 : (g < 1 ? 1 : g));
}

int wsi_upsample_concat_dispatch(const void* x, const void* skip, void* out, int n, int h, int w, int cx, int cs, int planes, hipStream_t st) {
    const int bpc = planes == 1 ? 2 : 4, cpl = planes == 1 ? 64 : 32;
    if (n <= 0 || h <= 0 || w <= 0 || cx <= 0 || cx % cpl || cs % cpl || cs < 0 || (cs && !skip)) return WSI_EINVAL;
    const PFGeom gx = pf_geom(n, h, w, cx), go = pf_geom(n, 2 * h, 2 * w, cx + cs);
    hipLaunchKernelGGL(upsample_concat_kernel, dim3(grid_for((long long)n * 4 * h * w * ((cx + cs) * bpc / 16))), dim3(256), 0, st,
                       (const char*)x, (const char*)skip, (char*)out, gx, go, cx * bpc, cs * bpc);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_nhwc_to_pf_dispatch(const float* in, void* out, int n, int h, int w, int c, int planes, hipStream_t st) {
    const int cpl = planes == 1 ? 64 : 32;
    if (n <= 0 || c % cpl || planes < 1 || planes > 3) return WSI_EINVAL;
    const PFGeom g = pf_geom(n, h, w, c);
    const int grid = grid_for((long long)n * h * w * (c / cpl));
    if (planes == 3) hipLaunchKernelGGL(nhwc_to_pf_kernel<3>, dim3(grid), dim3(256), 0, st, in, (char*)out, g);
    else if (planes == 2) hipLaunchKernelGGL(nhwc_to_pf_kernel<2>, dim3(grid), dim3(256), 0, st, in, (char*)out, g);
    else hipLaunchKernelGGL(nhwc_to_pf_kernel<1>, dim3(grid), dim3(256), 0, st, in, (char*)out, g);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_unet_head_dispatch(const void* in, int n, int h, int w, int c_pf, const float* wt, const float* b, int cin, int k, float* out,
                           int planes, hipStream_t st) {
    if (n <= 0 || cin <= 0 || cin > 64 || cin > c_pf || k <= 0 || k > 16 || planes < 1 || planes > 3) return WSI_EINVAL;
    const PFGeom g = pf_geom(n, h, w, c_pf);
    const int grid = grid_for((long long)n * h * w);
#define HEAD_LAUNCH(KM)                                                                                                              \
    do {                                                                                                                            \
        if (planes == 3) hipLaunchKernelGGL((unet_head_kernel<3, KM>), dim3(grid), dim3(256), 0, st, (const char*)in, g, wt, b, cin, k, out);      \
        else if (planes == 2) hipLaunchKernelGGL((unet_head_kernel<2, KM>), dim3(grid), dim3(256), 0, st, (const char*)in, g, wt, b, cin, k, out); \
        else hipLaunchKernelGGL((unet_head_kernel<1, KM>), dim3(grid), dim3(256), 0, st, (const char*)in, g, wt, b, cin, k, out);                  \
    } while (0)
    if (k <= 4) HEAD_LAUNCH(4);
    else if (k <= 8) HEAD_LAUNCH(8);
    else if (k <= 16) HEAD_LAUNCH(16);
    else return WSI_EINVAL;
#undef HEAD_LAUNCH
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_resize_nearest_dispatch(const float* src, long long planes_n, int hs, int ws, float* dst, int hd, int wd, hipStream_t st) {
    if (planes_n <= 0 || hs <= 0 || ws <= 0 || hd <= 0 || wd <= 0) return WSI_EINVAL;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for(planes_n * hd * wd)), dim3(256), 0, st, src, planes_n, hs, ws, dst, hd, wd);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}
