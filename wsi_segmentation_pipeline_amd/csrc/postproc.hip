// Tumour-bed post-process of the stitched map, on the device.
// Replaces, bit-exactly against oracle/postprocess_oracle.py (the published algorithms of the absent third-party calls):
//   /root/reference/utils/eval.py:66-71    cv2.resize(pred[c], level_dimensions[2])            resize_bilinear_f64_kernel
//   /root/reference/utils/eval.py:82       np.argmax(pred, 0)                                  argmax_kernel
//   /root/reference/utils/eval.py:90-91    (p >= 2), cv2.morphologyEx(MORPH_OPEN, ones(20,20)) threshold_kernel, morph_pass_kernel
//   /root/reference/utils/eval.py:92       skimage convex_hull_image (chull)                   hull_* kernels
//   /root/reference/utils/eval.py:94-95    mahotas bwperim, cv2.dilate(ones(20,20))            bwperim_kernel, morph_pass_kernel
//   /root/reference/utils/eval.py:104-121  tumour-bed IoU, accuracy / score sums               count kernels (exact integers)
//   /root/reference/paper_tools/overlay_tb_wsi.py:46-64  the same chain from a u8 heat map
//   /root/reference/contour_ordering.py:33-60            evenly_spaced_points_on_a_contour      esp_* kernels
// Everything here is HBM-bound byte / integer work on maps of a few MB (2500 x 2500 at level 2): plain coalesced
// kernels, no MFMA.  The convex hull runs on per-row extremes only (<= 3 doubled rows per image row), its two monotone
// chains are built by one lane each out of LDS, and the fill is one interval per row in exact 64-bit integers.
#include "common.h"
#include <limits.h>

// ------------------------------------------------------------------------------------------ resize / argmax / threshold
__global__ __launch_bounds__(256) void resize_bilinear_f64_kernel(const double* src, int C, int Hs, int Ws, double* dst, int Hd, int Wd) {
    const long long total = (long long)C * Hd * Wd;
    const double sy = (double)Hs / (double)Hd, sx = (double)Ws / (double)Wd;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % Wd);
        long long t = i / Wd;
        const int y = (int)(t % Hd), c = (int)(t / Hd);
        double fy = ((double)y + 0.5) * sy - 0.5, fx = ((double)x + 0.5) * sx - 0.5;
        double y0f = floor(fy), x0f = floor(fx);
        double wy = fy - y0f, wx = fx - x0f;
        long long y0 = (long long)y0f, x0 = (long long)x0f;
        if (y0 < 0) { y0 = 0; wy = 0.0; }
        if (x0 < 0) { x0 = 0; wx = 0.0; }
        long long y1 = y0 + 1 < Hs ? y0 + 1 : Hs - 1, x1 = x0 + 1 < Ws ? x0 + 1 : Ws - 1;
        if (y0 >= Hs - 1) { y0 = Hs - 1; y1 = Hs - 1; wy = 0.0; }
        if (x0 >= Ws - 1) { x0 = Ws - 1; x1 = Ws - 1; wx = 0.0; }
        const double* p = src + (size_t)c * Hs * Ws;
        const double a = p[y0 * Ws + x0], b = p[y0 * Ws + x1], cc = p[y1 * Ws + x0], d = p[y1 * Ws + x1];
        const double top = a * (1.0 - wx) + b * wx;           // -ffp-contract=off: separate multiply and add, like NumPy
        const double bot = cc * (1.0 - wx) + d * wx;
        dst[i] = top * (1.0 - wy) + bot * wy;
    }
}

__global__ __launch_bounds__(256) void argmax_kernel(const double* pred, int C, long long HW, uint8_t* classes) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        int best = 0;
        double bv = pred[i];
        bool nan_seen = bv != bv;
        for (int c = 1; c < C; ++c) {
            const double v = pred[(size_t)c * HW + i];
            if (!nan_seen && (v > bv || v != v)) { bv = v; best = c; nan_seen = v != v; }   // first maximum; NaN wins like np.argmax
        }
        classes[i] = (uint8_t)best;
    }
}

// dst = (src >= lo) as 0/1; lo is an integer threshold on u8 codes (class >= 2; heat >= ceil(0.9 * 255))
__global__ __launch_bounds__(256) void threshold_kernel(const uint8_t* src, long long n, int lo, uint8_t* dst) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i] >= lo ? 1 : 0;
}

// ------------------------------------------------------------------------------------------ rectangular morphology
// One separable pass of cv2.erode / cv2.dilate with a k x k rectangle (anchor k/2): offsets -k/2 .. k - k/2 - 1 along
// `axis` (0 = rows / vertical, 1 = columns / horizontal); positions outside the image are ignored.
__global__ __launch_bounds__(256) void morph_pass_kernel(const uint8_t* src, uint8_t* dst, int H, int W, int k, int axis, int take_min) {
    const long long total = (long long)H * W;
    const int h = k / 2;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)(i / W);
        int acc = take_min ? 1 : 0;
        if (axis) {
            const int a = max(0, x - h), e = min(W - 1, x + k - h - 1);
            for (int xx = a; xx <= e; ++xx) { const int v = src[(size_t)y * W + xx] != 0; acc = take_min ? (acc & v) : (acc | v); }
        } else {
            const int a = max(0, y - h), e = min(H - 1, y + k - h - 1);
            for (int yy = a; yy <= e; ++yy) { const int v = src[(size_t)yy * W + x] != 0; acc = take_min ? (acc & v) : (acc | v); }
        }
        dst[i] = (uint8_t)acc;
    }
}

__global__ __launch_bounds__(256) void bwperim_kernel(const uint8_t* src, uint8_t* dst, int H, int W) {
    const long long total = (long long)H * W;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)(i / W);
        const bool v = src[i] != 0;
        const bool n = y > 0 && src[i - W] != 0, s = y + 1 < H && src[i + W] != 0;
        const bool w = x > 0 && src[i - 1] != 0, e = x + 1 < W && src[i + 1] != 0;
        dst[i] = (v && !(n && s && w && e)) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------ convex hull image
// workspace (ints): lo[2H+1] | hi[2H+1] | counts[4] (nleft, nright, npoly, unused) | left[(2H+1)*2] | right[(2H+1)*2]
struct HullWs {
    int* lo; int* hi; int* counts; int* left; int* right;
};
static __host__ __device__ inline HullWs hull_ws(void* ws, int H) {
    HullWs w;
    const size_t n = (size_t)2 * H + 1;
    w.lo = (int*)ws; w.hi = w.lo + n; w.counts = w.hi + n; w.left = w.counts + 4; w.right = w.left + 2 * n;
    return w;
}

__global__ __launch_bounds__(256) void hull_init_kernel(HullWs w, int H) {
    const int n = 2 * H + 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { w.lo[i] = INT_MAX; w.hi[i] = INT_MIN; }
    if (blockIdx.x == 0 && threadIdx.x < 4) w.counts[threadIdx.x] = 0;
}

// one wave per image row: first / last foreground column -> the three doubled rows 2r-1, 2r, 2r+1 (index R + 1)
__global__ __launch_bounds__(64) void hull_rows_kernel(const uint8_t* img, int H, int W, HullWs w) {
    const int r = blockIdx.x, lane = threadIdx.x;
    int cmin = INT_MAX, cmax = INT_MIN;
    for (int c = lane; c < W; c += 64)
        if (img[(size_t)r * W + c]) { cmin = min(cmin, c); cmax = max(cmax, c); }
#pragma unroll
    for (int o = 32; o; o >>= 1) { cmin = min(cmin, __shfl_xor(cmin, o)); cmax = max(cmax, __shfl_xor(cmax, o)); }
    if (lane == 0 && cmin <= cmax) {
        const int R = 2 * r;
        atomicMin(&w.lo[R + 1], 2 * cmin - 1); atomicMax(&w.hi[R + 1], 2 * cmax + 1);
        atomicMin(&w.lo[R], 2 * cmin);         atomicMax(&w.hi[R], 2 * cmax);
        atomicMin(&w.lo[R + 2], 2 * cmin);     atomicMax(&w.hi[R + 2], 2 * cmax);
    }
}

static __device__ inline long long cross3(int oR, int oC, int aR, int aC, int bR, int bC) {
    return (long long)(aR - oR) * (long long)(bC - oC) - (long long)(aC - oC) * (long long)(bR - oR);
}

// Andrew's monotone chain over the doubled rows (already sorted by R): lane 0 builds the left chain (smallest C),
// lane 64 (second wave) the right chain.  The stacks live in global memory (L2-resident, a few thousand points).
__global__ __launch_bounds__(128) void hull_chain_kernel(HullWs w, int H) {
    const int n = 2 * H + 1;
    const int side = threadIdx.x >> 6;
    if (threadIdx.x & 63) return;
    int* st = side ? w.right : w.left;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const int lo = w.lo[i], hi = w.hi[i];
        if (lo > hi) continue;
        const int R = i - 1, Cc = side ? hi : lo;
        while (cnt >= 2) {
            const long long cr = cross3(st[2 * (cnt - 2)], st[2 * (cnt - 2) + 1], st[2 * (cnt - 1)], st[2 * (cnt - 1) + 1], R, Cc);
            if (side ? cr >= 0 : cr <= 0) --cnt; else break;
        }
        st[2 * cnt] = R; st[2 * cnt + 1] = Cc;
        ++cnt;
    }
    w.counts[side] = cnt;
}

static __device__ inline long long floor_div(long long a, long long b) {   // b > 0
    long long q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

// where the chain crosses doubled row R: exact rational num / den (den > 0)
static __device__ inline void chain_cross(const int* ch, int cnt, int R, long long& num, long long& den) {
    int a = 0, b = cnt - 1;                                  // last vertex with R_v <= R
    while (a < b) { const int m = (a + b + 1) >> 1; if (ch[2 * m] <= R) a = m; else b = m - 1; }
    if (a == cnt - 1 || ch[2 * a] == R) { num = ch[2 * a + 1]; den = 1; return; }
    const long long R0 = ch[2 * a], C0 = ch[2 * a + 1], R1 = ch[2 * a + 2], C1 = ch[2 * a + 3];
    den = R1 - R0;
    num = C0 * den + (C1 - C0) * (R - R0);
}

// one workgroup per image row: columns ceil(l / 2) .. floor(r / 2) of the hull's cut with that row
__global__ __launch_bounds__(256) void hull_fill_kernel(uint8_t* out, int H, int W, HullWs w) {
    const int r = blockIdx.x;
    __shared__ int span[2];
    if (threadIdx.x == 0) {
        int lo = 1, hi = 0;
        const int nl = w.counts[0], nr = w.counts[1];
        if (nl > 0) {
            const int R = 2 * r, Rmin = w.left[0], Rmax = w.left[2 * (nl - 1)];
            if (R >= Rmin && R <= Rmax) {
                long long ln, ld, rn, rd;
                chain_cross(w.left, nl, R, ln, ld);
                chain_cross(w.right, nr, R, rn, rd);
                long long cl = -floor_div(-ln, 2 * ld), ch = floor_div(rn, 2 * rd);
                if (cl < 0) cl = 0;
                if (ch > W - 1) ch = W - 1;
                lo = (int)cl; hi = (int)ch;
            }
        }
        span[0] = lo; span[1] = hi;
    }
    __syncthreads();
    const int lo = span[0], hi = span[1];
    for (int c = threadIdx.x; c < W; c += 256) out[(size_t)r * W + c] = (c >= lo && c <= hi) ? 1 : 0;
}

// closed outline polygon (x, y) float64 in pixel units: down the right chain, back up the left chain, consecutive
// duplicates dropped, first point repeated at the end (oracle hull_polygon); count -> counts[2]
__global__ void hull_polygon_kernel(HullWs w, double* out_xy, int cap) {
    if (threadIdx.x || blockIdx.x) return;
    const int nl = w.counts[0], nr = w.counts[1];
    int cnt = 0, pR = 0, pC = 0, fR = 0, fC = 0;
    auto push = [&](int R, int Cc) {
        if (cnt && R == pR && Cc == pC) return;
        if (!cnt) { fR = R; fC = Cc; }
        if (cnt < cap) { out_xy[2 * cnt] = (double)Cc / 2.0; out_xy[2 * cnt + 1] = (double)R / 2.0; }
        ++cnt; pR = R; pC = Cc;
    };
    for (int i = 0; i < nr; ++i) push(w.right[2 * i], w.right[2 * i + 1]);
    for (int i = nl - 1; i >= 0; --i) push(w.left[2 * i], w.left[2 * i + 1]);
    if (cnt && !(pR == fR && pC == fC)) push(fR, fC);
    w.counts[2] = cnt;
}

// ------------------------------------------------------------------------------------------ exact counts
// out[0] += sum(a & b), out[1] += sum(a | b) over 0/1 masks (a is compared > 0)   (utils/eval.py:104)
__global__ __launch_bounds__(256) void iou_counts_kernel(const uint8_t* a, const uint8_t* b, long long n, unsigned long long* out) {
    unsigned long long inter = 0, uni = 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const bool x = a[i] != 0, y = b[i] != 0;
        inter += x && y; uni += x || y;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { inter += __shfl_xor(inter, o); uni += __shfl_xor(uni, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], inter); atomicAdd(&out[1], uni); }
}

// The integer sums behind utils/eval.py:107-121 for a class map p (optionally times a 0/1 mask), ground truth gt:
//   out[0] = #(gt > 0)            out[1] = #(p == gt and gt > 0)        out[2] = sum |p - gt|
//   out[3] = sum max(gt, |gt - 3|) * (1 - (1 - (p > 0)) * (gt != 1))
//            the reference writes `(1 - gt > 0)` = `(1 - gt) > 0` with gt the uint8 array of a PIL image, so 1 - gt WRAPS
//            (0 -> 1, 1 -> 0, 2 -> 255, 3 -> 254) and the factor is gt != 1, not gt == 0 (r02 had the int64 reading)
//   out[4] = #((p > 0) and (gt > 0))   out[5] = #((p > 0) or (gt > 0))
__global__ __launch_bounds__(256) void score_counts_kernel(const uint8_t* p, const uint8_t* gt, const uint8_t* mask, long long n,
                                                           unsigned long long* out) {
    unsigned long long s[6] = {0, 0, 0, 0, 0, 0};
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int pv = mask ? (int)p[i] * (int)mask[i] : (int)p[i], g = gt[i];
        s[0] += g > 0;
        s[1] += (pv == g) && g > 0;
        s[2] += (unsigned long long)abs(pv - g);
        const int mx = max(g, abs(g - 3));
        s[3] += (unsigned long long)(mx * (1 - (1 - (pv > 0)) * (g != 1)));
        s[4] += (pv > 0) && (g > 0);
        s[5] += (pv > 0) || (g > 0);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int o = 32; o; o >>= 1) s[k] += __shfl_xor(s[k], o);
        if ((threadIdx.x & 63) == 0 && s[k]) atomicAdd(&out[k], s[k]);
    }
}

// ------------------------------------------------------------------------------------------ esp
// u[0] = 0, u[i] = u[i-1] + sqrt(dx^2 + dy^2): the same sequential float64 sum as np.cumsum (one lane; n is a contour)
__global__ void esp_arclength_kernel(const double* pts, int n, double* u) {
    if (threadIdx.x || blockIdx.x) return;
    double acc = 0.0;
    u[0] = 0.0;
    for (int i = 1; i < n; ++i) {
        const double xd = pts[2 * i] - pts[2 * i - 2], yd = pts[2 * i + 1] - pts[2 * i - 1];
        const double d = sqrt(xd * xd + yd * yd);
        acc = (i == 1) ? d : acc + d;
        u[i] = acc;
    }
}

// np.linspace(0, u.max(), num) + np.interp(t, u, x / y)  (NumPy's arr_interp: slope * (t - u[j]) + f[j], exact hits return f[j])
__global__ __launch_bounds__(256) void esp_interp_kernel(const double* pts, const double* u, int n, int num, double* out) {
    const double umax = u[n - 1];                            // u is non-decreasing: its maximum is its last element
    const double step = num > 1 ? umax / (double)(num - 1) : 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < num; i += gridDim.x * 256) {
        double t = (step == 0.0 && num > 1) ? ((double)i / (double)(num - 1)) * umax : (double)i * step;
        if (num > 1 && i == num - 1) t = umax;
        int a = 0, b = n - 1;                                // largest j with u[j] <= t   (t >= u[0] = 0)
        while (a < b) { const int m = (a + b + 1) >> 1; if (u[m] <= t) a = m; else b = m - 1; }
        const int j = a;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double f0 = pts[2 * j + k];
            double v;
            if (t > umax) v = pts[2 * (n - 1) + k];
            else if (j == n - 1 || u[j] == t) v = f0;
            else {
                const double f1 = pts[2 * j + 2 + k];
                const double slope = (f1 - f0) / (u[j + 1] - u[j]);
                v = slope * (t - u[j]) + f0;
                if (v != v) {
                    v = slope * (t - u[j + 1]) + f1;
                    if (v != v && f0 == f1) v = f0;
                }
            }
            out[2 * i + k] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ dispatch
static int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT)

int wsi_resize_dispatch(const double* src, int C, int Hs, int Ws, double* dst, int Hd, int Wd, hipStream_t st) {
    if (C <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return WSI_EINVAL;
    hipLaunchKernelGGL(resize_bilinear_f64_kernel, dim3(grid_for((long long)C * Hd * Wd)), dim3(256), 0, st, src, C, Hs, Ws, dst, Hd, Wd);
    return LAUNCH_OK();
}
int wsi_argmax_dispatch(const double* pred, int C, long long HW, uint8_t* classes, hipStream_t st) {
    if (C <= 0 || C > 255 || HW <= 0) return WSI_EINVAL;
    hipLaunchKernelGGL(argmax_kernel, dim3(grid_for(HW)), dim3(256), 0, st, pred, C, HW, classes);
    return LAUNCH_OK();
}
int wsi_threshold_dispatch(const uint8_t* src, long long n, int lo, uint8_t* dst, hipStream_t st) {
    if (n <= 0) return WSI_EINVAL;
    hipLaunchKernelGGL(threshold_kernel, dim3(grid_for(n)), dim3(256), 0, st, src, n, lo, dst);
    return LAUNCH_OK();
}
// op: 0 erode, 1 dilate, 2 open (erode then dilate).  tmp: H*W bytes (two for open: tmp and dst alternate)
int wsi_morph_dispatch(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int H, int W, int k, int op, hipStream_t st) {
    if (H <= 0 || W <= 0 || k <= 0 || op < 0 || op > 2 || src == dst || src == tmp || dst == tmp) return WSI_EINVAL;
    const int g = grid_for((long long)H * W);
    auto pass2 = [&](const uint8_t* s, uint8_t* d, int take_min) {        // horizontal into tmp, vertical into d
        hipLaunchKernelGGL(morph_pass_kernel, dim3(g), dim3(256), 0, st, s, tmp, H, W, k, 1, take_min);
        hipLaunchKernelGGL(morph_pass_kernel, dim3(g), dim3(256), 0, st, (const uint8_t*)tmp, d, H, W, k, 0, take_min);
    };
    if (op == 0) pass2(src, dst, 1);
    else if (op == 1) pass2(src, dst, 0);
    else {
        // erode: src -> tmp -> dst; dilate: dst -> tmp -> dst needs a third buffer, so run the last vertical pass in place
        // is NOT safe (a column reads its neighbours' rows): erode src -> dst, then dilate dst -> tmp (h) and tmp -> dst (v)
        pass2(src, dst, 1);
        pass2(dst, dst, 0);                                   // horizontal dst -> tmp, vertical tmp -> dst: dst is only read before it is rewritten
    }
    return LAUNCH_OK();
}
int wsi_bwperim_dispatch(const uint8_t* src, uint8_t* dst, int H, int W, hipStream_t st) {
    if (H <= 0 || W <= 0 || src == dst) return WSI_EINVAL;
    hipLaunchKernelGGL(bwperim_kernel, dim3(grid_for((long long)H * W)), dim3(256), 0, st, src, dst, H, W);
    return LAUNCH_OK();
}
size_t wsi_hull_ws_bytes(int H) { return H <= 0 ? 0 : ((size_t)(2 * H + 1) * 6 + 4) * sizeof(int); }
int wsi_hull_dispatch(const uint8_t* src, uint8_t* dst, int H, int W, void* ws, hipStream_t st) {
    if (H <= 0 || W <= 0 || !ws || W > (1 << 29)) return WSI_EINVAL;
    const HullWs w = hull_ws(ws, H);
    hipLaunchKernelGGL(hull_init_kernel, dim3((2 * H + 256) / 256), dim3(256), 0, st, w, H);
    hipLaunchKernelGGL(hull_rows_kernel, dim3(H), dim3(64), 0, st, src, H, W, w);
    hipLaunchKernelGGL(hull_chain_kernel, dim3(1), dim3(128), 0, st, w, H);
    if (dst) hipLaunchKernelGGL(hull_fill_kernel, dim3(H), dim3(256), 0, st, dst, H, W, w);
    return LAUNCH_OK();
}
int wsi_hull_polygon_dispatch(void* ws, int H, double* out_xy, int cap, hipStream_t st) {
    if (!ws || H <= 0 || !out_xy || cap <= 0) return WSI_EINVAL;
    hipLaunchKernelGGL(hull_polygon_kernel, dim3(1), dim3(64), 0, st, hull_ws(ws, H), out_xy, cap);
    return LAUNCH_OK();
}
int wsi_iou_counts_dispatch(const uint8_t* a, const uint8_t* b, long long n, unsigned long long* out, hipStream_t st) {
    if (n <= 0) return WSI_EINVAL;
    if (hipMemsetAsync(out, 0, 2 * sizeof(unsigned long long), st) != hipSuccess) return WSI_EFAULT;
    hipLaunchKernelGGL(iou_counts_kernel, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, a, b, n, out);
    return LAUNCH_OK();
}
int wsi_score_counts_dispatch(const uint8_t* p, const uint8_t* gt, const uint8_t* mask, long long n, unsigned long long* out, hipStream_t st) {
    if (n <= 0) return WSI_EINVAL;
    if (hipMemsetAsync(out, 0, 6 * sizeof(unsigned long long), st) != hipSuccess) return WSI_EFAULT;
    hipLaunchKernelGGL(score_counts_kernel, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, p, gt, mask, n, out);
    return LAUNCH_OK();
}
int wsi_esp_dispatch(const double* pts, int n, int num, double* out, double* scratch, hipStream_t st) {
    if (n < 1 || num < 1) return WSI_EINVAL;
    hipLaunchKernelGGL(esp_arclength_kernel, dim3(1), dim3(64), 0, st, pts, n, scratch);
    hipLaunchKernelGGL(esp_interp_kernel, dim3(grid_for(num)), dim3(256), 0, st, pts, (const double*)scratch, n, num, out);
    return LAUNCH_OK();
}
