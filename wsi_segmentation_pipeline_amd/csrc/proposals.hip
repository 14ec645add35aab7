// Region-proposal generation on the device: the steps in front of the bag path.
//   hsv_mask      find_nuclei(mode='hsv') (/root/reference/utils/preprocessing.py:94-98): skimage rgb2hsv saturation > 0.1, float64
//   cc_*          cv2.connectedComponentsWithStats of the ground-truth thumbnail (/root/reference/scannet.py:55): 8-connected
//                 union-find labelling, labels renumbered 1.. in raster order of each component's first pixel
//   kmeans_*      the key points of /root/reference/utils/regiontools.py:68-102 by deterministic Lloyd iterations (the reference's
//                 sklearn KMeans is version- and RNG-dependent: own spec, oracle/proposals_oracle.py)
// Thumbnail-sized, irregular, HBM / latency-bound integer work: plain kernels, int64 / float64 arithmetic that NumPy reproduces
// bit for bit (-ffp-contract=off).
#include "common.h"

// ------------------------------------------------------------------------------------------ HSV saturation mask
__global__ __launch_bounds__(256) void hsv_mask_kernel(const uint8_t* rgb, long long npix, int stride, double thresh, uint8_t* mask) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const uint8_t* p = rgb + i * stride;
        const double r = (double)p[0] / 255.0, g = (double)p[1] / 255.0, b = (double)p[2] / 255.0;
        const double v = fmax(r, fmax(g, b)), mn = fmin(r, fmin(g, b));
        const double delta = v - mn;
        const double s = delta == 0.0 ? 0.0 : delta / v;
        mask[i] = s > thresh ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------ connected components
// label[p] = p + 1 for foreground (0 = background); roots are minimal flat indices.  Union by atomicMin on roots
// (Playne-Hawick style label equivalence), iterated to a fixed point by the host loop (a changed flag), then flattened.
static __device__ inline int cc_find(const int* L, int a) {
    int r = a;
    while (L[r] - 1 != r) r = L[r] - 1;
    return r;
}
__global__ __launch_bounds__(256) void cc_init_kernel(const uint8_t* mask, int* L, long long n) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) L[i] = mask[i] ? (int)i + 1 : 0;
}
__global__ __launch_bounds__(256) void cc_merge_kernel(int* L, int H, int W, int* changed) {
    const long long n = (long long)H * W;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (!L[i]) continue;
        const int y = (int)(i / W), x = (int)(i % W);
        int ra = cc_find(L, (int)i);
        // the four already-scanned 8-neighbours (the other four are covered from their side)
        const int dy[4] = {-1, -1, -1, 0}, dx[4] = {-1, 0, 1, -1};
        for (int k = 0; k < 4; ++k) {
            const int yy = y + dy[k], xx = x + dx[k];
            if (yy < 0 || xx < 0 || xx >= W) continue;
            const long long q = (long long)yy * W + xx;
            if (!L[q]) continue;
            int rb = cc_find(L, (int)q);
            while (ra != rb) {                                // hook the larger root under the smaller one
                const int hi = max(ra, rb), lo = min(ra, rb);
                const int old = atomicMin(&L[hi], lo + 1);
                if (old == hi + 1) { *changed = 1; ra = rb = lo; }
                else { ra = cc_find(L, old - 1); rb = lo; }   // somebody re-hooked `hi` meanwhile: merge their root with ours
            }
        }
    }
}
__global__ __launch_bounds__(256) void cc_flatten_kernel(int* L, long long n, int* is_root) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (!L[i]) { is_root[i] = 0; continue; }
        const int r = cc_find(L, (int)i);
        L[i] = r + 1;
        is_root[i] = (r == (int)i);
    }
}
// exclusive prefix sum of 0/1 flags in three kernels (block sums, scan of block sums by one workgroup, add back)
__global__ __launch_bounds__(256) void scan_block_kernel(const int* in, int* out, int* block_sums, long long n) {
    __shared__ int s[256];
    const long long base = (long long)blockIdx.x * 1024;
    int v[4], sum = 0;
    for (int k = 0; k < 4; ++k) { const long long i = base + threadIdx.x * 4 + k; v[k] = i < n ? in[i] : 0; sum += v[k]; }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) { int t = threadIdx.x >= o ? s[threadIdx.x - o] : 0; __syncthreads(); s[threadIdx.x] += t; __syncthreads(); }
    int run = s[threadIdx.x] - sum;
    for (int k = 0; k < 4; ++k) { const long long i = base + threadIdx.x * 4 + k; if (i < n) out[i] = run; run += v[k]; }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = s[255];
}
__global__ __launch_bounds__(256) void scan_sums_kernel(int* block_sums, int nb, int* total) {
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    __shared__ int s[256];
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? block_sums[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) { int t = threadIdx.x >= o ? s[threadIdx.x - o] : 0; __syncthreads(); s[threadIdx.x] += t; __syncthreads(); }
        if (i < nb) block_sums[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void cc_rank_kernel(const int* L, const int* excl, const int* block_sums, long long n, int* out) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int l = L[i];
        if (!l) { out[i] = 0; continue; }
        const long long r = l - 1;                            // the root's flat index: its rank among roots + 1 is the label
        out[i] = excl[r] + block_sums[r >> 10] + 1;
    }
}

// ------------------------------------------------------------------------------------------ k-means (Lloyd, deterministic)
// pts (N,2) int32 (x, y); centres (k,2) float64.  assign: nearest centre in float64, first minimum; sums in int64 atomics (exact)
__global__ __launch_bounds__(256) void kmeans_assign_kernel(const int* pts, int n, const double* centres, int k, int* labels,
                                                            unsigned long long* sums, int* changed) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double px = (double)pts[2 * i], py = (double)pts[2 * i + 1];
        int best = 0;
        double bd = 0.0;
        for (int j = 0; j < k; ++j) {
            const double dx = px - centres[2 * j], dy = py - centres[2 * j + 1];
            const double d = dx * dx + dy * dy;
            if (j == 0 || d < bd) { bd = d; best = j; }
        }
        if (labels[i] != best) { labels[i] = best; *changed = 1; }
        atomicAdd(&sums[3 * best], (unsigned long long)(long long)pts[2 * i]);
        atomicAdd(&sums[3 * best + 1], (unsigned long long)(long long)pts[2 * i + 1]);
        atomicAdd(&sums[3 * best + 2], 1ull);
    }
}
__global__ void kmeans_update_kernel(double* centres, int k, const unsigned long long* sums) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const long long c = (long long)sums[3 * j + 2];
    if (c) {
        centres[2 * j] = (double)(long long)sums[3 * j] / (double)c;
        centres[2 * j + 1] = (double)(long long)sums[3 * j + 1] / (double)c;
    }
}

// ------------------------------------------------------------------------------------------ tile grid (sliding-window path)
// /root/reference/utils/dataset.py:143-166: candidates in the reference's order (interior raster, right-edge column, bottom-edge
// row, no corner), kept iff the level-2 mask window mask[yp:yp+dy, xp:xp+dx] (numpy slice clipping) is >= thresh nonzero.
struct TileGridArgs {
    int iw, ih, ph, pw, sh, sw, ny, nx, dx, dy, MH, MW;
    double m, thresh;
    const uint8_t* mask;
};
static __device__ inline void tile_at(const TileGridArgs& g, long long i, int& x, int& y) {
    const long long interior = (long long)g.ny * g.nx;
    if (i < interior) { y = 1 + (int)(i / g.nx) * g.sh; x = 1 + (int)(i % g.nx) * g.sw; }
    else if (i < interior + g.ny) { x = g.iw - 1 - g.pw; y = 1 + (int)(i - interior) * g.sh; }
    else { y = g.ih - 1 - g.ph; x = 1 + (int)(i - interior - g.ny) * g.sw; }
}
__global__ __launch_bounds__(256) void tile_keep_kernel(TileGridArgs g, long long n, int* flags) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int keep = 1;
        if (g.mask) {
            int x, y;
            tile_at(g, i, x, y);
            const long long yp = (long long)((double)y * g.m), xp = (long long)((double)x * g.m);      // int(ypos * m)
            const long long y1 = min(yp + g.dy, (long long)g.MH), x1 = min(xp + g.dx, (long long)g.MW);
            const long long rows = y1 - yp, cols = x1 - xp;
            keep = 0;
            if (rows > 0 && cols > 0) {
                long long cnt = 0;
                for (long long r = yp; r < y1; ++r) {
                    const uint8_t* row = g.mask + (size_t)r * g.MW;
                    for (long long c = xp; c < x1; ++c) cnt += row[c] != 0;
                }
                keep = (double)cnt / (double)(rows * cols) >= g.thresh;
            }
        }
        flags[i] = keep;
    }
}
__global__ __launch_bounds__(256) void tile_emit_kernel(TileGridArgs g, long long n, const int* flags, const int* excl, const int* block_sums, int* out_xy) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (!flags[i]) continue;
        int x, y;
        tile_at(g, i, x, y);
        const long long o = (long long)excl[i] + block_sums[i >> 10];
        out_xy[2 * o] = x; out_xy[2 * o + 1] = y;
    }
}

// ------------------------------------------------------------------------------------------ dispatch
static int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT)

int wsi_hsv_mask_dispatch(const uint8_t* rgb, long long npix, int stride, double thresh, uint8_t* mask, hipStream_t st) {
    if (npix <= 0 || stride < 3) return WSI_EINVAL;
    hipLaunchKernelGGL(hsv_mask_kernel, dim3(grid_for(npix)), dim3(256), 0, st, rgb, npix, stride, thresh, mask);
    return LAUNCH_OK();
}

// scratch: ints [n labels L | n flags | n exclusive scan | nb block sums | 2 (changed, total)]
size_t wsi_cc_scratch_bytes(int H, int W) {
    const long long n = (long long)H * W, nb = (n + 1023) / 1024;
    return (size_t)(3 * n + nb + 4) * sizeof(int);
}
int wsi_cc_dispatch(const uint8_t* mask, int H, int W, int* labels_out, int* count_out, void* scratch, hipStream_t st) {
    if (H <= 0 || W <= 0 || (long long)H * W > 0x7ffffff0LL) return WSI_EINVAL;
    const long long n = (long long)H * W, nb = (n + 1023) / 1024;
    int* L = (int*)scratch;
    int *flags = L + n, *excl = flags + n, *bsum = excl + n, *misc = bsum + nb;
    const int g = grid_for(n);
    hipLaunchKernelGGL(cc_init_kernel, dim3(g), dim3(256), 0, st, mask, L, n);
    // union-find with atomic hooks converges in one sweep for most images; sweep until a pass changes nothing (bounded)
    for (int it = 0; it < 64; ++it) {
        if (hipMemsetAsync(misc, 0, sizeof(int), st) != hipSuccess) return WSI_EFAULT;
        hipLaunchKernelGGL(cc_merge_kernel, dim3(g), dim3(256), 0, st, L, H, W, misc);
        int changed = 0;
        if (hipMemcpyAsync(&changed, misc, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return WSI_EFAULT;
        if (!changed) break;
    }
    hipLaunchKernelGGL(cc_flatten_kernel, dim3(g), dim3(256), 0, st, L, n, flags);
    hipLaunchKernelGGL(scan_block_kernel, dim3((int)nb), dim3(256), 0, st, (const int*)flags, excl, bsum, n);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, st, bsum, (int)nb, misc + 1);
    hipLaunchKernelGGL(cc_rank_kernel, dim3(g), dim3(256), 0, st, (const int*)L, (const int*)excl, (const int*)bsum, n, labels_out);
    if (count_out && hipMemcpyAsync(count_out, misc + 1, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) return WSI_EFAULT;
    return LAUNCH_OK();
}

// centres: k x 2 doubles, initialised by the caller; labels: n ints (any content: set to -1 here); sums scratch: 3k u64 + 1 int
int wsi_kmeans_dispatch(const int* pts, int n, double* centres, int k, int iters, int* labels, void* scratch, hipStream_t st) {
    if (n <= 0 || k <= 0 || iters <= 0) return WSI_EINVAL;
    unsigned long long* sums = (unsigned long long*)scratch;
    int* changed = (int*)(sums + 3 * (size_t)k);
    if (hipMemsetAsync(labels, 0xff, (size_t)n * sizeof(int), st) != hipSuccess) return WSI_EFAULT;
    for (int it = 0; it < iters; ++it) {
        if (hipMemsetAsync(sums, 0, 3 * (size_t)k * sizeof(unsigned long long) + sizeof(int), st) != hipSuccess) return WSI_EFAULT;
        hipLaunchKernelGGL(kmeans_assign_kernel, dim3(grid_for(n)), dim3(256), 0, st, pts, n, (const double*)centres, k, labels, sums, changed);
        int ch = 0;
        if (hipMemcpyAsync(&ch, changed, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return WSI_EFAULT;
        if (!ch) break;                                       // no assignment changed: the centres are final
        hipLaunchKernelGGL(kmeans_update_kernel, dim3((k + 63) / 64), dim3(64), 0, st, centres, k, (const unsigned long long*)sums);
    }
    return LAUNCH_OK();
}

static int range_len(int lo, int hi, int step) { return hi > lo ? (hi - lo + step - 1) / step : 0; }
long long wsi_tile_grid_candidates_impl(int iw, int ih, int ph, int pw, int sh, int sw) {
    if (sh <= 0 || sw <= 0) return -1;
    const long long ny = range_len(1, ih - 1 - ph, sh), nx = range_len(1, iw - 1 - pw, sw);
    return ny * nx + ny + nx;
}
size_t wsi_tile_grid_scratch_bytes_impl(long long n) { return (size_t)(2 * n + (n + 1023) / 1024 + 4) * sizeof(int); }
int wsi_tile_grid_dispatch(int iw, int ih, int ph, int pw, int sh, int sw, const uint8_t* mask, int MH, int MW, double m, double thresh,
                           int* out_xy, int* count_out, void* scratch, hipStream_t st) {
    const long long n = wsi_tile_grid_candidates_impl(iw, ih, ph, pw, sh, sw);
    if (n < 0 || n > 0x7ffffff0LL || ph <= 0 || pw <= 0 || (mask && (MH <= 0 || MW <= 0 || !(m > 0.0)))) return WSI_EINVAL;
    if (n == 0) return hipMemsetAsync(count_out, 0, sizeof(int), st) == hipSuccess ? WSI_OK : WSI_EFAULT;
    TileGridArgs g;
    g.iw = iw; g.ih = ih; g.ph = ph; g.pw = pw; g.sh = sh; g.sw = sw;
    g.ny = range_len(1, ih - 1 - ph, sh); g.nx = range_len(1, iw - 1 - pw, sw);
    g.dx = (int)((double)pw * m); g.dy = (int)((double)ph * m);
    g.MH = MH; g.MW = MW; g.m = m; g.thresh = thresh; g.mask = mask;
    const long long nb = (n + 1023) / 1024;
    int* flags = (int*)scratch;
    int *excl = flags + n, *bsum = excl + n, *total = bsum + nb;
    hipLaunchKernelGGL(tile_keep_kernel, dim3(grid_for(n)), dim3(256), 0, st, g, n, flags);
    hipLaunchKernelGGL(scan_block_kernel, dim3((int)nb), dim3(256), 0, st, (const int*)flags, excl, bsum, n);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, st, bsum, (int)nb, total);
    hipLaunchKernelGGL(tile_emit_kernel, dim3(grid_for(n)), dim3(256), 0, st, g, n, (const int*)flags, (const int*)excl, (const int*)bsum, out_xy);
    if (hipMemcpyAsync(count_out, total, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) return WSI_EFAULT;
    return LAUNCH_OK();
}
