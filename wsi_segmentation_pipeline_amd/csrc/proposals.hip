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
__global__ __launch_bounds__(256) void cc_merge_kernel(int* L, int H, int W, int* changed, int conn4) {
    const long long n = (long long)H * W;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (!L[i]) continue;
        const int y = (int)(i / W), x = (int)(i % W);
        int ra = cc_find(L, (int)i);
        // the four already-scanned 8-neighbours (the other four are covered from their side); 4-connectivity: up and left only
        const int dy[4] = {-1, -1, -1, 0}, dx[4] = {-1, 0, 1, -1};
        for (int k = 0; k < 4; ++k) {
            if (conn4 && (k == 0 || k == 2)) continue;
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

// Farthest-point seeds (r04; oracle/proposals_oracle.py kmeans_seed_farthest): the point farthest from the floor-mean, then k - 1
// times the point whose squared distance to the nearest seed is largest; exact integers, every tie to the lower raster index.
// One workgroup (the point sets are region thumbnails: 10^2 .. 10^5 points, k <= a few dozen); dmin: n int64 of scratch.
__global__ __launch_bounds__(1024) void kmeans_seed_farthest_kernel(const int* pts, int n, int k, double* centres, long long* dmin) {
    __shared__ long long rv[1024];
    __shared__ int ri[1024];
    __shared__ long long sh[2];
    const int tid = threadIdx.x;
    auto argmax_block = [&](long long v, int i) {             // largest value, lowest index among equals -> ri[0]
        rv[tid] = v; ri[tid] = i;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if (tid < s) {
                const long long v2 = rv[tid + s];
                const int i2 = ri[tid + s];
                if (v2 > rv[tid] || (v2 == rv[tid] && i2 < ri[tid])) { rv[tid] = v2; ri[tid] = i2; }
            }
            __syncthreads();
        }
    };
    long long sx = 0, sy = 0;
    for (int i = tid; i < n; i += 1024) { sx += pts[2 * i]; sy += pts[2 * i + 1]; }
    rv[tid] = sx;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (tid < s) rv[tid] += rv[tid + s]; __syncthreads(); }
    if (tid == 0) sh[0] = rv[0];
    __syncthreads();
    rv[tid] = sy;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (tid < s) rv[tid] += rv[tid + s]; __syncthreads(); }
    if (tid == 0) sh[1] = rv[0];
    __syncthreads();
    // floor division (the sums of pixel coordinates are >= 0)
    const long long cx = sh[0] / n, cy = sh[1] / n;
    long long bv = -1;
    int bi = 0x7fffffff;
    for (int i = tid; i < n; i += 1024) {
        const long long dx = pts[2 * i] - cx, dy = pts[2 * i + 1] - cy, d = dx * dx + dy * dy;
        if (d > bv) { bv = d; bi = i; }                       // ascending i per thread: the first maximum stays
    }
    argmax_block(bv, bi);
    int seed = ri[0];
    __syncthreads();
    if (tid == 0) { centres[0] = (double)pts[2 * seed]; centres[1] = (double)pts[2 * seed + 1]; }
    for (int c = 1; c <= k; ++c) {                            // pass c: fold seed c - 1 into dmin, pick seed c (c < k)
        const long long qx = pts[2 * seed], qy = pts[2 * seed + 1];
        bv = -1; bi = 0x7fffffff;
        for (int i = tid; i < n; i += 1024) {
            const long long dx = pts[2 * i] - qx, dy = pts[2 * i + 1] - qy, d = dx * dx + dy * dy;
            const long long m = c == 1 ? d : (d < dmin[i] ? d : dmin[i]);
            dmin[i] = m;
            if (m > bv) { bv = m; bi = i; }
        }
        if (c == k) break;
        argmax_block(bv, bi);
        seed = ri[0];
        __syncthreads();
        if (tid == 0) { centres[2 * c] = (double)pts[2 * seed]; centres[2 * c + 1] = (double)pts[2 * seed + 1]; }
    }
}
int wsi_kmeans_seed_farthest_dispatch(const int* pts, int n, int k, double* centres, void* scratch, hipStream_t st) {
    if (n <= 0 || k <= 0 || k > n) return WSI_EINVAL;
    hipLaunchKernelGGL(kmeans_seed_farthest_kernel, dim3(1), dim3(1024), 0, st, pts, n, k, centres, (long long*)scratch);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

// ------------------------------------------------------------------------------------------ find_nuclei, the other modes
// mode 'lab' (/root/reference/utils/preprocessing.py:88-92): a = rgb2lab(image)[..., 1]; mask = a > (1 + mu_percent) * mean(a).
// skimage is absent: own deterministic specification (oracle/wsi_oracle.py find_nuclei_lab) - `a` in 2^-20 fixed point so that the
// mean is an exact integer sum / count.  Two kernels: a -> int32 + sum; threshold.
#define LAB_FIX 1048576.0
static __device__ __forceinline__ double lab_a_of_rgb(const uint8_t* p) {
    double v[3], f[2];
    for (int c = 0; c < 3; ++c) {
        const double a = (double)p[c] / 255.0;
        v[c] = a > 0.04045 ? pow((a + 0.055) / 1.055, 2.4) : a / 12.92;
    }
    const double M[2][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}};
    const double white[2] = {0.95047, 1.0};
    for (int r = 0; r < 2; ++r) {
        double t = v[0] * M[r][0];
        t += v[1] * M[r][1];
        t += v[2] * M[r][2];
        t = t / white[r];
        f[r] = t > 0.008856 ? cbrt(t) : 7.787 * t + 16.0 / 116.0;
    }
    return 500.0 * (f[0] - f[1]);
}
__global__ __launch_bounds__(256) void lab_a_kernel(const uint8_t* rgb, long long npix, int stride, int* aq, unsigned long long* sum) {
    long long s = 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const int q = (int)llrint(lab_a_of_rgb(rgb + i * stride) * LAB_FIX);
        aq[i] = q;
        s += q;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(sum, (unsigned long long)s);
}
__global__ __launch_bounds__(256) void lab_thresh_kernel(const int* aq, long long npix, const unsigned long long* sum, double mu_percent, uint8_t* mask) {
    const double mu = ((double)(long long)*sum / LAB_FIX) / (double)npix;
    const double thr = (1 + mu_percent) * mu;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < npix; i += (long long)gridDim.x * 256)
        mask[i] = ((double)aq[i] / LAB_FIX > thr) ? 1 : 0;
}
// binary_fill_holes (scipy.ndimage, default structure = 4-connectivity of the BACKGROUND): background components that touch
// no image border are holes.  labels = 4-connected components of the inverted mask (wsi_cc_dispatch conn4); touch[label] = 1 for
// labels on the border; out = mask | (label && !touch[label]).
__global__ __launch_bounds__(256) void holes_touch_kernel(const int* labels, int H, int W, int* touch) {
    const int n = 2 * (H + W);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        int y, x;
        if (i < W) { y = 0; x = i; }
        else if (i < 2 * W) { y = H - 1; x = i - W; }
        else if (i < 2 * W + H) { y = i - 2 * W; x = 0; }
        else { y = i - 2 * W - H; x = W - 1; }
        const int l = labels[(size_t)y * W + x];
        if (l) touch[l] = 1;
    }
}
__global__ __launch_bounds__(256) void holes_fill_kernel(const uint8_t* mask, const int* labels, const int* touch, long long n, uint8_t* out) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (mask[i] || (labels[i] && !touch[labels[i]])) ? 1 : 0;
}
size_t wsi_cc_scratch_bytes(int H, int W);
size_t wsi_fill_holes_scratch_bytes_impl(int H, int W) {
    const size_t n = (size_t)H * W;
    return wsi_cc_scratch_bytes(H, W) + n /* inverted mask */ + 2 * (n + 2) * sizeof(int) /* labels, touch */ + 512;
}

// ------------------------------------------------------------------------------------------ SLIC superpixels
// skimage.segmentation.slic as /root/reference/slic.py:43 calls it (n_segments = 200, compactness = 20, sigma = 5,
// enforce_connectivity = False) on a 2-D RGB thumbnail; skimage is absent, so this follows its published algorithm
// (slic_superpixels.py / _slic.pyx of the 0.15 line) as an own deterministic specification, oracle/proposals_oracle.py slic_labels:
//   img_as_float -> scipy.ndimage.gaussian_filter (reflect, truncate 4; the size-1 depth axis is filtered too) -> rgb2lab ->
//   * (1 / compactness) -> [spec] rounded to 2^-20 fixed point, so that the cluster sums are exact integers and do not depend on
//   the summation order -> 10 rounds of {assign: every centre claims the pixels of its (4 step + 1)^2 window it is nearest to,
//   ties to the lower centre; update: centre = mean position, mean colour}.  Centres start on skimage's regular grid with
//   colour 0.  [spec] a centre that lost all its pixels is dropped (skimage divides by zero there).
// Thumbnail-sized float64 work; one thread per pixel, centres in LDS.
#define SLIC_FIX 1048576.0                                            // 2^20
#define SLIC_MAXK 2048
static __device__ __forceinline__ int slic_reflect(int p, int n) {   // scipy 'reflect': d c b a | a b c d | d c b a
    const int m = 2 * n;
    p %= m;
    if (p < 0) p += m;
    return p < n ? p : m - 1 - p;
}
// one separable pass of correlate1d with a symmetric kernel, in scipy's summation order: centre first, then the pairs from
// the outermost inwards.  axis 0 = the size-1 depth axis (every neighbour reflects onto the pixel itself), 1 = rows, 2 = columns.
__global__ __launch_bounds__(256) void slic_blur_kernel(const double* in, const uint8_t* rgb, double* out, int H, int W, const double* fw,
                                                        int radius, int axis) {
    const long long n = (long long)H * W * 3;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % 3);
        const long long px = i / 3;
        const int x = (int)(px % W), y = (int)(px / W);
        auto at = [&](int yy, int xx) -> double {
            const long long j = ((long long)yy * W + xx) * 3 + c;
            return rgb ? (double)rgb[j] / 255.0 : in[j];              // img_as_float of a uint8 image
        };
        double tmp = at(y, x) * (radius ? fw[radius] : 1.0);          // radius 0 (sigma = 0): skimage skips the filter
        for (int jj = -radius; jj < 0; ++jj) {
            double a, b;
            if (axis == 0) { a = at(y, x); b = a; }
            else if (axis == 1) { a = at(slic_reflect(y + jj, H), x); b = at(slic_reflect(y - jj, H), x); }
            else { a = at(y, slic_reflect(x + jj, W)); b = at(y, slic_reflect(x - jj, W)); }
            tmp += (a + b) * fw[radius + jj];
        }
        out[i] = tmp;
    }
}
// skimage rgb2lab (sRGB, D65, 2 degree observer) * ratio, rounded to 2^-20 fixed point
__global__ __launch_bounds__(256) void slic_lab_kernel(const double* rgb, long long npix, double ratio, int* q) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        double v[3], xyz[3];
        for (int c = 0; c < 3; ++c) {
            const double a = rgb[i * 3 + c];
            v[c] = a > 0.04045 ? pow((a + 0.055) / 1.055, 2.4) : a / 12.92;
        }
        const double M[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
        const double white[3] = {0.95047, 1.0, 1.08883};
        for (int r = 0; r < 3; ++r) {
            double t = v[0] * M[r][0];                                 // arr @ M.T: a sequential dot product
            t += v[1] * M[r][1];
            t += v[2] * M[r][2];
            t = t / white[r];
            xyz[r] = t > 0.008856 ? cbrt(t) : 7.787 * t + 16.0 / 116.0;
        }
        const double lab[3] = {116.0 * xyz[1] - 16.0, 500.0 * (xyz[0] - xyz[1]), 200.0 * (xyz[1] - xyz[2])};
        for (int c = 0; c < 3; ++c) q[i * 3 + c] = (int)llrint(lab[c] * ratio * SLIC_FIX);
    }
}
// segs[k] = {cy, cx, c0, c1, c2, alive}
__global__ __launch_bounds__(256) void slic_assign_kernel(const int* q, int H, int W, const double* segs, int K, int step_y, int step_x,
                                                          double spatial_weight, int* labels) {
    __shared__ double sg[1024 * 6];
    const long long n = (long long)H * W;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    const int x = i < n ? (int)(i % W) : 0, y = i < n ? (int)(i / W) : 0;
    double col[3] = {0, 0, 0};
    if (i < n)
        for (int c = 0; c < 3; ++c) col[c] = (double)q[i * 3 + c] / SLIC_FIX;
    double best = 1.7976931348623157e308;
    int bk = i < n ? labels[i] : 0;
    for (int k0 = 0; k0 < K; k0 += 1024) {                            // 1024 centres (48 KB) per LDS round
        const int kn = min(1024, K - k0);
        __syncthreads();
        for (int j = threadIdx.x; j < kn * 6; j += 256) sg[j] = segs[(size_t)k0 * 6 + j];
        __syncthreads();
        if (i >= n) continue;
        for (int k = 0; k < kn; ++k) {
            const double* s = sg + k * 6;
            if (s[5] == 0.0) continue;
            const double cy = s[0], cx = s[1];
            const long long y_min = (long long)fmax(cy - 2 * step_y, 0.0), y_max = (long long)fmin(cy + 2 * step_y + 1, (double)H);
            const long long x_min = (long long)fmax(cx - 2 * step_x, 0.0), x_max = (long long)fmin(cx + 2 * step_x + 1, (double)W);
            if (y < y_min || y >= y_max || x < x_min || x >= x_max) continue;
            const double dy = (cy - y) * (cy - y);                      // dz = 0: depth 1, centre z = 0
            const double dx = (cx - x) * (cx - x);
            double d = (0.0 + dy + dx) * spatial_weight;
            double dc = 0.0;
            for (int c = 0; c < 3; ++c) dc += (col[c] - s[2 + c]) * (col[c] - s[2 + c]);
            d += dc;
            if (best > d) { best = d; bk = k0 + k; }
        }
    }
    if (i < n) labels[i] = bk;
}
__global__ __launch_bounds__(256) void slic_sum_kernel(const int* q, int H, int W, const int* labels, unsigned long long* sums) {
    const long long n = (long long)H * W;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned long long* s = sums + (size_t)labels[i] * 6;
        atomicAdd(&s[0], 1ull);
        atomicAdd(&s[1], (unsigned long long)(i / W));
        atomicAdd(&s[2], (unsigned long long)(i % W));
        for (int c = 0; c < 3; ++c) atomicAdd(&s[3 + c], (unsigned long long)(long long)q[i * 3 + c]);   // two's complement: exact signed sums
    }
}
__global__ void slic_update_kernel(const unsigned long long* sums, double* segs, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const long long cnt = (long long)sums[k * 6];
    double* s = segs + (size_t)k * 6;
    if (cnt == 0) { s[5] = 0.0; return; }
    s[0] = (double)(long long)sums[k * 6 + 1] / (double)cnt;
    s[1] = (double)(long long)sums[k * 6 + 2] / (double)cnt;
    for (int c = 0; c < 3; ++c) s[2 + c] = ((double)(long long)sums[k * 6 + 3 + c] / SLIC_FIX) / (double)cnt;
}

size_t wsi_slic_scratch_bytes_impl(int H, int W, int K) {
    if (H <= 0 || W <= 0 || K <= 0 || K > SLIC_MAXK) return 0;
    const size_t npix = (size_t)H * W;
    return 2 * npix * 3 * sizeof(double) + npix * 3 * sizeof(int) + (size_t)K * 6 * sizeof(unsigned long long) + 256;
}
// rgb (H, W, 3) u8; fw: the 2 radius + 1 Gaussian weights (device, float64; radius 0 = no filter); segs (K, 6) float64 in / out;
// labels (H, W) int32 out
int wsi_slic_dispatch(const uint8_t* rgb, int H, int W, const double* fw, int radius, double* segs, int K, int step_y, int step_x,
                      double step, double compactness, int iters, int* labels, void* scratch, hipStream_t st) {
    if (!rgb || !segs || !labels || !scratch || wsi_slic_scratch_bytes_impl(H, W, K) == 0 || iters < 1 || radius < 0 || (radius && !fw) ||
        step_y < 1 || step_x < 1 || !(step > 0) || !(compactness > 0)) return WSI_EINVAL;
    const long long npix = (long long)H * W;
    double* a = (double*)scratch;
    double* b = a + npix * 3;
    int* q = (int*)(b + npix * 3);
    unsigned long long* sums = (unsigned long long*)(((uintptr_t)(q + npix * 3) + 255) & ~(uintptr_t)255);
    const int g3 = (int)((npix * 3 + 255) / 256 > 65535 * 16 ? 65535 * 16 : (npix * 3 + 255) / 256), g1 = (int)((npix + 255) / 256);
    if (radius) {
        hipLaunchKernelGGL(slic_blur_kernel, dim3(g3), dim3(256), 0, st, (const double*)nullptr, rgb, a, H, W, fw, radius, 0);
        hipLaunchKernelGGL(slic_blur_kernel, dim3(g3), dim3(256), 0, st, (const double*)a, (const uint8_t*)nullptr, b, H, W, fw, radius, 1);
        hipLaunchKernelGGL(slic_blur_kernel, dim3(g3), dim3(256), 0, st, (const double*)b, (const uint8_t*)nullptr, a, H, W, fw, radius, 2);
    } else {                                                           // no filter: a = img_as_float(rgb)
        hipLaunchKernelGGL(slic_blur_kernel, dim3(g3), dim3(256), 0, st, (const double*)nullptr, rgb, a, H, W, fw, 0, 1);
    }
    hipLaunchKernelGGL(slic_lab_kernel, dim3(g1 > 65535 * 16 ? 65535 * 16 : g1), dim3(256), 0, st, (const double*)a, npix, 1.0 / compactness, q);
    if (hipMemsetAsync(labels, 0, (size_t)npix * sizeof(int), st) != hipSuccess) return WSI_EFAULT;
    const double spatial_weight = 1.0 / (step * step);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(slic_assign_kernel, dim3(g1), dim3(256), 0, st, (const int*)q, H, W, (const double*)segs, K, step_y, step_x, spatial_weight, labels);
        if (hipMemsetAsync(sums, 0, (size_t)K * 6 * sizeof(unsigned long long), st) != hipSuccess) return WSI_EFAULT;
        hipLaunchKernelGGL(slic_sum_kernel, dim3(g1 > 4096 ? 4096 : g1), dim3(256), 0, st, (const int*)q, H, W, (const int*)labels, sums);
        hipLaunchKernelGGL(slic_update_kernel, dim3((K + 255) / 256), dim3(256), 0, st, (const unsigned long long*)sums, segs, K);
    }
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
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

// scratch: npix int32 + one u64 (zeroed here)
int wsi_lab_mask_dispatch(const uint8_t* rgb, long long npix, int stride, double mu_percent, uint8_t* mask, void* scratch, hipStream_t st) {
    if (!rgb || !mask || !scratch || npix <= 0 || stride < 3) return WSI_EINVAL;
    unsigned long long* sum = (unsigned long long*)scratch;
    int* aq = (int*)(sum + 2);
    if (hipMemsetAsync(sum, 0, 16, st) != hipSuccess) return WSI_EFAULT;
    const int g = grid_for(npix) > 2048 ? 2048 : grid_for(npix);
    hipLaunchKernelGGL(lab_a_kernel, dim3(g), dim3(256), 0, st, rgb, npix, stride, aq, sum);
    hipLaunchKernelGGL(lab_thresh_kernel, dim3(g), dim3(256), 0, st, (const int*)aq, npix, (const unsigned long long*)sum, mu_percent, mask);
    return LAUNCH_OK();
}
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
int wsi_cc_dispatch(const uint8_t* mask, int H, int W, int* labels_out, int* count_out, void* scratch, hipStream_t st, int conn4) {
    if (H <= 0 || W <= 0 || (long long)H * W > 0x7ffffff0LL) return WSI_EINVAL;
    const long long n = (long long)H * W, nb = (n + 1023) / 1024;
    int* L = (int*)scratch;
    int *flags = L + n, *excl = flags + n, *bsum = excl + n, *misc = bsum + nb;
    const int g = grid_for(n);
    hipLaunchKernelGGL(cc_init_kernel, dim3(g), dim3(256), 0, st, mask, L, n);
    // union-find with atomic hooks converges in one sweep for most images; sweep until a pass changes nothing (bounded)
    for (int it = 0; it < 64; ++it) {
        if (hipMemsetAsync(misc, 0, sizeof(int), st) != hipSuccess) return WSI_EFAULT;
        hipLaunchKernelGGL(cc_merge_kernel, dim3(g), dim3(256), 0, st, L, H, W, misc, conn4);
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

__global__ __launch_bounds__(256) void invert_mask_kernel(const uint8_t* m, long long n, uint8_t* o) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) o[i] = m[i] ? 0 : 1;
}
int wsi_fill_holes_dispatch(const uint8_t* mask, int H, int W, uint8_t* out, void* scratch, hipStream_t st) {
    if (!mask || !out || !scratch || H <= 0 || W <= 0) return WSI_EINVAL;
    const long long n = (long long)H * W;
    char* p = (char*)scratch;
    void* cc = p; p += (wsi_cc_scratch_bytes(H, W) + 255) / 256 * 256;
    int* labels = (int*)p; p += (n + 2) * sizeof(int);
    int* touch = (int*)p; p += (n + 2) * sizeof(int);
    uint8_t* inv = (uint8_t*)p;
    const int g = grid_for(n);
    hipLaunchKernelGGL(invert_mask_kernel, dim3(g), dim3(256), 0, st, mask, n, inv);
    const int rc = wsi_cc_dispatch(inv, H, W, labels, nullptr, cc, st, 1);
    if (rc) return rc;
    if (hipMemsetAsync(touch, 0, (size_t)(n + 2) * sizeof(int), st) != hipSuccess) return WSI_EFAULT;
    hipLaunchKernelGGL(holes_touch_kernel, dim3((2 * (H + W) + 255) / 256), dim3(256), 0, st, (const int*)labels, H, W, touch);
    hipLaunchKernelGGL(holes_fill_kernel, dim3(g), dim3(256), 0, st, mask, (const int*)labels, (const int*)touch, n, out);
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
