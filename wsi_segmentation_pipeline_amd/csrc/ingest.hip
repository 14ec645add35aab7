// Slide ingestion in front of the tile producer (reference utils/dataset.py:171-185: scan.read_region(...).convert('RGB'),
// optional image.resize((tile_w, tile_h)), ToTensor + Normalize).
//   wsi_ring      : pinned-host slots -> device staging -> unpack (RGBA|RGB rows -> the HBM-resident RGB level) on a copy stream
//                   of its own, so the decoder threads, the H2D copies and the trunk on the compute stream overlap
//   resample      : Pillow's two-pass BICUBIC resize of u8 tiles, bit-exact (spec: oracle/resize_oracle.py, pinned against the
//                   installed Pillow): integer taps with 22 fractional bits, horizontal pass then vertical, u8 between passes
#include "common.h"
#include <cmath>
#include <vector>

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT)
static int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

// ------------------------------------------------------------------------------------------ unpack
// rows of `ch`-byte pixels (ch = 3 | 4, alpha dropped like PIL's convert('RGB')) -> packed RGB rows of the level image
__global__ __launch_bounds__(256) void unpack_rows_kernel(const uint8_t* src, long long src_pitch, int ch, int rows, int width,
                                                          uint8_t* dst, long long dst_pitch) {
    const long long total = (long long)rows * width;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % width), y = (int)(i / width);
        const uint8_t* s = src + (size_t)y * src_pitch + (size_t)x * ch;
        uint8_t* d = dst + (size_t)y * dst_pitch + (size_t)x * 3;
        if (ch == 4) {
            const uint32_t v = *(const uint32_t*)s;                       // slots and pitches are 4-byte aligned for RGBA
            d[0] = (uint8_t)v; d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)(v >> 16);
        } else { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
    }
}

struct wsi_ring {
    int slots = 0, device = 0;
    size_t slot_bytes = 0;
    std::vector<void*> host, dev;
    std::vector<hipEvent_t> done;
    std::vector<char> busy;
    hipStream_t copy = nullptr;
    hipEvent_t fence = nullptr;
};

int wsi_ring_create_impl(wsi_ring** out, int slots, size_t slot_bytes) {
    if (!out || slots < 1 || slots > 64 || slot_bytes == 0 || (slot_bytes & 3)) return WSI_EINVAL;
    wsi_ring* r = new wsi_ring();
    r->slots = slots; r->slot_bytes = slot_bytes;
    bool ok = hipGetDevice(&r->device) == hipSuccess && hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&r->fence, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < slots; ++i) {
        void *h = nullptr, *d = nullptr;
        hipEvent_t e = nullptr;
        ok = hipHostMalloc(&h, slot_bytes, hipHostMallocDefault) == hipSuccess && hipMalloc(&d, slot_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        r->host.push_back(h); r->dev.push_back(d); r->done.push_back(e); r->busy.push_back(0);
    }
    if (!ok) {
        for (void* h : r->host) if (h) (void)hipHostFree(h);
        for (void* d : r->dev) if (d) (void)hipFree(d);
        for (hipEvent_t e : r->done) if (e) (void)hipEventDestroy(e);
        if (r->fence) (void)hipEventDestroy(r->fence);
        if (r->copy) (void)hipStreamDestroy(r->copy);
        delete r;
        return WSI_EFAULT;
    }
    *out = r;
    return WSI_OK;
}

void* wsi_ring_host_slot_impl(wsi_ring* r, int slot) { return (r && slot >= 0 && slot < r->slots) ? r->host[slot] : nullptr; }

int wsi_ring_wait_slot_impl(wsi_ring* r, int slot) {
    if (!r || slot < 0 || slot >= r->slots) return WSI_EINVAL;
    if (r->busy[slot]) {
        if (hipEventSynchronize(r->done[slot]) != hipSuccess) return WSI_EFAULT;
        r->busy[slot] = 0;
    }
    return WSI_OK;
}

int wsi_ring_submit_impl(wsi_ring* r, int slot, int rows, int width, int channels, long long src_pitch, uint8_t* dst,
                         long long dst_pitch) {
    if (!r || slot < 0 || slot >= r->slots || rows <= 0 || width <= 0 || (channels != 3 && channels != 4) || !dst) return WSI_EINVAL;
    if (src_pitch < (long long)width * channels || dst_pitch < (long long)width * 3 || (channels == 4 && (src_pitch & 3))) return WSI_EINVAL;
    const size_t bytes = (size_t)rows * (size_t)src_pitch;
    if (bytes > r->slot_bytes || r->busy[slot]) return WSI_EINVAL;
    if (hipMemcpyAsync(r->dev[slot], r->host[slot], bytes, hipMemcpyHostToDevice, r->copy) != hipSuccess) return WSI_EFAULT;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid_for((long long)rows * width)), dim3(256), 0, r->copy,
                       (const uint8_t*)r->dev[slot], src_pitch, channels, rows, width, dst, dst_pitch);
    if (hipGetLastError() != hipSuccess || hipEventRecord(r->done[slot], r->copy) != hipSuccess) return WSI_EFAULT;
    r->busy[slot] = 1;
    return WSI_OK;
}

int wsi_ring_fence_impl(wsi_ring* r, hipStream_t compute) {
    if (!r) return WSI_EINVAL;
    if (hipEventRecord(r->fence, r->copy) != hipSuccess || hipStreamWaitEvent(compute, r->fence, 0) != hipSuccess) return WSI_EFAULT;
    return WSI_OK;
}

// The copy stream writes destination memory the caller allocated on its compute stream: a caching allocator may hand out a
// block whose last readers (kernels of the previous slide) are still queued THERE.  acquire orders the ring's copies after
// everything enqueued on `compute` so far (an event wait, no host block).
int wsi_ring_acquire_impl(wsi_ring* r, hipStream_t compute) {
    if (!r) return WSI_EINVAL;
    if (hipEventRecord(r->fence, compute) != hipSuccess || hipStreamWaitEvent(r->copy, r->fence, 0) != hipSuccess) return WSI_EFAULT;
    return WSI_OK;
}
int wsi_ring_device_impl(const wsi_ring* r) { return r ? r->device : -1; }

int wsi_ring_drain_impl(wsi_ring* r) {
    if (!r) return WSI_EINVAL;
    if (hipStreamSynchronize(r->copy) != hipSuccess) return WSI_EFAULT;
    for (auto& b : r->busy) b = 0;
    return WSI_OK;
}

void wsi_ring_destroy_impl(wsi_ring* r) {
    if (!r) return;
    (void)hipStreamSynchronize(r->copy);
    for (void* h : r->host) (void)hipHostFree(h);
    for (void* d : r->dev) (void)hipFree(d);
    for (hipEvent_t e : r->done) (void)hipEventDestroy(e);
    (void)hipEventDestroy(r->fence);
    (void)hipStreamDestroy(r->copy);
    delete r;
}

// ------------------------------------------------------------------------------------------ Pillow BICUBIC resample
#define RS_BITS 22
struct RsAxis { int in = 0, out = 0, ksize = 0; int* bounds = nullptr; int* kk = nullptr; };   // device arrays
struct wsi_resample_plan { RsAxis h, v; };

static double bicubic_w(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static int rs_axis(RsAxis& ax, int in, int out) {
    double scale = (double)in / (double)out, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    std::vector<int> bounds((size_t)out * 2), kk((size_t)out * ksize, 0);
    std::vector<double> w(ksize);
    for (int xx = 0; xx < out; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in) xmax = in;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = bicubic_w((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (double)(1 << RS_BITS)) : (int)(0.5 + v * (double)(1 << RS_BITS));
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    ax.in = in; ax.out = out; ax.ksize = ksize;
    if (hipMalloc((void**)&ax.bounds, bounds.size() * sizeof(int)) != hipSuccess || hipMalloc((void**)&ax.kk, kk.size() * sizeof(int)) != hipSuccess)
        return WSI_EFAULT;
    if (hipMemcpy(ax.bounds, bounds.data(), bounds.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ax.kk, kk.data(), kk.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return WSI_EFAULT;
    return WSI_OK;
}

void wsi_resample_plan_destroy_impl(wsi_resample_plan* p) {
    if (!p) return;
    for (RsAxis* ax : {&p->h, &p->v}) { if (ax->bounds) (void)hipFree(ax->bounds); if (ax->kk) (void)hipFree(ax->kk); }
    delete p;
}
int wsi_resample_plan_create_impl(wsi_resample_plan** out, int in_h, int in_w, int out_h, int out_w) {
    if (!out || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return WSI_EINVAL;
    wsi_resample_plan* p = new wsi_resample_plan();
    if (rs_axis(p->h, in_w, out_w) != WSI_OK || rs_axis(p->v, in_h, out_h) != WSI_OK) { wsi_resample_plan_destroy_impl(p); return WSI_EFAULT; }
    *out = p;
    return WSI_OK;
}

static __device__ inline uint8_t rs_clip(int acc) {
    const int v = acc >> RS_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// horizontal pass: tile n read from the slide at origins[n] (out-of-slide pixels are 0, like the tile gather) -> tmp (n, ph, tw, 3)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* slide, long long pitch, int SH, int SW, const int* origins, int N,
                                                         int ph, int tw, const int* bounds, const int* kk, int ksize, uint8_t* tmp) {
    const long long total = (long long)N * ph * tw;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % tw);
        long long p = i / tw;
        const int y = (int)(p % ph), n = (int)(p / ph);
        const int sy = origins[2 * n + 1] + y, sx0 = origins[2 * n] + bounds[2 * ox], cnt = bounds[2 * ox + 1];
        const int* k = kk + (size_t)ox * ksize;
        int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
        if (sy >= 0 && sy < SH) {
            const uint8_t* row = slide + (size_t)sy * pitch;
            for (int t = 0; t < cnt; ++t) {
                const int sx = sx0 + t;
                if (sx < 0 || sx >= SW) continue;
                const uint8_t* s = row + (size_t)sx * 3;
                const int c = k[t];
                a0 += s[0] * c; a1 += s[1] * c; a2 += s[2] * c;
            }
        }
        uint8_t* d = tmp + (size_t)i * 3;
        d[0] = rs_clip(a0); d[1] = rs_clip(a1); d[2] = rs_clip(a2);
    }
}
// vertical pass: tmp (n, ph, tw, 3) -> out (n, th, tw, 3)
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* tmp, int N, int ph, int th, int tw, const int* bounds, const int* kk,
                                                         int ksize, uint8_t* out) {
    const long long total = (long long)N * th * tw;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % tw);
        long long p = i / tw;
        const int oy = (int)(p % th), n = (int)(p / th);
        const int y0 = bounds[2 * oy], cnt = bounds[2 * oy + 1];
        const int* k = kk + (size_t)oy * ksize;
        const uint8_t* s = tmp + (((size_t)n * ph + y0) * tw + x) * 3;
        int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
        for (int t = 0; t < cnt; ++t, s += (size_t)tw * 3) {
            const int c = k[t];
            a0 += s[0] * c; a1 += s[1] * c; a2 += s[2] * c;
        }
        uint8_t* d = out + (size_t)i * 3;
        d[0] = rs_clip(a0); d[1] = rs_clip(a1); d[2] = rs_clip(a2);
    }
}

size_t wsi_resample_scratch_bytes_impl(const wsi_resample_plan* p, int n) {
    return (!p || n <= 0) ? 0 : (size_t)n * p->v.in * p->h.out * 3;
}
int wsi_resample_tiles_impl(const wsi_resample_plan* p, const uint8_t* slide, long long pitch, int SH, int SW, const int* origins, int N,
                            uint8_t* out, void* scratch, hipStream_t st) {
    if (!p || !slide || !origins || !out || N <= 0 || SH <= 0 || SW <= 0) return WSI_EINVAL;
    const int ph = p->v.in, th = p->v.out, tw = p->h.out;
    const bool need_h = p->h.in != p->h.out, need_v = ph != th;
    if (need_v && !scratch) return WSI_EINVAL;
    uint8_t* tmp = need_v ? (uint8_t*)scratch : out;
    // the horizontal kernel also performs the gather from the slide (with identity taps when the width is unchanged:
    // bicubic at scale 1 has the single tap 1.0 at offset 0 in fixed point, exactly like Pillow skipping the pass)
    (void)need_h;
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((long long)N * ph * tw)), dim3(256), 0, st, slide, pitch, SH, SW, origins, N, ph, tw,
                       (const int*)p->h.bounds, (const int*)p->h.kk, p->h.ksize, tmp);
    if (need_v)
        hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for((long long)N * th * tw)), dim3(256), 0, st, (const uint8_t*)tmp, N, ph, th, tw,
                           (const int*)p->v.bounds, (const int*)p->v.kk, p->v.ksize, out);
    return LAUNCH_OK();
}
