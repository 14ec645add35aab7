// Fused tail of the U-Net decoder (parity mode, planes = 2): the LAST decoder block - nearest x2 upsample, 3x3 conv + BN + ReLU,
// 3x3 conv + BN + ReLU on the full-resolution map - and the 1x1 head, in ONE kernel.  Replaces, for the dense 'seg' path
// (/root/reference/utils/eval.py:199-200, model.decoder = smp UnetDecoder: the fifth DecoderBlock + the segmentation head), three
// launches that r04-r05 measured at 837 + 825 + 232 us per 128 tiles of 256 x 256 against ~180 us of matrix work: with 32 -> 16 ->
// 16 channels on 8.4 M pixels per batch the unfused kernels move 1 GB tensors through HBM four times and run one workgroup per CU
// in load -> compute -> store order (profiles/r05_seg_parity_one_batch_trace.txt).
//
// Data flow of a workgroup (one band of map rows of one tile image, eight waves for 256-wide maps):
//   x4 (N, h, w, 32 ch; PF lines)  --LDS-DMA, one low-resolution row per step-->  ring L (4 rows, swizzled slab image)
//   conv1: the 3x3 conv on the UPSAMPLED map, as a polyphase filter on the low-resolution rows.  Output pixel (2y + py, 2x + px)
//     reads upsampled rows 2y + py - 1 .. 2y + py + 1 = low rows {y - 1, y, y} (py = 0) or {y, y, y + 1} (py = 1): the three
//     row taps collapse to two taps with SUMMED weights, likewise the columns.  One MFMA tile = 32 low-resolution columns of one
//     output row; its 32 A rows are the 16 output channels at px = 0 and the 16 at px = 1 (three column taps -1, 0, +1 with the
//     unused one zero), so 6 taps x 6 MFMAs make 64 output pixels - 18 per 32 pixels where the unfused kernel spends 54 (nine
//     taps, half its 32 A rows padding).  Weights are summed in float64 on the host (wsi_unet_tail_prepack) before the fp16-pair
//     split: the same conv up to the rounding of the fp32 weights, NOT the same bits as the unfused kernels
//     (tests/test_gpu_unet.py compares both with the fp32 specification and with each other).
//   conv1 epilogue: BN bias, ReLU, fp16-pair split (the value every parity-mode tensor stores)  --> ring M (4 full-resolution
//     rows, 64 bytes per pixel: [hi ch 0-7 | hi ch 8-15 | lo ch 0-7 | lo ch 8-15], never written to HBM)
//   conv2: one MFMA tile = 32 columns of TWO output rows (A rows 0-15: row 2k - 1, rows 16-31: row 2k), 4 input rows x 3 columns
//     = 12 taps x 3 MFMAs (K = 16 real channels: one k-step per plane product) - 18 per 32 pixels instead of 54.
//   head: relu(bn(conv2)) rounded to the fp16 pair like the stored tensor would be, 1x1 conv to <= 4 classes as fp32 FMAs, the
//     two half-sums of a pixel combined by v_permlane32_swap, fp32 logits stored in NCHW (128-byte runs per class row).
// Step k of a band: [A(k): conv1 rows 2k, 2k+1] barrier [B(k): conv2 + head rows 2k-1, 2k] barrier; the low row A(k+1) needs is
// requested at the start of step k.  HBM traffic per tile image: 2.1 MB of x4 in, 1 MB of logits out (the unfused path: ~17 MB).
#include <hip/hip_runtime.h>
#include "conv_dev.h"

struct TailArgs {
    const char* in;            // x4: PF tensor (N, h, w, 32 channels), planes 2
    PFGeom gl;                 // its geometry
    const char* blob;          // wsi_unet_tail_prepack
    float* out;                // logits (N, classes, 2h, 2w) fp32
    int classes, bands, rows_per_band;
};

#define TAIL_W1_BYTES (2 * 6 * 4 * 1024)
#define TAIL_W2_BYTES (12 * 2 * 1024)
#define TAIL_F_OFF (TAIL_W1_BYTES + TAIL_W2_BYTES)      // floats: wsi1[16] b1[16] wsi2[16] b2[16] head_w[4][16] head_b[4]

static __device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {      // a's lanes 32-63 <-> b's lanes 0-31
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
static __device__ __forceinline__ float pair_sum(float x) {                    // x(lane) + x(lane ^ 32)
    float y = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
static __device__ __forceinline__ unsigned pack_f16(_Float16 a, _Float16 b) {
    const f16x2 v = {a, b};
    return __builtin_bit_cast(unsigned, v);
}

template <int NW>                                              // waves per workgroup = 2 * w / 32 (w = low-resolution width)
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void unet_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = a.gl.W, hl = a.gl.H, W = 2 * w, H = 2 * hl;
    const int LROW = (w + 2) * 128, MROW = (W + 2) * 64;
    char* const L = smem;
    char* const M = smem + 4 * LROW;
    float* const fl = (float*)(smem + 4 * LROW + 4 * MROW);   // scales, biases, head (TAIL_F_OFF): 132 floats
    const int n = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
    const int k0 = band * a.rows_per_band, k1 = k0 + a.rows_per_band;

    for (int i = tid * 16; i < 4 * MROW; i += NT * 16) *(u32x4*)(M + i) = u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 132; i += NT) fl[i] = ((const float*)(a.blob + TAIL_F_OFF))[i];

    const int py = wave & 1, ct = wave >> 1;                   // conv1: this wave's output-row parity and low-resolution column tile
    bf16x8 w1[6][4];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int f = 0; f < 4; ++f) w1[t][f] = *(const bf16x8*)(a.blob + ((py * 6 + t) * 4 + f) * 1024 + lane * 16);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.blob + TAIL_W1_BYTES), 0, TAIL_W2_BYTES, 0x00020000);
    const size_t in_bytes = (size_t)pf_alloc_pixels(a.gl.N, hl, w) * 128;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)min(in_bytes, (size_t)0x7fffffff), 0x00020000);

    // low-resolution row r (-1 .. hl: pad rows included) of image n -> ring slot r & 3: w + 2 pixels from the pad left of x = 0
    auto dma_row = [&](int r) {
        const int q0 = a.gl.G + n * a.gl.S + r * a.gl.P - 1;
        const unsigned dst = lds_addr_of(L + (r & 3) * LROW);
        for (int i0 = wave * 64; i0 < (w + 2) * 8; i0 += NT) {
            const int i = i0 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
            if (i < (w + 2) * 8) dma16_buf_asm(xrs, dst + i0 * 16, (q0 + Pl) * 128 + sl * 16, 0);
        }
    };
    dma_row(k0 - 1);                                           // A(k0 - 1) runs its py = 1 waves only: low rows k0 - 1, k0
    dma_row(k0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int k = k0 - 1; k <= k1; ++k) {
        if (k + 1 <= k1 && k + 1 >= 0 && k + 1 < hl) dma_row(k + 2);     // A(k + 1) reads low rows k .. k + 2
        // ---------------------------------------------------------------- A(k): conv1 rows 2k (py = 0 waves), 2k + 1 (py = 1 waves)
        if (k >= 0 && k < hl) {
            if (ct * 32 < w && !(k == k0 - 1 && py == 0) && !(k == k1 && py == 1)) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const int Pl0 = 1 + 32 * ct + l31;
                bf16x8 xf[2][4];
                auto xload = [&](bf16x8(&x)[4], int t) {       // tap t = 3a + (ox + 1): low row k + py - 1 + a, column + ox
                    const int r = k + py - 1 + t / 3, Pl = Pl0 + t % 3 - 1;
                    const int off = (r & 3) * LROW + lds_xbase(Pl, h);
#pragma unroll
                    for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(smem + (off ^ (f << 5)));
                };
                xload(xf[0], 0);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    if (t < 5) xload(xf[(t + 1) & 1], t + 1);
                    mfma_step<2>(acc, w1[t], xf[t & 1]);
                }
                // epilogue: rows 8g + 4h + i of the tile = channel 8(g & 1) + 4h + i at px = g >> 1
                const int mrow = 4 * LROW + ((2 * k + py) & 3) * MROW;
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    unsigned hp[2][2], lp[2][2];               // [g][dword]: channels 4h .. 4h+3 (g = 0) and 8 + 4h .. (g = 1), two per dword
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        _Float16 hi[4], lo[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int c = 8 * g + 4 * h + i;
                            float v = acc[4 * (2 * px + g) + i] * fl[c] + fl[16 + c];
                            v = __builtin_amdgcn_fmed3f(v, 0.f, 65504.f);
                            hi[i] = (_Float16)v;
                            lo[i] = (_Float16)(v - (float)hi[i]);
                        }
                        hp[g][0] = pack_f16(hi[0], hi[1]); hp[g][1] = pack_f16(hi[2], hi[3]);
                        lp[g][0] = pack_f16(lo[0], lo[1]); lp[g][1] = pack_f16(lo[2], lo[3]);
                    }
                    // lanes h = 0 give their channels 8-11 for the partner's 4-7: h = 0 then holds channels 0-7, h = 1 channels 8-15
                    swap32(hp[0][0], hp[1][0]); swap32(hp[0][1], hp[1][1]); swap32(lp[0][0], lp[1][0]); swap32(lp[0][1], lp[1][1]);
                    const int Xb = 1 + 2 * (32 * ct + l31) + px;
                    const int off = mrow + Xb * 64 + ((h ^ ((Xb >> 2) & 3)) << 4);
                    *(u32x4*)(smem + off) = u32x4{hp[0][0], hp[0][1], hp[1][0], hp[1][1]};
                    *(u32x4*)(smem + (off ^ 32)) = u32x4{lp[0][0], lp[0][1], lp[1][0], lp[1][1]};
                }
            }
        } else if (k >= hl) {                                  // rows 2hl, 2hl + 1: the zero padding below the map (k < 0: the ring starts zeroed)
            for (int r = 0; r < 2; ++r) {
                char* mrow = M + ((2 * k + r) & 3) * MROW;
                for (int i = tid * 16; i < MROW; i += NT * 16) *(u32x4*)(mrow + i) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---------------------------------------------------------------- B(k): conv2 + head, rows 2k - 1 and 2k, columns 32 wave ..
        if (k >= k0) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            bf16x8 wq[3][2], xq[2][2];
            auto wload = [&](bf16x8(&wv)[2], int t) {
                wv[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane * 16, t * 2048, 0));
                wv[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, lane * 16 + 1024, t * 2048, 0));
            };
            auto xload = [&](bf16x8(&x)[2], int t) {           // tap t = 3r + dx: conv1 row 2k - 2 + r, column X + dx - 1
                const int Xb = 32 * wave + l31 + t % 3;
                const int off = 4 * LROW + ((2 * k - 2 + t / 3) & 3) * MROW + Xb * 64 + ((h ^ ((Xb >> 2) & 3)) << 4);
                x[0] = *(const bf16x8*)(smem + off);
                x[1] = *(const bf16x8*)(smem + (off ^ 32));
            };
            wload(wq[0], 0);
            xload(xq[0], 0);
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                if (t < 11) { wload(wq[(t + 1) % 3], t + 1); xload(xq[(t + 1) & 1], t + 1); }
                acc = mfma16<2>(wq[t % 3][1], xq[t & 1][0], acc);      // lo x hi, hi x lo, hi x hi (conv_dev.h mfma_step<2>)
                acc = mfma16<2>(wq[t % 3][0], xq[t & 1][1], acc);
                acc = mfma16<2>(wq[t % 3][0], xq[t & 1][0], acc);
            }
            float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};      // head partial sums of row 2k - 1 / row 2k
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 8 * (g & 1) + 4 * h + i;
                    float v = acc[4 * g + i] * fl[32 + c] + fl[48 + c];
                    v = __builtin_amdgcn_fmed3f(v, 0.f, 65504.f);
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const float act = (float)hi + (float)lo;   // the value the stored tensor would hold
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        if (g < 2) s0[c4] = fmaf(fl[64 + c4 * 16 + c], act, s0[c4]);
                        else s1[c4] = fmaf(fl[64 + c4 * 16 + c], act, s1[c4]);
                    }
                }
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                s0[c4] = pair_sum(s0[c4]);
                s1[c4] = pair_sum(s1[c4]);
            }
            const int Y = 2 * k - 1 + h, X = 32 * wave + l31;  // lanes h = 0 store row 2k - 1, lanes h = 1 row 2k
            if (Y >= 2 * k0 && Y < 2 * k1) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
                    if (c4 < a.classes) a.out[(((size_t)n * a.classes + c4) * H + Y) * W + X] = (h ? s1[c4] : s0[c4]) + fl[128 + c4];
            }
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------------------------------------------
// Second form (r05, the default for maps 64, 128 and 256 wide): the same arithmetic with the waves SPECIALISED - half of them run conv1
// only, half conv2 + head only, one of each per SIMD - so each role keeps its weights in registers for the whole band (no weight
// fetch per step; the first form streamed conv2's 24 KB per step and wave) and the two roles overlap: while the conv1 waves are in
// their MFMA loop the conv2 waves convert / exchange / store, and the other way round.  The first form's counters (profiles/
// r05_pmc_tail_v1.txt): matrix pipe busy 38 % of the kernel, waves waiting 41 % of their life, ~7 VALU instructions per MFMA.
// Interval t: conv1 waves A(t) = conv1 rows 2t (py = 0 waves), 2t + 1 (py = 1 waves) || conv2 waves B(t - 1) = output rows 2t - 3,
// 2t - 2 from conv1 rows 2t - 4 .. 2t - 1; ONE barrier per interval.  Ring M has five rows: row 2t + 1 takes the slot of row 2t - 4,
// which B(t - 1) reads in its first three taps only - the conv2 waves count those reads off in an LDS word right after issuing them
// (a wave's LDS operations execute in order) and a py = 1 wave polls that word before its first write of the interval (the reads are
// ~2000 cycles old by then; the poll is bounded, so a protocol error could corrupt a band but never hang the queue).
#ifdef WSI_STUDY
__device__ unsigned long long g_tail_stamps[16];               // [0] reports; conv1 wave 0: [1] MFMA loops [2] epilogue + writes [3] poll [4] barrier wait;
#define TSTAMP(...) __VA_ARGS__                                //              conv2 wave 0: [5] MFMA loop [6] epilogue + stores [7] barrier wait; [8] kernel cycles
extern "C" int wsi_study_tail_stamps(unsigned long long* out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tail_stamps), sizeof(g_tail_stamps)) != hipSuccess) return WSI_EFAULT;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_tail_stamps), z, sizeof(z)) != hipSuccess) return WSI_EFAULT;
    }
    return WSI_OK;
}
#else
#define TSTAMP(...)
#endif
template <int WT>                                              // low-resolution width / 32: 1, 2 or 4
__global__ __launch_bounds__(WT == 4 ? 512 : 256, WT == 4 ? 2 : 1) void unet_tail2_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int TPW = WT >= 2 ? 2 : 1;                       // MFMA tiles per wave and interval
    constexpr int NC1 = 2 * WT / TPW, NC2 = 2 * WT / TPW, NT = (NC1 + NC2) * 64;
    constexpr int w = 32 * WT, W = 2 * w, LROW = (w + 2) * 128, MROW = (W + 2) * 64, MOFF = 4 * LROW, FOFF = MOFF + 5 * MROW;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hl = a.gl.H, H = 2 * hl;
    int* const flag = (int*)(smem + FOFF);
    TSTAMP(const unsigned long long ts_begin = __builtin_readcyclecounter(); unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;)
    const int n = blockIdx.x / a.bands, band = blockIdx.x % a.bands;
    const int k0 = band * a.rows_per_band, k1 = k0 + a.rows_per_band;
    const float* const bf = (const float*)(a.blob + TAIL_F_OFF);

    for (int i = tid * 16; i < 5 * MROW; i += NT * 16) *(u32x4*)(smem + MOFF + i) = u32x4{0u, 0u, 0u, 0u};
    if (tid == 0) *flag = 0;
    // head weights as the conv2 lanes read them: [h][channel slot 4(g & 1) + i of channel 8(g & 1) + 4h + i][class] - one 16-byte read per channel
    f32x4* const hwl = (f32x4*)(smem + FOFF + 128);
    if (tid < 16) {
        const int hh = tid >> 3, sl = tid & 7, c = 8 * (sl >> 2) + 4 * hh + (sl & 3);
        hwl[tid] = f32x4{bf[64 + c], bf[80 + c], bf[96 + c], bf[112 + c]};
    }
    const size_t in_bytes = (size_t)pf_alloc_pixels(a.gl.N, hl, w) * 128;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)min(in_bytes, (size_t)0x7fffffff), 0x00020000);
    auto dma_row = [&](int r, int wv, int nwv) {               // low row r -> ring slot r & 3, by waves 0 .. nwv - 1 (wv = this wave's index)
        const int q0 = a.gl.G + n * a.gl.S + r * a.gl.P - 1;
        const unsigned dst = lds_addr_of(smem + (r & 3) * LROW);
        for (int i0 = wv * 64; i0 < (w + 2) * 8; i0 += nwv * 64) {
            const int i = i0 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
            if (i < (w + 2) * 8) dma16_buf_asm(xrs, dst + i0 * 16, (q0 + Pl) * 128 + sl * 16, 0);
        }
    };
    dma_row(k0 - 1, wave, NC1 + NC2);
    dma_row(k0, wave, NC1 + NC2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (wave < NC1) {
        // ============================================================ conv1 waves
        const int py = wave & 1, ctb = (wave >> 1) * TPW;
        bf16x8 w1[6][4];
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int f = 0; f < 4; ++f) w1[t][f] = *(const bf16x8*)(a.blob + ((py * 6 + t) * 4 + f) * 1024 + lane * 16);
        f32x2 sc[4], bi[4];                                    // channels 8g + 4h + {0,1}, {2,3}: index 2g + pair
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                sc[2 * g + q] = f32x2{bf[8 * g + 4 * h + 2 * q], bf[8 * g + 4 * h + 2 * q + 1]};
                bi[2 * g + q] = f32x2{bf[16 + 8 * g + 4 * h + 2 * q], bf[16 + 8 * g + 4 * h + 2 * q + 1]};
            }
        for (int t = k0 - 1; t <= k1 + 1; ++t) {
            if (t + 1 <= k1 && t + 1 >= 0 && t + 1 < hl) dma_row(t + 2, wave, NC1);       // A(t + 1) reads low rows t .. t + 2
            if (t <= k1 && !(t == k0 - 1 && py == 0) && !(t == k1 && py == 1)) {
                const bool inside = t >= 0 && t < hl;          // else: the zero rows above / below the map
                const int mrow = MOFF + ((2 * t + py + 10) % 5) * MROW;
                f32x16 acc[TPW];
                TSTAMP(tq = __builtin_readcyclecounter();)
#pragma unroll
                for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
                if (inside) {
                    bf16x8 xf[2][TPW][4];
                    auto xload = [&](bf16x8(&x)[4], int ti, int tp) {      // tap tp = 3a + (ox + 1): low row t + py - 1 + a, column + ox
                        const int r = t + py - 1 + tp / 3, Pl = 1 + 32 * (ctb + ti) + l31 + tp % 3 - 1;
                        const int off = (r & 3) * LROW + lds_xbase(Pl, h);
#pragma unroll
                        for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(smem + (off ^ (f << 5)));
                    };
#pragma unroll
                    for (int ti = 0; ti < TPW; ++ti) xload(xf[0][ti], ti, 0);
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp) {
                        if (tp < 5) {
#pragma unroll
                            for (int ti = 0; ti < TPW; ++ti) xload(xf[(tp + 1) & 1][ti], ti, tp + 1);
                        }
                        // the six MFMAs of mfma_step<2> (lo x hi, hi x lo, hi x hi per k-step), the tiles alternating MFMA by MFMA
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            const int wi = m < 2 ? 2 + m : m < 4 ? m - 2 : m - 4, xi = m < 2 ? m : m < 4 ? m : m - 4;
#pragma unroll
                            for (int ti = 0; ti < TPW; ++ti) acc[ti] = mfma16<2>(w1[tp][wi], xf[tp & 1][ti][xi], acc[ti]);
                        }
                    }
                }
                TSTAMP(asm volatile("s_nop 0" : "+v"(acc[0])); ts[1] += __builtin_readcyclecounter() - tq;)
                if (py == 1) {                                 // row 2t + 1 takes the slot of row 2t - 4: wait for B(t - 1)'s reads of it (the conv2
                    const int need = NC2 * (t - k0);           // waves request them right after their epilogue of B(t - 2))
                    TSTAMP(const unsigned long long tp0 = __builtin_readcyclecounter();)
                    for (int spin = 0; spin < (1 << 20) && *(volatile int*)flag < need; ++spin) __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                    TSTAMP(ts[3] += __builtin_readcyclecounter() - tp0;)
                }
                TSTAMP(tq = __builtin_readcyclecounter();)
#pragma unroll
                for (int ti = 0; ti < TPW; ++ti) {
                    const int ct = ctb + ti;
                    unsigned hp[2][2][2], lp[2][2][2];         // [px][g][dword]
#pragma unroll
                    for (int px = 0; px < 2; ++px)
#pragma unroll
                        for (int g = 0; g < 2; ++g)
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int r = 4 * (2 * px + g) + 2 * q;
                                f32x2 v = f32x2{acc[ti][r], acc[ti][r + 1]} * sc[2 * g + q] + bi[2 * g + q];
                                v[0] = inside ? __builtin_amdgcn_fmed3f(v[0], 0.f, 65504.f) : 0.f;
                                v[1] = inside ? __builtin_amdgcn_fmed3f(v[1], 0.f, 65504.f) : 0.f;
                                const f16x2 hh = __builtin_convertvector(v, f16x2);
                                const f32x2 lo = v - __builtin_convertvector(hh, f32x2);
                                hp[px][g][q] = __builtin_bit_cast(unsigned, hh);
                                lp[px][g][q] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, f16x2));
                            }
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        // lanes h = 0 give their channels 8-11 for the partner's 4-7: h = 0 then holds channels 0-7, h = 1 channels 8-15
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
                                     "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
                                     : "+v"(hp[px][0][0]), "+v"(hp[px][0][1]), "+v"(lp[px][0][0]), "+v"(lp[px][0][1]),
                                       "+v"(hp[px][1][0]), "+v"(hp[px][1][1]), "+v"(lp[px][1][0]), "+v"(lp[px][1][1]));
                        const int Xb = 1 + 2 * (32 * ct + l31) + px;
                        const int off = mrow + Xb * 64 + ((h ^ ((Xb >> 2) & 3)) << 4);
                        *(u32x4*)(smem + off) = u32x4{hp[px][0][0], hp[px][0][1], hp[px][1][0], hp[px][1][1]};
                        *(u32x4*)(smem + (off ^ 32)) = u32x4{lp[px][0][0], lp[px][0][1], lp[px][1][0], lp[px][1][1]};
                    }
                }
                TSTAMP(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts[2] += __builtin_readcyclecounter() - tq;)
            }
            TSTAMP(tq = __builtin_readcyclecounter();)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            TSTAMP(ts[4] += __builtin_readcyclecounter() - tq;)
        }
#ifdef WSI_STUDY
        if (wave == 1 && lane == 0 && (blockIdx.x & 63) == 0) {       // a py = 1 wave: it polls
            atomicAdd(&g_tail_stamps[0], 1ull);
            for (int i = 1; i <= 4; ++i) atomicAdd(&g_tail_stamps[i], ts[i]);
            atomicAdd(&g_tail_stamps[8], __builtin_readcyclecounter() - ts_begin);
        }
#endif
    } else {
        // ============================================================ conv2 + head waves
        const int dwv = wave - NC1;
        bf16x8 w2[12][2];
#pragma unroll
        for (int t = 0; t < 12; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) w2[t][p] = *(const bf16x8*)(a.blob + TAIL_W1_BYTES + (t * 2 + p) * 1024 + lane * 16);
        f32x2 sc[4], bi[4];
        float hb[4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                sc[2 * g + q] = f32x2{bf[32 + 8 * g + 4 * h + 2 * q], bf[32 + 8 * g + 4 * h + 2 * q + 1]};
                bi[2 * g + q] = f32x2{bf[48 + 8 * g + 4 * h + 2 * q], bf[48 + 8 * g + 4 * h + 2 * q + 1]};
            }
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) hb[c4] = bf[128 + c4];
        // Both roles open an interval with their MFMA loops and close it with their epilogues.  Measured alternatives (r05 stamps,
        // profiles/r05_tail_stamps.txt): this wave's epilogue moved to the start of the NEXT interval, against the conv1 waves' MFMA
        // loops, takes 2.3x as long there (vector instructions of one wave barely issue while the SIMD's other wave keeps the matrix
        // pipe saturated) and delays the read counter the py = 1 waves poll: 313 us against 289 us per 128 tiles.
        f32x16 acc[TPW];
        auto epilogue = [&](int j) {
#pragma unroll
            for (int ti = 0; ti < TPW; ++ti) {
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};      // head partial sums per class: row 2j - 1 / row 2j
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x2 v = f32x2{acc[ti][4 * g + 2 * q], acc[ti][4 * g + 2 * q + 1]} * sc[2 * (g & 1) + q] + bi[2 * (g & 1) + q];
                        v[0] = __builtin_amdgcn_fmed3f(v[0], 0.f, 65504.f);      // ReLU; the head reads the fp32 value (the three-launch path
                        v[1] = __builtin_amdgcn_fmed3f(v[1], 0.f, 65504.f);      // rounds it to the fp16 pair in between: 2^-22 relative)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f32x4 hv = hwl[8 * h + 4 * (g & 1) + 2 * q + e];
                            const f32x4 vv = {v[e], v[e], v[e], v[e]};
                            if (g < 2) s0 = __builtin_elementwise_fma(hv, vv, s0);      // (v_pk_fma_f32: two classes per instruction)
                            else s1 = __builtin_elementwise_fma(hv, vv, s1);
                        }
                    }
                {                                              // s(lane) + s(lane ^ 32): eight v_permlane32_swap in one block
                    float x0 = s0[0], x1 = s0[1], x2 = s0[2], x3 = s0[3], x4 = s1[0], x5 = s1[1], x6 = s1[2], x7 = s1[3];
                    float y0 = x0, y1 = x1, y2 = x2, y3 = x3, y4 = x4, y5 = x5, y6 = x6, y7 = x7;
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\t"
                                 "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
                                 "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15\n\ts_nop 1"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),
                                   "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
                    s0 = f32x4{x0 + y0, x1 + y1, x2 + y2, x3 + y3};
                    s1 = f32x4{x4 + y4, x5 + y5, x6 + y6, x7 + y7};
                }
                const int Y = 2 * j - 1 + h, X = 32 * (dwv * TPW + ti) + l31;      // lanes h = 0 store row 2j - 1, lanes h = 1 row 2j
                if (Y >= 2 * k0 && Y < 2 * k1) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
                        if (c4 < a.classes) a.out[(((size_t)n * a.classes + c4) * H + Y) * W + X] = (h ? s1[c4] : s0[c4]) + hb[c4];
                }
            }
        };
        for (int t = k0 - 1; t <= k1 + 1; ++t) {
            const int j = t - 1;                               // B(j): output rows 2j - 1, 2j
            if (j >= k0 && j <= k1) {
#pragma unroll
                for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
                auto xload = [&](bf16x8(&x)[2], int ti, int tp) {      // tap tp = 3r + dx: conv1 row 2j - 2 + r, column X + dx - 1
                    const int Xb = 32 * (dwv * TPW + ti) + l31 + tp % 3;
                    const int off = MOFF + ((2 * j - 2 + tp / 3 + 10) % 5) * MROW + Xb * 64 + ((h ^ ((Xb >> 2) & 3)) << 4);
                    x[0] = *(const bf16x8*)(smem + off);
                    x[1] = *(const bf16x8*)(smem + (off ^ 32));
                };
                // the wave's tiles advance together (independent accumulator chains), pixel fragments are requested TWO taps ahead
                // (a tap is 3 MFMAs per tile: one tap of distance does not cover the LDS latency)
                bf16x8 xq[3][TPW][2];
                TSTAMP(tq = __builtin_readcyclecounter();)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                    for (int ti = 0; ti < TPW; ++ti) xload(xq[tp][ti], ti, tp);
#pragma unroll
                for (int tp = 0; tp < 12; ++tp) {
                    if (tp + 2 < 12) {
#pragma unroll
                        for (int ti = 0; ti < TPW; ++ti) xload(xq[(tp + 2) % 3][ti], ti, tp + 2);
                    }
                    if (tp == 0 && lane == 0)                  // taps 0-2 (conv1 row 2j - 2) are requested: count them off behind the reads
                        __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // lo x hi, hi x lo, hi x hi (conv_dev.h mfma_step<2>), the tiles alternating MFMA by MFMA: a dependent MFMA waits out
                    // its predecessor's passes, two chains hide that
#pragma unroll
                    for (int ti = 0; ti < TPW; ++ti) acc[ti] = mfma16<2>(w2[tp][1], xq[tp % 3][ti][0], acc[ti]);
#pragma unroll
                    for (int ti = 0; ti < TPW; ++ti) acc[ti] = mfma16<2>(w2[tp][0], xq[tp % 3][ti][1], acc[ti]);
#pragma unroll
                    for (int ti = 0; ti < TPW; ++ti) acc[ti] = mfma16<2>(w2[tp][0], xq[tp % 3][ti][0], acc[ti]);
                }
                TSTAMP(asm volatile("s_nop 0" : "+v"(acc[0])); ts[5] += __builtin_readcyclecounter() - tq;)
                TSTAMP(tq = __builtin_readcyclecounter();)
                epilogue(j);
                TSTAMP(ts[6] += __builtin_readcyclecounter() - tq;)
            }
            TSTAMP(tq = __builtin_readcyclecounter();)
            __syncthreads();
            TSTAMP(ts[7] += __builtin_readcyclecounter() - tq;)
        }
#ifdef WSI_STUDY
        if (wave == NC1 && lane == 0 && (blockIdx.x & 63) == 0)
            for (int i = 5; i <= 7; ++i) atomicAdd(&g_tail_stamps[i], ts[i]);
#endif
    }
}

int g_unet_tail_form = 2;                                      // A/B: wsi_conv_set_mode +4194304 -> the first form (unet_tail_kernel)

size_t wsi_unet_tail_lds_bytes(int w) { return (size_t)4 * (w + 2) * 128 + (size_t)4 * (2 * w + 2) * 64 + 132 * 4; }
size_t wsi_unet_tail2_lds_bytes(int w) { return (size_t)4 * (w + 2) * 128 + (size_t)5 * (2 * w + 2) * 64 + 128 + 256; }

int wsi_unet_tail_dispatch(const void* x4, const void* blob, int n, int h, int w, int classes, float* logits, hipStream_t st) {
    if (!x4 || !blob || !logits || n <= 0 || h <= 0 || w % 32 || w < 32 || w > 128 || classes < 1 || classes > 4) return WSI_EINVAL;
    if ((size_t)pf_alloc_pixels(n, h, w) * 128 > (size_t)0x7fffffff) return WSI_EINVAL;     // 32-bit buffer offsets
    TailArgs a;
    a.in = (const char*)x4; a.gl = pf_geom(n, h, w, 32); a.blob = (const char*)blob; a.out = logits; a.classes = classes;
    // Bands per image: one workgroup per CU runs a whole band, so the grid should fill whole rounds of the chip's CUs (528 images in
    // one band each = three rounds on 256 CUs, the last one 6 % full) while every band pays two overlap steps: pick the power of two
    // that maximises (fill of the last round) x rows / (rows + 2), bands of at least 8 low-resolution rows
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    int bands = 1;
    double best = 0.0;
    for (int b = 1; h % b == 0 && h / b >= 8; b *= 2) {
        const long long wgs = (long long)n * b, rounds = (wgs + cus - 1) / cus;
        const double score = (double)wgs / (double)(rounds * cus) * (double)(h / b) / (double)(h / b + 2);
        if (score > best * 1.02) { best = score; bands = b; }      // (ties and near-ties: the fewer bands)
    }
    a.bands = bands; a.rows_per_band = h / bands;
    if (g_unet_tail_form == 2 && (w == 32 || w == 64 || w == 128)) {
        const size_t lds2 = wsi_unet_tail2_lds_bytes(w);
#define TAIL2_LAUNCH(WT)                                                                                                             \
    do {                                                                                                                            \
        auto kfn = unet_tail2_kernel<WT>;                                                                                           \
        if (lds2 > 64 * 1024 && hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) \
            return WSI_EINVAL;                                                                                                      \
        hipLaunchKernelGGL(kfn, dim3(n * bands), dim3(WT == 4 ? 512 : 256), lds2, st, a);                                           \
    } while (0)
        if (w == 32) TAIL2_LAUNCH(1);
        else if (w == 64) TAIL2_LAUNCH(2);
        else TAIL2_LAUNCH(4);
#undef TAIL2_LAUNCH
        return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
    }
    const size_t lds = wsi_unet_tail_lds_bytes(w);
#define TAIL_LAUNCH(NW)                                                                                                              \
    do {                                                                                                                            \
        auto kfn = unet_tail_kernel<NW>;                                                                                            \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return WSI_EINVAL;                                                                                                      \
        hipLaunchKernelGGL(kfn, dim3(n * bands), dim3(NW * 64), lds, st, a);                                                        \
    } while (0)
    switch (w / 32) {
        case 1: TAIL_LAUNCH(2); break;
        case 2: TAIL_LAUNCH(4); break;
        case 3: TAIL_LAUNCH(6); break;
        case 4: TAIL_LAUNCH(8); break;
        default: return WSI_EINVAL;
    }
#undef TAIL_LAUNCH
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}
