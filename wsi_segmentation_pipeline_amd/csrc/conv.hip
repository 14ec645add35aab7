// 3x3 / 1x1 convolution + folded BatchNorm + optional residual + ReLU on gfx950 MFMA.
// Replaces the torch.nn ops of BasicBlock.forward (/root/reference/resnets_shift.py:49-65) and of
// the downsample branch (resnets_shift.py:173-177) in eval mode.
//
// GEMM orientation: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel]   (k = tap x cin)
//   A operand = weights  (row = cout  = lane&31, 8 consecutive cin per lane half)
//   B operand = pixels   (col = pixel = lane&31, the same 8 cin)
// so an accumulator register group holds 4 consecutive output channels of ONE pixel and the
// epilogue stores 8-byte channel runs into the pixel's 128-byte line.
//
// Precision: PLANES=2 multiplies bf16 hi/lo splits in three MFMA passes (hi*hi + hi*lo + lo*hi,
// fp32 accumulate): ~2^-16 relative operand error, needed for the 1e-3 logit contract
// (single-pass bf16 is 1.9e-2 off).  PLANES=1 is the single-pass bf16
// speed mode.
#include "conv_dev.h"

// --------------------------------------------------------------------------------------------
// Generic gather kernel (3x3 stride 2, 1x1 stride 2, also stride 1): per (line, tap) step the BM
// input pixels are gathered into an LDS tile by per-lane source addresses.  Pad output positions
// gather pixel 0 (always a zero guard) and are dropped by the epilogue.
template <int MT, int WM, int WN, int PLANES>
__global__ __launch_bounds__(WM* WN * 64) void conv_gather_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = WM * MT * 32;
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int PPT = BM * 8 / NTHREADS;          // 16-byte pieces per thread per step
    constexpr int PIX_STEP = NTHREADS / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int nblocks = a.go.C / (WN * 32);
    const int nb = blockIdx.x % nblocks;
    const int mtile = blockIdx.x / nblocks;
    const int q0 = a.go.G + mtile * BM;
    const int ntile = nb * WN + wn;
    const int NC = a.gi.C / PFmt<PLANES>::CPL;
    const int KS = a.ksize, NT = KS * KS, pad = KS / 2;
    const size_t in_pixstride = (size_t)a.gi.C * PFmt<PLANES>::BPC;
    const bf16x8* wbase = (const bf16x8*)a.wpk + (size_t)ntile * NC * NT * 4 * 64 + lane;

    // input pixel (tap 0,0) of each output pixel this thread stages
    int inq[PPT];
    const int sp = tid & 7;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int m = (tid >> 3) + k * PIX_STEP;
        const int q = q0 + m;
        int r = q - a.go.G;
        int v = -1;
        if (r >= 0 && r < a.go.NS) {
            const int n = r / a.go.S;
            r -= n * a.go.S;
            const int y = r / a.go.P, x = r - y * a.go.P;
            if (x != a.go.W && y != a.go.H)
                v = a.gi.G + n * a.gi.S + (y * a.stride - pad) * a.gi.P + (x * a.stride - pad);
        }
        inq[k] = v;
    }

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    int xbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xbase[mt] = lds_xbase(wm * MT * 32 + mt * 32 + l31, h);

    for (int c = 0; c < NC; ++c) {
        for (int t = 0; t < NT; ++t) {
            const int toff = (t / KS) * a.gi.P + (t % KS);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int m = (tid >> 3) + k * PIX_STEP;
                const int s = sp ^ ((m >> 1) & 7);
                const int src = inq[k] < 0 ? 0 : inq[k] + toff;
                *(uint4*)(smem + (size_t)(m * 8 + sp) * 16) =
                    *(const uint4*)((const char*)a.in + (size_t)src * in_pixstride + c * 128 + s * 16);
            }
            bf16x8 wf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) wf[f] = wbase[((size_t)(c * NT + t) * 4 + f) * 64];
            __syncthreads();
            mfma_line<MT, PLANES>(acc, wf, smem, xbase);
        }
    }
    conv_epilogue<MT, PLANES>(a, acc, q0 + wm * MT * 32, ntile, lane);
}

// --------------------------------------------------------------------------------------------
// Study builds: where a wave of the wide kernel spends its cycles (tools/wide_stamps.py).  s_memtime stamps around the
// phases, summed over the waves of a launch: [0] waves, [1] lifetime, [2] tile set-up, [3] first line's slab + weights
// (prologue), [4] later lines' slab waits, [5] tap-end waits (weight DMA + barrier), [6] tail (extra K segment, residual,
// encode, stores).  The multiply phases are the rest.
#ifdef WSI_STUDY
__device__ unsigned long long g_wide_stamps[8];
#define WSTAMP(...) __VA_ARGS__
extern "C" int wsi_study_wide_stamps(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wide_stamps), sizeof(g_wide_stamps)) != hipSuccess) return WSI_EFAULT;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_wide_stamps), z, sizeof(z)) != hipSuccess) return WSI_EFAULT;
    }
    return WSI_OK;
}
#else
#define WSTAMP(...)
#endif

// --------------------------------------------------------------------------------------------
// Slab kernel, software-pipelined form.  Per 128-byte line the nine taps are straight-line code:
//   * weight fragments come through buffer loads (scalar offset per tap, no address VALU) into a
//     3-slot register ring indexed statically (tap % 3): no copies, prefetch distance one tap;
//   * pixel fragments are requested one (tap, m-tile) step ahead - including across taps - so the
//     LDS latency always hides behind six MFMAs;
//   * a compiler scheduling fence per tap keeps hipcc from hoisting all nine taps' loads.

// DENSE: the tile enumerates REAL pixels only (n, y, x raster order) instead of consecutive PF
// positions, so no MFMA work is spent on pad positions ((H+1)(W+1)/HW = 27 % at 8x8 maps); the
// slab is still the contiguous range from the first to the last pixel's neighbourhood and each
// lane simply carries its own slab offset.
// ABL: compile-time ablation for bottleneck studies (tools/tune_conv.py cfg 50-53): 8 = no MFMA, 16 = no weight
// prefetch (stale registers), 32 = no pixel-fragment LDS reads (stale registers).  0 in every shipped configuration.
// PAIR (launcher: dense tiles of a map whose width is a multiple of 64, whole tiles only): pixel tile 2j+1 of a wave lies 32
// pixels to the right of tile 2j in the SAME map row, so its fragments sit exactly 4096 bytes behind tile 2j's in the slab
// (the slot swizzle has period 16 pixels) - its four LDS reads reuse the even tile's address registers with an immediate
// offset and cost no address arithmetic (r03: layer 1 spent ~650 of its ~1650 vector instructions per wave on these addresses).
template <int MT, int WM, int WN, int PLANES, int MINW, bool DENSE, int ABL = 0, bool PAIR = false>
__global__ __launch_bounds__(WM* WN * 64, MINW) void conv3x3s1_slab3_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = WM * MT * 32;
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int RESID_NBUF = MT < 2 ? MT : 2;               // residual tiles per batch and wave (mode 3; launch_slab3 sizes the LDS)
    constexpr bool SCHED = MINW < 3;                          // (the 168-register three-waves-per-SIMD form spills with the fences: it keeps the compiler's order)
    const int tid = threadIdx.x, lane = tid & 63;
    if (CONV_STUDY(a, CONV_ABL_DISPATCH_ONLY)) return;         // study builds: dispatch cost only
    WSTAMP(const unsigned long long st_begin = __builtin_readcyclecounter(); unsigned long long st_pro = 0, st_line = 0, st_tap = 0, st_a = 0;)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int nblocks = a.go.C / (WN * 32);
    // Workgroup id -> (pixel tile, channel block).  Ids are dealt round-robin to the 8 XCDs (own L2 each): with
    // bit 512 the channel blocks of one pixel tile get ids 8 apart, i.e. the SAME XCD, so the slab they share is
    // fetched from HBM once instead of once per channel block.
    int nb = blockIdx.x % nblocks, mtile = blockIdx.x / nblocks;
    if (a.flags & CONV_XCD_ORDER) {
        const int per = 8 * nblocks, r = blockIdx.x % per;
        nb = r >> 3;
        mtile = (blockIdx.x / per) * 8 + (r & 7);
        const int mtiles = DENSE ? (a.gi.N * a.gi.H * a.gi.W + BM - 1) / BM : (a.gi.NS + BM - 1) / BM;
        if (mtile >= mtiles) return;
    } else if (a.flags & CONV_XCD_RANGES) {
        // XCD-contiguous ranges: XCD x (= id & 7) walks tiles [x*chunk, (x+1)*chunk) in dispatch order, so the halo rows
        // two neighbouring pixel tiles share are still in THAT XCD's L2 when the second one asks for them
        const int chunk = gridDim.x >> 3, lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        nb = lin % nblocks;
        mtile = lin / nblocks;
        const int mtiles = DENSE ? (a.gi.N * a.gi.H * a.gi.W + BM - 1) / BM : (a.gi.NS + BM - 1) / BM;
        if (mtile >= mtiles) return;
    }
    const int P = a.gi.P;
    const int ntile = nb * WN + wn;
    const int NC = CONV_STUDY(a, CONV_ABL_NO_MAINLOOP) ? 0 : a.gi.C / PFmt<PLANES>::CPL;  // study builds: epilogue only
    const bool in96 = PLANES == 3 && (a.flags & CONV_IN96);   // 96-byte input lines (common.h): line-planar in memory, 128-byte lines in LDS
    const int ncup = a.in_up ? a.up_c / PFmt<PLANES>::CPL : 0;    // leading lines that come from the half-size tensor (U-Net decoder)
    const size_t in_pixstride = in96 ? (size_t)96 : (size_t)(a.gi.C - (a.in_up ? a.up_c : 0)) * PFmt<PLANES>::BPC;
    const int in_line = in96 ? 0 : 128;                       // (96-byte lines: the line's plane is part of the resource base)
    int xoff[MT], qs[MT];                                     // slab-local pixel / PF position of each tile row
    bool valid[MT];
    int slab0, npieces;
    if constexpr (DENSE) {
        const int HW = a.gi.H * a.gi.W, R = a.gi.N * HW;
        auto pos = [&](int i) { return pf_pos_of_index(a.gi, i); };
        const int i0 = mtile * BM, i1 = min(i0 + BM, R) - 1;
        slab0 = pos(i0) - P - 1;
        npieces = (pos(i1) + P + 1 - slab0 + 1) * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int i = i0 + wm * MT * 32 + mt * 32 + (a.gi.W < 32 ? dense_lane_pixel(l31) : l31);   // (conv_dev.h: LDS bank conflicts)
            valid[mt] = i < R;
            qs[mt] = pos(valid[mt] ? i : i1);
            xoff[mt] = qs[mt] - slab0 - (P + 1);             // tap (0,0) adds toff relative to q - P - 1
        }
    } else {
        const int q0 = a.gi.G + mtile * BM;
        slab0 = q0 - P - 1;
        npieces = (BM + 2 * P + 2) * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            xoff[mt] = wm * MT * 32 + mt * 32 + l31;
            qs[mt] = q0 + xoff[mt];
            valid[mt] = pf_is_pixel(a.go, qs[mt]);
        }
    }
    static_assert((NTHREADS / 8) % 16 == 0, "whole swizzle periods per DMA round");
    const size_t slab_byte0 = (size_t)slab0 * in_pixstride;
    const size_t in_bytes = (size_t)pf_alloc_pixels(a.gi.N, a.gi.H, a.gi.W) * in_pixstride;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.in + slab_byte0), 0, a.in ? (int)min(in_bytes - slab_byte0, (size_t)0x7fffffff) : 0, 0x00020000);
    int xvoff, upl, usl;
    bool xact = true;                                         // 96-byte lines: the lanes of the two hi6 slots fetch nothing
    {
        const int i = wave * 64 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
        xvoff = Pl * (int)in_pixstride + (in96 ? mx96_piece(sl) : sl) * 16;
        if (in96) xact = mx96_stored(sl);
        upl = Pl; usl = sl;
    }
    const int up_pixstride = a.up_c * PFmt<PLANES>::BPC;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.in_up, 0, a.in_up ? (int)((size_t)pf_alloc_pixels(a.gup.N, a.gup.H, a.gup.W) * up_pixstride) : 0, 0x00020000);
    // study hook (tools/tune_conv.py --wcopies): bits 10-13 of relu = number of back-to-back copies of the packed
    // weights minus one; workgroups spread over the copies (do hot weight lines serialise on few L2 channels?)
    const size_t wcopy = (size_t)(mtile % ((CONV_STUDY(a, 15 << CONV_WCOPIES_SHIFT) >> CONV_WCOPIES_SHIFT) + 1)) * (size_t)(a.go.C / 32) * NC * 9 * 4096;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wpk + wcopy + (size_t)ntile * NC * 9 * 4096), 0, NC * 9 * 4096, 0x00020000);
    const int wvoff = lane * 16;

    f32x16 acc[1][MT];                                        // [1]: the shape conv_residual_mx takes
    if constexpr (PLANES == 3) acc_init_bias<MT>(acc[0], a.bias, ntile, lane);
    else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][mt][r] = 0.f;
    }
    auto wload = [&](bf16x8(&w)[4], int soff) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
            w[f] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + f * 1024, soff, 0));
    };
    int xa[4];                                                // fragment addresses of the last even tile (PAIR)
    auto xload_m = [&](bf16x8(&x)[4], int Pl, int m) {        // m: the tile index (compile-time after unrolling)
        if (PAIR && (m & 1)) {
#pragma unroll
            for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(smem + xa[f] + 4096);
            return;
        }
        const int base = lds_xbase(Pl, h);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            xa[f] = base ^ (f << 5);
            x[f] = *(const bf16x8*)(smem + xa[f]);
        }
    };

    WSTAMP(const unsigned long long st_setup = __builtin_readcyclecounter();)
    for (int c = 0; c < NC; ++c) {
        WSTAMP(st_a = __builtin_readcyclecounter();)
        const int sline = c * 9 * 4096;
        int Pc = P;                                           // opaque per line: stops LICM from keeping all
        asm volatile("" : "+s"(Pc));                          // 36 tap/tile LDS addresses live in registers
        bf16x8 wbuf[3][4], xf[2][4];
        wload(wbuf[0], sline);                                // W(c,0) flies during the slab DMA
        if (c) __syncthreads();
        // slab pieces by buffer addressing: a round of NTHREADS pieces is NTHREADS / 8 pixels, a multiple of 16, so the swizzle
        // term ((Pl >> 1) & 7) of a lane does not depend on the round - one per-lane byte offset (computed once per tile),
        // everything else scalar (r02: the 64-bit per-piece address arithmetic was ~10 % of this kernel's vector instructions)
        if (c < ncup) {
            // line of the up-sampled tensor (ConvArgs.in_up): the piece of slab pixel Pl comes from pixel (y >> 1, x >> 1) of the
            // half-size tensor; pad / guard positions of the slab read pixel 0 of it (a zero guard).  One magic division pair per piece
            for (int i0 = wave * 64, r = 0; i0 < npieces; i0 += NTHREADS, ++r) {
                const int rel = slab0 + upl + r * (NTHREADS / 8) - a.gi.G;
                int src = 0;
                if (rel >= 0 && rel < a.gi.NS) {
                    const int n = fd_div(rel, a.gi.dS), rem = rel - n * a.gi.S;
                    const int y = fd_div(rem, a.gi.dP), x = rem - y * P;
                    if (x != a.gi.W && y != a.gi.H) src = a.gup.G + n * a.gup.S + (y >> 1) * a.gup.P + (x >> 1);
                }
                dma16_buf(urs, smem + (size_t)i0 * 16, src * up_pixstride + usl * 16, c * 128);
            }
        } else {
            const __amdgpu_buffer_rsrc_t xr = in96 ? __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (size_t)c * (size_t)a.plane96 + slab_byte0), 0,
                                                                                        (int)min(in_bytes - slab_byte0, (size_t)0x7fffffff), 0x00020000)
                                                   : xrs;
            for (int i0 = wave * 64, r = 0; i0 < npieces; i0 += NTHREADS, ++r)
                if (xact) dma16_buf(xr, smem + (size_t)i0 * 16, xvoff, (c - ncup) * in_line + r * (NTHREADS / 8) * (int)in_pixstride);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        WSTAMP(if (c) st_line += __builtin_readcyclecounter() - st_a; else st_pro = __builtin_readcyclecounter() - st_a; st_a = __builtin_readcyclecounter();)
        if (in96) {                                           // the slab has landed: rebuild its hi6 plane in place
            mx96_rebuild_hi6(smem, npieces >> 3, tid, NTHREADS);
            __syncthreads();
        }
        WSTAMP(st_tap += __builtin_readcyclecounter() - st_a;)
        xload_m(xf[0], xoff[0], 0);                           // step (t=0, mt=0): toff(0) = 0
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            asm volatile("" ::: "memory");                    // scheduling fence: keep later taps' loads below
            if (t < 8 && !(ABL & 16)) wload(wbuf[(t + 1) % 3], sline + (t + 1) * 4096);
            asm volatile("" ::: "memory");                    // ...and keep THIS prefetch above the tap's MFMAs: without
                                                              // it hipcc sinks the loads to the end of the tap (distance 0)
            const int toff = (t / 3) * Pc + (t % 3);
            const int toff_next = ((t + 1) / 3) * Pc + ((t + 1) % 3);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int k = t * MT + mt;
                if constexpr (!(ABL & 32)) {
                    if (mt + 1 < MT) xload_m(xf[(k + 1) & 1], xoff[mt + 1] + toff, mt + 1);
                    else if (t < 8) xload_m(xf[(k + 1) & 1], xoff[0] + toff_next, 0);
                }
                // r04: without this fence hipcc sinks every fragment read next to its MFMA (plain memory fences do not order MFMAs:
                // the r03 build of the layer-1 configuration held 76 `ds_read; s_waitcnt lgkmcnt(0); v_mfma` triples per line)
                if constexpr (SCHED) __builtin_amdgcn_sched_barrier(0);
                const bf16x8(&w)[4] = wbuf[t % 3];
                const bf16x8(&x)[4] = xf[k & 1];
                if constexpr (ABL & 8) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[0][mt][f] += (float)w[f][0] * (float)x[f][0];   // keeps the loads alive
                } else if constexpr (PLANES == 3) {
                    acc[0][mt] = mfma_mx6(acc[0][mt], w, x);
                } else if constexpr (PLANES == 2) {
                    mfma_step<2>(acc[0][mt], w, x);                   // lo*hi, hi*lo, hi*hi (fp16 pair)
                } else {
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[0][mt] = mfma_bf16(w[f], x[f], acc[0][mt]);
                }
                if constexpr (SCHED) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    WSTAMP(const unsigned long long st_loop_end = __builtin_readcyclecounter();)
    if constexpr (PLANES == 3) {
        if (a.resid) __syncthreads();                         // every wave is done reading pixel fragments: slab memory becomes
        conv_tail_mx<1, MT, RESID_NBUF>(a, acc, qs, valid, ntile, lane, smem + wave * (RESID_NBUF * 4096), slab0);   // the waves' residual staging
    } else {
        char* scratch = nullptr;
        if (PLANES == 2 && a.resid && !(a.flags & CONV_RESID_DIRECT)) {
            __syncthreads();
            scratch = smem + wave * 8192;
        }
        conv_epilogue_q<MT, PLANES>(a, acc[0], qs, valid, ntile, lane, scratch);
    }
#ifdef WSI_STUDY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && (blockIdx.x & 63) == 0) {                // one workgroup in 64 reports (same slots as the wide kernel; [5] = the hi6 rebuild passes here)
        const unsigned long long st_end = __builtin_readcyclecounter();
        atomicAdd(&g_wide_stamps[0], 1ull);
        atomicAdd(&g_wide_stamps[1], st_end - st_begin);
        atomicAdd(&g_wide_stamps[2], st_setup - st_begin);
        atomicAdd(&g_wide_stamps[3], st_pro);
        atomicAdd(&g_wide_stamps[4], st_line);
        atomicAdd(&g_wide_stamps[5], st_tap);
        atomicAdd(&g_wide_stamps[6], st_end - st_loop_end);
    }
#endif
}

// exact largest slab (pixels) over the dense tiles of BM real pixels
long long dense_max_slab_pixels(const ConvArgs& a, int BM) {
    const long long R = (long long)a.gi.N * a.gi.H * a.gi.W;
    const int mtiles = (int)((R + BM - 1) / BM);
    const int HW = a.gi.H * a.gi.W;
    auto pos = [&](long long i) { const long long n = i / HW, rem = i - n * HW; return (long long)a.gi.G + n * a.gi.S + (rem / a.gi.W) * a.gi.P + rem % a.gi.W; };
    auto span = [&](int m) { const long long i0 = (long long)m * BM, i1 = (i0 + BM < R ? i0 + BM : R) - 1; return pos(i1) + a.gi.P + 1 - (pos(i0) - a.gi.P - 1) + 1; };
    long long maxpix = span(mtiles - 1);
    int g = BM, r = HW;                                       // tile starts repeat (mod H*W) every H*W / gcd(BM, H*W) tiles
    while (r) { const int t = g % r; g = r; r = t; }
    const int period = HW / g + 1;
    const int scan = mtiles < period ? mtiles : period;
    for (int m = 0; m < scan; ++m) { const long long px = span(m); if (px > maxpix) maxpix = px; }
    return maxpix;
}

template <int MT, int WM, int WN, int PLANES, int MINW, bool DENSE, int ABL = 0, bool PAIR = false>
static int launch_slab3(const ConvArgs& a, hipStream_t st) {
    constexpr int BM = WM * MT * 32, NTHREADS = WM * WN * 64;
    if (a.go.C % (WN * 32)) return WSI_EINVAL;
    const int nblocks = a.go.C / (WN * 32);
    int mtiles, maxpix;
    if (DENSE) {
        const long long R = (long long)a.gi.N * a.gi.H * a.gi.W;
        mtiles = (int)((R + BM - 1) / BM);
        maxpix = (int)dense_max_slab_pixels(a, BM);          // exact (a bound of BM + pads + 2P + 2 costs a workgroup per CU at 64x64 maps)
    } else {
        mtiles = (a.gi.NS + BM - 1) / BM;
        maxpix = BM + 2 * a.gi.P + 2;
    }
    size_t lds = (size_t)((maxpix * 8 + NTHREADS - 1) / NTHREADS * NTHREADS) * 16;
    if (lds < (size_t)WM * WN * 8192) lds = (size_t)WM * WN * 8192;          // the epilogue stages residual tiles there (8 KB per wave)
    if (lds > 160 * 1024) return WSI_EINVAL;
    if (PAIR && (!DENSE || MT % 2 || a.gi.W % 64 || ((long long)a.gi.N * a.gi.H * a.gi.W) % BM)) return WSI_EINVAL;
    if (a.in_up) {                                           // fused upsample + concat input (common.h): 32-bit offsets into the half-size tensor
        constexpr int CPL = PFmt<PLANES>::CPL;
        if (a.up_c <= 0 || a.up_c % CPL || a.up_c > a.gi.C || (a.up_c < a.gi.C && !a.in) || a.gup.H * 2 != a.gi.H || a.gup.W * 2 != a.gi.W ||
            a.gup.N != a.gi.N || (a.flags & (CONV_IN96 | CONV_RESID96)) || a.in2 ||
            (size_t)pf_alloc_pixels(a.gup.N, a.gup.H, a.gup.W) * a.up_c * PFmt<PLANES>::BPC > (size_t)0x7fffffff)
            return WSI_EINVAL;
    }
    auto k = conv3x3s1_slab3_kernel<MT, WM, WN, PLANES, MINW, DENSE, ABL, PAIR>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return WSI_EINVAL;
    }
    const int grid = (a.flags & CONV_XCD_ORDER) ? (mtiles + 7) / 8 * 8 * nblocks : (a.flags & CONV_XCD_RANGES) ? (mtiles * nblocks + 7) / 8 * 8 : mtiles * nblocks;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}


// --------------------------------------------------------------------------------------------
// Row-stacked slab kernel (r04; mode 3, 64-wide maps: the trunk's 64-channel layer 1 at 256 x 256 patches).
// Same slab, same weight path (buffer loads into a register ring), same tail as conv3x3s1_slab3_kernel<4, 2, 2, 3, ..>, but a
// wave's four pixel tiles are the SAME 32 columns of four consecutive map rows instead of 128 consecutive pixels.  In the
// padded-flat layout the next map row is exactly P pixels further, so tap (dy, dx) of tile row mt reads the pixel fragments of
// slab row j = mt + dy at column shift dx: the fragment set (j, dx) serves up to three (tile, tap) steps.  Loop order per
// 32-channel line: dx outer, slab row j = 0..5, steps (mt, dy = j - mt) inner -
//   * 18 fragment sets (72 ds_read_b128) per line and wave instead of 36 (144): the r03 counters showed this kernel latency-bound
//     with nothing saturated (matrix pipe 34 %, LDS 27 %, waves waiting 45 % of their life), each step waiting for its own
//     four reads; now a set is requested while the previous one feeds up to nine MFMAs;
//   * half the LDS address arithmetic (one swizzled base per set);
//   * the three taps of a column (dy = 0..2 at one dx) sit in the three slots of the weight ring; slot dy is refilled with the
//     next column's tap right after its last use (rows 3, 4, 5), two to three steps ahead of its first use.
// Accumulation order per output is (line, dx, dy) instead of the (line, dy, dx) of every other stride-1 kernel: the results differ
// from theirs in the last bits of the fp32 sums (same products, same precision); tests compare this kernel with the fp32
// reference like the others, and with conv3x3s1_slab3_kernel to a few ulps.
// NCT: input lines known at compile time (2 = the 64-channel layer 1: the line loop unrolls), 0 = any.  IN96: 96-byte input lines
// (hi6 plane rebuilt in LDS after the slab has landed).  With 128-byte lines there is no rebuild pass whose ~45 temporaries compete
// with the weight ring for registers, so a line's first column of weights is requested across the line boundary (during the
// previous line's last column; line 0: before the slab DMA) instead of after the slab wait.
template <int MINW, int NCT = 0, bool IN96 = true>
__global__ __launch_bounds__(256, MINW) void conv3x3s1_rows_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = 4, WN = 2, BM = 256, NTHREADS = 256, RESID_NBUF = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    WSTAMP(const unsigned long long st_begin = __builtin_readcyclecounter(); unsigned long long st_pro = 0, st_line = 0, st_tap = 0, st_a = 0;)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;                 // wm: column half of the 64-wide map, wn: channel tile
    const int l31 = lane & 31, h = lane >> 5;
    const int nblocks = a.go.C / (WN * 32);
    int nb = blockIdx.x % nblocks, mtile = blockIdx.x / nblocks;
    if (a.flags & CONV_XCD_RANGES) {                          // XCD-contiguous tile ranges (see conv3x3s1_slab3_kernel)
        const int chunk = gridDim.x >> 3, lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        nb = lin % nblocks;
        mtile = lin / nblocks;
        if (mtile >= (a.gi.N * a.gi.H * a.gi.W) / BM) return;
    }
    const int P = a.gi.P;                                     // 65
    const int ntile = nb * WN + wn;
    const int NC = NCT ? NCT : a.gi.C / 32;
    constexpr bool in96 = IN96;
    const size_t in_pixstride = in96 ? (size_t)96 : (size_t)a.gi.C * 4;     // 96-byte lines are line-planar (ConvArgs.plane96)
    const int in_line = in96 ? 0 : 128;                       // (96-byte lines: the line's plane is part of the resource base)
    // tile = four whole rows of one image (the launcher checks W == 64, H % 4 == 0)
    const int p0 = pf_pos_of_index(a.gi, mtile * BM);         // first pixel of the tile's first row
    const int slab0 = p0 - P - 1;
    const int npieces = (5 * P + 66) * 8;                     // rows -1 .. 4 of the tile: p0 + 3P + 63 + P + 1 - slab0 + 1 pixels
    const int xoff0 = wm * 32 + l31;                          // slab-local pixel of tap (0, 0) of tile row 0
    const size_t slab_byte0 = (size_t)slab0 * in_pixstride;
    const size_t in_bytes = (size_t)pf_alloc_pixels(a.gi.N, a.gi.H, a.gi.W) * in_pixstride;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.in + slab_byte0), 0, (int)min(in_bytes - slab_byte0, (size_t)0x7fffffff), 0x00020000);
    int xvoff;
    bool xact = true;                                         // 96-byte lines: the lanes of the two hi6 slots fetch nothing
    {
        const int i = wave * 64 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
        xvoff = Pl * (int)in_pixstride + (in96 ? mx96_piece(sl) : sl) * 16;
        if (in96) xact = mx96_stored(sl);
    }
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wpk + (size_t)ntile * NC * 9 * 4096), 0, NC * 9 * 4096, 0x00020000);
    const int wvoff = lane * 16;

    f32x16 acc[1][MT];
    auto wload = [&](bf16x8(&w)[4], int soff) {
#pragma unroll
        for (int f = 0; f < 4; ++f)                           // fragment offset in the SCALAR operand: one address VGPR instead of four
            w[f] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, soff + f * 1024, 0));
    };
    auto xload = [&](bf16x8(&x)[4], int Pl) {
        asm volatile("" : "+v"(Pl));                          // opaque: each set's address arithmetic stays at the set (no hoisting, no spills)
        const int base = lds_xbase(Pl, h);
#pragma unroll
        for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(smem + (base ^ (f << 5)));
    };

    bf16x8 wbuf[3][4], xf[2][4];
    // packed weights: tap (dy, dx) of line c at ((c * 9) + dy * 3 + dx) * 4096
    constexpr bool XLINE = !IN96 && NCT > 0;                  // (a runtime line loop would carry the ring across its back edge: spills)
    if constexpr (XLINE) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) wload(wbuf[dy], (dy * 3) * 4096);       // line 0's first column flies during the slab DMA
    }
    WSTAMP(const unsigned long long st_setup = __builtin_readcyclecounter();)
#pragma unroll NCT ? NCT : 1
    for (int c = 0; c < (NCT ? NCT : NC); ++c) {
        WSTAMP(st_a = __builtin_readcyclecounter();)
        if (c) __syncthreads();
        {
            const __amdgpu_buffer_rsrc_t xr = in96 ? __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (size_t)c * (size_t)a.plane96 + slab_byte0), 0,
                                                                                        (int)min(in_bytes - slab_byte0, (size_t)0x7fffffff), 0x00020000)
                                                   : xrs;
            for (int i0 = wave * 64, r = 0; i0 < npieces; i0 += NTHREADS, ++r)
                if (xact) dma16_buf(xr, smem + (size_t)i0 * 16, xvoff, c * in_line + r * (NTHREADS / 8) * (int)in_pixstride);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        WSTAMP(if (c) st_line += __builtin_readcyclecounter() - st_a; else st_pro = __builtin_readcyclecounter() - st_a; st_a = __builtin_readcyclecounter();)
        if (in96) {                                           // the slab has landed: rebuild its hi6 plane in place
            mx96_rebuild_hi6(smem, npieces >> 3, tid, NTHREADS);
            __syncthreads();
        }
        WSTAMP(st_tap += __builtin_readcyclecounter() - st_a;)
        int Pc = P;                                           // opaque per line (see conv3x3s1_slab3_kernel)
        asm volatile("" : "+s"(Pc));
        const int sline = c * 9 * 4096;
        // the line's first column of weights is requested HERE, after the hi6 rebuild pass: fetched across the slab wait it would
        // be 48 more live registers beside the rebuild's ~45 temporaries and the 64 accumulators (168 = three waves per SIMD)
        if constexpr (!XLINE) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) wload(wbuf[dy], sline + (dy * 3) * 4096);
        }
        if (c == 0) acc_init_bias<MT>(acc[0], a.bias, ntile, lane);
        xload(xf[0], xoff0);                                  // set (dx = 0, j = 0)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int k = dx * 6 + j;
                // the NEXT set's four reads are in flight while this set's steps multiply: hipcc's scheduler otherwise moves the
                // MFMAs (no memory operands) up across plain fences and waits for every set right after requesting it
                if (j < 5) xload(xf[(k + 1) & 1], xoff0 + (j + 1) * Pc + dx);
                else if (dx < 2) xload(xf[(k + 1) & 1], xoff0 + dx + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int dy = j - mt;
                    if (dy >= 0 && dy < 3) acc[0][mt] = mfma_mx6(acc[0][mt], wbuf[dy], xf[k & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                // slot j - 3 has seen its last step of this column: refill it with the next column's tap (next line's first column
                // after dx = 2; nothing after the last line)
                if (j >= 3 && (dx < 2 || (XLINE && c + 1 < NC))) {
                    wload(wbuf[j - 3], dx < 2 ? sline + ((j - 3) * 3 + dx + 1) * 4096 : sline + (9 + (j - 3) * 3) * 4096);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    WSTAMP(const unsigned long long st_loop_end = __builtin_readcyclecounter();)
    int qs[MT];                                               // (computed here: four registers less across the main loop)
    bool valid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        qs[mt] = p0 + mt * P + wm * 32 + l31;
        valid[mt] = true;
    }
    if (a.resid) __syncthreads();                             // every wave is done reading pixel fragments: slab memory becomes
    conv_tail_mx<1, MT, RESID_NBUF>(a, acc, qs, valid, ntile, lane, smem + wave * (RESID_NBUF * 4096), slab0);   // the waves' residual staging
#ifdef WSI_STUDY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && (blockIdx.x & 63) == 0) {
        const unsigned long long st_end = __builtin_readcyclecounter();
        atomicAdd(&g_wide_stamps[0], 1ull);
        atomicAdd(&g_wide_stamps[1], st_end - st_begin);
        atomicAdd(&g_wide_stamps[2], st_setup - st_begin);
        atomicAdd(&g_wide_stamps[3], st_pro);
        atomicAdd(&g_wide_stamps[4], st_line);
        atomicAdd(&g_wide_stamps[5], st_tap);
        atomicAdd(&g_wide_stamps[6], st_end - st_loop_end);
    }
#endif
}

template <int MINW>
static int launch_rows(const ConvArgs& a, hipStream_t st) {
    const bool nc2 = a.gi.C == 64;
    constexpr int BM = 256, NTHREADS = 256;
    if (a.go.C % 64 || a.gi.C % 32 || a.gi.W != 64 || a.gi.H % 4 || a.in2) return WSI_EINVAL;
    const int nblocks = a.go.C / 64;
    const long long R = (long long)a.gi.N * a.gi.H * a.gi.W;
    const int mtiles = (int)(R / BM);                        // whole tiles: H % 4 == 0
    const int npix = 5 * a.gi.P + 66;
    size_t lds = (size_t)((npix * 8 + NTHREADS - 1) / NTHREADS * NTHREADS) * 16;
    if (lds < (size_t)4 * 8192) lds = (size_t)4 * 8192;      // residual staging of the tail (8 KB per wave)
    const bool i96 = a.flags & CONV_IN96;
    auto k = nc2 ? (i96 ? conv3x3s1_rows_kernel<MINW, 2, true> : conv3x3s1_rows_kernel<MINW, 2, false>)
                 : (i96 ? conv3x3s1_rows_kernel<MINW, 0, true> : conv3x3s1_rows_kernel<MINW, 0, false>);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return WSI_EINVAL;
    const int grid = (a.flags & CONV_XCD_RANGES) ? (mtiles * nblocks + 7) / 8 * 8 : mtiles * nblocks;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}


// --------------------------------------------------------------------------------------------
// Persistent, producer-fed form of the row-stacked kernel (r05; mode 3, 64-wide maps with 96-byte lines, 64 output channels: the
// trunk's layer 1).  Why: the r04 stamps and the r05 residual / no-residual comparison show conv3x3s1_rows_kernel bound by how many
// bytes a CU keeps in flight, not by a pipe - each of its three workgroups per CU runs [slab load][multiply][slab load][multiply]
// [residual round trips + encode] strictly one after the other, the three time-slice the matrix pipe (which triples the time every
// slab occupies the LDS), and 160 KB of LDS cannot hold a second slab per workgroup: ~24 KB in flight per CU, 3.0 TB/s.  Here ONE
// workgroup per CU walks its share of the tiles: four compute waves (2 column halves x 2 channel tiles, four map rows each, exactly
// the rows kernel's multiply loop) and one PRODUCER wave that issues every LDS-DMA of the workgroup - the slab of the next
// (tile, line) unit into the second slab buffer and, on a tile's last line, the sixteen residual tiles - one unit ahead of their
// use.  The compute waves issue no DMA, so their in-order vmcnt only ever tracks their own weight fetches and stores (a DMA issued
// by a compute wave would sit in front of every later weight load: the wave would wait for the slab before it may use the
// weights); the producer waits vmcnt(0) and the unit's barrier publishes what landed.  Per unit two barriers:
//   B1: slab u + 1 (and the tile's residual tiles) have landed, every compute wave is done with slab u
//       -> the producer requests slab u + 2 into the buffer of slab u; all five waves rebuild the hi6 plane of slab u + 1
//   B2: slab u + 1 is complete -> on a tile's last line the compute waves add the residual (identity MFMA), encode and store
// Residual tiles are staged in a 96-byte pitch (six stored slots per pixel; the identity step never reads the hi6 plane), 48 KB for
// the 16 tiles of a workgroup: LDS = 2 x 49 KB + 48 KB.  Same operands in the same order as conv3x3s1_rows_kernel: identical bits.
// Requires at least two input lines per tile (the staging area of tile k is refilled one unit before tile k + 1 ends).
constexpr int L1P_SLAB_PIX = 392;                             // 5 P + 66 = 391 slab pixels at P = 65, rounded to whole 16-byte slots of 8
constexpr int L1P_SLAB = L1P_SLAB_PIX * 128;
constexpr int L1P_STAGE = 16 * 3072;                          // sixteen residual tiles of 32 pixels x 96 bytes
constexpr int L1P_LDS = 2 * L1P_SLAB + L1P_STAGE;
template <int NCT>
__global__ __launch_bounds__(320, 1) void conv3x3s1_l1p_kernel(ConvArgs a) {
    static_assert(NCT >= 2 && NCT % 2 == 0, "the slab buffer of a unit is its line's parity; the staging area needs two lines per tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = 4, P = 65, NPIX = 391;
    char* const stage = smem + 2 * L1P_SLAB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave == 4;
    const int wm = (wave >> 1) & 1, wn = wave & 1;            // compute waves: column half of the 64-wide map, channel tile
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int NC = NCT;                                   // input lines (the line loop is unrolled: accumulators live inside one tile iteration)
    const int in_pixstride = 96;                              // 96-byte lines, line-planar (ConvArgs.plane96)
    // this workgroup's tiles: XCD x (= id & 7) owns a contiguous range of tiles, its workgroups take them round-robin, so the tiles in
    // flight on one XCD are neighbours (they share halo rows in that XCD's L2)
    const int mtiles = (a.gi.N * a.gi.H * a.gi.W) >> 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int chunk = (mtiles + 7) >> 3;
    const int t_lo = xcd * chunk, t_hi = min(t_lo + chunk, mtiles);
    const int ntl = t_hi - t_lo > slot ? (t_hi - t_lo - slot + per - 1) / per : 0;
    const int nunits = ntl * NC;
    if (nunits == 0) return;                                  // (uniform: the whole workgroup leaves)
    auto tile_p0 = [&](int k) { return pf_pos_of_index(a.gi, (t_lo + slot + k * per) << 8); };   // first pixel of the tile's first row

    // ---- producer: per-lane constants of its DMA patterns ----
    // slab: instruction r moves pieces 64 r .. 64 r + 63 = slab pixels 8 r .. 8 r + 7; the swizzle key ((Pl >> 1) & 7) of pixel
    // Pl = 8 r + (lane >> 3) depends on the parity of r: two per-lane offsets / activity masks
    int sv[2];
    bool sact[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int Pl = 8 * par + (lane >> 3), sl = (lane & 7) ^ ((Pl >> 1) & 7);
        sv[par] = (lane >> 3) * in_pixstride + mx96_piece(sl) * 16;
        sact[par] = mx96_stored(sl);
    }
    // residual tile: 192 pieces (32 pixels x 6 stored slots, 96-byte pitch in the staging area) in three instructions
    const int r_pixstride = 96;
    int rv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int g = 64 * k + lane;
        rv[k] = (g / 6) * r_pixstride + (g % 6) * 16;
    }
    auto dma_slab = [&](int u) {                              // unit u -> slab buffer u & 1
        const int k = u / NC, c = u - k * NC;
        const size_t byte0 = (size_t)(tile_p0(k) - P - 1) * in_pixstride;
        const size_t in_bytes = (size_t)pf_alloc_pixels(a.gi.N, a.gi.H, a.gi.W) * in_pixstride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (size_t)c * (size_t)a.plane96 + byte0), 0,
                                                                             (int)min(in_bytes - byte0, (size_t)0x7fffffff), 0x00020000);
        const unsigned dst = lds_addr_of(smem + (u & 1) * L1P_SLAB);
        for (int r = 0; r < (NPIX * 8 + 63) / 64; ++r) {
            const bool on = sact[r & 1] && (r * 64 + lane) < NPIX * 8;
            if (on) dma16_buf_asm(rs, dst + r * 1024, sv[r & 1], r * 8 * in_pixstride);
        }
    };
    auto dma_resid = [&](int k) {                             // the sixteen residual tiles of tile k -> staging area
        const int p0 = tile_p0(k);
        const size_t byte0 = (size_t)p0 * r_pixstride;
        const size_t r_bytes = (size_t)pf_alloc_pixels(a.go.N, a.go.H, a.go.W) * r_pixstride;
        for (int w = 0; w < 4; ++w) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.resid + (size_t)(w & 1) * (size_t)a.plane96 + byte0), 0,
                                                                                 (int)min(r_bytes - byte0, (size_t)0x7fffffff), 0x00020000);   // line wn's plane
            for (int mt = 0; mt < MT; ++mt) {
                const int soff = (mt * P + (w >> 1) * 32) * r_pixstride;                     // tile (wave w, row mt): pixels p0 + mt P + 32 wm + 0..31
                const unsigned dst = lds_addr_of(stage + (w * MT + mt) * 3072);
#pragma unroll
                for (int k3 = 0; k3 < 3; ++k3) dma16_buf_asm(rs, dst + k3 * 1024, rv[k3], soff);
            }
        }
    };
    auto barrier = [&]() {                                    // raw barrier: LDS traffic of this wave is complete, nothing else is drained
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- compute waves: weights, accumulators ----
    const int ntile = wn;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wpk + (size_t)ntile * NC * 9 * 4096), 0, NC * 9 * 4096, 0x00020000);
    const int wvoff = lane * 16;
    f32x16 acc[1][MT];
    bf16x8 wbuf[3][4], xf[2][4];
    auto wload = [&](bf16x8(&w)[4], int soff) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
            w[f] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, soff + f * 1024, 0));
    };
    const int xoff0 = wm * 32 + l31;                          // slab-local pixel of tap (0, 0) of tile row 0

    // ---- prologue: slab 0 (and slab 1) ----
    if (producer) {
        dma_slab(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    barrier();
    if (producer && nunits > 1) dma_slab(1);
    mx96_rebuild_hi6(smem, NPIX, tid, 320);
    barrier();

    for (int k = 0; k < ntl; ++k) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int u = k * NC + c;
        const bool last_line = c == NC - 1;
        if (producer) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slab u + 1 (and this tile's residual tiles) have landed
        } else {
            const char* const xl = smem + (c & 1) * L1P_SLAB;
            auto xload = [&](bf16x8(&x)[4], int Pl) {
                asm volatile("" : "+v"(Pl));                  // opaque: each set's address arithmetic stays at the set
                const int base = lds_xbase(Pl, h);
#pragma unroll
                for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(xl + (base ^ (f << 5)));
            };
            if (c == 0) acc_init_bias<MT>(acc[0], a.bias, ntile, lane);
            // the line's first column of weights is requested HERE, after the barriers: kept in flight across the barriers, the hi6
            // rebuild or the tail (r05 compile experiments: prefetch from the previous unit's last column, from the start of the tail,
            // from before the rebuild) the ring's 48 registers make hipcc spill weight fragments INSIDE the multiply loops (316-448
            // bytes of scratch at 256 registers; this form: 116 bytes, all in the tail)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) wload(wbuf[dy], (c * 9 + dy * 3) * 4096);
            int Pc = P;
            asm volatile("" : "+s"(Pc));
            const int sline = c * 9 * 4096;
            xload(xf[0], xoff0);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int kk = dx * 6 + j;
                    if (j < 5) xload(xf[(kk + 1) & 1], xoff0 + (j + 1) * Pc + dx);
                    else if (dx < 2) xload(xf[(kk + 1) & 1], xoff0 + dx + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int dy = j - mt;
                        if (dy >= 0 && dy < 3) acc[0][mt] = mfma_mx6(acc[0][mt], wbuf[dy], xf[kk & 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // slot j - 3 has seen its last step of this column: the next column's tap
                    if (j >= 3 && dx < 2) {
                        wload(wbuf[j - 3], sline + ((j - 3) * 3 + dx + 1) * 4096);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        barrier();                                            // B1
        if (producer) {
            if (u + 2 < nunits) dma_slab(u + 2);              // into the buffer the compute waves have just left
            if (a.resid && u + 1 < nunits && (u + 1) % NC == NC - 1) dma_resid((u + 1) / NC);   // staging is free: the previous tile's tail ended before B1
        }
        if (u + 1 < nunits) mx96_rebuild_hi6(smem + ((c + 1) & 1) * L1P_SLAB, NPIX, tid, 320);
        barrier();                                            // B2
        if (!producer && last_line) {
            const int p0 = tile_p0(k);
            int qs[MT];
            bool valid[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                qs[mt] = p0 + mt * P + wm * 32 + l31;
                valid[mt] = true;
            }
            if (a.resid) {
                // acc += residual tile through the matrix pipe (conv_dev.h conv_tail_mx: identity A operand); the tiles are resident:
                // no DMA, no waits.  Fragment f of pixel p: stored slot 2 f + h -> compact slot {0, 1, 2, 3, 4, -, 5, -}
                bf16x8 iw[4];
                {
                    int lo = l31;
                    asm volatile("" : "+v"(lo));              // opaque: built here for every tile, not kept alive across the tile loop (16 registers)
                    const int pp = mx_line_pos(lo), f0 = mx6_field_of_pos(pp);
                    f16x8 k0, k1;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        k0[jj] = (pp == 8 * h + jj) ? (_Float16)1.0f : (_Float16)0.0f;
                        k1[jj] = (pp == 16 + 8 * h + jj) ? (_Float16)1.0f : (_Float16)0.0f;
                    }
                    unsigned q[6];
#pragma unroll
                    for (int d = 0; d < 6; ++d) {
                        const int bit = 6 * f0 + 3 - 32 * d;
                        q[d] = (h == 0 && bit >= 0 && bit < 32) ? (1u << (bit & 31)) : 0u;
                    }
                    iw[0] = __builtin_bit_cast(bf16x8, k0);
                    iw[1] = __builtin_bit_cast(bf16x8, k1);
                    iw[2] = __builtin_bit_cast(bf16x8, u32x4{q[0], q[1], q[2], q[3]});
                    iw[3] = __builtin_bit_cast(bf16x8, u32x4{q[4], q[5], 127u, 0u});
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const char* t = stage + (wave * MT + mt) * 3072 + l31 * 96;
                    bf16x8 x[4];
                    x[0] = *(const bf16x8*)(t + h * 16);
                    x[1] = *(const bf16x8*)(t + (2 + h) * 16);
                    if (h == 0) {
                        x[2] = *(const bf16x8*)(t + 4 * 16);
                        x[3] = *(const bf16x8*)(t + 5 * 16);
                    } else {                                  // (the identity's lo6 half is zero, but a stale scale byte could be NaN)
                        x[2] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
                        x[3] = x[2];
                    }
                    acc[0][mt] = mfma_mx6(acc[0][mt], iw, x);
                }
            }
            conv_epilogue_mx<MT>(a, acc[0], qs, valid, ntile, lane);
        }
    }
    }
}

static int g_l1p_grid = 0;                               // workgroups of the persistent kernel: one per CU (multiple of 8)
int g_l1p = 0;                                           // A/B (wsi_conv_set_mode +1048576 ON): the persistent layer-1 kernel (cfg 42) instead of the rows kernel (cfg 40); r05: 45 % slower, off
static int launch_l1p(const ConvArgs& a, hipStream_t st) {
    if (a.go.C != 64 || a.gi.C % 32 || a.gi.C < 64 || a.gi.W != 64 || a.gi.H % 4 || a.in2 || a.in_up || !(a.flags & CONV_IN96) ||
        (a.resid && !(a.flags & CONV_RESID96)))
        return WSI_EINVAL;
    if (!g_l1p_grid) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
            return WSI_EFAULT;
        g_l1p_grid = cus / 8 * 8;
    }
    const long long mtiles = (long long)a.gi.N * a.gi.H * a.gi.W / 256;
    int grid = g_l1p_grid;
    if (mtiles < grid) grid = (int)((mtiles + 7) / 8 * 8);
    if (a.gi.C != 64) return WSI_EINVAL;                      // (two input lines: the only instantiation)
    auto k = conv3x3s1_l1p_kernel<2>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, L1P_LDS) != hipSuccess) return WSI_EINVAL;
    hipLaunchKernelGGL(k, dim3(grid), dim3(320), L1P_LDS, st, a);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}


// --------------------------------------------------------------------------------------------
// "Wide" dense slab kernel (Cout % 128 == 0): every wave owns 64 output channels x 128 pixels (2 x 4 MFMA tiles,
// 128 accumulator registers), so a pixel-fragment set read from LDS feeds SIX MFMAs instead of three and the
// operand-load instructions per MFMA drop from 1.67 to 1.0 (the r01 ablations showed the slab3 loop is bound by
// issuing and waiting for operand loads, not by MFMA, LDS bandwidth or HBM).  Workgroup = 2 x 2 waves = 256 pixels x
// 128 channels.  The weights of one (line, tap) for the workgroup's four channel tiles (16 KB) are staged ONCE per
// workgroup by LDS-DMA, double-buffered, one barrier per tap; the two waves that share a channel half read them
// from LDS, the two waves that share a pixel half read the same slab.  No VMEM load returns to registers in the
// main loop, so vmcnt only ever tracks DMA.
// General shape: WM x WN waves, each MT = 4 pixel tiles x NT channel tiles; the default (2, 2, 2) is the 256 px x 128
// couts form above, (4, 2, 1) = 512 px x 64 couts serves the 64-channel layer 1 (four waves share each weight stage).
// D8 (r05; 8 x 8 maps = the trunk's layer 4 at 256 x 256 patches, BM = 256 = four whole images): the slab image in LDS has rows of
// EIGHT pixels - the pad column of the padded-flat layout is not copied - so the 16 lanes the hardware serves together on a
// ds_read_b128 (two map rows of a dense tile, conv_dev.h dense_lane_pixel) are 16 CONSECUTIVE slab pixels for every tap and hit 16
// different bank groups.  With the 9-pixel pitch of the r01-r04 image every such group held two pixels that are equal modulo 16
// (residues 0 and 2 occur three times in a 32-pixel tile: no lane order avoids it) - 40 % of the layer-4 launches' LDS cycles were
// conflict cycles (profiles/r04_pmc).  Rows still follow each other as in memory (image rows, then the zero row two images share),
// so one DMA round of 32 slab pixels = 4 rows = 36 memory pixels: per-lane offset + scalar offset per round, as before.  What the pad
// column gave for free - zeros left of x = 0 and right of x = 7 - becomes a redirect: a lane whose tap falls outside its row reads
// one of 16 zero pixels behind the slab, the one with the residue (mod 16) of the pixel it would have read, which keeps the group
// conflict-free (3 vector instructions on six of the nine taps).
template <int PLANES, int MINW, int ABL = 0, int WM = 2, int WN = 2, int NT = 2, bool D8 = false>
__global__ __launch_bounds__(WM* WN * 64, MINW) void conv3x3s1_wide_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = 4, BM = WM * 128, NTHREADS = WM * WN * 64, NTILES = WN * NT, WB = NTILES * 4096;
    constexpr int D8_ZB = 304;                                // D8: first of the 16 zero pixels (37 rows of 8 = 296 slab pixels, next multiple of 16)
    static_assert(!D8 || (BM == 256 && NTHREADS == 256), "D8: four whole 8 x 8 images per tile, 32 slab pixels per DMA round");
    constexpr int RESID_NBUF = 4;                             // residual tiles in flight per wave (mode 3; launch_wide sizes the LDS)
    char* const wl = smem;                                    // 2 weight buffers of NTILES x 4 KB
    char* const xl = smem + 2 * WB;                           // pixel slab
    const int tid = threadIdx.x, lane = tid & 63;
    WSTAMP(const unsigned long long st_begin = __builtin_readcyclecounter(); unsigned long long st_pro = 0, st_line = 0, st_tap = 0, st_a = 0;)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int nblocks = a.go.C / (NTILES * 32);
    int nb = blockIdx.x % nblocks, mtile = blockIdx.x / nblocks;
    if (a.flags & CONV_XCD_RANGES) {                           // XCD-contiguous tile ranges (see conv3x3s1_slab3_kernel)
        const int chunk = gridDim.x >> 3, lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        nb = lin % nblocks;
        mtile = lin / nblocks;
        if (mtile >= (a.gi.N * a.gi.H * a.gi.W + BM - 1) / BM) return;
    }
    const int P = a.gi.P;
    const int NC = a.gi.C / PFmt<PLANES>::CPL;
    const size_t in_pixstride = (size_t)a.gi.C * PFmt<PLANES>::BPC;
    int xoff[MT];                                             // slab-local pixel of tap (0, 0) per tile row; the PF position is
    bool valid[MT];                                           // xoff + slab0 + P + 1 (rebuilt after the main loop: four registers less in it)
    int slab0, npieces;
    const int lpix = a.gi.W < 32 ? dense_lane_pixel(l31) : l31;      // bank-conflict-free lane order on narrow maps (conv_dev.h)
    const int Rtot = a.gi.N * a.gi.H * a.gi.W;
    if constexpr (D8) {
        // slab row r = memory row (image n0, y = -1) + r; pixel (n, y, x) of the tile sits at slab pixel 8 (1 + 9 (n - n0) + y) + x, so
        // tap (ty, tx) of it is slab pixel xoff + 8 ty + tx with xoff = 8 (9 (n - n0) + y) + x - 1
        const int i0 = mtile * BM, i1 = min(i0 + BM, Rtot) - 1, n0 = i0 >> 6;
        slab0 = a.gi.G + n0 * a.gi.S - P;                     // memory pixel of (n0, -1, 0): the zero row in front of the image (or the guard)
        npieces = (1 + 9 * ((i1 >> 6) - n0 + 1)) * 64;       // rows x 8 pixels x 8 pieces
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int i = i0 + wm * MT * 32 + mt * 32 + lpix;
            valid[mt] = i < Rtot;
            const int ii = valid[mt] ? i : i1;
            xoff[mt] = 8 * (9 * ((ii >> 6) - n0) + ((ii >> 3) & 7)) + (ii & 7) - 1;
        }
    } else {
        auto pos = [&](int i) { return pf_pos_of_index(a.gi, i); };
        const int i0 = mtile * BM, i1 = min(i0 + BM, Rtot) - 1;
        slab0 = pos(i0) - P - 1;
        npieces = (pos(i1) + P + 1 - slab0 + 1) * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int i = i0 + wm * MT * 32 + mt * 32 + lpix;
            valid[mt] = i < Rtot;
            xoff[mt] = pos(valid[mt] ? i : i1) - slab0 - (P + 1);
        }
    }
    // D8: lanes in map column 0 / 7 read a zero pixel on the taps that leave their row (the tile's 32 pixels are four whole rows:
    // the column is the same for every tile of the lane)
    const bool d8_left = D8 && (lpix & 7) == 0, d8_right = D8 && (lpix & 7) == 7;
    constexpr int RPIX = D8 ? 36 : NTHREADS / 8;              // memory pixels one DMA round advances
    // slab DMA by buffer addressing (see conv3x3s1_slab3_kernel): one per-lane byte offset, scalar offsets per line / round
    static_assert((NTHREADS / 8) % 16 == 0, "whole swizzle periods per DMA round");
    const size_t slab_byte0 = (size_t)slab0 * in_pixstride;
    const size_t in_bytes = (size_t)pf_alloc_pixels(a.gi.N, a.gi.H, a.gi.W) * in_pixstride;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.in + slab_byte0), 0, (int)min(in_bytes - slab_byte0, (size_t)0x7fffffff), 0x00020000);
    int xvoff;
    {
        const int i = wave * 64 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
        xvoff = (D8 ? (Pl >> 3) * 9 + (Pl & 7) : Pl) * (int)in_pixstride + sl * 16;
    }
    if constexpr (D8) {                                       // the 16 zero pixels (2 KB; the first line's barrier publishes them)
        if (tid < 128) *(u32x4*)(xl + D8_ZB * 128 + tid * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    // weights of (line c, tap t) for the workgroup's NTILES channel tiles -> buffer wb: 256 pieces of 16 B per tile.  Main-loop DMA
    // (weights one tap ahead, slab) goes through dma16_buf_asm: a pending LDS-DMA *builtin* makes hipcc wait lgkmcnt(0) before
    // every MFMA group, i.e. for the pixel fragments it has just requested for the NEXT group (conv_dev.h)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wpk + (size_t)(nb * NTILES) * NC * 9 * 4096), 0, NTILES * NC * 9 * 4096, 0x00020000);
    const int wvoff = (tid & 255) * 16;
    const unsigned wl_addr = lds_addr_of(wl), xl_addr = lds_addr_of(xl);
    auto wdma = [&](int c, int t, int wboff) {
#pragma unroll
        for (int p0 = 0; p0 < NTILES * 256; p0 += NTHREADS) {
            static_assert((NTILES * 256) % NTHREADS == 0, "whole DMA rounds");
            const int pw = p0 + wave * 64;                   // first piece of this wave's 1 KB chunk (wave-uniform)
            dma16_buf_asm(wrs, wl_addr + wboff + pw * 16, wvoff, (((pw >> 8) * NC + c) * 9 + t) * 4096);   // piece (tid & 255) of tile pw >> 8
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (PLANES == 3) acc_init_bias<MT>(acc[nt], a.bias, nb * NTILES + wn * NT + nt, lane);
        else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
        }
    }

    auto xload = [&](bf16x8(&x)[4], int Pl, int tx = 1) {     // tx: the tap's column (D8: 0 / 2 leave the row on the edge lanes)
        asm volatile("" : "+v"(Pl));                          // opaque: each step's address arithmetic stays at the step (no hoisting, no spills)
        if constexpr (D8) {
            if (tx == 0) Pl = d8_left ? D8_ZB + (Pl & 15) : Pl;
            if (tx == 2) Pl = d8_right ? D8_ZB + (Pl & 15) : Pl;
        }
        const int base = lds_xbase(Pl, h);
#pragma unroll
        for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(xl + (base ^ (f << 5)));
    };
    auto wread = [&](bf16x8(&w)[4], const char* wb, int nt) {
        const char* src = wb + (wn * NT + nt) * 4096 + lane * 16;
#pragma unroll
        for (int f = 0; f < 4; ++f) w[f] = *(const bf16x8*)(src + f * 1024);
    };

    int kpar = 0;                                             // weight buffer of the current tap
    WSTAMP(const unsigned long long st_setup = __builtin_readcyclecounter();)
    for (int c = 0; c < NC; ++c) {
        WSTAMP(st_a = __builtin_readcyclecounter();)
        if (c) __syncthreads();                               // slab and weight buffers are free again
        for (int i0 = wave * 64, r = 0; i0 < npieces; i0 += NTHREADS, ++r)
            dma16_buf_asm(xrs, xl_addr + i0 * 16, xvoff, c * 128 + r * RPIX * (int)in_pixstride);
        wdma(c, 0, kpar * WB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        WSTAMP(if (c) st_line += __builtin_readcyclecounter() - st_a; else st_pro = __builtin_readcyclecounter() - st_a;)
        int Pc = D8 ? 8 : P;                                  // slab pitch
        asm volatile("" : "+s"(Pc));
        bf16x8 xf[2][4], wf[NT][4];
        xload(xf[0], xoff[0], 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const char* wb = wl + kpar * WB;
            if (t < 8) wdma(c, t + 1, (kpar ^ 1) * WB);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wread(wf[nt], wb, nt);
            const int toff = (t / 3) * Pc + (t % 3);
            const int toff_next = ((t + 1) / 3) * Pc + ((t + 1) % 3);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int k = t * MT + mt;
                if constexpr (!(ABL & 32)) {
                    if (mt + 1 < MT) xload(xf[(k + 1) & 1], xoff[mt + 1] + toff, t % 3);
                    else if (t < 8) xload(xf[(k + 1) & 1], xoff[0] + toff_next, (t + 1) % 3);
                }
                __builtin_amdgcn_sched_barrier(0);            // next pixel fragments requested before this tile's MFMAs
                const bf16x8(&x)[4] = xf[k & 1];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    mfma_step<PLANES>(acc[nt][mt], wf[nt], x);
                }
            }
            if (t < 8) {
                WSTAMP(st_a = __builtin_readcyclecounter();)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tap's weights have landed ...
                __syncthreads();                                    // ... for everyone, and this tap's buffer is free
                WSTAMP(st_tap += __builtin_readcyclecounter() - st_a;)
            }
            kpar ^= 1;
        }
    }
    WSTAMP(const unsigned long long st_loop_end = __builtin_readcyclecounter();)
    int qs[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if constexpr (D8) {                                   // PF position of the lane's pixel (rows past the end repeat the tile's last pixel)
            const int i = mtile * BM + wm * MT * 32 + mt * 32 + lpix;
            qs[mt] = pf_pos_of_index(a.gi, valid[mt] ? i : min(mtile * BM + BM, Rtot) - 1);
        } else {
            qs[mt] = xoff[mt] + slab0 + (P + 1);
        }
    }
    if constexpr (PLANES == 3) {
        if (a.in2) {
            // Extra K segment (common.h ConvArgs.in2): the strided block's 1x1 downsample of the block input, one centre tap per
            // 32-channel line of `in2` (same pixel geometry as the output: the same slab shape and offsets), accumulated on top
            // of the 3x3 conv; its BN bias joins the accumulators here.
            const int NC2 = a.in2_c / 32;
            const size_t ps2 = (size_t)a.in2_c * 4, sb2 = (size_t)slab0 * ps2;
            const size_t bytes2 = (size_t)pf_alloc_pixels(a.gi.N, a.gi.H, a.gi.W) * ps2;
            const __amdgpu_buffer_rsrc_t xrs2 = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)a.in2 + sb2), 0, (int)min(bytes2 - sb2, (size_t)0x7fffffff), 0x00020000);
            int xvoff2;
            {
                const int i = wave * 64 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
                xvoff2 = (D8 ? (Pl >> 3) * 9 + (Pl & 7) : Pl) * (int)ps2 + sl * 16;
            }
            const char* wsrc2 = (const char*)a.wpk2 + (size_t)(nb * NTILES) * NC2 * 4096 + (size_t)(tid & 255) * 16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x16 b2[1];
                acc_init_bias<1>(b2, a.bias2, nb * NTILES + wn * NT + nt, lane);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nt][mt][r] += b2[0][r];
            }
            for (int c = 0; c < NC2; ++c) {
                __syncthreads();                              // slab and weight buffers are free again
                for (int i0 = wave * 64, r = 0; i0 < npieces; i0 += NTHREADS, ++r)
                    dma16_buf(xrs2, xl + (size_t)i0 * 16, xvoff2, c * 128 + r * RPIX * (int)ps2);
#pragma unroll
                for (int p0 = 0; p0 < NTILES * 256; p0 += NTHREADS) {
                    const int pw = p0 + wave * 64;
                    dma16(wsrc2 + (size_t)((pw >> 8) * NC2 + c) * 4096, wl + pw * 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                bf16x8 wf2[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wread(wf2[nt], wl, nt);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    bf16x8 x2[4];
                    xload(x2, xoff[mt] + (D8 ? 8 : P) + 1);   // the centre tap
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mfma_step<PLANES>(acc[nt][mt], wf2[nt], x2);
                }
            }
        }
        if (a.resid) __syncthreads();                         // weight stages + slab become the waves' residual staging (NBUF tiles each)
        conv_tail_mx<NT, MT, RESID_NBUF>(a, acc, qs, valid, nb * NTILES + wn * NT, lane, smem + wave * (RESID_NBUF * 4096), slab0);
    } else {
        char* scratch = nullptr;
        if (PLANES == 2 && a.resid && !(a.flags & CONV_RESID_DIRECT)) {    // slab memory becomes the waves' residual staging (epilogues)
            __syncthreads();
            scratch = xl + wave * 8192;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) conv_epilogue_q<MT, PLANES>(a, acc[nt], qs, valid, nb * NTILES + wn * NT + nt, lane, scratch);
    }
#ifdef WSI_STUDY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the stores count as tail)
    if (lane == 0 && (blockIdx.x & 63) == 0) {                // one workgroup in 64 reports: 10^5 waves adding to seven words distort the launch
        const unsigned long long st_end = __builtin_readcyclecounter();
        atomicAdd(&g_wide_stamps[0], 1ull);
        atomicAdd(&g_wide_stamps[1], st_end - st_begin);
        atomicAdd(&g_wide_stamps[2], st_setup - st_begin);
        atomicAdd(&g_wide_stamps[3], st_pro);
        atomicAdd(&g_wide_stamps[4], st_line);
        atomicAdd(&g_wide_stamps[5], st_tap);
        atomicAdd(&g_wide_stamps[6], st_end - st_loop_end);
    }
#endif
}


int g_wide_d8 = 1;                                       // A/B (wsi_conv_set_mode +131072 off): 8-pixel slab rows on 8 x 8 maps (r05)
template <int PLANES, int MINW, int ABL = 0, int WM = 2, int WN = 2, int NT = 2, bool D8 = false>
static int launch_wide(const ConvArgs& a, hipStream_t st) {
    constexpr int BM = WM * 128, NTHREADS = WM * WN * 64, BN = WN * NT * 32;
    if (a.go.C % BN) return WSI_EINVAL;
    if constexpr (!D8 && ABL == 0 && WM == 2 && WN == 2 && NT == 2) {
        if (g_wide_d8 && a.gi.W == 8 && a.gi.H == 8) return launch_wide<PLANES, MINW, 0, 2, 2, 2, true>(a, st);
    }
    const int nblocks = a.go.C / BN;
    const long long R = (long long)a.gi.N * a.gi.H * a.gi.W;
    const int mtiles = (int)((R + BM - 1) / BM);
    size_t xbytes = D8 ? (size_t)(304 + 16) * 128       // 37 rows of 8 pixels (rounded to 16) + the 16 zero pixels
                       : (size_t)((dense_max_slab_pixels(a, BM) * 8 + NTHREADS - 1) / NTHREADS * NTHREADS) * 16;
    if (xbytes < (size_t)WM * WN * 8192) xbytes = (size_t)WM * WN * 8192;    // residual staging of the epilogue (mode 2: in the slab)
    size_t lds = 2 * (BN / 32) * 4096 + xbytes;
    if (PLANES == 3 && lds < (size_t)WM * WN * 4 * 4096) lds = (size_t)WM * WN * 4 * 4096;   // mode 3: four residual tiles per wave, from smem + 0
    if (lds > 160 * 1024) return WSI_EINVAL;
    auto k = conv3x3s1_wide_kernel<PLANES, MINW, ABL, WM, WN, NT, D8>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return WSI_EINVAL;
    const int grid = (a.flags & CONV_XCD_RANGES) ? (mtiles * nblocks + 7) / 8 * 8 : mtiles * nblocks;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}


// --------------------------------------------------------------------------------------------
// Stride-2 3x3 conv (+ fused 1x1 stride-2 downsample) in slab form.
// in(2y+kh-1, 2x+kw-1) = phase image I[py][px](y+dy, x+dx) with (py,dy) = (1,-1),(0,0),(1,0) for
// kh = 0,1,2 (same for kw): each of the four input phase images, viewed in the OUTPUT's padded-flat
// geometry, is accessed with stride 1, so one contiguous region per phase is staged per 128-byte
// line (sizes BM, BM+1, BM+P, BM+P+1 pixels) and the nine taps are LDS address shifts.  The 1x1
// stride-2 downsample conv reads exactly the centre tap's pixels: it runs as a 10th tap with its
// own weights into a second accumulator set (FUSE), so the block's two stride-2 convs cost one
// pass over the input.  Ten taps are an even count: a 2-slot weight ring stays phase-aligned
// across lines.
template <int MT, int WM, int WN, int PLANES, int MINW, bool FUSE, int PMAX = 34, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64, MINW) void conv3x3s2_slab_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = WM * MT * 32;
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int KMAX = ((4 * BM + 2 * PMAX + 2) * 8 + NTHREADS - 1) / NTHREADS;    // pieces per thread, P <= PMAX
    constexpr int NT = FUSE ? 10 : 9;
    constexpr int RING = FUSE ? 2 : 3;                        // weight ring slots; NT % RING == 0 keeps lines aligned
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int nblocks = a.go.C / (WN * 32);
    int nb = blockIdx.x % nblocks, mtile = blockIdx.x / nblocks;
    if (a.flags & CONV_XCD_ORDER) {                            // XCD-aware order, see conv3x3s1_slab3_kernel
        const int per = 8 * nblocks, r = blockIdx.x % per;
        nb = r >> 3;
        mtile = (blockIdx.x / per) * 8 + (r & 7);
        if (mtile >= (a.go.NS + BM - 1) / BM) return;
    }
    const int P = a.go.P;
    const int q0 = a.go.G + mtile * BM;
    const int R01 = BM, R10 = 2 * BM + 1, R11 = 3 * BM + 1 + P;                      // region bases (pixels)
    const int npix = 4 * BM + 2 * P + 2;
    const int npieces = npix * 8;
    const int ntile = nb * WN + wn;
    const int NC = a.gi.C / PFmt<PLANES>::CPL;
    const size_t in_pixstride = (size_t)a.gi.C * PFmt<PLANES>::BPC;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wpk + (size_t)ntile * NC * 9 * 4096), 0, NC * 9 * 4096, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(FUSE ? a.wpk2 : a.wpk) + (size_t)ntile * NC * 4096), 0, NC * 4096, 0x00020000);
    const int wvoff = lane * 16;

    // source pixel (input PF index, 0 = a zero guard pixel) of every 16-byte piece this thread stages
    int srcpix[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int i = wave * 64 + k * NTHREADS + lane;
        const int Pl = i >> 3;
        int reg = 0, j = Pl;
        if (Pl >= R11) { reg = 3; j = Pl - R11; }
        else if (Pl >= R10) { reg = 2; j = Pl - R10; }
        else if (Pl >= R01) { reg = 1; j = Pl - R01; }
        const int py = reg >> 1, px = reg & 1;
        const int v = q0 - (py ? P : 0) - px + j;                                    // virtual position in the output geometry
        int src = 0;
        int r = v - a.go.G;
        if (Pl < npix && r >= 0 && r < a.go.NS) {
            const int n = r / a.go.S;
            r -= n * a.go.S;
            const int y = r / P, x = r - y * P;
            if (x != a.go.W && y != a.go.H) src = a.gi.G + n * a.gi.S + (2 * y + py) * a.gi.P + 2 * x + px;
        }
        const int sl = (i & 7) ^ ((Pl >> 1) & 7);                                    // swizzled 16-byte slot
        srcpix[k] = (int)((unsigned)src * (unsigned)in_pixstride + (unsigned)sl * 16u);   // byte offset (< 4 GiB, host-checked)
    }
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0xffffffff, 0x00020000);

    f32x16 acc[MT], accd[FUSE ? MT : 1];
    if constexpr (PLANES == 3) {                               // mode 3: accumulators start from the folded BN bias
        acc_init_bias<MT>(acc, a.bias, ntile, lane);
        if constexpr (FUSE) acc_init_bias<MT>(accd, a.bias2, ntile, lane);
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        if constexpr (FUSE) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) accd[mt][r] = 0.f;
        }
    }
    int xoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xoff[mt] = wm * MT * 32 + mt * 32 + l31;

    auto wload = [&](bf16x8(&w)[4], int c, int t) {           // t in [0,9): 3x3 tap, t == 9: downsample weights
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const u32x4 v = (t < 9) ? __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + f * 1024, (c * 9 + t) * 4096, 0)
                                    : __builtin_amdgcn_raw_buffer_load_b128(wrd, wvoff + f * 1024, c * 4096, 0);
            w[f] = __builtin_bit_cast(bf16x8, v);
        }
    };
    auto xload = [&](bf16x8(&x)[4], int Pl) {
        const int base = lds_xbase(Pl, h);
#pragma unroll
        for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(smem + (base ^ (f << 5)));
    };
    // LDS pixel offset of tap t for tile row 0: region base + back + dy*P + dx
    auto tap_off = [&](int t, int Pc) {
        if (t == 9) return 0;                                  // centre pixels: region 00
        const int kh = t / 3, kw = t % 3;
        const int py = kh != 1, px = kw != 1;
        const int base = py ? (px ? 3 * BM + 1 + Pc : 2 * BM + 1) : (px ? BM : 0);
        const int back = (py ? Pc : 0) + px;
        return base + back + (kh == 0 ? -Pc : 0) + (kw == 0 ? -1 : 0);
    };

    bf16x8 wbuf[RING][4], xf[2][4];
    wload(wbuf[0], 0, 0);
    for (int c = 0; c < NC; ++c) {
        int Pc = P;
        asm volatile("" : "+s"(Pc));
        if (c) __syncthreads();
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i0 = wave * 64 + k * NTHREADS;
            if (i0 < npieces)
                dma16_buf(irs, smem + (size_t)i0 * 16, srcpix[k], c * 128);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        xload(xf[0], xoff[0] + tap_off(0, Pc));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            asm volatile("" ::: "memory");
            if constexpr (!(ABL & 16)) {                      // ABL 16: bottleneck study, stale weight registers
                if (t + 1 < NT) wload(wbuf[(t + 1) % RING], c, t + 1);
                else if (c + 1 < NC) wload(wbuf[0], c + 1, 0);
            }
            asm volatile("" ::: "memory");                    // pin the prefetch above this tap's MFMAs
            const int toff = tap_off(t, Pc);
            const int toff_next = t + 1 < NT ? tap_off(t + 1, Pc) : 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int k = t * MT + mt;
                if (mt + 1 < MT) xload(xf[(k + 1) & 1], xoff[mt + 1] + toff);
                else if (t + 1 < NT) xload(xf[(k + 1) & 1], xoff[0] + toff_next);
                const bf16x8(&w)[4] = wbuf[t % RING];
                const bf16x8(&x)[4] = xf[k & 1];
                f32x16& d = (FUSE && t == 9) ? accd[FUSE ? mt : 0] : acc[mt];
                if constexpr (PLANES == 3) {
                    d = mfma_mx6(d, w, x);
                } else if constexpr (PLANES == 2) {
                    mfma_step<2>(d, w, x);
                } else {
#pragma unroll
                    for (int f = 0; f < 4; ++f) d = mfma_bf16(w[f], x[f], d);
                }
            }
        }
    }
    conv_epilogue_any<MT, PLANES>(a, acc, q0 + wm * MT * 32, ntile, lane);
    if constexpr (FUSE) {
        ConvArgs a2 = a;
        a2.out = a.out2; a2.bias = a.bias2; a2.resid = nullptr; a2.relu = 0;
        a2.wpk = a.wpk2; a2.ksize = 1;                        // (mode 2: the 1x1 pack's own channel scales follow ITS blocks)
        conv_epilogue_any<MT, PLANES>(a2, accd, q0 + wm * MT * 32, ntile, lane);
    }
}

template <int MT, int WM, int WN, int PLANES, int MINW, bool FUSE, int PMAX = 34, int ABL = 0>
static int launch_s2slab(const ConvArgs& a, hipStream_t st) {
    constexpr int BM = WM * MT * 32, NTHREADS = WM * WN * 64;
    if (a.go.C % (WN * 32) || a.go.P > PMAX) return WSI_EINVAL;
    if ((unsigned long long)pf_alloc_pixels(a.gi.N, a.gi.H, a.gi.W) * a.gi.C * PFmt<PLANES>::BPC >= 0xffffffffull) return WSI_EINVAL;
    const int mtiles = (a.go.NS + BM - 1) / BM;
    const int nblocks = a.go.C / (WN * 32);
    const int npieces = (4 * BM + 2 * a.go.P + 2) * 8;
    const size_t lds = (size_t)((npieces + NTHREADS - 1) / NTHREADS * NTHREADS) * 16;
    if (lds > 160 * 1024) return WSI_EINVAL;
    auto k = conv3x3s2_slab_kernel<MT, WM, WN, PLANES, MINW, FUSE, PMAX, ABL>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return WSI_EINVAL;
    }
    const int grid = (a.flags & CONV_XCD_ORDER) ? (mtiles + 7) / 8 * 8 * nblocks : mtiles * nblocks;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

// --------------------------------------------------------------------------------------------
// Stride-2 3x3 conv + fused 1x1 stride-2 downsample over a PHASE-SPLIT input (common.h ConvArgs): each of the four
// input phase images is an ordinary PF tensor in the OUTPUT geometry, so the pixels a tile of 256 output positions
// needs from one phase are ONE contiguous range (plain LDS-DMA, no gather) and the taps of that phase are LDS address
// shifts.  The r01 ablation showed the phase-slab kernel spends 40-55 % of its time re-streaming weights (every wave
// fetches 3.25 KB per tap for 64 pixels); here 8 waves = 4 pixel quarters x 2 channel halves share one 16 KB weight
// stage per tap (LDS-DMA, double-buffered): 3.25x less weight traffic per pixel.  Work items per 32-channel line:
// phase 00: tap (1,1) + the downsample tap; 01: (1,0) (1,2); 10: (0,1) (2,1); 11: (0,0) (0,2) (2,0) (2,2).  The
// next item's pixels are fetched while the current one multiplies (two slab buffers); waves 0-3 issue the pixel
// DMA, waves 4-7 the weight DMA, so each wave's in-order vmcnt tracks one kind only.  One barrier per tap.
// NT (r03): channel tiles per wave.  2 = the form above (workgroup = 256 px x 128 couts, 16 KB weight stages in a ring of four, units of
// up to two steps between barriers).  4 (DS = false only: 128 accumulator registers) = 256 px x 256 couts: every pixel buffer feeds twice
// the MFMAs - half the slab DMA pieces and pixel-fragment reads per MFMA on the layer-3 / layer-4 entries - with 32 KB weight stages
// in a ring of two, one step per unit.
template <int PLANES, bool DENSE, bool DS, int NT = 2>
__global__ __launch_bounds__(512, 1) void conv3x3s2_wide_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(NT == 2 || (NT == 4 && !DS), "four channel tiles per wave: no second accumulator set");
    constexpr int MT = 2, BM = 256, NTILES = 2 * NT, ULEN = NT == 2 ? 2 : 1;
    constexpr int WBUF = NTILES * 4096, XB = 45056;                   // X: up to 352 pixels (256 real ones + their pads + P + 1), whole DMA rounds
    constexpr int NWB = 2 * ULEN;                             // weight ring: the next unit (<= ULEN steps) is requested while this one multiplies
    char* const wl = smem;                                    // NWB weight buffers
    char* const xl0 = smem + NWB * WBUF;                      // 2 pixel buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int nblocks = a.go.C / (NTILES * 32);
    // XCD-contiguous ranges (ids are dealt round-robin to the 8 XCDs): XCD x walks workgroups [x*chunk, (x+1)*chunk) in
    // dispatch order, so the channel blocks of one pixel tile - which read the same input pixels - and neighbouring tiles'
    // halo rows meet in ONE L2 instead of being fetched by up to four (r01 PMC: 1.53x the algorithmic bytes)
    const int chunk = gridDim.x >> 3, lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    const int nb = lin % nblocks, mtile = lin / nblocks;
    if (mtile >= a.mtiles) return;                    // grid is rounded up to a multiple of 8 (whole workgroup leaves)
    const int P = a.go.P;
    // dense tile: 256 REAL output pixels (raster order); the pixels a phase needs are still one contiguous PF range,
    // from the first pixel (minus the phase's back-shift) to the last; no MFMA work on pad positions
    int q0, qlast, qs[MT];
    bool valid[MT];
    if constexpr (!DENSE) {                                   // tiny maps (a dense tile would span too many pad rows): 256 consecutive positions
        q0 = a.go.G + mtile * BM;
        qlast = q0 + BM - 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            qs[mt] = q0 + wm * MT * 32 + mt * 32 + l31;
            valid[mt] = pf_is_pixel(a.go, qs[mt]);
        }
    } else {
        const int HW = a.go.H * a.go.W, R = a.go.N * HW;
        auto pos = [&](int i) { return pf_pos_of_index(a.go, i); };
        const int i0 = mtile * BM, i1 = min(i0 + BM, R) - 1;
        q0 = pos(i0);
        qlast = pos(i1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int i = i0 + wm * MT * 32 + mt * 32 + (a.go.W < 32 ? dense_lane_pixel(l31) : l31);   // (conv_dev.h: LDS bank conflicts)
            valid[mt] = i < R;
            qs[mt] = pos(valid[mt] ? i : i1);
        }
    }
    const int NC = a.gi.C / PFmt<PLANES>::CPL;
    const size_t in_pixstride = (size_t)a.gi.C * PFmt<PLANES>::BPC;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wpk + (size_t)(nb * NTILES) * NC * 9 * 4096), 0, NTILES * NC * 9 * 4096, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)(DS ? a.wpk2 : a.wpk) + (size_t)(nb * NTILES) * NC * 4096), 0, NTILES * NC * 4096, 0x00020000);
    const int wvoff = lane * 16;
    // weights of (line c, tap t; t == 9: downsample) -> buffer wb; wave w in 4..7 moves quarter w-4 of each channel tile
    auto wdma = [&](int c, int t, char* wb) {
#pragma unroll
        for (int j = 0; j < NTILES; ++j) {
            // (asm DMA: a pending LDS-DMA builtin turns every counted LDS wait of the loop into lgkmcnt(0), conv_dev.h)
            if (t < 9) dma16_buf_asm(wrs, lds_addr_of(wb) + j * 4096 + (wave - 4) * 1024, wvoff, ((j * NC + c) * 9 + t) * 4096 + (wave - 4) * 1024);
            else dma16_buf_asm(wrd, lds_addr_of(wb) + j * 4096 + (wave - 4) * 1024, wvoff, (j * NC + c) * 4096 + (wave - 4) * 1024);
        }
    };
    // pixels of (line c, phase ph) -> buffer xb (waves 0..3)
    auto xdma = [&](int c, int ph, char* xb) {
        const int back = ((ph & 2) ? P : 0) + (ph & 1);
        const char* src = (const char*)a.in + ((size_t)ph * a.in_split_pixels + (size_t)(q0 - back)) * in_pixstride + c * 128;
        const int npieces = (qlast - q0 + 1 + back) * 8;
        for (int i0 = wave * 64; i0 < npieces; i0 += 256) {
            const int i = i0 + lane;
            const int Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
            dma16_asm(src + (size_t)Pl * in_pixstride + sl * 16, lds_addr_of(xb) + i0 * 16);
        }
    };
    auto wread = [&](bf16x8(&w)[NT][4], const char* wb) {
        const char* src = wb + (wn * NT) * 4096 + lane * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int f = 0; f < 4; ++f) w[nt][f] = *(const bf16x8*)(src + nt * 4096 + f * 1024);
        }
    };

    f32x16 acc[NT][MT], accd[DS ? NT : 1][DS ? MT : 1];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (PLANES == 3) {                           // mode 3: accumulators start from the folded BN bias
            acc_init_bias<MT>(acc[nt], a.bias, nb * NTILES + wn * NT + nt, lane);
            if constexpr (DS) acc_init_bias<MT>(accd[nt], a.bias2, nb * NTILES + wn * NT + nt, lane);
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[nt][mt][r] = 0.f;
                    if constexpr (DS) accd[nt][mt][r] = 0.f;
                }
        }
    }

    // Steps of one 32-channel line in execution order: (phase, 3x3 tap; tap 9 = the fused 1x1 downsample, which reads the
    // centre pixels = phase 00).  Phase 00: tap (1,1) [+ downsample]; 01: (1,0) (1,2); 10: (0,1) (2,1); 11: the four corners.
    // The LDS pixel shift of tap (kh, kw) inside its phase buffer is (kw == 2) + (kh == 2) * P.  Steps run in UNITS of one or two
    // between barriers; the weights of the next unit are requested at the start of the current one and waited for at its end
    // (ring of four buffers indexed by the running step number: at most two units are alive).
    constexpr int SPL = DS ? 10 : 9;                                         // steps per line
    constexpr int STEP_TAP[10] = {4, DS ? 9 : 3, DS ? 3 : 5, DS ? 5 : 1, DS ? 1 : 7, DS ? 7 : 0, DS ? 0 : 2, DS ? 2 : 6, DS ? 6 : 8, 8};
    constexpr int UNIT0[4] = {0, DS ? 2 : 1, DS ? 4 : 3, DS ? 6 : 5};        // first step of each phase
    constexpr int NSTEP[4] = {DS ? 2 : 1, 2, 2, 4};
    const int NG = NC * SPL;
    auto wdma_step = [&](int c, int j) {                      // j may run past the line: wraps into the next one
        const int cc = c + j / SPL, g = cc * SPL + j % SPL;
        if (g < NG) wdma(cc, STEP_TAP[j % SPL], wl + (g & (NWB - 1)) * WBUF);
    };
    if (wave < 4) xdma(0, 0, xl0);
    else { wdma_step(0, 0); if (NSTEP[0] == 2 && ULEN == 2) wdma_step(0, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const char* xl = xl0 + (size_t)(ph & 1) * XB;     // 4 items per line: buffer parity = phase parity
            if (wave < 4) {                                   // next item's pixels: land while this item multiplies
                if (ph < 3) xdma(c, ph + 1, xl0 + (size_t)((ph + 1) & 1) * XB);
                else if (c + 1 < NC) xdma(c + 1, 0, xl0);
            }
#pragma unroll
            for (int kp = 0; kp < NSTEP[ph]; kp += ULEN) {
                const int j0 = UNIT0[ph] + kp;                // first step of the unit
                const int ulen = NSTEP[ph] - kp >= ULEN ? ULEN : 1;
                // the next unit: the rest of this phase, or the first unit of the next phase / line
                const int jn = j0 + ulen;
                const int nlen = ULEN == 1 ? 1 : (kp + 2 < NSTEP[ph]) ? 2 : (ph < 3 ? (NSTEP[ph + 1] >= 2 ? 2 : 1) : (NSTEP[0] >= 2 ? 2 : 1));
                if (wave >= 4) { wdma_step(c, jn); if (nlen == 2) wdma_step(c, jn + 1); }
#pragma unroll
                for (int k = 0; k < ulen; ++k) {
                    const int g = c * SPL + j0 + k;
                    const int t = STEP_TAP[j0 + k];
                    constexpr bool PREF = NT == 2 && !DS;      // (with four channel tiles, or the second accumulator set, the fenced order needs > 256 registers)
                    bf16x8 wf[NT][4], xf[PREF ? 2 : MT][4];
                    wread(wf, wl + (g & (NWB - 1)) * WBUF);
                    const int sh = (t < 9 && t % 3 == 2 ? 1 : 0) + (t < 9 && t / 3 == 2 ? P : 0);
                    auto xload = [&](bf16x8(&x)[4], int mt) {
                        int ql = qs[mt] - q0;
                        asm volatile("" : "+v"(ql));           // opaque: the address arithmetic of every (line, step) is done HERE (hoisted
                        const int Pl = ql + sh;                // out of the line loop it costs ~70 registers and spills into the loop)
                        const int base = lds_xbase(Pl, h);
#pragma unroll
                        for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(xl + (base ^ (f << 5)));
                    };
                    if constexpr (PREF) {
                        // r04: the pixel fragments of tile mt + 1 are requested before the MFMAs of tile mt and stay in flight under
                        // them (r03 read all four tiles' fragments, waited for everything, then multiplied)
                        xload(xf[0], 0);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if (mt + 1 < MT) xload(xf[(mt + 1) & 1], mt + 1);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) mfma_step<PLANES>(acc[nt][mt], wf[nt], xf[mt & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) xload(xf[mt], mt);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                if constexpr (DS) mfma_step<PLANES>(t == 9 ? accd[nt][mt] : acc[nt][mt], wf[nt], xf[mt]);
                                else mfma_step<PLANES>(acc[nt][mt], wf[nt], xf[mt]);
                            }
                    }
                }
                // end of the unit: the weight waves wait for the next unit's stages, the pixel waves (at an item's last unit)
                // for the next item's pixels
                if (wave >= 4 || kp + ULEN >= NSTEP[ph]) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ntile = nb * NTILES + wn * NT + nt;
        if constexpr (PLANES == 3) {
            conv_epilogue_mx<MT>(a, acc[nt], qs, valid, ntile, lane);
        } else {
            conv_epilogue_q<MT, PLANES>(a, acc[nt], qs, valid, ntile, lane);
        }
        if constexpr (DS) {
            ConvArgs a2 = a;
            a2.out = a.out2; a2.bias = a.bias2; a2.resid = nullptr; a2.relu = 0;
            a2.wpk = a.wpk2; a2.ksize = 1;                    // (mode 2: the 1x1 pack's own channel scales follow ITS blocks)
            if constexpr (PLANES == 3) conv_epilogue_mx<MT>(a2, accd[nt], qs, valid, ntile, lane);
            else conv_epilogue_q<MT, PLANES>(a2, accd[nt], qs, valid, ntile, lane);
        }
    }
}

int g_s2_nt4 = 1;                                        // A/B (wsi_conv_set_mode +32768 off): 256-cout workgroups in the wide stride-2 kernel
template <int PLANES>
static int launch_s2wide(const ConvArgs& a, hipStream_t st) {
    constexpr int BM = 256, XB = 45056;
    const bool ds = a.out2 != nullptr;                                       // fused 1x1 downsample branch, or the 3x3 conv alone
    if (a.go.C % 128 || a.go.P > 34 || !a.in_split_pixels || (ds && (!a.wpk2 || !a.bias2)) || a.out_split_pixels) return WSI_EINVAL;
    const long long R = (long long)a.go.N * a.go.H * a.go.W;
    const bool nt4 = !ds && PLANES == 3 && g_s2_nt4 && a.go.C % 256 == 0;      // 256 couts per workgroup (r03)
    const int nblocks = a.go.C / (nt4 ? 256 : 128);
    // dense tiles if the span of 256 real pixels (+ the largest phase back-shift) fits one pixel buffer
    ConvArgs g = a;
    g.gi = a.go;
    const long long span = dense_max_slab_pixels(g, BM) - (a.go.P + 1);          // that helper adds 2P + 2 of halo; a phase needs P + 1
    const bool dense = (span * 8 + 255) / 256 * 256 * 16 <= XB;
    const int mtiles = dense ? (int)((R + BM - 1) / BM) : (a.go.NS + BM - 1) / BM;
    const size_t lds = 4 * 16384 + 2 * XB;                                     // (NT = 4: two stages of 32 KB)
    auto k = nt4 ? (dense ? conv3x3s2_wide_kernel<PLANES, true, false, 4> : conv3x3s2_wide_kernel<PLANES, false, false, 4>)
           : dense ? (ds ? conv3x3s2_wide_kernel<PLANES, true, true> : conv3x3s2_wide_kernel<PLANES, true, false>)
                   : (ds ? conv3x3s2_wide_kernel<PLANES, false, true> : conv3x3s2_wide_kernel<PLANES, false, false>);
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return WSI_EINVAL;
    ConvArgs b = a;
    b.mtiles = mtiles;
    hipLaunchKernelGGL(k, dim3((mtiles * nblocks + 7) / 8 * 8), dim3(512), lds, st, b);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int g_s2_ablate = 0;
int g_l1_rows = 1;                                       // A/B (wsi_conv_set_mode +1024 off): row-stacked layer-1 kernel (cfg 40) instead of slab3 (cfg 38)
int g_slab_pair = 1;                                     // A/B (wsi_conv_set_mode +4096 off): paired-tile LDS addressing of the layer-1 kernel
int g_xcd_ranges = 2;                                    // XCD-contiguous tile ranges: 1 = the 64-channel layer only, 2 = every stride-1 layer (r01: ~-1 % overall)
int g_xcd_order = 0;                                     // 1: CONV_XCD_ORDER for multi-channel-block launches
int g_s2_small_tiles = 1;                                // r01: 64-pixel tiles measured ~10 % faster (3 workgroups per CU)
// stride-2 3x3 (+ optional fused downsample) dispatch; cfg 0 = gather kernel (unfused only)
int wsi_s2_dispatch(const ConvArgs& a_in, int planes, hipStream_t st) {
    ConvArgs a = a_in;
    if (a.in_split_pixels)                                   // phase-split input: the wide kernel is the only reader
        return planes == 3 ? launch_s2wide<3>(a, st) : planes == 2 ? launch_s2wide<2>(a, st) : WSI_EINVAL;
    if (g_xcd_order && a.go.C > 128) a.flags |= CONV_XCD_ORDER;
    if (a.gi.C % 64 || a.go.C % 128 || planes < 1 || planes > 3) return WSI_EINVAL;
    if (a.go.H * 2 != a.gi.H || a.go.W * 2 != a.gi.W || a.gi.N != a.go.N) return WSI_EINVAL;
    const bool fuse = a.out2 != nullptr;
    if (fuse && (!a.wpk2 || !a.bias2)) return WSI_EINVAL;
    if (a.go.P > 34) {                                       // output maps wider than 33 (patches > 256): 64-pixel tiles, MINW 2
        if (planes == 3) return fuse ? launch_s2slab<2, 1, 4, 3, 2, true, 130>(a, st) : launch_s2slab<2, 1, 4, 3, 2, false, 130>(a, st);
        if (planes == 2) return fuse ? launch_s2slab<2, 1, 4, 2, 2, true, 130>(a, st) : launch_s2slab<2, 1, 4, 2, 2, false, 130>(a, st);
        return WSI_EINVAL;                                   // speed mode: gather kernel
    }
#ifdef WSI_STUDY
    if (g_s2_ablate && planes == 3 && fuse) return launch_s2slab<2, 1, 4, 3, 3, true, 34, 16>(a, st);   // weight loads off (wrong results)
#endif
    if (g_s2_small_tiles) {                                  // 64-pixel tiles: smaller slabs, more workgroups per CU
        if (planes == 3) return fuse ? launch_s2slab<2, 1, 4, 3, 3, true>(a, st) : launch_s2slab<2, 1, 4, 3, 3, false>(a, st);
        if (planes == 2) return fuse ? launch_s2slab<2, 1, 4, 2, 3, true>(a, st) : launch_s2slab<2, 1, 4, 2, 3, false>(a, st);
    }
    if (planes == 3) return fuse ? launch_s2slab<4, 1, 4, 3, 2, true>(a, st) : launch_s2slab<4, 1, 4, 3, 2, false>(a, st);
    if (planes == 2) return fuse ? launch_s2slab<4, 1, 4, 2, 2, true>(a, st) : launch_s2slab<4, 1, 4, 2, 2, false>(a, st);
    return fuse ? launch_s2slab<4, 1, 4, 1, 2, true>(a, st) : launch_s2slab<4, 1, 4, 1, 2, false>(a, st);
}

template <int MT, int WM, int WN, int PLANES>
static int launch_gather(const ConvArgs& a, hipStream_t st) {
    constexpr int BM = WM * MT * 32;
    if (a.go.C % (WN * 32)) return WSI_EINVAL;
    const int mtiles = (a.go.NS + BM - 1) / BM;
    const int nblocks = a.go.C / (WN * 32);
    const size_t lds = (size_t)BM * 128;
    hipLaunchKernelGGL((conv_gather_kernel<MT, WM, WN, PLANES>), dim3(mtiles * nblocks), dim3(WM * WN * 64), lds, st, a);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

// Slab tile configurations (cfg index -> MT, WM, WN, min waves/SIMD, dense).  Every wave owns 32 output channels and
// MT*32 pixels; WN waves share one pixel slab; BM = WM*MT*32, BN = WN*32.  (cfg 0-9, the first slab kernel, are gone.)
// software-pipelined variants: cfg 20 + index
#define SLAB3_CFGS(X) \
    X(20, 4, 1, 4, 2, false) \
    X(21, 4, 2, 2, 2, false) \
    X(25, 2, 2, 4, 3, false) \
    X(27, 4, 4, 2, 1, false) \
    X(28, 2, 2, 2, 3, true) \
    X(29, 2, 2, 2, 4, true) \
    X(39, 4, 4, 2, 1, true) \
    X(26, 4, 1, 2, 3, false) \
    X(30, 4, 1, 4, 2, true) \
    X(31, 4, 2, 2, 2, true) \
    X(32, 2, 2, 4, 3, true) \
    X(33, 8, 1, 4, 2, true) \
    X(34, 8, 1, 2, 2, true) \
    X(35, 8, 2, 2, 2, true) \
    X(36, 4, 2, 4, 2, true) \
    X(37, 4, 1, 4, 3, true) \
    X(38, 4, 2, 2, 3, true)


int wsi_pp_dispatch(const ConvArgs& a, int planes, int cfg, hipStream_t st);      // conv_pp.hip

int wsi_slab_dispatch_cfg(const ConvArgs& a_in, int planes, int cfg, hipStream_t st) {
    ConvArgs a = a_in;
    if (g_xcd_order && cfg >= 20 && cfg < 40 && !CONV_STUDY(a, ~7)) a.flags |= CONV_XCD_ORDER;     // slab3 family only
    if (g_xcd_ranges && !CONV_STUDY(a, ~7) && ((cfg >= 20 && cfg < 42) || cfg == 60 || cfg == 90 || cfg == 91 || (cfg >= 70 && cfg < 90)) && (g_xcd_ranges == 2 || a.go.C == 64)) a.flags |= CONV_XCD_RANGES;
    if ((a.flags & CONV_IN96) && !((cfg >= 20 && cfg <= 42) || cfg == 90 || cfg == 91)) return WSI_EINVAL;   // 96-byte input lines: slab3 / row-stacked kernels only
    if (a.in_up && !((cfg >= 20 && cfg < 40) || cfg == 90 || cfg == 91)) return WSI_EINVAL;   // fused upsample + concat input: slab3 kernels only
    if (cfg < 20) return WSI_EINVAL;                         // (cfg 0-9 were the first slab kernel, removed)
    if (cfg >= 70 && cfg < 90) return wsi_pp_dispatch(a, planes, cfg, st);           // ping-pong kernels (conv_pp.hip)
    // cfg 90: 512 px x 32 couts (8 x 1 waves, two pixel tiles each) for 32-channel outputs (U-Net decoder levels 4-5)
    if (cfg == 90) return planes == 3 ? launch_slab3<2, 8, 1, 3, 1, true>(a, st) : planes == 2 ? launch_slab3<2, 8, 1, 2, 1, true>(a, st) : WSI_EINVAL;
    // cfg 91: 256 px x 32 couts, the fallback where a 512-pixel tile's slab exceeds the LDS (tiles straddling two images of a wide map)
    if (cfg == 91) return planes == 3 ? launch_slab3<2, 4, 1, 3, 2, true>(a, st) : planes == 2 ? launch_slab3<2, 4, 1, 2, 2, true>(a, st) : WSI_EINVAL;
    // cfg 40 / 41: row-stacked layer-1 kernel (mode 3, 64-wide maps), three / two workgroups per CU
    if (cfg == 42) return planes == 3 ? launch_l1p(a, st) : WSI_EINVAL;         // r05: persistent, producer-fed form (96-byte lines only)
    if (cfg == 40) return planes == 3 ? launch_rows<3>(a, st) : WSI_EINVAL;
    if (cfg == 41) return planes == 3 ? launch_rows<2>(a, st) : WSI_EINVAL;
    if (cfg == 60) return planes == 3 ? launch_wide<3, 2>(a, st) : planes == 2 ? launch_wide<2, 2>(a, st) : launch_wide<1, 2>(a, st);
#ifdef WSI_STUDY
    if (cfg == 61 && planes == 3) return launch_wide<3, 2, 32>(a, st);               // ablation: no pixel-fragment reads
    if (cfg == 67) return planes == 3 ? launch_wide<3, 1, 0, 4, 2, 1>(a, st) : planes == 2 ? launch_wide<2, 1, 0, 4, 2, 1>(a, st) : WSI_EINVAL;   // 512 px x 64 couts
    if (cfg == 68) return planes == 3 ? launch_wide<3, 2, 0, 2, 2, 1>(a, st) : planes == 2 ? launch_wide<2, 2, 0, 2, 2, 1>(a, st) : WSI_EINVAL;   // 256 px x 64 couts
    if (cfg >= 50 && cfg <= 53 && planes == 3) {             // ablation builds of cfg 30 (bottleneck studies only)
        switch (cfg) {
        case 50: return launch_slab3<4, 1, 4, 3, 2, true, 16>(a, st);
        case 51: return launch_slab3<4, 1, 4, 3, 2, true, 32>(a, st);
        case 52: return launch_slab3<4, 1, 4, 3, 2, true, 48>(a, st);
        case 53: return launch_slab3<4, 1, 4, 3, 2, true, 8>(a, st);
        }
    }
    switch (cfg) {
#define X(id, MT, WM, WN, MINW, DENSE) \
    case id: return planes == 3 ? launch_slab3<MT, WM, WN, 3, MINW, DENSE>(a, st) \
                  : planes == 2 ? launch_slab3<MT, WM, WN, 2, MINW, DENSE>(a, st) : launch_slab3<MT, WM, WN, 1, MINW, DENSE>(a, st);
        SLAB3_CFGS(X)
#undef X
    }
#else
    // product build: only the tuned configurations are instantiated (cfg 30 / 31: slab3 for 128-multiple / 64-channel outputs)
    if (cfg == 30) return planes == 3 ? launch_slab3<4, 1, 4, 3, 2, true>(a, st) : planes == 2 ? launch_slab3<4, 1, 4, 2, 2, true>(a, st) : launch_slab3<4, 1, 4, 1, 2, true>(a, st);
    if (cfg == 31) return planes == 3 ? launch_slab3<4, 2, 2, 3, 2, true>(a, st) : planes == 2 ? launch_slab3<4, 2, 2, 2, 2, true>(a, st) : launch_slab3<4, 2, 2, 1, 2, true>(a, st);
    // cfg 38: cfg 31 held to 168 registers = three waves per SIMD, three workgroups per CU (r03 A/B on layer 1)
    if (cfg == 38) {
        if (planes != 3) return WSI_EINVAL;
        const int rc = g_slab_pair ? launch_slab3<4, 2, 2, 3, 3, true, 0, true>(a, st) : WSI_EINVAL;     // paired-tile addressing where the map allows it
        return rc != WSI_EINVAL ? rc : launch_slab3<4, 2, 2, 3, 3, true>(a, st);
    }
    // cfg 39: 512 px x 64 couts (4 x 2 waves) for 64-channel layers on maps wider than 128 (the U-Net decoder's last level): a
    // 256-pixel tile of a 256-wide map is ONE row under a three-row slab; two rows per tile cut the halo from 3x to 2x
    if (cfg == 39) return planes == 3 ? launch_slab3<4, 4, 2, 3, 1, true>(a, st) : planes == 2 ? launch_slab3<4, 4, 2, 2, 1, true>(a, st) : launch_slab3<4, 4, 2, 1, 1, true>(a, st);
#endif
    return WSI_EINVAL;
}

// default config per layer shape (tuned on MI355X, tools/tune_conv.py)
int g_wide_min_c = 128;                                  // channel count from which the wide kernel (cfg 60) is the default
                                                         // (r01: 3-8 % faster than cfg 30 on layers 2-4; A/B via wsi_conv_set_mode)
static int slab_default_cfg(const ConvArgs& a, int planes, bool fallback) {           // r01 / r02 tunes: profiles/r0*_tune_conv*.log
    if (a.go.C % 64) return fallback ? 91 : 90;              // 32 output channels
    // fused upsample + concat input (U-Net decoder, ConvArgs.in_up): the slab3 kernels have the two-source slab DMA; on the
    // 128-multiple shapes they are ~8 % slower than the wide kernel (r04 tune) and save the pass that writes the concatenated tensor
    if (a.in_up) return a.go.C % 128 == 0 ? 30 : (a.gi.W > 128 && !fallback ? 39 : 31);
    // maps up to 4 x 4 (64 x 64 crops of the region-bag path: 25-56 % of a slab is padding): the small slab3 tiles keep four
    // workgroups per CU where the wide / ping-pong slabs leave one (r02 tune, n = 32000: 1.13 vs 1.28 ms at 4 x 4, 0.94 vs 1.15 at 2 x 2)
    if (a.gi.W <= 4 && a.go.C % 128 == 0 && planes == 3) return 30;
    // r03 (fp6 line format, residual through the matrix pipe): in mx the 4-wave wide kernel - two workgroups per CU, so one
    // workgroup's tail runs beside the other's main loop - ties the ping-pong kernel without a residual and beats it by 2-3 %
    // with one (profiles/r03_tune_conv_mx.log).  Single-pass bf16 mode:
    // (r03 tune of that mode, n = 2000, with / without residual: wide 0.60 / 0.70 (layer 2), 0.50 / 0.56, 0.47 / 0.50 ms against
    // slab3 0.62 / 0.74 and ping-pong 0.52 / 0.59, 0.47 / 0.50: the wide kernel everywhere except 8x8 maps, where the two tie)
    // (r04: on 8x8 maps in single-pass bf16 the wide kernel with asm DMA now beats the ping-pong kernel too: 0.499 / 0.471 ms against
    // 0.507 / 0.477, profiles/r04_tune_parity.log - the rule that sent them to cfg 70 is gone)
    // parity mode (r02 tune, n = 2000): the ping-pong kernel in its 256 px x 128 couts shape on layers 3-4 (1.48 / 1.38 vs 1.55 / 1.43 ms
    // for the wide kernel), slab3 on layer 2 (1.69 vs 1.78 ms)
    // (r04: with its main-loop DMA through inline asm the wide kernel wins in parity mode too - n = 2000, with / without residual:
    // layer 2 1.569 / 1.470 ms against slab3 1.628 / 1.528, layer 3 1.363 / 1.318 against ping-pong 1.434 / 1.363, layer 4 1.257 / 1.237
    // against 1.288 / 1.256, profiles/r04_tune_parity.log - so the two parity rules of r02 are gone and the line below decides)
    if (a.go.C % 128 == 0 && a.go.C >= g_wide_min_c && !(planes == 1 && a.gi.W > 33)) return 60;
    if (a.go.C % 128 != 0 && a.gi.W > 128 && !fallback) return 39;       // r02 tune, C = 64 at 256 x 256: 0.94 vs 1.21 ms (cfg 31); at 128 x 128 cfg 31 wins
    if (a.go.C == 64 && a.gi.C == 64 && planes == 3 && !fallback && g_l1_rows && g_l1p && a.gi.W == 64 && a.gi.H % 4 == 0 && !a.in2 && (a.flags & CONV_IN96) &&
        (!a.resid || (a.flags & CONV_RESID96)))
        return 42;                                           // r05 study route (wsi_conv_set_mode +1048576): persistent producer-fed kernel, measured 45 % SLOWER than cfg 40
    if (a.go.C % 128 != 0 && planes == 3 && !fallback && g_l1_rows && a.gi.W == 64 && a.gi.H % 4 == 0 && !a.in2) return 40;   // r04: row-stacked tiles (A/B: wsi_conv_set_mode +1024 off)
    // r04: with the scheduling fences the two-waves-per-SIMD form (cfg 31) beats the fence-less 168-register form (cfg 38) on
    // 16 x 16 maps (cfg4's layer 1, n = 32000: 1.552 vs 1.693 ms); 64-wide maps whose height is no multiple of four keep cfg 38
    // (1.545 vs 1.590 ms; profiles/r04_tune_slab3_fences.log)
    if (a.go.C % 128 != 0 && planes == 3 && !fallback && a.gi.W != 64) return 31;
    if (a.go.C % 128 != 0 && planes == 3 && !fallback) return 38;         // r03 tune, layer 1: 1.416 / 1.619 ms vs 1.435 / 1.666 (cfg 31), n = 2000
    return a.go.C % 128 == 0 ? 30 : 31;
}

// Host dispatch.  cfg < 0 selects the tuned default.
int wsi_conv_dispatch(const ConvArgs& a, int planes, int cfg, hipStream_t st) {
    const int cout = a.go.C;
    if (planes < 1 || planes > 3) return WSI_EINVAL;
    const int cmul = planes == 1 ? 64 : 32;                  // whole 128-byte lines; 32-channel tensors: stride-1 3x3 only (cfg 90)
    if (a.gi.C % cmul || cout % cmul) return WSI_EINVAL;
    if ((a.gi.C % 64 || cout % 64) && !(a.ksize == 3 && a.stride == 1)) return WSI_EINVAL;
    if (planes == 3 && !(a.ksize == 3 && a.stride == 1)) return WSI_EINVAL;      // mode 3: slab kernels only
    if (a.ksize == 3 && a.stride == 1) {
        if (a.gi.H != a.go.H || a.gi.W != a.go.W || a.gi.N != a.go.N) return WSI_EINVAL;
        if (cfg >= 0) return wsi_slab_dispatch_cfg(a, planes, cfg, st);
        const int rc = wsi_slab_dispatch_cfg(a, planes, slab_default_cfg(a, planes, false), st);
        return rc != WSI_EINVAL ? rc : wsi_slab_dispatch_cfg(a, planes, slab_default_cfg(a, planes, true), st);   // (in_up: EINVAL unless the shape's kernel is slab3)
    }
    if ((a.ksize == 3 || a.ksize == 1) && (a.stride == 1 || a.stride == 2)) {
        if (a.go.H * a.stride != a.gi.H || a.go.W * a.stride != a.gi.W || a.gi.N != a.go.N) return WSI_EINVAL;
        if (cout % 128 == 0) return planes == 2 ? launch_gather<8, 1, 4, 2>(a, st) : launch_gather<8, 1, 4, 1>(a, st);
        return planes == 2 ? launch_gather<4, 2, 2, 2>(a, st) : launch_gather<4, 2, 2, 1>(a, st);
    }
    return WSI_EINVAL;
}
