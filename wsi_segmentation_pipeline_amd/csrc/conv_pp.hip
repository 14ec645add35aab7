// Stride-1 3x3 convolution + folded BN (+ residual) (+ ReLU), "ping-pong" form for layers with >= 128 output channels
// (/root/reference/resnets_shift.py:41-45,49-65 in eval mode): the dominant kernel of the trunk.
//
// r01's wide kernel (conv.hip) ran two 4-wave workgroups per CU with one `vmcnt(0)` + barrier per tap, the slab of every
// 32-channel line loaded with nothing behind it inside the workgroup, and every wave's fragment reads exposed right after
// each barrier: its counters showed the MFMA pipe busy 34-50 % and the waves waiting 37-46 % of their lifetime.  This form
// keeps the same data layout, weights, operand formats and per-wave register tile (64 couts x MT*32 pixels, so a
// pixel-fragment set still feeds six MFMA steps) but changes the schedule:
//   * ONE 8-wave workgroup per CU = two groups of four waves (one wave of each group per SIMD).  The groups run half a
//     tap apart (group B passes one extra barrier before the loop): while group A multiplies tap t out of registers
//     (pure MFMA, raised priority), group B reads ITS fragments of tap t from LDS and issues DMA, then they swap - the
//     matrix pipe of every SIMD always has one wave in its multiply phase and LDS reads never sit in front of an MFMA.
//   * the pixel slab is double-buffered: group B fetches the slab of line c+1 by LDS-DMA in eight slices behind the
//     taps of line c; group A fetches the weight stage of tap t+1 during tap t (double-buffered).  The two kinds of DMA
//     are issued by different waves, so each wave's in-order vmcnt tracks one kind; nobody ever waits for a DMA that has
//     had less than a full multiply phase to land.
//   * barriers are raw s_barrier (no vmcnt drain); LDS-DMA stays in flight across them.
// Workgroup tile: WM x WN waves, wave = 64 couts x MT*32 px.  (2, 4, 4) = 256 px x 256 couts serves layers 3-4,
// (4, 2, 3) = 384 px x 128 couts layer 2: both fit two slabs + two weight stages in the 160 KB LDS.
// Bit-identical to the wide / slab3 kernels (same per-output accumulation order: lines, taps, K fragments).
#include "conv_dev.h"

// raw barriers (LDS-DMA stays in flight across them); the _VM form first drains the wave's own DMA.  No control flow may
// sit between a multiply phase and its barrier: hipcc then sinks MFMAs past the barrier into the next load phase.
#define PP_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_BARRIER_VM() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

#ifdef WSI_STUDY
#define PP_STAMP(x) do { if constexpr (STAMP) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); x; } } while (0)
#else
#define PP_STAMP(x) do { } while (0)
#endif

// STAMP (study builds only): workgroups 0 and 300 time their phases with s_memtime and write, per wave, the summed cycles of
// {load phase, wait at its barrier, multiply phase, wait at its barrier} to a.out2 (8 waves x 4 u64 per stamped workgroup)
template <int PLANES, int WM, int WN, int MT, bool STAMP = false, int DMODE = 0, bool PRIO = false>
__global__ __launch_bounds__(512, 2) void conv3x3s1_pp_kernel(ConvArgs a, int xbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 2, BM = WM * MT * 32, NTILES = WN * NT, WB = NTILES * 4096;
    constexpr int NF = 4;                                     // 16-byte fragments per operand set
    constexpr int RESID_NBUF = 4;                             // mode 3: residual tiles in flight per wave (epilogue; 128 KB of the dead LDS)
    static_assert(WM * WN == 8, "eight waves: two groups of four, one wave of each group per SIMD");
    char* const wl = smem;                                    // 2 weight stages of NTILES x 4 KB
    char* const xl = smem + 2 * WB;                           // 2 pixel slabs of xbytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                // 0: group A (weight DMA), 1: group B (slab DMA)
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int nblocks = a.go.C / (NTILES * 32);
    int nb = blockIdx.x % nblocks, mtile = blockIdx.x / nblocks;
    if (a.flags & CONV_XCD_RANGES) {                          // XCD-contiguous tile ranges (see conv3x3s1_slab3_kernel)
        const int chunk = gridDim.x >> 3, lin = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        nb = lin % nblocks;
        mtile = lin / nblocks;
        if (mtile >= (a.gi.N * a.gi.H * a.gi.W + BM - 1) / BM) return;
    }
    const int P = a.gi.P;
    const int NC = a.gi.C / PFmt<PLANES>::CPL;
    const size_t in_pixstride = (size_t)a.gi.C * PFmt<PLANES>::BPC;
    int xoff[MT], qs[MT];
    bool valid[MT];
    int slab0, npieces;
    const int lpix = dense_lane_pixel(l31);                   // lane -> pixel of a 32-pixel tile: conv_dev.h (LDS bank conflicts)
    {
        const int HW = a.gi.H * a.gi.W, R = a.gi.N * HW;
        auto pos = [&](int i) { return pf_pos_of_index(a.gi, i); };
        const int i0 = mtile * BM, i1 = min(i0 + BM, R) - 1;
        slab0 = pos(i0) - P - 1;
        npieces = (pos(i1) + P + 1 - slab0 + 1) * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int i = i0 + wm * MT * 32 + mt * 32 + lpix;
            valid[mt] = i < R;
            qs[mt] = pos(valid[mt] ? i : i1);
            xoff[mt] = qs[mt] - slab0 - (P + 1);
        }
    }
    // Both DMA streams use buffer addressing: one per-lane byte offset each (computed once), everything that changes with
    // the line / tap / round is a scalar offset - no 64-bit per-lane address arithmetic inside the loop.
    // group A: weights of (line c, tap t) for the workgroup's NTILES channel tiles -> stage wb; 256 pieces of 16 B per tile
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wpk + (size_t)(nb * NTILES) * NC * 9 * 4096), 0, NTILES * NC * 9 * 4096, 0x00020000);
    const int wvoff = (tid & 255) * 16;
    auto wdma = [&](int c, int t, char* wb, int half) {       // channel tiles [half * NTILES / 2, (half + 1) * NTILES / 2); half 2: all
#pragma unroll
        for (int j = 0; j < NTILES; ++j)
            if (half == 2 || (j >= NTILES / 2) == (half == 1))
                dma16_buf(wrs, wb + j * 4096 + (wave & 3) * 1024, wvoff, ((j * NC + c) * 9 + t) * 4096);
    };
    // slab pieces: piece i = 16-byte slot (i & 7) of slab pixel i >> 3, stored swizzled (source-side XOR, like every slab
    // kernel).  A round of 256 pieces is 32 pixels, so the swizzle term ((Pl >> 1) & 7) does not depend on the round:
    // per-lane offset of round 0 + scalar 32 * round * pixstride.
    const size_t slab_byte0 = (size_t)slab0 * in_pixstride;
    const size_t in_bytes = (size_t)pf_alloc_pixels(a.gi.N, a.gi.H, a.gi.W) * in_pixstride;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.in + slab_byte0), 0, (int)min(in_bytes - slab_byte0, (size_t)0x7fffffff), 0x00020000);
    int xvoff, xvoff8;                                        // group B rounds (4 waves) / prologue rounds (8 waves, 64 pixels)
    {
        const int i = (wave & 3) * 64 + lane, Pl = i >> 3, sl = (i & 7) ^ ((Pl >> 1) & 7);
        xvoff = Pl * (int)in_pixstride + sl * 16;
        const int i8 = wave * 64 + lane, Pl8 = i8 >> 3, sl8 = (i8 & 7) ^ ((Pl8 >> 1) & 7);
        xvoff8 = Pl8 * (int)in_pixstride + sl8 * 16;
    }
    const int round_bytes = 32 * (int)in_pixstride;           // source bytes between two 256-piece rounds

    f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (PLANES == 3) acc_init_bias<MT>(acc[nt], a.bias, nb * NTILES + wn * NT + nt, lane);   // mode 3: start from the folded BN bias
        else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
        }
    }

    // LDS byte addresses of one tap's pixel fragments (and scale dwords), relative to smem: computed one step ahead,
    // inside the previous multiply phase (VALU beside MFMA is nearly free; in the load phase these ~12 instructions per
    // tile row delayed the reads: r02 counters)
    int xaddr[MT][NF];
    auto tap_addrs = [&](int slab_off, int toff) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int xo = xoff[mt];
            asm volatile("" : "+v"(xo));                      // opaque: computed HERE, every step (hipcc otherwise hoists all nine
            const int Pl = xo + toff;                         // taps' addresses out of the loop and spills them)
            const int base = lds_xbase(Pl, h);
#pragma unroll
            for (int f = 0; f < NF; ++f) xaddr[mt][f] = slab_off + (base ^ (f << 5));
            // ... and finished here (a second pin: hipcc otherwise sinks the arithmetic down to the reads, past the barrier)
            asm volatile("" : "+v"(xaddr[mt][0]), "+v"(xaddr[mt][1]), "+v"(xaddr[mt][2]), "+v"(xaddr[mt][3]));
        }
    };
    const int wlane = (wn * NT) * 4096 + lane * 16;           // this lane's slot inside a weight stage

    // prologue: slab of line 0 (all waves) + weight stage of step 0 (both groups, half the channel tiles each)
    for (int i0 = wave * 64, r = 0; i0 < npieces; i0 += 512, ++r) dma16_buf(xrs, xl + (size_t)i0 * 16, xvoff8, 2 * r * round_bytes);
    wdma(0, 0, wl, grp);
    if constexpr (DMODE == 1) wdma(0, 1, wl + WB, grp);       // (all later stages: group B's multiply phases, two steps ahead)
    if constexpr (DMODE == 2) { if (grp == 1) wdma(0, 1, wl + WB, 1); }   // B's half of step 1 (A requests its half in load phase 0)
    tap_addrs(2 * WB, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    if (grp == 1) PP_BARRIER();                               // group B runs half a tap behind group A from here on

    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    (void)tsum; (void)tprev;
    PP_STAMP(tprev = t_);
    for (int c = 0; c < NC; ++c) {
        char* xnext = xl + (size_t)((c + 1) & 1) * xbytes;
        int Pc = P;
        asm volatile("" : "+s"(Pc));                          // opaque per line: keeps later taps' address arithmetic out of registers
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const char* wcur = wl + ((c + t) & 1) * WB;       // step g = 9c + t: stage parity (c + t) & 1
            char* wnxt = wl + ((c + t + 1) & 1) * WB;
            // ---------------- load phase: this group's half of the next step's weight stage (+ group A: two slices of the
            // next line's slab), then every fragment of this tap into registers
            auto slab_slices = [&]() {                        // rounds t and t + 8 of the next line's slab (<= 16 rounds: 64 KB)
                if (c + 1 < NC && t < 8) {
                    const int i0 = t * 256 + (wave & 3) * 64;
                    if (i0 < npieces) dma16_buf(xrs, xnext + (size_t)i0 * 16, xvoff, t * round_bytes + (c + 1) * 128);
                    if (i0 + 2048 < npieces) dma16_buf(xrs, xnext + (size_t)(i0 + 2048) * 16, xvoff, (t + 8) * round_bytes + (c + 1) * 128);
                }
            };
            auto next_weights = [&](int half) {               // this wave's share of the NEXT step's stage (half 2 = all tiles)
                if (t < 8) wdma(c, t + 1, wnxt, half);
                else if (c + 1 < NC) wdma(c + 1, 0, wnxt, half);
            };
            if constexpr (DMODE == 0) {                       // both groups: their half of the weights in the load phase; A: slab
                next_weights(grp);
                if (grp == 0) slab_slices();
            } else if constexpr (DMODE == 2) {                // A: its half + slab in the load phase (B's half: B's multiply phases)
                if (grp == 0) { next_weights(0); slab_slices(); }
            }
            PP_STAMP(tsum[4] += t_ - tprev; tprev = t_);      // DMA issue part of the load phase
            bf16x8 wf[NT][4], xf[MT][4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const char* src = wcur + wlane + nt * 4096;
#pragma unroll
                for (int f = 0; f < NF; ++f) wf[nt][f] = *(const bf16x8*)(src + f * 1024);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int f = 0; f < NF; ++f) xf[mt][f] = *(const bf16x8*)(smem + xaddr[mt][f]);
            }
            // group B's share of the next stage must be complete before group A reads it, one barrier from here (B has been
            // waiting for its fragments meanwhile); group A's share and slab slices get the whole multiply phase below
            if (grp == 1 || DMODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (DMODE 1: what group A requested beside its last MFMAs: slab slices)
            PP_STAMP(tsum[0] += t_ - tprev; tprev = t_);
            PP_BARRIER();
            PP_STAMP(tsum[1] += t_ - tprev; tprev = t_);
            // ---------------- multiply phase: registers only (+ the address arithmetic of the next step)
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
            // DMODE 1 / 2: DMA issued beside the MFMAs.  The stage of step g+2 may be requested by group B here: its buffer
            // (that of step g) was last read in B's own load phase just before; it lands during the next two phases and is
            // waited for at the end of B's next load phase.  Group A requests the slab slices here.
            if constexpr (DMODE != 0) {
                if (grp == 1) {
                    char* w2 = wl + ((c + t) & 1) * WB;       // stage of step g + 2 = the one just consumed
                    const int half = DMODE == 1 ? 2 : 1;
                    if (t < 7) wdma(c, t + 2, w2, half);
                    else if (c + 1 < NC) wdma(c + 1, t - 7, w2, half);
                } else if (DMODE == 1) {
                    slab_slices();
                }
            }
            if (t < 8) tap_addrs(2 * WB + (c & 1) * xbytes, ((t + 1) / 3) * Pc + ((t + 1) % 3));
            else tap_addrs(2 * WB + ((c + 1) & 1) * xbytes, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mfma_step<PLANES>(acc[nt][mt], wf[nt], xf[mt]);
            // pin every accumulator here: without a data dependence hipcc sinks the MX MFMAs (readnone calls) below the
            // barrier statement, into the next load phase
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(acc[0][mt]), "+v"(acc[1][mt]));
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            PP_STAMP(tsum[2] += t_ - tprev; tprev = t_);
            // group A's load-phase requests have had this multiply phase to land; DMA requested DURING a multiply phase is
            // waited for at the end of the requester's next load phase instead
            if (grp == 0 && DMODE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PP_BARRIER();
            PP_STAMP(tsum[3] += t_ - tprev; tprev = t_);
        }
    }
#ifdef WSI_STUDY
    if constexpr (STAMP) {
        if ((blockIdx.x == 0 || blockIdx.x == 300) && lane == 0 && a.out2) {
            unsigned long long* o = (unsigned long long*)a.out2 + ((blockIdx.x ? 1 : 0) * 8 + wave) * 6;
            for (int k = 0; k < 6; ++k) o[k] = tsum[k];
        }
    }
#endif
    if (grp == 0) PP_BARRIER();                               // group A's matching extra barrier
    if constexpr (PLANES == 3) {                              // weight stages and slabs are dead: residual staging, NBUF tiles per wave
        constexpr int NBUF = RESID_NBUF % MT ? MT : (RESID_NBUF < NT * MT ? RESID_NBUF : NT * MT);
        conv_tail_mx<NT, MT, NBUF>(a, acc, qs, valid, nb * NTILES + wn * NT, lane, smem + wave * (RESID_NBUF * 4096), slab0);
    } else {
        char* scratch = nullptr;
        if (PLANES == 2 && a.resid && !(a.flags & CONV_RESID_DIRECT)) scratch = xl + wave * 8192;   // slabs are dead: residual staging
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) conv_epilogue_q<MT, PLANES>(a, acc[nt], qs, valid, nb * NTILES + wn * NT + nt, lane, scratch);
    }
}

long long dense_max_slab_pixels(const ConvArgs& a, int BM);

template <int PLANES, int WM, int WN, int MT, bool STAMP = false, int DMODE = 0, bool PRIO = false>
static int launch_pp(const ConvArgs& a, hipStream_t st) {
    constexpr int BM = WM * MT * 32, BN = WN * 2 * 32;
    if (a.go.C % BN) return WSI_EINVAL;
    const int nblocks = a.go.C / BN;
    const long long R = (long long)a.gi.N * a.gi.H * a.gi.W;
    const int mtiles = (int)((R + BM - 1) / BM);
    size_t xbytes = (size_t)((dense_max_slab_pixels(a, BM) * 8 + 63) / 64 * 64) * 16;         // whole 1 KB DMA instructions
    if (2 * xbytes < 8 * 8192) xbytes = 8 * 8192 / 2;                                          // residual staging of the epilogue (mode 2: in the slabs)
    size_t lds = 2 * (size_t)(BN / 32) * 4096 + 2 * xbytes;
    if (PLANES == 3 && lds < 8 * 4 * 4096) lds = 8 * 4 * 4096;                                // mode 3: four residual tiles per wave, from smem + 0
    if (lds > 160 * 1024 || xbytes > 65536) return WSI_EINVAL;                                 // (group B moves a slab in <= 16 rounds)
    auto k = conv3x3s1_pp_kernel<PLANES, WM, WN, MT, STAMP, DMODE, PRIO>;
    static bool lds_ok = false;                              // once per instantiation: up to the whole 160 KB
    if (!lds_ok) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return WSI_EINVAL;
        lds_ok = true;
    }
    const int grid = (a.flags & CONV_XCD_RANGES) ? (mtiles * nblocks + 7) / 8 * 8 : mtiles * nblocks;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a, (int)xbytes);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

// cfg 70: 256 px x 256 couts (2 x 4 waves, MT 4); cfg 71: 384 px x 128 couts (4 x 2 waves, MT 3); cfg 72: 512 px x 128 couts
// (4 x 2 waves, MT 4; small maps only: two 512-pixel slabs must fit)
int wsi_pp_dispatch(const ConvArgs& a, int planes, int cfg, hipStream_t st) {
    switch (cfg) {
    case 70: return planes == 3 ? launch_pp<3, 2, 4, 4>(a, st) : planes == 2 ? launch_pp<2, 2, 4, 4>(a, st) : launch_pp<1, 2, 4, 4>(a, st);
    case 71: return planes == 3 ? launch_pp<3, 4, 2, 3>(a, st) : planes == 2 ? launch_pp<2, 4, 2, 3>(a, st) : launch_pp<1, 4, 2, 3>(a, st);
    case 73: return planes == 3 ? launch_pp<3, 2, 4, 4, false, 1>(a, st) : planes == 1 ? launch_pp<1, 2, 4, 4, false, 1>(a, st) : WSI_EINVAL;
    case 74: return planes == 3 ? launch_pp<3, 2, 4, 4, false, 2>(a, st) : planes == 1 ? launch_pp<1, 2, 4, 4, false, 2>(a, st) : WSI_EINVAL;
    case 77: return planes == 3 ? launch_pp<3, 4, 2, 3, false, 1>(a, st) : planes == 1 ? launch_pp<1, 4, 2, 3, false, 1>(a, st) : WSI_EINVAL;
    case 78: return planes == 3 ? launch_pp<3, 4, 2, 3, false, 2>(a, st) : planes == 1 ? launch_pp<1, 4, 2, 3, false, 2>(a, st) : WSI_EINVAL;
#ifdef WSI_STUDY
    case 75: return planes == 3 ? launch_pp<3, 2, 4, 4, true>(a, st) : WSI_EINVAL;       // cfg 70 with phase stamps (a.out2 = 512-byte debug buffer)
    case 76: return planes == 3 ? launch_pp<3, 4, 2, 3, true>(a, st) : WSI_EINVAL;       // cfg 71 with phase stamps
    case 79: return planes == 3 ? launch_pp<3, 8, 1, 1>(a, st) : WSI_EINVAL;             // 256 px x 64 couts (layer 1 study)
    case 82: return planes == 2 ? launch_pp<2, 2, 4, 3>(a, st) : WSI_EINVAL;             // parity mode: 192 px x 256 couts (MT 4 spills)
    case 84: return planes == 2 ? launch_pp<2, 2, 4, 2>(a, st) : WSI_EINVAL;             // parity mode: 128 px x 256 couts
#endif
    case 83: return planes == 2 ? launch_pp<2, 4, 2, 2>(a, st) : WSI_EINVAL;             // parity mode: 256 px x 128 couts (MT 4 spills there)
    case 72: return planes == 3 ? launch_pp<3, 2, 4, 4, false, 0, true>(a, st) : planes == 1 ? launch_pp<1, 2, 4, 4, false, 0, true>(a, st) : WSI_EINVAL;   // A/B: raised priority in the multiply phase
    }
    return WSI_EINVAL;
}
