// Device helpers shared by the convolution kernels (conv.hip, conv_pp.hip): LDS-DMA wrappers, the swizzled LDS pixel
// image, the per-precision MFMA step and the fused epilogues (bias + residual + ReLU + operand split for the next layer).
// Replaces the tail of BasicBlock.forward (/root/reference/resnets_shift.py:57-63: bn2, += residual, relu) in eval mode.
#pragma once
#include "common.h"

// --------------------------------------------------------------------------------------------
// Fused epilogue: bias (+ residual) (+ ReLU), split to bf16 planes, store.  acc[mt] covers
// pixels q_base + mt*32 + (lane&31) and channels ntile*32 + 8g + 4h + i.
// --------------------------------------------------------------------------------------------
// 16-byte LDS-DMA: lane i writes LDS [lds_wave_base + 16*i] from its own global address.
static __device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Buffer-addressed form: source = resource base + 32-bit per-lane byte offset + scalar offset (no
// 64-bit address VALU).  Kept in a plain __device__ function: used directly inside a kernel
// template, this builtin makes hipcc's host pass drop the kernel stub (ROCm 7.2).
static __device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// The same transfer issued from inline asm, for DMA that stays in flight WHILE the wave reads other parts of the LDS (weight stages
// requested one tap ahead, slabs one line ahead).  r04 finding: with an LDS-DMA builtin pending, hipcc's wait-count pass stops
// counting LDS reads - every `s_waitcnt lgkmcnt(N)` in the loop becomes `lgkmcnt(0)`, so pixel fragments requested one step ahead
// are awaited right after their request (the wide kernel's main loop held 6 such full waits per tap where the same loop without
// the pending DMA has lgkmcnt(4) / (7) / (11)).  An asm DMA is invisible to that pass: the reads keep their counted waits, and the
// DMA is awaited where the kernel says so (`s_waitcnt vmcnt(0)` + barrier before the stage is read - the kernels did that
// explicitly already).  Any vmcnt wait the compiler computes for its own loads only gets more conservative by unknown younger
// requests (in-order completion), never unsafe.  `lds` = byte address of the wave's 1 KiB destination (wave-uniform).
// M0: hipcc treats m0 as a reserved register - a "m0" clobber is rejected with a warning and changes nothing - while its own
// LDS-DMA builtins (the in2 segment, the residual staging of the tails) keep m0 values alive and merge equal initialisations
// (SIFixSGPRCopies).  So the asm leaves m0 exactly as it found it: saved to a scratch SGPR, set, used, restored (M0 is read
// when the DMA issues; two scalar moves per transfer).
static __device__ __forceinline__ unsigned lds_addr_of(const char* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
static __device__ __forceinline__ void dma16_asm(const void* gsrc, unsigned lds) {       // per-lane 64-bit source address
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(gsrc) : "memory");
}
static __device__ __forceinline__ void dma16_buf_asm(__amdgpu_buffer_rsrc_t rs, unsigned lds, int voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// Residual tile (32 pixels x one 128-byte line) -> 4 KB of LDS at dst by LDS-DMA, eight lanes per line: piece
// i = j*64 + lane is slot (i & 7) of tile pixel i >> 3, stored swizzled like the pixel slabs (source-side XOR).  q = this
// lane's own pixel (PF index; lanes p and p+32 hold the same); the owning lanes hand it out by ds_bpermute.
static __device__ __forceinline__ void resid_tile_dma(const void* resid, int q_own, size_t pixstride, size_t line_off, int lane, char* dst) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pp = 8 * j + (lane >> 3);
        const int q = __shfl(q_own, pp);
        const int sl = (lane & 7) ^ ((pp >> 1) & 7);
        dma16((const char*)resid + (size_t)q * pixstride + line_off + sl * 16, dst + j * 1024);
    }
}

// Buffer-addressed form for the mode-3 tail: `rs` = the residual tensor from the first pixel of the workgroup's slab on,
// q_rel = this lane's pixel relative to it (both tensors of a stride-1 conv share one geometry), so the per-lane offset
// is 32-bit arithmetic and the channel-tile offset is scalar.
static __device__ __forceinline__ void resid_tile_dma_buf(__amdgpu_buffer_rsrc_t rs, int q_rel, int pixstride, int line_off, int lane, char* dst) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pp = 8 * j + (lane >> 3);
        const int q = __shfl(q_rel, pp);
        const int sl = (lane & 7) ^ ((pp >> 1) & 7);
        dma16_buf(rs, dst + j * 1024, q * pixstride + sl * 16, line_off);
    }
}

// ---- 96-byte lines (common.h CONV_IN96 / OUT96 / RESID96) -------------------------------------------------------------
// Logical LDS slot s of a line <-> 16-byte piece of the 96-byte memory line: slots 0-4 -> pieces 0-4, slot 6 -> piece 5,
// slots 5 and 7 (the hi6 plane) are not stored.
static __device__ __forceinline__ bool mx96_stored(int s) { return s != 5 && s != 7; }
static __device__ __forceinline__ int mx96_piece(int s) { return s == 6 ? 5 : s; }

// Rebuild the hi6 plane of `npix` pixel-lines of a swizzled slab image in LDS (128-byte pitch, slot s of pixel Pl at slot
// s ^ ((Pl >> 1) & 7)) from their fp16 planes: fp6 field 2i = plane position i, field 2i + 1 = position 16 + i
// (common.h mx6_field_of_pos), scale = the producer's scale_hi byte (dword 3 of slot 6), all 32 values by ONE
// v_cvt_scalef32_pk32_fp6_f16 - bit-identical to what the producers' f32 path stores in a 128-byte line
// (tools/probes/fp6_from_f16.hip: every fp16 bit pattern, every scale).  One thread per pixel-line; the caller fences with
// barriers on both sides.
static __device__ __forceinline__ void mx96_rebuild_hi6(char* slab, int npix, int tid, int nthreads) {
    for (int Pl = tid; Pl < npix; Pl += nthreads) {
        const int key = (Pl >> 1) & 7;
        char* line = slab + Pl * 128;
        unsigned D[16];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const u32x4 d = *(const u32x4*)(line + ((s4 ^ key) << 4));
            D[4 * s4] = d[0]; D[4 * s4 + 1] = d[1]; D[4 * s4 + 2] = d[2]; D[4 * s4 + 3] = d[3];
        }
        const int sh = (int)(*(const unsigned*)(line + ((6 ^ key) << 4) + 12) & 255u);
        u32x16 e;
#pragma unroll
        for (int k = 0; k < 16; ++k)                          // register k = {position k, position 16 + k}
            e[k] = __builtin_amdgcn_perm(D[8 + (k >> 1)], D[k >> 1], (k & 1) ? 0x07060302u : 0x05040100u);
        u32x6 q;
        const float scale = sh ? mx_scale_value(sh) : 1.f;
        asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(q) : "v"(e), "v"(scale));
        *(u32x4*)(line + ((5 ^ key) << 4)) = u32x4{q[0], q[1], q[2], q[3]};
        *(u32x4*)(line + ((7 ^ key) << 4)) = u32x4{q[4], q[5], (unsigned)sh, 0u};
    }
}

// resid_tile_dma_buf for a residual stored in 96-byte lines: the lanes of the two hi6 slots stay idle
static __device__ __forceinline__ void resid_tile_dma_buf96(__amdgpu_buffer_rsrc_t rs, int q_rel, int pixstride, int line_off, int lane, char* dst) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pp = 8 * j + (lane >> 3);
        const int q = __shfl(q_rel, pp);
        const int sl = (lane & 7) ^ ((pp >> 1) & 7);
        if (mx96_stored(sl)) dma16_buf(rs, dst + j * 1024, q * pixstride + mx96_piece(sl) * 16, line_off);
    }
}

template <int MT, int PLANES>
static __device__ __forceinline__ void conv_epilogue_q(const ConvArgs& a, f32x16 (&acc)[MT], const int (&qs)[MT],
                                                       const bool (&valid)[MT], int ntile, int lane, char* scratch = nullptr) {
    typedef typename PairElem<PLANES>::T E;                                       // fp16 in mode 2 (fp16 pair), bf16 in mode 1
    typedef __attribute__((ext_vector_type(4))) E Ex4;
    const int h = lane >> 5, l31 = lane & 31;
    const size_t pixstride = (size_t)a.go.C * PFmt<PLANES>::BPC;
    const size_t chan_off = (size_t)ntile * (32 * PFmt<PLANES>::BPC) + (size_t)(4 * h) * 2;
    size_t poff[MT], ooff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        poff[mt] = (size_t)(valid[mt] ? qs[mt] : a.go.G) * pixstride + chan_off;   // invalid rows read a real pixel, store nothing
        ooff[mt] = a.out_split_pixels ? pf_out_offset(a.go, a.out_split_pixels, valid[mt] ? qs[mt] : a.go.G, pixstride) + chan_off : poff[mt];
    }

    // residual: with `scratch` (8 KB of wave-private LDS, split precision) tile by tile through LDS-DMA, line-contiguous
    // (see conv_epilogue_mx); otherwise every residual load of the tile in flight at once (branch-free)
    const bool via_lds = PLANES == 2 && a.resid && scratch;
    Ex4 rh[MT][4], rl[MT][4];
    if (via_lds) {
        resid_tile_dma(a.resid, valid[0] ? qs[0] : a.go.G, pixstride, (size_t)ntile * 128, lane, scratch);
    } else if (a.resid) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const char* rp = (const char*)a.resid + poff[mt] + 16 * g;
                if (CONV_STUDY(a, CONV_NONTEMPORAL)) {
                    rh[mt][g] = __builtin_nontemporal_load((const Ex4*)rp);
                    if constexpr (PLANES == 2) rl[mt][g] = __builtin_nontemporal_load((const Ex4*)(rp + 64));
                } else {
                    rh[mt][g] = *(const Ex4*)rp;
                    if constexpr (PLANES == 2) rl[mt][g] = *(const Ex4*)(rp + 64);
                }
            }
    }
    float bias[16], wsi[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[g * 4 + i] = a.bias[ntile * 32 + 8 * g + 4 * h + i];
    if constexpr (PLANES == 2) {                                                  // per-channel inverse weight scales (common.h conv_wscale_inv)
        const float* ws = conv_wscale_inv(a.wpk, a.go.C, a.gi.C, a.ksize) + ntile * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(ws + 8 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) wsi[g * 4 + i] = v[i];
        }
    }
    // phase 2: bias + residual + ReLU, split, store
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (via_lds) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // tile mt has landed (and the previous tile's stores)
            if (mt + 1 < MT)
                resid_tile_dma(a.resid, valid[mt + 1] ? qs[mt + 1] : a.go.G, pixstride, (size_t)ntile * 128, lane, scratch + ((mt + 1) & 1) * 4096);
            const char* t = scratch + (mt & 1) * 4096 + l31 * 128 + 8 * h;
            const int sw = (l31 >> 1) & 7;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                rh[mt][g] = *(const Ex4*)(t + ((g ^ sw) << 4));
                rl[mt][g] = *(const Ex4*)(t + (((4 + g) ^ sw) << 4));
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (PLANES == 2) v[i] = acc[mt][4 * g + i] * wsi[4 * g + i] + bias[4 * g + i];
                else v[i] = acc[mt][4 * g + i] + bias[4 * g + i];
            }
            if (a.resid) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += (float)rh[mt][g][i];
                if constexpr (PLANES == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)rl[mt][g][i];
                }
            }
            Ex4 hi, lo;
            if constexpr (PLANES == 2) {
                const float lo_clamp = a.relu ? 0.f : -65504.f;                    // ReLU (if any) + fp16-range clamp in one op
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = __builtin_amdgcn_fmed3f(v[i], lo_clamp, 65504.f);
                    hi[i] = (E)v[i];
                    lo[i] = (E)(v[i] - (float)hi[i]);
                }
            } else {
                if (a.relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hi[i] = (E)v[i];
                    lo[i] = (E)(v[i] - (float)hi[i]);
                }
            }
            if (valid[mt] && !CONV_STUDY(a, CONV_ABL_NO_STORE)) {
                char* op = (char*)a.out + ooff[mt] + 16 * g;
                if (CONV_STUDY(a, CONV_NONTEMPORAL)) {
                    __builtin_nontemporal_store(hi, (Ex4*)op);
                    if constexpr (PLANES == 2) __builtin_nontemporal_store(lo, (Ex4*)(op + 64));
                } else {
                    *(Ex4*)op = hi;
                    if constexpr (PLANES == 2) *(Ex4*)(op + 64) = lo;
                }
            }
        }
    }
}

// one (pixel tile, channel tile, 32-channel line, tap) step of mode 3: two fp16 MFMAs on the hi planes + one block-scaled
// fp6 MFMA whose K halves are the two cross terms (lanes h=0: W hi6 x X lo6, lanes h=1: W lo6 x X hi6).  w / x = the four
// 16-byte fragments of a line as this lane reads them (slots 2f + h): f 0, 1 fp16 k-steps; f 2 = dwords 0-3 of the
// lane's fp6 plane; f 3 = {dwords 4-5, scale byte, pad}.
static __device__ __forceinline__ f32x16 mfma_mx6(f32x16 d, const bf16x8 (&w)[4], const bf16x8 (&x)[4]) {
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, x[0]), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[1]), __builtin_bit_cast(f16x8, x[1]), d, 0, 0, 0);
    const i32x4 wq = __builtin_bit_cast(i32x4, w[2]), wr = __builtin_bit_cast(i32x4, w[3]);
    const i32x4 xq = __builtin_bit_cast(i32x4, x[2]), xr = __builtin_bit_cast(i32x4, x[3]);
    // fp6 operands are 192 bits: the instruction reads registers 0-5 of each operand tuple (the backend selects the 6-register
    // form for cbsz = blgp = 2), so the tuple is simply the two 16-byte fragments back to back - no zero filling, hence no copies
    // of a fragment into a fresh 8-register tuple (r04: those copies cost the row-stacked kernel 40 registers)
    const i32x8 wa = {wq[0], wq[1], wq[2], wq[3], wr[0], wr[1], wr[2], wr[3]}, xa = {xq[0], xq[1], xq[2], xq[3], xr[0], xr[1], xr[2], xr[3]};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, xa, d, 2, 2, 0, wr[2], 0, xr[2]);
}

// Accumulator start values of mode 3: the folded BN bias of this lane's 16 output channels (the epilogue then adds nothing)
template <int MT>
static __device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[MT], const float* bias, int ntile, int lane) {
    const int h = lane >> 5;
    float b[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(bias + ntile * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[4 * g + i] = v[i];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = b[r];
}

// Mode-3 epilogue (fp16 hi + MX-fp6): acc (bias and residual already inside: acc_init_bias, conv_tail_mx) -> clamp /
// ReLU -> the next layer's operands, for tiles mt0 .. mt0 + MTN - 1.  Lane (pixel, h) owns line positions 16h .. 16h+15 of the fp16 plane (32 contiguous
// bytes) and, after one exchange with lane ^ 32 (v_permlane32_swap per register), ALL 32 values of one fp6 plane (h = 0:
// lo6, h = 1: hi6), which one v_cvt_scalef32_2xpk16_fp6_f32 converts.  Four 16-byte stores per 32x32 tile; the whole
// 128-byte line is written.  (`a.resid` is not read here: conv_tail_mx adds the residual.)
template <int MT, int MTN = MT>
static __device__ __forceinline__ void conv_epilogue_mx(const ConvArgs& a, f32x16 (&acc)[MT], const int (&qs)[MT],
                                                        const bool (&valid)[MT], int ntile, int lane, int mt0 = 0) {
    const int h = lane >> 5;
    const bool o96 = a.flags & CONV_OUT96;                                            // 96-byte lines (ordinary PF only, never phase-split): line-planar
    const size_t pixstride = o96 ? (size_t)96 : (size_t)a.go.C * 4;
    const float lo_clamp = a.relu ? 0.f : -65504.f;
#pragma unroll
    for (int mt = mt0; mt < mt0 + MTN; ++mt) {
        const size_t loff = (size_t)(valid[mt] ? qs[mt] : a.go.G) * pixstride + (o96 ? (size_t)ntile * (size_t)a.plane96 : (size_t)ntile * 128);
        f32x16 v = acc[mt];
        f32x16 hi, lo;
        f16x8 hv[2];
        float mv = 0.f, ml = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float v0 = __builtin_amdgcn_fmed3f(v[r], lo_clamp, 65504.f);      // ReLU (if any) + fp16-range clamp in one op
            const float v1 = __builtin_amdgcn_fmed3f(v[r + 1], lo_clamp, 65504.f);
            const f16x2 hh = __builtin_convertvector(f32x2{v0, v1}, f16x2);          // v_cvt_pk_f16_f32 (RNE)
            hv[r >> 3][r & 7] = hh[0];
            hv[r >> 3][(r & 7) + 1] = hh[1];
            hi[r] = (float)hh[0];
            hi[r + 1] = (float)hh[1];
            lo[r] = v0 - hi[r];
            lo[r + 1] = v1 - hi[r + 1];
            mv = fmaxf(mv, fmaxf(fabsf(hi[r]), fabsf(hi[r + 1])));
            ml = fmaxf(ml, fmaxf(fabsf(lo[r]), fabsf(lo[r + 1])));
        }
        pair_max2(mv, ml);                                                            // block maxima over the pixel's two lanes
        const int sh = mx6_scale_byte(mv), sl = mx6_scale_byte(ml);
        const int sb = h ? sh : sl;                                                   // this lane's plane: h = 0 lo6, h = 1 hi6
        swap32_halves(lo, hi);                                                        // lo := (h ? partner's hi : own lo), hi := (h ? own hi : partner's lo)
        const u32x6 q = mx6_pack32(lo, hi, sb ? mx_scale_value(sb) : 1.f);
        if (valid[mt] && !CONV_STUDY(a, CONV_ABL_NO_STORE)) {
            char* ol = (char*)a.out + (a.out_split_pixels ? pf_out_offset(a.go, a.out_split_pixels, qs[mt], pixstride) + (size_t)ntile * 128 : loff);
            *(f16x8*)(ol + 32 * h) = hv[0];
            *(f16x8*)(ol + 32 * h + 16) = hv[1];
            if (!o96) {
                *(u32x4*)(ol + MX6_PLANE_LO(0) + 16 * h) = u32x4{q[0], q[1], q[2], q[3]};
                *(u32x4*)(ol + MX6_PLANE_HI(0) + 16 * h) = u32x4{q[4], q[5], (unsigned)sb, 0u};
            } else if (h == 0) {                                                      // the lo6 plane and both scales; hi6 is rebuilt by the reader
                *(u32x4*)(ol + 64) = u32x4{q[0], q[1], q[2], q[3]};
                *(u32x4*)(ol + 80) = u32x4{q[4], q[5], (unsigned)sl, (unsigned)sh};
            }
        }
    }
}

// Mode-3 tail of a conv kernel: (residual) + clamp / ReLU + line encode + store of all NT x MT accumulator tiles of a wave.
// Residual: acc[nt][mt] += residual tile, through the matrix pipe.  The tiles (32 pixels x one 128-byte line each) are
// fetched by LDS-DMA into `scratch` (NBUF x 4 KB, private to the wave), eight lanes per line, in the swizzled slab image
// (resid_tile_dma); a tile is then the B operand of one mode-3 MFMA step whose A operand is the identity: 1.0 at the K
// position of the lane's own channel in the fp16 plane and in the hi6 plane (K half 0, which multiplies the residual's
// lo6 plane), zeros in the lo6 plane - so the step adds exactly hi + lo6 * 2^(scale_lo - 127), the value the line stores.
// Tiles go in batches of NBUF (channel tile major): the DMA of batch b + 1 is in flight while batch b is encoded and
// stored, so a wave pays one exposed round trip to memory instead of one per tile (r02: vector decode of ~50 instructions
// per tile and 4-8 round trips in a row per wave, +20 % on a residual launch).  Only `vmcnt(0)` waits: spill code or
// stores the compiler places between the DMAs cannot break a counted wait.
template <int NT, int MT, int NBUF>
static __device__ __forceinline__ void conv_tail_mx(const ConvArgs& a, f32x16 (&acc)[NT][MT], const int (&qs)[MT], const bool (&valid)[MT],
                                                    int ntile0, int lane, char* scratch, int slab0) {
    constexpr int T = NT * MT;
    static_assert(MT % NBUF == 0 || NBUF % MT == 0, "a batch is part of one channel tile or whole channel tiles");
    static_assert(NBUF <= T && T % NBUF == 0, "whole batches");
    if (!a.resid) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) conv_epilogue_mx<MT>(a, acc[nt], qs, valid, ntile0 + nt, lane);
        return;
    }
    const int h = lane >> 5, l31 = lane & 31;
    const bool r96 = a.flags & CONV_RESID96;                    // the residual tensor has 96-byte lines (line-planar: ConvArgs.plane96)
    const int pixstride = r96 ? 96 : a.go.C * 4;
    // the residual from the slab's first pixel on (slab0 = the workgroup's first input pixel = a pixel of the residual too)
    const size_t rbase = (size_t)slab0 * pixstride, rbytes = (size_t)pf_alloc_pixels(a.go.N, a.go.H, a.go.W) * pixstride;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.resid + rbase), 0,
                                                                          (int)min(rbytes - rbase, (size_t)0x7fffffff), 0x00020000);
    bf16x8 iw[4];
    {
        const int p = mx_line_pos(l31), f0 = mx6_field_of_pos(p);
        f16x8 k0, k1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            k0[j] = (p == 8 * h + j) ? (_Float16)1.0f : (_Float16)0.0f;
            k1[j] = (p == 16 + 8 * h + j) ? (_Float16)1.0f : (_Float16)0.0f;
        }
        unsigned q[6];
#pragma unroll
        for (int d = 0; d < 6; ++d) {                           // code 8 (= 1.0) in field f0 of the h = 0 lanes: bit 6 f0 + 3
            const int bit = 6 * f0 + 3 - 32 * d;
            q[d] = (h == 0 && bit >= 0 && bit < 32) ? (1u << (bit & 31)) : 0u;
        }
        iw[0] = __builtin_bit_cast(bf16x8, k0);
        iw[1] = __builtin_bit_cast(bf16x8, k1);
        iw[2] = __builtin_bit_cast(bf16x8, u32x4{q[0], q[1], q[2], q[3]});
        iw[3] = __builtin_bit_cast(bf16x8, u32x4{q[4], q[5], 127u, 0u});
    }
    auto rdma = [&](int k) {
        const int nt = k / MT, mt = k % MT;
        int qr = qs[mt] - slab0;                                // (rows past the end repeat the tile's last pixel: a real one)
        asm volatile("" : "+v"(qr));                            // opaque: offsets are rebuilt per batch, not kept alive across the encodes
        if (r96) {                                              // the channel tile's own line plane (a plane can exceed the 2 GB a resource spans)
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const char*)a.resid + (size_t)(ntile0 + nt) * (size_t)a.plane96 + rbase), 0, (int)min(rbytes - rbase, (size_t)0x7fffffff), 0x00020000);
            resid_tile_dma_buf96(prs, qr, pixstride, 0, lane, scratch + (k % NBUF) * 4096);
        }
        else resid_tile_dma_buf(rrs, qr, pixstride, (ntile0 + nt) * 128, lane, scratch + (k % NBUF) * 4096);
    };
    const int xb = l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4);
#pragma unroll
    for (int k = 0; k < NBUF; ++k) rdma(k);
#pragma unroll
    for (int b0 = 0; b0 < T; b0 += NBUF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // batch b0 has landed (and the previous batch's stores)
#pragma unroll
        for (int k = b0; k < b0 + NBUF; ++k) {
            const char* t = scratch + (k % NBUF) * 4096;
            bf16x8 x[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) x[f] = *(const bf16x8*)(t + (xb ^ (f << 5)));
            if (r96 && h) {                                         // 96-byte lines carry no hi6 plane: the identity's lo6 half (zeros)
                x[2] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});   // multiplies it anyway, but a stale scale byte could be NaN
                x[3] = x[2];
            }
            acc[k / MT][k % MT] = mfma_mx6(acc[k / MT][k % MT], iw, x);
        }
        if (b0 + NBUF < T) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the batch is in registers: its buffers take the next one
#pragma unroll
            for (int k = b0 + NBUF; k < b0 + 2 * NBUF; ++k) rdma(k);
        }
        // encode + store this batch while the next one is in flight
        if constexpr (NBUF >= MT) {
#pragma unroll
            for (int nt = b0 / MT; nt < (b0 + NBUF) / MT; ++nt) conv_epilogue_mx<MT>(a, acc[nt], qs, valid, ntile0 + nt, lane);
        } else {
            conv_epilogue_mx<MT, NBUF>(a, acc[b0 / MT], qs, valid, ntile0 + b0 / MT, lane, b0 % MT);
        }
    }
}

// contiguous-position form: tile rows are PF positions q_base + mt*32 + (lane&31), pads filtered here
template <int MT, int PLANES>
static __device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT], int q_base, int ntile,
                                                     int lane) {
    int qs[MT];
    bool valid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        qs[mt] = q_base + mt * 32 + (lane & 31);
        valid[mt] = pf_is_pixel(a.go, qs[mt]);
    }
    conv_epilogue_q<MT, PLANES>(a, acc, qs, valid, ntile, lane);
}

template <int MT, int PLANES>
static __device__ __forceinline__ void conv_epilogue_any(const ConvArgs& a, f32x16 (&acc)[MT], int q_base, int ntile, int lane) {
    if constexpr (PLANES == 3) {
        int qs[MT];
        bool valid[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            qs[mt] = q_base + mt * 32 + (lane & 31);
            valid[mt] = pf_is_pixel(a.go, qs[mt]);
        }
        conv_epilogue_mx<MT>(a, acc, qs, valid, ntile, lane);                  // (acc started from the bias: acc_init_bias)
    } else {
        conv_epilogue<MT, PLANES>(a, acc, q_base, ntile, lane);
    }
}

// One 128-byte line of K for MT pixel tiles: 4 fragments per operand; the pixel fragments of tile
// mt+1 are requested before the MFMAs of tile mt so the LDS latency hides behind them.
template <int MT, int PLANES>
static __device__ __forceinline__ void mfma_line(f32x16 (&acc)[MT], const bf16x8 (&wf)[4], const char* smem,
                                                 const int (&xbase)[MT]) {
    bf16x8 xf[2][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) xf[0][f] = *(const bf16x8*)(smem + (xbase[0] ^ (f << 5)));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const bf16x8(&x)[4] = xf[mt & 1];
        if (mt + 1 < MT) {
#pragma unroll
            for (int f = 0; f < 4; ++f) xf[(mt + 1) & 1][f] = *(const bf16x8*)(smem + (xbase[mt + 1] ^ (f << 5)));
        }
        if constexpr (PLANES == 2) {
            acc[mt] = mfma16<2>(wf[2], x[0], acc[mt]);   // lo*hi
            acc[mt] = mfma16<2>(wf[3], x[1], acc[mt]);
            acc[mt] = mfma16<2>(wf[0], x[2], acc[mt]);   // hi*lo
            acc[mt] = mfma16<2>(wf[1], x[3], acc[mt]);
            acc[mt] = mfma16<2>(wf[0], x[0], acc[mt]);   // hi*hi
            acc[mt] = mfma16<2>(wf[1], x[1], acc[mt]);
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[mt] = mfma_bf16(wf[f], x[f], acc[mt]);
        }
    }
}

// Lane -> pixel of a 32-pixel MFMA tile for DENSE tiles.  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27} and
// {4-11, 16-19, 28-31} (+32 for the upper half) and is conflict-free when a group's 16 pixels are distinct modulo 16 in the
// slab.  With lane = pixel, the pad column a dense tile skips puts two of a group's pixels on one bank on maps narrower than
// 32 (r03 counters of the wide kernel: 39 % of its LDS cycles on 16x16 and 8x8 maps were conflict cycles, none on 32x32); giving
// each group 16 CONSECUTIVE pixels (one row of a 16-wide map, two of an 8-wide one) removes it at 16 and leaves one 2-way
// conflict per read at 8.  Any lane order is legal: a lane is just a column of the MFMA tile, and the epilogue stores by the
// same map.
static __device__ __forceinline__ int dense_lane_pixel(int l31) {
    return l31 < 4 ? l31 : l31 < 12 ? l31 + 12 : l31 < 16 ? l31 - 8 : l31 < 20 ? l31 + 8 : l31 < 28 ? l31 - 12 : l31;
}

// LDS byte offset of slot-pair base for slab-local pixel Pl and lane half h (swizzled):
// slot s = 2f + h is stored at slot s ^ ((Pl>>1)&7); fragment f is reached by XOR (f<<5).
static __device__ __forceinline__ int lds_xbase(int Pl, int h) { return Pl * 128 + ((h ^ ((Pl >> 1) & 7)) << 4); }
// one (pixel tile, channel tile, 32-channel line, tap) step of every precision mode
template <int PLANES>
static __device__ __forceinline__ void mfma_step(f32x16& d, const bf16x8 (&w)[4], const bf16x8 (&x)[4]) {
    if constexpr (PLANES == 3) {
        d = mfma_mx6(d, w, x);
    } else if constexpr (PLANES == 2) {
        d = mfma16<2>(w[2], x[0], d);
        d = mfma16<2>(w[3], x[1], d);
        d = mfma16<2>(w[0], x[2], d);
        d = mfma16<2>(w[1], x[3], d);
        d = mfma16<2>(w[0], x[0], d);
        d = mfma16<2>(w[1], x[1], d);
    } else {
#pragma unroll
        for (int f = 0; f < 4; ++f) d = mfma_bf16(w[f], x[f], d);
    }
}

