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

template <int MT, int PLANES>
static __device__ __forceinline__ void conv_epilogue_q(const ConvArgs& a, f32x16 (&acc)[MT], const int (&qs)[MT],
                                                       const bool (&valid)[MT], int ntile, int lane, char* scratch = nullptr) {
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
    bf16x4 rh[MT][4], rl[MT][4];
    if (via_lds) {
        resid_tile_dma(a.resid, valid[0] ? qs[0] : a.go.G, pixstride, (size_t)ntile * 128, lane, scratch);
    } else if (a.resid) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const char* rp = (const char*)a.resid + poff[mt] + 16 * g;
                if (CONV_STUDY(a, CONV_NONTEMPORAL)) {
                    rh[mt][g] = __builtin_nontemporal_load((const bf16x4*)rp);
                    if constexpr (PLANES == 2) rl[mt][g] = __builtin_nontemporal_load((const bf16x4*)(rp + 64));
                } else {
                    rh[mt][g] = *(const bf16x4*)rp;
                    if constexpr (PLANES == 2) rl[mt][g] = *(const bf16x4*)(rp + 64);
                }
            }
    }
    float bias[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[g * 4 + i] = a.bias[ntile * 32 + 8 * g + 4 * h + i];
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
                rh[mt][g] = *(const bf16x4*)(t + ((g ^ sw) << 4));
                rl[mt][g] = *(const bf16x4*)(t + (((4 + g) ^ sw) << 4));
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[mt][4 * g + i] + bias[4 * g + i];
            if (a.resid) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += (float)rh[mt][g][i];
                if constexpr (PLANES == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)rl[mt][g][i];
                }
            }
            if (a.relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            bf16x4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hi[i] = (__bf16)v[i];
                lo[i] = (__bf16)(v[i] - (float)hi[i]);
            }
            if (valid[mt] && !CONV_STUDY(a, CONV_ABL_NO_STORE)) {
                char* op = (char*)a.out + ooff[mt] + 16 * g;
                if (CONV_STUDY(a, CONV_NONTEMPORAL)) {
                    __builtin_nontemporal_store(hi, (bf16x4*)op);
                    if constexpr (PLANES == 2) __builtin_nontemporal_store(lo, (bf16x4*)(op + 64));
                } else {
                    *(bf16x4*)op = hi;
                    if constexpr (PLANES == 2) *(bf16x4*)(op + 64) = lo;
                }
            }
        }
    }
}

// Mode-3 epilogue (fp16 hi + MX-fp4).  Lane (pixel, h) owns line positions 16h .. 16h+15 (common.h mx_line_pos):
// 32 contiguous bytes of fp16 and 8 bytes of each fp4 plane.  Block maxima need one exchange with lane^32; the
// fp4 planes are swapped between the two lanes so each writes one 16-byte piece (h=0: lo4 of all 32, h=1: hi4).
// Four store instructions per 32x32 tile (2 x 16 B fp16, 16 B fp4, 4 B scale) and four loads for a residual.
// Residual: `scratch` (8 KB of LDS private to the wave, or null) selects how the residual tile is read.  Read straight
// from memory, a load instruction touches 32 different 128-byte lines (one per pixel) and the four loads of a tile cost
// four TCP look-ups per line: measured 3.8 TB/s on the residual bytes and -18 % on a layer-1 launch when the same bytes
// are fetched line-contiguously (r01 study).  With scratch the tile (32 lines = 4 KB) is fetched by LDS-DMA, eight lanes
// per line (pixel indices come from the owning lanes by ds_bpermute; slot swizzle applied on the source side), the
// next tile's DMA in flight while this one is converted, and each lane then reads its share from LDS.
template <int MT>
static __device__ __forceinline__ void conv_epilogue_mx(const ConvArgs& a, f32x16 (&acc)[MT], const int (&qs)[MT],
                                                        const bool (&valid)[MT], int ntile, int lane, char* scratch = nullptr) {
    const int h = lane >> 5, l31 = lane & 31;
    const size_t pixstride = (size_t)a.go.C * 4;
    float bias[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[g * 4 + i] = a.bias[ntile * 32 + 8 * g + 4 * h + i];
    const float lo_clamp = a.relu ? 0.f : -65504.f;
    const bool via_lds = a.resid && scratch;
    auto rdma = [&](int mt) {                                 // residual tile mt -> scratch buffer mt & 1
        resid_tile_dma(a.resid, valid[mt] ? qs[mt] : a.go.G, pixstride, (size_t)ntile * 128, lane, scratch + (mt & 1) * 4096);
    };
    if (via_lds) rdma(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const size_t loff = (size_t)(valid[mt] ? qs[mt] : a.go.G) * pixstride + (size_t)ntile * 128;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[mt][r] + bias[r];
        if (a.resid) {
            f16x8 r0, r1;
            uint2 nib;
            unsigned rs;
            if (via_lds) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // tile mt has landed (and the previous tile's stores)
                if (mt + 1 < MT) rdma(mt + 1);
                const char* t = scratch + (mt & 1) * 4096 + l31 * 128;
                const int sw = (l31 >> 1) & 7;
                r0 = *(const f16x8*)(t + (((2 * h) ^ sw) << 4));
                r1 = *(const f16x8*)(t + (((2 * h + 1) ^ sw) << 4));
                nib = *(const uint2*)(t + ((4 ^ sw) << 4) + 8 * h);
                rs = *(const unsigned*)(t + ((6 ^ sw) << 4)) & 255u;
            } else {
                const char* rl = (const char*)a.resid + loff;
                r0 = *(const f16x8*)(rl + 32 * h);
                r1 = *(const f16x8*)(rl + 32 * h + 16);
                nib = *(const uint2*)(rl + 64 + 8 * h);                              // lo4 of this lane's 16 positions
                rs = *(const unsigned*)(rl + 96) & 255u;                             // residual's scale_lo
            }
            const float rscale = rs ? mx4_scale_value((int)rs) : 0.f;
            float d[16];
            mx4_unpack8(nib.x, rscale, d);
            mx4_unpack8(nib.y, rscale, d + 8);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                v[r] += (float)r0[r] + d[r];
                v[8 + r] += (float)r1[r] + d[8 + r];
            }
        }
        float lo[16], mh = 0.f, ml = 0.f;
        f16x8 hv[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = __builtin_amdgcn_fmed3f(v[r], lo_clamp, 65504.f);              // ReLU (if any) + fp16-range clamp in one op
            const _Float16 hh = (_Float16)v[r];
            hv[r >> 3][r & 7] = hh;
            lo[r] = v[r] - (float)hh;
            v[r] = (float)hh;                                                       // v now holds hi
            mh = fmaxf(mh, fabsf(v[r]));
            ml = fmaxf(ml, fabsf(lo[r]));
        }
        mh = fmaxf(mh, __shfl_xor(mh, 32));
        ml = fmaxf(ml, __shfl_xor(ml, 32));
        const int sh = mx4_scale_byte(mh), sl = mx4_scale_byte(ml);
        const float fh = sh ? mx4_scale_value(sh) : 1.f, fl = sl ? mx4_scale_value(sl) : 1.f;
        const unsigned ql[2] = {mx4_pack8(lo, fl), mx4_pack8(lo + 8, fl)}, qh[2] = {mx4_pack8(v, fh), mx4_pack8(v + 8, fh)};
        // lane h=0 keeps lo4 and receives the partner's lo4; lane h=1 keeps hi4 and receives the partner's hi4
        const unsigned s0 = __shfl_xor(h ? ql[0] : qh[0], 32), s1 = __shfl_xor(h ? ql[1] : qh[1], 32);
        const u32x4 q4 = h ? u32x4{s0, s1, qh[0], qh[1]} : u32x4{ql[0], ql[1], s0, s1};
        if (valid[mt] && !CONV_STUDY(a, CONV_ABL_NO_STORE)) {
            char* ol = (char*)a.out + (a.out_split_pixels ? pf_out_offset(a.go, a.out_split_pixels, qs[mt], pixstride) + (size_t)ntile * 128 : loff);
            *(f16x8*)(ol + 32 * h) = hv[0];
            *(f16x8*)(ol + 32 * h + 16) = hv[1];
            *(u32x4*)(ol + 64 + 16 * h) = q4;
            const unsigned sc = (unsigned)(h ? sh : sl);                          // replicated: the whole 128-byte line is written
            *(u32x4*)(ol + 96 + 16 * h) = u32x4{sc, sc, sc, sc};              // (no partial-line writes), readers pick any dword
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
        conv_epilogue_mx<MT>(a, acc, qs, valid, ntile, lane);
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
            acc[mt] = mfma_bf16(wf[2], x[0], acc[mt]);   // lo*hi
            acc[mt] = mfma_bf16(wf[3], x[1], acc[mt]);
            acc[mt] = mfma_bf16(wf[0], x[2], acc[mt]);   // hi*lo
            acc[mt] = mfma_bf16(wf[1], x[3], acc[mt]);
            acc[mt] = mfma_bf16(wf[0], x[0], acc[mt]);   // hi*hi
            acc[mt] = mfma_bf16(wf[1], x[1], acc[mt]);
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[mt] = mfma_bf16(wf[f], x[f], acc[mt]);
        }
    }
}

// LDS byte offset of slot-pair base for slab-local pixel Pl and lane half h (swizzled):
// slot s = 2f + h is stored at slot s ^ ((Pl>>1)&7); fragment f is reached by XOR (f<<5).
static __device__ __forceinline__ int lds_xbase(int Pl, int h) { return Pl * 128 + ((h ^ ((Pl >> 1) & 7)) << 4); }
// Mode 3: the block-scale dword of pixel Pl (slot 6 + h, replicated in all four dwords of the slot).  Reading dword
// (Pl & 1) + 2 * ((Pl >> 4) & 1) spreads 32 consecutive pixels over all 32 banks of a ds_read_b32.
static __device__ __forceinline__ bf16x8 lds_xscale(const char* smem, int base, int Pl) {
    const unsigned sc = *(const unsigned*)(smem + (base ^ (3 << 5)) + 4 * ((Pl & 1) + 2 * ((Pl >> 4) & 1)));
    return __builtin_bit_cast(bf16x8, u32x4{sc, 0u, 0u, 0u});
}

// one (pixel tile, channel tile, 32-channel line, tap) step of every precision mode
template <int PLANES>
static __device__ __forceinline__ void mfma_step(f32x16& d, const bf16x8 (&w)[4], const bf16x8 (&x)[4]) {
    if constexpr (PLANES == 3) {
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, x[0]), d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w[1]), __builtin_bit_cast(f16x8, x[1]), d, 0, 0, 0);
        const i32x4 wq = __builtin_bit_cast(i32x4, w[2]), xq = __builtin_bit_cast(i32x4, x[2]);
        const i32x8 wa = {wq[0], wq[1], wq[2], wq[3], 0, 0, 0, 0}, xa = {xq[0], xq[1], xq[2], xq[3], 0, 0, 0, 0};
        d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, xa, d, 4, 4, 0, __builtin_bit_cast(i32x4, w[3])[0], 0,
                                                            __builtin_bit_cast(i32x4, x[3])[0]);
    } else if constexpr (PLANES == 2) {
        d = mfma_bf16(w[2], x[0], d);
        d = mfma_bf16(w[3], x[1], d);
        d = mfma_bf16(w[0], x[2], d);
        d = mfma_bf16(w[1], x[3], d);
        d = mfma_bf16(w[0], x[0], d);
        d = mfma_bf16(w[1], x[1], d);
    } else {
#pragma unroll
        for (int f = 0; f < 4; ++f) d = mfma_bf16(w[f], x[f], d);
    }
}

