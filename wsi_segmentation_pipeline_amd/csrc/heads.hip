// Heads and layout converters.
//   avgpool_fc : AdaptiveAvgPool2d((1,1)) + flatten + Linear(F -> K)
//                (/root/reference/resnets_shift.py:206-208 `fc0`; /root/reference/models/models.py:32-38 Classifier)
//   linear     : y = act(x W^T + b), fp32 (/root/reference/resnets_shift.py:135-139,215 `fc`;
//                /root/reference/models/models.py:46-50 Regressor)
//   pf_pack / pf_unpack : f32 NCHW <-> padded-flat bf16 planes (API boundary + tests)
#include "pf_lines.h"

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup per image. feat[n][c] = mean over H*W of (hi+lo); logits[n][k] = feat . w[k] + b[k].
template <int PLANES>
__global__ __launch_bounds__(256) void avgpool_fc_kernel(const void* in, PFGeom g, const float* w, const float* b, int K,
                                                         float* feat, float* logits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* f = (float*)smem;                            // C floats
    const int n = blockIdx.x, tid = threadIdx.x;
    const size_t pixstride = (size_t)g.C * PFmt<PLANES>::BPC;
    const float inv = 1.0f / (float)(g.H * g.W);
    if constexpr (PLANES == 3) {
        // Thread t owns one 16-byte slice (8 line positions) of 32-channel line (t >> 2) mod NL, for pixels t / (4 NL),
        // + 256 / (4 NL), ...: whole lines are read by 4 neighbouring lanes (coalesced 64 B of fp16 + the lo6 plane / scale),
        // partial sums meet in LDS.  (The per-channel form read 2 bytes per access: 0.125 ms per 1000 patches.)
        const int NL = g.C / 32, HW = g.H * g.W;
        float* part = f + g.C;                                  // [256 / (4 NL) pixel groups][C] partial sums (launcher sizes it)
        const int slice = tid & 3, line = (tid >> 2) % NL, pg = tid / (4 * NL), npg = 256 / (4 * NL);
        float sm[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (pg < npg) {
            for (int p = pg; p < HW; p += npg) {
                const int y = p / g.W, x = p - y * g.W;
                const char* L = (const char*)in + (size_t)(g.G + n * g.S + y * g.P + x) * pixstride + (size_t)line * 128;
                const f16x8 hi = *(const f16x8*)(L + 16 * slice);
                const u32x4 p0 = *(const u32x4*)(L + MX6_PLANE_LO(0)), p1 = *(const u32x4*)(L + MX6_PLANE_HI(0));   // lo6 plane + {rest, scale_lo}
                const unsigned sl = p1[2] & 255u;
                const f32x32 dd = mx6_unpack32(u32x6{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1]}, sl ? mx_scale_value((int)sl) : 0.f);
                float d[8];                                                           // position 8 slice + i = field 16 (slice & 1) + 2 i + (slice >> 1)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float e0 = (slice >> 1) ? dd[2 * i + 1] : dd[2 * i], e1 = (slice >> 1) ? dd[16 + 2 * i + 1] : dd[16 + 2 * i];
                    d[i] = (slice & 1) ? e1 : e0;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) sm[i] += (float)hi[i] + d[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) part[pg * g.C + line * 32 + mx_line_chan(8 * slice + i)] = sm[i];
        }
        __syncthreads();
        for (int c = tid; c < g.C; c += 256) {
            float t = 0.f;
            for (int k = 0; k < npg; ++k) t += part[k * g.C + c];
            f[c] = t * inv;
            if (feat) feat[(size_t)n * g.C + c] = t * inv;
        }
    } else
    for (int c4 = tid; c4 < g.C / 4; c4 += 256) {
        const int c = c4 * 4;
        const size_t coff = PLANES == 2 ? (size_t)(c >> 5) * 128 + (c & 31) * 2 : (size_t)c * 2;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int y = 0; y < g.H; ++y)
            for (int x = 0; x < g.W; ++x) {
                const char* p = (const char*)in + (size_t)(g.G + n * g.S + y * g.P + x) * pixstride + coff;
                typedef __attribute__((ext_vector_type(4))) typename PairElem<PLANES>::T Ex4;   // mode 2: fp16 pair, 1: bf16
                const Ex4 hi = *(const Ex4*)p;
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += (float)hi[k];
                if constexpr (PLANES == 2) {
                    const Ex4 lo = *(const Ex4*)(p + 64);
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] += (float)lo[k];
                }
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[c + k] = s[k] * inv;
            if (feat) feat[(size_t)n * g.C + c + k] = s[k] * inv;
        }
    }
    __syncthreads();
    if (!logits) return;
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = wave; k < K; k += 4) {
        float acc = 0.f;
        for (int c = lane; c < g.C; c += 64) acc += f[c] * w[(size_t)k * g.C + c];
        acc = wave_sum(acc);
        if (lane == 0) logits[(size_t)n * K + k] = acc + b[k];
    }
}

// y[b][j] = act(sum_k x[b][k] * w[j][k] + bias[j]).  One wave per output j, up to BT rows of x per
// pass (weights are streamed once per pass, coalesced 16 B per lane).
template <int BT>
__global__ __launch_bounds__(256) void linear_kernel(const float* x, const float* w, const float* bias, float* y, int B,
                                                     int K, int J, int relu) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= J) return;
    const f32x4* wr = (const f32x4*)(w + (size_t)j * K);
    for (int b0 = 0; b0 < B; b0 += BT) {
        float acc[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t) acc[t] = 0.f;
        for (int k4 = lane; k4 < K / 4; k4 += 64) {
            const f32x4 wv = wr[k4];
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                if (b0 + t < B) {
                    const f32x4 xv = ((const f32x4*)(x + (size_t)(b0 + t) * K))[k4];
                    acc[t] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const float s = wave_sum(acc[t]);
            if (lane == 0 && b0 + t < B) {
                float v = s + (bias ? bias[j] : 0.f);
                if (relu) v = fmaxf(v, 0.f);
                y[(size_t)(b0 + t) * J + j] = v;
            }
        }
    }
}

// Few outputs, many rows (the bag head's second layer Linear(4096 -> 4) over thousands of bags, models.py Classifier /
// Regressor tails): one wave per ROW, the J <= 16 weight rows (J x K floats, L2-resident) streamed by all waves.
// linear_kernel above puts one wave on each OUTPUT and walks the rows serially: 22 ms for 2000 bags (r02 rocprof).
template <int JMAX>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* x, const float* w, const float* bias, float* y, int B, int K,
                                                          int J, int relu) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const f32x4* xr = (const f32x4*)(x + (size_t)b * K);
    float acc[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) acc[j] = 0.f;
    for (int k4 = lane; k4 < K / 4; k4 += 64) {
        const f32x4 xv = xr[k4];
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            if (j < J) {
                const f32x4 wv = ((const f32x4*)(w + (size_t)j * K))[k4];
                acc[j] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        if (j < J) {
            const float s = wave_sum(acc[j]);
            if (lane == 0) {
                float v = s + (bias ? bias[j] : 0.f);
                if (relu) v = fmaxf(v, 0.f);
                y[(size_t)b * J + j] = v;
            }
        }
    }
}

// y = act(x W^T + b) as an fp32-input MFMA GEMM (v_mfma_f32_32x32x2_f32: exact fp32 multiply-add chains, no operand
// split needed) for the bag head `fc.0` = Linear(8192 -> 4096) + ReLU (/root/reference/resnets_shift.py:133-139,214-215).
// r01 ran it on the wave-per-output VALU kernel above, which re-streams the 134 MB weight matrix once per 8 bags; here a
// workgroup owns 64 bags x 128 outputs, every weight element is fetched once per 64 bags and meets them in the matrix
// pipe.  Wave w: outputs [32w, 32w+32) x 64 bags = two 32x32 accumulators; K in steps of 32 through LDS (k-major tiles:
// a lane's operand is one conflict-free ds_read_b32), next step's global loads in flight during the current step's MFMAs.
__global__ __launch_bounds__(256) void linear_mfma_f32_kernel(const float* x, const float* w, const float* bias, float* y, int B,
                                                              int K, int J, int relu) {
    constexpr int BM = 64, BN = 128, KB = 32;
    __shared__ float Xs[KB][BM];
    __shared__ float Ws[KB][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    // staging roles: X tile 64 rows x 32 k (8 k per thread), W tile 128 rows x 32 k (16 k per thread)
    const int xr = tid >> 2, xk = (tid & 3) * 8;
    const int wr = tid >> 1, wk = (tid & 1) * 16;
    const float* xp = x + (size_t)min(m0 + xr, B - 1) * K + xk;
    const float* wp = w + (size_t)(n0 + wr) * K + wk;
    f32x4 xv[2], wv[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) xv[q] = *(const f32x4*)(xp + k0 + 4 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[q] = *(const f32x4*)(wp + k0 + 4 * q);
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) Xs[xk + 4 * q + e][xr] = xv[q][e];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) Ws[wk + 4 * q + e][wr] = wv[q][e];
    };
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KB) {
        __syncthreads();                                      // the previous step's reads are done
        commit();
        __syncthreads();
        if (k0 + KB < K) fetch(k0 + KB);                      // in flight behind this step's 32 MFMAs
#pragma unroll
        for (int kk = 0; kk < KB / 2; ++kk) {
            const float b = Ws[2 * kk + h][wave * 32 + l31];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[2 * kk + h][mt * 32 + l31], b, acc[mt], 0, 0, 0);
        }
    }
    const int j = n0 + wave * 32 + l31;
    const float bj = bias ? bias[j] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < B) {
                float v = acc[mt][r] + bj;
                if (relu) v = fmaxf(v, 0.f);
                y[(size_t)row * J + j] = v;
            }
        }
}

template <int PLANES>
__global__ __launch_bounds__(256) void pf_pack_kernel(const float* in, void* out, PFGeom g) {
    const long long total = (long long)g.N * g.H * g.W * g.C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        long long p = i / g.C;
        const int x = (int)(p % g.W); p /= g.W;
        const int y = (int)(p % g.H);
        const int n = (int)(p / g.H);
        const float v = in[(((size_t)n * g.C + c) * g.H + y) * g.W + x];
        char* o = (char*)out + (size_t)(g.G + n * g.S + y * g.P + x) * ((size_t)g.C * PLANES * 2);
        if constexpr (PLANES == 2) {                                          // fp16 pair (common.h split_f16)
            _Float16 hi, lo;
            split_f16(v, hi, lo);
            *(_Float16*)(o + (c >> 5) * 128 + (c & 31) * 2) = hi;
            *(_Float16*)(o + (c >> 5) * 128 + 64 + (c & 31) * 2) = lo;
        } else {
            *(__bf16*)(o + c * 2) = (__bf16)v;
        }
    }
}

template <int PLANES>
__global__ __launch_bounds__(256) void pf_unpack_kernel(const void* in, float* out, PFGeom g) {
    const long long total = (long long)g.N * g.H * g.W * g.C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        long long p = i / g.C;
        const int x = (int)(p % g.W); p /= g.W;
        const int y = (int)(p % g.H);
        const int n = (int)(p / g.H);
        const char* o = (const char*)in + (size_t)(g.G + n * g.S + y * g.P + x) * ((size_t)g.C * PLANES * 2);
        float v;
        if constexpr (PLANES == 2)
            v = (float)*(const _Float16*)(o + (c >> 5) * 128 + (c & 31) * 2) +
                (float)*(const _Float16*)(o + (c >> 5) * 128 + 64 + (c & 31) * 2);
        else
            v = (float)*(const __bf16*)(o + c * 2);
        out[(((size_t)n * g.C + c) * g.H + y) * g.W + x] = v;
    }
}

__global__ __launch_bounds__(256) void pf_pack_mx_kernel(const float* in, void* out, PFGeom g) {
    const long long total = (long long)g.N * g.H * g.W * (g.C / 32);
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int l = (int)(i % (g.C / 32));
        long long p = i / (g.C / 32);
        const int x = (int)(p % g.W); p /= g.W;
        const int y = (int)(p % g.H);
        const int n = (int)(p / g.H);
        float v[32];
        for (int c = 0; c < 32; ++c) v[c] = in[(((size_t)n * g.C + 32 * l + c) * g.H + y) * g.W + x];
        mx_line_encode(v, (char*)out + (size_t)(g.G + n * g.S + y * g.P + x) * ((size_t)g.C * 4) + (size_t)l * 128);
    }
}

__global__ __launch_bounds__(256) void pf_unpack_mx_kernel(const void* in, float* out, PFGeom g) {
    const long long total = (long long)g.N * g.H * g.W * g.C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % g.C);
        long long p = i / g.C;
        const int x = (int)(p % g.W); p /= g.W;
        const int y = (int)(p % g.H);
        const int n = (int)(p / g.H);
        const char* line = (const char*)in + (size_t)(g.G + n * g.S + y * g.P + x) * ((size_t)g.C * 4) + (size_t)(c >> 5) * 128;
        out[(((size_t)n * g.C + c) * g.H + y) * g.W + x] = mx_line_decode(line, c & 31);
    }
}

static int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

int wsi_avgpool_fc_dispatch(const void* in, const PFGeom& g, const float* w, const float* b, int K, float* feat,
                            float* logits, int planes, hipStream_t st) {
    if (g.C % 4 || g.N <= 0 || planes < 1 || planes > 3 || (planes == 3 && g.C % 32)) return WSI_EINVAL;
    size_t lds = (size_t)g.C * 4;
    if (planes == 3) {                                      // + partial sums of the 256 / (4 C/32) pixel groups
        if (g.C > 2048) return WSI_EINVAL;
        lds += (size_t)(256 / (4 * (g.C / 32))) * g.C * 4;
    }
    if (planes == 3)
        hipLaunchKernelGGL(avgpool_fc_kernel<3>, dim3(g.N), dim3(256), lds, st, in, g, w, b, K, feat, logits);
    else if (planes == 2)
        hipLaunchKernelGGL(avgpool_fc_kernel<2>, dim3(g.N), dim3(256), lds, st, in, g, w, b, K, feat, logits);
    else
        hipLaunchKernelGGL(avgpool_fc_kernel<1>, dim3(g.N), dim3(256), lds, st, in, g, w, b, K, feat, logits);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_linear_dispatch(const float* x, const float* w, const float* bias, float* y, int B, int K, int J, int relu,
                        hipStream_t st) {
    if (K % 4 || B <= 0 || J <= 0) return WSI_EINVAL;
    if (J % 128 == 0 && K % 32 == 0 && B >= 8 && (long long)J * K >= (1 << 20)) {     // the bag head fc.0 (4096 x 8192): matrix pipe
        hipLaunchKernelGGL(linear_mfma_f32_kernel, dim3(J / 128, (B + 63) / 64), dim3(256), 0, st, x, w, bias, y, B, K, J, relu);
        return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
    }
    if (J <= 16 && B >= 32) {                                                          // few outputs, many rows: a wave per row
        if (J <= 4) hipLaunchKernelGGL(linear_rows_kernel<4>, dim3((B + 3) / 4), dim3(256), 0, st, x, w, bias, y, B, K, J, relu);
        else hipLaunchKernelGGL(linear_rows_kernel<16>, dim3((B + 3) / 4), dim3(256), 0, st, x, w, bias, y, B, K, J, relu);
        return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
    }
    hipLaunchKernelGGL(linear_kernel<8>, dim3((J + 3) / 4), dim3(256), 0, st, x, w, bias, y, B, K, J, relu);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_pf_pack_dispatch(const float* in, void* out, const PFGeom& g, int planes, hipStream_t st) {
    const long long total = (long long)g.N * g.H * g.W * g.C;
    if (planes == 3)
        hipLaunchKernelGGL(pf_pack_mx_kernel, dim3(grid_for(total / 32)), dim3(256), 0, st, in, out, g);
    else if (planes == 2)
        hipLaunchKernelGGL(pf_pack_kernel<2>, dim3(grid_for(total)), dim3(256), 0, st, in, out, g);
    else
        hipLaunchKernelGGL(pf_pack_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, in, out, g);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}

int wsi_pf_unpack_dispatch(const void* in, float* out, const PFGeom& g, int planes, hipStream_t st) {
    const long long total = (long long)g.N * g.H * g.W * g.C;
    if (planes == 3)
        hipLaunchKernelGGL(pf_unpack_mx_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, out, g);
    else if (planes == 2)
        hipLaunchKernelGGL(pf_unpack_kernel<2>, dim3(grid_for(total)), dim3(256), 0, st, in, out, g);
    else
        hipLaunchKernelGGL(pf_unpack_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, in, out, g);
    return hipGetLastError() == hipSuccess ? WSI_OK : WSI_EFAULT;
}
