// Store-pattern probe for the conv epilogues (r03): does it matter to the memory path whether a wave's 16-byte stores complete
// whole 128-byte lines per instruction?
//   A  "epilogue": lane (p = l & 31, h = l >> 5) owns pixel line p of a 32-line tile and stores its four 16-byte pieces at
//      {32h, 32h+16, 64+16h, 96+16h}: every instruction touches 32 lines with 32 bytes each (what conv_epilogue_mx does).
//   B  "lines"   : instruction j stores piece (l & 7) of line 8j + (l >> 3): 8 whole lines per instruction.
//   A_nt / B_nt  : the same with nontemporal stores.
//   RW variants  : each tile also reads one 4 KB tile (coalesced, as the slab DMA does) - the layer-1 mix of 1.5 : 1.
// Prints GB/s per pattern for `bytes` written by 256 CUs x 8 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT, bool NT, bool RD>
__global__ __launch_bounds__(256) void k_store(char* out, const char* in, long long tiles) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    for (long long t = wave0; t < tiles; t += nw) {
        char* o = out + t * 4096;
        if (RD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 r = *(const u32x4*)(in + t * 4096 + j * 1024 + lane * 16);
                v[j] ^= r[0] + r[1] + r[2] + r[3];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4* dst;
            if (PAT == 0) {
                const int p = lane & 31, h = lane >> 5;
                const int off = j == 0 ? 32 * h : j == 1 ? 32 * h + 16 : j == 2 ? 64 + 16 * h : 96 + 16 * h;
                dst = (u32x4*)(o + p * 128 + off);
            } else {
                dst = (u32x4*)(o + (8 * j + (lane >> 3)) * 128 + (lane & 7) * 16);
            }
            if (NT) __builtin_nontemporal_store(v, dst);
            else *dst = v;
        }
    }
}

int main(int argc, char** argv) {
    const long long bytes = (argc > 1 ? atoll(argv[1]) : 4096) * (1ll << 20);
    const long long tiles = bytes / 4096;
    char *out, *in;
    if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&in, bytes) != hipSuccess) return 1;
    hipMemset(in, 1, bytes);
    hipMemset(out, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, void (*k)(char*, const char*, long long), double moved) {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, out, in, tiles);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it && ms < best) best = ms;
        }
        printf("%-28s %8.3f ms  %7.1f GB/s moved\n", name, best, moved / best * 1e-6);
    };
    run("A epilogue pattern", k_store<0, false, false>, (double)bytes);
    run("B whole lines", k_store<1, false, false>, (double)bytes);
    run("A epilogue pattern, nt", k_store<0, true, false>, (double)bytes);
    run("B whole lines, nt", k_store<1, true, false>, (double)bytes);
    run("A + read", k_store<0, false, true>, 2.0 * bytes);
    run("B + read", k_store<1, false, true>, 2.0 * bytes);
    run("A nt + read", k_store<0, true, true>, 2.0 * bytes);
    run("B nt + read", k_store<1, true, true>, 2.0 * bytes);
    return 0;
}
