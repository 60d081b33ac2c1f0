// Probe: how fast does one CU drain a 256 x 256 output tile for different lane -> address mappings of the epilogue
// stores?  Persistent grid of 256 workgroups x 8 waves (2 x 4, 128 x 64 per wave) walking the tiles of an [M, N] output,
// exactly the store stream of gemm_pp_kernel's epilogue without the main loop (optionally with an idle gap per tile).
//   build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern.bin ; run: ./store_pattern.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int PAT, int ESZ>
__global__ __launch_bounds__(512) void k(char* out, int M, int N, int ntn, int ntiles, int gap) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 2, wc = w & 3;
    const int frow = lane & 15, fq = lane >> 4, hi = (frow >> 3) & 1;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, M * N * ESZ, 0x00020000);
    const int ld = N * ESZ;
    u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / ntn, tn = t - tm * ntn;
        const int base = (tm * 256 + wr * 128) * ld + (tn * 256 + wc * 64) * ESZ;  // wave's 128 rows x (64 * ESZ) bytes
        if constexpr (ESZ == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (PAT == 0) {  // permuted layout, direct: row frow, 16-byte pieces at 32-byte stride
                    const int off = base + (16 * i + frow) * ld + fq * 32;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 16, 0, 0);
                } else if constexpr (PAT == 1) {  // exchange with lane ^ 8: 8 rows x 128 bytes per instruction, lanes row-scattered
                    const int off = base + (16 * i + (frow & 7)) * ld + fq * 32 + hi * 16;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 8 * ld, 0, 0);
                } else {  // line-major: consecutive lanes = consecutive 16-byte pieces
                    const int off = base + (16 * i + (lane >> 3)) * ld + (lane & 7) * 16;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 8 * ld, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (PAT == 0) {  // permuted: lane owns 64 contiguous bytes
                    const int off = base + (16 * i + frow) * ld + fq * 64;
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 16 * j, 0, 0);
                } else if constexpr (PAT == 1) {  // natural: 4 lanes x 16 bytes contiguous per row and instruction
                    const int off = base + (16 * i + frow) * ld + fq * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 64 * j, 0, 0);
                } else if constexpr (PAT == 2) {  // natural + exchange: 8 rows x 128 bytes
                    const int off = base + (16 * i + (frow & 7)) * ld + fq * 16 + hi * 64;
#pragma unroll
                    for (int J = 0; J < 2; ++J) {
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 128 * J, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 128 * J + 8 * ld, 0, 0);
                    }
                } else {  // line-major: 4 rows x 256 bytes per instruction
                    const int off = base + (16 * i + (lane >> 4)) * ld + (lane & 15) * 16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 4 * j * ld, 0, 0);
                }
            }
        }
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(32);  // ~1 us each
        if (gap) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
}

template <int PAT, int ESZ>
float run(char* buf, int M, int N, int gap, int grid = 256) {
    const int ntn = N / 256, ntiles = ((M + 255) / 256) * ntn;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<PAT, ESZ>), dim3(grid), dim3(512), 0, 0, buf, M, N, ntn, ntiles, gap);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    const int M = 51456;
    char* buf;
    hipMalloc(&buf, (size_t)65536 * 3072 * 4);  // covers every (M, N, element size) used below
    const int Ns[3] = {768, 2304, 3072};
    for (int gap = 0; gap <= 12; gap += 12)
        for (int ni = 0; ni < 3; ++ni) {
            const int N = Ns[ni];
            const int waves = (((M + 255) / 256) * (N / 256) + 255) / 256;
            printf("N=%4d gap=%2d us (%2d tile waves)  T(2B): direct %7.1f  xchg %7.1f  line %7.1f us | f32: perm %7.1f  nat %7.1f  nat+xchg %7.1f  line %7.1f us\n", N, gap, waves,
                   run<0, 2>(buf, M, N, gap), run<1, 2>(buf, M, N, gap), run<2, 2>(buf, M, N, gap), run<0, 4>(buf, M, N, gap), run<1, 4>(buf, M, N, gap),
                   run<2, 4>(buf, M, N, gap), run<3, 4>(buf, M, N, gap));
            fflush(stdout);
        }
    // per-CU or aggregate limit?  the same 10 tiles per workgroup on fewer CUs
    for (int grid = 256; grid >= 8; grid /= 2) {
        const int Mg = grid * 256;  // N = 2560: 10 column tiles, so ntiles = 10 * grid (Mg <= 65536 rows: inside the allocation)
        const float t2 = run<0, 2>(buf, Mg, 2560, 0, grid), t4 = run<1, 4>(buf, Mg, 2560, 0, grid);
        const double tiles = (double)grid * 10;
        printf("grid %3d: T direct %7.1f us = %6.1f GB/s/CU | f32 nat %7.1f us = %6.1f GB/s/CU\n", grid, t2, tiles * 131072 / grid / t2 * 1e-3, t4, tiles * 262144 / grid / t4 * 1e-3);
    }
    return 0;
}
