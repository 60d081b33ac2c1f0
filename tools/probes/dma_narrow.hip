// Probe: would a 256 x 128 tile (half the B rows per K-step, twice the column tiles sharing an A panel) make gemm_pp's operand stream
// cheap enough?  Same LDS-DMA issue pattern as dma_pattern.hip (pattern 2: the `share` workgroups of a panel on one XCD), with NB units
// of 16 KiB of B per K-step: NB = 2 is the 256 x 256 tile (64 KiB per K-step), NB = 1 the 256 x 128 tile (48 KiB per K-step, half the MFMA work).
//   build: hipcc --offload-arch=gfx950 -O3 dma_narrow.hip -o dma_narrow.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
using lptr_t = __attribute__((address_space(3))) void*;

template <int NB>
__global__ __launch_bounds__(512) void k(const char* A, const char* B, int K, int npanels, int share, int* sink, int pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, srow = lane >> 3, sslot = lane & 7;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), 0, 0x7fffffff, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B), 0, 0x7fffffff, 0x00020000);
    const int nkt = K / 64, ldb = K * 2 + pad;
    int slot = 0;
    const int vb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int col = vb % share;  // this workgroup's column tile: its own NB * 128 rows of B
    for (int panel = vb / share; panel < npanels; panel += gridDim.x / share) {
        for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
            for (int u = 0; u < 2 + NB; ++u) {  // units of 16 KiB: A rows 0-127, B ..., A rows 128-255
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int lr = (2 * w + q) * 8 + srow;
                    const bool isA = (u == 0 || u == 1 + NB);
                    int off;
                    if (isA) off = (panel * 256 + lr + (u ? 128 : 0)) * ldb + kt * 128 + sslot * 16;
                    else off = (col * NB * 128 + (u - 1) * 128 + lr) * ldb + kt * 128 + sslot * 16;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsA : rsB, (lptr_t)(smem + slot * 16384 + (2 * w + q) * 1024), 16, off, 0, 0, 0);
                }
                slot = slot == 9 ? 0 : slot + 1;
                asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (*(int*)(smem + threadIdx.x * 4) == 0x12345678) sink[0] = 1;
}

template <int NB>
void run(const char* A, const char* B, int* sink, int K, int share, int grid, int pad = 0) {
    const int M = 51456, npanels = M / 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NB>), dim3(grid), dim3(512), 163840, 0, A, B, K, npanels, share, sink, pad);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const int rounds = (npanels + grid / share - 1) / (grid / share);
    const double per = best * 1e3 / (rounds * (K / 64));
    printf("  pad %4d K=%4d tile 256x%d share %2d: %7.1f us = %.2f us per K-step per CU (%.2f us per 256x256x64 of MFMA work), %5.1f GB/s per CU\n", pad, K, NB * 128, share, best * 1e3,
           per, per * 2 / NB, (2 + NB) * 16384.0 / per * 1e-3);
}

int main() {
    char *A, *B; int* sink;
    (void)hipMalloc(&A, (size_t)51456 * (3072 * 2 + 1024) + (1 << 20)); (void)hipMemset(A, 1, (size_t)51456 * (3072 * 2 + 1024));
    (void)hipMalloc(&B, (size_t)3072 * (3072 * 2 + 1024) + (1 << 20)); (void)hipMemset(B, 1, (size_t)3072 * (3072 * 2 + 1024));
    (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int Ks[2] = {768, 3072};
    for (int ki = 0; ki < 2; ++ki) {
        run<2>(A, B, sink, Ks[ki], 3, 240);   // N = 768 today: 3 column tiles of 256
        run<1>(A, B, sink, Ks[ki], 6, 240);   // N = 768 as 6 column tiles of 128
        run<2>(A, B, sink, Ks[ki], 6, 240);   // N = 3072 today (GN = 6)
        run<1>(A, B, sink, Ks[ki], 12, 240);  // N = 3072 as groups of 12 narrow tiles
        run<2>(A, B, sink, Ks[ki], 12, 240);
        for (int pad : {64, 128, 256, 512, 1024}) run<2>(A, B, sink, Ks[ki], 3, 240, pad);
    }
    return 0;
}
