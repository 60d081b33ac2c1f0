// Probe: LDS-DMA throughput of gemm_pp_kernel's operand stream as a function of the A access pattern.
// Every workgroup (8 waves) walks row panels of an A[M, K] bf16 matrix: per K-step it brings the 256 x 64 slice of its
// panel (32 pieces of 8 rows x 128 bytes) plus a 256 x 64 slice of a small shared B into LDS, 10 units of 16 KiB in
// flight, exactly the issue pattern of the GEMM without MFMA / ds_read / barriers.
//   pattern 0: A row-major, row stride K * 2 bytes (what the GEMM does)
//   pattern 1: A packed per (panel, K-step): each 32 KiB slice contiguous
//   share: number of consecutive workgroups that read the same panel (column tiles of the GEMM)
//   build: hipcc --offload-arch=gfx950 -O3 dma_pattern.hip -o dma_pattern.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
using lptr_t = __attribute__((address_space(3))) void*;

template <int PATTERN>
__global__ __launch_bounds__(512) void k(const char* A, const char* B, int K, int npanels, int share, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, srow = lane >> 3, sslot = lane & 7;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), 0, 0x7fffffff, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B), 0, 0x7fffffff, 0x00020000);
    const int nkt = K / 64, ldb = K * 2;
    int slot = 0;
    // pattern 2: row-major A, but the `share` workgroups of a panel sit on ONE XCD (block b runs on XCD b % 8)
    const int vb = PATTERN == 2 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    for (int panel = vb / share; panel < npanels; panel += gridDim.x / share) {
        for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // 4 units of 16 KiB per K-step: A, B, B, A
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int lr = (2 * w + q) * 8 + srow;  // unit-local row 0..127
                    int off;
                    const bool isA = (u == 0 || u == 3);
                    const int row = lr + (u == 3 || u == 2 ? 128 : 0);
                    if (!isA) off = row * ldb + kt * 128 + sslot * 16;
                    else if (PATTERN != 1) off = (panel * 256 + row) * ldb + kt * 128 + sslot * 16;
                    else off = (panel * nkt + kt) * 32768 + row * 128 + sslot * 16;
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

template <int PATTERN>
void run(const char* A, const char* B, int* sink, int K, int share, int grid) {
    const int M = 51456, npanels = M / 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<PATTERN>), dim3(grid), dim3(512), 163840, 0, A, B, K, npanels, share, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const int rounds = (npanels + grid / share - 1) / (grid / share);  // panels per workgroup (rounded up)
    printf("  K=%4d share %2d grid %3d pattern %d: %7.1f us  = %.2f us per K-step per CU, %5.1f GB/s per CU, A from memory %.2f TB/s\n", K, share, grid, PATTERN, best * 1e3,
           best * 1e3 / (rounds * (K / 64)), 65536.0 * rounds * (K / 64) / best * 1e-6, (double)M * K * 2 / best * 1e-9);
}

int main() {
    char *A, *B; int* sink;
    (void)hipMalloc(&A, (size_t)51456 * 3072 * 2 + (1 << 20)); (void)hipMemset(A, 1, (size_t)51456 * 3072 * 2);
    (void)hipMalloc(&B, 256 * 3072 * 2 + (1 << 20)); (void)hipMemset(B, 1, 256 * 3072 * 2);
    (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int Ks[2] = {3072, 768};
    for (int ki = 0; ki < 2; ++ki)
        for (int share = 1; share <= 12; share *= (share == 1 ? 3 : 4)) {  // 1, 3, 12
            run<0>(A, B, sink, Ks[ki], share, 240);
            run<1>(A, B, sink, Ks[ki], share, 240);
            run<2>(A, B, sink, Ks[ki], share, 240);
        }
    return 0;
}
