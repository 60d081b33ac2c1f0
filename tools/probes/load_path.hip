// Probe: per-CU bandwidth of the ways to bring an L2-resident operand tile into a CU (gfx950):
//   0  buffer_load_dwordx4 -> VGPR (consumed by a dummy xor)
//   1  buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB per wave instruction)
//   2  global_load_dwordx4 -> VGPR -> ds_write_b128
// Every workgroup (8 waves) streams the same `span` bytes again and again (span << L2), 8 full 128-byte lines per wave
// instruction like gemm_pp_kernel's operand units.  Reports GB/s per CU for 256 / 32 workgroups.
//   build: hipcc --offload-arch=gfx950 -O3 load_path.hip -o load_path.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
using lptr_t = __attribute__((address_space(3))) void*;

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, unsigned* sink, int span, int iters, int wg_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
    const int base = blockIdx.x * wg_stride;  // workgroups of one XCD overlap (wg_stride small) or not
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // one "K-step": 64 KiB per workgroup = 8 pieces of 1 KiB per wave
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int off = base + ((it * 65536 + (w * 8 + q) * 1024) % span) + lane * 16;
            if constexpr (MODE == 0) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                acc ^= v;
            } else if constexpr (MODE == 1) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (w * 8 + q) * 1024), 16, off, 0, 0, 0);
            } else {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                *(u32x4*)(smem + (w * 8 + q) * 1024 + lane * 16) = v;
            }
        }
        if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 0) acc[0] ^= *(unsigned*)(smem + threadIdx.x * 4);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
void run(const char* src, unsigned* sink, int grid, int span, int wg_stride, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 65536, 0, src, sink, span, iters, wg_stride);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    printf("  %-28s grid %3d span %4d KiB stride %6d: %7.1f GB/s per CU  (%.2f us per 64 KiB)\n", name, grid, span >> 10, wg_stride, 65536.0 * iters / best * 1e-6, best * 1e3 / iters);
}

int main() {
    char* src; unsigned* sink;
    (void)hipMalloc(&src, 512u << 20); (void)hipMemset(src, 1, 512u << 20); (void)hipMalloc(&sink, 4);
    for (int grid = 256; grid >= 32; grid /= 8)
        for (int stride = 0; stride <= 1 << 20; stride += 1 << 20) {  // 0: all workgroups read the same 256 KiB; 1 MiB: disjoint
            const int span = 256 << 10;
            run<0>(src, sink, grid, span, stride, "load dwordx4 -> VGPR");
            run<1>(src, sink, grid, span, stride, "load dwordx4 -> LDS (DMA)");
            run<2>(src, sink, grid, span, stride, "load -> VGPR -> ds_write");
        }
    return 0;
}
