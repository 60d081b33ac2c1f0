// Probe for the fp8 "lo" pass of the split-operand GEMMs: v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands.
//   probe_cvt   : v_cvt_pk_fp8_f32 of a float array (is it OCP e4m3fn with round-to-nearest-even and saturation?)
//   probe_mfma  : one MFMA on A[16][128], B[16][128] bytes, lane l holding the 32 bytes k = 32 (l >> 4) .. + 31 of row l & 15 of both
//                 operands, E8M0 scales sa / sb in byte 0 of the scale registers -> D[16][16]
//   probe_rate  : independent-accumulator loops of the scaled fp8 MFMA and of v_mfma_f32_16x16x32_f16 on every CU (relative rate)
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/mfma_f8.hip -o tools/probes/mfma_f8.so   (driver: tools/probes/mfma_f8.py)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__global__ void cvt_kernel(const float* x, uint8_t* y, int n) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n + 1) return;
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], i + 1 < n ? x[i + 1] : 0.f, 0, false);
    y[i] = (uint8_t)(w & 0xff);
    if (i + 1 < n) y[i + 1] = (uint8_t)((w >> 8) & 0xff);
}
__global__ void mfma_kernel(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    const i32x8 a = *(const i32x8*)(A + r * 128 + 32 * g);
    const i32x8 b = *(const i32x8*)(B + r * 128 + 32 * g);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    // C/D map of the 16x16 shapes: column l & 15 (the B operand's row index), rows 4 (l >> 4) + i (the A operand's)
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); b[i] = 0x3c343c34 ^ (threadIdx.x << (i & 7)); }
    f16x8 ha = __builtin_bit_cast(f16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)), hb = __builtin_bit_cast(f16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 127, 0, 127);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}
extern "C" int probe_cvt(const float* x, uint8_t* y, int n) {
    hipLaunchKernelGGL(cvt_kernel, dim3((n / 2 + 256) / 256), dim3(256), 0, 0, x, y, n);
    return (int)hipDeviceSynchronize();
}
extern "C" int probe_mfma(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, A, B, D, sa, sb);
    return (int)hipDeviceSynchronize();
}
// returns milliseconds of `grid` workgroups x 4 waves x iters x 8 MFMAs
extern "C" float probe_rate(int mode, int grid, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
