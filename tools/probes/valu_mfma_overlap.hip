// Probe (round 4): do VALU instructions and MFMAs of the waves of one SIMD overlap on gfx950?  The attention kernels' time equals the SUM of
// their VALU and matrix-pipe cycles; this measures the two instruction classes alone, mixed in one wave, and split over the
// waves of a SIMD.  256 workgroups x 16 waves (4 per SIMD), registers only, no memory traffic in the loop.
//   build: hipcc --offload-arch=gfx950 -O3 valu_mfma_overlap.hip -o valu_mfma_overlap.bin ; run: ./valu_mfma_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;

// per loop iteration and wave: NM MFMAs (16x16x32 bf16, 8 independent accumulators), NV fp32 FMAs (8 independent chains), NE v_exp_f32
template <int NM, int NV, int NE, int SPLIT>
__global__ __launch_bounds__(1024) void k(const float* src, float* sink, int iters) {
    const int t = blockIdx.x * 1024 + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    bf8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)src[(t + i) & 1023]; b[i] = (__bf16)src[(t + 2 * i) & 1023]; }
    f4 acc[8];
    float v[8], e[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; v[i] = src[(t + i) & 1023]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = src[(t + 3 * i) & 1023] * 0.01f;
    const float c0 = src[5], c1 = src[7];
    // SPLIT: waves 0..7 (two per SIMD) run the MFMAs only, waves 8..15 the VALU work only
    const bool do_m = !SPLIT || wave < 8, do_v = !SPLIT || wave >= 8;
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], c0, c1);
#pragma unroll
            for (int i = 0; i < NE; ++i) e[i & 3] = __builtin_amdgcn_exp2f(e[i & 3]) * 0.5f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += e[i];
    if (s == 12345.678f) sink[t] = s;
}

// the same with integer VALU work (v_add_u32 / v_xor) and v_max3_f32 / v_cvt_pk_bf16_f32 instead of FMAs: KIND 0 int, 1 max3, 2 cvt_pk
template <int NM, int NV, int KIND>
__global__ __launch_bounds__(1024) void k2(const float* src, float* sink, int iters) {
    const int t = blockIdx.x * 1024 + threadIdx.x;
    bf8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)src[(t + i) & 1023]; b[i] = (__bf16)src[(t + 2 * i) & 1023]; }
    f4 acc[8];
    float v[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; v[i] = src[(t + i) & 1023]; u[i] = __float_as_uint(v[i]); }
    const float c0 = src[5], c1 = src[7];
    const unsigned k0 = __float_as_uint(c0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (KIND == 0) u[i & 7] = (u[i & 7] + k0) ^ (unsigned)it;
            else if constexpr (KIND == 1) v[i & 7] = __builtin_fmaxf(__builtin_fmaxf(v[i & 7], c0), v[(i + 1) & 7] * 0.f + c1);
            else { typedef __attribute__((ext_vector_type(2))) __bf16 b2; b2 p = {(__bf16)v[i & 7], (__bf16)v[(i + 3) & 7]}; v[i & 7] += __builtin_bit_cast(float, p) * 1e-30f; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i] + (float)u[i];
    if (s == 12345.678f) sink[t] = s;
}
template <int NM, int NV, int KIND>
static void run2(const char* name, const float* src, float* sink) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<NM, NV, KIND>), dim3(256), dim3(1024), 0, 0, src, sink, 10);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k2<NM, NV, KIND>), dim3(256), dim3(1024), 0, 0, src, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-64s %8.1f ns per loop iteration (%d MFMA, %d ops of kind %d per wave)\n", name, best * 1e6 / iters, NM, NV, KIND);
}

template <int NM, int NV, int NE, int SPLIT>
static void run(const char* name, const float* src, float* sink) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, NE, SPLIT>), dim3(256), dim3(1024), 0, 0, src, sink, 10);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NM, NV, NE, SPLIT>), dim3(256), dim3(1024), 0, 0, src, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // ns per iteration of the whole SIMD (4 waves): at ~2.1 GHz one cycle is ~0.48 ns
    printf("%-64s %8.1f ns per loop iteration (%d MFMA, %d FMA, %d exp per wave)\n", name, best * 1e6 / iters, NM, NV, NE);
}

int main() {
    float *src, *sink;
    hipMalloc(&src, 4096); hipMalloc(&sink, 256 * 1024 * 4);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97) - 0.03f;
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    run<32, 0, 0, 0>("MFMA only, 4 waves per SIMD", src, sink);
    run<0, 128, 0, 0>("FMA only", src, sink);
    run<0, 0, 32, 0>("exp only (+ 1 mul each)", src, sink);
    run<32, 128, 0, 0>("MFMA + FMA in every wave", src, sink);
    run<32, 64, 16, 0>("MFMA + FMA + exp in every wave", src, sink);
    run<64, 256, 0, 1>("2 waves MFMA only (x2 work), 2 waves FMA only (x2 work)", src, sink);
    run<64, 0, 0, 1>("  ... the MFMA waves alone (VALU waves idle)", src, sink);
    run<0, 256, 0, 1>("  ... the FMA waves alone", src, sink);
    // a light matrix load (no power throttling): 8 MFMAs against 128 FMAs per wave
    run<8, 0, 0, 0>("MFMA only, light", src, sink);
    run<8, 128, 0, 0>("MFMA light + FMA in every wave", src, sink);
    run<16, 256, 0, 1>("2 waves MFMA light (x2), 2 waves FMA (x2)", src, sink);
    run<0, 0, 32, 0>("exp only", src, sink);
    run<8, 0, 32, 0>("MFMA light + exp in every wave", src, sink);
    run2<0, 128, 0>("integer add + xor only", src, sink);
    run2<8, 128, 0>("MFMA light + integer add + xor", src, sink);
    run2<0, 64, 1>("max + mul + max only", src, sink);
    run2<8, 64, 1>("MFMA light + max + mul + max", src, sink);
    run2<0, 64, 2>("cvt_pk_bf16 + fma only", src, sink);
    run2<8, 64, 2>("MFMA light + cvt_pk_bf16 + fma", src, sink);
    return 0;
}
