// Small and HBM-bound kernels of the MuDPT path: patch im2col, prompt-row writes (the reference's
// torch.cat splices, clip/model.py:288,296,530,536, become in-place row writes), the deterministic
// batch reduction that is the splice's backward, the fp32 helpers for the prompt projections
// (trainers/mudpt.py:127-128, clip/model.py:539), the cosine-logit / cross-entropy head
// (trainers/mudpt.py:178-182,250) and the fused SGD update.
#include "kernels.h"

namespace mudpt {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4v;

// ---- patchify: images fp32 [B,3,S,S] -> patches T [B*P, 3*p*p], inner order (c, py, px) = conv1.weight.reshape(out, -1)
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, typename T::elem* __restrict__ out, int B,
                                                       int S, int p) {
    using elem = typename T::elem;
    const int x8n = S / 8;
    const size_t total = (size_t)B * 3 * S * x8n;
    const int g = S / p;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x8 = idx % x8n;
        size_t t = idx / x8n;
        const int y = t % S; t /= S;
        const int c = t % 3;
        const int b = t / 3;
        const f32x4* src = (const f32x4*)(img + (((size_t)b * 3 + c) * S + y) * S + x8 * 8);
        const f32x4 lo = src[0], hi = src[1];
        const int gy = y / p, py = y % p, x = x8 * 8, gx = x / p, px = x % p;
        typename T::vec8 v = {(elem)lo[0], (elem)lo[1], (elem)lo[2], (elem)lo[3], (elem)hi[0], (elem)hi[1], (elem)hi[2], (elem)hi[3]};
        *(typename T::vec8*)(out + ((size_t)b * g * g + gy * g + gx) * (3 * p * p) + (c * p + py) * p + px) = v;
    }
}

// Any patch size (ViT-L/14: p = 14, 3 p p = 588): one thread per output element, rows padded with zeros to ldk (a multiple
// of 64, the GEMM's K granularity; the weight rows are padded the same way).
template <typename T>
__global__ __launch_bounds__(256) void patchify_any_kernel(const float* __restrict__ img, typename T::elem* __restrict__ out, int B, int S, int p, int ldk) {
    using elem = typename T::elem;
    const int g = S / p, K0 = 3 * p * p;
    const size_t total = (size_t)B * g * g * ldk;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = idx % ldk;
        const size_t patch = idx / ldk;
        float v = 0.f;
        if (k < K0) {
            const int c = k / (p * p), py = (k / p) % p, px = k % p;
            const int gx = patch % g, gy = (patch / g) % g;
            const size_t b = patch / ((size_t)g * g);
            v = img[((b * 3 + c) * S + gy * p + py) * S + gx * p + px];
        }
        out[idx] = (elem)v;
    }
}

int launch_patchify(int dtype, const float* images, void* patches, int B, int S, int p, int ldk, hipStream_t s) {
    ARG_CHECK(images && patches && B > 0 && S > 0 && p > 0 && S % p == 0 && ldk >= 3 * p * p, "patchify: bad arguments S=%d p=%d ldk=%d", S, p, ldk);
    if (p % 8 == 0 && ldk == 3 * p * p) {
        const size_t total = (size_t)B * 3 * S * (S / 8);
        const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (dtype == DT_BF16) hipLaunchKernelGGL(patchify_kernel<BF16>, dim3(grid), dim3(256), 0, s, images, (__bf16*)patches, B, S, p);
        else if (dtype == DT_F16) hipLaunchKernelGGL(patchify_kernel<F16>, dim3(grid), dim3(256), 0, s, images, (_Float16*)patches, B, S, p);
        else { set_error("patchify: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    } else {
        const size_t total = (size_t)B * (S / p) * (S / p) * ldk;
        const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
        if (dtype == DT_BF16) hipLaunchKernelGGL(patchify_any_kernel<BF16>, dim3(grid), dim3(256), 0, s, images, (__bf16*)patches, B, S, p, ldk);
        else if (dtype == DT_F16) hipLaunchKernelGGL(patchify_any_kernel<F16>, dim3(grid), dim3(256), 0, s, images, (_Float16*)patches, B, S, p, ldk);
        else { set_error("patchify: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// Split-operand form (parity mode, common.h LoMode): hi = T(pixel) to `out`, the remainder pixel - hi to `out_lo` as T or as e4m3 bytes; four pixel slots per thread.
template <typename T>
__global__ __launch_bounds__(256) void patchify_split_kernel(const float* __restrict__ img, typename T::elem* __restrict__ out, void* __restrict__ out_lo, int lo_mode,
                                                             int B, int S, int p, int ldk) {
    using elem = typename T::elem;
    const int g = S / p, K0 = 3 * p * p;
    const size_t total = (size_t)B * g * g * (ldk / 4);  // four consecutive k per thread: one 4-byte store of e4m3 remainders
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k0 = (int)(idx % (ldk / 4)) * 4;
        const size_t patch = idx / (ldk / 4);
        const int gx = patch % g, gy = (patch / g) % g;
        const size_t b = patch / ((size_t)g * g);
        float rem[4];
        typename T::vec4 hv, lv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j;
            float v = 0.f;
            if (k < K0) {
                const int c = k / (p * p), py = (k / p) % p, px = k % p;
                v = img[((b * 3 + c) * S + gy * p + py) * S + gx * p + px];
            }
            elem hi;
            rem[j] = split_rem(v, hi);
            hv[j] = hi;
            lv[j] = (elem)rem[j];
        }
        *(typename T::vec4*)(out + patch * ldk + k0) = hv;
        if (lo_mode == LO_F8) *(uint32_t*)((char*)out_lo + patch * ldk * 2 + k0) = pack_lo8(rem[0], rem[1], rem[2], rem[3]);
        else *(typename T::vec4*)((elem*)out_lo + patch * ldk + k0) = lv;
    }
}

int launch_patchify_split(int dtype, const float* images, void* patches, void* patches_lo, int lo_mode, int B, int S, int p, int ldk, hipStream_t s) {
    ARG_CHECK(images && patches && patches_lo && B > 0 && S > 0 && p > 0 && S % p == 0 && ldk >= 3 * p * p && ldk % 4 == 0, "patchify: bad arguments S=%d p=%d ldk=%d", S, p, ldk);
    ARG_CHECK(lo_mode == LO_F16 || lo_mode == LO_F8, "patchify: bad lo_mode %d", lo_mode);
    const size_t total = (size_t)B * (S / p) * (S / p) * (ldk / 4);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == DT_BF16) hipLaunchKernelGGL(patchify_split_kernel<BF16>, dim3(grid), dim3(256), 0, s, images, (__bf16*)patches, patches_lo, lo_mode, B, S, p, ldk);
    else if (dtype == DT_F16) hipLaunchKernelGGL(patchify_split_kernel<F16>, dim3(grid), dim3(256), 0, s, images, (_Float16*)patches, patches_lo, lo_mode, B, S, p, ldk);
    else { set_error("patchify: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- set_rows: x[b, row0 + i, :] = rows[i, :] (+ add[i, :])
__global__ __launch_bounds__(256) void set_rows_kernel(float* __restrict__ x, int L, int d, int row0, int n, const float* __restrict__ rows,
                                                       const float* __restrict__ add) {
    const int b = blockIdx.x / n, i = blockIdx.x % n;
    f32x4* dst = (f32x4*)(x + ((size_t)b * L + row0 + i) * d);
    const f32x4* r = (const f32x4*)(rows + (size_t)i * d);
    const f32x4* a = add ? (const f32x4*)(add + (size_t)i * d) : nullptr;
    for (int k = threadIdx.x; k < d / 4; k += blockDim.x) dst[k] = a ? r[k] + a[k] : r[k];
}

int launch_set_rows(float* x, int B, int L, int d, int row0, int n, const float* rows, const float* add, hipStream_t s) {
    ARG_CHECK(x && rows && B > 0 && n > 0 && row0 >= 0 && row0 + n <= L && d % 4 == 0, "set_rows: bad arguments");
    hipLaunchKernelGGL(set_rows_kernel, dim3(B * n), dim3(256), 0, s, x, L, d, row0, n, rows, add);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- move_rows: dst[dmap(r)] = src[smap(r)] for whole rows of row_bytes (multiple of 16), map = rows[r] or r: the gather /
// scatter of the one row per sequence (CLS / EOT token) that the last block's tail works on
__global__ __launch_bounds__(256) void move_rows_kernel(const char* __restrict__ src, size_t src_stride, const int* __restrict__ src_rows,
                                                        char* __restrict__ dst, size_t dst_stride, const int* __restrict__ dst_rows, int row_bytes) {
    const int r = blockIdx.x;
    const u32x4v* in = (const u32x4v*)(src + (size_t)(src_rows ? src_rows[r] : r) * src_stride);
    u32x4v* out = (u32x4v*)(dst + (size_t)(dst_rows ? dst_rows[r] : r) * dst_stride);
    for (int k = threadIdx.x; k < row_bytes / 16; k += blockDim.x) out[k] = in[k];
}

int launch_gather_rows(const void* src, size_t src_stride, const int* rows, void* dst, size_t dst_stride, int nrows, int row_bytes, hipStream_t s) {
    ARG_CHECK(src && dst && rows && nrows > 0 && row_bytes > 0 && row_bytes % 16 == 0 && src_stride % 16 == 0 && dst_stride % 16 == 0, "gather_rows: bad arguments");
    hipLaunchKernelGGL(move_rows_kernel, dim3(nrows), dim3(256), 0, s, (const char*)src, src_stride, rows, (char*)dst, dst_stride, (const int*)nullptr, row_bytes);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

int launch_scatter_rows(const void* src, size_t src_stride, const int* rows, void* dst, size_t dst_stride, int nrows, int row_bytes, hipStream_t s) {
    ARG_CHECK(src && dst && rows && nrows > 0 && row_bytes > 0 && row_bytes % 16 == 0 && src_stride % 16 == 0 && dst_stride % 16 == 0, "scatter_rows: bad arguments");
    hipLaunchKernelGGL(move_rows_kernel, dim3(nrows), dim3(256), 0, s, (const char*)src, src_stride, (const int*)nullptr, (char*)dst, dst_stride, rows, row_bytes);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- add_rows: dst[rows[r]] += src[r] in T (the single query's share of the last block's input gradient)
template <typename T>
__global__ __launch_bounds__(128) void add_rows_kernel(const typename T::elem* __restrict__ src, const int* __restrict__ rows, typename T::elem* __restrict__ dst, int d) {
    using vec8 = typename T::vec8;
    const int r = blockIdx.x;
    const vec8* in = (const vec8*)(src + (size_t)r * d);
    vec8* out = (vec8*)(dst + (size_t)rows[r] * d);
    for (int k = threadIdx.x; k < d / 8; k += blockDim.x) {
        vec8 a = out[k];
        const vec8 b = in[k];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (typename T::elem)((float)a[i] + (float)b[i]);
        out[k] = a;
    }
}
int launch_add_rows(int dtype, const void* src, const int* rows, void* dst, int nrows, int d, hipStream_t s) {
    ARG_CHECK(src && rows && dst && nrows > 0 && d > 0 && d % 8 == 0, "add_rows: bad arguments");
    if (dtype == DT_BF16) hipLaunchKernelGGL(add_rows_kernel<BF16>, dim3(nrows), dim3(128), 0, s, (const __bf16*)src, rows, (__bf16*)dst, d);
    else if (dtype == DT_F16) hipLaunchKernelGGL(add_rows_kernel<F16>, dim3(nrows), dim3(128), 0, s, (const _Float16*)src, rows, (_Float16*)dst, d);
    else { set_error("add_rows: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- reduce_rows: out[i, c] (+)= sum_b src[b, row0 + i, c] (b ascending: bitwise reproducible); optional zeroing
template <typename T>
__global__ __launch_bounds__(256) void reduce_rows_kernel(float* __restrict__ src, typename T::elem* __restrict__ src_lp, int B, int L, int d,
                                                          int row0, int n, float* __restrict__ out, bool zero_src, bool accumulate,
                                                          float scale) {
    // block = 32 columns x 8 batch segments; segment partials are combined in segment order: the summation tree
    // depends only on (B, launch shape), never on timing -> bitwise reproducible.
    __shared__ float part[8][33];
    const int cx = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + cx;
    const bool live = idx < n * d;
    const int i = live ? idx / d : 0, c = live ? idx % d : 0;
    const int per = (B + 7) / 8, b0 = seg * per, b1 = b0 + per < B ? b0 + per : B;
    float acc = 0.f;
    if (live) {
        // 32 loads in flight per thread (B = 256: the whole segment in one round trip; the loop is a pure latency chain), added in ascending b
        const size_t step = (size_t)L * d, o0 = ((size_t)row0 + i) * d + c;
        for (int bb = b0; bb < b1; bb += 32) {
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const size_t o = (size_t)(bb + k) * step + o0;
                v[k] = bb + k < b1 ? (src ? src[o] : (float)src_lp[o]) : 0.f;  // src == null: the gradient stream lives in T only
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                acc += v[k];
                if (zero_src && bb + k < b1) {
                    const size_t o = (size_t)(bb + k) * step + o0;
                    if (src) src[o] = 0.f;
                    if (src_lp) src_lp[o] = (typename T::elem)0.f;
                }
            }
        }
    }
    part[seg][cx] = acc;
    __syncthreads();
    if (seg == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) t += part[s2][cx];
        t *= scale;
        out[idx] = accumulate ? out[idx] + t : t;
    }
}

int launch_reduce_rows(int dtype, float* src, void* src_lp, int B, int L, int d, int row0, int n, float* out, bool zero_src,
                       bool accumulate, float scale, hipStream_t s) {
    ARG_CHECK((src || src_lp) && out && B > 0 && n > 0 && row0 >= 0 && row0 + n <= L, "reduce_rows: bad arguments");
    const int grid = (n * d + 31) / 32;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(reduce_rows_kernel<BF16>, dim3(grid), dim3(256), 0, s, src, (__bf16*)src_lp, B, L, d, row0, n, out, zero_src, accumulate, scale);
    else if (dtype == DT_F16)
        hipLaunchKernelGGL(reduce_rows_kernel<F16>, dim3(grid), dim3(256), 0, s, src, (_Float16*)src_lp, B, L, d, row0, n, out, zero_src, accumulate, scale);
    else { set_error("reduce_rows: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- small fp32 GEMM: C = alpha * op(A) op(B) + bias + beta * C (the prompt projections, M = 4..44 rows; the feature projections)
// One workgroup per 16 x 16 output tile, and the tile is a LATENCY chain over K (a few global round trips), not a throughput problem:
// the 8 waves split K between them, each wave feeds v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate) straight from global
// memory -- lane (r, g) supplies A[m0 + r][k] and B[k][n0 + r] for k = k0 + 4 g + u, any partition of k works as long as A and B agree --
// with all of its loads of up to 96 k in flight before the first MFMA, and the 8 partial tiles are summed through LDS in wave order
// (fixed order: bitwise reproducible).  K = 768: one round trip of 24 loads per lane instead of 12 dependent chunk round trips.
constexpr int SG_WAVES = 8, SG_STEPS = 6;  // waves per tile; 16-k steps whose loads travel together
__global__ __launch_bounds__(SG_WAVES * 64) void sgemm_kernel(bool tA, bool tB, int M, int N, int K, float alpha, const float* __restrict__ A, int lda,
                                                              const float* __restrict__ B, int ldb, float beta, float* __restrict__ C, int ldc,
                                                              const float* __restrict__ bias) {
    __shared__ float part[SG_WAVES][16][17];
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int per = ((K + SG_WAVES - 1) / SG_WAVES + 15) & ~15;  // k per wave, a multiple of the 16-k step
    const int kb = wave * per, ke = kb + per < K ? kb + per : K;
    const bool row_ok = m0 + r < M, col_ok = n0 + r < N;
    const float* Ar = tA ? A + m0 + r : A + (size_t)(m0 + r) * lda;  // element (m0 + r, k): Ar[k * lda] (tA) / Ar[k]
    const float* Bc = tB ? B + (size_t)(n0 + r) * ldb : B + n0 + r;  // element (k, n0 + r): Bc[k] (tB) / Bc[k * ldb]
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = kb; k0 < ke; k0 += 16 * SG_STEPS) {
        float av[SG_STEPS][4], bv[SG_STEPS][4];
#pragma unroll
        for (int st = 0; st < SG_STEPS; ++st)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 16 * st + 4 * g + u;
                av[st][u] = (row_ok && k < ke) ? (tA ? Ar[(size_t)k * lda] : Ar[k]) : 0.f;
                bv[st][u] = (col_ok && k < ke) ? (tB ? Bc[k] : Bc[(size_t)k * ldb]) : 0.f;
            }
#pragma unroll
        for (int st = 0; st < SG_STEPS; ++st)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][u], bv[st][u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) part[wave][4 * g + q][r] = acc[q];  // D: column lane & 15, rows 4 (lane >> 4) + q
    __syncthreads();
    if (threadIdx.x < 256) {
        const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15, m = m0 + ty, n = n0 + tx;
        if (m < M && n < N) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < SG_WAVES; ++w) v += part[w][ty][tx];
            v *= alpha;
            if (bias) v += bias[n];
            if (beta != 0.f) v += beta * C[(size_t)m * ldc + n];
            C[(size_t)m * ldc + n] = v;
        }
    }
}

int launch_sgemm(bool tA, bool tB, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb, float beta,
                 float* C, int ldc, const float* bias, hipStream_t s) {
    ARG_CHECK(A && B && C && M > 0 && N > 0 && K > 0, "sgemm: bad arguments M=%d N=%d K=%d", M, N, K);
    hipLaunchKernelGGL(sgemm_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(SG_WAVES * 64), 0, s, tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int M, int N, int lda, float* __restrict__ out, bool accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc += A[(size_t)m * lda + n];
    out[n] = accumulate ? out[n] + acc : acc;
}

int launch_colsum(const float* A, int M, int N, int lda, float* out, bool accumulate, hipStream_t s) {
    ARG_CHECK(A && out && M > 0 && N > 0, "colsum: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256), dim3(256), 0, s, A, M, N, lda, out, accumulate);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}

int launch_add(const float* a, const float* b, float* y, size_t n, hipStream_t s) {
    ARG_CHECK(a && b && y && n > 0, "add: bad arguments");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(add_kernel, dim3(grid), dim3(256), 0, s, a, b, y, n);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, typename T::elem* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (typename T::elem)x[i];
}

int launch_cast(int dtype, const float* x, void* y, size_t n, hipStream_t s) {
    ARG_CHECK(x && y && n > 0, "cast: bad arguments");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (dtype == DT_BF16) hipLaunchKernelGGL(cast_kernel<BF16>, dim3(grid), dim3(256), 0, s, x, (__bf16*)y, n);
    else if (dtype == DT_F16) hipLaunchKernelGGL(cast_kernel<F16>, dim3(grid), dim3(256), 0, s, x, (_Float16*)y, n);
    else { set_error("cast: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- head -------------------------------------------------------------------------------------------
// y = x / ||x|| per row, inv = 1 / ||x||   (one wave per row)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv, int rows, int e) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int i = lane; i < e; i += 64) { const float v = x[(size_t)r * e + i]; s += v * v; }
    const float iv = 1.0f / sqrtf(wave_sum(s));
    for (int i = lane; i < e; i += 64) y[(size_t)r * e + i] = x[(size_t)r * e + i] * iv;
    if (lane == 0) inv[r] = iv;
}
// dx = inv * (dy - y * <dy, y>), in place on dy
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ inv, int rows, int e) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float s = 0.f;
    for (int i = lane; i < e; i += 64) s += dy[(size_t)r * e + i] * y[(size_t)r * e + i];
    s = wave_sum(s);
    const float iv = inv[r];
    for (int i = lane; i < e; i += 64) dy[(size_t)r * e + i] = iv * (dy[(size_t)r * e + i] - y[(size_t)r * e + i] * s);
}
// per-row cross entropy: row_loss[b] = lse(logits[b]) - logits[b, label]; dlogits = (softmax - onehot) * gscale
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ row_loss,
                                                      float* __restrict__ dlogits, int B, int C, float gscale) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;
    const float* z = logits + (size_t)r * C;
    float m = -INFINITY;
    for (int i = lane; i < C; i += 64) m = fmaxf(m, z[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += __expf(z[i] - m);
    s = wave_sum(s);
    // a label outside [0, C) (torch's F.cross_entropy asserts on it) must not become an out-of-bounds read: the row's loss
    // is NaN, which the trainer's non-finite-loss check reports, and no one-hot is subtracted
    const int64_t y64 = labels[r];
    const bool y_ok = y64 >= 0 && y64 < (int64_t)C;
    const int y = y_ok ? (int)y64 : -1;
    if (lane == 0) row_loss[r] = y_ok ? (m + __logf(s)) - z[y] : __builtin_nanf("");
    if (dlogits) {
        const float is = 1.f / s;
        for (int i = lane; i < C; i += 64) dlogits[(size_t)r * C + i] = (__expf(z[i] - m) * is - (i == y ? 1.f : 0.f)) * gscale;
    }
}
// loss = mean of row_loss in fixed order (one block, tree over a fixed partition: reproducible)
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float part[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0] / n;
}

int launch_l2norm(const float* x, float* y, float* inv, int rows, int e, hipStream_t s) {
    ARG_CHECK(x && y && inv && rows > 0 && e > 0, "l2norm: bad arguments");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, y, inv, rows, e);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

static int check_head(const HeadArgs& a) {
    ARG_CHECK(a.img && a.txt && a.logits && a.img_n && a.txt_n && a.img_inv && a.txt_inv, "head: null operand");
    ARG_CHECK(a.B > 0 && a.C > 0 && a.e > 0, "head: bad shape B=%d C=%d e=%d", a.B, a.C, a.e);
    return MUDPT_OK;
}

int launch_head_fwd(const HeadArgs& a, hipStream_t s) {
    if (int e = check_head(a)) return e;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a.img, a.img_n, a.img_inv, a.B, a.e);
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((a.C + 3) / 4), dim3(256), 0, s, a.txt, a.txt_n, a.txt_inv, a.C, a.e);
    HIP_TRY(hipGetLastError());
    // logits = scale * img_n . txt_n^T
    return launch_sgemm(false, true, a.B, a.C, a.e, a.scale, a.img_n, a.e, a.txt_n, a.e, 0.f, a.logits, a.C, nullptr, s);
}

int launch_head_bwd(const HeadArgs& a, hipStream_t s) {
    if (int e = check_head(a)) return e;
    ARG_CHECK(a.labels && a.loss && a.dlogits && a.row_loss && a.dimg && a.dtxt, "head bwd: null operand");
    hipLaunchKernelGGL(ce_rows_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a.logits, a.labels, a.row_loss, a.dlogits, a.B, a.C, a.grad_scale / a.B);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, a.row_loss, a.B, a.loss);
    HIP_TRY(hipGetLastError());
    // d img_n = scale * dlogits . txt_n ; d txt_n = scale * dlogits^T . img_n
    if (int e = launch_sgemm(false, false, a.B, a.e, a.C, a.scale, a.dlogits, a.C, a.txt_n, a.e, 0.f, a.dimg, a.e, nullptr, s)) return e;
    if (int e = launch_sgemm(true, false, a.C, a.e, a.B, a.scale, a.dlogits, a.C, a.img_n, a.e, 0.f, a.dtxt, a.e, nullptr, s)) return e;
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a.dimg, a.img_n, a.img_inv, a.B, a.e);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((a.C + 3) / 4), dim3(256), 0, s, a.dtxt, a.txt_n, a.txt_inv, a.C, a.e);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- CoCoOp head: every image has its OWN text features (trainers/cocoop.py:187-196) ---------------------------------
// logits[i, c] = scale * <img_n[i], txt_n[i * C + c]>   (one wave per (i, c))
__global__ __launch_bounds__(256) void pair_logits_kernel(const float* __restrict__ img_n, const float* __restrict__ txt_n, float* __restrict__ logits,
                                                          int B, int C, int e, float scale) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * C) return;
    const float* a = img_n + (size_t)(r / C) * e;
    const float* t = txt_n + (size_t)r * e;
    float s = 0.f;
    for (int k = lane; k < e; k += 64) s += a[k] * t[k];
    s = wave_sum(s);
    if (lane == 0) logits[r] = scale * s;
}
// d txt_n[i * C + c, :] = scale * dlogits[i, c] * img_n[i, :]
__global__ __launch_bounds__(256) void pair_dtxt_kernel(const float* __restrict__ dlogits, const float* __restrict__ img_n, float* __restrict__ dtxt,
                                                        int B, int C, int e, float scale) {
    const int r = blockIdx.x;
    const float g = scale * dlogits[r];
    const float* a = img_n + (size_t)(r / C) * e;
    for (int k = threadIdx.x; k < e; k += blockDim.x) dtxt[(size_t)r * e + k] = g * a[k];
}

// a.txt / txt_n / txt_inv / dtxt have B * C rows (row i * C + c); a.dimg is not produced (the image encoder is frozen and
// meta_net's input is a constant of the step: nothing upstream of the image features trains, trainers/cocoop.py:222-226)
int launch_mean(const float* v, int n, float* out, hipStream_t s) {
    ARG_CHECK(v && out && n > 0, "mean: bad arguments");
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, v, n, out);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// the image features are normalised by the caller (they also feed meta_net): a.img_n / a.img_inv are inputs here
int launch_pair_head_fwd(const HeadArgs& a, hipStream_t s) {
    if (int e = check_head(a)) return e;
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((a.B * a.C + 3) / 4), dim3(256), 0, s, a.txt, a.txt_n, a.txt_inv, a.B * a.C, a.e);
    hipLaunchKernelGGL(pair_logits_kernel, dim3((a.B * a.C + 3) / 4), dim3(256), 0, s, a.img_n, a.txt_n, a.logits, a.B, a.C, a.e, a.scale);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

int launch_pair_head_bwd(const HeadArgs& a, hipStream_t s) {
    if (int e = check_head(a)) return e;
    ARG_CHECK(a.labels && a.loss && a.dlogits && a.row_loss && a.dtxt, "pair head bwd: null operand");
    const int Bt = a.B_total > 0 ? a.B_total : a.B;  // chunked over the images: the mean is over the whole batch, taken by the caller
    hipLaunchKernelGGL(ce_rows_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a.logits, a.labels, a.row_loss, a.dlogits, a.B, a.C, a.grad_scale / Bt);
    if (a.B_total <= 0) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, a.row_loss, a.B, a.loss);
    hipLaunchKernelGGL(pair_dtxt_kernel, dim3(a.B * a.C), dim3(128), 0, s, a.dlogits, a.img_n, a.dtxt, a.B, a.C, a.e, a.scale);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((a.B * a.C + 3) / 4), dim3(256), 0, s, a.dtxt, a.txt_n, a.txt_inv, a.B * a.C, a.e);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- CoCoOp prompt construction (trainers/cocoop.py:148-165) + positional embedding (:52) ---------------------------------
// x0[(i, c), l, :] = emb_pos[c, l, :] for l outside 1..n;  = ctx[l-1] + bias[i] + pos[l] for the n context rows
__global__ __launch_bounds__(128) void cocoop_prompts_kernel(float* __restrict__ x0, const float* __restrict__ emb_pos, const float* __restrict__ ctx,
                                                             const float* __restrict__ bias, const float* __restrict__ pos, int C, int L, int d, int n) {
    const int row = blockIdx.x;  // (i * C + c) * L + l
    const int l = row % L, seq = row / L, c = seq % C, i = seq / C;
    f32x4* dst = (f32x4*)(x0 + (size_t)row * d);
    if (l >= 1 && l <= n) {
        const f32x4 *a = (const f32x4*)(ctx + (size_t)(l - 1) * d), *b = (const f32x4*)(bias + (size_t)i * d), *p = (const f32x4*)(pos + (size_t)l * d);
        for (int k = threadIdx.x; k < d / 4; k += blockDim.x) dst[k] = a[k] + b[k] + p[k];
    } else {
        const f32x4* src = (const f32x4*)(emb_pos + ((size_t)c * L + l) * d);
        for (int k = threadIdx.x; k < d / 4; k += blockDim.x) dst[k] = src[k];
    }
}
int launch_cocoop_prompts(float* x0, const float* emb_pos, const float* ctx, const float* bias, const float* pos, int B, int C, int L, int d, int n, hipStream_t s) {
    ARG_CHECK(x0 && emb_pos && ctx && bias && pos && B > 0 && C > 0 && n > 0 && 1 + n < L && d % 4 == 0, "cocoop_prompts: bad arguments");
    hipLaunchKernelGGL(cocoop_prompts_kernel, dim3(B * C * L), dim3(128), 0, s, x0, emb_pos, ctx, bias, pos, C, L, d, n);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}
// d bias[i, :] = scale * sum_c sum_r dx[(i, c), 1 + r, :]  (c, then r ascending: a fixed order, bitwise reproducible)
template <typename T>
__global__ __launch_bounds__(128) void cocoop_dbias_kernel(const float* __restrict__ dx, const typename T::elem* __restrict__ dx_lp, float* __restrict__ dbias,
                                                           int C, int L, int d, int n, float scale) {
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float acc = 0.f;
        for (int c = 0; c < C; ++c)
            for (int r = 0; r < n; ++r) {
                const size_t o = (((size_t)i * C + c) * L + 1 + r) * d + k;
                acc += dx ? dx[o] : (float)dx_lp[o];
            }
        dbias[(size_t)i * d + k] = acc * scale;
    }
}
int launch_cocoop_dbias(int dtype, const float* dx, const void* dx_lp, float* dbias, int B, int C, int L, int d, int n, float scale, hipStream_t s) {
    ARG_CHECK((dx || dx_lp) && dbias && B > 0 && C > 0 && n > 0 && 1 + n < L, "cocoop_dbias: bad arguments");
    if (dtype == DT_BF16) hipLaunchKernelGGL(cocoop_dbias_kernel<BF16>, dim3(B), dim3(128), 0, s, dx, (const __bf16*)dx_lp, dbias, C, L, d, n, scale);
    else if (dtype == DT_F16) hipLaunchKernelGGL(cocoop_dbias_kernel<F16>, dim3(B), dim3(128), 0, s, dx, (const _Float16*)dx_lp, dbias, C, L, d, n, scale);
    else { set_error("cocoop_dbias: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}
// ReLU in place, and its backward: dy *= (y > 0)   (meta_net, trainers/cocoop.py:103-107)
__global__ __launch_bounds__(256) void relu_kernel(float* __restrict__ y, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) y[i] = fmaxf(y[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dy[i] = y[i] > 0.f ? dy[i] : 0.f;
}
int launch_relu(float* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(relu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, n);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}
int launch_relu_bwd(float* dy, const float* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, y, n);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ---- SGD (torch.optim.SGD update rule) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n, float lr,
                                                  float momentum, float wd, float dampening, bool nesterov, bool first) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float d = g[i] + wd * p[i];
        if (momentum != 0.f) {
            const float b = first ? d : momentum * buf[i] + (1.f - dampening) * d;
            buf[i] = b;
            d = nesterov ? d + momentum * b : b;
        }
        p[i] -= lr * d;
    }
}

int launch_sgd(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd, float dampening, bool nesterov,
               bool first, hipStream_t s) {
    ARG_CHECK(p && g && n > 0 && (momentum == 0.f || buf), "sgd: bad arguments");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, s, p, g, buf, n, lr, momentum, wd, dampening, nesterov, first);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

}  // namespace mudpt
