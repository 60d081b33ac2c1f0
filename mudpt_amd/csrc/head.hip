// Fused cosine-logit / cross-entropy head for gfx950 (trainers/mudpt.py:178-182 and F.cross_entropy at :250).
//
//   logits[B, C] = exp(logit_scale) * normalise(img)[B, e] . normalise(txt)[C, e]^T,  loss = mean_b CE(logits[b], label[b])
//
// The contraction runs on the matrix cores in EXACT fp32 (v_mfma_f32_16x16x4_f32: a k-ordered fp32 fma chain, bit for bit; the
// logits feed a 1e-3 parity bound at logit scale 14-100, so bf16 / fp16 operands are not an option here), and the row softmax /
// cross-entropy / dlogits never leave the workgroup that made the logit rows:
//   head_rows_kernel  (one workgroup per 16 images): normalise the 16 image rows -> LDS; logit tile [16, C] by MFMA -> LDS + HBM;
//       [training] per-row log-sum-exp, loss row, dlogits = (softmax - onehot) * gscale in place; d(normalised image features) =
//       scale * dlogits . txt_n by MFMA; backward of the normalisation -> dimg.
//   head_dtxt_kernel  (one workgroup per 16 classes, training): d(normalised text features) = scale * dlogits^T . img_n by MFMA
//       (images in ascending order: a fixed summation order), backward of the normalisation -> dtxt.
// plus the existing row-normalisation of the C text features and the fixed-order mean of the loss rows: 4 launches instead of 9.
// MFMA operand maps (16x16x4 f32): lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]; D: column l & 15, rows 4 (l >> 4) + r.
// Four consecutive k are fetched as one float4 per operand and fed to four MFMAs; any partition of k works as long as A and B agree.
#include "kernels.h"

namespace mudpt {

constexpr int HEAD_ROWS = 16, HEAD_WAVES = 8, HEAD_RPW = HEAD_ROWS / HEAD_WAVES;  // 16 rows per workgroup, 8 waves, 2 rows per wave in the row-wise phases

// acc[t] += A[16, K] . B[K, 16 tile t]: A from `a_row` (this lane's row, k contiguous: LDS or global), B rows from `b_base + k * ldb`
// (k-major operand: element (k, col) at b[k * ldb + col]); K a multiple of 16; rows of B beyond k_valid read as zero.
template <int NT>
__device__ inline void mfma_kmajor(f32x4 (&acc)[NT], const float* a_row, int a_stride_k, const float* b_base, size_t ldb, const int (&col)[NT],
                                   const bool (&col_ok)[NT], int K, int k_valid, int g) {
    // 4 steps' operands are requested before the first MFMA (the loop is otherwise one dependent memory round trip per step); K % 16 == 0
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[4], bv[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u + g;
            av[u] = a_row[(size_t)k * a_stride_k];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[u][t] = (col_ok[t] && k < k_valid) ? b_base[(size_t)k * ldb + col[t]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][t], acc[t], 0, 0, 0);
    }
}

// y = inv * (d - n * <d, n>) for 4 rows per wave: backward of n = x / ||x||; d and n are LDS / global rows of length e
__device__ inline void l2norm_bwd_row(const float* d, const float* n, float inv, float* out, int e, int lane) {
    float s = 0.f;
    for (int k = lane; k < e; k += 64) s += d[k] * n[k];
    s = wave_sum(s);
    for (int k = lane; k < e; k += 64) out[k] = inv * (d[k] - n[k] * s);
}

template <bool TRAIN>
__global__ __launch_bounds__(HEAD_WAVES * 64) void head_rows_kernel(HeadArgs p, int Cpad) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float* In = hsm;                          // [16][e] normalised image rows
    float* Zt = In + HEAD_ROWS * p.e;         // [16][Cpad] logits, then dlogits
    float* Dn = Zt + HEAD_ROWS * Cpad;        // [16][e] d(normalised image features)   (TRAIN only)
    __shared__ float inv_s[HEAD_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * HEAD_ROWS, e = p.e, C = p.C, B = p.B;

    // ---- 1. normalise the image rows (HEAD_RPW per wave) ----------------------------------------------------------------------------
    for (int i = 0; i < HEAD_RPW; ++i) {
        const int rl = wave * HEAD_RPW + i, r = r0 + rl;
        float s = 0.f;
        if (r < B)
            for (int k = lane; k < e; k += 64) { const float v = p.img[(size_t)r * e + k]; s += v * v; }
        const float iv = r < B ? 1.0f / sqrtf(wave_sum(s)) : 0.f;
        for (int k = lane; k < e; k += 64) {
            const float v = r < B ? p.img[(size_t)r * e + k] * iv : 0.f;
            In[rl * e + k] = v;
            if (r < B && p.img_n) p.img_n[(size_t)r * e + k] = v;
        }
        if (lane == 0) {
            inv_s[rl] = iv;
            if (r < B && p.img_inv) p.img_inv[r] = iv;
        }
    }
    __syncthreads();

    // ---- 2. logits: class tiles of 16 over the waves; k = e in float4 pieces (four MFMAs each) ----------------------------------
    for (int ct = wave; ct < Cpad / 16; ct += HEAD_WAVES) {
        const int cls = ct * 16 + c;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* ar = In + c * e + 4 * g;
        const float* br = p.txt_n + (size_t)(cls < C ? cls : 0) * e + 4 * g;
        for (int j0 = 0; j0 < e; j0 += 128) {  // 8 float4 pairs requested before the first MFMA: the loads overlap instead of chaining
            f32x4 a4[8], b4[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 16 * u;
                const bool in = j < e;
                a4[u] = in ? *(const f32x4*)(ar + j) : f32x4{0.f, 0.f, 0.f, 0.f};
                b4[u] = (in && cls < C) ? *(const f32x4*)(br + j) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t], b4[u][t], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = 4 * g + r;
            const float z = p.scale * acc[r];
            Zt[rl * Cpad + cls] = z;
            if (r0 + rl < B && cls < C) p.logits[(size_t)(r0 + rl) * C + cls] = z;
        }
    }
    if constexpr (!TRAIN) return;
    __syncthreads();

    // ---- 3. cross-entropy rows; dlogits = (softmax - onehot) * gscale, in place ---------------------------------------------------
    const float gscale = p.grad_scale / (p.B_total > 0 ? p.B_total : B);
    for (int i = 0; i < HEAD_RPW; ++i) {
        const int rl = wave * HEAD_RPW + i, r = r0 + rl;
        float* z = Zt + rl * Cpad;
        if (r >= B) {  // padding rows contribute nothing to the gradients
            for (int k = lane; k < Cpad; k += 64) z[k] = 0.f;
            continue;
        }
        float m = -INFINITY;
        for (int k = lane; k < C; k += 64) m = fmaxf(m, z[k]);
        m = wave_max(m);
        float s = 0.f;
        for (int k = lane; k < C; k += 64) s += __expf(z[k] - m);
        s = wave_sum(s);
        // a label outside [0, C) (torch's F.cross_entropy asserts on it): NaN loss row, no one-hot, never an out-of-bounds access
        const int64_t y64 = p.labels[r];
        const bool y_ok = y64 >= 0 && y64 < (int64_t)C;
        const int y = y_ok ? (int)y64 : -1;
        if (lane == 0) p.row_loss[r] = y_ok ? (m + __logf(s)) - z[y] : __builtin_nanf("");
        const float is = 1.f / s;
        for (int k = lane; k < Cpad; k += 64) {
            const float d = k < C ? (__expf(z[k] - m) * is - (k == y ? 1.f : 0.f)) * gscale : 0.f;
            z[k] = d;
            if (k < C && p.dlogits) p.dlogits[(size_t)r * C + k] = d;
        }
    }
    __syncthreads();

    // ---- 4. d(normalised image features)[16, e] = scale * dlogits[16, C] . txt_n[C, e]: e tiles over the waves, 4 at a time -------
    {
        const int ne = e / 16;
        for (int et0 = wave * 4; et0 < ne; et0 += 4 * HEAD_WAVES) {
            f32x4 acc[4];
            int col[4];
            bool ok[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; col[t] = (et0 + t) * 16 + c; ok[t] = et0 + t < ne; }
            mfma_kmajor<4>(acc, Zt + c * Cpad, 1, p.txt_n, (size_t)e, col, ok, Cpad, C, g);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (ok[t])
#pragma unroll
                    for (int r = 0; r < 4; ++r) Dn[(4 * g + r) * e + col[t]] = p.scale * acc[t][r];
        }
    }
    __syncthreads();
    // ---- 5. backward of the normalisation -> gradient of the raw image features ------------------------------------------------
    for (int i = 0; i < HEAD_RPW; ++i) {
        const int rl = wave * HEAD_RPW + i, r = r0 + rl;
        if (r < B) l2norm_bwd_row(Dn + rl * e, In + rl * e, inv_s[rl], p.dimg + (size_t)r * e, e, lane);
    }
}

// d(raw text features) of 16 classes: scale * dlogits^T[16, B] . img_n[B, e], then the backward of the normalisation
__global__ __launch_bounds__(HEAD_WAVES * 64) void head_dtxt_kernel(HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    float* Dn = hsm;  // [16][e]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.x * HEAD_ROWS, e = p.e, C = p.C, B = p.B;
    const int Bpad = (B + 3) & ~3, ne = e / 16;
    const int cls = c0 + c;
    // A[class c][k = image] = dlogits[image][class]: this lane's "row" is a column of dlogits, stride C between consecutive k
    const float* a_row = p.dlogits + (cls < C ? cls : 0);
    for (int et0 = wave * 4; et0 < ne; et0 += 4 * HEAD_WAVES) {
        f32x4 acc[4];
        int col[4];
        bool ok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; col[t] = (et0 + t) * 16 + c; ok[t] = et0 + t < ne; }
        // the sum over the images is a latency chain (one dependent global round trip per step of 4 images): 8 steps' operands are
        // requested before the first MFMA so the loads overlap; the MFMA order (images ascending) is unchanged
        for (int k0 = 0; k0 < Bpad; k0 += 32) {
            float av[8], bv[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 4 * u + g;
                av[u] = (k < B && cls < C) ? a_row[(size_t)k * C] : 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[u][t] = (ok[t] && k < B) ? p.img_n[(size_t)k * e + col[t]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][t], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (ok[t])
#pragma unroll
                for (int r = 0; r < 4; ++r) Dn[(4 * g + r) * e + col[t]] = p.scale * acc[t][r];
    }
    __syncthreads();
    for (int i = 0; i < HEAD_RPW; ++i) {
        const int rl = wave * HEAD_RPW + i, r = c0 + rl;
        if (r < C) l2norm_bwd_row(Dn + rl * e, p.txt_n + (size_t)r * e, p.txt_inv[r], p.dtxt + (size_t)r * e, e, lane);
    }
}

static int head_lds_bytes(const HeadArgs& a, bool train) {
    const int Cpad = (a.C + 15) / 16 * 16;
    return (HEAD_ROWS * a.e * (train ? 2 : 1) + HEAD_ROWS * Cpad) * 4;
}
// true if the fused kernels can hold a 16-row problem in LDS (otherwise the caller keeps the unfused fp32 path of elementwise.hip)
bool head_fused_fits(const HeadArgs& a, bool train) { return a.e % 16 == 0 && head_lds_bytes(a, train) + 256 <= 163840; }

// logits (and the normalised features the backward needs).  txt_n / txt_inv are (re)computed unless a.txt is null (cached text features).
int launch_head_fused_fwd(const HeadArgs& a, hipStream_t s) {
    ARG_CHECK(a.img && a.logits && a.txt_n && a.txt_inv && a.B > 0 && a.C > 0 && a.e > 0, "head: bad arguments");
    if (a.txt)
        if (int rc = launch_l2norm(a.txt, a.txt_n, a.txt_inv, a.C, a.e, s)) return rc;
    const int Cpad = (a.C + 15) / 16 * 16, lds = head_lds_bytes(a, false);
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)head_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840 - 256));
        HIP_TRY(hipFuncSetAttribute((const void*)head_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840 - 256));
        HIP_TRY(hipFuncSetAttribute((const void*)head_dtxt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        pd.done[dev] = true;
    }
    hipLaunchKernelGGL(head_rows_kernel<false>, dim3((a.B + HEAD_ROWS - 1) / HEAD_ROWS), dim3(HEAD_WAVES * 64), lds, s, a, Cpad);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// forward + mean cross-entropy + backward to the RAW features in one pass over the logit rows (logits are written too)
int launch_head_fused_train(const HeadArgs& a, hipStream_t s) {
    ARG_CHECK(a.img && a.logits && a.txt_n && a.txt_inv && a.img_n && a.img_inv && a.labels && a.loss && a.dlogits && a.row_loss && a.dimg && a.dtxt,
              "head: null operand");
    ARG_CHECK(a.B > 0 && a.C > 0 && a.e > 0 && a.e <= 1024, "head: bad shape B=%d C=%d e=%d", a.B, a.C, a.e);
    if (a.txt)
        if (int rc = launch_l2norm(a.txt, a.txt_n, a.txt_inv, a.C, a.e, s)) return rc;
    const int Cpad = (a.C + 15) / 16 * 16, lds = head_lds_bytes(a, true);
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)head_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840 - 256));
        HIP_TRY(hipFuncSetAttribute((const void*)head_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840 - 256));
        HIP_TRY(hipFuncSetAttribute((const void*)head_dtxt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        pd.done[dev] = true;
    }
    hipLaunchKernelGGL(head_rows_kernel<true>, dim3((a.B + HEAD_ROWS - 1) / HEAD_ROWS), dim3(HEAD_WAVES * 64), lds, s, a, Cpad);
    HIP_TRY(hipGetLastError());
    if (a.B_total <= 0)
        if (int rc = launch_mean(a.row_loss, a.B, a.loss, s)) return rc;
    hipLaunchKernelGGL(head_dtxt_kernel, dim3((a.C + HEAD_ROWS - 1) / HEAD_ROWS), dim3(HEAD_WAVES * 64), HEAD_ROWS * a.e * 4, s, a);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

}  // namespace mudpt
