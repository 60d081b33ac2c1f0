// Single-query attention for the LAST block of a tower (gfx950).
//
// Only one row per sequence of the last block's output is ever used -- the CLS token (clip/model.py:549) or the EOT token
// (trainers/mudpt.py:154) -- so that block needs ONE query per sequence against all (causal: the first pos + 1) keys.  The general
// kernels (attention.hip) would compute all L queries and, in the backward, sweep a dO that is zero except on one row.  Here one
// wave handles one (sequence, head) pair:
//   forward : s[k] = q . K[k] (key on the lane), softmax over the wave, o = sum_k p[k] V[k] (head dimension on the lane);
//   backward: p and dP = dO . V[k] recomputed (key on the lane), dS = p (dP - delta); then with the head dimension on the lane
//             dq = sum_k dS[k] K[k], dK[k] = dS[k] q, dV[k] = p[k] dO written for EVERY key row (zeros beyond a causal limit), so
//             the dX GEMM that follows reads a fully defined [M, 2 d] operand.
// q / o / dO / dq are compact [nseq, H * 64] arrays (one row per sequence); K and V are read from the packed qkv buffer
// [nseq, L, 3 * H * 64] whose q third is never touched.  HBM-bound: K and V are read once (forward) / twice with the second pass from
// L2 (backward), dK and dV written once.  Sums run in a fixed order: bitwise reproducible.
#include "kernels.h"

namespace mudpt {

constexpr float LOG2E_S = 1.4426950408889634f;
constexpr float SC_S = 0.125f * LOG2E_S;

template <typename T>
__device__ inline float dot64(const typename T::elem* row, const float* qs) {
    using vec8 = typename T::vec8;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const vec8 v = *(const vec8*)(row + 8 * c);
#pragma unroll
        for (int i = 0; i < 8; ++i) s = __builtin_fmaf((float)v[i], qs[8 * c + i], s);
    }
    return s;
}

// grid: ceil(nseq * H / 4) workgroups of 4 waves; dynamic LDS: 4 waves x (64 + Lpad) floats
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_single_kernel(AttnArgs p, const void* q_sel, void* out_sel, void* out_lo, int ld_out, float* lse_sel, int Lpad) {
    using elem = typename T::elem;
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wv;
    if (pair >= p.B * p.H) return;  // whole waves leave together; no barrier below
    float* qs = ssm + wv * (64 + Lpad);
    float* ps = qs + 64;
    const int b = pair / p.H, hd = pair - b * p.H, HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const int pos = p.sel_rows[b] - b * L;
    const int nk = p.causal ? pos + 1 : L;
    const elem* Kb = (const elem*)p.qkv + (size_t)b * L * ld + HD + hd * 64;
    const elem* Vb = Kb + HD;
    qs[lane] = (float)((const elem*)q_sel)[(size_t)b * HD + hd * 64 + lane];
    // ---- scores, key on the lane ----
    float m = -INFINITY;
    for (int k = lane; k < nk; k += 64) {
        const float s = dot64<T>(Kb + (size_t)k * ld, qs);
        ps[k] = s;
        m = fmaxf(m, s);
    }
    m = wave_max(m);
    float l = 0.f;
    for (int k = lane; k < nk; k += 64) {
        const float e = __builtin_amdgcn_exp2f((ps[k] - m) * SC_S);
        ps[k] = e;
        l += e;
    }
    l = wave_sum(l);
    // ---- o[dim] = sum_k p[k] V[k][dim], head dimension on the lane; 8 keys' loads in flight ----
    float o = 0.f;
    int k = 0;
    for (; k + 8 <= nk; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (float)Vb[(size_t)(k + u) * ld + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) o = __builtin_fmaf(ps[k + u], v[u], o);
    }
    for (; k < nk; ++k) o = __builtin_fmaf(ps[k], (float)Vb[(size_t)k * ld + lane], o);
    o *= 1.f / l;
    const size_t oo = (size_t)b * ld_out + hd * 64 + lane;
    const elem oe = (elem)o;
    ((elem*)out_sel)[oo] = oe;
    if (out_lo) ((elem*)out_lo)[oo] = (elem)(o - (float)oe);
    if (lane == 0) lse_sel[pair] = m * 0.125f + __logf(l);
}

// dynamic LDS: 4 waves x (128 + 2 Lpad) floats
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_single_kernel(AttnArgs p, const void* q_sel, const void* out_sel, int ld_out, const void* dout_sel,
                                                               const float* lse_sel, void* dq_sel, int Lpad) {
    using elem = typename T::elem;
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wv;
    if (pair >= p.B * p.H) return;
    float* qs = ssm + wv * (128 + 2 * Lpad);
    float* gs = qs + 64;    // dO row
    float* ps = gs + 64;    // p[k]
    float* dss = ps + Lpad;  // dS[k]
    const int b = pair / p.H, hd = pair - b * p.H, HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const int pos = p.sel_rows[b] - b * L;
    const int nk = p.causal ? pos + 1 : L;
    const elem* Kb = (const elem*)p.qkv + (size_t)b * L * ld + HD + hd * 64;
    const elem* Vb = Kb + HD;
    elem* dKb = (elem*)p.dqkv + (size_t)b * L * ld + HD + hd * 64;
    elem* dVb = dKb + HD;
    const float qd = (float)((const elem*)q_sel)[(size_t)b * HD + hd * 64 + lane];
    const float gd = (float)((const elem*)dout_sel)[(size_t)b * HD + hd * 64 + lane];
    const float od = (float)((const elem*)out_sel)[(size_t)b * ld_out + hd * 64 + lane];
    qs[lane] = qd;
    gs[lane] = gd;
    const float delta = wave_sum(gd * od);
    const float nlse = -lse_sel[pair] * LOG2E_S;
    // ---- p, dP, dS with the key on the lane ----
    for (int k = lane; k < nk; k += 64) {
        const float s = dot64<T>(Kb + (size_t)k * ld, qs);
        const float dp = dot64<T>(Vb + (size_t)k * ld, gs);
        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s, SC_S, nlse));
        ps[k] = pr;
        dss[k] = pr * (dp - delta);
    }
    // ---- head dimension on the lane: dq, and the dK / dV rows of every key ----
    float dq = 0.f;
    int k = 0;
    for (; k + 8 <= nk; k += 8) {
        float kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kv[u] = (float)Kb[(size_t)(k + u) * ld + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float ds = dss[k + u];
            dq = __builtin_fmaf(ds, kv[u], dq);
            dKb[(size_t)(k + u) * ld + lane] = (elem)(ds * qd * 0.125f);
            dVb[(size_t)(k + u) * ld + lane] = (elem)(ps[k + u] * gd);
        }
    }
    for (; k < nk; ++k) {
        const float ds = dss[k];
        dq = __builtin_fmaf(ds, (float)Kb[(size_t)k * ld + lane], dq);
        dKb[(size_t)k * ld + lane] = (elem)(ds * qd * 0.125f);
        dVb[(size_t)k * ld + lane] = (elem)(ps[k] * gd);
    }
    for (; k < L; ++k) {  // keys behind a causal limit: no gradient, but the rows are operands of the dX GEMM
        dKb[(size_t)k * ld + lane] = (elem)0.f;
        dVb[(size_t)k * ld + lane] = (elem)0.f;
    }
    ((elem*)dq_sel)[(size_t)b * HD + hd * 64 + lane] = (elem)(dq * 0.125f);
}

static int single_attrs() {  // the per-wave score arrays can exceed the default 64 KiB of dynamic LDS (L up to 4096)
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)attn_bwd_single_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)attn_bwd_single_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)attn_fwd_single_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)attn_fwd_single_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        pd.done[dev] = true;
    }
    return MUDPT_OK;
}

static int check_single(const AttnArgs& a) {
    ARG_CHECK(a.qkv && a.sel_rows && a.B > 0 && a.L > 0 && a.H > 0, "attention (single query): bad arguments B=%d L=%d H=%d", a.B, a.L, a.H);
    ARG_CHECK(a.L <= 4096, "attention (single query): L=%d exceeds the supported 4096 rows", a.L);
    ARG_CHECK((uintptr_t)a.qkv % 16 == 0 && (a.H * 64 * 3) % 8 == 0, "attention (single query): qkv must be 16-byte aligned");
    return MUDPT_OK;
}

// q_sel [B, H*64] (T); out_sel (T, row stride ld_out elements; out_lo optional low half), lse_sel [B, H]
int launch_attn_fwd_single(int dtype, const AttnArgs& a, const void* q_sel, void* out_sel, void* out_lo, int ld_out, float* lse_sel, hipStream_t s) {
    if (int e = check_single(a)) return e;
    ARG_CHECK(q_sel && out_sel && lse_sel && ld_out >= a.H * 64, "attention (single query) fwd: null operand");
    const int Lpad = (a.L + 63) & ~63, lds = 4 * (64 + Lpad) * 4, grid = (a.B * a.H + 3) / 4;
    if (int e = single_attrs()) return e;
    if (dtype == DT_BF16) hipLaunchKernelGGL(attn_fwd_single_kernel<BF16>, dim3(grid), dim3(256), lds, s, a, q_sel, out_sel, out_lo, ld_out, lse_sel, Lpad);
    else if (dtype == DT_F16) hipLaunchKernelGGL(attn_fwd_single_kernel<F16>, dim3(grid), dim3(256), lds, s, a, q_sel, out_sel, out_lo, ld_out, lse_sel, Lpad);
    else { set_error("attention: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// writes the k and v thirds of a.dqkv for every row and dq_sel [B, H*64]; the q third of a.dqkv is NOT written
int launch_attn_bwd_single(int dtype, const AttnArgs& a, const void* q_sel, const void* out_sel, int ld_out, const void* dout_sel, const float* lse_sel,
                           void* dq_sel, hipStream_t s) {
    if (int e = check_single(a)) return e;
    ARG_CHECK(q_sel && out_sel && dout_sel && lse_sel && dq_sel && a.dqkv && ld_out >= a.H * 64, "attention (single query) bwd: null operand");
    const int Lpad = (a.L + 63) & ~63, lds = 4 * (128 + 2 * Lpad) * 4, grid = (a.B * a.H + 3) / 4;
    ARG_CHECK(lds <= 160 * 1024, "attention (single query) bwd: L too large");
    if (int e = single_attrs()) return e;
    if (dtype == DT_BF16) hipLaunchKernelGGL(attn_bwd_single_kernel<BF16>, dim3(grid), dim3(256), lds, s, a, q_sel, out_sel, ld_out, dout_sel, lse_sel, dq_sel, Lpad);
    else if (dtype == DT_F16) hipLaunchKernelGGL(attn_bwd_single_kernel<F16>, dim3(grid), dim3(256), lds, s, a, q_sel, out_sel, ld_out, dout_sel, lse_sel, dq_sel, Lpad);
    else { set_error("attention: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

}  // namespace mudpt
