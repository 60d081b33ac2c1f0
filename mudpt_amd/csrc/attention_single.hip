// Single-query attention for the LAST block of a tower (gfx950).
//
// Only one row per sequence of the last block's output is ever used -- the CLS token (clip/model.py:549) or the EOT token
// (trainers/mudpt.py:154) -- so that block needs ONE query per sequence against all (causal: the first pos + 1) keys.  The general
// kernels (attention.hip) would compute all L queries and, in the backward, sweep a dO that is zero except on one row.  Here one
// wave handles one (sequence, head) pair, 8 lanes per key row (each lane 8 head dimensions = one 16-byte access, so a wave instruction
// moves 8 whole 128-byte rows):
//   forward : s[k] = q . K[k] (8 multiply-adds per lane + an exchange inside the row's 8 lanes), softmax over the wave (scores in LDS),
//             o = sum_k p[k] V[k] accumulated per lane over its row group's keys and summed over the 8 groups at the end;
//   backward: ONE pass over K and V: p, dP = dO . V[k], dS = p (dP - delta) per key; dq = sum_k dS[k] K[k] accumulates in registers,
//             dK[k] = dS[k] q and dV[k] = p[k] dO are stored as they are made, for EVERY key row (zeros beyond a causal limit), so
//             the dX GEMM that follows reads a fully defined [M, 2 d] operand.
// q / o / dO / dq are compact [nseq, H * 64] arrays (one row per sequence); K and V are read from the packed qkv buffer
// [nseq, L, 3 * H * 64] whose q third is never touched.  HBM-bound: K and V read once, dK and dV written once.  Sums run in a fixed
// order: bitwise reproducible.  (The first version put a key on every lane -- 64 rows per instruction, 16 bytes of each -- and read
// K twice: 148 us for the vision tower's last block against ~55 us of HBM time.)
#include "kernels.h"

namespace mudpt {

constexpr float LOG2E_S = 1.4426950408889634f;
constexpr float SC_S = 0.125f * LOG2E_S;

// Row layout of both kernels: 8 lanes per key row, each lane owns 8 consecutive head dimensions (one 16-byte access), so ONE wave
// instruction moves 8 whole 128-byte rows; a dot product is 8 multiply-adds per lane and a 3-step exchange inside the 8 lanes of a row.
__device__ inline float row_sum8(float v) {  // sum over the 8 lanes that share lane >> 3 (all of them receive it)
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ inline float group_sum8(float v) {  // sum over the 8 row groups: lanes that share lane & 7
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
template <typename T>
__device__ inline void load8(const typename T::elem* p, bool valid, float (&out)[8]) {
    typename T::vec8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (typename T::elem)0.f;
    if (valid) v = *(const typename T::vec8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}

constexpr int SQ_UNROLL = 4;  // row groups (of 8 keys) whose loads are in flight together

// grid: ceil(nseq * H / 4) workgroups of 4 waves; dynamic LDS: 4 waves x Lpad floats (the scores of a pair)
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_single_kernel(AttnArgs p, const void* q_sel, void* out_sel, void* out_lo, int ld_out, float* lse_sel, int Lpad) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ch = lane & 7, rg = lane >> 3;
    const int pair = blockIdx.x * 4 + wv;
    if (pair >= p.B * p.H) return;  // whole waves leave together; no barrier below
    float* ps = ssm + wv * Lpad;
    const int b = pair / p.H, hd = pair - b * p.H, HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const int pos = p.sel_rows[b] - b * L;
    const int nk = p.causal ? pos + 1 : L;
    const elem* Kb = (const elem*)p.qkv + (size_t)b * L * ld + HD + hd * 64 + ch * 8;
    const elem* Vb = Kb + HD;
    float qf[8];
    load8<T>((const elem*)q_sel + (size_t)b * HD + hd * 64 + ch * 8, true, qf);
    // ---- scores: 8 keys per step, SQ_UNROLL steps' loads in flight ----
    float m = -INFINITY;
    for (int k0 = 0; k0 < nk; k0 += 8 * SQ_UNROLL) {
        float kv[SQ_UNROLL][8];
#pragma unroll
        for (int u = 0; u < SQ_UNROLL; ++u) load8<T>(Kb + (size_t)(k0 + 8 * u + rg) * ld, k0 + 8 * u + rg < nk, kv[u]);
#pragma unroll
        for (int u = 0; u < SQ_UNROLL; ++u) {
            const int k = k0 + 8 * u + rg;
            float sc = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sc = __builtin_fmaf(kv[u][j], qf[j], sc);
            sc = row_sum8(sc);
            if (k < nk) {
                if (ch == 0) ps[k] = sc;
                m = fmaxf(m, sc);
            }
        }
    }
    m = wave_max(m);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's score stores have landed (a wave's LDS operations complete in order)
    // ---- o = sum_k p[k] V[k]: each lane accumulates its 8 dimensions over its row group's keys; the 8 groups are summed at the end ----
    float l = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    for (int k0 = 0; k0 < nk; k0 += 8 * SQ_UNROLL) {
        float vv[SQ_UNROLL][8];
#pragma unroll
        for (int u = 0; u < SQ_UNROLL; ++u) load8<T>(Vb + (size_t)(k0 + 8 * u + rg) * ld, k0 + 8 * u + rg < nk, vv[u]);
#pragma unroll
        for (int u = 0; u < SQ_UNROLL; ++u) {
            const int k = k0 + 8 * u + rg;
            const float e = k < nk ? __builtin_amdgcn_exp2f((ps[k] - m) * SC_S) : 0.f;
            l += e;  // every lane of a row adds the same e: the wave sum below counts each key 8 times
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(e, vv[u][j], o[j]);
        }
    }
    l = wave_sum(l) * 0.125f;
    const float inv = 1.f / l;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = group_sum8(o[j]) * inv;
    if (rg == 0) {
        const size_t oo = (size_t)b * ld_out + hd * 64 + ch * 8;
        vec8 hi, lo;
        float rem[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { elem hv; rem[j] = split_rem(o[j], hv); hi[j] = hv; lo[j] = (elem)rem[j]; }
        *(vec8*)((elem*)out_sel + oo) = hi;
        if (out_lo) {  // the low half of a split operand (common.h LoMode)
            if (p.lo_mode == LO_F8) {
                uint32_t* d8 = (uint32_t*)((char*)out_lo + (size_t)b * ld_out * 2 + hd * 64 + ch * 8);
                d8[0] = pack_lo8(rem[0], rem[1], rem[2], rem[3]);
                d8[1] = pack_lo8(rem[4], rem[5], rem[6], rem[7]);
            } else {
                *(vec8*)((elem*)out_lo + oo) = lo;
            }
        }
    }
    if (lane == 0) lse_sel[pair] = m * 0.125f + __logf(l);
}

// ONE pass over K and V: p, dP, dS of 8 keys per step; dq accumulates in registers, the dK / dV rows are stored as they are made
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_single_kernel(AttnArgs p, const void* q_sel, const void* out_sel, int ld_out, const void* dout_sel,
                                                               const float* lse_sel, void* dq_sel) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ch = lane & 7, rg = lane >> 3;
    const int pair = blockIdx.x * 4 + wv;
    if (pair >= p.B * p.H) return;
    const int b = pair / p.H, hd = pair - b * p.H, HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const int pos = p.sel_rows[b] - b * L;
    const int nk = p.causal ? pos + 1 : L;
    const elem* Kb = (const elem*)p.qkv + (size_t)b * L * ld + HD + hd * 64 + ch * 8;
    const elem* Vb = Kb + HD;
    elem* dKb = (elem*)p.dqkv + (size_t)b * L * ld + HD + hd * 64 + ch * 8;
    elem* dVb = dKb + HD;
    float qf[8], gf[8], of[8];
    load8<T>((const elem*)q_sel + (size_t)b * HD + hd * 64 + ch * 8, true, qf);
    load8<T>((const elem*)dout_sel + (size_t)b * HD + hd * 64 + ch * 8, true, gf);
    load8<T>((const elem*)out_sel + (size_t)b * ld_out + hd * 64 + ch * 8, true, of);
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) delta = __builtin_fmaf(gf[j], of[j], delta);
    delta = row_sum8(delta);
    const float nlse = -lse_sel[pair] * LOG2E_S;
    float dq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dq[j] = 0.f;
    for (int k0 = 0; k0 < nk; k0 += 8 * SQ_UNROLL) {
        float kv[SQ_UNROLL][8], vv[SQ_UNROLL][8];
#pragma unroll
        for (int u = 0; u < SQ_UNROLL; ++u) {
            const int k = k0 + 8 * u + rg;
            load8<T>(Kb + (size_t)k * ld, k < nk, kv[u]);
            load8<T>(Vb + (size_t)k * ld, k < nk, vv[u]);
        }
#pragma unroll
        for (int u = 0; u < SQ_UNROLL; ++u) {
            const int k = k0 + 8 * u + rg;
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc = __builtin_fmaf(kv[u][j], qf[j], sc); dp = __builtin_fmaf(vv[u][j], gf[j], dp); }
            sc = row_sum8(sc);
            dp = row_sum8(dp);
            const float pr = k < nk ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc, SC_S, nlse)) : 0.f;
            const float ds = pr * (dp - delta);
            vec8 dk, dv;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dq[j] = __builtin_fmaf(ds, kv[u][j], dq[j]);
                dk[j] = (elem)(ds * qf[j] * 0.125f);
                dv[j] = (elem)(pr * gf[j]);
            }
            if (k < nk) {
                *(vec8*)(dKb + (size_t)k * ld) = dk;
                *(vec8*)(dVb + (size_t)k * ld) = dv;
            }
        }
    }
    {  // keys behind a causal limit: no gradient, but the rows are operands of the dX GEMM
        vec8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (elem)0.f;
        for (int k = nk + rg; k < L; k += 8) {
            *(vec8*)(dKb + (size_t)k * ld) = z;
            *(vec8*)(dVb + (size_t)k * ld) = z;
        }
    }
    vec8 dqo;
#pragma unroll
    for (int j = 0; j < 8; ++j) dqo[j] = (elem)(group_sum8(dq[j]) * 0.125f);
    if (rg == 0) *(vec8*)((elem*)dq_sel + (size_t)b * HD + hd * 64 + ch * 8) = dqo;
}

static int single_attrs() {  // the per-wave score arrays can exceed the default 64 KiB of dynamic LDS (L up to 4096)
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
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
    ARG_CHECK((uintptr_t)q_sel % 16 == 0 && (uintptr_t)out_sel % 16 == 0 && (uintptr_t)out_lo % 16 == 0 && ld_out % 8 == 0, "attention (single query) fwd: operands must be 16-byte aligned");
    const int Lpad = (a.L + 63) & ~63, lds = 4 * Lpad * 4, grid = (a.B * a.H + 3) / 4;
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
    ARG_CHECK((uintptr_t)a.dqkv % 16 == 0 && (uintptr_t)q_sel % 16 == 0 && (uintptr_t)dout_sel % 16 == 0 && (uintptr_t)out_sel % 16 == 0 && (uintptr_t)dq_sel % 16 == 0 && ld_out % 8 == 0,
              "attention (single query) bwd: operands must be 16-byte aligned");
    const int grid = (a.B * a.H + 3) / 4;
    if (dtype == DT_BF16) hipLaunchKernelGGL(attn_bwd_single_kernel<BF16>, dim3(grid), dim3(256), 0, s, a, q_sel, out_sel, ld_out, dout_sel, lse_sel, dq_sel);
    else if (dtype == DT_F16) hipLaunchKernelGGL(attn_bwd_single_kernel<F16>, dim3(grid), dim3(256), 0, s, a, q_sel, out_sel, ld_out, dout_sel, lse_sel, dq_sel);
    else { set_error("attention: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

}  // namespace mudpt
