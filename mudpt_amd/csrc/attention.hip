// Attention forward / backward for gfx950, head dim 64, whole sequence of one (sequence, head) on chip.
//
// Replaces nn.MultiheadAttention's per-head softmax(Q K^T / 8 [+ causal mask]) V and its autograd
// (clip/model.py:271-273; mask clip/model.py:810-816).  Sequence lengths on this path are short and
// fixed (vision 197 + n_ctx = 201, text 77; SURVEY.md §5 "long-context: absent"), so one workgroup holds
// K and V of a (sequence, head) pair in LDS and every wave owns one 32-row block; the deep-prompt rows
// are ordinary rows of the same tile.  L is padded to NB * 32 inside the kernels only (masked keys,
// unstored query rows); activations in HBM are never padded.
//
// MFMA plan (v_mfma_f32_32x32x16, C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)):
//   fwd / dq pass: "query on the lane":  S^T = K Q^T  (A = K rows from LDS, B = Q fragment from HBM);
//     the whole score column of a query sits in one lane's registers (plus its lane ^ 32 partner), so the
//     softmax needs one cross-lane exchange; the converted accumulator is directly the B operand of
//     O^T = V^T P^T and dQ^T = K^T dS^T (A = transposed V / K image in LDS, read in the accumulator's
//     k order: element j of lane half h is row 16 s + 8 (j >> 2) + 4 h + (j & 3)).
//   dk/dv pass: "key on the lane": S = Q K^T, dP = dO V^T (B = K / V fragments from HBM, kept in
//     registers for the whole pass), dV^T += dO^T P, dK^T += Q^T dS with P / dS taken from the accumulator.
// dQ is produced by a second sweep with the roles swapped instead of atomics or a cross-wave reduction:
// bitwise reproducible, at the price of recomputing S and dP once (attention is < 10 % of step FLOPs).
#include "kernels.h"

namespace mudpt {

constexpr int KS = 72;  // row stride (elements) of a row-major [Lp][64] LDS image: 144 B, conflict-free b128 reads
constexpr float LOG2E = 1.4426950408889634f;

int attn_padded_len(int L) { return (L + 31) / 32 * 32; }

template <typename T>
struct Stager {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    // row-major image rows[key][KS] and/or transposed image cols[d][TS] of src[L][64] (row stride ld elements)
    template <bool ROWS, bool COLS>
    __device__ static inline void run(elem* rows, elem* cols, int TS, const elem* src, size_t ld, int L, int Lp, int tid,
                                      int nthreads) {
        for (int idx = tid; idx < Lp * 8; idx += nthreads) {
            const int key = idx >> 3, ch = idx & 7;
            vec8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (elem)0.f;
            if (key < L) v = *(const vec8*)(src + (size_t)key * ld + ch * 8);
            if constexpr (ROWS) *(vec8*)(rows + key * KS + ch * 8) = v;
            if constexpr (COLS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) cols[(ch * 8 + i) * TS + key] = v[i];
            }
        }
    }
};

// A operand (8 elements) for the k-step s of 32-key tile kt out of a transposed image img[d][TS]:
// elements 0..3 = keys kt*32 + 16 s + 4 h + 0..3, elements 4..7 = the same + 8.
template <typename T>
__device__ inline typename T::vec8 tr_frag(const typename T::elem* img, int TS, int d, int kt, int s, int h) {
    using vec4 = typename T::vec4;
    const typename T::elem* p = img + d * TS + kt * 32 + 16 * s + 4 * h;
    const vec4 lo = *(const vec4*)p;
    const vec4 hi = *(const vec4*)(p + 8);
    typename T::vec8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = lo[i]; r[4 + i] = hi[i]; }
    return r;
}

template <typename T>
__device__ inline typename T::vec8 pack8(const f32x16& x, int s) {
    typename T::vec8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (typename T::elem)x[8 * s + i];
    return r;
}

__device__ inline int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int NB, bool CAUSAL>
__global__ __launch_bounds__(NB * 64) void attn_fwd_kernel(AttnArgs p) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;
    constexpr int Lp = NB * 32, TS = Lp + 4, NT = NB * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem* Ks = (elem*)smem;          // [Lp][KS]
    elem* Vt = Ks + Lp * KS;         // [64][TS]

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;

    Stager<T>::template run<true, false>(Ks, nullptr, 0, base + HD, ld, L, Lp, tid, NT);
    Stager<T>::template run<false, true>(nullptr, Vt, TS, base + 2 * HD, ld, L, Lp, tid, NT);

    const int q = wave * 32 + c;
    vec8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[ks][i] = (elem)0.f;
        if (q < L) qf[ks] = *(const vec8*)(base + (size_t)q * ld + ks * 16 + h * 8);
    }
    __syncthreads();

    f32x16 S[NB];
#pragma unroll
    for (int kt = 0; kt < NB; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
        if (CAUSAL && kt > wave) continue;  // whole tile above the diagonal
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const vec8 kf = *(const vec8*)(Ks + (kt * 32 + c) * KS + ks * 16 + h * 8);
            S[kt] = T::mfma32(kf, qf[ks], S[kt]);
        }
    }
    // softmax over the keys of column q: registers of this lane and of lane ^ 32
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NB; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + acc_row(r, h);
            const bool dead = key >= L || (CAUSAL && key > q);
            S[kt][r] = dead ? -INFINITY : S[kt][r];
            m = fmaxf(m, S[kt][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float sc = 0.125f * LOG2E;  // 1 / sqrt(64) folded into the exponent
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NB; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f((S[kt][r] - m) * sc);
            S[kt][r] = e;
            l += e;
        }
    l += __shfl_xor(l, 32, 64);

    f32x16 O[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NB; ++kt) {
        if (CAUSAL && kt > wave) continue;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const vec8 pb = pack8<T>(S[kt], s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) O[dt] = T::mfma32(tr_frag<T>(Vt, TS, dt * 32 + c, kt, s, h), pb, O[dt]);
        }
    }
    if (q < L) {
        const float inv = 1.f / l;
        elem* o = (elem*)p.out + ((size_t)b * L + q) * HD + hd * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 v = {(elem)(O[dt][4 * g] * inv), (elem)(O[dt][4 * g + 1] * inv), (elem)(O[dt][4 * g + 2] * inv),
                          (elem)(O[dt][4 * g + 3] * inv)};
                *(vec4*)(o + dt * 32 + 8 * g + 4 * h) = v;
            }
    }
    if (h == 0 && p.lse) p.lse[((size_t)b * p.H + hd) * Lp + q] = q < L ? m * 0.125f + __logf(l) : 0.f;
}

// ------------------------------------------------------------------------------------------------
// backward, pass 1: dQ (and delta = rowsum(dO * O)); query on the lane
// ------------------------------------------------------------------------------------------------
template <typename T, int NB, bool CAUSAL>
__global__ __launch_bounds__(NB * 64) void attn_bwd_dq_kernel(AttnArgs p, const void* fwd_out) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;
    constexpr int Lp = NB * 32, TS = Lp + 4, NT = NB * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem* Ks = (elem*)smem;        // [Lp][KS]
    elem* Vs = Ks + Lp * KS;       // [Lp][KS]
    elem* Kt = Vs + Lp * KS;       // [64][TS]

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;

    Stager<T>::template run<true, true>(Ks, Kt, TS, base + HD, ld, L, Lp, tid, NT);
    Stager<T>::template run<true, false>(Vs, nullptr, 0, base + 2 * HD, ld, L, Lp, tid, NT);

    const int q = wave * 32 + c;
    vec8 qf[4], gf[4];
    float delta = 0.f;
    {
        const elem* dO = (const elem*)p.dout + ((size_t)b * L + q) * HD + hd * 64;
        const elem* Of = (const elem*)fwd_out + ((size_t)b * L + q) * HD + hd * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { qf[ks][i] = (elem)0.f; gf[ks][i] = (elem)0.f; }
            if (q < L) {
                qf[ks] = *(const vec8*)(base + (size_t)q * ld + ks * 16 + h * 8);
                gf[ks] = *(const vec8*)(dO + ks * 16 + h * 8);
                const vec8 of = *(const vec8*)(Of + ks * 16 + h * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) delta += (float)gf[ks][i] * (float)of[i];
            }
        }
    }
    delta += __shfl_xor(delta, 32, 64);
    const size_t stat = ((size_t)b * p.H + hd) * Lp + q;
    if (h == 0) p.delta[stat] = delta;
    const float nlse = -p.lse[stat] * LOG2E;
    __syncthreads();

    f32x16 dQ[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[dt][r] = 0.f;
    const float sc = 0.125f * LOG2E;
#pragma unroll 1
    for (int kt = 0; kt < NB; ++kt) {
        if (CAUSAL && kt > wave) continue;
        f32x16 S, dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const vec8 kf = *(const vec8*)(Ks + (kt * 32 + c) * KS + ks * 16 + h * 8);
            const vec8 vf = *(const vec8*)(Vs + (kt * 32 + c) * KS + ks * 16 + h * 8);
            S = T::mfma32(kf, qf[ks], S);
            dP = T::mfma32(vf, gf[ks], dP);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + acc_row(r, h);
            const bool dead = key >= L || q >= L || (CAUSAL && key > q);
            const float pr = dead ? 0.f : __builtin_amdgcn_exp2f(S[r] * sc + nlse);
            S[r] = pr * (dP[r] - delta) * 0.125f;  // dS^T (already times 1/sqrt(d))
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const vec8 db = pack8<T>(S, s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dQ[dt] = T::mfma32(tr_frag<T>(Kt, TS, dt * 32 + c, kt, s, h), db, dQ[dt]);
        }
    }
    if (q < L) {
        elem* o = (elem*)p.dqkv + ((size_t)b * L + q) * ld + hd * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 v = {(elem)dQ[dt][4 * g], (elem)dQ[dt][4 * g + 1], (elem)dQ[dt][4 * g + 2], (elem)dQ[dt][4 * g + 3]};
                *(vec4*)(o + dt * 32 + 8 * g + 4 * h) = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, pass 2: dK, dV; key on the lane
// ------------------------------------------------------------------------------------------------
template <typename T, int NB, bool CAUSAL>
__global__ __launch_bounds__(NB * 64) void attn_bwd_dkv_kernel(AttnArgs p) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;
    constexpr int Lp = NB * 32, TS = Lp + 4, NT = NB * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem* Qs = (elem*)smem;         // [Lp][KS]
    elem* Gs = Qs + Lp * KS;        // [Lp][KS]   dO rows
    elem* Qt = Gs + Lp * KS;        // [64][TS]
    elem* Gt = Qt + 64 * TS;        // [64][TS]
    float* lse_s = (float*)(Gt + 64 * TS);  // [Lp]  -lse * log2(e)
    float* del_s = lse_s + Lp;              // [Lp]

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;

    Stager<T>::template run<true, true>(Qs, Qt, TS, base, ld, L, Lp, tid, NT);
    Stager<T>::template run<true, true>(Gs, Gt, TS, dO, (size_t)HD, L, Lp, tid, NT);
    for (int i = tid; i < Lp; i += NT) {
        const size_t stat = ((size_t)b * p.H + hd) * Lp + i;
        lse_s[i] = -p.lse[stat] * LOG2E;
        del_s[i] = p.delta[stat];
    }
    const int key = wave * 32 + c;
    vec8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { kf[ks][i] = (elem)0.f; vf[ks][i] = (elem)0.f; }
        if (key < L) {
            kf[ks] = *(const vec8*)(base + HD + (size_t)key * ld + ks * 16 + h * 8);
            vf[ks] = *(const vec8*)(base + 2 * HD + (size_t)key * ld + ks * 16 + h * 8);
        }
    }
    __syncthreads();

    f32x16 dK[2], dV[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dK[dt][r] = 0.f; dV[dt][r] = 0.f; }
    const float sc = 0.125f * LOG2E;
#pragma unroll 1
    for (int qt = 0; qt < NB; ++qt) {
        if (CAUSAL && qt < wave) continue;  // every query of the tile precedes every key of this block
        f32x16 S, dP, nl;  // row constants of the tile's 16 query rows: -lse * log2(e), and -delta as dP's initial value
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *(const f32x4*)(lse_s + qt * 32 + 8 * g + 4 * h);
            const f32x4 d4 = *(const f32x4*)(del_s + qt * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) { S[4 * g + i] = 0.f; dP[4 * g + i] = -d4[i]; nl[4 * g + i] = l4[i]; }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const vec8 qa = *(const vec8*)(Qs + (qt * 32 + c) * KS + ks * 16 + h * 8);
            const vec8 ga = *(const vec8*)(Gs + (qt * 32 + c) * KS + ks * 16 + h * 8);
            S = T::mfma32(qa, kf[ks], S);
            dP = T::mfma32(ga, vf[ks], dP);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = qt * 32 + acc_row(r, h);
            const bool dead = key >= L || q >= L || (CAUSAL && key > q);
            const float pr = dead ? 0.f : __builtin_amdgcn_exp2f(S[r] * sc + nl[r]);
            S[r] = pr;
            dP[r] = pr * dP[r] * 0.125f;  // dS (delta already subtracted through the accumulator init)
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const vec8 pb = pack8<T>(S, s);
            const vec8 db = pack8<T>(dP, s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dV[dt] = T::mfma32(tr_frag<T>(Gt, TS, dt * 32 + c, qt, s, h), pb, dV[dt]);
                dK[dt] = T::mfma32(tr_frag<T>(Qt, TS, dt * 32 + c, qt, s, h), db, dK[dt]);
            }
        }
    }
    if (key < L) {
        elem* ok = (elem*)p.dqkv + ((size_t)b * L + key) * ld + HD + hd * 64;
        elem* ov = ok + HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vec4 a = {(elem)dK[dt][4 * g], (elem)dK[dt][4 * g + 1], (elem)dK[dt][4 * g + 2], (elem)dK[dt][4 * g + 3]};
                vec4 v = {(elem)dV[dt][4 * g], (elem)dV[dt][4 * g + 1], (elem)dV[dt][4 * g + 2], (elem)dV[dt][4 * g + 3]};
                *(vec4*)(ok + dt * 32 + 8 * g + 4 * h) = a;
                *(vec4*)(ov + dt * 32 + 8 * g + 4 * h) = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int check(const AttnArgs& a, bool bwd) {
    ARG_CHECK(a.qkv && a.B > 0 && a.L > 0 && a.H > 0, "attention: bad arguments B=%d L=%d H=%d", a.B, a.L, a.H);
    ARG_CHECK(a.L <= 224, "attention: L=%d exceeds the on-chip limit of 224 rows", a.L);
    ARG_CHECK((uintptr_t)a.qkv % 16 == 0, "attention: qkv must be 16-byte aligned");
    if (!bwd) ARG_CHECK(a.out && (uintptr_t)a.out % 16 == 0, "attention: null/unaligned out");
    if (bwd) ARG_CHECK(a.out && a.dout && a.dqkv && a.lse && a.delta, "attention bwd: null operand");
    return MUDPT_OK;
}

template <typename K>
static int set_lds(K kern, int bytes) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return MUDPT_OK;
}

template <typename T, int NB, bool CAUSAL>
static int fwd_cfg(const AttnArgs& a, hipStream_t s) {
    constexpr int Lp = NB * 32, TS = Lp + 4;
    constexpr int lds = (Lp * KS + 64 * TS) * 2;
    auto kern = attn_fwd_kernel<T, NB, CAUSAL>;
    static bool once = false;
    if (!once) { if (int e = set_lds(kern, lds)) return e; once = true; }
    hipLaunchKernelGGL(kern, dim3(a.B * a.H), dim3(NB * 64), lds, s, a);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

template <typename T, int NB, bool CAUSAL>
static int bwd_cfg(const AttnArgs& a, hipStream_t s) {
    constexpr int Lp = NB * 32, TS = Lp + 4;
    constexpr int lds1 = (2 * Lp * KS + 64 * TS) * 2;
    constexpr int lds2 = (2 * Lp * KS + 2 * 64 * TS) * 2 + 2 * Lp * 4;
    auto k1 = attn_bwd_dq_kernel<T, NB, CAUSAL>;
    auto k2 = attn_bwd_dkv_kernel<T, NB, CAUSAL>;
    static bool once = false;
    if (!once) {
        if (int e = set_lds(k1, lds1)) return e;
        if (int e = set_lds(k2, lds2)) return e;
        once = true;
    }
    hipLaunchKernelGGL(k1, dim3(a.B * a.H), dim3(NB * 64), lds1, s, a, (const void*)a.out);
    hipLaunchKernelGGL(k2, dim3(a.B * a.H), dim3(NB * 64), lds2, s, a);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

template <typename T, bool BWD>
static int dispatch(const AttnArgs& a, hipStream_t s) {
    const int nb = attn_padded_len(a.L) / 32;
#define MUDPT_ATTN_CASE(N)                                                                     \
    case N:                                                                                    \
        if (a.causal) return BWD ? bwd_cfg<T, N, true>(a, s) : fwd_cfg<T, N, true>(a, s);      \
        return BWD ? bwd_cfg<T, N, false>(a, s) : fwd_cfg<T, N, false>(a, s);
    switch (nb) {
        MUDPT_ATTN_CASE(1)
        MUDPT_ATTN_CASE(2)
        MUDPT_ATTN_CASE(3)
        MUDPT_ATTN_CASE(4)
        MUDPT_ATTN_CASE(5)
        MUDPT_ATTN_CASE(6)
        MUDPT_ATTN_CASE(7)
    }
#undef MUDPT_ATTN_CASE
    set_error("attention: unsupported padded length %d", nb * 32);
    return MUDPT_ERR_ARG;
}

int launch_attn_fwd(int dtype, const AttnArgs& a, hipStream_t s) {
    if (int e = check(a, false)) return e;
    if (dtype == DT_BF16) return dispatch<BF16, false>(a, s);
    if (dtype == DT_F16) return dispatch<F16, false>(a, s);
    set_error("attention: unknown dtype %d", dtype);
    return MUDPT_ERR_ARG;
}

int launch_attn_bwd(int dtype, const AttnArgs& a, hipStream_t s) {
    if (int e = check(a, true)) return e;
    if (dtype == DT_BF16) return dispatch<BF16, true>(a, s);
    if (dtype == DT_F16) return dispatch<F16, true>(a, s);
    set_error("attention: unknown dtype %d", dtype);
    return MUDPT_ERR_ARG;
}

}  // namespace mudpt
