// Exact-fp32 attention forward for gfx950: the attention core of the "fp32" (exact) mode, what PREC = "fp32" selects.
//
// The reference's CPU path is fp32 end to end (clip/clip.py:142-143 floats the model; nn.MultiheadAttention at clip/model.py:271-273),
// and its logits are exp(logit_scale) = 100 times a cosine with pretrained weights: north_star's 1e-3 logit bound is then 1e-5 on the
// cosine, which q, k, v and P rounded to 11-bit fp16 do not hold (measured: 3e-4 of relative text-feature error from those four
// roundings alone).  This kernel keeps q, k, v, the scores, P and the output in fp32 and contracts on the matrix cores with
// v_mfma_f32_16x16x4_f32 (a k-ordered fp32 fma chain, bit for bit; 1/16 of the fp16 MFMA rate: ~0.25 ms per ViT-B/16 layer at B 256).
//
//   in : qkv32 [B, L, 3 H 64] fp32 (q | k | v thirds, heads contiguous inside a third): the in_proj GEMM's fp32 epilogue
//   out: out [B, L, ld_out] fp16 = hi(O) and out_lo = fp16(O - hi): the split operand of the out_proj GEMM; lse [B, H, Lp];
//        qkv_lp [B, L, 3 H 64] fp16: the copy the (fp16-operand) backward kernels read
//
// One workgroup = 8 waves = 128 queries of one (sequence, head); a wave owns 16 queries.  Keys / values stream through LDS in tiles of
// 32 keys (fp32 rows of 256 bytes, double buffered, staged through registers).  Operand maps of the 16x16x4 f32 MFMA: lane l holds
// A[l & 15][l >> 4], B[l >> 4][l & 15], D[4 (l >> 4) + r][l & 15].  With n = l & 15, g = l >> 4:
//   S^T[key][query] = K . Q^T : A = K[key n][d = 16 g + s], B = Q[query n][d = 16 g + s], s = 0..15 (any partition of d over the
//       16 steps x 4 k-slots works as long as both operands agree); the lane ends up with S[query n][keys 4 g .. 4 g + 3].
//   O^T[d][query] = V^T . P^T : B = P[query n][key 4 g + i] -- the registers the lane already holds, no data movement --,
//       A = V[key 4 g + i][d = 4 n + dt] for the accumulator dt = 0..3; the lane ends up with O[query n][16 g .. 16 g + 15].
// LDS reads are ds_read_b128: K rows with chunk c of row n stored in slot c ^ kswz(n) (conflict-free for the lane groups of a
// ds_read_b128), V rows plain (the 16 lanes of a group read 16 different chunks).
#include "kernels.h"

namespace mudpt {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4e;
constexpr int XKT = 32;  // keys per staged tile
constexpr int XW = 8, XQ = XW * 16;  // waves / queries per workgroup (round 3: 4 / 64 read K and V of a pair four times over: 1.4 GB per ViT-B/16 layer at B 256)

// slot swizzle of the K image: within a ds_read_b128 lane group {n in 0-3, 12-15 with k-slot g} + {n in 4-11 with k-slot g ^ 1}
// the slots (4 g + j) ^ kswz(n) are 16 distinct chunks of the 256-byte bank row
__device__ inline int kswz(int n) { return n ^ ((((n >> 2) ^ (n >> 3)) & 1) << 2); }

template <bool CAUSAL>
__global__ __launch_bounds__(XW * 64) void attn_fwd_exact_kernel(AttnArgs p, int nchunks) {
    __shared__ __attribute__((aligned(16))) float sm[2][2][XKT * 64];  // [buffer][K | V][key][64]: 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H, qc = blockIdx.y;
    const int L = p.L, HD = p.H * 64, ld = 3 * HD;
    const int Lp = (L + 31) / 32 * 32;
    const int ld_out = p.ld_out ? p.ld_out : HD;
    const float* base = p.qkv32 + (size_t)b * L * ld + h * 64;
    _Float16* lp = (_Float16*)p.qkv_lp;
    if (lp) lp += (size_t)b * L * ld + h * 64;

    const int q0 = qc * XQ + w * 16, qrow = q0 + n;
    const bool qvalid = qrow < L;
    // the padded tail [L, Lp) of the log-sum-exp row reads as 0, like the fp16 forward kernels leave it: the whole-pair backward kernels
    // load those rows unguarded (their dQ is never stored, but the invariant "no uninitialised value enters the arithmetic" holds).
    // Lp <= nchunks * XQ (XQ is a multiple of 32), so every padded row belongs to some wave of the grid, live or not.
    if (g == 0 && p.lse && !qvalid && qrow < Lp) p.lse[((size_t)b * p.H + h) * Lp + qrow] = 0.f;
    // Q fragment: Q[qrow][16 g + s]
    float qf[16];
    {
        const f32x4* src = (const f32x4*)(base + (size_t)(qvalid ? qrow : 0) * ld + 16 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = qvalid ? src[j] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[4 * j + e] = v[e];
        }
        if (lp && qvalid) {
            f16x8 a, c;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)qf[e]; c[e] = (_Float16)qf[8 + e]; }
            f16x8* dst = (f16x8*)(lp + (size_t)qrow * ld + 16 * g);
            dst[0] = a; dst[1] = c;
        }
    }

    // keys this workgroup needs: all (non-causal) / up to its last query (causal)
    const int klast = CAUSAL ? (qc * XQ + XQ - 1 < L - 1 ? qc * XQ + XQ - 1 : L - 1) : L - 1;
    const int nt = klast / XKT + 1;
    const bool copy_kv = lp && qc == nchunks - 1;  // the last query chunk stages every key of the sequence: it writes the fp16 copies of k, v

    // staging: thread -> SR chunks of K and of V per tile (chunk e = tid + 64 XW r: key e >> 4, 16-byte chunk e & 15)
    constexpr int SR = XKT * 16 / (XW * 64);
    static_assert(SR >= 1 && SR * XW * 64 == XKT * 16, "a tile's 16-byte chunks must split evenly over the threads");
    f32x4 stg[SR][2];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const int e = tid + XW * 64 * r, key = t * XKT + (e >> 4), c = e & 15;
            const bool ok = key < L;
            const float* src = base + (size_t)(ok ? key : 0) * ld + 4 * c;
            stg[r][0] = ok ? *(const f32x4*)(src + HD) : f32x4{0.f, 0.f, 0.f, 0.f};
            stg[r][1] = ok ? *(const f32x4*)(src + 2 * HD) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_write = [&](int t, int buf) {
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const int e = tid + XW * 64 * r, kl = e >> 4, c = e & 15, key = t * XKT + kl;
            *(f32x4*)&sm[buf][0][kl * 64 + ((c ^ kswz(kl & 15)) << 2)] = stg[r][0];
            *(f32x4*)&sm[buf][1][kl * 64 + (c << 2)] = stg[r][1];
            if (copy_kv && key < L) {
                f16x4 kk = {(_Float16)stg[r][0][0], (_Float16)stg[r][0][1], (_Float16)stg[r][0][2], (_Float16)stg[r][0][3]};
                f16x4 vv = {(_Float16)stg[r][1][0], (_Float16)stg[r][1][1], (_Float16)stg[r][1][2], (_Float16)stg[r][1][3]};
                *(f16x4*)(lp + (size_t)key * ld + HD + 4 * c) = kk;
                *(f16x4*)(lp + (size_t)key * ld + 2 * HD + 4 * c) = vv;
            }
        }
    };

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, lsum = 0.f;
    const float sc = 0.125f * 1.4426950408889634f;  // 1 / sqrt(64) (nn.MultiheadAttention scales q by head_dim^-1/2) in the log2 domain
    const bool wave_live = q0 < L;

    stage_load(0);
    stage_write(0, 0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1, k0 = t * XKT;
        if (t + 1 < nt) stage_load(t + 1);
        if (wave_live && (!CAUSAL || k0 <= q0 + 15)) {
            const float* Ks = sm[buf][0];
            const float* Vs = sm[buf][1];
            f32x4 st[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x4 kf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) kf[j] = *(const f32x4*)&Ks[(sub * 16 + n) * 64 + (((4 * g + j) ^ kswz(n)) << 2)];
                st[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 16; ++s) st[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s >> 2][s & 3], qf[s], st[sub], 0, 0, 0);
            }
            // st[sub][i] = S[query n][key k0 + 16 sub + 4 g + i]
            float tmax = -1e30f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = k0 + 16 * sub + 4 * g + i;
                    const bool dead = key >= L || (CAUSAL && key > qrow);
                    st[sub][i] = dead ? -1e30f : st[sub][i] * sc;
                    tmax = fmaxf(tmax, st[sub][i]);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    st[sub][i] = __builtin_amdgcn_exp2f(st[sub][i] - m_new);
                    psum += st[sub][i];
                }
            lsum = lsum * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 vf = *(const f32x4*)&Vs[(sub * 16 + 4 * g + i) * 64 + 4 * n];
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dt], st[sub][i], o[dt], 0, 0, 0);
                }
        }
        if (t + 1 < nt) stage_write(t + 1, buf ^ 1);
        __syncthreads();
    }
    if (!wave_live) return;
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (!qvalid) return;
    const float inv = 1.0f / lsum;
    if (g == 0 && p.lse) p.lse[((size_t)b * p.H + h) * Lp + qrow] = (m + __builtin_amdgcn_logf(lsum)) * 0.6931471805599453f;
    // o[dt][r] = O[qrow][16 g + 4 r + dt]
    f16x8 hi[2], lo[2];
    float rem[16];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int c = 4 * r + dt;
            _Float16 hv;
            rem[c] = split_rem(o[dt][r] * inv, hv);
            hi[c >> 3][c & 7] = hv;
            lo[c >> 3][c & 7] = (_Float16)rem[c];
        }
    const size_t off = ((size_t)b * L + qrow) * ld_out + h * 64 + 16 * g;
    f16x8* dh = (f16x8*)((_Float16*)p.out + off);
    dh[0] = hi[0]; dh[1] = hi[1];
    if (p.out_lo) {  // the low half of the split operand (common.h LoMode): fp16, or e4m3 bytes at the same row stride in bytes
        if (p.lo_mode == LO_F8) {
            u32x4e w;
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = pack_lo8(rem[4 * c], rem[4 * c + 1], rem[4 * c + 2], rem[4 * c + 3]);
            *(u32x4e*)((char*)p.out_lo + ((size_t)b * L + qrow) * ld_out * 2 + h * 64 + 16 * g) = w;
        } else {
            f16x8* dl = (f16x8*)((_Float16*)p.out_lo + off);
            dl[0] = lo[0]; dl[1] = lo[1];
        }
    }
}

}  // namespace

int launch_attn_fwd_exact(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    ARG_CHECK(a.qkv32 && a.out && a.B > 0 && a.L > 0 && a.H > 0, "attention (exact): bad arguments");
    ARG_CHECK(a.L <= 4096, "attention (exact): L=%d exceeds 4096", a.L);
    const int ld_out = a.ld_out ? a.ld_out : a.H * 64;
    ARG_CHECK(ld_out % 8 == 0 && ((uintptr_t)a.out % 16 == 0) && ((uintptr_t)a.out_lo % 16 == 0) && ((uintptr_t)a.qkv32 % 16 == 0) && ((uintptr_t)a.qkv_lp % 16 == 0),
              "attention (exact): operands must be 16-byte aligned");
    ARG_CHECK((size_t)a.B * a.H < 0x7fffffffull, "attention (exact): too many (sequence, head) pairs");
    const int nchunks = (a.L + XQ - 1) / XQ;
    const dim3 grid((unsigned)(a.B * a.H), (unsigned)nchunks);
    if (a.causal) MUDPT_LAUNCH(attn_fwd_exact_kernel<true>, grid, dim3(XW * 64), 0, s, prof, a, nchunks);
    else MUDPT_LAUNCH(attn_fwd_exact_kernel<false>, grid, dim3(XW * 64), 0, s, prof, a, nchunks);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

}  // namespace mudpt
