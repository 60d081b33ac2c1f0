// Attention for 224 < L <= ~620, head dim 64: the "resident" kernels.  One workgroup per (sequence, head) pair holds BOTH streamed operands
// of the pair in LDS (2 x Lr x 128 bytes, Lr = L rounded up to 32: 152 KB of the CU's 160 KB at ViT-L/14@336's L = 581), staged once, and its
// waves then run with no barrier at all.  Same mathematics and interfaces as the staged kernels of attention.hip (which remain for longer
// sequences, the window / single-row forms and A/B runs); replaces nn.MultiheadAttention's per-head softmax(Q K^T / 8 [+ mask]) V and its
// autograd (clip/model.py:271-273; mask clip/model.py:810-816).
//
// Built with -fno-slp-vectorize (mudpt_amd/build.py): under plain -O3 the SLP pass pairs the per-element multiplies of the softmax epilogue
// into v_pk_mul_f32 on misaligned register pairs and then repairs the pairing with v_mov / v_perm / v_alignbit (MI355X_MICROARCH.md's
// constants table prices packed f32 VALU beside MFMAs as an anti-lever).
#include "kernels.h"

namespace mudpt {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float SC = 0.125f * LOG2E;  // 1 / sqrt(64) folded into the base-2 exponent
typedef __attribute__((ext_vector_type(4))) short s16x4;
using lds_s16x4 = __attribute__((address_space(3))) s16x4;
__device__ inline int attn_padded_len_dev(int L) { return (L + 31) / 32 * 32; }

// Long-sequence forward, round 3 second form ("resident"): the pair's whole K and V sit in LDS (2 x Lr x 128 bytes, Lr = L rounded up to 32:
// 152 KB of the CU's 160 KB at L = 581), staged ONCE per (image, head) pair instead of once per 128-query workgroup, and the waves of the
// one workgroup per CU then stream over them with no barrier at all.  Timing ablations of the staged kernels showed why: with the per-stage
// global -> register -> LDS copy removed the 16-query-block core ran 18 % faster and the 32-query core below 49 % faster.
// Each wave owns 32 queries on v_mfma_f32_32x32x16: S^T[32 keys][32 queries] = K Q^T puts ONE query on a lane (column l & 31) with 16 of the
// tile's 32 keys in its registers (rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)); the other 16 sit in lane l ^ 32, so the row maximum / sum need
// one cross-lane exchange per 64 keys instead of two per 16-query block, a 32 x 32 tile costs half the MFMA and LDS-read instructions of
// two 16 x 16 x 32 blocks, and registers 8 s .. 8 s + 7 of a tile, packed to T, ARE the B operand of O^T = V^T P^T for the 16-key slab s
// (the k order inside a slab is rows 16 s + 8 (j >> 2) + 4 h + (j & 3): the A operand follows it by reading V with two hardware-transposed
// 4 x 16 block reads at rows 16 s + 4 h and 16 s + 8 + 4 h).  Query blocks are dealt round-robin to the RES_W-or-fewer waves.
// LDS images are [Lr keys][64] with 16-byte chunk c of row r in slot c ^ fK(r) / c ^ fV(r): fK(r) = (r >> 1) & 7 is conflict-free for the
// 32-row ds_read_b128 fragments, fV(r) = bits (0, 1, 2) of r sent to bits (0, 2, 1) for the transposed 64-bit block reads (brute-force
// checked against the bank maps of MI355X_MICROARCH.md's LDS table).
__device__ inline int flash_fk(int r) { return (r >> 1) & 7; }
__device__ inline int flash_fv(int r) { return (r & 1) | ((r & 2) << 1) | ((r & 4) >> 1); }
constexpr int RES_W_MAX = 16;               // launch bound (128 VGPRs: the build bounded at 12 waves / 168 VGPRs measured 17 % slower at 12 waves)
constexpr int RES_W = 12;                   // waves launched at most: 3 per SIMD; 19 query blocks (L = 581) then load the SIMDs 5 / 5 / 5 / 4
constexpr int RES_W_BWD = 8;                // dK/dV kernel: 2 waves per SIMD with 256 VGPRs each (at 168 the compiler serialises every LDS read
                                            // behind its MFMA); 19 blocks load the SIMDs 5 / 5 / 5 / 4 as with 12 waves
constexpr float RES_DEFER = 6.f;            // deferred rescale threshold, base-2 exponent units
constexpr int RES_LDS_MAX = 160 * 1024;     // gfx950: one workgroup may own the CU's whole LDS

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(RES_W_MAX * 64) void attn_fwd_resident_kernel(AttnArgs p) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;
    extern __shared__ __attribute__((aligned(16))) unsigned char res_lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int pair = blockIdx.x, b = pair / p.H, hd = pair - b * p.H;
    const int HD = p.H * 64, L = p.L, Lp = attn_padded_len_dev(L), Lr = (L + 31) & ~31;
    elem* Ks = (elem*)res_lds;
    elem* Vs = Ks + (size_t)Lr * 64;
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    // stage the pair: Lr rows x 8 chunks of K and of V, four chunks of each in flight per thread; rows >= L are zero
    for (int i0 = tid; i0 < Lr * 8; i0 += 4 * nthr) {
        vec8 sk[4], sv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = i0 + k * nthr, row = idx >> 3, ch = idx & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) { sk[k][i] = (elem)0.f; sv[k][i] = (elem)0.f; }
            if (row < L) {
                sk[k] = *(const vec8*)(base + HD + (size_t)row * ld + ch * 8);
                sv[k] = *(const vec8*)(base + 2 * HD + (size_t)row * ld + ch * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = i0 + k * nthr, row = idx >> 3, ch = idx & 7;
            if (row < Lr) {
                *(vec8*)(Ks + row * 64 + ((ch ^ flash_fk(row)) << 3)) = sk[k];
                *(vec8*)(Vs + row * 64 + ((ch ^ flash_fv(row)) << 3)) = sv[k];
            }
        }
    }
    __syncthreads();  // the only barrier: K and V are read-only from here on
    const int gp = (lane >> 4) & 1, i16 = lane & 15;
    const int nqb = (L + 31) >> 5, ntile = Lr >> 5;
    const size_t ldo = p.ld_out ? (size_t)p.ld_out : (size_t)HD;
    for (int qb = wave; qb < nqb; qb += nw) {
        const int q0 = qb * 32, q = q0 + n;
        vec8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[ks][i] = (elem)0.f;
            if (q < L) qf[ks] = *(const vec8*)(base + (size_t)q * ld + 16 * ks + 8 * h);
        }
        f32x16 O[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
        float m = -INFINITY, l = 0.f;
        const int tend = CAUSAL ? (qb + 1 < ntile ? qb + 1 : ntile) : ntile;  // 32-key tiles this block sees
        // NT = 2: a 64-key step (one maximum / rescale for two tiles); NT = 1: the odd tile at the end
        auto step = [&](auto nt_c, int t0) {
            constexpr int NT = decltype(nt_c)::value;
            const int key0 = t0 * 32;
            f32x16 S[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[t][r] = 0.f;
                const int row = key0 + 32 * t + n;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    S[t] = T::mfma32(*(const vec8*)(Ks + row * 64 + (((2 * ks + h) ^ flash_fk(row)) << 3)), qf[ks], S[t]);
            }
            if (CAUSAL || key0 + 32 * NT > L) {  // only the step that holds the end of the sequence (and causal steps) needs the per-key mask
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (key >= L || (CAUSAL && key > q)) S[t][r] = -INFINITY;
                    }
            }
            float mloc = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, S[t][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float mnew = fmaxf(m, mloc);  // finite from the first step on: key 0 is visible to every query
            // Deferred rescale: m is the REFERENCE of the exponentials, not necessarily the running maximum.  It moves (and l and O are
            // rescaled, 34 multiplies) only when some row of the wave outgrew it by more than 2^RES_DEFER; until then p <= 2^RES_DEFER, exact
            // in fp32 and at the same relative precision in T.  The first step (m = -inf) always takes the branch.
            if (__builtin_amdgcn_ballot_w64((mnew - m) * SC > RES_DEFER) != 0) {
                const float alpha = mnew == -INFINITY ? 1.f : __builtin_amdgcn_exp2f((m - mnew) * SC);
                m = mnew;
                l *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) O[dt] *= alpha;
            }
            const float nm = m == -INFINITY ? 0.f : -m * SC;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[t][r], SC, nm));
                    S[t][r] = e;
                    l += e;
                }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    vec8 pb;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[j] = (elem)S[t][8 * sl + j];
                    const int rr = key0 + 32 * t + 16 * sl + 4 * h + (i16 >> 2);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int col = 32 * dt + 16 * gp + 4 * (i16 & 3);
                        const elem* vp = Vs + rr * 64 + ((((col >> 3) ^ flash_fv(rr)) << 3) + (col & 7));
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)vp);
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp + 8 * 64));
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const vec8 vf = __builtin_bit_cast(vec8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                        O[dt] = T::mfma32(vf, pb, O[dt]);
                    }
                }
        };
        int t0 = 0;
        for (; t0 + 2 <= tend; t0 += 2) step(std::integral_constant<int, 2>{}, t0);
        if (t0 < tend) step(std::integral_constant<int, 1>{}, t0);
        l += __shfl_xor(l, 32, 64);
        if (q < L) {
            const float inv = 1.f / l;
            const size_t off = ((size_t)b * L + q) * ldo + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int d0 = 32 * dt + 8 * rr + 4 * h;  // registers 4 rr .. 4 rr + 3 of tile dt are d0 .. d0 + 3 of this lane's query
                    vec4 hv, lv;
                    float rem[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        elem a;
                        rem[e] = split_rem(O[dt][4 * rr + e] * inv, a);
                        hv[e] = a; lv[e] = (elem)rem[e];
                    }
                    *(vec4*)((elem*)p.out + off + d0) = hv;
                    if (p.out_lo) {  // the low half of a split operand (common.h LoMode): T, or e4m3 bytes at the same row stride in bytes
                        if (p.lo_mode == LO_F8) *(uint32_t*)((char*)p.out_lo + ((size_t)b * L + q) * ldo * 2 + hd * 64 + d0) = pack_lo8(rem[0], rem[1], rem[2], rem[3]);
                        else *(vec4*)((elem*)p.out_lo + off + d0) = lv;
                    }
                }
        }
        if (h == 0 && p.lse && q < Lp) p.lse[(size_t)pair * Lp + q] = q < L ? m * 0.125f + __logf(l) : 0.f;
    }
}

// Resident backward (224 < L, the operands of a pair fit the CU's LDS): the same structure as the resident forward, 32 rows of the lane
// operand per wave on v_mfma_f32_32x32x16, no barrier after the staging.  Both LDS images are read row-wise (ds_read_b128 fragments of the
// S / dP products) AND transposed (ds_read_b64_tr_b16 blocks of the dQ / dK / dV products), so they share one swizzle that is conflict-free
// for both: res_f(r) = bit 1 of r -> bit 2, bits 2..3 of r -> bits 0..1 (brute-force checked against the bank maps of
// MI355X_MICROARCH.md's LDS table: 4 x 16-lane groups for b128, 2 x 32 for the transposed 64-bit reads, 8 x 8 for the b128 stores).
__device__ inline int res_f(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// stage rows [0, Lr) of two [.., 64]-wide operands (row strides ld0 / ld1 elements) into swizzled images; rows >= L are zero
template <typename T>
__device__ inline void res_stage2(typename T::elem* I0, typename T::elem* I1, const typename T::elem* s0, size_t ld0, const typename T::elem* s1,
                                  size_t ld1, int L, int Lr, int tid, int nthr) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    for (int i0 = tid; i0 < Lr * 8; i0 += 4 * nthr) {
        vec8 a[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = i0 + k * nthr, row = idx >> 3, ch = idx & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[k][i] = (elem)0.f; c[k][i] = (elem)0.f; }
            if (row < L) {
                a[k] = *(const vec8*)(s0 + (size_t)row * ld0 + ch * 8);
                c[k] = *(const vec8*)(s1 + (size_t)row * ld1 + ch * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = i0 + k * nthr, row = idx >> 3, ch = idx & 7;
            if (row < Lr) {
                *(vec8*)(I0 + row * 64 + ((ch ^ res_f(row)) << 3)) = a[k];
                *(vec8*)(I1 + row * 64 + ((ch ^ res_f(row)) << 3)) = c[k];
            }
        }
    }
}

// A operand of a transposed product from a res_f image: the 16-row slab at row0 (rows row0 + 4 h + {0..3} and + 8), columns 32 dt .. + 31
template <typename T>
__device__ inline typename T::vec8 res_tr(const typename T::elem* img, int row0, int dt, int lane) {
    using vec8 = typename T::vec8;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int h = lane >> 5, gp = (lane >> 4) & 1, i16 = lane & 15;
    const int r0 = row0 + 4 * h + (i16 >> 2), r1 = r0 + 8, col = 32 * dt + 16 * gp + 4 * (i16 & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r0 * 64 + ((((col >> 3) ^ res_f(r0)) << 3) + (col & 7))));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + r1 * 64 + ((((col >> 3) ^ res_f(r1)) << 3) + (col & 7))));
    return __builtin_bit_cast(vec8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// store a [64 dims x 32 rows]^T accumulator pair (registers 4 rr .. 4 rr + 3 of tile dt = dims 32 dt + 8 rr + 4 h .. + 3 of this lane's row)
template <typename T>
__device__ inline void res_store_t(typename T::elem* dst, const f32x16 (&acc)[2], float scale, int h) {
    using elem = typename T::elem;
    using vec4 = typename T::vec4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            vec4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (elem)(acc[dt][4 * rr + e] * scale);
            *(vec4*)(dst + 32 * dt + 8 * rr + 4 * h) = v;
        }
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(RES_W * 64) void attn_bwd_dq_resident_kernel(AttnArgs p, const void* fwd_out) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char res_lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int pair = blockIdx.x, b = pair / p.H, hd = pair - b * p.H;
    const int HD = p.H * 64, L = p.L, Lp = attn_padded_len_dev(L), Lr = Lp;
    elem* Ks = (elem*)res_lds;
    elem* Vs = Ks + (size_t)Lr * 64;
    const size_t ld = (size_t)3 * HD, ldof = p.ld_out ? (size_t)p.ld_out : (size_t)HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;
    const elem* Of = (const elem*)fwd_out + (size_t)b * L * ldof + hd * 64;
    res_stage2<T>(Ks, Vs, base + HD, ld, base + 2 * HD, ld, L, Lr, tid, nthr);
    __syncthreads();
    const int nqb = Lr >> 5, ntile = Lr >> 5;
    for (int qb = wave; qb < nqb; qb += nw) {
        const int q = qb * 32 + n;
        vec8 qf[4], gf[4];
        float delta = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            vec8 of;
#pragma unroll
            for (int i = 0; i < 8; ++i) { qf[ks][i] = (elem)0.f; gf[ks][i] = (elem)0.f; of[i] = (elem)0.f; }
            if (q < L) {
                qf[ks] = *(const vec8*)(base + (size_t)q * ld + 16 * ks + 8 * h);
                gf[ks] = *(const vec8*)(dO + (size_t)q * HD + 16 * ks + 8 * h);
                of = *(const vec8*)(Of + (size_t)q * ldof + 16 * ks + 8 * h);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) delta += (float)gf[ks][i] * (float)of[i];
        }
        delta += __shfl_xor(delta, 32, 64);
        const size_t stat = (size_t)pair * Lp + q;  // q < Lp always: Lp = 32 nqb
        if (h == 0) p.delta[stat] = delta;
        const float nlse = q < L ? -p.lse[stat] * LOG2E : 0.f;
        f32x16 dQ[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQ[dt][r] = 0.f;
        const int tend = CAUSAL ? (qb + 1 < ntile ? qb + 1 : ntile) : ntile;
        auto tile = [&](auto masked_c, int t0) {
            constexpr bool MASKED = decltype(masked_c)::value;
            const int key0 = t0 * 32, row = key0 + n;
            f32x16 S, dP;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = row * 64 + (((2 * ks + h) ^ res_f(row)) << 3);
                S = T::mfma32(*(const vec8*)(Ks + off), qf[ks], S);
                dP = T::mfma32(*(const vec8*)(Vs + off), gf[ks], dP);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], SC, nlse));
                if (MASKED) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= L || (CAUSAL && key > q)) pr = 0.f;
                }
                S[r] = pr * (dP[r] - delta);
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                vec8 db;
#pragma unroll
                for (int j = 0; j < 8; ++j) db[j] = (elem)S[8 * sl + j];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dQ[dt] = T::mfma32(res_tr<T>(Ks, key0 + 16 * sl, dt, lane), db, dQ[dt]);
            }
        };
        // tiles below the diagonal (causal) / before the end of the sequence need no per-key mask
        const int tfull = CAUSAL ? (qb < (L >> 5) ? qb : (L >> 5)) : (L >> 5);
        int t0 = 0;
        for (; t0 < tfull; ++t0) tile(std::false_type{}, t0);
        for (; t0 < tend; ++t0) tile(std::true_type{}, t0);
        if (q < L) res_store_t<T>((elem*)p.dqkv + ((size_t)b * L + q) * ld + hd * 64, dQ, 0.125f, h);
    }
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(RES_W_BWD * 64) void attn_bwd_dkv_resident_kernel(AttnArgs p) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char res_lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int pair = blockIdx.x, b = pair / p.H, hd = pair - b * p.H;
    const int HD = p.H * 64, L = p.L, Lp = attn_padded_len_dev(L), Lr = Lp;
    elem* Qs = (elem*)res_lds;
    elem* Gs = Qs + (size_t)Lr * 64;
    // per query: -8 lse (the S accumulator starts there: exp2(SC (q.k - 8 lse)) = p; -inf on the padding queries: p = 0) and -delta (the dP
    // accumulator starts there): ds_read_b128 straight into the MFMA C tuples, no per-element add
    float* lse_s = (float*)(Gs + (size_t)Lr * 64);
    float* del_s = lse_s + Lr;
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;
    res_stage2<T>(Qs, Gs, base, ld, dO, (size_t)HD, L, Lr, tid, nthr);
    for (int i = tid; i < Lr; i += nthr) {
        lse_s[i] = i < L ? -8.f * p.lse[(size_t)pair * Lp + i] : -INFINITY;
        del_s[i] = i < L ? -p.delta[(size_t)pair * Lp + i] : 0.f;
    }
    __syncthreads();
    const int nkb = Lr >> 5, ntile = Lr >> 5;
    for (int kb = wave; kb < nkb; kb += nw) {
        const int key = kb * 32 + n;
        vec8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { kf[ks][i] = (elem)0.f; vf[ks][i] = (elem)0.f; }
            if (key < L) {
                kf[ks] = *(const vec8*)(base + HD + (size_t)key * ld + 16 * ks + 8 * h);
                vf[ks] = *(const vec8*)(base + 2 * HD + (size_t)key * ld + 16 * ks + 8 * h);
            }
        }
        f32x16 dK[2], dV[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dK[dt][r] = 0.f; dV[dt][r] = 0.f; }
        for (int t0 = CAUSAL ? kb : 0; t0 < ntile; ++t0) {  // causal: query tiles that hold a query >= this block's first key
            const int q0 = t0 * 32, row = q0 + n;
            f32x16 S, dP;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 l4 = *(const f32x4*)(lse_s + q0 + 8 * j + 4 * h), d4 = *(const f32x4*)(del_s + q0 + 8 * j + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { S[4 * j + e] = l4[e]; dP[4 * j + e] = d4[e]; }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = row * 64 + (((2 * ks + h) ^ res_f(row)) << 3);
                S = T::mfma32(*(const vec8*)(Qs + off), kf[ks], S);
                dP = T::mfma32(*(const vec8*)(Gs + off), vf[ks], dP);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = __builtin_amdgcn_exp2f(S[r] * SC);
                if (CAUSAL && key > q0 + (r & 3) + 8 * (r >> 2) + 4 * h) pr = 0.f;
                S[r] = pr;
                dP[r] *= pr;
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                vec8 pb, db;
#pragma unroll
                for (int j = 0; j < 8; ++j) { pb[j] = (elem)S[8 * sl + j]; db[j] = (elem)dP[8 * sl + j]; }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dV[dt] = T::mfma32(res_tr<T>(Gs, q0 + 16 * sl, dt, lane), pb, dV[dt]);
                    dK[dt] = T::mfma32(res_tr<T>(Qs, q0 + 16 * sl, dt, lane), db, dK[dt]);
                }
            }
        }
        if (key < L) {
            elem* ok = (elem*)p.dqkv + ((size_t)b * L + key) * ld + HD + hd * 64;
            res_store_t<T>(ok, dK, 0.125f, h);
            res_store_t<T>(ok + HD, dV, 1.f, h);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool attn_resident_fits(int L, bool bwd) {
    const int Lr = (L + 31) & ~31;
    return L > 224 && Lr * (256 + (bwd ? 8 : 0)) <= RES_LDS_MAX;  // L <= 224: the whole-pair kernels of attention.hip (resident forms measured 28 / 52 % slower at L = 201)
}

template <typename T>
static int fwd_resident(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    const int Lr = (a.L + 31) & ~31, lds = Lr * 256;
    const int nqb = Lr / 32, nwv = nqb <= RES_W_MAX ? nqb : RES_W;  // one round if the blocks fit the workgroup, else 3 waves per SIMD
    static PerDevice pd[2];
    const int dev = current_device();
    if (!pd[a.causal].done[dev]) {
        const void* k = a.causal ? (const void*)attn_fwd_resident_kernel<T, true> : (const void*)attn_fwd_resident_kernel<T, false>;
        HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_MAX));
        pd[a.causal].done[dev] = true;
    }
    if (a.causal) MUDPT_LAUNCH((attn_fwd_resident_kernel<T, true>), dim3(a.B * a.H), dim3(nwv * 64), lds, s, prof, a);
    else MUDPT_LAUNCH((attn_fwd_resident_kernel<T, false>), dim3(a.B * a.H), dim3(nwv * 64), lds, s, prof, a);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// K | V (dQ kernel, which also writes delta), then Q | dO | lse | delta (dK/dV kernel) of a pair in LDS
template <typename T>
static int bwd_resident(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    const LaunchProf p1{prof ? prof->start : nullptr, nullptr}, p2{nullptr, prof ? prof->stop : nullptr};
    const int Lr = (a.L + 31) & ~31, lds1 = Lr * 256, lds2 = Lr * (256 + 8), nb = Lr / 32;
    const int nw1 = nb <= RES_W ? nb : RES_W, nw2 = nb <= RES_W_BWD ? nb : RES_W_BWD;  // dQ: 164 VGPRs fit three waves per SIMD
    static PerDevice pd[2];
    const int dev = current_device();
    if (!pd[a.causal].done[dev]) {
        const void* k1 = a.causal ? (const void*)attn_bwd_dq_resident_kernel<T, true> : (const void*)attn_bwd_dq_resident_kernel<T, false>;
        const void* k2 = a.causal ? (const void*)attn_bwd_dkv_resident_kernel<T, true> : (const void*)attn_bwd_dkv_resident_kernel<T, false>;
        HIP_TRY(hipFuncSetAttribute(k1, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_MAX));
        HIP_TRY(hipFuncSetAttribute(k2, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_MAX));
        pd[a.causal].done[dev] = true;
    }
    const dim3 gp(a.B * a.H);
    if (a.causal) {
        MUDPT_LAUNCH((attn_bwd_dq_resident_kernel<T, true>), gp, dim3(nw1 * 64), lds1, s, &p1, a, (const void*)a.out);
        MUDPT_LAUNCH((attn_bwd_dkv_resident_kernel<T, true>), gp, dim3(nw2 * 64), lds2, s, &p2, a);
    } else {
        MUDPT_LAUNCH((attn_bwd_dq_resident_kernel<T, false>), gp, dim3(nw1 * 64), lds1, s, &p1, a, (const void*)a.out);
        MUDPT_LAUNCH((attn_bwd_dkv_resident_kernel<T, false>), gp, dim3(nw2 * 64), lds2, s, &p2, a);
    }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

int launch_attn_fwd_resident(int dtype, const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    return dtype == DT_BF16 ? fwd_resident<BF16>(a, s, prof) : fwd_resident<F16>(a, s, prof);
}
int launch_attn_bwd_resident(int dtype, const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    return dtype == DT_BF16 ? bwd_resident<BF16>(a, s, prof) : bwd_resident<F16>(a, s, prof);
}

}  // namespace mudpt
