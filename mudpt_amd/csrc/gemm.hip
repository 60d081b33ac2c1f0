// MFMA GEMM for gfx950:  C[M,N] = A[M,K] . B[N,K]^T, bf16/fp16 operands, fp32 accumulate, fused epilogues.
//
// Replaces the reference's nn.Linear / conv-as-GEMM calls on the path (clip/model.py:257-263 in_proj,
// out_proj, c_fc, c_proj; :527 conv1) and their autograd dX products.
//
// Structure (v1): BM x BN x 64 block tile, WM x WN waves, v_mfma_f32_16x16x32 tiles.  Both operand tiles
// are staged global -> LDS with 16-byte global_load_lds (no VGPR round trip), double buffered.  The LDS
// image of a tile is [rows][64] (128-byte rows); a wave instruction fills 8 rows.  global_load_lds writes
// lane-linear, so the bank swizzle (16-byte chunk c of row r sits in slot c ^ (r & 7)) is applied on the
// per-lane SOURCE address and again on the ds_read_b128 address; with it every 16-lane group of a
// ds_read_b128 touches 16 distinct slots of the 256-byte bank row (conflict free).
// The MFMA computes the transposed tile D[n][m] = B-frag x A-frag so that a lane owns 4 consecutive
// output columns of one row: the epilogue stores 8 (T) or 16 (fp32) contiguous bytes per lane.
#include <type_traits>

#include "kernels.h"

namespace mudpt {

using gptr_t = const __attribute__((address_space(1))) void*;
using lptr_t = __attribute__((address_space(3))) void*;

// BK = 128 (round 4, small grids): 256-byte LDS rows, a wave instruction fills 4 of them, chunk c of row r sits in slot c ^ (r & 15) -- half the
// K-steps, i.e. half the wait -> barrier -> fragment-read -> MFMA chains of a workgroup that has the CU almost to itself.  No split operand.
template <typename T, int BM, int BN, int WM, int WN, int EPI, int NS = 2, int BK = 64>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(GemmArgs p) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int NW = WM * WN;
    static_assert(BK == 64 || BK == 128, "K-tile");
    constexpr int ROWB = BK * 2;      // bytes of an LDS row
    constexpr int RPI = 1024 / ROWB;  // rows a wave's DMA instruction fills
    constexpr int CPR = ROWB / 16;    // 16-byte chunks per row
    constexpr int TM = BM / WM / 16;  // 16-row sub-tiles per wave along M
    constexpr int TN = BN / WN / 16;
    constexpr int IA = BM / RPI / NW;   // global_load_lds instructions per wave for the A tile
    constexpr int IB = BN / RPI / NW;
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "tile rows must split over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int ntn = (p.N + BN - 1) / BN;
    const int wg = (p.flags & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / ntn) * BM;
    const int n0 = (wg % ntn) * BN;

    // split K: slice blockIdx.y of the contraction (gridDim.y = 1 and ksplit = 0 otherwise)
    const int kz = blockIdx.y, kspan = p.ksplit ? p.ksplit : p.K;
    const elem* __restrict__ A = (const elem*)p.A + (size_t)kz * kspan;
    const elem* __restrict__ Bw = (const elem*)p.B + (size_t)kz * kspan;

    // per-lane source pointers (k offset advances by BK per tile); rows clamped at the ragged edge
    const int srow = lane / CPR;                      // row inside the RPI-row group this lane fills
    const elem* asrc[IA];
    const elem* bsrc[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int tr = (wave + NW * i) * RPI + srow;  // row of the tile; swizzled source chunk for LDS slot lane % CPR
        int r = m0 + tr;
        r = r < p.M ? r : p.M - 1;
        asrc[i] = A + (size_t)r * p.lda + ((lane % CPR) ^ (tr & (CPR - 1))) * 8;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int tr = (wave + NW * i) * RPI + srow;
        int r = n0 + tr;
        r = r < p.N ? r : p.N - 1;
        bsrc[i] = Bw + (size_t)r * p.ldb + ((lane % CPR) ^ (tr & (CPR - 1))) * 8;
    }

    // Split A operand (common.h LoMode): after the nt1 K-tiles of the first pass (hi against B) a second pass contracts the low half --
    // LO_F16: nt1 more tiles of A_lo against the same B; LO_F8: K / 128 tiles of e4m3 bytes, A_lo against B8.  Both low buffers have the row
    // stride of their T counterparts IN BYTES, and every K-tile of either pass is 128 bytes of a row: a tile of the second pass is the
    // first pass's address plus a uniform byte offset (dA / dB) -- the staging below is otherwise unchanged.
    const int nt1 = kspan / BK;
    const bool f8 = p.lo_mode == LO_F8;
    const int nt = nt1 + (p.lo_mode == LO_NONE ? 0 : (f8 ? nt1 / 2 : nt1));
    const ptrdiff_t dA = p.A_lo ? (const char*)p.A_lo - (const char*)p.A : 0;
    const ptrdiff_t dB = f8 ? (const char*)p.B8 - (const char*)p.B : 0;
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
        const bool second = kt >= nt1;
        const ptrdiff_t oa = second ? dA + (ptrdiff_t)(kt - nt1) * ROWB : (ptrdiff_t)kt * ROWB;
        const ptrdiff_t ob = second ? dB + (ptrdiff_t)(kt - nt1) * ROWB : (ptrdiff_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)asrc[i] + oa), (lptr_t)(base + (wave + NW * i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < IB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)bsrc[i] + ob), (lptr_t)(base + BM * ROWB + (wave + NW * i) * 1024), 16, 0, 0);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = sub-tile base + (lane & 15), chunk = 4 * kstep + (lane >> 4)
    const int frow = lane & 15;
    const int fq = lane >> 4;
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) aoff[i] = (wm * (BM / WM) + i * 16 + frow) * ROWB;
#pragma unroll
    for (int j = 0; j < TN; ++j) boff[j] = BM * ROWB + (wn * (BN / WN) + j * 16 + frow) * ROWB;
    const int sw = frow & (CPR - 1);  // (row & (CPR - 1)): sub-tile bases are multiples of 16

    // NS-deep ring of stages: the loads of k-tiles kt + 1 .. kt + NS - 1 are in flight while tile kt is multiplied.  NS = 2 is the
    // plain double buffer (2 workgroups per CU hide each other's waits); NS = 4 is for grids smaller than the chip (the text
    // tower's 99-row GEMMs: a handful of workgroups, each a pure latency chain over K).
    static_assert(NS >= 2 && (NS & (NS - 1)) == 0, "ring depth must be a power of two");
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
        if (st < nt) stage(st, st);

    // One K-tile: wait for it, refill the stage freed by the previous tile, multiply.  The e4m3 tiles of the second pass run in their OWN loop
    // after the fp16 tiles (same ring, same staging), not behind a branch in one loop: a diamond around the in-place accumulation costs a
    // second set of accumulator registers (gemm_pp.hip; here it made the 256 x 256 patch-embed tile 3.4x slower).
    auto tile = [&](auto f8tag, int kt) {
        const int cur = kt & (NS - 1);
        // tile kt has landed once at most the NS - 2 younger stages are outstanding (loads retire in issue order)
        if (kt + NS - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (IA + IB)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // ... for every wave; and every wave is done reading the stage refilled next (read in step kt - 1)
        if (kt + NS - 1 < nt) stage((kt + NS - 1) & (NS - 1), kt + NS - 1);
        const char* base = smem + cur * STAGE_BYTES;
        if constexpr (decltype(f8tag)::value && BK == 64) {
            // e4m3 tile: 128 k per row; lane (frow, fq) owns bytes 32 fq .. 32 fq + 31 of its row = chunks 2 fq, 2 fq + 1 (swizzled like every
            // tile), ONE K = 128 matrix instruction per sub-tile pair (operand pairing and block scales: tools/probes/mfma_f8_layout.py)
            const int c = ((2 * fq) ^ sw) * 16;
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            i32x8 b8[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b8[j] = __builtin_shufflevector(*(const i32x4*)(base + boff[j] + c), *(const i32x4*)(base + boff[j] + (c ^ 16)), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const i32x8 a8 = __builtin_shufflevector(*(const i32x4*)(base + aoff[i] + c), *(const i32x4*)(base + aoff[i] + (c ^ 16)), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8, acc[i][j], 0, 0, 0, p.b8_scale, 0, LO8_SCALE_E8M0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                const int c = ((ks * 4 + fq) ^ sw) * 16;
                vec8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const vec8*)(base + aoff[i] + c);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *(const vec8*)(base + boff[j] + c);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = T::mfma16(bf[j], af[i], acc[i][j]);
            }
        }
    };
    const int nt16 = f8 ? nt1 : nt;  // LO_F16: the second pass is nt1 more tiles of the same kind
    for (int kt = 0; kt < nt16; ++kt) tile(std::false_type{}, kt);
    for (int kt = nt16; kt < nt; ++kt) tile(std::true_type{}, kt);

    // ---- epilogue: lane holds out[m][n .. n+3], m = sub-tile row (lane & 15), n = 4 * (lane >> 4) ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + frow;
        if (m >= p.M) continue;
        size_t orow = (size_t)m;
        const float* posrow = nullptr;
        if constexpr (EPI == EPI_PATCH) {
            const int b = m / p.patches, pp = m - b * p.patches;
            orow = (size_t)b * p.seq_len + 1 + pp;
            posrow = p.pos + (size_t)(1 + pp) * p.N;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 16 + fq * 4;
            if (n >= p.N) continue;  // N % 16 == 0 is checked on the host: sub-tiles are all-in or all-out
            f32x4 v = acc[i][j];
            if (p.bias) {
                const f32x4 b4 = *(const f32x4*)(p.bias + n);
                v += b4;
            }
            if constexpr (EPI == EPI_STORE) {
                typename T::vec4 o = {round_to<elem>(v[0]), round_to<elem>(v[1]), round_to<elem>(v[2]), round_to<elem>(v[3])};
                *(typename T::vec4*)((elem*)p.out0 + orow * p.ldo0 + n) = o;
            } else if constexpr (EPI == EPI_GELU) {
                if (p.gelu_q8) {  // the backward's QuickGELU'(u) in 8 bits instead of u (common.h)
                    *(uint32_t*)((char*)p.out0 + orow * p.ldo0 + n) = gelu_grad_q8x4(v[0], v[1], v[2], v[3]);
                } else {
                    typename T::vec4 u = {round_to<elem>(v[0]), round_to<elem>(v[1]), round_to<elem>(v[2]), round_to<elem>(v[3])};
                    *(typename T::vec4*)((elem*)p.out0 + orow * p.ldo0 + n) = u;
                }
                if (p.out1_lo) {  // split operand (common.h LoMode): the next GEMM's second pass contracts over the low half
                    typename T::vec4 g, lo;
                    float rem[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { elem hv; rem[c] = split_rem(quick_gelu(v[c]), hv); g[c] = hv; lo[c] = (elem)rem[c]; }
                    *(typename T::vec4*)((elem*)p.out1 + orow * p.ldo1 + n) = g;
                    if (p.out1_lo_mode == LO_F8) *(uint32_t*)((char*)p.out1_lo + orow * p.ldo1 * 2 + n) = pack_lo8(rem[0], rem[1], rem[2], rem[3]);
                    else *(typename T::vec4*)((elem*)p.out1_lo + orow * p.ldo1 + n) = lo;
                } else {
                    typename T::vec4 g = {round_to<elem>(quick_gelu(v[0])), round_to<elem>(quick_gelu(v[1])), round_to<elem>(quick_gelu(v[2])), round_to<elem>(quick_gelu(v[3]))};
                    *(typename T::vec4*)((elem*)p.out1 + orow * p.ldo1 + n) = g;
                }
            } else if constexpr (EPI == EPI_RESIDUAL) {
                const f32x4 r4 = *(const f32x4*)((const float*)p.aux + orow * p.ldaux + n);
                *(f32x4*)((float*)p.out0 + orow * p.ldo0 + n) = v + r4;
            } else if constexpr (EPI == EPI_GELU_BWD) {
                typename T::vec4 o;
                if (p.gelu_q8) {
                    const uint32_t w = *(const uint32_t*)((const char*)p.aux + orow * p.ldaux + n);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = round_to<elem>(v[c] * gelu_grad_from_q8(w, c));
                } else {
                    const typename T::vec4 u = *(const typename T::vec4*)((const elem*)p.aux + orow * p.ldaux + n);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = round_to<elem>(v[c] * quick_gelu_grad((float)u[c]));
                }
                *(typename T::vec4*)((elem*)p.out0 + orow * p.ldo0 + n) = o;
            } else if constexpr (EPI == EPI_PATCH) {
                const f32x4 q4 = *(const f32x4*)(posrow + n);
                *(f32x4*)((float*)p.out0 + orow * p.ldo0 + n) = v + q4;
            } else {  // EPI_STORE_F32
                *(f32x4*)((float*)p.out0 + (size_t)kz * p.split_stride + orow * p.ldo0 + n) = v;
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int EPI, int NS = 2, int BK = 64>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int lds = NS * (BM + BN) * BK * 2;
    auto kern = gemm_nt_kernel<T, BM, BN, WM, WN, EPI, NS, BK>;
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        pd.done[dev] = true;
    }
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(ntm * ntn, a.ksplit ? a.K / a.ksplit : 1), dim3(WM * WN * 64), lds, s, a);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// 64 x 64 tiles while they are all resident at once (five workgroups per CU x 256 CUs), or twice that for the short contractions of the
// width-512 text tower; beyond that the 128 x 128 tile's halved operand traffic wins (tools/gemm_bench.py --set small: M = 9 000 / 15 000,
// N = 768: 48.5 vs 61.2 and 80.2 vs 107.8 us at K = 3072)
static inline bool small_tiles(const GemmArgs& a) {
    const size_t t64 = (size_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
    return t64 <= 1280 || (t64 <= 2560 && a.K <= 512);
}

static int device_cus() {
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        if (hipDeviceGetAttribute(&pd.ncu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 256;
        pd.done[dev] = true;
    }
    return pd.ncu[dev];
}

// Round 4: 128-deep K-tiles for the 64 x 64 kernel while ALL its workgroups are resident at two per CU (64 KB of LDS each) -- these grids are
// latency chains (wait -> barrier -> fragment reads -> MFMAs per K-tile), and half as many links is what pays: qkv 804 x 2304 x 768 9.8 -> 8.6 us,
// proj 804 x 768 x 3072 in 3 slices 15.3 -> 13.8; the text tower's 1000 x 512 x 2048 11.2 -> 10.9.  Results are bit-identical to the 64-deep form
// (same order of the k-steps).  Grids beyond two workgroups per CU lose (fc 804 x 3072 x 768, 624 tiles: 13.7 -> 16.2): they keep 64 (a 64 x 96 x 128
// tile that would make them fit lost on every shape: fc 13.9 -> 14.3, qkv 8.9 -> 10.0; 256-deep tiles at one workgroup per CU: proj 804 x 768 x 3072
// 16.2 us unsplit against 14.1 for three slices of 128-deep tiles + their sum, the text tower's shapes +-0).
static inline bool deep_k_tiles(const GemmArgs& a, int slices, int variant) {
    const size_t t64 = (size_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
    return (variant & 0xff) != 12 && a.lo_mode == LO_NONE && (a.K / slices) % 128 == 0 && t64 * slices <= (size_t)2 * device_cus();
}

// variant: tuning knob (mudpt_model_set "gemm_variant" / mudpt_gemm's last argument): 0 = default kernel choice, 1/2/4 = force a simple tile,
// 12 = the default choice without the 128-deep K-tiles (A/B)
template <typename T, int EPI>
static int launch_epi(const GemmArgs& a, hipStream_t s, int variant) {
    // Large problems that do not go to the persistent ping-pong kernel (gemm_pp.hip): 256 x 256 tile on 8 waves;
    // small problems (text tower, tiny shapes): 128 x 128 on 4 waves.
    if ((size_t)a.M * a.N >= (size_t)256 * 128 * 512) {
        switch (variant & 0xff) {
            case 2: return launch_cfg<T, 128, 256, 2, 4, EPI>(a, s);
            case 4: return launch_cfg<T, 256, 128, 4, 2, EPI>(a, s);
            default: return launch_cfg<T, 256, 256, 2, 4, EPI>(a, s);
        }
    }
    // fewer 128 x 128 tiles than half the CUs: every workgroup is a latency chain over K -- narrower tiles (twice the
    // workgroups) and a 4-deep ring.  gemm_variant 5 / 6 force the shallow / deep form (A/B runs).
    const size_t t128 = (size_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const int v = variant & 0xff;
    // Round 3: on small grids (small_tiles: M = 804 at the reference's training batch of 4, the text tower up to ~6000 rows) a 64 x 64
    // tile on 4 waves with the plain double buffer wins on every shape measured (tools/gemm_bench.py --set small / text: sum of a block's
    // GEMMs 217 -> 167 us at M = 804): 32 KB of LDS lets five workgroups share a CU, and these grids are latency chains, not MFMA-bound.
    // gemm_variant 5 / 6 force the earlier 128 x 128 shallow / 128 x 64 deep forms, 9 this one (A/B runs).
    if (((v == 0 && small_tiles(a)) || v == 10) && deep_k_tiles(a, 1, variant)) return launch_cfg<T, 64, 64, 2, 2, EPI, 2, 128>(a, s);
    if (((v == 0 || v == 12) && small_tiles(a)) || v == 9 || v == 10) return launch_cfg<T, 64, 64, 2, 2, EPI, 2>(a, s);
    if ((t128 <= 128 && v != 5) || v == 6) return launch_cfg<T, 128, 64, 2, 2, EPI, 4>(a, s);
    return launch_cfg<T, 128, 128, 2, 2, EPI>(a, s);
}

template <typename T>
static int launch_t(int epi, const GemmArgs& a, hipStream_t s, int variant) {
    switch (epi) {
        case EPI_STORE: return launch_epi<T, EPI_STORE>(a, s, variant);
        case EPI_GELU: return launch_epi<T, EPI_GELU>(a, s, variant);
        case EPI_RESIDUAL: return launch_epi<T, EPI_RESIDUAL>(a, s, variant);
        case EPI_GELU_BWD: return launch_epi<T, EPI_GELU_BWD>(a, s, variant);
        case EPI_PATCH: return launch_epi<T, EPI_PATCH>(a, s, variant);
        case EPI_STORE_F32: return launch_epi<T, EPI_STORE_F32>(a, s, variant);
    }
    set_error("gemm: unknown epilogue %d", epi);
    return MUDPT_ERR_ARG;
}

// out[m][n] = sum over the S split-K slices (in slice order: bitwise reproducible) of part[z][m][n] (+ bias[n]), as T or fp32
template <typename T, bool OUT_F32>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, size_t stride, int M, int N, const float* __restrict__ bias,
                                                            void* __restrict__ out, int ldo) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n4 = (size_t)N / 4;
    if (i >= (size_t)M * n4) return;
    const size_t m = i / n4, n = (i - m * n4) * 4;
    f32x4 v = *(const f32x4*)(part + m * N + n);
    for (int z = 1; z < S; ++z) v += *(const f32x4*)(part + (size_t)z * stride + m * N + n);
    if (bias) v += *(const f32x4*)(bias + n);
    if constexpr (OUT_F32) {
        *(f32x4*)((float*)out + m * ldo + n) = v;
    } else {
        using elem = typename T::elem;
        typename T::vec4 o = {round_to<elem>(v[0]), round_to<elem>(v[1]), round_to<elem>(v[2]), round_to<elem>(v[3])};
        *(typename T::vec4*)((elem*)out + m * ldo + n) = o;
    }
}

// Split K for the store GEMMs whose grid is a fraction of the chip and whose contraction is long (small batches: M = B L = 804 rows at
// the reference's training batch of 4, K = 2304 / 3072): a 128 x 64 tile per workgroup leaves 2/3 of the CUs idle while every workgroup
// walks 36-48 K-steps.  S slices of K fill the chip; their fp32 partials go through the caller's scratch and are summed in slice order.
static int split_k_slices(int epi, const GemmArgs& a, const GemmOpts& o, bool& deep) {
    deep = false;
    const int v = o.variant & 0xff;
    if (!(epi == EPI_STORE || epi == EPI_STORE_F32) || !o.scratch || (v != 0 && v != 12) || a.lo_mode != LO_NONE) return 1;
    const int ncu = device_cus();
    const size_t tiles = (size_t)((a.M + 63) / 64) * ((a.N + 63) / 64);  // 64 x 64 tiles, five workgroups to a CU
    if (tiles * 2 > (size_t)ncu * 5 || a.K < 1536) return 1;
    // with 128-deep K-tiles two workgroups share a CU: three or four slices of those beat four of the 64-deep form (deep_k_tiles)
    int S = (int)((size_t)ncu * 2 / tiles);
    if (S > 4) S = 4;
    while (S > 1 && (a.K % (S * 128) != 0 || a.K / S < 512)) --S;
    if (S >= 3 && deep_k_tiles(a, S, o.variant) && (size_t)S * a.M * a.N <= o.scratch_elems) { deep = true; return S; }
    S = (int)((size_t)ncu * 5 / tiles);
    if (S > 4) S = 4;
    while (S > 1 && (a.K % (S * 64) != 0 || a.K / S < 512)) --S;
    if ((size_t)S * a.M * a.N > o.scratch_elems) return 1;
    return S;
}

// default dispatch: the persistent ping-pong kernel takes the big GEMMs whose epilogue needs no operand load besides bias / u
bool gemm_uses_pp(int epi, const GemmArgs& a, int variant) {
    const bool pp_epi = epi == EPI_STORE || epi == EPI_GELU || epi == EPI_GELU_BWD || epi == EPI_STORE_F32;
    const int v = variant & 0xff;
    // Size threshold (round 4: half of what it was): from 128 tiles of 256 x 256 on -- half the CUs with a tile each -- the persistent kernel beats the
    // 128 x 128 / 64 x 64 kernels on the shapes between the small grids and the vision tower's (tools/gemm_bench.py --set mid: the text tower at 1000
    // classes, 19 000 rows: width 768 proj 97 -> 80, dfc 95 -> 74, dqkv 75 -> 59 us, width 512 dfc 57 -> 49; CoCoOp's vision tower, 12 608 rows: dfc
    // 81 -> 69; its text tower's N = 2048 shapes 34 -> 26); at 50 tiles (6 336 x 512) it loses (proj 25 -> 34).  Same results bit for bit either way.
    const size_t tiles = (size_t)((a.M + 255) / 256) * ((a.N + 255) / 256);
    return (v == 0 || v == 3 || v == 5 || v == 6 || v == 12) && pp_epi && tiles >= (v == 3 ? 256 : 128) && a.ldo0 % 8 == 0 && (epi != EPI_GELU || a.ldo1 % 8 == 0) &&
           (epi != EPI_GELU_BWD || a.ldaux % 8 == 0);
}

int launch_gemm(int dtype, int epi, const GemmArgs& a, hipStream_t s, const GemmOpts& o) {
    const int variant = o.variant;
    ARG_CHECK(a.A && a.B && a.out0, "gemm: null operand");
    ARG_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty shape M=%d N=%d K=%d", a.M, a.N, a.K);
    ARG_CHECK(a.K % 64 == 0, "gemm: K=%d must be a multiple of 64", a.K);
    ARG_CHECK(a.N % 16 == 0, "gemm: N=%d must be a multiple of 16", a.N);
    ARG_CHECK(a.lda >= a.K && a.ldb >= a.K && a.lda % 8 == 0 && a.ldb % 8 == 0, "gemm: bad lda/ldb %d/%d", a.lda, a.ldb);
    ARG_CHECK(a.ldo0 >= a.N && a.ldo0 % 4 == 0, "gemm: bad ldo0 %d", a.ldo0);
    ARG_CHECK(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.B % 16 == 0) && ((uintptr_t)a.out0 % 16 == 0), "gemm: operands must be 16-byte aligned");
    if (epi == EPI_GELU) ARG_CHECK(a.out1 && a.ldo1 >= a.N && a.ldo1 % 4 == 0, "gemm: gelu epilogue needs out1");
    ARG_CHECK(!a.out1_lo || a.out1_lo_mode == LO_F16 || a.out1_lo_mode == LO_F8, "gemm: bad out1_lo_mode %d", a.out1_lo_mode);
    ARG_CHECK(!a.gelu_q8 || ((epi == EPI_GELU && !a.out1_lo) || epi == EPI_GELU_BWD), "gemm: gelu_q8 belongs to the QuickGELU epilogues (1 without a split output, 3)");
    ARG_CHECK(a.lo_mode == LO_NONE || a.lo_mode == LO_F16 || a.lo_mode == LO_F8, "gemm: bad lo_mode %d", a.lo_mode);
    if (a.lo_mode != LO_NONE) ARG_CHECK(a.A_lo && (uintptr_t)a.A_lo % 16 == 0, "gemm: a split operand needs its low half (16-byte aligned)");
    if (a.lo_mode == LO_F8) {
        ARG_CHECK(dtype == DT_F16, "gemm: the e4m3 second pass exists for fp16 operands only");
        ARG_CHECK(a.B8 && (uintptr_t)a.B8 % 16 == 0 && a.K % 128 == 0, "gemm: the e4m3 second pass needs B8 and K %% 128 == 0 (K = %d)", a.K);
        ARG_CHECK(a.b8_scale > 0 && a.b8_scale < 255, "gemm: bad E8M0 weight scale %d", a.b8_scale);
    }
    if (epi == EPI_RESIDUAL || epi == EPI_GELU_BWD) ARG_CHECK(a.aux && a.ldaux >= a.N && a.ldaux % 4 == 0, "gemm: epilogue needs aux");
    if (epi == EPI_PATCH) ARG_CHECK(a.pos && a.patches > 0 && a.seq_len > a.patches && a.M % a.patches == 0, "gemm: bad patch epilogue args");
    GemmArgs b = a;
    if (variant & 0x100) b.flags |= 1;
    if (variant & 0x200) b.flags |= 2;  // gemm_pp: no half tiles in the last wave
    b.flags |= ((variant >> 10) & 3) << 4;  // gemm_pp timing-only ablations (bits 10, 11 of the knob): no LDS fragment reads / no operand DMA
    b.flags |= ((variant >> 12) & 0xff) << 8;  // bits 12..19 of the knob: column-tile group width GN of gemm_pp (0 = default)
    // default: the persistent ping-pong kernel for the big GEMMs whose epilogue needs no operand load besides bias / u
    if (gemm_uses_pp(epi, a, o.variant)) return launch_gemm_pp(dtype, epi, b, s, o);
    bool deep = false;
    if (const int S = split_k_slices(epi, a, o, deep); S > 1) {
        GemmArgs q = b;
        q.bias = nullptr; q.out0 = o.scratch; q.ldo0 = a.N; q.ksplit = a.K / S; q.split_stride = (size_t)a.M * a.N;
        if (dtype == DT_BF16) { if (int rc = deep ? launch_cfg<BF16, 64, 64, 2, 2, EPI_STORE_F32, 2, 128>(q, s) : launch_cfg<BF16, 64, 64, 2, 2, EPI_STORE_F32, 2>(q, s)) return rc; }
        else if (dtype == DT_F16) { if (int rc = deep ? launch_cfg<F16, 64, 64, 2, 2, EPI_STORE_F32, 2, 128>(q, s) : launch_cfg<F16, 64, 64, 2, 2, EPI_STORE_F32, 2>(q, s)) return rc; }
        else { set_error("gemm: unknown dtype %d", dtype); return MUDPT_ERR_ARG; }
        const unsigned grid = (unsigned)(((size_t)a.M * (a.N / 4) + 255) / 256);
        if (epi == EPI_STORE_F32) hipLaunchKernelGGL((splitk_reduce_kernel<BF16, true>), dim3(grid), dim3(256), 0, s, o.scratch, S, q.split_stride, a.M, a.N, a.bias, a.out0, a.ldo0);
        else if (dtype == DT_BF16) hipLaunchKernelGGL((splitk_reduce_kernel<BF16, false>), dim3(grid), dim3(256), 0, s, o.scratch, S, q.split_stride, a.M, a.N, a.bias, a.out0, a.ldo0);
        else hipLaunchKernelGGL((splitk_reduce_kernel<F16, false>), dim3(grid), dim3(256), 0, s, o.scratch, S, q.split_stride, a.M, a.N, a.bias, a.out0, a.ldo0);
        HIP_TRY(hipGetLastError());
        return MUDPT_OK;
    }
    if (dtype == DT_BF16) return launch_t<BF16>(epi, b, s, o.variant);
    if (dtype == DT_F16) return launch_t<F16>(epi, b, s, o.variant);
    set_error("gemm: unknown dtype %d", dtype);
    return MUDPT_ERR_ARG;
}

}  // namespace mudpt
