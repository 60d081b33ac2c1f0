// Shared device/host definitions for the MuDPT gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace mudpt {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// Operand dtype of the MFMA GEMMs / attention (fp32 accumulate, fp32 residual stream either way).
enum DType : int { DT_BF16 = 0, DT_F16 = 1 };

struct BF16 {
    using elem = __bf16;
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    static constexpr int id = DT_BF16;
    __device__ static inline f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static inline f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
struct F16 {
    using elem = _Float16;
    using vec8 = f16x8;
    using vec4 = f16x4;
    static constexpr int id = DT_F16;
    __device__ static inline f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    __device__ static inline f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// sigmoid(1.702 u) = 1 / (1 + 2^(-1.702 log2(e) u)) from the two hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each) instead
// of expf + an IEEE division: the division alone is ~10 VALU instructions (v_div_scale x 2, v_rcp, 5 fma, v_div_fmas, v_div_fixup), and a
// 256 x 256 MLP tile evaluates this 65 536 times in an epilogue during which the matrix cores idle (measured: DESIGN.md 4).  The
// result is rounded to T (8 / 11 bits) right away; u -> +-inf gives exactly 0 / 1.
__device__ inline float sigmoid_1702(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * u));
}
__device__ inline float quick_gelu(float u) {  // clip/model.py:173-175  x * sigmoid(1.702 x)
    return u * sigmoid_1702(u);
}
__device__ inline float quick_gelu_grad(float u) {
    const float s = sigmoid_1702(u);
    return s * (1.0f + 1.702f * u * (1.0f - s));
}

// QuickGELU'(u) in 8 bits (GemmArgs::gelu_q8; the bf16 throughput mode): the backward needs u only through QuickGELU'(u), which lives in
// [-0.0998, 1.0998].  c_fc's epilogue stores code = rint((g' + GQ8_OFF) * GQ8_SCALE) in [0, 254] instead of u in T (1 byte instead of 2 per
// MLP activation, written once and read once per step), and the d(QuickGELU) epilogue multiplies by the decoded value: absolute error
// <= 0.5 / GQ8_SCALE = 2.4e-3, the size of bf16's own rounding of a value near 1 (2^-9 = 2.0e-3).
constexpr float GQ8_SCALE = 212.f, GQ8_OFF = 0.1f;
__device__ inline uint32_t gelu_grad_q8x4(float u0, float u1, float u2, float u3) {  // byte j = code of u_j
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32((quick_gelu_grad(u0) + GQ8_OFF) * GQ8_SCALE, 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32((quick_gelu_grad(u1) + GQ8_OFF) * GQ8_SCALE, 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32((quick_gelu_grad(u2) + GQ8_OFF) * GQ8_SCALE, 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32((quick_gelu_grad(u3) + GQ8_OFF) * GQ8_SCALE, 3, w);
    return w;
}
__device__ inline float gelu_grad_from_q8(uint32_t word, int j) {  // decoded QuickGELU' of byte j
    return __builtin_fmaf((float)((word >> (8 * j)) & 0xffu), 1.f / GQ8_SCALE, -GQ8_OFF);
}

// Split operands (Tower::split_mode).  A forward GEMM's A operand v is stored as hi = T(v) in the ordinary operand buffer plus the
// remainder lo = v - hi in a SECOND buffer with the same row stride in BYTES, in one of two forms:
//   LO_F16: lo as T.  hi + lo carries 22 bits; the GEMM runs a second pass over lo against the same weights (what the text tower uses: it
//           needs every bit, tests/precision_ablation.py);
//   LO_F8 : lo * 2^LO8_SHIFT as OCP e4m3 (the first K bytes of each row; the rest of the row is padding so that byte offsets into hi and lo
//           agree).  The second pass runs on the MX-scaled fp8 matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4, 2x the fp16 rate, K = 128
//           per instruction) against an e4m3 copy of the weights: the pair carries ~15 bits and the pass costs half of the first one.
// LO8_SHIFT: |lo| <= 2^-11 |v|, so lo * 2^12 stays below e4m3's maximum 448 for |v| < 224 (beyond: saturates, the element falls back towards
// fp16 precision) and inside e4m3's normal range (>= 2^-6) for |v| >= 2^-7.  The hardware conversion returns NaN beyond 448
// (tools/probes/mfma_f8.py): the clamp is explicit.
enum LoMode : int { LO_NONE = 0, LO_F16 = 1, LO_F8 = 2 };
constexpr int LO8_SHIFT = 12;
constexpr int LO8_SCALE_E8M0 = 127 - LO8_SHIFT;  // the MX block scale 2^-LO8_SHIFT of the fp8 pass's activation operand

// T(v) of a value an epilogue has just computed, through an opaque register: the conversion then rounds the fp32 VALUE.  With -ffp-contract=fast
// (hipcc's default) the compiler may fold the multiply / add that produced v into the conversion (v_fma_mixlo_f16: ONE rounding of the exact
// result) in one kernel and not in another, and the same operands then come out one ulp of T apart depending on which kernel ran the shape --
// round 4: the QuickGELU epilogues of gemm_nt and gemm_pp, which made a trimmed text tower (small-tile kernel) differ from the untrimmed one
// (persistent kernel) by 2e-3 of a gradient's rms where they used to agree to 1e-6.
template <typename E>
__device__ inline E round_to(float v) {
    asm volatile("" : "+v"(v));
    return (E)v;
}

// hi = T(v); returns the fp32 remainder v - hi.  The value passes through an opaque register first: with -ffp-contract=fast (hipcc's
// default) the compiler may form `hi` from the UNROUNDED product that made v (v_fma_mixlo_f16: one rounding) in one place and from the
// rounded v in another; on exact ties the two disagree and hi + lo is then off by a whole ulp of T (measured in round 3: 1 element in
// ~8 000 of the exact attention output).
template <typename E>
__device__ inline float split_rem(float v, E& hi) {
    asm volatile("" : "+v"(v));
    hi = (E)v;
    float h = (float)hi;
    asm volatile("" : "+v"(h));
    return v - h;
}
template <typename E>
__device__ inline void split_hi_lo(float v, E& hi, E& lo) {
    lo = (E)split_rem(v, hi);
}
// four remainders -> four e4m3 bytes (byte j = element j), scaled by 2^LO8_SHIFT, saturating
__device__ inline uint32_t pack_lo8(float a, float b, float c, float d) {
    constexpr float S = (float)(1 << LO8_SHIFT);
    auto sat = [](float x) { return __builtin_fminf(__builtin_fmaxf(x * S, -448.f), 448.f); };
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(sat(a), sat(b), 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(sat(c), sat(d), w, true);
    return (uint32_t)w;
}

// Wave-wide reductions on the DPP path (round 4): a butterfly inside each row of 16 lanes (quad_perm, row_half_mirror, row_mirror), the
// two row broadcasts that carry the partial results into the last row, and a v_readlane of lane 63 -- six single-cycle-issue VALU
// operations with no LDS round trip.  The __shfl_xor form they replace lowers to six DEPENDENT ds_bpermute_b32 (~60 clocks each through
// the LDS crossbar): two such chains sat between the load and the store phase of every LayerNorm row.
template <int CTRL, int ROW_MASK = 0xf>
__device__ inline float dpp_mov(float v, float old) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ inline float wave_sum(float v) {
    v += dpp_mov<0xb1>(v, 0.f);         // quad_perm [1, 0, 3, 2]
    v += dpp_mov<0x4e>(v, 0.f);         // quad_perm [2, 3, 0, 1]
    v += dpp_mov<0x141>(v, 0.f);        // row_half_mirror
    v += dpp_mov<0x140>(v, 0.f);        // row_mirror: every lane of a row holds the row's sum
    v += dpp_mov<0x142, 0xa>(v, 0.f);   // row_bcast:15 into rows 1, 3
    v += dpp_mov<0x143, 0xc>(v, 0.f);   // row_bcast:31 into rows 2, 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ inline float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xb1>(v, v));
    v = fmaxf(v, dpp_mov<0x4e>(v, v));
    v = fmaxf(v, dpp_mov<0x141>(v, v));
    v = fmaxf(v, dpp_mov<0x140>(v, v));
    v = fmaxf(v, dpp_mov<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_mov<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Bijective XCD-aware remap of a 1-D grid: blocks that share an XCD (bid % 8) get one contiguous
// chunk of work ids, so neighbouring tiles hit the same per-XCD L2 (speed only, never correctness).
__device__ inline int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

}  // namespace mudpt

// ---- host-side error plumbing (no exceptions cross the C ABI) -------------------------------
#define MUDPT_OK 0
#define MUDPT_ERR_ARG 1
#define MUDPT_ERR_HIP 2
#define MUDPT_ERR_STATE 3

namespace mudpt {
void set_error(const char* fmt, ...);
}

namespace mudpt {
// One-time-per-DEVICE launch setup (function attributes and CU counts belong to a device; one process may drive several GPUs).
struct PerDevice {
    bool done[64] = {};
    int ncu[64] = {};
};
// Optional HIP events that ride on a kernel's own dispatch packet (measurement legs of bench.py): start is recorded when the kernel
// begins, stop when it ends; no marker packets between kernels.
struct LaunchProf {
    hipEvent_t start = nullptr, stop = nullptr;
};
#define MUDPT_LAUNCH(kern, grid, block, lds, stream, prof, ...)                                                          \
    do {                                                                                                                 \
        const mudpt::LaunchProf* _lp = (prof);                                                                           \
        if (_lp && (_lp->start || _lp->stop)) hipExtLaunchKernelGGL(kern, grid, block, lds, stream, _lp->start, _lp->stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                             \
    } while (0)
inline int current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 63;
}
}  // namespace mudpt

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            mudpt::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return MUDPT_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

#define ARG_CHECK(cond, ...)                                                                \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            mudpt::set_error(__VA_ARGS__);                                                  \
            return MUDPT_ERR_ARG;                                                           \
        }                                                                                   \
    } while (0)
