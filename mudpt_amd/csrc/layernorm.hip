// LayerNorm forward / backward (fp32 statistics, eps 1e-5) for gfx950: HBM-bound, one wave per row,
// 16-byte accesses, the row held in registers between the reduction and the normalisation.
//
// Replaces clip/model.py:164-170 (LayerNorm evaluated in fp32) and its autograd.  gamma/beta are frozen
// on this path, so the backward produces dx only.  The backward also folds in the residual-stream add
// (dx = dres + LN'(dy)) and emits the low-precision copy the next dX GEMM consumes.
#include "kernels.h"

namespace mudpt {

constexpr int LN_MAXV = 4;  // float4 per lane: d <= 1024

template <typename T, bool OUT_F32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdArgs p) {
    using elem = typename T::elem;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const size_t xr = p.row_index ? (size_t)p.row_index[r] : (size_t)r;
    const f32x4* x = (const f32x4*)(p.x + xr * p.ldx);
    const int d4 = p.d >> 2;
    f32x4 v[LN_MAXV];
    float s = 0.f;
    // optional prompt splice: this row is replaced by a prompt row (wave-uniform decision)
    const f32x4* ov = nullptr;
    if (p.ov_rows) {
        const int pos = r % p.ov_L - p.ov_row0;
        if (pos >= 0 && pos < p.ov_n) ov = (const f32x4*)(p.ov_rows + (size_t)pos * p.d);
    }
    const f32x4* add = (p.add && !ov) ? (const f32x4*)(p.add + xr * p.ldadd) : nullptr;
    const typename T::vec4* add_lp = (p.add_lp && !ov) ? (const typename T::vec4*)((const elem*)p.add_lp + xr * p.ldadd) : nullptr;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + 64 * k;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < d4) {
            v[k] = ov ? ov[i] : x[i];
            if (add) v[k] += add[i];
            if (add_lp) { const typename T::vec4 a = add_lp[i]; v[k] += f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]}; }
            if (p.xout) ((f32x4*)(p.xout + xr * p.ldxout))[i] = v[k];
        }
        s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
    const float mean = wave_sum(s) / p.d;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + 64 * k;
        if (i < d4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float c = v[k][j] - mean; q += c * c; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / p.d + 1e-5f);
    if (lane == 0) {
        if (p.mean) p.mean[r] = mean;
        if (p.rstd) p.rstd[r] = rstd;
    }
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + 64 * k;
        if (i < d4) {
            const f32x4 g = ((const f32x4*)p.gamma)[i], b = ((const f32x4*)p.beta)[i];
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[k][j] - mean) * rstd * g[j] + b[j];
            if constexpr (OUT_F32) {
                ((f32x4*)((float*)p.out + (size_t)r * p.ldo))[i] = y;
            } else {
                if (p.out_lo) {  // split operand (common.h): hi here, the remainder as T or as e4m3 in a second buffer with the same row stride in bytes
                    typename T::vec4 o, lo;
                    float rem[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { elem hv; rem[j] = split_rem(y[j], hv); o[j] = hv; lo[j] = (elem)rem[j]; }
                    ((typename T::vec4*)((elem*)p.out + (size_t)r * p.ldo))[i] = o;
                    if (p.lo_mode == LO_F8) ((uint32_t*)((char*)p.out_lo + (size_t)r * p.ldo * 2))[i] = pack_lo8(rem[0], rem[1], rem[2], rem[3]);
                    else ((typename T::vec4*)((elem*)p.out_lo + (size_t)r * p.ldo))[i] = lo;
                } else {
                    typename T::vec4 o = {(elem)y[0], (elem)y[1], (elem)y[2], (elem)y[3]};
                    ((typename T::vec4*)((elem*)p.out + (size_t)r * p.ldo))[i] = o;
                }
            }
        }
    }
}

template <typename T, bool DY_F32>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs p) {
    using elem = typename T::elem;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const size_t t = p.row_index ? (size_t)p.row_index[r] : (size_t)r;
    const f32x4* x = (const f32x4*)(p.x + t * p.ldx);
    const int d4 = p.d >> 2;
    const size_t sr = p.by_token ? t : (size_t)r;  // row of dy
    const size_t st = (p.by_token || p.stats_by_token) ? t : (size_t)r;  // row of the statistics
    const float mean = p.mean[st], rstd = p.rstd[st];
    f32x4* side_row = nullptr;  // wave-uniform: this row is a spliced prompt position
    if (p.side) {
        const int pos = (int)(t % (size_t)p.side_L) - p.side_row0;
        if (pos >= 0 && pos < p.side_n) side_row = (f32x4*)(p.side + (t / (size_t)p.side_L) * p.side_ldb + (size_t)pos * p.d);
    }
    f32x4 xh[LN_MAXV], g[LN_MAXV];
    typename T::vec4 rv[LN_MAXV];  // the T residual gradient, fetched with the other operands so its latency hides behind the reduction
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + 64 * k;
        xh[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        g[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < d4) {
            const f32x4 xv = x[i];
            const f32x4 gm = ((const f32x4*)p.gamma)[i];
            if (p.dres_lp) rv[k] = ((const typename T::vec4*)((const elem*)p.dres_lp + t * p.lddres))[i];
            f32x4 dy;
            if constexpr (DY_F32) {
                dy = ((const f32x4*)((const float*)p.dy + sr * p.lddy))[i];
            } else {
                const typename T::vec4 dv = ((const typename T::vec4*)((const elem*)p.dy + sr * p.lddy))[i];
                dy = f32x4{(float)dv[0], (float)dv[1], (float)dv[2], (float)dv[3]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[k][j] = (xv[j] - mean) * rstd;
                g[k][j] = dy[j] * gm[j];
                s1 += g[k][j];
                s2 += g[k][j] * xh[k][j];
            }
        }
    }
    const float c1 = wave_sum(s1) / p.d, c2 = wave_sum(s2) / p.d;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int i = lane + 64 * k;
        if (i < d4) {
            f32x4 dx;
#pragma unroll
            for (int j = 0; j < 4; ++j) dx[j] = rstd * (g[k][j] - c1 - xh[k][j] * c2);
            if (p.dres) dx += ((const f32x4*)(p.dres + t * p.lddres))[i];
            if (p.dres_lp) {  // the gradient stream itself is kept in T (bf16 mode): no fp32 copy to read or write
                dx += f32x4{(float)rv[k][0], (float)rv[k][1], (float)rv[k][2], (float)rv[k][3]};
            }
            if (side_row) { side_row[i] = dx; dx = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (p.dx) ((f32x4*)(p.dx + t * p.lddx))[i] = dx;
            if (p.dx_lp) {
                typename T::vec4 o = {(elem)dx[0], (elem)dx[1], (elem)dx[2], (elem)dx[3]};
                ((typename T::vec4*)((elem*)p.dx_lp + t * p.lddx_lp))[i] = o;
            }
        }
    }
}

int launch_ln_fwd(int dtype, const LnFwdArgs& a, hipStream_t s, const LaunchProf* prof) {
    ARG_CHECK(a.x && a.gamma && a.beta && a.out, "ln_fwd: null operand");
    ARG_CHECK(a.rows > 0 && a.d > 0 && a.d % 4 == 0 && a.d <= 256 * LN_MAXV, "ln_fwd: bad shape rows=%d d=%d", a.rows, a.d);
    ARG_CHECK(a.ldx % 4 == 0 && a.ldo % 4 == 0 && a.ldx >= a.d && a.ldo >= a.d, "ln_fwd: bad strides %d/%d", a.ldx, a.ldo);
    const dim3 grid((a.rows + 3) / 4), block(256);
    if (dtype == DT_BF16) {
        if (a.out_f32) MUDPT_LAUNCH((ln_fwd_kernel<BF16, true>), grid, block, 0, s, prof, a);
        else MUDPT_LAUNCH((ln_fwd_kernel<BF16, false>), grid, block, 0, s, prof, a);
    } else if (dtype == DT_F16) {
        if (a.out_f32) MUDPT_LAUNCH((ln_fwd_kernel<F16, true>), grid, block, 0, s, prof, a);
        else MUDPT_LAUNCH((ln_fwd_kernel<F16, false>), grid, block, 0, s, prof, a);
    } else {
        set_error("ln_fwd: unknown dtype %d", dtype);
        return MUDPT_ERR_ARG;
    }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

int launch_ln_bwd(int dtype, const LnBwdArgs& a, hipStream_t s, const LaunchProf* prof) {
    ARG_CHECK(a.dy && a.x && a.mean && a.rstd && a.gamma && (a.dx || a.dx_lp), "ln_bwd: null operand");
    ARG_CHECK(!(a.dres && a.dres_lp), "ln_bwd: dres and dres_lp are exclusive");
    ARG_CHECK(!a.side || (!a.row_index && a.side_L > 0 && a.side_n > 0), "ln_bwd: the fused splice backward needs the identity row map");
    ARG_CHECK(a.rows > 0 && a.d > 0 && a.d % 4 == 0 && a.d <= 256 * LN_MAXV, "ln_bwd: bad shape rows=%d d=%d", a.rows, a.d);
    ARG_CHECK(a.ldx % 4 == 0 && a.lddy % 4 == 0 && a.lddx % 4 == 0, "ln_bwd: strides must be multiples of 4");
    const dim3 grid((a.rows + 3) / 4), block(256);
    if (dtype == DT_BF16) {
        if (a.dy_f32) MUDPT_LAUNCH((ln_bwd_kernel<BF16, true>), grid, block, 0, s, prof, a);
        else MUDPT_LAUNCH((ln_bwd_kernel<BF16, false>), grid, block, 0, s, prof, a);
    } else if (dtype == DT_F16) {
        if (a.dy_f32) MUDPT_LAUNCH((ln_bwd_kernel<F16, true>), grid, block, 0, s, prof, a);
        else MUDPT_LAUNCH((ln_bwd_kernel<F16, false>), grid, block, 0, s, prof, a);
    } else {
        set_error("ln_bwd: unknown dtype %d", dtype);
        return MUDPT_ERR_ARG;
    }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

}  // namespace mudpt
