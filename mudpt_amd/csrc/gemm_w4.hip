// Four-wave persistent MFMA GEMM for gfx950 (round 4 experiment; tools/gemm_bench.py --variants 0,1048576):
//   C[M,N] = A[M,K] . B[N,K]^T, 256 x 256 x 64 block tile on FOUR waves (2 x 2, 128 x 128 per wave, one wave per SIMD, up to 512 registers each).
//
// Why: gemm_pp.hip's eight waves (128 x 64 each) pull 192 KiB of fragments out of LDS per K-step on top of the 64 KiB the operand DMA writes
// into it -- 256 KiB at 128 B/clk = the 2 048 clocks the matrix pipe needs for the same K-step, i.e. an LDS port that is never idle (DESIGN.md 4.1:
// the fragment reads are worth 11 % of a launch).  A 128 x 128 wave tile reads 128 KiB.  The vendor library's kernel for these shapes has this form
// (256 x 256 x 64, 256 threads, 130 KiB of LDS).
//
// Structure: the same LDS images, swizzles, LDS-DMA staging, persistent tile walk and T-output epilogue as gemm_pp.hip; no second wave group, so the
// K-step is software-pipelined inside the wave over two register sets of fragments (k-halves of 32):
//   half 0:  ds_read fragments (s, k 32..63)           | 64 MFMA on fragments (s, k 0..31)
//   --- lgkmcnt(0), vmcnt: DMA(s + 1) has landed, ONE s_barrier per K-step: stage s is free, stage s + 1 is visible ---
//   half 1:  issue DMA(s + 2) into stage s, ds_read fragments (s + 1, k 0..31) | 64 MFMA on fragments (s, k 32..63)
#include <hip/hip_ext.h>

#include "kernels.h"

namespace mudpt {

using lptr_w4 = __attribute__((address_space(3))) void*;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_w4;

template <typename T, int EPI, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmArgs p, int ntn, int ntiles) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    static_assert(EPI == EPI_STORE, "gemm_w4: T-output store epilogue only");
    constexpr int STAGE = 65536, BOFF = 32768, NST = 32;
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int frow = lane & 15, fq = lane >> 4;

    const int nkt = p.K >> 6;
    const int G = gridDim.x;
    const int my = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int total = my * nkt;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((size_t)p.M * p.lda * 2 < 0xffffffffull ? (size_t)p.M * p.lda * 2 : 0xffffffffull), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)((size_t)p.N * p.ldb * 2 < 0xffffffffull ? (size_t)p.N * p.ldb * 2 : 0xffffffffull), 0x00020000);
    const auto rsOut0 = __builtin_amdgcn_make_buffer_rsrc(p.out0, 0, p.M * p.ldo0 * 2, 0x00020000);
    const auto rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    bool epi_pending = false;

    // ---- DMA: wave w fills row groups 8 w .. 8 w + 7 (8 rows = 1 KiB of LDS each) of the A image and of the B image -----------------
    // A image: chunk c of row r sits in slot c ^ (r & 7); B image (rows permuted for the T epilogue, see gemm_pp.hip): slot c ^ ((r & 3) | ((r >> 4) & 1) << 2)
    // Row group q of wave w starts at row 64 w + 8 q: the lane part of its source offset is that of group 0 plus q x 8 rows -- a uniform
    // step added at the issue (one VALU add in an MFMA's shadow) instead of 16 long-lived registers; the B swizzle has bit 4 of the row
    // in it, so groups with (q >> 1) odd use a second lane constant.
    const int srow = lane >> 3, sslot = lane & 7;
    const int row0 = w * 64 + srow;
    const int offA0 = (row0 * p.lda + ((sslot ^ (row0 & 7)) << 3)) * 2;
    const int offB0 = (row0 * p.ldb + ((sslot ^ (row0 & 3)) << 3)) * 2, offB1 = (row0 * p.ldb + ((sslot ^ ((row0 & 3) | 4)) << 3)) * 2;
    const int stepA = 8 * p.lda * 2, stepB = 8 * p.ldb * 2;

    const int GN = ntn >= 12 ? 6 : (ntn >= 6 && ntn % 3 == 0 ? 3 : ntn);
    const int ntm = ntiles / ntn;
    auto tile_mn = [&](int tile, int& tm, int& tn) {
        const int tpg = ntm * GN, full = ntn / GN, ng = tile / tpg;
        if (ng < full) {
            const int rem = tile - ng * tpg;
            tm = rem / GN;
            tn = ng * GN + (rem - tm * GN);
        } else {
            const int wl = ntn - full * GN, rem = tile - full * tpg;
            tm = rem / wl;
            tn = full * GN + (rem - tm * wl);
        }
    };
    auto item_tile = [&](int it, int& tm, int& tn) { tile_mn(xcd_remap((int)blockIdx.x + it * G, ntiles), tm, tn); };

    int n_kt = 0, n_it = 0, n_soff = 0;
    int voA = 0, voB0 = 0, voB1 = 0;  // vector offsets of the tile being prefetched: lane part + tile base (bounds-checked; the K offset rides in soffset)
    auto set_next_item = [&](int it) {
        int tm, tn;
        item_tile(it, tm, tn);
        const int baseA = tm * 256 * p.lda * 2, baseB = tn * 256 * p.ldb * 2;
        voA = offA0 + baseA; voB0 = offB0 + baseB; voB1 = offB1 + baseB;
    };
    auto advance_next = [&]() {  // only called while another step exists
        n_soff += 128;
        if (++n_kt == nkt) {
            n_kt = 0; n_soff = 0;
            set_next_item(++n_it);
        }
    };
    auto issue = [&](int stage) {
        if constexpr (ABL & 2) return;
        char* dst = smem + stage * STAGE + w * 8192;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_w4)(dst + q * 1024), 16, voA + q * stepA, n_soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_w4)(dst + BOFF + q * 1024), 16, ((q >> 1) & 1 ? voB1 : voB0) + q * stepB, n_soff, 0, 0);
        }
    };

    // ---- fragment reads --------------------------------------------------------------------------------------------------------------
    const int swA = frow & 7, swB = (frow & 3) | (((frow >> 2) & 1) << 2);
    const int a_base = (wr * 128 + frow) * 128;
    const int b_base = BOFF + (wc * 128 + (frow >> 2) * 16 + (frow & 3)) * 128;
    auto read_frags = [&](const char* st, int h, vec8 (&af)[8], vec8 (&bf)[8]) {
        if constexpr (ABL & 1) return;
        const int ca = ((h * 4 + fq) ^ swA) << 4, cb = ((h * 4 + fq) ^ swB) << 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *(const vec8*)(st + a_base + i * 2048 + ca);
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[j] = *(const vec8*)(st + b_base + (j >> 2) * 8192 + (j & 3) * 512 + cb);
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma64 = [&](const vec8 (&af)[8], const vec8 (&bf)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = T::mfma16(bf[j], af[i], acc[i][j]);
    };

    vec8 ax[8], bx[8], ay[8], by[8];
    if constexpr (ABL & 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ax[i][e] = (elem)(float)(lane + e); bx[i][e] = (elem)(float)(lane - e); ay[i][e] = (elem)(float)(lane * e); by[i][e] = (elem)(float)(i + e); }
    }

    // ---- prologue ----------------------------------------------------------------------------------------------------------------------
    // n_* always describe a VALID step: once the block's last step has been issued it is issued again (into a stage that is free) instead of
    // branching around the issue -- the halves of a K-step stay single basic blocks the scheduler can interleave.
    int n_idx = 0;
    auto issue_and_advance = [&](int stage) {
        issue(stage);
        if (n_idx + 1 < total) { ++n_idx; advance_next(); }
    };
    set_next_item(0);
    issue_and_advance(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(smem, 0, ax, bx);
    issue_and_advance(1);

    int s = 0, c_it = 0;
    for (int it = 0; it < my; ++it) {
        for (int k = 0; k < nkt; ++k) {
            const int cur = s & 1;
            const char* st = smem + cur * STAGE;
            // ---- half 0 ----
            read_frags(st, 1, ay, by);
            mfma64(ax, bx);
            // one LDS read in the shadow of every MFMA while there are reads (a lone wave per SIMD: nobody else fills the matrix pipe)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave is done reading stage s
            // DMA(s + 1) has landed (it was issued before the last epilogue's stores)
            if (epi_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            epi_pending = false;
            if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- half 1 ----
            issue(cur);
            read_frags(smem + (cur ^ 1) * STAGE, 0, ax, bx);
            mfma64(ay, by);
#pragma unroll
            for (int g = 0; g < 16; ++g) {  // the DMA writes LDS: the compiler keeps every fragment read behind all of them
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (n_idx + 1 < total) { ++n_idx; advance_next(); }
            ++s;
        }
        // ---- epilogue: lane owns out[m][n .. n + 15] for m = sub-tile row (lane & 15), per 64-column group -----------------------------
        int tm, tn;
        item_tile(c_it++, tm, tn);
        const int m_base = tm * 256 + wr * 128 + frow;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            const int n = tn * 256 + wc * 128 + gq * 64 + fq * 16;
            f32x4 bias4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bias4[j] = p.bias ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBias, (n + 4 * j) * 4, 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                vec8 o0, o1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o0[c] = round_to<elem>(acc[i][gq * 4 + 0][c] + bias4[0][c]); o0[4 + c] = round_to<elem>(acc[i][gq * 4 + 1][c] + bias4[1][c]);
                    o1[c] = round_to<elem>(acc[i][gq * 4 + 2][c] + bias4[2][c]); o1[4 + c] = round_to<elem>(acc[i][gq * 4 + 3][c] + bias4[3][c]);
                }
                const int off = (n < p.N ? (m_base * p.ldo0 + n) * 2 : OOB) + i * (16 * p.ldo0 * 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_w4, o0), rsOut0, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_w4, o1), rsOut0, off, 16, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][gq * 4 + j] = f32x4{0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_sched_barrier(0);  // or all 256 accumulators are read out at once and the loop's registers are spilled around it
            }
        }
        epi_pending = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-issued last step must not outlive the workgroup's LDS
}

template <typename T, int EPI, int ABL = 0>
static int launch_w4(const GemmArgs& a, hipStream_t s, const GemmOpts& o) {
    constexpr int lds = 2 * 65536;
    auto kern = gemm_w4_kernel<T, EPI, ABL>;
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipDeviceGetAttribute(&pd.ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        pd.done[dev] = true;
    }
    const int ncu = pd.ncu[dev];
    const int ntm = (a.M + 255) / 256, ntn = (a.N + 255) / 256, ntiles = ntm * ntn;
    const int grid = ntiles < ncu ? ntiles : ncu;
    if (o.ev_start && o.ev_stop) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, o.ev_start, o.ev_stop, 0, a, ntn, ntiles);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, ntn, ntiles);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// Arguments are validated by launch_gemm (gemm.hip) before it dispatches here.
int launch_gemm_w4(int dtype, int epi, const GemmArgs& a, hipStream_t s, const GemmOpts& o) {
    ARG_CHECK(epi == EPI_STORE && a.lo_mode == LO_NONE, "gemm_w4: store epilogue, no split operand");
    ARG_CHECK((size_t)a.M * a.lda * 2 < 0xffffffffull && (size_t)a.N * a.ldb * 2 < 0xffffffffull, "gemm_w4: operand larger than 4 GiB");
    ARG_CHECK(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldo0 % 8 == 0, "gemm_w4: strides must be multiples of 8");
    ARG_CHECK((size_t)a.M * a.ldo0 * 2 < 0x7fffffffull, "gemm_w4: output larger than 2 GiB");
    if (dtype == DT_BF16) {
        switch ((a.flags >> 4) & 7) {  // timing-only ablations (knob bits 10, 11, 21)
            case 1: return launch_w4<BF16, EPI_STORE, 1>(a, s, o);
            case 2: return launch_w4<BF16, EPI_STORE, 2>(a, s, o);
            case 3: return launch_w4<BF16, EPI_STORE, 3>(a, s, o);
            case 4: return launch_w4<BF16, EPI_STORE, 4>(a, s, o);
            case 7: return launch_w4<BF16, EPI_STORE, 7>(a, s, o);
        }
        return launch_w4<BF16, EPI_STORE>(a, s, o);
    }
    if (dtype == DT_F16) return launch_w4<F16, EPI_STORE>(a, s, o);
    set_error("gemm: unknown dtype %d", dtype);
    return MUDPT_ERR_ARG;
}

}  // namespace mudpt
