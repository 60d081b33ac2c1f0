// Persistent ping-pong MFMA GEMM for gfx950 (the large-M GEMMs of the vision tower):
//   C[M,N] = A[M,K] . B[N,K]^T, 256 x 256 x 64 block tile, 8 waves (2 along M x 4 along N, 128 x 64 per wave),
//   v_mfma_f32_16x16x32, fp32 accumulate, the same fused epilogues as gemm.hip.
//
// Structure (one workgroup per CU, grid = min(tiles, CUs), each block walks tiles b, b + grid, ...):
//  * K-step = 4 phases, one 64 x 32 accumulator quadrant each:  [ds_read fragments | issue one 16 KiB operand
//    unit of the NEXT K-step with buffer_load ... lds | counted vmcnt] s_barrier [16 MFMA] s_barrier.
//    Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section while its
//    partner reads LDS / issues DMA: the matrix pipe alternates between the two and never waits on LDS.
//  * Operands go global -> LDS by LDS-DMA (no VGPR round trip), two 64 KiB stages.  The four units of a K-step are
//    ordered by first use -- A rows of the first quadrant pair, B rows of sub-tiles 0-1, B rows of sub-tiles 2-3,
//    A rows of the second pair -- and each stays >= 2 phases in flight behind a counted s_waitcnt vmcnt(4); a unit
//    is read one phase after the wait that retires it (two wave groups are a barrier apart).
//  * The prefetch runs across output tiles: the first K-step of the next tile streams in during the epilogue.
//  * LDS image [rows][64] (128-byte rows), XOR swizzle on the DMA source address and on the ds_read_b128 address,
//    conflict-free for the A reads and for the permuted B reads.  B fragment rows are permuted so that a lane
//    ends up with 16 CONSECUTIVE output columns of one row: the epilogue moves 32 (T) / 64 (fp32) contiguous bytes
//    per lane and 128 / 256 contiguous bytes per row per store instruction.
//  * Buffer descriptors bound-check the operand reads (rows >= M read as zero): no clamping, ragged M is free.
#include <hip/hip_ext.h>

#include <type_traits>

#include "kernels.h"

namespace mudpt {

using lptr_t = __attribute__((address_space(3))) void*;

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// s_waitcnt vmcnt(N): at most N of this wave's vector-memory operations (DMA, loads AND stores, in issue order)
// may still be outstanding.
template <int N>
__device__ inline void vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate is 6 bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ST / LD: cache policy of the epilogue's output stores / of EPI_GELU_BWD's one-touch aux read (aux bits of the buffer instructions on
// gfx950: 1 = sc0, 2 = nt, 16 = sc1).  Measured in round 2 (tools/gemm_bench.py, A/B in one process, B = 256 shapes): write-through
// sc1 / sc0 sc1 stores are 8-60 % SLOWER (fc 315 -> 515 us: the L2 no longer absorbs the store bursts), nt stores and nt aux loads
// are within +-1 % over the 8 GEMMs of a block.  Counters with nt stores over the whole step (round 2, separate --pmc passes): fc's HBM-side reads 412 -> 332 MB per launch, but its
// writes 632 -> 900 MB (partial lines are no longer combined in the L2) and the step 25.10 -> 25.38 ms.  Plain policy everywhere; the parameters stay for the next experiment.
constexpr int EPI_GELU_SPLIT = 100;   // EPI_GELU with out1 as a split operand: out1 = hi(gelu(u)), out1_lo = the remainder as T (common.h LO_F16)
constexpr int EPI_GELU_SPLIT8 = 101;  // ... the remainder as e4m3 bytes (LO_F8)
constexpr int EPI_GELU_Q8 = 102;      // EPI_GELU with out0 = 8-bit codes of QuickGELU'(u) instead of u (GemmArgs::gelu_q8, common.h)
constexpr int EPI_GELU_BWD_Q8 = 103;  // EPI_GELU_BWD reading such codes from aux

// Split A operand (common.h LoMode; GemmArgs::A_lo): a tile's K loop is two passes -- K / 64 steps of A against B, then the low half:
// LO_F16: K / 64 more steps of A_lo against the same B (fp16 MFMA; the text tower, which needs all 22 bits);
// LO_F8 (template F8): K / 128 steps of e4m3 bytes, A_lo against B8, on v_mfma_scale_f32_16x16x128_f8f6f4 -- a K-step is again 128 bytes per row and
// 64 KiB per stage, 32 matrix instructions of twice the length per wave: the same pipe time, DMA volume, LDS image, swizzle and vmcnt
// accounting as an fp16 step.  Both low buffers have the row stride of their T counterparts in bytes, so only the buffer descriptors change
// between the passes.
// ABL: timing-only ablations of the K loop (WRONG results; tools/gemm_bench.py --variants 1024,2048,3072): bit 0 = no fragment reads from LDS
// (the MFMAs run on whatever the registers hold), bit 1 = no operand DMA.  Where the K-step's 1.56 us go: DESIGN.md 4.1.
template <typename T, int EPI, bool F8 = false, int ST = 0, int LD = 0, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs p, int ntn, int ntiles, int rem_half) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int STAGE = 65536, BOFF = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int frow = lane & 15, fq = lane >> 4;

    const int nkt1 = p.K >> 6;                                                       // K-steps of the first pass
    const int nkt = nkt1 + (p.lo_mode == LO_NONE ? 0 : (F8 ? nkt1 >> 1 : nkt1));     // ... of both passes of a tile
    const int G = gridDim.x;
    // Work items of this block: my_full whole 256 x 256 tiles (b, b + G, ...), then -- when the host split the last, partial
    // wave of tiles (rem_half of them, 2 rem_half <= G) -- one HALF tile of 128 rows x 256 columns: blocks b and b + rem_half
    // take the upper / lower half of remainder tile b, so the tail costs half a tile time instead of a whole one
    // (603 tiles on 256 CUs: 2.5 instead of 3 tile times).
    const int my_full = rem_half ? ntiles / G : (ntiles - (int)blockIdx.x + G - 1) / G;
    const bool has_half = rem_half && (int)blockIdx.x < 2 * rem_half;
    const int total = (my_full + (has_half ? 1 : 0)) * nkt;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((size_t)p.M * p.lda * 2 < 0xffffffffull ? (size_t)p.M * p.lda * 2 : 0xffffffffull), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)((size_t)p.N * p.ldb * 2 < 0xffffffffull ? (size_t)p.N * p.ldb * 2 : 0xffffffffull), 0x00020000);
    // second pass: the low half of A; B again or its e4m3 copy (same extents in bytes)
    const auto rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A_lo ? p.A_lo : p.A), 0, (int)((size_t)p.M * p.lda * 2 < 0xffffffffull ? (size_t)p.M * p.lda * 2 : 0xffffffffull), 0x00020000);
    const auto rsB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(F8 ? p.B8 : p.B), 0, (int)((size_t)p.N * p.ldb * 2 < 0xffffffffull ? (size_t)p.N * p.ldb * 2 : 0xffffffffull), 0x00020000);

    // epilogue operands through bounds-checked descriptors as well (out-of-range lanes get offset OOB: dropped / read 0)
    static_assert(EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_GELU_SPLIT || EPI == EPI_GELU_SPLIT8 || EPI == EPI_GELU_BWD || EPI == EPI_STORE_F32 ||
                  EPI == EPI_GELU_Q8 || EPI == EPI_GELU_BWD_Q8, "epilogue not built for gemm_pp");
    constexpr bool GELU = (EPI == EPI_GELU || EPI == EPI_GELU_SPLIT || EPI == EPI_GELU_SPLIT8 || EPI == EPI_GELU_Q8);
    constexpr int OOB = (int)0x80000000;
    constexpr bool OUT_F32 = (EPI == EPI_STORE_F32);
    // B fragment rows: permuted (a lane ends up with 16 consecutive columns = 32 bytes of T) for the T outputs, natural
    // (4 consecutive columns per sub-tile = 16 bytes of fp32) for the fp32 output; see the epilogue.
    constexpr bool NAT = OUT_F32;
    constexpr int NST = (EPI == EPI_STORE || EPI == EPI_GELU_BWD || EPI == EPI_GELU_BWD_Q8) ? 16 : (EPI == EPI_GELU_SPLIT ? 48 : (EPI == EPI_GELU_SPLIT8 ? 40 : (EPI == EPI_GELU_Q8 ? 24 : 32)));  // stores per wave per tile, exact
    const auto rsOut0 = __builtin_amdgcn_make_buffer_rsrc(p.out0, 0, p.M * p.ldo0 * (OUT_F32 ? 4 : (EPI == EPI_GELU_Q8 ? 1 : 2)), 0x00020000);
    const auto rsOut1 = __builtin_amdgcn_make_buffer_rsrc(p.out1, 0, GELU ? p.M * p.ldo1 * 2 : 0, 0x00020000);
    const auto rsOut1Lo = __builtin_amdgcn_make_buffer_rsrc(p.out1_lo, 0, (EPI == EPI_GELU_SPLIT || EPI == EPI_GELU_SPLIT8) ? p.M * p.ldo1 * 2 : 0, 0x00020000);
    const auto rsAux = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.aux), 0, EPI == EPI_GELU_BWD ? p.M * p.ldaux * 2 : (EPI == EPI_GELU_BWD_Q8 ? p.M * p.ldaux : 0), 0x00020000);
    const auto rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    bool epi_pending = false;  // the previous step ended with an epilogue: NST stores sit in the VMEM queue

    // ---- DMA bookkeeping: 4 units x 2 row-groups (8 rows, 1 KiB in LDS) per wave --------------------------------
    // unit 0: A row-groups {0..7, 16..23} (rows of quadrant pair 0 of both wave rows); unit 3: A {8..15, 24..31}
    // unit 1: B even row-groups (fragment sub-tiles 0, 1);                           unit 2: B odd row-groups
    int grp[4][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = 2 * w + q;
        grp[0][q] = k < 8 ? k : k + 8;
        grp[3][q] = k < 8 ? k + 8 : k + 16;
        // permuted B: sub-tiles 0, 1 read the even 8-row groups; natural B: the first four groups of every 64 rows
        grp[1][q] = NAT ? (k >> 2) * 8 + (k & 3) : 2 * k;
        grp[2][q] = NAT ? (k >> 2) * 8 + (k & 3) + 4 : 2 * k + 1;
    }
    const int srow = lane >> 3, sslot = lane & 7;
    // lane part of the source offset (elements): row * ld + chunk * 8, chunk = slot ^ swizzle(row)
    int lane_off[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool isA = (u == 0 || u == 3);
            const int row = grp[u][q] * 8 + srow;
            const int f = (isA || NAT) ? (row & 7) : ((row & 3) | (((row >> 4) & 1) << 2));
            lane_off[u][q] = (row * (isA ? p.lda : p.ldb) + ((sslot ^ f) << 3)) * 2;  // bytes
        }

    // Tile order inside an XCD's contiguous chunk: groups of GN column tiles, all row panels of a group before the next
    // group, column fastest.  The ~32 tiles an XCD runs at once then cover ~32/GN row panels x GN weight tiles, so the GN
    // weight tiles (GN x 393 KB at K = 768) stay in the 4 MiB L2 for the whole sweep over M instead of all N/256 tiles
    // thrashing it.  Measured (A/B in one process): N = 3072 (12 column tiles) GN = 6: -4 %; N = 2304 (9) GN = 3: -5 %;
    // N = 768 (3): one group.
    const int gn_knob = (p.flags >> 8) & 0xff;
    const int GN = gn_knob > 0 ? gn_knob : (ntn >= 12 ? 6 : (ntn >= 6 && ntn % 3 == 0 ? 3 : ntn));
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
    // tile (and half: 0 upper / 1 lower 128 rows) of this block's it-th work item
    auto item_tile = [&](int it, int& tm, int& tn, int& h) {
        if (it < my_full) {
            tile_mn(xcd_remap((int)blockIdx.x + it * G, ntiles), tm, tn);
            h = 0;
        } else {
            h = (int)blockIdx.x >= rem_half ? 1 : 0;
            tile_mn(xcd_remap(my_full * G + (int)blockIdx.x - h * rem_half, ntiles), tm, tn);
        }
    };
    // The step being prefetched ("next"): k-tile n_kt of this block's n_it-th item; all scalar, updated incrementally
    // (no division in the loop: the DMA issue sits in the read section that must stay shorter than 16 MFMAs).
    int n_kt = 0, n_it = 0, n_baseA = 0, n_baseB = 0, n_soff = 0;  // n_soff: byte offset of the step's 128 bytes inside its pass's rows
    bool n_second = false;  // the step being prefetched belongs to the second pass (low half of A): the *2 descriptors
    bool n_half = false;  // the item being prefetched is the half tile: its 128 A rows are unit 0 alone, unit 3 is not issued
    auto set_next_item = [&](int it) {
        int tm, tn, h;
        item_tile(it, tm, tn, h);
        n_half = it >= my_full;
        // bytes (operands < 4 GiB are checked on the host).  Unit 0's second half (waves 4-7) lands in LDS rows 128-191 and
        // normally comes from tile rows 128-191; in a half tile it is rows 64-127: 64 rows less.
        n_baseA = (tm * 256 + (n_half ? h * 128 - (w >= 4 ? 64 : 0) : 0)) * p.lda * 2;
        n_baseB = tn * 256 * p.ldb * 2;
    };
    auto advance_next = [&]() {
        n_soff += 128;
        if (++n_kt == nkt) {
            n_kt = 0; n_soff = 0; n_second = false;
            if ((++n_it) * nkt < total) set_next_item(n_it);
        } else if (n_kt == nkt1) {
            n_soff = 0; n_second = true;
        }
    };
    // issue unit u of the next step into LDS stage `stage`
    auto issue = [&](int u, int stage) {
        if constexpr (ABL & 2) return;
        const bool isA = (u == 0 || u == 3);
        const int base = isA ? n_baseA : n_baseB;
        char* dst = smem + stage * STAGE + (isA ? 0 : BOFF);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? (n_second ? rsA2 : rsA) : (n_second ? rsB2 : rsB), (lptr_t)(dst + grp[u][q] * 1024), 16, lane_off[u][q] + base, n_soff, 0, 0);
    };

    // ---- fragment read offsets -----------------------------------------------------------------------------------
    const int sw = frow & 7;
    const int c0 = (fq ^ sw) << 4;           // k-step 0 chunk byte offset; k-step 1 is c0 ^ 64
    const int c8 = ((2 * fq) ^ sw) << 4;     // e4m3 step: the lane's 32 bytes k = 32 fq .. 32 fq + 31 are chunks 2 fq (c8) and 2 fq + 1 (c8 ^ 16)
    int ca = c0, cb = c0 ^ 64;               // the two chunk offsets of the step being computed
    const int a_base = (wr * 128 + frow) * 128;
    const int b_base = BOFF + (NAT ? wc * 64 + frow : wc * 64 + (frow >> 2) * 16 + (frow & 3)) * 128;
    constexpr int BJ = NAT ? 2048 : 512;  // byte step between the B fragments of consecutive sub-tiles

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: step 0 completely, then offset the second wave group by one barrier ----------------------------
    if (total > 0) {
        set_next_item(0);
        issue(0, 0); issue(1, 0); issue(2, 0);
        if (!n_half) issue(3, 0);
        advance_next();
    }
    VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    // A fragments of the current quadrant pair, B fragments of all 4 sub-tiles: 32 bytes each = the two k-steps of an fp16 step (halves 0 / 1)
    // or the ONE 8-register operand of an e4m3 step -- kept as 8-register tuples so that the latter needs no copies
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    i32x8 af[4], bf[4];
    if constexpr (ABL & 1) {  // defined operands for the timing-only build
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[i] = i32x8{lane, 1, 2, 3, 4, 5, 6, 7} * 0x3c003c00; bf[i] = i32x8{7, 6, 5, 4, 3, 2, 1, lane} * 0x3c003c00; }
    }
    auto half = [](const i32x8& v, int ks) -> vec8 {
        return __builtin_bit_cast(vec8, ks ? __builtin_shufflevector(v, v, 4, 5, 6, 7) : __builtin_shufflevector(v, v, 0, 1, 2, 3));
    };
    int c_it = 0;              // the tile being computed: this block's c_it-th item

    // 16 MFMAs (8 of twice the length in an e4m3 step) of one accumulator quadrant: rows 16 (i0 + i), sub-tiles j0, j0 + 1
    auto mfma_quadrant = [&](auto f8tag, int i0, int j0) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (decltype(f8tag)::value) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[j0 + j], af[i], acc[i0 + i][j0 + j], 0, 0, 0, p.b8_scale, 0, LO8_SCALE_E8M0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i0 + i][j0 + j] = T::mfma16(half(bf[j0 + j], ks), half(af[i], ks), acc[i0 + i][j0 + j]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_a = [&](const char* st, int i0) {
        if constexpr (ABL & 1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            af[i] = __builtin_shufflevector(*(const i32x4*)(st + a_base + (i0 + i) * 2048 + ca), *(const i32x4*)(st + a_base + (i0 + i) * 2048 + cb), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto read_b = [&](const char* st, int j0) {
        if constexpr (ABL & 1) return;
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j)
            bf[j] = __builtin_shufflevector(*(const i32x4*)(st + b_base + j * BJ + ca), *(const i32x4*)(st + b_base + j * BJ + cb), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto epilogue = [&]() {
            // ---- epilogue of this output tile: lane owns out[m][n0 .. n0 + 15], m = sub-tile row (lane & 15) ----
            // Branch-free: buffer descriptors drop out-of-range lanes (ragged M, N edge), so every wave issues exactly
            // NST stores.  They are NOT waited for here: the next tile's main loop runs while they drain, and its first
            // two counted waits allow NST more operations in flight (epi_pending).
            int tm, tn, h;
            const bool c_half = c_it >= my_full;  // a half tile: rows h * 128 + wr * 64 + 16 i, i < 4
            item_tile(c_it++, tm, tn, h);
            const int ni = c_half ? 4 : 8;
            // T outputs: lane owns columns n .. n + 15 of row m (two 16-byte pieces, 16 bytes apart);
            // fp32 output: lane owns columns n + 16 j .. + 3 of row m for j = 0..3 (pieces of a pair (2J, 2J+1) are 64 bytes apart)
            const int n = tn * 256 + wc * 64 + (NAT ? fq * 4 : fq * 16);
            f32x4 bias4[4];  // the bias descriptor ends at N: columns beyond it read 0
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bias4[j] = p.bias ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBias, (n + (NAT ? 16 : 4) * j) * 4, 0, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int m_base = tm * 256 + (c_half ? h * 128 + wr * 64 : wr * 128) + frow;
            // byte offset of column `col` in row (m_base + 16 i) of a [rows][ld] array.  Rows >= M lie beyond the descriptor's
            // range (dropped / read as 0); a column beyond N moves the lane out of range.
            auto row_off = [&](int i, int ld, int esz, int col) { return (col < p.N ? (m_base * ld + col) * esz : OOB) + i * (16 * ld * esz); };
            // ---- pass 1: everything that needs a LOAD is folded into the accumulators, all loads before any store
            // (vmcnt retires in issue order: a load waited for behind a store also waits for that store's completion)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += bias4[j];
            if constexpr (EPI == EPI_GELU_BWD) {
                vec8 u[8][2];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int off = i < ni ? row_off(i, p.ldaux, 2, n) : OOB;
                    u[i][0] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rsAux, off, 0, LD));
                    u[i][1] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(rsAux, off, 16, LD));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[i][j][c] *= quick_gelu_grad((float)u[i][j >> 1][4 * (j & 1) + c]);
            }
            if constexpr (EPI == EPI_GELU_BWD_Q8) {
                u32x4 q[8];  // 16 byte codes: this lane's 16 columns of row i
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = __builtin_amdgcn_raw_buffer_load_b128(rsAux, i < ni ? row_off(i, p.ldaux, 1, n) : OOB, 0, LD);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[i][j][c] *= gelu_grad_from_q8(q[i][j], c);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- pass 2: stores only (exactly NST per wave).  One CU drains ~16 bytes per clock (tools/probes/store_pattern.hip:
            // 31-33 GB/s per CU whatever the lane -> address map), so a 128 KiB tile costs ~4 us that nothing overlaps.
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < ni) {  // uniform: a half tile (always the block's last item) stores 4 of the 8 sub-tile rows
                if constexpr (OUT_F32) {
                    // natural layout: the 4 lanes of a row write 64 contiguous bytes per instruction (permuted: 16-byte pieces
                    // 64 bytes apart, measured 1.55x slower)
#pragma unroll
                    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsOut0, row_off(i, p.ldo0, 4, n + 16 * j), 0, ST);
                } else {
                    vec8 o0, o1;
                    if constexpr (EPI == EPI_GELU_Q8) {  // 16 byte codes of QuickGELU'(u): one 16-byte store instead of u's two
                        u32x4 q;
#pragma unroll
                        for (int j = 0; j < 4; ++j) q[j] = gelu_grad_q8x4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                        __builtin_amdgcn_raw_buffer_store_b128(q, rsOut0, row_off(i, p.ldo0, 1, n), 0, ST);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {  // round_to: the conversion rounds the fp32 value (no fused multiply-convert; common.h)
                            o0[c] = round_to<elem>(acc[i][0][c]); o0[4 + c] = round_to<elem>(acc[i][1][c]);
                            o1[c] = round_to<elem>(acc[i][2][c]); o1[4 + c] = round_to<elem>(acc[i][3][c]);
                        }
                        const int off = row_off(i, p.ldo0, 2, n);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rsOut0, off, 0, ST);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rsOut0, off, 16, ST);
                    }
                    if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_Q8) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            o0[c] = round_to<elem>(quick_gelu(acc[i][0][c])); o0[4 + c] = round_to<elem>(quick_gelu(acc[i][1][c]));
                            o1[c] = round_to<elem>(quick_gelu(acc[i][2][c])); o1[4 + c] = round_to<elem>(quick_gelu(acc[i][3][c]));
                        }
                        const int off1 = row_off(i, p.ldo1, 2, n);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rsOut1, off1, 0, ST);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rsOut1, off1, 16, ST);
                    }
                    if constexpr (EPI == EPI_GELU_SPLIT8) {
                        u32x4 l8;
                        float rem[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                elem hv;
                                rem[c] = split_rem(quick_gelu(acc[i][j][c]), hv);
                                if (j < 2) o0[4 * j + c] = hv; else o1[4 * (j - 2) + c] = hv;
                            }
                            l8[j] = pack_lo8(rem[0], rem[1], rem[2], rem[3]);
                        }
                        const int off1 = row_off(i, p.ldo1, 2, n);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rsOut1, off1, 0, ST);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rsOut1, off1, 16, ST);
                        // e4m3 remainders: byte column n of a row whose stride in bytes is that of out1 (2 ldo1)
                        __builtin_amdgcn_raw_buffer_store_b128(l8, rsOut1Lo, (n < p.N ? m_base * p.ldo1 * 2 + n : OOB) + i * (32 * p.ldo1), 0, ST);
                    }
                    if constexpr (EPI == EPI_GELU_SPLIT) {
                        vec8 l0, l1;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            elem hv, lv;
                            split_hi_lo(quick_gelu(acc[i][0][c]), hv, lv); o0[c] = hv; l0[c] = lv;
                            split_hi_lo(quick_gelu(acc[i][1][c]), hv, lv); o0[4 + c] = hv; l0[4 + c] = lv;
                            split_hi_lo(quick_gelu(acc[i][2][c]), hv, lv); o1[c] = hv; l1[c] = lv;
                            split_hi_lo(quick_gelu(acc[i][3][c]), hv, lv); o1[4 + c] = hv; l1[4 + c] = lv;
                        }
                        const int off1 = row_off(i, p.ldo1, 2, n);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rsOut1, off1, 0, ST);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rsOut1, off1, 16, ST);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, l0), rsOut1Lo, off1, 0, ST);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, l1), rsOut1Lo, off1, 16, ST);
                    }
                }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            epi_pending = true;
    };

    // One K-step of a full tile (4 phases) / of the half tile (2 phases), in two flavours: an fp16 step and an e4m3 step of the second pass.
    // The two flavours are SEPARATE loops per tile (first pass, then second pass), not one loop with a branch around the MFMAs: a diamond
    // around an in-place accumulation made the compiler give every quadrant a second set of accumulator registers (+34 VGPRs, copies).
    int s = 0;  // the block's running K-step count (stage = s & 1)
    auto full_step = [&](auto f8tag) {
        constexpr bool F8S = decltype(f8tag)::value;
        const int cur = s & 1;
        const char* st = smem + cur * STAGE;
        const bool nxt = s + 1 < total;
        ca = F8S ? c8 : c0;
        cb = F8S ? (c8 ^ 16) : (c0 ^ 64);
        // ================= phase 0: quadrant (rows 0-63, sub-tiles 0-1) =================
        read_a(st, 0);
        read_b(st, 0);
        // retire unit 2 of this step (read in phase 1): younger operations = unit 3 [2] (+ the last epilogue's stores) (+ unit 0' [2])
        if (nxt) {
            issue(0, cur ^ 1);
            if (epi_pending) vmcnt<NST + 4>(); else vmcnt<4>();
        } else {
            if (epi_pending) vmcnt<NST + 2>(); else vmcnt<2>();
        }
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(f8tag, 0, 0);
        __builtin_amdgcn_s_barrier();
        // ================= phase 1: quadrant (rows 0-63, sub-tiles 2-3) =================
        read_b(st, 2);
        // retire unit 3 of this step (read in phase 2): younger = (stores) + unit 0' [2] + unit 1' [2]
        if (nxt) {
            issue(1, cur ^ 1);
            if (epi_pending) vmcnt<NST + 4>(); else vmcnt<4>();
        } else {
            if (epi_pending) vmcnt<NST>(); else vmcnt<0>();
        }
        epi_pending = false;  // phase 3's wait (units 0', 1' are younger than the stores) covers the stores themselves
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(f8tag, 0, 2);
        __builtin_amdgcn_s_barrier();
        // ================= phase 2: quadrant (rows 64-127, sub-tiles 2-3) =================
        read_a(st, 4);
        if (nxt) issue(2, cur ^ 1);
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(f8tag, 4, 2);
        __builtin_amdgcn_s_barrier();
        // ================= phase 3: quadrant (rows 64-127, sub-tiles 0-1) =================
        // retires units 0, 1 of the next step (read in its phase 0); a half tile has no unit 3
        if (nxt) {
            if (!n_half) { issue(3, cur ^ 1); advance_next(); VMCNT(4); } else { advance_next(); VMCNT(2); }
        }
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(f8tag, 4, 0);
        __builtin_amdgcn_s_barrier();
        ++s;
    };
    auto half_step = [&](auto f8tag) {
        constexpr bool F8S = decltype(f8tag)::value;
        const int cur = s & 1;
        const char* st = smem + cur * STAGE;
        const bool nxt = s + 1 < total;
        ca = F8S ? c8 : c0;
        cb = F8S ? (c8 ^ 16) : (c0 ^ 64);
        // ---- a K-step of the half tile: 64 rows per wave, two phases, units 0 (A), 1, 2 (B) only ----
        // ================= phase 0: (rows 0-63, sub-tiles 0-1) =================
        read_a(st, 0);
        read_b(st, 0);
        // retire unit 2 of this step (read in phase 1): younger = (the last full tile's stores) + units 0', 1' [4]
        if (nxt) {
            issue(0, cur ^ 1);
            issue(1, cur ^ 1);
            if (epi_pending) vmcnt<NST + 4>(); else vmcnt<4>();
        } else {
            if (epi_pending) vmcnt<NST>(); else vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(f8tag, 0, 0);
        __builtin_amdgcn_s_barrier();
        // ================= phase 1: (rows 0-63, sub-tiles 2-3) =================
        read_b(st, 2);
        if (nxt) { issue(2, cur ^ 1); advance_next(); VMCNT(2); }  // retires units 0', 1' (and any stores before them)
        epi_pending = false;
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(f8tag, 0, 2);
        __builtin_amdgcn_s_barrier();
        ++s;
    };
    const int nkt2 = nkt - nkt1;  // K-steps of the second pass (0 without a split operand)
    for (int it = 0; it < my_full; ++it) {
        if constexpr (F8) {
            for (int k = 0; k < nkt1; ++k) full_step(std::false_type{});
            for (int k = 0; k < nkt2; ++k) full_step(std::true_type{});
        } else {
            for (int k = 0; k < nkt; ++k) full_step(std::false_type{});  // LO_F16: the second pass is nkt1 more steps of the same kind
        }
        epilogue();
    }
    if (has_half) {
        if constexpr (F8) {
            for (int k = 0; k < nkt1; ++k) half_step(std::false_type{});
            for (int k = 0; k < nkt2; ++k) half_step(std::true_type{});
        } else {
            for (int k = 0; k < nkt; ++k) half_step(std::false_type{});
        }
        epilogue();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the second group's extra barrier
}

template <typename T, int EPI, bool F8 = false, int ST = 0, int LD = 0, int ABL = 0>
static int launch_pp(const GemmArgs& a, hipStream_t s, const GemmOpts& o) {
    constexpr int lds = 2 * 65536;
    auto kern = gemm_pp_kernel<T, EPI, F8, ST, LD, ABL>;
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipDeviceGetAttribute(&pd.ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        pd.done[dev] = true;
    }
    const int ncu = pd.ncu[dev];
    const int ntm = (a.M + 255) / 256, ntn = (a.N + 255) / 256, ntiles = ntm * ntn;
    // grid = CUs; a last partial wave of R tiles with 2 R <= grid is run as 2 R half tiles (see the kernel); fewer tiles
    // than half the CUs: every tile is split
    int grid = ntiles < ncu ? (2 * ntiles <= ncu ? 2 * ntiles : ntiles) : ncu;
    if (a.flags & 2) grid = ntiles < ncu ? ntiles : ncu;  // tuning knob: no half tiles
    const int rem = ntiles % grid, rem_half = (rem > 0 && 2 * rem <= grid && !(a.flags & 2)) ? rem : 0;
    if (o.ev_start && o.ev_stop) {
        // measurement mode: the two events ride on the kernel's own dispatch packet (no marker packets between kernels, which
        // cost ~7 us per pair and serialise the queue)
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, o.ev_start, o.ev_stop, 0, a, ntn, ntiles, rem_half);
    } else {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, ntn, ntiles, rem_half);
    }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

template <typename T>
static int launch_pp_t(int epi, const GemmArgs& a, hipStream_t s, const GemmOpts& o) {
    if constexpr (T::id == DT_F16) {
        if (a.lo_mode == LO_F8) {  // the e4m3 second pass: forward GEMMs of a split tower only
            switch (epi) {
                case EPI_STORE: return launch_pp<T, EPI_STORE, true>(a, s, o);
                case EPI_GELU:
                    if (!a.out1_lo) return launch_pp<T, EPI_GELU, true>(a, s, o);
                    return a.out1_lo_mode == LO_F8 ? launch_pp<T, EPI_GELU_SPLIT8, true>(a, s, o) : launch_pp<T, EPI_GELU_SPLIT, true>(a, s, o);
                case EPI_STORE_F32: return launch_pp<T, EPI_STORE_F32, true>(a, s, o);
            }
            set_error("gemm_pp: epilogue %d is not built with the e4m3 second pass", epi);
            return MUDPT_ERR_ARG;
        }
    }
    if constexpr (T::id == DT_BF16) {  // timing-only ablations (flags bits 4, 5), bf16 store epilogue only
        if (epi == EPI_STORE && (a.flags & 0x30)) {
            switch ((a.flags >> 4) & 3) {
                case 1: return launch_pp<T, EPI_STORE, false, 0, 0, 1>(a, s, o);
                case 2: return launch_pp<T, EPI_STORE, false, 0, 0, 2>(a, s, o);
                default: return launch_pp<T, EPI_STORE, false, 0, 0, 3>(a, s, o);
            }
        }
    }
    switch (epi) {
        case EPI_STORE: return launch_pp<T, EPI_STORE>(a, s, o);
        case EPI_GELU:
            if (!a.out1_lo) return a.gelu_q8 ? launch_pp<T, EPI_GELU_Q8>(a, s, o) : launch_pp<T, EPI_GELU>(a, s, o);
            return a.out1_lo_mode == LO_F8 ? launch_pp<T, EPI_GELU_SPLIT8>(a, s, o) : launch_pp<T, EPI_GELU_SPLIT>(a, s, o);
        case EPI_GELU_BWD: return a.gelu_q8 ? launch_pp<T, EPI_GELU_BWD_Q8>(a, s, o) : launch_pp<T, EPI_GELU_BWD>(a, s, o);
        case EPI_STORE_F32: return launch_pp<T, EPI_STORE_F32>(a, s, o);
    }
    set_error("gemm_pp: epilogue %d is not built for the ping-pong kernel", epi);
    return MUDPT_ERR_ARG;
}

// Arguments are validated by launch_gemm (gemm.hip) before it dispatches here.
int launch_gemm_pp(int dtype, int epi, const GemmArgs& a, hipStream_t s, const GemmOpts& o) {
    ARG_CHECK((size_t)a.M * a.lda * 2 < 0xffffffffull && (size_t)a.N * a.ldb * 2 < 0xffffffffull, "gemm_pp: operand larger than 4 GiB");
    ARG_CHECK(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldo0 % 8 == 0, "gemm_pp: strides must be multiples of 8");
    ARG_CHECK(!a.out1_lo || (uintptr_t)a.out1_lo % 16 == 0, "gemm_pp: out1_lo must be 16-byte aligned");
    // epilogue offsets are 32-bit and rely on the descriptors' range check for rows >= M
    ARG_CHECK(!a.gelu_q8 || (a.lo_mode == LO_NONE && a.ldo0 % 16 == 0 && a.ldaux % 16 == 0), "gemm_pp: gelu_q8 needs 16-byte row strides and no split operand");
    ARG_CHECK((size_t)a.M * a.ldo0 * (epi == EPI_STORE_F32 ? 4 : 2) < 0x7fffffffull && (size_t)a.M * (size_t)(a.ldo1 > a.ldaux ? a.ldo1 : a.ldaux) * 2 < 0x7fffffffull, "gemm_pp: output larger than 2 GiB");
    if (dtype == DT_BF16) return launch_pp_t<BF16>(epi, a, s, o);
    if (dtype == DT_F16) return launch_pp_t<F16>(epi, a, s, o);
    set_error("gemm: unknown dtype %d", dtype);
    return MUDPT_ERR_ARG;
}

}  // namespace mudpt
