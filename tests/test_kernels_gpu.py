"""GPU parity of the single HIP kernels against the CPU oracle, called through the C ABI (ctypes)."""
import ctypes as C

import pytest
import torch

from oracle import mudpt_oracle as O

pytestmark = pytest.mark.gpu

DT = {"bf16": (0, torch.bfloat16), "fp16": (1, torch.float16)}
EPS = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}  # half ulp relative


@pytest.fixture(scope="module")
def lib():
    from mudpt_amd import capi
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return capi.load()


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def ok(lib, rc):
    assert rc == 0, lib.mudpt_last_error().decode()


def gemm(lib, dt, epi, A, B, bias=None, out0=None, out1=None, aux=None, patches=0, seq_len=0, pos=None, variant=0):
    M, K = A.shape
    N = B.shape[0]
    ok(lib, lib.mudpt_gemm(dt, epi, M, N, K, P(A), A.stride(0), P(B), B.stride(0), P(bias), P(out0), out0.stride(0),
                           P(out1), out1.stride(0) if out1 is not None else 0, P(aux), aux.stride(0) if aux is not None else 0,
                           patches, seq_len, P(pos), variant, None))
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemm_exact_integers(lib, dtype):
    """Small-integer operands: every product and sum is exact in fp32, so the result must be bit-exact.
    A is a shifted identity-like pattern and B is asymmetric, which catches swapped row/col maps."""
    dt, tt = DT[dtype]
    M, N, K = 200, 144, 128
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = (torch.arange(N).view(N, 1) % 5 - 2 + (torch.arange(K).view(1, K) % 3)).float()  # asymmetric
    ref = A @ B.t()
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    gemm(lib, dt, 5, A.cuda().to(tt), B.cuda().to(tt), out0=out)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemm_exact_integers_full_size(lib, dtype):
    """BASELINE config 2's row count (M = 256 x 201): 603 tiles = two full waves + 91 tiles run as 182 half tiles.
    Small-integer operands make every fp32 sum exact, so full and half tiles must be bit-exact."""
    dt, tt = DT[dtype]
    M, N, K = 51456, 768, 256
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = (torch.arange(N).view(N, 1) % 7 - 3 + (torch.arange(K).view(1, K) % 4)).float()
    ref = A @ B.t()
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    gemm(lib, dt, 5, A.cuda().to(tt), B.cuda().to(tt), out0=out)
    assert torch.equal(out.cpu(), ref)
    outT = torch.empty(M, N, device="cuda", dtype=tt)  # |values| <= 3 * 6 * 256 is exact in bf16? no: compare after the same rounding
    gemm(lib, dt, 0, A.cuda().to(tt), B.cuda().to(tt), out0=outT)
    assert torch.equal(outT.cpu(), ref.to(tt))


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(804, 2304, 768), (847, 512, 2048), (77, 128, 64), (4096, 768, 3072), (33000, 768, 768),
                                   (22100, 768, 768), (51400, 768, 192), (99, 512, 192), (99, 2048, 512)])
def test_gemm_epilogues(lib, dtype, shape):
    """All fused epilogues against a float64 product.  The last two shapes leave a partial last wave of 256 x 256 tiles on 256
    CUs (261 = 256 + 5 and 603 = 2 * 256 + 91 tiles), which the persistent kernel runs as half tiles; both have a ragged
    last row panel (22100: the lower half tile is entirely out of range).  The 99-row shapes are the trimmed text tower's: the
    small-grid kernel with its 4-deep operand ring (K = 192: exactly the ring's prologue depth)."""
    dt, tt = DT[dtype]
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(tt)
    B = (torch.randn(N, K, generator=g) * K ** -0.5).to(tt)
    bias = torch.randn(N, generator=g)
    acc = A.double() @ B.double().t()
    Ad, Bd, bd = A.cuda(), B.cuda(), bias.cuda()
    tol = dict(atol=4 * EPS[dtype], rtol=4 * EPS[dtype])
    # 0: store T with bias
    out = torch.empty(M, N, device="cuda", dtype=tt)
    gemm(lib, dt, 0, Ad, Bd, bias=bd, out0=out)
    torch.testing.assert_close(out.cpu().double(), acc + bias.double(), **tol)
    # 5: fp32 store, no bias
    o32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    gemm(lib, dt, 5, Ad, Bd, out0=o32)
    torch.testing.assert_close(o32.cpu().double(), acc, atol=2e-5 * K ** 0.5, rtol=1e-5)
    # 1: bias + QuickGELU, both outputs
    u, gl = torch.empty(M, N, device="cuda", dtype=tt), torch.empty(M, N, device="cuda", dtype=tt)
    gemm(lib, dt, 1, Ad, Bd, bias=bd, out0=u, out1=gl)
    uref = acc + bias.double()
    torch.testing.assert_close(u.cpu().double(), uref, **tol)
    torch.testing.assert_close(gl.cpu().double(), uref * torch.sigmoid(1.702 * uref), **tol)
    # 2: residual fp32
    res = torch.randn(M, N, generator=g)
    gemm(lib, dt, 2, Ad, Bd, bias=bd, out0=o32, aux=res.cuda())
    torch.testing.assert_close(o32.cpu().double(), res.double() + uref, atol=2e-5 * K ** 0.5, rtol=1e-5)
    # 3: QuickGELU backward
    upre = torch.randn(M, N, generator=g).to(tt)
    gemm(lib, dt, 3, Ad, Bd, out0=out, aux=upre.cuda())
    ud = upre.double()
    s = torch.sigmoid(1.702 * ud)
    torch.testing.assert_close(out.cpu().double(), acc * (s * (1 + 1.702 * ud * (1 - s))), **tol)
    # 1 / 3 with QuickGELU' in 8 bits (variant bit 17; what the bf16 mode keeps for the backward instead of u): the forward writes byte codes
    # rint((g' + 0.1) * 212), the backward multiplies by the decoded value -- absolute error <= 0.5 / 212 on a factor in [-0.1, 1.1]
    Q8 = 0x20000
    codes = torch.zeros(M, N, device="cuda", dtype=torch.uint8)
    gl2 = torch.empty(M, N, device="cuda", dtype=tt)
    gemm(lib, dt, 1, Ad, Bd, bias=bd, out0=codes, out1=gl2, variant=Q8)
    assert torch.equal(gl2, gl)
    sr = torch.sigmoid(1.702 * uref)
    gp = sr * (1 + 1.702 * uref * (1 - sr))
    want = torch.round((gp + 0.1) * 212.0)
    assert (codes.cpu().double() - want).abs().max().item() <= 1 and ((codes.cpu().double() - want).abs() > 0).double().mean().item() < 0.02  # ties / 1-ulp exp
    assert ((codes.cpu().double() / 212.0 - 0.1) - gp).abs().max().item() <= 0.5 / 212 + 1e-3
    out8 = torch.empty(M, N, device="cuda", dtype=tt)
    cq = torch.randint(0, 255, (M, N), generator=g, dtype=torch.uint8)
    gemm(lib, dt, 3, Ad, Bd, out0=out8, aux=cq.cuda(), variant=Q8)
    torch.testing.assert_close(out8.cpu().double(), acc * (cq.double() / 212.0 - 0.1), **tol)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemm_rows_do_not_depend_on_the_kernel_that_ran_them(lib, dtype):
    """A row of a GEMM must come out bit-identical whichever kernel ran the shape -- the small-tile kernel (64 x 64 or 128 x 128 tiles) or the
    persistent one: every kernel contracts k in the same order, and (round 4) every T conversion of an epilogue rounds the fp32 VALUE
    (common.h round_to: with -ffp-contract=fast the compiler folded the QuickGELU multiply into the conversion in one kernel and not in the
    other, one ulp apart).  That is what makes a trimmed / bucketed text tower equal the untrimmed one (tests/test_manyclass_gpu.py).
    The text tower's shapes at 208 class prompts: 5 408 rows (trimmed) against 16 016 (untrimmed); and 450 rows, a grid small enough for the
    128-deep K-tiles of round 4 (two workgroups per CU: gemm.hip deep_k_tiles)."""
    dt, tt = DT[dtype]
    M2, M1, M0 = 450, 5408, 16016
    g = torch.Generator().manual_seed(11)
    for name, N, K, epi in (("qkv", 1536, 512, 0), ("out", 512, 512, 5), ("fc", 2048, 512, 1), ("proj", 512, 2048, 5), ("dgelu", 2048, 512, 3), ("dfc", 512, 2048, 0)):
        A = torch.randn(M0, K, generator=g).to(tt).cuda()
        B = (torch.randn(N, K, generator=g) * K ** -0.5).to(tt).cuda()
        bias = torch.randn(N, generator=g).cuda()
        auxf = torch.randn(M0, N, generator=g).to(tt).cuda()
        res = {}
        for M in (M2, M1, M0):
            out0 = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 5 else tt)
            out1 = torch.zeros(M, N, device="cuda", dtype=tt) if epi == 1 else None
            gemm(lib, dt, epi, A[:M], B, bias=bd_or_none(bias, epi), out0=out0, out1=out1, aux=auxf[:M] if epi == 3 else None)
            res[M] = (out0[:M1].clone(), out1[:M1].clone() if out1 is not None else None)
        assert torch.equal(res[M1][0], res[M0][0]), name
        assert torch.equal(res[M2][0], res[M0][0][:M2]), name
        if epi == 1:
            assert torch.equal(res[M1][1], res[M0][1]), name
            assert torch.equal(res[M2][1], res[M0][1][:M2]), name


def bd_or_none(bias, epi):
    return None if epi == 3 else bias


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(804, 768, 3072), (804, 768, 2304), (450, 512, 2048), (201, 768, 3072), (1000, 768, 1536)])
def test_gemm_split_k(lib, dtype, shape):
    """Small grids with a long contraction (the reference's training batch of 4: M = 804) split K over up to 4 slices whose fp32 partials are
    summed in slice order.  Small-integer operands make every sum exact, so the split result is BIT-exact (both store epilogues, bias
    included); random operands agree with the unsplit kernel to fp32 rounding; run to run bit for bit."""
    dt, tt = DT[dtype]
    M, N, K = shape
    g = torch.Generator().manual_seed(M + K)
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    B = (torch.arange(N).view(N, 1) % 5 - 2 + (torch.arange(K).view(1, K) % 3)).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    ref = A @ B.t() + bias
    Ad, Bd, bd = A.cuda().to(tt), B.cuda().to(tt), bias.cuda()
    o32 = torch.empty(M, N, device="cuda")
    gemm(lib, dt, 5, Ad, Bd, bias=bd, out0=o32, variant=0x10000)
    assert torch.equal(o32.cpu(), ref)
    oT = torch.empty(M, N, device="cuda", dtype=tt)
    gemm(lib, dt, 0, Ad, Bd, bias=bd, out0=oT, variant=0x10000)
    assert torch.equal(oT.cpu(), ref.to(tt))
    Ar, Br = torch.randn(M, K, generator=g).to(tt).cuda(), (torch.randn(N, K, generator=g) * K ** -0.5).to(tt).cuda()
    a32, b32, c32 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    gemm(lib, dt, 5, Ar, Br, out0=a32)
    gemm(lib, dt, 5, Ar, Br, out0=b32, variant=0x10000)
    gemm(lib, dt, 5, Ar, Br, out0=c32, variant=0x10000)
    torch.cuda.synchronize()
    torch.testing.assert_close(b32, a32, atol=2e-5 * K ** 0.5, rtol=1e-5)
    assert torch.equal(b32, c32)
    assert not torch.equal(b32, a32) or K < 1536  # the split path really ran (a different summation order shows in the last bits)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemm_patch_epilogue(lib, dtype):
    """Patch-embed GEMM: row m of the im2col matrix lands on token row (m / P) * L + 1 + m % P, plus pos-emb."""
    dt, tt = DT[dtype]
    Bn, Pn, L, N, K = 3, 4, 7, 192, 768
    g = torch.Generator().manual_seed(5)
    A = torch.randn(Bn * Pn, K, generator=g).to(tt)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(tt)
    pos = torch.randn(1 + Pn, N, generator=g)
    out = torch.full((Bn * L, N), 7.0, device="cuda")
    gemm(lib, dt, 4, A.cuda(), W.cuda(), out0=out, patches=Pn, seq_len=L, pos=pos.cuda())
    ref = torch.full((Bn, L, N), 7.0, dtype=torch.float64)
    ref[:, 1:1 + Pn] = (A.double() @ W.double().t()).view(Bn, Pn, N) + pos[1:].double()
    torch.testing.assert_close(out.cpu().double().view(Bn, L, N), ref, atol=1e-4, rtol=1e-5)


def test_gemm_rejects_bad_shapes(lib):
    A = torch.zeros(64, 96, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    rc = lib.mudpt_gemm(0, 0, 64, 64, 96, P(A), 96, P(A), 96, None, P(out), 64, None, 0, None, 0, 0, 0, None, 0, None)
    assert rc == 1 and b"multiple of 64" in lib.mudpt_last_error()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("rows,d", [(804, 768), (847, 512), (21, 192), (5, 128), (4, 1024)])
def test_layernorm_fwd_bwd(lib, dtype, rows, d):
    dt, tt = DT[dtype]
    g = torch.Generator().manual_seed(rows + d)
    x = torch.randn(rows, d, generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    xr = x.clone().requires_grad_(True)
    y = O.layer_norm(xr, gamma, beta)
    out = torch.empty(rows, d, device="cuda", dtype=tt)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    xc, gc, bc = x.cuda(), gamma.cuda(), beta.cuda()  # keep device tensors alive across the async launches
    ok(lib, lib.mudpt_layernorm_fwd(dt, P(xc), d, None, P(gc), P(bc), P(out), d, 0, P(mean), P(rstd), rows, d, None))
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu().float(), y.detach(), atol=4 * EPS[dtype], rtol=4 * EPS[dtype])
    o32 = torch.empty(rows, d, device="cuda")
    ok(lib, lib.mudpt_layernorm_fwd(dt, P(xc), d, None, P(gc), P(bc), P(o32), d, 1, None, None, rows, d, None))
    torch.cuda.synchronize()
    torch.testing.assert_close(o32.cpu(), y.detach(), atol=2e-5, rtol=1e-5)
    # backward: dx = dres + LN'(dy), dy in T
    dy = torch.randn(rows, d, generator=g).to(tt)
    dres = torch.randn(rows, d, generator=g)
    (dx_ref,) = torch.autograd.grad(y, xr, dy.float())
    dx = torch.empty(rows, d, device="cuda")
    dx_lp = torch.empty(rows, d, device="cuda", dtype=tt)
    dyc, drc = dy.cuda(), dres.cuda()
    ok(lib, lib.mudpt_layernorm_bwd(dt, P(dyc), d, 0, P(xc), d, None, P(mean), P(rstd), P(gc), P(drc), d, P(dx), d,
                                    P(dx_lp), d, rows, d, None))
    torch.cuda.synchronize()
    torch.testing.assert_close(dx.cpu(), dx_ref + dres, atol=2e-5, rtol=2e-5)
    torch.testing.assert_close(dx_lp.cpu().float(), dx_ref + dres, atol=4 * EPS[dtype] * 4, rtol=4 * EPS[dtype])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("add_kind", ["f32", "lp", "none"])
def test_layernorm_fused_add_and_prompt_splice(lib, dtype, add_kind):
    """What block_fwd fuses into LayerNorm (clip/model.py:281-301): v = x + add (fp32 addend, or the T-precision update stream),
    rows 1..n of every L-row sequence REPLACED by the deep-prompt rows (no add on those), v written back as the block input
    (bit-exact: one fp32 add), then normalised."""
    dt, tt = DT[dtype]
    nseq, L, d, row0, n = 5, 13, 512, 9, 4   # vision-style: the last n rows of every sequence
    rows = nseq * L
    g = torch.Generator().manual_seed(17)
    x = torch.randn(rows, d, generator=g) * 1.5 + 0.3
    add = torch.randn(rows, d, generator=g) * 0.5
    add_t = add.to(tt)
    ov = torch.randn(n, d, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    v = x + (add if add_kind == "f32" else add_t.float() if add_kind == "lp" else 0)
    v = v.view(nseq, L, d).clone()
    v[:, row0:row0 + n] = ov
    v = v.view(rows, d)
    y = O.layer_norm(v, gamma, beta)
    xc, gc, bc, oc = x.cuda(), gamma.cuda(), beta.cuda(), ov.cuda()
    a32 = add.cuda() if add_kind == "f32" else None
    alp = add_t.cuda() if add_kind == "lp" else None
    xout = torch.full((rows, d), float("nan"), device="cuda")
    out = torch.empty(rows, d, device="cuda", dtype=tt)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ok(lib, lib.mudpt_layernorm_fwd_fused(dt, P(xc), d, P(a32), P(alp), d, P(oc), row0, n, L, P(xout), d, P(gc), P(bc), P(out), d, 0,
                                          P(mean), P(rstd), rows, d, None))
    torch.cuda.synchronize()
    assert torch.equal(xout.cpu(), v)                      # the saved block input: exact
    torch.testing.assert_close(out.cpu().float(), y, atol=4 * EPS[dtype], rtol=4 * EPS[dtype])
    torch.testing.assert_close(mean.cpu(), v.mean(-1), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(rstd.cpu(), (v.var(-1, unbiased=False) + 1e-5).rsqrt(), atol=1e-5, rtol=2e-5)
    # fp32 output and no splice: every row is x + add
    o32 = torch.empty(rows, d, device="cuda")
    ok(lib, lib.mudpt_layernorm_fwd_fused(dt, P(xc), d, P(a32), P(alp), d, None, 0, 0, 1, None, 0, P(gc), P(bc), P(o32), d, 1, None, None, rows, d, None))
    torch.cuda.synchronize()
    v2 = x + (add if add_kind == "f32" else add_t.float() if add_kind == "lp" else 0)
    torch.testing.assert_close(o32.cpu(), O.layer_norm(v2, gamma, beta), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,C,e", [(4, 11, 512), (256, 1000, 512), (5, 3, 128), (3, 1, 64)])
def test_head_matches_torch_cross_entropy(lib, B, C, e):
    """Cosine logits + mean cross-entropy, forward and backward (trainers/mudpt.py:178-182,250) against torch autograd in float64,
    at the benchmark's C = 11 and at BASELINE configs[2]'s C = 1000 (CE over 1000 columns)."""
    g = torch.Generator().manual_seed(B * 1000 + C)
    img, txt = torch.randn(B, e, generator=g) * 3, torch.randn(C, e, generator=g) * 0.2
    labels = torch.randint(0, C, (B,), generator=g)
    scale, gscale = 14.2857, 0.5
    i64, t64 = img.double().requires_grad_(True), txt.double().requires_grad_(True)
    ref_logits = scale * torch.nn.functional.normalize(i64, dim=-1) @ torch.nn.functional.normalize(t64, dim=-1).t()
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, labels)
    (gscale * ref_loss).backward()
    ic, tc, lc = img.cuda(), txt.cuda(), labels.cuda()
    logits, loss = torch.empty(B, C, device="cuda"), torch.empty(1, device="cuda")
    dimg, dtxt = torch.empty(B, e, device="cuda"), torch.empty(C, e, device="cuda")
    ok(lib, lib.mudpt_head(P(ic), P(tc), P(lc), scale, gscale, B, C, e, P(logits), P(loss), P(dimg), P(dtxt), None))
    torch.testing.assert_close(logits.cpu().double(), ref_logits.detach(), atol=2e-5, rtol=1e-5)
    assert abs(loss.item() - ref_loss.item()) <= 2e-6 * max(1.0, abs(ref_loss.item()))
    for got, ref in ((dimg, i64.grad), (dtxt, t64.grad)):
        rms = ref.pow(2).mean().sqrt().item()
        assert (got.cpu().double() - ref).abs().max().item() <= 2e-5 * rms + 1e-12
    # forward only (model_inference): no labels
    logits2 = torch.empty(B, C, device="cuda")
    ok(lib, lib.mudpt_head(P(ic), P(tc), None, scale, 1.0, B, C, e, P(logits2), None, None, None, None))
    assert torch.equal(logits2, logits)


def test_head_label_out_of_range_gives_nan_loss_not_a_fault(lib):
    B, C, e = 4, 7, 64
    img, txt = torch.randn(B, e).cuda(), torch.randn(C, e).cuda()
    labels = torch.tensor([0, 7, 3, -1]).cuda()  # 7 and -1 are outside [0, C)
    logits, loss = torch.empty(B, C, device="cuda"), torch.empty(1, device="cuda")
    dimg, dtxt = torch.empty(B, e, device="cuda"), torch.empty(C, e, device="cuda")
    ok(lib, lib.mudpt_head(P(img), P(txt), P(labels), 10.0, 1.0, B, C, e, P(logits), P(loss), P(dimg), P(dtxt), None))
    assert torch.isnan(loss).all() and torch.isfinite(logits).all()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,L,d,row0,n", [(256, 201, 768, 197, 4), (1000, 20, 512, 1, 4), (3, 7, 64, 2, 1), (33, 5, 128, 0, 2)])
def test_reduce_rows(lib, dtype, B, L, d, row0, n):
    """Backward of the prompt splice: sum over the batch of the n prompt rows.  Against a float64 sum, bit for bit run to run,
    from the fp32 stream and from its T copy, with accumulate and with clearing of the summed rows."""
    dt, tt = DT[dtype]
    g = torch.Generator().manual_seed(B + d)
    src = torch.randn(B, L, d, generator=g)
    ref = src[:, row0:row0 + n].double().sum(0)
    sc = src.cuda()
    out = torch.empty(n, d, device="cuda")
    ok(lib, lib.mudpt_reduce_rows(dt, P(sc), None, B, L, d, row0, n, P(out), 0, 0, 0.25, None))
    torch.cuda.synchronize()
    tol = 4e-6 * B ** 0.5  # fp32 summation of B unit-variance values: a few ulp of sqrt(B)
    assert (out.cpu().double() - 0.25 * ref).abs().max().item() <= tol
    again = torch.empty(n, d, device="cuda")
    for _ in range(3):
        ok(lib, lib.mudpt_reduce_rows(dt, P(sc), None, B, L, d, row0, n, P(again), 0, 0, 0.25, None))
        torch.cuda.synchronize()
        assert torch.equal(again, out)   # fixed-order tree: bitwise reproducible
    # accumulate: out += sum
    ok(lib, lib.mudpt_reduce_rows(dt, P(sc), None, B, L, d, row0, n, P(again), 0, 1, 0.25, None))
    torch.cuda.synchronize()
    torch.testing.assert_close(again, 2 * out, atol=1e-6, rtol=1e-6)
    # T-precision source (the bf16 gradient stream), cleared afterwards together with the fp32 copy
    lp = src.to(tt).cuda()
    ref_lp = lp.cpu().float()[:, row0:row0 + n].double().sum(0)
    out2 = torch.empty(n, d, device="cuda")
    ok(lib, lib.mudpt_reduce_rows(dt, None, P(lp), B, L, d, row0, n, P(out2), 1, 0, 1.0, None))
    torch.cuda.synchronize()
    assert (out2.cpu().double() - ref_lp).abs().max().item() <= tol
    assert torch.count_nonzero(lp[:, row0:row0 + n]) == 0
    keep = torch.ones(L, dtype=torch.bool); keep[row0:row0 + n] = False
    assert torch.equal(lp[:, keep.cuda()].cpu(), src.to(tt)[:, keep])  # other rows untouched


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,C,L,d,n", [(64, 11, 9, 512, 4), (3, 100, 20, 512, 4), (2, 5, 7, 128, 2)])
def test_cocoop_dbias(lib, dtype, B, C, L, d, n):
    """Gradient of meta_net's per-image context shift (trainers/cocoop.py:141-146): the sum over an image's C prompts and n context rows
    of the text-input gradient; from the fp32 stream and from its T copy, against a float64 sum, bit for bit run to run."""
    dt, tt = DT[dtype]
    g = torch.Generator().manual_seed(B * C + L)
    dx = torch.randn(B * C, L, d, generator=g)
    ref = dx.view(B, C, L, d)[:, :, 1:1 + n].double().sum((1, 2))
    dxc, out = dx.cuda(), torch.empty(B, d, device="cuda")
    ok(lib, lib.mudpt_cocoop_dbias(dt, P(dxc), None, P(out), B, C, L, d, n, 0.5, None))
    torch.cuda.synchronize()
    tol = 4e-6 * (C * n) ** 0.5
    assert (out.cpu().double() - 0.5 * ref).abs().max().item() <= tol
    lp = dx.to(tt).cuda()
    ref_lp = lp.cpu().float().view(B, C, L, d)[:, :, 1:1 + n].double().sum((1, 2))
    out2, out3 = torch.empty(B, d, device="cuda"), torch.empty(B, d, device="cuda")
    ok(lib, lib.mudpt_cocoop_dbias(dt, None, P(lp), P(out2), B, C, L, d, n, 1.0, None))
    ok(lib, lib.mudpt_cocoop_dbias(dt, None, P(lp), P(out3), B, C, L, d, n, 1.0, None))
    torch.cuda.synchronize()
    assert (out2.cpu().double() - ref_lp).abs().max().item() <= 2 * tol and torch.equal(out2, out3)


@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(4, 768, 512), (44, 512, 768), (17, 33, 70), (256, 11, 512)])
def test_sgemm_all_transpose_forms(lib, tA, tB, M, N, K):
    """fp32 C = alpha op(A) op(B) + bias + beta C: the prompt projections (trainers/mudpt.py:127-128, clip/model.py:539), their
    weight / input gradients and the logit contraction; every transpose form, ragged shapes."""
    g = torch.Generator().manual_seed(M * N + K + 2 * tA + tB)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    bias, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    opA, opB = (A.t() if tA else A).double(), (B.t() if tB else B).double()
    ref = 0.7 * (opA @ opB) + bias.double() + 0.3 * C0.double()
    Ac, Bc, bc, Cc = A.cuda(), B.cuda(), bias.cuda(), C0.clone().cuda()
    ok(lib, lib.mudpt_sgemm(tA, tB, M, N, K, 0.7, P(Ac), A.stride(0), P(Bc), B.stride(0), 0.3, P(Cc), N, P(bc), None))
    torch.cuda.synchronize()
    torch.testing.assert_close(Cc.cpu().double(), ref, atol=3e-5 * K ** 0.5, rtol=1e-5)
    Cd = torch.full((M, N), float("nan"), device="cuda")   # beta = 0 must not read C
    ok(lib, lib.mudpt_sgemm(tA, tB, M, N, K, 1.0, P(Ac), A.stride(0), P(Bc), B.stride(0), 0.0, P(Cd), N, None, None))
    torch.cuda.synchronize()
    torch.testing.assert_close(Cd.cpu().double(), opA @ opB, atol=3e-5 * K ** 0.5, rtol=1e-5)


def test_layernorm_gather_scatter(lib):
    """row_index: LN of selected token rows (ln_post on CLS rows, ln_final on EOT rows) and the scatter of its gradient."""
    rows, d, total = 6, 256, 40
    g = torch.Generator().manual_seed(3)
    x = torch.randn(total, d, generator=g)
    idx = torch.tensor([0, 7, 14, 21, 28, 39], dtype=torch.int32)
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    xr = x[idx.long()].clone().requires_grad_(True)
    y = O.layer_norm(xr, gamma, beta)
    out, mean, rstd = torch.empty(rows, d, device="cuda"), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    xc, ic, gc, bc = x.cuda(), idx.cuda(), gamma.cuda(), beta.cuda()
    ok(lib, lib.mudpt_layernorm_fwd(0, P(xc), d, P(ic), P(gc), P(bc), P(out), d, 1, P(mean), P(rstd), rows, d, None))
    dy = torch.randn(rows, d, generator=g)
    (dref,) = torch.autograd.grad(y, xr, dy)
    dx = torch.zeros(total, d, device="cuda")
    dyc = dy.cuda()
    ok(lib, lib.mudpt_layernorm_bwd(0, P(dyc), d, 1, P(xc), d, P(ic), P(mean), P(rstd), P(gc), None, 0, P(dx), d, None, 0, rows, d, None))
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu(), y.detach(), atol=2e-5, rtol=1e-5)
    full = torch.zeros(total, d)
    full[idx.long()] = dref
    torch.testing.assert_close(dx.cpu(), full, atol=2e-5, rtol=2e-5)


ATTN_CASES = [(3, 201, 12, False), (11, 77, 8, True), (2, 7, 3, False), (2, 33, 2, True), (1, 224, 1, False), (2, 64, 2, True),
              # L > 224: the tiled (online-softmax) kernels; 581 = ViT-L/14@336's 577 tokens + 4 prompt rows (BASELINE configs[4])
              # (forward: K and V of a pair resident in LDS up to L = 640, one workgroup per pair; above that the staged 16-query-block form)
              (2, 300, 2, False), (1, 581, 3, False), (2, 260, 2, True), (1, 225, 1, False), (1, 640, 2, False), (1, 641, 1, False),
              (2, 513, 2, True), (70, 300, 4, False),
              # backward: Q | dO | lse | delta of a pair resident up to L = 608; 609 .. 640: resident forward, staged backward
              (1, 608, 1, False), (1, 609, 1, True),
              # more (sequence, head) pairs than resident workgroups: the persistent loops of the forward and the fused backward walk
              # several pairs per workgroup (images of the next pair stream in while the current one is computed)
              (150, 201, 2, False), (700, 20, 8, True)]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,L,H,causal", ATTN_CASES)
def test_attention_fwd_bwd(lib, dtype, B, L, H, causal):
    dt, tt = DT[dtype]
    g = torch.Generator().manual_seed(B * 1000 + L)
    qkv = torch.randn(B, L, 3 * H * 64, generator=g).to(tt)
    dout = torch.randn(B, L, H * 64, generator=g).to(tt)
    q32 = qkv.float().requires_grad_(True)
    ref = O.attention(q32, H, O.causal_mask(L) if causal else None)
    (dref,) = torch.autograd.grad(ref, q32, dout.float())
    Lp = lib.mudpt_attention_padded_len(L)
    assert Lp % 32 == 0 and Lp >= L
    qc = qkv.cuda()
    out = torch.empty(B, L, H * 64, device="cuda", dtype=tt)
    lse = torch.empty(B, H, Lp, device="cuda")
    ok(lib, lib.mudpt_attention_fwd(dt, P(qc), P(out), P(lse), B, L, H, int(causal), None))
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu().float(), ref.detach(), atol=6 * EPS[dtype], rtol=6 * EPS[dtype])
    # LSE against the definition
    q, k, _ = q32.detach().split(H * 64, dim=-1)
    s = (q.view(B, L, H, 64).transpose(1, 2) @ k.view(B, L, H, 64).transpose(1, 2).transpose(-1, -2)) / 8
    if causal:
        s = s + O.causal_mask(L)
    torch.testing.assert_close(lse.cpu()[:, :, :L], torch.logsumexp(s, dim=-1), atol=1e-3, rtol=1e-4)
    dqkv = torch.zeros(B, L, 3 * H * 64, device="cuda", dtype=tt)
    delta = torch.empty(B, H, Lp, device="cuda")
    doc = dout.cuda()
    ok(lib, lib.mudpt_attention_bwd(dt, P(qc), P(out), P(doc), P(lse), P(delta), P(dqkv), B, L, H, int(causal), None))
    torch.cuda.synchronize()
    scale = dref.abs().max().item()
    torch.testing.assert_close(dqkv.cpu().float(), dref, atol=12 * EPS[dtype] * scale, rtol=8 * EPS[dtype])
    # Every form (fixed summation order, no atomics) is bit for bit the same run to run.  The two-kernel form (bit 1), the fused single
    # pass (bit 3: Q, K, V, dO resident once) and the fused pass with two blocks per wave (bit 2) do the same per-block arithmetic in the
    # same order, compiled separately (fma contraction may differ): the same gradients to T-precision rounding.
    again = torch.zeros_like(dqkv)
    ok(lib, lib.mudpt_attention_bwd(dt, P(qc), P(out), P(doc), P(lse), P(delta), P(again), B, L, H, int(causal), None))
    torch.cuda.synchronize()
    assert torch.equal(again, dqkv)
    # bit 4: the single-sweep kernel (non-causal, L <= 224): S / dP / exp computed once, dS rounded to T crosses LDS for dQ -- the same
    # rounding point as the other forms (they round dS to T for the dQ product too)
    forms = (2, 4, 8) if causal or L > 224 else (2, 4, 8, 16)
    for form in forms:
        other = torch.full_like(dqkv, float("nan"))
        ok(lib, lib.mudpt_attention_bwd(dt, P(qc), P(out), P(doc), P(lse), P(delta), P(other), B, L, H, int(causal) | form, None))
        torch.cuda.synchronize()
        torch.testing.assert_close(other.cpu().float(), dref, atol=12 * EPS[dtype] * scale, rtol=8 * EPS[dtype])
        torch.testing.assert_close(other.float(), dqkv.float(), atol=2 * EPS[dtype] * scale, rtol=2 * EPS[dtype])
        twice = torch.zeros_like(dqkv)
        ok(lib, lib.mudpt_attention_bwd(dt, P(qc), P(out), P(doc), P(lse), P(delta), P(twice), B, L, H, int(causal) | form, None))
        torch.cuda.synchronize()
        assert torch.equal(twice, other), form


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,L,H,causal,gain", [(2, 581, 3, False, 1.0), (2, 581, 3, False, 3.0), (3, 400, 2, True, 2.5), (1, 640, 1, False, 0.05)])
def test_attention_long_forward_forms(lib, dtype, B, L, H, causal, gain):
    """224 < L <= 640: the resident forward (32 queries per wave, rescale deferred until a row's maximum outgrows its reference by 2^6)
    against the staged 16-query-block kernel (flag bit 1) and the definition.  gain 3: scores of +-100 and more, the reference moves at
    most steps and by far more than the threshold; gain 0.05: it never moves after the first step."""
    dt, tt = DT[dtype]
    g = torch.Generator().manual_seed(L + int(gain * 10))
    qkv = (torch.randn(B, L, 3 * H * 64, generator=g) * gain)
    qkv[..., 2 * H * 64:] /= gain  # values stay O(1)
    qkv = qkv.to(tt)
    Lp = lib.mudpt_attention_padded_len(L)
    qc = qkv.cuda()
    outs, lses = [], []
    for flag in (0, 2):
        out = torch.full((B, L, H * 64), float("nan"), device="cuda", dtype=tt)
        lse = torch.full((B, H, Lp), float("nan"), device="cuda")
        ok(lib, lib.mudpt_attention_fwd(dt, P(qc), P(out), P(lse), B, L, H, int(causal) | flag, None))
        torch.cuda.synchronize()
        outs.append(out.cpu().float()); lses.append(lse.cpu())
    ref = O.attention(qkv.float(), H, O.causal_mask(L) if causal else None)
    for out in outs:
        torch.testing.assert_close(out, ref, atol=6 * EPS[dtype], rtol=6 * EPS[dtype])
    torch.testing.assert_close(outs[0], outs[1], atol=3 * EPS[dtype], rtol=3 * EPS[dtype])
    torch.testing.assert_close(lses[0][:, :, :L], lses[1][:, :, :L], atol=2e-4, rtol=1e-5)
    assert (lses[0][:, :, L:] == 0).all()  # the padded tail the backward reads


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,L,H,causal", [(5, 201, 12, False), (11, 26, 8, True), (3, 581, 4, False), (4, 77, 2, True), (2, 7, 1, False)])
def test_attention_single_query(lib, dtype, B, L, H, causal):
    """The last block's single-query attention (one query per sequence: CLS row 0 for the vision tower, a different EOT position per
    sequence under the causal mask for the text tower) against the oracle's full attention restricted to that row: output, log-sum-exp,
    dq, and the dK / dV rows of EVERY key (zeros behind the causal limit); bit for bit run to run."""
    dt, tt = DT[dtype]
    g = torch.Generator().manual_seed(B * 100 + L)
    qkv = torch.randn(B, L, 3 * H * 64, generator=g).to(tt)
    pos = torch.randint(1, L, (B,), generator=g) if causal else torch.zeros(B, dtype=torch.long)
    sel = (torch.arange(B) * L + pos).to(torch.int32)
    dsel = torch.randn(B, H * 64, generator=g).to(tt)
    q32 = qkv.float().requires_grad_(True)
    ref = O.attention(q32, H, O.causal_mask(L) if causal else None)      # [B, L, H*64]
    ref_sel = ref[torch.arange(B), pos]
    dout = torch.zeros(B, L, H * 64)
    dout[torch.arange(B), pos] = dsel.float()
    (dref,) = torch.autograd.grad(ref, q32, dout)
    qc, sc = qkv.cuda(), sel.cuda()
    q_sel = qkv[torch.arange(B), pos, :H * 64].contiguous().cuda()
    out_sel, lse_sel = torch.empty(B, H * 64, device="cuda", dtype=tt), torch.empty(B, H, device="cuda")
    ok(lib, lib.mudpt_attention_fwd_single(dt, P(qc), P(q_sel), P(sc), P(out_sel), P(lse_sel), B, L, H, int(causal), None))
    torch.cuda.synchronize()
    torch.testing.assert_close(out_sel.cpu().float(), ref_sel.detach(), atol=6 * EPS[dtype], rtol=6 * EPS[dtype])
    qh, kh = q32.detach()[..., :H * 64].view(B, L, H, 64), q32.detach()[..., H * 64:2 * H * 64].view(B, L, H, 64)
    sco = torch.einsum("bhd,blhd->bhl", qh[torch.arange(B), pos], kh) / 8
    if causal:
        sco = sco.masked_fill(torch.arange(L).view(1, 1, L) > pos.view(B, 1, 1), float("-inf"))
    torch.testing.assert_close(lse_sel.cpu(), torch.logsumexp(sco, dim=-1), atol=1e-3, rtol=1e-4)
    dqkv = torch.full((B, L, 3 * H * 64), 7.0, device="cuda", dtype=tt)  # the q third must stay untouched
    dq_sel = torch.empty(B, H * 64, device="cuda", dtype=tt)
    dc = dsel.cuda()
    ok(lib, lib.mudpt_attention_bwd_single(dt, P(qc), P(q_sel), P(sc), P(out_sel), P(dc), P(lse_sel), P(dqkv), P(dq_sel), B, L, H, int(causal), None))
    torch.cuda.synchronize()
    scale = dref.abs().max().item()
    tol = dict(atol=12 * EPS[dtype] * scale, rtol=8 * EPS[dtype])
    torch.testing.assert_close(dq_sel.cpu().float(), dref[torch.arange(B), pos, :H * 64], **tol)
    torch.testing.assert_close(dqkv.cpu().float()[..., H * 64:], dref[..., H * 64:], **tol)
    assert (dqkv[..., :H * 64] == 7.0).all()
    again, dq2 = torch.zeros_like(dqkv), torch.empty_like(dq_sel)
    ok(lib, lib.mudpt_attention_bwd_single(dt, P(qc), P(q_sel), P(sc), P(out_sel), P(dc), P(lse_sel), P(again), P(dq2), B, L, H, int(causal), None))
    torch.cuda.synchronize()
    assert torch.equal(again[..., H * 64:], dqkv[..., H * 64:]) and torch.equal(dq2, dq_sel)


@pytest.mark.parametrize("B,L,H,causal,row0,n", [(3, 201, 12, False, 197, 4), (2, 77, 8, True, 1, 4), (3, 150, 2, False, 14, 4), (2, 581, 4, False, 577, 4),
                                                 (2, 581, 2, False, 62, 5), (4, 26, 8, True, 1, 16)])
def test_attention_backward_window_form(lib, B, L, H, causal, row0, n):
    """Block 0 of a tower needs d(qkv) on the prompt rows only: the window form computes the 16-row blocks (L > 224: 128-row groups) that hold
    rows row0 .. row0 + n - 1 of every sequence -- dQ from all keys, dK / dV from all queries -- and leaves the other rows unwritten.  The
    wanted rows equal the two-kernel form's bit for bit (same sums, same order); the rest of the buffer keeps its previous contents outside
    the computed blocks."""
    dt, tt = DT["bf16"]
    g = torch.Generator().manual_seed(L * 7 + row0)
    qkv = torch.randn(B, L, 3 * H * 64, generator=g).to(tt).cuda()
    dout = torch.randn(B, L, H * 64, generator=g).to(tt).cuda()
    Lp = lib.mudpt_attention_padded_len(L)
    out, lse, delta = torch.empty(B, L, H * 64, device="cuda", dtype=tt), torch.zeros(B, H, Lp, device="cuda"), torch.zeros(B, H, Lp, device="cuda")
    ok(lib, lib.mudpt_attention_fwd(dt, P(qkv), P(out), P(lse), B, L, H, int(causal), None))
    full = torch.empty_like(qkv)
    ok(lib, lib.mudpt_attention_bwd(dt, P(qkv), P(out), P(dout), P(lse), P(delta), P(full), B, L, H, int(causal) | 2, None))
    torch.cuda.synchronize()
    delta_full = delta.clone()
    win = torch.full_like(qkv, 3.0)
    delta.zero_()
    ok(lib, lib.mudpt_attention_bwd(dt, P(qkv), P(out), P(dout), P(lse), P(delta), P(win), B, L, H, int(causal) | (row0 << 8) | (n << 20), None))
    torch.cuda.synchronize()
    assert torch.equal(win[:, row0:row0 + n], full[:, row0:row0 + n])
    assert torch.equal(delta[..., :L], delta_full[..., :L])  # delta of every query feeds the dK / dV pass
    gran = 128 if L > 224 else 16
    lo, hi = row0 // gran * gran, min(L, -(-(row0 + n) // gran) * gran)
    untouched = torch.ones(L, dtype=torch.bool)
    untouched[lo:hi] = False
    assert (win[:, untouched] == 3.0).all(), "rows outside the window's blocks must not be written"


def test_attention_softmax_extremes(lib):
    """Large score spread: one key dominates a row (exp underflow for the rest) -- no NaN, matches the oracle."""
    B, L, H = 1, 201, 1
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, L, 192, generator=g)
    qkv[0, 5, :64] *= 30  # query 5: huge logits
    qkv[0, 17, 64:128] *= 30  # key 17: huge against every query
    qkv = qkv.to(torch.float16)
    ref = O.attention(qkv.float(), H, None)
    out = torch.empty(B, L, 64, device="cuda", dtype=torch.float16)
    lse = torch.empty(B, H, 224, device="cuda")
    qc = qkv.cuda()
    ok(lib, lib.mudpt_attention_fwd(1, P(qc), P(out), P(lse), B, L, H, 0, None))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(lse[:, :, :L]).all()
    torch.testing.assert_close(out.cpu().float(), ref, atol=4e-3, rtol=4e-3)


@pytest.mark.parametrize("variant", [0, 1, 2, 4])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemm_variants_exact_integers(lib, variant, dtype):
    """Every large-problem GEMM kernel (0 = default: persistent ping-pong kernel; 1, 2, 4 = simple 256x256, 128x256,
    256x128 tiles): exact small-integer operands, ragged M, asymmetric B, K spanning several tiles -> bit-exact."""
    dt, tt = DT[dtype]
    M, N, K = 16500, 1024, 192
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = (torch.arange(N).view(N, 1) % 7 - 3 + (torch.arange(K).view(1, K) % 3)).float()
    ref = A @ B.t()
    out = torch.full((M, N), -1.0, device="cuda", dtype=torch.float32)
    gemm(lib, dt, 5, A.cuda().to(tt), B.cuda().to(tt), out0=out, variant=variant)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(33000, 768, 768), (8200, 2304, 768), (22000, 768, 3072)])
def test_gemm_pingpong_epilogues(lib, dtype, shape):
    dt, tt = DT[dtype]
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(tt)
    B = (torch.randn(N, K, generator=g) * K ** -0.5).to(tt)
    bias = torch.randn(N, generator=g)
    acc = A.float() @ B.float().t()  # fp32 CPU reference (fp64 is too slow at this size); tolerances account for it
    Ad, Bd, bd = A.cuda(), B.cuda(), bias.cuda()
    tol = dict(atol=4 * EPS[dtype], rtol=4 * EPS[dtype])
    f32tol = dict(atol=3e-5 * K ** 0.5, rtol=2e-5)
    # default dispatch (variant 0): ping-pong kernel for store / GELU / GELU' / fp32 store, simple 256x256 for residual
    out = torch.empty(M, N, device="cuda", dtype=tt)
    gemm(lib, dt, 0, Ad, Bd, bias=bd, out0=out)
    torch.testing.assert_close(out.cpu().float(), acc + bias, **tol)
    o32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    gemm(lib, dt, 5, Ad, Bd, out0=o32)
    torch.testing.assert_close(o32.cpu(), acc, **f32tol)
    u, gl = torch.empty(M, N, device="cuda", dtype=tt), torch.empty(M, N, device="cuda", dtype=tt)
    gemm(lib, dt, 1, Ad, Bd, bias=bd, out0=u, out1=gl)
    uref = acc + bias
    torch.testing.assert_close(u.cpu().float(), uref, **tol)
    torch.testing.assert_close(gl.cpu().float(), uref * torch.sigmoid(1.702 * uref), **tol)
    res = torch.randn(M, N, generator=g)
    resd = res.cuda()
    gemm(lib, dt, 2, Ad, Bd, bias=bd, out0=o32, aux=resd)
    torch.testing.assert_close(o32.cpu(), res + uref, **f32tol)
    upre = torch.randn(M, N, generator=g).to(tt)
    upred = upre.cuda()
    gemm(lib, dt, 3, Ad, Bd, out0=out, aux=upred)
    s = torch.sigmoid(1.702 * upre.float())
    torch.testing.assert_close(out.cpu().float(), acc * (s * (1 + 1.702 * upre.float() * (1 - s))), **tol)
    # repeated launches give identical bits (no race in the DMA / barrier protocol shows up as run-to-run change)
    ref_bits = out.clone()
    for _ in range(5):
        gemm(lib, dt, 3, Ad, Bd, out0=out, aux=upred)
        assert torch.equal(out, ref_bits)
