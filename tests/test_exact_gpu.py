"""The exact mode (dtype "fp32" = MUDPT_F32, what PREC = "fp32" selects) through the C ABI, against the reference fixtures at the logit scale
pretrained CLIP checkpoints carry: exp(logit_scale) = 100 (the reference multiplies the cosine by it, trainers/mudpt.py:181-182,
trainers/cocoop.py:180,191; the init value of clip/model.py:777 is 14.29).  north_star's bound -- logits within 1e-3 of the reference's
CPU path -- is then 1e-5 on the cosine: every forward GEMM operand is a [hi | lo] fp16 pair and the attention forward runs in fp32."""
import ctypes as C
import math

import pytest
import torch

from tests.helpers import GoldenCase
from tests.test_model_gpu import build

pytestmark = pytest.mark.gpu

LOGIT_ATOL_EXACT = 1e-3   # north_star, on the MAXIMUM over the logits, at logit scale 100
TINY_SLACK = 1.5          # the 3-layer tiny shapes (embed 128): a relative feature error moves the cosine by eps / sqrt(e), tests/test_model_gpu.py
GRAD_RTOL = 2e-2          # the backward is the fp16 mode's: same bound as tests/test_model_gpu.py GRAD_RTOL["fp16"]


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("B,L,H,causal", [(3, 201, 12, False), (11, 9, 8, True), (5, 77, 8, True), (2, 581, 4, False), (2, 64, 2, False),
                                           (3, 65, 1, True), (1, 1, 1, True), (2, 33, 3, False), (700, 9, 8, True)])
def test_exact_attention_forward(B, L, H, causal):
    """attention_exact.hip against a float64 softmax(QK^T / 8 [+ causal mask]) V: output (hi + lo) to fp32 rounding, the log-sum-exp, and the
    fp16 copy of q | k | v it leaves for the backward."""
    from mudpt_amd import capi
    lib = capi.load()
    g = torch.Generator().manual_seed(L * 7 + H)
    d = H * 64
    qkv = (torch.randn(B, L, 3 * d, generator=g) * 1.5).cuda()
    if L >= 33:
        qkv[0, 5, d:d + 64] *= 6.0  # one key far above the rest: the running maximum jumps inside a later tile (rescale path)
        qkv[0, L - 1, d:d + 64] *= 5.0
    Lp = lib.mudpt_attention_padded_len(L)
    hi = torch.zeros(B, L, 2 * d, device="cuda", dtype=torch.float16)  # [hi | lo] rows
    lse = torch.full((B, H, Lp), float("nan"), device="cuda")  # the kernel must write every row, the padded tail included
    lp = torch.zeros(B, L, 3 * d, device="cuda", dtype=torch.float16)
    rc = lib.mudpt_attention_fwd_exact(P(qkv), P(lp), P(hi), C.c_void_p(hi.data_ptr() + d * 2), 2 * d, P(lse), B, L, H, int(causal), None)
    assert rc == 0, lib.mudpt_last_error().decode()
    torch.cuda.synchronize()
    q, k, v = (t.reshape(B, L, H, 64).transpose(1, 2).double().cpu() for t in qkv.split(d, dim=-1))
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), dtype=torch.float64).triu_(1)
    ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, d)
    got = hi[..., :d].double().cpu() + hi[..., d:].double().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"exact attention B {B} L {L} H {H} causal {causal}: max err {err:.2e} (max |O| {scale:.2f})")
    assert err <= 8e-6 * max(scale, 1.0)  # fp32 score rounding at |s| ~ 50 (torch fp32 on the CPU: 6e-6 on this input) + the [hi | lo] pair's 2^-22
    ref_lse = torch.logsumexp(s, dim=-1)
    assert (lse[:, :, :L].double().cpu() - ref_lse).abs().max().item() <= 2e-5
    assert (lse[:, :, L:] == 0).all()  # the padded tail the whole-pair backward kernels read (as the fp16 forward kernels leave it)
    assert torch.equal(lp.cpu(), qkv.half().cpu())
    # hi is the fp16 rounding of the value, lo the remainder: |lo| <= half an ulp of hi
    ulp = torch.ldexp(torch.ones(()), torch.frexp(hi[..., :d].float().abs().clamp_min(6.2e-5)).exponent - 11).cpu()
    assert (hi[..., d:].float().abs().cpu() <= 0.5 * ulp * (1 + 1e-3)).all()


@pytest.mark.parametrize("name", ["mudpt_tiny_s100", "mudpt_vitb16_b4_s100", "mudpt_vitb16_c208_b2_s100", "mudpt_vitl14_336_b1_s100"])
def test_logits_at_scale_100_within_1e_3(name):
    case = GoldenCase(name)
    assert abs(case.frozen["logit_scale"].exp().item() - 100.0) < 1e-3
    slack = TINY_SLACK if case.cfg.v_layers < 12 else 1.0
    m = build(case, "fp32")
    logits = m(case.images).cpu()
    err = (logits - case.logits).abs().max().item()
    rms = (logits - case.logits).pow(2).mean().sqrt().item()
    print(f"{name} exact mode: |logit - reference| max {err:.3e} rms {rms:.3e}")
    assert err <= slack * LOGIT_ATOL_EXACT, err
    loss, lg = m.forward_backward(case.images, case.labels, return_logits=True)
    torch.cuda.synchronize()
    assert torch.equal(lg.cpu(), logits)  # the training step's forward is the inference forward
    assert abs(loss.item() - case.loss) <= slack * LOGIT_ATOL_EXACT
    # gradients against the REFERENCE's own (the fixture: every tensor in full, the three big projection weights as the [::8, ::8] sample
    # gen_golden.py stores; the oracle is held to the same fixtures on the CPU, tests/test_oracle_golden.py)
    for k, g in m.grads().items():
        full, sample = case.grad(k), case.grad_sample(k)
        r, g = (full, g.detach().cpu()) if full is not None else (sample, g.detach().cpu()[::8, ::8])
        rms_g = r.pow(2).mean().sqrt().item()
        e = (g - r).abs().max().item()
        print(f"  {k}: rms {rms_g:.3e} max err {e:.3e}")
        assert math.isfinite(e) and e <= GRAD_RTOL * rms_g * 4 + 1e-9, (k, e, rms_g)
        assert torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item() > 0.9995, k
    m.close()
    # for the record: the fast modes at this scale (fp16 = split text tower only; bf16 = the benchmark mode) -- sanity-bounded only
    for dtype, bound in (("fp16", 1.2e-2), ("bf16", 0.25)):
        m = build(case, dtype)
        e = (m(case.images).cpu() - case.logits).abs().max().item()
        print(f"{name} {dtype}: max {e:.3e}")
        assert e <= slack * bound
        m.close()


@pytest.mark.parametrize("name", ["mudpt_tiny", "mudpt_vitb16_b4", "mudpt_vitb16_c208_b2", "mudpt_vitl14_336_b1"])
def test_exact_mode_at_init_scale(name):
    """The same mode on the scale-14.29 fixtures (208 class prompts of mixed length: the length buckets; ViT-L/14@336: the tiled L = 581)."""
    case = GoldenCase(name)
    m = build(case, "fp32")
    logits = m(case.images).cpu()
    err = (logits - case.logits).abs().max().item()
    print(f"{name} exact mode: max {err:.3e}")
    assert err <= 2e-4 * (TINY_SLACK if case.cfg.v_layers < 12 else 1.0)
    m.close()


@pytest.mark.parametrize("name", ["cocoop_tiny_s100", "cocoop_vitb16_b2_s100"])
def test_cocoop_logits_at_scale_100_within_1e_3(name):
    from oracle import cocoop_oracle as CO
    from tests.test_cocoop_gpu import build as build_cocoop
    case = GoldenCase(name)
    slack = TINY_SLACK * 1.5 if case.cfg.v_layers < 12 else 1.0  # CoCoOp's image-feature error enters twice (tests/test_cocoop_gpu.py)
    m = build_cocoop(case.cfg, case.frozen, case.tokens, case.params, "fp32", len(case.labels))
    m.eval()
    logits = m(case.images).cpu()
    err = (logits - case.logits).abs().max().item()
    print(f"{name} exact mode: max {err:.3e}")
    assert err <= slack * LOGIT_ATOL_EXACT
    m.train()
    loss = m.forward_backward(case.images, case.labels)
    torch.cuda.synchronize()
    assert abs(loss.item() - case.loss) <= slack * LOGIT_ATOL_EXACT
    for k, g in m.grads().items():
        r = case.grad(k)
        assert torch.isfinite(g).all()
        assert torch.nn.functional.cosine_similarity(g.detach().cpu().flatten(), r.flatten(), dim=0).item() > 0.995, k
    m.close()


@pytest.mark.parametrize("M,N,K", [(804, 3072, 768), (6000, 3072, 1536), (22000, 2048, 1024)])
def test_split_gelu_epilogue(M, N, K):
    """c_fc of a split-operand tower: u (fp16) and QuickGELU(u) as a [hi | lo] pair, through the small-tile kernel (M 804) and the persistent
    kernel (its EPI_GELU_SPLIT epilogue: 48 stores per wave and tile), K = 2 x width as in the exact mode.  hi + lo must carry the fp32
    value of QuickGELU(acc + bias) to 2^-21; u is its fp16 rounding."""
    from mudpt_amd import capi
    lib = capi.load()
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g)).cuda().half()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().half()
    bias = torch.randn(N, generator=g).cuda()
    u = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    gg = torch.zeros(M, 2 * N, device="cuda", dtype=torch.float16)  # [hi | lo] rows
    rc = lib.mudpt_gemm_gelu_split(1, M, N, K, P(A), K, P(W), K, P(bias), P(u), N, P(gg), C.c_void_p(gg.data_ptr() + 2 * N), 2 * N, None)
    assert rc == 0, lib.mudpt_last_error().decode()
    torch.cuda.synchronize()
    acc = (A.double() @ W.double().t() + bias.double()).cpu()
    ref = acc * torch.sigmoid(1.702 * acc)
    got = gg[:, :N].double().cpu() + gg[:, N:].double().cpu()
    # fp32 accumulation of K products + the hardware exp / rcp of QuickGELU (1 ulp each): ~1e-6 relative; the pair itself 2^-22
    err = (got - ref).abs().max().item()
    print(f"split GELU {M}x{N}x{K}: max err {err:.2e} (max |g| {ref.abs().max():.2f})")
    assert err <= 3e-6 * max(1.0, ref.abs().max().item())
    assert (u.double().cpu() - acc).abs().max().item() <= 2.0 ** -11 * acc.abs().max().item() + 1e-6
    assert (gg[:, :N].float() - (gg[:, :N].float() + gg[:, N:].float())).abs().max().item() <= 2.0 ** -11 * gg[:, :N].float().abs().max().item()
