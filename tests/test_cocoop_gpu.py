"""GPU parity of the CoCoOp path (BASELINE configs[3]) through the C ABI: the HIP library with
``variant = "cocoop"`` against the golden vectors of the reference's own ``trainers/cocoop.py`` and the CPU oracle."""
import pytest
import torch

from oracle import cocoop_oracle as CO
from oracle import mudpt_oracle as O
from tests.helpers import GoldenCase, assert_training_forward_is_the_inference_forward

pytestmark = pytest.mark.gpu

# Same bounds as tests/test_model_gpu.py: fp16 (text tower with split operands) within north_star's 1e-3 on the maximum for
# ViT-B/16 (measured max 3.1e-4), 1.5x that for the tiny shape (measured 9.5e-4); logit scale 14.29.
LOGIT_RMS = {"fp16": 5e-4, "bf16": 1.6e-2}
LOGIT_ATOL = {"fp16": 1e-3, "bf16": 3.2e-2}
TINY_SLACK = 1.5
# Gradients.  Error model: every gradient here is a SUM of per-(image, class) terms -- d ctx = sum_i sum_c T[i, c], d bias_i = sum_c T[i, c]
# (T[i, c] = the n_ctx context rows of the text-input gradient of prompt (i, c)), and meta_net's gradients are linear images of d bias.
# Each term comes out of the T-precision text-tower backward with a relative error delta_T (the same figure the non-cancelling MuDPT
# gradients are held to: tests/test_model_gpu.py GRAD_RTOL); the errors of different terms are independent, so the error of a sum has
# RMS delta_T * sqrt(sum T^2), while the signal is |sum T|.  Because sum_c dlogits[i, c] = 0 the terms nearly cancel:
# kappa = ||sqrt(sum T^2)|| / ||sum T|| >= 1 measures by how much (computed below from the ORACLE's own terms, per tensor).  Bounds:
# RMS error <= delta_T * kappa * rms(gradient); single elements <= 4 * delta_T * kappa * max|gradient| (meta_net's weight gradients
# are outer products with heavy tails: an element's error scales with the element, not with the tensor's RMS).
# delta_T = relative error of ONE term out of the 12-layer T-precision backward: measured 0.8-1.6e-3 (fp16) and 1.1-2.5e-2 (bf16) on the
# two fixtures (the printed rms err / kappa); the bounds leave a factor 1.6-2.5.
DELTA_T = {"fp16": 4e-3, "bf16": 4e-2}


def cancellation_factors(cfg, dprompts):
    """kappa per trainable from the per-(image, class) terms T = dprompts[:, :, 1 : 1 + n_ctx] ([B, C, n, d])."""
    T = dprompts[:, :, 1:1 + cfg.n_ctx].double()
    quad_ctx, sig_ctx = T.pow(2).sum((0, 1)).sqrt(), T.sum((0, 1))          # [n, d]
    quad_b, sig_b = T.pow(2).sum((1, 2)).sqrt(), T.sum((1, 2))              # [B, d]
    k_ctx = (quad_ctx.norm() / sig_ctx.norm()).item()
    k_b = (quad_b.norm() / sig_b.norm()).item()
    return {"prompt_learner.ctx": k_ctx, "meta": k_b}


def build(cfg, frozen, tokens, params, dtype, max_batch, knobs=None):
    from mudpt_amd.model import CustomCLIP, ModelShape
    shape = ModelShape(cfg.image_size, cfg.patch, cfg.v_width, cfg.v_layers, cfg.v_heads, cfg.t_width, cfg.t_layers, cfg.t_heads,
                       cfg.ctx_len, cfg.embed_dim, cfg.n_ctx, 1)
    m = CustomCLIP(shape, frozen, tokens, max_batch=max_batch, dtype=dtype, variant="cocoop", knobs=knobs)
    assert m.param_names == CO.TRAINABLE_ORDER  # the reference's prompt_learner.* names, flat-bucket order
    m.set_params(params)
    return m


# cocoop_vitb16_c48_b2: 48 class names of 1-9 words from the reference (EOT rows 7..25): the trimmed text tower runs 26 positions with a different
# EOT row per class, 96 sequences per step
@pytest.fixture(scope="module", params=["cocoop_tiny", "cocoop_vitb16_b2", "cocoop_vitb16_c48_b2", "cocoop_vitb32_b1"])  # the last: the reference's CoCoOp yaml (ViT-B/32, batch 1)
def case(request):
    return GoldenCase(request.param)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_logits_loss_grads_match_reference(case, dtype):
    m = build(case.cfg, case.frozen, case.tokens, case.params, dtype, len(case.labels))
    m.eval()
    logits = m(case.images).cpu()  # eval mode: logits (trainers/cocoop.py:198)
    err, rms = (logits - case.logits).abs().max().item(), (logits - case.logits).pow(2).mean().sqrt().item()
    print(f"{dtype}: |logit - reference| max {err:.3e} rms {rms:.3e}")
    slack = TINY_SLACK if case.cfg.v_layers < 12 else 1.0
    assert rms <= slack * LOGIT_RMS[dtype] and err <= slack * LOGIT_ATOL[dtype]
    m.train()
    loss, logits2 = m.forward_backward(case.images, case.labels, return_logits=True)  # training mode: CE inside forward (:196-197)
    torch.cuda.synchronize()
    assert_training_forward_is_the_inference_forward(logits2, logits, dtype)  # same kernels; a tiny training batch may split K (tests/helpers.py)
    assert abs(loss.item() - case.loss) <= slack * LOGIT_ATOL[dtype]
    got = {k: v.detach().cpu() for k, v in m.grads().items()}
    taps = {}
    CO.forward_backward(case.cfg, case.frozen, case.params, case.class_embedding, case.eot, case.images, case.labels, taps)
    kappa = cancellation_factors(case.cfg, taps["dprompts"])
    bad = []
    for k in CO.TRAINABLE_ORDER:
        r = case.grad(k)
        rms_g = r.pow(2).mean().sqrt().item()
        e, er = (got[k] - r).abs().max().item(), (got[k] - r).pow(2).mean().sqrt().item()
        kap = kappa.get(k, kappa["meta"])
        bound, gmax = DELTA_T[dtype] * kap, r.abs().max().item()
        print(f"{dtype} {k}: rms {rms_g:.3e} kappa {kap:.2f} rms err {er / rms_g:.3e} (bound {bound:.3e}) max err / max|g| {e / gmax:.3e} (bound {4 * bound:.3e})")
        if er > bound * rms_g + 1e-9 or e > 4 * bound * gmax + 1e-9:
            bad.append(k)
    m.close()
    assert not bad, bad


@pytest.mark.parametrize("chunk", [0, 4, 1])
def test_larger_batch_against_oracle_and_sgd(chunk):
    """B = 6 images x 11 classes = 66 text sequences of the tiny shape, smaller batch than max_batch, then two SGD steps
    (torch.optim.SGD semantics, as Dassl's build_optimizer) tracked against the oracle.  chunk: images per text-tower pass
    (0 = all that fit; 4 = a full and a partial pass; 1 = the reference's own image-by-image loop, trainers/cocoop.py:187-194)."""
    cfg = O.TINY
    frozen = O.make_frozen_state(cfg, 31)
    tok = O.synthetic_tokens(cfg, 11).long()
    params = CO.make_trainable_state(cfg, 32)
    g = torch.Generator().manual_seed(33)
    images, labels = torch.randn(6, 3, cfg.image_size, cfg.image_size, generator=g), torch.randint(0, 11, (6,), generator=g)
    emb, eot = frozen["token_embedding.weight"][tok], tok.argmax(-1)
    m = build(cfg, frozen, tok, params, "fp16", max_batch=8, knobs={"cocoop_chunk": chunk})
    p, bufs = {k: v.clone() for k, v in params.items()}, {k: None for k in params}
    for step in range(2):
        loss, logits = m.forward_backward(images, labels, return_logits=True)
        ref_loss, ref_logits, ref = CO.forward_backward(cfg, frozen, p, emb, eot, images, labels)
        # logit error model: logit = scale * cos(img, txt); a relative feature error eps in a random direction changes the cosine of two
        # e-dimensional unit vectors by ~eps / sqrt(e), so at equal eps the tiny shape (e = 128) is sqrt(512 / 128) = 2 x worse than
        # ViT-B/16's 1e-3 (north_star), and CoCoOp's image-feature error enters twice (directly and through meta_net's context shift): 1.5 x
        assert (logits.cpu() - ref_logits).abs().max().item() <= 1e-3 * (512 / cfg.embed_dim) ** 0.5 * 1.5
        assert abs(loss.item() - ref_loss.item()) <= 2e-3
        for k in CO.TRAINABLE_ORDER:
            r = ref[k]
            assert (m.grads()[k].cpu() - r).pow(2).mean().sqrt().item() <= 2e-2 * r.pow(2).mean().sqrt().item() + 1e-9, (step, k)
        m.sgd_step(0.002)
        for k in CO.TRAINABLE_ORDER:
            p[k], bufs[k] = O.sgd_step(p[k], ref[k], bufs[k], 0.002)
    torch.cuda.synchronize()
    for k, v in m.named_parameters():
        assert (v.detach().cpu() - p[k]).abs().max().item() <= 1e-4, k
    m.close()


@pytest.mark.parametrize("n_cls,max_batch", [(100, 100), (1000, 100)])
def test_reference_config_sizes_fit_and_run(n_cls, max_batch):
    """The reference's own CoCoOp config is train batch 1 / test batch 100 (configs/trainers/CoCoOp/vit_b16_c4_ep10_batch1_ctxv1.yaml)
    on class sets up to ImageNet's 1000: B * C = 100 000 prompts per test batch.  The text tower's activations are sized for a chunk
    of images, not for max_batch * C * 77 rows (1.1 TB): create succeeds, a 100-image eval batch runs in several passes, chunks of the
    batch give the same logits bit for bit, and a batch-1 training step works on the same handle."""
    from mudpt_amd import synth
    from mudpt_amd.model import CustomCLIP, ModelShape
    shape = ModelShape(depth=1)
    m = CustomCLIP(shape, synth.random_clip_state(shape, seed=0), synth.synthetic_tokenized_prompts(n_cls), ctx_token_ids=synth.CTX_INIT_TOKENS,
                   max_batch=max_batch, dtype="bf16", seed=1, variant="cocoop")
    g = torch.Generator().manual_seed(3)
    images = torch.randn(max_batch, 3, 224, 224, generator=g).cuda()
    m.eval()
    full = m(images)
    assert full.shape == (max_batch, n_cls) and torch.isfinite(full).all()
    part = torch.cat([m(images[i:i + 25]) for i in range(0, max_batch, 25)])
    assert torch.equal(full, part)
    m.train()
    loss = m.forward_backward(images[:1], torch.tensor([n_cls - 1]).cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and torch.isfinite(m.flat_grads).all() and m.flat_grads.abs().sum() > 0
    m.close()
