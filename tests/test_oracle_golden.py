"""Pin the CPU oracle to outputs of the reference's own modules (tests/golden/gen_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import mudpt_oracle as O
from tests.helpers import GoldenCase


@pytest.fixture(scope="module", params=["mudpt_tiny", "mudpt_vitb16_b4", "mudpt_vitl14_336_b1", "mudpt_vitb16_n2_d9_b2"])  # the last: n_ctx 2, depth 9 -- what the reference's scripts train
def case(request):
    c = GoldenCase(request.param)
    c.check_recipe()
    return c


def test_forward_backward_matches_reference(case):
    loss, logits, grads = O.forward_backward(case.cfg, case.frozen, case.params, case.class_embedding, case.eot,
                                             case.images, case.labels)
    # fp32 on both sides; only the summation order differs (batch-first vs LND, explicit softmax)
    torch.testing.assert_close(logits, case.logits, atol=2e-5, rtol=1e-5)
    assert abs(loss.item() - case.loss) < 1e-5
    for k in O.TRAINABLE_ORDER:
        g = grads[k]
        s = case.z["grad_sum." + k]
        assert abs(g.double().pow(2).sum().sqrt().item() - s[1]) <= 1e-4 * s[1] + 1e-9, k
        full, sample = case.grad(k), case.grad_sample(k)
        scale = float(s[1]) / max(g.numel() ** 0.5, 1.0)  # rms of the gradient tensor
        if full is not None:
            torch.testing.assert_close(g, full, atol=1e-3 * scale + 1e-9, rtol=1e-4)
        else:
            torch.testing.assert_close(g[::8, ::8], sample, atol=1e-3 * scale + 1e-9, rtol=1e-4)


def test_many_class_fixture_matches_reference():
    """208 class prompts of mixed length (EOT 7..25), B = 2: the shape of work of BASELINE configs[2] (C = 1000) at fixture size."""
    c = GoldenCase("mudpt_vitb16_c208_b2")
    c.check_recipe()
    assert c.tokens.shape == (208, 77) and int(c.eot.min()) == 7 and int(c.eot.max()) >= 16 and len(set(c.eot.tolist())) >= 10
    test_forward_backward_matches_reference(c)


@pytest.mark.parametrize("name", ["mudpt_tiny_s100", "mudpt_vitb16_b4_s100", "mudpt_vitb16_c208_b2_s100", "mudpt_vitl14_336_b1_s100"])
def test_scale100_fixture_matches_reference(name):
    """exp(logit_scale) = 100, what every released CLIP checkpoint holds (the reference multiplies the cosine by it, trainers/mudpt.py:181-182;
    init value 1/0.07 = 14.29, clip/model.py:777): the fixtures come from the reference's modules with that one parameter changed."""
    import math
    c = GoldenCase(name)
    c.check_recipe()
    assert abs(c.frozen["logit_scale"].exp().item() - 100.0) < 1e-3 and abs(float(c.z["logit_scale"]) - math.log(100.0)) < 1e-6
    loss, logits, grads = O.forward_backward(c.cfg, c.frozen, c.params, c.class_embedding, c.eot, c.images, c.labels)
    err = (logits - c.logits).abs().max().item()
    print(f"{name}: oracle vs reference logits max {err:.3e}")
    # fp32 on both sides: the summation-order noise of the cosine (~1e-6) times 100
    torch.testing.assert_close(logits, c.logits, atol=1.5e-4, rtol=1e-5)
    assert abs(loss.item() - c.loss) < (2e-5 if c.logits.shape[1] <= 11 else 1e-4)  # 208 classes: the loss sums the logits' fp32 noise over more terms
    for k in O.TRAINABLE_ORDER:
        g, s = grads[k], c.z["grad_sum." + k]
        scale = float(s[1]) / max(g.numel() ** 0.5, 1.0)
        full, sample = c.grad(k), c.grad_sample(k)
        if full is not None:
            torch.testing.assert_close(g, full, atol=1e-3 * scale + 1e-9, rtol=1e-4)
        else:
            torch.testing.assert_close(g[::8, ::8], sample, atol=1e-3 * scale + 1e-9, rtol=1e-4)


def test_c1000_oracle_fixture_is_current():
    """tests/golden/oracle_vitb16_c1000_b2.npz holds the ORACLE's own outputs at C = 1000 (the GPU suite compares against it instead of
    recomputing them: 65 s).  Re-derive its logits here (forward only, ~15 s) so that a change to the oracle cannot leave it stale."""
    import os
    from tests.golden import gen_oracle_c1000 as G
    cfg, frozen, tok, params, images, labels = G.inputs()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vitb16_c1000_b2.npz"), allow_pickle=False)
    assert int(z["tokens_checksum"]) == int(tok.sum())
    with torch.no_grad():
        logits = O.forward(cfg, frozen, params, frozen["token_embedding.weight"][tok], tok.argmax(-1), images)
    torch.testing.assert_close(logits, torch.from_numpy(z["logits"]), atol=1e-6, rtol=1e-6)
    loss = torch.nn.functional.cross_entropy(logits, labels).item()
    assert abs(loss - float(z["loss"])) < 1e-6


def test_block_outputs_match_reference(case):
    taps = {}
    with torch.no_grad():
        O.forward(case.cfg, case.frozen, case.params, case.class_embedding, case.eot, case.images, taps)
    n = 0
    for k in case.z.files:
        if not k.startswith("tap."):
            continue
        ref = torch.from_numpy(case.z[k])
        got = taps[k[4:]]
        if got.shape != ref.shape:
            got = got[:, ::8, ::16]
        torch.testing.assert_close(got, ref, atol=2e-4, rtol=1e-4)
        n += 1
    assert n >= (6 if case.cfg.depth <= case.cfg.t_layers else 5)  # ViT-L/14: the sampled tap of block 23 exists for the vision tower only


def test_tokenizer_fixture_shape(case):
    tok = case.tokens
    assert tok.shape == (11, 77)
    assert (tok[:, 0] == 49406).all() and (tok.max(dim=-1).values == 49407).all()
    # "a photo of a <name>." -> EOT at 7 for one-token names, 8 for "binocular" (n_ctx 4)
    if case.cfg.n_ctx == 4:
        assert case.eot.tolist() == [7] * 10 + [8]


def test_flat_bucket_roundtrip():
    cfg = O.TINY
    p = O.make_trainable_state(cfg, 3)
    flat = O.flatten(p)
    assert flat.numel() == sum(v.numel() for v in p.values())
    q = O.unflatten(flat, cfg)
    for k in O.TRAINABLE_ORDER:
        assert torch.equal(p[k], q[k])
    assert O.flatten(O.make_trainable_state(O.VIT_B16, 1)).numel() == 1243136


@pytest.mark.parametrize("name", ["cocoop_tiny", "cocoop_vitb16_b2", "cocoop_vitb16_c48_b2", "cocoop_vitb32_b1", "cocoop_tiny_s100", "cocoop_vitb16_b2_s100"])
def test_cocoop_forward_backward_matches_reference(name):
    """The CoCoOp restatement (oracle/cocoop_oracle.py) against the reference's trainers/cocoop.py CustomCLIP: eval-mode
    logits, training-mode loss (cross-entropy inside forward) and the gradients of ctx and meta_net."""
    from oracle import cocoop_oracle as CO
    case = GoldenCase(name)
    case.check_recipe()
    loss, logits, grads = CO.forward_backward(case.cfg, case.frozen, case.params, case.class_embedding, case.eot, case.images, case.labels)
    s100 = name.endswith("_s100")  # the cosine's summation-order noise is multiplied by 100 instead of 14.29
    torch.testing.assert_close(logits, case.logits, atol=1.5e-4 if s100 else 2e-5, rtol=1e-5)
    assert abs(loss.item() - case.loss) < (2e-5 if s100 else 1e-5)
    for k in CO.TRAINABLE_ORDER:
        ref = case.grad(k)
        scale = ref.pow(2).mean().sqrt().item()
        torch.testing.assert_close(grads[k], ref, atol=1e-3 * scale + 1e-9, rtol=1e-4)
    assert CO.flatten(case.params).numel() == sum(int(np.prod(s)) for s in CO.trainable_shapes(case.cfg).values())


@pytest.mark.parametrize("nesterov,dampening", [(False, 0.0), (True, 0.0), (False, 0.1)])
def test_sgd_restatement_equals_torch_optim(nesterov, dampening):
    """oracle.sgd_step -- the checker of the library's fused SGD -- against torch.optim.SGD itself (what Dassl's build_optimizer("sgd") runs
    behind trainers/mudpt.py:225,251) over four steps with momentum, weight decay, and the nesterov / dampening variants."""
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(257, generator=g)
    grads = [torch.randn(257, generator=g) for _ in range(4)]
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=5e-4, dampening=dampening, nesterov=nesterov)
    q, buf = p0.clone(), None
    for gr in grads:
        p.grad = gr.clone()
        opt.step()
        q, buf = O.sgd_step(q, gr, buf, 0.01, 0.9, 5e-4, dampening, nesterov)
        torch.testing.assert_close(q, p.detach(), atol=1e-7, rtol=1e-6)
