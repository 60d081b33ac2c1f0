"""Edge cases and full-size properties of the MuDPT path on the GPU (through the C ABI).

* shapes the golden vectors do not cover: depth 1 (no deep prompts), a single context token, batch 1, one class, a batch smaller
  than max_batch after a larger one (stale buffers);
* BASELINE configs[1]'s full size (ViT-B/16, 256 images, bf16), where the CPU oracle is too slow: size-independent properties --
  images are independent units, so the logits of a batch equal the logits of its chunks BITWISE and follow a permutation of the
  images; the batch gradient is the mean of the chunk gradients."""
import dataclasses

import pytest
import torch

from oracle import mudpt_oracle as O

pytestmark = pytest.mark.gpu


def build(cfg, frozen, tokens, params, dtype, max_batch):
    from mudpt_amd.model import CustomCLIP, ModelShape
    shape = ModelShape(cfg.image_size, cfg.patch, cfg.v_width, cfg.v_layers, cfg.v_heads, cfg.t_width, cfg.t_layers, cfg.t_heads,
                       cfg.ctx_len, cfg.embed_dim, cfg.n_ctx, cfg.depth)
    m = CustomCLIP(shape, frozen, tokens, max_batch=max_batch, dtype=dtype)
    m.set_params(params)
    return m


@pytest.mark.parametrize("n_ctx,depth,batch,n_cls", [(1, 1, 1, 3), (2, 1, 2, 1), (3, 3, 1, 2), (2, 5, 2, 4)])
def test_small_shapes_against_the_oracle(n_ctx, depth, batch, n_cls):
    """depth 1 = no deep prompts at all; depth 5 > the tiny towers' 3 layers = surplus prompts never consumed (zero gradient,
    SURVEY appendix A.7); one class = zero loss and zero gradients; batch 1."""
    cfg = dataclasses.replace(O.TINY, n_ctx=n_ctx, depth=depth)
    frozen = O.make_frozen_state(cfg, 41)
    tok = O.synthetic_tokens(cfg, n_cls).long()
    params = O.make_trainable_state(cfg, 42)
    g = torch.Generator().manual_seed(43)
    images, labels = torch.randn(batch, 3, cfg.image_size, cfg.image_size, generator=g), torch.randint(0, n_cls, (batch,), generator=g)
    ref_loss, ref_logits, ref = O.forward_backward(cfg, frozen, params, frozen["token_embedding.weight"][tok], tok.argmax(-1), images, labels)
    m = build(cfg, frozen, tok, params, "fp16", batch)
    loss, logits = m.forward_backward(images, labels, return_logits=True)
    torch.cuda.synchronize()
    # logit error model (tests/test_cocoop_gpu.py): a relative feature error eps moves the cosine of two e-dimensional unit vectors by
    # ~eps / sqrt(e): the tiny shape (e = 128) gets sqrt(512 / 128) = 2 x ViT-B/16's 1e-3 (north_star)
    tol = 1e-3 * (512 / cfg.embed_dim) ** 0.5
    assert (logits.cpu() - ref_logits).abs().max().item() <= tol
    assert abs(loss.item() - ref_loss.item()) <= tol
    for k, gr in m.grads().items():
        r = ref[k]
        assert r.shape == gr.shape, k
        if r.numel() == 0:  # depth 1: the deep-prompt tensors are empty
            continue
        assert (gr.cpu() - r).abs().max().item() <= 3e-2 * r.pow(2).mean().sqrt().item() + 1e-7, k
    m.close()


def test_smaller_batch_after_a_larger_one_is_unaffected_by_stale_buffers():
    cfg = O.TINY
    frozen = O.make_frozen_state(cfg, 51)
    tok = O.synthetic_tokens(cfg, 5).long()
    params = O.make_trainable_state(cfg, 52)
    g = torch.Generator().manual_seed(53)
    big, small = torch.randn(6, 3, cfg.image_size, cfg.image_size, generator=g), torch.randn(2, 3, cfg.image_size, cfg.image_size, generator=g)
    yb, ys = torch.randint(0, 5, (6,), generator=g), torch.randint(0, 5, (2,), generator=g)
    m = build(cfg, frozen, tok, params, "fp16", 6)
    m.forward_backward(big, yb)
    loss_a, logits_a = m.forward_backward(small, ys, return_logits=True)
    grads_a = {k: v.clone() for k, v in m.grads().items()}
    fresh = build(cfg, frozen, tok, params, "fp16", 2)
    loss_b, logits_b = fresh.forward_backward(small, ys, return_logits=True)
    torch.cuda.synchronize()
    assert torch.equal(logits_a, logits_b) and loss_a.item() == loss_b.item()
    for k, v in fresh.grads().items():
        assert torch.equal(v, grads_a[k]), k
    m.close()
    fresh.close()


def test_full_size_batch_properties_bf16():
    """BASELINE configs[1]: B = 256, ViT-B/16, 11 classes, bf16 (the benchmark configuration)."""
    from mudpt_amd import synth
    from mudpt_amd.model import CustomCLIP, ModelShape
    shape = ModelShape()
    m = CustomCLIP(shape, synth.random_clip_state(shape, seed=0), synth.bench_tokenized_prompts(), ctx_token_ids=synth.CTX_INIT_TOKENS,
                   max_batch=256, dtype="bf16", seed=1)
    g = torch.Generator().manual_seed(7)
    images = torch.randn(256, 3, 224, 224, generator=g).cuda()
    labels = torch.randint(0, 11, (256,), generator=g).cuda()
    m.eval()
    full = m(images)
    assert torch.isfinite(full).all() and full.shape == (256, 11)
    # images are independent units: chunks of the batch give the same rows, bit for bit (every kernel's per-row arithmetic order is
    # independent of the row's position: GEMM tiles incl. the half-tile tail, LayerNorm rows, attention per (image, head))
    chunks = torch.cat([m(images[i:i + 64]) for i in range(0, 256, 64)])
    assert torch.equal(full, chunks)
    perm = torch.randperm(256, generator=g).cuda()
    assert torch.equal(m(images[perm]), full[perm])
    # the loss is the mean over the images and the gradient the mean of the chunk gradients
    m.train()
    loss = m.forward_backward(images, labels).item()
    grad = m.flat_grads.clone()
    acc, losses = torch.zeros_like(grad), []
    for i in range(0, 256, 64):
        losses.append(m.forward_backward(images[i:i + 64], labels[i:i + 64]).item())
        acc += m.flat_grads / 4
    assert abs(loss - sum(losses) / 4) <= 1e-5 * max(1.0, abs(loss))
    # Both sides are bf16-mode gradients (8-bit token gradients, 3-7 % RMS error against fp32 each, tests/test_model_gpu.py), computed
    # from differently scaled sums (the text tower sees the sum over 256 vs 64 images): they agree to that noise, tensor by tensor.
    off = 0
    for k, p in m.named_parameters():
        a, b = grad[off:off + p.numel()], acc[off:off + p.numel()]
        off += p.numel()
        rms = a.pow(2).mean().sqrt().item()
        assert rms > 0, k
        assert (a - b).pow(2).mean().sqrt().item() <= 0.1 * rms, k
    m.close()


def test_split_operand_knobs_are_refused_where_the_buffers_do_not_exist():
    """The split-operand knobs (include/mudpt.h MUDPT_F32) only exist where the mode keeps low halves: a bf16 handle has none, an fp16 handle
    only in its text tower, and e4m3 remainders need the e4m3 weight copies (vision tower of the parity mode)."""
    from mudpt_amd import capi
    cfg = O.TINY
    frozen = O.make_frozen_state(cfg, 61)
    tok = O.synthetic_tokens(cfg, 3).long()
    params = O.make_trainable_state(cfg, 62)
    m = build(cfg, frozen, tok, params, "bf16", 1)
    for knob, v in (("vis_lo", 1), ("txt_lo", 1), ("vis_exact_attn", 1), ("txt_exact_attn", 1)):
        with pytest.raises(capi.MudptError):
            m.set_knob(knob, v)
    m.set_knob("vis_lo", 0)   # switching OFF is always fine
    m.close()
    m = build(cfg, frozen, tok, params, "fp16", 1)
    m.set_knob("txt_lo", 0)
    m.set_knob("txt_lo", 1)
    with pytest.raises(AssertionError):   # e4m3 remainders: no e4m3 weights in the text tower (bad argument)
        m.set_knob("txt_lo", 2)
    with pytest.raises(capi.MudptError):
        m.set_knob("vis_lo", 2)
    with pytest.raises(capi.MudptError):
        m.set_knob("txt_exact_attn", 1)  # the fp32 attention forward belongs to the parity mode
    m.close()
    m = build(cfg, frozen, tok, params, "fp32", 1)
    for knob, v in (("vis_lo", 0), ("vis_lo", 1), ("vis_lo", 2), ("vis_sites", 12), ("txt_sites", 15), ("vis_exact_attn", 1), ("txt_exact_attn", 0), ("txt_lo", 0)):
        m.set_knob(knob, v)
    with pytest.raises(AssertionError):
        m.set_knob("txt_lo", 2)
    g = torch.Generator().manual_seed(63)
    assert torch.isfinite(m(torch.randn(1, 3, cfg.image_size, cfg.image_size, generator=g))).all()  # and the handle still runs with the knobs moved
    m.close()
