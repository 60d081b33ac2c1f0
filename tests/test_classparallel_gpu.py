"""Class-parallel text tower (SURVEY 8e second axis; include/mudpt.h mudpt_set_class_shard / mudpt_cp_*).

The reference runs all C class prompts through the text tower on every replica (trainers/mudpt.py:142-156 inside nn.DataParallel,
:230-233).  Here rank r may encode classes [c0, c1) only; two sums of a [C, embed] table (features forward, their gradient backward)
complete the step.  Pins:

* the phases on an UNSHARDED handle equal the monolithic step bit for bit (same kernels, same order);
* two sharded handles in ONE process with the exchanges done by hand (uneven shards 6 + 5 of 11 classes, half the batch each) give the
  reference fixture's logits / loss / gradients within the usual bounds, and equal the unsharded run up to the summation order;
* two PROCESSES (gloo, both on the test box's one GPU) through ``CustomCLIP(class_shard=...)`` and torch.distributed on the 208-class
  fixture: logits, loss, gradients against the reference's numbers; sharded eval forward with and without the text cache;
* a sharded handle refuses the monolithic entry points.
"""
import os
import subprocess
import sys

import pytest
import torch

from oracle import mudpt_oracle as O
from tests.helpers import FWD_SPLIT_TOL, GoldenCase
from tests.test_manyclass_gpu import build, check_grads, LOGIT_ATOL, GRAD_RTOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shape_of(cfg):
    from mudpt_amd.model import ModelShape
    return ModelShape(cfg.image_size, cfg.patch, cfg.v_width, cfg.v_layers, cfg.v_heads, cfg.t_width, cfg.t_layers, cfg.t_heads,
                      cfg.ctx_len, cfg.embed_dim, cfg.n_ctx, cfg.depth)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_phases_on_an_unsharded_handle_equal_the_monolithic_step(dtype):
    c = GoldenCase("mudpt_vitb16_b4")
    m = build(c.cfg, c.frozen, c.tokens, c.params, dtype, 4)
    loss, logits = m.forward_backward(c.images, c.labels, return_logits=True)
    torch.cuda.synchronize()
    ref = (loss.item(), logits.clone(), m.flat_grads.clone())
    loss, logits = m.forward_backward_cp(c.images, c.labels, return_logits=True)
    torch.cuda.synchronize()
    assert loss.item() == ref[0] and torch.equal(logits, ref[1]) and torch.equal(m.flat_grads, ref[2])
    m.close()


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_two_sharded_handles_with_manual_exchange_match_reference_and_unsharded(dtype):
    from mudpt_amd import capi, parallel
    from mudpt_amd.model import CustomCLIP
    c = GoldenCase("mudpt_vitb16_b4")
    C, B, world = len(c.tokens), len(c.labels), 2
    full = build(c.cfg, c.frozen, c.tokens, c.params, dtype, B)
    # the unsharded handle on each rank's half batch: the logits the sharded ranks must reproduce BIT FOR BIT (same images per launch, so the
    # vision tower runs the same kernels; what differs is only who encoded which class) ...
    logits_h = []
    for r in range(world):
        _, lg = full.forward_backward(c.images[r * (B // world):(r + 1) * (B // world)], c.labels[r * (B // world):(r + 1) * (B // world)], return_logits=True)
        torch.cuda.synchronize()
        logits_h.append(lg.cpu().clone())
    logits_h = torch.cat(logits_h)
    # ... and on the whole batch: loss and gradients of the job.  (Its logits agree with the half-batch ones to rounding only: a TRAINING
    # forward of a few images splits the contraction of c_proj into a number of slices that depends on the row count)
    loss_f, logits_f = full.forward_backward(c.images, c.labels, return_logits=True)
    torch.cuda.synchronize()
    grads_f = {k: g.detach().cpu().clone() for k, g in full.grads().items()}
    logits_f = logits_f.cpu()
    full.close()

    ranks = []
    for r in range(world):
        shard = parallel.class_range(C, r, world)
        m = CustomCLIP(shape_of(c.cfg), c.frozen, c.tokens, max_batch=B // world, dtype=dtype, class_shard=shard)
        m.set_params(c.params)
        ranks.append(m)
    assert [m.class_shard for m in ranks] == [(0, 6), (6, 11)]
    lib, st = ranks[0].lib, ranks[0]._stream()
    per = B // world
    img = [c.images[r * per:(r + 1) * per].cuda().contiguous() for r in range(world)]
    lab = [c.labels[r * per:(r + 1) * per].cuda().contiguous() for r in range(world)]
    for r, m in enumerate(ranks):
        capi.check(lib.mudpt_cp_forward(m._h, capi.ptr(img[r]), per, capi.FWD_TRAINING, st), "cp_forward")
    torch.cuda.synchronize()
    # exchange 1: rows of the other rank are zero -> the sum is the gather
    for r, m in enumerate(ranks):
        c0, c1 = m.class_shard
        assert (m._cp_feat[:c0] == 0).all() and (m._cp_feat[c1:] == 0).all() and (m._cp_feat[c0:c1] != 0).any()
    table = ranks[0]._cp_feat + ranks[1]._cp_feat
    for m in ranks:
        m._cp_feat.copy_(table)
    logits, losses = [], []
    for r, m in enumerate(ranks):
        lg = torch.empty(per, C, device="cuda")
        capi.check(lib.mudpt_cp_head(m._h, capi.ptr(lab[r]), per, 1.0 / world, capi.ptr(m._loss), capi.ptr(lg), 0, st), "cp_head")
        capi.check(lib.mudpt_cp_backward(m._h, capi.CP_VISION, st), "cp_backward")
        logits.append(lg)
    torch.cuda.synchronize()
    dtable = ranks[0]._cp_dfeat + ranks[1]._cp_dfeat  # exchange 2
    for m in ranks:
        m._cp_dfeat.copy_(dtable)
        capi.check(lib.mudpt_cp_backward(m._h, capi.CP_TEXT, st), "cp_backward")
    torch.cuda.synchronize()
    logits = torch.cat(logits).cpu()
    loss = sum(m._loss[0].item() for m in ranks) / world
    bucket = ranks[0].flat_grads + ranks[1].flat_grads  # the gradient all-reduce
    grads = {}
    for k, p in ranks[0].named_parameters():
        off = p.data_ptr() - ranks[0].flat_params.data_ptr()
        grads[k] = bucket[off // 4:off // 4 + p.numel()].view_as(p).cpu()
    for m in ranks:
        m.close()
    # against the reference fixture
    assert (logits - c.logits).abs().max().item() <= LOGIT_ATOL[dtype]
    assert abs(loss - c.loss) <= (2e-4 if dtype == "fp16" else 5e-3) * max(1.0, abs(c.loss))
    ref = {k: c.grad(k) for k in O.TRAINABLE_ORDER}
    if all(v is not None for v in ref.values()):
        check_grads(grads, ref, dtype, "class-parallel")
    # against the unsharded run.  Forward: bit for bit.  Backward: d(features) summed over the two half batches equals the full-batch
    # table to ~1e-7, but the T-precision roundings of the 12-block text backward re-randomise under ANY perturbation of their input
    # (each flipped rounding moves every later one), so the text-side gradients agree at the rounding-noise level only -- the level both
    # runs hold against the reference.  The exact statement (same batch, same table: sharded == unsharded to 1e-5) is the next test.
    assert torch.equal(logits, logits_h), "logits do not depend on which rank encoded a class"
    assert (logits - logits_f).abs().max().item() <= FWD_SPLIT_TOL[dtype]
    assert abs(loss - loss_f.item()) <= (1e-6 if FWD_SPLIT_TOL[dtype] == 0 else FWD_SPLIT_TOL[dtype]) * max(1.0, abs(loss))
    for k, g in grads_f.items():
        rms = g.pow(2).mean().sqrt().item()
        err = (grads[k] - g).abs().max().item()
        print(f"{dtype} {k}: sharded - unsharded {err / rms:.3e} x rms")
        assert err <= 4 * GRAD_RTOL[dtype] * rms + 1e-12, k


@pytest.mark.parametrize("shards", [[(0, 6), (6, 11)], [(0, 1), (1, 11)], [(0, 4), (4, 8), (8, 11)]])
def test_text_tower_sharding_is_exact_given_the_same_tables(shards):
    """Same images on every handle, the feature table and its gradient taken from the unsharded handle: the text tower's forward rows are
    bit-identical whichever handle encodes a class (also when a shard trims to a shorter max(EOT)), and the purely text-side gradients
    (visual_ctx_deep_projections.*: fed by d_txt_deep only) sum over the shards to the unsharded ones up to the order of the fp32 sums
    over classes."""
    from mudpt_amd import capi
    from mudpt_amd.model import CustomCLIP
    c = GoldenCase("mudpt_vitb16_b4")
    B = len(c.labels)
    img, lab = c.images.cuda(), c.labels.cuda()
    keys = ["image_encoder.visual_ctx_deep_projections.weight", "image_encoder.visual_ctx_deep_projections.bias"]
    vis_keys = ["image_encoder.visual_ctx", "mudpt_prompt_learner.embed_projection.weight", "mudpt_prompt_learner.deep_projections.weight"]

    def run(shard, table=None, dtable=None):
        m = CustomCLIP(shape_of(c.cfg), c.frozen, c.tokens, max_batch=B, dtype="fp16", class_shard=shard)
        m.set_params(c.params)
        st = m._stream()
        capi.check(m.lib.mudpt_cp_forward(m._h, capi.ptr(img), B, capi.FWD_TRAINING, st), "cp_forward")
        torch.cuda.synchronize()
        feat = m._cp_feat.clone()
        if table is not None:
            m._cp_feat.copy_(table)
        capi.check(m.lib.mudpt_cp_head(m._h, capi.ptr(lab), B, 1.0, capi.ptr(m._loss), None, 0, st), "cp_head")
        capi.check(m.lib.mudpt_cp_backward(m._h, capi.CP_VISION, st), "cp_backward")
        torch.cuda.synchronize()
        dfeat = m._cp_dfeat.clone()
        if dtable is not None:
            m._cp_dfeat.copy_(dtable)
        capi.check(m.lib.mudpt_cp_backward(m._h, capi.CP_TEXT, st), "cp_backward")
        torch.cuda.synchronize()
        g = {k: v.detach().clone() for k, v in m.grads().items()}
        m.close()
        return feat, dfeat, g

    feat_u, dfeat_u, g_u = run(None)
    res = [run(sh, feat_u, dfeat_u) for sh in shards]
    assert torch.equal(sum(r[0] for r in res), feat_u), "feature rows differ between the sharded and the unsharded text tower"
    for r in res:
        assert torch.equal(r[1], dfeat_u)
        for k in vis_keys:  # same images, same tables: the vision side does not know about the shard
            assert torch.equal(r[2][k], g_u[k]), k
    for k in keys:
        rms = g_u[k].pow(2).mean().sqrt().item()
        assert (sum(r[2][k] for r in res) - g_u[k]).abs().max().item() <= 2e-5 * rms, k


def test_sharded_handle_refuses_monolithic_entry_points():
    from mudpt_amd import capi
    from mudpt_amd.model import CustomCLIP
    c = GoldenCase("mudpt_tiny")
    nb = len(c.labels)
    m = CustomCLIP(shape_of(c.cfg), c.frozen, c.tokens, max_batch=nb, dtype="fp16", class_shard=(0, 2))
    m.set_params(c.params)
    img, lab = c.images.cuda(), c.labels.cuda()
    lg = torch.empty(nb, len(c.tokens), device="cuda")
    assert m.lib.mudpt_forward(m._h, capi.ptr(img), nb, capi.ptr(lg), m._stream()) != 0
    assert b"mudpt_cp_" in m.lib.mudpt_last_error()
    assert m.lib.mudpt_forward_backward(m._h, capi.ptr(img), capi.ptr(lab), nb, 1.0, capi.ptr(m._loss), None, m._stream()) != 0
    assert m.lib.mudpt_cp_backward(m._h, capi.CP_VISION, m._stream()) != 0  # no step in flight
    assert m.lib.mudpt_set_class_shard(m._h, 3, 3) != 0 and m.lib.mudpt_set_class_shard(m._h, 0, len(c.tokens) + 1) != 0
    m.close()


@pytest.mark.parametrize("dtype", ["fp16"])
def test_two_processes_gloo_208_classes_match_reference(tmp_path, dtype):
    c = GoldenCase("mudpt_vitb16_c208_b2")
    out = tmp_path / "cp.pt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "cp_worker.py"), "mudpt_vitb16_c208_b2", dtype, str(out)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = torch.load(out, weights_only=True)  # written by this test's own worker: tensors, floats, tuples
    assert z["shard"] == (0, 104)
    logits = torch.cat([rk["logits"] for rk in z["ranks"]])
    loss = sum(rk["loss"] for rk in z["ranks"]) / 2
    assert (logits - c.logits).abs().max().item() <= LOGIT_ATOL[dtype]
    assert abs(loss - c.loss) <= 2e-4 * max(1.0, abs(c.loss))
    _, _, og = O.forward_backward(c.cfg, c.frozen, c.params, c.class_embedding, c.eot, c.images, c.labels)
    check_grads(z["grads"], og, dtype, "2 ranks, 208 classes")
    for rk in z["ranks"]:  # eval mode: same logits as the training forward (dropout-free model), cache reuse changes nothing
        assert torch.equal(rk["eval"], rk["eval_reuse"])
        # a TRAINING forward of <= 8 images splits the contraction of out_proj / c_proj (a differently associated fp32 sum), an inference
        # forward never does (its logits must not depend on the test batch's size): the two agree to that rounding (tests/helpers.py)
        assert (rk["eval"] - rk["logits"]).abs().max().item() <= 6e-4


def test_plugin_runs_class_parallel_at_world_2(tmp_path):
    """The trainer PLUGIN under torch.distributed.run with two ranks (gloo, both on the box's one GPU), once with the class-parallel text
    tower (MUDPT_CLASS_PARALLEL=1) and once replicated: `build_model` shards the classes, `forward_backward` and the test pass go through
    the phases, the replicas hold bit-identical parameters after two steps, and the two modes agree on losses, accuracy and parameters up to
    the rounding noise of two differently ordered backward passes."""
    res = {}
    for mode in ("1", "0"):
        out = str(tmp_path / f"plugin{mode}")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MUDPT_CLASS_PARALLEL=mode)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
               os.path.join(ROOT, "tests", "plugin_cp_worker.py"), out]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        r0, r1 = torch.load(out + ".r0", weights_only=True), torch.load(out + ".r1", weights_only=True)
        assert torch.equal(r0["params"], r1["params"]), "replicas diverged"
        assert r0["acc"] == r1["acc"]
        res[mode] = (r0, r1)
    assert res["1"][0]["shard"] == (0, 6) and res["1"][1]["shard"] == (6, 11) and res["0"][0]["shard"] is None
    a, b = res["1"][0], res["0"][0]
    assert abs(a["losses"][0] - b["losses"][0]) <= 1e-5 * max(1.0, abs(b["losses"][0]))  # same forward: the first loss is the same number
    assert abs(a["losses"][1] - b["losses"][1]) <= 5e-3
    assert a["acc"] == b["acc"]
    moved = (b["params"] - a["params"]).abs().max().item()
    assert moved <= 2e-3, moved
