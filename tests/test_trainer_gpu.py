"""The drop-in trainer surface on the GPU: build_model -> forward_backward -> model_inference -> save / load_model."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_trainer_steps_and_checkpoint_roundtrip(tmp_path):
    from mudpt_amd import dassl_lite, trainer
    cfg = dassl_lite.default_cfg()
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.OPTIM.MAX_EPOCH = 2
    cfg.OPTIM.WARMUP_EPOCH = 0
    cfg.OPTIM.LR = 0.02
    cfg.DATASET.NUM_TRAIN, cfg.DATASET.NUM_TEST = 8, 8
    cfg.DATALOADER.TEST.BATCH_SIZE = 8
    cfg.TRAINER.MUDPT.N_CTX, cfg.TRAINER.MUDPT.DEEP_PROMPT_DEPTH = 4, 12
    t = dassl_lite.build_trainer(cfg)
    assert type(t).__name__ == "MuDPT" and t.get_model_names() == ["MultimodalDeepPromptTuning"]
    assert sorted(t.model.state_dict()) == sorted(t.model.param_names) and len(t.model.param_names) == 10
    assert sum(p.numel() for p in t.model.parameters()) == 1243136
    batch = t.train_loader_x[0]
    t.batch_idx, t.num_batches = 0, 99
    losses = [t.forward_backward(batch)["loss"] for _ in range(6)]  # same batch: SGD must make progress on it
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    logits = t.model_inference(batch["img"].cuda())
    assert logits.shape == (4, 11) and torch.isfinite(logits).all()
    t.save_model(1, str(tmp_path))
    before = {k: v.clone() for k, v in t.model.state_dict().items()}
    with torch.no_grad():
        for p in t.model.parameters():
            p.add_(1.0)
    t.load_model(str(tmp_path), epoch=2)
    for k, v in t.model.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert torch.allclose(t.model_inference(batch["img"].cuda()), logits, atol=1e-5)
    acc = t.test()
    assert 0.0 <= acc <= 100.0


def test_cocoop_trainer_steps_and_checkpoint_roundtrip(tmp_path):
    """trainers/cocoop.py's surface: only prompt_learner is registered / optimised; checkpoints hold ctx + meta_net.*."""
    from mudpt_amd import cocoop, dassl_lite  # noqa: F401  (registers CoCoOp)
    cfg = dassl_lite.default_cfg()
    cfg.TRAINER.NAME = "CoCoOp"
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.OPTIM.MAX_EPOCH, cfg.OPTIM.WARMUP_EPOCH, cfg.OPTIM.LR = 2, 0, 0.02
    cfg.DATASET.NUM_TRAIN, cfg.DATASET.NUM_TEST = 8, 8
    cfg.DATALOADER.TRAIN_X.BATCH_SIZE, cfg.DATALOADER.TEST.BATCH_SIZE = 2, 4
    t = dassl_lite.build_trainer(cfg)
    assert type(t).__name__ == "CoCoOp" and t.get_model_names() == ["prompt_learner"]
    assert sorted(t.model.prompt_learner.state_dict()) == ["ctx", "meta_net.linear1.bias", "meta_net.linear1.weight",
                                                           "meta_net.linear2.bias", "meta_net.linear2.weight"]
    assert sum(p.numel() for p in t.model.parameters()) == 4 * 512 + 32 * 512 + 32 + 512 * 32 + 512
    batch = t.train_loader_x[0]
    t.batch_idx, t.num_batches = 0, 99
    losses = [t.forward_backward(batch)["loss"] for _ in range(6)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    logits = t.model_inference(batch["img"].cuda())
    assert logits.shape == (2, 11) and torch.isfinite(logits).all()
    t.save_model(1, str(tmp_path))
    before = {k: v.clone() for k, v in t.model.state_dict().items()}
    with torch.no_grad():
        for p in t.model.parameters():
            p.add_(1.0)
    t.load_model(str(tmp_path), epoch=2)
    for k, v in t.model.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert torch.allclose(t.model_inference(batch["img"].cuda()), logits, atol=1e-5)


def test_backbone_from_checkpoint_files_equals_the_in_memory_path(tmp_path):
    """MODEL.BACKBONE.PATH (trainers/mudpt.py:20-38 -> clip/clip.py:95-144): a plain state-dict file and a torch.jit archive, both
    holding convert_weights' fp16 tensors (clip/model.py:857-878), give the same logits BIT FOR BIT as the same weights handed over
    in memory -- shape inference (clip/model.py:885-904), fp16 -> fp32 widening and key routing are the only things in between."""
    from mudpt_amd import dassl_lite, synth, trainer
    from mudpt_amd.model import ModelShape
    from tests.test_checkpoint_cpu import _Tree, as_checkpoint
    shape = ModelShape(n_ctx=4, depth=12)
    ck = as_checkpoint(synth.random_clip_state(shape, seed=0))
    ck["input_resolution"] = torch.tensor(224)
    plain, jit = tmp_path / "ViT-B-16.state.pt", tmp_path / "ViT-B-16.jit.pt"
    torch.save(ck, plain)
    torch.jit.save(torch.jit.script(_Tree(ck)), str(jit))
    del ck
    logits = {}
    for tag, path in (("memory", ""), ("plain", str(plain)), ("jit", str(jit))):
        cfg = dassl_lite.default_cfg()
        cfg.OUTPUT_DIR = str(tmp_path)
        cfg.DATASET.NUM_TRAIN, cfg.DATASET.NUM_TEST = 4, 4
        cfg.DATALOADER.TEST.BATCH_SIZE = 4
        cfg.MODEL.BACKBONE.PATH = path
        cfg.TRAINER.MUDPT.N_CTX, cfg.TRAINER.MUDPT.DEEP_PROMPT_DEPTH = 4, 12
        t = dassl_lite.build_trainer(cfg)
        assert t.model.shape == shape
        t.set_model_mode("eval")
        logits[tag] = t.model_inference(t.train_loader_x[0]["img"].cuda()).clone()
        t.model.close()
    assert torch.isfinite(logits["memory"]).all()
    assert torch.equal(logits["plain"], logits["memory"]) and torch.equal(logits["jit"], logits["memory"])


def test_c_abi_allreduce_over_a_one_rank_rccl_communicator():
    """mudpt_allreduce_grads (include/mudpt.h; the exchange a non-Python host would call): with a world-size-1 RCCL communicator the
    sum over ranks is the identity, so the bound gradient bucket must come back bit for bit -- this exercises the lazy RCCL
    resolution, the count / dtype / op arguments and the stream hand-over.  N > 1 is covered by bench.py under torch.distributed.run."""
    import ctypes as C
    import os
    from oracle import mudpt_oracle as O
    from tests.helpers import GoldenCase
    from tests.test_model_gpu import build
    rccl = None
    for cand in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = C.CDLL(cand)
            break
        except OSError:
            continue
    assert rccl is not None, "RCCL not found"

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    case = GoldenCase("mudpt_tiny")
    m = build(case, "fp16")
    m.forward_backward(case.images, case.labels)
    torch.cuda.synchronize()
    before = m.flat_grads.clone()
    assert before.abs().sum().item() > 0
    from mudpt_amd import capi
    capi.check(m.lib.mudpt_allreduce_grads(m._h, comm, m._stream()), "allreduce_grads")
    torch.cuda.synchronize()
    assert torch.equal(m.flat_grads, before)
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    m.close()


def test_prefetched_epoch_equals_the_plain_loop(tmp_path):
    """build_model wraps train_loader_x in the DevicePrefetcher (next batch's host -> device copy on a side stream during the current
    step).  An epoch through it gives the same losses, bit for bit, as feeding the raw host batches one by one to a second trainer."""
    from mudpt_amd import dassl_lite, trainer  # noqa: F401
    from mudpt_amd.prefetch import DevicePrefetcher

    def make():
        cfg = dassl_lite.default_cfg()
        cfg.OUTPUT_DIR = str(tmp_path)
        cfg.OPTIM.MAX_EPOCH, cfg.OPTIM.WARMUP_EPOCH, cfg.OPTIM.LR = 1, 0, 0.01
        cfg.DATASET.NUM_TRAIN, cfg.DATASET.NUM_TEST = 20, 4
        cfg.DATALOADER.TRAIN_X.BATCH_SIZE, cfg.DATALOADER.TEST.BATCH_SIZE = 4, 4
        cfg.TRAINER.MUDPT.N_CTX, cfg.TRAINER.MUDPT.DEEP_PROMPT_DEPTH = 4, 12
        return dassl_lite.build_trainer(cfg)
    a, b = make(), make()
    assert isinstance(a.train_loader_x, DevicePrefetcher) and len(a.train_loader_x) == 5
    a.set_model_mode("train")
    a.num_batches = len(a.train_loader_x)
    la = []
    for a.batch_idx, batch in enumerate(a.train_loader_x):
        assert batch["img"].is_cuda and batch["_mudpt_sharded"]
        la.append(a.forward_backward(batch)["loss"])
    b.set_model_mode("train")
    b.num_batches = 5
    lb = []
    for b.batch_idx in range(5):
        lb.append(b.forward_backward(b.train_loader_x[b.batch_idx])["loss"])  # raw host batch: the synchronous .to(device) path
    assert la == lb and all(x == x for x in la)
    assert torch.equal(a.model.flat_params, b.model.flat_params)
    a.model.close()
    b.model.close()


def test_prefetcher_keeps_staged_batches_alive_for_the_consumer_stream():
    """A loader that yields fp16 images (CustomCLIP then converts on the consumer's stream) and a caller that drops every batch WITHOUT a
    host sync: the staged tensors were allocated on the prefetcher's side stream, so without record_stream the caching allocator could
    give a dropped batch's block to the next batch's copy while the conversion / patch gather still reads it.  Logits through the
    prefetcher must equal the unprefetched path for every batch."""
    from mudpt_amd.prefetch import DevicePrefetcher
    from tests.helpers import GoldenCase
    from tests.test_model_gpu import build
    case = GoldenCase("mudpt_tiny")
    g = torch.Generator().manual_seed(5)
    S = case.cfg.image_size
    batches = [{"img": torch.randn(3, 3, S, S, generator=g).half(), "label": torch.randint(0, 11, (3,), generator=g)} for _ in range(12)]
    m = build(case, "fp16")
    m.eval()
    want = [m(b["img"].float().cuda()).clone() for b in batches]
    torch.cuda.synchronize()
    got = []
    for batch in DevicePrefetcher(batches, device="cuda:0"):
        assert batch["img"].is_cuda and batch["img"].dtype == torch.float16
        got.append(m(batch["img"]))  # no sync, no reference kept to the staged batch beyond this iteration
        del batch
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    m.close()


def test_gradient_overflow_skips_the_step_and_halves_the_loss_scale(tmp_path):
    """The backward runs on per-sample gradients times a power-of-two loss scale inside the library (include/mudpt.h mudpt_set_loss_scale).
    With an absurd scale the fp16 copies of the token gradients overflow: the gradients come back non-finite, the plugin skips the
    optimizer step and halves the scale (torch.cuda.amp.GradScaler's rule, the reference's amp path trainers/mudpt.py:228,243-246) until
    a step goes through -- and the parameters never move on garbage."""
    from mudpt_amd import dassl_lite, trainer  # noqa: F401
    cfg = dassl_lite.default_cfg()
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.OPTIM.MAX_EPOCH, cfg.OPTIM.WARMUP_EPOCH, cfg.OPTIM.LR = 1, 0, 0.01
    cfg.DATASET.NUM_TRAIN, cfg.DATASET.NUM_TEST = 8, 4
    cfg.DATALOADER.TRAIN_X.BATCH_SIZE, cfg.DATALOADER.TEST.BATCH_SIZE = 4, 4
    cfg.TRAINER.MUDPT.N_CTX, cfg.TRAINER.MUDPT.DEEP_PROMPT_DEPTH, cfg.TRAINER.MUDPT.PREC = 4, 12, "fp16"
    t = dassl_lite.build_trainer(cfg)
    batch = t.train_loader_x[0]
    t.batch_idx, t.num_batches = 0, 99
    ref_loss = t.forward_backward(batch)["loss"]              # a clean step at the default scale
    t.model.set_loss_scale(2.0 ** 40)
    t._loss_scale_state = {"scale": 2.0 ** 40, "clean": 0, "skipped": 0}
    before = t.model.flat_params.clone()
    skipped = 0
    for _ in range(60):
        out = t.forward_backward(batch)
        assert out["loss"] == out["loss"]                      # the LOSS stays finite: only the scaled backward overflowed
        if not torch.equal(t.model.flat_params, before):
            break
        skipped += 1
    assert 1 <= skipped < 60 and t._loss_scale_state["skipped"] == skipped
    assert t.model.loss_scale == 2.0 ** (40 - skipped) and torch.isfinite(t.model.flat_params).all() and torch.isfinite(t.model.flat_grads).all()
    assert abs(out["loss"] - ref_loss) < 0.5                    # and training continues from where it was
    t.model.close()
