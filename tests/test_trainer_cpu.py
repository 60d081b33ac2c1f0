"""Host logic of the drop-in trainer plugin that needs no GPU: registration, cfg checks, error behaviour, schedules."""
import math

import pytest
import torch

from mudpt_amd import dassl_lite, trainer


def test_plugin_is_registered_under_the_reference_name():
    assert "MuDPT" in trainer.TRAINER_REGISTRY.registered_names()  # train.py --trainer MuDPT
    for hook in ("check_cfg", "build_model", "forward_backward", "parse_batch_train", "load_model", "model_inference"):
        assert callable(getattr(trainer.MuDPT, hook))


def test_cocoop_plugin_is_registered_under_the_reference_name():
    from mudpt_amd import cocoop
    assert "CoCoOp" in trainer.TRAINER_REGISTRY.registered_names()  # train.py --trainer CoCoOp
    for hook in ("check_cfg", "build_model", "forward_backward", "parse_batch_train", "load_model", "model_inference"):
        assert callable(getattr(cocoop.CoCoOp, hook))
    cfg = dassl_lite.default_cfg()
    t = object.__new__(cocoop.CoCoOp)
    t.check_cfg(cfg)
    cfg.TRAINER.COCOOP.PREC = "int8"
    with pytest.raises(AssertionError):  # trainers/cocoop.py:204
        t.check_cfg(cfg)
    t._models = {"prompt_learner": torch.nn.Linear(2, 2)}
    assert t.load_model("") is None
    with pytest.raises(FileNotFoundError):  # trainers/cocoop.py:296-297
        t.load_model("/nonexistent", epoch=3)


def test_check_cfg_matches_reference_assert():
    cfg = dassl_lite.default_cfg()
    t = object.__new__(trainer.MuDPT)
    t.check_cfg(cfg)
    cfg.TRAINER.MUDPT.PREC = "int8"
    with pytest.raises(AssertionError):  # trainers/mudpt.py:190
        t.check_cfg(cfg)


def test_load_model_error_behaviour(tmp_path):
    t = object.__new__(trainer.MuDPT)
    t._models = {"MultimodalDeepPromptTuning": torch.nn.Linear(2, 2)}
    assert t.load_model("") is None  # "skipped as no pretrained model is given"
    with pytest.raises(FileNotFoundError):  # trainers/mudpt.py:286-287
        t.load_model(str(tmp_path), epoch=3)


def test_missing_backbone_path_is_an_explicit_error():
    cfg = dassl_lite.default_cfg()
    cfg.MODEL.BACKBONE.SYNTHETIC_SEED = None
    with pytest.raises(RuntimeError, match="MODEL.BACKBONE.PATH"):
        trainer.load_clip_state_dict(cfg)


def test_benchmark_prompts_tokenize_without_the_vocabulary():
    tok = trainer.tokenize_prompts(["a photo of a face.", "a photo of a binocular."])
    assert tok.shape == (2, 77) and tok[0, :8].tolist() == [49406, 320, 1125, 539, 320, 1710, 269, 49407]
    assert tok.argmax(-1).tolist() == [7, 8]
    with pytest.raises(RuntimeError):
        trainer.tokenize_prompts(["a photo of a zebra."])


def test_constant_warmup_cosine_schedule():
    p = torch.nn.Parameter(torch.zeros(1))
    cfg = dassl_lite.default_cfg().OPTIM
    opt = dassl_lite.build_optimizer(torch.nn.ParameterList([p]), cfg)
    sch = dassl_lite.build_lr_scheduler(opt, cfg)
    lrs = []
    for _ in range(cfg.MAX_EPOCH):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert lrs[0] == pytest.approx(1e-5)  # WARMUP_CONS_LR for WARMUP_EPOCH = 1
    assert lrs[1] == pytest.approx(0.5 * 0.0025 * (1 + math.cos(math.pi * 1 / 10)))
    assert lrs[-1] < lrs[1]


def test_load_pretrained_weights_matches_names_and_shapes(tmp_path):
    """MODEL.INIT_WEIGHTS (trainers/mudpt.py:220-221, trainers/cocoop.py:234-235): Dassl's name + shape matching."""
    src, dst = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    path = tmp_path / "w.pth.tar"
    torch.save({"state_dict": {"weight": src.weight.detach(), "bias": torch.zeros(5), "extra": torch.ones(1)}, "epoch": 3}, path)
    before_bias = dst.bias.detach().clone()
    dassl_lite.load_pretrained_weights(dst, str(path))
    assert torch.equal(dst.weight, src.weight) and torch.equal(dst.bias, before_bias)  # wrong-shaped bias is discarded


def test_device_prefetcher_is_transparent_without_a_gpu():
    """On a box without a HIP device the prefetcher passes the loader through: same batches, same order, same length, attributes
    of the wrapped loader reachable (Dassl reads train_loader_x.dataset / len())."""
    from mudpt_amd.prefetch import DevicePrefetcher

    class Loader(list):
        dataset = "the dataset"
    batches = Loader({"img": torch.full((2, 3, 4, 4), float(i)), "label": torch.tensor([i, i + 1]), "impath": [str(i)]} for i in range(5))
    pf = DevicePrefetcher(batches, device=None if not torch.cuda.is_available() else "cuda:0")
    assert len(pf) == 5 and pf.dataset == "the dataset"
    for epoch in range(2):  # re-iterable, like a DataLoader
        got = list(pf)
        assert len(got) == 5
        for i, b in enumerate(got):
            assert b["impath"] == [str(i)] and torch.equal(b["img"].cpu(), batches[i]["img"]) and torch.equal(b["label"].cpu(), batches[i]["label"])
    assert list(DevicePrefetcher(Loader())) == []


def test_prec_fp16_on_a_real_checkpoint_scale_is_called_out(capsys):
    """PREC = "fp16" (the reference yamls' default) with exp(logit_scale) = 100, what every released CLIP checkpoint holds: the fp16 mode is
    ~4e-3 from the reference's logits there, so the plugins say so once and name the setting that holds 1e-3;
    random-init weights (scale 14.29), the other PREC values and a missing logit_scale stay silent."""
    import math
    import torch
    from mudpt_amd import trainer
    trainer._warned_fp16_scale = False
    real, init = {"logit_scale": torch.tensor(math.log(100.0))}, {"logit_scale": torch.tensor(math.log(1 / 0.07))}
    assert not trainer.warn_if_fp16_misses_the_bound("fp16", init)
    assert not trainer.warn_if_fp16_misses_the_bound("fp32", real)
    assert not trainer.warn_if_fp16_misses_the_bound("amp", real)
    assert not trainer.warn_if_fp16_misses_the_bound("fp16", None)
    assert not trainer.warn_if_fp16_misses_the_bound("fp16", {})
    assert capsys.readouterr().out == ""
    assert trainer.warn_if_fp16_misses_the_bound("fp16", real)
    out = capsys.readouterr().out
    assert 'PREC "fp32"' in out and "4e-3" in out and "100.0" in out
    assert trainer.warn_if_fp16_misses_the_bound("fp16", real)  # still reported to the caller ...
    assert capsys.readouterr().out == ""                          # ... but printed once per process
    assert trainer.precision_to_dtype("fp32") == "fp32" and trainer.precision_to_dtype("amp") == "bf16"
