"""The native BPE tokenizer (mudpt_amd/tokenizer.py) against token ids recorded from the reference's own tokenizer
(tests/golden/tokenizer_cases.json, written by tests/golden/gen_golden.py --tokenizer-only).

The merge table is data that ships with CLIP, not with this repository: the test looks for it through the tokenizer's own
search (MUDPT_BPE_VOCAB, an importable clip package) and at the reference checkout of the build container, and is skipped
where no copy exists (the GPU box)."""
import json
import os

import pytest
import torch

from mudpt_amd import tokenizer

HERE = os.path.dirname(os.path.abspath(__file__))


def _vocab():
    for cand in (os.environ.get("MUDPT_BPE_VOCAB"), "/root/reference/clip/" + tokenizer.VOCAB_FILE):
        if cand and os.path.isfile(cand):
            return cand
    try:
        return tokenizer.find_vocab()
    except RuntimeError:
        pytest.skip("CLIP's bpe_simple_vocab_16e6.txt.gz is not available on this machine")


def test_ids_match_the_reference_tokenizer():
    tok = tokenizer.BPETokenizer(_vocab())
    spec = json.load(open(os.path.join(HERE, "golden", "tokenizer_cases.json"), encoding="utf-8"))
    assert len(spec["cases"]) >= 25
    for c in spec["cases"]:
        row = tok([c["text"]], spec["context_length"], truncate=spec["truncate"])[0]
        n = len(c["ids"])
        assert row[:n].tolist() == c["ids"], c["text"]
        assert (row[n:] == 0).all()
        assert int(row[n - 1]) == 49407
    assert tok.sot_id == 49406 and tok.eot_id == 49407


def test_too_long_prompt_raises_like_clip_tokenize():
    tok = tokenizer.BPETokenizer(_vocab())
    with pytest.raises(RuntimeError, match="too long for context length"):  # clip/clip.py:235
        tok(["word " * 100])
    out = tok(["word " * 100], truncate=True)
    assert out.shape == (1, 77) and out[0, -1] == 49407


def test_missing_vocab_is_an_explicit_error(tmp_path, monkeypatch):
    import importlib.util
    if importlib.util.find_spec("clip") is not None:
        pytest.skip("a clip package with its merge table is importable here")
    monkeypatch.delenv("MUDPT_BPE_VOCAB", raising=False)
    with pytest.raises(RuntimeError, match="merge table"):
        tokenizer.find_vocab(str(tmp_path / "nope.gz"), near=str(tmp_path / "ViT-B-16.pt"))


def test_benchmark_prompts_agree_with_the_recorded_ids():
    from mudpt_amd import synth
    tok = tokenizer.BPETokenizer(_vocab())
    prompts = [f"a photo of a {n}." for n in synth.BENCH_CLASSNAMES]
    assert torch.equal(tok(prompts), synth.bench_tokenized_prompts())
