"""ORACLE fixture (not a reference fixture): the CPU oracle's logits, loss and gradients for ViT-B/16 with 1000 synthetic class prompts, B = 2.

    python tests/golden/gen_oracle_c1000.py      # writes tests/golden/oracle_vitb16_c1000_b2.npz (~65 s of CPU)

The reference's own fixtures stop at C = 208 (mudpt_vitb16_c208_b2, which pins the oracle on CPU); beyond that size parity against the
reference itself is unpinned and the oracle is the checker.  The oracle is deterministic, so its outputs at this size are stored instead
of being recomputed inside the GPU suite (65 s of the driver's GPU-test budget).  tests/test_oracle_golden.py re-derives a slice of
this file on the CPU so that a change to the oracle cannot leave a stale fixture behind.  The three big Linear weight gradients are
stored as [::4, ::4] samples; everything else in full."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mudpt_oracle as O  # noqa: E402
from mudpt_amd import synth  # noqa: E402

N_CLS, TRAIN_SEED, IMAGE_SEED, LABELS = 1000, 9, 97, [3, 977]
SAMPLE = 4  # stride of the stored samples of the big 2-D gradients


def inputs():
    cfg = O.VIT_B16
    frozen = O.make_frozen_state(cfg, 0)
    tok = synth.synthetic_tokenized_prompts(N_CLS).long()
    params = O.make_trainable_state(cfg, TRAIN_SEED, frozen, synth.CTX_INIT_TOKENS)
    g = torch.Generator().manual_seed(IMAGE_SEED)
    images, labels = torch.randn(2, 3, 224, 224, generator=g), torch.tensor(LABELS)
    return cfg, frozen, tok, params, images, labels


def main():
    cfg, frozen, tok, params, images, labels = inputs()
    loss, logits, grads = O.forward_backward(cfg, frozen, params, frozen["token_embedding.weight"][tok], tok.argmax(-1), images, labels)
    out = {"logits": logits.numpy(), "loss": np.array(loss.item(), dtype=np.float64), "tokens_checksum": np.array(int(tok.sum())),
           "meta": np.array(repr(dict(n_cls=N_CLS, train_seed=TRAIN_SEED, image_seed=IMAGE_SEED, labels=LABELS, sample=SAMPLE)))}
    for k in O.TRAINABLE_ORDER:
        g = grads[k]
        out["grad_stats." + k] = np.array([g.double().pow(2).mean().sqrt().item(), g.double().abs().max().item()])
        if g.dim() == 2 and g.numel() > 100000:
            out["grad_sample." + k] = g[::SAMPLE, ::SAMPLE].numpy()
        else:
            out["grad." + k] = g.numpy()
    path = os.path.join(ROOT, "tests", "golden", "oracle_vitb16_c1000_b2.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: loss {loss.item():.6f}, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
