"""Shared test helpers: rebuild a golden case (weights from the seeded recipe) for the oracle."""
import ast
import os

import numpy as np
import torch

from oracle import mudpt_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class GoldenCase:
    """name "mudpt_*": trainers/mudpt.py fixtures; "cocoop_*": trainers/cocoop.py fixtures (5 trainables, cocoop_oracle)."""

    def __init__(self, name: str):
        self.cocoop = name.startswith("cocoop")
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.z = z
        self.cfg = O.Config(**ast.literal_eval(str(z["config"])))
        fs, ts, is_ = (int(v) for v in z["seeds"])
        self.frozen = O.make_frozen_state(self.cfg, fs)
        if "logit_scale" in z.files:  # the *_s100 fixtures: exp(logit_scale) = 100, what pretrained CLIP checkpoints hold (clip/model.py:919)
            self.frozen["logit_scale"] = torch.tensor(float(z["logit_scale"]))
        self.tokens = torch.from_numpy(z["tokenized_prompts"]).long()
        self.eot = self.tokens.argmax(dim=-1)  # trainers/mudpt.py:154
        self.class_embedding = self.frozen["token_embedding.weight"][self.tokens]
        if self.cocoop:
            from oracle import cocoop_oracle as CO
            self.params = CO.make_trainable_state(self.cfg, ts, self.frozen, [int(v) for v in z["ctx_token_ids"]])
        else:
            self.params = O.make_trainable_state(self.cfg, ts, self.frozen, [int(v) for v in z["ctx_token_ids"]])
        self.labels = torch.from_numpy(z["labels"])
        g = torch.Generator().manual_seed(is_)
        B = len(self.labels)
        self.images = torch.randn(B, 3, self.cfg.image_size, self.cfg.image_size, generator=g)
        self.logits = torch.from_numpy(z["logits"])
        self.loss = float(z["loss"])

    def check_recipe(self):
        """The seeded recipe reproduced the tensors the fixture was generated from."""
        img = self.images.double()
        np.testing.assert_allclose([img.sum().item(), img.abs().sum().item()], self.z["images_checksum"], rtol=1e-12)
        f = self.frozen
        if "frozen_checksum" not in self.z.files:
            return
        np.testing.assert_allclose(
            [f["visual.transformer.resblocks.0.attn.in_proj_weight"].double().sum().item(),
             f["token_embedding.weight"].double().abs().sum().item()], self.z["frozen_checksum"], rtol=1e-12)

    def grad(self, key):
        k = "grad." + key
        return torch.from_numpy(self.z[k]) if k in self.z.files else None

    def grad_sample(self, key):
        k = "grad_sample." + key
        return torch.from_numpy(self.z[k]) if k in self.z.files else None


# A TRAINING forward of a tiny batch (<= 8 ViT-B images: <= 320 tiles of 64 x 64) splits the contraction of the vision tower's out_proj /
# c_proj over K slices (a differently associated fp32 sum flips the T rounding of ~1 % of their outputs); an INFERENCE forward never does, so
# that the logits of an image do not depend on the size of the test batch it arrives in.  At such batches the two forwards
# agree to that rounding only (measured 2.8e-4 fp16 / 5e-3 bf16 at logit scale 14.29); above them, and in the parity mode, bit for bit.
FWD_SPLIT_TOL = {"fp16": 6e-4, "bf16": 1.2e-2, "fp32": 0.0}  # (CoCoOp, where the image features enter twice: 9.3e-3 bf16)


def assert_training_forward_is_the_inference_forward(train_logits, eval_logits, dtype):
    a, b = train_logits.detach().float().cpu(), eval_logits.detach().float().cpu()
    if torch.equal(a, b):
        return
    d = (a - b).abs().max().item()
    assert d <= FWD_SPLIT_TOL[dtype], f"training vs inference logits differ by {d:.3e} ({dtype})"
