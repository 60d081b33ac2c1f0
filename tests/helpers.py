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
