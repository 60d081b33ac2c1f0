import sys, os
sys.path.insert(0, os.getcwd())
import torch
from oracle import mudpt_oracle as O
from tests.helpers import GoldenCase
from tests.test_model_gpu import build
case = GoldenCase("mudpt_vitb16_b4")
taps = {}
with torch.no_grad():
    ref = O.forward(case.cfg, case.frozen, case.params, case.class_embedding, case.eot, case.images, taps)
m = build(case, "fp16")
lg = m(case.images).cpu()
fi = m.debug_read("image_features", 4).view(4, -1).double(); ft = m.debug_read("text_features", 4).view(11, -1).double()
ri, rt = taps["image_features"].double(), taps["text_features"].double()
def logits(i, t): return 14.285714285714286 * (i / i.norm(dim=-1, keepdim=True)) @ (t / t.norm(dim=-1, keepdim=True)).t()
print("scale", case.frozen["logit_scale"].exp().item())
print("GPU logits vs ref            :", (lg - ref).abs().max().item(), (lg - ref).pow(2).mean().sqrt().item())
print("fp64 head on GPU features vs GPU logits:", (logits(fi, ft).float() - lg).abs().max().item())
print("fp64 head(GPU img, GPU txt) vs ref:", (logits(fi, ft).float() - ref).abs().max().item())
print("fp64 head(GPU img, ref txt) vs ref:", (logits(fi, rt).float() - ref).abs().max().item())
print("fp64 head(ref img, GPU txt) vs ref:", (logits(ri, ft).float() - ref).abs().max().item())
# decompose the text feature error into radial / tangential parts
d = ft - rt
rad = (d * rt).sum(-1, keepdim=True) / (rt * rt).sum(-1, keepdim=True) * rt
print("text error: radial rel", (rad.norm(dim=-1) / rt.norm(dim=-1)).tolist())
print("text error: tangential rel", ((d - rad).norm(dim=-1) / rt.norm(dim=-1)).tolist())
# common-mode: is the error the same across classes?
print("mean error vector norm / mean per-class error norm:", (d.mean(0).norm() / d.norm(dim=-1).mean()).item())
m.close()
