"""Is the B 4 / C 50 step bound by the host's launch rate?  Enqueue time of a step (the C call returns after queueing its ~460 launches)
against the synchronised step time (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mudpt_amd import synth
from mudpt_amd.model import CustomCLIP, ModelShape

B, C = 4, 50
shape = ModelShape()
m = CustomCLIP(shape, synth.random_clip_state(shape, seed=0), synth.synthetic_tokenized_prompts(C), ctx_token_ids=synth.CTX_INIT_TOKENS, max_batch=B, dtype="bf16",
               device="cuda:0", seed=1)
g = torch.Generator().manual_seed(1)
images = torch.randn(B, 3, 224, 224, generator=g).cuda(); labels = torch.randint(0, C, (B,), generator=g).cuda()
def step():
    m.forward_backward(images, labels); m.sgd_step(0.0025, momentum=0.9, weight_decay=5e-4)
for _ in range(10): step()
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, until the GPU drained {1e3 * (t2 - t0) / N:.3f} ms/step", flush=True)
# each step synchronised: the GPU-side latency chain alone
t0 = time.perf_counter()
for _ in range(N):
    step(); torch.cuda.synchronize()
print(f"synchronised every step: {1e3 * (time.perf_counter() - t0) / N:.3f} ms/step", flush=True)
