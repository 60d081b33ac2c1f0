"""Experiment: do two half-batch instances on two streams beat one full-batch instance? (micro-batch overlap potential)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from mudpt_amd.model import CustomCLIP, ModelShape
from mudpt_amd import synth

shape = ModelShape()
state = synth.random_clip_state(shape, seed=0)
tok = synth.bench_tokenized_prompts()


def run(nmodels, B, steps=8, warm=3, knobs=None):
    models = [CustomCLIP(shape, state, tok, ctx_token_ids=synth.CTX_INIT_TOKENS, max_batch=B, dtype="bf16", seed=1, knobs=knobs) for _ in range(nmodels)]
    streams = [torch.cuda.Stream() for _ in range(nmodels)]
    g = torch.Generator().manual_seed(0)
    imgs = [torch.randn(B, 3, 224, 224, generator=g).cuda() for _ in range(nmodels)]
    labs = [torch.randint(0, 11, (B,), generator=g).cuda() for _ in range(nmodels)]
    torch.cuda.synchronize()

    def step():
        for m, s, x, y in zip(models, streams, imgs, labs):
            with torch.cuda.stream(s):
                m.forward_backward(x, y)
                m.sgd_step(0.0025)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{nmodels} x batch {B} {knobs or ''}: {dt * 1e3:.2f} ms per step -> {nmodels * B / dt:.0f} img/s", flush=True)
    for m in models:
        m.close()


run(1, 256)
run(1, 128)
run(2, 128)
run(4, 64)
# Round 2, one box: 1 x 256: 26.5-26.9 ms; 2 x 128 on two streams: 25.9-28.4 ms (+4 % .. -7 %, run to run); with every persistent GEMM
# confined to half of the CUs: 30.2-31.8 ms.  Half-batch pipelining inside one handle is not worth its second set of scratch buffers.
