import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mudpt_amd import capi
lib = capi.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
for B, Cn, e in ((256, 11, 512), (256, 1000, 512), (128, 1000, 768)):
    img, txt = torch.randn(B, e, device="cuda"), torch.randn(Cn, e, device="cuda")
    labels = torch.randint(0, Cn, (B,), device="cuda")
    logits, loss = torch.empty(B, Cn, device="cuda"), torch.empty(1, device="cuda")
    dimg, dtxt = torch.empty(B, e, device="cuda"), torch.empty(Cn, e, device="cuda")
    def f():
        assert lib.mudpt_head(P(img), P(txt), P(labels), 14.3, 1.0, B, Cn, e, P(logits), P(loss), P(dimg), P(dtxt), None) == 0
    for _ in range(3): f()
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize()
    print(f"head fwd+bwd B={B} C={Cn} e={e}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call (incl. scratch malloc/free + sync of the test hook)")
