"""Probe: how fast does the vendor f16 GEMM run the P-form contraction (10000 x 10000 x 528)?  Upper bound for a
fused filter kernel in that form (the vendor kernel also writes the 200 MB result)."""
import torch
torch.manual_seed(0)
dev = "cuda:0"
for (m, n, k) in [(10000, 10000, 528), (10240, 10240, 528), (10240, 10240, 544), (10240, 10240, 512), (8192, 8192, 512)]:
    a = (torch.randn(m, k, device=dev) * 0.2).half()
    b = (torch.randn(n, k, device=dev) * 0.2).half()
    bt = b.t().contiguous()
    for name, fn in [("A@B^T", lambda: a @ b.t()), ("A@Bt", lambda: a @ bt)]:
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"{m}x{n}x{k} {name}: {ms*1e3:.1f} us  {2*m*n*k/ms/1e9:.0f} TF/s")
