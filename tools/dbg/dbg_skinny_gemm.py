"""Diagnostic (GPU box): library GEMM time for the tall, thin Linear layers of the VSS blocks (fp16, autocast shapes)."""
import torch


def t(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for T, K, N in [(524288, 16, 64), (524288, 32, 16), (131072, 32, 128), (131072, 64, 32), (32768, 64, 256), (32768, 128, 64),
                (524288, 64, 128), (524288, 32, 64)]:
    x = torch.randn(T, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16)
    dy = torch.randn(T, N, device="cuda", dtype=torch.float16)
    tf = t(lambda: torch.nn.functional.linear(x, w))
    tb = t(lambda: dy @ w)
    byt = 2 * T * (K + N)
    print(f"T={T} K={K} N={N}: fwd {tf*1e3:7.1f} us ({byt/tf/1e6:6.0f} GB/s)  dX {tb*1e3:7.1f} us ({byt/tb/1e6:6.0f} GB/s)",
          flush=True)
