"""What does a plain streaming kernel reach on this board?  (Context for the roofline fractions: read + write mixes.)"""
import torch
dev = "cuda:0"
n = 1 << 29                      # 1 GiB of bf16 per tensor
a = torch.randn(n, device=dev, dtype=torch.bfloat16); b = torch.randn(n, device=dev, dtype=torch.bfloat16); c = torch.empty_like(a)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
by = n * 2
for name, fn, traffic in (("copy (1R 1W)", lambda: c.copy_(a), 2 * by), ("add out-of-place (2R 1W)", lambda: torch.add(a, b, out=c), 3 * by),
                          ("add in-place (2R 1W)", lambda: a.add_(b), 3 * by), ("read-only sum (1R)", lambda: a.sum(), by),
                          ("fill (1W)", lambda: c.zero_(), by)):
    s = t(fn)
    print("%-28s %7.1f us  %5.2f TB/s" % (name, s * 1e6, traffic / s / 1e12))
