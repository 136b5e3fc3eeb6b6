"""What a copy of SEVEN arrays at once reaches (the scatter pass of the headline step moves 6 doubles + the ids of every slot,
read once and written once: 14 streams): torch's multi-tensor copy (one kernel walking all seven) against seven separate copies
and one copy of the same total size.    python tools/dbg/multi_copy.py [n]"""
import sys, time
import torch
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
src = [torch.rand(n, device=dev, dtype=torch.float64) for _ in range(6)] + [torch.zeros(n // 2, device=dev, dtype=torch.float64)]
dst = [torch.empty_like(s) for s in src]
big_s = torch.rand(int(6.5 * n), device=dev, dtype=torch.float64)
big_d = torch.empty_like(big_s)
nbytes = 2 * sum(s.numel() * 8 for s in src)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
t1 = timed(lambda: torch._foreach_copy_(dst, src))
t2 = timed(lambda: [d.copy_(s) for d, s in zip(dst, src)])
t3 = timed(lambda: big_d.copy_(big_s))
for name, t in (("one multi-tensor kernel over 7 arrays", t1), ("7 separate copies", t2), ("one array of the same total size", t3)):
    print(f"{name:40s} {t * 1e3:7.3f} ms  {nbytes / t / 1e12:5.2f} TB/s")
