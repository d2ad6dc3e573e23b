import torch
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (315, 630, 1260):
    n = mb * 1000000 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); ts = t(lambda: a.sum())
    print("%4d MB: fill %6.1f us %.2f TB/s | copy %6.1f us %.2f TB/s (r+w) | sum %6.1f us %.2f TB/s" % (mb, tf, mb / tf, tc, 2 * mb / tc, ts, mb / ts))
