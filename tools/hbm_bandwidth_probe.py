import torch, time
x = torch.empty(4_000_000_000, dtype=torch.float32, device='cuda')
y = torch.empty(4_000_000_000, dtype=torch.float32, device='cuda')
for name, fn, nbytes in (("fill (write)", lambda: x.fill_(1.0), 16e9), ("copy (read+write)", lambda: y.copy_(x), 32e9), ("sum (read)", lambda: x.sum(), 16e9)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print("%s: %.3f ms  %.2f TB/s" % (name, ms, nbytes / ms / 1e9))
