import torch, time
x = torch.empty(20 * 1920 * 1080 * 4, dtype=torch.float32, device="cuda")
for name, fn in (("fill_", lambda: x.fill_(1.5)), ("zero_", lambda: x.zero_()), ("copy", None)):
    if fn is None:
        y = torch.empty_like(x)
        fn = lambda: y.copy_(x)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20): fn()
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    print(name, "%.1f us  %.2f TB/s (bytes written)" % (ms * 1e3, x.numel() * 4 / ms / 1e9))
