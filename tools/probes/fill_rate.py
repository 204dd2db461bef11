"""Pure-write and copy rates of the card (torch fill_ / copy_ on 512 MiB .. 1 GiB tensors): what an HBM-bound store stream can reach."""
import torch
torch.cuda.init()
for mb in (512, 1024):
    n = mb * 2**20 // 4
    a = torch.empty(n, dtype=torch.float32, device='cuda'); b = torch.empty_like(a)
    for name, fn, nbytes in (('fill', lambda: a.fill_(1.0), 4 * n), ('copy', lambda: b.copy_(a), 8 * n)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%s %4d MiB: %.1f us, %.2f TB/s' % (name, mb, ms * 1e3, nbytes / ms / 1e9))
