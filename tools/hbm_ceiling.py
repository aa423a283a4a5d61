"""Measured streaming ceilings of the box beside the nominal 8 TB/s: fill and device-to-device copy of
the frame batch's size (48 MiB = 4096 x 64 x 64 x 3) and of 1 GiB, timed with HIP events; device facts from torch."""
import torch
dev = torch.device('cuda:0')
p = torch.cuda.get_device_properties(0)
print('device: %s, %d CUs, %.0f GiB, warp %d, LDS/block %d KiB, clock %s MHz' % (
    p.name, p.multi_processor_count, p.total_memory / 2**30, p.warp_size,
    p.shared_memory_per_block // 1024, getattr(p, 'clock_rate', 0) // 1000))
for nbytes, label in ((4096 * 64 * 64 * 3, 'frame batch 48 MiB'), (2**30, '1 GiB')):
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for name, fn, moved in (('fill', lambda: a.fill_(7), nbytes), ('copy', lambda: b.copy_(a), 2 * nbytes)):
        for _ in range(5):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 50
        ev[0].record()
        for _ in range(n):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1e3 / n
        print('%-20s %-5s %8.1f us  %6.2f TB/s (bytes moved %d)' % (label, name, us, moved / us / 1e6, moved))
