"""ceiling of a dependency-free random-row gather + scatter on this GPU, by row size: what an interleaved [row][param | slot] layout
(1 KB per touch instead of two 512-byte rows in two tables) could buy K2 at large batches"""
import torch, time
dev = torch.device('cuda')
n = 1 << 20
g = torch.Generator(device=dev); g.manual_seed(1)
for width in (128, 256, 512):
    src = torch.randn((n * 128 // width, width), device=dev)
    dst = torch.empty_like(src)
    rows = src.shape[0]
    for m in (1 << 14, 1 << 16, 1 << 18):
        idx = torch.randint(0, rows, (m,), device=dev, generator=g)
        for _ in range(3):
            dst.index_copy_(0, idx, src.index_select(0, idx))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            t = src.index_select(0, idx)
        e1.record(); torch.cuda.synchronize()
        gather = e0.elapsed_time(e1) / 20
        print('row %4d B, %7d rows: gather %.1f us = %.2f TB/s (read + write of the gathered copy)' %
              (width * 4, m, gather * 1e3, 2 * m * width * 4 / gather / 1e9), flush=True)
