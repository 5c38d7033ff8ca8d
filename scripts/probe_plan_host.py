"""does planning the next chunk block the host?  Per _next_chunk call at a large batch size: host time, and the time the step launches of a chunk take to enqueue"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
log = []
for name in ('_next_chunk', '_plan_chunk'):
    f = getattr(eng, name)
    def g(*a, _f=f, _n=name, **k):
        t0 = time.perf_counter()
        try:
            return _f(*a, **k)
        finally:
            log.append((_n, (time.perf_counter() - t0) * 1e3))
    setattr(eng, name, g)
if os.environ.get('PROBE_BENCH_WARMUP') == '1':    # bench.timed_run's warm-up: eight single-batch calls onto an idle queue, then the rest
    for _ in range(8):
        eng.run_batches(csr, 1, B, want_loss=True)
        torch.cuda.synchronize()
    eng.run_batches(csr, 504, B, want_loss=True)
elif os.environ.get('PROBE_BENCH_WARMUP') == '2':  # ... through bench.Loop
    lp = bench.Loop(eng, csr, B, 10 ** 9, 1)
    lp.run(512)
else:
    eng.run_batches(csr, 512, B, want_loss=True)
torch.cuda.synchronize()
if os.environ.get('PROBE_SETTLE') == '1':          # as bench.timed_run: what the warm-up planned ahead is dropped, the timed call plans its first chunk in order
    eng.settle()
    torch.cuda.synchronize()
del log[:]
if os.environ.get('PROBE_EVENTS') == '1':          # as bench.timed_run: a HIP event pair around every step call
    eng.step_events = []
t0 = time.perf_counter()
eng.run_batches(csr, int(os.environ.get('PROBE_STEPS', '1024')), B, want_loss=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('B %d: host %.2f ms, total %.2f ms (%.2f us per batch)' % (B, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / int(os.environ.get('PROBE_STEPS', '1024'))))
print(' '.join('%s %.2f' % x for x in log))
