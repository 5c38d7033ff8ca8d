"""where the host spends the timed region of `bench.py --steps 20 --warmup 5`: bench.timed_run with timers around the engine's calls"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch
import bench, tkr_hip
from single import _engine

dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
log = []


def timed(obj, name):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            log.append((name, (time.perf_counter() - t0) * 1e6))
    setattr(obj, name, g)


timed(eng, '_next_chunk')
timed(eng, '_plan_chunk')
timed(eng, 'prepare')
timed(eng, '_run')
timed(tkr_hip, 'plan_caller')
orig_step_fn = eng.step_fn


def step_fn(B):
    st = orig_step_fn(B)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return st(*a, **k)
        finally:
            log.append(('step', (time.perf_counter() - t0) * 1e6))
    g.takes_events = getattr(st, 'takes_events', False)
    g.assigns_loss = getattr(st, 'assigns_loss', False)
    if getattr(st, 'plan_and_run', None) is not None:
        def par(*a, **k):
            t0 = time.perf_counter()
            try:
                return st.plan_and_run(*a, **k)
            finally:
                log.append(('plan_and_run', (time.perf_counter() - t0) * 1e6))
        g.plan_and_run = par
    return g


eng.step_fn = step_fn
steps, warm = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 5
if os.environ.get('TKR_PROBE_SPLIT_WARMUP') == '1':          # the warm-up as single-batch calls
    lp = bench.Loop(eng, csr, 256, 10 ** 9, 1)
    for _ in range(warm - 1):
        lp.run(1)
    wall, step_ms, loop = bench.timed_run(eng, csr, 256, steps, 1, 10 ** 9, 1, loop=lp)
else:
    wall, step_ms, loop = bench.timed_run(eng, csr, 256, steps, warm, 10 ** 9, 1)
print('wall %.1f us' % (wall * 1e6))
for name, us in log:
    print('  %-14s %.1f us' % (name, us))
for rep in range(3):
    del log[:]
    wall, step_ms, loop = bench.timed_run(eng, csr, 256, steps, warm, 10 ** 9, 1)
    print('timed_run again: wall %.1f | %s' % (wall * 1e6, ' '.join('%s %.0f' % (n, u) for n, u in log)))
for rep in range(2):
    del log[:]
    eng.settle(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_batches(csr, steps, 256, want_loss=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('call %d: wall %.1f host %.1f | %s' % (rep + 2, (t2 - t0) * 1e6, (t1 - t0) * 1e6, ' '.join('%s %.0f' % (n, u) for n, u in log)))
