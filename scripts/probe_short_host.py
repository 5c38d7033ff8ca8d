"""the driver's timed 20-batch call, split at the moment the engine's call returns: host time in front of / inside the launch vs
launch latency + kernel + the wake-up of the synchronise.  One line per process run (the driver's line is one call of one process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import torch
import bench

dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem('ml10m', 128, 0, 1, dev)
steps, warm = 20, int(os.environ.get('PROBE_WARMUPS', '3'))
inner = []
loop = bench.Loop(eng, csr, 256, 10 ** 9, 1)
for w in range(warm):
    if os.environ.get('PROBE_WARM') == 'singles':
        loop.run(1)
    else:
        eng.run_batches(csr, steps, 256, want_loss=False)
    bench._fence(1)
region = bench._recorded_pair()
if os.environ.get('PROBE_INNER') == '1':
    st = eng._step
    real = st.plan_and_run

    def par(*a):
        t = time.perf_counter()
        real(*a)
        inner.append((time.perf_counter() - t) * 1e6)
    st.plan_and_run = par
if os.environ.get('PROBE_SETTLE') == '1':         # bench.timed_run settles (and checks the status word: a copy to the host) in front of the timed call
    eng.settle()
if os.environ.get('PROBE_SETTLE') == '2':
    eng.settle(check=False)
if os.environ.get('PROBE_RESERVE') == '1':
    eng.reserve_events(12)
bench._fence(1)
ev = os.environ.get('PROBE_EVENT') == '1'
if ev:
    region[0].record()
t0 = time.perf_counter()
if os.environ.get('PROBE_LOOP') == '1':
    loop.run(steps)
else:
    eng.run_batches(csr, steps, 256, want_loss=False)
t1 = time.perf_counter()
if ev:
    region[1].record()
torch.cuda.synchronize(dev)
t2 = time.perf_counter()
print('host %.1f us  then %.1f us  total %.1f us  (C call %s)' % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e6, ' '.join('%.1f' % x for x in inner)))
