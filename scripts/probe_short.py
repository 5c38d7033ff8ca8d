"""the driver's short call (bench.py --steps 20 --warmup 5) taken apart, without a profiler:
python scripts/probe_short.py [steps] [reps] [B] [shape]
wall = synchronize .. run_batches(steps) .. synchronize; host = when run_batches returned; floor = one empty launch + synchronize"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch
import bench, tkr_hip
from single import _engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
shape = sys.argv[4] if len(sys.argv) > 4 else 'ml10m'
dev = torch.device('cuda', 0)
r, csr, eng, nnz = bench.build_problem(shape, 128, 0, 1, dev)
eng.run_batches(csr, 5, B, want_loss=False)
torch.cuda.synchronize()
x = torch.zeros(64, device=dev)
fl = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x.add_(1.0)
    torch.cuda.synchronize()
    fl.append(time.perf_counter() - t0)
walls, hosts, devs = [], [], []
inner = os.environ.get('TKR_PROBE_EVENTS') == '1'    # bench.py's HIP events around the step launch, inside the call
if inner:
    eng.reserve_events(4)
for _ in range(reps):
    if inner:
        eng.settle()
        eng.step_events = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    eng.run_batches(csr, steps, B, want_loss=False)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    walls.append(t2 - t0); hosts.append(t1 - t0); devs.append(e0.elapsed_time(e1) * 1e-3)
eng.check()
print('first calls (us):', ' '.join('%.0f' % (w * 1e6) for w in walls[:8]), flush=True)
med = lambda v: float(np.median(v)) * 1e6
print('%s B %d steps %d: wall %.1f us (min %.1f)  host returns at %.1f us  events %.1f us  floor (1 launch + sync) %.1f us  -> %.1f M triplets/s'
      % (shape, B, steps, med(walls), min(walls) * 1e6, med(hosts), med(devs), med(fl), steps * B / med(walls)), flush=True)
if os.environ.get('TKR_PROBE_IDLE') == '1':         # what does an idle device cost the next call?  (clock ramp, cold caches)
    a = torch.randn(4096, 4096, device=dev)
    for label, prep in (('after 0.5 s idle', lambda: time.sleep(0.5)),
                        ('after 0.5 s idle + 20 ms of matmul', lambda: (time.sleep(0.5), [a @ a for _ in range(40)])),
                        ('after 0.5 s idle + 5 batches', lambda: (time.sleep(0.5), eng.run_batches(csr, 5, B, want_loss=False))),
                        ('after 0.5 s idle + 512 batches', lambda: (time.sleep(0.5), eng.run_batches(csr, 512, B, want_loss=False)))):
        ws = []
        for _ in range(6):
            prep()
            eng.settle()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run_batches(csr, steps, B, want_loss=False)
            torch.cuda.synchronize()
            ws.append((time.perf_counter() - t0) * 1e6)
        print('%-40s %s' % (label, ' '.join('%.0f' % w for w in ws)), flush=True)
    eng.check()
if os.environ.get('TKR_PROBE_FRESH') == '1':        # bench.py's sequence on fresh engines: 5 warm-up batches, then the timed call
    for trial in range(4):
        r2, csr2, eng2, _ = bench.build_problem(shape, 128, 0, 1, dev)
        eng2.run_batches(csr2, 5, B, want_loss=False)
        eng2.settle()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng2.run_batches(csr2, steps, B, want_loss=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        eng2.settle(); torch.cuda.synchronize()
        t3 = time.perf_counter()
        eng2.run_batches(csr2, steps, B, want_loss=False)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print('fresh engine %d: first timed call %.0f us (host %.0f), second %.0f us' % (trial, (t2 - t0) * 1e6, (t1 - t0) * 1e6, (t4 - t3) * 1e6), flush=True)
if os.environ.get('TKR_K1_PROF') == '1':            # a library built with -DTKR_K1_PROF (TKR_HIP_LIB): phases of workgroup 0 of K1's kernels
    import ctypes as C
    out = (C.c_uint64 * 32)()
    acc = np.zeros(32)
    n = 10
    for _ in range(n):
        eng.settle(); torch.cuda.synchronize()
        eng.run_batches(csr, steps, B, want_loss=False)
        torch.cuda.synchronize()
        assert tkr_hip.lib().tkr_debug_k1_prof(out) == 0
        acc += np.array(out[:], dtype=np.float64)
    v = acc / n / 100.0                              # 100 MHz -> us
    names = ['draw', 'user sort', 'user tasks', 'user occ', 'item sort', 'item tasks', 'item occ + tail']
    print('sample_plan (us): ' + '  '.join('%s %.1f' % (nm, v[i + 1] - v[i]) for i, nm in enumerate(names)) + '   total %.1f' % (v[7] - v[0]))
    print('gap sample_plan end -> resolve_flow start %.1f us' % (v[8] - v[7]))
    names = ['occurrence versions', 'owner order', 'records']
    print('resolve_flow (us): ' + '  '.join('%s %.1f' % (nm, v[9 + i] - v[8 + i]) for i, nm in enumerate(names)) + '   total %.1f' % (v[11] - v[8]), flush=True)
if os.environ.get('TKR_PROBE_EVCOST') == '1':       # host cost of an event record right behind three kernel launches
    import ctypes as C
    hip = C.CDLL('libamdhip64.so')
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(64)]
    for e in evs:
        e.record()
    torch.cuda.synchronize()
    strm = torch.cuda.current_stream().cuda_stream
    for mode in ('torch', 'raw', 'torch', 'raw'):
        ts = []
        for r_ in range(12):
            torch.cuda.synchronize()
            x.add_(1.0); x.add_(1.0); x.add_(1.0)
            e = evs[r_]
            t0 = time.perf_counter()
            if mode == 'torch':
                e.record()
            else:
                hip.hipEventRecord(C.c_void_p(e.cuda_event), C.c_void_p(strm))
            ts.append((time.perf_counter() - t0) * 1e6)
        print('event record behind 3 launches, %s: %s us' % (mode, ' '.join('%.1f' % t for t in ts)), flush=True)
if os.environ.get('TKR_PLAN_STAMP') == '1':         # a library built with -DTKR_PLAN_STAMP (TKR_HIP_LIB): the planner prologue of the step, workgroups 0 and last
    n = 10
    acc = np.zeros((2, 12))
    for _ in range(n):
        eng.settle(); torch.cuda.synchronize()
        eng.ctl[tkr_hip.FLOW_CTL_PROF:tkr_hip.FLOW_CTL_PROF + 64] = 0
        eng.run_batches(csr, steps, B, want_loss=False)
        torch.cuda.synchronize()
        v = eng.ctl[tkr_hip.FLOW_CTL_PROF:tkr_hip.FLOW_CTL_PROF + 64].cpu().numpy().view(np.uint64).astype(np.float64).reshape(2, 16)[:, :12]
        acc += v - v[0, 0]
    v = acc / n / 100.0
    names = ['phase A', 'drain A', 'rendezvous A', 'phase B', 'drain B', 'rendezvous B', 'commit+drain', 'rendezvous C', 'acquire', 'queue set-up', 'step']
    print('workgroup 0 (us): ' + '  '.join('%s %.1f' % (nm, v[0, i + 1] - v[0, i]) for i, nm in enumerate(names)) + '   total %.1f' % (v[0, 11] - v[0, 0]))
    import ctypes as C
    out = (C.c_uint64 * 32)()
    if tkr_hip.lib().tkr_debug_own_k1_prof(out) == 0:
        w = np.array(out[:], dtype=np.float64) / 100.0
        print('phase A (us, last call): draw %.1f  sorts %.1f  head scan %.1f | users: heads %.1f occ %.1f | items: heads %.1f occ %.1f | tail %.1f' %
              (w[1] - w[0], w[2] - w[1], w[3] - w[2], w[4] - w[3], w[5] - w[4], w[17] - w[16], w[18] - w[17], w[7] - max(w[5], w[18])))
        print('phase B (us, last call): loads+versions %.1f  owner order %.1f  headers to LDS %.1f  records %.1f' % (w[9] - w[8], w[10] - w[9], w[12] - w[10], w[11] - w[12]))
    print('last workgroup: start %.1f  waits until %.1f  acquire %.1f  set-up %.1f  step %.1f  end %.1f' %
          (v[1, 0], v[1, 8], v[1, 9] - v[1, 8], v[1, 10] - v[1, 9], v[1, 11] - v[1, 10], v[1, 11]), flush=True)
