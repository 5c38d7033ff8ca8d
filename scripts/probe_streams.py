"""dev probe: aggregate B=256 throughput of S independent engines (user shards) on S HIP streams of one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'top-k-rec_amd')]
import numpy as np, torch, synth, tkr_hip
from single import _engine
r = synth.make_ratings(**synth.ML10M, seed=42)
row_ptr, pos, srt, tr_users = synth.positives_csr(r)
dev = torch.device('cuda', 0)
n_users, n_items, k, B, nb = r['n_users'], r['n_in'] + r['n_out'], 128, 256, 512
hp = dict(lu=2.5e-3, li=2.5e-3, lj=2.5e-4, lb=0.0, lr=1e-4, mode='l2')
for S in (1, 2, 4, 8, 16):
    engs, csrs, plans, streams = [], [], [], []
    for s in range(S):
        eng = _engine.BprEngine(n_users, n_items, k, hp, dev, seed=s)
        csr = _engine.TrainingCSR.from_arrays(row_ptr, pos, tr_users[s::S], dev)
        plan = _engine.PlanBuffers(nb, B, dev)
        tkr_hip.sample_plan(csr, n_users, n_items, s, 0, nb, B, eng.cnt, plan)
        engs.append(eng); csrs.append(csr); plans.append(plan); streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    states = [e.state() for e in engs]
    def run(reps):
        for _ in range(reps):
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    tkr_hip.bpr_run(states[s], plans[s], B, nb, None)
    run(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(6); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('S=%2d streams: %.2f us per batch per stream, aggregate %.1f M triplets/s' % (S, dt / (6 * nb) * 1e6, 6 * nb * S * B / dt / 1e6), flush=True)
    del engs, csrs, plans
