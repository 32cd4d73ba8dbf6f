"""DEVELOPMENT AID (GPU): the pipelined host path (optimize_stream) batch by batch with host timestamps and device events: the loop of
GlobalReconOptimizer._stream_loop spelled out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ensure_assets, build_model
from glamr_amd.utils import synth
root = ensure_assets(); dev = torch.device('cuda:0')
m = build_model(root, dev)
md = synth.make_smpl_model()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = [synth.make_in_dict(seed=s, num_frames=300, num_persons=1, smpl_model=md) for s in range(B)]
for _ in m.optimize_stream([base] * 5):
    pass
torch.cuda.synchronize()
computes = m._compute_streams; down = m._download_stream
T0 = time.time()
now = lambda: (time.time() - T0) * 1e3
log = []
def stage(k):
    t = now()
    with torch.cuda.stream(down):
        r = m.stage_inputs(base)
    return r, t, now()
rin, a, b = stage(0)
log.append(('stage', 0, a, b))
flight = []
ev = []
NB = 10
def fetch_collect(k, datas, packed, done):
    t = now(); down.wait_event(done); f = m._fetch_async(packed, down); t1 = now()
    m.collect(datas, packed, f); log.append(('fetch_enq', k, t, t1)); log.append(('collect', k, t1, now()))
for k in range(NB):
    cs = computes[k % 2]
    t_e0 = now()
    cs.wait_event(rin.upload_done)
    with torch.cuda.stream(cs):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        datas, packed = m._resident_for_stream(rin, None)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        done = torch.cuda.Event(); done.record()
    log.append(('enqueue', k, t_e0, now()))
    ev.append((e0, e1))
    flight.append((k, datas, packed, done))
    if k < NB - 1:
        rin, a, b = stage(k + 1); log.append(('stage', k + 1, a, b))
    if len(flight) > 2:
        fetch_collect(*flight.pop(0))
while flight:
    fetch_collect(*flight.pop(0))
torch.cuda.synchronize()
print('total %.1f ms for %d batches = %.1f ms per batch' % (now(), NB, now() / NB))
for name, k, a, b in log:
    print('%-10s batch %d  %7.1f -> %7.1f  (%5.1f ms)' % (name, k, a, b, b - a))
base_ev = ev[0][0]
for k, (e0, e1) in enumerate(ev):
    print('device compute batch %d: start +%.1f ms, duration %.1f ms' % (k, base_ev.elapsed_time(e0), e0.elapsed_time(e1)))
