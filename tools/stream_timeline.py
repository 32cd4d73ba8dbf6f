"""DEVELOPMENT AID (GPU): the pipelined host path (optimize_stream) batch by batch with host timestamps and device events: where a batch's
72 ms go when the device work is 51-55 ms."""
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
    with torch.cuda.stream(computes[k % 2]):
        r = m.stage_inputs(base)
    return r, t, now()
rin, a, b = stage(0)
log.append(('stage', 0, a, b))
prev = None
ev = []
NB = 8
for k in range(NB):
    cs = computes[k % 2]
    t_e0 = now()
    with torch.cuda.stream(cs):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        datas, packed = m._resident_for_stream(rin, None)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        done = torch.cuda.Event(); done.record()
    t_e1 = now()
    down.wait_event(done)
    fetched = m._fetch_async(packed, down)
    t_f = now()
    ev.append((e0, e1))
    log.append(('enqueue', k, t_e0, t_e1)); log.append(('fetch_enq', k, t_e1, t_f))
    if k < NB - 1:
        rin, a, b = stage(k + 1); log.append(('stage', k + 1, a, b))
    if prev is not None:
        t = now(); m.collect(*prev); log.append(('collect', k - 1, t, now()))
    prev = (datas, packed, fetched)
t = now(); m.collect(*prev); log.append(('collect', NB - 1, t, now()))
torch.cuda.synchronize()
print('total %.1f ms for %d batches = %.1f ms per batch' % (now(), NB, now() / NB))
for name, k, a, b in log:
    print('%-10s batch %d  %7.1f -> %7.1f  (%5.1f ms)' % (name, k, a, b, b - a))
base_ev = ev[0][0]
for k, (e0, e1) in enumerate(ev):
    print('device compute batch %d: start +%.1f ms, duration %.1f ms' % (k, base_ev.elapsed_time(e0), e0.elapsed_time(e1)))
