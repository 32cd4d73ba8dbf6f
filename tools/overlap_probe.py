"""Development probe: does a smaller optimiser arena (LDS left to co-resident kernels) + narrow GEMM tiles (116 registers) let the
prior networks of one batch run UNDER the optimiser stage of the previous one?  Sweeps the two runtime knobs in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glamr_amd.utils import synth
from glamr_amd import parallel

B = int(os.environ.get('GLAMR_PROBE_BATCH', '1024'))
bench.NUM_FRAMES = int(os.environ.get('GLAMR_PROBE_FRAMES', bench.NUM_FRAMES))
dev = torch.device('cuda:0')
asset_root = bench.ensure_assets()
model = bench.build_model(asset_root, dev)
md = synth.make_smpl_model()
in_dicts = [synth.make_in_dict(seed=sd, num_frames=bench.NUM_FRAMES, num_persons=1, smpl_model=md) for sd in range(B)]
rin = model.stage_inputs(in_dicts)
torch.cuda.synchronize()


POOL = [torch.cuda.Stream(device=dev) for _ in range(3)]      # created ONCE: later streams of torch's pool may share a hardware queue


def run(nstreams, steps=6, warmup=2):
    streams = POOL[:nstreams]
    keep = []
    def step(i):
        with torch.cuda.stream(streams[i % nstreams]):
            _, packed = model.optimize_resident(rin)
        keep.append(packed)
        if len(keep) > 2 * nstreams:
            keep.pop(0)
        return packed
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        p = step(i)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    return ms, [round(model.launch_ms(ws), 1) for ws in p.stage_ws]


CONFIGS = [(150, 0, 1), (150, 0, 2), (120, 0, 2), (120, 1, 2), (120, 1, 1), (96, 1, 2), (96, 1, 3), (64, 1, 2), (64, 0, 2), (64, 1, 1)]
if os.environ.get('GLAMR_PROBE_CONFIGS'):
    CONFIGS = [tuple(int(v) for v in c.split(',')) for c in os.environ['GLAMR_PROBE_CONFIGS'].split(';')]
for cfg in CONFIGS:
    lds, narrow, ns = cfg[:3]
    if len(cfg) > 3:
        os.environ['GLAMR_GRECON_THREADS_RT'] = str(cfg[3])
    else:
        os.environ.pop('GLAMR_GRECON_THREADS_RT', None)
    os.environ.pop('GLAMR_GRECON_LDS_KB_RT', None)          # 0 = the launcher's own policy
    if lds:
        os.environ['GLAMR_GRECON_LDS_KB_RT'] = str(lds)
    if narrow:
        os.environ['GLAMR_GEMM_NARROW'] = '1'
    else:
        os.environ.pop('GLAMR_GEMM_NARROW', None)
    ms, k = run(ns)
    print('threads %s lds %3d KB narrow %d streams %d : %.1f ms/step  %.0f seq/s  last stage launches %s' % (cfg[3] if len(cfg) > 3 else 'auto', lds, narrow, ns, ms, B / ms * 1e3, k), flush=True)
