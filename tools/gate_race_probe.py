"""DEVELOPMENT AID (GPU): which arrays of the two streams' replayed steps differ from ONE plain step when the step is cut into three graphs
(GLAMR_GATE_PREP=early, see GlobalReconOptimizer.init_resident): 1024 x 300-frame sequences, alternating replays with the same generator state.
usage: GLAMR_GATE_PREP=early python tools/gate_race_probe.py [n_sequences]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glamr_amd.utils import synth
from glamr_amd.global_recon.models.global_recon_model import PipelineGate
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = bench.build_model(bench.ensure_assets(), dev)
md = synth.make_smpl_model()
rin = model.stage_inputs([synth.make_in_dict(seed=s, num_frames=bench.NUM_FRAMES, num_persons=1, smpl_model=md) for s in range(B)])
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
torch.cuda.synchronize()
model.pipeline_gate = PipelineGate()
for st in streams:
    with torch.cuda.stream(st):
        model.optimize_resident(rin)
torch.cuda.synchronize()
graphs = [model.capture_resident(rin, stream=st, check=True) for st in streams]
torch.cuda.synchronize()
seed = 7
torch.manual_seed(seed)
with torch.cuda.stream(streams[0]):
    _, ref = model.optimize_resident(rin)
torch.cuda.synchronize()
keys = [k for k, v in ref.t.items() if torch.is_tensor(v) and v.numel() > 1]
want = {k: ref.t[k].clone() for k in keys}
want_lat = [x.clone() for x in ref.latents]
for trial in range(3):
    for i in range(8):
        torch.manual_seed(seed)
        graphs[i % 2].replay()
    torch.cuda.synchronize()
    for gi, g in enumerate(graphs):
        bad = []
        for k in keys:
            got = g.packed.t.get(k)
            if got is None or got.shape != want[k].shape:
                continue
            d = (got.float() - want[k].float()).abs()
            if not torch.equal(got, want[k]):
                bad.append('%s %.3g (%d of %d values)' % (k, float(d.max()), int((d > 0).sum()), d.numel()))
        if 'j_local' in want and not torch.equal(g.packed.t['j_local'], want['j_local']):
            d = (g.packed.t['j_local'] - want['j_local']).abs().flatten(2).max(dim=2).values      # (slots, T)
            rows = torch.nonzero(d > 0)
            print('   j_local differs in %d (slot, frame) rows: slots %s frames %s .. %s' % (rows.shape[0], sorted(set(rows[:, 0].tolist()))[:12], rows[:, 1].min().item(), rows[:, 1].max().item()),
                  rows[:20].tolist())
        pa_g, pa_r = getattr(g.packed, 'person_arrays', None), getattr(ref, 'person_arrays', None)
        if pa_g and pa_r:
            for k in pa_g:
                if torch.is_tensor(pa_g[k]) and k in pa_r and pa_g[k].shape == pa_r[k].shape and not torch.equal(pa_g[k], pa_r[k]):
                    bad.append('pa.%s %.3g (%d)' % (k, float((pa_g[k] - pa_r[k]).abs().max()), int((pa_g[k] != pa_r[k]).sum())))
        lat = ['latent%d %.3g' % (j, float((a - b).abs().max())) for j, (a, b) in enumerate(zip(g.packed.latents, want_lat)) if not torch.equal(a, b)]
        print('trial %d graph %d: %s' % (trial, gi, '; '.join(bad + lat) if bad or lat else 'all arrays bit-identical'))
