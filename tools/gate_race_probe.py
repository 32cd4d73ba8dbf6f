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
            dj = (g.packed.t['j_local'] - want['j_local']).abs().max(dim=3).values      # (slots, T, 26)
            for r in rows[:6].tolist():
                js = torch.nonzero(dj[r[0], r[1]] > 0).flatten().tolist()
                print('      slot %d frame %d: output joints %s (their sources in the 45-joint table: %s)' % (r[0], r[1], js, [int(model.smpl.joint_map[j]) for j in js]))
            print('   j_local differs in %d (slot, frame) rows: slots %s frames %s .. %s' % (rows.shape[0], sorted(set(rows[:, 0].tolist()))[:12], rows[:, 1].min().item(), rows[:, 1].max().item()),
                  rows[:20].tolist())
        pa_g, pa_r = getattr(g.packed, 'person_arrays', None), getattr(ref, 'person_arrays', None)
        if pa_g and pa_r:
            for k in pa_g:
                if torch.is_tensor(pa_g[k]) and k in pa_r and pa_g[k].shape == pa_r[k].shape and not torch.equal(pa_g[k], pa_r[k]):
                    bad.append('pa.%s %.3g (%d)' % (k, float((pa_g[k] - pa_r[k]).abs().max()), int((pa_g[k] != pa_r[k]).sum())))
        lat = ['latent%d %.3g' % (j, float((a - b).abs().max())) for j, (a, b) in enumerate(zip(g.packed.latents, want_lat)) if not torch.equal(a, b)]
        print('trial %d graph %d: %s' % (trial, gi, '; '.join(bad + lat) if bad or lat else 'all arrays bit-identical'))

dbg = getattr(model.smpl, '_dbg_chain', None)
if dbg:
    caps = {k: v for k, v in dbg.items() if k[0]}
    (ka, a), (kb, b) = sorted(caps.items())[:2] if len(caps) >= 2 else (list(caps.items()) * 2)[:2]
    for name, x, y in (('chain joints', a[0], b[0]), ('skinning input pose', a[1], b[1]), ('skinning input betas', a[2], b[2]), ('skinning input pose, snapshot BEFORE the skinning', a[3], b[3])):
        d = (x - y).abs()
        print('snapshots of the two graphs after the last trial: %s differ in %d values (max %.3g)' % (name, int((d > 0).sum()), float(d.max())))
    d = (a[0] - b[0]).abs().max(dim=2).values      # (frames, 24)
    rows = torch.nonzero(d.max(dim=1).values > 0).flatten().tolist()
    for r in rows[:8]:
        print('   frame row %d (slot %d frame %d): chain joints that differ %s' % (r, r // 300, r % 300, torch.nonzero(d[r] > 0).flatten().tolist()))
    dp = (a[3] - b[3]).abs()
    for r in torch.nonzero(dp.max(dim=1).values > 0).flatten().tolist()[:8]:
        ks = torch.nonzero(dp[r] > 0).flatten().tolist()
        print('   pose-before row %d (slot %d frame %d): floats %d .. %d differ (%d of them), i.e. from joint %d on' % (r, r // 300, r % 300, ks[0], ks[-1], len(ks), ks[0] // 3 + 1))
