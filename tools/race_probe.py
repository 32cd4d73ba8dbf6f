"""DEVELOPMENT AID (GPU): the two-stream pipeline checked pair by pair (the tool that found round 6's packed-fp32 corruption,
profiles/r06_pipeline_corruption.log; tests/test_pipeline_soak_gpu.py is its permanent form).
Two step graphs (one per stream) are replayed as a PAIR, the device is synchronised, and every array of both is compared bit for bit with a plain
step of the same seed; the seeds alternate between iterations so that a value left over from the previous replay is a wrong one.
To see the corruption again build a library WITH packed-fp32 instructions: GLAMR_VARIANT_PACKED=" " tools/build_variant_files.sh packed
"smpl.hip init.hip nets.hip eval.hip"; GLAMR_LIB_PATH=tools/_lib_packed.so GLAMR_SKIN_AFTER_PRIORS=1 python tools/race_probe.py
knobs: GLAMR_GATE_PREP=late, GLAMR_SKIN_AFTER_PRIORS=1, GLAMR_PROBE_TWO_INPUTS=1, GLAMR_PROBE_LATENTS=1, GLAMR_PROBE_SNAPSHOT=1 (the skinning's workspace
copied inside the graphs), GLAMR_PROBE_HOST_WAIT=1, GLAMR_PROBE_SPIN_US=n, GLAMR_PROBE_B=graph|nets|mm|add|sleep|none (what the second stream runs)
usage: python tools/race_probe.py [n_sequences] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from glamr_amd import _lib
from glamr_amd.utils import synth
from glamr_amd.global_recon.models.global_recon_model import PipelineGate

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = _lib.lib()

model = bench.build_model(bench.ensure_assets(), dev)
md = synth.make_smpl_model()
# GLAMR_PROBE_TWO_INPUTS=1: the two streams work on DIFFERENT batches -- a wrong value that equals the other stream's right one came from there
two_inputs = os.environ.get('GLAMR_PROBE_TWO_INPUTS') == '1'
def staged(seed0):
    dicts = [synth.make_in_dict(seed=seed0 + s, num_frames=bench.NUM_FRAMES, num_persons=1, smpl_model=md) for s in range(B)]
    lat = None
    if os.environ.get('GLAMR_PROBE_LATENTS') == '1':      # GIVEN latents (copied into the priors' arrays inside the step graph) instead of draws
        from glamr_amd.models.prior_models import num_windows, NZ
        rng = np.random.default_rng(seed0)
        lat = [{idx: {'motion': rng.standard_normal((num_windows(bench.NUM_FRAMES), NZ)).astype(np.float32), 'traj': rng.standard_normal(NZ).astype(np.float32)} for idx in d['est']} for d in dicts]
    return model.stage_inputs(dicts, lat)


rin = staged(0)
rins = [rin, staged(5000) if two_inputs else rin]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
torch.cuda.synchronize()
model.pipeline_gate = PipelineGate()
if os.environ.get('GLAMR_PROBE_HOST_WAIT') == '1':
    # the gate's wait on the HOST (no cross-queue barrier packet on the device): does the corruption need the device-side wait?
    def host_before(stream, gate=model.pipeline_gate):
        if gate.last is not None:
            gate.last.synchronize()
    model.pipeline_gate.before = host_before
SPIN_US = int(os.environ.get('GLAMR_PROBE_SPIN_US', '0'))
if SPIN_US:
    # the other stream's first kernels SPIN_US later: torch._sleep cycles after the gate's wait
    orig_before = model.pipeline_gate.before
    def delayed_before(stream):
        orig_before(stream)
        with torch.cuda.stream(stream):
            torch.cuda._sleep(int(SPIN_US * 2100))
    model.pipeline_gate.before = delayed_before

# GLAMR_PROBE_SNAPSHOT=1: the skinning's workspace (what smpl_prep_kernel / smpl_lbs_kernel / smpl_finish_kernel hand to each other) is copied
# right after the three kernels, in stream order, inside the graphs as well: which array is the FIRST wrong one
snapshot = os.environ.get('GLAMR_PROBE_SNAPSHOT') == '1'
snap_latest = {}
if snapshot:
    def rrj(body_pose, betas, smpl=model.smpl):
        Bf = body_pose.shape[0]
        z = smpl.__dict__.get('_origin')
        if z is None or z.shape[0] < Bf:
            z = smpl.__dict__['_origin'] = torch.zeros((Bf, 3), dtype=torch.float32, device=dev)
            torch.cuda.current_stream(dev).synchronize()
        h = smpl._handle(dev)
        joints = torch.empty((Bf, smpl.n_out, 3), device=dev, dtype=torch.float32)
        ws = torch.empty(L.glamr_smpl_workspace_bytes(h, Bf), device=dev, dtype=torch.uint8)
        pose_in = body_pose.clone()
        _lib.check(L.glamr_smpl_forward(h, Bf, _lib.ptr(body_pose), _lib.ptr(betas), _lib.ptr(z), None, None, _lib.ptr(joints), 2, _lib.ptr(ws), _lib.current_stream()))
        bp = (Bf + 31) // 32 * 32
        al = lambda x: (x + 255) // 256 * 256
        off, out = 0, {'pose_in': pose_in, 'pose_after': body_pose.clone()}
        for name, nfl in (('feat', bp * 224), ('feat_h', bp * 224), ('askin', bp * 288), ('askin_h', bp * 384), ('chain', bp * 72), ('picked', bp * 63), ('partial', 3 * bp * 6), ('pivot', bp * 3)):
            if name not in ('feat', 'askin'):
                out[name] = ws[off:off + nfl * 4].clone()
            off = al(off + nfl * 4)
        assert off == ws.numel(), (off, ws.numel())
        snap_latest.clear()
        snap_latest.update(out)
        return joints
    model.smpl.root_relative_joints = rrj
for st, r in zip(streams, rins):
    with torch.cuda.stream(st):
        model.optimize_resident(r)
torch.cuda.synchronize()
graphs, graph_snaps = [], []
for st, r in zip(streams, rins):
    graphs.append(model.capture_resident(r, stream=st, check=False))
    graph_snaps.append(dict(snap_latest))
torch.cuda.synchronize()
ref_snaps = {}
seeds = (7, 11)
refs = {}
for gi, r in enumerate(rins):
    for seed in seeds:
        torch.manual_seed(seed)
        with torch.cuda.stream(streams[gi]):
            _, ref = model.optimize_resident(r)
        torch.cuda.synchronize()
        keys = [k for k, v in ref.t.items() if torch.is_tensor(v) and v.numel() > 1]
        ref_snaps[(gi, seed)] = {k: v.clone() for k, v in snap_latest.items()}
        refs[(gi, seed)] = ({k: ref.t[k].clone() for k in keys}, {k: v.clone() for k, v in ref.person_arrays.items() if torch.is_tensor(v)}, [x.clone() for x in ref.latents])
model.pipeline_gate.last = None
print('gate cut: %s, skinning %s, %d sequences' % (os.environ.get('GLAMR_GATE_PREP', 'late'), 'after the priors (old order)' if os.environ.get('GLAMR_SKIN_AFTER_PRIORS') == '1' else 'before the predictor', B))


def compare(g, ref):
    want, want_pa, want_lat = ref
    bad = []
    for k, w in want.items():
        got = g.packed.t.get(k)
        if got is None or got.shape != w.shape or torch.equal(got, w):
            continue
        d = (got.float() - w.float()).abs()
        bad.append('%s %.3g (%d values)' % (k, float(d.max()), int((d > 0).sum())))
    for k, w in want_pa.items():
        got = g.packed.person_arrays.get(k)
        if got is not None and got.shape == w.shape and not torch.equal(got, w):
            bad.append('pa.%s (%d values)' % (k, int((got != w).sum())))
    for j, (a, b) in enumerate(zip(g.packed.latents, want_lat)):
        if not torch.equal(a, b):
            bad.append('latent%d' % j)
    rows = None
    if 'j_local' in want and not torch.equal(g.packed.t['j_local'], want['j_local']):
        d = (g.packed.t['j_local'] - want['j_local']).abs().flatten(2).max(dim=2).values
        rows = torch.nonzero(d > 0).tolist()
    return bad, rows


# GLAMR_PROBE_B: what the second stream of a pair runs once the first one's gate opens -- 'graph' (its own step graph, default), 'mm' (rocBLAS
# GEMMs), 'add' (elementwise kernels), 'nets' (this library's priors alone, plain launches), 'none'
B_KIND = os.environ.get('GLAMR_PROBE_B', 'graph')
if B_KIND != 'graph':
    xa = torch.randn(8192, 8192, device=dev); xb = torch.randn(8192, 8192, device=dev); xc = torch.empty(8192, 8192, device=dev)
    big = torch.zeros(64 << 20, device=dev)


STATE = {}


def substitute(gi_first):
    st = streams[1 - gi_first]
    with torch.cuda.stream(st):
        st.wait_event(model.pipeline_gate.last)
        if B_KIND == 'mm':
            for _ in range(6):
                torch.mm(xa, xb, out=xc)
        elif B_KIND == 'add':
            for _ in range(40):
                big.add_(1.0)
        elif B_KIND == 'sleep':
            torch.cuda._sleep(10_000_000)
        elif B_KIND == 'nets':
            # the priors alone, plain launches, on buffers of their own
            g2 = graphs[1 - gi_first].packed
            n_slots = g2.S * g2.P
            if 'rs3' not in STATE:
                rs_b = model.mt_model.handle.resident_set(n_slots, g2.T, g2.latents[0].shape[1])
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    STATE['rs3'] = model.mt_model.handle.resident_set(n_slots, g2.T, g2.latents[0].shape[1])
                for k in ('nets_pose', 'nets_vis', 'meps', 'teps'):
                    STATE['rs3'][k].copy_(rs_b[k])
                STATE['rs3']['persistent'] = True
            rs3 = STATE['rs3']
            model.mt_model.infer_padded(rs3['nets_pose'], rs3['nets_vis'], rins[1 - gi_first].lens, rs3['meps'], rs3['teps'], buffers=rs3, coschedule=True)


n_bad = 0
for it in range(N):
    order = (0, 1) if it % 2 == 0 else (1, 0)
    seed_of = {order[0]: seeds[it % 2], order[1]: seeds[(it + 1) % 2]}
    torch.cuda.synchronize()
    for k, gi in enumerate(order):
        if k == 1 and B_KIND != 'graph':
            substitute(order[0])
            continue
        torch.manual_seed(seed_of[gi])
        graphs[gi].replay()
    torch.cuda.synchronize()
    for gi in (order if B_KIND == 'graph' else order[:1]):
        bad, rows = compare(graphs[gi], refs[(gi, seed_of[gi])])
        n_bad += bool(bad)
        print('iteration %d graph %d (%s of the pair, seed %d): %s' % (it, gi, 'first' if gi == order[0] else 'second', seed_of[gi], '; '.join(bad) if bad else 'bit-identical'))
        if rows:
            got, want = graphs[gi].packed.t['j_local'], refs[(gi, seed_of[gi])][0]['j_local']
            other = refs[(1 - gi, seed_of[1 - gi])][0]['j_local']
            for (sl, fr) in rows[:10]:
                dj = (got[sl, fr] - want[sl, fr]).abs().max(dim=1).values
                js = torch.nonzero(dj > 0).flatten().tolist()
                eq_other = bool(torch.equal(got[sl, fr][js], other[sl, fr][js]))
                near = [d for d in range(-8, 9) if d and 0 <= fr + d < want.shape[1] and torch.equal(got[sl, fr][js], want[sl, fr + d][js])]
                print('      slot %d frame %d: %d output joints wrong %s max %.3g m; equal to the OTHER stream right values: %s; equal to this stream frame +d for d in %s'
                      % (sl, fr, len(js), js, float(dj.max()), eq_other, near))
            if snapshot:
                gs, rs = graph_snaps[gi], ref_snaps[(gi, seed_of[gi])]
                bad_frames = sorted({sl * 300 + fr for sl, fr in rows})
                bp = B * 300
                for name in ('pose_in', 'pose_after', 'chain', 'picked', 'pivot', 'partial', 'feat_h', 'askin_h'):
                    a_, b_ = gs[name], rs[name]
                    if torch.equal(a_, b_):
                        print('      snapshot %s: identical' % name)
                        continue
                    if name in ('pose_in', 'pose_after'):
                        fr_bad = torch.nonzero((a_ != b_).any(dim=1)).flatten().tolist()
                    elif name in ('chain', 'picked', 'pivot'):
                        fr_bad = torch.nonzero((a_.view(torch.float32).view(bp, -1) != b_.view(torch.float32).view(bp, -1)).any(dim=1)).flatten().tolist()
                    elif name == 'partial':
                        fr_bad = sorted(set(torch.nonzero((a_.view(torch.float32).view(3, bp, 6) != b_.view(torch.float32).view(3, bp, 6)).any(dim=2))[:, 1].tolist()))
                    else:      # fragment-major fp16 planes: 32-frame tiles
                        per = a_.numel() // (bp // 32)
                        tl = torch.nonzero((a_.view(bp // 32, per) != b_.view(bp // 32, per)).any(dim=1)).flatten().tolist()
                        fr_bad = ['tile %d (frames %d..%d), %d halves differ' % (t, 32 * t, 32 * t + 31, int((a_.view(bp // 32, per)[t].view(torch.int16) != b_.view(bp // 32, per)[t].view(torch.int16)).sum())) for t in tl]
                    print('      snapshot %s: DIFFERS in %d frames/tiles, first %s; j_local bad frames %s' % (name, len(fr_bad), fr_bad[:8], bad_frames[:8]))
                    if name == 'chain':
                        x, y = a_.view(torch.float32).view(bp, 24, 3), b_.view(torch.float32).view(bp, 24, 3)
                        for f in fr_bad[:6]:
                            dj = (x[f] - y[f]).abs().max(dim=1).values
                            print('         chain frame %d (block %d, frame-in-block %d): joints wrong %s max %.3g' % (f, f // 8, f % 8, torch.nonzero(dj > 0).flatten().tolist(), float(dj.max())))
            print('   j_local rows (slot, frame): %d, first %s; blocks of smpl_prep_kernel %s' % (len(rows), rows[:12], sorted({(r[0] * 300 + r[1]) // 8 for r in rows})[:24]))
print('SUMMARY: %d of %d graph replays differed from the plain step' % (n_bad, 2 * N))
