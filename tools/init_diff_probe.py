"""DEVELOPMENT AID (GPU).  How far is the device init_data (init.hip) from the numpy variant (= the reference's own initial state), array
by array, on BASELINE configs[1] with the detection gap -- and which of the differences decides the solution the optimiser reaches?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import ensure_assets, build_model
from oracle import make_golden as mg
from glamr_amd.utils import synth

root = ensure_assets(); dev = torch.device('cuda:0')
m = build_model(root, dev)
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
lat = mg.latents_for(in_dict, 0)
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'full_glamr_dynamic_T300.npz')))
vis = g['p0_vis_frames']
_, ph = m.init_data_batch_host([in_dict], [lat])
_, pd = m.init_data_batch([in_dict], [lat])
keys = [k for k in ph.t if ph.t[k] is not None and ph.t[k].dtype == torch.float32 and k in pd.t and pd.t[k] is not None and ph.t[k].shape == pd.t[k].shape]
for k in keys:
    a, b = ph.t[k].cpu().numpy(), pd.t[k].cpu().numpy()
    d = np.abs(a - b)
    print('%-22s max abs diff %.3e (max |value| %.3g)' % (k, d.max(), np.abs(a).max()))

def run(packed):
    m.run_schedule(packed)
    torch.cuda.synchronize()
    kp = packed.t['kp_2d_pred'][0, :300].cpu().numpy()
    d = np.abs(kp - g['p0_kp_2d_pred'])[vis].max(axis=(1, 2))
    return d.max(), int((d > 1).sum())

for swap in ([], ['cam_pose'], ['traj_local_pred'], ['j_local'], ['orient_cam'], ['kp_2d'], ['cam_pose', 'traj_local_pred'], keys):
    _, pd = m.init_data_batch([in_dict], [lat])
    for k in swap:
        pd.t[k].copy_(ph.t[k])
    print('device init with %s taken from the host init: %s' % (swap if len(swap) < 6 else 'EVERYTHING', run(pd)))
