"""DEVELOPMENT AID (GPU): the two-person latent-optimisation fixture key by key, per person."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from oracle import make_golden as mg
from glamr_amd.utils import synth
from glamr_amd.global_recon.configs import get_config
from glamr_amd.global_recon.models import model_dict
from tests.grecon_common import _rot_err

dev = torch.device('cuda:0')
base = bench.build_model(bench.ensure_assets(), dev)
cfg = get_config('glamr_dynamic_multi')
cfg['grecon_model_specs'].update(flag_opt_motion_latent=True, flag_opt_traj_latent=True)
model = model_dict['global_recon_model'](cfg, dev, None, smpl=base.smpl, mt_model=base.mt_model)
g = np.load(os.path.join(mg.GOLD, 'grecon_latent_glamr_dynamic_multi_T90_P2.npz'))
in_dict = synth.make_in_dict(seed=3, num_frames=90, num_persons=2, smpl_model=synth.make_smpl_model(), gap=mg.LATENT_GAP.get(('glamr_dynamic_multi', 90, 2)))
lat = mg.latents_for(in_dict, 3)
out = model.optimize(in_dict, latents=lat, max_iters=5)
for pi in range(2):
    pd = out['person_data'][pi]
    vis = g['init_p%d_vis_frames' % pi]
    for k in ('motion_latent', 'smpl_pose', 'traj_local_pred', 'root_trans_world', 'kp_2d_pred', 'traj_local_xy', 'traj_local_heading', 'traj_local_rot', 'world_dheading'):
        if 'opt_p%d_%s' % (pi, k) in g.files and k in pd:
            d = np.abs(np.asarray(pd[k], np.float64).reshape(g['opt_p%d_%s' % (pi, k)].shape) - g['opt_p%d_%s' % (pi, k)])
            print('person %d %-20s max %.3e at %s' % (pi, k, d.max(), np.unravel_index(d.argmax(), d.shape)))
    print('person %d orientation (as rotation) %.3e' % (pi, _rot_err(pd['smpl_orient_world'], g['opt_p%d_smpl_orient_world' % pi])))
    d = np.abs(pd['kp_2d_pred'] - g['opt_p%d_kp_2d_pred' % pi]).max(axis=(1, 2))
    print('   frames with kp error > 1 px:', np.flatnonzero(d > 1).tolist()[:30], 'visible there:', vis[np.flatnonzero(d > 1)][:30].tolist())
for k in ('cam_pose',):
    print(k, np.abs(out[k] - g['opt_' + k]).max())
