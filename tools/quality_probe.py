import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import ensure_assets
from oracle import make_golden as mg
from glamr_amd.utils import synth
from glamr_amd.global_recon.models import model_dict
from glamr_amd.global_recon.configs import get_config
from glamr_amd.lib.models.smpl import SMPL
from glamr_amd.models.prior_models import MotionTrajJointModel
root = ensure_assets(); dev = torch.device('cuda:0')
smpl = SMPL(os.path.join(root, 'data', 'body_models', 'smpl'), pose_type='body26fk', extra_regressor_path=os.path.join(root, 'data', 'J_regressor_extra.npy')).to(dev)
mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(root, 'results'))
for tag, gap in (('', None), ('_nogap', (0, 0))):
    g = np.load('tests/golden/full_glamr_dynamic_T300%s.npz' % tag)
    in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model(), gap=gap)
    lat = mg.latents_for(in_dict, 0)
    vis = g['p0_vis_frames']
    obs = np.zeros((300, 24, 2)); obs[vis] = in_dict['est'][0]['kp_2d'][:, :24, :2]
    def repro(kp): return np.linalg.norm(kp[vis][:, :24] - obs[vis], axis=-1)
    r = repro(g['p0_kp_2d_pred'])
    print('== %s reference: reprojection error mean %.3f median %.3f max %.2f px' % (tag or 'gap', r.mean(), np.median(r), r.max()))
    for name in ('device', 'host'):
        m = model_dict['global_recon_model'](get_config('glamr_dynamic'), dev, None, smpl=smpl, mt_model=mt)
        if name == 'host': m.init_data_batch = m.init_data_batch_host
        out = m.optimize(in_dict, latents=lat)
        pd = out['person_data'][0]
        r = repro(pd['kp_2d_pred'])
        d = np.abs(pd['kp_2d_pred'] - g['p0_kp_2d_pred'])[vis].max(axis=(1, 2))
        print('   %s init: reprojection mean %.3f median %.3f max %.2f | vs golden kp max %.3f median-over-frames %.3f, frames > 1px: %d %s | kp loss %.2f' % (
            name, r.mean(), np.median(r), r.max(), d.max(), np.median(d), (d > 1).sum(), np.where(vis)[0][d > 1][:12], float(m.last_losses[0][0])))
