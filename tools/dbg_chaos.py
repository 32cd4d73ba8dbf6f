import os, sys, copy
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
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
lat = mg.latents_for(in_dict, 0)
m = model_dict['global_recon_model'](get_config('glamr_dynamic'), dev, None, smpl=smpl, mt_model=mt)
m.init_data_batch = m.init_data_batch_host
ref = m.optimize(in_dict, latents=lat)
vis = ref['person_data'][0]['vis_frames']
for eps in (1e-7, 1e-6, 1e-5):
    d2 = copy.deepcopy(in_dict)
    d2['est'][0]['root_trans'] = (d2['est'][0]['root_trans'] * (1 + eps)).astype(np.float32)
    o = m.optimize(d2, latents=lat)
    e = np.abs(o['person_data'][0]['kp_2d_pred'] - ref['person_data'][0]['kp_2d_pred'])
    ec = np.abs(o['cam_pose'] - ref['cam_pose'])
    print('perturb %.0e: kp diff vis %.3f px, all %.3f ; cam diff vis %.3e gap %.3e ; frames with kp diff > 1px: %s' % (eps, e[vis].max(), e.max(), ec[vis].max(), ec[~vis].max(), np.where(e.max(axis=(1, 2)) > 1)[0][:12]))
