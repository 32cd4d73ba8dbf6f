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
cfg_id, T, P, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = dict(np.load('tests/golden/grecon_%s_T%d_P%d.npz' % (cfg_id, T, P)))
in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model())
m = model_dict['global_recon_model'](get_config(cfg_id), dev, None, smpl=smpl, mt_model=mt)
lat = mg.latents_for(in_dict, 3)
for name in ('device', 'host'):
    if name == 'host': m.init_data_batch = m.init_data_batch_host
    out = m.optimize(in_dict, latents=lat, max_iters=K)
    pd = out['person_data'][0]
    for key in ('smpl_orient_world', 'root_trans_world', 'kp_2d_pred', 'world_dheading', 'traj_local_rot', 'traj_local_z', 'traj_local_dxy'):
        if 'opt_p0_' + key in g and key in pd:
            e = np.abs(np.asarray(pd[key]) - g['opt_p0_' + key])
            print(name, key, e.max(), np.unravel_index(e.argmax(), e.shape))
    print(name, 'cam', np.abs(out['cam_pose'] - g['opt_cam_pose']).max())
