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
in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=synth.make_smpl_model())
lat = mg.latents_for(in_dict, 0)
for K in (2, 3):
    outs = []
    for name in ('device', 'host'):
        m = model_dict['global_recon_model'](get_config('glamr_dynamic'), dev, None, smpl=smpl, mt_model=mt)
        if name == 'host': m.init_data_batch = m.init_data_batch_host
        outs.append(m.optimize(in_dict, latents=lat, max_iters=K))
    a, b = outs
    e = np.abs(a['cam_pose'] - b['cam_pose']).reshape(300, -1).max(1)
    bad = np.where(e > 1e-3)[0]
    print('K', K, 'cam diff frames', bad[:20], 'max', e.max())
    for key in ('cam_rot_6d', 'cam_trans'):
        e2 = np.abs(a[key] - b[key]).reshape(300, -1).max(1)
        print('   ', key, np.where(e2 > 1e-4)[0][:20], e2.max())
    np.set_printoptions(linewidth=200, precision=6, suppress=False)
    for t in list(bad[:4]) + [99, 160]:
        print('   frame', t, 'device 6d', a['cam_rot_6d'][t], 'tr', a['cam_trans'][t])
        print('   frame', t, 'host   6d', b['cam_rot_6d'][t], 'tr', b['cam_trans'][t])
    for k in a['person_data'][0]:
        x, y = a['person_data'][0][k], b['person_data'][0][k]
        if isinstance(x, np.ndarray) and x.dtype.kind == 'f' and x.shape == y.shape and x.size:
            e = np.abs(x - y)
            if e.max() > 1e-4: print('   person', k, e.max(), np.unravel_index(e.argmax(), e.shape))
