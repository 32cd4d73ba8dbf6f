"""DEVELOPMENT AID (GPU).  Perturbs the initial camera poses of THIS repository's numpy init path (not the reference) and reports how far
the kernel's result moves.  The reference's own sensitivity is in tests/golden/full_glamr_dynamic_T300_family.npz (oracle/make_golden.py
gen_full_family); round 1 mis-attributed this probe's output to the reference."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import ensure_assets
from oracle import make_golden as mg
from glamr_amd.utils import synth
from glamr_amd.global_recon.models import model_dict
from glamr_amd.global_recon.configs import get_config
from glamr_amd.global_recon import packing
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
orig = packing.PackedScenes.set_cam_pose
for eps in (1e-7, 1e-6, 1e-5):
    for trial in range(3):
        rng = np.random.RandomState(trial)
        def patched(self, cam_poses):
            cam_poses = [c * (1 + eps * rng.uniform(-1, 1, c.shape)).astype(np.float32) for c in cam_poses]
            return orig(self, cam_poses)
        packing.PackedScenes.set_cam_pose = patched
        o = m.optimize(in_dict, latents=lat)
        packing.PackedScenes.set_cam_pose = orig
        e = np.abs(o['person_data'][0]['kp_2d_pred'] - ref['person_data'][0]['kp_2d_pred'])
        ec = np.abs(o['cam_pose'] - ref['cam_pose'])
        print('cam_pose rel. perturbation %.0e trial %d: kp diff vis %.3f px, all %.3f ; cam diff vis %.3e gap %.3e' % (eps, trial, e[vis].max(), e.max(), ec[vis].max(), ec[~vis].max()))
