"""End-to-end MI355X tests of the drop-in entry point GlobalReconOptimizer.optimize (host preprocessing + prior kernels + SMPL
kernel + fused optimiser) against fixtures produced by the UNMODIFIED reference on the same synthetic inputs and latents."""
import os
import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from glamr_amd.utils import synth

pytestmark = pytest.mark.gpu
INDEX_KEYS = ('visible', 'visible_orig', 'exist_frames', 'vis_frames', 'invis_frames', 'kp_2d_score')


@pytest.fixture(scope='module')
def make_model(asset_root):
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.lib.models.smpl import SMPL
    from glamr_amd.models.prior_models import MotionTrajJointModel
    dev = torch.device('cuda:0')
    smpl = SMPL(os.path.join(asset_root, 'data', 'body_models', 'smpl'), pose_type='body26fk',
                extra_regressor_path=os.path.join(asset_root, 'data', 'J_regressor_extra.npy')).to(dev)
    mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(asset_root, 'results'))

    def make(cfg_id):
        return model_dict['global_recon_model'](get_config(cfg_id), dev, None, smpl=smpl, mt_model=mt)
    return make


@pytest.mark.parametrize('cfg_id,T,P,K', mg.GRECON_CASES)
def test_optimize_matches_reference_fixture(make_model, golden, cfg_id, T, P, K):
    g = golden('grecon_%s_T%d_P%d' % (cfg_id, T, P))
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=md)
    model = make_model(cfg_id)
    # state right after init_data: indices bit-exact, continuous quantities to fp32 round-off
    data = model.init_data(in_dict, latents=mg.latents_for(in_dict, 3))
    for pi in range(P):
        pd = data['person_data'][pi]
        for key in INDEX_KEYS:
            assert np.array_equal(np.asarray(pd[key]), g['init_p%d_%s' % (pi, key)]), 'frame/visibility indexing must be bit-exact: ' + key
        assert int(pd['fr_start']) == int(g['init_p%d_fr_start' % pi]) and int(pd['fr_end']) == int(g['init_p%d_fr_end' % pi])
        for key, tol in (('smpl_pose', 1e-4), ('traj_local_pred', 1e-4), ('smpl_orient_world', 2e-4), ('root_trans_world', 2e-4), ('kp_2d_pred', 5e-2)):
            err = np.abs(np.asarray(pd[key], dtype=np.float64) - g['init_p%d_%s' % (pi, key)]).max()
            assert err < tol, 'init %s: %g' % (key, err)
    seen = g['init_p0_vis_frames']
    assert np.abs(np.asarray(data['cam_pose'])[seen] - g['init_cam_pose'][seen]).max() < 2e-4
    # K iterations per stage
    out = model.optimize(in_dict, latents=mg.latents_for(in_dict, 3), max_iters=K)
    for pi in range(P):
        pd = out['person_data'][pi]
        vis = g['init_p%d_vis_frames' % pi] & g['init_p0_vis_frames']
        err = np.abs(pd['kp_2d_pred'] - g['opt_p%d_kp_2d_pred' % pi])[vis].max()
        assert err < 0.5, 'kp_2d_pred after optimisation: %g px' % err
        if cfg_id != 'glamr_3dpw':
            for key in ('smpl_orient_world', 'root_trans_world'):
                err = np.abs(pd[key] - g['opt_p%d_%s' % (pi, key)]).max()
                assert err < 1e-2, '%s: %g' % (key, err)
    assert out['cam_pose'].shape == (T, 4, 4) and out['seq_len'] == T


def test_full_schedule_300_frames(make_model, golden):
    """BASELINE.json configs[1]: 300 frames, 1 person, dynamic camera, the full 500-iteration schedule."""
    g = golden('full_glamr_dynamic_T300')
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=md)
    out = make_model('glamr_dynamic').optimize(in_dict, latents=mg.latents_for(in_dict, 0))
    pd = out['person_data'][0]
    vis = g['p0_vis_frames']
    e_kp = np.abs(pd['kp_2d_pred'] - g['p0_kp_2d_pred'])[vis].max()
    e_cam = np.abs(out['cam_pose'] - g['cam_pose'])[vis].max()
    # joints in the camera frame are what the loss sees: compare the root in camera coordinates
    def root_cam(cam, trans):
        return np.einsum('tij,tj->ti', cam[:, :3, :3], trans) + cam[:, :3, 3]
    e_root = np.abs(root_cam(out['cam_pose'], pd['root_trans_world']) - root_cam(g['cam_pose'], g['p0_root_trans_world']))[vis].max()
    print('full schedule: kp %.3f px, cam %.2e, root-in-camera %.2e m' % (e_kp, e_cam, e_root))
    assert e_kp < 1.0 and e_root < 2e-2
