"""LATENT-OPTIMISATION mode on the MI355X (flag_opt_motion_latent / flag_opt_traj_latent, global_recon_model.py:43-44,155-158,434-437,
619-622): the priors run inside the Adam loop and the latent draws are parameters -- SURVEY.md 8f row 4.  Against fixtures the UNMODIFIED
reference produced with both flags switched on (oracle/make_golden.py gen_grecon_latent): the first iteration's gradient w.r.t.
`motion_latent` (reprojection loss -> stage kernel's dL/d j_local -> glamr_smpl_backward -> glamr_nets_infill_backward through every window)
and the state after K iterations of every stage."""
import os
import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from glamr_amd.utils import synth
from tests.grecon_common import kp_err

pytestmark = pytest.mark.gpu


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

    def make(cfg_id, **flags):
        cfg = get_config(cfg_id)
        cfg['grecon_model_specs'].update(flags)
        return model_dict['global_recon_model'](cfg, dev, None, smpl=smpl, mt_model=mt)
    return make


@pytest.mark.parametrize('cfg_id,T,P,K', mg.LATENT_CASES)
def test_latent_optimisation_matches_the_reference(make_model, golden, cfg_id, T, P, K):
    g = golden('grecon_latent_%s_T%d_P%d' % (cfg_id, T, P))
    in_dict = synth.make_in_dict(seed=3, num_frames=T, num_persons=P, smpl_model=synth.make_smpl_model())
    lat = mg.latents_for(in_dict, 3)
    model = make_model(cfg_id, flag_opt_motion_latent=True, flag_opt_traj_latent=True)
    model.latent_trace = {}
    out = model.optimize(in_dict, latents=lat, max_iters=K)
    tr = model.latent_trace
    stage = next(iter(model.opt_stage_specs))
    # first iteration: what the re-run priors produce, and the gradient that reaches the motion latent
    n = int(g['init_p0_exist_len'])
    assert np.abs(tr['smpl_pose'][0, :T] - g['%s_fwd_p0_smpl_pose' % stage]).max() < 1e-4
    assert np.abs(tr['traj_local_pred'][0, :n] - g['%s_fwd_p0_traj_local_pred' % stage]).max() < 1e-4
    ref = g['%s_grad_p0_motion_latent' % stage]
    got = tr['g_motion_latent'][0, :ref.shape[0]]
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    print('%s T=%d: d loss / d motion_latent, first iteration: relative error %.2e (largest %.3g); traj_latent: gradient None in the reference = %s'
          % (cfg_id, T, err, np.abs(ref).max(), bool(g['%s_gradnone_p0_traj_latent' % stage])))
    assert err < 1e-3
    assert bool(g['%s_gradnone_p0_traj_latent' % stage])
    # after K iterations of every stage
    pd = out['person_data'][0]
    vis = g['init_p0_vis_frames']
    e_lat = np.abs(pd['motion_latent'] - g['opt_p0_motion_latent']).max()
    e_pose = np.abs(pd['smpl_pose'] - g['opt_p0_smpl_pose']).max()
    e_kp = kp_err(pd['kp_2d_pred'], g['opt_p0_kp_2d_pred'], vis)
    e_tr = np.abs(pd['root_trans_world'] - g['opt_p0_root_trans_world']).max()
    print('after %d iterations per stage: motion_latent %.2e (moved %.2e), smpl_pose %.2e, kp_2d_pred %.3f px, root_trans_world %.2e'
          % (K, e_lat, np.abs(g['opt_p0_motion_latent'] - lat[0]['motion']).max(), e_pose, e_kp, e_tr))
    assert np.array_equal(pd['traj_latent'], lat[0]['traj'])                      # never updated: its gradient is None, Adam skips it
    assert np.abs(g['opt_p0_traj_latent'] - lat[0]['traj']).max() == 0.0           # ... in the reference too
    assert np.abs(g['opt_p0_motion_latent'] - lat[0]['motion']).max() > 1e-3       # the motion latent did move
    tol = LATENT_TOL[(cfg_id, T)]
    assert e_lat < tol[0] and e_pose < tol[1] and e_kp < tol[2] and e_tr < tol[3]


# (motion_latent, smpl_pose rad, projected keypoints px, root_trans_world m) after K iterations per stage
LATENT_TOL = {('glamr_dynamic', 100): (1e-4, 1e-5, 0.1, 1e-4),       # achieved 3.4e-6 (moved 1.2e-2), 6.2e-8, 0.013 px, 2.4e-7
              ('glamr_static', 130): (1e-4, 1e-5, 0.1, 1e-4)}        #          1.4e-6 (moved 8.1e-3), 5.7e-8, 0.012 px, 4.2e-6


def test_shipped_configs_are_unaffected_and_flags_are_read(make_model):
    m = make_model('glamr_dynamic')
    assert not m.latent_mode
    assert make_model('glamr_dynamic', flag_opt_motion_latent=True).latent_mode and make_model('glamr_dynamic', flag_opt_traj_latent=True).latent_mode
