"""MI355X: seeded random scenes -- configuration, length (12 .. 340 frames: partial waves, more frames than a workgroup has threads), persons, detection gaps,
persons that enter late or leave early -- through optimize() against the CPU restatement (oracle/port, pinned to the reference on the fixture cases):
init_data's frame bookkeeping bit for bit, then 3 iterations per stage.  Shapes nobody picked by hand."""
import copy

import numpy as np
import pytest

from glamr_amd.utils import synth
from oracle import make_golden as mg
from tests.grecon_common import kp_err, _rot_err

pytestmark = pytest.mark.gpu

CFGS = [('glamr_dynamic', 1), ('glamr_static', 1), ('glamr_3dpw', 1), ('glamr_dynamic_multi', 2), ('glamr_static_multi', 3), ('glamr_h36m', 2), ('glamr_dynamic_multi', 4),
        ('glamr_static_multi', 2)]


def _case(k):
    rng = np.random.RandomState(1000 + k)
    cfg_id, P = CFGS[k % len(CFGS)]
    T = int(rng.choice([12, 31, 64, 65, 100, 127, 129, 200, 257, 300, 305, 340]))
    gap = None
    if T >= 40 and rng.rand() < 0.6:
        a = int(rng.randint(5, T - 20))
        gap = (a, a + int(rng.randint(3, 12)))
    trims = []
    for pi in range(1, P):
        if T >= 40 and rng.rand() < 0.5:
            first = int(rng.randint(0, T // 3))
            last = int(rng.randint(2 * T // 3, T + 1))
            trims.append((pi, first, last))
    return cfg_id, P, T, gap, trims


@pytest.mark.parametrize('k', range(12))
def test_random_scene_matches_the_oracle(asset_root, k):
    import os
    import torch
    from oracle.port import build
    from glamr_amd.global_recon.configs import get_config
    from glamr_amd.global_recon.models import model_dict
    from glamr_amd.lib.models.smpl import SMPL
    from glamr_amd.models.prior_models import MotionTrajJointModel
    cfg_id, P, T, gap, trims = _case(k)
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=500 + k, num_frames=T, num_persons=P, smpl_model=md, gap=gap)
    for tr in trims:
        synth.trim_person(in_dict, *tr)
    lat = mg.latents_for(in_dict, 500 + k)
    K = 3
    cfg = get_config(cfg_id)
    for spec in cfg['opt_stage_specs'].values():
        spec['opt_niters'] = K
    ora = build.load_optimizer(asset_root, cfg)
    ref = ora.optimize(copy.deepcopy(in_dict), latents=lat)
    dev = torch.device('cuda:0')
    smpl = SMPL(os.path.join(asset_root, 'data', 'body_models', 'smpl'), pose_type='body26fk', extra_regressor_path=os.path.join(asset_root, 'data', 'J_regressor_extra.npy')).to(dev)
    mt = MotionTrajJointModel(None, dev, None, smpl=smpl, results_root=os.path.join(asset_root, 'results'))
    model = model_dict['global_recon_model'](get_config(cfg_id), dev, None, smpl=smpl, mt_model=mt)
    out = model.optimize(in_dict, latents=lat, max_iters=K)
    worst = [0.0, 0.0, 0.0]
    for pi in range(P):
        a, b = out['person_data'][pi], ref['person_data'][pi]
        assert int(a['fr_start']) == int(b['fr_start']) and int(a['fr_end']) == int(b['fr_end']), (cfg_id, T, pi)
        for key in ('visible', 'visible_orig', 'exist_frames', 'vis_frames', 'invis_frames'):
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key]).astype(np.asarray(a[key]).dtype)), (cfg_id, T, pi, key)
        vis = np.asarray(b['vis_frames']).astype(bool) & np.asarray(ref['person_data'][0]['vis_frames']).astype(bool)
        if vis.any():
            worst[0] = max(worst[0], kp_err(a['kp_2d_pred'], np.asarray(b['kp_2d_pred']), vis))
        if cfg_id != 'glamr_3dpw':          # (there the world pose is a gauge: the camera rides on the person)
            worst[1] = max(worst[1], float(np.abs(np.asarray(a['root_trans_world'], np.float64) - np.asarray(b['root_trans_world'])).max()))
            worst[2] = max(worst[2], _rot_err(a['smpl_orient_world'], np.asarray(b['smpl_orient_world'])))
    print('random scene %d: %s, %d frames, %d person(s), gap %s, trims %s: kp %.4f px, root_trans_world %.2e m, smpl_orient_world %.2e' % (k, cfg_id, T, P, gap, trims, *worst))
    assert worst[0] < 0.15 and worst[1] < 1e-4 and worst[2] < 1e-2          # achieved over the 12 scenes: at most 0.034 px / 1.1e-5 m / 2.5e-3
