"""Device filter_pose (csrc/init.hip prep_person_kernel; global_recon_model.py:250-271) on the MI355X against the UNMODIFIED reference, on
inputs WITH root-orientation jumps.  The threshold is an acos evaluated in float32 at pi / 3: the cases within 1e-4 /
5e-5 / 2e-5 of it on either side are where a bit-exact visibility mask could flip."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden as mg

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

    def make(**specs):
        cfg = get_config('glamr_dynamic')
        cfg['grecon_model_specs'].update(specs)
        return model_dict['global_recon_model'](cfg, dev, None, smpl=smpl, mt_model=mt)
    return make


KEYS = ('visible', 'visible_orig', 'exist_frames', 'vis_frames', 'invis_frames')


@pytest.mark.parametrize('case', mg.FILTER_CASES, ids=[c[0] for c in mg.FILTER_CASES])
def test_device_filter_pose_on_injected_jumps(make_model, golden, case):
    """init_data on the device (the default path): visible / vis_frames / invis_frames / exist range `array_equal` to the reference's."""
    g = golden('filter_pose')
    name, T, gap, events, attrs = case
    in_dict, seed = mg.filter_inputs(case)
    model = make_model(**attrs)
    data = model.init_data(in_dict, latents=mg.latents_for(in_dict, seed))
    pd = data['person_data'][0]
    for key in KEYS:
        assert np.array_equal(np.asarray(pd[key]), g['%s_%s' % (name, key)]), '%s: %s' % (name, key)
    assert int(pd['fr_start']) == int(g[name + '_fr_start']) and int(pd['fr_end']) == int(g[name + '_fr_end'])
    gone = np.flatnonzero(np.asarray(pd['visible']) != np.asarray(pd['visible_orig']))
    print('%s: frames made invisible %s' % (name, gone.tolist()))
    if 'thr_minus' not in name:
        assert len(gone) > 0                                   # the case did exercise the filter


def test_all_jump_cases_in_one_ragged_batch(make_model, golden):
    """The same sequences as ONE batch (120- and 300-frame sequences padded to 300): a person slot's filter must not see its neighbours."""
    g = golden('filter_pose')
    cases = [c for c in mg.FILTER_CASES if not c[4]]
    ins = [mg.filter_inputs(c) for c in cases]
    model = make_model()
    outs = model.optimize_batch([i for i, _ in ins], [mg.latents_for(i, s) for i, s in ins], max_iters=1)
    for c, out in zip(cases, outs):
        for key in KEYS:
            assert np.array_equal(np.asarray(out['person_data'][0][key]), g['%s_%s' % (c[0], key)]), '%s: %s' % (c[0], key)


def test_keypoint_count_filter_default_minimum_on_the_device(make_model, golden):
    """flag_make_invis_with_keypoint with the reference's default minimum (15 > the 14 joints HybrIK scores): glamr_init_prepare alone (the
    priors cannot run on a sequence without a visible frame -- the reference cannot either) leaves every frame invisible, like the reference's
    filter_pose on the same arrays."""
    import ctypes
    from glamr_amd import _lib
    g = golden('filter_pose')
    in_dict, seed = mg.filter_inputs(mg.FILTER_CASES[0])
    model = make_model(flag_make_invis_with_keypoint=True)
    assert model.make_invis_keypoint_min_num == 15 and model.make_invis_keypoint_min_score == 0.6
    vis = model.prepare_only([in_dict])
    assert np.array_equal(vis[0], g['kp_default_visible'].astype(np.float32))
    model14 = make_model(flag_make_invis_with_keypoint=True, make_invis_keypoint_min_num=14)
    vis14 = model14.prepare_only([in_dict])
    assert np.array_equal(vis14[0], g['isolated_spike_visible'].astype(np.float32))
