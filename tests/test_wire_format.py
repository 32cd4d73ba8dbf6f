"""Input wire format: pose.pkl of pose_est/hybrik_demo/demo.py:317-354."""
import copy
import os
import pickle

import numpy as np
import pytest

from glamr_amd.utils import synth, wire


@pytest.fixture(scope='module')
def est():
    return synth.make_in_dict(seed=5, num_frames=90, num_persons=2, smpl_model=synth.make_smpl_model())['est']


def test_synthetic_sequences_follow_the_contract(est):
    out = wire.normalise_est(est)
    for pid, d in out.items():
        n = int(d['bboxes_dict']['exist'].sum())
        assert d['smpl_pose_quat_wroot'].shape == (n, 54, 4) and d['smpl_pose_quat_wroot'].dtype == np.float32
        assert d['kp_2d'].shape[0] == n and d['cam_K'].shape == (n, 3, 3)
        assert np.array_equal(d['frames'], np.flatnonzero(d['bboxes_dict']['exist']))
        assert d['frame2ind'][int(d['frames'][3])] == 3
        assert np.array_equal(d['smpl_beta'], est[pid]['smpl_beta'])


def test_flat_rotation_layout_is_accepted(est):
    e = copy.deepcopy(est)
    e[0]['smpl_pose_quat_wroot'] = e[0]['smpl_pose_quat_wroot'].reshape(len(e[0]['smpl_beta']), 216)
    assert wire.normalise_est(e)[0]['smpl_pose_quat_wroot'].shape[1:] == (54, 4)


@pytest.mark.parametrize('mutate, msg', [
    (lambda e: e[0].pop('kp_2d'), 'missing key'),
    (lambda e: e[0].__setitem__('smpl_beta', e[0]['smpl_beta'][:-1]), 'smpl_beta must have shape'),
    (lambda e: e[0].__setitem__('kp_2d', e[0]['kp_2d'][:, :20]), 'kp_2d must have shape'),
    (lambda e: e[0]['root_trans'].__setitem__((3, 1), np.nan), 'non-finite'),
    (lambda e: e[0].__setitem__('smpl_pose_quat_wroot', e[0]['smpl_pose_quat_wroot'] * 1.5), 'rotation matrices'),
    (lambda e: e[1]['bboxes_dict'].__setitem__('exist', e[1]['bboxes_dict']['exist'][:-5]), 'hold 24 rotation matrices|share the video length'),
    (lambda e: e[0]['bboxes_dict'].__setitem__('exist', e[0]['bboxes_dict']['exist'] * 2), '0/1'),
])
def test_malformed_inputs_fail_loudly(est, mutate, msg):
    e = copy.deepcopy(est)
    mutate(e)
    with pytest.raises(wire.WireFormatError, match=msg):
        wire.normalise_est(e)


def test_pose_pkl_round_trip(est, tmp_path):
    d = tmp_path / 'walk_01' / 'pose_est'
    os.makedirs(d)
    with open(d / 'pose.pkl', 'wb') as f:
        pickle.dump(est, f)
    in_dict = wire.load_pose_pkl(str(d / 'pose.pkl'), seq_name='walk_01')
    assert in_dict['seq_name'] == 'walk_01' and in_dict['gt'] == {} and set(in_dict['est']) == {0, 1}


def test_ground_truth_pickle_contract():
    in_dict = synth.make_in_dict(seed=6, num_frames=80, num_persons=2, smpl_model=synth.make_smpl_model(), with_gt=True)
    gt, meta = wire.normalise_gt({'person_data': in_dict['gt'], 'meta': {'cam_K': np.eye(3)}}, num_frames=80)
    assert set(gt) == {0, 1} and gt[0]['pose'].shape == (80, 72) and gt[0]['shape'].shape == (10,) and 'cam_K' in meta
    bad = {'person_data': {0: dict(in_dict['gt'][0], pose=in_dict['gt'][0]['pose'][:, :69])}}
    with pytest.raises(wire.WireFormatError, match='72'):
        wire.normalise_gt(bad)
    with pytest.raises(wire.WireFormatError, match='covers 80 frames'):
        wire.normalise_gt({'person_data': in_dict['gt']}, num_frames=81)
