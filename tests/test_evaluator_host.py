"""Host-side pieces of the evaluator (no GPU): Procrustes alignment, heading alignment, metric bookkeeping."""
import numpy as np

from glamr_amd.global_recon.utils import evaluator as ev
from glamr_amd.lib.utils import np_transform as nt


def test_procrustes_recovers_a_similarity_transform():
    rng = np.random.default_rng(0)
    X = rng.normal(size=(5, 14, 3)).astype(np.float32)
    q = nt.aa_to_quat(rng.normal(size=(5, 3)).astype(np.float32))
    R = nt.quat_to_rotmat(q)
    Y = 1.7 * np.einsum('nab,njb->nja', R, X) + rng.normal(size=(5, 1, 3)).astype(np.float32)
    assert np.abs(ev.batch_compute_similarity_transform(X, Y) - Y).max() < 1e-4


def test_heading_alignment_starts_at_the_origin_facing_x():
    rng = np.random.default_rng(1)
    T = 40
    heading = np.linspace(0.7, 1.9, T).astype(np.float32)
    q = nt.quat_mul(nt.heading_to_quat(heading), np.broadcast_to(ev.BASE_ORIENT, (T, 4)))
    trans = np.stack([np.cumsum(np.cos(heading)) * 0.02 + 3.0, np.cumsum(np.sin(heading)) * 0.02 - 1.0, np.full(T, 0.9)], 1).astype(np.float32)
    qa, ta = ev.convert_traj_world2heading(q, trans, apply_base_orient_after=True)
    assert np.abs(ta[0, :2]).max() < 1e-6 and abs(ta[0, 2] - 0.9) < 1e-6
    nobase = nt.quat_mul(qa, nt.quat_conj(np.broadcast_to(ev.BASE_ORIENT, (T, 4))))
    assert abs(float(nt.heading_of(nobase[:1])[0])) < 1e-5                       # first frame heads along +x
    step = ta[1, :2] - ta[0, :2]
    assert step[0] > 0 and abs(step[1]) < 0.2 * step[0]


def test_metric_bookkeeping():
    vis = np.array([True, True, False, True])
    d = {'person_data': {0: {'eval_joints_world': np.zeros((4, 14, 3), np.float32), 'eval_joints_world_PA': np.zeros((4, 14, 3), np.float32),
                             'aligned_eval_joints_world': np.zeros((4, 14, 3), np.float32), 'vis_frames': vis, 'invis_frames': ~vis}},
         'gt': {0: {'eval_joints_world': np.full((4, 14, 3), 0.001, np.float32), 'aligned_eval_joints_world': np.full((4, 14, 3), 0.002, np.float32)}}}
    v, info = ev.compute_PAMPJPE(d, 'invis')
    assert info['num_data'] == 1 and abs(v - 1000 * 0.001 * np.sqrt(3)) < 1e-3
    v, info = ev.compute_Global_MPJPE(d)
    assert info['num_data'] == 4 and abs(v - 1000 * 0.002 * np.sqrt(3)) < 1e-3
    vals, info = ev.compute_sample_PAMPJPE_invis(d)
    assert vals.shape == (1,)
    m = ev.AverageMeter()
    m.update(2.0, 3); m.update(4.0, 1)
    assert abs(m.avg - 2.5) < 1e-12 and m.count == 4
