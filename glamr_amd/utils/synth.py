"""Synthetic stand-ins for the assets GLAMR needs but that cannot be shipped or downloaded here:

* an SMPL-shaped body model file (`SMPL_NEUTRAL.pkl`, V=6890, 24 joints, public key layout) plus
  `J_regressor_extra.npy` (9 x V) -- the reference reads them at lib/models/smpl.py:285 / smplx.SMPL.__init__;
* Lightning-style checkpoints (`{'state_dict': ...}`) for the motion infiller and the trajectory predictor, laid out
  where `find_last_version` / `get_checkpoint_path` look (lib/utils/tools.py:41-45,94-104);
* HybrIK-format pose dictionaries, the wire format of `GlobalReconOptimizer.optimize` (pose_est/hybrik_demo/demo.py:317-354),
  shaped like AMASS clips (30 fps, `trans 3 / pose 72 / shape 10`, motion_infiller/data/amass_dataset.py:65-67).

Everything is drawn from numpy `default_rng(seed)` streams so that the build container and the GPU box regenerate
bit-identical assets.  This is data generation only -- no reference arithmetic lives here.
"""
import os
import pickle
import numpy as np

NUM_VERTS = 6890
NUM_FACES = 13776
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], dtype=np.int64)

# rough rest-pose skeleton (metres, y up, facing +z) so that the synthetic body has human proportions
_REST_JOINTS = np.array([
    [0.00, -0.24, 0.02], [0.07, -0.33, 0.01], [-0.07, -0.33, 0.01], [0.00, -0.12, -0.01],
    [0.10, -0.71, 0.01], [-0.10, -0.71, 0.01], [0.00, 0.02, 0.00], [0.09, -1.11, -0.03],
    [-0.09, -1.11, -0.03], [0.00, 0.07, 0.02], [0.11, -1.17, 0.09], [-0.11, -1.17, 0.09],
    [0.00, 0.28, -0.02], [0.08, 0.19, -0.01], [-0.08, 0.19, -0.01], [0.00, 0.37, 0.03],
    [0.17, 0.22, -0.02], [-0.17, 0.22, -0.02], [0.43, 0.21, -0.04], [-0.43, 0.21, -0.04],
    [0.68, 0.22, -0.04], [-0.68, 0.22, -0.04], [0.77, 0.21, -0.05], [-0.77, 0.21, -0.05]], dtype=np.float64)

# vertex ids smplx picks as extra joints on the SMPL topology, with the body part each should sit on: (vertex id, joint, offset)
_PICKED = [(332, 15, (0.0, 0.02, 0.11)), (6260, 15, (-0.03, 0.05, 0.09)), (2800, 15, (0.03, 0.05, 0.09)),
           (4071, 15, (-0.07, 0.03, 0.0)), (583, 15, (0.07, 0.03, 0.0)),
           (3216, 10, (0.01, -0.03, 0.08)), (3226, 10, (0.05, -0.03, 0.05)), (3387, 7, (0.0, -0.08, -0.05)),
           (6617, 11, (-0.01, -0.03, 0.08)), (6624, 11, (-0.05, -0.03, 0.05)), (6787, 8, (0.0, -0.08, -0.05)),
           (2746, 22, (0.03, 0.0, 0.04)), (2319, 22, (0.09, 0.01, 0.02)), (2445, 22, (0.10, 0.0, 0.0)),
           (2556, 22, (0.09, -0.01, -0.02)), (2673, 22, (0.07, -0.02, -0.03)),
           (6191, 23, (-0.03, 0.0, 0.04)), (5782, 23, (-0.09, 0.01, 0.02)), (5905, 23, (-0.10, 0.0, 0.0)),
           (6016, 23, (-0.09, -0.01, -0.02)), (6133, 23, (-0.07, -0.02, -0.03))]

# which chain joint each of the 9 "extra" regressed joints (indices 45..53 of JOINT_MAP, lib/models/smpl.py:35-57) hugs
_EXTRA_ANCHOR = [2, 1, 12, 15, 0, 9, 3, 15, 15]


def make_smpl_model(seed=1234):
    """Returns a dict with the public SMPL pickle keys (float32 / integer numpy arrays) and 'J_regressor_extra'."""
    rng = np.random.default_rng(seed)
    V = NUM_VERTS
    seg_w = np.array([3, 2, 2, 3, 3, 3, 3, 2, 2, 4, 1, 1, 1, 1.5, 1.5, 3, 2, 2, 2, 2, 1, 1, 0.7, 0.7])
    primary = rng.choice(24, size=V, p=seg_w / seg_w.sum())
    sigma = np.array([.07, .06, .06, .08, .05, .05, .09, .04, .04, .10, .03, .03, .04, .05, .05, .07,
                      .05, .05, .04, .04, .03, .03, .025, .025])
    v = _REST_JOINTS[primary] + rng.normal(size=(V, 3)) * sigma[primary][:, None]
    # stretch limb vertices towards the child joint so segments look like bones, not blobs
    child = {p: c for c, p in enumerate(SMPL_PARENTS) if p >= 0}
    child.update({0: 3, 9: 12, 12: 15})
    for j, c in child.items():
        idx = np.where(primary == j)[0]
        v[idx] += rng.uniform(0, 0.8, size=(len(idx), 1)) * (_REST_JOINTS[c] - _REST_JOINTS[j])
    for vid, j, off in _PICKED:
        primary[vid] = j
        v[vid] = _REST_JOINTS[j] + np.asarray(off)

    def regressor(anchor_sets):
        R = np.zeros((len(anchor_sets), V))
        for r, joints in enumerate(anchor_sets):
            m = np.isin(primary, joints)
            w = rng.random(V) ** 8 * m
            R[r] = w / w.sum()
        return R

    J_regressor = regressor([[j] for j in range(24)])
    J_extra = regressor([[j] for j in _EXTRA_ANCHOR])
    W = rng.random((V, 24)) ** 8 * 0.02
    W[np.arange(V), primary] += 0.7
    par = SMPL_PARENTS[primary]
    par[par < 0] = 3
    W[np.arange(V), par] += 0.3
    W /= W.sum(1, keepdims=True)
    kintree = np.stack([SMPL_PARENTS.copy(), np.arange(24)]).astype(np.int64)
    kintree[0, 0] = 4294967295
    faces = np.stack([np.arange(NUM_FACES) % V, (np.arange(NUM_FACES) * 7 + 1) % V, (np.arange(NUM_FACES) * 13 + 2) % V], 1)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return {
        'v_template': f32(v),
        'shapedirs': f32(rng.normal(size=(V, 3, 10)) * 0.01),
        'posedirs': f32(rng.normal(size=(V, 3, 207)) * 0.002),
        'J_regressor': f32(J_regressor),
        'weights': f32(W),
        'kintree_table': kintree.astype(np.uint32),
        'f': faces.astype(np.uint32),
        'J_regressor_extra': f32(J_extra),
    }


def make_h36m_regressor(md, seed=4321):
    """A synthetic stand-in for `data/J_regressor_h36m.npy` (17 x V; lib/models/smpl.py:29): every H36M joint regressed from the
    vertices of one body part.  Drawn from its own generator so the SMPL model above does not change."""
    rng = np.random.default_rng(seed)
    V = md['v_template'].shape[0]
    primary = np.argmax(md['weights'], axis=1)
    # H36M order: pelvis, rhip, rknee, rankle, lhip, lknee, lankle, spine, thorax, neck/nose, head, lshoulder, lelbow, lwrist, rshoulder, relbow, rwrist
    anchors = [0, 2, 5, 8, 1, 4, 7, 6, 9, 12, 15, 16, 18, 20, 17, 19, 21]
    R = np.zeros((17, V))
    for r, j in enumerate(anchors):
        w = rng.random(V) ** 8 * (primary == j)
        R[r] = w / w.sum()
    return R.astype(np.float32)


def write_smpl_assets(root, seed=1234):
    """Writes `<root>/data/body_models/smpl/SMPL_NEUTRAL.pkl` and `<root>/data/J_regressor_extra.npy`
    (the relative paths hard-wired at lib/models/smpl.py:28-31).  Returns the model dict."""
    md = make_smpl_model(seed)
    mdir = os.path.join(root, 'data', 'body_models', 'smpl')
    os.makedirs(mdir, exist_ok=True)
    with open(os.path.join(mdir, 'SMPL_NEUTRAL.pkl'), 'wb') as f:
        pickle.dump({k: v for k, v in md.items() if k != 'J_regressor_extra' and not k.startswith('_')}, f, protocol=2)
    np.save(os.path.join(root, 'data', 'J_regressor_extra.npy'), md['J_regressor_extra'])
    np.save(os.path.join(root, 'data', 'J_regressor_h36m.npy'), make_h36m_regressor(md))
    return md


# ---------------------------------------------------------------------------------------------------------------------
# network weights
# ---------------------------------------------------------------------------------------------------------------------

def make_state_dict(layout, seed, final_bias=None):
    """layout: ordered list of (key, shape).  Linear/attention weights ~ U(+-1/sqrt(fan_in)), biases ~ U(+-0.05),
    LayerNorm gains ~ 1 + 0.1 N, learned tokens ~ 0.01 N.  `final_bias`: {key: vector} overrides (used to centre the
    last layer on a plausible output so random networks still emit body-like motion).  Returns {key: float32 ndarray}."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in layout:
        shape = tuple(shape)
        if key.endswith('token'):
            a = rng.normal(size=shape) * 0.01
        elif '.norm' in key and key.endswith('weight'):
            a = 1.0 + 0.1 * rng.normal(size=shape)
        elif len(shape) >= 2:
            bound = 1.0 / np.sqrt(shape[-1])
            a = rng.uniform(-bound, bound, size=shape)
            if key.endswith('out_fc.weight'):
                a *= 0.2
        else:
            a = rng.uniform(-0.05, 0.05, size=shape)
        sd[key] = np.ascontiguousarray(a, dtype=np.float32)
    for key, val in (final_bias or {}).items():
        sd[key] = np.ascontiguousarray(val, dtype=np.float32)
    return sd


TRAJ_OUT_BIAS = np.array([0.0, 0.03, 0.92, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0], dtype=np.float32)


def write_checkpoints(root, infiller_layout, trajpred_layout, seed=1):
    """Writes the two `.ckpt` files under `<root>/results/...` where the reference loader globs for them
    (motion_infiller/models/motion_traj_joint_model.py:37-44,58-65)."""
    import torch
    out = {}
    specs = [('motion_filler/motion_infiller_demo', infiller_layout, seed, None),
             ('traj_pred/traj_pred_demo', trajpred_layout, seed + 1, {'data_decoder.out_fc.bias': TRAJ_OUT_BIAS})]
    for sub, layout, s, fb in specs:
        sd = make_state_dict(layout, s, fb)
        d = os.path.join(root, 'results', sub, 'version_0', 'checkpoints')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, 'model-best-epoch=0000.ckpt')
        torch.save({'state_dict': {k: torch.from_numpy(v.copy()) for k, v in sd.items()}}, path)
        out[sub] = sd
    return out


# ---------------------------------------------------------------------------------------------------------------------
# HybrIK-format synthetic sequences
# ---------------------------------------------------------------------------------------------------------------------

def _rodrigues(r):
    """(...,3) -> (...,3,3), float64."""
    th = np.linalg.norm(r, axis=-1, keepdims=True)
    k = r / np.maximum(th, 1e-12)
    K = np.zeros(r.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def _fk_joints(md, betas, rotmats):
    """Chain joints of the synthetic model.  betas (T,10), rotmats (T,24,3,3) -> (T,24,3) with the root at its rest place."""
    # rest joints are linear in the shape: J = J_regressor (v_template + shapedirs beta); the two regressed factors are cached
    if '_J_template64' not in md:
        Jr = md['J_regressor'].astype(np.float64)
        md['_J_template64'] = Jr @ md['v_template'].astype(np.float64)
        md['_J_shapedirs64'] = np.einsum('jv,vkl->jkl', Jr, md['shapedirs'].astype(np.float64))
    J = md['_J_template64'][None] + np.einsum('jkl,tl->tjk', md['_J_shapedirs64'], betas)
    G = [None] * 24
    pos = np.zeros_like(J)
    for j in range(24):
        p = SMPL_PARENTS[j]
        if p < 0:
            G[j] = rotmats[:, j]
            pos[:, j] = J[:, j]
        else:
            G[j] = G[p] @ rotmats[:, j]
            pos[:, j] = pos[:, p] + np.einsum('tab,tb->ta', G[p], J[:, j] - J[:, p])
    return pos


_BASE_R = np.array([[0., 0., 1.], [1., 0., 0.], [0., 1., 0.]])   # quaternion (0.5,0.5,0.5,0.5): body y-up -> world z-up


def make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=None, dynamic_cam=True, gap=None,
                 kp_noise=1.5, seq_name=None, with_gt=False):
    """One synthetic sequence in the dictionary format `GlobalReconOptimizer.optimize` consumes
    (global_recon/run_demo.py:78-81): {'est': {idx: hybrik_dict}, 'gt': {}, 'gt_meta': {}, 'seq_name': str}.

    World: z up; people walk on the ground plane with slowly turning heading.  Camera: y down / z forward, placed a few
    metres away, panning and swaying when `dynamic_cam`.  `gap=(a,b)`: frames [a,b) of person 0 are undetected
    (`exist` = 0); default [100,160) for T>=200, [40,60) for T>=80.  Person p>0 gets a shifted gap."""
    T = num_frames
    md = smpl_model if smpl_model is not None else make_smpl_model()
    rng = np.random.default_rng(10_000 + seed)
    t = np.arange(T) / 30.0
    if gap is None:
        gap = (100, 160) if T >= 200 else ((40, 60) if T >= 80 else None)

    # camera-to-world trajectory
    cam_pos = np.array([-4.2, 0.0, 1.3])[None] + (np.stack([0.25 * np.sin(0.5 * t + 0.3), 0.5 * np.sin(0.35 * t),
                                                              0.05 * np.sin(0.9 * t)], 1) if dynamic_cam else 0.0)
    pan = (0.15 * np.sin(0.4 * t + rng.uniform(0, 6.28)) if dynamic_cam else np.zeros(T)) + 0.02
    tilt = (0.03 * np.sin(0.7 * t) if dynamic_cam else np.zeros(T)) + 0.04
    R0 = np.array([[0., 0., 1.], [-1., 0., 0.], [0., -1., 0.]])     # columns: right, down, forward in world axes
    Rc2w = _rodrigues(np.stack([0 * pan, 0 * pan, pan], 1)) @ R0[None] @ _rodrigues(np.stack([tilt, 0 * tilt, 0 * tilt], 1))
    K = np.eye(3, dtype=np.float32)
    K[0, 0] = K[1, 1] = 1000.0
    K[0, 2], K[1, 2] = 960.0, 540.0

    est, gt = {}, {}
    for p in range(num_persons):
        prng = np.random.default_rng(20_000 + 97 * seed + p)
        freqs = prng.uniform(0.3, 0.8, size=(3, 69))
        phase = prng.uniform(0, 2 * np.pi, size=(3, 69))
        amp = prng.uniform(0.0, 0.1, size=(3, 69))
        body_pose = sum(amp[k] * np.sin(2 * np.pi * freqs[k] * t[:, None] + phase[k]) for k in range(3))
        betas = np.repeat(0.5 * prng.normal(size=(1, 10)), T, 0)
        heading = 0.3 * np.sin(0.25 * t + prng.uniform(0, 6.28)) + prng.uniform(-0.4, 0.4)
        speed = 0.6 + 0.2 * np.sin(0.5 * t + prng.uniform(0, 6.28))
        xy = np.cumsum(np.stack([np.cos(heading), np.sin(heading)], 1) * speed[:, None] / 30.0, 0)
        xy += np.array([0.5, 1.0 * p - 0.5 * (num_persons - 1)])
        z = 0.92 + 0.02 * np.sin(2 * np.pi * 1.6 * t + prng.uniform(0, 6.28))
        trans_w = np.concatenate([xy, z[:, None]], 1)
        sway = _rodrigues(0.05 * np.stack([np.sin(1.1 * t), np.sin(0.7 * t + 1.0), 0 * t], 1))
        Rw = _rodrigues(np.stack([0 * t, 0 * t, heading], 1)) @ _BASE_R[None] @ sway
        if with_gt:      # the ground truth in the layout the evaluator reads (global_recon/utils/evaluator.py:238-262)
            from glamr_amd.lib.utils.np_transform import rotmat_to_rotvec_nearest
            gt[p] = {'pose': np.concatenate([rotmat_to_rotvec_nearest(Rw.astype(np.float32)), body_pose.astype(np.float32)], 1),
                     'shape': betas[0].astype(np.float32), 'root_trans': trans_w.astype(np.float32)}
        # into the camera frame
        Rc = np.transpose(Rc2w, (0, 2, 1)) @ Rw
        tc = np.einsum('tba,tb->ta', Rc2w, trans_w - cam_pos)
        rot_local = _rodrigues(body_pose.reshape(T, 23, 3))
        rotmats = np.concatenate([Rc[:, None], rot_local], 1)
        joints = _fk_joints(md, betas, rotmats)
        joints = joints - joints[:, :1] + tc[:, None]
        uvw = np.einsum('ab,tjb->tja', K.astype(np.float64), joints)
        kp24 = uvw[..., :2] / uvw[..., 2:]
        kp = np.concatenate([kp24, kp24[:, [15, 22, 23, 10, 11]]], 1) + prng.normal(size=(T, 29, 2)) * kp_noise
        # estimator noise on the 3-D quantities
        rotmats[:, 0] = rotmats[:, 0] @ _rodrigues(0.02 * prng.normal(size=(T, 3)))
        tc_noisy = tc + prng.normal(size=(T, 3)) * np.array([0.01, 0.01, 0.04])
        exist = np.ones(T, dtype=np.float64)
        if gap is not None:
            a, b = gap
            shift = 37 * p
            exist[min(a + shift, T - 2):min(b + shift, T - 1)] = 0.0
        vis = np.where(exist == 1)[0]
        est[p] = {
            'smpl_pose_quat_wroot': rotmats[vis].reshape(len(vis), -1, 4).astype(np.float32),
            'smpl_beta': betas[vis].astype(np.float32),
            'root_trans': tc_noisy[vis].astype(np.float32),
            'kp_2d': kp[vis].astype(np.float32),
            'cam_K': np.repeat(K[None], len(vis), 0),
            'frames': vis,
            'frame2ind': {int(f): i for i, f in enumerate(vis)},
            'bboxes_dict': {'id': p, 'exist': exist, 'start': int(vis[0]), 'end': int(vis[-1]),
                            'num_frames': float(exist.sum()), 'exist_frames': vis},
        }
    return {'est': est, 'gt': gt, 'gt_meta': dict(),
            'seq_name': seq_name or 'synth_s%d_T%d_P%d' % (seed, T, num_persons)}


def inject_orient_jumps(in_dict, idx, events):
    """Root-orientation discontinuities in person `idx`'s HybrIK rotations -- what `filter_pose` (global_recon_model.py:250-262) exists for
    (a pose estimator flipping a person by ~180 degrees for a few frames).  `events`, applied in order, frames are VIDEO frames that must be
    detected:
      ('spike', frame, angle, n)   frames [frame, frame + n) are turned by `angle` about the body's own up axis (right-multiplied): two
                                   discontinuities, at `frame` and at `frame + n`;
      ('step', frame, angle)       the root rotation of `frame` becomes EXACTLY that of `frame - 1` (the previous DETECTED frame's) turned by
                                   `angle` about a fixed oblique axis, and every later frame carries the same left factor along (no second
                                   discontinuity): the quaternion angle between the two frames is `angle` to float32 rounding -- for
                                   thresholds at pi / 3 +- 1e-4."""
    src = in_dict['est'][idx]
    rot = src['smpl_pose_quat_wroot'].reshape(src['smpl_pose_quat_wroot'].shape[0], -1, 3, 3).astype(np.float64)
    f2i = src['frame2ind']
    for ev in events:
        if ev[0] == 'spike':
            _, frame, angle, n = ev
            D = _rodrigues(np.array([[0.0, angle, 0.0]]))[0]
            for f in range(frame, frame + n):
                rot[f2i[f], 0] = rot[f2i[f], 0] @ D
        elif ev[0] == 'step':
            _, frame, angle = ev
            i = f2i[frame]
            axis = np.array([0.36, 0.8, 0.48])
            target = rot[i - 1, 0] @ _rodrigues((axis * angle)[None])[0]
            L = target @ rot[i, 0].T
            rot[i:, 0] = L[None] @ rot[i:, 0]
        else:
            raise KeyError(ev[0])
    src['smpl_pose_quat_wroot'] = rot.reshape(rot.shape[0], -1, 4).astype(np.float32)
    return in_dict


def trim_person(in_dict, idx, first, last):
    """Person `idx` is detected only in frames [first, last): appears late / leaves early (ragged existence inside a sequence).  In place."""
    src = in_dict['est'][idx]
    ex = np.asarray(src['bboxes_dict']['exist']).copy()
    keep_frames = np.flatnonzero(ex)
    keep = (keep_frames >= first) & (keep_frames < last)
    ex[:first] = 0
    ex[last:] = 0
    for k in ('smpl_pose_quat_wroot', 'smpl_beta', 'root_trans', 'kp_2d', 'cam_K'):
        src[k] = src[k][keep]
    frames = np.flatnonzero(ex)
    src['frames'] = frames
    src['frame2ind'] = {int(f): i for i, f in enumerate(frames)}
    src['bboxes_dict'] = dict(src['bboxes_dict'], exist=ex, start=int(frames[0]), end=int(frames[-1]), num_frames=float(ex.sum()), exist_frames=frames)
    return in_dict
