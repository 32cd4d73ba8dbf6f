"""Evaluation of a global reconstruction against ground truth: PA-MPJPE (all / visible / occluded frames), G-MPJPE, G-MPVE, ACCEL and
the per-frame sample metric -- the reference's `global_recon/utils/evaluator.py` with the same class, method and metric names
(SURVEY.md 8f rank 1).

What runs where: everything per frame runs on the device through libglamr_hip.so -- the four SMPL evaluations per person (ground truth
and estimate, each in world coordinates and with the trajectory re-aligned to its heading every `align_freq` frames,
evaluator.py:202-327) on the HIP skinning kernel with vertices, the 17-joint H36M regression from the vertices
(glamr_eval_regress_joints), the per-chunk heading alignment (glamr_eval_heading_align: `convert_traj_world2heading`,
traj_pred/utils/traj_utils.py:97-107) and the Procrustes alignment (glamr_eval_procrustes: one 3x3 SVD per frame,
lib/utils/torch_transform.py:282-345).  The host keeps the dictionaries and the final reductions of a few numbers per frame
(means over frames, the min / mean over seeds).  The numpy twins (`convert_traj_world2heading`, `batch_compute_similarity_transform`)
are kept as cross-checks for the tests.
"""
from collections import defaultdict

import numpy as np
import torch

from glamr_amd import _lib
from glamr_amd.lib.models.smpl import SMPL, SMPL_MODEL_DIR
from glamr_amd.lib.utils import np_transform as nt

# lib/models/smpl.py:23-29
H36M_TO_J17 = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10, 0, 7, 9]
H36M_TO_J15 = [H36M_TO_J17[14]] + H36M_TO_J17[:14]
JOINT_REGRESSOR_H36M = 'data/J_regressor_h36m.npy'
BASE_ORIENT = np.array([0.5, 0.5, 0.5, 0.5], np.float32)


class AverageMeter(object):
    """lib/utils/tools.py:9-33"""

    def __init__(self, avg=None, count=1):
        self.reset()
        if avg is not None:
            self.val = avg
            self.avg = avg
            self.count = count
            self.sum = avg * count

    def __repr__(self):
        return '%.4f' % self.avg

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        if n > 0:
            self.val = val
            self.sum += val * n
            self.count += n
            self.avg = self.sum / self.count


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _select(pose_dict, mode, *arrays):
    if mode == 'vis':
        return [a[pose_dict['vis_frames']] for a in arrays]
    if mode == 'invis':
        return [a[pose_dict['invis_frames']] for a in arrays]
    return list(arrays)


def _mean_joint_error(pairs, mode, data, est_key, gt_key):
    """Shared body of compute_MPJPE / compute_MPVE / compute_PAMPJPE (evaluator.py:15-38, 96-118, 41-66): per-frame mean point distance
    in millimetres, summed over frames and persons, divided by the number of frames."""
    num_data, total = 0, 0.0
    for idx, pose_dict in data['person_data'].items():
        est, gt = _select(pose_dict, mode, pose_dict[est_key], data['gt'][idx][gt_key])
        if gt.shape[0] == 0:
            continue
        dist = np.linalg.norm(est - gt, axis=2)
        total += float(dist.mean(axis=1).sum()) * 1000
        num_data += est.shape[0]
    return (total / num_data if num_data > 0 else 0.0), {'num_data': num_data}


def compute_MPJPE(data, mode='all', aligned=False):
    key = 'aligned_eval_joints_world' if aligned else 'eval_joints_world'
    return _mean_joint_error(None, mode, data, key, key)


def compute_MPVE(data, mode='all', aligned=False):
    key = 'aligned_eval_verts_world' if aligned else 'eval_verts_world'
    return _mean_joint_error(None, mode, data, key, key)


def compute_PAMPJPE(data, mode='all'):
    return _mean_joint_error(None, mode, data, 'eval_joints_world_PA', 'eval_joints_world')


def compute_PAMPJPE_seq(data, mode='all'):
    """evaluator.py:69-93: the per-frame values (concatenated over persons) instead of their mean."""
    num_data, vals = 0, []
    for idx, pose_dict in data['person_data'].items():
        est, gt = _select(pose_dict, mode, pose_dict['eval_joints_world_PA'], data['gt'][idx]['eval_joints_world'])
        if gt.shape[0] == 0:
            vals.append(np.zeros((0,), np.float32))
            continue
        vals.append(np.linalg.norm(est - gt, axis=2).mean(axis=1) * 1000)
        num_data += est.shape[0]
    return np.concatenate(vals), {'num_data': num_data}


def compute_PAMPJPE_all(data): return compute_PAMPJPE(data, 'all')
def compute_PAMPJPE_vis(data): return compute_PAMPJPE(data, 'vis')
def compute_PAMPJPE_invis(data): return compute_PAMPJPE(data, 'invis')
def compute_sample_PAMPJPE_all(data): return compute_PAMPJPE_seq(data, 'all')
def compute_sample_PAMPJPE_vis(data): return compute_PAMPJPE_seq(data, 'vis')
def compute_sample_PAMPJPE_invis(data): return compute_PAMPJPE_seq(data, 'invis')
def compute_Global_MPJPE(data): return compute_MPJPE(data, 'all', aligned=True)
def compute_Global_MPVE(data): return compute_MPVE(data, 'all', aligned=True)


def compute_accel_error(data):
    """evaluator.py:145-160: second finite difference of the root-relative joints."""
    num_data, total = 0, 0.0
    for idx, pose_dict in data['person_data'].items():
        j, g = pose_dict['eval_joints_world'], data['gt'][idx]['eval_joints_world']
        diff = (j[:-2] - 2 * j[1:-1] + j[2:]) - (g[:-2] - 2 * g[1:-1] + g[2:])
        total += float(np.linalg.norm(diff, axis=2).mean(axis=1).sum()) * 1000
        num_data += diff.shape[0]
    return total / num_data, {'num_data': num_data}


def quat_apply(q, v):
    """lib/utils/torch_transform.py:39-45"""
    xyz = q[..., 1:]
    t = np.cross(xyz, v) * 2
    return v + q[..., :1] * t + np.cross(xyz, t)


def convert_traj_world2heading(orient_q, trans, apply_base_orient_after=False):
    """traj_pred/utils/traj_utils.py:97-107: rotate the trajectory about z so that its first frame heads along +x, and move the first
    frame's xy to the origin."""
    base = np.broadcast_to(BASE_ORIENT, orient_q.shape)
    nobase = nt.quat_mul(orient_q, nt.quat_conj(base))
    inv_heading = np.broadcast_to(nt.quat_conj(nt.heading_quat_of(nobase[0])), nobase.shape)
    orient_heading = nt.quat_mul(inv_heading, nobase)
    local = trans.copy()
    local[..., :2] -= trans[0, ..., :2]
    trans_heading = quat_apply(inv_heading, local)
    if apply_base_orient_after:
        orient_heading = nt.quat_mul(orient_heading, base)
    return orient_heading.astype(np.float32), trans_heading.astype(np.float32)


def batch_compute_similarity_transform(S1, S2):
    """Procrustes alignment of every frame of S1 (N, J, 3) onto S2: lib/utils/torch_transform.py:282-345 (from VIBE)."""
    A, B = np.swapaxes(S1, 1, 2).astype(np.float64), np.swapaxes(S2, 1, 2).astype(np.float64)       # (N, 3, J)
    mu1, mu2 = A.mean(axis=2, keepdims=True), B.mean(axis=2, keepdims=True)
    X1, X2 = A - mu1, B - mu2
    var1 = (X1 ** 2).sum(axis=(1, 2))
    K = X1 @ np.swapaxes(X2, 1, 2)
    U, _, Vt = np.linalg.svd(K)
    V = np.swapaxes(Vt, 1, 2)
    Z = np.tile(np.eye(3)[None], (K.shape[0], 1, 1))
    Z[:, -1, -1] *= np.sign(np.linalg.det(U @ Vt))
    R = V @ (Z @ np.swapaxes(U, 1, 2))
    scale = np.trace(R @ K, axis1=1, axis2=2) / var1
    t = mu2 - scale[:, None, None] * (R @ mu1)
    return np.swapaxes(scale[:, None, None] * (R @ A) + t, 1, 2).astype(np.float32)


class Evaluator:
    """Same constructor and methods as the reference's (evaluator.py:163-391).  `device` must be a HIP device; `smpl` / `j_regressor_h36m`
    may be handed in (tests, non-standard working directories), otherwise they are read from the reference's relative paths."""

    def __init__(self, algo='', dataset='', device=None, log_file='nofile', align_freq=250, compute_sample=True, smpl=None,
                 j_regressor_h36m=None, log=None):
        self.algo, self.dataset = algo, dataset
        self.device = device if device is not None else torch.device('cuda', 0)
        self.align_freq = align_freq
        self.compute_sample = compute_sample
        self.log = log
        self.smpl = smpl if smpl is not None else SMPL(SMPL_MODEL_DIR, pose_type='body26fk', create_transl=False).to(self.device)
        J = j_regressor_h36m if j_regressor_h36m is not None else np.load(JOINT_REGRESSOR_H36M)
        self.J_regressor = torch.as_tensor(np.asarray(J), dtype=torch.float32, device=self.device)
        self.metrics_func = {'PA-MPJPE': compute_PAMPJPE_all, 'PA-MPJPE-vis': compute_PAMPJPE_vis, 'PA-MPJPE-invis': compute_PAMPJPE_invis,
                             'G-MPJPE': compute_Global_MPJPE, 'G-MPVE': compute_Global_MPVE, 'ACCEL': compute_accel_error}
        self.sample_metrics_func = {'sample_PA-MPJPE-invis': compute_sample_PAMPJPE_invis}
        if self.compute_sample:
            self.metrics_func.update(self.sample_metrics_func)
        self.metrics_name = list(self.metrics_func.keys())
        self.seed_min_metrics = ['PA-MPJPE-invis']
        self.reset()

    def reset(self):
        self.metrics_dict_collection = dict()
        self.acc_metrics_dict = {'metrics': defaultdict(AverageMeter)}

    # -- device part ----------------------------------------------------------------------------------------------------------------
    def _smpl_eval(self, orient, body_pose, betas, trans, scale=None):
        """One skinning call with vertices + the H36M joints regressed from them.  Returns (verts (T,V,3), joint_15 (T,15,3)) as numpy."""
        dev = self.device
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
        with torch.no_grad():
            out = self.smpl(global_orient=t(orient), body_pose=t(body_pose), betas=t(betas), root_trans=t(trans),
                            root_scale=None if scale is None else t(scale), return_full_pose=True)
            verts = out.vertices.contiguous()
            B, V = verts.shape[:2]
            j17 = torch.empty((B, self.J_regressor.shape[0], 3), dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().glamr_eval_regress_joints(B, V, self.J_regressor.shape[0], _lib.ptr(verts), _lib.ptr(self.J_regressor), _lib.ptr(j17),
                                                            _lib.current_stream()))
            return verts.cpu().numpy(), j17[:, H36M_TO_J15].cpu().numpy(), out.joints.cpu().numpy()

    # -- alignment (device kernels) + host bookkeeping ------------------------------------------------------------------------------------------------------------------
    def get_aligned_orient_trans(self, pose_dict):
        """evaluator.py:202-216: heading alignment in chunks of `align_freq` frames that overlap by one frame."""
        dev = self.device
        t = lambda a: torch.as_tensor(np.ascontiguousarray(_np(a), dtype=np.float32), device=dev)
        orient, trans = t(pose_dict['smpl_orient_world']), t(pose_dict['root_trans_world'])
        n = orient.shape[0]
        o_aa, o_tr, o_q = (torch.empty((n, k), dtype=torch.float32, device=dev) for k in (3, 3, 4))
        _lib.check(_lib.lib().glamr_eval_heading_align(n, int(self.align_freq), _lib.ptr(orient), _lib.ptr(trans), _lib.ptr(o_aa), _lib.ptr(o_tr), _lib.ptr(o_q),
                                                       _lib.current_stream()))
        pose_dict['aligned_orient_q'] = o_q.cpu().numpy()
        pose_dict['aligned_orient'] = o_aa.cpu().numpy()
        pose_dict['aligned_trans'] = o_tr.cpu().numpy()

    def procrustes(self, S1, S2):
        """batch_compute_similarity_transform_torch (lib/utils/torch_transform.py:282-345) on the device: (N, J, 3) onto (N, J, 3)."""
        dev = self.device
        a = torch.as_tensor(np.ascontiguousarray(S1, dtype=np.float32), device=dev)
        b = torch.as_tensor(np.ascontiguousarray(S2, dtype=np.float32), device=dev)
        out = torch.empty_like(a)
        _lib.check(_lib.lib().glamr_eval_procrustes(a.shape[0], a.shape[1], _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.current_stream()))
        return out.cpu().numpy()

    def prepare_seq(self, data):
        use_keys = ['pose', 'pose_cam', 'root_trans', 'root_trans_cam', 'smpl_orient_cam', 'smpl_orient_world', 'smpl_pose', 'smpl_beta',
                    'root_trans_cam', 'root_trans_world', 'scale', 'vis_frames', 'invis_frames', 'visible', 'j3d_h36m', 'kp']
        exclude_keys = ['smpl_pose_rotmat']
        for idx, pose_dict in data['person_data'].items():          # evaluator.py:224-236: keep the frames the person exists in
            if 'exist_frames' in pose_dict:
                ex = _np(pose_dict['exist_frames']).astype(bool)
                for d in (pose_dict, data['gt'][idx]):
                    for key in list(d.keys()):
                        if not any(x in key for x in use_keys) or key in exclude_keys or d[key] is None:
                            continue
                        d[key] = _np(d[key])[ex]
        # ground truth (:238-289)
        for idx, gt in data['gt'].items():
            if 'pose' not in gt:
                continue
            visible = _np(data['person_data'][idx]['visible_orig'])
            gt['vis_frames'], gt['invis_frames'] = visible == 1, visible == 0
            pose = _np(gt['pose']).astype(np.float32)
            orient, trans = pose[:, :3], _np(gt['root_trans']).astype(np.float32)
            if self.dataset == '3DPW':                              # y-up recordings -> z-up (:251-255)
                q = np.broadcast_to(nt.aa_to_quat(np.array([np.pi * 0.5, 0, 0], np.float32)), (pose.shape[0], 4))
                orient = nt.quat_to_aa(nt.quat_mul(q, nt.aa_to_quat(orient))).astype(np.float32)
                trans = quat_apply(q, trans).astype(np.float32)
            gt['smpl_orient_world'], gt['root_trans_world'] = orient, trans
            betas = np.repeat(_np(gt['shape']).astype(np.float32).reshape(1, -1), pose.shape[0], axis=0)
            verts, j15, joints = self._smpl_eval(orient, pose[:, 3:], betas, trans)
            gt['smpl_verts_world'], gt['smpl_joints_world'] = verts, joints
            pelvis = (j15[:, [3]] + j15[:, [4]]) * 0.5
            gt['eval_joints_world'], gt['eval_verts_world'] = j15[:, 1:] - pelvis, verts - pelvis
            gt['smpl_pose'] = pose[:, 3:]
            self.get_aligned_orient_trans(gt)
            verts, j15, _ = self._smpl_eval(gt['aligned_orient'], pose[:, 3:], betas, gt['aligned_trans'])
            gt['aligned_eval_joints_world'], gt['aligned_eval_verts_world'] = j15[:, 1:], verts
        # estimate (:291-327)
        for idx, pd in data['person_data'].items():
            visible = _np(pd['visible_orig'])
            pd['vis_frames'], pd['invis_frames'] = visible == 1, visible == 0
            scale = None if pd.get('scale') is None else _np(pd['scale'])
            args = (_np(pd['smpl_pose']), _np(pd['smpl_beta']))
            verts, j15, joints = self._smpl_eval(_np(pd['smpl_orient_world']), args[0], args[1], _np(pd['root_trans_world']), scale)
            pd['smpl_verts_world'], pd['smpl_joints_world'] = verts, joints
            pelvis = (j15[:, [3]] + j15[:, [4]]) * 0.5
            pd['eval_joints_world'], pd['eval_verts_world'] = j15[:, 1:] - pelvis, verts - pelvis
            self.get_aligned_orient_trans(pd)
            pd['eval_joints_world_PA'] = self.procrustes(pd['eval_joints_world'], data['gt'][idx]['eval_joints_world'])
            verts, j15, _ = self._smpl_eval(pd['aligned_orient'], args[0], args[1], pd['aligned_trans'], scale)
            pd['aligned_eval_joints_world'], pd['aligned_eval_verts_world'] = j15[:, 1:], verts

    def compute_sequence_metrics(self, data, name=None, accumulate=True):
        """One sequence: every configured metric as an AverageMeter weighted by the number of samples behind it (evaluator.py:329-339)."""
        self.prepare_seq(data)
        data['log'], data['name'] = self.log, name
        meters = {}
        for metric_name, metric in self.metrics_func.items():
            value, info = metric(data)
            meters[metric_name] = AverageMeter(value, info['num_data'])
        result = defaultdict(dict, seq_len=data['seq_len'], metrics=meters)
        if accumulate:
            self.update_accumulated_metrics(result, name)
        return result

    def update_accumulated_metrics(self, metrics_dict, name=None):
        """Folds a sequence's meters into the running totals (evaluator.py:341-346); named sequences are also kept individually."""
        if name is not None:
            self.metrics_dict_collection[name] = metrics_dict
        totals, seq = self.acc_metrics_dict['metrics'], metrics_dict['metrics']
        for key in self.metrics_name:
            totals[key].update(seq[key].avg, seq[key].count)
        return self.acc_metrics_dict

    def metrics_from_multiple_seeds(self, metrics_dict_arr):
        """evaluator.py:352-379: best-of-seeds for the sample metric and PA-MPJPE-invis, mean over seeds otherwise."""
        first = metrics_dict_arr[0]
        combined = {}
        for key in self.metrics_name:
            count = first['metrics'][key].count
            per_seed = [m['metrics'][key].avg for m in metrics_dict_arr]
            if 'sample' in key or 'mean' in key:          # per-frame arrays: reduce over seeds first, then over frames
                stacked = np.stack(per_seed)
                reduced = stacked.min(axis=0) if 'sample' in key else stacked.mean(axis=0)
                value = reduced.mean() if count != 0 else 0
            else:
                per_seed = np.asarray(per_seed)
                value = per_seed.min() if key in self.seed_min_metrics else per_seed.mean()
            combined[key] = AverageMeter(value, count)
        return defaultdict(dict, seq_len=first['seq_len'], metrics=combined)

    def print_metrics(self, metrics_dict=None, fmt='.3f', prefix='', print_accum=True):
        """One log line `<prefix><algo> --- name: running average (last value) ...` (evaluator.py:381-391); array-valued meters are skipped."""
        meters = (self.acc_metrics_dict if metrics_dict is None else metrics_dict)['metrics']
        number = '{:' + fmt + '}'
        cell = ('{}: ' + number + ' (' + number + ')') if print_accum else ('{}: ' + number)
        parts = [cell.format(key, m.avg, m.val) for key, m in meters.items() if not isinstance(m.avg, np.ndarray)]
        line = '%s%s --- %s' % (prefix, self.algo, ' '.join(parts))
        if 'sample_PA-MPJPE-invis' not in meters:
            line += ' sample_PA-MPJPE-invis: None (need multiple seeds)'
        (self.log.info if self.log is not None else print)(line)
        return line
