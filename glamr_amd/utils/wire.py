"""The input wire format of the global reconstruction: what `pose_est/hybrik_demo/demo.py:317-354` pickles as `pose.pkl` and
`global_recon/run_demo.py:78-81` wraps as `in_dict = {'est': est_dict, 'gt': {}, 'gt_meta': {}, 'seq_name': ...}`.

    est_dict[person_id] = {
        'smpl_pose_quat_wroot': (Tvis, 54, 4)   24 rotation matrices per detection, flattened row-major and regrouped by 4 (demo.py:320)
        'smpl_beta':            (Tvis, 10)
        'root_trans':           (Tvis, 3)       camera coordinates, metres
        'kp_2d':                (Tvis, 29, 2)   HybrIK's 29 keypoints in pixels (the first 24 are the SMPL joints)
        'cam_K':                (Tvis, 3, 3)
        'frames':               (Tvis,)         video frame of every detection
        'frame2ind':            {frame: row}
        'bboxes_dict':          {'exist': (T,) 0/1 per video frame, 'bbox', 'start', 'end', 'num_frames', 'exist_frames', 'id'}
    }

`normalise_est` checks a dictionary against this contract (the reference trusts its producer and fails later with shape errors
deep inside `init_data`), converts the arrays to the dtypes the device path expects and returns a clean copy; nothing here
touches the device."""
import pickle

import numpy as np

MAX_PERSONS = 32          # per scene (csrc/grecon_wide.hip; 8 on the instances with the on-chip arena)
REQUIRED = ('smpl_pose_quat_wroot', 'smpl_beta', 'root_trans', 'kp_2d', 'cam_K', 'bboxes_dict')


class WireFormatError(ValueError):
    pass


def _fail(pid, msg):
    raise WireFormatError('person %r: %s' % (pid, msg))


def normalise_person(pid, src, check_rotations=True, check_values=True):
    """check_values=False leaves the value checks (finite numbers, orthonormal rotations) to the caller: the batched device path runs them on
    the uploaded arrays instead of person by person on the host (GlobalReconOptimizer.stage_inputs)."""
    for k in REQUIRED:
        if k not in src:
            _fail(pid, 'missing key %r (have %s)' % (k, sorted(src.keys())))
    bb = src['bboxes_dict']
    if 'exist' not in bb:
        _fail(pid, "bboxes_dict has no 'exist' mask")
    exist = np.asarray(bb['exist'])
    if exist.ndim != 1 or exist.size < 2:
        _fail(pid, "bboxes_dict['exist'] must be a 1-D mask over the video frames, got shape %s" % (exist.shape,))
    exist01 = (exist != 0)
    if not np.array_equal(exist01.astype(exist.dtype), exist):
        _fail(pid, "bboxes_dict['exist'] must hold 0/1 values")
    n_vis = int(exist01.sum())
    if n_vis < 2:
        _fail(pid, 'needs at least two detections to interpolate between (has %d)' % n_vis)
    rot = np.asarray(src['smpl_pose_quat_wroot'], dtype=np.float32)
    if rot.shape[0] != n_vis or rot.size != n_vis * 24 * 9:
        _fail(pid, "smpl_pose_quat_wroot must hold 24 rotation matrices for each of the %d detections, got shape %s" % (n_vis, rot.shape))
    rot = rot.reshape(n_vis, 54, 4)
    out = {'smpl_pose_quat_wroot': rot}
    for key, tail in (('smpl_beta', (10,)), ('root_trans', (3,)), ('cam_K', (3, 3))):
        a = np.asarray(src[key], dtype=np.float32)
        if a.shape != (n_vis,) + tail:
            _fail(pid, '%s must have shape %s, got %s' % (key, (n_vis,) + tail, a.shape))
        out[key] = a
    kp = np.asarray(src['kp_2d'])
    if kp.ndim != 3 or kp.shape[0] != n_vis or kp.shape[1] < 24 or kp.shape[2] != 2:
        _fail(pid, 'kp_2d must have shape (%d, >=24, 2), got %s' % (n_vis, kp.shape))
    out['kp_2d'] = kp
    for key, a in list(out.items()) if check_values else ():
        if not np.all(np.isfinite(a)):
            _fail(pid, '%s contains non-finite values' % key)
    if check_rotations and check_values:
        R = rot.reshape(n_vis * 24, 3, 3)
        r = np.ascontiguousarray(R.reshape(-1, 9).T)                                       # nine component vectors: |R R^T - I| from the six row products
        dot = lambda i, j: r[3 * i] * r[3 * j] + r[3 * i + 1] * r[3 * j + 1] + r[3 * i + 2] * r[3 * j + 2]
        err = max(float(np.abs(dot(i, j) - (1.0 if i == j else 0.0)).max()) for i in range(3) for j in range(i, 3))
        if err > 1e-2:
            _fail(pid, 'smpl_pose_quat_wroot does not hold rotation matrices (|R R^T - I| = %.3g); the field is named after quaternions but '
                       'carries 24 x 3 x 3 matrices regrouped by 4 (demo.py:320)' % err)
    frames = np.flatnonzero(exist01)
    if 'frames' in src and not np.array_equal(np.asarray(src['frames']).reshape(-1), frames):
        _fail(pid, "'frames' disagrees with bboxes_dict['exist']")
    out['frames'] = frames
    out['frame2ind'] = {int(f): i for i, f in enumerate(frames)}
    nb = dict(bb)
    nb['exist'] = exist
    nb.setdefault('start', int(frames[0]))
    nb.setdefault('end', int(frames[-1]))
    nb.setdefault('num_frames', n_vis)
    nb.setdefault('exist_frames', frames)
    out['bboxes_dict'] = nb
    for k, v in src.items():          # anything else the producer added travels unchanged
        out.setdefault(k, v)
    return out


def check_layout(est):
    """The structural half of normalise_est -- keys, shapes, the 0/1 detection mask, a common video length -- WITHOUT building the
    normalised copies: the batched device path (GlobalReconOptimizer.stage_inputs) scatters straight from the producer's arrays and
    checks the values on the device.  Returns the video length."""
    if not isinstance(est, dict) or not est:
        raise WireFormatError('est_dict must be a non-empty {person_id: dict}')
    if len(est) > MAX_PERSONS:
        raise WireFormatError('at most %d persons per sequence are supported (got %d)' % (MAX_PERSONS, len(est)))
    n_fr = None
    for pid, src in est.items():
        for k in REQUIRED:
            if k not in src:
                _fail(pid, 'missing key %r (have %s)' % (k, sorted(src.keys())))
        bb = src['bboxes_dict']
        if 'exist' not in bb:
            _fail(pid, "bboxes_dict has no 'exist' mask")
        exist = np.asarray(bb['exist'])
        if exist.ndim != 1 or exist.size < 2:
            _fail(pid, "bboxes_dict['exist'] must be a 1-D mask over the video frames, got shape %s" % (exist.shape,))
        n_vis = int(np.count_nonzero(exist))
        if n_vis != int(np.count_nonzero(exist == 1)):
            _fail(pid, "bboxes_dict['exist'] must hold 0/1 values")
        if n_vis < 2:
            _fail(pid, 'needs at least two detections to interpolate between (has %d)' % n_vis)
        rot = src['smpl_pose_quat_wroot']
        if np.shape(rot)[0] != n_vis or np.size(rot) != n_vis * 216:
            _fail(pid, "smpl_pose_quat_wroot must hold 24 rotation matrices for each of the %d detections, got shape %s" % (n_vis, np.shape(rot)))
        for key, tail in (('smpl_beta', (10,)), ('root_trans', (3,)), ('cam_K', (3, 3))):
            if np.shape(src[key]) != (n_vis,) + tail:
                _fail(pid, '%s must have shape %s, got %s' % (key, (n_vis,) + tail, np.shape(src[key])))
        kps = np.shape(src['kp_2d'])
        if len(kps) != 3 or kps[0] != n_vis or kps[1] < 24 or kps[2] != 2:
            _fail(pid, 'kp_2d must have shape (%d, >=24, 2), got %s' % (n_vis, kps))
        if n_fr is None:
            n_fr = exist.size
        elif exist.size != n_fr:
            raise WireFormatError("all persons must share the video length; bboxes_dict['exist'] lengths differ (%d vs %d)" % (n_fr, exist.size))
    return n_fr


def normalise_est(est, check_rotations=True, check_values=True):
    """Validated copy of `est_dict`.  All persons must cover the same number of video frames (global_recon_model.py:85)."""
    if not isinstance(est, dict) or not est:
        raise WireFormatError('est_dict must be a non-empty {person_id: dict}')
    if len(est) > MAX_PERSONS:
        raise WireFormatError('at most %d persons per sequence are supported (got %d)' % (MAX_PERSONS, len(est)))
    out = {pid: normalise_person(pid, src, check_rotations, check_values) for pid, src in est.items()}
    lens = {pid: len(d['bboxes_dict']['exist']) for pid, d in out.items()}
    if len(set(lens.values())) != 1:
        raise WireFormatError("all persons must share the video length; bboxes_dict['exist'] lengths: %s" % lens)
    return out


def make_in_dict(est, seq_name, gt=None, gt_meta=None, validate=True):
    """`in_dict` of run_demo.py:80 / run_dataset.py:98-103."""
    return {'est': normalise_est(est) if validate else est, 'gt': gt if gt is not None else {}, 'gt_meta': gt_meta if gt_meta is not None else {},
            'seq_name': seq_name}


def load_pose_pkl(path, seq_name=None):
    with open(path, 'rb') as f:
        est = pickle.load(f)
    import os
    return make_in_dict(est, seq_name or os.path.splitext(os.path.basename(os.path.dirname(os.path.abspath(path))))[0])


# ---- ground truth (evaluation) ------------------------------------------------------------------------------------------------------

def normalise_gt(gt_dict, num_frames=None):
    """The ground-truth pickle `preprocess/preprocess_3dpw.py:115-153` writes: {'person_data': {pid: {'pose' (T,72), 'shape' (10,),
    'root_trans' (T,3), 'pose_cam', 'root_trans_cam', 'visible', ...}}, 'meta': {'cam_pose', 'cam_K', ...}}.  Returns
    (person_data, meta) as run_dataset.py:98-101 hands them to optimize(); only the fields the evaluator reads
    (global_recon/utils/evaluator.py:238-262) are required."""
    if not isinstance(gt_dict, dict) or 'person_data' not in gt_dict:
        raise WireFormatError("ground truth must be {'person_data': {...}, 'meta': {...}}")
    out = {}
    for pid, src in gt_dict['person_data'].items():
        for k in ('pose', 'shape', 'root_trans'):
            if k not in src:
                _fail(pid, 'ground truth lacks %r' % k)
        pose = np.asarray(src['pose'], dtype=np.float32)
        if pose.ndim != 2 or pose.shape[1] != 72:
            _fail(pid, 'ground-truth pose must be (T, 72) axis-angle, got %s' % (pose.shape,))
        T = pose.shape[0]
        if num_frames is not None and T != num_frames:
            _fail(pid, 'ground truth covers %d frames, the estimate %d' % (T, num_frames))
        shape = np.asarray(src['shape'], dtype=np.float32).reshape(-1)
        if shape.size < 10:
            _fail(pid, 'ground-truth shape needs 10 coefficients, got %d' % shape.size)
        trans = np.asarray(src['root_trans'], dtype=np.float32)
        if trans.shape != (T, 3):
            _fail(pid, 'ground-truth root_trans must be (%d, 3), got %s' % (T, trans.shape))
        d = dict(src)
        d.update(pose=pose, shape=shape[:10], root_trans=trans)
        for k in ('pose', 'shape', 'root_trans'):
            if not np.all(np.isfinite(d[k])):
                _fail(pid, 'ground-truth %s contains non-finite values' % k)
        out[pid] = d
    return out, dict(gt_dict.get('meta', {}))
