"""Host-side (numpy, float32) rigid-transform helpers used ONCE per sequence while GlobalReconOptimizer.init_data assembles a
scene (initial camera, heading initialisation, relative transforms).  They follow the reference's conventions -- quaternions
(w,x,y,z), kornia-style conversions with their epsilons -- so that the initial state matches
global_recon/models/global_recon_model.py:166-169,178-183,273-317.  The per-iteration maths lives in the HIP kernels
(glamr_amd/csrc/rotmath.hpp), not here.

Reference counterparts: lib/utils/torch_transform.py (quat_mul :10-28, get_heading :172-185, rot6d :214-227, make/inverse
transform :246-279), lib/utils/konia_transform.py (aa<->R<->quat :234-313, :349-443, :560-630, :753-826),
traj_pred/utils/traj_utils.py (traj_global2local_heading :44-62, interp_orient_q_sep_heading :120-141).
"""
import numpy as np

F = np.float32
_BASE_CONJ = np.array([0.5, -0.5, -0.5, -0.5], dtype=F)
_BASE = np.array([0.5, 0.5, 0.5, 0.5], dtype=F)


def _sdiv(num, den, eps=1e-6):
    den = np.where(np.abs(den) < eps, den + F(eps), den)
    return num / den


def safe_atan2(y, x, eps=1e-6):
    y = np.where((np.abs(y) < eps) & (np.abs(x) < eps), y + F(eps), y)
    return np.arctan2(y, x).astype(F)


def aa_to_rotmat(aa):
    aa = np.asarray(aa, dtype=F)
    th2 = (aa * aa).sum(-1, keepdims=True)
    th = np.sqrt(np.maximum(th2, F(1e-6)))
    w = aa / (th + F(1e-6))
    wx, wy, wz = w[..., 0:1], w[..., 1:2], w[..., 2:3]
    c, s = np.cos(th), np.sin(th)
    k = F(1.0) - c
    normal = np.concatenate([c + wx * wx * k, wx * wy * k - wz * s, wy * s + wx * wz * k,
                             wz * s + wx * wy * k, c + wy * wy * k, -wx * s + wy * wz * k,
                             -wy * s + wx * wz * k, wx * s + wy * wz * k, c + wz * wz * k], axis=-1)
    rx, ry, rz = aa[..., 0:1], aa[..., 1:2], aa[..., 2:3]
    one = np.ones_like(rx)
    taylor = np.concatenate([one, -rz, ry, rz, one, -rx, -ry, rx, one], axis=-1)
    out = np.where(th2 > F(1e-6), normal, taylor)
    return out.reshape(aa.shape[:-1] + (3, 3)).astype(F)


def rotmat_to_quat(R, eps=1e-6):
    m = np.asarray(R, dtype=F).reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [m[..., i:i + 1] for i in range(9)]
    tr = m00 + m11 + m22
    sq = np.sqrt(np.maximum(tr + F(1.0), F(eps))) * F(2.0)
    q0 = np.concatenate([F(0.25) * sq, _sdiv(m21 - m12, sq), _sdiv(m02 - m20, sq), _sdiv(m10 - m01, sq)], -1)
    sq = np.sqrt(np.maximum(F(1.0) + m00 - m11 - m22, F(eps))) * F(2.0)
    q1 = np.concatenate([_sdiv(m21 - m12, sq), F(0.25) * sq, _sdiv(m01 + m10, sq), _sdiv(m02 + m20, sq)], -1)
    sq = np.sqrt(np.maximum(F(1.0) + m11 - m00 - m22, F(eps))) * F(2.0)
    q2 = np.concatenate([_sdiv(m02 - m20, sq), _sdiv(m01 + m10, sq), F(0.25) * sq, _sdiv(m12 + m21, sq)], -1)
    sq = np.sqrt(np.maximum(F(1.0) + m22 - m00 - m11, F(eps))) * F(2.0)
    q3 = np.concatenate([_sdiv(m10 - m01, sq), _sdiv(m02 + m20, sq), _sdiv(m12 + m21, sq), F(0.25) * sq], -1)
    inner = np.where(m11 > m22, q2, q3)
    mid = np.where((m00 > m11) & (m00 > m22), q1, inner)
    return np.where(tr > 0.0, q0, mid).astype(F)


def quat_to_rotmat(q):
    q = np.asarray(q, dtype=F)
    q = q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), F(1e-12))
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    out = np.stack([1 - (ty * y + tz * z), ty * x - tz * w, tz * x + ty * w,
                    ty * x + tz * w, 1 - (tx * x + tz * z), tz * y - tx * w,
                    tz * x - ty * w, tz * y + tx * w, 1 - (tx * x + ty * y)], axis=-1)
    return out.reshape(q.shape[:-1] + (3, 3)).astype(F)


def quat_to_aa(q, eps=1e-6):
    q = np.asarray(q, dtype=F)
    c, v = q[..., 0], q[..., 1:]
    s2 = (v * v).sum(-1)
    s = np.sqrt(np.maximum(s2, F(eps)))
    tt = F(2.0) * np.where(c < 0, safe_atan2(-s, -c), safe_atan2(s, c))
    k = np.where(s2 > 0, _sdiv(tt, s, eps), F(2.0))
    return (v * k[..., None]).astype(F)


def aa_to_quat(aa, eps=1e-6):
    aa = np.asarray(aa, dtype=F)
    th2 = (aa * aa).sum(-1, keepdims=True)
    th = np.sqrt(np.maximum(th2, F(eps)))
    half = th * F(0.5)
    pos = th2 > 0
    k = np.where(pos, _sdiv(np.sin(half), th, eps), F(0.5))
    w = np.where(pos, np.cos(half), F(1.0))
    return np.concatenate([w, aa * k], axis=-1).astype(F)


def quat_mul(a, b):
    w1, x1, y1, z1 = [a[..., i] for i in range(4)]
    w2, x2, y2, z2 = [b[..., i] for i in range(4)]
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=-1).astype(F)


def quat_conj(q):
    return np.concatenate([q[..., :1], -q[..., 1:]], axis=-1)


def quat_angle_between(q1, q2, eps=1e-6):
    w = quat_mul(q1, quat_conj(q2))[..., 0]
    return np.arccos(np.clip(2 * w * w - 1, -1 + eps, 1 - eps))


def unit(x, eps=1e-9):
    return x / np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), F(eps))


def sixd_to_rotmat(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = unit(a1)
    b2 = unit(a2 - (b1 * a2).sum(-1, keepdims=True) * b1)
    return np.stack([b1, b2, np.cross(b1, b2)], axis=-1).astype(F)


def rotmat_to_6d(R):
    return np.concatenate([R[..., 0], R[..., 1]], axis=-1)


def make_transform(aa, trans):
    T = np.zeros(aa.shape[:-1] + (4, 4), dtype=F)
    T[..., :3, :3] = aa_to_rotmat(aa)
    T[..., :3, 3] = trans
    T[..., 3, 3] = 1.0
    return T


def invert_transform(T):
    out = np.zeros_like(T)
    out[..., :3, :3] = np.swapaxes(T[..., :3, :3], -1, -2)
    out[..., :3, 3] = -np.einsum('...ji,...j->...i', T[..., :3, :3], T[..., :3, 3])
    out[..., 3, 3] = 1.0
    return out


def heading_of(q):
    return F(2.0) * safe_atan2(q[..., 3], q[..., 0])


def heading_quat_of(q):
    z = np.zeros_like(q[..., 0])
    return unit(np.stack([q[..., 0], z, z, q[..., 3]], axis=-1))


def heading_to_quat(theta):
    z = np.zeros_like(theta)
    return aa_to_quat(np.stack([z, z, theta], axis=-1))


def heading_to_vec(theta):
    return np.stack([np.cos(theta), np.sin(theta)], axis=-1).astype(F)


def _rot2d(xy, th):
    c, s = np.cos(th), np.sin(th)
    return np.stack([xy[..., 0] * c - xy[..., 1] * s, xy[..., 0] * s + xy[..., 1] * c], axis=-1)


def global_to_local_traj(trans, q):
    """traj_global2local_heading (6d): (T,3), (T,4) -> (T,11)."""
    q = quat_mul(q, np.broadcast_to(_BASE_CONJ, q.shape))
    h = heading_of(q)
    hq = heading_quat_of(q)
    local6 = rotmat_to_6d(quat_to_rotmat(quat_mul(quat_conj(hq), q)))
    xy, z = trans[..., :2], trans[..., 2]
    dh = np.concatenate([h[:1], h[1:] - h[:-1]])
    dxy = np.concatenate([xy[:1], _rot2d(xy[1:] - xy[:-1], -h[:-1])])
    return np.concatenate([dxy, z[..., None], local6, heading_to_vec(dh)], axis=-1).astype(F)


def lerp_extrapolate(idx, values, n):
    """scipy interp1d(idx, values, axis=0, assume_sorted=True, fill_value='extrapolate') evaluated on arange(n); float64 result."""
    from scipy.interpolate import interp1d
    f = interp1d(idx.astype(np.float32), values, axis=0, assume_sorted=True, fill_value='extrapolate')
    return f(np.arange(n, dtype=np.float32))


def interp_orient_sep_heading(q_vis, vis_frames):
    """interp_orient_q_sep_heading: heading vector and de-headed 6D orientation are interpolated separately over invisible frames."""
    q = quat_mul(q_vis, np.broadcast_to(_BASE_CONJ, q_vis.shape))
    hq = heading_quat_of(q)
    hvec = heading_to_vec(heading_of(q))
    loc6 = rotmat_to_6d(quat_to_rotmat(quat_mul(quat_conj(hq), q)))
    n = vis_frames.shape[0]
    idx = np.where(vis_frames)[0]
    hv = lerp_extrapolate(idx, hvec, n).astype(F)
    l6 = lerp_extrapolate(idx, loc6, n).astype(F)
    qi = quat_mul(heading_to_quat(safe_atan2(hv[..., 1], hv[..., 0])), rotmat_to_quat(sixd_to_rotmat(l6)))
    return quat_mul(qi, np.broadcast_to(_BASE, qi.shape))


def rotmat_to_rotvec_nearest(M):
    """(N,3,3) approximately-orthogonal float matrices -> (N,3) rotation vectors (float64), equal to
    scipy `Rotation.from_matrix(M).as_rotvec()` (global_recon_model.py:105-108) to double-precision round-off, without its
    per-call LAPACK SVD: the nearest rotation (polar factor) is reached by three Newton steps X <- (X + X^-T)/2, which converge
    quadratically from HybrIK's float32-accurate matrices; the quaternion is then extracted from the dominant diagonal pattern and
    mapped to angle * axis with the positive-w convention."""
    X = np.asarray(M, dtype=np.float64)
    for _ in range(3):
        a, b, c = X[:, 0, 0], X[:, 0, 1], X[:, 0, 2]
        d, e, f = X[:, 1, 0], X[:, 1, 1], X[:, 1, 2]
        g, h, i = X[:, 2, 0], X[:, 2, 1], X[:, 2, 2]
        cof = np.stack([np.stack([e * i - f * h, f * g - d * i, d * h - e * g], -1),
                        np.stack([c * h - b * i, a * i - c * g, b * g - a * h], -1),
                        np.stack([b * f - c * e, c * d - a * f, a * e - b * d], -1)], -2)      # cofactor matrix = det * X^-T
        det = a * cof[:, 0, 0] + b * cof[:, 0, 1] + c * cof[:, 0, 2]
        X = 0.5 * (X + cof / det[:, None, None])
    m00, m11, m22 = X[:, 0, 0], X[:, 1, 1], X[:, 2, 2]
    tr = m00 + m11 + m22
    q = np.empty((X.shape[0], 4))                                   # (x, y, z, w)
    choice = np.argmax(np.stack([m00, m11, m22, tr], -1), axis=-1)
    for k in range(3):
        idx = np.where(choice == k)[0]
        if len(idx) == 0:
            continue
        i_, j_, k_ = k, (k + 1) % 3, (k + 2) % 3
        q[idx, i_] = 1 - tr[idx] + 2 * X[idx, i_, i_]
        q[idx, j_] = X[idx, j_, i_] + X[idx, i_, j_]
        q[idx, k_] = X[idx, k_, i_] + X[idx, i_, k_]
        q[idx, 3] = X[idx, k_, j_] - X[idx, j_, k_]
    idx = np.where(choice == 3)[0]
    q[idx, 0] = X[idx, 2, 1] - X[idx, 1, 2]
    q[idx, 1] = X[idx, 0, 2] - X[idx, 2, 0]
    q[idx, 2] = X[idx, 1, 0] - X[idx, 0, 1]
    q[idx, 3] = 1 + tr[idx]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[q[:, 3] < 0] *= -1
    s = np.linalg.norm(q[:, :3], axis=1)
    angle = 2 * np.arctan2(s, q[:, 3])
    small = angle <= 1e-3
    a2 = angle * angle
    scale = np.where(small, 2 + a2 / 12 + 7 * a2 * a2 / 2880, angle / np.where(small, 1.0, np.sin(angle / 2)))
    return q[:, :3] * scale[:, None]
