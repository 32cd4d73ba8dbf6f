"""TEST INFRASTRUCTURE ONLY -- CPU (torch) restatement of the rotation / rigid-transform maths on the GLAMR hot path.

Each function names the reference lines it follows (paths relative to /root/reference).  Quaternions are (w, x, y, z).
Epsilon placement, `where` branches and clamp positions are kept because the optimiser differentiates through them
(SURVEY.md Appendix C items 8, 9).  Pinned against the unmodified reference in tests/test_oracle_vs_reference.py.
"""
import math
import torch


# ---- small helpers -------------------------------------------------------------------------------------------------

def safe_atan2(y, x, eps=1e-6):
    """lib/utils/torch_transform.py:63-67 -- nudge y by +eps where both arguments are within eps of zero."""
    tiny = (y.abs() < eps) & (x.abs() < eps)
    return torch.atan2(torch.where(tiny, y + eps, y), x)


def _safe_div(num, den, eps=1e-6):
    """lib/utils/konia_transform.py:340-343 -- add eps to denominators closer to zero than eps."""
    return num / torch.where(den.abs() < eps, den + eps, den)


def unit(x, eps=1e-9):
    """lib/utils/torch_transform.py:6-7"""
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


# ---- quaternion algebra (lib/utils/torch_transform.py:10-60) ---------------------------------------------------------

def quat_mul(a, b):
    """:10-28 -- Hamilton product in the reference's 9-multiplication arrangement (kept for rounding parity)."""
    w1, x1, y1, z1 = a.unbind(-1)
    w2, x2, y2, z2 = b.unbind(-1)
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    return torch.stack([qq - ww + (z1 - y1) * (y2 - z2), qq - xx + (x1 + w1) * (x2 + w2),
                        qq - yy + (w1 - x1) * (y2 + z2), qq - zz + (z1 + y1) * (w2 - x2)], dim=-1)


def quat_conj(q):
    """:31-35"""
    return torch.cat([q[..., :1], -q[..., 1:]], dim=-1)


def quat_rotate(q, v):
    """:38-45 quat_apply"""
    xyz = q[..., 1:]
    t = torch.cross(xyz, v, dim=-1) * 2
    return v + q[..., :1] * t + torch.cross(xyz, t, dim=-1)


def quat_angle_between(q1, q2, eps=1e-6):
    """:48-60 quat_angle(quat_mul(q1, conj(q2)))"""
    w = quat_mul(q1, quat_conj(q2))[..., 0]
    return torch.acos((2 * w * w - 1).clamp(-1 + eps, 1 - eps))


# ---- heading helpers (:172-211) ---------------------------------------------------------------------------------------

def heading_of(q, eps=1e-6):
    """:172-177 get_heading"""
    return 2 * safe_atan2(q[..., 3], q[..., 0], eps)


def heading_quat_of(q):
    """:180-185 get_heading_q -- keep (w, z), renormalise."""
    z = torch.zeros_like(q[..., 0])
    return unit(torch.stack([q[..., 0], z, z, q[..., 3]], dim=-1))


def heading_to_vec(theta):
    """:188-191"""
    return torch.stack([torch.cos(theta), torch.sin(theta)], dim=-1)


def vec_to_heading(v):
    """:194-197"""
    return safe_atan2(v[..., 1], v[..., 0])


def heading_to_quat(theta):
    """:200-204 -- axis-angle (0,0,theta) through the kornia converter."""
    z = torch.zeros_like(theta)
    return aa_to_quat(torch.stack([z, z, theta], dim=-1))


def remove_heading(q, hq=None):
    """:207-211 deheading_quat"""
    return quat_mul(quat_conj(heading_quat_of(q) if hq is None else hq), q)


# ---- kornia conversions (lib/utils/konia_transform.py) ---------------------------------------------------------------

def aa_to_rotmat(aa):
    """:234-313 angle_axis_to_rotation_matrix.  First-order Taylor branch when theta^2 <= 1e-6; the main branch divides by
    (theta + 1e-6), so its output is not exactly orthonormal."""
    shape = aa.shape[:-1]
    v = aa.reshape(-1, 3)
    theta2 = (v * v).sum(-1, keepdim=True)
    theta = torch.sqrt(theta2.clamp_min(1e-6))
    w = v / (theta + 1e-6)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    k = 1.0 - c
    normal = torch.cat([c + wx * wx * k, wx * wy * k - wz * s, wy * s + wx * wz * k,
                        wz * s + wx * wy * k, c + wy * wy * k, -wx * s + wy * wz * k,
                        -wy * s + wx * wz * k, wx * s + wy * wz * k, c + wz * wz * k], dim=1)
    rx, ry, rz = v[:, 0:1], v[:, 1:2], v[:, 2:3]
    one = torch.ones_like(rx)
    taylor = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1)
    big = (theta2 > 1e-6).to(v.dtype)
    return (big * normal + (1 - big) * taylor).view(shape + (3, 3))


def rotmat_to_quat(R, eps=1e-6):
    """:349-443 rotation_matrix_to_quaternion -- all four candidates are evaluated, then selected by `where`."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [m[..., i:i + 1] for i in range(9)]
    tr = m00 + m11 + m22

    sq = torch.sqrt((tr + 1.0).clamp_min(eps)) * 2.0
    q_tr = torch.cat([0.25 * sq, _safe_div(m21 - m12, sq), _safe_div(m02 - m20, sq), _safe_div(m10 - m01, sq)], dim=-1)
    sq = torch.sqrt((1.0 + m00 - m11 - m22).clamp_min(eps)) * 2.0
    q_x = torch.cat([_safe_div(m21 - m12, sq), 0.25 * sq, _safe_div(m01 + m10, sq), _safe_div(m02 + m20, sq)], dim=-1)
    sq = torch.sqrt((1.0 + m11 - m00 - m22).clamp_min(eps)) * 2.0
    q_y = torch.cat([_safe_div(m02 - m20, sq), _safe_div(m01 + m10, sq), 0.25 * sq, _safe_div(m12 + m21, sq)], dim=-1)
    sq = torch.sqrt((1.0 + m22 - m00 - m11).clamp_min(eps)) * 2.0
    q_z = torch.cat([_safe_div(m10 - m01, sq), _safe_div(m02 + m20, sq), _safe_div(m12 + m21, sq), 0.25 * sq], dim=-1)
    inner = torch.where(m11 > m22, q_y, q_z)
    mid = torch.where((m00 > m11) & (m00 > m22), q_x, inner)
    return torch.where(tr > 0.0, q_tr, mid)


def quat_to_rotmat(q):
    """:470-555 quaternion_to_rotation_matrix (normalises with eps 1e-12 first)."""
    qn = torch.nn.functional.normalize(q, p=2.0, dim=-1, eps=1e-12)
    w, x, y, z = qn.unbind(-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.tensor(1.0)
    return torch.stack([one - (tyy + tzz), txy - twz, txz + twy,
                        txy + twz, one - (txx + tzz), tyz - twx,
                        txz - twy, tyz + twx, one - (txx + tyy)], dim=-1).view(q.shape[:-1] + (3, 3))


def quat_to_aa(q, eps=1e-6):
    """:560-630 quaternion_to_angle_axis"""
    c, q1, q2, q3 = q.unbind(-1)
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    s = torch.sqrt(s2.clamp_min(eps))
    two_theta = 2.0 * torch.where(c < 0.0, safe_atan2(-s, -c), safe_atan2(s, c))
    k = torch.where(s2 > 0.0, _safe_div(two_theta, s, eps), 2.0 * torch.ones_like(s))
    return torch.stack([q1 * k, q2 * k, q3 * k], dim=-1)


def aa_to_quat(aa, eps=1e-6):
    """:753-826 angle_axis_to_quaternion"""
    th2 = (aa * aa).sum(-1, keepdim=True)
    th = torch.sqrt(th2.clamp_min(eps))
    half = th * 0.5
    pos = th2 > 0.0
    k = torch.where(pos, _safe_div(torch.sin(half), th, eps), 0.5 * torch.ones_like(half))
    w = torch.where(pos, torch.cos(half), torch.ones_like(half))
    return torch.cat([w, aa * k], dim=-1)


def rotmat_to_aa(R):
    """:316-339"""
    return quat_to_aa(rotmat_to_quat(R))


# ---- 6D representation and rigid transforms (lib/utils/torch_transform.py:214-279) -----------------------------------

def rotmat_to_6d(R):
    """:214-217 -- first two COLUMNS, concatenated."""
    return torch.cat([R[..., 0], R[..., 1]], dim=-1)


def sixd_to_rotmat(d6):
    """:220-227 Gram-Schmidt; result columns b1 b2 b3."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = unit(a1)
    b2 = unit(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    return torch.stack([b1, b2, torch.cross(b1, b2, dim=-1)], dim=-1)


def aa_to_6d(aa):
    """:230-231"""
    return rotmat_to_6d(aa_to_rotmat(aa))


def quat_to_6d(q):
    """:238-239"""
    return rotmat_to_6d(quat_to_rotmat(q))


def sixd_to_quat(d6):
    """:242-243"""
    return rotmat_to_quat(sixd_to_rotmat(d6))


def make_transform(rot, trans, rot_type=None):
    """:246-254"""
    if rot_type == 'axis_angle':
        rot = aa_to_rotmat(rot)
    elif rot_type == '6d':
        rot = sixd_to_rotmat(rot)
    M = torch.eye(4, dtype=trans.dtype, device=trans.device).repeat(rot.shape[:-2] + (1, 1))
    M[..., :3, :3] = rot
    M[..., :3, 3] = trans
    return M


def apply_transform(M, p):
    """:257-262 transform_trans -- p (...,3) through M (4x4), broadcasting M over extra point dims."""
    ph = torch.cat([p, torch.ones_like(p[..., :1])], dim=-1)[..., None, :]
    while M.dim() < ph.dim():
        M = M.unsqueeze(-3)
    return torch.matmul(ph, M.transpose(-2, -1))[..., 0, :3]


def rotate_aa(M, aa):
    """:265-271 transform_rot -- left-multiply the rotation of an axis-angle by M's rotation block."""
    R = aa_to_rotmat(aa)
    while M.dim() < R.dim():
        M = M.unsqueeze(-3)
    return rotmat_to_aa(torch.matmul(M[..., :3, :3], R))


def invert_transform(M):
    """:274-279"""
    out = torch.zeros_like(M)
    out[..., :3, :3] = M[..., :3, :3].transpose(-2, -1)
    out[..., :3, 3] = -torch.matmul(M[..., :3, 3].unsqueeze(-2), M[..., :3, :3]).squeeze(-2)
    out[..., 3, 3] = 1.0
    return out


def project(p3d, K):
    """lib/utils/geometry.py:23-25 perspective_projection (t_form=None).  p3d (B,N,3), K (B,3,3)."""
    h = torch.matmul(K, p3d.transpose(2, 1)).transpose(2, 1)
    return h[:, :, :2] / (h[:, :, 2:] + 1e-8)


# ---- heading-frame trajectory representation (traj_pred/utils/traj_utils.py) -----------------------------------------

_BASE = (0.5, 0.5, 0.5, 0.5)


def _rot2d(xy, th):
    """:7-11"""
    c, s = torch.cos(th), torch.sin(th)
    return torch.stack([xy[..., 0] * c - xy[..., 1] * s, xy[..., 0] * s + xy[..., 1] * c], dim=-1)


def global_to_local_traj(trans, q):
    """:44-62 traj_global2local_heading (6d local orient).  trans (T,...,3), q (T,...,4) -> (T,...,11):
    [d_xy in the previous frame's heading coords (row 0: absolute xy), z, local 6d, (cos, sin) of d_heading (row 0: absolute)]."""
    base = torch.tensor(_BASE, device=q.device)
    q = quat_mul(q, quat_conj(base).expand_as(q))
    h = heading_of(q)
    hq = heading_quat_of(q)
    local6 = quat_to_6d(remove_heading(q, hq))
    xy, z = trans[..., :2], trans[..., 2]
    dh = torch.cat([h[[0]], h[1:] - h[:-1]])
    dxy = torch.cat([xy[[0]], _rot2d(xy[1:] - xy[:-1], -h[:-1])])
    return torch.cat([dxy, z.unsqueeze(-1), local6, heading_to_vec(dh)], dim=-1)


def local_to_global_traj(local, local_heading=True):
    """:65-88 traj_local2global_heading (6d, no deheading).  Two prefix sums over the time axis (dim 0)."""
    base = torch.tensor(_BASE, device=local.device)
    dxy_h, z = local[..., :2], local[..., 2]
    d6, dh_vec = local[..., 3:-2], local[..., -2:]
    dh = vec_to_heading(dh_vec)
    h = torch.cumsum(dh, dim=0) if local_heading else dh
    dxy = dxy_h.clone()
    dxy[1:] = _rot2d(dxy_h[1:], h[:-1])
    xy = torch.cumsum(dxy, dim=0)
    trans = torch.cat([xy, z.unsqueeze(-1)], dim=-1)
    q = quat_mul(heading_to_quat(h), sixd_to_quat(d6))
    q = quat_mul(q, base.expand_as(q))
    return trans, q


def world_to_heading_frame(q, trans):
    """:97-107 convert_traj_world2heading (apply_base_orient_after=False)."""
    base = torch.tensor(_BASE, device=q.device)
    qn = quat_mul(q, quat_conj(base).expand_as(q))
    inv_h = quat_conj(heading_quat_of(qn[0])).expand_as(qn)
    t = trans.clone()
    t[..., :2] -= trans[0, ..., :2]
    return quat_mul(inv_h, qn), quat_rotate(inv_h, t)


def interp_orient_sep_heading(q_vis, vis_frames):
    """:120-141 interp_orient_q_sep_heading -- linear inter/extrapolation (scipy interp1d on the host, float64 result cast to
    float32) of the heading vector and of the de-headed 6D orientation across invisible frames."""
    import numpy as np
    from scipy.interpolate import interp1d
    dev = q_vis.device
    base = torch.tensor(_BASE, device=dev)
    q = quat_mul(q_vis, quat_conj(base).expand_as(q_vis))
    hq = heading_quat_of(q)
    hvec = heading_to_vec(heading_of(q))
    loc6 = quat_to_6d(remove_heading(q, hq))
    n = vis_frames.shape[0]
    idx = torch.where(vis_frames)[0].cpu().numpy()
    grid = np.arange(n, dtype=np.float32)

    def lerp(x):
        f = interp1d(idx, x.cpu().numpy(), axis=0, assume_sorted=True, fill_value='extrapolate')
        return torch.tensor(f(grid), device=dev, dtype=torch.float32)

    qi = quat_mul(heading_to_quat(vec_to_heading(lerp(hvec))), sixd_to_quat(lerp(loc6)))
    return quat_mul(qi, base.expand_as(qi))
