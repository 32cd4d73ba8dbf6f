"""Device handle of the two motion priors (glamr_nets_* in include/glamr_hip.h): weight hand-over and the batched inference call.
All arithmetic is in the HIP kernels (glamr_amd/csrc/nets.hip, nn_kernels.hpp); this module only marshals tensors."""
import ctypes

import numpy as np
import torch

from .. import _lib
from .layouts import INFILLER_LAYOUT, TRAJPRED_LAYOUT

NETS_INFILL, NETS_TRAJ = 1, 2
PAST, CUR, NZ = 10, 30, 128


def num_windows(seq_len):
    """ceil((T - past) / cur)  (motion_infiller_vae.py:625)"""
    return int(np.ceil((seq_len - PAST) / CUR))


def _pack(state_dict, layout):
    blob, desc, off = [], (_lib.TensorDesc * len(layout))(), 0
    for i, (key, shape) in enumerate(layout):
        if key not in state_dict:
            raise KeyError('checkpoint lacks %r' % key)
        a = np.ascontiguousarray(np.asarray(state_dict[key].detach().cpu().numpy() if hasattr(state_dict[key], 'detach') else state_dict[key],
                                            dtype=np.float32))
        if tuple(a.shape) != tuple(shape):
            raise ValueError('%s has shape %s, expected %s' % (key, a.shape, shape))
        desc[i].offset, desc[i].rows, desc[i].cols = off, shape[0], (shape[1] if len(shape) > 1 else 0)
        blob.append(a.reshape(-1))
        off += a.size
    return np.concatenate(blob), desc


class MotionPriorsHandle:
    """One per device.  infiller_sd / trajpred_sd: state_dicts with the reference's key names."""

    def __init__(self, infiller_sd, trajpred_sd, rest_joints, parents, device):
        if device.type != 'cuda':
            raise RuntimeError('the motion priors run on an MI355X only; there is no CPU fallback')
        self.device = device
        L = _lib.lib()
        ib, idesc = _pack(infiller_sd, INFILLER_LAYOUT)
        tb, tdesc = _pack(trajpred_sd, TRAJPRED_LAYOUT)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(L.glamr_nets_create(ctypes.byref(h), _lib.ptr(ib), idesc, len(INFILLER_LAYOUT), _lib.ptr(tb), tdesc, len(TRAJPRED_LAYOUT),
                                           _lib.ptr(np.ascontiguousarray(rest_joints, dtype=np.float32)),
                                           _lib.ptr(np.ascontiguousarray(parents, dtype=np.int32))))
        self.h = h

    def infer(self, body_pose, visible, lens, motion_eps=None, traj_eps=None, infill=True, traj=True):
        """body_pose (B,T,69) fp32 device, visible (B,T) 1/0, lens list[int].  Returns dict of device tensors."""
        L = _lib.lib()
        B, T = body_pose.shape[:2]
        dev = body_pose.device
        body_pose = body_pose.float().contiguous()
        lens_np = np.ascontiguousarray(lens, dtype=np.int32)
        out = {}
        flags = (NETS_INFILL if infill else 0) | (NETS_TRAJ if traj else 0)
        n_win_max = 0
        if infill:
            visible = visible.float().contiguous()
            n_win_max = motion_eps.shape[1]
            motion_eps = motion_eps.float().contiguous()
            out['pose'] = torch.empty((B, T, 69), device=dev)
        if traj:
            traj_eps = traj_eps.float().contiguous()
            out['local_traj'] = torch.empty((B, T, 11), device=dev)
            out['trans'] = torch.empty((B, T, 3), device=dev)
            out['orient'] = torch.empty((B, T, 3), device=dev)
        ws = torch.empty(L.glamr_nets_workspace_bytes(self.h, B, T), dtype=torch.uint8, device=dev)
        _lib.check(L.glamr_nets_infer(self.h, B, T, _lib.ptr(lens_np), _lib.ptr(body_pose), _lib.ptr(visible) if infill else None,
                                      _lib.ptr(motion_eps) if infill else None, n_win_max, _lib.ptr(traj_eps) if traj else None,
                                      _lib.ptr(out.get('pose')), _lib.ptr(out.get('local_traj')), _lib.ptr(out.get('trans')),
                                      _lib.ptr(out.get('orient')), flags, _lib.ptr(ws), _lib.current_stream()))
        return out
