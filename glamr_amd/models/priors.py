"""Device handle of the two motion priors (glamr_nets_* in include/glamr_hip.h): weight hand-over and the batched inference call.
All arithmetic is in the HIP kernels (glamr_amd/csrc/nets.hip, nn_kernels.hpp); this module only marshals tensors."""
import ctypes
import logging
import os
import sys

import numpy as np
import torch

from .. import _lib
from .layouts import INFILLER_LAYOUT, TRAJPRED_LAYOUT

NETS_INFILL, NETS_TRAJ, NETS_PERSISTENT, NETS_COSCHEDULE = 1, 2, 4, 8
VAE_INFER, VAE_TRAIN, VAE_RECON = 0, 1, 2
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
        # Which kernel family this checkpoint runs on is decided from its weights (glamr_nets_precision): said out loud, once per handle -- a
        # trained checkpoint that falls out of the fp16 planes' range is slower, not wrong, and the user should know which one they have.
        wc = (ctypes.c_double * 2)()
        self.fp32_only = bool(L.glamr_nets_precision(h, wc))
        self.precision_bound, self.largest_weight = float(wc[0]), float(wc[1])
        msg = ('motion priors on %s: %s (worst-case magnitude the fp16 operand planes would have to hold %.4g, largest weight %.4g; limit 3e4)'
               % (device, 'plain fp32 kernels (fp32 MFMA GEMMs, fp32 attention, small-batch LSTM; no fused blocks)' if self.fp32_only
                  else 'fp32-grade products on the fp16 matrix cores (two fp16 planes per operand, three MFMAs per k step)',
                  self.precision_bound, self.largest_weight))
        logging.getLogger('glamr_amd').log(logging.WARNING if self.fp32_only else logging.INFO, msg)
        if self.fp32_only or os.environ.get('GLAMR_VERBOSE'):
            print('[glamr_amd] ' + msg, file=sys.stderr)

    def close(self):
        """Releases the device weights and the captured launch graphs of this handle (glamr_nets_destroy).  Pending work must have
        completed; the handle cannot be used afterwards."""
        h, self.h = self.__dict__.get('h'), None
        if h:
            torch.cuda.synchronize(self.device)
            _lib.lib().glamr_nets_destroy(h)

    RESIDENT_GEOMETRIES = 3          # per stream

    def resident_set(self, B, T, n_win):
        """Persistent buffers for the batched pipeline (GlobalReconOptimizer.init_resident): inputs, outputs and workspace of one
        glamr_nets_infer call at FIXED addresses, one set per HIP stream and batch geometry.  The set is only ever reused by a later call on
        the SAME stream, after everything that consumed it was enqueued there, so stream order keeps it safe; fixed addresses are what
        lets the library replay the call's ~450 launches as one captured HIP graph.  At most RESIDENT_GEOMETRIES sets per stream are kept
        (least recently used dropped: a dataset of videos of many lengths must not grow device memory without bound); a set tells the
        library to capture (`persistent`) only from its SECOND use on -- a geometry seen once never pays capture + instantiation."""
        L = _lib.lib()
        sid = torch.cuda.current_stream(self.device).cuda_stream
        ring = self.__dict__.setdefault('_ring', {})
        per_stream = ring.setdefault(sid, {})          # insertion order = recency
        key = (B, T, n_win)
        cur = per_stream.pop(key, None)
        if cur is None:
            while len(per_stream) >= self.RESIDENT_GEOMETRIES:
                # work that still reads the dropped set is already enqueued on this stream; the caching allocator hands the blocks to later
                # allocations of the same stream only, which run after it
                per_stream.pop(next(iter(per_stream)))
            f32 = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)
            cur = dict(nets_pose=f32(B, T, 69), nets_vis=f32(B, T), meps=f32(B, n_win, NZ), teps=f32(B, NZ), pose=f32(B, T, 69),
                       local_traj=f32(B, T, 11), trans=f32(B, T, 3), orient=f32(B, T, 3),
                       ws=torch.empty(L.glamr_nets_workspace_bytes(self.h, B, T), dtype=torch.uint8, device=self.device), uses=0)
        cur['uses'] += 1
        cur['persistent'] = cur['uses'] >= 2
        per_stream[key] = cur
        return cur

    def infer(self, body_pose, visible, lens, motion_eps=None, traj_eps=None, infill=True, traj=True, buffers=None, coschedule=False):
        """body_pose (B,T,69) fp32 device, visible (B,T) 1/0, lens list[int].  Returns dict of device tensors.  `buffers`: a
        resident_set() whose output / workspace tensors are used instead of fresh allocations.  `coschedule`: the caller pipelines batches over
        two streams (GLAMR_NETS_COSCHEDULE, include/glamr_hip.h): the infiller runs on the kernels that fit beside a resident optimiser stage."""
        L = _lib.lib()
        B, T = body_pose.shape[:2]
        dev = body_pose.device
        body_pose = body_pose.float().contiguous()
        lens_np = np.ascontiguousarray(lens, dtype=np.int32)
        out = {}
        flags = (NETS_INFILL if infill else 0) | (NETS_TRAJ if traj else 0) | (NETS_PERSISTENT if buffers is not None and buffers.get('persistent', True) else 0) \
            | (NETS_COSCHEDULE if coschedule else 0)
        n_win_max = 0
        new = (lambda name, *shape: buffers[name]) if buffers is not None else (lambda name, *shape: torch.empty(shape, device=dev))
        if infill:
            visible = visible.float().contiguous()
            n_win_max = motion_eps.shape[1]
            motion_eps = motion_eps.float().contiguous()
            out['pose'] = new('pose', B, T, 69)
        if traj:
            traj_eps = traj_eps.float().contiguous()
            out['local_traj'] = new('local_traj', B, T, 11)
            out['trans'] = new('trans', B, T, 3)
            out['orient'] = new('orient', B, T, 3)
        ws = buffers['ws'] if buffers is not None else torch.empty(L.glamr_nets_workspace_bytes(self.h, B, T), dtype=torch.uint8, device=dev)
        _lib.check(L.glamr_nets_infer(self.h, B, T, _lib.ptr(lens_np), _lib.ptr(body_pose), _lib.ptr(visible) if infill else None,
                                      _lib.ptr(motion_eps) if infill else None, n_win_max, _lib.ptr(traj_eps) if traj else None,
                                      _lib.ptr(out.get('pose')), _lib.ptr(out.get('local_traj')), _lib.ptr(out.get('trans')),
                                      _lib.ptr(out.get('orient')), flags, _lib.ptr(ws), _lib.current_stream()))
        return out


    # -- the infiller inside an optimisation loop (latent-optimisation mode) ----------------------------------------------------------------
    def infill_taped(self, body_pose, visible, lens, motion_eps):
        """The infiller half of infer() with every activation kept: returns (out_pose (B,T,69), tape).  `tape` goes to infill_backward."""
        L = _lib.lib()
        B, T = body_pose.shape[:2]
        dev = body_pose.device
        body_pose, visible, motion_eps = body_pose.float().contiguous(), visible.float().contiguous(), motion_eps.float().contiguous()
        lens_np = np.ascontiguousarray(lens, dtype=np.int32)
        out_pose = torch.empty((B, T, 69), device=dev)
        buf = torch.empty(L.glamr_nets_tape_bytes(self.h, B, T), dtype=torch.uint8, device=dev)
        _lib.check(L.glamr_nets_infill_taped(self.h, B, T, _lib.ptr(lens_np), _lib.ptr(body_pose), _lib.ptr(visible), _lib.ptr(motion_eps), motion_eps.shape[1],
                                             _lib.ptr(out_pose), _lib.ptr(buf), _lib.current_stream()))
        return out_pose, {'buf': buf, 'lens': lens_np, 'eps': motion_eps, 'B': B, 'T': T}

    def infill_backward(self, tape, g_out_pose):
        """dL/d motion_eps (B, n_win_max, 128) for dL/d out_pose (B,T,69) of the taped call."""
        L = _lib.lib()
        eps = tape['eps']
        g_out_pose = g_out_pose.float().contiguous()
        g_eps = torch.empty_like(eps)
        _lib.check(L.glamr_nets_infill_backward(self.h, tape['B'], tape['T'], _lib.ptr(tape['lens']), _lib.ptr(eps), eps.shape[1], _lib.ptr(g_out_pose),
                                                _lib.ptr(g_eps), _lib.ptr(tape['buf']), _lib.current_stream()))
        return g_eps

    # -- training-mode / reconstruction passes (forward(data), inference(recon=True)) -----------------------------------------------------
    def infiller_window(self, mode, in_body_pose, frame_mask, eps=None, body_pose=None, want_context=True):
        """One 50-frame window per sequence through context encoder, (posterior encoder,) prior and decoder.  in_body_pose / body_pose
        (B,50,69), frame_mask (B,50) 1 = visible, eps (B,128).  Returns dict: out_body_pose (B,30,69), p_z (B,2,128), z (B,128)
        [, context (B,50,256), q_z (B,2,128)]."""
        L = _lib.lib()
        B = in_body_pose.shape[0]
        dev = in_body_pose.device
        if tuple(in_body_pose.shape[1:]) != (50, 69):
            raise ValueError('the motion infiller takes windows of 50 frames (past 10 / current 30 / future 10), got %s' % (tuple(in_body_pose.shape),))
        f = lambda t: None if t is None else t.to(dev).float().contiguous()
        in_body_pose, frame_mask, eps, body_pose = f(in_body_pose), f(frame_mask), f(eps), f(body_pose)
        out = {'out_body_pose': torch.empty((B, 30, 69), device=dev), 'p_z': torch.empty((B, 2, NZ), device=dev), 'z': torch.empty((B, NZ), device=dev)}
        if want_context:
            out['context'] = torch.empty((B, 50, 256), device=dev)
        if mode != VAE_INFER:
            out['q_z'] = torch.empty((B, 2, NZ), device=dev)
        io = _lib.InfillerIO(_lib.ptr(in_body_pose), _lib.ptr(body_pose), _lib.ptr(frame_mask), _lib.ptr(eps), _lib.ptr(out.get('context')),
                             _lib.ptr(out.get('q_z')), _lib.ptr(out['p_z']), _lib.ptr(out['z']), _lib.ptr(out['out_body_pose']))
        ws = torch.empty(L.glamr_nets_workspace_bytes(self.h, B, 50), dtype=torch.uint8, device=dev)
        _lib.check(L.glamr_nets_infiller_window(self.h, B, mode, ctypes.byref(io), _lib.ptr(ws), _lib.current_stream()))
        return out

    def traj_clip(self, mode, in_body_pose=None, in_joint_pos=None, trans=None, orient=None, eps=None, valid_len=0, init_row=None,
                  want=('trans', 'orient', 'orient_q')):
        """One clip per sequence through the trajectory predictor.  in_body_pose or in_joint_pos (B,T,69); trans / orient (B,T,3) for the
        posterior encoder and the first output row; eps (B,128).  Returns dict of device tensors (batch-major)."""
        L = _lib.lib()
        src = in_body_pose if in_body_pose is not None else in_joint_pos
        B, T = src.shape[:2]
        dev = src.device
        f = lambda t: None if t is None else t.to(dev).float().contiguous()
        in_body_pose, in_joint_pos, trans, orient, eps, init_row = f(in_body_pose), f(in_joint_pos), f(trans), f(orient), f(eps), f(init_row)
        out = {'out_local_traj': torch.empty((B, T, 11), device=dev), 'out_orig_local_traj': torch.empty((B, T, 11), device=dev),
               'p_z': torch.empty((B, 2 * NZ), device=dev), 'z': torch.empty((B, NZ), device=dev)}
        if trans is not None:
            out['local_traj'] = torch.empty((B, T, 11), device=dev)
        if mode != VAE_INFER:
            out['q_z'] = torch.empty((B, 2 * NZ), device=dev)
        for k, w in (('trans', 3), ('orient', 3), ('orient_q', 4)):
            if k in want:
                out['out_' + k] = torch.empty((B, T, w), device=dev)
        io = _lib.TrajIO(_lib.ptr(in_body_pose), _lib.ptr(in_joint_pos), _lib.ptr(trans), _lib.ptr(orient), _lib.ptr(eps), _lib.ptr(init_row), int(valid_len),
                         _lib.ptr(out.get('local_traj')), _lib.ptr(out.get('q_z')), _lib.ptr(out['p_z']), _lib.ptr(out['z']),
                         _lib.ptr(out['out_orig_local_traj']), _lib.ptr(out['out_local_traj']), _lib.ptr(out.get('out_trans')),
                         _lib.ptr(out.get('out_orient')), _lib.ptr(out.get('out_orient_q')))
        ws = torch.empty(L.glamr_nets_workspace_bytes(self.h, B, T), dtype=torch.uint8, device=dev)
        _lib.check(L.glamr_nets_traj_clip(self.h, B, T, mode, ctypes.byref(io), _lib.ptr(ws), _lib.current_stream()))
        return out


def local_to_global(local_traj):
    """traj_local2global_heading on the device: local_traj (B,T,11) -> trans (B,T,3), orient axis-angle (B,T,3), orient quaternion (B,T,4)."""
    L = _lib.lib()
    local_traj = local_traj.float().contiguous()
    B, T = local_traj.shape[:2]
    dev = local_traj.device
    trans, orient, q = torch.empty((B, T, 3), device=dev), torch.empty((B, T, 3), device=dev), torch.empty((B, T, 4), device=dev)
    ws = torch.empty(L.glamr_traj_local_to_global_workspace_bytes(B, T), dtype=torch.uint8, device=dev)
    _lib.check(L.glamr_traj_local_to_global(B, T, _lib.ptr(local_traj), _lib.ptr(trans), _lib.ptr(orient), _lib.ptr(q), _lib.ptr(ws), _lib.current_stream()))
    return trans, orient, q
