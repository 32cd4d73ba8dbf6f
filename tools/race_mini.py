"""DEVELOPMENT AID (GPU): the smallest form of the two-stream corruption found so far.
Stream A: the skinning alone (SMPL.root_relative_joints: smpl_prep_kernel, smpl_lbs_kernel, smpl_finish_kernel) on STATIC inputs, plain launches.
Stream B: the priors' launch sequence cut after its k-th part (GLAMR_NETS_PROBE_STOP of a probe build, tools/build_variant_files.sh spin nets.hip
-DGLAMR_RACE_PROBE), plain launches, started `delay` microseconds into A's call.  Every A result against A alone, bit for bit.
usage: GLAMR_LIB_PATH=tools/_lib_spin.so python tools/race_mini.py [stop position, 0 = whole priors] [runs] [B kind: nets | mm | none]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from glamr_amd.utils import synth
from glamr_amd.models.prior_models import num_windows, NZ

dev = torch.device('cuda:0')
stop = int(sys.argv[1]) if len(sys.argv) > 1 else 5
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
kind = sys.argv[3] if len(sys.argv) > 3 else 'nets'
S, T = 1024, 300
model = bench.build_model(bench.ensure_assets(), dev)
g = torch.Generator(device=dev).manual_seed(3)
pose = 0.3 * torch.randn((S * T, 69), device=dev, generator=g)
betas = 0.5 * torch.randn((S * T, 10), device=dev, generator=g)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
with torch.cuda.stream(sa):
    ref = model.smpl.root_relative_joints(pose, betas).clone()
    ref2 = model.smpl.root_relative_joints(pose, betas).clone()
torch.cuda.synchronize()
assert torch.equal(ref, ref2)
nw = num_windows(T)
h = model.mt_model.handle
with torch.cuda.stream(sb):
    rs = h.resident_set(S, T, nw)
    rs['nets_pose'].copy_(0.2 * torch.randn((S, T, 69), device=dev, generator=g))
    rs['nets_vis'].fill_(1.0)
    rs['nets_vis'][:, 100:160] = 0.0
    rs['meps'].copy_(torch.randn((S, nw, NZ), device=dev, generator=g))
    rs['teps'].copy_(torch.randn((S, NZ), device=dev, generator=g))
    rs['persistent'] = True
lens = np.full(S, T, np.int32)
lens_dev = torch.zeros(S, dtype=torch.int32, device=dev)
gx = 0.3 * torch.randn(51200, 96, device=dev); grb = torch.randn(50, 256, device=dev); gws = (0.1 * torch.randn(2 * 256 * 96, device=dev)).half(); gy = torch.empty(51200, 256, device=dev)
xa = torch.randn(8192, 8192, device=dev); xb = torch.randn(8192, 8192, device=dev); xc = torch.empty(8192, 8192, device=dev)
torch.cuda.synchronize()
if stop:
    os.environ['GLAMR_NETS_PROBE_STOP'] = str(stop)


STATE = {}


def b_work():
    if kind == 'nets':
        model.mt_model.infer_padded(rs['nets_pose'], rs['nets_vis'], lens, rs['meps'], rs['teps'], buffers=rs, coschedule=True)
    elif kind == 'mm':
        torch.mm(xa, xb, out=xc)
    elif kind == 'lin':      # the first GEMM with the handle's own weights (GLAMR_MINI_LIN=0), or with COPIES of them in torch arrays (=1)
        import ctypes
        from glamr_amd import _lib
        fn = _lib.lib().glamr_debug_first_lin
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        which = int(os.environ.get('GLAMR_MINI_LIN', '0'))
        if which and 'copied' not in STATE:
            fn(h.h, gx.data_ptr(), gy.data_ptr(), 51200, gws.data_ptr(), grb.data_ptr(), 2, _lib.current_stream())
            STATE['copied'] = True
        X, Y = gx.data_ptr(), gy.data_ptr()
        place = os.environ.get('GLAMR_MINI_PLACE', '')      # 'x' / 'y' / 'xy': that operand at the address the priors' own call uses (inside its workspace)
        if place:
            ox, oh = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.lib().glamr_debug_ws_offsets(S, T, ctypes.byref(ox), ctypes.byref(oh))
            if 'x' in place:
                X = rs['ws'].data_ptr() + ox.value
            if 'y' in place:
                Y = rs['ws'].data_ptr() + oh.value
            if 'printed' not in STATE:
                STATE['printed'] = print('first lin operands: X %#x Y %#x (workspace %#x + %d / + %d)' % (X, Y, rs['ws'].data_ptr(), ox.value, oh.value))
        fn(h.h, X, Y, 51200, gws.data_ptr(), grb.data_ptr(), which & 1, _lib.current_stream())
    elif kind == 'gemm':      # one launch of gemm_free_kernel<6, 2, 1, 4> on random arrays, nothing else of the priors
        import ctypes
        from glamr_amd import _lib
        fn = _lib.lib().glamr_debug_gemm_free
        fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
        if os.environ.get('GLAMR_MINI_H2D'):      # a small pageable host -> device copy ahead of it, like the upload of the sequence lengths
            lens_dev.copy_(torch.from_numpy(lens), non_blocking=True)
        X, Y = (gx, gy) if not os.environ.get('GLAMR_MINI_WS_BUFFERS') else (rs['ws'].view(torch.float32)[:51200 * 96], rs['ws'].view(torch.float32)[64 << 20:(64 << 20) + 51200 * 256])
        for _ in range(int(os.environ.get('GLAMR_MINI_GEMMS', '1'))):
            fn(X.data_ptr(), grb.data_ptr(), gws.data_ptr(), Y.data_ptr(), 51200, _lib.current_stream())


with torch.cuda.stream(sb):
    b_work(); b_work()
torch.cuda.synchronize()
if os.environ.get('GLAMR_MINI_NO_LENS'):
    os.environ['GLAMR_NETS_PROBE_NO_LENS'] = '1'
if os.environ.get('GLAMR_MINI_SKIP'):      # from here on B's call leaves out its small kernels (bit 0 pose_in, bit 1 window_gather); new graph-cache key through the flags
    os.environ['GLAMR_NETS_PROBE_SKIP'] = os.environ['GLAMR_MINI_SKIP']
    rs['persistent'] = False
if os.environ.get('GLAMR_MINI_ZERO_WS'):   # the GEMM's operand rows all zero / all ones
    with torch.cuda.stream(sb):
        rs['ws'].view(torch.float32)[:] = float(os.environ['GLAMR_MINI_ZERO_WS'])
    torch.cuda.synchronize()
n_bad = 0
for it in range(runs):
    delay = 20 + 40 * (it % 12)
    with torch.cuda.stream(sb):
        torch.cuda._sleep(int(delay * 2100))
        b_work()
    with torch.cuda.stream(sa):
        got = model.smpl.root_relative_joints(pose, betas)
    torch.cuda.synchronize()
    d = (got != ref).flatten(1).any(dim=1)
    nb = int(d.sum())
    n_bad += nb > 0
    print('run %d (B %d us after A): %d of %d frames differ%s' % (it, delay, nb, S * T, (', first ' + str(torch.nonzero(d).flatten()[:8].tolist())) if nb else ''))
print('SUMMARY: %d of %d runs differ from the skinning alone (B = %s%s)' % (n_bad, runs, kind, (', cut at position %d' % stop) if stop and kind == 'nets' else ''))
