"""Development probe: an LDS-free matrix-core kernel (tools/filler_probe.hip) on a second stream while the optimiser stage of 1024
300-frame scenes is resident.  Prints the stage launch alone, the filler alone, and both together (stage first / filler first; one long
filler launch / a train of short ones), so that the sum, the maximum and what was measured can be compared."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch
from glamr_amd import _lib
from glamr_amd.global_recon import packing
from glamr_amd.global_recon.configs import get_config
from glamr_amd.utils import synth
from oracle.port import build
from oracle import make_golden as mg
from tests.grecon_common import j_local_from_oracle


def filler_lib():
    so = os.path.join(HERE, '_filler_probe.so')
    src = os.path.join(HERE, 'filler_probe.hip')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', src, '-o', so])
    lib = ctypes.CDLL(so)
    lib.filler_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def main():
    dev = torch.device('cuda:0')
    root = build.ensure_synthetic_assets('/tmp/glamr_bench_assets')
    cfg = get_config('glamr_dynamic')
    md = synth.make_smpl_model()
    in_dict = synth.make_in_dict(seed=0, num_frames=300, num_persons=1, smpl_model=md)
    ora = build.load_optimizer(root, cfg)
    data = ora.init_data(in_dict, latents=mg.latents_for(in_dict, 0))
    jl = j_local_from_oracle(ora.smpl, data)
    L = _lib.lib()
    F = filler_lib()
    S = int(os.environ.get('GLAMR_PROBE_SCENES', '1024'))
    spec = cfg['opt_stage_specs']['init_opt']
    packed = packing.PackedScenes([data] * S, [jl] * S, dev)
    sd = packing.stage_desc(spec, cfg['grecon_model_specs'], False, niters=500)
    sb = packed.struct()
    ws = torch.empty(L.glamr_grecon_workspace_bytes(packed.S, packed.P, packed.T), dtype=torch.uint8, device=dev)
    nfrag = 4096                                                   # 4096 fragments x 2 KB = 8 MB: L2 / MALL resident
    W = torch.randn(nfrag * 512, device=dev) * 0.05
    out = torch.zeros(1 << 22, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def stage():
        _lib.check(L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, _lib.ptr(ws), ctypes.c_void_p(s1.cuda_stream)))

    def fill(variant, grid, iters, launches, lds):
        for _ in range(launches):
            rc = F.filler_launch(variant, grid, W.data_ptr(), out.data_ptr(), iters, nfrag, lds, ctypes.c_void_p(s2.cuda_stream))
            assert rc == 0, rc

    def timed(fn_first, fn_second):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record(torch.cuda.current_stream())
        s1.wait_event(ev[0])
        s2.wait_event(ev[0])
        for fn, st in (fn_first, fn_second):
            if fn is None:
                continue
            with torch.cuda.stream(st):
                fn()
        e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e1.record(s1)
        e2.record(s2)
        torch.cuda.synchronize()
        return ev[0].elapsed_time(e1), ev[0].elapsed_time(e2)

    stage()
    torch.cuda.synchronize()
    t_stage = min(timed((stage, s1), (None, s2))[0] for _ in range(2))
    print('stage alone (%d scenes, 500 iterations): %.2f ms' % (S, t_stage), flush=True)
    names = {0: 'MPL8', 1: 'MPL8+split', 2: 'MPL16+split(spills)', 3: 'MPL8+split, 4-wave WG', 4: 'MPL2+split', 5: 'MPL8+split, 2-wave WG', 6: 'MPL8+split, 3-wave WG'}
    mfma = {0: 8, 1: 10, 2: 21, 3: 10, 4: 2, 5: 10, 6: 10}
    # (variant, grid in waves, iterations per launch, launches, LDS bytes)
    cases = [(1, 6144, 20000, 1, 0), (1, 6144, 150, 130, 0), (0, 6144, 20000, 1, 0), (3, 6144, 20000, 1, 0), (3, 6144, 20000, 1, 4096),
             (4, 6144, 40000, 1, 0), (1, 1536, 80000, 1, 0), (1, 24576, 5000, 1, 0), (1, 6144, 20000, 1, 1024)]
    if os.environ.get('GLAMR_PROBE_CASES'):
        cases = [tuple(int(v) for v in c.split(',')) for c in os.environ['GLAMR_PROBE_CASES'].split(';')]
    for (v, grid, iters, launches, lds) in cases:
        f = lambda: fill(v, grid, iters, launches, lds)
        f()
        torch.cuda.synchronize()
        t_fill = min(timed((None, s1), (f, s2))[1] for _ in range(2))
        a = timed((stage, s1), (f, s2))
        b = timed((f, s2), (stage, s1))
        tf = mfma[v] * 16384.0 * 64 / 64 * grid * iters * launches / 1e12      # TFLOP of the filler (16x16x32: 16 384 flop per instruction)
        print('%-24s grid %5d x %6d it x %3d launches lds %4d | alone %.2f ms (%.0f TF/s) | stage first: stage %.2f filler %.2f | filler first: stage %.2f filler %.2f | sum %.2f'
              % (names[v], grid, iters, launches, lds, t_fill, tf / t_fill * 1e3, a[0], a[1], b[0], b[1], t_stage + t_fill), flush=True)


if __name__ == '__main__':
    main()
