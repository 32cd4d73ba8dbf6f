"""MI355X: the workgroup prefix / suffix sums of the optimiser stage (glamr_amd/csrc/block_rt.hpp DeviceRT::scan_multi -- DPP row shifts by
default, ds_bpermute shuffles with -DGLAMR_SCAN_SHUFFLE) against a sequential sum on integer-valued floats, where every summation order must
give the same bits: lengths 1..700 (partial waves, several chunks), 1 / 2 / 4 channels, strides 1 and 2, both directions, arrays in LDS and in
global memory (tools/scan_probe.hip: 320 cases)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('variant', ['dpp', 'shuffle'])
def test_block_scans_are_exact_on_integer_data(tmp_path, variant):
    exe = str(tmp_path / ('scan_probe_' + variant))
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-result', os.path.join(ROOT, 'tools', 'scan_probe.hip'), '-o', exe]
    if variant == 'shuffle':
        cmd.insert(1, '-DGLAMR_SCAN_SHUFFLE')
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert '320 cases, 0 with mismatches' in out.stdout
    # DeviceRT::scan_regs (round 5: scan operands and results in registers, compile-time wave count): exact on integers and bit-identical to
    # scan_multi on arbitrary floats, in both summation orders of the wave part
    assert 'scan_regs: ' in out.stdout and ' cases, 0 with mismatches' in out.stdout.split('scan_regs: ')[1], out.stdout[-2000:]
