"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the entry points the header declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'glamr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(glamr_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    from glamr_amd import _lib
    assert _declared() == _lib.exported_symbols()


def test_library_builds_and_exports_every_declared_symbol():
    from glamr_amd import build, _lib
    build.build_library()
    L = _lib.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert L.glamr_version() >= 100
    assert isinstance(L.glamr_last_error(), bytes)


def test_struct_sizes_match_the_header():
    """ctypes mirrors of the ABI structs must have the C layout (checked against a gcc-compiled probe)."""
    import subprocess
    import tempfile
    from glamr_amd import _lib
    code = '#include <stdio.h>\n#include "glamr_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(glamr_tensor_desc), ' \
           'sizeof(glamr_stage_desc), sizeof(glamr_scene_batch), sizeof(glamr_param_layout), sizeof(glamr_raw_batch), ' \
           'sizeof(glamr_person_arrays));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'p.c'), 'w').write(code)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 'p.c'), '-o', os.path.join(d, 'p')])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, 'p')]).split()]
    assert sizes == [ctypes.sizeof(_lib.TensorDesc), ctypes.sizeof(_lib.StageDesc), ctypes.sizeof(_lib.SceneBatch), ctypes.sizeof(_lib.ParamLayout),
                     ctypes.sizeof(_lib.RawBatch), ctypes.sizeof(_lib.PersonArrays)]


def test_product_never_imports_the_oracle():
    """The product package must not route through the checker (oracle/) or any CPU fallback."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'glamr_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_entry_points_reject_bad_arguments_without_touching_the_gpu():
    """Error convention of the boundary (include/glamr_hip.h): a negative GLAMR_E_* code, never an exception or an exit, and a
    thread-local message.  These checks run before any HIP call, so they hold on a box without a GPU."""
    from glamr_amd import build, _lib
    build.build_library()
    L = _lib.lib()
    E_INVALID = -1
    # parameter layout / workspace size
    lay = _lib.ParamLayout()
    assert L.glamr_grecon_param_layout(0, 300, ctypes.byref(lay)) == E_INVALID
    assert L.glamr_grecon_param_layout(33, 300, ctypes.byref(lay)) == E_INVALID
    assert L.glamr_grecon_param_layout(9, 300, ctypes.byref(lay)) == 0            # more than 8 persons: csrc/grecon_wide.hip
    assert L.glamr_grecon_param_layout(1, 1, ctypes.byref(lay)) == E_INVALID
    assert b'max_persons' in L.glamr_last_error()
    assert L.glamr_grecon_param_layout(1, 300, ctypes.byref(lay)) == 0
    # the layout holds every variable group of every stage: camera 6T + 3T, camera residuals 6T + 3T, then one block per person
    assert lay.person0 == 18 * 300 and lay.scene_stride == lay.person0 + lay.person_stride and lay.world_dheading + 300 == lay.person_stride
    assert L.glamr_grecon_workspace_bytes(0, 1, 300) == 0 and L.glamr_grecon_workspace_bytes(4, 33, 300) == 0
    assert L.glamr_grecon_workspace_bytes(4, 12, 300) > 12 * L.glamr_grecon_workspace_bytes(4, 1, 300) // 2      # (the wide instances' workspace)
    assert L.glamr_grecon_workspace_bytes(4, 1, 300) > 4 * 3 * lay.scene_stride * 4      # Adam moments + gradient per scene at least
    # stage launch: null pointers, bad geometry, missing arrays, too long a sequence
    sd, sb = _lib.StageDesc(), _lib.SceneBatch()
    assert L.glamr_grecon_run_stage(None, ctypes.byref(sd), None, None, None) == E_INVALID
    assert L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, ctypes.c_void_p(16), None) == E_INVALID
    assert b'bad batch geometry' in L.glamr_last_error()
    sb.n_scenes, sb.max_persons, sb.max_len, sb.n_joints = 1, 1, 300, 26
    assert L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, ctypes.c_void_p(16), None) == E_INVALID
    assert b'NULL' in L.glamr_last_error()
    for name, _ in _lib.SceneBatch._fields_[4:]:
        setattr(sb, name, 16)                           # any non-null value: the geometry checks come first
    sb.n_joints = 25
    assert L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, ctypes.c_void_p(16), None) == E_INVALID
    assert b'n_joints' in L.glamr_last_error()
    sb.n_joints, sb.max_len = 26, 40000
    assert L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, ctypes.c_void_p(16), None) == E_INVALID
    assert b'exceeds' in L.glamr_last_error()
    sb.max_len, sd.niters = 300, -1
    assert L.glamr_grecon_run_stage(ctypes.byref(sb), ctypes.byref(sd), None, ctypes.c_void_p(16), None) == E_INVALID
    assert b'niters' in L.glamr_last_error()
    # the Python wrapper turns the code into an exception carrying the message
    try:
        _lib.check(L.glamr_grecon_param_layout(0, 300, ctypes.byref(lay)))
    except RuntimeError as e:
        assert 'max_persons' in str(e)
    else:
        raise AssertionError('no exception')
