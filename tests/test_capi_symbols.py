"""The C-ABI library builds, loads, and exports every symbol include/sst_amd.h declares (no compute)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'sst_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sst_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_path():
    names = _declared_symbols()
    for required in ('sst_dynamic_voxelize_f32', 'sst_unique_rows', 'sst_segment_reduce_fwd_f32',
                     'sst_segment_reduce_bwd_f32', 'sst_window_coors', 'sst_region_batching',
                     'sst_sra_attn_fwd_f32', 'sst_sra_attn_bwd_f32', 'sst_ingroup_rank_i64'):
        assert required in names


def test_library_exports_every_declared_symbol():
    from sst_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared_symbols() if not hasattr(lib, n)]
    assert not missing, f'declared in include/sst_amd.h but not exported: {missing}'


def test_python_binding_covers_every_declared_symbol():
    from sst_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared_symbols()
    lib = _lib.load()
    assert _lib.version().startswith('sst_amd')
    assert lib.sst_scan_workspace_bytes(1000) > 0


def test_ops_fail_loudly_without_gpu_tensors():
    import pytest
    import torch
    import sst_amd
    pts = torch.zeros(10, 3)
    with pytest.raises(RuntimeError):
        sst_amd.voxelization(pts, [0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], -1, -1)
    with pytest.raises(RuntimeError):
        sst_amd.dynamic_point_to_voxel_forward(torch.zeros(4, 3), torch.zeros(4, 3, dtype=torch.int32), 'max')
    with pytest.raises(RuntimeError):
        sst_amd.get_inner_win_inds(torch.zeros(4, dtype=torch.long))


def test_grid_helper_matches_reference_ceil():
    from sst_amd import kernels as K
    assert K.voxel_grid([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4]) == [468, 468, 1]
    assert K.voxel_grid([0.25, 0.25, 0.2], [-80, -80, -2, 80, 80, 4]) == [640, 640, 30]


def test_bench_cli_contract_and_workloads_registry():
    """bench.py's command line (the driver's contract: --gpus / --steps / --warmup, defaults N = 1) and the workload
    registry, without a GPU"""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--help'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ('--gpus', '--steps', '--warmup', '--workload', '--precision', '--no-cpu-baseline'):
        assert flag in out.stdout
    for w in ('sst_bs2', 'sst_bev', 'fsd', 'fsdv2'):
        assert w in out.stdout
    sys.path.insert(0, root)
    import bench_workloads
    assert set(bench_workloads.WORKLOADS) == {'fsd', 'fsdv2'}
    for spec in bench_workloads.WORKLOADS.values():
        assert {'cls', 'points', 'metric', 'name'} <= set(spec)


def test_bf16_shadow_slots_follow_the_parameter_object():
    """the bf16 copies of the fp32 master weights (host logic, no GPU): a slot belongs to one parameter OBJECT, is never
    handed to another parameter that reuses a dead one's id, is dropped with the parameter, and making a copy of a CPU
    parameter fails loudly (the copies are made by a kernel: tests/test_gpu_bf16.py)"""
    import gc
    import pytest
    import torch
    from sst_amd import bf16
    p = torch.nn.Parameter(torch.randn(6, 4))
    slot = bf16._shadow_slot(p)
    assert bf16._shadow_slot(p) is slot and slot == {}
    slot[(None, False)] = 'marker'
    before = len(bf16._shadows)
    del p
    gc.collect()
    assert len(bf16._shadows) == before - 1
    for _ in range(20):   # a new parameter never sees another one's slot, whatever id / address it lands on
        q = torch.nn.Parameter(torch.randn(6, 4))
        assert bf16._shadow_slot(q) == {}
        bf16._shadow_slot(q)[(None, False)] = 'marker'
        del q
    q = torch.nn.Parameter(torch.randn(6, 4))
    with pytest.raises(RuntimeError):
        bf16.shadow(q)
    bf16.invalidate_shadows()
    assert len(bf16._shadows) == 0



def test_header_is_plain_c_and_cpp(tmp_path):
    """include/sst_amd.h is the drop-in boundary: it must compile on its own as C99 and as C++ (extern "C", plain pointers and
    sizes, the argument structs of the layer entry points), with nothing but <stdint.h> behind it"""
    import shutil
    import subprocess
    src = tmp_path / 'hdr.c'
    src.write_text('#include "sst_amd.h"\nint main(void) { return (int)sizeof(sst_encoder_layer_fwd_args) == 0; }\n')
    inc = os.path.join(ROOT, 'include')
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-fsyntax-only', '-I', inc, str(src)], check=True)
    subprocess.run(['g++', '-std=c++17', '-Wall', '-Werror', '-fsyntax-only', '-I', inc, '-x', 'c++', str(src)], check=True)
