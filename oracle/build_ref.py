"""TEST INFRASTRUCTURE ONLY — compiles the reference's own CPU voxelization extension from the sources
where they lie under /root/reference (nothing is copied into this repo), straight with g++ (the
reference's setup.py / build system is not run).  Output: oracle/_ref/voxel_layer_ref.so (git-ignored;
it travels to the GPU box with the snapshot like our own built .so).

Sources (3 files, torch + pybind11 headers only):
  mmdet3d/ops/voxel/src/voxelization.cpp       pybind module: hard_voxelize, dynamic_voxelize, ...
  mmdet3d/ops/voxel/src/voxelization_cpu.cpp   dynamic_voxelize_cpu / hard_voxelize_cpu
  mmdet3d/ops/voxel/src/scatter_points_cpu.cpp dynamic_point_to_voxel_cpu (legacy, unused by the bindings)
The reference's DynamicScatter (dynamic_point_to_voxel_forward/backward) has NO CPU implementation
(voxelization.h:106,127 "do not support cpu yet"): only dynamic_voxelize / hard_voxelize are usable from
this build; DynamicScatter is restated in oracle/voxel_oracle.py.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, '_ref')
OUT = os.path.join(OUT_DIR, 'voxel_layer_ref.so')
REF_ROOT = os.environ.get('SST_REFERENCE_ROOT', '/root/reference')
SRC_DIR = os.path.join(REF_ROOT, 'mmdet3d', 'ops', 'voxel', 'src')
SOURCES = ['voxelization.cpp', 'voxelization_cpu.cpp', 'scatter_points_cpu.cpp']


def reference_available():
    return all(os.path.exists(os.path.join(SRC_DIR, s)) for s in SOURCES)


def build(force=False, verbose=False):
    """Returns the path of the built module, or None when the reference tree is not present."""
    if os.path.exists(OUT) and not force:
        return OUT
    if not reference_available():
        return None
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    incs = cpp_extension.include_paths() + [sysconfig.get_paths()['include']]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(OUT_DIR, s.replace('.cpp', '.o'))
        objs.append(o)
        cmd = ['g++', '-O2', '-fPIC', '-std=c++17', '-w', f'-D_GLIBCXX_USE_CXX11_ABI={abi}',
               '-DTORCH_EXTENSION_NAME=voxel_layer_ref', '-DTORCH_API_INCLUDE_EXTENSION_H']
        cmd += [f'-I{i}' for i in incs] + ['-c', os.path.join(SRC_DIR, s), '-o', o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('oracle/_ref build failed:\n' + ' '.join(cmd) + '\n' + out[-3000:])
    link = ['g++', '-shared', '-o', OUT] + objs + [f'-L{torch_lib}', '-ltorch', '-ltorch_cpu', '-lc10',
                                                   '-ltorch_python', f'-Wl,-rpath,{torch_lib}']
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('oracle/_ref link failed:\n' + r.stdout + r.stderr)
    for o in objs:
        os.remove(o)
    if verbose:
        print('built', OUT)
    return OUT


def load():
    """Imports oracle/_ref/voxel_layer_ref.so (must have been built); returns the module or None."""
    if not os.path.exists(OUT):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location('voxel_layer_ref', OUT)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# ---- second target: the reference's CPU points-in-boxes routine (box convention pin for the dynamic point pool) ----
PIB_OUT = os.path.join(OUT_DIR, 'points_in_boxes_ref.so')
PIB_SRC = os.path.join(REF_ROOT, 'mmdet3d', 'ops', 'roiaware_pool3d', 'src', 'points_in_boxes_cpu.cpp')
PIB_BINDING = os.path.join(HERE, 'ref_points_in_boxes_binding.cpp')


def build_points_in_boxes(force=False, verbose=False):
    """oracle/_ref/points_in_boxes_ref.so = the reference's points_in_boxes_cpu.cpp (compiled where it lies) +
    oracle/ref_points_in_boxes_binding.cpp (our pybind shim).  None when the reference tree is absent."""
    if os.path.exists(PIB_OUT) and not force:
        return PIB_OUT
    if not os.path.exists(PIB_SRC):
        return None
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    incs = cpp_extension.include_paths() + [sysconfig.get_paths()['include']]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    objs, procs = [], []
    for src in (PIB_SRC, PIB_BINDING):
        o = os.path.join(OUT_DIR, 'pib_' + os.path.basename(src).replace('.cpp', '.o'))
        objs.append(o)
        cmd = ['g++', '-O2', '-fPIC', '-std=c++17', '-w', f'-D_GLIBCXX_USE_CXX11_ABI={abi}',
               '-DTORCH_EXTENSION_NAME=points_in_boxes_ref', '-DTORCH_API_INCLUDE_EXTENSION_H']
        cmd += [f'-I{i}' for i in incs] + ['-c', src, '-o', o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('oracle/_ref build failed:\n' + ' '.join(cmd) + '\n' + out[-3000:])
    link = ['g++', '-shared', '-o', PIB_OUT] + objs + [f'-L{torch_lib}', '-ltorch', '-ltorch_cpu', '-lc10',
                                                       '-ltorch_python', f'-Wl,-rpath,{torch_lib}']
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('oracle/_ref link failed:\n' + r.stdout + r.stderr)
    for o in objs:
        os.remove(o)
    if verbose:
        print('built', PIB_OUT)
    return PIB_OUT


def load_points_in_boxes():
    if not os.path.exists(PIB_OUT):
        return None
    import importlib.util
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location('points_in_boxes_ref', PIB_OUT)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# ---- third target: the CPU rulebook templates of the reference's vendored spconv (header-only, geometry.h) ----
SRB_OUT = os.path.join(OUT_DIR, 'spconv_rulebook_ref.so')
SRB_INC = os.path.join(REF_ROOT, 'mmdet3d', 'ops', 'spconv', 'include')
SRB_BINDING = os.path.join(HERE, 'ref_spconv_rulebook_binding.cpp')
SRB_STUBS = os.path.join(HERE, 'ref_stubs')  # an empty cuda_runtime_api.h: tensorview.h includes it unconditionally


def build_spconv_rulebook(force=False, verbose=False):
    """oracle/_ref/spconv_rulebook_ref.so = oracle/ref_spconv_rulebook_binding.cpp instantiating getIndicePairsConv /
    SubM / DeConv <int, int, 3> from the reference's include/spconv/geometry.h, + the reference's src/maxpool.cc (CPU
    max-pool functors), both compiled where they lie."""
    if os.path.exists(SRB_OUT) and not force:
        return SRB_OUT
    if not os.path.exists(os.path.join(SRB_INC, 'spconv', 'geometry.h')):
        return None
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    incs = cpp_extension.include_paths() + [sysconfig.get_paths()['include'], SRB_STUBS, SRB_INC]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    objs = []
    for src in (SRB_BINDING, os.path.join(REF_ROOT, 'mmdet3d', 'ops', 'spconv', 'src', 'maxpool.cc')):
        obj = os.path.join(OUT_DIR, 'srb_' + os.path.basename(src).rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        cmd = ['g++', '-O2', '-fPIC', '-std=c++17', '-w', f'-D_GLIBCXX_USE_CXX11_ABI={abi}',
               '-DTORCH_EXTENSION_NAME=spconv_rulebook_ref', '-DTORCH_API_INCLUDE_EXTENSION_H']
        cmd += [f'-I{i}' for i in incs] + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('oracle/_ref build failed:\n' + ' '.join(cmd) + '\n' + (r.stdout + r.stderr)[-3000:])
    link = ['g++', '-shared', '-o', SRB_OUT] + objs + [f'-L{torch_lib}', '-ltorch', '-ltorch_cpu', '-lc10',
                                                       '-ltorch_python', f'-Wl,-rpath,{torch_lib}']
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('oracle/_ref link failed:\n' + r.stdout + r.stderr)
    for obj in objs:
        os.remove(obj)
    if verbose:
        print('built', SRB_OUT)
    return SRB_OUT


def load_spconv_rulebook():
    if not os.path.exists(SRB_OUT):
        return None
    import importlib.util
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location('spconv_rulebook_ref', SRB_OUT)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


if __name__ == '__main__':
    p = build(force='--force' in sys.argv, verbose=True)
    print(p if p else 'reference tree not present; nothing built')
    p = build_points_in_boxes(force='--force' in sys.argv, verbose=True)
    print(p if p else 'reference tree not present; points_in_boxes not built')
    p = build_spconv_rulebook(force='--force' in sys.argv, verbose=True)
    print(p if p else 'reference tree not present; spconv rulebook not built')
