"""Builds libsc2amd.so (HIP kernels + C-ABI, gfx950 only) in-tree with hipcc.

The shared library is the product: ``sc2-benchmark_amd/libsc2amd.so`` travels with the source tree
(git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libsc2amd.so')
# (the implicit-GEMM templates of conv_igemm_impl.h are instantiated in eight translation units so that they compile in
#  parallel: the slowest ones first)
SOURCES = ['conv_inst_e.hip', 'conv_inst_f.hip', 'conv_inst_g.hip', 'conv_inst_h.hip', 'conv_inst_a.hip', 'conv_inst_b.hip',
           'conv_inst_c.hip', 'conv_inst_d.hip', 'conv_dec_persist.hip', 'conv_igemm.hip', 'conv_gdn512.hip', 'gdn512_rows.hip', 'bn.hip', 'gdn96_strips.hip', 'conv0_gdn96.hip', 'conv2_gdn48.hip', 'conv2x2_c48.hip', 'conv1x1_stream.hip', 'conv1x1_pair.hip', 'conv1x1_kres.hip', 'conv3x3_win.hip', 'conv1x1_win.hip', 'conv2x2_win.hip',
           'conv_f32.hip', 'loss.hip', 'conv_wgrad.hip', 'gdn_bwd.hip', 'entropy.hip', 'gaussian.hip', 'rans.hip', 'layout.hip', 'diag.hip', 'abi.cpp', 'cdf_host.cpp', 'rans_host.cpp']
HEADERS = [os.path.join(CSRC, 'sc2_common.h'), os.path.join(CSRC, 'conv_igemm_impl.h'),
           os.path.join(HERE, '..', 'include', 'sc2_bottleneck.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-D__HIP_PLATFORM_AMD__=1']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + '.o')
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + FLAGS + ['-x', 'hip', '-c', path, '-o', obj]
        subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        subprocess.check_call(cmd)
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
