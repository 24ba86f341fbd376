"""Build libiouaware_hip.so (gfx950) in-tree with hipcc.

    python iou-aware-single-stage-object-detector_amd/csrc/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off and the
(default) correctly rounded fp32 divide/sqrt are part of the numerical
contract of ia_math.hpp; do not add -ffast-math.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['capi.hip', 'decode.hip', 'select.hip', 'nms.hip', 'loss.hip', 'elementwise.hip', 'assign.hip',
           'softnms.hip', 'preproc.hip', 'wino.hip', 'gemm.hip', 'lazynms.hip', 'headloss.hip', 'gconv.hip',
           'trainops.hip', 'bignms.hip', 'conv3x3_bf16.hip', 'conv1x1_stream.hip', 'im2col.hip', 'nms64.hip', 'stem.hip']
HEADERS = ['ia_math.hpp', 'ia_block.hpp', 'ia_internal.hpp', 'ia_loss.hpp', 'ia_rowmax_dev.hpp', 'ia_gather_dev.hpp', 'ia_nms.hpp', 'ia_conv3.hpp', '../../include/iouaware.h']
OUT = os.path.join(HERE, 'libiouaware_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fhip-fp32-correctly-rounded-divide-sqrt', '-fno-gpu-rdc',
         '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace('.hip', '.o'))
        cmd = [_hipcc()] + FLAGS + ['-c', os.path.join(HERE, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    # gemm.hip calls hipBLASLt (library GEMM with fused epilogue); inside a torch process the
    # loader resolves libhipblaslt.so.1 to the copy torch already mapped
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs + [
        '-L/opt/rocm/lib', '-lhipblaslt', '-Wl,-rpath,/opt/rocm/lib']
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
