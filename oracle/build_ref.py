"""TEST INFRASTRUCTURE ONLY.  Build the REAL reference NMS (C++) into oracle/_ref/.

Source: /root/reference/mmdet/ops/nms/src/nms_cpu.cpp, compiled from where it
lies (never copied into this repo).  torch >= 2 removed the
`AT_DISPATCH_FLOATING_TYPES(DeprecatedTypeProperties, ...)` overload the file
uses at line 63, so ONE token is rewritten in the compiler's input stream
(`dets.type()` -> `dets.scalar_type()` on that dispatch line); the patched
text only ever exists in a pipe / the git-ignored oracle/_ref/ build dir.

Output: oracle/_ref/nms_cpu_ref.so -- a pybind11 torch extension exposing
`nms(dets: Tensor(n,5), threshold: float) -> LongTensor` (nms_cpu.cpp:61-71).
It travels to the GPU box with the snapshot (git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
import sysconfig

REF_SRC = '/root/reference/mmdet/ops/nms/src/nms_cpu.cpp'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, '_ref')
OUT = os.path.join(OUT_DIR, 'nms_cpu_ref.so')


def build(force=False):
    if not os.path.exists(REF_SRC):
        return None                     # GPU box: use the prebuilt file if any
    if os.path.exists(OUT) and not force and \
            os.path.getmtime(OUT) >= os.path.getmtime(REF_SRC):
        return OUT
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(REF_SRC) as f:
        text = f.read()
    patched = text.replace('AT_DISPATCH_FLOATING_TYPES(dets.type()',
                           'AT_DISPATCH_FLOATING_TYPES(dets.scalar_type()')
    assert patched != text, 'reference source changed; patch point not found'
    inc = []
    for p in cpp_extension.include_paths():
        inc += ['-isystem', p]
    inc += ['-isystem', sysconfig.get_paths()['include']]
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = ['g++', '-x', 'c++', '-', '-O2', '-std=c++17', '-fPIC', '-shared', '-w',
           '-DTORCH_EXTENSION_NAME=nms_cpu_ref', '-DTORCH_API_INCLUDE_EXTENSION_H',
           '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc + [
           '-L' + libdir, '-lc10', '-ltorch', '-ltorch_cpu', '-ltorch_python',
           '-Wl,-rpath,' + libdir, '-o', OUT]
    subprocess.run(cmd, input=patched.encode(), check=True)
    return OUT


SOFT_SRC = '/root/reference/mmdet/ops/nms/src/soft_nms_cpu.pyx'
# NOT under oracle/_ref/: that directory travels to the GPU box, and the reference's Python (the
# .pyx is Python-side source) never travels in any form.  oracle/_ref_local/ is listed in
# .gitignore AND .gpurunignore; the committed tests/golden/soft_nms.npz is the pin on the GPU box.
SOFT_DIR = os.path.join(HERE, '_ref_local')
SOFT_OUT = os.path.join(SOFT_DIR, 'soft_nms_cpu.so')


def build_soft(force=False):
    """The reference's own soft-NMS (Cython) -> oracle/_ref_local/soft_nms_cpu.so (build container
    only, never shipped).  The .pyx is translated from where it lies; the generated C only exists
    in a temporary directory."""
    if not os.path.exists(SOFT_SRC):
        return None
    if os.path.exists(SOFT_OUT) and not force and \
            os.path.getmtime(SOFT_OUT) >= os.path.getmtime(SOFT_SRC):
        return SOFT_OUT
    import tempfile
    import numpy
    os.makedirs(SOFT_DIR, exist_ok=True)
    stale = os.path.join(OUT_DIR, 'soft_nms_cpu.so')        # location of rounds <= 4
    if os.path.exists(stale):
        os.remove(stale)
    with tempfile.TemporaryDirectory() as tmp:
        c_file = os.path.join(tmp, 'soft_nms_cpu.c')
        subprocess.run([sys.executable, '-m', 'cython', '-3', SOFT_SRC, '-o', c_file], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.run(['gcc', '-O2', '-fPIC', '-shared', '-w', c_file, '-o', SOFT_OUT,
                        '-I' + numpy.get_include(), '-I' + sysconfig.get_paths()['include']],
                       check=True)
    return SOFT_OUT


def load_soft():
    """the reference module exposing soft_nms_cpu(boxes, iou_thr, method, sigma, min_score)"""
    if not os.path.exists(SOFT_OUT):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location('soft_nms_cpu', SOFT_OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load():
    """Import the built module (None when it is not available)."""
    if not os.path.exists(OUT):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location('nms_cpu_ref', OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    for fn in (build, build_soft):
        out = fn(force='--force' in sys.argv)
        print('built' if out else 'reference not present; skipped', out or '')
