"""Register / scratch use of every device kernel in libiouaware_hip.so, from the code objects' own metadata.

    python tools/kernel_resources.py [--all]

The shared library carries one clang offload bundle per translation unit in its `.hip_fatbin` section; each
bundle holds the gfx950 code object (an ELF) whose NT_AMDGPU_METADATA note lists, per kernel, `.vgpr_count`,
`.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size` (scratch bytes per work-item) and LDS.
VERDICT r5 item 6: no shipped kernel may spill (tests/test_kernel_resources.py asserts it on the CPU).
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'iou-aware-single-stage-object-detector_amd', 'csrc', 'libiouaware_hip.so')
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(so=SO):
    """-> list of gfx950 code objects (bytes) found in the library's offload bundles"""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, 'fat.bin')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', so, fat],
                       check=True)
        data = open(fat, 'rb').read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        n, = struct.unpack_from('<Q', data, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx950' in triple and size:
                out.append(data[base + off:base + off + size])
    return out


def kernels(so=SO):
    """-> list of dicts: name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds"""
    rows = []
    for co in code_objects(so):
        with tempfile.NamedTemporaryFile(suffix='.co') as fh:
            fh.write(co)
            fh.flush()
            txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', fh.name], check=True,
                                 capture_output=True, text=True).stdout
        for blk in re.split(r'\n\s*- \.agpr_count:', txt)[1:]:
            blk = '.agpr_count:' + blk

            def num(key, blk=blk):
                m = re.search(r'\.%s:\s+(\d+)' % key, blk)
                return int(m.group(1)) if m else 0
            name = re.search(r'\.name:\s+(\S+)', blk).group(1)
            rows.append(dict(name=name, vgpr=num('vgpr_count'), agpr=num('agpr_count'), sgpr=num('sgpr_count'),
                             vgpr_spill=num('vgpr_spill_count'), sgpr_spill=num('sgpr_spill_count'),
                             scratch=num('private_segment_fixed_size'), lds=num('group_segment_fixed_size')))
    return rows


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True,
                         text=True).stdout.split('\n')
    return dict(zip(names, out))


if __name__ == '__main__':
    rows = kernels()
    dm = demangle([r['name'] for r in rows])
    bad = [r for r in rows if r['vgpr_spill'] or r['sgpr_spill'] or r['scratch']]
    print('%d kernels, %d with spills / scratch' % (len(rows), len(bad)))
    for r in (rows if '--all' in sys.argv else bad):
        print('%-90s vgpr %3d agpr %3d sgpr %3d | vgpr_spill %3d sgpr_spill %3d scratch %4d B | lds %6d'
              % (dm[r['name']][:90], r['vgpr'], r['agpr'], r['sgpr'], r['vgpr_spill'], r['sgpr_spill'],
                 r['scratch'], r['lds']))
