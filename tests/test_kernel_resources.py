"""CPU: no shipped kernel of this build spills.

The code objects inside libiouaware_hip.so carry per-kernel metadata (NT_AMDGPU_METADATA); tools/kernel_resources.py
reads it with llvm-readelf.  Every `ia::` kernel must have `.vgpr_spill_count == 0` and
`.private_segment_fixed_size == 0` (no scratch at all: no spill, no dynamically indexed private array).  Kernels
instantiated from rocPRIM (the radix sort behind ia_nms_f64 / the big-n NMS route) are library code and are listed,
not gated.  SGPR spills go to VGPR lanes (v_writelane), not to memory: reported, capped."""
import importlib.util
import os

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _tool():
    spec = importlib.util.spec_from_file_location('kernel_resources', os.path.join(ROOT, 'tools', 'kernel_resources.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope='module')
def rows():
    import __graft_entry__ as g
    g.build()
    t = _tool()
    if not os.path.exists(os.path.join(t.LLVM, 'llvm-readelf')):
        pytest.skip('no llvm-readelf in this image')
    rows = t.kernels()
    names = t.demangle([r['name'] for r in rows])
    for r in rows:
        r['pretty'] = names[r['name']]
    return rows


def test_every_translation_unit_contributes_kernels(rows):
    own = [r for r in rows if 'ia::' in r['pretty']]
    assert len(own) >= 140, len(own)                       # 20 translation units with device code


def test_no_own_kernel_spills_or_uses_scratch(rows):
    own = [r for r in rows if 'ia::' in r['pretty']]
    bad = [(r['pretty'], r['vgpr_spill'], r['scratch']) for r in own if r['vgpr_spill'] or r['scratch']]
    assert not bad, bad


def test_sgpr_spills_are_few(rows):
    own = [r for r in rows if 'ia::' in r['pretty'] and r['sgpr_spill']]
    # (lane writes into a VGPR, no memory traffic; k_wino_out<true> holds the per-level tables of 5 levels + 8
    # segments in SGPRs)
    assert all(r['sgpr_spill'] <= 24 for r in own), [(r['pretty'], r['sgpr_spill']) for r in own]
    assert len(own) <= 2, [(r['pretty'], r['sgpr_spill']) for r in own]
