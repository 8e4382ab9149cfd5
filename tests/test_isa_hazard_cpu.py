"""The generated gfx950 code must not contain the packed-fp32 form that misreads an
operand next to bf16 MFMAs (Makefile: ``-fno-slp-vectorize``; DESIGN.md section 5;
``tools/mfma_valu_hazard.hip`` reproduces the hazard on the GPU): VOP3P
``v_pk_fma_f32`` / ``v_pk_mul_f32`` / ``v_pk_add_f32`` with OP_SEL[1] = 1, i.e. the
LOW result lane reading the HIGH half of the second source.  hipcc cross-compiles the
device code of every source file to assembly here (no GPU needed)."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'garage_amd', 'csrc')
HIPCC = '/opt/rocm/bin/hipcc'
HAZARD = re.compile(r'v_pk_(fma|mul|add)_f32\b.*\bop_sel:\[[01],1')


def _makefile_flags():
    text = open(os.path.join(ROOT, 'Makefile')).read()
    line = [l for l in text.split('\n') if l.startswith('FLAGS :=')][0]
    flags = line.split(':=', 1)[1].replace('$(ARCH)', 'gfx950').replace('$(EXTRA)', '')
    return [f for f in flags.split() if f not in ('-fPIC', )]


def _asm(path, tmp):
    out = os.path.join(tmp, os.path.basename(path) + '.s')
    cmd = [HIPCC] + _makefile_flags() + ['--cuda-device-only', '-S', '-x', 'hip', path,
                                         '-o', out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return path, open(out).read()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_no_packed_fp32_instruction_reads_the_high_half_of_src1(tmp_path):
    assert '-fno-slp-vectorize' in _makefile_flags()
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith('.hip'))
    with ThreadPoolExecutor(max_workers=6) as ex:
        results = list(ex.map(lambda s: _asm(s, str(tmp_path)), srcs))
    bad = []
    for path, text in results:
        assert 's_endpgm' in text, path  # device code was generated
        kernel = '?'
        for line in text.split('\n'):
            if line.startswith('_Z') and line.rstrip().endswith(':') is False and ':' in line:
                kernel = line.split(':')[0]
            if HAZARD.search(line):
                bad.append((os.path.basename(path), kernel[:80], line.strip()))
    assert not bad, bad[:10]


def test_the_pattern_catches_the_forms_the_reproducer_found():
    hit = ['v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0]',
           'v_pk_mul_f32 v[14:15], v[14:15], s[0:1] op_sel:[0,1]',
           'v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]',
           'v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,1,0] op_sel_hi:[1,0,1]']
    miss = ['v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel_hi:[1,0,1]',
            'v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,0,0]',
            'v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,0,1]',
            'v_pk_mul_f32 v[0:1], v[2:3], v[4:5]',
            'v_pk_add_f32 v[0:1], s[8:9], v[2:3] op_sel:[1,0]']
    assert all(HAZARD.search(l) for l in hit)
    assert not any(HAZARD.search(l) for l in miss)
