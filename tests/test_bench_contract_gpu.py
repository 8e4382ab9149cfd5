"""bench.py prints ONE JSON line with the contract's keys (driver-facing)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_line_has_the_contract_fields():
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1',
         '--warmup', '1', '--cpu-envs', '8'],
        cwd=ROOT, capture_output=True, text=True, timeout=550)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup',
                'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == 1 and line['warmup'] == 1
    assert line['scaling'] == 'weak' and line['vs_baseline'] is None
    # (GARAGE_AMD_SPLIT_BF16=1 in the environment -- the whole suite run with the opt-in
    # split-operand k-loops -- spells the arithmetic out instead)
    assert line['dtype'] == 'f32' or os.environ.get('GARAGE_AMD_SPLIT_BF16') == '1'
    assert line['data'] == 'synthetic'
    assert line['value'] > 0 and line['higher_is_better'] is True
    w = line['config']['workload']
    assert '{' not in w and 'HalfCheetah' in w and 'PPO E=10 x 32' in w, w
    r = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s'
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    c = line['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in c, key
    assert c['kind'] == 'port' and c['value'] > 0
    g = line['roofline_gae_scan']
    assert g['bound'] == 'hbm' and g['unit'] == 'GB/s'
    assert line['grad_allreduce'].startswith('none') and \
        line['rccl_ranks'] is None


@pytest.mark.timeout(900)
def test_bench_gpus_2_starts_two_ranks_by_itself():
    """The driver's one-line command for N > 1 without torchrun: the parent spawns
    the ranks before touching the GPU and relays rank 0's line.  On a one-GPU box
    the two ranks share the device and exchange over gloo (the RCCL path needs two
    GPUs), which the line must say -- the fallback is never silent."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['GARAGE_AMD_BACKEND'] = 'gloo'
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
         '--config', 'c1', '--steps', '1', '--warmup', '1', '--cpu-envs', '0',
         '--no-roofline'],
        cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['parallelism'] == 'dp2'
    assert 'FALLBACK' in line['grad_allreduce'] and 'gloo' in line['grad_allreduce']
    assert line['rccl_ranks'] is None
    assert line['value'] > 0
