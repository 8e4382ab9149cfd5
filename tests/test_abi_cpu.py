"""CPU checks of the boundary: the library loads and exports every symbol the
header declares; the product never imports the oracle."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'garage_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ga_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from garage_amd import _lib
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    # the ctypes table and the header must describe the same set
    assert sorted(_lib.SIGNATURES) == names
    assert lib.ga_abi_version() == 4


def test_argument_errors_are_reported_without_a_gpu():
    from garage_amd import _lib
    with pytest.raises(_lib.GarageAmdError) as e:
        _lib.call('ga_gae_scan_f32', None, None, None, None, None, 1, 1, 1, 1,
                  0, 1, 0.99, 0.97, 0.0, 0.0, None, None, None)
    assert 'null pointer' in str(e.value)


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'garage_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                src = open(os.path.join(base, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_host_loops_under_address_sanitizer():
    """``make asan-host``: the C++ epoch and rollout loops (update.cpp,
    rollout_loop.cpp) compiled with ``-fsanitize=address,undefined`` for the CPU
    and run against recording fakes of every kernel entry point
    (tests/host/update_loop_harness.cpp): minibatch ranges, the data-parallel even
    split and per-step scales, phase 1, the interleaving of two passes, argument
    errors, the fused step's scratch layout and regions, the rollout loop's buffer
    ping-pong.  GPU sanitizers are not available on this pool."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(['make', '-C', root, 'asan-host'], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'host loops ok' in out.stdout


def test_driver_build_entry_point():
    """``__graft_entry__.build()`` -- what the driver runs on a CPU-only machine --
    compiles (a no-op when the library is current), imports the package and agrees
    with the loader on the ABI version."""
    import __graft_entry__ as entry
    from garage_amd import _lib
    entry.build()
    assert _lib.load().ga_abi_version() == _lib.ABI_VERSION
