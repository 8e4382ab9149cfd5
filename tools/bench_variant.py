"""Developer tool: run bench.py against a variant build of the library
(tools/build_variants.sh):  python tools/bench_variant.py <lib.so> [bench args]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from garage_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
