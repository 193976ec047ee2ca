"""Same-box A/B of two builds of the library: run bench.py's main() against the shared object given as argv[1] (the rest of argv goes
to bench.py).  Used for in-box comparisons, since two gpurun boxes differ by ~1 %."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
lib_path = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
from ming_univision_amd import _lib
_lib.LIB_PATH = lib_path
import bench
bench.main()
