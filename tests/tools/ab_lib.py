"""Dev tool: run a script of this repo against ANOTHER build of the library (same-box A/B of two kernel versions):

    python tests/tools/ab_lib.py pytorchcv_amd/csrc/ab/libpcv_amd_old.so bench.py --workload resnet50_bs256 --no-cpu-baseline

The product has no such switch: this sets pytorchcv_amd._lib.LIB_PATH before the library is first opened and then runs the script.
"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
assert os.path.exists(lib), lib
from pytorchcv_amd import _lib
_lib.LIB_PATH = lib
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
