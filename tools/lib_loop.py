"""Diagnostic: tools/bench_train_loop.py --one <mode> against an alternative build of the library:
python tools/lib_loop.py <lib.so> <mode> [bench_train_loop args]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train_loop.py")
sys.argv = [script, "--one", sys.argv[2]] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
