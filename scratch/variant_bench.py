"""Diagnostic: run bench.py against a variant build of the library (RWK_VARIANT_LIB=path)."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_amd import _lib
_lib.LIB_PATH = os.environ["RWK_VARIANT_LIB"]
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
