"""Profiling helper (not a test): bench.py on another build of libojf.so (OJF_LIB_PATH), tolerating builds that lack newer entry points
(the ctypes table is trimmed to what the library exports) - for A/B kernel traces of an older library with today's tools."""
import ctypes, os, sys
import torch  # first: the process must bind torch's HIP runtime before the library pulls in another copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_joint_depthfusion_and_semantic_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
for name in list(_lib.SIGNATURES):
    if not hasattr(lib, name):
        del _lib.SIGNATURES[name]
import bench
bench.main()
