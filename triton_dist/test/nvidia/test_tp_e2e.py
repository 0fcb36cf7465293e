"""Same command line as the AMD-named script (the implementation underneath is the same on MI355X):
runs triton_dist/test/amd/test_tp_e2e.py.  (Reference: python/triton_dist/test/nvidia/test_tp_e2e.py.)"""
import os
import runpy

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "amd", "test_tp_e2e.py"), run_name="__main__")
