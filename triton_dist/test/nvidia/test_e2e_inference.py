"""Same command line as the AMD-named script (the implementation underneath is the same on MI355X):
runs triton_dist/test/amd/test_e2e_inference.py.  (Reference: python/triton_dist/test/nvidia/test_e2e_inference.py.)"""
import os
import runpy

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "amd", "test_e2e_inference.py"), run_name="__main__")
