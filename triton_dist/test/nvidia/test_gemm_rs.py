"""Same command line as test/amd/test_gemm_rs_intra_node.py.  (Reference:
python/triton_dist/test/nvidia/test_gemm_rs.py.)"""
import os
import runpy

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "amd", "test_gemm_rs_intra_node.py"),
               run_name="__main__")
