"""Same command line as test/amd/test_ag_gemm_intra_node.py through the NVIDIA-named entry points
(``create_ag_gemm_context`` / ``ag_gemm``).  (Reference: python/triton_dist/test/nvidia/test_ag_gemm.py.)"""
import os
import runpy

runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "amd", "test_ag_gemm_intra_node.py"),
               run_name="__main__")
