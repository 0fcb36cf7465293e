"""Sequence-parallel all-gather attention through the inter-node entry points.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_sp_ag_attention_inter_node.py
    python triton_dist/test/nvidia/test_sp_ag_attention_inter_node.py --virtual-ranks 4 --check

The reference has a 2D (node x local rank) variant for multi-node jobs
(kernels/nvidia/sp_ag_attention_inter_node.py); one MI355X node is a single xGMI full mesh, so
triton_dist.kernels.nvidia.sp_ag_attention_inter_node resolves to the same full-mesh kernel.
This script checks exactly that: the module imports under the reference's name and produces the
intra-node result.  (CLI of reference test_sp_ag_attention_inter_node.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import runpy  # noqa: E402

import triton_dist.kernels.nvidia.sp_ag_attention_inter_node as inter  # noqa: E402
import triton_dist.kernels.nvidia.sp_ag_attention_intra_node as intra  # noqa: E402


def main():
    assert inter.fused_sp_ag_attn_intra_node is intra.fused_sp_ag_attn_intra_node
    here = os.path.dirname(os.path.abspath(__file__))
    runpy.run_path(os.path.join(here, "test_sp_ag_attention_intra_node.py"), run_name="__main__")


if __name__ == "__main__":
    main()
