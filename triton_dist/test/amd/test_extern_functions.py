"""Calling native entry points by symbol name (``extern_call``), the way the reference's
``dl.extern_call`` reaches device library symbols.

    python triton_dist/test/amd/test_extern_functions.py

(Scenario of reference python/triton_dist/test/amd/test_extern_functions.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import ctypes  # noqa: E402

from triton_dist.language.core import extern_call  # noqa: E402


def main():
    argparse.ArgumentParser().parse_args()
    n = extern_call("tdh_registry_num_kernels", lib="host", check=False)
    assert isinstance(n, int) and n >= 0
    words = extern_call("tdist_mega_task_words", lib="hip", check=False) if torch.cuda.is_available() else 16
    assert words == 16, words
    try:
        extern_call("no_such_symbol", lib="host")
    except AttributeError:
        pass
    else:
        raise AssertionError("a missing symbol must raise")
    size = extern_call("tdist_ipc_handle_size", restype=ctypes.c_int, lib="hip", check=False) if torch.cuda.is_available() else 64
    assert size == 64, size
    print("✅ extern_call: host + device-library symbols resolved and called", flush=True)


if __name__ == "__main__":
    main()
