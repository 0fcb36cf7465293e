"""AOT kernel registry and CLI (reference: python/triton_dist/tools/)."""
from .compile_aot import (aot_compile_spaces, bind_launcher, distributed, install_triton_alias,  # noqa: F401
                          parse_signature, register_kernel, registered_kernels, select_algo, validate_spec)
