"""AOT kernel registry, CLI and the user-kernel loader (reference: python/triton_dist/tools/)."""
from .compile_aot import (aot_compile_spaces, bind_launcher, distributed, install_triton_alias,  # noqa: F401
                          parse_signature, register_kernel, registered_kernels, select_algo, validate_spec)
from .runtime import (Kernel, Module, compile_hip, f32, f64, i32, i64, load_module, load_workspace,  # noqa: F401
                      marshal_args, u64)
