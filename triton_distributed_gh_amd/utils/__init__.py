from .distributed import (DistEnv, finalize_distributed, get_env, init_seed, initialize_distributed,  # noqa: F401
                          is_initialized)
from .symm import SymmetricHeap, SymmCtxStruct  # noqa: F401
