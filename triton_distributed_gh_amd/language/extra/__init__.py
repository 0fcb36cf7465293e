from . import libshmem_device  # noqa: F401
