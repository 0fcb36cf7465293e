"""Kernel entry points.  Import the submodules you need; nothing heavy happens at import time."""
