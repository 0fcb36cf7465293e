"""pip-installable wrapper around the in-tree build: ``pip install -e .`` or ``python setup.py build_ext --inplace``
compile libtdist_hip.so / libtdist_host.so next to the package (reference: python/setup.py, which
patches and builds Triton + rocSHMEM; nothing of that is needed here)."""
from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py


def _build_native():
    from triton_distributed_gh_amd import build

    build.build_all(verbose=True)


class BuildExt(Command):
    description = "compile the HIP / C++ libraries for gfx950"
    user_options = [("inplace", "i", "accepted for compatibility; the build is always in-tree")]

    def initialize_options(self):
        self.inplace = 1

    def finalize_options(self):
        pass

    def run(self):
        _build_native()


class BuildPy(build_py):

    def run(self):
        _build_native()
        super().run()


setup(
    name="triton_dist",
    version="0.1.0",
    description="MI355X-native compute-communication overlap library (triton_dist-compatible API)",
    packages=find_packages(include=["triton_distributed_gh_amd*", "triton_dist*", "pyrocshmem*"]),
    package_data={"triton_distributed_gh_amd": ["*.so", "csrc/**/*"]},
    cmdclass={"build_ext": BuildExt, "build_py": BuildPy},
    python_requires=">=3.10",
)
