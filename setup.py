"""pip install -e . : builds rii_amd/librii_amd.so with hipcc (gfx950) through rii_amd/csrc/Makefile and installs the
pure-Python host layer.  (The reference's own build system, setup.py there, is out of scope; this is only a convenience
for users who switch `import rii` to `import rii_amd as rii`.)"""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "rii_amd", "csrc"), "-j4"])
        super().run()


setup(
    name="rii_amd",
    version="0.1.0",
    description="MI355X-native IVFPQ query engine behind the rii.Rii API",
    packages=["rii_amd"],
    package_data={"rii_amd": ["librii_amd.so", "csrc/*"]},
    python_requires=">=3.8",
    install_requires=["numpy"],
    cmdclass={"build_py": BuildWithHip},
)
