"""Builds libsdnet_hip.so (hipcc, gfx950) before the Python package is collected: `pip install -e .` / `python -m build`.
All metadata lives in pyproject.toml."""
import os
import subprocess
from pathlib import Path

from setuptools import setup
from setuptools.command.build_py import build_py

CSRC = Path(__file__).resolve().parent / "structuredetector_amd" / "csrc"


class BuildWithNativeLibrary(build_py):
    def run(self):
        env = dict(os.environ)
        env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.run(["make", "-C", str(CSRC), "-j4"], check=True, env=env)
        super().run()


setup(cmdclass={"build_py": BuildWithNativeLibrary})
