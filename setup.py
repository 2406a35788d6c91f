"""Packaging for `pip install -e .` from a checkout.  The HIP library is built in-tree by
`python __graft_entry__.py` (or `make -C bipymc_amd/csrc`) and shipped as package data."""
from setuptools import find_packages, setup

setup(
    name="bipymc_amd",
    version="0.1.0",
    description="MI355X-native DE-MC / DREAM population sampler (drop-in for wgurecky/bipymc's DeMcMpi / DreamMpi)",
    packages=find_packages(include=["bipymc_amd", "bipymc_amd.*"]),
    package_data={"bipymc_amd": ["libbipymc_hip.so", "csrc/*"]},
    install_requires=["numpy>=1.20"],
    python_requires=">=3.8",
)
