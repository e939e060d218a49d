"""pathtracer_amd — MI355X render core for andreasots/pathtracer behind a C ABI.

`core` binds libptcore.so (HIP kernels + host-side mirrors), `abi` holds the plain-data layouts of
include/ptcore.h, `workloads` the synthetic scenes of SURVEY.md §8(d), `sharding` the sample-split
multi-GPU plumbing.
"""
from . import abi, core, workloads, sharding  # noqa: F401

__all__ = ["abi", "core", "workloads", "sharding"]
