"""chromap_amd -- MI355X (gfx950) implementation of Chromap's per-read mapping hot path.

Python is only a thin convenience layer over the C ABI in include/chromap_amd.h
(chromap_amd/libchromap_amd.so, hand-written HIP).  See DESIGN.md / INTEGRATION.md.
"""
from ._capi import Batch, IndexView, Params, Record, RefView, Stats, default_params, lib  # noqa: F401
from .mapper import ChromapError, ChromapGPU, read_fastq_pairs, read_fastx  # noqa: F401

__all__ = ["ChromapGPU", "default_params", "read_fastx", "read_fastq_pairs", "lib"]
