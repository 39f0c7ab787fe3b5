"""slamem_amd -- MI355X-native MEM retrieval engine behind slaMEM's command line.

The product is ``csrc/libslamem_hip.so`` (hand-written HIP kernels for gfx950 behind the C ABI of
``include/slamem_hip.h``) and the C front end in ``host/``.  This Python package is the thin harness
around the C ABI used by the tests and ``bench.py``; importing it does not load the library, every
compute call does and fails loudly when it is missing (there is no CPU fallback).
"""
__all__ = ["capi", "engine", "shard", "synth"]
