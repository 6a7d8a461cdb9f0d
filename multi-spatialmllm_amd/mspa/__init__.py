"""mspa -- host layer of the MI355X-native MultiSPA geometry engine.

``mspa._lib``    ctypes binding of libmspa.so (the C ABI of include/mspa.h; hand-written HIP, gfx950)
``mspa.engine``  PyTorch-ROCm tensor front end: device-resident frame sets, the three kernels
``mspa.shard``   one-process-per-GPU sharding of pairs/scenes and the RCCL collation step
``mspa.synth``   seeded synthetic RGB-D scenes / track sets (inputs for tests and bench)

There is no CPU fallback anywhere in this package: if libmspa.so is missing or no GPU is
visible the calls raise.  The NumPy/C restatements of the reference live under oracle/ and are
test infrastructure only.
"""
