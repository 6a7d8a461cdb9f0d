"""mspa -- host layer of the MI355X-native MultiSPA geometry engine.

``mspa._lib``    ctypes binding of libmspa.so (the C ABI of include/mspa.h; hand-written HIP, gfx950)
``mspa.engine``  PyTorch-ROCm tensor front end: device-resident frame sets, the three kernels
``mspa.shard``   one-process-per-GPU sharding of pairs/scenes and the RCCL collation step
``mspa.synth``   seeded synthetic RGB-D scenes / track sets (inputs for tests and bench)

There is no CPU fallback anywhere in this package: if libmspa.so is missing or no GPU is
visible the calls raise.  The NumPy/C restatements of the reference live under oracle/ and are
test infrastructure only.
"""

import os as _os

# The streaming sweeps keep ten scenes' depth-decode kernels in flight on ten HIP streams (mspa/upload.py).  The ROCm runtime
# multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default, i.e. at most 4 kernels side by side:
# 12 k frames/s instead of 29 k (tools/device_ingest_streams.py, profiles/r06_device_ingest.md).  The runtime reads the variable
# when it initialises, so it is set here, on import, unless the caller has chosen a value; importing this package after the
# first HIP call leaves the default in place (everything still runs, the from-disk sweeps just decode with less overlap).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
