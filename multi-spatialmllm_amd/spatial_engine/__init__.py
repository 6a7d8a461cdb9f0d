"""Drop-in mirror of the reference's ``spatial_engine`` package for the geometry hot path.

Module paths, function names, arguments and return types follow
facebookresearch/Multi-SpatialMLLM; the arithmetic runs in libmspa.so (hand-written HIP, gfx950)
through ``mspa.engine``.  Put ``multi-spatialmllm_amd/`` on PYTHONPATH *instead of* the reference
tree.  These call-compatible wrappers move NumPy arrays over PCIe on every call; pipelines that
care about throughput keep data resident and use ``mspa.engine`` / ``mspa.scene`` directly.
"""
