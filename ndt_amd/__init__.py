"""ndt_amd -- MI355X-native ray-trace core for the ndt N-dimensional tracer.

Host-side Python is plumbing only: scene files <-> the C ABI of libndt_hip.so
(include/ndt_hip.h), device buffers, and torch.distributed for the multi-GPU image gather.
"""
from .flat_scene import (FlatScene, load_scene, RenderParams, RenderStats, shard_rows, OBJ_TYPES,  # noqa: F401
                         ABI_VERSION)
