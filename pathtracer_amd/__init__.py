"""pathtracer_amd: MI355X-native path-tracing compute pass behind the reference's render seam.

`device.PathTracer` drives the HIP kernels through the C ABI in include/ptx355.h; `host` mirrors the
reference's scene ingest and frame loop.  The native libraries are mandatory: nothing here computes
on the CPU in their place."""
from .records import CAMERA_DT, MATERIAL_DT, NODE_DT, SHAPE_DT, TEXINFO_DT  # noqa: F401
