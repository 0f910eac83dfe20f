"""numpy dtypes of the ABI records in include/ptx355.h (byte-identical to the reference's #[repr(C)]
structs: src/scene/shape.rs:223-248, src/scene/material.rs:108-119, src/accel/bvh.rs:12-19,
src/io/texture_atlas.rs:10-18, src/camera/camera.rs:150-171)."""
import numpy as np

SHAPE_DT = np.dtype([
    ("shape_type", "<u4"), ("material_idx", "<u4"), ("radius", "<f4"), ("radius2", "<f4"),
    ("position", "<f4", 3), ("height", "<f4"),
    ("normal", "<f4", 3), ("csg_op", "<u4"),
    ("rotation", "<f4", 3), ("texture_scale", "<f4"),
    ("v0", "<f4", 3), ("pad2", "<u4"),
    ("v1", "<f4", 3), ("pad3", "<u4"),
    ("v2", "<f4", 3), ("pad4", "<u4"),
])
MATERIAL_DT = np.dtype([
    ("base_color", "<f4", 3), ("metallic", "<f4"),
    ("emission", "<f4", 3), ("roughness", "<f4"),
    ("emission_strength", "<f4"), ("ior", "<f4"), ("transmission", "<f4"), ("texture_id", "<i4"),
])
NODE_DT = np.dtype([("aabb_min", "<f4", 3), ("left_or_prim", "<u4"), ("aabb_max", "<f4", 3), ("prim_count", "<u4")])
TEXINFO_DT = np.dtype([("width", "<u4"), ("height", "<u4"), ("offset", "<u4"), ("pad", "<u4")])
CAMERA_DT = np.dtype([
    ("position", "<f4", 3), ("focal_length", "<f4"),
    ("right", "<f4", 3), ("aspect", "<f4"),
    ("up", "<f4", 3), ("exposure", "<f4"),
    ("forward", "<f4", 3), ("frame_index", "<u4"),
    ("width", "<u4"), ("height", "<u4"), ("sample_count", "<u4"), ("max_bounces", "<u4"),
    ("tone_mapper", "<u4"), ("fractal_march_steps", "<u4"), ("firefly_clamp", "<f4"), ("skybox_brightness", "<f4"),
    ("skybox_color", "<f4", 3), ("pad2", "<f4"),
])
assert SHAPE_DT.itemsize == 112 and MATERIAL_DT.itemsize == 48 and NODE_DT.itemsize == 32
assert TEXINFO_DT.itemsize == 16 and CAMERA_DT.itemsize == 112
