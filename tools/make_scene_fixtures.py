"""Writes data/scenes/<name>.npz from the reference's scene files (run in the build container, where the
reference checkout is mounted; the GPU box never sees it).  A fixture is DATA: the authored shape list
(after OBJ expansion) in figure/material record containers, the camera block and decoded textures.
Textures larger than 128x128 (teapot.yaml's 1.9 MB wood.jpg, the planets' maps, the skyboxes) are replaced by a 64x64
procedural texture: the renders of the fixtures are compared with the oracle's renders of the same fixtures, not with
images of the reference.  A model whose OBJ file tobj would refuse (aelita.obj) is left out, as the reference leaves it out.

Usage: python tools/make_scene_fixtures.py [/root/reference]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import host_prep as hp  # noqa: E402  (fixture generation is test tooling)

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
import glob  # noqa: E402

# every scene file the reference ships (30): the four benchmark scenes, one scene per primitive type, the textured and the model scenes
SCENES = sorted(os.path.basename(p)[:-5] for p in glob.glob(f"{REF}/resources/scenes/*.yaml"))


def authored_records(shapes):
    n = len(shapes)
    figs = np.zeros(n, dtype=hp.SHAPE_DT)
    mats = np.zeros(n, dtype=hp.MATERIAL_DT)
    for i, s in enumerate(shapes):
        g, m = figs[i], mats[i]
        g["shape_type"] = hp.TYPE_ID[s.shape_type]
        g["radius"], g["radius2"], g["height"] = s.radius, s.radius2, s.height
        g["position"], g["normal"], g["rotation"] = s.position, s.normal, s.rotation
        g["csg_op"] = 1 if s.negative else 0
        g["texture_scale"] = 1.0 if s.texture_scale is None else s.texture_scale
        g["v0"] = (s.power, float(s.max_iterations), 0.0) if s.shape_type in ("mandelbulb", "julia") else s.v0
        g["v1"], g["v2"] = s.v1, s.v2
        g["pad2"], g["pad3"], g["pad4"] = (hp.pack_f16x2(*s.uv0), hp.pack_f16x2(*s.uv1), hp.pack_f16x2(*s.uv2))
        mm = s.material
        m["base_color"], m["metallic"], m["emission"], m["roughness"] = mm.base_color, mm.metallic, mm.emission, mm.roughness
        m["emission_strength"], m["ior"], m["transmission"], m["texture_id"] = mm.emission_strength, mm.ior, mm.transmission, mm.texture_id
    return figs, mats


def procedural_texture(size=64):
    y, x = np.mgrid[0:size, 0:size]
    img = np.zeros((size, size, 4), dtype=np.uint8)
    img[..., 0] = (x * 4) & 0xFF
    img[..., 1] = (y * 4) & 0xFF
    img[..., 2] = ((x ^ y) * 4) & 0xFF
    img[..., 3] = 255
    return img


def main():
    out_dir = os.path.join(ROOT, "data", "scenes")
    os.makedirs(out_dir, exist_ok=True)
    for name in SCENES:
        path = f"{REF}/resources/scenes/{name}.yaml"
        if not os.path.exists(path):
            print("skip", name)
            continue
        scene = hp.load_scene(path, REF)
        shapes = hp.expand_models(scene)
        figs, mats = authored_records(shapes)
        c = scene.camera
        cam = np.array(list(c.position) + list(c.rotation) + [c.fov, c.exposure, c.max_bounces, c.firefly_clamp]
                       + list(c.skybox_color) + [c.skybox_brightness, c.tone_mapper, c.fractal_march_steps], dtype=np.float64)
        tex_paths = []
        for s in shapes:
            if s.texture and s.texture not in tex_paths:
                tex_paths.append(s.texture)
        data = {"figs": figs, "mats": mats, "camera": cam, "n_textures": np.int64(len(tex_paths))}
        for t, tp in enumerate(tex_paths):
            data[f"tex{t}_shapes"] = np.array([i for i, s in enumerate(shapes) if s.texture == tp], dtype=np.int64)
            try:
                from PIL import Image
                img = np.asarray(Image.open(tp).convert("RGBA"), dtype=np.uint8)
                if img.shape[0] * img.shape[1] > 128 * 128:
                    img = procedural_texture()
            except Exception:
                img = procedural_texture()
            data[f"tex{t}_rgba8"] = img
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **data)
        print(name, len(shapes), "shapes", len(tex_paths), "textures")


if __name__ == "__main__":
    main()
