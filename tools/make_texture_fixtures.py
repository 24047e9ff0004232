#!/usr/bin/env python3
"""Decode the reference's texture assets (standalone/textures/*, data files, public domain per the
reference LICENSE) with PIL and store the TEXELS as fixtures: tests/golden/textures/<name>.rgb8.z =
8-byte little-endian (width, height) + zlib(RGB8, row 0 = top). The reference decodes with
image 0.23.8 (texture.rs:22); JPEG decoders may differ by an LSB or two, no reference test pins
texel values, so the build DEFINES texels as this decode — "parity unpinned" for JPEG,
decoder-independent for the PNG. Run in the authoring container only (needs /root/reference)."""
import hashlib, json, os, sys, zlib
import numpy as np
from PIL import Image
import PIL

SRC = "/root/reference/standalone/textures"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "textures")
os.makedirs(DST, exist_ok=True)
manifest = {"decoder": f"PIL {PIL.__version__}", "files": {}}
for name in ("bergsjostolen.jpg", "earth.jpg", "chessboard.png"):
    with Image.open(os.path.join(SRC, name)) as im:
        mode = im.mode
        rgb = np.asarray(im.convert("RGB"))
    h, w, _ = rgb.shape
    blob = np.array([w, h], dtype="<u4").tobytes() + zlib.compress(rgb.tobytes(), 9)
    stem = os.path.splitext(name)[0]
    with open(os.path.join(DST, stem + ".rgb8.z"), "wb") as f:
        f.write(blob)
    manifest["files"][stem] = {"source": name, "source_mode": mode, "width": w, "height": h,
                               "sha256_rgb8": hashlib.sha256(rgb.tobytes()).hexdigest(), "bytes": len(blob)}
    print(stem, w, h, mode, len(blob))
with open(os.path.join(DST, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1)
