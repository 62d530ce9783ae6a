"""Decodes the reference's own texture assets (/root/reference/Assets/Textures: Alien-Metal_{Albedo,Metallic,Roughness}.png,
{Earth,Moon}_{BaseColor,Normal}.jpg -- used at Source/MyScene.ixx:161-166, 285-295) with PIL in the build container and writes them,
reduced 8x by a box filter, as raw RGBA fixtures tests/golden/textures/<stem>.ptex ("PTEX", width, height, pixels: host/Texture.hpp
LoadRawTexture).  These are DATA (decoded pixels); no reference source is copied.  The reference itself cannot travel to the GPU box,
the fixtures can.   Run from the repo root:  python tests/golden/make_textures.py"""
import glob
import os
import struct

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/Assets/Textures"


def main():
    out_dir = os.path.join(HERE, "textures")
    os.makedirs(out_dir, exist_ok=True)
    for f in sorted(glob.glob(os.path.join(SRC, "*"))):
        im = Image.open(f).convert("RGBA")
        w, h = im.size[0] // 8, im.size[1] // 8
        im = im.resize((w, h), Image.BOX)
        px = np.asarray(im, dtype=np.uint8)
        stem = os.path.splitext(os.path.basename(f))[0]
        with open(os.path.join(out_dir, stem + ".ptex"), "wb") as o:
            o.write(b"PTEX" + struct.pack("<II", w, h) + px.tobytes())
        print(stem, (w, h), "mean rgba", px.reshape(-1, 4).mean(0).round(1))


if __name__ == "__main__":
    main()
