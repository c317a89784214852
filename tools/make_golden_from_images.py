"""Extracts small numeric fixtures from the reference repository's published renders (images/*.bmp).

Run once in the build container (the reference tree is not available on the GPU box):
    python tools/make_golden_from_images.py /root/reference/images tests/golden/published_image_patches.json
For each image it stores 9x9-pixel mean values (8-bit, tone-mapped as published) on a grid of positions in
the part of the picture above the objects (rows 0..250 of 600), where only box, light and camera matter.
Positions are (col, row) with row 0 at the top, as an image viewer shows them.
"""
import json
import struct
import sys

import numpy as np


def read_bmp(path):
    d = open(path, "rb").read()
    off, = struct.unpack_from("<I", d, 10)
    w, h = struct.unpack_from("<ii", d, 18)
    bpp, = struct.unpack_from("<H", d, 28)
    bypp = bpp // 8
    stride = (w * bypp + 3) & ~3
    img = np.zeros((abs(h), w, 3), np.uint8)
    for r in range(abs(h)):
        row = np.frombuffer(d, np.uint8, w * bypp, off + r * stride).reshape(w, bypp)
        y = abs(h) - 1 - r if h > 0 else r  # positive height = bottom-up
        img[y] = row[:, 2::-1]  # B,G,R(,A) -> R,G,B
    return img


def main(src, dst):
    out = {"note": "9x9 means of the 8-bit published renders of willalexander/pathtracer (images/*.bmp); "
                   "positions are (col,row), row 0 = top; extracted by tools/make_golden_from_images.py",
           "images": {}}
    cols = [40, 100, 170, 300, 430, 500, 560]
    rows = [20, 130, 200, 245]
    for name in ["cornellBox.bmp", "dragonChrome.bmp", "dragonGlass.bmp", "dragonMilk.bmp", "dragonKetchup.bmp"]:
        img = read_bmp(f"{src}/{name}").astype(np.float64)
        patches = []
        for r in rows:
            for c in cols:
                p = img[r - 4:r + 5, c - 4:c + 5].reshape(-1, 3).mean(0)
                patches.append({"col": c, "row": r, "rgb": [round(float(x), 2) for x in p]})
        out["images"][name] = {"width": img.shape[1], "height": img.shape[0], "patches": patches}
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
