#!/usr/bin/env python3
"""Pre-decode the scenes' PNG textures into the .rtex container the loaders read.

The reference decodes PNG to RGBA8 through the Corona library
(/root/reference/mesh.h:117-135, pixel format PF_R8G8B8A8, top row first).
PNG is lossless, so any conforming decoder yields the same bytes; this tool
uses Pillow once, in the build container, and the result is committed so that
no PNG decoder is needed where the renderer runs.

.rtex layout (little endian): magic "RTEX", u32 width, u32 height,
u32 zlib_len, then a zlib stream of width*height*4 bytes (RGBA8, row major,
top row first).

usage: decode_textures.py <src_dir_with_png> <dst_dir>
"""
import os
import struct
import sys
import zlib

from PIL import Image


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for name in sorted(os.listdir(src)):
        if not name.lower().endswith(".png"):
            continue
        im = Image.open(os.path.join(src, name)).convert("RGBA")
        raw = im.tobytes()
        assert len(raw) == im.width * im.height * 4
        z = zlib.compress(raw, 9)
        with open(os.path.join(dst, name + ".rtex"), "wb") as f:
            f.write(b"RTEX" + struct.pack("<III", im.width, im.height, len(z)) + z)
        print(f"{name}: {im.width}x{im.height} -> {len(z)} bytes")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
