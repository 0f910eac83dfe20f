"""tests/golden/reference_cornell_box_blocks.npy: 8x8 block means (sRGB, 0..255, float32, shape 135 x 240 x 3) of the one rendered image the
reference repository holds, resources/cornell_box.png (its README's picture: 1920x1080, sample count unknown, made from an earlier
state of cornell_box.yaml -- see tests/test_gpu_reference_image.py for what differs).  Run where /root/reference exists.
The PNG is decoded here with zlib and the five PNG row filters; nothing but pixel data is read from the file."""
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
    w, h, depth, ctype, _, _, interlace = hdr
    assert depth == 8 and interlace == 0 and ctype in (2, 6), hdr
    ch = 4 if ctype == 6 else 3
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:  # 1 sub, 3 average, 4 Paeth: the left neighbour is the row being built
            cur = np.zeros(w * ch, dtype=np.int32)
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, ch)


def block_means(img, b=8):
    h, w, _ = img.shape
    return img[:, :, :3].astype(np.float64).reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3)).astype(np.float32)


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/resources/cornell_box.png"
    img = read_png(src)
    print(img.shape, img[540, 300], img[540, 1600], img[200, 960], img[900, 960])
    np.save(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_blocks.npy"), block_means(img))
