"""Image files either side of the render path.

The reference writes its test renders with ``imageio.imwrite`` (render_utils.py:312-315) and reads
dataset frames with ``imageio.imread`` (load_blender.py:69, load_llff.py:95-111); imageio is not
part of this image.  PNG needs nothing but zlib, so it is handled here directly (8-bit gray / RGB /
RGBA, non-interlaced); other formats (the JPEGs of LLFF scenes) go through Pillow when it is
installed.

``AsyncImageWriter`` takes the encoding off the render loop: frames are compressed and written by a
small thread pool (zlib releases the GIL) while the GPU renders the next pose.
"""
import concurrent.futures
import os
import struct
import zlib

import numpy as np

_PNG_MAGIC = b"\x89PNG\r\n\x1a\n"
_COLOR_TYPE = {1: 0, 2: 4, 3: 2, 4: 6}          # channels -> PNG colour type
_CHANNELS = {0: 1, 4: 2, 2: 3, 6: 4}


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png(img, level=6):
    """uint8 [H, W] / [H, W, 1|2|3|4] -> PNG bytes (filter 0 rows, one IDAT)."""
    a = np.asarray(img)
    if a.dtype != np.uint8:
        raise TypeError("encode_png expects uint8, got %s (quantise with utils.to8b first)" % a.dtype)
    if a.ndim == 2:
        a = a[..., None]
    if a.ndim != 3 or a.shape[2] not in _COLOR_TYPE:
        raise ValueError("encode_png expects [H, W] or [H, W, 1..4], got %s" % (a.shape,))
    h, w, c = a.shape
    rows = np.empty((h, 1 + w * c), np.uint8)
    rows[:, 0] = 0
    rows[:, 1:] = a.reshape(h, w * c)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, _COLOR_TYPE[c], 0, 0, 0)
    return _PNG_MAGIC + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(rows.tobytes(), level)) + _chunk(b"IEND", b"")


def write_png(path, img, level=6):
    data = encode_png(img, level)
    tmp = path + ".part"
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)           # a reader never sees a half-written frame


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def decode_png(data):
    """PNG bytes -> uint8 [H, W, C] (8-bit, non-interlaced, colour types 0/2/4/6)."""
    if data[:8] != _PNG_MAGIC:
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    if hdr is None:
        raise ValueError("PNG without IHDR")
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in _CHANNELS or interlace != 0:
        raise ValueError("unsupported PNG (bit depth %d, colour type %d, interlace %d): install Pillow" % (depth, ctype, interlace))
    c = _CHANNELS[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), np.uint8)
    prev = np.zeros(w * c, np.uint8)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:]
        if ft == 0:
            cur = line.copy()
        elif ft == 2:
            cur = line + prev                                     # uint8 wrap-around = mod 256
        elif ft == 1:
            cur = np.cumsum(line.reshape(w, c), axis=0, dtype=np.uint8).reshape(-1)
        elif ft in (3, 4):
            cur = np.zeros(w * c, np.uint8)
            li, pv = line.tolist(), prev.tolist()
            res = [0] * (w * c)
            for i in range(w * c):
                left = res[i - c] if i >= c else 0
                up = pv[i]
                ul = pv[i - c] if i >= c else 0
                pred = ((left + up) >> 1) if ft == 3 else _paeth(left, up, ul)
                res[i] = (li[i] + pred) & 255
            cur[:] = res
        else:
            raise ValueError("bad PNG filter type %d" % ft)
        out[y] = cur
        prev = cur
    return out.reshape(h, w, c)


def read_image(path):
    """Image file -> uint8 [H, W, C] (what imageio.imread returns for 8-bit files).  PNGs that the
    built-in decoder covers need no third-party package; everything else goes through Pillow."""
    try:
        from PIL import Image
    except ImportError:
        Image = None
    if Image is not None:
        with Image.open(path) as im:
            if im.mode not in ("L", "LA", "RGB", "RGBA"):
                im = im.convert("RGBA" if "A" in im.getbands() or im.mode == "P" and "transparency" in im.info else "RGB")
            a = np.asarray(im)
        return a[..., None] if a.ndim == 2 else a
    if not path.lower().endswith(".png"):
        raise ImportError("reading %s needs Pillow (only PNG is decoded natively)" % path)
    with open(path, "rb") as f:
        return decode_png(f.read())


class AsyncImageWriter:
    """PNG encoding + file writes on worker threads.  ``submit(path, frame)`` returns at once; the
    frame (uint8 host array) must not be modified until ``wait()``/``close()`` or until the
    returned future is done.  Errors surface in ``wait()``."""

    def __init__(self, workers=4, level=6):
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self._futures = []
        self._level = level

    def submit(self, path, frame, before=None):
        """`before`: optional callable run on the worker first (e.g. wait for the frame's D2H copy)."""
        def job():
            if before is not None:
                before()
            write_png(path, frame, self._level)
        fut = self._pool.submit(job)
        self._futures.append(fut)
        return fut

    def wait(self):
        futs, self._futures = self._futures, []
        for f in futs:
            f.result()

    def close(self):
        try:
            self.wait()
        finally:
            self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
