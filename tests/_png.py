"""A small PNG writer for the tests (the format's own specification, RFC 2083 / ISO 15948): any of the
five row filters, gray / gray+alpha, 1..16 bits.  Independent of the decoder under test."""
import struct
import zlib

import numpy as np


def _chunk(typ, data):
    return struct.pack(">I", len(data)) + typ + data + struct.pack(">I", zlib.crc32(typ + data) & 0xffffffff)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    if pa <= pb and pa <= pc:
        return a
    return b if pb <= pc else c


def pack_rows(img, depth, channels=1):
    """img: [h][w] (or [h][w][channels]) integer samples < 2**depth -> list of raw row byte arrays."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    assert c == channels
    rows = []
    for y in range(h):
        s = a[y].reshape(-1).astype(np.uint32)
        if depth == 8:
            rows.append(s.astype(np.uint8).tobytes())
        elif depth == 16:
            rows.append(s.astype(">u2").tobytes())
        else:
            per = 8 // depth
            pad = (-len(s)) % per
            s = np.concatenate([s, np.zeros(pad, np.uint32)]).reshape(-1, per)
            shifts = np.array([8 - depth * (i + 1) for i in range(per)], np.uint32)
            rows.append((s << shifts).sum(1).astype(np.uint8).tobytes())
    return rows


def encode(img, depth=8, color_type=0, filters=(0, 1, 2, 3, 4), idat_split=0, level=6, extra_chunks=()):
    channels = {0: 1, 4: 2}[color_type]
    a = np.asarray(img)
    h, w = a.shape[:2]
    rows = pack_rows(a, depth, channels)
    bpp = max(1, channels * depth // 8)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        out = bytearray(len(row))
        for i in range(len(row)):
            left = row[i - bpp] if i >= bpp else 0
            up = prev[i]
            ul = prev[i - bpp] if i >= bpp else 0
            pred = (0, left, up, (left + up) >> 1, _paeth(left, up, ul))[ft]
            out[i] = (row[i] - pred) & 255
        raw.append(ft)
        raw += out
        prev = row
    z = zlib.compress(bytes(raw), level)
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    for typ, data in extra_chunks:
        png += _chunk(typ, data)
    if idat_split:
        for i in range(0, len(z), idat_split):
            png += _chunk(b"IDAT", z[i:i + idat_split])
    else:
        png += _chunk(b"IDAT", z)
    return png + _chunk(b"IEND", b"")
