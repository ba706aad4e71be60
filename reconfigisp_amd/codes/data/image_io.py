"""Image file reading without cv2 (absent from this image; the reference reads every frame with
``cv2.imread(path, cv2.IMREAD_UNCHANGED)``, e.g. data/sid_sony_ratio_rggb2bgr_dataset.py:112-117).

``read_image(path)`` returns what that call returns for the files the datasets hold:
  * ``.png``  8- or 16-bit, grey / RGB / RGBA / palette, non-interlaced -> (H,W) or (H,W,3|4) uint8 / uint16 with the
              colour channels in **BGR(A)** order (cv2's convention; the ground-truth frames are BGR),
  * ``.npy``  the stored array as is (a convenient container for 14-bit mosaics).
The PNG decoder is zlib + numpy: chunk walk, inflate, undo the five scan-line filters (Sub / Up vectorised over the row,
Average / Paeth per byte column), assemble samples."""
import struct
import zlib

import numpy as np

_SIG = b'\x89PNG\r\n\x1a\n'
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _unfilter(raw, height, stride, bpp):
    out = np.zeros((height + 1, stride), np.uint8)                # row 0 = the all-zero "previous" row
    pos = 0
    for y in range(1, height + 1):
        ftype = raw[pos]
        line = np.frombuffer(raw, np.uint8, stride, pos + 1)
        pos += stride + 1
        prev, cur = out[y - 1], out[y]
        if ftype == 0:
            cur[:] = line
        elif ftype == 2:                                          # Up
            cur[:] = line + prev
        elif ftype == 1:                                          # Sub: running sum per byte lane
            lanes = line.reshape(-1, bpp).astype(np.uint32) if stride % bpp == 0 else None
            if lanes is not None:
                cur[:] = (np.cumsum(lanes, axis=0) & 0xFF).astype(np.uint8).reshape(-1)
            else:
                cur[:] = line
                for i in range(bpp, stride):
                    cur[i] = (int(cur[i]) + int(cur[i - bpp])) & 0xFF
        elif ftype in (3, 4):                                     # Average / Paeth: sequential in x
            ln, pv = line.tolist(), prev.tolist()
            row = [0] * stride
            for i in range(stride):
                a = row[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ftype == 3:
                    pred = (a + b) >> 1
                else:
                    c = pv[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                row[i] = (ln[i] + pred) & 0xFF
            cur[:] = row
        else:
            raise ValueError('PNG: unknown filter type %d' % ftype)
    return out[1:]


def read_png(path):
    with open(path, 'rb') as f:
        data = f.read()
    if data[:8] != _SIG:
        raise ValueError('%s is not a PNG file' % path)
    pos, idat, header, palette = 8, [], None, None
    while pos < len(data):
        length, kind = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b'IHDR':
            header = struct.unpack('>IIBBBBB', body)
        elif kind == b'PLTE':
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b'IDAT':
            idat.append(body)
        elif kind == b'IEND':
            break
    width, height, depth, ctype, _, _, interlace = header
    if interlace:
        raise NotImplementedError('%s: interlaced PNG' % path)
    if depth not in (8, 16) or ctype not in _CHANNELS:
        raise NotImplementedError('%s: PNG bit depth %d / colour type %d' % (path, depth, ctype))
    ch = _CHANNELS[ctype]
    bpp = ch * depth // 8
    rows = _unfilter(zlib.decompress(b''.join(idat)), height, width * bpp, bpp)
    if depth == 16:
        img = rows.reshape(height, width, ch, 2)
        img = (img[..., 0].astype(np.uint16) << 8) | img[..., 1]             # big-endian samples
    else:
        img = rows.reshape(height, width, ch)
    if ctype == 3:
        img = palette[img[..., 0]]
        ch = 3
    if ch == 1:
        return np.ascontiguousarray(img[..., 0])
    if ch == 2:                                                   # grey + alpha: IMREAD_UNCHANGED gives BGRA
        return np.ascontiguousarray(np.stack([img[..., 0]] * 3 + [img[..., 1]], axis=-1))
    order = [2, 1, 0] + ([3] if ch == 4 else [])                  # RGB(A) -> BGR(A)
    return np.ascontiguousarray(img[..., order])


def write_png(path, img):
    """(H,W) or (H,W,3) uint8 / uint16, colour in BGR order -> PNG (filter 0; for fixtures and result dumps)."""
    img = np.asarray(img)
    if img.dtype not in (np.uint8, np.uint16) or img.ndim not in (2, 3):
        raise ValueError('write_png: uint8 / uint16 (H,W) or (H,W,3) arrays only')
    if img.ndim == 3:
        img = img[..., ::-1]
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    depth = 8 * img.dtype.itemsize
    body = img.astype('>u2' if depth == 16 else np.uint8).reshape(h, -1).view(np.uint8)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), body], axis=1).tobytes()

    def chunk(kind, payload):
        return struct.pack('>I', len(payload)) + kind + payload + struct.pack('>I', zlib.crc32(kind + payload) & 0xFFFFFFFF)

    with open(path, 'wb') as f:
        f.write(_SIG + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, {1: 0, 3: 2}[ch], 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def read_image(path):
    low = path.lower()
    if low.endswith('.npy'):
        return np.load(path)
    if low.endswith('.png'):
        return read_png(path)
    raise NotImplementedError('%s: only .png and .npy frames can be read in this build (no cv2)' % path)
