"""CPU: the PNG grid writer of the sample.py drop-ins (stand-in for torchvision.utils.save_image,
mnist/sample.py:115-116): layout and a byte-exact decode of the file it writes."""
import struct
import zlib

import numpy as np

import mvae_amd  # noqa: F401
from mvae_amd.sample_common import CELEBA_ATTRS, make_grid, write_png


def _decode_png(path):
    data = open(path, 'rb').read()
    assert data[:8] == b'\x89PNG\r\n\x1a\n'
    pos, chunks = 8, []
    while pos < len(data):
        n, tag = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xffffffff
        chunks.append((tag, body))
        pos += 12 + n
    w, h, depth, ctype = struct.unpack('>IIBB', chunks[0][1][:10])
    assert (depth, ctype) == (8, 2) and chunks[-1][0] == b'IEND'
    raw = zlib.decompress(b''.join(b for t, b in chunks if t == b'IDAT'))
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


def test_grid_layout_and_png_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    imgs = rng.rand(11, 1, 28, 28).astype(np.float32)
    grid = make_grid(imgs)                       # 8 per row, 2 px padding -> 2 rows
    assert grid.shape == (2 * 30 + 2, 8 * 30 + 2, 3)
    assert (grid[:2] == 0).all() and (grid[:, :2] == 0).all()
    k = 9                                        # second row, second column
    tile = grid[30 + 2:30 + 2 + 28, 30 + 2:30 + 2 + 28, 0]
    assert np.array_equal(tile, np.clip(imgs[k, 0] * 255 + 0.5, 0, 255).astype(np.uint8))
    assert (grid[32:60, 3 * 30 + 2:, :] == 0).all()     # cells 11..15 stay empty
    path = str(tmp_path / 'g.png')
    write_png(path, grid)
    assert np.array_equal(_decode_png(path), grid)
    rgb = make_grid(rng.rand(3, 3, 64, 64).astype(np.float32))
    assert rgb.shape == (68, 3 * 66 + 2, 3)


def test_celeba_attribute_names():
    assert len(CELEBA_ATTRS) == 18 and CELEBA_ATTRS[9] == 'Male' and CELEBA_ATTRS[14] == 'Smiling'
