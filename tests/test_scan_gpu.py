"""Direct tests of the device-wide scans (hagrid_amd/csrc/wave_prims.h) that replace cub::DeviceScan::ExclusiveSum
(reference: parallel.cuh:31-42; call sites build.cu:487,557,597,659, merge.cu:310-311, flatten.cu:136, compress.cu:51):
tile-boundary sizes, the device carry, pairs of ints (the two-word publish of the look-back form), both forms."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TILE = 256 * 8          # kScanTile


@pytest.fixture(scope="module")
def mem():
    from hagrid_amd import api
    m = api.MemManager(keep=True)
    yield m
    m.close()


def scan(mem, values, words, carry, lookback):
    from hagrid_amd import api
    values = np.ascontiguousarray(values, dtype=np.int32)
    n = values.size // words
    out = np.empty_like(values); total = np.zeros(words, np.int32)
    c = None if carry is None else np.ascontiguousarray(carry, dtype=np.int32)
    api._check(mem, mem._K.hagrid_kat_scan(mem._ctx, values.ctypes.data_as(C.c_void_p), n, words, None if c is None else c.ctypes.data_as(C.c_void_p),
                                           lookback, out.ctypes.data_as(C.c_void_p), total.ctypes.data_as(C.c_void_p)), "kat_scan")
    return out, total


@pytest.mark.parametrize("lookback", [1, 0, 2, 1 | 4, 2 | 4, 0 | 4])      # + 4: in place (a helper must not sum a tile its owner is overwriting)
@pytest.mark.parametrize("words", [1, 2])
def test_scan_sizes_and_carry(mem, words, lookback):
    rng = np.random.default_rng(7)
    for n in (0, 1, 63, 64, 65, 255, 256, TILE - 1, TILE, TILE + 1, 64 * TILE - 1, 64 * TILE, 64 * TILE + 1, 65 * TILE + 77, 1_000_003, 10_000_000):
        v = rng.integers(0, 9, size=(n, words), dtype=np.int32)
        for carry in (None, np.array([5, 1_000_000][:words], np.int32)):
            out, total = scan(mem, v.reshape(-1), words, carry, lookback)
            base = np.zeros(words, np.int64) if carry is None else carry.astype(np.int64)
            incl = np.cumsum(v.astype(np.int64), axis=0) + base
            want = np.concatenate([base[None, :], incl[:-1]]) if n else np.zeros((0, words), np.int64)
            assert (out.reshape(n, words) == want).all(), (n, words, lookback, carry is not None)
            assert (total == (incl[-1] if n else base)).all(), (n, words, lookback)


def test_scan_lookback_many_calls_share_status_words(mem):
    """Status words are re-used across calls (epochs instead of clearing): alternate sizes and value types."""
    rng = np.random.default_rng(11)
    for it in range(40):
        words = 1 + (it & 1)
        n = int(rng.integers(1, 300_000))
        v = rng.integers(0, 1000, size=(n, words), dtype=np.int32)
        out, total = scan(mem, v.reshape(-1), words, None, 1)
        incl = np.cumsum(v.astype(np.int64), axis=0)
        assert (out.reshape(n, words)[1:] == incl[:-1]).all() and (out.reshape(n, words)[0] == 0).all() and (total == incl[-1]).all()


def test_scan_negative_values_and_wraparound(mem):
    """The look-back status carries a 32-bit value: negative partial sums and two's-complement wrap must survive."""
    v = np.array([-5, 7, -(1 << 30), -(1 << 30), -(1 << 30), 3] * 5000, np.int32)
    for lookback in (1, 0):
        out, total = scan(mem, v, 1, None, lookback)
        want = (np.concatenate([[0], np.cumsum(v.astype(np.int64))[:-1]]) & 0xffffffff).astype(np.uint32).view(np.int32)
        assert (out == want).all()


def test_scan_in_place_forced_helping_repeated(mem):
    """The helping path on scans that overwrite their input (merge.hip tile sums, ray_order.hip bin table, trav_image.hip sizes): every
    lane helps at its first miss while the owners of those tiles are alive and storing outputs over the items -- a helper that summed a
    half-overwritten tile would publish a wrong aggregate.  Many repetitions: the window is a race."""
    rng = np.random.default_rng(23)
    for it in range(60):
        words = 1 + (it & 1)
        n = int(rng.integers(200 * TILE, 900 * TILE))
        v = rng.integers(1, 1000, size=(n, words), dtype=np.int32)
        out, total = scan(mem, v.reshape(-1), words, None, 2 | 4)
        incl = np.cumsum(v.astype(np.int64), axis=0)
        assert (out.reshape(n, words)[1:] == incl[:-1]).all() and (out.reshape(n, words)[0] == 0).all() and (total == incl[-1]).all(), it
