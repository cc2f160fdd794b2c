"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes binding of ``oracle/rans_oracle.c`` (the plain-C restatement of
CompressAI's ``pmf_to_quantized_cdf`` and rANS coder).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this.

PARITY UNPINNED (see rans_oracle.c header and DESIGN.md).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle_rans.so')
_lib = None


def build(force=False):
    """Compiles oracle/rans_oracle.c with gcc (no GPU, no reference needed)."""
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'rans_oracle.c')):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.oracle_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.oracle_pmf_to_quantized_cdf.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int,
                                                  ctypes.POINTER(ctypes.c_uint32)]
        L.oracle_rans_encode_with_indexes.restype = ctypes.c_long
        L.oracle_rans_encode_with_indexes.argtypes = [i32p, i32p, ctypes.c_long, i32p, ctypes.c_int, i32p, i32p,
                                                      u8p, ctypes.c_long]
        L.oracle_rans_decode_with_indexes.restype = ctypes.c_int
        L.oracle_rans_decode_with_indexes.argtypes = [u8p, ctypes.c_long, i32p, ctypes.c_long, i32p, ctypes.c_int,
                                                      i32p, i32p, i32p]
        _lib = L
    return _lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    cdf = np.zeros(pmf.size + 1, dtype=np.uint32)
    rc = lib().oracle_pmf_to_quantized_cdf(_p(pmf, ctypes.c_float), int(pmf.size), int(precision),
                                           _p(cdf, ctypes.c_uint32))
    if rc == -1:
        raise ValueError('Invalid `pmf`, non-finite or negative element found')
    if rc == -2:
        raise ValueError('Invalid `pmf`: at least one element must have a non-zero probability.')
    if rc != 0:
        raise AssertionError('pmf_to_quantized_cdf: no symbol to steal from')
    return cdf


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    """symbols/indexes: 1-D int sequences; cdfs: 2-D int table [n_cdfs, stride]. Returns bytes."""
    symbols, indexes = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
    cdfs, cdf_sizes, offsets = _i32(cdfs), _i32(cdf_sizes).reshape(-1), _i32(offsets).reshape(-1)
    assert cdfs.ndim == 2 and symbols.size == indexes.size
    cap = 4 * (4 * symbols.size + 16) + 64
    out = np.empty(cap, dtype=np.uint8)
    i32 = ctypes.c_int32
    nb = lib().oracle_rans_encode_with_indexes(_p(symbols, i32), _p(indexes, i32), int(symbols.size),
                                               _p(cdfs, i32), int(cdfs.shape[1]), _p(cdf_sizes, i32),
                                               _p(offsets, i32), _p(out, ctypes.c_uint8), int(cap))
    if nb < 0:
        raise RuntimeError('oracle encode failed: {}'.format(nb))
    return out[:nb].tobytes()


def decode_with_indexes(encoded, indexes, cdfs, cdf_sizes, offsets):
    indexes = _i32(indexes).reshape(-1)
    cdfs, cdf_sizes, offsets = _i32(cdfs), _i32(cdf_sizes).reshape(-1), _i32(offsets).reshape(-1)
    enc = np.frombuffer(bytes(encoded), dtype=np.uint8).copy()
    out = np.empty(indexes.size, dtype=np.int32)
    i32 = ctypes.c_int32
    rc = lib().oracle_rans_decode_with_indexes(_p(enc, ctypes.c_uint8), int(enc.size), _p(indexes, i32),
                                               int(indexes.size), _p(cdfs, i32), int(cdfs.shape[1]),
                                               _p(cdf_sizes, i32), _p(offsets, i32), _p(out, i32))
    if rc != 0:
        raise RuntimeError('oracle decode failed: {}'.format(rc))
    return out
