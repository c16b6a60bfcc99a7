"""gym 0.21 ``utils/seeding.py`` restated from memory (UNVERIFIED against the real package).

``np_random(seed)`` returns a ``numpy.random.RandomState`` seeded with the 32-bit words of the first
8 bytes of SHA-512(str(seed)).  The build never relies on this hash for parity: reset noise is an
explicit, logged input of every golden trace.
"""
import hashlib
import os
import struct

import numpy as np


def _bigint_from_bytes(data):
    sizeof_int = 4
    padding = sizeof_int - len(data) % sizeof_int
    data += b"\0" * padding
    int_count = len(data) // sizeof_int
    unpacked = struct.unpack("{}I".format(int_count), data)
    accum = 0
    for i, val in enumerate(unpacked):
        accum += 2 ** (sizeof_int * 8 * i) * val
    return accum


def _int_list_from_bigint(bigint):
    if bigint < 0:
        raise ValueError("Seed must be non-negative, not {}".format(bigint))
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def create_seed(a=None, max_bytes=8):
    if a is None:
        a = _bigint_from_bytes(os.urandom(max_bytes))
    elif isinstance(a, int):
        a = a % 2 ** (8 * max_bytes)
    else:
        raise ValueError("Invalid type for seed: {}".format(type(a)))
    return a


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(digest[:max_bytes])


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, int) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
