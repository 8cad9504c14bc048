import hashlib
import os

import numpy as np


def hash_seed(seed=None, max_bytes=8):
    """little-endian integer of the first max_bytes bytes of sha512(str(seed))"""
    if seed is None:
        seed = int.from_bytes(os.urandom(max_bytes), "little")
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()
    return int.from_bytes(digest[:max_bytes], "little")


def _int_list_from_bigint(bigint):
    if bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def np_random(seed=None):
    if seed is None:
        seed = int.from_bytes(os.urandom(8), "big")
    seed = int(seed) % 2 ** 64
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed
