import hashlib
import os

import numpy as np


def hash_seed(seed=None, max_bytes=8):
    """little-endian integer of the first max_bytes bytes of sha512(str(seed))"""
    if seed is None:
        seed = int.from_bytes(os.urandom(max_bytes), "little")
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()
    return int.from_bytes(digest[:max_bytes], "little")


def np_random(seed=None):
    if seed is None:
        seed = int.from_bytes(os.urandom(4), "little")
    seed = int(seed)
    h = hash_seed(seed)
    rng = np.random.RandomState()
    rng.seed([(h >> (32 * i)) & 0xFFFFFFFF for i in range(2)])
    return rng, seed
