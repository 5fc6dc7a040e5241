"""Hash-feature ids (oracle side, NumPy / pure Python).

TEST INFRASTRUCTURE (see oracle/__init__.py).  BASELINE.json configs[4] asks for "30 categories as hash
features"; the reference names no hash function ("Not specified anywhere ... hash
function for category features") and has no fixture for one, so parity of the bucket ids is UNPINNED: this
file is the definition, ``csrc/gather.hip::hash_bucket_kernel`` must reproduce it bit for bit.

    bucket(s) = FNV-1a-64(utf-8 bytes of s) mod n_buckets

(The TF-side analogue, ``tf.keras.layers.Hashing``, uses FarmHash64; it is not reproduced here.)
The category strings are the ``category`` column written by
/root/reference/scripts/data_processing/prepare_training_data.py:47 (or ``main_category``,
/root/reference/src/data/preprocessor.py:478-491).
"""
from __future__ import annotations

import numpy as np

FNV_OFFSET = 0xCBF29CE484222325
FNV_PRIME = 0x100000001B3
_MASK = (1 << 64) - 1


def fnv1a64(data: bytes) -> int:
    h = FNV_OFFSET
    for c in data:
        h = ((h ^ c) * FNV_PRIME) & _MASK
    return h


def hash_buckets(values, n_buckets: int) -> np.ndarray:
    """int64 bucket ids of a sequence of str."""
    return np.array([fnv1a64(v.encode("utf-8")) % n_buckets for v in values], dtype=np.int64)
