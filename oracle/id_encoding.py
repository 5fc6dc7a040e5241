"""Restatement of the reference's integer id encoding (oracle; PINNED by
tests/golden/id_encoding_*.npz — see oracle/__init__.py).

Follows
* ``/root/reference/scripts/data_processing/prepare_training_data.py:113-123``
  (``create_user_item_mappings``: ``sorted(unique)`` then ``enumerate``) and
  ``:209-210`` (``Series.map`` of those dicts -> ``user_idx`` / ``item_idx``);
* ``/root/reference/src/data/preprocessor.py:478-491``
  (``LabelEncoder().fit_transform`` -> ``user_id_encoded`` / ``item_id_encoded``
  / ``category_encoded`` with NaN categories replaced by "Unknown" first).

Both give id = rank of the string among the sorted distinct strings
(Python ``str`` order = Unicode code-point order = UTF-8 byte order), int64.
"""
from __future__ import annotations

import numpy as np


def encode_ids(values) -> tuple[np.ndarray, list]:
    """Returns (int64 codes, sorted vocabulary)."""
    vals = list(values)
    vocab = sorted(set(vals))
    lut = {v: i for i, v in enumerate(vocab)}
    return np.fromiter((lut[v] for v in vals), dtype=np.int64, count=len(vals)), vocab


def encode_categories(values, unknown="Unknown") -> tuple[np.ndarray, list]:
    """``df["main_category"].fillna("Unknown")`` then LabelEncoder (preprocessor.py:485-489)."""
    vals = [unknown if (v is None or (isinstance(v, float) and v != v)) else v for v in values]
    return encode_ids(vals)


def encode_ids_utf8(values) -> np.ndarray:
    """Same ranks computed the way a byte-oriented (C/GPU) encoder would: sort the
    UTF-8 encodings bytewise.  Must equal encode_ids (code-point order == UTF-8
    byte order); checked in tests/test_oracle_cpu.py."""
    enc = [v.encode("utf-8") for v in values]
    vocab = sorted(set(enc))
    lut = {v: i for i, v in enumerate(vocab)}
    return np.fromiter((lut[v] for v in enc), dtype=np.int64, count=len(enc))
