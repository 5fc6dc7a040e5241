"""Counter-based deterministic synthetic inputs (oracle side, NumPy).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The same generator is
implemented in HIP (``csrc/fill.hip``) so multi-GB tables are regenerated on
the GPU box and never shipped; SURVEY.md §8(d) "Synthetic inputs".

Definition (all arithmetic mod 2**64):

    mix(x)     = splitmix64 finalizer of (x + 0x9E3779B97F4A7C15)
    stream     = mix(mix(seed) ^ (tensor_id * 0xD6E8FEB86659FD93))
    u64(i)     = mix(stream + i)

    f32 uniform : u = (u64 >> 40) * 2**-24   (exact in f32)
                  v = fl32(fl32(u * scale) + lo)          (two roundings, no fma)
    id uniform  : id = ((u64 >> 32) * N) >> 32            (pure integer)
    id powerlaw : u = (u64 >> 11) * 2**-53 (f64); u2 = u*u; u4 = u2*u2
                  id = min(N-1, floor(N * u4))            (IEEE f64 products only)

There is no reference counterpart: the reference has no synthetic data beyond a
100-row ``np.random.seed(42)`` frame in ``tests/unit/test_preprocessor.py:277-292``.
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_TID = np.uint64(0xD6E8FEB86659FD93)


def mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 step on uint64 array(s) (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def stream_key(seed: int, tensor_id: int) -> np.uint64:
    with np.errstate(over="ignore"):
        s = mix(np.array([seed], dtype=np.uint64))
        t = np.array([tensor_id], dtype=np.uint64) * _TID
        return mix(s ^ t)[0]


def raw_u64(seed: int, tensor_id: int, n: int, start: int = 0) -> np.ndarray:
    key = stream_key(seed, tensor_id)
    idx = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return mix(key + idx)


def uniform_f32(seed: int, tensor_id: int, n: int, lo: float, scale: float,
                start: int = 0) -> np.ndarray:
    """v = fl32(fl32(u*scale) + lo), u in [0,1) with 24 bits."""
    x = raw_u64(seed, tensor_id, n, start)
    u = (x >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return (u * np.float32(scale)).astype(np.float32) + np.float32(lo)


def ids_uniform(seed: int, tensor_id: int, n: int, num_rows: int, start: int = 0) -> np.ndarray:
    x = raw_u64(seed, tensor_id, n, start)
    hi = x >> np.uint64(32)
    return ((hi * np.uint64(num_rows)) >> np.uint64(32)).astype(np.int64)


def ids_powerlaw(seed: int, tensor_id: int, n: int, num_rows: int, start: int = 0) -> np.ndarray:
    x = raw_u64(seed, tensor_id, n, start)
    u = (x >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    u2 = u * u
    u4 = u2 * u2
    ids = np.floor(np.float64(num_rows) * u4).astype(np.int64)
    return np.minimum(ids, num_rows - 1)


# ---- tensor-id convention shared with the product's synthetic initialiser ----
TID_USER_TABLE = 1
TID_ITEM_TABLE = 2
TID_USER_IDS = 3
TID_ITEM_IDS = 4
TID_DENSE_BASE = 16          # + 2*layer (+1 for the item tower); user tower even, item tower odd... see dense_tid
TID_CATEGORY_TABLE = 5
TID_CATEGORY_IDS = 6
TID_SAMPLE_WEIGHT = 7
TID_CAND_PROB = 8


TID_DROPOUT_BASE = 64


def dropout_tid(tower: int, layer: int) -> int:
    return TID_DROPOUT_BASE + 2 * layer + tower


def dropout_keep(seed: int, tensor_id: int, row_start: int, rows: int, n: int, rate: float):
    """(keep mask [rows, n] bool, scale f32) of the product's inverted dropout: element (r, c) of the layer output
    whose global batch row is row_start + r is dropped iff the top 24 bits of u64(counter (row_start+r)*n + c)
    are < round(rate * 2**24).  Configs: /root/reference/configs/data_config.yaml:58 (dropout_rate: 0.1)."""
    x = raw_u64(seed, tensor_id, rows * n, start=row_start * n)
    p24 = np.uint64(int(float(np.float32(rate)) * 16777216.0 + 0.5))
    keep = ((x >> np.uint64(40)) >= p24).reshape(rows, n)
    return keep, np.float32(1.0) / (np.float32(1.0) - np.float32(rate))


def dense_tid(tower: int, layer: int) -> int:
    """tower 0 = user, 1 = item."""
    return TID_DENSE_BASE + 2 * layer + tower


def embedding_table(seed: int, tensor_id: int, num_rows: int, dim: int,
                    row_start: int = 0, row_count: int | None = None) -> np.ndarray:
    """Keras ``Embedding`` default init U(-0.05, 0.05)."""
    if row_count is None:
        row_count = num_rows - row_start
    flat = uniform_f32(seed, tensor_id, row_count * dim, lo=-0.05, scale=0.1,
                       start=row_start * dim)
    return flat.reshape(row_count, dim)


def glorot_limit(fan_in: int, fan_out: int) -> np.float32:
    return np.float32(np.sqrt(6.0 / (fan_in + fan_out)))


def dense_kernel(seed: int, tensor_id: int, fan_in: int, fan_out: int) -> np.ndarray:
    """Keras ``Dense`` default Glorot-uniform kernel, shape [in, out]."""
    lim = glorot_limit(fan_in, fan_out)
    scale = np.float32(lim + lim)
    return uniform_f32(seed, tensor_id, fan_in * fan_out, lo=-lim, scale=scale).reshape(fan_in, fan_out)


def batch_ids(seed: int, tensor_id: int, step: int, batch: int, num_rows: int,
              variant: str = "U") -> np.ndarray:
    """Ids of training step ``step``: counter offset = step*batch."""
    if variant == "U":
        return ids_uniform(seed, tensor_id, batch, num_rows, start=step * batch)
    if variant == "Z":
        return ids_powerlaw(seed, tensor_id, batch, num_rows, start=step * batch)
    raise ValueError(f"unknown id variant {variant!r}")
