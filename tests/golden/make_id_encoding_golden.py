"""Generates tests/golden/id_encoding_small.npz by running the REFERENCE's own
id-encoding code in the build container (never on the GPU box; /root/reference
does not travel).

  a7: scripts/data_processing/prepare_training_data.py:113-123,209-210
      (create_user_item_mappings + Series.map)
  a6: src/data/preprocessor.py:478-491 — that module cannot be imported here
      (``import nltk`` at preprocessor.py:14 fails: nltk is absent), so the
      statement it executes, ``sklearn.preprocessing.LabelEncoder().fit_transform``,
      is run directly on the same columns.

Usage: python tests/golden/make_id_encoding_golden.py
"""
import importlib.util
import pathlib

import numpy as np
import pandas as pd
from sklearn.preprocessing import LabelEncoder

REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parent / "id_encoding_small.npz"


def load_reference_script():
    spec = importlib.util.spec_from_file_location(
        "ref_prepare_training_data", REF / "scripts/data_processing/prepare_training_data.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synthetic_frame(seed=1234, n_rows=4000, n_users=600, n_items=450):
    rng = np.random.default_rng(seed)
    alphabet = list("ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789")
    extra = ["é", "ß", "Ω", "中", "𝔘", "_", "-", "a", "z"]          # non-ASCII + case mix
    def rand_id(prefix, k):
        body = "".join(rng.choice(alphabet, size=k))
        if rng.random() < 0.15:
            body += str(rng.choice(extra))
        return prefix + body
    users = sorted({rand_id("A", int(rng.integers(6, 20))) for _ in range(n_users)})
    items = sorted({rand_id("B0", int(rng.integers(4, 9))) for _ in range(n_items)})
    cats = ["Books", "Electronics", "All_Beauty", "Toys_and_Games", None, "Movies_and_TV"]
    df = pd.DataFrame({
        "user_id": rng.choice(users, size=n_rows),
        "parent_asin": rng.choice(items, size=n_rows),
        "main_category": rng.choice(np.array(cats, dtype=object), size=n_rows),
        "rating": rng.integers(1, 6, size=n_rows).astype(np.float64),
    })
    return df


def main():
    ref = load_reference_script()
    df = synthetic_frame()
    user_to_idx, item_to_idx = ref.create_user_item_mappings(df)          # :113-123
    user_idx = df["user_id"].map(user_to_idx).to_numpy()                  # :209
    item_idx = df["parent_asin"].map(item_to_idx).to_numpy()              # :210
    assert user_idx.dtype == np.int64 and item_idx.dtype == np.int64
    user_enc = LabelEncoder().fit_transform(df["user_id"])                # preprocessor.py:481
    item_enc = LabelEncoder().fit_transform(df["parent_asin"])            # :482
    cat_enc = LabelEncoder().fit_transform(df["main_category"].fillna("Unknown"))  # :485-489
    assert np.array_equal(user_idx, user_enc) and np.array_equal(item_idx, item_enc)
    np.savez_compressed(
        OUT,
        user_id=np.array(df["user_id"].tolist(), dtype=np.str_),
        parent_asin=np.array(df["parent_asin"].tolist(), dtype=np.str_),
        main_category=np.array(["<NaN>" if c is None else c for c in df["main_category"].tolist()], dtype=np.str_),
        user_idx=user_idx.astype(np.int64), item_idx=item_idx.astype(np.int64),
        user_id_encoded=np.asarray(user_enc, dtype=np.int64),
        item_id_encoded=np.asarray(item_enc, dtype=np.int64),
        category_encoded=np.asarray(cat_enc, dtype=np.int64),
    )
    print("wrote", OUT, "rows", len(df), "users", len(user_to_idx), "items", len(item_to_idx))


if __name__ == "__main__":
    main()
