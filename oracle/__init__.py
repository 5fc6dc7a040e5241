"""CPU oracle for the two-tower retrieval training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.

PARITY STATUS
-------------
* Hot path (embedding lookup, MLP towers, dot-product scorer, in-batch
  sampled-softmax loss, sparse/dense optimizers): **parity unpinned**.  The
  reference repository contains no implementation of it
  (``/root/reference/src/models/__init__.py:1`` and
  ``src/training/__init__.py:1`` are docstring stubs) and no test or fixture
  pins any of its outputs.  The arithmetic would live in the un-vendored,
  un-pinned third-party packages ``tensorflow>=2.15.0`` and
  ``tensorflow-recommenders>=0.7.3`` (``pyproject.toml:22,24``), neither of
  which is installed or installable here.  This restatement follows the
  published TFRS 0.7.x / Keras 2.15 semantics and the
  hyper-parameter contract in ``configs/data_config.yaml:54-71``.
* Id encoding (``src/data/preprocessor.py:478-491``,
  ``scripts/data_processing/prepare_training_data.py:113-123,209-210``):
  **pinned** by ``tests/golden/id_encoding_*.npz``, generated in the build
  container by importing the reference's own ``create_user_item_mappings``
  and sklearn's ``LabelEncoder`` (script: ``tests/golden/make_id_encoding_golden.py``).
"""
