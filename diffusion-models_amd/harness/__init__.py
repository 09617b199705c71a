"""Step harnesses around the drop-in model: the training step (train.py:429-452 semantics, data-parallel) and
batch-sharded generation (train.py:476-504 / save_fid_eval.py:163-172 semantics)."""
