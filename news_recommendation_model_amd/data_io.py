"""The reference's processed-data container and loader: the format either side of the hot path.

Container (reference tool/process_data.py:449-462): one file = ONE zstd frame of ``pickle.dumps(obj)``.
  head file   ``<name>``                obj = [subvolume_num, total_data_number, max_user_id, user_num]      (:270, :291)
  subvolume   ``<name>.subvolume<i>``   obj = list of records                                               (:255-262, :289)
  record      [impression_id, user_id, history [H, P+16] f64, inview [T, P+14] f64, global [T, 3] f64,
               label_true [T] f64, label_id [T] f64 (-1 = padding), n_padding]                               (:252)
``torch.utils.data.DataLoader`` default-collates a list of such records into the 8 tensors ``train.py:67`` unpacks.

PARITY UNPINNED.  The reference module imports ``zstandard`` at its top, which is not installed here, so it cannot be
imported and none of its files hold a processed-data fixture; everything below is restated from the source text.  zstd
frames are read and written with pyarrow's codec (standard frames, the same bytes ``zstandard`` produces/accepts).
Unpickling is restricted to what a record holds (lists, numbers, numpy arrays); anything else in a file is refused.
"""
from __future__ import annotations

import io
import os
import pickle

import numpy as np

RECORD_FIELDS = ("impression_id", "user_id", "x_history", "x_target", "x_global", "label", "label_id", "empty_num")


# ------------------------------------------------------------------------------------------------ zstd + pickle
def _codec(level=11):
    import pyarrow as pa
    if not pa.Codec.is_available("zstd"):
        raise RuntimeError("this pyarrow build has no zstd codec")
    return pa.Codec("zstd", compression_level=level)


def zstd_compress(raw: bytes, level: int = 11) -> bytes:
    """One standard zstd frame (``ZstdCompressor(level=11).compress``, process_data.py:460-461)."""
    return _codec(level).compress(raw, asbytes=True)


def zstd_decompress(blob: bytes) -> bytes:
    """``ZstdDecompressor().decompress`` (:451-452).  Streams, so the frame need not carry its content size."""
    import pyarrow as pa
    with pa.CompressedInputStream(pa.BufferReader(blob), "zstd") as stream:
        return stream.read()


_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
}


class _RecordUnpickler(pickle.Unpickler):
    """Lists / tuples / numbers / strings are built by the pickle VM itself; the only globals a processed-data file
    needs are numpy's array and scalar reconstructors."""

    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"processed-data files hold lists, numbers and numpy arrays only; refusing {module}.{name}")


def export_processed_data(data, path, need_copy=False, level=11):
    """process_data.py:455-462."""
    if need_copy:
        data = list(data)
    with open(path, "wb") as f:
        f.write(zstd_compress(pickle.dumps(data), level))


def import_processed_data(path):
    """process_data.py:449-453 (restricted unpickling)."""
    with open(path, "rb") as f:
        raw = zstd_decompress(f.read())
    return _RecordUnpickler(io.BytesIO(raw)).load()


# ------------------------------------------------------------------------------------------------ dataset files
def write_processed_dataset(records, head_path, subvolume_item_num=30000, max_user_id=None, user_num=None):
    """Head file + subvolumes as ``process_dataset`` leaves them (:255-291): full subvolumes of ``subvolume_item_num``
    records, a last partial one, head = [subvolume_num, total, max_user_id, user_num]."""
    records = list(records)
    users = {int(r[1]) for r in records}
    max_user_id = max(users) if max_user_id is None and users else (max_user_id or 0)
    user_num = len(users) if user_num is None else user_num
    n_sub = 0
    for lo in range(0, len(records), subvolume_item_num):
        export_processed_data(records[lo:lo + subvolume_item_num], f"{head_path}.subvolume{n_sub}")
        n_sub += 1
    export_processed_data([n_sub, len(records), int(max_user_id), int(user_num)], head_path)
    return head_path


def _iter_subvolumes(head_file_path, n_sub):
    """Records of ``<head>.subvolume0 .. <n_sub-1>`` in file order; a missing subvolume is skipped (:110)."""
    for i in range(n_sub):
        path = f"{head_file_path}.subvolume{i}"
        if os.path.isfile(path):
            yield from import_processed_data(path)


class _UserQuota:
    """Per-user state of the balanced subset: ``pending`` holds the user's records until its block of ``quota`` is
    full; ``taken`` counts what the user has contributed to the block/extra slots so far."""
    __slots__ = ("pending", "taken")

    def __init__(self, first):
        self.pending = [first]
        self.taken = 1


def _balanced_subset(stream, want, quota, extras):
    """The selection rule of process_data.py:120-143 as a state machine over the record stream.

    * every user first collects a block of ``quota`` records; the block is emitted, whole and in arrival order, at the
      moment its last record arrives (a block can only be completed by a record that is NOT the user's first -- with
      quota == 1 no block is ever emitted, as in the reference);
    * the record after a user's full block is emitted on the spot while the global budget of ``extras`` lasts
      (one per user); anything later from that user is dropped;
    * the scan stops as soon as ``want`` records are out; if the stream ends first, the users whose block never
      filled hand in what they hold, in order of first appearance (this tail is not capped at ``want``)."""
    out, users = [], {}
    for rec in stream:
        uid = rec[1]
        st = users.get(uid)
        if st is None:
            users[uid] = _UserQuota(rec)
        elif st.taken < quota:
            st.pending.append(rec)
            st.taken += 1
            if st.taken == quota:
                out.extend(st.pending)
                st.pending = None
        elif st.taken == quota and extras > 0:
            out.append(rec)
            st.taken += 1
            extras -= 1
        if len(out) >= want:
            return out
    for st in users.values():
        if st.taken < quota:
            out.extend(st.pending)
    return out


def load_processed_dataset(head_file_path, load_data_number=-1, user_min_data_num=2):
    """Behaviour of process_data.py:92-145.  ``load_data_number`` < 0 or >= the total: the first ``total`` records of
    the subvolumes, in file order.  Otherwise a per-user balanced subset (``_balanced_subset``): block size
    ``max(load_data_number // user_num, user_min_data_num)``, and ``load_data_number - block * user_num`` users may add
    one record beyond their block.  Returns (records, max_user_id).  PARITY UNPINNED (module docstring)."""
    n_sub, total, max_user_id, n_users = import_processed_data(head_file_path)
    stream = _iter_subvolumes(head_file_path, n_sub)
    if load_data_number < 0 or load_data_number >= total:
        records = []
        for rec in stream:
            if len(records) >= total:
                break
            records.append(rec)
        return records, max_user_id
    quota = max(int(load_data_number / n_users), user_min_data_num)
    return _balanced_subset(stream, load_data_number, quota, load_data_number - quota * n_users), max_user_id


# ------------------------------------------------------------------------------------------------ batching
def collate(records):
    """What DataLoader's default_collate makes of a list of records (train.py:40,67), as a dict of numpy arrays with
    the field names the rest of this package uses (``synth.make_batch``)."""
    cols = list(zip(*records))
    out = {}
    for name, col in zip(RECORD_FIELDS, cols):
        if name in ("impression_id", "user_id"):
            out[name] = np.asarray(col, dtype=np.int64)
        elif name == "empty_num":
            out[name] = np.asarray(col).astype(np.int64)
        else:
            out[name] = np.stack([np.asarray(a, dtype=np.float64) for a in col])
    return out


def iter_batches(records, batch_size, shuffle=True, seed=0):
    """``DataLoader(dataset=records, batch_size=B, shuffle=True)`` (train.py:40): the last batch may be short."""
    order = np.arange(len(records))
    if shuffle:
        np.random.default_rng(seed).shuffle(order)
    for lo in range(0, len(order), batch_size):
        yield collate([records[i] for i in order[lo:lo + batch_size]])


def records_from_batch(batch):
    """The inverse of ``collate`` for a ``synth.make_batch`` dict: a list of 8-field records (label_id = the candidate
    index where a row is real, -1 on padding rows, as process_data.py:216-235 fills it)."""
    B = len(batch["user_id"])
    T = batch["x_target"].shape[1]
    recs = []
    for b in range(B):
        n_pad = int(batch["empty_num"][b]) if "empty_num" in batch else 0
        label_id = np.arange(T, dtype=np.float64)
        if n_pad:
            label_id[T - n_pad:] = -1
        recs.append([int(batch.get("impression_id", np.arange(B))[b]), int(batch["user_id"][b]),
                     np.asarray(batch["x_history"][b], dtype=np.float64), np.asarray(batch["x_target"][b], dtype=np.float64),
                     np.asarray(batch["x_global"][b], dtype=np.float64), np.asarray(batch["label"][b], dtype=np.float64),
                     label_id, np.float64(n_pad)])
    return recs
