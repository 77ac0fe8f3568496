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


def load_processed_dataset(head_file_path, load_data_number=-1, user_min_data_num=2):
    """process_data.py:92-145, statement for statement (including its quirks): everything when ``load_data_number`` < 0
    or >= total, otherwise a per-user balanced subset -- every user contributes at most ``max_data_num - 1`` records in
    blocks (a block is released when it fills), plus one extra record for the first ``max_data_user_num`` users that
    overflow, and the partial blocks at the end if the quota is still open.  Returns (records, max_user_id)."""
    subvolume_num, total_data_number, max_user_id, user_num = import_processed_data(head_file_path)
    if load_data_number < 0:
        load_data_number = total_data_number
        max_data_num = total_data_number
        max_data_user_num = total_data_number
    else:
        load_data_number = min(total_data_number, load_data_number)
        max_data_num = max(int(load_data_number / user_num), user_min_data_num) + 1
        max_data_user_num = load_data_number - (max_data_num - 1) * user_num
    processed_data = []
    user_id_dict = {}
    for i in range(subvolume_num):
        subvolume_path = "{}.subvolume{}".format(head_file_path, i)
        if not os.path.isfile(subvolume_path):
            continue
        part = import_processed_data(subvolume_path)
        if load_data_number == total_data_number:
            part = part[0:min(load_data_number - len(processed_data), len(part))]
            processed_data = processed_data + part
        else:
            for data in part:
                user_id = data[1]
                if user_id in user_id_dict:
                    held = user_id_dict[user_id]
                    if len(held) == max_data_num - 1 and max_data_user_num > 0:
                        processed_data.append(data)
                        held.append(0)
                        max_data_user_num -= 1
                    elif len(held) <= max_data_num - 2:
                        held.append(data)
                        if len(held) == max_data_num - 1:
                            processed_data += held
                            user_id_dict[user_id] = [0] * (max_data_num - 1)       # placeholders: block already released
                else:
                    user_id_dict[user_id] = [data]
                if len(processed_data) >= load_data_number:
                    break
        if len(processed_data) >= load_data_number:
            break
    if len(processed_data) < load_data_number:
        for data_list in user_id_dict.values():
            if len(data_list) < max_data_num - 1:
                processed_data += data_list
    return processed_data, max_user_id


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
