"""Speech feature batches on the device: the collate step of the reference's speech path and the online
alternative to its offline feature files.

  * `collate_features`: `utils.tokens_as_tensor` for floating-point sequences (pasero/utils.py:709-736: zero padding to
    the longest sequence, cast to the model dtype, lengths) — but the ragged rows cross PCIe once, concatenated in the
    dtype they have on disk (NumpyFile rows are fp16, pasero/files.py:103-175), and `pk_pad_rows` pads and converts on
    the GPU.  Same values as the reference: fp16 -> bf16/fp32 conversion is exact-then-rounded exactly like `.to(dtype)`.
  * `NumpyFile` + `collate_from_file`: the reference's feature-file reader (pasero/files.py:103-193: a pickled header
    {positions, lengths, dim, dtype} followed by the raw rows) with the same iteration surface, plus a batched read
    that puts the rows of a batch STRAIGHT into the pinned staging buffer (`readinto`, no intermediate bytes / numpy
    copies: the reference's `__next__` makes two per row, then `pad_sequence` a third and `.to(dtype)` a fourth).
  * `wav_to_log_mel`: 16 kHz waveforms -> (B, 3000, 80) Whisper log-mel features with `pk_logmel`, instead of running
    examples/Whisper/extract-features.py offline and reading its fp16 file.
"""
import io
import pickle
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import functional as F
from . import lib
from .lib import check, dtype_code, ptr, stream_ptr

_SRC_CODE = {np.dtype('float32'): 0, np.dtype('float16'): 2}


def collate_features(token_list: Sequence[np.ndarray], dtype: torch.dtype, device) -> Tuple[Tensor, Tensor]:
    """token_list: B arrays (T_b, D) of float16 / float32 features -> (tokens (B, Tmax, D) `dtype` on `device`, zero
    padded; lengths (B,) int64 on the CPU like the reference's)"""
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('pasero_amd.features.collate_features needs a GPU (no CPU fallback)')
    arrays = [np.ascontiguousarray(a) for a in token_list]
    assert arrays and all(a.ndim == 2 and a.shape[1] == arrays[0].shape[1] and a.dtype == arrays[0].dtype for a in arrays)
    if arrays[0].dtype not in _SRC_CODE:
        raise TypeError(f'features must be float16 or float32, got {arrays[0].dtype}')
    D = arrays[0].shape[1]
    lengths = torch.tensor([a.shape[0] for a in arrays], dtype=torch.int64)
    offsets = torch.zeros(len(arrays) + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(lengths, 0)
    total, Tmax = int(offsets[-1]), int(lengths.max())
    staging = torch.empty(max(total, 1), D, dtype=torch.float16 if arrays[0].dtype == np.float16 else torch.float32,
                          pin_memory=True)
    view = staging.numpy()
    for a, o in zip(arrays, offsets[:-1].tolist()):
        view[o:o + a.shape[0]] = a
    return _pad_on_device(staging, lengths, D, dtype, device), lengths


def _pad_on_device(staging: Tensor, lengths: Tensor, D: int, dtype: torch.dtype, device) -> Tensor:
    """staging: pinned (total, D) rows of the batch back to back -> (B, Tmax, D) zero padded, `dtype`, on `device`"""
    offsets = torch.zeros(lengths.numel() + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(lengths, 0)
    Tmax = int(lengths.max()) if lengths.numel() else 0
    src = staging.to(device, non_blocking=True)
    off_dev = offsets.to(device, non_blocking=True)
    out = torch.empty(lengths.numel(), Tmax, D, dtype=dtype, device=device)
    src_code = _SRC_CODE[np.dtype('float16') if staging.dtype == torch.float16 else np.dtype('float32')]
    check(lib.load().pk_pad_rows(ptr(src), src_code, ptr(off_dev), ptr(out), dtype_code(out), lengths.numel(), Tmax, D,
                                 stream_ptr()), 'pk_pad_rows')
    return out


class NumpyFile:
    """Reader of the reference's 'numpy' corpus format (pasero/files.py:103-193; written by `NumpyFile.build`, e.g. by
    examples/Whisper/extract-features.py:164).  Same surface as the reference class for what the data pipeline calls:
    `get_positions()` -> (indices, lengths), `seek(index)`, `tell()`, `next()` -> (length, dim) array, iteration,
    `close()` / `reopen()`.  `source` is a path or the file's bytes (the reference's `store_files_under` in-memory
    mode, files.py:39-47)."""

    def __init__(self, source, store_files_under: Optional[int] = None):
        self._path = None if isinstance(source, (bytes, bytearray, memoryview)) else source
        self._file = io.BytesIO(bytes(source)) if self._path is None else open(self._path, 'rb')
        if self._path is not None and store_files_under:
            self._file.seek(0, io.SEEK_END)
            size = self._file.tell()
            self._file.seek(0)
            if size <= store_files_under:
                content = self._file.read()
                self._file.close()
                self._file = io.BytesIO(content)
        header = pickle.load(self._file)
        self._dim = int(header['dim'])
        self._dtype = np.dtype(header['dtype'])
        if self._dtype not in _SRC_CODE:
            raise TypeError(f'feature files must hold float16 or float32 rows, got {self._dtype}')
        self._itemsize = self._dim * self._dtype.itemsize
        pos = np.asarray(header['positions'], dtype=np.int64)
        keep = pos > 0  # slots that were never written (num_feats larger than the corpus, files.py:113-116)
        self._positions = pos[keep]
        self._lengths = np.asarray(header['lengths'], dtype=np.int64)[keep]
        self._indices = np.arange(len(self._positions))
        self._index = 0

    dim = property(lambda self: self._dim)
    dtype = property(lambda self: self._dtype)

    def __len__(self) -> int:
        return len(self._positions)

    def get_positions(self) -> Tuple[np.ndarray, np.ndarray]:
        return self._indices, self._lengths

    def close(self):
        if not isinstance(self._file, io.BytesIO) and not self._file.closed:
            self._file.close()

    def reopen(self):
        if self._file.closed:
            self._file = open(self._path, 'rb')
            if self._index < len(self._positions):
                self._file.seek(self._positions[self._index])

    def seek(self, offset, whence=0):
        self._index = int(offset)
        self.reopen()
        self._file.seek(self._positions[self._index], whence)

    def tell(self) -> int:
        return self._index

    def __next__(self) -> np.ndarray:
        self.reopen()
        if self._index >= len(self._positions):
            raise StopIteration
        length = int(self._lengths[self._index])
        x = np.empty(length * self._dim, dtype=self._dtype)
        got = self._file.readinto(memoryview(x).cast('B')) if x.size else 0
        if got != x.nbytes:
            raise EOFError(f'feature file truncated: row {self._index} needs {x.nbytes} bytes, got {got}')
        if self._dim > 1:
            x = x.reshape(length, self._dim)
        self._index += 1
        return x

    def __iter__(self):
        while True:
            try:
                yield next(self)
            except StopIteration:
                break

    def read_rows_into(self, indices: Sequence[int], staging: np.ndarray) -> np.ndarray:
        """rows `indices` (in this order) back to back into `staging` ((>= total, dim) array of the file's dtype,
        typically the numpy view of a pinned tensor) with one positioned read each -> their lengths"""
        self.reopen()
        lengths = self._lengths[np.asarray(indices, dtype=np.int64)]
        flat = memoryview(staging).cast('B')
        o = 0
        for i, n in zip(indices, lengths.tolist()):
            nbytes = n * self._itemsize
            if nbytes:
                self._file.seek(self._positions[i])
                got = self._file.readinto(flat[o:o + nbytes])
                if got != nbytes:
                    raise EOFError(f'feature file truncated: row {i} needs {nbytes} bytes, got {got}')
            o += nbytes
        self._index = int(indices[-1]) + 1 if len(indices) else self._index
        return lengths


def collate_from_file(file: NumpyFile, indices: Sequence[int], dtype: torch.dtype, device) -> Tuple[Tensor, Tensor]:
    """`utils.tokens_as_tensor([file rows ...], dtype=dtype)` (pasero/utils.py:709-736) for the rows `indices` of a
    feature file: file -> pinned staging (one `readinto` per row) -> one host-to-device copy -> `pk_pad_rows` pads and
    converts on the GPU.  -> (tokens (B, Tmax, D) on `device`, lengths (B,) int64 on the CPU like the reference's)"""
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('pasero_amd.features.collate_from_file needs a GPU (no CPU fallback)')
    if file.dim <= 1:
        raise ValueError('collate_from_file: the file holds scalars per position, not feature rows')
    lengths_np = file.get_positions()[1][np.asarray(indices, dtype=np.int64)]
    total = int(lengths_np.sum())
    staging = torch.empty(max(total, 1), file.dim, dtype=torch.float16 if file.dtype == np.float16 else torch.float32,
                          pin_memory=True)
    file.read_rows_into(indices, staging.numpy())
    lengths = torch.from_numpy(lengths_np.astype(np.int64))
    return _pad_on_device(staging, lengths, file.dim, dtype, device), lengths


def wav_to_log_mel(wavs: List[np.ndarray], device, dtype: torch.dtype = torch.float32) -> Tuple[Tensor, Tensor]:
    """wavs: B mono 16 kHz float waveforms (any lengths; > 30 s is truncated like the feature extractor does) ->
    (features (B, 3000, 80), lengths (B,) = 3000 frames each: Whisper pads every clip to 30 s)"""
    n = 480000
    batch = torch.zeros(len(wavs), n, dtype=torch.float32, pin_memory=True)
    lens = torch.empty(len(wavs), dtype=torch.int64)
    for i, w in enumerate(wavs):
        w = np.asarray(w, dtype=np.float32).reshape(-1)[:n]
        batch[i, :len(w)] = torch.from_numpy(w)
        lens[i] = len(w)
    feats = F.log_mel(batch.to(device, non_blocking=True), lens.to(device, non_blocking=True))
    return feats.to(dtype), torch.full((len(wavs),), 3000, dtype=torch.int64)
