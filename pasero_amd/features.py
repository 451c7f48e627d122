"""Speech feature batches on the device: the collate step of the reference's speech path and the online
alternative to its offline feature files.

  * `collate_features`: `utils.tokens_as_tensor` for floating-point sequences (pasero/utils.py:709-736: zero padding to
    the longest sequence, cast to the model dtype, lengths) — but the ragged rows cross PCIe once, concatenated in the
    dtype they have on disk (NumpyFile rows are fp16, pasero/files.py:103-175), and `pk_pad_rows` pads and converts on
    the GPU.  Same values as the reference: fp16 -> bf16/fp32 conversion is exact-then-rounded exactly like `.to(dtype)`.
  * `wav_to_log_mel`: 16 kHz waveforms -> (B, 3000, 80) Whisper log-mel features with `pk_logmel`, instead of running
    examples/Whisper/extract-features.py offline and reading its fp16 file.
"""
from typing import List, Sequence, Tuple

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
    src = staging.to(device, non_blocking=True)
    off_dev = offsets.to(device, non_blocking=True)
    out = torch.empty(len(arrays), Tmax, D, dtype=dtype, device=device)
    check(lib.load().pk_pad_rows(ptr(src), _SRC_CODE[arrays[0].dtype], ptr(off_dev), ptr(out), dtype_code(out),
                                 len(arrays), Tmax, D, stream_ptr()), 'pk_pad_rows')
    return out, lengths


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
