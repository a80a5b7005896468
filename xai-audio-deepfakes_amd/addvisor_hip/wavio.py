"""Minimal RIFF/WAVE reader and writer (the reference uses torchaudio.load / torchaudio.save / librosa.load:
audioprocessor.py:50, hifigan.py:156, 226-229; neither is needed here).

Reads PCM 8/16/24/32-bit, IEEE float 32/64-bit and WAVE_FORMAT_EXTENSIBLE wrappers of those, any channel count;
writes IEEE float32 (what ``torchaudio.save`` emits for a float tensor) or PCM16.  SURVEY.md §8(f) rank 3."""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np
import torch


def read_wav(path) -> Tuple[torch.Tensor, int]:
    """-> ``(audio [channels, frames] fp32 in [-1, 1), sample_rate)`` like ``torchaudio.load``."""
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, raw = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:            # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the tag
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            raw = body
        pos += 8 + size + (size & 1)
    if fmt is None or raw is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1:                                             # integer PCM
        if bits == 8:
            a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(raw[:len(raw) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            a = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
        elif bits == 32:
            a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:                                           # IEEE float
        if bits not in (32, 64):
            raise ValueError(f"{path}: unsupported float width {bits}")
        a = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    a = a[:len(a) // ch * ch].reshape(-1, ch).T
    return torch.from_numpy(np.ascontiguousarray(a)), int(sr)


def write_wav(path, audio: torch.Tensor, sample_rate: int, encoding: str = "float32") -> None:
    """``audio [channels, frames]`` or ``[frames]`` -> RIFF/WAVE.  ``encoding``: ``"float32"`` (format tag 3, the
    ``torchaudio.save`` default for float tensors) or ``"pcm16"`` (clipped to [-1, 1))."""
    a = audio.detach().to("cpu", torch.float32)
    if a.dim() == 1:
        a = a[None]
    if a.dim() != 2:
        raise ValueError("audio must be [channels, frames] or [frames]")
    ch, n = a.shape
    inter = np.ascontiguousarray(a.numpy().T)
    if encoding == "float32":
        tag, bits, payload = 3, 32, inter.astype("<f4").tobytes()
    elif encoding == "pcm16":
        tag, bits = 1, 16
        payload = np.clip(np.round(inter * 32768.0), -32768, 32767).astype("<i2").tobytes()
    else:
        raise ValueError(encoding)
    block = ch * bits // 8
    fmt = struct.pack("<HHIIHH", tag, ch, int(sample_rate), int(sample_rate) * block, block, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if tag == 3:
        chunks += b"fact" + struct.pack("<II", 4, n)
    chunks += b"data" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
