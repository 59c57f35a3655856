"""Serialise a CompiledModel into the flat, name-tagged blob `vnl_model_create` takes.

Layout (little endian):
    char[8]  magic "VNLMDL01"
    u32      nsections, u32 reserved
    nsections x { char name[24]; u32 dtype (1=f64, 2=i32); u32 count; u64 offset }
    payload (each section 8-byte aligned)
Scalars are 1-element f64 sections.  Consumers look sections up by name, so the
blob can grow without breaking older readers.
"""
from __future__ import annotations

import struct

import numpy as np

from .mjcf import CompiledModel

MAGIC = b"VNLMDL01"
_ENTRY = struct.Struct("<24sIIQ")


def to_blob(model: CompiledModel) -> bytes:
    sections = []
    for k, v in model.scalars.items():
        sections.append((k, 1, np.array([v], dtype="<f8")))
    for k, v in model.arrays.items():
        v = np.asarray(v)
        if v.dtype.kind in "iu":
            sections.append((k, 2, np.ascontiguousarray(v, dtype="<i4").ravel()))
        else:
            sections.append((k, 1, np.ascontiguousarray(v, dtype="<f8").ravel()))
    header_len = 16 + _ENTRY.size * len(sections)
    offset = (header_len + 7) // 8 * 8
    entries, payload = [], bytearray()
    for name, dtype, arr in sections:
        raw = arr.tobytes()
        nm = name.encode()
        if len(nm) > 23:
            raise ValueError(f"section name too long: {name}")
        entries.append(_ENTRY.pack(nm, dtype, arr.size, offset + len(payload)))
        payload += raw
        payload += b"\0" * (-len(raw) % 8)
    head = MAGIC + struct.pack("<II", len(sections), 0) + b"".join(entries)
    head += b"\0" * (offset - len(head))
    return bytes(head + payload)
