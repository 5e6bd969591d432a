"""Host-side mirror of the reference's schema types (names and values identical).

Reference: cloudini_lib/include/cloudini_lib/basic_types.hpp:28-67 (FieldType, PointField),
cloudini_lib/include/cloudini_lib/cloudini.hpp:33-111 (EncodingOptions, CompressionOption,
kEncodingVersion, EncodingInfo).
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import List, Optional

kEncodingVersion = 5  # cloudini.hpp:63
kPointsPerChunk = 32 * 1024  # src/codec_common.hpp:28
kDecodeButSkipStore = 0xFFFFFFFF  # basic_types.hpp:71


class FieldType(enum.IntEnum):
    UNKNOWN = 0
    INT8 = 1
    UINT8 = 2
    INT16 = 3
    UINT16 = 4
    INT32 = 5
    UINT32 = 6
    FLOAT32 = 7
    FLOAT64 = 8
    INT64 = 9
    UINT64 = 10


class EncodingOptions(enum.IntEnum):
    NONE = 0
    LOSSY = 1
    LOSSLESS = 2


class CompressionOption(enum.IntEnum):
    NONE = 0
    LZ4 = 1
    ZSTD = 2


_SIZEOF = {
    FieldType.INT8: 1, FieldType.UINT8: 1, FieldType.INT16: 2, FieldType.UINT16: 2,
    FieldType.INT32: 4, FieldType.UINT32: 4, FieldType.FLOAT32: 4, FieldType.FLOAT64: 8,
    FieldType.INT64: 8, FieldType.UINT64: 8,
}


def SizeOf(t: FieldType) -> int:
    """basic_types.hpp:73-95."""
    return _SIZEOF.get(FieldType(t), 0)


@dataclass
class PointField:
    name: str
    offset: int = 0
    type: FieldType = FieldType.UNKNOWN
    resolution: Optional[float] = None  # max quantisation error is 0.5 * resolution


@dataclass
class EncodingInfo:
    fields: List[PointField] = field(default_factory=list)
    width: int = 0
    height: int = 1
    point_step: int = 0
    encoding_opt: EncodingOptions = EncodingOptions.LOSSY
    encoding_config: str = ""
    compression_opt: CompressionOption = CompressionOption.ZSTD
    use_threads: bool = True
    version: int = kEncodingVersion

    def copy(self, **kw) -> "EncodingInfo":
        import copy as _copy

        out = _copy.deepcopy(self)
        for k, v in kw.items():
            setattr(out, k, v)
        return out
