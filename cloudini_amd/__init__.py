"""cloudini_amd: MI355X-native stage-1 point-cloud codec behind Cloudini's PointcloudEncoder/Decoder API.

Importing the package is cheap (schema types only). The HIP library is loaded on first use by
`cloudini_amd.native`; if it has not been built (`python -c "import __graft_entry__ as g; g.build()"`)
that load raises -- there is no CPU fallback in the product path.
"""
from .schema import (CompressionOption, EncodingInfo, EncodingOptions, FieldType, PointField, SizeOf,
                     kEncodingVersion, kPointsPerChunk)

__all__ = ["CompressionOption", "EncodingInfo", "EncodingOptions", "FieldType", "PointField", "SizeOf",
           "kEncodingVersion", "kPointsPerChunk"]
