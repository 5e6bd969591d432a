#!/usr/bin/env python3
"""Freeze the reference's applyVizLossyPreprocessing outputs for tests/test_viz_preprocess.py::_schema_gate_cases
(needs oracle/_ref, i.e. /root/reference): python tests/golden/make_viz_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import test_viz_preprocess as t  # noqa: E402
from oracle.binding import RefLib  # noqa: E402

ref = RefLib()
arrays = {}
for name, info, data in t._schema_gate_cases():
    out, res, w, h = ref.viz_preprocess(info, data)
    arrays[name + "/out"] = out
    arrays[name + "/res"] = np.array(res, dtype=np.float32)
    arrays[name + "/shape"] = np.array([w, h], dtype=np.int64)
np.savez_compressed(os.path.join(HERE, "viz_golden.npz"), **arrays)
print("wrote", len(arrays) // 3, "cases")
