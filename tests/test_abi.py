"""The C-ABI library loads and exports every symbol include/kimi_hip.h declares (no compute, no GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    from kimimaro_amd import _abi, build
    if not os.path.exists(_abi.LIB_PATH):
        build.build()
    L = _abi.lib()
    header = open(os.path.join(ROOT, "include", "kimi_hip.h")).read()
    declared = set(re.findall(r"\b(kh_[a-z0-9_]+)\s*\(", header))
    declared.discard("kh_label_t")
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libkimi_hip.so does not export %s" % name
    assert declared == set(_abi.SYMBOLS)
    assert L.kh_version() >= 100


def test_label_struct_matches_header():
    from kimimaro_amd import _abi
    header = open(os.path.join(ROOT, "include", "kimi_hip.h")).read()
    body = header[header.index("typedef struct kh_label_t {"):header.index("} kh_label_t;")]
    fields = re.findall(r"\b(?:uint32_t|float)\s+([A-Za-z_0-9, ]+);", body)
    names = [n.strip() for f in fields for n in f.split(",")]
    assert names == list(_abi.LABEL_T.names)


def test_product_fails_loudly_without_gpu():
    import numpy as np
    import torch
    import kimimaro_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(kimimaro_amd.HipUnavailableError):
        kimimaro_amd.skeletonize(np.ones((16, 16, 16), np.uint32), dust_threshold=0)


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "kimimaro_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), fn
