"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    import kalign_amd
    from kalign_amd import api
    if not os.path.exists(api.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    L = kalign_amd.load_library()
    hdr = open(os.path.join(ROOT, "include", "kalign_amd.h")).read()
    declared = set(re.findall(r"\b(ka_[a-z_]+)\s*\(", hdr))
    assert declared == set(api.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.ka_abi_version() >= 1


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product must fail loudly, not compute on the CPU."""
    import kalign_amd
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(kalign_amd.KalignAmdError):
        kalign_amd.Context(0)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kalign_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libkalign_oracle" not in src \
                    and "libkalign_ref" not in src, f
