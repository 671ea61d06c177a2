"""CPU checks: the C-ABI library loads and exports every symbol include/c2mi.h declares (no compute)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "c2mi.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char\*)\s+(c2_[a-z0-9_]+)\(", hdr, flags=re.M))
    assert len(declared) >= 25
    lib = ctypes.CDLL(os.path.join(ROOT, "celligner2_amd", "lib", "libc2mi.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in c2mi.h but not exported"
    from celligner2_amd import _lib
    assert set(_lib.EXPORTED) == declared, set(_lib.EXPORTED) ^ declared


def test_no_cpu_fallback():
    """The product refuses CPU tensors instead of computing on the host."""
    import pytest
    import torch
    from celligner2_amd import ops
    from celligner2_amd._lib import C2Error
    with pytest.raises(C2Error):
        ops.prep_input(torch.zeros(2, 4), torch.zeros(2, 4), True)
    from celligner2_amd.model import Celligner2
    m = Celligner2(16, [8, 4], 2, 0.0, conditions=["a", "b"])
    with pytest.raises(C2Error):
        m(x=torch.zeros(3, 16), batch=torch.tensor([[1, 0]] * 3), sizefactor=torch.ones(3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "celligner2_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn
