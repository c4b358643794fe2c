"""CPU: the C-ABI library builds, loads and exports every symbol include/llmseg_hip.h declares (no compute calls)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "llmseg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(llmseg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_bound_and_exported():
    from llmseg_amd import _lib
    decl = _declared()
    assert len(decl) >= 18
    assert sorted(_lib.SIGNATURES) == decl, (sorted(set(decl) ^ set(_lib.SIGNATURES)))
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.llmseg_version() >= 1


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under llmseg_amd/ may reference it."""
    pkg = os.path.join(ROOT, "llmseg_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "from .. import oracle" not in txt and "import oracle" not in txt, f
