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
    assert lib.llmseg_version() == _lib.ABI_VERSION == int(re.search(r"#define LLMSEG_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "llmseg_hip.h")).read()).group(1))


def _header_struct_fields(name):
    """Field names of `typedef struct { ... } name;` in declaration order."""
    src = open(os.path.join(ROOT, "include", "llmseg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    end = re.search(r"\}\s*" + name + r"\s*;", src).start()
    body = src[src.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        out.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1])
        out += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", r)[-1] for r in rest]
    return out


def test_struct_layouts_header_binding_library_and_integration_stub():
    """the binding's structs are the header's field for field, their sizes are the built library's, and the ctypes stub
    INTEGRATION.md shows a reference maintainer declares the same struct (it was 16 bytes short in round 2)."""
    import ctypes as C
    from llmseg_amd import _lib
    lib = _lib.load()
    for which, (cname, st) in enumerate((("llmseg_gemm_args", _lib.GemmArgs), ("llmseg_attn_args", _lib.AttnArgs),
                                         ("llmseg_attn_bwd_args", _lib.AttnBwdArgs), ("llmseg_dropout", _lib.Dropout))):
        assert [f[0] for f in st._fields_] == _header_struct_fields(cname), cname
        assert lib.llmseg_struct_size(which) == C.sizeof(st), cname
    assert lib.llmseg_struct_size(99) == -1
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"class GemmArgs\(C\.Structure\):.*?\n(?=_lib\.llmseg_struct_size)", md, flags=re.S).group(0)
    ns = {"C": C}
    exec(block, ns)
    stub = ns["GemmArgs"]
    assert [f[0] for f in stub._fields_] == [f[0] for f in _lib.GemmArgs._fields_]
    assert [C.sizeof(f[1]) for f in stub._fields_] == [C.sizeof(f[1]) for f in _lib.GemmArgs._fields_]
    assert C.sizeof(stub) == C.sizeof(_lib.GemmArgs) == lib.llmseg_struct_size(0)
    assert "struct_size=C.sizeof(GemmArgs)" in md


def test_abi_guard_rejects_a_stale_struct():
    """A caller built against an older header (round 2's struct: no struct_size, 3 fewer tail fields) is refused with LLMSEG_EINVAL
    before any field is read -- no device needed: the guard is the first statement of the entry point."""
    import ctypes as C
    from llmseg_amd import _lib
    lib = _lib.load()
    g = _lib.GemmArgs(M=1, N=1, K=8)
    assert g.struct_size == C.sizeof(_lib.GemmArgs)
    g.struct_size -= 16
    assert lib.llmseg_gemm_bf16(C.byref(g), None) == -1 and b"ABI mismatch" in lib.llmseg_last_error()
    a = _lib.AttnArgs()
    a.struct_size = 0
    assert lib.llmseg_attn_fwd(C.byref(a), None) == -1 and b"ABI mismatch" in lib.llmseg_last_error()
    b = _lib.AttnBwdArgs()
    b.struct_size += 8
    assert lib.llmseg_attn_bwd(C.byref(b), None) == -1 and b"ABI mismatch" in lib.llmseg_last_error()


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under llmseg_amd/ may reference it."""
    pkg = os.path.join(ROOT, "llmseg_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "from .. import oracle" not in txt and "import oracle" not in txt, f
