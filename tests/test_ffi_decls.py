"""The Rust side of the seam, checked mechanically (there is no rustc here): include/mzk_ffi.rs -- the
generated, complete `extern "C"` block -- and the annotated excerpt in INTEGRATION.md section 2 are parsed back and compared
with every prototype of include/mzk.h: name, arity, and each parameter's kind (integer width / pointer depth / const-ness /
opaque handle / callback).  A signature change in the header that is not carried into the Rust declarations fails here."""
import os, re, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as g


@pytest.fixture(scope="module")
def c_protos():
    return {name: (ret, [t for _, t in params]) for name, ret, params in g.c_prototypes()}


def test_generated_ffi_file_is_current_and_complete(c_protos):
    path = os.path.join(ROOT, "include", "mzk_ffi.rs")
    assert open(path).read() == g.emit(), "include/mzk_ffi.rs is stale: run python tools/gen_rust_ffi.py"
    decls = g.rust_decls(open(path).read())
    import myzkp_amd
    assert sorted(decls) == sorted(c_protos) == sorted(myzkp_amd.DECLARED_SYMBOLS)
    assert decls == c_protos


def test_c_type_mapping_examples():
    assert g.c_type("const uint64_t*") == "*const u64"
    assert g.c_type("uint64_t *") == "*mut u64"
    assert g.c_type("const void* const*") == "*const *const c_void"
    assert g.c_type("void* const*") == "*const *mut c_void"
    assert g.c_type("mzk_merkle**") == "*mut *mut c_void"
    assert g.c_type("const mzk_merkle* const*") == "*const *const c_void"
    assert g.c_type("const char*") == "*const c_char"
    assert g.c_param("uint64_t out_xy[8]", 0) == ("out_xy", "*mut u64")
    assert g.c_param("const uint64_t alpha[4]", 0) == ("alpha", "*const u64")
    assert g.c_param("mzk_fri_challenge_fn challenge", 0)[1].startswith('extern "C" fn(')


def _kind(t):
    """what must agree between a hand-written declaration and the header: spaces and `/* comments */` aside, the type itself"""
    return " ".join(re.sub(r"/\*.*?\*/", "", t).split())


def test_integration_md_extern_block_matches_the_header(c_protos):
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = md.index('extern "C" {', md.index("## 2."))
    block = md[start:md.index("\n}\n", start)]
    decls = g.rust_decls(block)
    assert len(decls) >= 50
    problems = []
    for name, (ret, types) in decls.items():
        if name not in c_protos:
            problems.append("%s: not in include/mzk.h" % name)
            continue
        cret, ctypes_ = c_protos[name]
        if ret != cret:
            problems.append("%s: returns %s, header says %s" % (name, ret, cret))
        if len(types) != len(ctypes_):
            problems.append("%s: %d parameters, header has %d" % (name, len(types), len(ctypes_)))
            continue
        for i, (a, b) in enumerate(zip(types, ctypes_)):
            if _kind(a) != _kind(b):
                problems.append("%s: parameter %d is %s, header says %s" % (name, i, _kind(a), _kind(b)))
    assert not problems, "\n".join(problems)
