"""A machine check that julia/CoordinateDescentHIP.jl and include/cdhip.h agree (no `julia` exists in this
pipeline, so the binding never runs here; tests/test_binding_call_sequence.py replays its call sequences, and
this file pins what a replay cannot: that every `ccall((:name, libcdhip), Ret, (types...), ...)` in the .jl
names an exported function, with the arity and the C types of its prototype, and that the two mirrored structs
carry the header's fields in the header's order)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "CoordinateDescentHIP.jl")
HDR = os.path.join(ROOT, "include", "cdhip.h")

# C parameter type (qualifiers and names stripped) -> the Julia ccall types that may stand for it
C_TO_JULIA = {
    "cdh_handle": {"Ptr{Cvoid}"},
    "cdh_handle*": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "void*": {"Ptr{Cvoid}"},
    "int32_t": {"Int32"},
    "int64_t": {"Int64"},
    "uint64_t": {"UInt64"},
    "double": {"Float64"},
    "int32_t*": {"Ptr{Int32}", "Ref{Int32}"},
    "int64_t*": {"Ptr{Int64}", "Ref{Int64}"},
    "double*": {"Ptr{Float64}", "Ref{Float64}"},
    "cdh_options*": {"Ref{CdhOptions}", "Ptr{CdhOptions}"},
    "cdh_stats*": {"Ref{CdhStats}", "Ptr{CdhStats}"},
    "char*": {"Cstring"},
}
C_FIELD_TO_JULIA = {"int64_t": "Int64", "int32_t": "Int32", "uint64_t": "UInt64", "double": "Float64"}


def _strip_comments(txt):
    return re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


def _ctype(param):
    """'const int64_t *idx1' -> 'int64_t*'; 'cdh_handle h' -> 'cdh_handle'."""
    t = param.replace("const", " ").strip()
    stars = t.count("*")
    t = t.replace("*", " ")
    words = t.split()
    base = words[0]
    return base + "*" * stars


def header_prototypes():
    txt = _strip_comments(open(HDR).read())
    txt = "\n".join(l for l in txt.splitlines() if not l.strip().startswith("typedef int32_t (*"))
    protos = {}
    for ret, name, params in re.findall(r"(int32_t|const char \*)\s*(cdh_\w+)\s*\(([^)]*)\)\s*;", txt):
        ps = [p.strip() for p in params.split(",") if p.strip() and p.strip() != "void"]
        protos[name] = ("char*" if "char" in ret else "int32_t", [_ctype(p) for p in ps])
    return protos


def header_struct(name):
    txt = _strip_comments(open(HDR).read())
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, flags=re.S).group(1)
    return [(t, f) for t, f in re.findall(r"(\w+)\s+(\w+)\s*;", body)]


def julia_ccalls():
    src = open(JL).read()
    src = "\n".join(l.split("#", 1)[0] if not l.lstrip().startswith('"') else l for l in src.splitlines())
    calls = []
    for name, ret, types in re.findall(r"ccall\(\(:(\w+),\s*libcdhip\),\s*(\w+),\s*\(([^)]*)\)", src, flags=re.S):
        ts = [t.strip() for t in types.split(",") if t.strip()]
        calls.append((name, ret, ts))
    return calls


def julia_struct(name):
    src = open(JL).read()
    body = re.search(r"struct %s\b(.*?)\nend" % name, src, flags=re.S).group(1)
    body = "\n".join(l.split("#", 1)[0] for l in body.splitlines())
    return [(f, t) for f, t in re.findall(r"(\w+)::(\w+)", body)]


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    assert len(protos) >= 45 and "cdh_coordinate_descent" in protos and "cdh_cache_drift" in protos
    calls = julia_ccalls()
    assert len(calls) >= 25
    for name, ret, types in calls:
        assert name in protos, f"{name} is called by the binding but not declared in cdhip.h"
        cret, cparams = protos[name]
        assert ret in C_TO_JULIA[cret], (name, ret, cret)
        assert len(types) == len(cparams), f"{name}: {len(types)} ccall argument types, prototype has {len(cparams)}"
        for i, (jt, ct) in enumerate(zip(types, cparams)):
            assert jt in C_TO_JULIA[ct], f"{name}: argument {i + 1} is {jt} in the binding, `{ct}` in cdhip.h"


def test_the_binding_covers_the_entry_points_the_front_ends_need():
    called = {c[0] for c in julia_ccalls()}
    need = {"cdh_create", "cdh_destroy", "cdh_set_X_cols", "cdh_set_y", "cdh_set_obs_weights", "cdh_set_loss",
            "cdh_set_penalty", "cdh_initialize", "cdh_set_iterate", "cdh_gradient", "cdh_descend",
            "cdh_coordinate_descent", "cdh_get_support", "cdh_get_beta", "cdh_get_residual", "cdh_col_rms",
            "cdh_xt_r", "cdh_gram", "cdh_resid_moments", "cdh_set_reuse_residual", "cdh_set_gradient_cache",
            "cdh_get_gradient_cache", "cdh_last_error"}
    assert need <= called, need - called


def test_mirrored_structs_carry_the_headers_fields_in_order():
    for cname, jname in (("cdh_options", "CdhOptions"), ("cdh_stats", "CdhStats")):
        c = header_struct(cname)
        j = julia_struct(jname)
        assert [f for _, f in c] == [f for f, _ in j], (cname, c, j)
        for (ct, cf), (jf, jt) in zip(c, j):
            assert C_FIELD_TO_JULIA[ct] == jt, (cname, cf, ct, jt)


def test_owner_is_a_strong_reference_not_an_object_id():
    """objectid of a Vector is its address and can be reused by the next copy(y)."""
    src = open(JL).read()
    assert "owner::Any" in src and "X.owner === f.r" in src
    code = "\n".join(l.split("#", 1)[0] for l in src.splitlines())
    assert "objectid(" not in code
