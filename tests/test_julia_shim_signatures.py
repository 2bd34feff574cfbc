"""The Julia shims (slam.jl_amd/julia/*.jl) cannot run in this image (no Julia): every `ccall` in them is checked statically against the prototype
of the same symbol in include/slamhip.h -- the symbol exists, the return type, the number of arguments, each argument's class (pointer / 32-bit
integer / 64-bit integer / double) and, where the Julia side names one, the pointee type; and the call passes as many values as its type tuple has
entries.  A mismatch here is a segmentation fault or a silently wrong argument in a Julia process."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "slamhip.h")
SHIMS = [os.path.join(ROOT, "slam.jl_amd", "julia", f) for f in ("SLAMHip.jl", "SLAMHipStreams.jl")]
DOCS = [os.path.join(ROOT, "INTEGRATION.md")]                        # the binding snippets a maintainer would paste


def split_top(s, sep=","):
    """split at top-level separators (outside (), [], {}, strings)"""
    out, depth, cur, i, in_str = [], 0, [], 0, False
    while i < len(s):
        c = s[i]
        if in_str:
            cur.append(c)
            if c == "\\":
                cur.append(s[i + 1]); i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True; cur.append(c)
        elif c in "([{":
            depth += 1; cur.append(c)
        elif c in ")]}":
            depth -= 1; cur.append(c)
        elif c == sep and depth == 0:
            out.append("".join(cur).strip()); cur = []
        else:
            cur.append(c)
        i += 1
    if "".join(cur).strip():
        out.append("".join(cur).strip())
    return out


def matching(s, i):
    """index of the bracket closing the one at s[i] (strings skipped)"""
    depth, in_str = 0, False
    for k in range(i, len(s)):
        c = s[k]
        if in_str:
            if c == '"' and s[k - 1] != "\\":
                in_str = False
        elif c == '"':
            in_str = True
        elif c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return k
    raise ValueError("unbalanced")


C_INT32 = {"int", "int32_t", "unsigned", "unsigned int", "uint32_t"}
C_INT64 = {"int64_t", "long long", "uint64_t", "size_t", "unsigned long long"}


def c_class(t):
    """(class, pointee) of a C parameter type (the parameter name already removed)"""
    t = re.sub(r"\bconst\b", "", t).strip()
    t = re.sub(r"\s+", " ", t)
    if "*" in t:
        base = t.replace("*", "").strip()
        n = t.count("*")
        if n > 1:
            return "ptr", "ptr"
        return "ptr", base
    if t in C_INT32:
        return "i32", None
    if t in C_INT64:
        return "i64", None
    if t == "double":
        return "f64", None
    if t == "void":
        return "void", None
    raise AssertionError(f"header type not understood: {t!r}")


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(slam_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        if "typedef" in ret or "(" in ret:
            continue
        ps = []
        for p in split_top(params):
            p = re.sub(r"\s+", " ", p).strip()
            if p in ("void", ""):
                continue
            arr = re.match(r"^(.*?)([A-Za-z_]\w*)\s*\[[^\]]*\]$", p)          # array parameters decay to pointers
            if arr:
                ps.append(c_class(arr.group(1).strip() + " *")); continue
            mm = re.match(r"^(.*?)([A-Za-z_]\w*)?$", p)
            ty = mm.group(1).strip() if mm.group(2) and mm.group(1).strip() else p
            ps.append(c_class(ty))
        protos[name] = (c_class(ret), ps)
    return protos


JL_PTR_BASE = {"Cvoid": None, "Float64": "double", "Cdouble": "double", "Int32": "int32_t", "Cint": "int32_t", "Int64": "int64_t", "Clonglong": "int64_t",
               "UInt64": "uint64_t", "UInt8": "uint8_t", "Cuchar": "uint8_t", "Cchar": "char", "UInt32": "uint32_t", "Bool": "uint8_t"}


def jl_class(t):
    t = t.strip()
    m = re.match(r"^(Ptr|Ref)\{(.*)\}$", t)
    if m:
        inner = m.group(2).strip()
        if inner.startswith("Ptr{") or inner.startswith("Ref{"):
            return "ptr", "ptr"
        if inner not in JL_PTR_BASE:
            return "ptr", inner                                       # a Julia struct passed by reference (checked by name below)
        return "ptr", JL_PTR_BASE[inner]
    if t == "Cstring":
        return "ptr", "char"
    if t in ("Cint", "Int32", "Cuint", "UInt32"):
        return "i32", None
    if t in ("Int64", "Clonglong", "Csize_t", "UInt64", "Culonglong"):
        return "i64", None
    if t in ("Cdouble", "Float64"):
        return "f64", None
    if t == "Cvoid":
        return "void", None
    raise AssertionError(f"Julia ccall type not understood: {t!r}")


def shim_ccalls(paths=None):
    calls = []
    for path in (paths or SHIMS):
        src = open(path).read()
        if path.endswith(".md"):                                      # only the fenced code blocks
            src = "\n".join(b if k % 2 else "\n" * b.count("\n") for k, b in enumerate(src.split("```")))
        src = "\n".join(line.split("#", 1)[0] if '"' not in line else line for line in src.split("\n"))     # comments (lines with strings kept whole)
        for m in re.finditer(r"\bccall\(", src):
            end = matching(src, m.end() - 1)
            args = split_top(src[m.end():end])
            sym = re.match(r"^\(\s*:(\w+)\s*,", args[0])
            assert sym, f"{os.path.basename(path)}: ccall without a (:symbol, lib) tuple: {args[0]!r}"
            if not sym.group(1).startswith("slam_"):
                continue                                              # (the HIP runtime's own entry points: hipMalloc, hipMemcpyAsync ... of libamdhip64)
            tup = args[2].strip()
            assert tup.startswith("(") and tup.endswith(")"), (path, sym.group(1), tup)
            types = split_top(tup[1:-1])
            line = src.count("\n", 0, m.start()) + 1
            calls.append((os.path.basename(path), line, sym.group(1), args[1].strip(), types, args[3:]))
    return calls


def compatible(c, j):
    (cc, cb), (jc, jb) = c, j
    if cc != jc:
        return False
    if cc != "ptr" or jb is None or cb in ("void",):                  # Ptr{Cvoid} / void * match any pointer
        return True
    if cb == "ptr" or jb == "ptr":
        return cb == jb or cb.startswith("slam_") or jb is None
    if cb == jb:
        return True
    if cb == "int" and jb == "int32_t":
        return True
    if cb.startswith("slam_") or cb.startswith("struct"):             # opaque handles are Ptr{Cvoid}; config structs are Julia structs of the same name
        return jb is None or jb.lower().replace("_", "") in cb.lower().replace("_", "") or cb.replace("slam_", "").replace("_", "") in jb.lower()
    return False


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    assert len(protos) > 60, len(protos)
    calls = shim_ccalls()
    assert len(calls) >= 40, len(calls)
    bad = []
    for fname, line, sym, ret, types, values in calls:
        where = f"{fname}:{line} {sym}"
        if sym not in protos:
            bad.append(f"{where}: not declared in include/slamhip.h"); continue
        cret, cparams = protos[sym]
        if not compatible(cret, jl_class(ret)):
            bad.append(f"{where}: return type {ret} vs header {cret}")
        if len(types) != len(cparams):
            bad.append(f"{where}: {len(types)} argument types, the header declares {len(cparams)}"); continue
        if len(values) != len(types):
            bad.append(f"{where}: {len(values)} values passed for {len(types)} argument types")
        for k, (t, cp) in enumerate(zip(types, cparams)):
            if not compatible(cp, jl_class(t)):
                bad.append(f"{where}: argument {k + 1} is {t}, the header declares {cp}")
    assert not bad, "\n".join(bad)


def test_integration_md_snippets_match_the_header():
    protos = header_prototypes()
    calls = shim_ccalls(DOCS)
    assert len(calls) >= 10, len(calls)
    bad = []
    for fname, line, sym, ret, types, values in calls:
        where = f"{fname}:{line} {sym}"
        if sym not in protos:
            bad.append(f"{where}: not declared in include/slamhip.h"); continue
        cret, cparams = protos[sym]
        if not compatible(cret, jl_class(ret)):
            bad.append(f"{where}: return type {ret} vs header {cret}")
        if len(types) != len(cparams):
            bad.append(f"{where}: {len(types)} argument types, the header declares {len(cparams)}"); continue
        if values and values != ["..."] and len(values) != len(types):
            bad.append(f"{where}: {len(values)} values passed for {len(types)} argument types")
        for k, (t, cp) in enumerate(zip(types, cparams)):
            if not compatible(cp, jl_class(t)):
                bad.append(f"{where}: argument {k + 1} is {t}, the header declares {cp}")
    assert not bad, "\n".join(bad)


def test_the_six_seams_are_bound():
    syms = {c[2] for c in shim_ccalls()}
    for s in ("slam_detect", "slam_describe", "slam_pyr_update", "slam_fb_track", "slam_triangulate", "slam_local_ba", "slam_local_ba_batch", "slam_frontend_step"):
        assert s in syms, s


def test_frontend_config_struct_matches_field_for_field():
    """slam_frontend_config is passed by reference from a Julia mutable struct: same fields, same order, same widths"""
    hdr = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct slam_frontend_config\s*\{(.*?)\}\s*slam_frontend_config\s*;", hdr, flags=re.S).group(1)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        c_fields += [(n.strip(), ty) for n in names.split(",")]
    src = open(SHIMS[1]).read()
    jl = re.search(r"mutable struct FrontEndConfig\n(.*?)\nend", src, flags=re.S).group(1)
    j_fields = [(m.group(1), m.group(2)) for m in re.finditer(r"(\w+)::(\w+)", jl)]
    width = {"int32_t": "Int32", "double": "Float64"}
    assert [(n, width[t]) for n, t in c_fields] == j_fields
    # ... and the Python mirror (the ctypes Structure of slam.jl_amd/frontend.py)
    import ctypes as C
    from slam_jl_amd.frontend import FrontEndConfig
    py = [(n, {C.c_int32: "Int32", C.c_double: "Float64", C.c_int: "Int32"}[t]) for n, t in FrontEndConfig._fields_]
    assert py == j_fields
