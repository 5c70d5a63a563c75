"""The Panama binding under java/ cannot be compiled here (no JVM in this image), so nothing else would notice a downcall descriptor that
drifted away from include/mecano_hip.h.  This test parses both sides as text and diffs them:

* every `handle("mh_...", <descriptor>)` of MecanoHipNative.java against the C prototype of that name (return kind, argument kinds in
  order: pointer -> ADDRESS, int64_t / size_t -> JAVA_LONG, int32_t / enums -> JAVA_INT, double -> JAVA_DOUBLE);
* the StructLayouts of mh_options and mh_model_desc against the C structs (member names, kinds, order, natural-alignment padding), and the
  byte offsets MecanoHipNative.options(...) writes to;
* the ABI version constant of the binding against MH_ABI_VERSION, and the ctypes mirror (mecano_amd/_lib.py) against the same structs.
"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "mecano_hip.h")).read()
NATIVE = open(os.path.join(ROOT, "java", "us", "ihmc", "mecano", "hip", "MecanoHipNative.java")).read()
MODEL = open(os.path.join(ROOT, "java", "us", "ihmc", "mecano", "hip", "HipMultiBodyModel.java")).read()


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_kind(decl):
    """'const double *q' -> ADDRESS, 'int64_t B' -> JAVA_LONG, ..."""
    decl = decl.strip()
    if "*" in decl or "[" in decl or re.match(r"(const\s+)?mh_\w+_t\b", decl):  # opaque handles (mh_model_t, mh_timer_t) are pointers
        return "ADDRESS"
    base = decl.split()[0] if decl.split()[0] != "const" else decl.split()[1]
    return {"int64_t": "JAVA_LONG", "size_t": "JAVA_LONG", "uint64_t": "JAVA_LONG", "int32_t": "JAVA_INT", "uint32_t": "JAVA_INT", "int": "JAVA_INT",
            "mh_status": "JAVA_INT", "mh_layout": "JAVA_INT", "mh_joint_type": "JAVA_INT", "double": "JAVA_DOUBLE", "float": "JAVA_FLOAT"}[base]


def c_prototypes():
    protos = {}
    text = strip_comments(HEADER)
    for m in re.finditer(r"\b([A-Za-z_][\w \*]*?)\b(mh_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret:
            continue
        args = [] if args in ("", "void") else [a for a in args.split(",")]
        ret_kind = "VOID" if ret == "void" else c_kind(ret + " x")
        protos[name] = (ret_kind, [c_kind(a) for a in args])
    return protos


def java_descriptors():
    text = strip_comments(NATIVE)
    kinds = r"(?:ADDRESS|JAVA_LONG|JAVA_INT|JAVA_DOUBLE|JAVA_FLOAT)"

    def parse(expr):
        expr = " ".join(expr.split())
        m = re.fullmatch(r"status\((.*)\)", expr)
        if m:
            return "JAVA_INT", [a.strip() for a in m.group(1).split(",") if a.strip()]
        m = re.fullmatch(r"FunctionDescriptor\.ofVoid\((.*)\)", expr)
        if m:
            return "VOID", [a.strip() for a in m.group(1).split(",") if a.strip()]
        m = re.fullmatch(r"FunctionDescriptor\.of\((.*)\)", expr)
        if m:
            parts = [a.strip() for a in m.group(1).split(",") if a.strip()]
            return parts[0], parts[1:]
        raise AssertionError(f"descriptor expression not understood: {expr}")

    named = {}
    for m in re.finditer(r"FunctionDescriptor\s+(\w+)\s*=\s*(status\([^;]*\))\s*;", text, flags=re.S):
        named[m.group(1)] = parse(m.group(2))
    out = {}
    for m in re.finditer(r'handle\(\s*"(mh_\w+)"\s*,\s*((?:[^()]|\([^()]*\))*?)\)\s*;', text, flags=re.S):
        name, expr = m.group(1), m.group(2).strip()
        out[name] = named[expr] if expr in named else parse(expr)
        assert all(re.fullmatch(kinds, k) for k in out[name][1]), (name, out[name])
    return out


def test_every_downcall_descriptor_matches_its_c_prototype():
    protos, java = c_prototypes(), java_descriptors()
    assert len(java) >= 30, sorted(java)
    for name, (ret, args) in sorted(java.items()):
        assert name in protos, f"{name}: bound in MecanoHipNative.java but not declared in include/mecano_hip.h"
        c_ret, c_args = protos[name]
        assert ret == c_ret, f"{name}: returns {c_ret} in C, {ret} in the binding"
        assert args == c_args, f"{name}: C arguments {c_args}, binding {args}"


def c_struct(name):
    m = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), strip_comments(HEADER), flags=re.S)
    assert m, name
    fields = []
    for decl in [d.strip() for d in m.group(1).split(";") if d.strip()]:
        arr = re.search(r"(\w+)\[(\d+)\]$", decl)
        if arr:
            fields.append((arr.group(1), c_kind(decl.split("[")[0]), int(arr.group(2))))
        else:
            fields.append((re.search(r"(\w+)$", decl).group(1), c_kind(decl), 1))
    return fields


def java_struct(text, const):
    m = re.search(r"StructLayout\s+%s\s*=\s*MemoryLayout\.structLayout\((.*?)\)\s*;" % const, strip_comments(text), flags=re.S)
    assert m, const
    fields, pads = [], 0
    for item in re.finditer(r'MemoryLayout\.sequenceLayout\((\d+),\s*(\w+)\)\.withName\("(\w+)"\)|(\w+)\.withName\("(\w+)"\)|MemoryLayout\.paddingLayout\((\d+)\)',
                            m.group(1)):
        if item.group(3):
            fields.append((item.group(3), item.group(2), int(item.group(1))))
        elif item.group(5):
            fields.append((item.group(5), item.group(4), 1))
        else:
            fields.append(("<pad>", "PAD", int(item.group(6))))
    return fields


SIZE = {"JAVA_INT": 4, "JAVA_LONG": 8, "JAVA_DOUBLE": 8, "ADDRESS": 8, "JAVA_FLOAT": 4}


def with_natural_padding(fields):
    """What a C compiler lays out: each member aligned to its own size; explicit <pad> entries where padding is inserted."""
    out, ofs = [], 0
    for name, kind, count in fields:
        align = SIZE[kind]
        if ofs % align:
            out.append(("<pad>", "PAD", align - ofs % align))
            ofs += align - ofs % align
        out.append((name, kind, count))
        ofs += SIZE[kind] * count
    return out, ofs


@pytest.mark.parametrize("c_name, text, const", [("mh_options", NATIVE, "OPTIONS"), ("mh_model_desc", MODEL, "DESC")])
def test_struct_layouts_match_the_header(c_name, text, const):
    want, _ = with_natural_padding(c_struct(c_name))
    assert java_struct(text, const) == want


def test_options_helper_writes_at_the_struct_offsets():
    fields, size = with_natural_padding(c_struct("mh_options"))
    offsets, ofs = {}, 0
    for name, kind, count in fields:
        offsets[name] = ofs
        ofs += (count if kind == "PAD" else SIZE[kind] * count)
    body = re.search(r"static MemorySegment options\(Arena arena, boolean considerCoriolis, boolean considerAccelerations, double\[\] rootAcceleration, MemorySegment context\)\s*\{(.*?)\n   \}",
                     NATIVE, flags=re.S).group(1)
    sets = re.findall(r"options\.set\((\w+),\s*([^,]+),", body)
    assert (("JAVA_INT", str(offsets["consider_coriolis"])) in sets and ("JAVA_INT", str(offsets["consider_accelerations"])) in sets
            and ("JAVA_INT", str(offsets["layout"])) in sets and ("JAVA_INT", str(offsets["use_root_acceleration"])) in sets
            and ("ADDRESS", str(offsets["stream"])) in sets), sets
    assert ("JAVA_DOUBLE", f"{offsets['root_acceleration']} + 8L * k") in sets, sets
    assert ("ADDRESS", str(offsets["context"])) in sets, sets  # mh_options.context (MH_ABI_VERSION 4): the calls of a HipDeviceBatch name its context
    assert size == 80


def test_abi_version_constants_agree():
    version = int(re.search(r"#define MH_ABI_VERSION (\d+)", HEADER).group(1))
    assert int(re.search(r"static final int ABI = (\d+);", NATIVE).group(1)) == version
    assert f"ABI version {version}" in NATIVE


def test_ctypes_mirror_matches_the_header_structs():
    from mecano_amd import _lib
    kinds = {ctypes.c_int32: "JAVA_INT", ctypes.c_void_p: "ADDRESS", ctypes.c_double: "JAVA_DOUBLE", ctypes.c_int64: "JAVA_LONG"}
    for c_name, cls in (("mh_options", _lib.MhOptions), ("mh_model_desc", _lib.MhModelDesc)):
        want = c_struct(c_name)
        got = []
        for name, tp in cls._fields_:
            if hasattr(tp, "_length_"):
                got.append((name, kinds[tp._type_], tp._length_))
            else:
                got.append((name, kinds[tp], 1))
        assert got == want, c_name
        assert ctypes.sizeof(cls) == with_natural_padding(want)[1]
