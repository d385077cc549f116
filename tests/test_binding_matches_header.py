"""The Rust binding INTEGRATION.md shows (and the ctypes binding vgen_amd/api.py is) must describe the header as the C compiler sees it.

No rustc in the image, so the document's `#[repr(C)]` structs and `extern "C"` declarations are PARSED here and held against
include/vgen_hip.h three ways:
  * every struct: field names, order and types against the header's own declaration, and the repr(C) layout those Rust types imply
    (offset of every field, size of the struct) against `offsetof` / `sizeof` printed by a C program compiled against the header;
  * every function: argument count, each argument's type (pointer-ness, const-ness, width, signedness) and the return type, through
    a C-type -> Rust-type mapping;
  * the ctypes Structures of vgen_amd/api.py against the same offsets.
A swapped pair of fields, a widened integer, a dropped argument or a `*const` where the library writes fails here, on the CPU."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vgen_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")

# Rust struct <-> C struct names
STRUCTS = {"VgenParams": "vgen_params", "VgenMatch": "vgen_match", "VgenMemoryInfo": "vgen_memory_info", "VgenScanConfig": "vgen_scan_config",
           "VgenGenerated": "vgen_generated", "VgenScanResult": "vgen_scan_result"}
OPAQUE = {"vgen_ctx": "VgenCtx", "vgen_filter": "VgenFilter"}
PRIM = {"uint8_t": "u8", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int": "c_int", "char": "c_char", "float": "f32", "double": "f64",
        "size_t": "usize", "void": "c_void"}
RUST_SIZE = {"u8": 1, "i8": 1, "c_char": 1, "u32": 4, "i32": 4, "c_int": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "usize": 8}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_type_to_rust(ctype, array=None):
    """'const uint8_t *' -> '*const u8'; arrays in ARGUMENT position decay to pointers (array given as the [n] text)."""
    t = ctype.replace("volatile", " ").strip()
    stars = t.count("*")
    t = t.replace("*", " ")
    words = t.split()
    const = "const" in words
    words = [w for w in words if w not in ("const", "struct", "enum")]
    assert len(words) == 1, (ctype, words)
    base = words[0]
    if base in PRIM:
        r = PRIM[base]
    elif base in OPAQUE:
        r = OPAQUE[base]
    elif base in {v: k for k, v in STRUCTS.items()}:
        r = {v: k for k, v in STRUCTS.items()}[base]
    elif base == "vgen_progress_cb":
        return 'Option<extern "C" fn(u64, *mut c_void)>'
    else:
        raise AssertionError("unknown C type " + ctype)
    if array is not None:   # argument arrays decay: `const uint8_t key[32]` is `const uint8_t *`
        stars += 1
    # const applies to the pointee of the innermost pointer (the header never writes `T *const`)
    for i in range(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def parse_header():
    text = strip_c_comments(open(HEADER).read())
    text = re.sub(r"#.*", "", text)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.*?)(\w+)\s*(\[(\d+)\])?$", decl)
            assert fm, decl
            ctype, fname, arr = fm.group(1).strip(), fm.group(2), fm.group(4)
            rust = c_type_to_rust(ctype)
            if arr:
                rust = "[%s; %s]" % (rust, arr)
            fields.append((fname, rust))
        structs[name] = fields
    text_nostruct = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    text_nostruct = re.sub(r"typedef\s+enum\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text_nostruct, flags=re.S)
    funcs = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(vgen_\w+)\s*\(([^;{}]*?)\)\s*;", text_nostruct, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret:
            continue
        rret = None if ret == "void" else c_type_to_rust(ret)
        rargs = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.*?)(\w+)\s*(\[\d*\])?$", a)
                assert am, a
                rargs.append(c_type_to_rust(am.group(1).strip(), am.group(3)))
        funcs[name] = (rargs, rret)
    return structs, funcs


def rust_blocks():
    doc = open(DOC).read()
    return "\n".join(re.findall(r"```rust\n(.*?)```", doc, flags=re.S))


def strip_rust_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def split_top(s, sep=","):
    """Split at separators that are not inside (), [] or <>."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def parse_rust():
    text = strip_rust_comments(rust_blocks())
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub\s+struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in split_top(" ".join(m.group(2).split())):
            fm = re.match(r"pub\s+(\w+)\s*:\s*(.+)$", f)
            assert fm, f
            fields.append((fm.group(1), " ".join(fm.group(2).split())))
        structs[m.group(1)] = fields
    funcs = {}
    for m in re.finditer(r"pub\s+fn\s+(vgen_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", text, flags=re.S):
        args = []
        for a in split_top(" ".join(m.group(2).split())):
            am = re.match(r"\w+\s*:\s*(.+)$", a)
            assert am, a
            args.append(" ".join(am.group(1).split()))
        ret = " ".join(m.group(3).split()) if m.group(3) else None
        funcs[m.group(1)] = (args, ret)
    return structs, funcs


def rust_layout(fields):
    """repr(C) layout of a field list: [(name, offset, size)], total size."""
    def size_align(t):
        am = re.match(r"\[(.+);\s*(\d+)\]$", t)
        if am:
            s, a = size_align(am.group(1))
            return s * int(am.group(2)), a
        if t.startswith("*") or t.startswith("Option<"):
            return 8, 8
        return RUST_SIZE[t], RUST_SIZE[t]
    off, maxa, out = 0, 1, []
    for name, t in fields:
        s, a = size_align(t)
        off = (off + a - 1) // a * a
        out.append((name, off, s))
        off += s
        maxa = max(maxa, a)
    return out, (off + maxa - 1) // maxa * maxa


@pytest.fixture(scope="module")
def c_layout(tmp_path_factory):
    """offsetof / sizeof of every field of every struct of the header, from a C program compiled against it."""
    hstructs, _ = parse_header()
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vgen_hip.h"', 'int main(void) {']
    for sname, fields in hstructs.items():
        lines.append('  printf("%s * %%zu 0\\n", sizeof(%s));' % (sname, sname))
        for fname, _ in fields:
            lines.append('  printf("%s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));' % (sname, fname, sname, fname, sname, fname))
    lines += ['  return 0;', '}']
    d = tmp_path_factory.mktemp("layout")
    src = d / "layout.c"
    src.write_text("\n".join(lines))
    exe = d / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        s, f, a, b = ln.split()
        out.setdefault(s, {})[f] = (int(a), int(b))
    return out


def test_every_struct_of_the_rust_binding_is_the_headers(c_layout):
    hstructs, _ = parse_header()
    rstructs, _ = parse_rust()
    assert set(rstructs) == set(STRUCTS), (sorted(rstructs), sorted(STRUCTS))
    assert set(STRUCTS.values()) == set(hstructs), (sorted(hstructs),)
    for rname, cname in STRUCTS.items():
        assert rstructs[rname] == hstructs[cname], "%s: the document declares\n  %s\nthe header\n  %s" % (rname, rstructs[rname], hstructs[cname])
        layout, total = rust_layout(rstructs[rname])
        assert total == c_layout[cname]["*"][0], (rname, total, c_layout[cname]["*"])
        for fname, off, size in layout:
            assert (off, size) == c_layout[cname][fname], (rname, fname, (off, size), c_layout[cname][fname])


def test_every_function_of_the_rust_binding_has_the_headers_signature():
    _, hfuncs = parse_header()
    _, rfuncs = parse_rust()
    assert set(rfuncs) == set(hfuncs), (sorted(set(hfuncs) - set(rfuncs)), sorted(set(rfuncs) - set(hfuncs)))
    assert len(hfuncs) >= 44
    for name, (hargs, hret) in hfuncs.items():
        rargs, rret = rfuncs[name]
        assert rret == hret, (name, "returns", rret, "header", hret)
        assert len(rargs) == len(hargs), (name, rargs, hargs)
        for i, (ra, ha) in enumerate(zip(rargs, hargs)):
            assert ra == ha, "%s: argument %d is %s in INTEGRATION.md, %s by the header" % (name, i, ra, ha)


def test_the_parser_sees_a_swapped_field_a_widened_integer_and_a_lost_argument(c_layout, monkeypatch, tmp_path):
    """The test above must be able to fail: three mutations of the document, each caught."""
    doc = open(DOC).read()
    import tests.test_binding_matches_header as me

    def with_doc(text, fn):
        p = tmp_path / "doc.md"
        p.write_text(text)
        monkeypatch.setattr(me, "DOC", str(p))
        try:
            fn()
        except AssertionError:
            return True
        finally:
            monkeypatch.setattr(me, "DOC", DOC)
        return False
    swapped = doc.replace("pub shard: u32, pub n_shards: u32", "pub n_shards: u32, pub shard: u32")
    widened = doc.replace("pub frames: u32, pub match_cap: u32", "pub frames: u64, pub match_cap: u32")
    lost = doc.replace("pub fn vgen_set_match_cap(ctx: *mut VgenCtx, match_cap: u32) -> c_int;", "pub fn vgen_set_match_cap(ctx: *mut VgenCtx) -> c_int;")
    const = doc.replace("pub fn vgen_key_add(key_be: *const u8, amount: u64, out_be: *mut u8)", "pub fn vgen_key_add(key_be: *const u8, amount: u64, out_be: *const u8)")
    assert swapped != doc and widened != doc and lost != doc and const != doc
    assert with_doc(swapped, lambda: me.test_every_struct_of_the_rust_binding_is_the_headers(c_layout))
    assert with_doc(widened, lambda: me.test_every_struct_of_the_rust_binding_is_the_headers(c_layout))
    assert with_doc(lost, me.test_every_function_of_the_rust_binding_has_the_headers_signature)
    assert with_doc(const, me.test_every_function_of_the_rust_binding_has_the_headers_signature)
    assert not with_doc(doc, lambda: me.test_every_struct_of_the_rust_binding_is_the_headers(c_layout))


def test_the_ctypes_structures_have_the_headers_layout(c_layout):
    sys.path.insert(0, ROOT)
    from vgen_amd import api
    pairs = {"vgen_params": api._Params, "vgen_match": api._Match, "vgen_memory_info": api._MemoryInfo, "vgen_scan_config": api._ScanConfig,
             "vgen_generated": api._Generated, "vgen_scan_result": api._ScanResult}
    assert set(pairs) == set(c_layout)
    for cname, st in pairs.items():
        assert ctypes.sizeof(st) == c_layout[cname]["*"][0], (cname, ctypes.sizeof(st), c_layout[cname]["*"])
        names = [f[0] for f in st._fields_]
        assert names == [f for f in c_layout[cname] if f != "*"], (cname, names)
        for fname in names:
            d = getattr(st, fname)
            assert (d.offset, d.size) == c_layout[cname][fname], (cname, fname, (d.offset, d.size), c_layout[cname][fname])
