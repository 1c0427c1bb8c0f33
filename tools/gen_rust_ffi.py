"""Prints Rust `extern "C"` declarations for the entry points of include/vdbmi.h that rust/vectordb-mi355x/src/ffi.rs
does not declare yet (a mechanical type mapping; the image has no rustc, so the output is checked by eye)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, "include/vdbmi.h")).read(), flags=re.S)
have = set(re.findall(r'fn (vdb_[a-z0-9_]+)', open(os.path.join(ROOT, "rust/vectordb-mi355x/src/ffi.rs")).read()))
SCALAR = {"size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "int64_t": "i64", "int32_t": "i32", "int": "c_int",
          "float": "f32", "double": "c_double", "char": "c_char", "void": "c_void", "vdb_status": "c_int", "vdb_metric": "c_int",
          "vdb_scalar_op": "c_int", "vdb_list_query_kind": "c_int", "vdb_value_kind": "c_int"}
def ty(c):
    c = c.strip()
    const = "const" in c.split()
    base = " ".join(t for t in c.replace("*", " ").split() if t != "const" and t != "struct")
    stars = c.count("*")
    r = SCALAR.get(base, base)                    # vdb_index, vdb_layer_csr ... keep their names
    if stars == 0:
        return r
    # innermost pointer carries the constness of the pointee; outer levels are written as *const for `T* const*`, else *mut
    out = r
    for level in range(stars):
        out = ("*const " if (const and level == 0) else "*mut ") + out
    if stars == 2 and re.search(r'\*\s*const\s*\*', c):
        out = "*const *const " + r
    return out
for ret, name, args in re.findall(r'\n([A-Za-z_][A-Za-z0-9_ \*]*?)\b(vdb_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', h):
    if name in have:
        continue
    params = []
    if args.strip() not in ("", "void"):
        for a in args.split(","):
            a = " ".join(a.split())
            m = re.match(r'(.*?)([A-Za-z_][A-Za-z0-9_]*)$', a)
            t, n = m.group(1), m.group(2)
            if n in ("type", "in", "ref", "fn", "loop", "match", "move", "self"):
                n += "_"
            params.append(f"{n}: {ty(t)}")
    r = ty(ret)
    print(f"    pub fn {name}({', '.join(params)})" + ("" if r == "c_void" else f" -> {r}") + ";")
