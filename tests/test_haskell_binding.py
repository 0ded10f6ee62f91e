"""The Haskell side of the boundary as files (haskell/: VERDICT r4 item 7).  The image has no GHC, so the modules cannot be compiled here; what CAN be
checked is the contract: every `foreign import ccall` names a function that include/am.h declares, with the same arity and the same C types
(Ptr a -> pointer, CInt -> int, CSize -> size_t, Word64 -> uint64_t, ...), the storable records have the sizes of their C structs, and the package
description names the library and the header.  Reference for what the modules stand in for: src/Data/Text/AhoCorasick/Automaton.hs:32-44,
Searcher.hs:14-24, Replacer.hs:20-28 (export lists), benchmark/rust-ffi/app/Main.hs:28-52 (the reference's own FFI shape)."""
import os
import re

from tests.conftest import ROOT

HS = os.path.join(ROOT, "haskell")

# Haskell FFI type -> C type class
HS_TO_C = {"CInt": "int", "CSize": "size_t", "CLong": "long", "Word8": "uint8_t", "Word32": "uint32_t", "Word64": "uint64_t", "Int64": "int64_t", "()": "void"}


def _c_prototypes():
    src = open(os.path.join(ROOT, "include", "am.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"AM_API\s+([^;{}]+?)\b(am_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
        protos[name] = (_c_class(ret), [_c_class(p) for p in params])
    return protos


def _c_class(decl):
    """`const am_slice* hay` -> 'ptr'; `uint64_t max_length` -> 'uint64_t'; `int` -> 'int'; `uint8_t id_out[128]` -> 'ptr'."""
    decl = decl.strip()
    if "*" in decl or "[" in decl:
        return "ptr"
    toks = [t for t in decl.replace("const", " ").split() if t]
    for t in toks:
        if t in ("int", "void", "size_t", "long", "uint8_t", "uint32_t", "uint64_t", "int64_t"):
            return t
    raise AssertionError("unclassified C declaration: %r" % decl)


def _hs_imports():
    out = []
    for base, _, files in os.walk(HS):
        for f in files:
            if not f.endswith(".hs"):
                continue
            text = open(os.path.join(base, f)).read()
            text = re.sub(r"--[^\n]*", "", text)
            for m in re.finditer(r'foreign\s+import\s+ccall\s+(safe|unsafe)\s+"(&?)(\w+)"\s+(\w+)\s*::\s*(.*?)(?=\nforeign\s|\n\n|\n[a-z]\w*\s*::|\ndata\s|\ninstance\s)', text, flags=re.S):
                safety, amp, cname, hsname, sig = m.groups()
                out.append((f, safety, bool(amp), cname, hsname, " ".join(sig.split())))
    return out


def _split_arrows(sig):
    parts, depth, cur = [], 0, ""
    i = 0
    while i < len(sig):
        ch = sig[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if depth == 0 and sig.startswith("->", i):
            parts.append(cur.strip()); cur = ""; i += 2
            continue
        cur += ch
        i += 1
    parts.append(cur.strip())
    return parts


def _hs_class(t):
    t = t.strip()
    while t.startswith("(") and t.endswith(")") and t != "()":
        t = t[1:-1].strip()
    if t.startswith("Ptr ") or t.startswith("FunPtr "):
        return "ptr"
    if t.startswith("IO "):
        return _hs_class(t[3:])
    return HS_TO_C[t]


def test_every_foreign_import_matches_the_header():
    protos = _c_prototypes()
    imports = _hs_imports()
    assert len(imports) >= 17, imports
    seen = set()
    for f, safety, amp, cname, hsname, sig in imports:
        assert cname in protos, "%s: %s is not declared in include/am.h" % (f, cname)
        ret, params = protos[cname]
        seen.add(cname)
        if amp:                                            # "&am_x_destroy" :: FunPtr (Ptr X -> IO ()): a finaliser = one pointer argument, void
            inner = re.match(r"FunPtr\s*\((.*)\)$", sig).group(1)
            parts = _split_arrows(inner)
            assert [_hs_class(p) for p in parts[:-1]] == params == ["ptr"] and _hs_class(parts[-1]) == ret == "void", (cname, sig)
            continue
        parts = _split_arrows(sig)
        got_params, got_ret = [_hs_class(p) for p in parts[:-1]], _hs_class(parts[-1])
        assert got_params == params, "%s %s: Haskell %s vs C %s" % (f, cname, got_params, params)
        assert got_ret == ret, "%s %s: returns %s vs C %s" % (f, cname, got_ret, ret)
        assert parts[-1].startswith("IO "), (cname, "every import is in IO")
        # calls that launch GPU work and wait for it must not block the Haskell runtime's capability: `safe`
        if cname in ("am_run", "am_count", "am_contains_any", "am_contains_all", "am_replacer_run", "am_matches_data"):
            assert safety == "safe", cname
    # the entry points a drop-in for the three reference modules needs
    need = {"am_automaton_create_ex", "am_automaton_destroy", "am_run", "am_count", "am_contains_any", "am_matches_size", "am_matches_data", "am_matches_free",
            "am_last_error", "am_needle_ids_create", "am_needle_ids_destroy", "am_contains_all", "am_replacer_create", "am_replacer_destroy", "am_replacer_run",
            "am_replaced_get", "am_replaced_free"}
    assert need <= seen, need - seen


def test_storable_records_have_the_c_layouts():
    src = open(os.path.join(HS, "src", "Data", "Text", "AhoCorasick", "Automaton", "Device.hs")).read()
    rep = open(os.path.join(HS, "src", "Data", "Text", "AhoCorasick", "Replacer", "Device.hs")).read()
    # am_slice {ptr, size_t, size_t} = 24, am_match {u64, u32, u32} = 16, am_payload {i64, u32, u32, u64, u32, u32} = 32 (include/am.h)
    for text, name, size, offsets in ((src, "AmSlice", 24, (0, 8, 16)), (src, "AmMatch", 16, (0, 8, 12)), (rep, "AmPayload", 32, (0, 8, 12, 16, 24))):
        inst = text[text.index("instance Storable %s" % name):]
        inst = inst[:inst.index("\n\n")]
        assert re.search(r"sizeOf _ = %d\b" % size, inst), (name, inst)
        assert tuple(int(x) for x in re.findall(r"peekByteOff p (\d+)", inst)) == offsets, (name, inst)


def test_package_description_links_the_library():
    cabal = open(os.path.join(HS, "alfred-margaret-device.cabal")).read()
    assert re.search(r"extra-libraries:\s*am\b", cabal) and re.search(r"include-dirs:\s*\.\./include", cabal)
    mods = re.findall(r"^\s+(Data\.Text\.AhoCorasick\.\S+)$", cabal, flags=re.M)
    assert len(mods) == 3
    for m in mods:
        path = os.path.join(HS, "src", *m.split(".")) + ".hs"
        assert os.path.exists(path), path
        assert re.search(r"^module %s\b" % re.escape(m), open(path).read(), flags=re.M)
