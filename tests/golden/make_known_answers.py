#!/usr/bin/env python3
"""Writes tests/golden/reference_known_answers.json.

These are the reference's own known-answer tests for the hot path, restated as DATA
(inputs and expected outputs only).  The reference is Haskell and cannot be executed in
this image, so nothing here was produced by running it; each entry cites the reference
test or README line that states the expected value.  Paths are relative to /root/reference.
"""
import json
import os

S = "tests/Data/Text/AhoCorasickSpec.hs"
U = "tests/Data/Text/Utf8Spec.hs"
CS, IC = "CaseSensitive", "IgnoreCase"

iliad = "Ἄνδρα μοι ἔννεπε, Μοῦσα, πολύτροπον, ὃς μάλα πολλὰ"
iliad_upper = "ἌΝΔΡΑ ΜΟΙ ἜΝΝΕΠΕ, ΜΟΥ͂ΣΑ, ΠΟΛΎΤΡΟΠΟΝ, ὋΣ ΜΆΛΑ ΠΟΛΛᾺ"
tshirt = ["tshirt", "shirts", "shorts"]

data = {
    "utf8_encoding": [  # S:43-47
        {"src": S + ":43", "text": "$", "bytes": [0x24]},
        {"src": S + ":44", "text": "€", "bytes": [0xE2, 0x82, 0xAC]},
        {"src": S + ":45", "text": "£", "bytes": [0xC2, 0xA3]},
        {"src": S + ":46", "text": "𐍈", "bytes": [0xF0, 0x90, 0x8D, 0x88]},
        {"src": S + ":47", "text": "$€£𐍈", "bytes": [0x24, 0xE2, 0x82, 0xAC, 0xC2, 0xA3, 0xF0, 0x90, 0x8D, 0x88]},
    ],
    "count_matches": [
        {"src": S + ":53", "case": CS, "needles": ["abc", "rst", "xyz"], "haystack": "abcdefghijklmnopqrstuvwxyz", "count": 3},
        {"src": S + ":56", "case": CS, "needles": ["$", "£"], "haystack": "$€£𐍈", "count": 2},
        {"src": S + ":62", "case": IC, "needles": ["abc", "rst", "xyz"], "haystack": "abcdefghijklmnopqrstuvwxyz", "count": 3},
        {"src": S + ":65", "case": IC, "needles": ["ABC", "Rst", "xYZ"], "haystack": "abcdefghijklmnopqrstuvwxyz", "count": 0},
        {"src": S + ":68", "case": IC, "needles": ["groß", "öffnung", "tür"], "haystack": "Großfräsmaschinenöffnungstür", "count": 3},
        {"src": S + ":69", "case": IC, "needles": ["groß", "öffnung", "tür"], "haystack": "GROẞFRÄSMASCHINENÖFFNUNGSTÜR", "count": 3},
        {"src": S + ":255 (needles [] -> 0)", "case": CS, "needles": [], "haystack": "abc", "count": 0},
    ],
    "contains_any": [
        {"src": S + ":173", "case": CS, "needles": tshirt, "haystack": "short tshirts", "expected": True},
        {"src": S + ":174", "case": CS, "needles": tshirt, "haystack": "long shirt", "expected": False},
        {"src": S + ":175", "case": CS, "needles": tshirt, "haystack": "Short TSHIRTS", "expected": False},
        {"src": S + ":179", "case": IC, "needles": tshirt, "haystack": "Short TSHIRTS", "expected": True},
        {"src": S + ":183-187", "case": CS, "needles": ["μοι"], "haystack": iliad, "expected": True},
        {"src": S + ":183-187", "case": CS, "needles": ["Ὀδυσεύς"], "haystack": iliad, "expected": False},
        {"src": S + ":190-192", "case": IC, "needles": ["μοι"], "haystack": iliad_upper, "expected": True},
    ],
    # README.md:87-100 prints the fold result newest-first; stored here oldest-first
    # (= the order in which the fold function is called).  value = the needle text.
    "match_lists": [
        {"src": "README.md:90-93", "case": CS, "needles": tshirt, "haystack": "short tshirts",
         "matches": [[12, "tshirt"], [13, "shirts"]]},
        {"src": "README.md:95-101", "case": CS, "needles": tshirt, "haystack": "sweatshirts and shirtshirts",
         "matches": [[10, "tshirt"], [11, "shirts"], [22, "shirts"], [26, "tshirt"], [27, "shirts"]]},
    ],
    "replacer": [
        {"src": "README.md:67-68", "case": CS, "pairs": [["tshirt", "banana"], ["shirt", "pear"]], "haystack": "tshirts for sale", "expected": "bananas for sale"},
        {"src": "README.md:70-71", "case": CS, "pairs": [["tshirt", "banana"], ["shirt", "pear"]], "haystack": "tshirts and shirts for sale", "expected": "bananas and pears for sale"},
        {"src": "README.md:73-74", "case": CS, "pairs": [["tshirt", "banana"], ["shirt", "pear"]], "haystack": "sweatshirts and shirtshirts", "expected": "sweabananas and shirbananas"},
        {"src": "README.md:76-77", "case": CS, "pairs": [["tshirt", "banana"], ["shirt", "pear"]], "haystack": "sweatshirts and shirttshirts", "expected": "sweabananas and pearbananas"},
        {"src": S + ":89", "case": CS, "pairs": [["A", "B"]], "haystack": "AXAXB", "expected": "BXBXB"},
        {"src": S + ":90", "case": CS, "pairs": [["A", "B"], ["X", "Y"]], "haystack": "AXAXB", "expected": "BYBYB"},
        {"src": S + ":91", "case": CS, "pairs": [["aaa", ""], ["b", "c"]], "haystack": "aaabaaa", "expected": "c"},
        {"src": S + ":93", "case": CS, "pairs": [["A", "B"], ["Q", "r"], ["Z", ""]], "haystack": "AXAXB", "expected": "BXBXB"},
        {"src": S + ":96", "case": CS, "pairs": [["aa", "zz"], ["bb", "w"]], "haystack": "aaabbb", "expected": "zzawb"},
        {"src": S + ":97", "case": CS, "pairs": [["aaa", ""]], "haystack": "aaaaa", "expected": "aa"},
        {"src": S + ":100", "case": CS, "pairs": [["A", ""], ["BBBB", "bingo"]], "haystack": "BBABB", "expected": "bingo"},
        {"src": S + ":101", "case": CS, "pairs": [["BB", ""], ["BBBB", "bingo"]], "haystack": "BBBB", "expected": ""},
        {"src": S + ":104-105", "case": CS, "pairs": [["\U0001f574", "levitating man in business suit"]], "haystack": "the \U0001f574", "expected": "the levitating man in business suit"},
        {"src": S + ":109", "case": IC, "pairs": [["A", "B"]], "haystack": "AXAXB", "expected": "BXBXB"},
        {"src": S + ":110", "case": IC, "pairs": [["A", "B"]], "haystack": "axaxb", "expected": "BxBxb"},
        {"src": S + ":111", "case": IC, "pairs": [["a", "b"]], "haystack": "AXAXB", "expected": "bXbXB"},
        {"src": S + ":113", "case": IC, "pairs": [["A", "B"], ["X", "Y"]], "haystack": "AXAXB", "expected": "BYBYB"},
        {"src": S + ":114", "case": IC, "pairs": [["A", "B"], ["X", "Y"]], "haystack": "axaxb", "expected": "BYBYb"},
        {"src": S + ":115", "case": IC, "pairs": [["a", "b"], ["x", "y"]], "haystack": "AXAXB", "expected": "bybyB"},
        {"src": S + ":118", "case": IC, "pairs": [["foo", "BAR"], ["bar", "BAZ"]], "haystack": "Foo", "expected": "BAZ"},
        {"src": S + ":121", "case": IC, "pairs": [["éclair", "lightning"]], "haystack": "Éclair", "expected": "lightning"},
        {"src": S + ":124", "case": IC, "pairs": [["å", "b"]], "haystack": "åÅÅ", "expected": "bbb"},
        {"src": S + ":125", "case": IC, "pairs": [["k", "m"]], "haystack": "KkK", "expected": "mmm"},
        {"src": S + ":126", "case": IC, "pairs": [["ǳ", "z"]], "haystack": "ǳǲǱ", "expected": "zzz"},
        {"src": S + ":127", "case": IC, "pairs": [["bèta", "α"], ["Α", "alpha"]], "haystack": "BÈTA", "expected": "alpha"},
        {"src": S + ":128", "case": IC, "pairs": [["ßèta", "sseta"]], "haystack": "ßèta", "expected": "sseta"},
        {"src": S + ":129", "case": IC, "pairs": [["ßèta", "sseta"]], "haystack": "ẞÈTA", "expected": "sseta"},
        {"src": S + ":134-135", "case": IC, "pairs": [["\U0001f574", "levitating man in business suit"]], "haystack": "the \U0001f574", "expected": "the levitating man in business suit"},
    ],
    # Searcher.containsAll with the empty needle is never true (S:196-200): pins the quirk that
    # root values are only emitted after a successful goto transition.
    "contains_all_empty_needle": [
        {"src": S + ":196-200", "case": CS, "needles": [""], "haystack": h, "expected": False}
        for h in ["", "a", "abc", "éè", "💩"]
    ],
    "skip_code_points_backwards": (  # U:115-154
        [{"src": U + ":117-120", "text": "abcd", "index": 3, "n": n, "expected": 3 - n} for n in range(4)]
        + [{"src": U + ":123-130", "text": "💩💩", "index": i, "n": 0, "expected": 0 if i < 4 else 4} for i in range(8)]
        + [{"src": U + ":133-136", "text": "💩💩", "index": i, "n": 1, "expected": 0} for i in range(4, 8)]
        + [{"src": U + ":140-150", "text": "aİẞ💩ẞİa", "index": i, "n": n, "expected": e} for (i, n, e) in
           [(15, 0, 15), (15, 1, 13), (15, 2, 10), (15, 3, 6), (15, 4, 3), (15, 5, 1), (15, 6, 0),
            (14, 2, 6), (13, 2, 6), (10, 3, 1), (9, 3, 0)]]
        + [{"src": U + ":153", "text": "💩💩", "index": 8, "n": 0, "expected": "error"},
           {"src": U + ":154", "text": "💩💩", "index": 7, "n": 2, "expected": "error"}]
    ),
    # src/Data/Text/Utf8/Unlower.hs:45-55 (doc table) and U:52-62: inverse of Char.toLower.
    # Sets, not lists: the reference's list order is a HashMap artefact.
    "unlower": [
        {"src": U + ":53", "cp": "A", "set": []},
        {"src": U + ":54", "cp": "ẞ", "set": []},
        {"src": U + ":57", "cp": "1", "set": ["1"]},
        {"src": U + ":60", "cp": "a", "set": ["a", "A"]},
        {"src": U + ":61", "cp": "ß", "set": ["ẞ", "ß"]},
        {"src": U + ":62", "cp": "i", "set": ["İ", "i", "I"]},
        {"src": "src/Data/Text/Utf8/Unlower.hs:46", "cp": "k", "set": ["K", "k", "K"]},
        {"src": "src/Data/Text/Utf8/Unlower.hs:48", "cp": "å", "set": ["Å", "å", "Å"]},
        {"src": "src/Data/Text/Utf8/Unlower.hs:49", "cp": "ǆ", "set": ["ǆ", "ǅ", "Ǆ"]},
        {"src": "src/Data/Text/Utf8/Unlower.hs:50", "cp": "ǉ", "set": ["ǉ", "ǈ", "Ǉ"]},
        {"src": "src/Data/Text/Utf8/Unlower.hs:51", "cp": "ǌ", "set": ["ǌ", "ǋ", "Ǌ"]},
        {"src": "src/Data/Text/Utf8/Unlower.hs:52", "cp": "ǳ", "set": ["ǳ", "ǲ", "Ǳ"]},
        {"src": "src/Data/Text/Utf8/Unlower.hs:53", "cp": "θ", "set": ["ϴ", "θ", "Θ"]},
        {"src": "src/Data/Text/Utf8/Unlower.hs:54", "cp": "ω", "set": ["Ω", "ω", "Ω"]},
    ],
    # Splitter (SURVEY 8f "next" row), S:224-244
    "splitter": [
        {"src": S + ":228", "sep": "bob", "ignore_case": False, "haystack": "C++bobobCOBOLbobScala", "expected": ["C++", "obCOBOL", "Scala"]},
        {"src": S + ":229", "sep": "bob", "ignore_case": True, "haystack": "C++bobobCOBOLbobScala", "expected": ["C++", "obCOBOL", "Scala"]},
        {"src": S + ":230", "sep": "bob", "ignore_case": True, "haystack": "C++BOBOBCOBOLBOBSCALA", "expected": ["C++", "OBCOBOL", "SCALA"]},
        {"src": S + ":235", "sep": ", ", "ignore_case": False, "haystack": iliad, "expected": ["Ἄνδρα μοι ἔννεπε", "Μοῦσα", "πολύτροπον", "ὃς μάλα πολλὰ"]},
        {"src": S + ":237", "sep": ", ", "ignore_case": True, "haystack": iliad, "expected": ["Ἄνδρα μοι ἔννεπε", "Μοῦσα", "πολύτροπον", "ὃς μάλα πολλὰ"]},
        {"src": S + ":243", "sep": "å", "ignore_case": True, "haystack": "aaåbbÅccÅdd", "expected": ["aa", "bb", "cc", "dd"]},
    ],
    # benchmark/data-utf8/example.txt (the only checked-in data file; format: needles, blank
    # line, haystack -- benchmark/README.md:20-32).  The reference states no expected count for
    # it; the count below is from the independent naive oracle (oracle/naive.py).
    "benchmark_example_file": {
        "src": "benchmark/data-utf8/example.txt",
        "needles": ["Henk", "Piet", "Klaas", "Sjaak", "Marieke"],
        "haystack": "Henk eet een appel en Piet eet kaas.\nKlaas eet ook kaas.\nKaas is baas.\n"
                    "Mari en Marieke wandelen door het bos.\nDe auto van Sjaak heeft geen trekhaak en die van Klaas ook niet.\n",
        "count_naive": 6,
    },
}

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "reference_known_answers.json"), "w") as f:
    json.dump(data, f, indent=1, ensure_ascii=True)
print({k: (len(v) if isinstance(v, list) else 1) for k, v in data.items()})
