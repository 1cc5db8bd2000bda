"""The formulas of the path that are PURE TEXT - Mat4.inverse's twenty cofactor expressions and Mat4.fromQuat's nine
elements - compared token for token between the reference's engine/math3d.zig and the two restatements that must equal
it: the oracle (oracle/yume_oracle.c) and the kernels' (yume_amd/csrc/yh_math.hpp).  Oracle and kernels were written by
the same hand, so a transcription slip shared by both would pass every parity test; this test reads the reference's own
text instead.  It does not lift "parity unpinned" (only a Zig compiler could), it removes the blind spot for what can be
checked without one.  This container only: skips where /root/reference is absent (the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZIG = "/root/reference/engine/math3d.zig"

pytestmark = pytest.mark.skipif(not os.path.exists(ZIG), reason="no reference tree on this box")


def canon(expr):
    """One expression as a token string: no blanks, no C float suffix, the two spellings of 1/det unified."""
    e = re.sub(r"\s+", "", expr)
    e = re.sub(r"(?<![A-Za-z_])(\d+\.\d+|\d+)f\b", r"\1", e)  # 1.0f -> 1.0
    e = re.sub(r"\bdet_inv\b", "di", e)
    e = re.sub(r"(?<![\w.])(\d+)(?![\w.])", r"\1.0", e)  # Zig's comptime ints in float context: 1 / det -> 1.0 / det
    return e.rstrip(",;")


def function_body(text, start_pat, end_pat):
    a = re.search(start_pat, text)
    assert a, start_pat
    b = re.search(end_pat, text[a.end():])
    assert b, end_pat
    return text[a.end(): a.end() + b.start()]


def inverse_from(text, start_pat, end_pat, elem_pat, out_pat):
    """-> (element names by index, [t11..t14, det], di, 16 output expressions) of one inverse implementation."""
    body = function_body(text, start_pat, end_pat)
    names = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(n[1-4][1-4])\s*=\s*" + elem_pat, body)}
    defs = {}
    for m in re.finditer(r"\b(t1[1-4]|det|det_inv|di)\s*=(?!=)\s*([^;]+);", body):
        defs.setdefault(m.group(1), canon(m.group(2)))
    outs = [canon(m.group(1)) for m in re.finditer(out_pat, body)]
    return names, [defs[k] for k in ("t11", "t12", "t13", "t14", "det")], defs.get("det_inv", defs.get("di")), outs


def test_inverse_is_the_references_text():
    zig = open(ZIG).read()
    zig_body_outs = r"^\s+((?:t1[1-4]|\(n[^\n]+\))\s*\*\s*det_inv),\s*$"
    ref = inverse_from(zig, r"pub fn inverse\(self: Mat4\)[^{]*\{", r"\n    \}\n", r"self\.values\[(\d+)\]", "(?m)" + zig_body_outs)
    assert len(ref[0]) == 16 and len(ref[3]) == 16 and sorted(ref[0].values()) == list(range(16))
    assert ref[2] == "1.0/det"
    oracle = open(os.path.join(ROOT, "oracle", "yume_oracle.c")).read()
    got_o = inverse_from(oracle, r"int yo_inverse\([^)]*\)\s*\{", r"\n\}\n", r"m->v\[(\d+)\]", r"r\.v\[\d+\]\s*=\s*([^;]+);")
    kern = open(os.path.join(ROOT, "yume_amd", "csrc", "yh_math.hpp")).read()
    got_k = inverse_from(kern, r"bool m4_inverse\([^)]*\)\s*\{", r"\n\}\n", r"m\.v\[(\d+)\]", r"out\.v\[\d+\]\s*=\s*([^;]+);")
    for who, got in (("oracle/yume_oracle.c yo_inverse", got_o), ("yh_math.hpp m4_inverse", got_k)):
        assert got[0] == ref[0], f"{who}: n_rc names map to other elements than math3d.zig:765-780"
        assert got[1] == ref[1], f"{who}: t11..t14 / det differ from math3d.zig:782-787"
        assert got[2] == ref[2], f"{who}: det_inv differs from math3d.zig:791"
        assert len(got[3]) == 16
        for i, (a, b) in enumerate(zip(got[3], ref[3])):
            assert a == b, f"{who}: output element {i} differs from math3d.zig:794-812:\n  {a}\n  {b}"
    # the stores go to elements 0..15 in the reference's order
    for text, pat in ((oracle, r"r\.v\[(\d+)\]\s*=\s*[^;]+\* di;"), (kern, r"out\.v\[(\d+)\]\s*=\s*[^;]+\* di;")):
        assert [int(x) for x in re.findall(pat, text)][:16] == list(range(16))
    # the zero test is exact equality in all three (math3d.zig:789)
    assert "if (det == 0) return error.DeterminantZero;" in zig
    assert "if (det == 0.0f) return 0;" in oracle and "if (det == 0.0f) return false;" in kern


def from_quat_from(text, start_pat, end_pat, out_pat, zig=False):
    body = function_body(text, start_pat, end_pat)
    prods = {m.group(1): canon(m.group(2)) for m in re.finditer(r"\b(xx|yy|zz|xy|xz|yz|wx|wy|wz)\s*=\s*([wxyz]\s*\*\s*[wxyz])", body)}
    if zig:  # three Vec4.make(a, b, c, 0.0) rows = columns i, j, k: elements 0,1,2 / 4,5,6 / 8,9,10
        rows = re.findall(r"Vec4\.make\((.*), 0\.0\),", body)
        assert len(rows) == 3
        elems = {}
        for c, row in enumerate(rows):
            depth, cur, parts = 0, "", []
            for ch in row:  # split on top-level commas
                if ch == "(":
                    depth += 1
                if ch == ")":
                    depth -= 1
                if ch == "," and depth == 0:
                    parts.append(cur)
                    cur = ""
                else:
                    cur += ch
            parts.append(cur)
            assert len(parts) == 3
            for r, e in enumerate(parts):
                elems[c * 4 + r] = canon(e)
        return prods, elems
    return prods, {int(m.group(1)): canon(m.group(2)) for m in re.finditer(out_pat, body) if int(m.group(1)) in (0, 1, 2, 4, 5, 6, 8, 9, 10)}


def test_from_quat_is_the_references_text():
    zig = open(ZIG).read()
    ref = from_quat_from(zig, r"pub fn fromQuat\(quat: Quat\) Mat4 \{", r"\n    \}\n", None, zig=True)
    assert len(ref[0]) == 9 and len(ref[1]) == 9
    oracle = open(os.path.join(ROOT, "oracle", "yume_oracle.c")).read()
    got_o = from_quat_from(oracle, r"void yo_from_quat\([^)]*\)\s*\{", r"\n\}\n", r"out->v\[(\d+)\]\s*=\s*([^;]+);")
    kern = open(os.path.join(ROOT, "yume_amd", "csrc", "yh_math.hpp")).read()
    got_k = from_quat_from(kern, r"M4 m4_from_quat\([^)]*\)\s*\{", r"\n\}\n", r"r\.v\[(\d+)\]\s*=\s*([^;]+);")
    for who, got in (("oracle/yume_oracle.c yo_from_quat", got_o), ("yh_math.hpp m4_from_quat", got_k)):
        assert got[0] == ref[0], f"{who}: the nine products differ from math3d.zig:733-741"
        assert got[1] == ref[1], f"{who}: elements differ from math3d.zig:744-746: {got[1]} vs {ref[1]}"


def test_decompose_spells_the_epsilon_tests_as_the_reference_does():
    """math3d.zig:854-856 uses epsilonNotEql (false for NaN), not !epsilonEql (true for NaN): both restatements must."""
    zig = open(ZIG).read()
    assert re.search(r"if \(epsilonNotEql\(m\.unnamed\[0\]\[3\], 0\) or\s+epsilonNotEql\(m\.unnamed\[1\]\[3\], 0\) or\s+epsilonNotEql\(m\.unnamed\[2\]\[3\], 0\)\)", zig)
    assert "return @abs(a - b) > epsilon;" in zig and "return @abs(a - b) <= epsilon;" in zig
    oracle = open(os.path.join(ROOT, "oracle", "yume_oracle.c")).read()
    kern = open(os.path.join(ROOT, "yume_amd", "csrc", "yh_math.hpp")).read()
    assert "if (eps_not_eql(m[3], 0.0f) || eps_not_eql(m[7], 0.0f) || eps_not_eql(m[11], 0.0f))" in oracle
    assert "return fabsf(a - b) > YO_EPS;" in oracle and "return fabsf(a - b) <= YO_EPS;" in oracle
    assert "if (not_eql0(m[3]) || not_eql0(m[7]) || not_eql0(m[11]))" in kern
    assert "__builtin_fabsf(a - 0.0f) > eps" in kern and "__builtin_fabsf(a - 0.0f) <= eps" in kern
    assert "!eps_eql(" not in oracle and "!eql0(" not in kern
