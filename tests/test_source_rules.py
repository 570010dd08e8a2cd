"""Static rules for the inline assembly in ctc_amd/csrc (no GPU needed).

hipcc treats an `asm` statement as one opaque instruction: it neither counts its memory operations nor pads its hazards
(/opt/skills/guides/cdna_hip_programming.md 5.7).  Two of those hazards bit this code base in round 4 ("The
hazard"); these checks keep them from coming back unnoticed."""
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ctc_amd", "csrc")


def _asm_statements():
    for name in sorted(os.listdir(CSRC)):
        if name.startswith("noblank_km"):                    # diagnostics-only experiment (generated loops, own rules)
            continue
        text = open(os.path.join(CSRC, name)).read()
        for m in re.finditer(r'asm\s+volatile\s*\(\s*((?:"(?:[^"\\]|\\.)*"\s*)+)', text):
            body = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', m.group(1)))
            yield name, text.count("\n", 0, m.start()) + 1, body


def test_wide_asm_stores_keep_their_wait_states():
    """a vector-memory store of more than 64 bits reads its data registers over several cycles: 2 wait states before a
    VALU write of them (gfx940+), inside the string"""
    seen = 0
    for name, line, body in _asm_statements():
        insns = [i.strip() for i in body.replace("\\n", "\n").replace("\\t", " ").split("\n") if i.strip()]
        for k, ins in enumerate(insns):
            if re.match(r"(global|buffer|flat|scratch)_store_dwordx[34]\b", ins):
                seen += 1
                assert k + 1 < len(insns) and re.match(r"s_nop\s+[1-9]", insns[k + 1]), \
                    "%s:%d: %s is not followed by s_nop 1 inside the asm string" % (name, line, ins.split()[0])
    assert seen >= 1                                          # (the write-through store of common.hpp)


def test_asm_vector_memory_with_scalar_base_is_settled():
    """an "s" operand of an asm vector-memory instruction may be a scalar the compiler has just reloaded from a spill lane
    (a VALU write of an SGPR: 5 wait states): every such statement's base goes through bin_row_base_settled first"""
    for name, line, body in _asm_statements():
        if re.search(r"(global|buffer)_(store|load)\w*\s", body) and re.search(r"%\d+\s+offset", body) and "s_nop 4" not in body:
            text = open(os.path.join(CSRC, name)).read()
            # the only such statements are bin_store_at's; their callers settle the row base once per row
            assert "bin_store_at" in text and "bin_row_base_settled" in text, "%s:%d" % (name, line)
    users = [n for n in os.listdir(CSRC) if "bin_store_col<" in open(os.path.join(CSRC, n)).read()]
    for n in users:
        text = open(os.path.join(CSRC, n)).read()
        rows = len(re.findall(r"float \*g = p\.grad \+ \(\(int64_t\)t[lt]\[i\]", text))
        assert rows >= 1 and text.count("bin_row_base_settled(g);") == rows, n
