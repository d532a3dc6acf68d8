"""The hand-placed operand registers of the three-product chunk (tomahawk_amd/csrc/hip/ld_count.hip.h, contract3_chunk) - checked as text, without a GPU:
what the GPU tests cannot see is whether an edit of the tables still keeps every product group's four words (hA qA hB qB) in four VGPR banks (a
v_bitop3_b32 with two sources in one bank costs ~3 % of the headline), and a wrong table would only show up there as a slower kernel."""
import os
import re

HDR = os.path.join(os.path.dirname(__file__), "..", "tomahawk_amd", "csrc", "hip", "ld_count.hip.h")


def _macros():
    text = open(HDR).read().replace("\\\n", " ")
    return {m.group(1): m.group(2) for m in re.finditer(r"^#define (TWK_\w+)(?:\([^)]*\))? (.*)$", text, re.M)}


def _g12(body):
    return [tuple(int(x) for x in m.group(1).split(",")) for m in re.finditer(r"TWK_G12\(([^)]*)\)", body)]


def test_every_product_group_reads_four_banks():
    mac = _macros()
    halves = {k: _g12(v) for k, v in mac.items() if k.startswith("TWK_HALF")}
    assert sorted(halves) == ["TWK_HALF0_P", "TWK_HALF0_R", "TWK_HALF1_P", "TWK_HALF1_R"]
    for name, groups in halves.items():
        assert len(groups) == 4, name                      # two words x two B variants
        for g in groups:
            h, q, hb, qb, v = g[0:4], g[4:8], g[8], g[9], g[10]
            assert v in (0, 1)
            for s in range(4):
                banks = {h[s] % 4, q[s] % 4, hb % 4, qb % 4}
                assert len(banks) == 4, (name, g)          # v_and hA hB / v_bitop3 qA hB qB / v_bitop3 qB hA qA: no bank twice
            assert len({x - h[0] for x in h} ^ {0, 4, 8, 12}) == 0 and len({x - q[0] for x in q} ^ {0, 4, 8, 12}) == 0      # the four A variants' tuples, 4 registers apart


def test_reads_fill_exactly_the_registers_the_products_use():
    mac = _macros()

    def read_regs(body):
        regs = set()
        for lo, hi in re.findall(r"TWK_RD_[AB]\((\d+), (\d+)", body):
            regs |= set(range(int(lo), int(hi) + 1))
        return regs

    def product_regs(*names):
        regs = set()
        for n in names:
            for g in _g12(mac[n]):
                regs |= set(g[:10])
        return regs

    a_p, a_r = read_regs(mac["TWK_READ_P"]), read_regs(mac["TWK_READ_R"])
    bx, by = read_regs(mac["TWK_READ_BX"]), read_regs(mac["TWK_READ_BY"])
    assert len(a_p) == len(a_r) == 32 and len(bx) == len(by) == 8
    assert not (a_p & a_r) and not (bx & by) and not ((a_p | a_r) & (bx | by))
    assert product_regs("TWK_HALF0_P", "TWK_HALF1_P") == a_p | bx | by
    assert product_regs("TWK_HALF0_R", "TWK_HALF1_R") == a_r | bx | by
    assert product_regs("TWK_HALF0_P") <= a_p | bx and product_regs("TWK_HALF1_P") <= a_p | by
    # the clobber lists name exactly those registers
    clob = lambda n: {int(x) for x in re.findall(r'"v(\d+)"', mac[n])}
    assert clob("TWK_CLOBBER_P") == a_p and clob("TWK_CLOBBER_R") == a_r and clob("TWK_CLOBBER_B") == bx | by
    # H tuples on registers = 0, Q tuples on registers = 2 (mod 4); B pairs: H on = 0, Q on = 2
    for body in (mac["TWK_READ_P"], mac["TWK_READ_R"]):
        los = [int(lo) for lo, _ in re.findall(r"TWK_RD_A\((\d+), (\d+)", body)]
        assert [lo % 4 for lo in los] == [0, 2] * 4
    for body in (mac["TWK_READ_BX"], mac["TWK_READ_BY"]):
        los = [int(lo) for lo, _ in re.findall(r"TWK_RD_B\((\d+), (\d+)", body)]
        assert [lo % 4 for lo in los] == [0, 2, 0, 2]
