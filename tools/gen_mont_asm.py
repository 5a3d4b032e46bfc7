#!/usr/bin/env python3
"""Generate gkr-mimc_amd/csrc/fr_mont_gen.inc: the column-scheduled (product-scanning, FIPS)
BN254-Fr Montgomery multiplication body for gfx950.

Why generated: gfx950 has no carry-chained multiply-add.  The cheapest exact schedule is one
`v_mad_u64_u32` (32x32+64 -> 64, carry-out in VCC) plus one `v_addc_co_u32` (carry into a third
accumulator word) per limb product, walking the 15 columns of the 8x8 limb product with the
Montgomery m_i*q terms interleaved.  hipcc cannot be made to use the MAD's carry-out from C++ and
pads every inline-asm statement boundary with an s_nop, so each column is emitted as ONE asm
statement (split only where the 30-operand limit of inline asm would be exceeded).  The portable
branch (host compilers) executes the identical schedule with u64 arithmetic; the CPU unit test of
the schedule exercises it.

acc = (lo,hi) 64-bit pair + ovf word.  A MAD needs its ADDC only when the running column sum can reach
2^64.  The generator bounds that sum statically -- limbs of a, b and m_i up to 2^32-1 (or the tighter top limb a
precondition gives), the limbs of q exact -- orders each column's products by their largest possible value and
leaves the ADDC out for the leading products whose cumulative bound stays below 2^64 (BN254's q has four limbs
below 2^31, so three of a column's m_i*q_j products usually fit): 33 of the 136 ADDCs of a product disappear.
The first ADDC a column does execute is the e64 form 0 + 0 + carry, which initialises ovf without a register
zeroing; a column that needs none hands over acc >> 32.  The portable branch runs the identical schedule
(FR_MADN = multiply-add without carry tracking; with -DFR_CHECK_SKIPS it counts a wrap-around as a failure, which
is how the bounds are unit-tested on the CPU).
"""
import os

NL = 8
M32 = (1 << 32) - 1
# The modulus is a parameter: BN254's scalar field Fr (the sumcheck / NTT kernels) and its base field Fp (the curve
# arithmetic of the multi-scalar multiplications, g1.hip.h).  set_field() switches the limbs every bound is computed from and
# the prefix of the constants the generated text names (FRQ0.. / FR_QINV32 or FPQ0.. / FP_QINV32).
FIELDS = {
    "FR": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "FP": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
}
PFX = "FR"


def set_field(pfx):
    global PFX, QL, Q_INT, TOP_LT3Q, TOP_LTQ, TOP_LT2Q
    PFX = pfx
    Q_INT = FIELDS[pfx]
    QL = [(Q_INT >> (32 * i)) & M32 for i in range(NL)]
    TOP_LT3Q = (3 * Q_INT - 1) >> 224        # largest top limb of a value below 3q
    TOP_LTQ = (Q_INT - 1) >> 224             # ... of a canonical value
    TOP_LT2Q = (2 * Q_INT - 1) >> 224        # ... of a lazy product of operands below 2q


set_field("FR")
assert QL == [0xf0000001, 0x43e1f593, 0x79b97091, 0x2833e848, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72]


def opmax(o, bounds):
    k, i = o
    if k == "q":
        return QL[i]
    if k == "one":
        return 1
    return bounds.get((k, i), M32)


def plan(t_max, prods, bounds):
    """Order a column's products (smallest bound first) and decide which need their ADDC.
    Returns ([(x, y, track)], bound of the column sum)."""
    items = sorted(prods, key=lambda p: opmax(p[0], bounds) * opmax(p[1], bounds))
    s, out, tracking = t_max, [], False
    for x, y in items:
        pm = opmax(x, bounds) * opmax(y, bounds)
        if not tracking and s + pm < (1 << 64):
            out.append((x, y, False))
        else:
            tracking = True
            out.append((x, y, True))
        s += pm
    return out, s
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "gkr-mimc_amd", "csrc", "fr_mont_gen.inc")


def cexpr(o):
    k, i = o
    if k == "one":
        return "1u"
    return {"a": "a.v[%d]", "b": "b.v[%d]", "m": "m%d", "q": PFX + "Q%d", "A": "A[%d]", "ca": "ca.v[%d]", "cb": "cb.v[%d]",
            "d": "d%d", "e": "e%d"}[k] % i


def emit_asm(products, ovf_live):
    """One asm statement for [(x, y, track)]; ovf_live = the column's ovf word has been initialised already.
    Returns (text, ovf_live afterwards)."""
    ops = []
    for x, y, _t in products:
        for o in (x, y):
            if o not in ops and o[0] != "one":
                ops.append(o)
    uses_ovf = any(t for _x, _y, t in products)
    base = 2 if uses_ovf else 1
    names = {o: "%%%d" % (base + n) for n, o in enumerate(ops)}
    names[("one", 0)] = "1"          # inline constant
    lines = []
    inits = False
    for x, y, track in products:
        lines.append("v_mad_u64_u32 %%0, vcc, %s, %s, %%0" % (names[x], names[y]))
        if not track:
            continue
        if not ovf_live:
            lines.append("v_addc_co_u32_e64 %1, vcc, 0, 0, vcc")
            inits = True
            ovf_live = True
        else:
            lines.append("v_addc_co_u32_e32 %1, vcc, 0, %1, vcc")
    outs = '"+v"(acc)'
    if uses_ovf:
        outs += ', "=&v"(ovf)' if inits else ', "+v"(ovf)'
    ins = ", ".join('"%s"(%s)' % ("s" if o[0] in ("q", "ca", "cb") else "v", cexpr(o)) for o in ops)
    body = '"' + '\\n\\t"\n        "'.join(lines) + '"'
    assert len(ops) + base <= 30
    return '    asm(%s\n        : %s\n        : %s\n        : "vcc");\n' % (body, outs, ins), ovf_live


def emit_portable(products):
    s = ""
    for x, y, track in products:
        s += "    %s(acc, %s%s, %s);\n" % ("FR_MADC" if track else "FR_MADN", "ovf, " if track else "", cexpr(x), cexpr(y))
    return s


def split(products, limit=30):
    chunks, cur = [], []
    for p in products:
        trial = cur + [p]
        ops = set()
        for x, y, _t in trial:
            ops.add(x)
            ops.add(y)
        if len(ops) + 2 > limit and cur:
            chunks.append(cur)
            cur = [p]
        else:
            cur = trial
    if cur:
        chunks.append(cur)
    return chunks


def column(dev, host, t_max, prods, bounds, last=None):
    """Emit one column: the planned products, then (optionally) the product `last` that can only be formed once the
    others are in (m_c * q_0).  Returns (dev, host, bound of the column sum, whether ovf is live)."""
    planned, s_max = plan(t_max, prods, bounds)
    live = False
    host += "    ovf = 0;\n"
    for ch in split(planned):
        t, live = emit_asm(ch, live)
        dev += t
        host += emit_portable(ch)
    if last is not None:
        pm = opmax(last[1], bounds) * opmax(last[2], bounds)
        track = s_max + pm >= (1 << 64)
        t = last[0]
        e, live = emit_asm([(last[1], last[2], track)], live)
        dev += t + e
        host += t + emit_portable([(last[1], last[2], track)])
        s_max += pm
    if not live:
        dev += "    ovf = 0;\n"
    return dev, host, s_max


def gen_mul(bounds=None):
    bounds = dict(bounds or {})
    dev = host = "    u64 acc = (u64)a.v[0] * b.v[0];\n    u32 ovf;\n"
    t_max = M32 * M32
    for c in range(2 * NL - 1):
        prods = []
        lo_i, hi_i = max(0, c - (NL - 1)), min(c, NL - 1)
        if c > 0:
            for i in range(lo_i, hi_i + 1):
                prods.append((("a", i), ("b", c - i)))
        for i in range(lo_i, hi_i + 1):
            if c < NL and i == c:
                continue  # m_c * q_0 is added once m_c is known
            prods.append((("m", i), ("q", c - i)))
        last = ("    const u32 m%d = (u32)acc * %s_QINV32;\n" % (c, PFX), ("m", c), ("q", 0)) if c < NL else None
        dev, host, s_max = column(dev, host, t_max, prods, bounds, last)
        if c >= NL:
            host += "    r.v[%d] = (u32)acc;\n" % (c - NL)
            # device: copy the finished limb out of the accumulator pair with an explicit move, otherwise
            # hipcc keeps every result limb in the low half of its own 64-bit register tuple (2x VGPRs)
            dev += '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (c - NL)
        sh = "    acc = (acc >> 32) | ((u64)ovf << 32);\n"
        dev += sh
        host += sh
        t_max = s_max >> 32
    fin = "    r.v[%d] = (u32)acc;\n" % (NL - 1)
    dfin = '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (NL - 1)
    return dev + dfin, host + fin


# ------------------------------------------------------------------------------------------------
# squaring: the 28 cross products a_i*a_j (i < j) are taken ONCE, against the doubled operand, so the plain half of
# the product costs 36 limb products instead of 64 (the Montgomery half is unchanged: 100 instead of 128 in all).
#     a^2 = sum_i a_i^2 2^(64 i) + sum_i a_i 2^(32 i) * 2 (a - a mod 2^(32 (i + 1)))
# and the limbs of 2 (a - a mod 2^(32 (i + 1))) are those of 2a above position i + 1 (d_j = a_j << 1 | a_(j-1) >> 31)
# and, at position i + 1, the doubled limb without the bit that came up from a_i (e_(i+1) = a_(i+1) << 1).
# PRECONDITION a < 2q < 2^255: 2a fits eight limbs.  The result is the same integer a*a/2^256 + (m*q)/2^256 the
# general schedule produces (same T, same m_i), bit for bit.
# ------------------------------------------------------------------------------------------------
def gen_sqr():
    bounds = {("a", NL - 1): TOP_LT2Q, ("d", NL - 1): (4 * Q_INT - 1) >> 224}
    for j in range(1, NL):
        bounds[("e", j)] = M32 - 1
    pro = ""
    for j in range(1, NL):
        pro += "    const u32 e%d = a.v[%d] << 1;\n" % (j, j)
    for j in range(2, NL):
        pro += "    const u32 d%d = (a.v[%d] << 1) | (a.v[%d] >> 31);\n" % (j, j, j - 1)
    dev = host = pro + "    u64 acc = (u64)a.v[0] * a.v[0];\n    u32 ovf;\n"
    t_max = M32 * M32
    for c in range(2 * NL - 1):
        prods = []
        lo_i, hi_i = max(0, c - (NL - 1)), min(c, NL - 1)
        if c > 0:
            for i in range(lo_i, hi_i + 1):
                j = c - i
                if i > j:
                    continue
                if i == j:
                    prods.append((("a", i), ("a", i)))
                elif j == i + 1:
                    prods.append((("a", i), ("e", j)))
                else:
                    prods.append((("a", i), ("d", j)))
        for i in range(lo_i, hi_i + 1):
            if c < NL and i == c:
                continue
            prods.append((("m", i), ("q", c - i)))
        last = ("    const u32 m%d = (u32)acc * %s_QINV32;\n" % (c, PFX), ("m", c), ("q", 0)) if c < NL else None
        dev, host, s_max = column(dev, host, t_max, prods, bounds, last)
        if c >= NL:
            host += "    r.v[%d] = (u32)acc;\n" % (c - NL)
            dev += '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (c - NL)
        sh = "    acc = (acc >> 32) | ((u64)ovf << 32);\n"
        dev += sh
        host += sh
        t_max = s_max >> 32
    fin = "    r.v[%d] = (u32)acc;\n" % (NL - 1)
    dfin = '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (NL - 1)
    return dev + dfin, host + fin


# ------------------------------------------------------------------------------------------------
# mul2: two independent products with their instruction streams interleaved one-for-one, so that a wave
# that is alone on its SIMD (the small sumcheck rounds) still has an independent instruction to issue
# while the previous MAD/ADDC of the other chain is in flight.  Chain X carries in VCC, chain Y in a
# scratch SGPR pair (VOP3 forms of v_mad_u64_u32 / v_addc_co_u32 take any SGPR pair for the carry).
# ------------------------------------------------------------------------------------------------
def cexpr2(o, ch):
    k, i = o
    if k == "q":
        return PFX + "Q%d" % i
    return {"a": "a%d.v[%d]", "b": "b%d.v[%d]", "m": "m%d_%d"}[k] % (ch, i)


def emit_asm2(products, ovf_live):
    ops = []          # (operand, chain) ; q operands are shared (chain None)
    def key(o, ch):
        return (o, None) if o[0] == "q" else (o, ch)
    for x, y, _t in products:
        for ch in (0, 1):
            for o in (x, y):
                if key(o, ch) not in ops:
                    ops.append(key(o, ch))
    uses_ovf = any(t for _x, _y, t in products)
    # outputs: %0 acc0, [ovf0], acc1, [ovf1], sgpr carry pair of chain 1
    if uses_ovf:
        n_acc0, n_ovf0, n_acc1, n_ovf1, n_sc, base = "%0", "%1", "%2", "%3", "%4", 5
    else:
        n_acc0, n_ovf0, n_acc1, n_ovf1, n_sc, base = "%0", None, "%1", None, "%2", 3
    names = {k: "%%%d" % (base + n) for n, k in enumerate(ops)}
    lines = []
    inits = False
    for x, y, track in products:
        lines.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (n_acc0, names[key(x, 0)], names[key(y, 0)], n_acc0))
        lines.append("v_mad_u64_u32 %s, %s, %s, %s, %s" % (n_acc1, n_sc, names[key(x, 1)], names[key(y, 1)], n_acc1))
        if not track:
            continue
        if not ovf_live:
            lines.append("v_addc_co_u32_e64 %s, vcc, 0, 0, vcc" % n_ovf0)
            lines.append("v_addc_co_u32_e64 %s, %s, 0, 0, %s" % (n_ovf1, n_sc, n_sc))
            inits = True
            ovf_live = True
        else:
            lines.append("v_addc_co_u32_e32 %s, vcc, 0, %s, vcc" % (n_ovf0, n_ovf0))
            lines.append("v_addc_co_u32_e64 %s, %s, 0, %s, %s" % (n_ovf1, n_sc, n_ovf1, n_sc))
    if uses_ovf:
        outs = ('"+v"(acc0), "=&v"(ovf0), "+v"(acc1), "=&v"(ovf1)' if inits else '"+v"(acc0), "+v"(ovf0), "+v"(acc1), "+v"(ovf1)')
    else:
        outs = '"+v"(acc0), "+v"(acc1)'
    ins = ", ".join('"%s"(%s)' % ("s" if o[0] == "q" else "v", cexpr2(o, ch)) for (o, ch) in ops)
    body = '"' + '\\n\\t"\n        "'.join(lines) + '"'
    assert len(ops) + base <= 30, len(ops)
    return '    asm(%s\n        : %s, "=&s"(sc)\n        : %s\n        : "vcc");\n' % (body, outs, ins), ovf_live


def split2(products, limit=30):
    chunks, cur = [], []
    for p in products:
        trial = cur + [p]
        ops = set()
        for x, y, _t in trial:
            for o in (x, y):
                if o[0] == "q":
                    ops.add((o, None))
                else:
                    ops.add((o, 0))
                    ops.add((o, 1))
        if len(ops) + 5 > limit and cur:
            chunks.append(cur)
            cur = [p]
        else:
            cur = trial
    if cur:
        chunks.append(cur)
    return chunks


def gen_mul2():
    bounds = {}
    dev = ("    u64 acc0 = (u64)a0.v[0] * b0.v[0], acc1 = (u64)a1.v[0] * b1.v[0];\n"
           "    u32 ovf0, ovf1;\n    unsigned long long sc;\n")
    t_max = M32 * M32
    for c in range(2 * NL - 1):
        prods = []
        lo_i, hi_i = max(0, c - (NL - 1)), min(c, NL - 1)
        if c > 0:
            for i in range(lo_i, hi_i + 1):
                prods.append((("a", i), ("b", c - i)))
        for i in range(lo_i, hi_i + 1):
            if c < NL and i == c:
                continue
            prods.append((("m", i), ("q", c - i)))
        planned, s_max = plan(t_max, prods, bounds)
        live = False
        for ch in split2(planned):
            t, live = emit_asm2(ch, live)
            dev += t
        if c < NL:
            pm = M32 * QL[0]
            track = s_max + pm >= (1 << 64)
            dev += "    const u32 m0_%d = (u32)acc0 * %s_QINV32, m1_%d = (u32)acc1 * %s_QINV32;\n" % (c, PFX, c, PFX)
            t, live = emit_asm2([(("m", c), ("q", 0), track)], live)
            dev += t
            s_max += pm
        else:
            dev += '    r0.v[%d] = FR_LIMB_COPY((u32)acc0);\n' % (c - NL)
            dev += '    r1.v[%d] = FR_LIMB_COPY((u32)acc1);\n' % (c - NL)
        if not live:
            dev += "    ovf0 = 0;\n    ovf1 = 0;\n"
        dev += "    acc0 = (acc0 >> 32) | ((u64)ovf0 << 32);\n    acc1 = (acc1 >> 32) | ((u64)ovf1 << 32);\n"
        t_max = s_max >> 32
    dev += '    r0.v[%d] = FR_LIMB_COPY((u32)acc0);\n' % (NL - 1)
    dev += '    r1.v[%d] = FR_LIMB_COPY((u32)acc1);\n' % (NL - 1)
    return dev


# ------------------------------------------------------------------------------------------------
# wide multiply-accumulate: A (17 limbs, un-reduced) += a*b as a plain 512-bit product -- no Montgomery
# reduction.  Used for the products that only feed a sum: the sum is reduced once per thread instead of
# once per product (half of a multiplication's limb products are its reduction).  Column c adds the
# accumulator limb A[c] with one more MAD (x1), so the cost is 64+15 MAD/ADDC pairs against 136.
# ------------------------------------------------------------------------------------------------
def gen_mac_wide():
    """Device: the plain product a*b column by column (64 MADs, as in the multiplication without its Montgomery
    half); every finished limb is added straight into the accumulator limb by ONE carry-chained add whose carry
    lives in an SGPR pair (VCC belongs to the column arithmetic).  Host: same columns, u64.
    PRECONDITION a, b < 3q (every call site multiplies lazy Montgomery products or canonical elements): the top
    limbs are below 2^31.2, so the first product of the columns 7..14 needs no ADDC."""
    bounds = {("a", NL - 1): TOP_LT3Q, ("b", NL - 1): TOP_LT3Q}
    dev = "    u64 acc = (u64)a.v[0] * b.v[0];\n    u32 ovf;\n    unsigned long long sc;\n"
    dev += '    asm("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(A[0]), "=&s"(sc) : "v"((u32)acc));\n'
    dev += "    acc >>= 32;\n"
    host = "    u64 acc = (u64)a.v[0] * b.v[0];\n    u32 ovf;\n    u64 cy;\n"
    host += "    cy = (u64)A[0] + (u32)acc;\n    A[0] = (u32)cy;\n    cy >>= 32;\n    acc >>= 32;\n"
    t_max = (M32 * M32) >> 32
    for c in range(1, 2 * NL - 1):
        lo_i, hi_i = max(0, c - (NL - 1)), min(c, NL - 1)
        prods = []
        for i in range(lo_i, hi_i + 1):
            prods.append((("a", i), ("b", c - i)))
        dev, host, s_max = column(dev, host, t_max, prods, bounds)
        dev += '    asm("v_addc_co_u32_e64 %%0, %%1, %%0, %%2, %%1" : "+v"(A[%d]), "+s"(sc) : "v"((u32)acc));\n' % c
        host += "    cy += (u64)A[%d] + (u32)acc;\n    A[%d] = (u32)cy;\n    cy >>= 32;\n" % (c, c)
        sh = "    acc = (acc >> 32) | ((u64)ovf << 32);\n"
        dev += sh
        host += sh
        t_max = s_max >> 32
    # the last column's carry (acc < 2^32 now) and the chain carry go into limbs 15 and 16
    dev += '    asm("v_addc_co_u32_e64 %0, %1, %0, %2, %1" : "+v"(A[15]), "+s"(sc) : "v"((u32)acc));\n'
    dev += '    asm("v_addc_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(A[16]), "+s"(sc));\n'
    host += "    cy += (u64)A[15] + (u32)acc;\n    A[15] = (u32)cy;\n    cy >>= 32;\n    A[16] += (u32)cy;\n"
    return dev, host


# ------------------------------------------------------------------------------------------------
# multiplication by a launch-wide constant c (the fold challenge): with x = x_lo + 2^128 x_hi and the two
# host-prepared images ca = c * 2^-128 mod q, cb = c (both uniform, in SGPRs),
#     x * c / 2^256  ==  (x_lo * ca + x_hi * cb) / 2^128   (mod q)
# so only FOUR Montgomery steps are needed: 32 + 32 + 32 limb products instead of 64 + 64.  Result < 3q.
# ------------------------------------------------------------------------------------------------
def gen_mul_const2():
    """PRECONDITION a < 3q; ca, cb canonical (the host and the pyramid kernel store them reduced): the limbs a_7,
    ca_7, cb_7 are small, which the carry planning uses."""
    H = NL // 2
    bounds = {("a", NL - 1): TOP_LT3Q, ("ca", NL - 1): TOP_LTQ, ("cb", NL - 1): TOP_LTQ}
    dev = host = "    u64 acc = (u64)a.v[0] * ca.v[0];\n    u32 ovf;\n"
    ncol = H + NL - 1            # columns 0 .. 10 carry products; limbs H .. H+7 are the result
    t_max = M32 * M32
    for c in range(ncol):
        prods = []
        for i in range(H):
            j = c - i
            if 0 <= j < NL:
                if not (c == 0 and i == 0):
                    prods.append((("a", i), ("ca", j)))
                prods.append((("a", H + i), ("cb", j)))
        for i in range(H):
            j = c - i
            if 0 <= j < NL and not (c < H and i == c):
                prods.append((("m", i), ("q", j)))
        last = ("    const u32 m%d = (u32)acc * %s_QINV32;\n" % (c, PFX), ("m", c), ("q", 0)) if c < H else None
        dev, host, s_max = column(dev, host, t_max, prods, bounds, last)
        if c >= H:
            host += "    r.v[%d] = (u32)acc;\n" % (c - H)
            dev += '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (c - H)
        sh = "    acc = (acc >> 32) | ((u64)ovf << 32);\n"
        dev += sh
        host += sh
        t_max = s_max >> 32
    fin = "    r.v[%d] = (u32)acc;\n" % (NL - 1)
    dfin = '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (NL - 1)
    return dev + dfin, host + fin


def main():
    outs = os.path.join(os.path.dirname(OUT), "fr_sqr_gen.inc")
    dev, host = gen_sqr()
    with open(outs, "w") as f:
        f.write("// GENERATED by tools/gen_mont_asm.py -- do not edit.  Body of fr_mont_sqr_raw() in fr_bn254.h:\n"
                "// input `a` < 2q, output `r` = a*a/2^256 mod q in [0, 2q): 36 + 64 limb products (cross products once,\n"
                "// against the doubled operand).\n")
        f.write("#if defined(__HIP_DEVICE_COMPILE__)\n" + dev + "#else\n" + host + "#endif\n")
    print("wrote", outs)
    outc = os.path.join(os.path.dirname(OUT), "fr_mulc2_gen.inc")
    dev, host = gen_mul_const2()
    with open(outc, "w") as f:
        f.write("// GENERATED by tools/gen_mont_asm.py -- do not edit.  Body of fr_mul_const2() in fr_bn254.h:\n"
                "// r = (a_lo * ca + a_hi * cb) / 2^128 mod q with four Montgomery steps; r < 3q.\n")
        f.write("#if defined(__HIP_DEVICE_COMPILE__)\n" + dev + "#else\n" + host + "#endif\n")
    print("wrote", outc)
    outw = os.path.join(os.path.dirname(OUT), "fr_mac_wide_gen.inc")
    dev, host = gen_mac_wide()
    with open(outw, "w") as f:
        f.write("// GENERATED by tools/gen_mont_asm.py -- do not edit.  Body of fr_mac_wide() in fr_bn254.h:\n"
                "// A (17 x u32 limbs, un-reduced) += a*b as a plain 512-bit integer product.\n")
        f.write("#if defined(__HIP_DEVICE_COMPILE__)\n" + dev + "#else\n" + host + "#endif\n")
    print("wrote", outw)
    out2 = os.path.join(os.path.dirname(OUT), "fr_mont2_gen.inc")
    with open(out2, "w") as f:
        f.write("// GENERATED by tools/gen_mont_asm.py -- do not edit.  Device body of fr_mont_mul2_raw() in fr_bn254.h:\n"
                "// two independent lazy Montgomery products r0 = a0*b0/2^256, r1 = a1*b1/2^256 (both in [0, 2q)) with\n"
                "// their instruction streams interleaved (chain 0 carries in VCC, chain 1 in the SGPR pair `sc`).\n")
        f.write(gen_mul2())
    print("wrote", out2)
    dev, host = gen_mul()
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_mont_asm.py -- do not edit.  Body of fr_mont_mul_raw() in fr_bn254.h:\n"
                "// inputs `a`, `b` (Fr limbs, a*b < q*2^256), output `r` = a*b/2^256 mod q in [0, 2q).\n"
                "// Needs: u32/u64 typedefs, FRQ0..FRQ7, FR_QINV32, FR_MADC (portable branch).\n")
        f.write("#if defined(__HIP_DEVICE_COMPILE__)\n" + dev + "#else\n" + host + "#endif\n")
    print("wrote", OUT)
    # the base field Fp of the curve arithmetic (fp_bn254.h): every Fp value a kernel holds is below 2p, which the carry
    # planning of the product uses (top limbs below 2^31)
    set_field("FP")
    outp = os.path.join(os.path.dirname(OUT), "fp_mont_gen.inc")
    dev, host = gen_mul({("a", NL - 1): TOP_LT2Q, ("b", NL - 1): TOP_LT2Q})
    with open(outp, "w") as f:
        f.write("// GENERATED by tools/gen_mont_asm.py -- do not edit.  Body of fp_mul() in fp_bn254.h:\n"
                "// inputs `a`, `b` < 2p (BN254 base field), output `r` = a*b/2^256 mod p in [0, 2p).\n")
        f.write("#if defined(__HIP_DEVICE_COMPILE__)\n" + dev + "#else\n" + host + "#endif\n")
    print("wrote", outp)
    outp = os.path.join(os.path.dirname(OUT), "fp_sqr_gen.inc")
    dev, host = gen_sqr()
    with open(outp, "w") as f:
        f.write("// GENERATED by tools/gen_mont_asm.py -- do not edit.  Body of fp_sqr() in fp_bn254.h:\n"
                "// input `a` < 2p, output `r` = a*a/2^256 mod p in [0, 2p), the integer fp_mul(a, a) returns.\n")
        f.write("#if defined(__HIP_DEVICE_COMPILE__)\n" + dev + "#else\n" + host + "#endif\n")
    print("wrote", outp)
    set_field("FR")


if __name__ == "__main__":
    main()
