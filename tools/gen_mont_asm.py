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

acc = (lo,hi) 64-bit pair + ovf word.  Every product can carry out of the 64-bit pair (a shifted-in
accumulator of up to ~2^37 plus a product of up to 2^64-2^33+1), so every MAD is followed by an
ADDC; the first ADDC of a column is the e64 form 0 + 0 + carry, which initialises ovf without a
register zeroing.  Column 0 starts from the exact product a0*b0 (no carry possible).
"""
import os

NL = 8
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "gkr-mimc_amd", "csrc", "fr_mont_gen.inc")


def cexpr(o):
    k, i = o
    if k == "one":
        return "1u"
    return {"a": "a.v[%d]", "b": "b.v[%d]", "m": "m%d", "q": "FRQ%d", "A": "A[%d]", "ca": "ca.v[%d]", "cb": "cb.v[%d]"}[k] % i


def emit_asm(products, pos0):
    """One asm statement; pos0 = index of its first product within the column."""
    ops = []
    for x, y in products:
        for o in (x, y):
            if o not in ops and o[0] != "one":
                ops.append(o)
    names = {o: "%%%d" % (2 + n) for n, o in enumerate(ops)}
    names[("one", 0)] = "1"          # inline constant
    lines = []
    inits = False
    for k, (x, y) in enumerate(products):
        pos = pos0 + k
        lines.append("v_mad_u64_u32 %%0, vcc, %s, %s, %%0" % (names[x], names[y]))
        if pos == 0:
            lines.append("v_addc_co_u32_e64 %1, vcc, 0, 0, vcc")
            inits = True
        else:
            lines.append("v_addc_co_u32_e32 %1, vcc, 0, %1, vcc")
    ovf = '"=&v"(ovf)' if inits else '"+v"(ovf)'
    ins = ", ".join('"%s"(%s)' % ("s" if o[0] in ("q", "ca", "cb") else "v", cexpr(o)) for o in ops)
    body = '"' + '\\n\\t"\n        "'.join(lines) + '"'
    assert len(ops) + 2 <= 30
    return '    asm(%s\n        : "+v"(acc), %s\n        : %s\n        : "vcc");\n' % (body, ovf, ins)


def emit_portable(products, pos0):
    s = ""
    for k, (x, y) in enumerate(products):
        if pos0 + k == 0:
            s += "    ovf = 0;\n"
        s += "    FR_MADC(acc, ovf, %s, %s);\n" % (cexpr(x), cexpr(y))
    return s


def split(products, pos0, limit=30):
    chunks, cur = [], []
    for p in products:
        trial = cur + [p]
        ops = set()
        for x, y in trial:
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


def gen_mul():
    dev = host = "    u64 acc = (u64)a.v[0] * b.v[0];\n    u32 ovf;\n"
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
        pos = 0
        for ch in split(prods, pos):
            dev += emit_asm(ch, pos)
            host += emit_portable(ch, pos)
            pos += len(ch)
        if c < NL:
            t = "    const u32 m%d = (u32)acc * FR_QINV32;\n" % c
            dev += t + emit_asm([(("m", c), ("q", 0))], pos)
            host += t + emit_portable([(("m", c), ("q", 0))], pos)
        else:
            host += "    r.v[%d] = (u32)acc;\n" % (c - NL)
            # device: copy the finished limb out of the accumulator pair with an explicit move, otherwise
            # hipcc keeps every result limb in the low half of its own 64-bit register tuple (2x VGPRs)
            dev += '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (c - NL)
        sh = "    acc = (acc >> 32) | ((u64)ovf << 32);\n"
        dev += sh
        host += sh
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
        return "FRQ%d" % i
    return {"a": "a%d.v[%d]", "b": "b%d.v[%d]", "m": "m%d_%d"}[k] % (ch, i)


def emit_asm2(products, pos0):
    ops = []          # (operand, chain) ; q operands are shared (chain None)
    def key(o, ch):
        return (o, None) if o[0] == "q" else (o, ch)
    for x, y in products:
        for ch in (0, 1):
            for o in (x, y):
                if key(o, ch) not in ops:
                    ops.append(key(o, ch))
    base = 5      # %0 acc0, %1 ovf0, %2 acc1, %3 ovf1, %4 sgpr carry pair of chain 1
    names = {k: "%%%d" % (base + n) for n, k in enumerate(ops)}
    lines = []
    inits = False
    for k, (x, y) in enumerate(products):
        pos = pos0 + k
        lines.append("v_mad_u64_u32 %%0, vcc, %s, %s, %%0" % (names[key(x, 0)], names[key(y, 0)]))
        lines.append("v_mad_u64_u32 %%2, %%4, %s, %s, %%2" % (names[key(x, 1)], names[key(y, 1)]))
        if pos == 0:
            lines.append("v_addc_co_u32_e64 %1, vcc, 0, 0, vcc")
            lines.append("v_addc_co_u32_e64 %3, %4, 0, 0, %4")
            inits = True
        else:
            lines.append("v_addc_co_u32_e32 %1, vcc, 0, %1, vcc")
            lines.append("v_addc_co_u32_e64 %3, %4, 0, %3, %4")
    ovf = '"=&v"(ovf0), "+v"(acc1), "=&v"(ovf1)' if inits else '"+v"(ovf0), "+v"(acc1), "+v"(ovf1)'
    ins = ", ".join('"%s"(%s)' % ("s" if o[0] == "q" else "v", cexpr2(o, ch)) for (o, ch) in ops)
    body = '"' + '\\n\\t"\n        "'.join(lines) + '"'
    assert len(ops) + base <= 30, len(ops)
    return '    asm(%s\n        : "+v"(acc0), %s, "=&s"(sc)\n        : %s\n        : "vcc");\n' % (body, ovf, ins)


def split2(products, limit=30):
    chunks, cur = [], []
    for p in products:
        trial = cur + [p]
        ops = set()
        for x, y in trial:
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
    dev = ("    u64 acc0 = (u64)a0.v[0] * b0.v[0], acc1 = (u64)a1.v[0] * b1.v[0];\n"
           "    u32 ovf0, ovf1;\n    unsigned long long sc;\n")
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
        pos = 0
        for ch in split2(prods):
            dev += emit_asm2(ch, pos)
            pos += len(ch)
        if c < NL:
            dev += "    const u32 m0_%d = (u32)acc0 * FR_QINV32, m1_%d = (u32)acc1 * FR_QINV32;\n" % (c, c)
            dev += emit_asm2([(("m", c), ("q", 0))], pos)
        else:
            dev += '    r0.v[%d] = FR_LIMB_COPY((u32)acc0);\n' % (c - NL)
            dev += '    r1.v[%d] = FR_LIMB_COPY((u32)acc1);\n' % (c - NL)
        dev += "    acc0 = (acc0 >> 32) | ((u64)ovf0 << 32);\n    acc1 = (acc1 >> 32) | ((u64)ovf1 << 32);\n"
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
    """Device: the plain product a*b column by column (64 MAD/ADDC pairs, as in the multiplication without its
    Montgomery half); every finished limb is added straight into the accumulator limb by ONE carry-chained
    add whose carry lives in an SGPR pair (VCC belongs to the column arithmetic).  Host: same columns, u64."""
    dev = "    u64 acc = (u64)a.v[0] * b.v[0];\n    u32 ovf;\n    unsigned long long sc;\n"
    dev += '    asm("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(A[0]), "=&s"(sc) : "v"((u32)acc));\n'
    dev += "    acc >>= 32;\n"
    host = "    u64 acc = (u64)a.v[0] * b.v[0];\n    u32 ovf;\n    u64 cy;\n"
    host += "    cy = (u64)A[0] + (u32)acc;\n    A[0] = (u32)cy;\n    cy >>= 32;\n    acc >>= 32;\n"
    for c in range(1, 2 * NL - 1):
        lo_i, hi_i = max(0, c - (NL - 1)), min(c, NL - 1)
        prods = []
        for i in range(lo_i, hi_i + 1):
            prods.append((("a", i), ("b", c - i)))
        pos = 0
        for ch in split(prods, pos):
            dev += emit_asm(ch, pos)
            host += emit_portable(ch, pos)
            pos += len(ch)
        dev += '    asm("v_addc_co_u32_e64 %%0, %%1, %%0, %%2, %%1" : "+v"(A[%d]), "+s"(sc) : "v"((u32)acc));\n' % c
        host += "    cy += (u64)A[%d] + (u32)acc;\n    A[%d] = (u32)cy;\n    cy >>= 32;\n" % (c, c)
        sh = "    acc = (acc >> 32) | ((u64)ovf << 32);\n"
        dev += sh
        host += sh
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
    H = NL // 2
    dev = host = "    u64 acc = (u64)a.v[0] * ca.v[0];\n    u32 ovf;\n"
    ncol = H + NL - 1            # columns 0 .. 10 carry products; limbs H .. H+7 are the result
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
        pos = 0
        # column 0 starts from the exact product a0*ca0: its first listed product must use the carry-tracking form
        first = True
        for ch in split(prods, pos):
            if c == 0 and first:
                dev += emit_asm(ch, 0)
                host += emit_portable(ch, 0)
            else:
                dev += emit_asm(ch, pos if pos else (0 if c else 1))
                host += emit_portable(ch, pos if pos else (0 if c else 1))
            first = False
            pos += len(ch)
        if c < H:
            t = "    const u32 m%d = (u32)acc * FR_QINV32;\n" % c
            dev += t + emit_asm([(("m", c), ("q", 0))], max(pos, 1))
            host += t + emit_portable([(("m", c), ("q", 0))], max(pos, 1))
        else:
            host += "    r.v[%d] = (u32)acc;\n" % (c - H)
            dev += '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (c - H)
        sh = "    acc = (acc >> 32) | ((u64)ovf << 32);\n"
        dev += sh
        host += sh
    fin = "    r.v[%d] = (u32)acc;\n" % (NL - 1)
    dfin = '    r.v[%d] = FR_LIMB_COPY((u32)acc);\n' % (NL - 1)
    return dev + dfin, host + fin


def main():
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


if __name__ == "__main__":
    main()
