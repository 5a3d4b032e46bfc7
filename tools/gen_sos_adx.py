# emits C++ inline asm: (A) the 8-limb product T = x*y, (B) one 256-bit Montgomery step on T in registers
def lines_to_asm(L):
    return "\n".join('        "%s\\n\\t"' % s for s in L)

def gen_product():
    L = []; e = L.append
    T = ["%%[t%d]" % i for i in range(8)]
    Y = ["%d(%%[y])" % (8*j) for j in range(4)]
    for i in range(4):
        e(f"movq {8*i}(%[x]), %%rdx")
        if i == 0:
            e(f"mulx {Y[0]}, {T[0]}, {T[1]}")
            e(f"mulx {Y[1]}, %%rax, {T[2]}"); e(f"addq %%rax, {T[1]}")
            e(f"mulx {Y[2]}, %%rax, {T[3]}"); e(f"adcq %%rax, {T[2]}")
            e(f"mulx {Y[3]}, %%rax, {T[4]}"); e(f"adcq %%rax, {T[3]}")
            e(f"adcq $0, {T[4]}")
        else:
            e("xorl %%eax, %%eax")
            for j in range(3):
                e(f"mulx {Y[j]}, %%rax, %[s]")
                e(f"adox %%rax, {T[i+j]}")
                e(f"adcx %[s], {T[i+j+1]}")
            e(f"mulx {Y[3]}, %%rax, {T[i+4]}")
            e(f"adox %%rax, {T[i+3]}")
            e("movl $0, %%eax")
            e(f"adox %%rax, {T[i+4]}")
            e(f"adcx %%rax, {T[i+4]}")
    return L

def gen_reduce():
    L = []; e = L.append
    T = ["%%[t%d]" % i for i in range(8)]
    M = ["%%[m%d]" % i for i in range(4)]
    N = ["%%[n%d]" % i for i in range(4)]
    Qc = ["%%[q%d]" % i for i in range(4)]
    # phase 2: m = T_lo * N mod 2^256 (triangular product)
    # (imul writes CF and OF: the single products are taken before each carry chain starts)
    e(f"movq {T[0]}, %%rdx")
    e(f"movq {T[0]}, {M[3]}"); e(f"imulq {N[3]}, {M[3]}")
    e(f"mulx {N[0]}, {M[0]}, {M[1]}")
    e(f"mulx {N[1]}, %%rax, {M[2]}"); e(f"addq %%rax, {M[1]}")
    e(f"mulx {N[2]}, %%rax, %[s]"); e(f"adcq %%rax, {M[2]}"); e(f"adcq %[s], {M[3]}")
    e(f"movq {T[1]}, %%rdx")
    e(f"movq {T[1]}, %%rax"); e(f"imulq {N[2]}, %%rax"); e(f"addq %%rax, {M[3]}")
    e("xorl %%eax, %%eax")
    e(f"mulx {N[0]}, %%rax, %[s]"); e(f"adox %%rax, {M[1]}"); e(f"adcx %[s], {M[2]}")
    e(f"mulx {N[1]}, %%rax, %[s]"); e(f"adox %%rax, {M[2]}"); e(f"adcx %[s], {M[3]}")
    e("movl $0, %%eax"); e(f"adox %%rax, {M[3]}")
    e(f"movq {T[2]}, %%rdx")
    e(f"movq {T[2]}, %%rax"); e(f"imulq {N[1]}, %%rax"); e(f"addq %%rax, {M[3]}")
    e(f"mulx {N[0]}, %%rax, %[s]"); e(f"addq %%rax, {M[2]}"); e(f"adcq %[s], {M[3]}")
    e(f"movq {T[3]}, %%rax"); e(f"imulq {N[0]}, %%rax"); e(f"addq %%rax, {M[3]}")
    # phase 3: T += m * q, row by row; carries rippled to T7
    for i in range(4):
        e(f"movq {M[i]}, %%rdx")
        e("xorl %%eax, %%eax")
        for j in range(4):
            e(f"mulx {Qc[j]}, %%rax, %[s]")
            e(f"adox %%rax, {T[i+j]}")
            e(f"adcx %[s], {T[i+j+1]}")
        e("movl $0, %%eax")
        # OF belongs to T[i+4], CF to T[i+5]
        if i + 4 <= 7: e(f"adox %%rax, {T[i+4]}")
        for k in range(i + 5, 8):
            e(f"adcx %%rax, {T[k]}")
            e(f"adox %%rax, {T[k]}")
    return L

prod = lines_to_asm(gen_product())
red = lines_to_asm(gen_reduce())
print(f'''static const u64 QINV256[4] = {{0xc2e1f593efffffffULL, 0x6586864b4c6911b3ULL, 0xe39a982899062391ULL, 0x73f82f1d0d8341b2ULL}};
static inline E mul_sos256_adx(const E& x, const E& y) {{
    u64 t0, t1, t2, t3, t4, t5, t6, t7, s, m0, m1, m2, m3;
    asm(
{prod}
        : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [t4] "=&r"(t4), [t5] "=&r"(t5), [t6] "=&r"(t6), [t7] "=&r"(t7), [s] "=&r"(s)
        : [x] "r"(x.l), [y] "r"(y.l), "m"(*(const u64(*)[4])x.l), "m"(*(const u64(*)[4])y.l)
        : "rax", "rdx", "cc");
    asm(
{red}
        : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4), [t5] "+&r"(t5), [t6] "+&r"(t6), [t7] "+&r"(t7), [s] "=&r"(s),
          [m0] "=&r"(m0), [m1] "=&r"(m1), [m2] "=&r"(m2), [m3] "=&r"(m3)
        : [n0] "m"(QINV256[0]), [n1] "m"(QINV256[1]), [n2] "m"(QINV256[2]), [n3] "m"(QINV256[3]),
          [q0] "m"(Q[0]), [q1] "m"(Q[1]), [q2] "m"(Q[2]), [q3] "m"(Q[3])
        : "rax", "rdx", "cc");
    E r = {{{{t4, t5, t6, t7}}}};
    return r;
}}
''')
