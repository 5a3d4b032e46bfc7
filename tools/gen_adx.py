# emits a C++ inline-asm Montgomery product (no-carry CIOS, mulx + adcx/adox dual carry chains)
def gen(name="mul_adx"):
    L = []
    e = L.append
    T = ["%[t0]", "%[t1]", "%[t2]", "%[t3]"]
    Y = ["%[y0]", "%[y1]", "%[y2]", "%[y3]"]
    A = "%[A]"
    for i in range(4):
        e(f"movq {8*i}(%[x]), %%rdx")
        if i == 0:
            e(f"mulx {Y[0]}, {T[0]}, {T[1]}")
            e(f"mulx {Y[1]}, %%rax, {T[2]}")
            e(f"addq %%rax, {T[1]}")
            e(f"mulx {Y[2]}, %%rax, {T[3]}")
            e(f"adcq %%rax, {T[2]}")
            e(f"mulx {Y[3]}, %%rax, {A}")
            e(f"adcq %%rax, {T[3]}")
            e(f"adcq $0, {A}")
        else:
            e("xorl %%eax, %%eax")
            for j in range(3):
                e(f"mulx {Y[j]}, %%rax, %[s]")
                e(f"adox %%rax, {T[j]}")
                e(f"adcx %[s], {T[j+1]}")
            e(f"mulx {Y[3]}, %%rax, {A}")
            e(f"adox %%rax, {T[3]}")
            e("movl $0, %%eax")
            e(f"adox %%rax, {A}")
            e(f"adcx %%rax, {A}")
        # reduction row
        e(f"movq {T[0]}, %%rdx")
        e("imulq %[qinv], %%rdx")
        e("xorl %%eax, %%eax")
        e(f"mulx 0(%[q]), %%rax, %[s]")
        e(f"adcx {T[0]}, %%rax")
        e(f"movq %[s], {T[0]}")
        for j in range(1, 4):
            e(f"adcx {T[j]}, {T[j-1]}")
            e(f"mulx {8*j}(%[q]), %%rax, {T[j]}")
            e(f"adox %%rax, {T[j-1]}")
        e("movl $0, %%eax")
        e(f"adcx %%rax, {T[3]}")
        e(f"adox {A}, {T[3]}")
    body = "\n".join('        "%s\\n\\t"' % s for s in L)
    return f'''static inline E {name}(const E& x, const E& y) {{
    u64 t0, t1, t2, t3, A, s;
    asm(
{body}
        : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [A] "=&r"(A), [s] "=&r"(s)
        : [x] "r"(x.l), [y0] "r"(y.l[0]), [y1] "r"(y.l[1]), [y2] "r"(y.l[2]), [y3] "r"(y.l[3]), [q] "r"(Q), [qinv] "m"(QINV),
          "m"(*(const u64(*)[4])x.l)
        : "rax", "rdx", "cc");
    E r = {{{{t0, t1, t2, t3}}}};
    return r;
}}
'''
print(gen())
