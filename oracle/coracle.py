"""ctypes binding of oracle/libgkr_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY.

Field-element arrays are numpy uint64 arrays of shape (n, 4): the gnark-crypto `fr.Element` memory
image (little-endian Montgomery limbs), i.e. exactly what a Go `[]fr.Element` holds.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_here = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_here, "libgkr_oracle.so")

GATE_IDENTITY, GATE_CIPHER, GATE_ADD, GATE_SUM, GATE_SUM_POW7 = 0, 1, 2, 3, 4


def build():
    subprocess.check_call(["make", "-C", _here, "-s"])


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (a container can
    see 256 logical CPUs while being allowed a handful; OpenMP would then oversubscribe badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, n)


def _load():
    # make is a no-op when the library is current; a stale one (sources changed, .so git-ignored) is rebuilt.  On a box
    # without make/gcc the prebuilt library that travelled with the tree is used as it is.
    try:
        build()
    except Exception:
        if not os.path.exists(_SO):
            raise
    # libgomp reads these when it is first loaded: sleep instead of spinning at barriers
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("OMP_PROC_BIND", "false")
    lib = C.CDLL(_SO)
    P = C.c_void_p
    sig = {
        "oracle_fr_from_u64": (None, [P, C.c_uint64]),
        "oracle_fr_mul": (None, [P, P, P]),
        "oracle_fr_mul_generic": (None, [P, P, P]),
        "oracle_bench_fr_mul": (None, [C.c_long, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "oracle_fr_add": (None, [P, P, P]),
        "oracle_fr_sub": (None, [P, P, P]),
        "oracle_fr_inverse": (None, [P, P]),
        "oracle_fr_to_regular": (None, [P, P]),
        "oracle_fr_from_regular": (None, [P, P]),
        "oracle_mimc_hash": (None, [P, P, C.c_size_t]),
        "oracle_mimc_keyed_permutation": (None, [P, P, P]),
        "oracle_random_fr_array": (None, [P, C.c_size_t]),
        "oracle_get_ark": (None, [P, C.c_int]),
        "oracle_fold": (None, [P, C.c_size_t, P]),
        "oracle_evaluate": (None, [P, P, C.c_size_t, P, C.c_int]),
        "oracle_eval_eq": (None, [P, P, P, C.c_int]),
        "oracle_folded_eq_table": (None, [P, P, C.c_int, P]),
        "oracle_chunk_of_eq_table": (None, [P, C.c_size_t, C.c_size_t, P, C.c_int, P]),
        "oracle_eval_univariate": (None, [P, P, C.c_int, P]),
        "oracle_lagrange_coefficient": (None, [P, C.c_int]),
        "oracle_interpolate_on_range": (C.c_int, [P, P, C.c_int]),
        "oracle_gate_eval_batch": (None, [C.c_int, P, P, P, C.c_int, C.c_size_t]),
        "oracle_sumcheck_prove": (C.c_int, [C.c_int, P, C.c_int, C.c_int, P, P, C.c_int, P, C.c_int, P, P, P]),
        "oracle_sumcheck_verify": (C.c_int, [P, C.c_int, P, C.c_int, C.c_int, P, P, P]),
        "oracle_evaluation": (None, [P, C.c_int, P, P, C.c_int, C.c_int, P, C.c_int, P, C.c_int]),
        "oracle_mimc_proof_len": (C.c_size_t, [C.c_int]),
        "oracle_gkr_prove_mimc": (C.c_int, [C.c_int, P, P, P, P, P, P]),
        "oracle_gkr_verify_mimc": (C.c_int, [C.c_int, P, P, P, P, P]),
        "oracle_circuit_proof_len": (C.c_size_t, [P, C.c_int, C.c_int]),
        "oracle_gkr_prove_circuit": (C.c_int, [P, C.c_int, C.c_int, P, C.c_int, P, P, P, P]),
        "oracle_gkr_verify_circuit": (C.c_int, [P, C.c_int, C.c_int, P, P, C.c_int, P, P]),
        "oracle_g1_on_curve": (C.c_int, [P]),
        "oracle_g1_scalar_mul": (None, [P, P, P]),
        "oracle_g1_batch_scalar_mul": (None, [P, P, P, C.c_size_t]),
        "oracle_g1_add": (None, [P, P, P]),
        "oracle_g1_msm": (None, [P, P, P, C.c_size_t]),
        "oracle_num_threads": (C.c_int, []),
        "oracle_set_num_threads": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    return lib


lib = _load()
lib.oracle_set_num_threads(int(os.environ.get("GKR_ORACLE_THREADS", min(usable_cpus(), 64))))


def bench_fr_mul(n=2_000_000, generic=False):
    """(ns per dependent multiplication, ns per independent multiplication) on one core."""
    a, b = C.c_double(0), C.c_double(0)
    lib.oracle_bench_fr_mul(n, 1 if generic else 0, C.byref(a), C.byref(b))
    return a.value, b.value


def fr(n=1):
    return np.zeros((n, 4), dtype=np.uint64)


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def _ptr_array(tables):
    arr = (C.c_void_p * len(tables))(*[t.ctypes.data for t in tables])
    return arr


def from_u64(v):
    o = fr()
    lib.oracle_fr_from_u64(_p(o), int(v))
    return o


def from_ints(vals):
    """Regular-form Python ints -> Montgomery (n,4) array."""
    vals = list(vals)
    reg = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            reg[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    out = fr(len(vals))
    for i in range(len(vals)):
        lib.oracle_fr_from_regular(out[i:].ctypes.data, reg[i:].ctypes.data)
    return out


def to_ints(arr):
    arr = np.ascontiguousarray(arr).reshape(-1, 4)
    out = []
    tmp = np.zeros(4, dtype=np.uint64)
    for i in range(arr.shape[0]):
        lib.oracle_fr_to_regular(tmp.ctypes.data, arr[i:].ctypes.data)
        out.append(sum(int(tmp[k]) << (64 * k) for k in range(4)))
    return out


def mimc_hash(arr):
    o = fr()
    arr = np.ascontiguousarray(arr)
    lib.oracle_mimc_hash(_p(o), _p(arr), arr.shape[0])
    return o


def mimc_keyed_permutation(x, key):
    o = fr()
    lib.oracle_mimc_keyed_permutation(_p(o), _p(x), _p(key))
    return o


def random_fr_array(n):
    o = fr(n)
    lib.oracle_random_fr_array(_p(o), n)
    return o


def fold(tbl, r):
    t = np.array(tbl, copy=True)
    lib.oracle_fold(_p(t), t.shape[0], _p(r))
    return t[: t.shape[0] // 2].copy()


def evaluate(tbl, coords):
    o = fr()
    lib.oracle_evaluate(_p(o), _p(np.ascontiguousarray(tbl)), tbl.shape[0], _p(np.ascontiguousarray(coords)),
                        coords.shape[0])
    return o


def eval_eq(q, h):
    o = fr()
    lib.oracle_eval_eq(_p(o), _p(q), _p(h), q.shape[0])
    return o


def folded_eq_table(q, mult=None):
    n = q.shape[0]
    o = fr(1 << n)
    lib.oracle_folded_eq_table(_p(o), _p(np.ascontiguousarray(q)), n, _p(mult))
    return o


def chunked_eq_table(q, chunk_size, mult=None):
    n = q.shape[0]
    o = fr(1 << n)
    for cid in range((1 << n) // chunk_size):
        lib.oracle_chunk_of_eq_table(_p(o), cid, chunk_size, _p(q), n, _p(mult))
    return o


def eval_univariate(coeffs, x):
    o = fr()
    lib.oracle_eval_univariate(_p(o), _p(np.ascontiguousarray(coeffs)), coeffs.shape[0], _p(x))
    return o


def lagrange_coefficient(domain):
    o = fr(domain * domain)
    lib.oracle_lagrange_coefficient(_p(o), domain)
    return o.reshape(domain, domain, 4)


def interpolate_on_range(values):
    o = fr(values.shape[0])
    rc = lib.oracle_interpolate_on_range(_p(o), _p(np.ascontiguousarray(values)), values.shape[0])
    assert rc == 0
    return o


def gate_eval_batch(gate, ark, xs):
    """Gate.EvalBatch (circuit/gates.go:16): res[i] = gate(xs[0][i], xs[1][i], ...)."""
    xs = [np.ascontiguousarray(x) for x in xs]
    res = fr(xs[0].shape[0])
    ark = fr() if ark is None else np.ascontiguousarray(ark)
    lib.oracle_gate_eval_batch(gate, _p(ark), _p(res), _ptr_array(xs), len(xs), xs[0].shape[0])
    return res


def gate_degree(gate):
    return 7 if gate in (GATE_CIPHER, GATE_SUM_POW7) else 1


def sumcheck_prove(gate, ark, X, qprimes, claims):
    """X: list of (2^bN,4) arrays (copied; the C function consumes its inputs). qprimes: (nq,bN,4)."""
    qprimes = np.ascontiguousarray(qprimes)
    nq, bN = qprimes.shape[0], qprimes.shape[1]
    Xc = [np.array(x, copy=True) for x in X]
    nc = gate_degree(gate) + 2
    proof, chal, final = fr(max(bN * nc, 1)), fr(max(bN, 1)), fr(len(X) + 1)
    claims = np.ascontiguousarray(claims).reshape(-1, 4)
    ark = fr() if ark is None else ark
    rc = lib.oracle_sumcheck_prove(gate, _p(ark), len(Xc), bN, _ptr_array(Xc), _p(qprimes), nq,
                                   _p(claims) if claims.shape[0] else None, claims.shape[0],
                                   _p(proof), _p(chal), _p(final))
    if rc != 0:
        raise RuntimeError("oracle_sumcheck_prove rc=%d" % rc)
    return proof[: bN * nc].reshape(bN, nc, 4), chal[:bN], final


def sumcheck_verify(claims, proof):
    bN, nc = proof.shape[0], proof.shape[1]
    chal, final, recomb = fr(max(bN, 1)), fr(), fr()
    claims = np.ascontiguousarray(claims).reshape(-1, 4)
    rc = lib.oracle_sumcheck_verify(_p(claims), claims.shape[0], _p(np.ascontiguousarray(proof)), bN, nc,
                                    _p(chal), _p(final), _p(recomb))
    return rc, chal[:bN], final, recomb


def evaluation(gate, ark, qprimes, claims, X):
    qprimes = np.ascontiguousarray(qprimes)
    nq, bN = qprimes.shape[0], qprimes.shape[1]
    claims = np.ascontiguousarray(claims).reshape(-1, 4)
    o = fr()
    ark = fr() if ark is None else ark
    Xc = [np.ascontiguousarray(x) for x in X]
    lib.oracle_evaluation(_p(o), gate, _p(ark), _p(qprimes), nq, bN, _p(claims) if claims.shape[0] else None,
                          claims.shape[0], _ptr_array(Xc), len(Xc))
    return o


def mimc_proof_len(bN):
    return lib.oracle_mimc_proof_len(bN)


def gkr_prove_mimc(bN, in0, in1, qprime, want_outputs=True):
    flat = fr(mimc_proof_len(bN))
    outs = fr(1 << bN) if want_outputs else None
    secs = C.c_double(0.0)
    qprime = np.ascontiguousarray(qprime).reshape(-1, 4)
    rc = lib.oracle_gkr_prove_mimc(bN, _p(np.ascontiguousarray(in0)), _p(np.ascontiguousarray(in1)),
                                   _p(qprime) if bN else None, _p(flat), _p(outs), C.byref(secs))
    if rc != 0:
        raise RuntimeError("oracle_gkr_prove_mimc rc=%d" % rc)
    return flat, outs, secs.value


def gkr_verify_mimc(bN, flat, in0, in1, outputs, qprime):
    qprime = np.ascontiguousarray(qprime).reshape(-1, 4)
    return lib.oracle_gkr_verify_mimc(bN, _p(np.ascontiguousarray(flat)), _p(np.ascontiguousarray(in0)),
                                      _p(np.ascontiguousarray(in1)), _p(np.ascontiguousarray(outputs)),
                                      _p(qprime) if bN else None)


class LayerDesc(C.Structure):
    _fields_ = [("gate", C.c_int), ("n_in", C.c_int), ("in_", C.c_int * 4), ("ark", C.c_uint64 * 4)]


def circuit_descs(pycircuit):
    """pyoracle circuit (list of Layer) -> C array of oracle_layer_desc."""
    import pyoracle as o
    kinds = {"identity": GATE_IDENTITY, "cipher": GATE_CIPHER, "add": GATE_ADD, "sum": GATE_SUM, "sum_pow7": GATE_SUM_POW7}
    arr = (LayerDesc * len(pycircuit))()
    for i, lay in enumerate(pycircuit):
        arr[i].gate = -1 if lay.gate is None else kinds[lay.gate.kind]
        arr[i].n_in = len(lay.In)
        for k, v in enumerate(lay.In):
            arr[i].in_[k] = v
        limbs = o.to_mont_limbs(getattr(lay.gate, "ark", 0)) if lay.gate is not None else [0, 0, 0, 0]
        for k in range(4):
            arr[i].ark[k] = limbs[k]
    return arr


def gkr_prove_circuit(descs, bN, inputs, qprime):
    n_layers = len(descs)
    flat = fr(lib.oracle_circuit_proof_len(C.cast(descs, C.c_void_p), n_layers, bN))
    outs = fr(1 << bN)
    secs = C.c_double(0.0)
    ins = [np.ascontiguousarray(x) for x in inputs]
    qprime = np.ascontiguousarray(qprime).reshape(-1, 4)
    rc = lib.oracle_gkr_prove_circuit(C.cast(descs, C.c_void_p), n_layers, bN, _ptr_array(ins), len(ins),
                                      _p(qprime) if bN else None, _p(flat), _p(outs), C.byref(secs))
    if rc != 0:
        raise RuntimeError("oracle_gkr_prove_circuit rc=%d" % rc)
    return flat, outs, secs.value


def gkr_verify_circuit(descs, bN, flat, inputs, outputs, qprime):
    ins = [np.ascontiguousarray(x) for x in inputs]
    qprime = np.ascontiguousarray(qprime).reshape(-1, 4)
    return lib.oracle_gkr_verify_circuit(C.cast(descs, C.c_void_p), len(descs), bN, _p(np.ascontiguousarray(flat)),
                                         _ptr_array(ins), len(ins), _p(np.ascontiguousarray(outputs)),
                                         _p(qprime) if bN else None)


# ---- BN254 G1 (g1_oracle.c; parity unpinned: see the header there) -------------------------------------------------
G1_GEN = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f,
                   0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e], dtype=np.uint64)   # (1, 2), Montgomery


def g1_on_curve(pt):
    pt = np.ascontiguousarray(pt, dtype=np.uint64)
    return bool(lib.oracle_g1_on_curve(_p(pt)))


def g1_scalar_mul(base, scalar):
    out = np.zeros(8, dtype=np.uint64)
    lib.oracle_g1_scalar_mul(_p(out), _p(np.ascontiguousarray(base, dtype=np.uint64)), _p(np.ascontiguousarray(scalar, dtype=np.uint64)))
    return out


def g1_batch_scalar_mul(base, scalars):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros((scalars.shape[0], 8), dtype=np.uint64)
    lib.oracle_g1_batch_scalar_mul(_p(out), _p(np.ascontiguousarray(base, dtype=np.uint64)), _p(scalars), scalars.shape[0])
    return out


def g1_add(a, b):
    out = np.zeros(8, dtype=np.uint64)
    lib.oracle_g1_add(_p(out), _p(np.ascontiguousarray(a, dtype=np.uint64)), _p(np.ascontiguousarray(b, dtype=np.uint64)))
    return out


def g1_msm(points, scalars):
    points = np.ascontiguousarray(points, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    assert points.shape[0] == scalars.shape[0]
    out = np.zeros(8, dtype=np.uint64)
    lib.oracle_g1_msm(_p(out), _p(points), _p(scalars), points.shape[0])
    return out
