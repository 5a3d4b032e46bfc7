"""Python mirror of the reference's Go API for the hot path, over the C ABI of libgkrhip.so
(include/gkrhip.h).  Names and argument meaning follow the Go functions they stand for:

    fold(table, r)                    (*MultiLin).Fold            poly/multilin.go:19-23
    evaluate(table, coords)           MultiLin.Evaluate           poly/multilin.go:59-66
    folded_eq_table(q, mult=None)     poly.FoldedEqTable          poly/eq.go:41-59
    gate_eval_batch(gate, ark, xs)    Gate.EvalBatch              circuit/gates.go:16
    sumcheck_prove(X, qPrimes, claims, gate, ark)   sumcheck.Prove   sumcheck/prover.go:46-90
    gkr_prove_mimc(in0, in1, qPrime)  Assign + gkr.Prove on examples.MimcCircuit (gkr/prover.go:21-47)
    MimcSession                       resident assignment, repeated Prove (benchmarks)

Field-element arrays are numpy uint64 arrays of shape (n, 4): the memory image of a Go
`[]fr.Element`.  Errors raise GkrHipError (the reference panics).  There is no CPU fallback: if the
library or a gfx950 GPU is missing, the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

GATE_IDENTITY, GATE_CIPHER, GATE_ADD = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgkrhip.so")
_lib = None

# every symbol include/gkrhip.h declares: (restype, argtypes)
_P, _SZ, _I, _U64, _D = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_double
ABI = {
    "gkrhip_init": (_I, [_I]),
    "gkrhip_shutdown": (None, []),
    "gkrhip_device_count": (_I, []),
    "gkrhip_last_error": (C.c_char_p, []),
    "gkrhip_last_error_r": (_SZ, [_I, C.c_char_p, _SZ]),
    "gkrhip_sumcheck_verify": (_I, [_P, _I, _P, _I, _I, _P, _P, _P]),
    "gkrhip_version": (C.c_char_p, []),
    "gkrhip_build_id": (C.c_char_p, []),
    "gkrhip_device_synchronize": (_I, []),
    "gkrhip_mem_info": (_I, [C.POINTER(_SZ), C.POINTER(_SZ)]),
    "gkrhip_host_alloc": (_I, [C.POINTER(C.c_void_p), _SZ]),
    "gkrhip_host_free": (None, [C.c_void_p]),
    "gkrhip_reserve_lanes": (_I, [_I]),
    "gkrhip_set_option": (_I, [C.c_char_p, C.c_long]),
    "gkrhip_fold": (_I, [_P, _SZ, _P]),
    "gkrhip_evaluate": (_I, [_P, _P, _SZ, _P, _I]),
    "gkrhip_eq_table": (_I, [_P, _P, _I, _P]),
    "gkrhip_chunk_of_eq_table": (_I, [_P, _SZ, _SZ, _P, _I, _P]),
    "gkrhip_gate_eval_batch": (_I, [_I, _P, _P, _P, _I, _SZ]),
    "gkrhip_sumcheck_prove": (_I, [_I, _P, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P]),
    "gkrhip_mimc_proof_len": (_SZ, [_I]),
    "gkrhip_gkr_prove_mimc": (_I, [_I, _P, _P, _P, _P, _P]),
    "gkrhip_gkr_prove_mimc_regular": (_I, [_I, _P, _P, _P, _P, _P]),
    "gkrhip_mimc_session_create": (_I, [C.POINTER(_P), _I]),
    "gkrhip_mimc_session_load_inputs": (_I, [_P, _P, _P]),
    "gkrhip_mimc_session_synth_inputs": (_I, [_P, _U64, _U64]),
    "gkrhip_mimc_session_assign": (_I, [_P]),
    "gkrhip_mimc_session_prove": (_I, [_P, _P, _P]),
    "gkrhip_mimc_session_prove_group": (_I, [_I, _P, _P, _P, _P]),
    "gkrhip_mimc_session_outputs": (_I, [_P, _P]),
    "gkrhip_mimc_session_evaluate_layer": (_I, [_P, _I, _P, _P]),
    "gkrhip_mimc_session_destroy": (None, [_P]),
    "gkrhip_session_create": (_I, [C.POINTER(_P), _P, _I, _I]),
    "gkrhip_session_load_input": (_I, [_P, _I, _P]),
    "gkrhip_session_proof_len": (_SZ, [_P]),
    "gkrhip_session_num_inputs": (_I, [_P]),
    "gkrhip_gmimc_t2_circuit": (_I, [_P, _I]),
    "gkrhip_gmimc_circuit": (_I, [_I, _P, _I, _P]),
    "gkrhip_gmimc_hash_circuit": (_I, [_I, _I, _P, _I, _P]),
    "gkrhip_gate_register": (_I, [_P, C.POINTER(_I)]),
    "gkrhip_gate_lookup": (_I, [_I, _P]),
    "gkrhip_gkr_verify": (_I, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "gkrhip_gkr_prove": (_I, [_P, _I, _I, _P, _I, _P, _P, _P]),
    "gkrhip_gkr_verify_mimc": (_I, [_I, _P, _P, _P, _P, _P]),
    "gkrhip_mimc_session_verify": (_I, [_P, _P, _P]),
    "gkrhip_to_regular": (_I, [_P, _SZ]),
    "gkrhip_from_regular": (_I, [_P, _SZ]),
    "gkrhip_mimc_permutation_batch": (_I, [_P, _P, _P, _SZ]),
    "gkrhip_comm_unique_id": (_I, [_P]),
    "gkrhip_comm_init": (_I, [_I, _I, _P]),
    "gkrhip_comm_init_shm": (_I, [_I, _I, C.c_char_p]),
    "gkrhip_comm_init_lanes": (_I, [_I, _I, _I, _P]),
    "gkrhip_comm_init_shm_lanes": (_I, [_I, _I, _I, C.c_char_p]),
    "gkrhip_comm_init_tick": (_I, [_I, _I, _I, _P]),
    "gkrhip_comm_init_tick_shm": (_I, [_I, _I, _I, C.c_char_p]),
    "gkrhip_comm_tick_stats": (_I, [C.POINTER(_U64), C.POINTER(_U64)]),
    "gkrhip_comm_destroy": (_I, []),
    "gkrhip_comm_info": (_I, [C.POINTER(_I), C.POINTER(_I)]),
    "gkrhip_host_shard_seed": (_I, [_P, _P, _I, _I]),
    "gkrhip_host_limbsplit_reduce": (_I, [_P, _P, _I]),
    "gkrhip_host_mimc_hash": (_I, [_P, _P, _SZ]),
    "gkrhip_host_cipher_round_coeffs": (_I, [_P, _P, _P, _P]),
    "gkrhip_bench_fold": (_I, [_SZ, _I, _I, _I, C.POINTER(_D), C.POINTER(_D)]),
    "gkrhip_bench_sumcheck": (_I, [_I, _I, _I, _I, _I, C.POINTER(_D), _P]),
    "gkrhip_bench_partial_eval": (_I, [_I, _I, _I, C.POINTER(_D), _P]),
    "gkrhip_compute_h": (_I, [_P, _P, _P, _P, _SZ, _SZ]),
    "gkrhip_bench_compute_h": (_I, [_I, _I, _I, C.POINTER(_D), C.POINTER(_I), C.POINTER(_D)]),
    "gkrhip_g1_bases_create": (_I, [C.POINTER(_P), _P, _SZ]),
    "gkrhip_g1_bases_generate": (_I, [C.POINTER(_P), _P, _P, _SZ, _I]),
    "gkrhip_g1_bases_len": (_SZ, [_P]),
    "gkrhip_g1_bases_read": (_I, [_P, _P, _SZ, _SZ]),
    "gkrhip_g1_bases_destroy": (None, [_P]),
    "gkrhip_msm_g1": (_I, [_P, _P, _P, _SZ, _I]),
    "gkrhip_msm_g1_once": (_I, [_P, _P, _P, _SZ, _I]),
    "gkrhip_msm_g1_set_window": (_I, [_P, _I]),
    "gkrhip_msm_g1_precompute": (_I, [_P, _I]),
    "gkrhip_msm_g2_precompute": (_I, [_P, _I]),
    "gkrhip_bench_msm_g1_fixed_base": (_I, [_I, _I, _I, _I, C.POINTER(_D), C.POINTER(_D), C.POINTER(_I), C.POINTER(_D), C.POINTER(_D), _P]),
    "gkrhip_bench_msm_g2_fixed_base": (_I, [_I, _I, _I, _I, C.POINTER(_D), C.POINTER(_D), C.POINTER(_I), C.POINTER(_D), C.POINTER(_D), _P]),
    "gkrhip_g1_batch_scalar_mul": (_I, [_P, _P, _P, _SZ, _I]),
    "gkrhip_bench_msm_g1": (_I, [_I, _I, _I, _I, C.POINTER(_D), C.POINTER(_D), C.POINTER(_I), C.POINTER(_D), _P]),
    "gkrhip_compute_h_msm_g1": (_I, [_P, _P, _P, _P, _P, _SZ, _SZ, _P]),
    "gkrhip_g2_bases_create": (_I, [C.POINTER(_P), _P, _SZ]),
    "gkrhip_g2_bases_generate": (_I, [C.POINTER(_P), _P, _P, _SZ, _I]),
    "gkrhip_g2_bases_len": (_SZ, [_P]),
    "gkrhip_g2_bases_read": (_I, [_P, _P, _SZ, _SZ]),
    "gkrhip_g2_bases_destroy": (None, [_P]),
    "gkrhip_msm_g2": (_I, [_P, _P, _P, _SZ, _I]),
    "gkrhip_msm_g1_g2": (_I, [_P, _P, _P, _P, _P, _SZ, _I]),
    "gkrhip_msm_shared": (_I, [_P, _P, _P, _SZ, _P, _SZ, _P, _SZ, _I]),
    "gkrhip_msm_g2_once": (_I, [_P, _P, _P, _SZ, _I]),
    "gkrhip_msm_g2_set_window": (_I, [_P, _I]),
    "gkrhip_g2_batch_scalar_mul": (_I, [_P, _P, _P, _SZ, _I]),
    "gkrhip_g2_generator": (_I, [_P]),
    "gkrhip_bench_msm_g2": (_I, [_I, _I, _I, _I, C.POINTER(_D), C.POINTER(_D), C.POINTER(_I), C.POINTER(_D), _P]),
    "gkrhip_profile_reset": (_I, [_SZ]),
    "gkrhip_profile_get": (_I, [C.POINTER(_U64), C.POINTER(_D), C.POINTER(_D), C.POINTER(_U64), C.POINTER(_D), C.POINTER(_D)]),
    "gkrhip_profile_host": (_I, [C.POINTER(_U64), C.POINTER(_D), C.POINTER(_D), C.POINTER(_D), C.POINTER(_D)]),
    "gkrhip_profile_latency": (_I, [C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64)]),
    "gkrhip_profile_counter": (_I, [C.c_char_p, C.POINTER(_U64)]),
    "gkrhip_host_sumcheck_closes": (_I, [_I, _P, _I, _I, _P, _I, _P, _I, _I, _P, _P, _P, C.POINTER(C.c_int)]),
    "gkrhip_host_ahead_contract": (_I, [_P, _P, _P, _I]),
    "gkrhip_host_group_selftest": (_I, [_I, _I, _I, _I, _P, _P]),
}


class GkrHipError(RuntimeError):
    pass


def library_path():
    return _SO


def load():
    """Load libgkrhip.so (no GPU needed for loading; compute calls need a gfx950 device)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise GkrHipError("libgkrhip.so not built: run `python __graft_entry__.py` (or gkr-mimc_amd/build.py); "
                          "there is no CPU fallback")
    from . import build as _build
    # the hash compiled into the binary itself (not a tracked side file) against the sources on disk
    if _build.binary_sha(_SO) != _build.source_sha():
        raise GkrHipError("libgkrhip.so was not built from the sources in this tree (the source hash compiled into it "
                          "differs): run `python __graft_entry__.py`; there is no CPU fallback")
    lib = C.CDLL(_SO)
    lib.gkrhip_build_id.restype = C.c_char_p
    if lib.gkrhip_build_id().decode() != _build.source_sha():
        raise GkrHipError("libgkrhip.so reports a different build id than its file holds")
    for name, (res, args) in ABI.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def error_message(rc):
    """The message of the failure that returned `rc` (thread-independent: gkrhip_last_error_r)."""
    buf = C.create_string_buffer(1024)
    load().gkrhip_last_error_r(int(rc), buf, len(buf))
    return buf.value.decode(errors="replace")


def _check(rc):
    if rc != 0:
        raise GkrHipError(error_message(rc) or "gkrhip error %d" % rc)


def init(device=0):
    _check(load().gkrhip_init(int(device)))


def shutdown():
    load().gkrhip_shutdown()


def device_count():
    return load().gkrhip_device_count()


def set_option(key, value):
    _check(load().gkrhip_set_option(key.encode(), int(value)))


def mem_info():
    f, t = C.c_size_t(0), C.c_size_t(0)
    _check(load().gkrhip_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


class PinnedArray:
    """A (rows, words) uint64 array in page-locked host memory (gkrhip_host_alloc): `.a` is the numpy view; uploads from it are
    plain DMA transfers.  Keep the object alive while the view is in use; close() (or the context manager) frees the memory."""

    def __init__(self, rows, words=4):
        self._p = C.c_void_p()
        _check(load().gkrhip_host_alloc(C.byref(self._p), int(rows) * int(words) * 8))
        buf = (C.c_uint64 * (max(1, int(rows) * int(words)))).from_address(self._p.value)
        self.a = np.frombuffer(buf, dtype=np.uint64)[: int(rows) * int(words)].reshape(int(rows), int(words))

    def close(self):
        if self._p:
            self.a = None
            load().gkrhip_host_free(self._p)
            self._p = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass


def synchronize():
    _check(load().gkrhip_device_synchronize())


def reserve_lanes(n):
    """Create n lanes ahead of the first burst of concurrent calls (a lane takes ~3 ms to create)."""
    _check(load().gkrhip_reserve_lanes(int(n)))


def _fr(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data


def fold(table, r):
    """(*MultiLin).Fold(r): returns the folded table (len n/2); the input array is not modified."""
    t = np.array(_fr(table), copy=True)
    _check(load().gkrhip_fold(_ptr(t), t.shape[0], _ptr(_fr(r))))
    return t[: t.shape[0] // 2].copy()


def evaluate(table, coords):
    table, coords = _fr(table), _fr(coords).reshape(-1, 4)
    out = np.zeros((1, 4), np.uint64)
    _check(load().gkrhip_evaluate(_ptr(out), _ptr(table), table.shape[0], _ptr(coords) if coords.shape[0] else None,
                                  coords.shape[0]))
    return out


def folded_eq_table(q, mult=None):
    q = _fr(q).reshape(-1, 4)
    bN = q.shape[0]
    out = np.zeros((1 << bN, 4), np.uint64)
    m = None if mult is None else _fr(mult)
    _check(load().gkrhip_eq_table(_ptr(out), _ptr(q) if bN else None, bN, _ptr(m)))
    return out


def chunk_of_eq_table(table, chunk_id, chunk_size, q, mult=None):
    """poly.ChunkOfEqTable: fills chunk `chunk_id` of `table` (2^len(q) elements) in place."""
    q = _fr(q).reshape(-1, 4)
    assert table.dtype == np.uint64 and table.flags.c_contiguous and table.shape == (1 << q.shape[0], 4)
    m = None if mult is None else _fr(mult)
    _check(load().gkrhip_chunk_of_eq_table(_ptr(table), chunk_id, chunk_size, _ptr(q) if q.shape[0] else None, q.shape[0], _ptr(m)))


def gate_eval_batch(gate, ark, xs):
    xs = [_fr(x) for x in xs]
    n = xs[0].shape[0]
    res = np.zeros((n, 4), np.uint64)
    arr = (C.c_void_p * len(xs))(*[x.ctypes.data for x in xs])
    a = None if ark is None else _fr(ark)
    _check(load().gkrhip_gate_eval_batch(gate, _ptr(a), _ptr(res), arr, len(xs), n))
    return res


def gate_degree(gate):
    if gate in (GATE_IDENTITY, GATE_ADD):
        return 1
    return 7 if gate == GATE_CIPHER else gate_lookup(gate)["power"]


def sumcheck_prove(X, q_primes, claims, gate, ark=None):
    """sumcheck.Prove(X, qPrimes, claims, gate) -> (proof[bN][deg+2], challenges[bN], finalClaims)."""
    X = [_fr(x) for x in X]
    q_primes = _fr(q_primes)
    assert q_primes.ndim == 3
    nq, bN = q_primes.shape[0], q_primes.shape[1]
    claims = _fr(claims).reshape(-1, 4)
    nc = gate_degree(gate) + 2
    proof = np.zeros((max(bN * nc, 1), 4), np.uint64)
    chal = np.zeros((max(bN, 1), 4), np.uint64)
    final = np.zeros((len(X) + 1, 4), np.uint64)
    for i, x in enumerate(X):
        if x.shape[0] != 1 << bN:  # sumcheck/prover.go:52-56
            raise GkrHipError("inconsistent sizes : bn is %d but table %d has size %d" % (bN, i, x.shape[0]))
    arr = (C.c_void_p * len(X))(*[x.ctypes.data for x in X])
    a = None if ark is None else _fr(ark)
    _check(load().gkrhip_sumcheck_prove(gate, _ptr(a), len(X), bN, arr, _ptr(q_primes) if bN else None, nq,
                                        _ptr(claims) if claims.shape[0] else None, claims.shape[0],
                                        _ptr(proof), _ptr(chal), _ptr(final)))
    return proof[: bN * nc].reshape(bN, nc, 4), chal[:bN], final


def sumcheck_verify(claims, proof):
    """sumcheck.Verify(claims, proof) -> (challenges[bN], finalClaim, recombChal); raises GkrHipError with the reference's
    message when a round's check fails (the reference returns err).  Host-only: needs no GPU."""
    claims = _fr(claims).reshape(-1, 4)
    proof = _fr(proof)
    assert proof.ndim == 3
    bN, nc = proof.shape[0], proof.shape[1]
    chal = np.zeros((max(bN, 1), 4), np.uint64)
    fin, rec = np.zeros((1, 4), np.uint64), np.zeros((1, 4), np.uint64)
    flat = np.ascontiguousarray(proof.reshape(-1, 4)) if bN else np.zeros((1, 4), np.uint64)
    rc = load().gkrhip_sumcheck_verify(_ptr(claims) if claims.shape[0] else None, claims.shape[0], _ptr(flat), bN, max(nc, 1),
                                       _ptr(chal), _ptr(fin), _ptr(rec))
    if rc > 0:
        raise GkrHipError(load().gkrhip_last_error().decode())
    _check(rc)
    return chal[:bN], fin, rec


def mimc_proof_len(bN):
    return load().gkrhip_mimc_proof_len(bN)


def gkr_prove_mimc(in0, in1, q_prime, want_outputs=True, regular=False, out=None):
    """Circuit.Assign(in0, in1) + gkr.Prove(MimcCircuit, a, qPrime); returns (flat proof, outputs).  regular=True: every
    buffer holds regular-form values (the hint interface's big.Int words) instead of Montgomery fr.Elements.  With a
    communicator installed in0/in1/outputs are this rank's shard and bN (= len(q_prime)) is the global size.
    out=(flat, outputs): caller-allocated result arrays, as the hint's pre-allocated `oups` (prover/gadget/hints.go:197)."""
    in0, in1 = _fr(in0), _fr(in1)
    n = in0.shape[0]
    q_prime = _fr(q_prime).reshape(-1, 4)
    bN = q_prime.shape[0] if q_prime.shape[0] else n.bit_length() - 1
    assert n & (n - 1) == 0 and n <= 1 << bN and in1.shape[0] == n
    if out is not None:
        flat, outs = out
        assert flat.shape == (mimc_proof_len(bN), 4) and flat.dtype == np.uint64 and flat.flags.c_contiguous
        assert outs is None or (outs.shape == (n, 4) and outs.dtype == np.uint64 and outs.flags.c_contiguous)
    else:
        flat = np.zeros((mimc_proof_len(bN), 4), np.uint64)
        outs = np.zeros((n, 4), np.uint64) if want_outputs else None
    fn = load().gkrhip_gkr_prove_mimc_regular if regular else load().gkrhip_gkr_prove_mimc
    _check(fn(bN, _ptr(in0), _ptr(in1), _ptr(q_prime) if bN else None, _ptr(flat), _ptr(outs)))
    return flat, outs


def gkr_verify_mimc(flat, in0, in1, outputs, q_prime):
    """gkr.Verify(MimcCircuit, proof, inputs, outputs, qPrime): True if accepted, False if rejected."""
    in0, in1, outputs, flat = _fr(in0), _fr(in1), _fr(outputs), _fr(flat)
    bN = in0.shape[0].bit_length() - 1
    q_prime = _fr(q_prime).reshape(-1, 4)
    rc = load().gkrhip_gkr_verify_mimc(bN, _ptr(flat), _ptr(in0), _ptr(in1), _ptr(outputs), _ptr(q_prime) if bN else None)
    if rc < 0:
        _check(rc)
    return rc == 0


def to_regular(arr):
    """Montgomery limbs -> regular value limbs (fr.Element.ToBigIntRegular for a whole slice)."""
    a = np.array(_fr(arr), copy=True)
    _check(load().gkrhip_to_regular(_ptr(a), a.reshape(-1, 4).shape[0]))
    return a


def from_regular(arr):
    a = np.array(_fr(arr), copy=True)
    _check(load().gkrhip_from_regular(_ptr(a), a.reshape(-1, 4).shape[0]))
    return a


def mimc_permutation_batch(x, key):
    """hash.MimcKeyedPermutation(x[i], key[i]) for every i (HashHint.Call for a batch)."""
    x, key = _fr(x), _fr(key)
    out = np.zeros_like(x)
    _check(load().gkrhip_mimc_permutation_batch(_ptr(out), _ptr(x), _ptr(key), x.shape[0]))
    return out


MAX_GATE_INPUTS = 4


class LayerDesc(C.Structure):
    _fields_ = [("gate", C.c_int), ("n_in", C.c_int), ("in_", C.c_int * MAX_GATE_INPUTS), ("ark", C.c_uint64 * 4)]


class GateDesc(C.Structure):
    _fields_ = [("id", C.c_char * 32), ("n_in", C.c_int), ("sum_mask", C.c_uint), ("power", C.c_int)]


def gate_register(gate_id, n_in, sum_mask, power):
    """Add a gate of the family (sum of the inputs selected by sum_mask + Ark)^power, power 1 or 7, to the library's
    gate table (the native circuit.Gate plug point); returns its gate id."""
    d = GateDesc(gate_id.encode(), n_in, sum_mask, power)
    out = C.c_int(-1)
    _check(load().gkrhip_gate_register(C.byref(d), C.byref(out)))
    return out.value


def gate_lookup(gate):
    d = GateDesc()
    _check(load().gkrhip_gate_lookup(gate, C.byref(d)))
    return {"id": d.id.decode(), "n_in": d.n_in, "sum_mask": d.sum_mask, "power": d.power}


def _layers_to_list(arr):
    return [(l.gate, [l.in_[k] for k in range(l.n_in)], [int(v) for v in l.ark]) for l in arr]


def _layers_from_list(layers):
    arr = (LayerDesc * len(layers))()
    for i, (gate, ins, ark) in enumerate(layers):
        arr[i].gate, arr[i].n_in = gate, len(ins)
        for k, v in enumerate(ins):
            arr[i].in_[k] = v
        for k in range(4):
            arr[i].ark[k] = 0 if ark is None else int(ark[k])
    return arr


def gmimc_t2_circuit():
    """The library's GMiMC (t = 2) circuit description as a list of (gate, [inputs], ark limbs)."""
    n = load().gkrhip_gmimc_t2_circuit(None, 0)
    arr = (LayerDesc * n)()
    assert load().gkrhip_gmimc_t2_circuit(C.cast(arr, C.c_void_p), n) == n
    return _layers_to_list(arr)


def gmimc_circuit(t):
    """GMiMC compression circuit for t = 2, 4, 8: (layers, input_map); input layer k is state[j] when
    input_map[k] = j < t and block[j - t] otherwise."""
    n = load().gkrhip_gmimc_circuit(t, None, 0, None)
    if n < 0:
        _check(n)
    arr = (LayerDesc * n)()
    imap = (C.c_int * (2 * t))(*([-1] * (2 * t)))
    assert load().gkrhip_gmimc_circuit(t, C.cast(arr, C.c_void_p), n, C.cast(imap, C.c_void_p)) == n
    layers = _layers_to_list(arr)
    n_in = sum(1 for l in layers if l[0] < 0)
    return layers, [imap[k] for k in range(n_in)]


def gmimc_hash_circuit(t, nblocks):
    """The sponge GMimcT{t}.Hash over nblocks * t message elements as a circuit: (layers, input_map); input layer k is
    msg[input_map[k]]."""
    n = load().gkrhip_gmimc_hash_circuit(t, nblocks, None, 0, None)
    if n < 0:
        _check(n)
    arr = (LayerDesc * n)()
    imap = (C.c_int * (t * nblocks))(*([-1] * (t * nblocks)))
    assert load().gkrhip_gmimc_hash_circuit(t, nblocks, C.cast(arr, C.c_void_p), n, C.cast(imap, C.c_void_p)) == n
    layers = _layers_to_list(arr)
    n_in = sum(1 for l in layers if l[0] < 0)
    return layers, [imap[k] for k in range(n_in)]


def circuit_proof_len(layers, bN):
    """GkrProverHint.NbOutputs for a circuit given as a layer list (prover/gadget/hints.go:76-116)."""
    outs = [0] * len(layers)
    for gate, ins, _ark in layers:
        for p in ins:
            outs[p] += 1
    sc = sum(bN * (gate_degree(g) + 2) for g, _i, _a in layers if g >= 0)
    return sc + sum(outs) + bN * sum(outs) + bN


def gkr_prove(layers, inputs, q_prime, want_outputs=True):
    """Circuit.Assign + gkr.Prove for any circuit of library gates on host tables: (flat proof, outputs)."""
    arr = _layers_from_list(layers)
    inputs = [_fr(x) for x in inputs]
    n = inputs[0].shape[0]
    bN = n.bit_length() - 1
    q_prime = _fr(q_prime).reshape(-1, 4)
    flat = np.zeros((circuit_proof_len(layers, bN), 4), np.uint64)
    outs = np.zeros((n, 4), np.uint64) if want_outputs else None
    ptrs = (C.c_void_p * len(inputs))(*[x.ctypes.data for x in inputs])
    _check(load().gkrhip_gkr_prove(C.cast(arr, C.c_void_p), len(layers), bN, ptrs, len(inputs), _ptr(q_prime) if bN else None,
                                   _ptr(flat), _ptr(outs)))
    return flat, outs


def gkr_verify(layers, flat, inputs, outputs, q_prime):
    """gkr.Verify for any circuit of library gates on host tables: True if accepted, False if rejected."""
    arr = _layers_from_list(layers)
    inputs = [_fr(x) for x in inputs]
    outputs, flat = _fr(outputs), _fr(flat)
    bN = outputs.shape[0].bit_length() - 1
    q_prime = _fr(q_prime).reshape(-1, 4)
    ptrs = (C.c_void_p * len(inputs))(*[x.ctypes.data for x in inputs])
    rc = load().gkrhip_gkr_verify(C.cast(arr, C.c_void_p), len(layers), bN, _ptr(flat), ptrs, len(inputs), _ptr(outputs),
                                  _ptr(q_prime) if bN else None)
    if rc < 0:
        _check(rc)
    return rc == 0


class MimcSession:
    """Resident assignment on the GPU (examples.MimcCircuit, or any circuit given as `layers` =
    [(gate, [inputs], ark limbs or None), ...]); prove() can be repeated."""

    def __init__(self, bN, layers=None):
        self.bN = bN
        self._h = C.c_void_p()
        w, r = C.c_int(1), C.c_int(0)
        load().gkrhip_comm_info(C.byref(w), C.byref(r))
        self.world, self.rank = w.value, r.value
        self.local_n = (1 << bN) // self.world
        if layers is None:
            _check(load().gkrhip_mimc_session_create(C.byref(self._h), bN))
        else:
            arr = _layers_from_list(layers)
            _check(load().gkrhip_session_create(C.byref(self._h), C.cast(arr, C.c_void_p), len(layers), bN))
        self.proof_len = load().gkrhip_session_proof_len(self._h)
        self.num_inputs = load().gkrhip_session_num_inputs(self._h)

    def load_input(self, index, table):
        table = _fr(table)
        assert table.shape[0] == self.local_n
        _check(load().gkrhip_session_load_input(self._h, index, _ptr(table)))

    def load_inputs(self, in0, in1):
        in0, in1 = _fr(in0), _fr(in1)
        assert in0.shape[0] == self.local_n and in1.shape[0] == self.local_n
        _check(load().gkrhip_mimc_session_load_inputs(self._h, _ptr(in0), _ptr(in1)))

    def synth_inputs(self, stride=None, offset=None):
        """RandomFrArray inputs generated on the device; defaults to this rank's shard."""
        stride = self.world if stride is None else stride
        offset = self.rank if offset is None else offset
        _check(load().gkrhip_mimc_session_synth_inputs(self._h, stride, offset))

    def assign(self):
        _check(load().gkrhip_mimc_session_assign(self._h))

    def prove(self, q_prime):
        q_prime = _fr(q_prime).reshape(-1, 4)
        flat = np.zeros((self.proof_len, 4), np.uint64)
        _check(load().gkrhip_mimc_session_prove(self._h, _ptr(q_prime) if self.bN else None, _ptr(flat)))
        return flat

    @staticmethod
    def prove_group(sessions, q_primes):
        """gkr.Prove for up to eight sessions of the same shape from the calling thread, in lock-step: proof i is bit for bit
        sessions[i].prove(q_primes[i]), the round kernels of the proofs go to the GPU as one launch (gkrhip_mimc_session_prove_group)."""
        n = len(sessions)
        assert n == len(q_primes) and n >= 1
        qs = [_fr(q).reshape(-1, 4) for q in q_primes]
        flats = [np.zeros((s.proof_len, 4), np.uint64) for s in sessions]
        hs = (C.c_void_p * n)(*[s._h for s in sessions])
        qp = (C.c_void_p * n)(*[_ptr(q) if s.bN else None for s, q in zip(sessions, qs)])
        fp = (C.c_void_p * n)(*[_ptr(f) for f in flats])
        rcs = (C.c_int * n)()
        _check(load().gkrhip_mimc_session_prove_group(n, hs, qp, fp, rcs))
        return flats

    def outputs(self):
        out = np.zeros((self.local_n, 4), np.uint64)
        _check(load().gkrhip_mimc_session_outputs(self._h, _ptr(out)))
        return out

    def verify(self, q_prime, flat):
        """gkr.Verify against the resident inputs/outputs: True if accepted."""
        q_prime = _fr(q_prime).reshape(-1, 4)
        rc = load().gkrhip_mimc_session_verify(self._h, _ptr(q_prime) if self.bN else None, _ptr(_fr(flat)))
        if rc < 0:
            _check(rc)
        return rc == 0

    def evaluate_layer(self, layer, coords):
        coords = _fr(coords).reshape(-1, 4)
        out = np.zeros((1, 4), np.uint64)
        _check(load().gkrhip_mimc_session_evaluate_layer(self._h, layer, _ptr(coords) if self.bN else None, _ptr(out)))
        return out

    def close(self):
        if self._h:
            load().gkrhip_mimc_session_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    """128-byte RCCL unique id (call on rank 0, broadcast to the other ranks)."""
    buf = np.zeros(128, np.uint8)
    _check(load().gkrhip_comm_unique_id(_ptr(buf)))
    return buf


def comm_init(world, rank, unique_id):
    uid = None if unique_id is None else np.ascontiguousarray(unique_id, dtype=np.uint8)
    _check(load().gkrhip_comm_init(world, rank, _ptr(uid)))


def comm_init_shm(world, rank, name):
    _check(load().gkrhip_comm_init_shm(world, rank, name.encode()))


def comm_init_lanes(world, rank, unique_ids):
    """unique_ids: (nlanes, 128) uint8 -- one RCCL unique id per lane."""
    ids = np.ascontiguousarray(unique_ids, dtype=np.uint8).reshape(-1, 128)
    _check(load().gkrhip_comm_init_lanes(world, rank, ids.shape[0], _ptr(ids)))


def comm_init_shm_lanes(world, rank, nlanes, name):
    _check(load().gkrhip_comm_init_shm_lanes(world, rank, nlanes, name.encode()))


def comm_init_tick(world, rank, nlanes, unique_id):
    """nlanes lanes over ONE RCCL communicator (the ticker): deterministic order of collectives on every rank."""
    uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
    _check(load().gkrhip_comm_init_tick(world, rank, nlanes, _ptr(uid)))


def comm_init_tick_shm(world, rank, nlanes, name):
    """The ticker with its tick all-reduce over host shared memory (several ranks on one GPU: tests)."""
    _check(load().gkrhip_comm_init_tick_shm(world, rank, nlanes, name.encode()))


def comm_tick_stats():
    a, b = C.c_uint64(0), C.c_uint64(0)
    _check(load().gkrhip_comm_tick_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


def comm_destroy():
    _check(load().gkrhip_comm_destroy())


def host_shard_seed(q_tail, rank):
    q_tail = _fr(q_tail).reshape(-1, 4)
    out = np.zeros((1, 4), np.uint64)
    _check(load().gkrhip_host_shard_seed(_ptr(out), _ptr(q_tail) if q_tail.shape[0] else None, q_tail.shape[0], rank))
    return out


def host_limbsplit_reduce(lanes):
    lanes = np.ascontiguousarray(lanes, dtype=np.uint64)
    out = np.zeros((1, 4), np.uint64)
    _check(load().gkrhip_host_limbsplit_reduce(_ptr(out), _ptr(lanes), lanes.shape[0]))
    return out


def host_mimc_hash(arr):
    arr = _fr(arr).reshape(-1, 4)
    out = np.zeros((1, 4), np.uint64)
    _check(load().gkrhip_host_mimc_hash(_ptr(out), _ptr(arr), arr.shape[0]))
    return out


def host_cipher_round_coeffs(M, c, qk):
    out = np.zeros((9, 4), np.uint64)
    _check(load().gkrhip_host_cipher_round_coeffs(_ptr(out), _ptr(_fr(M)), _ptr(_fr(c)), _ptr(_fr(qk))))
    return out


def host_sumcheck_closes(gate, ark, arity, q_primes, claims, proof, challenges, final_claims, claims_are_sums=True):
    """The prover's own check of a finished sumcheck on host data (no GPU): 0 closes, 1 + i round i, -1 closing identity,
    -2 finalClaims[0]."""
    q_primes = _fr(q_primes)
    nq, bN = q_primes.shape[0], q_primes.shape[1]
    claims = _fr(claims).reshape(-1, 4)
    v = C.c_int(0)
    _check(load().gkrhip_host_sumcheck_closes(gate, _ptr(_fr(ark)) if ark is not None else None, arity, bN, _ptr(q_primes) if bN else None, nq,
                                              _ptr(claims) if claims.shape[0] else None, claims.shape[0], 1 if claims_are_sums else 0,
                                              _ptr(_fr(proof)), _ptr(_fr(challenges)) if bN else None, _ptr(_fr(final_claims)), C.byref(v)))
    return v.value


def host_ahead_contract(class_sums, q_low):
    q_low = _fr(q_low).reshape(-1, 4)
    out = np.zeros((7, 4), np.uint64)
    _check(load().gkrhip_host_ahead_contract(_ptr(out), _ptr(_fr(class_sums)), _ptr(q_low) if q_low.shape[0] else None, q_low.shape[0]))
    return out


def bench_fold(n, ntab=1, warmup=3, iters=20, isolated=False):
    """ms per fold of `ntab` tables of n elements: back-to-back launches, or (isolated=True) the pair
    (back-to-back, one launch at a time on an idle GPU)."""
    ms, iso = C.c_double(0), C.c_double(0)
    _check(load().gkrhip_bench_fold(n, ntab, warmup, iters, C.byref(ms), C.byref(iso) if isolated else None))
    return (ms.value, iso.value) if isolated else ms.value


def bench_sumcheck(kind, bn, ninstance=1, warmup=1, iters=3):
    """sumcheck.Prove micro-benchmark (kind 0: BenchmarkWithCipherGate, 1: BenchmarkMultiIdentity): (ms per Prove, finalClaims[0])."""
    ms = C.c_double(0)
    fin = np.zeros((1, 4), np.uint64)
    _check(load().gkrhip_bench_sumcheck(kind, bn, ninstance, warmup, iters, C.byref(ms), _ptr(fin)))
    return ms.value, fin


def compute_h(a, b, c, cardinality=0):
    """computeH(a, b, c, domain) of prover/gadget/prove.go:308-359: Montgomery (n, 4) arrays in, the domain's `cardinality`
    REGULAR-form values out (the reference's order: coefficients of H at bit-reversed positions)."""
    a, b, c = _fr(a), _fr(b), _fr(c)
    n = a.shape[0]
    assert b.shape[0] == n and c.shape[0] == n and n >= 1
    card = cardinality or (1 << (n - 1).bit_length())
    h = np.zeros((card, 4), np.uint64)
    _check(load().gkrhip_compute_h(_ptr(h), _ptr(a), _ptr(b), _ptr(c), n, cardinality))
    return h


def bench_compute_h(logn, warmup=1, iters=3):
    """(ms per computeH on device-resident vectors, passes over HBM, HBM bytes moved by those passes)."""
    ms, np_, by = C.c_double(0), C.c_int(0), C.c_double(0)
    _check(load().gkrhip_bench_compute_h(logn, warmup, iters, C.byref(ms), C.byref(np_), C.byref(by)))
    return ms.value, np_.value, by.value


# ---- G1 multi-scalar multiplication (gnark-crypto's MultiExp as called at prover/gadget/prove.go:76,91,189,202,221) ----
MSM_SCALARS_MONT = 1


def _pts(a, words):
    """(n, words) uint64: the memory image of a Go []bn254.G1Affine (8 words per point) or []bn254.G2Affine (16)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(1, words)
    assert a.ndim == 2 and a.shape[1] == words
    return a


class _Bases:
    """Bases of a multi-scalar multiplication resident in HBM (a proving-key vector such as pk.G1.A, fixed across proofs)."""
    GROUP, WORDS = "g1", 8

    def __init__(self, points=None, base=None, scalars=None, scalars_mont=False):
        self._h = _P()
        lib = load()
        if points is not None:
            points = _pts(points, self.WORDS)
            _check(self._f("%s_bases_create")(C.byref(self._h), _ptr(points), points.shape[0]))
        else:       # bases[i] = [scalars[i]] base, generated on the device
            base, scalars = _pts(base, self.WORDS), _fr(scalars)
            _check(self._f("%s_bases_generate")(C.byref(self._h), _ptr(base), _ptr(scalars), scalars.shape[0],
                                                MSM_SCALARS_MONT if scalars_mont else 0))

    def _f(self, pattern):
        return getattr(load(), "gkrhip_" + pattern % self.GROUP)

    def __len__(self):
        return int(self._f("%s_bases_len")(self._h))

    def read(self, first=0, count=None):
        count = len(self) - first if count is None else count
        out = np.zeros((count, self.WORDS), dtype=np.uint64)
        _check(self._f("%s_bases_read")(self._h, _ptr(out), first, count))
        return out

    def set_window(self, c):
        _check(self._f("msm_%s_set_window")(self._h, int(c)))

    def precompute(self, c=0):
        """Fixed-base tables [2^(c j)] P_i (once per key); the handle's multi_exp then sorts all windows into one bucket space.
        c = 0: by the number of points, 8..22 forced, -1 drops the tables."""
        _check(self._f("msm_%s_precompute")(self._h, int(c)))

    def multi_exp(self, scalars, scalars_mont=False):
        """sum_i [scalars[i]] bases[i] as an affine image: (*G1Affine).MultiExp / (*G2Affine).MultiExp(points, scalars, config)."""
        scalars = _fr(scalars) if len(scalars) else np.zeros((0, 4), dtype=np.uint64)
        out = np.zeros(self.WORDS, dtype=np.uint64)
        _check(self._f("msm_%s")(_ptr(out), self._h, _ptr(scalars) if scalars.shape[0] else None, scalars.shape[0],
                                 MSM_SCALARS_MONT if scalars_mont else 0))
        return out

    def close(self):
        if self._h:
            self._f("%s_bases_destroy")(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class G1Bases(_Bases):
    GROUP, WORDS = "g1", 8

    def compute_h_multi_exp(self, a, b, c, cardinality=0, want_h=False):
        """h = computeH(a, b, c, domain); krs2.MultiExp(self, h) (prove.go:128,221) with H never leaving the device.
        Returns the G1Affine image (and H, regular form, when want_h)."""
        a, b, c = _fr(a), _fr(b), _fr(c)
        n = a.shape[0]
        assert b.shape[0] == n and c.shape[0] == n and n >= 1
        card = cardinality or max(2, 1 << (n - 1).bit_length())
        out = np.zeros(8, dtype=np.uint64)
        h = np.zeros((card, 4), np.uint64) if want_h else None
        _check(load().gkrhip_compute_h_msm_g1(_ptr(out), self._h, _ptr(a), _ptr(b), _ptr(c), n, card, _ptr(h) if want_h else None))
        return (out, h) if want_h else out


class G2Bases(_Bases):
    GROUP, WORDS = "g2", 16


def _multi_exp(group, words, points, scalars, scalars_mont):
    points = _pts(points, words) if len(points) else np.zeros((0, words), dtype=np.uint64)
    scalars = _fr(scalars) if len(scalars) else np.zeros((0, 4), dtype=np.uint64)
    assert points.shape[0] == scalars.shape[0]
    out = np.zeros(words, dtype=np.uint64)
    n = points.shape[0]
    _check(getattr(load(), "gkrhip_msm_%s_once" % group)(_ptr(out), _ptr(points) if n else None, _ptr(scalars) if n else None, n,
                                                          MSM_SCALARS_MONT if scalars_mont else 0))
    return out


def multi_exp_g1_g2(g1_bases, g2_bases, scalars, scalars_mont=False):
    """bs1.MultiExp(pk.G1.B, wireValuesB) and Bs.MultiExp(pk.G2.B, wireValuesB) (prove.go:189,277) in one call: the two sums share
    the upload, the decoding and the sort of the scalars.  Returns (G1 affine image, G2 affine image)."""
    scalars = _fr(scalars) if len(scalars) else np.zeros((0, 4), dtype=np.uint64)
    o1, o2 = np.zeros(8, dtype=np.uint64), np.zeros(16, dtype=np.uint64)
    _check(load().gkrhip_msm_g1_g2(_ptr(o1), _ptr(o2), g1_bases._h, g2_bases._h, _ptr(scalars) if scalars.shape[0] else None,
                                   scalars.shape[0], MSM_SCALARS_MONT if scalars_mont else 0))
    return o1, o2


def multi_exp_shared(g1_bases, g2_bases, scalars, scalars_mont=False):
    """Several MSMs over one scalar vector (gkrhip_msm_shared): one upload, one sort.  g1_bases / g2_bases: lists of handles of
    equal length.  Returns (list of G1 affine images, list of G2 affine images)."""
    scalars = _fr(scalars) if len(scalars) else np.zeros((0, 4), dtype=np.uint64)
    k1, k2 = len(g1_bases), len(g2_bases)
    o1, o2 = np.zeros((max(k1, 1), 8), dtype=np.uint64), np.zeros((max(k2, 1), 16), dtype=np.uint64)
    h1 = (C.c_void_p * max(k1, 1))(*[b._h for b in g1_bases])
    h2 = (C.c_void_p * max(k2, 1))(*[b._h for b in g2_bases])
    _check(load().gkrhip_msm_shared(_ptr(o1), _ptr(o2), h1, k1, h2, k2, _ptr(scalars) if scalars.shape[0] else None, scalars.shape[0],
                                    MSM_SCALARS_MONT if scalars_mont else 0))
    return [o1[i] for i in range(k1)], [o2[i] for i in range(k2)]


def multi_exp_g1(points, scalars, scalars_mont=False):
    """(*G1Affine).MultiExp(points, scalars, config) in one call on host buffers."""
    return _multi_exp("g1", 8, points, scalars, scalars_mont)


def multi_exp_g2(points, scalars, scalars_mont=False):
    """(*G2Affine).MultiExp(points, scalars, config) (prove.go:277) in one call on host buffers."""
    return _multi_exp("g2", 16, points, scalars, scalars_mont)


def _batch_mul(group, words, base, scalars, scalars_mont):
    base, scalars = _pts(base, words), _fr(scalars)
    out = np.zeros((scalars.shape[0], words), dtype=np.uint64)
    _check(getattr(load(), "gkrhip_%s_batch_scalar_mul" % group)(_ptr(out), _ptr(base), _ptr(scalars), scalars.shape[0],
                                                                 MSM_SCALARS_MONT if scalars_mont else 0))
    return out


def batch_scalar_multiplication_g1(base, scalars, scalars_mont=False):
    """bn254.BatchScalarMultiplicationG1(base, scalars) (prove.go:177): (n, 8) G1Affine images."""
    return _batch_mul("g1", 8, base, scalars, scalars_mont)


def batch_scalar_multiplication_g2(base, scalars, scalars_mont=False):
    """bn254.BatchScalarMultiplicationG2(base, scalars): (n, 16) G2Affine images."""
    return _batch_mul("g2", 16, base, scalars, scalars_mont)


def g2_generator():
    out = np.zeros(16, dtype=np.uint64)
    _check(load().gkrhip_g2_generator(_ptr(out)))
    return out


def _bench_msm(group, words, logn, c, warmup, iters):
    ms, tail, cu = C.c_double(0), C.c_double(0), C.c_int(0)
    ph = (C.c_double * 5)()
    res = np.zeros(words, dtype=np.uint64)
    _check(getattr(load(), "gkrhip_bench_msm_%s" % group)(logn, c, warmup, iters, C.byref(ms), ph, C.byref(cu), C.byref(tail), _ptr(res)))
    return {"ms": ms.value, "c": cu.value, "host_tail_ms": tail.value, "result": res,
            "phases_ms": dict(zip(("sort", "accumulate", "big_buckets", "reduce", "copy"), list(ph)))}


def _bench_msm_fixed_base(group, words, logn, c, warmup, iters):
    ms, tail, pre, cu = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
    ph = (C.c_double * 5)()
    res = np.zeros(words, dtype=np.uint64)
    _check(getattr(load(), "gkrhip_bench_msm_%s_fixed_base" % group)(logn, c, warmup, iters, C.byref(ms), ph, C.byref(cu), C.byref(tail),
                                                                       C.byref(pre), _ptr(res)))
    return {"ms": ms.value, "c": cu.value, "host_tail_ms": tail.value, "precompute_ms": pre.value, "result": res,
            "phases_ms": dict(zip(("sort", "accumulate", "big_buckets", "reduce", "copy"), list(ph)))}


def bench_msm_g1_fixed_base(logn, c=0, warmup=1, iters=3):
    """The same MSM on fixed-base tables (gkrhip_msm_g1_precompute): + precompute_ms, the tables' one-time cost."""
    return _bench_msm_fixed_base("g1", 8, logn, c, warmup, iters)


def bench_msm_g2_fixed_base(logn, c=0, warmup=1, iters=3):
    return _bench_msm_fixed_base("g2", 16, logn, c, warmup, iters)


def bench_msm_g1(logn, c=0, warmup=1, iters=3):
    """MSM of 2^logn synthetic device-resident bases and scalars: dict(ms, phases_ms, c, host_tail_ms, result)."""
    return _bench_msm("g1", 8, logn, c, warmup, iters)


def bench_msm_g2(logn, c=0, warmup=1, iters=3):
    return _bench_msm("g2", 16, logn, c, warmup, iters)


def bench_partial_eval(bn, warmup=10, iters=200):
    """BenchmarkPartialEvalWithCipher's shape: (microseconds per dispatchPartialEvals, evals[0] as a (1, 4) array)."""
    us = C.c_double(0)
    e0 = np.zeros((1, 4), dtype=np.uint64)
    _check(load().gkrhip_bench_partial_eval(bn, warmup, iters, C.byref(us), _ptr(e0)))
    return us.value, e0


def host_group_selftest(n, steps, diverge_at=-1, leave_after=-1):
    """gkrhip_host_group_selftest (no GPU): (launches asked for, launches made, most proofs in one launch, verdict)."""
    counts = (C.c_uint64 * 3)()
    verdict = C.c_int(-1)
    _check(load().gkrhip_host_group_selftest(n, steps, diverge_at, leave_after, counts, C.byref(verdict)))
    return int(counts[0]), int(counts[1]), int(counts[2]), verdict.value


def profile_counter(name):
    v = C.c_uint64(0)
    _check(load().gkrhip_profile_counter(name.encode(), C.byref(v)))
    return v.value


def profile_reset(min_n):
    _check(load().gkrhip_profile_reset(min_n))


def profile_get():
    fl, pl = C.c_uint64(0), C.c_uint64(0)
    fm, fb, pm, pmm = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
    _check(load().gkrhip_profile_get(C.byref(fl), C.byref(fm), C.byref(fb), C.byref(pl), C.byref(pm), C.byref(pmm)))
    r, hh, hw, hl, ho = C.c_uint64(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
    _check(load().gkrhip_profile_host(C.byref(r), C.byref(hh), C.byref(hw), C.byref(hl), C.byref(ho)))
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _check(load().gkrhip_profile_latency(C.byref(a), C.byref(b), C.byref(c)))
    sp = C.c_uint64(0)
    _check(load().gkrhip_profile_counter(b"spec_rounds", C.byref(sp)))
    rt = C.c_uint64(0)
    _check(load().gkrhip_profile_counter(b"chal_retries", C.byref(rt)))
    lc, lf = C.c_uint64(0), C.c_uint64(0)
    _check(load().gkrhip_profile_counter(b"layer_checks", C.byref(lc)))
    _check(load().gkrhip_profile_counter(b"layer_check_failures", C.byref(lf)))
    ah = C.c_uint64(0)
    _check(load().gkrhip_profile_counter(b"ahead_round0", C.byref(ah)))
    return {"ahead_round0": ah.value, "layer_checks": lc.value, "layer_check_failures": lf.value, "spec_rounds": sp.value, "chal_retries": rt.value, "fold_launches": fl.value, "fold_ms": fm.value, "fold_bytes": fb.value,
            "peval_launches": pl.value, "peval_ms": pm.value, "peval_modmuls": pmm.value,
            "rounds": r.value, "host_hash_ms": hh.value, "host_wait_ms": hw.value, "host_launch_ms": hl.value,
            "host_other_ms": ho.value, "prelaunched_rounds": a.value, "lookahead_round0": b.value, "coop_rounds": c.value}
