"""Worker of the CPU multi-process test (torch.distributed, gloo, world_size >= 2): the SHARDED sumcheck
protocol of DESIGN.md section 7 -- shard on the lowest index bits, per-round all-reduce of limb-split
lanes, gather + redundant tail rounds -- restated in Python with the oracle's arithmetic for the
per-shard table work and the PRODUCT's host-side scalar helpers (libgkrhip.so loads without a GPU) for
the shard weight, the lane reduction, the round coefficients and Fiat-Shamir.  Every rank checks its
transcript against the un-sharded oracle."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyoracle as o  # noqa: E402

Q = o.Q


def to_fr(vals):
    out = np.zeros((len(vals), 4), np.uint64)
    for i, v in enumerate(vals):
        for k, l in enumerate(o.to_mont_limbs(v % Q)):
            out[i, k] = l
    return out


def from_fr(arr):
    return [o.from_mont_limbs([int(x) for x in row]) for row in np.asarray(arr).reshape(-1, 4)]


def lanes_of(vals):
    """limb-split lanes (8 x 32-bit limbs of the Montgomery residue) of a list of field values -> int64 tensor"""
    t = torch.zeros(len(vals) * 8, dtype=torch.int64)
    for i, v in enumerate(vals):
        m = (v % Q) * o.R % Q
        for j in range(8):
            t[8 * i + j] = (m >> (32 * j)) & 0xFFFFFFFF
    return t


def allreduce_elements(gk, vals):
    t = lanes_of(vals)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)       # exact integer sum: valid for packed limbs split into lanes
    out = []
    for i in range(len(vals)):
        lanes = np.array([int(x) for x in t[8 * i:8 * i + 8]], dtype=np.uint64)
        out.append(from_fr(gk.host_limbsplit_reduce(lanes))[0])
    return out


def allgather_elements(vals, world):
    t = lanes_of(vals)
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    res = []
    for r in range(world):
        row = []
        for i in range(len(vals)):
            m = sum(int(outs[r][8 * i + j]) << (32 * j) for j in range(8))
            row.append(m * o.R_INV % Q)
        res.append(row)
    return res


def cipher_rounds(gk, K, S, q, ark, seed, collective, c, proof, chal):
    """rounds over local tables (python ints) and coordinates q; returns (c, K[0], S[0]) after the last fold"""
    for k in range(len(q)):
        mid = len(K) // 2
        W = o.folded_eq_table(q[k + 1:], seed)
        M = [0] * 8
        for x in range(mid):
            u = (K[x] + S[x] + ark) % Q
            d = ((K[x + mid] - K[x]) + (S[x + mid] - S[x])) % Q
            for j in range(8):
                M[j] = (M[j] + W[x] * pow(u, 7 - j, Q) * pow(d, j, Q)) % Q
        if collective:
            M = allreduce_elements(gk, M)
        co = from_fr(gk.host_cipher_round_coeffs(to_fr(M), to_fr([c]), to_fr([q[k]])))
        r = from_fr(gk.host_mimc_hash(to_fr(co)))[0]
        proof.append(co)
        chal.append(r)
        c = c * ((1 + 2 * q[k] * r - q[k] - r) % Q) % Q
        K, S = o.fold(K, r), o.fold(S, r)
    return c, K[0], S[0]


def linear_rounds(gk, gate, X, q, seed, collective, c, proof, chal):
    """Fused rounds of a single-point LINEAR gate (identity, sums + Ark): the device returns the two sums
    M_0 = sum W u, M_1 = sum W d of S_k(t) = M_0 + M_1 t (linear_round.hip.h); the ranks add them; the three round
    coefficients are c ((1-q_k) + (2 q_k - 1) t) (M_0 + M_1 t).  Returns (c, [X_t folded on every challenge])."""
    X = [list(x) for x in X]
    for k in range(len(q)):
        mid = len(X[0]) // 2
        W = o.folded_eq_table(q[k + 1:], seed)
        M = [0, 0]
        for x in range(mid):
            u = gate.eval(*[t[x] for t in X])
            d = (gate.eval(*[t[x + mid] for t in X]) - u) % Q            # the gate's linear part of (hi - lo)
            M[0] = (M[0] + W[x] * u) % Q
            M[1] = (M[1] + W[x] * d) % Q
        if collective:
            M = allreduce_elements(gk, M)
        a0, a1 = (1 - q[k]) % Q, (2 * q[k] - 1) % Q
        co = [c * a0 * M[0] % Q, c * (a0 * M[1] + a1 * M[0]) % Q, c * a1 * M[1] % Q]
        r = from_fr(gk.host_mimc_hash(to_fr(co)))[0]
        proof.append(co)
        chal.append(r)
        c = c * ((1 + 2 * q[k] * r - q[k] - r) % Q) % Q
        X = [o.fold(t, r) for t in X]
    return c, [t[0] for t in X]


def generic_rounds(gk, gate, eq, X, m, collective, proof, chal):
    for _k in range(m):
        ev = o.partial_evals(eq, X, gate)
        if collective:
            ev = allreduce_elements(gk, ev)
        co = o.interpolate_on_range(ev)
        r = from_fr(gk.host_mimc_hash(to_fr(co)))[0]
        proof.append(co)
        chal.append(r)
        eq = o.fold(eq, r)
        X = [o.fold(x, r) for x in X]
    return [eq[0]] + [x[0] for x in X]


def main():
    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    gamma = world.bit_length() - 1
    gk = importlib.import_module("gkr-mimc_amd")
    for bn in (gamma, gamma + 1, 5, 7):
        m1 = bn - gamma
        # ---- cipher gate, one point (the 91 MiMC layers)
        X, claims, qs, gate = o.initialize_cipher_gate_instance(bn)
        X[1] = [(7 * v * v + 3) % Q for v in X[1]]
        claims = [o.evaluation(gate, qs, [], *X)]
        q = qs[0]
        want = o.sumcheck_prove(X, qs, claims, gate)
        K, S = X[0][rank::world], X[1][rank::world]          # shard: indices = rank (mod world)
        seed = from_fr(gk.host_shard_seed(to_fr(q[m1:]), rank))[0]
        assert seed == o.folded_eq_table(q[m1:])[rank]
        proof, chal = [], []
        c, kv, sv = cipher_rounds(gk, K, S, q[:m1], gate.ark, seed, True, 1, proof, chal)
        rows = allgather_elements([kv, sv], world)
        K2, S2 = [r[0] for r in rows], [r[1] for r in rows]
        c, kv, sv = cipher_rounds(gk, K2, S2, q[m1:], gate.ark, 1, False, c, proof, chal)
        assert proof == want[0] and chal == want[1] and [c, kv, sv] == want[2], ("cipher", bn, rank)
        # ---- a linear gate of three inputs, one point (sharded fused linear rounds: the add / sum layers of GMiMC)
        gate = o.SumGate(o.ARKS[3], 1)
        n = 1 << bn
        X = [[(i * i + 7 * t + 1) % Q for i in range(n)] for t in range(3)]
        q = o.random_fr_array(bn)
        claims = [o.evaluation(gate, [q], [], *X)]
        want = o.sumcheck_prove([list(x) for x in X], [q], claims, gate)
        seed = from_fr(gk.host_shard_seed(to_fr(q[m1:]), rank))[0]
        proof, chal = [], []
        c, vals = linear_rounds(gk, gate, [x[rank::world] for x in X], q[:m1], seed, True, 1, proof, chal)
        rows = allgather_elements(vals, world)
        c, vals = linear_rounds(gk, gate, [[rows[r][t] for r in range(world)] for t in range(3)], q[m1:], 1, False, c, proof, chal)
        assert proof == want[0] and chal == want[1] and [c] + vals == want[2], ("linear", bn, rank)
        # ---- identity gate, several claims (MiMC layer 2)
        X, claims, qs, gate = o.initialize_multi_instance(bn, 5)
        want = o.sumcheck_prove(X, qs, claims, gate)
        rho = from_fr(gk.host_mimc_hash(to_fr(claims)))[0]
        eq = [0] * (1 << m1)
        mult = 1
        for j, qj in enumerate(qs):
            sj = mult * from_fr(gk.host_shard_seed(to_fr(qj[m1:]), rank))[0] % Q
            eq = [(a + b) % Q for a, b in zip(eq, o.folded_eq_table(qj[:m1], sj))]
            mult = mult * rho % Q if j else rho
        Xl = [x[rank::world] for x in X]
        proof, chal = [], []
        last = generic_rounds(gk, gate, eq, Xl, m1, True, proof, chal)
        rows = allgather_elements(last, world)
        cols = [[rows[r][t] for r in range(world)] for t in range(len(last))]
        last = generic_rounds(gk, gate, cols[0], cols[1:], gamma, False, proof, chal)
        assert proof == want[0] and chal == want[1] and last == want[2], ("identity", bn, rank)
    dist.barrier()
    if rank == 0:
        print("DIST-OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
