// host_gates.hip.h -- the gate table: the native counterpart of the reference's circuit.Gate plug point
// (circuit/gates.go:9-21).  A gate is a DESCRIPTOR the kernels interpret,
//     Eval(xs...) = (sum of the inputs selected by mask + Ark)^power,  power = 1 or 7,  Degree() = power,
// so a new gate of the family needs no recompilation (gkrhip_gate_register).  Entries 0..2 are the built-in gates
// (IdentityGate circuit/gates/copy.go:9-32, CipherGate circuit/gates/cipher.go:11-70, and the add gate of the GMiMC
// circuits).  Included by gkrhip.hip inside its anonymous namespace.
#pragma once

struct GateDesc {
    std::string id;
    int n_in;
    unsigned mask;
    int power;
};
std::mutex g_gates_mu;
std::vector<GateDesc> g_gates = {
    {"CopyGate", 1, 1u, 1},     // IdentityGate.ID() (copy.go:12)
    {"CipherGate", 2, 3u, 7},   // the Ark is the layer's (cipher.go:22 appends it to the ID)
    {"AddArkGate", 2, 3u, 1},   // xs[0] + xs[1] + Ark
};

bool gate_get(int gate, GateDesc* out) {
    std::lock_guard<std::mutex> lk(g_gates_mu);
    if (gate < 0 || gate >= (int)g_gates.size()) return false;
    if (out) *out = g_gates[gate];
    return true;
}
int gate_degree(int gate) {   // Gate.Degree(): cipher.go:68-70, copy.go:30-32
    GateDesc d;
    return gate_get(gate, &d) ? d.power : 1;
}
// The gate as applied to `arity` tables.  IdentityGate takes any number of tables and returns xs[0]
// (copy.go:15-22; the reference's multi-instance tests pass [L, R], sumcheck/testing.go:28-57); every other
// gate takes exactly its n_in inputs.
int gate_resolve(int gate, int arity, GateDesc* out) {
    if (!gate_get(gate, out)) return fail("unknown gate id %d", gate);
    if (arity < 1 || arity > GKR_MAX_ARITY) return fail("arity %d not supported (1..%d)", arity, GKR_MAX_ARITY);
    if (gate == GKRHIP_GATE_IDENTITY) {
        out->n_in = arity;
        return 0;
    }
    if (arity != out->n_in) return fail("gate %s takes %d inputs, got %d", out->id.c_str(), out->n_in, arity);
    return 0;
}
// is this the shape the fused cipher round kernels are written for: (xs[0] + xs[1] + Ark)^7
inline bool gate_is_cipher2(const GateDesc& d) { return d.power == 7 && d.n_in == 2 && d.mask == 3u; }

int gate_register(const gkrhip_gate_desc* desc, int* id_out) {
    if (!desc || !id_out) return fail("gate_register: null argument");
    if (desc->n_in < 1 || desc->n_in > GKRHIP_MAX_GATE_INPUTS) return fail("gate_register: %d inputs (1..%d)", desc->n_in, GKRHIP_MAX_GATE_INPUTS);
    if (desc->power != 1 && desc->power != 7)
        return fail("gate_register: power %d -- the kernels evaluate (sum + Ark)^1 and (sum + Ark)^7 only", desc->power);
    const unsigned full = (1u << desc->n_in) - 1;
    if (desc->sum_mask == 0 || (desc->sum_mask & ~full)) return fail("gate_register: sum_mask 0x%x does not select inputs 0..%d", desc->sum_mask, desc->n_in - 1);
    char name[sizeof desc->id + 1];
    memcpy(name, desc->id, sizeof desc->id);
    name[sizeof desc->id] = 0;
    std::lock_guard<std::mutex> lk(g_gates_mu);
    for (size_t i = 3; i < g_gates.size(); i++) {
        const GateDesc& g = g_gates[i];
        if (g.n_in == desc->n_in && g.mask == desc->sum_mask && g.power == desc->power && g.id == name) {
            *id_out = (int)i;
            return 0;
        }
        if (g.id == name && name[0]) return fail("gate_register: a different gate is already registered under the ID %s", name);
    }
    g_gates.push_back(GateDesc{name, desc->n_in, desc->sum_mask, desc->power});
    *id_out = (int)g_gates.size() - 1;
    return 0;
}
// host evaluation of a gate on scalars (the verifier's final check, gkr/verifier.go:93-110)
E gate_eval_host(const GateDesc& d, const E& ark, const E* xs) {
    E s = ark;
    for (int k = 0; k < d.n_in; k++)
        if ((d.mask >> k) & 1u) s = hfr::add(s, xs[k]);
    return d.power == 7 ? hfr::pow7(s) : s;
}
