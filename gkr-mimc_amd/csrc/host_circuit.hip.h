// host_circuit.hip.h -- circuit description, layer wiring and proof layout (circuit/circuit.go:11-91,
// examples/mimc.go:10-37, prover/gadget/hints.go:76-116).  Included by gkrhip.hip inside its anonymous namespace.
#pragma once
// ---- circuit description (circuit/circuit.go:11-44) -------------------------------------------------
struct Layer {
    std::vector<int> in, out;
    int gate = -1;  // -1: input layer
    E ark = hfr::ZERO;
};
typedef std::vector<Layer> Circuit;

Circuit mimc_circuit() {  // examples/mimc.go:10-37
    Circuit c(94);
    c[2].in = {0};
    c[2].gate = GKRHIP_GATE_IDENTITY;
    for (int i = 0; i < 91; i++) {
        c[i + 3].in = {2, i == 0 ? 1 : i + 2};
        c[i + 3].gate = GKRHIP_GATE_CIPHER;
        c[i + 3].ark = hfr::ARKS[i];
    }
    for (size_t l = 0; l < c.size(); l++)  // BuildCircuit
        for (int p : c[l].in) c[p].out.push_back((int)l);
    return c;
}

// circuit from a flat description (circuit/circuit.go:11-44: In given, Out computed by BuildCircuit)
int circuit_from_layers(const gkrhip_layer* layers, int n, Circuit* out) {
    if (n < 2 || n > 4096) return fail("circuit: %d layers", n);
    Circuit c(n);
    bool seen_gate = false;
    for (int l = 0; l < n; l++) {
        const gkrhip_layer& d = layers[l];
        if (d.gate < 0) {
            if (seen_gate) return fail("circuit: input layer %d after a gate layer", l);
            if (d.n_in != 0) return fail("circuit: input layer %d has inputs", l);
            continue;
        }
        seen_gate = true;
        if (d.gate != GKRHIP_GATE_IDENTITY && d.gate != GKRHIP_GATE_CIPHER && d.gate != GKRHIP_GATE_ADD)
            return fail("circuit: layer %d has unknown gate %d", l, d.gate);
        const int want = d.gate == GKRHIP_GATE_IDENTITY ? 1 : 2;
        if (d.n_in != want) return fail("circuit: layer %d: gate %d takes %d inputs, got %d", l, d.gate, want, d.n_in);
        c[l].gate = d.gate;
        memcpy(c[l].ark.l, d.ark, 32);
        if (!hfr::is_canonical(c[l].ark)) return fail("circuit: layer %d: Ark is not a canonical element", l);
        for (int k = 0; k < d.n_in; k++) {
            if (d.in[k] < 0 || d.in[k] >= l) return fail("circuit: layer %d reads layer %d (must be an earlier layer)", l, d.in[k]);
            c[l].in.push_back(d.in[k]);
        }
    }
    if (c[0].gate >= 0) return fail("circuit: no input layer");
    if (c[n - 1].gate < 0) return fail("circuit: the last layer must be a gate layer (the output)");
    for (int l = 0; l < n; l++)
        for (int p : c[l].in) c[p].out.push_back(l);
    for (int l = 0; l < n; l++) {
        if (c[l].gate < 0 && c[l].out.size() > 1)   // circuit/circuit.go:36-41
            return fail("Layer %d is an input layer but has %zu outputs", l, c[l].out.size());
        if (l < n - 1 && c[l].out.empty()) return fail("circuit: layer %d feeds nothing (only the last layer may)", l);
    }
    *out = c;
    return 0;
}

// Build-defined circuit of one GMiMC (t = 2) compression, out = GMimcT2.UpdateInplace([s0,s1],[b0,b1])[0]
// (hash/gmimc.go:52-65): inputs 0..3 = s0, s1, b0, b1; per round one add layer x' = y + b1 + Ark_i and one
// cipher layer y' = (b0 + x + Ark_i)^7; explicit copy layers for the multi-use inputs; feed-forward by two add
// layers with Ark = 0; layers that do not reach the output are pruned.
std::vector<gkrhip_layer> gmimc_t2_layers() {
    struct Tmp {
        int gate, n_in, in[2];
        E ark;
    };
    std::vector<Tmp> L;
    auto add = [&](int gate, int a, int b, const E& ark) {
        Tmp t;
        t.gate = gate;
        t.n_in = gate < 0 ? 0 : (gate == GKRHIP_GATE_IDENTITY ? 1 : 2);
        t.in[0] = a;
        t.in[1] = b;
        t.ark = ark;
        L.push_back(t);
        return (int)L.size() - 1;
    };
    for (int i = 0; i < 4; i++) add(-1, 0, 0, hfr::ZERO);
    const int cs0 = add(GKRHIP_GATE_IDENTITY, 0, 0, hfr::ZERO);
    const int cb0 = add(GKRHIP_GATE_IDENTITY, 2, 0, hfr::ZERO);
    const int cb1 = add(GKRHIP_GATE_IDENTITY, 3, 0, hfr::ZERO);
    int x = cs0, y = 1;
    for (int i = 0; i < hfr::MIMC_ROUNDS; i++) {
        const int nx = add(GKRHIP_GATE_ADD, y, cb1, hfr::ARKS[i]);
        const int ny = add(GKRHIP_GATE_CIPHER, cb0, x, hfr::ARKS[i]);
        x = nx;
        y = ny;
    }
    const int t1 = add(GKRHIP_GATE_ADD, x, cs0, hfr::ZERO);
    add(GKRHIP_GATE_ADD, t1, cb0, hfr::ZERO);
    std::vector<char> need(L.size(), 0);
    need.back() = 1;
    for (int l = (int)L.size() - 1; l >= 0; l--)
        if (need[l])
            for (int k = 0; k < L[l].n_in; k++) need[L[l].in[k]] = 1;
    for (int i = 0; i < 4; i++) need[i] = 1;
    std::vector<int> ren(L.size(), -1);
    std::vector<gkrhip_layer> out;
    for (size_t l = 0; l < L.size(); l++) {
        if (!need[l]) continue;
        ren[l] = (int)out.size();
        gkrhip_layer d;
        memset(&d, 0, sizeof d);
        d.gate = L[l].gate;
        d.n_in = L[l].n_in;
        for (int k = 0; k < d.n_in; k++) d.in[k] = ren[L[l].in[k]];
        memcpy(d.ark, L[l].ark.l, 32);
        out.push_back(d);
    }
    return out;
}

size_t proof_len(const Circuit& c, int bN) {  // hints.go:76-116
    size_t sc = 0, cl = 0, qp = 0;
    for (const Layer& l : c) {
        if (l.gate >= 0) sc += (size_t)bN * (gate_degree(l.gate) + 2);
        cl += l.out.size();
        qp += (size_t)bN * l.out.size();
    }
    return sc + cl + qp + bN;
}
