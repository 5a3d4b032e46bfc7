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
        GateDesc g;
        if (!gate_get(d.gate, &g)) return fail("circuit: layer %d has unknown gate %d", l, d.gate);
        if (d.n_in != g.n_in) return fail("circuit: layer %d: gate %s takes %d inputs, got %d", l, g.id.c_str(), g.n_in, d.n_in);
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

// Build-defined circuits of one GMiMC compression, out = GMimcT{t}.UpdateInplace(state, block)[0]
// (hash/gmimc.go:52-65).  The reference has the hasher but no gate or circuit for it; a round
//     state[j] += block[j] + Ark_i  (all j);   state[0] = state[0]^7;   rotate left
// is, per wire, one add layer (x + b + Ark_i) for the t-1 linear branches and one cipher layer (b + x + Ark_i)^7 for
// the S-box branch; inputs used more than once sit behind explicit copy layers (as examples/mimc.go:20 does for the
// key); layers that do not reach the output are pruned.
struct GmimcBuilder {
    struct Tmp {
        int gate, n_in, in[GKRHIP_MAX_GATE_INPUTS];
        E ark;
    };
    std::vector<Tmp> L;
    int add(int gate, std::initializer_list<int> ins, const E& ark) {
        Tmp t;
        memset(&t, 0, sizeof t);
        t.gate = gate;
        t.n_in = (int)ins.size();
        int k = 0;
        for (int v : ins) t.in[k++] = v;
        t.ark = ark;
        L.push_back(t);
        return (int)L.size() - 1;
    }
    // prune what does not reach the last layer; kept_inputs (optional) receives the original indices of the input
    // layers that remain (an input layer without a consumer would have no claim to check, gkr/verifier.go:120-132)
    // (output: the layer whose table is the circuit's output, default the last one added; it is the last one kept)
    std::vector<gkrhip_layer> finish(std::vector<int>* kept_inputs = nullptr, int output = -1) {
        std::vector<char> need(L.size(), 0);
        if (output < 0) output = (int)L.size() - 1;
        need[output] = 1;
        L.resize(output + 1);
        for (int l = (int)L.size() - 1; l >= 0; l--)
            if (need[l])
                for (int k = 0; k < L[l].n_in; k++) need[L[l].in[k]] = 1;
        std::vector<int> ren(L.size(), -1);
        std::vector<gkrhip_layer> out;
        for (size_t l = 0; l < L.size(); l++) {
            if (!need[l]) continue;
            if (L[l].gate < 0 && kept_inputs) kept_inputs->push_back((int)l);
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
};

// t = 2 with two-input gates only (BASELINE config 5): inputs 0..3 = s0, s1, b0, b1; the feed-forward
// x_91 + s0 + b0 is two add layers with Ark = 0.
std::vector<gkrhip_layer> gmimc_t2_layers() {
    GmimcBuilder B;
    for (int i = 0; i < 4; i++) B.add(-1, {}, hfr::ZERO);
    const int cs0 = B.add(GKRHIP_GATE_IDENTITY, {0}, hfr::ZERO);
    const int cb0 = B.add(GKRHIP_GATE_IDENTITY, {2}, hfr::ZERO);
    const int cb1 = B.add(GKRHIP_GATE_IDENTITY, {3}, hfr::ZERO);
    int x = cs0, y = 1;
    for (int i = 0; i < hfr::MIMC_ROUNDS; i++) {
        const int nx = B.add(GKRHIP_GATE_ADD, {y, cb1}, hfr::ARKS[i]);
        const int ny = B.add(GKRHIP_GATE_CIPHER, {cb0, x}, hfr::ARKS[i]);
        x = nx;
        y = ny;
    }
    const int t1 = B.add(GKRHIP_GATE_ADD, {x, cs0}, hfr::ZERO);
    B.add(GKRHIP_GATE_ADD, {t1, cb0}, hfr::ZERO);
    return B.finish();
}

// any t in {2, 4, 8} (hash/gmimc.go:16-20); the feed-forward is ONE layer of the registered three-input gate "sum3"
// (state'[0] + state[0] + block[0]).  The reference's round never mixes the branches (every branch gets its own
// key and Ark added, branch 0 goes through the S-box, the state rotates), so state'[0] depends on ONE initial branch
// (number 91 mod t) besides the feed-forward operands: the circuit's input layers are exactly the operands that
// matter, input_map[k] = j for state[j], t + j for block[j] (t = 2: all four; t = 4: 6 of 8; t = 8: 10 of 16).
int gmimc_layers(int t, std::vector<gkrhip_layer>* out, std::vector<int>* input_map) {
    if (t != 2 && t != 4 && t != 8) return fail("gmimc circuit: t = %d (2, 4 or 8)", t);
    gkrhip_gate_desc d;
    memset(&d, 0, sizeof d);
    strcpy(d.id, "sum3");
    d.n_in = 3;
    d.sum_mask = 7;
    d.power = 1;
    int sum3 = -1;
    CHK(gate_register(&d, &sum3));
    GmimcBuilder B;
    for (int i = 0; i < 2 * t; i++) B.add(-1, {}, hfr::ZERO);
    std::vector<int> st(t), cb(t);
    const int cs0 = B.add(GKRHIP_GATE_IDENTITY, {0}, hfr::ZERO);      // state[0]: round 0 and the feed-forward
    for (int j = 0; j < t; j++) st[j] = j == 0 ? cs0 : j;
    for (int j = 0; j < t; j++) cb[j] = B.add(GKRHIP_GATE_IDENTITY, {t + j}, hfr::ZERO);   // block[j]: every round
    for (int i = 0; i < hfr::MIMC_ROUNDS; i++) {
        std::vector<int> nx(t);
        for (int j = 1; j < t; j++) nx[j - 1] = B.add(GKRHIP_GATE_ADD, {st[j], cb[j]}, hfr::ARKS[i]);
        nx[t - 1] = B.add(GKRHIP_GATE_CIPHER, {cb[0], st[0]}, hfr::ARKS[i]);
        st = nx;
    }
    B.add(sum3, {st[0], cs0, cb[0]}, hfr::ZERO);
    *out = B.finish(input_map);
    return 0;
}

// The whole sponge GMimcT{t}.Hash(msg) for messages of nblocks * t elements (hash/gmimc.go:29-49: state = 0; every
// block of t elements goes through UpdateInplace; the hash is state[0]).  The whole state is carried from block to
// block: from the second block on the feed-forward perm[j] + state[j] + block[j] is a "sum3" layer on EVERY branch (a
// layer with two consumers, round 0 and the feed-forward of the next block, gets two claims like any other).  In the
// first block the state is zero and the layers are the registered one-input gates "addark1" (x + Ark) and "pow7ark1"
// ((x + Ark)^7) and a two-input add for the feed-forward.  input_map[k] = index into msg of input layer k.
int gmimc_hash_layers(int t, int nblocks, std::vector<gkrhip_layer>* out, std::vector<int>* input_map) {
    if (t != 2 && t != 4 && t != 8) return fail("gmimc hash circuit: t = %d (2, 4 or 8)", t);
    if (nblocks < 1 || nblocks > 64) return fail("gmimc hash circuit: %d blocks (1..64)", nblocks);
    int sum3 = -1, add1 = -1, pow1 = -1;
    auto reg = [&](const char* id, int n_in, unsigned mask, int power, int* g) {
        gkrhip_gate_desc d;
        memset(&d, 0, sizeof d);
        strcpy(d.id, id);
        d.n_in = n_in;
        d.sum_mask = mask;
        d.power = power;
        return gate_register(&d, g);
    };
    CHK(reg("sum3", 3, 7u, 1, &sum3));
    CHK(reg("addark1", 1, 1u, 1, &add1));
    CHK(reg("pow7ark1", 1, 1u, 7, &pow1));
    GmimcBuilder B;
    for (int i = 0; i < t * nblocks; i++) B.add(-1, {}, hfr::ZERO);
    std::vector<int> st(t, -1);
    for (int b = 0; b < nblocks; b++) {
        std::vector<int> cb(t);
        for (int j = 0; j < t; j++) cb[j] = B.add(GKRHIP_GATE_IDENTITY, {b * t + j}, hfr::ZERO);
        const std::vector<int> old = st;
        std::vector<int> cur = st;
        for (int i = 0; i < hfr::MIMC_ROUNDS; i++) {
            std::vector<int> nx(t);
            for (int j = 1; j < t; j++)
                nx[j - 1] = cur[j] < 0 ? B.add(add1, {cb[j]}, hfr::ARKS[i]) : B.add(GKRHIP_GATE_ADD, {cur[j], cb[j]}, hfr::ARKS[i]);
            nx[t - 1] = cur[0] < 0 ? B.add(pow1, {cb[0]}, hfr::ARKS[i]) : B.add(GKRHIP_GATE_CIPHER, {cb[0], cur[0]}, hfr::ARKS[i]);
            cur = nx;
        }
        for (int j = 0; j < t; j++)
            st[j] = old[j] < 0 ? B.add(GKRHIP_GATE_ADD, {cur[j], cb[j]}, hfr::ZERO) : B.add(sum3, {cur[j], old[j], cb[j]}, hfr::ZERO);
    }
    *out = B.finish(input_map, st[0]);
    return 0;
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
