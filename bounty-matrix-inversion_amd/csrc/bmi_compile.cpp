// Compiler passes of the host scheduler on the flat (CSR) form of a traced circuit: dead look-up elimination and the
// width-aware level schedule.  Context-free, CPU only (no HIP call): the reference leaves these steps to Concrete's
// compiler inside fhe.Compiler(...).compile (matrix_inversion/main.py:53-66); here they are two C-ABI functions so that
// compiling the 8x8 inverse (2.6 M look-ups) costs seconds, not minutes of Python loops.
// Leaves are numbered inputs first, then one per node in creation order (leaf of node i = n_in + i); creation order is
// a topological order.  bmi_amd/program.py holds the same algorithms in Python (the test suite compares the two).
#include <stdint.h>

#include <algorithm>
#include <queue>
#include <vector>

#include "../../include/bmi_tfhe.h"

extern "C" {

int bmi_circuit_prune(uint32_t n_in, uint32_t n_nodes, const int64_t *node_ptr, const int32_t *term_leaf,
                      const int32_t *out_leaf, uint64_t n_out_terms, uint8_t *live_node) {
    if ((n_nodes && (!node_ptr || !live_node)) || (n_out_terms && !out_leaf)) return -1;
    std::fill(live_node, live_node + n_nodes, (uint8_t)0);
    for (uint64_t e = 0; e < n_out_terms; e++) {
        const int64_t t = out_leaf[e];
        if (t < 0 || t >= (int64_t)n_in + n_nodes) return -1;
        if (t >= (int64_t)n_in) live_node[t - n_in] = 1;
    }
    for (int64_t i = (int64_t)n_nodes - 1; i >= 0; i--) {
        if (!live_node[i]) continue;
        for (int64_t e = node_ptr[i]; e < node_ptr[i + 1]; e++) {
            const int64_t t = term_leaf[e];
            if (t < 0 || t >= (int64_t)n_in + i) return -1;   // a node may only read inputs and earlier nodes
            if (t >= (int64_t)n_in) live_node[t - n_in] = 1;
        }
    }
    return 0;
}

// Width-aware list schedule with the depth of the ASAP schedule (see program.schedule_levels for the rationale):
// levels are filled in order; a level takes every ready node whose ALAP level it is, then ready nodes in ALAP order while
// the kernel rounds the critical ones need anyway have room (round_ ciphertexts per latency-kernel round up to two rounds,
// wide_round per throughput-kernel round beyond).  level_out[i] in 1..depth.
int bmi_circuit_schedule(uint32_t n_in, uint32_t n_nodes, const int64_t *node_ptr, const int32_t *term_leaf,
                         uint32_t round_, uint32_t wide_round, int32_t *level_out, int32_t *depth_out) {
    if (!depth_out || (n_nodes && (!node_ptr || !term_leaf || !level_out)) || round_ == 0 || wide_round == 0) return -1;
    const int64_t nn = n_nodes;
    std::vector<int32_t> asap(nn), alap(nn), indeg(nn, 0);
    std::vector<int64_t> succ_ptr(nn + 1, 0);
    int32_t depth = 0;
    for (int64_t i = 0; i < nn; i++) {
        int32_t lv = 0;
        for (int64_t e = node_ptr[i]; e < node_ptr[i + 1]; e++) {
            const int64_t t = term_leaf[e];
            if (t < 0 || t >= (int64_t)n_in + i) return -1;
            if (t >= (int64_t)n_in) {
                lv = std::max(lv, asap[t - n_in]);
                succ_ptr[t - n_in + 1]++;
                indeg[i]++;
            }
        }
        asap[i] = lv + 1;
        depth = std::max(depth, asap[i]);
    }
    *depth_out = depth;
    if (nn == 0) return 0;
    for (int64_t i = 0; i < nn; i++) succ_ptr[i + 1] += succ_ptr[i];
    std::vector<int32_t> succ(succ_ptr[nn]);
    {
        std::vector<int64_t> fill(succ_ptr.begin(), succ_ptr.end() - 1);
        for (int64_t i = 0; i < nn; i++)
            for (int64_t e = node_ptr[i]; e < node_ptr[i + 1]; e++)
                if (term_leaf[e] >= (int64_t)n_in) succ[fill[term_leaf[e] - n_in]++] = (int32_t)i;
    }
    for (int64_t i = nn - 1; i >= 0; i--) {   // successors have larger indices: final before their producers
        int32_t a = depth;
        for (int64_t e = succ_ptr[i]; e < succ_ptr[i + 1]; e++) a = std::min(a, alap[succ[e]] - 1);
        alap[i] = a;
    }
    // ready nodes bucketed by ALAP level; a node is never ready later than its ALAP, so bucket[t] is level t's critical set
    std::vector<std::vector<int32_t>> bucket(depth + 2);
    std::priority_queue<int32_t, std::vector<int32_t>, std::greater<int32_t>> keys;   // ALAP values with a non-empty bucket
    auto push_ready = [&](int32_t i) {
        if (bucket[alap[i]].empty()) keys.push(alap[i]);
        bucket[alap[i]].push_back(i);
    };
    for (int64_t i = 0; i < nn; i++)
        if (indeg[i] == 0) push_ready((int32_t)i);
    int64_t done = 0;
    int32_t t = 0;
    std::vector<int32_t> chosen;
    while (done < nn) {
        t++;
        if (t > depth) return -2;
        const size_t must = bucket[t].size();
        const size_t cap = must <= 2 * (size_t)round_ ? std::max<size_t>(1, (must + round_ - 1) / round_) * round_
                                                      : (must + wide_round - 1) / wide_round * wide_round;
        chosen.clear();
        while (!keys.empty() && chosen.size() < cap) {
            const int32_t a = keys.top();
            std::vector<int32_t> &b = bucket[a];
            const size_t room = cap - chosen.size();
            if (b.size() <= room) {
                chosen.insert(chosen.end(), b.begin(), b.end());
                b.clear();
                keys.pop();
            } else {
                chosen.insert(chosen.end(), b.end() - room, b.end());
                b.resize(b.size() - room);
            }
        }
        for (int32_t i : chosen) level_out[i] = t;
        for (int32_t i : chosen)
            for (int64_t e = succ_ptr[i]; e < succ_ptr[i + 1]; e++)
                if (--indeg[succ[e]] == 0) push_ready(succ[e]);
        done += (int64_t)chosen.size();
    }
    return t == depth ? 0 : -2;
}

}  // extern "C"
