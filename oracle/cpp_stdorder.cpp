/* TEST INFRASTRUCTURE ONLY.  The one place where the oracle needs the C++ standard library the reference is built
 * against: TreeAgent::update_available (agents/cppmodule/agent.cpp:284-301) copies its `traversed` set, a
 * std::unordered_set<size_t>, into the `occupied` vector in the set's ITERATION order, and a later GC keeps a prefix of
 * that vector alive (see remove_nodes in agent_oracle.c).  The iteration order is a property of the platform's
 * libstdc++ hash table, so the restatement obtains it the way the reference does: from the same container. */
#include <cstddef>
#include <cstdint>
#include <deque>
#include <unordered_set>

extern "C" int orc_cpp_traversed_order(int root, const int32_t *child, int n_actions, int32_t *out) {
    std::deque<size_t> frontier;
    std::unordered_set<size_t> seen;
    frontier.push_back((size_t)root);
    for (size_t i = 0; i < frontier.size(); ++i) {
        size_t node = frontier[i];
        if (!seen.insert(node).second) continue;
        for (int c = 0; c < n_actions; ++c) {
            size_t k = (size_t)child[node * (size_t)n_actions + c]; /* follows child 0 as well: node 0 stays occupied */
            if (seen.count(k) == 0) frontier.push_back(k);
        }
    }
    int n = 0;
    for (size_t v : seen) out[n++] = (int32_t)v;
    return n;
}
