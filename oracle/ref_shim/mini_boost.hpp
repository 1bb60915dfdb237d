// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Stand-in for the two Boost pieces MeshUtil::removeLooseComponents touches [Boost is an un-vendored dependency of /root/reference, absent from this image]:
// boost::tuple<float, float, float> as an ordered map key, and boost::adjacency_list<vecS, vecS, undirectedS> with add_edge / num_vertices /
// connected_components.  Published behaviour restated: add_edge grows the vertex set to the larger endpoint; connected_components runs a depth-first
// search over the vertices in index order and numbers the components in the order their first vertex is met.  Nothing of this is reference code.
#pragma once
#include <cstddef>
#include <tuple>
#include <vector>

namespace boost {

struct vecS {}; struct undirectedS {};
template <class A, class B, class C> using tuple = std::tuple<A, B, C>;
namespace tuples { template <class A, class B, class C> inline std::tuple<A, B, C> make_tuple(A a, B b, C c) { return std::tuple<A, B, C>(a, b, c); } }

template <class OutEdgeList, class VertexList, class Directed>
struct adjacency_list { std::vector<std::vector<std::size_t>> adj; };

template <class G> inline void add_edge(std::size_t u, std::size_t v, G& g) {
    const std::size_t need = (u > v ? u : v) + 1;
    if (g.adj.size() < need) g.adj.resize(need);
    g.adj[u].push_back(v); g.adj[v].push_back(u);
}
template <class G> inline std::size_t num_vertices(const G& g) { return g.adj.size(); }
template <class G> inline int connected_components(const G& g, int* comp) {
    const std::size_t n = g.adj.size(); std::vector<char> seen(n, 0); std::vector<std::size_t> stack; int c = 0;
    for (std::size_t s = 0; s < n; ++s) {
        if (seen[s]) continue;
        stack.push_back(s); seen[s] = 1;
        while (!stack.empty()) { const std::size_t u = stack.back(); stack.pop_back(); comp[u] = c; for (std::size_t v : g.adj[u]) if (!seen[v]) { seen[v] = 1; stack.push_back(v); } }
        ++c;
    }
    return c;
}

}  // namespace boost
