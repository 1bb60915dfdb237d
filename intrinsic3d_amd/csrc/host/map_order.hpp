// Iteration order of the reference's voxel map without building the map.
//
// The reference keeps its voxels in std::unordered_map<Vec3i, T, hash> created with reserve(64) then max_load_factor(0.6)
// (sparse_voxel_grid.cpp:44-54), and everything it does "for each voxel" — the record order of a saved volume, correctSDF's in-place
// sweep, the edge set of the albedo regulariser — runs in that map's iteration order.  With libstdc++ the order is a pure function of the
// insertion sequence: nodes form one singly linked list, a node enters at the front of its bucket (or at the front of the whole list
// when the bucket is empty), and a rehash relinks the nodes in list order (bits/hashtable.h, _M_insert_bucket_begin / _M_rehash_aux).
// This replays exactly those list operations on index arrays — no node allocations, no key comparisons — and asks libstdc++'s own
// _Prime_rehash_policy when and to which prime bucket count to grow, so the bucket counts are the library's by construction.
#pragma once
#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <unordered_map>
#include <vector>
#include "../device/map_order_device.hpp"

namespace i3d {

inline size_t voxel_hash(int x, int y, int z) {                                 // mat.h:117-124: int -> size_t sign-extends before the multiply
    return ((size_t)x * 73856093) ^ ((size_t)y * 19349669) ^ ((size_t)z * 83492791);
}

// keys: n (x, y, z) in insertion order (`map[key] = i` for i = 0..n-1).  out[v] = the value held by the v-th visited element: its insertion
// index, or — when a key repeats and `distinct` is false — the index of its LAST occurrence (operator[] overwrites the payload, the node
// keeps its place).  With `distinct` the caller guarantees that no key repeats and the bucket chains are never searched.
inline void map_iteration_order_replay(const int* keys, size_t n, std::vector<int>& out, bool distinct = true) {
    constexpr int EMPTY = -1, BEFORE_BEGIN = -2;
    std::__detail::_Prime_rehash_policy pol(1.0f);
    // reserve(64) under the default load factor 1.0: _Hashtable::rehash(64)
    size_t nb = pol._M_next_bkt(std::max<size_t>(pol._M_bkt_for_elements(1), 64));
    pol = std::__detail::_Prime_rehash_policy(0.6f);                            // max_load_factor(0.6f) installs a fresh policy
    std::vector<int> bucket(nb, EMPTY), next(n, -1), payload;
    std::vector<size_t> code(n);
    if (!distinct) payload.resize(n);
    int head = -1; size_t count = 0;
    auto link_after = [&](int prev, int node) {                                 // node->next = prev->next; prev->next = node
        if (prev == BEFORE_BEGIN) { next[node] = head; head = node; } else { next[node] = next[prev]; next[prev] = node; }
    };
    for (size_t i = 0; i < n; ++i) {
        code[i] = voxel_hash(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]);
        if (!distinct) {                                                        // _M_find_before_node: walk the bucket's chain
            payload[i] = (int)i;
            const size_t b = code[i] % nb; bool found = false;
            if (bucket[b] != EMPTY)
                for (int p = bucket[b] == BEFORE_BEGIN ? head : next[bucket[b]]; p != -1 && code[p] % nb == b; p = next[p])
                    if (code[p] == code[i] && keys[3 * p] == keys[3 * i] && keys[3 * p + 1] == keys[3 * i + 1] && keys[3 * p + 2] == keys[3 * i + 2]) { payload[p] = (int)i; found = true; break; }
            if (found) continue;
        }
        const auto grow = pol._M_need_rehash(nb, count, 1); ++count;
        if (grow.first) {                                                       // _M_rehash_aux(unique keys)
            nb = grow.second;
            std::vector<int> nbk(nb, EMPTY);
            int p = head; head = -1; size_t bbegin = 0;
            while (p != -1) {
                const int nx = next[p]; const size_t b = code[p] % nb;
                if (nbk[b] == EMPTY) {
                    next[p] = head; head = p; nbk[b] = BEFORE_BEGIN;
                    if (next[p] != -1) nbk[bbegin] = p;
                    bbegin = b;
                } else link_after(nbk[b], p);
                p = nx;
            }
            bucket.swap(nbk);
        }
        const size_t b = code[i] % nb;                                          // _M_insert_bucket_begin
        if (bucket[b] != EMPTY) link_after(bucket[b], (int)i);
        else {
            next[i] = head; head = (int)i;
            if (next[i] != -1) bucket[code[next[i]] % nb] = (int)i;
            bucket[b] = BEFORE_BEGIN;
        }
    }
    out.clear(); out.reserve(count);
    for (int p = head; p != -1; p = next[p]) out.push_back(distinct ? p : payload[p]);
}

// The rehash schedule of n insertions into the reference's map (reserve(64), then max_load_factor 0.6), from libstdc++'s own policy object:
// epoch k holds bucket count nb and ends when the map has m_end elements (the insertion that triggers a rehash belongs to the NEXT epoch).
// _M_need_rehash does nothing (and changes nothing) while count + 1 <= _M_next_resize, so only the calls that can do work are made.
inline std::vector<MapEpoch> map_epochs(size_t n) {
    std::__detail::_Prime_rehash_policy pol(1.0f);
    size_t nb = pol._M_next_bkt(std::max<size_t>(pol._M_bkt_for_elements(1), 64));
    pol = std::__detail::_Prime_rehash_policy(0.6f);
    std::vector<MapEpoch> ep;
    size_t i = 0;
    while (i < n) {
        const auto grow = pol._M_need_rehash(nb, i, 1);
        if (grow.first) { ep.push_back(MapEpoch{i, nb}); nb = grow.second; }
        ++i;
        if (pol._M_next_resize > i) i = std::min(n, pol._M_next_resize);
    }
    ep.push_back(MapEpoch{n, nb});
    return ep;
}

// The closed form the device uses (device/map_order.hip), on the host with std::sort: after every rehash epoch the list is the elements sorted by
// (stamp of the arrival that created their bucket group, own stamp), both descending.  Test-only cross-check of the formulation (distinct keys).
inline void map_iteration_order_epochs(const int* keys, size_t n, std::vector<int>& out) {
    const std::vector<MapEpoch> ep = map_epochs(n);
    std::vector<size_t> code(n); std::vector<int> pos(n, 0), order;
    for (size_t i = 0; i < n; ++i) code[i] = voxel_hash(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]);
    size_t m_prev = 0;
    for (const MapEpoch& e : ep) {
        const size_t m = e.m_end;
        if (m == m_prev) continue;
        std::vector<int> gmin(e.nb, 0x7f7f7f7f), stamp(m);
        for (size_t i = 0; i < m; ++i) { stamp[i] = i < m_prev ? pos[i] : (int)i; int& g = gmin[code[i] % e.nb]; g = std::min(g, stamp[i]); }
        order.resize(m); for (size_t i = 0; i < m; ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { const int ga = gmin[code[a] % e.nb], gb = gmin[code[b] % e.nb]; return ga != gb ? ga > gb : stamp[a] > stamp[b]; });
        for (size_t r = 0; r < m; ++r) pos[order[r]] = (int)r;
        m_prev = m;
    }
    out = order;
}

// the same through a real std::unordered_map (used to cross-check the replay)
inline void map_iteration_order_stl(const int* keys, size_t n, std::vector<int>& out) {
    struct K { int x, y, z; bool operator==(const K& o) const { return x == o.x && y == o.y && z == o.z; } };
    struct H { size_t operator()(const K& k) const { return voxel_hash(k.x, k.y, k.z); } };
    std::unordered_map<K, int, H> m; m.reserve(64); m.max_load_factor(0.6f);
    for (size_t i = 0; i < n; ++i) m[K{keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]}] = (int)i;
    out.clear(); out.reserve(m.size());
    for (auto it = m.begin(); it != m.end(); ++it) out.push_back(it->second);
}

}  // namespace i3d
