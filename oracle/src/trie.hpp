// ORACLE — TEST INFRASTRUCTURE ONLY (see rng.hpp).
//
// trie.hpp — CPU restatement of the weighted choice map and address mask that the
// reference's effect handlers keep their bookkeeping in:
//   modppl/src/address.rs:8-49    SplitAddr (split at the FIRST '/', trim the head; regex ^(.*?)/(.*)$)
//   modppl/src/address.rs:52-146  AddrMap  (visit / search / complement / all_visited)
//   modppl/src/trie.rs:5-248      Trie<V>  (weight bookkeeping: w_observe, insert, remove, merge,
//                                           schema, collect)
// Only what DynGenFnHandler uses is restated.  std::map replaces HashMap (iteration order is
// unspecified in the reference; every consumer here is order-independent up to fp summation
// order of weights, which the reference does not fix either).
#pragma once
#include <any>
#include <map>
#include <memory>
#include <optional>
#include <string>

#include "dists.hpp"

namespace oracle {

inline std::string trim(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

// address.rs:24-35
struct SplitAddr {
    bool is_term;
    std::string first, rest;
    static SplitAddr from_addr(const std::string& addr) {
        const size_t p = addr.find('/');
        if (p == std::string::npos) return {true, trim(addr), ""};
        return {false, trim(addr.substr(0, p)), addr.substr(p + 1)};
    }
};

// address.rs:52-146
struct AddrMap {
    std::map<std::string, AddrMap> m;
    bool is_leaf() const { return m.empty(); }
    const AddrMap* search(const std::string& addr) const {
        const SplitAddr s = SplitAddr::from_addr(addr);
        auto it = m.find(s.first);
        if (it == m.end()) return nullptr;
        return s.is_term ? &it->second : it->second.search(s.rest);
    }
    void insert(const std::string& addr, AddrMap sub) { m[addr] = std::move(sub); }
    void visit(const std::string& addr) {
        const SplitAddr s = SplitAddr::from_addr(addr);
        AddrMap& sub = m[s.first];  // entry().or_insert(new)
        if (!s.is_term) sub.visit(s.rest);
    }
    bool all_visited(const AddrMap& other) const {
        for (const auto& [addr, sub] : other.m) {
            const AddrMap* sv = search(addr);
            if (!sv) return false;
            if (!sv->is_leaf() && !sv->all_visited(sub)) return false;
        }
        return true;
    }
    AddrMap complement(const AddrMap& mask) const {
        AddrMap c;
        for (const auto& [addr, sub] : m) {
            const AddrMap* sm = mask.search(addr);
            if (!sm) {
                c.visit(addr);
            } else if (!sub.is_leaf() && !sm->is_leaf()) {
                AddrMap sc = sub.complement(*sm);
                if (!sc.is_leaf()) c.insert(addr, std::move(sc));
            }
        }
        return c;
    }
    bool operator==(const AddrMap& o) const { return m == o.m; }
};

// trie.rs:5-248; V = std::shared_ptr<const std::any> plays Arc<dyn Any + Send + Sync>.
using DynValue = std::shared_ptr<const std::any>;
template <class T>
DynValue arc(T v) { return std::make_shared<const std::any>(std::move(v)); }

struct Trie {
    std::map<std::string, Trie> mapping;
    std::optional<DynValue> value;
    double weight_ = 0.;

    static Trie leaf(DynValue v, double w) { Trie t; t.value = std::move(v); t.weight_ = w; return t; }
    bool is_empty() const { return mapping.empty() && !value.has_value(); }
    bool is_leaf() const { return mapping.empty() && value.has_value(); }
    size_t len() const { return mapping.size(); }
    double weight() const { return weight_; }
    std::optional<DynValue> take_inner() { auto v = value; value.reset(); return v; }
    void replace_inner(DynValue v) { value = std::move(v); }
    DynValue expect_inner(const std::string& msg) const {
        if (!value) throw Panic(msg);
        return *value;
    }

    const Trie* search(const std::string& addr) const {  // trie.rs:87-96
        const SplitAddr s = SplitAddr::from_addr(addr);
        auto it = mapping.find(s.first);
        if (s.is_term) return it == mapping.end() ? nullptr : &it->second;
        if (it == mapping.end()) throw Panic("search: missing prefix \"" + s.first + "\"");  // mapping[first] panics
        return it->second.search(s.rest);
    }
    void observe(const std::string& addr, DynValue v) {  // trie.rs:99-115
        const SplitAddr s = SplitAddr::from_addr(addr);
        if (s.is_term) {
            if (mapping.count(s.first)) throw Panic("observe: attempted to put into occupied address \"" + s.first + "\"");
            mapping[s.first] = Trie::leaf(std::move(v), 0.0);
        } else {
            mapping[s.first].observe(s.rest, std::move(v));
        }
    }
    void w_observe(const std::string& addr, DynValue v, double w) {  // trie.rs:118-135
        weight_ += w;
        const SplitAddr s = SplitAddr::from_addr(addr);
        if (s.is_term) {
            if (mapping.count(s.first)) throw Panic("w_observe: attempted to put into occupied address \"" + s.first + "\"");
            mapping[s.first] = Trie::leaf(std::move(v), w);
        } else {
            mapping[s.first].w_observe(s.rest, std::move(v), w);
        }
    }
    void insert(const std::string& addr, Trie sub) {  // trie.rs:138-155
        weight_ += sub.weight_;
        const SplitAddr s = SplitAddr::from_addr(addr);
        if (s.is_term) {
            if (mapping.count(s.first)) throw Panic("insert: attempted to put into occupied address \"" + s.first + "\"");
            mapping[s.first] = std::move(sub);
        } else {
            mapping[s.first].insert(s.rest, std::move(sub));
        }
    }
    std::optional<Trie> remove(const std::string& addr) {  // trie.rs:158-184
        const SplitAddr s = SplitAddr::from_addr(addr);
        std::optional<Trie> sub;
        if (s.is_term) {
            auto it = mapping.find(s.first);
            if (it != mapping.end()) { sub = std::move(it->second); mapping.erase(it); }
        } else {
            auto it = mapping.find(s.first);
            if (it != mapping.end()) {
                sub = it->second.remove(s.rest);
                if (it->second.is_empty()) remove(s.first);  // also subtracts the emptied node's residual weight
            }
        }
        if (sub) weight_ -= sub->weight_;
        return sub;
    }
    void merge(Trie other) {  // trie.rs:187-203
        for (auto& [addr, osub] : other.mapping) {
            if (osub.is_leaf()) {
                w_observe(addr, *osub.value, osub.weight_);
            } else {
                auto it = mapping.find(addr);
                if (it != mapping.end()) it->second.merge(std::move(osub));
                else insert(addr, std::move(osub));
            }
        }
    }
    AddrMap schema() const {  // trie.rs:206-216
        AddrMap a;
        for (const auto& [addr, sub] : mapping) {
            if (sub.is_leaf()) a.visit(addr);
            else a.insert(addr, sub.schema());
        }
        return a;
    }
    // trie.rs:222-246: returns (remaining self, collected, collected weight)
    static void collect(Trie self, const AddrMap& mask, Trie& rest, Trie& collected, double& w) {
        collected = Trie();
        if (self.schema() == mask) {
            w = self.weight();
            rest = Trie();
            collected = std::move(self);
            return;
        } else if (!mask.is_leaf()) {
            for (const auto& [addr, submask] : mask.m) {
                std::optional<Trie> sub = self.remove(addr);
                if (!sub) throw Panic("collect: unreachable");
                if (submask.is_leaf()) {
                    collected.insert(addr, std::move(*sub));
                } else {
                    Trie r, c; double cw;
                    collect(std::move(*sub), submask, r, c, cw);
                    if (!r.is_empty()) self.insert(addr, std::move(r));
                    if (!c.is_empty()) collected.insert(addr, std::move(c));
                }
            }
        }
        w = collected.weight();
        rest = std::move(self);
    }

    template <class T>
    T read(const std::string& addr) const {  // dyngenfn.rs:17-35
        const Trie* t = search(addr);
        if (!t) throw Panic("read: failed when searching empty address \"" + addr + "\"");
        if (!t->value) throw Panic("read: no inner value at \"" + addr + "\"");
        const T* p = std::any_cast<T>(t->value->get());
        if (!p) throw Panic("read: failed when downcasting type at address \"" + addr + "\"");
        return *p;
    }
};
using DynTrie = Trie;

}  // namespace oracle
