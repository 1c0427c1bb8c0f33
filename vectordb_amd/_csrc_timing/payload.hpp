// Host-side payload model of libvdbmi: PayloadValue / Payload / Filter / PayloadIndex with the
// semantics of src/utils/payload.rs, src/payload_storage/filters.rs and stores.rs, plus the
// filter -> per-slot bitmap compiler that feeds the device masks.
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <optional>
#include <set>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

#include "common.hpp"

namespace vdb {

// Ordered f64 with OrderedFloat<f64> semantics (ordered-float 4.6): NaN equals NaN and sorts
// last; -0.0 equals 0.0.
struct OF64 {
    double v;
    static int cmp(double a, double b) {
        const bool an = std::isnan(a), bn = std::isnan(b);
        if (an | bn) return (int)an - (int)bn;
        return (a > b) - (a < b);
    }
    bool operator==(const OF64& o) const { return cmp(v, o.v) == 0; }
    bool operator!=(const OF64& o) const { return cmp(v, o.v) != 0; }
    bool operator<(const OF64& o) const { return cmp(v, o.v) < 0; }
    bool operator<=(const OF64& o) const { return cmp(v, o.v) <= 0; }
    bool operator>(const OF64& o) const { return cmp(v, o.v) > 0; }
    bool operator>=(const OF64& o) const { return cmp(v, o.v) >= 0; }
};
struct Bool8 {   // keeps std::vector<bool> out of the variant
    uint8_t v;
    bool operator==(const Bool8& o) const { return (v != 0) == (o.v != 0); }
    bool operator!=(const Bool8& o) const { return !(*this == o); }
    bool operator<(const Bool8& o) const { return (v != 0) < (o.v != 0); }
    bool operator<=(const Bool8& o) const { return (v != 0) <= (o.v != 0); }
    bool operator>(const Bool8& o) const { return (v != 0) > (o.v != 0); }
    bool operator>=(const Bool8& o) const { return (v != 0) >= (o.v != 0); }
};

// PayloadValue, payload.rs:8-18.  The variant index equals the Rust discriminant / vdb_value_kind.
using ValueData = std::variant<int64_t, OF64, std::string, Bool8, std::vector<int64_t>, std::vector<OF64>,
                               std::vector<std::string>, std::vector<Bool8>>;
}  // namespace vdb

struct vdb_value {
    vdb::ValueData d;
    int kind() const { return (int)d.index(); }
    bool scalar() const { return d.index() <= VDB_T_BOOL; }                // stores.rs:63-71
    bool operator==(const vdb_value& o) const { return d == o.d; }         // derived PartialEq
};

namespace vdb {

template <class T> inline bool cmp_apply(vdb_cmp_op op, const T& a, const T& b) {
    switch (op) {
        case VDB_OP_EQ: return a == b;
        case VDB_OP_NEQ: return a != b;
        case VDB_OP_LT: return a < b;
        case VDB_OP_LTE: return a <= b;
        case VDB_OP_GT: return a > b;
        default: return a >= b;
    }
}

// PayloadValue::compare_scalar, payload.rs:150-186 (nullopt = None).
inline std::optional<bool> compare_scalar(const vdb_value& a, vdb_cmp_op op, const vdb_value& b) {
    if (a.kind() != b.kind()) return std::nullopt;
    switch (a.kind()) {
        case VDB_T_INT: return cmp_apply(op, std::get<int64_t>(a.d), std::get<int64_t>(b.d));
        case VDB_T_FLOAT: return cmp_apply(op, std::get<OF64>(a.d), std::get<OF64>(b.d));
        case VDB_T_STR: return cmp_apply(op, std::get<std::string>(a.d), std::get<std::string>(b.d));
        case VDB_T_BOOL:
            if (op == VDB_OP_EQ || op == VDB_OP_NEQ) return cmp_apply(op, std::get<Bool8>(a.d), std::get<Bool8>(b.d));
            return std::nullopt;
        default: return std::nullopt;
    }
}

// PayloadValue::evaluate_list_query, payload.rs:188-229.
template <class T> inline std::optional<bool> list_elem(const std::vector<T>& l, size_t i, vdb_cmp_op op, const T& x) {
    if (i >= l.size()) return std::nullopt;
    return cmp_apply(op, l[i], x);
}
inline std::optional<bool> evaluate_list_query(const vdb_value& self, vdb_list_query kind, vdb_cmp_op op, size_t n, const vdb_value* v) {
    const int lk = self.kind();
    switch (kind) {
        case VDB_LQ_CONTAINS: {
            if (!v || lk < VDB_T_LIST_INT || v->kind() != lk - 4) return std::nullopt;
            return std::visit([&](const auto& list) -> std::optional<bool> {
                using L = std::decay_t<decltype(list)>;
                if constexpr (std::is_same_v<L, std::vector<int64_t>>) return std::find(list.begin(), list.end(), std::get<int64_t>(v->d)) != list.end();
                else if constexpr (std::is_same_v<L, std::vector<OF64>>) return std::find(list.begin(), list.end(), std::get<OF64>(v->d)) != list.end();
                else if constexpr (std::is_same_v<L, std::vector<std::string>>) return std::find(list.begin(), list.end(), std::get<std::string>(v->d)) != list.end();
                else if constexpr (std::is_same_v<L, std::vector<Bool8>>) return std::find(list.begin(), list.end(), std::get<Bool8>(v->d)) != list.end();
                else return std::nullopt;
            }, self.d);
        }
        case VDB_LQ_EQUALS:
            if (!v || v->kind() != lk) return std::nullopt;                // discriminants differ -> None
            return self == *v;
        case VDB_LQ_LENGTH:
            return std::visit([&](const auto& list) -> std::optional<bool> {
                using L = std::decay_t<decltype(list)>;
                if constexpr (std::is_same_v<L, int64_t> || std::is_same_v<L, OF64> || std::is_same_v<L, std::string> || std::is_same_v<L, Bool8>) return std::nullopt;
                else return cmp_apply(op, list.size(), n);
            }, self.d);
        default: {
            if (!v || lk < VDB_T_LIST_INT || v->kind() != lk - 4) return std::nullopt;
            switch (lk) {
                case VDB_T_LIST_INT: return list_elem(std::get<std::vector<int64_t>>(self.d), n, op, std::get<int64_t>(v->d));
                case VDB_T_LIST_FLOAT: return list_elem(std::get<std::vector<OF64>>(self.d), n, op, std::get<OF64>(v->d));
                case VDB_T_LIST_STR: return list_elem(std::get<std::vector<std::string>>(self.d), n, op, std::get<std::string>(v->d));
                default:
                    if (op != VDB_OP_EQ && op != VDB_OP_NEQ) return std::nullopt;       // payload.rs:219-225
                    return list_elem(std::get<std::vector<Bool8>>(self.d), n, op, std::get<Bool8>(v->d));
            }
        }
    }
}

// Outcome of a predicate on one payload: true / false / Err(InvalidPayload(msg)).
struct Tri {
    enum : uint8_t { False = 0, True = 1, Err = 2 } s;
    std::string msg;
    static Tri of(bool b) { return {b ? True : False, {}}; }
    static Tri err(std::string m) { return {Err, std::move(m)}; }
};

}  // namespace vdb

// Payload, payload.rs:21.  Fields are kept sorted by key: ascending byte order is this library's
// stand-in for HashMap iteration order (DET-3).
struct vdb_payload {
    std::vector<std::pair<std::string, vdb_value>> fields;
    const vdb_value* get(const std::string& k) const {                     // payload.rs:32-34
        auto it = std::lower_bound(fields.begin(), fields.end(), k, [](const auto& f, const std::string& key) { return f.first < key; });
        return (it != fields.end() && it->first == k) ? &it->second : nullptr;
    }
    void set(const std::string& k, const vdb_value& v) {                   // payload.rs:27-29
        auto it = std::lower_bound(fields.begin(), fields.end(), k, [](const auto& f, const std::string& key) { return f.first < key; });
        if (it != fields.end() && it->first == k) it->second = v;
        else fields.insert(it, {k, v});
    }
    // Payload::compare_field, payload.rs:36-114
    vdb::Tri compare_field(const std::string& field, vdb_cmp_op op, const vdb_value& other) const {
        using namespace vdb;
        const vdb_value* have = get(field);
        if (!have) return Tri::err("Missing field: " + field);
        if (have->kind() == VDB_T_LIST_STR) {
            const auto& l = std::get<std::vector<std::string>>(have->d);
            if (other.kind() == VDB_T_LIST_STR) {
                if (op == VDB_OP_EQ) return Tri::of(l == std::get<std::vector<std::string>>(other.d));
                if (op == VDB_OP_NEQ) return Tri::of(l != std::get<std::vector<std::string>>(other.d));
                return Tri::err("Invalid operation for ListStr");
            }
            if (other.kind() == VDB_T_STR) {
                const bool has = std::find(l.begin(), l.end(), std::get<std::string>(other.d)) != l.end();
                if (op == VDB_OP_EQ) return Tri::of(has);
                if (op == VDB_OP_NEQ) return Tri::of(!has);
                return Tri::err("Invalid operation for ListStr and Str");
            }
        }
        auto r = compare_scalar(*have, op, other);
        return r ? Tri::of(*r) : Tri::err("Type mismatch for field: " + field);
    }
    // Payload::evaluate_list_field, payload.rs:117-128
    vdb::Tri evaluate_list_field(const std::string& field, vdb_list_query kind, vdb_cmp_op op, size_t n, const vdb_value* v) const {
        using namespace vdb;
        const vdb_value* have = get(field);
        if (!have) return Tri::err("Missing field: " + field);
        auto r = evaluate_list_query(*have, kind, op, n, v);
        return r ? Tri::of(*r) : Tri::err("Invalid list operation on field: " + field);
    }
    std::vector<std::string> scalar_keys() const {                         // segment.rs:40-44
        std::vector<std::string> k;
        for (auto& f : fields) if (f.second.scalar()) k.push_back(f.first);
        return k;
    }
};

// Filter, filters.rs:5-19
struct vdb_filter {
    enum Kind : int { Match, Compare, And, Or, Not } kind;
    std::string key; vdb_cmp_op op = VDB_OP_EQ; vdb_value value;
    std::vector<vdb_filter> kids;

    // evaluate_filter, filters.rs:23-75 (short-circuit, errors propagate)
    vdb::Tri eval(const vdb_payload& p) const {
        using vdb::Tri;
        switch (kind) {
            case Match: { const vdb_value* a = p.get(key); return Tri::of(a && *a == value); }
            case Compare: return p.compare_field(key, op, value);
            case And:
                for (auto& k : kids) { Tri t = k.eval(p); if (t.s != Tri::True) return t; }
                return Tri::of(true);
            case Or:
                for (auto& k : kids) { Tri t = k.eval(p); if (t.s != Tri::False) return t; }
                return Tri::of(false);
            default: { Tri t = kids.at(0).eval(p); if (t.s == Tri::Err) return t; return Tri::of(t.s == Tri::False); }
        }
    }
    // Canonical text used as the cache key of compiled bitmaps.
    void serialise(std::string& out) const;
};

// PayloadIndex, stores.rs:6-83.  Posting sets are ordered (ascending id: DET-2, DET-4).
struct vdb_payload_index {
    struct ScalarLess {
        bool operator()(const vdb_value& a, const vdb_value& b) const { return a.d < b.d; }   // variant: index first, then value
    };
    std::map<std::string, std::map<vdb_value, std::set<uint64_t>, ScalarLess>> index;
    void insert(uint64_t id, const vdb_payload& p) {                       // stores.rs:18-31
        for (auto& f : p.fields) if (f.second.scalar()) index[f.first][f.second].insert(id);
    }
    void remove(uint64_t id, const vdb_payload& p) {                       // stores.rs:34-52
        for (auto& f : p.fields) {
            if (!f.second.scalar()) continue;
            auto k = index.find(f.first);
            if (k == index.end()) continue;
            auto v = k->second.find(f.second);
            if (v != k->second.end()) { v->second.erase(id); if (v->second.empty()) k->second.erase(v); }
            if (k->second.empty()) index.erase(k);
        }
    }
    const std::set<uint64_t>* query_exact(const std::string& key, const vdb_value& v) const {   // stores.rs:55-61
        if (!v.scalar()) return nullptr;
        auto k = index.find(key);
        if (k == index.end()) return nullptr;
        auto it = k->second.find(v);
        return it == k->second.end() ? nullptr : &it->second;
    }
    std::optional<std::set<uint64_t>> all_for_key(const std::string& key) const {                // stores.rs:74-82
        auto k = index.find(key);
        if (k == index.end()) return std::nullopt;
        std::set<uint64_t> u;
        for (auto& v : k->second) u.insert(v.second.begin(), v.second.end());
        return u;
    }
};
