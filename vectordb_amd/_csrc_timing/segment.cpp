// Segment (src/segment/segment.rs) over the device-resident index: id assignment, payload
// bookkeeping, tombstones and purge on the host; every distance on the GPU.
#include <algorithm>
#include <cstring>
#include <functional>

#include "index.hpp"

using namespace vdb;

void vdb_filter::serialise(std::string& out) const {
    auto put_value = [&](const vdb_value& v) {
        out += char('a' + v.kind());
        std::visit([&](const auto& x) {
            using T = std::decay_t<decltype(x)>;
            auto put_str = [&](const std::string& s) { out += std::to_string(s.size()); out += ':'; out += s; };
            if constexpr (std::is_same_v<T, int64_t>) out += std::to_string(x);
            else if constexpr (std::is_same_v<T, OF64>) { uint64_t b; double d = std::isnan(x.v) ? NAN : (x.v == 0 ? 0.0 : x.v); std::memcpy(&b, &d, 8); out += std::to_string(b); }
            else if constexpr (std::is_same_v<T, std::string>) put_str(x);
            else if constexpr (std::is_same_v<T, Bool8>) out += x.v ? '1' : '0';
            else if constexpr (std::is_same_v<T, std::vector<int64_t>>) for (auto e : x) { out += std::to_string(e); out += ','; }
            else if constexpr (std::is_same_v<T, std::vector<OF64>>) for (auto e : x) { uint64_t b; std::memcpy(&b, &e.v, 8); out += std::to_string(b); out += ','; }
            else if constexpr (std::is_same_v<T, std::vector<std::string>>) for (auto& e : x) put_str(e);
            else for (auto e : x) out += e.v ? '1' : '0';
        }, v.d);
        out += ';';
    };
    switch (kind) {
        case Match: out += "M"; out += std::to_string(key.size()); out += ':'; out += key; put_value(value); break;
        case Compare: out += "C"; out += char('0' + (int)op); out += std::to_string(key.size()); out += ':'; out += key; put_value(value); break;
        case And: case Or: out += (kind == And ? "A(" : "O("); for (auto& k : kids) k.serialise(out); out += ')'; break;
        default: out += "N("; kids.at(0).serialise(out); out += ')';
    }
}

namespace {

std::vector<vdb_scored_point> one_query(vdb_index& h, const float* q, size_t len, size_t top_k, const vdb_index::FilterArgs& fa) {
    if (len != h.dim) throw Failure(err_length_mismatch(h.dim, len));              // hnsw.rs:519-524 / :593-598
    std::vector<vdb_scored_point> out(std::max<size_t>(top_k, 1));
    uint32_t n = 0;
    h.search_batch(q, 1, top_k, 0, fa, out.data(), &n, nullptr);
    out.resize(n);
    return out;
}

// hnsw.rs:210-325 on index `h` with the given payload structures.
void filter_aware_edges(vdb_index& h, const vdb_payload_index& pidx, const std::unordered_map<uint64_t, vdb_payload>& payloads,
                        uint64_t point_id, const vdb_payload& payload, const std::vector<std::string>& keys) {
    const uint32_t self = h.slot_of.at(point_id);
    const size_t m = h.m;
    std::set<uint64_t> extra;                                                      // DET-5
    struct Scored { uint32_t slot; float raw; };
    auto sort_raw = [&](std::vector<Scored>& v) {                                  // hnsw.rs:250-256 / :291-297 (stable)
        if (h.metric == VDB_DOT) std::stable_sort(v.begin(), v.end(), [](const Scored& a, const Scored& b) { return a.raw > b.raw; });
        else std::stable_sort(v.begin(), v.end(), [](const Scored& a, const Scored& b) { return a.raw < b.raw; });
    };
    for (const std::string& key : keys) {
        const vdb_value* value = payload.get(key);
        if (!value) continue;
        if (const std::set<uint64_t>* id_set = pidx.query_exact(key, *value)) {
            // reservoir sampling of <= 100 same-value points (rand 0.9 choose_multiple; DET-2)
            std::vector<uint32_t> reservoir; size_t after = 0; const size_t amount = 100;
            for (uint64_t pid : *id_set) {
                if (pid == point_id) continue;
                auto it = h.slot_of.find(pid);
                if (it == h.slot_of.end() || h.deleted[it->second]) continue;
                if (reservoir.size() < amount) { reservoir.push_back(it->second); continue; }
                const uint64_t k = rng_draw(h.opts.seed, 2, h.sample_draws++) % (after + 1 + amount);
                if (k < amount) reservoir[k] = it->second;
                ++after;
            }
            std::vector<float> raw = h.score_rows(self, reservoir);
            std::vector<Scored> scored;
            for (size_t i = 0; i < reservoir.size(); ++i) scored.push_back({reservoir[i], raw[i]});
            sort_raw(scored);
            for (size_t i = 0; i < scored.size() && i < m; ++i) extra.insert(h.ids[scored[i].slot]);
            if (extra.size() >= m) break;
        }
        std::vector<Scored> candidates;
        if (h.current_max_level > 0) {                                              // hnsw.rs:268-273
            // query = the stored row, no second normalisation: a build-mode launch at node level 0
            std::vector<vdb_scored_point> res = h.search_from_row(self, h.slot_of.at(*h.entry_point), h.ef);
            for (auto& sp : res) candidates.push_back({h.slot_of.at(sp.id), sp.raw_score});
        } else {                                                                    // hnsw.rs:275-288 (DET-6)
            std::vector<float> raw = h.score_all_rows(self);
            for (uint64_t lid : h.live) {
                if (lid == point_id) continue;
                const uint32_t s = h.slot_of.at(lid);
                candidates.push_back({s, raw[s]});
            }
        }
        sort_raw(candidates);
        size_t taken = 0;
        for (const Scored& c : candidates) {
            if (taken >= m) break;
            auto pit = payloads.find(h.ids[c.slot]);
            if (pit == payloads.end()) continue;
            const vdb_value* v = pit->second.get(key);
            if (!v || !(*v == *value)) continue;
            extra.insert(h.ids[c.slot]); ++taken;
        }
        if (extra.size() >= m) break;
    }
    extra.insert(point_id);                                                         // hnsw.rs:318
    for (uint64_t nid : extra) h.add_bidirectional_edge(0, self, h.slot_of.at(nid));
}

}  // namespace

uint64_t vdb_segment::insert(const float* v, size_t len, const vdb_payload* p) {     // segment.rs:32-61
    const uint64_t pid = next_id;
    if (len != hnsw->dim) throw Failure(err_length_mismatch(hnsw->dim, len));
    hnsw->insert(pid, v, len);
    if (p) {
        payload_index.insert(pid, *p);
        payloads[pid] = *p;
        note_payload(hnsw->slot_of.at(pid), *p);
        ++payload_version;
        const std::vector<std::string> keys = p->scalar_keys();
        if (!keys.empty()) filter_aware_edges(*hnsw, payload_index, payloads, pid, *p, keys);
    }
    next_id += 1;
    return pid;
}

// Extension: vectors go through vdb_index::insert_batch (rounds), then payload bookkeeping and the
// filter-aware edges of each point run in id order against the finished graph.
void vdb_segment::insert_batch(const float* v, size_t n, size_t len, const vdb_payload* const* p, uint64_t* out_ids) {
    if (len != hnsw->dim) throw Failure(err_length_mismatch(hnsw->dim, len));
    std::vector<uint64_t> pids(n);
    for (size_t i = 0; i < n; ++i) pids[i] = next_id + i;
    hnsw->insert_batch(pids.data(), v, n, len);
    for (size_t i = 0; i < n; ++i) {
        if (p && p[i]) {
            payload_index.insert(pids[i], *p[i]);
            payloads[pids[i]] = *p[i];
            note_payload(hnsw->slot_of.at(pids[i]), *p[i]);
        }
    }
    ++payload_version;
    if (p) {
        const size_t chunk = 65536;                        // bounds the host copies of the search results
        for (size_t lo = 0; lo < n; lo += chunk) {
            const size_t hi = std::min(n, lo + chunk);
            filter_aware_edges_batch(std::vector<uint64_t>(pids.begin() + lo, pids.begin() + hi), p + lo);
        }
    }
    next_id += n;
    if (out_ids) std::memcpy(out_ids, pids.data(), n * sizeof(uint64_t));
}

// hnsw.rs:210-325 for every point of a batch with two launches instead of two per point and key:
// all sampled same-value pairs are scored at once, and every point's row searches the graph (as it
// stands after the batch was linked, before any of these extra edges) once.  Per point the rules
// are those of filter_aware_edges(): keys in ascending order, up to m nearest of <= 100 sampled
// same-value points, then up to m same-value points among the search results, stop at m extras,
// self-edge included (hnsw.rs:318).  The sample is drawn by position from the posting list
// (Floyd's algorithm on the counter-based stream 2) instead of a reservoir pass over it.
void vdb_segment::filter_aware_edges_batch(const std::vector<uint64_t>& pids, const vdb_payload* const* p) {
    vdb_index& h = *hnsw;
    if (h.current_max_level == 0) {                          // tiny graph: the per-point form scans every row (hnsw.rs:275-288)
        for (size_t i = 0; i < pids.size(); ++i)
            if (p[i]) { auto keys = p[i]->scalar_keys(); if (!keys.empty()) filter_aware_edges(h, payload_index, payloads, pids[i], *p[i], keys); }
        return;
    }
    struct Job { size_t point; std::string key; const vdb_value* value; size_t first, count; };   // slice of the pair arrays
    std::vector<size_t> pts; std::vector<std::vector<std::string>> keys_of;
    for (size_t i = 0; i < pids.size(); ++i) {
        if (!p[i]) continue;
        std::vector<std::string> keys = p[i]->scalar_keys();
        if (keys.empty()) continue;
        pts.push_back(i); keys_of.push_back(std::move(keys));
    }
    if (pts.empty()) return;
    const size_t m = h.m, amount = 100;
    std::map<const std::set<uint64_t>*, std::vector<uint32_t>> posting;      // live slots of a posting list, in id order
    std::vector<Job> jobs; std::vector<uint32_t> qs, ss;
    for (size_t t = 0; t < pts.size(); ++t) {
        const size_t i = pts[t];
        const uint32_t self = h.slot_of.at(pids[i]);
        for (const std::string& key : keys_of[t]) {
            const vdb_value* value = p[i]->get(key);
            if (!value) continue;
            Job j{t, key, value, qs.size(), 0};
            if (const std::set<uint64_t>* id_set = payload_index.query_exact(key, *value)) {
                auto it = posting.find(id_set);
                if (it == posting.end()) {
                    std::vector<uint32_t> live; live.reserve(id_set->size());
                    for (uint64_t pid : *id_set) { auto s = h.slot_of.find(pid); if (s != h.slot_of.end() && !h.deleted[s->second]) live.push_back(s->second); }
                    it = posting.emplace(id_set, std::move(live)).first;
                }
                const std::vector<uint32_t>& live = it->second;
                if (live.size() <= amount + 1) {
                    for (uint32_t s : live) if (s != self) { qs.push_back(self); ss.push_back(s); }
                } else {
                    // amount + 1 distinct positions, so that dropping `self` (if drawn) still leaves `amount`;
                    // membership in a 256-slot open-addressing table on the stack, positions then in ascending order
                    const size_t N = live.size(), want = amount + 1;
                    size_t table[256]; std::fill(table, table + 256, SIZE_MAX);
                    size_t chosen[amount + 1]; size_t nc = 0;
                    auto add = [&](size_t v) -> bool {
                        size_t hsh = (v * 0x9E3779B97F4A7C15ull) >> 56;
                        while (table[hsh] != SIZE_MAX) { if (table[hsh] == v) return false; hsh = (hsh + 1) & 255; }
                        table[hsh] = v; chosen[nc++] = v; return true;
                    };
                    for (size_t r = N - want; r < N; ++r) {
                        const size_t t2 = (size_t)(rng_draw(h.opts.seed, 2, h.sample_draws++) % (r + 1));
                        if (!add(t2)) add(r);
                    }
                    std::sort(chosen, chosen + nc);
                    size_t taken = 0;
                    for (size_t c = 0; c < nc; ++c) { const size_t pos = chosen[c]; if (live[pos] == self || taken == amount) continue; qs.push_back(self); ss.push_back(live[pos]); ++taken; }
                }
            }
            j.count = qs.size() - j.first;
            jobs.push_back(std::move(j));
        }
    }
    const std::vector<float> raw = h.score_pairs(qs, ss);
    std::vector<uint32_t> rows(pts.size());
    for (size_t t = 0; t < pts.size(); ++t) rows[t] = h.slot_of.at(pids[pts[t]]);
    std::vector<vdb_scored_point> found; std::vector<uint32_t> found_n;
    const size_t ef = h.ef;
    h.search_from_rows(rows, h.slot_of.at(*h.entry_point), ef, found, found_n);

    struct Scored { uint32_t slot; float raw; uint32_t at; };
    // the first `keep` entries of the stable sort of hnsw.rs:250-256 / :291-297: ties keep their arrival order
    auto best_first = [&](std::vector<Scored>& v, size_t keep) {
        for (size_t i = 0; i < v.size(); ++i) v[i].at = (uint32_t)i;
        keep = std::min(keep, v.size());
        if (h.metric == VDB_DOT) std::partial_sort(v.begin(), v.begin() + keep, v.end(), [](const Scored& a, const Scored& b) { return a.raw > b.raw || (a.raw == b.raw && a.at < b.at); });
        else std::partial_sort(v.begin(), v.begin() + keep, v.end(), [](const Scored& a, const Scored& b) { return a.raw < b.raw || (a.raw == b.raw && a.at < b.at); });
    };
    size_t jb = 0;
    for (size_t t = 0; t < pts.size(); ++t) {
        const uint64_t point_id = pids[pts[t]];
        const uint32_t self = rows[t];
        std::set<uint64_t> extra;                                                  // DET-5
        bool done = false;
        for (; jb < jobs.size() && jobs[jb].point == t; ++jb) {
            if (done) continue;
            const Job& j = jobs[jb];
            if (payload_index.query_exact(j.key, *j.value)) {
                std::vector<Scored> scored;
                for (size_t k = 0; k < j.count; ++k) scored.push_back({ss[j.first + k], raw[j.first + k], 0});
                best_first(scored, m);
                for (size_t k = 0; k < scored.size() && k < m; ++k) extra.insert(h.ids[scored[k].slot]);
                if (extra.size() >= m) { done = true; continue; }
            }
            std::vector<Scored> candidates;
            for (uint32_t k = 0; k < found_n[t]; ++k) { const vdb_scored_point& sp = found[t * ef + k]; candidates.push_back({h.slot_of.at(sp.id), sp.raw_score, 0}); }
            best_first(candidates, candidates.size());     // every entry may be needed: only same-value rows count
            size_t taken = 0;
            for (const Scored& c : candidates) {
                if (taken >= m) break;
                auto pit = payloads.find(h.ids[c.slot]);
                if (pit == payloads.end()) continue;
                const vdb_value* v = pit->second.get(j.key);
                if (!v || !(*v == *j.value)) continue;
                extra.insert(h.ids[c.slot]); ++taken;
            }
            if (extra.size() >= m) done = true;
        }
        extra.insert(point_id);                                                     // hnsw.rs:318
        for (uint64_t nid : extra) h.add_bidirectional_edge(0, self, h.slot_of.at(nid));
    }
}

const float* vdb_segment::get_vector(uint64_t id) const {                           // segment.rs:64-69
    if (deleted.count(id)) return nullptr;
    auto it = hnsw->slot_of.find(id);
    if (it == hnsw->slot_of.end() || hnsw->deleted[it->second]) return nullptr;
    return &hnsw->hvecs[(size_t)it->second * hnsw->dim];
}

void vdb_segment::del(uint64_t id) {                                                // segment.rs:71-99
    if (deleted.count(id) || !hnsw->slot_of.count(id)) return;
    auto pit = payloads.find(id);
    if (pit != payloads.end()) payload_index.remove(id, pit->second);
    deleted.insert(id);
    hnsw->mark_deleted(id);
    ++payload_version;
    const size_t deleted_count = deleted.size(), total = hnsw->len();
    if (deleted_count >= 100 && ((float)deleted_count / (float)total) >= 0.25f) purge();
}

void vdb_segment::purge() {                                                         // segment.rs:166-217
    auto nh = std::make_unique<vdb_index>(hnsw->metric, hnsw->m, hnsw->ef, hnsw->max_level_cap, hnsw->dim, hnsw->opts);
    nh->level_draws = hnsw->level_draws; nh->sample_draws = hnsw->sample_draws;      // the counter RNG continues
    vdb_payload_index npi; std::unordered_map<uint64_t, vdb_payload> npl;
    std::map<uint64_t, uint32_t> ordered(hnsw->slot_of.begin(), hnsw->slot_of.end());   // DET-6
    for (auto& kv : ordered) {
        const uint64_t id = kv.first;
        if (deleted.count(id)) continue;
        std::vector<float> vec(hnsw->hvecs.begin() + (size_t)kv.second * hnsw->dim, hnsw->hvecs.begin() + (size_t)(kv.second + 1) * hnsw->dim);
        nh->insert(id, vec.data(), vec.size());
        auto pit = payloads.find(id);
        if (pit != payloads.end()) {
            npi.insert(id, pit->second);
            npl[id] = pit->second;
            filter_aware_edges(*nh, npi, npl, id, pit->second, pit->second.scalar_keys());
        }
    }
    hnsw = std::move(nh); payload_index = std::move(npi); payloads = std::move(npl);
    deleted.clear();
    rebuild_columns();
    { std::lock_guard<std::mutex> lk(filter_mu); filter_cache.clear(); }
    ++payload_version;
}

std::vector<vdb_scored_point> vdb_segment::search(const float* q, size_t len, size_t top_k) const {   // segment.rs:103-119
    if (live_count() == 0) throw Failure(err_search("No active points available to search."));
    std::vector<vdb_scored_point> c = one_query(*hnsw, q, len, top_k * 2, {});
    std::vector<vdb_scored_point> out;
    for (auto& sp : c) { if (out.size() >= top_k) break; if (!deleted.count(sp.id)) out.push_back(sp); }
    return out;
}

std::optional<uint64_t> vdb_segment::find_entry_point_matching_filter(const vdb_filter& f) const {     // hnsw.rs:558-583
    switch (f.kind) {
        case vdb_filter::Match: {
            const std::set<uint64_t>* s = payload_index.query_exact(f.key, f.value);
            if (!s) return std::nullopt;
            for (uint64_t id : *s) {                                                // DET-4
                auto it = hnsw->slot_of.find(id);
                if (it != hnsw->slot_of.end() && !hnsw->deleted[it->second]) return id;
            }
            return std::nullopt;
        }
        case vdb_filter::And: case vdb_filter::Or:
            for (auto& k : f.kids) if (auto r = find_entry_point_matching_filter(k)) return r;
            return std::nullopt;
        case vdb_filter::Not: return find_entry_point_matching_filter(f.kids.at(0));
        default: return std::nullopt;
    }
}

// Masked mode: when at most this fraction of the rows passes, the passing rows are scanned exactly
// instead of walking the graph (a beam of fixed width meets too few passing rows to fill top_k).
static bool masked_scan_pays(size_t n_pass, size_t n) {
    const char* e = std::getenv("VDB_MASKED_SCAN");               // 0: never, 1: always
    if (e) return std::atoi(e) != 0;
    return n_pass <= 4096 || n_pass * 8 <= n;
}

// ------------------------------------------------------------------------------------------------
// Payload columns: what the device-side filter compiler reads
// ------------------------------------------------------------------------------------------------
void vdb_segment::note_payload(uint32_t slot, const vdb_payload& p) {
    if (has_bits.size() <= slot / 32) has_bits.resize(slot / 32 + 1, 0);
    has_bits[slot / 32] |= 1u << (slot & 31); has_dirty = true;
    for (const auto& f : p.fields) {
        PayloadColumn& c = columns[f.first];
        if (c.kind.size() <= slot) { c.kind.resize(slot + 1, 0); c.bits.resize(slot + 1, 0); }
        c.kind[slot] = (uint8_t)(1 + f.second.kind());
        int64_t b = 0;
        switch (f.second.kind()) {
            case VDB_T_INT: b = std::get<int64_t>(f.second.d); break;
            case VDB_T_FLOAT: { const double d = std::get<OF64>(f.second.d).v; std::memcpy(&b, &d, 8); break; }
            case VDB_T_BOOL: b = std::get<Bool8>(f.second.d).v != 0; break;
            case VDB_T_STR: {
                const std::string& str = std::get<std::string>(f.second.d);
                auto it = c.str_code.find(str);
                if (it == c.str_code.end()) { it = c.str_code.emplace(str, (uint32_t)c.strs.size()).first; c.strs.push_back(str); }
                b = it->second; break;
            }
            default: c.list_slots.push_back(slot);
        }
        c.bits[slot] = b;
        if (slot < c.on_device) c.on_device = slot;              // (a slot is written once; kept for safety)
    }
}

void vdb_segment::rebuild_columns() {                              // after a purge: every point has a new slot
    columns.clear(); has_bits.clear(); has_dirty = true;
    for (const auto& kv : payloads) {
        auto it = hnsw->slot_of.find(kv.first);
        if (it != hnsw->slot_of.end()) note_payload(it->second, kv.second);
    }
    for (auto& c : columns) std::sort(c.second.list_slots.begin(), c.second.list_slots.end());
}

namespace {
struct NodeBits { DevBuf<uint32_t> p, e; };
}

// Filter -> device bitmaps over the slots: pass (the filter is true) and err (evaluate_filter returned Err); a slot
// without a payload has neither bit (hnsw.rs:669 map_or(false), :391 continue).  Match on a scalar goes through the
// inverted index when its posting list is short (stores.rs:55-61: the list is scattered into the bitmap) and through the
// key's column otherwise; Compare scans the column; And / Or / Not combine bitmaps word-wise.
std::shared_ptr<const FilterBits> vdb_segment::compile(const vdb_filter& f) const {
    std::string key; f.serialise(key);
    const size_t n = hnsw->len();
    // one compilation at a time: two threads that ask for the same new filter get the same object
    std::lock_guard<std::mutex> lk(filter_mu);
    auto it = filter_cache.find(key);
    if (it != filter_cache.end() && it->second->version == payload_version && it->second->n_slots == n) {
        it->second->last_used = ++filter_clock;
        return it->second;
    }
    auto fresh = std::make_shared<FilterBits>();
    FilterBits& fb = *fresh;
    fb.n_slots = n;
    const uint32_t words = (uint32_t)((n + 31) / 32), w1 = std::max<uint32_t>(words, 1);
    {
        vdb_index& h = *hnsw;
        std::lock_guard<std::mutex> dl(h.mu);                     // the index's stream is shared with its searches
        h.use_device();
        h.sync_device();                                          // the tombstone bitmap the counts are taken against
        hipStream_t st = h.stream;
        // slots that own a payload
        has_bits.resize(w1, 0);
        if (has_dirty || has_words_on_device != w1) {
            d_has.reserve(w1, st, false);
            VDB_HIP(hipMemcpyAsync(d_has.p, has_bits.data(), (size_t)w1 * 4, hipMemcpyHostToDevice, st));
            VDB_HIP(hipStreamSynchronize(st));
            has_dirty = false; has_words_on_device = w1;
        }
        std::vector<std::unique_ptr<DevBuf<uint8_t>>> tables; std::vector<std::unique_ptr<DevBuf<uint32_t>>> lists;   // alive until the stream has run
        std::vector<std::vector<uint8_t>> host_keep; std::vector<std::vector<uint32_t>> host_keep32;
        auto column = [&](const std::string& k) -> PayloadColumn* {                   // device copy of a key's column, current up to slot n
            auto ci = columns.find(k);
            if (ci == columns.end()) return nullptr;
            PayloadColumn& c = ci->second;
            if (c.kind.size() < n) { c.kind.resize(n, 0); c.bits.resize(n, 0); }
            c.d_kind.reserve(std::max<size_t>(n, 1), st, true); c.d_bits.reserve(std::max<size_t>(n, 1), st, true);
            if (c.on_device < n) {
                VDB_HIP(hipMemcpyAsync(c.d_kind.p + c.on_device, c.kind.data() + c.on_device, n - c.on_device, hipMemcpyHostToDevice, st));
                VDB_HIP(hipMemcpyAsync(c.d_bits.p + c.on_device, c.bits.data() + c.on_device, (n - c.on_device) * 8, hipMemcpyHostToDevice, st));
                VDB_HIP(hipStreamSynchronize(st));                 // the host vectors may grow before the copy has run otherwise
                c.on_device = n;
            }
            return &c;
        };
        auto upload8 = [&](std::vector<uint8_t>&& v) -> const uint8_t* {
            tables.emplace_back(new DevBuf<uint8_t>()); host_keep.emplace_back(std::move(v));
            tables.back()->reserve(std::max<size_t>(host_keep.back().size(), 1), st, false);
            if (!host_keep.back().empty()) VDB_HIP(hipMemcpyAsync(tables.back()->p, host_keep.back().data(), host_keep.back().size(), hipMemcpyHostToDevice, st));
            return tables.back()->p;
        };
        auto upload32 = [&](std::vector<uint32_t>&& v) -> const uint32_t* {
            lists.emplace_back(new DevBuf<uint32_t>()); host_keep32.emplace_back(std::move(v));
            lists.back()->reserve(std::max<size_t>(host_keep32.back().size(), 1), st, false);
            if (!host_keep32.back().empty()) VDB_HIP(hipMemcpyAsync(lists.back()->p, host_keep32.back().data(), host_keep32.back().size() * 4, hipMemcpyHostToDevice, st));
            return lists.back()->p;
        };
        const char* force = std::getenv("VDB_FILTER_MATCH");      // tests: "postings" / "column" pin the path of Match leaves
        std::function<void(const vdb_filter&, NodeBits&)> eval = [&](const vdb_filter& node, NodeBits& out) {
            out.p.reserve(w1, st, false); out.e.reserve(w1, st, false);
            switch (node.kind) {
                case vdb_filter::Match: case vdb_filter::Compare: {
                    const bool is_match = node.kind == vdb_filter::Match;
                    // Match on a scalar with a short posting list: the inverted index names the passing points
                    if (is_match && node.value.scalar()) {
                        const std::set<uint64_t>* ids = payload_index.query_exact(node.key, node.value);
                        const size_t len = ids ? ids->size() : 0;
                        bool postings = len <= std::max<size_t>(4096, n / 64);
                        if (force) postings = std::string(force) == "postings";
                        if (postings) {
                            VDB_HIP(hipMemsetAsync(out.p.p, 0, (size_t)w1 * 4, st)); VDB_HIP(hipMemsetAsync(out.e.p, 0, (size_t)w1 * 4, st));
                            std::vector<uint32_t> slots; slots.reserve(len);
                            if (ids) for (uint64_t id : *ids) { auto si = h.slot_of.find(id); if (si != h.slot_of.end() && (has_bits[si->second / 32] >> (si->second & 31) & 1u)) slots.push_back(si->second); }
                            const uint32_t m = (uint32_t)slots.size();
                            launch_filter_scatter(upload32(std::move(slots)), m, out.p.p, st);
                            return;
                        }
                    }
                    PayloadColumn* c = column(node.key);
                    FilterLeafArgs a{};
                    a.kind = c ? c->d_kind.p : nullptr; a.bits = c ? c->d_bits.p : nullptr; a.n_col = c ? (uint32_t)n : 0;
                    a.n = (uint32_t)n; a.words = words; a.leaf = is_match ? FILTER_LEAF_MATCH : FILTER_LEAF_COMPARE; a.op = (int)node.op;
                    a.want_kind = node.value.kind(); a.has = d_has.p; a.pass = out.p.p; a.err = out.e.p;
                    switch (node.value.kind()) {
                        case VDB_T_INT: a.want_bits = std::get<int64_t>(node.value.d); break;
                        case VDB_T_FLOAT: { const double d = std::get<OF64>(node.value.d).v; std::memcpy(&a.want_bits, &d, 8); break; }
                        case VDB_T_BOOL: a.want_bits = std::get<Bool8>(node.value.d).v != 0; break;
                        case VDB_T_STR: {                           // one comparison per distinct string of the key
                            const std::string& want = std::get<std::string>(node.value.d);
                            std::vector<uint8_t> table(c ? c->strs.size() : 0);
                            for (size_t i = 0; i < table.size(); ++i) table[i] = is_match ? c->strs[i] == want : cmp_apply(node.op, c->strs[i], want);
                            a.table_n = (uint32_t)table.size(); a.table = upload8(std::move(table)); break;
                        }
                        default: break;
                    }
                    launch_filter_leaf(a, st);
                    // list-valued fields: the few leaves that can hold for a list are evaluated on the host for those slots
                    const bool list_leaf = is_match ? node.value.kind() >= VDB_T_LIST_INT
                                                    : (node.value.kind() == VDB_T_LIST_STR || node.value.kind() == VDB_T_STR);
                    if (c && list_leaf && !c->list_slots.empty()) {
                        std::vector<uint32_t> slots; std::vector<uint8_t> tri;
                        for (uint32_t s : c->list_slots) {
                            if (s >= n) continue;
                            auto pit = payloads.find(h.ids[s]);
                            if (pit == payloads.end()) continue;
                            const vdb_value* have = pit->second.get(node.key);
                            if (!have || (!is_match && have->kind() != VDB_T_LIST_STR)) continue;
                            const Tri t = node.eval(pit->second);
                            slots.push_back(s); tri.push_back((uint8_t)t.s);
                        }
                        const uint32_t m = (uint32_t)slots.size();
                        launch_filter_patch(upload32(std::move(slots)), upload8(std::move(tri)), m, out.p.p, out.e.p, st);
                    }
                    return;
                }
                case vdb_filter::And: case vdb_filter::Or: {
                    const bool is_and = node.kind == vdb_filter::And;
                    if (node.kids.empty()) { launch_filter_combine(is_and ? FILTER_TRUE : FILTER_FALSE, out.p.p, out.e.p, nullptr, nullptr, d_has.p, w1, st); return; }
                    eval(node.kids[0], out);
                    for (size_t k = 1; k < node.kids.size(); ++k) {
                        NodeBits rhs; eval(node.kids[k], rhs);
                        launch_filter_combine(is_and ? FILTER_AND : FILTER_OR, out.p.p, out.e.p, rhs.p.p, rhs.e.p, d_has.p, w1, st);
                        VDB_HIP(hipStreamSynchronize(st));         // rhs is freed when it leaves scope
                    }
                    return;
                }
                default:
                    eval(node.kids.at(0), out);
                    launch_filter_combine(FILTER_NOT, out.p.p, out.e.p, nullptr, nullptr, d_has.p, w1, st);
            }
        };
        NodeBits root;
        if (words == 0) { root.p.reserve(1, st, false); root.e.reserve(1, st, false); VDB_HIP(hipMemsetAsync(root.p.p, 0, 4, st)); VDB_HIP(hipMemsetAsync(root.e.p, 0, 4, st)); }
        else eval(f, root);
        DevBuf<uint32_t> counts; counts.reserve(2, st, false);
        VDB_HIP(hipMemsetAsync(counts.p, 0, 8, st));
        const uint32_t* dele = h.n_deleted ? h.d_deleted.p : nullptr;
        launch_filter_count(root.p.p, root.e.p, dele, (uint32_t)n, counts.p, st);
        uint32_t hc[2] = {0, 0};
        VDB_HIP(hipMemcpyAsync(hc, counts.p, 8, hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
        fb.n_pass = hc[0]; fb.n_err = hc[1];
        std::swap(fb.d_pass.p, root.p.p); std::swap(fb.d_pass.cap, root.p.cap);
        std::swap(fb.d_err.p, root.e.p); std::swap(fb.d_err.cap, root.e.cap);
        if (masked_scan_pays(fb.n_pass, n) && fb.n_pass) {         // the narrow-filter scan reads the passing rows through this list
            DevBuf<uint32_t> sums; sums.reserve(std::max<uint32_t>(filter_compact_blocks((uint32_t)n), 1), st, false);
            fb.d_rows.reserve(fb.n_pass, st, false);
            launch_filter_compact(fb.d_pass.p, dele, (uint32_t)n, sums.p, fb.d_rows.p, st);
            VDB_HIP(hipStreamSynchronize(st));
            fb.have_rows = true;
        }
    }
    fb.version = payload_version;
    fb.last_used = ++filter_clock;
    if (it != filter_cache.end()) it->second = fresh;          // a stale entry: searches still using it keep their reference
    else {
        while (filter_cache.size() >= FILTER_CACHE_MAX) {      // drop the least recently used entry
            auto victim = filter_cache.begin();
            for (auto c = filter_cache.begin(); c != filter_cache.end(); ++c) if (c->second->last_used < victim->second->last_used) victim = c;
            filter_cache.erase(victim);
        }
        filter_cache.emplace(std::move(key), fresh);
    }
    return fresh;
}

void vdb_segment::filter_bitmaps(const vdb_filter& f, uint32_t* pass, uint32_t* err) const {
    const auto b = compile(f);
    const size_t words = (b->n_slots + 31) / 32;
    if (!words) return;
    std::lock_guard<std::mutex> dl(hnsw->mu);
    hnsw->use_device();
    VDB_HIP(hipMemcpyAsync(pass, b->d_pass.p, words * 4, hipMemcpyDeviceToHost, hnsw->stream));
    VDB_HIP(hipMemcpyAsync(err, b->d_err.p, words * 4, hipMemcpyDeviceToHost, hnsw->stream));
    VDB_HIP(hipStreamSynchronize(hnsw->stream));
}

void vdb_segment::filtered_batch(const float* q, size_t nq, size_t len, size_t top_k, size_t useful_k, size_t ef_search, const vdb_filter* f,
                                 vdb_scored_point* out, uint32_t* out_n) const {
    vdb_index& h = *hnsw;
    if (len != h.dim) throw Failure(err_length_mismatch(h.dim, len));               // hnsw.rs:593-598
    vdb_index::FilterArgs fa;
    const std::shared_ptr<const FilterBits> bits = f ? compile(*f) : nullptr;     // alive until this call returns
    fa.bits = bits.get();
    if (h.opts.filter_mode == VDB_FILTER_MASKED) {
        if (f && (bits->have_rows || bits->n_pass == 0) && masked_scan_pays(bits->n_pass, h.len())) {
            // the scan returns no tombstone, so the caller's over-fetch is not needed; rows of out stay top_k apart
            const size_t k = std::min(top_k, useful_k);
            if (bits->n_pass == 0) { std::memset(out_n, 0, nq * sizeof(uint32_t)); return; }
            if (k == top_k) { h.exact_knn(q, nq, k, false, out, out_n, nullptr, bits->d_rows.p, bits->n_pass); return; }
            std::vector<vdb_scored_point> tmp(nq * k);
            h.exact_knn(q, nq, k, false, tmp.data(), out_n, nullptr, bits->d_rows.p, bits->n_pass);
            for (size_t i = 0; i < nq; ++i) std::memcpy(out + i * top_k, tmp.data() + i * k, out_n[i] * sizeof(vdb_scored_point));
            return;
        }
        fa.mode = f ? MODE_MASKED : MODE_SEARCH;
        h.search_batch(q, nq, top_k, ef_search, fa, out, out_n, nullptr);
        return;
    }
    // ---- literal: entry selection on the host (hnsw.rs:600-626; no distances involved) ----
    fa.mode = MODE_LITERAL;
    if (h.entry_point) {
        uint64_t id = *h.entry_point;
        if (f) {
            bool keep = false;
            auto pit = payloads.find(id);
            if (pit != payloads.end()) {
                const Tri t = f->eval(pit->second);
                if (t.s == Tri::Err) throw Failure(err_invalid_payload(t.msg));
                keep = t.s == Tri::True;
            }
            if (!keep) { if (auto alt = find_entry_point_matching_filter(*f)) id = *alt; }
        }
        fa.entry_slot = h.slot_of.at(id);
    } else {
        std::optional<uint64_t> alt = f ? find_entry_point_matching_filter(*f) : std::nullopt;
        if (!alt) { std::memset(out_n, 0, nq * sizeof(uint32_t)); return; }
        fa.entry_slot = h.slot_of.at(*alt);
    }
    if (h.deleted[fa.entry_slot]) throw Failure(err_panic("called `Option::unwrap()` on a `None` value"));   // hnsw.rs:636
    try {
        h.search_batch(q, nq, top_k, ef_search, fa, out, out_n, nullptr);
    } catch (const Failure& e) {
        if (e.e.code == VDB_ERR_INVALID_PAYLOAD && e.e.msg.rfind("@slot:", 0) == 0 && f) {
            const uint32_t s = (uint32_t)std::stoul(e.e.msg.substr(6));
            auto pit = payloads.find(h.ids.at(s));
            const Tri t = pit != payloads.end() ? f->eval(pit->second) : Tri::err("filter evaluation failed");
            throw Failure(err_invalid_payload(t.msg));
        }
        throw;
    }
}

std::vector<vdb_scored_point> vdb_segment::in_place_filtered_search(const float* q, size_t len, size_t top_k, const vdb_filter* f) const {
    std::vector<vdb_scored_point> out(std::max<size_t>(top_k, 1));
    uint32_t n = 0;
    if (top_k == 0) return {};
    filtered_batch(q, 1, len, top_k, top_k, 0, f, out.data(), &n);
    out.resize(n);
    return out;
}

std::vector<vdb_scored_point> vdb_segment::search_with_filter(const float* q, size_t len, size_t top_k, const vdb_filter* f) const {   // segment.rs:121-147
    if (live_count() == 0) throw Failure(err_search("No active points available to search."));
    std::vector<vdb_scored_point> c = in_place_filtered_search(q, len, top_k * 2, f);
    std::vector<vdb_scored_point> out;
    for (auto& sp : c) { if (out.size() >= top_k) break; if (!deleted.count(sp.id)) out.push_back(sp); }
    return out;
}

std::vector<vdb_scored_point> vdb_segment::post_filter(const float* q, size_t len, size_t top_k, const vdb_filter* f) const {          // segment.rs:221-249
    if (live_count() == 0) throw Failure(err_search("No active points available to search."));
    std::vector<vdb_scored_point> c = one_query(*hnsw, q, len, top_k * 4, {});
    std::vector<vdb_scored_point> out;
    for (auto& sp : c) {
        if (out.size() >= top_k) break;
        if (deleted.count(sp.id)) continue;
        if (f) {
            auto pit = payloads.find(sp.id);
            if (pit == payloads.end()) continue;
            if (f->eval(pit->second).s != Tri::True) continue;                      // Err counts as false (segment.rs:241)
        }
        out.push_back(sp);
    }
    return out;
}

// Batched Segment::search / search_with_filter: over-fetch 2 x top_k, drop tombstones, keep top_k.
void vdb_segment::search_batch(const float* q, size_t nq, size_t len, size_t top_k, size_t ef_search, const vdb_filter* f,
                               vdb_scored_point* out, uint32_t* out_n) const {
    if (live_count() == 0) throw Failure(err_search("No active points available to search."));
    if (len != hnsw->dim) throw Failure(err_length_mismatch(hnsw->dim, len));
    const size_t k2 = top_k * 2;
    std::vector<vdb_scored_point> tmp(nq * std::max<size_t>(k2, 1)); std::vector<uint32_t> cnt(nq, 0);
    if (k2) {
        if (f) filtered_batch(q, nq, len, k2, top_k, ef_search, f, tmp.data(), cnt.data());
        else hnsw->search_batch(q, nq, k2, ef_search, {}, tmp.data(), cnt.data(), nullptr);
    }
    for (size_t i = 0; i < nq; ++i) {
        uint32_t n = 0;
        for (uint32_t j = 0; j < cnt[i] && n < top_k; ++j) {
            const vdb_scored_point& sp = tmp[i * k2 + j];
            if (!deleted.count(sp.id)) out[i * top_k + n++] = sp;
        }
        out_n[i] = n;
    }
}

// Batched Segment::search / search_with_filter on device buffers.  The graph walk never returns a tombstone (hnsw.rs:475)
// and the scan reads live rows only, so the over-fetch of segment.rs:110 / :134 changes nothing here and is left out.
void vdb_segment::search_batch_dev(const float* d_q, size_t nq, size_t top_k, size_t ef_search, const vdb_filter* f,
                                   vdb_scored_point* d_out, uint32_t* d_out_n, vdb_query_stats* d_stats, hipStream_t s) const {
    if (live_count() == 0) throw Failure(err_search("No active points available to search."));
    vdb_index& h = *hnsw;
    if (!f) { h.search_batch_dev(d_q, nq, top_k, ef_search, d_out, d_out_n, d_stats, s); return; }
    if (h.opts.filter_mode != VDB_FILTER_MASKED)
        throw Failure(err_arg("filtered search on device buffers needs filter_mode = VDB_FILTER_MASKED (the literal walk picks its entry and reports errors on the host)"));
    const std::shared_ptr<const FilterBits> bits = compile(*f);
    if (bits->n_pass == 0) { h.use_device(); VDB_HIP(hipMemsetAsync(d_out_n, 0, nq * sizeof(uint32_t), s ? s : h.stream)); return; }
    if (bits->have_rows && masked_scan_pays(bits->n_pass, h.len())) {
        h.exact_knn(d_q, nq, top_k, true, d_out, d_out_n, s, bits->d_rows.p, bits->n_pass);
        if (d_stats) VDB_HIP(hipMemsetAsync(d_stats, 0, nq * sizeof(vdb_query_stats), s ? s : h.stream));    // a scan walks no graph
    }
    else {
        vdb_index::FilterArgs fa; fa.mode = MODE_MASKED; fa.bits = bits.get();
        h.search_batch_dev(d_q, nq, top_k, ef_search, d_out, d_out_n, d_stats, s, &fa);
    }
    hold_until_done(bits, s ? s : h.stream);         // the bitmaps must outlive the launches
}

void vdb_segment::hold_until_done(std::shared_ptr<const FilterBits> bits, hipStream_t s) const {
    hnsw->use_device();
    std::lock_guard<std::mutex> lk(filter_mu);
    for (auto it = in_flight.begin(); it != in_flight.end();) {      // launches of earlier calls that have finished, on any stream
        std::vector<InFlight>& l = it->second;
        size_t kept = 0;
        for (size_t i = 0; i < l.size(); ++i) {
            if (hipEventQuery(l[i].done) == hipSuccess) spare_events.push_back(l[i].done);
            else { if (kept != i) l[kept] = std::move(l[i]); ++kept; }
        }
        l.resize(kept);
        it = l.empty() && it->first != s ? in_flight.erase(it) : std::next(it);
    }
    std::vector<InFlight>& q = in_flight[s];
    if (!q.empty() && q.back().bits == bits) {        // the same bitmaps again: move their event behind the new launches
        VDB_HIP(hipEventRecord(q.back().done, s));
        return;
    }
    hipEvent_t ev;
    if (!spare_events.empty()) { ev = spare_events.back(); spare_events.pop_back(); }
    else VDB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (hipError_t e = hipEventRecord(ev, s); e != hipSuccess) { spare_events.push_back(ev); VDB_HIP(e); }
    q.push_back(InFlight{std::move(bits), ev});
}

vdb_segment::~vdb_segment() {
    if (in_flight.empty() && spare_events.empty()) return;
    (void)hipSetDevice(hnsw->opts.device);
    (void)hipDeviceSynchronize();
    for (auto& [st, q] : in_flight) for (InFlight& f : q) (void)hipEventDestroy(f.done);
    for (hipEvent_t e : spare_events) (void)hipEventDestroy(e);
}
