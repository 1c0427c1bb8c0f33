// Host side of the HNSW index: the literal graph the reference keeps (lists with self links and
// duplicates), the linking rules of hnsw.rs:119-208, and the device-resident packed copy every
// distance computation runs against.
#include "index.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace vdb;

namespace {
constexpr size_t WAVE_LANES = 64;
thread_local std::string g_last_error;
}
void vdb::set_last_error(const std::string& s) { g_last_error = s; }
extern "C" const char* vdb_last_error_message(void) { return g_last_error.c_str(); }

// ------------------------------------------------------------------------------------------------
vdb_index::vdb_index(vdb_metric metric_, size_t m_, size_t ef_, size_t cap_, size_t dim_, const vdb_options& o)
    : metric(metric_), m(m_), ef(ef_), max_level_cap(cap_), dim(dim_), level_scale(1.0 / std::log((double)m_)), opts(o) {   // hnsw.rs:77-94
    if (dim == 0 || dim > MAX_DIM) throw Failure(err_arg("dim must be in 1..2048"));
    if (ef == 0 || ef > MAX_EF) throw Failure(err_arg("ef must be in 1..1024"));
    if (m == 0) throw Failure(err_arg("m must be positive"));
    stride = (uint32_t)((dim + 3) & ~size_t(3));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Failure(err_device("no HIP device available; libvdbmi has no CPU path"));
    if (opts.device < 0 || opts.device >= ndev) throw Failure(err_device("device ordinal out of range"));
    use_device();
    VDB_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
}

vdb_index::~vdb_index() {
    if (stream) { (void)hipSetDevice(opts.device); (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
    if (h_pin) (void)hipHostFree(h_pin);
}

vdb_index::StreamScratch& vdb_index::stream_scratch(hipStream_t s) const {
    auto it = s_by_stream.find(s);
    if (it == s_by_stream.end()) {
        if (s_by_stream.size() >= STREAM_SCRATCH_MAX) {
            auto victim = s_by_stream.begin();
            for (auto c = s_by_stream.begin(); c != s_by_stream.end(); ++c) if (c->second.last_used < victim->second.last_used) victim = c;
            VDB_HIP(hipDeviceSynchronize());          // kernels of the victim's stream may still use its buffers
            s_by_stream.erase(victim);
        }
        it = s_by_stream.try_emplace(s).first;
    }
    it->second.last_used = ++s_stream_clock;
    return it->second;
}

void* vdb_index::pinned(size_t bytes) const {
    if (bytes > h_pin_bytes) {
        if (h_pin) VDB_HIP(hipHostFree(h_pin));
        h_pin = nullptr; h_pin_bytes = 0;
        const size_t want = std::max<size_t>(bytes, 64 * 1024);
        VDB_HIP(hipHostMalloc(&h_pin, want, hipHostMallocDefault));
        h_pin_bytes = want;
    }
    return h_pin;
}

void vdb_index::use_device() const { VDB_HIP(hipSetDevice(opts.device)); }

const std::vector<uint32_t>* vdb_index::list(size_t level, uint32_t slot) const {
    if (level >= layers.size()) return nullptr;
    const HostLayer& L = layers[level];
    if (slot >= L.idx.size() || L.idx[slot] < 0) return nullptr;
    return &L.lists[L.idx[slot]];
}

std::vector<uint32_t>& vdb_index::list_or_default(size_t level, uint32_t slot) {
    if (layers.size() <= level) layers.resize(level + 1);
    HostLayer& L = layers[level];
    if (L.idx.size() <= slot) L.idx.resize(std::max<size_t>(slot + 1, ids.size()), -1);
    if (L.idx[slot] < 0) { L.idx[slot] = (int32_t)L.lists.size(); L.lists.emplace_back(); L.owner.push_back(slot); }
    graph_dirty = true;
    if (level == 0) {                       // the device copy of layer 0 is patched list by list
        if (dirty0_mark.size() <= slot) dirty0_mark.resize(std::max<size_t>(slot + 1, ids.size()), 0);
        if (!dirty0_mark[slot]) { dirty0_mark[slot] = 1; dirty0.push_back(slot); }
    } else {
        upper_dirty = true;
        if (dirty_upper.size() < (size_t(1) << 22)) dirty_upper.push_back((uint64_t)level << 32 | slot); else dirty_upper_lost = true;
    }
    return L.lists[L.idx[slot]];
}

std::optional<uint32_t> vdb_index::first_live_slot() const {       // DET-1
    if (live.empty()) return std::nullopt;
    return slot_of.at(*live.begin());
}

size_t vdb_index::draw_level() {                                    // hnsw.rs:96-101
    const uint64_t h = rng_draw(opts.seed, 1, level_draws++);
    const double r = (double)(h >> 11) * (1.0 / 9007199254740992.0);
    const double x = std::floor(-std::log(r) * level_scale);
    size_t l;
    if (!(x == x) || x <= 0) l = 0;
    else if (x >= 1.8446744073709552e19) l = SIZE_MAX;
    else l = (size_t)x;
    return std::min(l, max_level_cap);
}

// Appends rows to the host mirror and the device (normalised on the device for Cosine,
// hnsw.rs:138-139), draws levels and creates the self links of hnsw.rs:143-148.
uint32_t vdb_index::append_rows(const uint64_t* pids, const float* v, size_t n) {
    use_device();
    const uint32_t first = (uint32_t)ids.size();
    const size_t total = ids.size() + n;
    if (total >= 0x7FFFFFFFull) throw Failure(err_arg("more than 2^31 points in one index"));
    d_vecs.reserve(total * stride, stream); d_ids.reserve(total, stream);
    std::vector<float> padded(n * stride, 0.f);
    for (size_t i = 0; i < n; ++i) std::memcpy(&padded[i * stride], v + i * dim, dim * sizeof(float));
    VDB_HIP(hipMemcpyAsync(d_vecs.p + (size_t)first * stride, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    VDB_HIP(hipMemcpyAsync(d_ids.p + first, pids, n * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    if (metric == VDB_COSINE) {
        launch_normalize_rows(d_vecs.p, stride, (uint32_t)dim, first, (uint32_t)n, stream);
        VDB_HIP(hipMemcpyAsync(padded.data(), d_vecs.p + (size_t)first * stride, padded.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    }
    VDB_HIP(hipStreamSynchronize(stream));
    hvecs.resize(total * dim);
    for (size_t i = 0; i < n; ++i) std::memcpy(&hvecs[(first + i) * dim], &padded[i * stride], dim * sizeof(float));
    for (size_t i = 0; i < n; ++i) {
        const uint32_t s = first + (uint32_t)i;
        ids.push_back(pids[i]); slot_of[pids[i]] = s;
        if (pending_deleted.erase(pids[i])) { deleted.push_back(1); ++n_deleted; }       // marked deleted before it existed (hnsw.rs:112)
        else { deleted.push_back(0); live.insert(pids[i]); }
        const size_t level = draw_level();
        levels.push_back((uint32_t)level);
        for (size_t l = 0; l <= level; ++l) list_or_default(l, s).push_back(s);
    }
    rows_on_device = total;
    graph_dirty = true; deleted_dirty = true;
    return first;
}

std::vector<float> vdb_index::maybe_normalize(const float* v) {     // hnsw.rs:752-764
    std::vector<float> out(v, v + dim);
    if (metric != VDB_COSINE) return out;
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    s_q.reserve(stride, stream, false);
    std::vector<float> padded(stride, 0.f);
    std::memcpy(padded.data(), v, dim * sizeof(float));
    VDB_HIP(hipMemcpyAsync(s_q.p, padded.data(), stride * sizeof(float), hipMemcpyHostToDevice, stream));
    launch_normalize_rows(s_q.p, stride, (uint32_t)dim, 0, 1, stream);
    VDB_HIP(hipMemcpyAsync(padded.data(), s_q.p, stride * sizeof(float), hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    std::memcpy(out.data(), padded.data(), dim * sizeof(float));
    return out;
}

// ------------------------------------------------------------------------------------------------
// Device copy of the graph: rows are already resident; lists are packed with self loops and
// duplicate ids removed (the visited set makes both invisible to every search, hnsw.rs:475).
void vdb_index::clean_list(const std::vector<uint32_t>& src, uint32_t self, std::vector<uint32_t>& out) {
    // drop self loops and repeats, keep order
    if (clean_stamp.size() < ids.size()) clean_stamp.resize(ids.size(), 0);
    if (++clean_epoch == 0) { std::fill(clean_stamp.begin(), clean_stamp.end(), 0); clean_epoch = 1; }
    out.clear();
    for (uint32_t x : src) { if (x == self || clean_stamp[x] == clean_epoch) continue; clean_stamp[x] = clean_epoch; out.push_back(x); }
}

// One layer-0 list as its device block; neighbours beyond the 62nd go to the end of `ovf`.
void vdb_index::fill_block(uint32_t* blk, const std::vector<uint32_t>& lst, std::vector<uint32_t>& ovf, size_t ovf_base) {
    std::fill(blk, blk + ADJ_BLOCK, 0u);
    blk[0] = (uint32_t)lst.size();
    const size_t inl = std::min<size_t>(lst.size(), ADJ_INLINE);
    std::copy(lst.begin(), lst.begin() + inl, blk + 1);
    if (lst.size() > ADJ_INLINE) { blk[63] = (uint32_t)(ovf_base + ovf.size()); ovf.insert(ovf.end(), lst.begin() + ADJ_INLINE, lst.end()); }
}

// Layer 0 from scratch.
void vdb_index::pack_layer0_full() {
    const size_t n = ids.size();
    std::vector<uint32_t> adj0(n * ADJ_BLOCK, 0), ovf, tmp;
    h_deg0.assign(n, 0);
    delta_slot.clear(); delta_pos.clear(); delta_val.clear();      // the lists already hold the appended entries
    size_t edges = 0;
    if (!layers.empty()) {
        const HostLayer& L0 = layers[0];
        for (size_t k = 0; k < L0.lists.size(); ++k) {
            const uint32_t s = L0.owner[k];
            clean_list(L0.lists[k], s, tmp);
            fill_block(&adj0[(size_t)s * ADJ_BLOCK], tmp, ovf, 0);
            h_deg0[s] = (uint32_t)tmp.size(); edges += tmp.size();
        }
    }
    l0_edges = edges; ovf_used = ovf.size(); ovf_live = ovf.size();
    const size_t n1 = std::max<size_t>(n, 1);
    d_adj0.reserve(n1 * ADJ_BLOCK, stream, false); d_ovf.reserve(std::max<size_t>(ovf.size(), 1), stream, false);
    if (n) VDB_HIP(hipMemcpyAsync(d_adj0.p, adj0.data(), adj0.size() * 4, hipMemcpyHostToDevice, stream));
    if (!ovf.empty()) VDB_HIP(hipMemcpyAsync(d_ovf.p, ovf.data(), ovf.size() * 4, hipMemcpyHostToDevice, stream));
    VDB_HIP(hipStreamSynchronize(stream));
}

// Layer 0 after a few edits: only the blocks of the lists that changed are rebuilt and written over
// the device copy; a list that outgrows its block gets a fresh overflow run at the end of `ovf` (the
// old run is garbage until the next full pack).
void vdb_index::patch_layer0() {
    const size_t n = ids.size();
    d_adj0.reserve(std::max<size_t>(n, 1) * ADJ_BLOCK, stream, true);
    h_deg0.resize(n, 0);
    std::vector<uint32_t> idx, blocks(dirty0.size() * ADJ_BLOCK), ovf, tmp;
    idx.reserve(dirty0.size());
    const HostLayer* L0 = layers.empty() ? nullptr : &layers[0];
    auto l0 = [&](size_t i) -> const std::vector<uint32_t>* {
        const uint32_t s = dirty0[i];
        return (L0 && s < L0->idx.size() && L0->idx[s] >= 0) ? &L0->lists[L0->idx[s]] : nullptr;
    };
    for (size_t di = 0; di < dirty0.size(); ++di) {
        const uint32_t s = dirty0[di];
        // the lists lie at random places of the host mirror: request the header 16 ahead, the contents 8 ahead
        if (di + 16 < dirty0.size()) { if (auto* v = l0(di + 16)) __builtin_prefetch(v); }
        if (di + 8 < dirty0.size()) { if (auto* v = l0(di + 8)) if (!v->empty()) { __builtin_prefetch(v->data()); __builtin_prefetch(v->data() + v->size() - 1); } }
        const std::vector<uint32_t>* lst = list(0, s);
        if (lst) clean_list(*lst, s, tmp); else tmp.clear();
        fill_block(&blocks[idx.size() * ADJ_BLOCK], tmp, ovf, ovf_used);
        // the overflow run of a hub list is written anew at the end of `ovf`; its old run becomes dead space
        ovf_live += tmp.size() > ADJ_INLINE ? tmp.size() - ADJ_INLINE : 0;
        ovf_live -= h_deg0[s] > ADJ_INLINE ? h_deg0[s] - ADJ_INLINE : 0;
        l0_edges += tmp.size(); l0_edges -= h_deg0[s]; h_deg0[s] = (uint32_t)tmp.size();
        idx.push_back(s);
    }
    if (!ovf.empty()) {
        d_ovf.reserve(ovf_used + ovf.size(), stream, true);
        VDB_HIP(hipMemcpyAsync(d_ovf.p + ovf_used, ovf.data(), ovf.size() * 4, hipMemcpyHostToDevice, stream));
        ovf_used += ovf.size();
    }
    // one upload for both kinds of patch (page-locked when small: a single insert pays no blocking copy); appended
    // entries go first, a list that was also edited some other way is then re-packed as a whole
    const size_t nd = delta_slot.size(), words = 3 * nd + idx.size() + blocks.size();
    if (words) {
        s_delta.reserve(words, stream, false);
        const std::pair<const uint32_t*, size_t> parts[5] = {{delta_slot.data(), nd}, {delta_pos.data(), nd}, {delta_val.data(), nd},
                                                             {idx.data(), idx.size()}, {blocks.data(), blocks.size()}};
        if (words * 4 <= (size_t(256) << 10)) {
            uint32_t* h = static_cast<uint32_t*>(pinned(words * 4));
            size_t o = 0;
            for (auto& pr : parts) { std::memcpy(h + o, pr.first, pr.second * 4); o += pr.second; }
            VDB_HIP(hipMemcpyAsync(s_delta.p, h, words * 4, hipMemcpyHostToDevice, stream));
        } else {                                  // a round of the batched builder: no second host copy of its blocks
            size_t o = 0;
            for (auto& pr : parts) { if (pr.second) VDB_HIP(hipMemcpyAsync(s_delta.p + o, pr.first, pr.second * 4, hipMemcpyHostToDevice, stream)); o += pr.second; }
        }
        if (nd) launch_patch_entries(d_adj0.p, s_delta.p, s_delta.p + nd, s_delta.p + 2 * nd, (uint32_t)nd, stream);
        if (!idx.empty()) launch_patch_blocks(d_adj0.p, s_delta.p + 3 * nd, s_delta.p + 3 * nd + idx.size(), (uint32_t)idx.size(), stream);
    }
    VDB_HIP(hipStreamSynchronize(stream));
    delta_slot.clear(); delta_pos.clear(); delta_val.clear();
}

// Diagnostic (VDB_VERIFY_DEVICE=1): every layer-0 block on the device must describe the packed form of the
// host list (degree, inline neighbours, overflow run), whatever mix of block patches and appended entries built it.
void vdb_index::verify_layer0() {
    const size_t n = ids.size();
    if (n == 0) return;
    std::vector<uint32_t> dev(n * ADJ_BLOCK), ovf(std::max<size_t>(ovf_used, 1)), tmp;
    VDB_HIP(hipMemcpyAsync(dev.data(), d_adj0.p, dev.size() * 4, hipMemcpyDeviceToHost, stream));
    if (ovf_used) VDB_HIP(hipMemcpyAsync(ovf.data(), d_ovf.p, ovf_used * 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    size_t edges = 0;
    for (uint32_t s = 0; s < n; ++s) {
        const std::vector<uint32_t>* lst = list(0, s);
        if (lst) clean_list(*lst, s, tmp); else tmp.clear();
        const uint32_t* blk = &dev[(size_t)s * ADJ_BLOCK];
        bool ok = blk[0] == tmp.size() && h_deg0[s] == tmp.size();
        for (size_t i = 0; ok && i < tmp.size(); ++i)
            ok = (i < ADJ_INLINE ? blk[1 + i] : ovf[blk[63] + (i - ADJ_INLINE)]) == tmp[i];
        if (!ok) throw Failure(err_device("layer-0 block of slot " + std::to_string(s) + " differs from the host list (device degree " +
                                          std::to_string(blk[0]) + ", host " + std::to_string(tmp.size()) + ")"));
        edges += tmp.size();
    }
    if (edges != l0_edges) throw Failure(err_device("layer-0 edge count " + std::to_string(l0_edges) + " differs from the lists' " + std::to_string(edges)));
}

// Upper layers from scratch (they hold about 1/M of the points): descriptor lists for the LDS-list
// kernels and the fused blocks of the greedy descent.
void vdb_index::pack_upper() {
    const size_t n = ids.size();
    std::vector<uint32_t> pool, upper_base(n, NONE32), tmp;
    std::vector<uint64_t> udesc;
    std::vector<uint8_t> upper_cnt(n, 0);
    // a slot's descriptors for layers 1..cnt are consecutive in udesc
    for (size_t l = 1; l < layers.size(); ++l)
        for (uint32_t s : layers[l].owner) upper_cnt[s] = std::max<uint8_t>(upper_cnt[s], (uint8_t)std::min<size_t>(l, 255));
    for (uint32_t s = 0; s < n; ++s) {
        if (!upper_cnt[s]) continue;
        upper_base[s] = (uint32_t)udesc.size();
        for (size_t l = 1; l <= upper_cnt[s]; ++l) {
            const std::vector<uint32_t>* lst = list(l, s);
            if (!lst) { udesc.push_back(make_desc(0, 0)); continue; }
            clean_list(*lst, s, tmp);
            udesc.push_back(make_desc(pool.size(), (uint32_t)tmp.size()));
            pool.insert(pool.end(), tmp.begin(), tmp.end());
        }
    }
    // fused upper-layer blocks (see GraphView::ublk); skipped when the packed reference cannot hold the counts
    std::vector<uint32_t> ublk, ublk_ref;
    have_ublk = udesc.size() < (size_t(1) << UREF_BITS) && pool.size() < (size_t(1) << 32);
    for (uint32_t s = 0; s < n && have_ublk; ++s) if (upper_cnt[s] > 63) have_ublk = false;
    if (have_ublk) {
        ublk.assign(udesc.size() * ADJ_BLOCK, 0); ublk_ref.assign(udesc.size() * ADJ_BLOCK, 0);
        for (size_t d = 0; d < udesc.size(); ++d) {
            const uint64_t off = udesc[d] & 0xFFFFFFFFFFull; const uint32_t deg = (uint32_t)(udesc[d] >> 40);
            uint32_t* blk = &ublk[d * ADJ_BLOCK]; uint32_t* ref = &ublk_ref[d * ADJ_BLOCK];
            blk[0] = deg; blk[63] = (uint32_t)(off + ADJ_INLINE);
            for (uint32_t i = 0; i < std::min<uint32_t>(deg, ADJ_INLINE); ++i) {
                const uint32_t nb = pool[off + i];
                blk[1 + i] = nb; ref[1 + i] = make_uref(upper_base[nb], upper_cnt[nb]);
            }
        }
    }
    upper_edges = pool.size(); upper_words = udesc.size() * 2 + pool.size() + (have_ublk ? ublk.size() * 2 : 0);
    const size_t n1 = std::max<size_t>(n, 1);
    d_upper_base.reserve(n1, stream, false); d_upper_cnt.reserve(n1, stream, false);
    d_udesc.reserve(std::max<size_t>(udesc.size(), 1), stream, false); d_pool.reserve(std::max<size_t>(pool.size(), 1), stream, false);
    if (n) {
        VDB_HIP(hipMemcpyAsync(d_upper_base.p, upper_base.data(), n * 4, hipMemcpyHostToDevice, stream));
        VDB_HIP(hipMemcpyAsync(d_upper_cnt.p, upper_cnt.data(), n, hipMemcpyHostToDevice, stream));
    }
    if (!udesc.empty()) VDB_HIP(hipMemcpyAsync(d_udesc.p, udesc.data(), udesc.size() * 8, hipMemcpyHostToDevice, stream));
    if (!pool.empty()) VDB_HIP(hipMemcpyAsync(d_pool.p, pool.data(), pool.size() * 4, hipMemcpyHostToDevice, stream));
    if (have_ublk && !ublk.empty()) {
        d_ublk.reserve(ublk.size(), stream, false); d_ublk_ref.reserve(ublk_ref.size(), stream, false);
        VDB_HIP(hipMemcpyAsync(d_ublk.p, ublk.data(), ublk.size() * 4, hipMemcpyHostToDevice, stream));
        VDB_HIP(hipMemcpyAsync(d_ublk_ref.p, ublk_ref.data(), ublk_ref.size() * 4, hipMemcpyHostToDevice, stream));
    }
    VDB_HIP(hipStreamSynchronize(stream));
    upper_slots = n;
    pool_used = pool_live = pool.size();
    h_upper_base.swap(upper_base); h_upper_cnt.swap(upper_cnt); h_udesc.swap(udesc);
    dirty_upper.clear(); dirty_upper_lost = false;
}

// Brings the packed upper layers up to date list by list: a changed list is written anew at the end of `pool` (its old
// run becomes dead space), its descriptor and its fused block are overwritten, a new owner gets descriptors and
// blocks at the end.  Returns false when a whole re-pack is the better or the only way (many lists changed, mostly
// dead pool, an owner that gained a layer, counts beyond the packed references).
bool vdb_index::patch_upper() {
    const size_t n = ids.size();
    if (!have_ublk || dirty_upper_lost || d_upper_base.cap == 0 || h_upper_base.size() < upper_slots || upper_slots > n) return false;
    if (dirty_upper.size() > 64 * (h_udesc.size() + 64)) return false;      // a builder round: not worth sorting
    std::sort(dirty_upper.begin(), dirty_upper.end());
    dirty_upper.erase(std::unique(dirty_upper.begin(), dirty_upper.end()), dirty_upper.end());
    if (dirty_upper.size() * 8 > h_udesc.size() + 64 || pool_used > 2 * pool_live + (size_t(1) << 20)) return false;
    // new owners first: every block written below may refer to them
    std::vector<uint32_t> fresh;
    // the mirror is edited in place: every `return false` below is followed by a whole re-pack, which rebuilds it
    std::vector<uint32_t>& nbase = h_upper_base; std::vector<uint8_t>& ncnt = h_upper_cnt; std::vector<uint64_t>& ndesc = h_udesc;
    nbase.resize(n, NONE32); ncnt.resize(n, 0);
    std::vector<uint64_t> todo;
    for (uint64_t key : dirty_upper) {
        const uint32_t l = (uint32_t)(key >> 32), s = (uint32_t)key;
        if (s >= n || !list(l, s)) return false;
        if (ncnt[s] == 0) {
            uint32_t cnt = 0;
            for (size_t k = 1; k < layers.size(); ++k) if (list(k, s)) cnt = (uint32_t)k;
            if (cnt > 63) return false;
            nbase[s] = (uint32_t)ndesc.size(); ncnt[s] = (uint8_t)cnt;
            for (uint32_t k = 1; k <= cnt; ++k) { ndesc.push_back(make_desc(0, 0)); todo.push_back((uint64_t)k << 32 | s); }
            fresh.push_back(s);
        } else if (l > ncnt[s]) return false;
        else todo.push_back(key);
    }
    std::sort(todo.begin(), todo.end());
    todo.erase(std::unique(todo.begin(), todo.end()), todo.end());
    if (ndesc.size() >= (size_t(1) << UREF_BITS)) return false;
    std::vector<uint32_t> tail, didx, tmp;
    size_t live = pool_live;
    for (uint64_t key : todo) {
        const uint32_t l = (uint32_t)(key >> 32), s = (uint32_t)key;
        const uint32_t di = nbase[s] + l - 1;
        const std::vector<uint32_t>* lst = list(l, s);
        if (lst) clean_list(*lst, s, tmp); else tmp.clear();
        live += tmp.size(); live -= (size_t)(ndesc[di] >> 40);
        ndesc[di] = make_desc(pool_used + tail.size(), (uint32_t)tmp.size());
        tail.insert(tail.end(), tmp.begin(), tmp.end());
        didx.push_back(di);
    }
    if (pool_used + tail.size() >= (size_t(1) << 32)) return false;
    // staged upload: descriptor indices, blocks, reference blocks, descriptor words (index, value), owner words
    const size_t nd = didx.size(), nf = fresh.size();
    std::vector<uint32_t> st(nd + 2 * nd * ADJ_BLOCK + 4 * nd + 2 * nf, 0);
    uint32_t* blk0 = st.data() + nd; uint32_t* ref0 = blk0 + nd * ADJ_BLOCK;
    uint32_t* widx = ref0 + nd * ADJ_BLOCK; uint32_t* wval = widx + 2 * nd; uint32_t* oidx = wval + 2 * nd; uint32_t* oval = oidx + nf;
    for (size_t i = 0; i < nd; ++i) {
        const uint32_t di = didx[i];
        const uint64_t off = ndesc[di] & 0xFFFFFFFFFFull; const uint32_t deg = (uint32_t)(ndesc[di] >> 40);
        const uint32_t* run = tail.data() + (off - pool_used);
        uint32_t* blk = blk0 + i * ADJ_BLOCK; uint32_t* ref = ref0 + i * ADJ_BLOCK;
        st[i] = di;
        blk[0] = deg; blk[63] = (uint32_t)(off + ADJ_INLINE);
        for (uint32_t j = 0; j < std::min<uint32_t>(deg, ADJ_INLINE); ++j) { blk[1 + j] = run[j]; ref[1 + j] = make_uref(nbase[run[j]], ncnt[run[j]]); }
        widx[2 * i] = 2 * di; widx[2 * i + 1] = 2 * di + 1;
        wval[2 * i] = (uint32_t)ndesc[di]; wval[2 * i + 1] = (uint32_t)(ndesc[di] >> 32);
    }
    for (size_t i = 0; i < nf; ++i) { oidx[i] = fresh[i]; oval[i] = nbase[fresh[i]]; }
    d_upper_base.reserve(n, stream, true); d_upper_cnt.reserve(n, stream, true);
    d_udesc.reserve(ndesc.size(), stream, true); d_ublk.reserve(ndesc.size() * ADJ_BLOCK, stream, true); d_ublk_ref.reserve(ndesc.size() * ADJ_BLOCK, stream, true);
    d_pool.reserve(std::max<size_t>(pool_used + tail.size(), 1), stream, true);
    if (upper_slots != n) {                       // slots added since the last sync own nothing until told otherwise
        VDB_HIP(hipMemsetAsync(d_upper_base.p + upper_slots, 0xFF, (n - upper_slots) * 4, stream));
        VDB_HIP(hipMemsetAsync(d_upper_cnt.p + upper_slots, 0, n - upper_slots, stream));
    }
    if (!tail.empty()) VDB_HIP(hipMemcpyAsync(d_pool.p + pool_used, tail.data(), tail.size() * 4, hipMemcpyHostToDevice, stream));
    if (!st.empty()) {
        uint32_t* h = st.data();
        if (st.size() * 4 <= (size_t(256) << 10)) { h = static_cast<uint32_t*>(pinned(st.size() * 4)); std::memcpy(h, st.data(), st.size() * 4); }
        s_delta.reserve(st.size(), stream, false);
        VDB_HIP(hipMemcpyAsync(s_delta.p, h, st.size() * 4, hipMemcpyHostToDevice, stream));
        const uint32_t* d = s_delta.p;
        launch_patch_blocks(d_ublk.p, d, d + nd, (uint32_t)nd, stream);
        launch_patch_blocks(d_ublk_ref.p, d, d + nd + nd * ADJ_BLOCK, (uint32_t)nd, stream);
        launch_patch_words(reinterpret_cast<uint32_t*>(d_udesc.p), d + nd + 2 * nd * ADJ_BLOCK, d + nd + 2 * nd * ADJ_BLOCK + 2 * nd, (uint32_t)(2 * nd), stream);
        launch_patch_words(d_upper_base.p, d + nd + 2 * nd * ADJ_BLOCK + 4 * nd, d + nd + 2 * nd * ADJ_BLOCK + 4 * nd + nf, (uint32_t)nf, stream);
    }
    for (uint32_t s : fresh) VDB_HIP(hipMemsetAsync(d_upper_cnt.p + s, ncnt[s], 1, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    pool_used += tail.size(); pool_live = live;
    upper_slots = n;
    upper_edges = pool_live; upper_words = h_udesc.size() * 2 + pool_used + h_udesc.size() * ADJ_BLOCK * 2;
    dirty_upper.clear();
    return true;
}

// Diagnostic (VDB_VERIFY_DEVICE=1): the packed upper layers on the device must describe the host lists, however
// many patches and re-packs produced them.
void vdb_index::verify_upper() {
    const size_t n = ids.size();
    if (n == 0 || !have_ublk || upper_slots != n) return;
    std::vector<uint32_t> base(n), pool(std::max<size_t>(pool_used, 1)), blk(std::max<size_t>(h_udesc.size(), 1) * ADJ_BLOCK), ref(blk.size()), tmp;
    std::vector<uint8_t> cnt(n); std::vector<uint64_t> desc(std::max<size_t>(h_udesc.size(), 1));
    VDB_HIP(hipMemcpyAsync(base.data(), d_upper_base.p, n * 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipMemcpyAsync(cnt.data(), d_upper_cnt.p, n, hipMemcpyDeviceToHost, stream));
    if (!h_udesc.empty()) {
        VDB_HIP(hipMemcpyAsync(desc.data(), d_udesc.p, h_udesc.size() * 8, hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipMemcpyAsync(blk.data(), d_ublk.p, h_udesc.size() * ADJ_BLOCK * 4, hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipMemcpyAsync(ref.data(), d_ublk_ref.p, h_udesc.size() * ADJ_BLOCK * 4, hipMemcpyDeviceToHost, stream));
    }
    if (pool_used) VDB_HIP(hipMemcpyAsync(pool.data(), d_pool.p, pool_used * 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    auto fail = [&](uint32_t s, size_t l, const char* what) {
        throw Failure(err_device("device copy of the upper layers differs from the host lists (slot " + std::to_string(s) + ", layer " + std::to_string(l) + ": " + what + ")"));
    };
    for (uint32_t s = 0; s < n; ++s) {
        uint32_t want = 0;
        for (size_t l = 1; l < layers.size(); ++l) if (list(l, s)) want = (uint32_t)l;
        if (cnt[s] != want) fail(s, want, "layer count");
        if (!want) continue;
        if ((size_t)base[s] + want > h_udesc.size()) fail(s, want, "descriptor range");
        for (uint32_t l = 1; l <= want; ++l) {
            const size_t di = (size_t)base[s] + l - 1;
            const std::vector<uint32_t>* lst = list(l, s);
            if (lst) clean_list(*lst, s, tmp); else tmp.clear();
            const uint64_t off = desc[di] & 0xFFFFFFFFFFull; const uint32_t deg = (uint32_t)(desc[di] >> 40);
            if (deg != tmp.size() || off + deg > pool_used) fail(s, l, "degree");
            if (!std::equal(tmp.begin(), tmp.end(), pool.begin() + off)) fail(s, l, "pool run");
            const uint32_t* b = &blk[di * ADJ_BLOCK]; const uint32_t* r = &ref[di * ADJ_BLOCK];
            if (b[0] != deg || b[63] != (uint32_t)(off + ADJ_INLINE)) fail(s, l, "block header");
            for (uint32_t j = 0; j < std::min<uint32_t>(deg, ADJ_INLINE); ++j)
                if (b[1 + j] != tmp[j] || r[1 + j] != make_uref(base[tmp[j]], cnt[tmp[j]])) fail(s, l, "block entry");
        }
    }
}

void vdb_index::sync_device(uint32_t upper_slack) {
    use_device();
    const size_t n = ids.size();
    if (graph_dirty) {
        // a whole re-pack only when most blocks changed: patching a block costs about as much as packing it
        const bool full = need_full || d_adj0.cap == 0 || dirty0.size() * 2 > n ||
                          ovf_used > 8 * ovf_live + (size_t(1) << 22);          // mostly dead overflow runs: pack anew
        if (full) pack_layer0_full(); else patch_layer0();
        bool upper_now = full || upper_dirty;
        if (upper_now && !full && upper_slack != 0 && d_upper_base.cap != 0) {
            // between rounds of the batched builder: the packed upper layers may lack up to 1/upper_slack of their lists
            // (they only route a walk towards its layer-0 start; layer 0 is always current) unless the entry point
            // moved; every search of a caller and every single insert sees them complete
            size_t lists = 0;
            for (size_t l = 1; l < layers.size(); ++l) lists += layers[l].lists.size();
            const uint64_t ep = entry_point ? *entry_point : ~0ull;
            if (ep == entry_at_pack && (lists - upper_lists_packed) * upper_slack <= upper_lists_packed) upper_now = false;
        }
        if (upper_now && !full && patch_upper()) upper_now = false, upper_dirty = false;
        else if (upper_now) { pack_upper(); upper_dirty = false; }
        if (!upper_dirty) {
            upper_lists_packed = 0;
            for (size_t l = 1; l < layers.size(); ++l) upper_lists_packed += layers[l].lists.size();
            entry_at_pack = entry_point ? *entry_point : ~0ull;
        }
        if (!upper_now && upper_slots != n) {    // new points without (packed) upper lists: default entries only
            d_upper_base.reserve(n, stream, true); d_upper_cnt.reserve(n, stream, true);
            static_assert(NONE32 == 0xFFFFFFFFu, "the byte fill below writes NONE32");
            VDB_HIP(hipMemsetAsync(d_upper_base.p + upper_slots, 0xFF, (n - upper_slots) * 4, stream));
            VDB_HIP(hipMemsetAsync(d_upper_cnt.p + upper_slots, 0, n - upper_slots, stream));
            upper_slots = n;
            h_upper_base.resize(n, NONE32); h_upper_cnt.resize(n, 0);
        }
        for (uint32_t s : dirty0) dirty0_mark[s] = 0;
        dirty0.clear(); need_full = false;
        if (getenv("VDB_VERIFY_DEVICE")) { verify_layer0(); if (!upper_dirty) verify_upper(); }
        device_bytes = n * (size_t)stride * 4 + n * 8 + n * ADJ_BLOCK * 4 + ovf_used * 4 + n * 5 + upper_words * 4;
        graph_dirty = upper_dirty;               // a postponed upper re-pack keeps the copy stale for other callers
    }
    if (deleted_dirty) {
        if (n_deleted) {
            std::vector<uint32_t> bits((n + 31) / 32, 0);
            for (size_t s = 0; s < n; ++s) if (deleted[s]) bits[s >> 5] |= 1u << (s & 31);
            d_deleted.reserve(bits.size(), stream, false);
            VDB_HIP(hipMemcpyAsync(d_deleted.p, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, stream));
            VDB_HIP(hipStreamSynchronize(stream));
        }
        deleted_dirty = false;
    }
}

GraphView vdb_index::view() const {
    GraphView g{};
    g.vecs = d_vecs.p; g.stride = stride; g.dim = (uint32_t)dim; g.n = (uint32_t)ids.size();
    g.ids = d_ids.p; g.adj0 = d_adj0.p; g.ovf = d_ovf.p; g.upper_base = d_upper_base.p; g.upper_cnt = d_upper_cnt.p;
    g.udesc = d_udesc.p; g.pool = d_pool.p; g.ublk = have_ublk ? d_ublk.p : nullptr; g.ublk_ref = have_ublk ? d_ublk_ref.p : nullptr; g.deleted = n_deleted ? d_deleted.p : nullptr; g.metric = (int)metric;
    return g;
}

uint32_t vdb_index::beam_width() const {
    int w = opts.beam_width;
    if (const char* e = std::getenv("VDB_BEAM_WIDTH")) w = std::atoi(e);
    if (w <= 0) w = 4;
    return (uint32_t)std::min(w, 4);
}

uint32_t vdb_index::pick_hash_log2(size_t, bool& global) const {
    // 4096 ids (16 KiB of LDS) keep eight search wavefronts resident per CU; a query that visits
    // more rows than that continues in a spill table in HBM (see Visited in kernels.hip)
    global = false;
    const char* fe = getenv("VDB_HASH_LOG2"); const int forced = fe ? atoi(fe) : 0;
    return (forced >= 10 && forced <= 14) ? (uint32_t)forced : 12;
}

// Launches the search.  Queries that outgrow the LDS visited table spill into HBM tables taken from
// a pool; if the pool or a spill table is exhausted the query is flagged and (on the host-buffer
// paths) repeated alone with a table sized for it.
void vdb_index::run_search(SearchArgs a, size_t nq, hipStream_t s, bool retry_on_overflow, const float* d_q_all) {
    bool global = false;
    a.hash_log2 = pick_hash_log2(a.ef, global);
    // 16-bit visited table of the register-resident beam: 4096 entries (8 KiB of LDS, sixteen wavefronts per CU);
    // the id-to-tag mapping needs slots below 2^(tag_log2 + 13), so very large shards take a larger table
    const char* te = getenv("VDB_TAG_LOG2"); const int forced_tags = te ? atoi(te) : 0;
    a.tag_log2 = 12;
    while (a.tag_log2 < 15 && ((uint64_t)a.g.n >> (a.tag_log2 + 13)) != 0) ++a.tag_log2;
    if (forced_tags >= 7 && forced_tags <= 15) a.tag_log2 = (uint32_t)forced_tags;      // tests shrink it to force forgetting
    if (const char* fg = getenv("VDB_FORCE_GENERIC")) { if (atoi(fg) != 0) a.force_generic = 1; }   // tests: the LDS-list kernel for every mode
    // Launches too small to fill the chip with one wavefront per query walk on a team of wavefronts per query.  A step
    // then expands four nodes per wavefront unless the index (or VDB_BEAM_WIDTH) names a width: that width is the total.
    a.team = team_size_for(a);
    if (a.team) {
        int w = opts.beam_width;
        if (const char* e = std::getenv("VDB_BEAM_WIDTH")) w = std::atoi(e);
        a.team_width = w > 0 ? (uint32_t)std::min<int>(w, 4 * (int)a.team) : 4u * a.team;
        if (const char* e = std::getenv("VDB_TEAM_WIDTH")) { const int t = std::atoi(e); if (t > 0) a.team_width = (uint32_t)std::min<int>(t, 4 * (int)a.team); }
        // the team shares one visited table: 8192 tags, 16 KiB (fewer ids forgotten and scored again)
        if (!(forced_tags >= 7 && forced_tags <= 15)) a.tag_log2 = std::max<uint32_t>(a.tag_log2, 13u);
    }
    if (a.mode == MODE_LITERAL) {
        // the literal walk visits every reachable row: one spill table per query of the launch, sized for all rows
        // (a pool of its own: the geometry differs from the shared one and follows the segment's size)
        a.spill_log2 = 15;
        while (a.spill_log2 < 30 && (size_t(3) << a.spill_log2) / 4 < (size_t)a.g.n + 4 * WAVE_LANES) ++a.spill_log2;
        a.spill_tables = (uint32_t)nq;
        const size_t words = (size_t)a.spill_tables + ((size_t)a.spill_tables << a.spill_log2);
        s_spill_lit.reserve(words, s, false);
        VDB_HIP(hipMemsetAsync(s_spill_lit.p, 0, (size_t)a.spill_tables * 4, s));
        a.spill_locks = s_spill_lit.p; a.spill_pool = s_spill_lit.p + a.spill_tables;
    } else {
    // one spill table per possibly resident wavefront (256 CUs x 16), 2^15 ids each
    a.spill_log2 = 15;
    a.spill_tables = 4096;
    const size_t spill_words = (size_t)a.spill_tables + ((size_t)a.spill_tables << a.spill_log2);
    if (s_spill.cap < spill_words) {
        // lock words start free and every query hands its table back, so they are cleared once:
        // launches on different streams may share the pool while they overlap
        s_spill.reserve(spill_words, s, false);
        VDB_HIP(hipMemsetAsync(s_spill.p, 0, (size_t)a.spill_tables * 4, s));
        VDB_HIP(hipStreamSynchronize(s));
    }
    a.spill_locks = s_spill.p; a.spill_pool = s_spill.p + a.spill_tables;
    }
    launch_search(a, s);
    // the register-resident beam never overflows (its visited set forgets instead): nothing to follow up — except during
    // construction, whose upper-layer beams keep the exact visited table and would leave a point unlinked on overflow
    if (reg_list_chunks(a) > 0 && a.mode != MODE_BUILD) return;
    if (!retry_on_overflow) return;
    std::vector<vdb_query_stats> st(nq);
    VDB_HIP(hipMemcpyAsync(st.data(), a.stats, nq * sizeof(vdb_query_stats), hipMemcpyDefault, s));   // a.stats may be pinned host memory (small batches)
    VDB_HIP(hipStreamSynchronize(s));
    std::vector<uint32_t> redo;
    for (size_t i = 0; i < nq; ++i) if (st[i].flags & QF_HASH_OVERFLOW) redo.push_back((uint32_t)i);
    uint32_t lg = 19;
    while (!redo.empty()) {
        if (lg > 27) throw Failure(err_device("visited table overflow beyond 2^27 entries"));
        // the query alone, with a private spill table sized for it
        s_hash.reserve(64 + (size_t(1) << lg), s, false);
        for (uint32_t qi : redo) {
            SearchArgs b = a;
            b.nq = 1; b.spill_locks = s_hash.p; b.spill_pool = s_hash.p + 64; b.spill_tables = 1; b.spill_log2 = lg;
            VDB_HIP(hipMemsetAsync(b.spill_locks, 0, 4, s));
            if (b.queries) b.queries = d_q_all + (size_t)qi * dim;
            if (b.query_slots) b.query_slots = a.query_slots + qi;
            if (b.q_levels) { b.q_levels = a.q_levels + qi; b.nb_off = a.nb_off + qi; }   // nb_off values stay absolute
            if (b.q_entry) b.q_entry = a.q_entry + qi;
            if (b.cand_scratch) b.cand_scratch = a.cand_scratch + (size_t)qi * a.cand_cap;
            b.out = a.out ? a.out + (size_t)qi * a.top_k : nullptr; b.out_n = a.out_n + qi; b.stats = a.stats + qi;
            launch_search(b, s);
        }
        VDB_HIP(hipMemcpyAsync(st.data(), a.stats, nq * sizeof(vdb_query_stats), hipMemcpyDefault, s));   // a.stats may be pinned host memory (small batches)
        VDB_HIP(hipStreamSynchronize(s));
        std::vector<uint32_t> again;
        for (uint32_t qi : redo) if (st[qi].flags & QF_HASH_OVERFLOW) again.push_back(qi);
        redo.swap(again); lg += 3;
    }
}

void vdb_index::search_batch(const float* q, size_t nq, size_t top_k, size_t ef_search, const FilterArgs& fa,
                             vdb_scored_point* out, uint32_t* out_n, vdb_query_stats* stats) {
    if (nq == 0) return;
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    const size_t ef_ = ef_search ? ef_search : ef;
    if (ef_ > MAX_EF) throw Failure(err_arg("ef_search must be <= 1024"));
    if (!entry_point && fa.entry_slot == NONE32) {                                   // hnsw.rs:515-518
        std::memset(out_n, 0, nq * sizeof(uint32_t));
        if (stats) std::memset(stats, 0, nq * sizeof(vdb_query_stats));
        return;
    }
    if (top_k == 0) { std::memset(out_n, 0, nq * sizeof(uint32_t)); return; }
    sync_device();
    s_q.reserve(nq * dim, stream, false); s_out.reserve(nq * top_k, stream, false);
    s_n.reserve(nq, stream, false); s_stats.reserve(nq, stream, false);
    // small batches go through pinned host memory: copies from and to pageable memory cost more than the search of one query
    const size_t b_q = nq * dim * sizeof(float), b_out = nq * top_k * sizeof(vdb_scored_point), b_n = nq * sizeof(uint32_t), b_st = nq * sizeof(vdb_query_stats);
    const bool small = b_q + b_out + b_n + b_st <= (size_t(1) << 20);
    // a handful of queries (one per call is the reference's shape): the kernel reads them from pinned host memory and writes
    // its results there, so the call is one launch and one wait — no copy is enqueued at all
    const size_t o_out = (b_q + 15) & ~size_t(15), o_n = o_out + ((b_out + 15) & ~size_t(15)), o_st = o_n + ((b_n + 15) & ~size_t(15));
    const bool zero = small && o_st + b_st <= (size_t(64) << 10) && fa.mode != MODE_LITERAL && !std::getenv("VDB_NO_ZERO_COPY");
    char* pin = small ? static_cast<char*>(pinned(zero ? o_st + b_st : b_q + b_out + b_n + b_st)) : nullptr;
    char* dpin = nullptr;
    if (zero) { std::memcpy(pin, q, b_q); VDB_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dpin), pin, 0)); }
    else if (small) { std::memcpy(pin, q, b_q); VDB_HIP(hipMemcpyAsync(s_q.p, pin, b_q, hipMemcpyHostToDevice, stream)); }
    else VDB_HIP(hipMemcpyAsync(s_q.p, q, nq * dim * sizeof(float), hipMemcpyHostToDevice, stream));
    SearchArgs a{};
    a.g = view(); a.mode = fa.mode; a.arith = opts.arith; a.queries = zero ? reinterpret_cast<const float*>(dpin) : s_q.p; a.nq = (uint32_t)nq;
    a.entry = fa.entry_slot != NONE32 ? fa.entry_slot : slot_of.at(*entry_point);
    a.max_level = (uint32_t)current_max_level;
    auto fl = first_live_slot(); a.fallback_live = fl ? *fl : NONE32;
    a.ef = (uint32_t)ef_; a.top_k = (uint32_t)top_k; a.out = s_out.p; a.out_n = s_n.p; a.stats = s_stats.p;
    if (zero) { a.out = reinterpret_cast<vdb_scored_point*>(dpin + o_out); a.out_n = reinterpret_cast<uint32_t*>(dpin + o_n); a.stats = reinterpret_cast<vdb_query_stats*>(dpin + o_st); }
    a.beam_width = beam_width();
    if (fa.bits) { a.pass_bits = fa.bits->d_pass.p; a.err_bits = fa.bits->d_err.p; }
    if (fa.mode == MODE_LITERAL) {
        // hnsw.rs:639-689 never prunes its candidate heap: it can hold every row of the segment (each row is scored at
        // most once), so the heap of a query has room for all of them and the walk runs at any segment size.  The
        // batch is cut into launches whose heaps and visited tables together stay below about 2 GiB.
        const char* ce = getenv("VDB_LITERAL_CAND_CAP");                  // tests: a queue that is too small is an error, never a wrong answer
        size_t cap = ids.size() + 1;
        if (ce && atoi(ce) > 0) cap = std::min<size_t>(cap, (size_t)atoi(ce));
        a.cand_cap = (uint32_t)cap;
        size_t table_bytes = size_t(4) << 15;
        while (table_bytes < (size_t(4) << 30) && table_bytes * 3 / 16 < ids.size() + 4 * WAVE_LANES) table_bytes <<= 1;
        const size_t per_query = cap * sizeof(vdb_scored_point) + table_bytes;
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(nq, (size_t(2) << 30) / per_query));
        s_cand.reserve(chunk * cap, stream, false);
        a.cand_scratch = s_cand.p;
        for (size_t q0 = 0; q0 < nq; q0 += chunk) {
            SearchArgs b = a;
            b.nq = (uint32_t)std::min(chunk, nq - q0);
            b.queries = a.queries + q0 * dim; b.out = a.out + q0 * top_k; b.out_n = a.out_n + q0; b.stats = a.stats + q0;
            run_search(b, b.nq, stream, true, b.queries);
        }
    } else
    run_search(a, nq, stream, true, a.queries);
    std::vector<vdb_query_stats> st(nq);
    if (zero) {
        VDB_HIP(hipStreamSynchronize(stream));
        std::memcpy(out, pin + o_out, b_out); std::memcpy(out_n, pin + o_n, b_n); std::memcpy(st.data(), pin + o_st, b_st);
    } else if (small) {
        VDB_HIP(hipMemcpyAsync(pin + b_q, s_out.p, b_out, hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipMemcpyAsync(pin + b_q + b_out, s_n.p, b_n, hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipMemcpyAsync(pin + b_q + b_out + b_n, s_stats.p, b_st, hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipStreamSynchronize(stream));
        std::memcpy(out, pin + b_q, b_out); std::memcpy(out_n, pin + b_q + b_out, b_n); std::memcpy(st.data(), pin + b_q + b_out + b_n, b_st);
    } else {
        VDB_HIP(hipMemcpyAsync(out, s_out.p, nq * top_k * sizeof(vdb_scored_point), hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipMemcpyAsync(out_n, s_n.p, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipMemcpyAsync(st.data(), s_stats.p, nq * sizeof(vdb_query_stats), hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipStreamSynchronize(stream));
    }
    for (size_t i = 0; i < nq; ++i) {
        if (st[i].flags & QF_CAND_OVERFLOW)
            throw Failure(err_device("literal filtered search: the candidate queue of query " + std::to_string(i) + " outgrew " + std::to_string(a.cand_cap) +
                                     " entries (once ef passing rows are found the reference's walk scores every reachable row); "
                                     "use VDB_FILTER_MASKED for segments of this size"));
        if (st[i].flags & QF_FILTER_ERROR) throw Failure(Error{VDB_ERR_INVALID_PAYLOAD, "@slot:" + std::to_string(st[i].expanded)});
    }
    if (stats) std::memcpy(stats, st.data(), nq * sizeof(vdb_query_stats));
}

void vdb_index::search_batch_dev(const float* d_q, size_t nq, size_t top_k, size_t ef_search, vdb_scored_point* d_out, uint32_t* d_out_n,
                                 vdb_query_stats* d_stats, hipStream_t s, const FilterArgs* fa) {
    if (nq == 0) return;
    std::lock_guard<std::mutex> lk(mu);
    search_dev_locked(d_q, nq, top_k, ef_search, d_out, d_out_n, d_stats, s, fa);
}

void vdb_index::search_dev_locked(const float* d_q, size_t nq, size_t top_k, size_t ef_search, vdb_scored_point* d_out, uint32_t* d_out_n,
                                  vdb_query_stats* d_stats, hipStream_t s, const FilterArgs* fa) {
    use_device();
    if (!s) s = stream;
    const size_t ef_ = ef_search ? ef_search : ef;
    if (ef_ > MAX_EF) throw Failure(err_arg("ef_search must be <= 1024"));
    if (!entry_point) { VDB_HIP(hipMemsetAsync(d_out_n, 0, nq * sizeof(uint32_t), s)); return; }
    sync_device();
    SearchArgs a{};
    a.g = view(); a.mode = fa ? fa->mode : MODE_SEARCH; a.arith = opts.arith; a.queries = d_q; a.nq = (uint32_t)nq;
    a.entry = slot_of.at(*entry_point); a.max_level = (uint32_t)current_max_level;
    auto fl = first_live_slot(); a.fallback_live = fl ? *fl : NONE32;
    a.ef = (uint32_t)ef_; a.top_k = (uint32_t)top_k; a.out = d_out; a.out_n = d_out_n; a.stats = d_stats;
    if (fa && fa->bits) { a.pass_bits = fa->bits->d_pass.p; a.err_bits = fa->bits->d_err.p; }
    if (!a.stats) {                       // the kernels report per-query flags: one private array per stream
        auto& buf = stream_scratch(s).stats;
        buf.reserve(nq, s, false);
        a.stats = buf.p;
    }
    a.beam_width = beam_width();
    // no host round trip on this path: a query of the LDS-list kernel (ef above 256) that outgrows every
    // spill table reports QF_HASH_OVERFLOW in d_stats
    run_search(a, nq, s, false, d_q);
}

void vdb_index::exact_knn(const float* q, size_t nq, size_t top_k, bool dev, vdb_scored_point* out, uint32_t* out_n, hipStream_t s,
                          const uint32_t* rowmap, size_t n_rows) {
    if (nq == 0) return;
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    if (!s) s = stream;
    sync_device();
    GraphView g = view();
    if (rowmap) { g.rowmap = rowmap; g.n = (uint32_t)n_rows; }
    if (g.n == 0 || top_k == 0) {
        if (dev) VDB_HIP(hipMemsetAsync(out_n, 0, nq * sizeof(uint32_t), s)); else std::memset(out_n, 0, nq * sizeof(uint32_t));
        return;
    }
    // FAST: MFMA candidate pass + strict re-rank; STRICT (and shapes the MFMA kernel does not take): every pair in strict arithmetic
    const bool mfma = opts.arith == VDB_ARITH_FAST && knn_mfma_supported(g, (uint32_t)top_k) && !std::getenv("VDB_KNN_STRICT");
    const uint32_t chunks = exact_knn_chunks(g.n);
    StreamScratch& sc = stream_scratch(s);
    DevBuf<uint32_t>& words = sc.knn_words; DevBuf<vdb_scored_point>& lists = sc.knn_lists;
    if (mfma) words.reserve(knn_mfma_scratch_words(g.n, (uint32_t)nq, g.stride), s, false);
    else lists.reserve(nq * chunks * top_k, s, false);
    // A graph over all rows answers first (a short beam): its k-th key bounds the true one, and the candidate pass that
    // starts from that bound admits a fraction of the rows it admits from +inf.  The exact answer does not depend on it.
    const bool hint = mfma && !rowmap && entry_point && ids.size() >= 65536 && top_k <= MAX_EF / 4 && !std::getenv("VDB_KNN_NO_HINT");
    auto run = [&](const float* dq, vdb_scored_point* dout, uint32_t* dn) {
        if (hint) search_dev_locked(dq, nq, top_k, std::max<size_t>(32, 2 * top_k), dout, dn, nullptr, s, nullptr);
        if (mfma) launch_exact_knn_mfma(g, dq, (uint32_t)nq, (uint32_t)top_k, dout, dn, words.p, s, hint);
        else launch_exact_knn(g, dq, (uint32_t)nq, (uint32_t)top_k, dout, dn, lists.p, chunks, s);
    };
    if (dev) { run(q, out, out_n); return; }
    s_q.reserve(nq * dim, s, false); s_out.reserve(nq * top_k, s, false); s_n.reserve(nq, s, false);
    VDB_HIP(hipMemcpyAsync(s_q.p, q, nq * dim * sizeof(float), hipMemcpyHostToDevice, s));
    run(s_q.p, s_out.p, s_n.p);
    VDB_HIP(hipMemcpyAsync(out, s_out.p, nq * top_k * sizeof(vdb_scored_point), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipMemcpyAsync(out_n, s_n.p, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    VDB_HIP(hipStreamSynchronize(s));
}

void vdb_index::exact_knn_static(const float* d_q, size_t nq, size_t top_k, vdb_scored_point* d_out, uint32_t* d_out_n, hipStream_t s) {
    if (nq == 0) return;
    if (nq > 0xFFFFFFFFull) throw Failure(err_arg("exact k-NN without allocation: more than 2^32 - 1 queries in one call"));
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    if (!s) s = stream;
    // bringing the device copy up to date may allocate: this entry point refuses instead
    if (deleted_dirty || rows_on_device != ids.size())
        throw Failure(err_arg("exact k-NN without allocation: the index has edits that are not on the device yet; call vdb_index_sync_device first"));
    const GraphView g = view();
    if (opts.arith != VDB_ARITH_FAST || !knn_mfma_supported(g, (uint32_t)top_k))
        throw Failure(err_arg("exact k-NN without allocation needs a FAST index that holds rows and top_k in 1..28"));
    launch_exact_knn_mfma_static(g, d_q, (uint32_t)nq, (uint32_t)top_k, d_out, d_out_n, s);
}

std::vector<float> vdb_index::score_rows(uint32_t query_slot, const std::vector<uint32_t>& slots) {
    std::vector<float> raw(slots.size());
    if (slots.empty()) return raw;
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    s_slots.reserve(slots.size(), stream, false); s_raw.reserve(slots.size(), stream, false);
    VDB_HIP(hipMemcpyAsync(s_slots.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, stream));
    launch_score_rows(view(), query_slot, s_slots.p, (uint32_t)slots.size(), opts.arith, s_raw.p, stream);
    VDB_HIP(hipMemcpyAsync(raw.data(), s_raw.p, raw.size() * 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    return raw;
}

std::vector<float> vdb_index::score_all_rows(uint32_t query_slot) {
    const size_t n = ids.size();
    std::vector<float> raw(n);
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    s_raw.reserve(n, stream, false); s_q.reserve(stride, stream, false);
    // the query is a stored row: copy it out of the table (dense, dim floats) and stream the table past it
    VDB_HIP(hipMemcpyAsync(s_q.p, d_vecs.p + (size_t)query_slot * stride, dim * sizeof(float), hipMemcpyDeviceToDevice, stream));
    launch_score_scan(s_q.p, 1, d_vecs.p, (uint32_t)n, (uint32_t)dim, stride, (int)metric, opts.arith, s_raw.p, stream);
    VDB_HIP(hipMemcpyAsync(raw.data(), s_raw.p, n * 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    return raw;
}

// Layer-0 beam for a query that is a stored row (no second normalisation), greedy descent from
// `entry_slot` first: the fallback search of hnsw.rs:268-273.  Returns the full sorted beam.
std::vector<vdb_scored_point> vdb_index::search_from_row(uint32_t row_slot, uint32_t entry_slot, size_t ef_search) {
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    sync_device();
    const uint32_t lvl = 0, off[2] = {0, 1};
    s_slots.reserve(1, stream, false); s_lvl.reserve(1, stream, false); s_off.reserve(2, stream, false);
    s_nb.reserve(m, stream, false); s_nbc.reserve(1, stream, false);
    s_out.reserve(ef_search, stream, false); s_n.reserve(1, stream, false); s_stats.reserve(1, stream, false);
    VDB_HIP(hipMemcpyAsync(s_slots.p, &row_slot, 4, hipMemcpyHostToDevice, stream));
    VDB_HIP(hipMemcpyAsync(s_lvl.p, &lvl, 4, hipMemcpyHostToDevice, stream));
    VDB_HIP(hipMemcpyAsync(s_off.p, off, 8, hipMemcpyHostToDevice, stream));
    SearchArgs a{};
    a.g = view(); a.mode = MODE_BUILD; a.arith = opts.arith; a.query_slots = s_slots.p; a.nq = 1;
    a.entry = entry_slot; a.max_level = (uint32_t)current_max_level;
    auto fl = first_live_slot(); a.fallback_live = fl ? *fl : NONE32;
    a.ef = (uint32_t)ef_search; a.top_k = (uint32_t)ef_search; a.out = s_out.p; a.out_n = s_n.p; a.stats = s_stats.p;
    a.beam_width = 1;
    a.q_levels = s_lvl.p; a.nb_off = s_off.p; a.m = (uint32_t)m; a.nb_out = s_nb.p; a.nb_cnt = s_nbc.p;
    run_search(a, 1, stream, true, nullptr);
    uint32_t n = 0;
    std::vector<vdb_scored_point> out(ef_search);
    VDB_HIP(hipMemcpyAsync(&n, s_n.p, 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipMemcpyAsync(out.data(), s_out.p, ef_search * sizeof(vdb_scored_point), hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    out.resize(n);
    return out;
}

void vdb_index::search_from_rows(const std::vector<uint32_t>& rows, uint32_t entry_slot, size_t ef_search,
                                 std::vector<vdb_scored_point>& out, std::vector<uint32_t>& cnt) {
    const size_t nq = rows.size();
    out.assign(nq * ef_search, vdb_scored_point{}); cnt.assign(nq, 0);
    if (nq == 0) return;
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    sync_device();
    std::vector<uint32_t> lv(nq, 0), off(nq + 1);
    for (size_t i = 0; i <= nq; ++i) off[i] = (uint32_t)i;
    s_slots.reserve(nq, stream, false); s_lvl.reserve(nq, stream, false); s_off.reserve(nq + 1, stream, false);
    s_nb.reserve(nq * m, stream, false); s_nbc.reserve(nq, stream, false);
    s_out.reserve(nq * ef_search, stream, false); s_n.reserve(nq, stream, false); s_stats.reserve(nq, stream, false);
    VDB_HIP(hipMemcpyAsync(s_slots.p, rows.data(), nq * 4, hipMemcpyHostToDevice, stream));
    VDB_HIP(hipMemcpyAsync(s_lvl.p, lv.data(), nq * 4, hipMemcpyHostToDevice, stream));
    VDB_HIP(hipMemcpyAsync(s_off.p, off.data(), (nq + 1) * 4, hipMemcpyHostToDevice, stream));
    SearchArgs a{};
    a.g = view(); a.mode = MODE_BUILD; a.arith = opts.arith; a.query_slots = s_slots.p; a.nq = (uint32_t)nq;
    a.entry = entry_slot; a.max_level = (uint32_t)current_max_level;
    auto fl = first_live_slot(); a.fallback_live = fl ? *fl : NONE32;
    a.ef = (uint32_t)ef_search; a.top_k = (uint32_t)ef_search; a.out = s_out.p; a.out_n = s_n.p; a.stats = s_stats.p;
    a.beam_width = 1;
    a.q_levels = s_lvl.p; a.nb_off = s_off.p; a.m = (uint32_t)m; a.nb_out = s_nb.p; a.nb_cnt = s_nbc.p;
    run_search(a, nq, stream, true, nullptr);
    VDB_HIP(hipMemcpyAsync(cnt.data(), s_n.p, nq * 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipMemcpyAsync(out.data(), s_out.p, nq * ef_search * sizeof(vdb_scored_point), hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
}

std::vector<float> vdb_index::score_pairs(const std::vector<uint32_t>& query_slots, const std::vector<uint32_t>& slots) {
    std::vector<float> raw(slots.size());
    if (slots.empty()) return raw;
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    sync_device();
    s_qslots.reserve(slots.size(), stream, false); s_slots.reserve(slots.size(), stream, false); s_raw.reserve(slots.size(), stream, false);
    VDB_HIP(hipMemcpyAsync(s_qslots.p, query_slots.data(), slots.size() * 4, hipMemcpyHostToDevice, stream));
    VDB_HIP(hipMemcpyAsync(s_slots.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, stream));
    launch_score_pairs(view(), s_qslots.p, s_slots.p, (uint32_t)slots.size(), s_raw.p, stream);
    VDB_HIP(hipMemcpyAsync(raw.data(), s_raw.p, raw.size() * 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    return raw;
}

// greedy_search_layer_unfiltered as a public accessor (hnsw.rs:333-368).  Searches run the descent
// inside the search kernel; this standalone form scores one adjacency list per step with the strict
// scan kernel and applies the first-improvement rule on the host.
uint32_t vdb_index::greedy_unfiltered(const float* q, uint32_t entry_slot, size_t level) {
    std::vector<float> qn(q, q + dim);
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    sync_device();
    s_q.reserve(stride, stream, false);
    VDB_HIP(hipMemcpyAsync(s_q.p, qn.data(), dim * sizeof(float), hipMemcpyHostToDevice, stream));
    uint32_t current = entry_slot; bool changed = true; int steps = 0;
    auto score_list = [&](const std::vector<uint32_t>& slots) {
        std::vector<float> raw(slots.size());
        s_slots.reserve(slots.size(), stream, false); s_raw.reserve(slots.size(), stream, false);
        // gather the rows and scan them against the query held in s_q
        DevBuf<float> rows; rows.reserve(slots.size() * stride, stream, false);
        for (size_t i = 0; i < slots.size(); ++i)
            VDB_HIP(hipMemcpyAsync(rows.p + i * stride, d_vecs.p + (size_t)slots[i] * stride, stride * sizeof(float), hipMemcpyDeviceToDevice, stream));
        launch_score_scan(s_q.p, 1, rows.p, (uint32_t)slots.size(), (uint32_t)dim, stride, (int)metric, VDB_ARITH_STRICT, s_raw.p, stream);
        VDB_HIP(hipMemcpyAsync(raw.data(), s_raw.p, raw.size() * 4, hipMemcpyDeviceToHost, stream));
        VDB_HIP(hipStreamSynchronize(stream));
        return raw;
    };
    while (changed && steps < 1000) {
        ++steps; changed = false;
        const std::vector<uint32_t>* nb = list(level, current);
        if (!nb) continue;
        std::vector<uint32_t> cand{current};
        for (uint32_t x : *nb) if (!deleted[x]) cand.push_back(x);
        if (cand.size() == 1) continue;
        std::vector<float> raw = score_list(cand);
        const float s_current = normalize_score(raw[0]);
        for (size_t i = 1; i < cand.size(); ++i)
            if (normalize_score(raw[i]) < s_current) { current = cand[i]; changed = true; break; }
    }
    return current;
}

// ------------------------------------------------------------------------------------------------
// Graph producer
// ------------------------------------------------------------------------------------------------

// One construction launch: for every new slot, greedy descent above its level and an ef-beam on
// each layer from its level down to 0 (hnsw.rs:169-199); returns the first M results per layer.
vdb_index::BuildResult vdb_index::build_search(const std::vector<uint32_t>& slots, uint32_t entry_slot, bool lazy_upper) {
    BuildResult br;
    const size_t nq = slots.size();
    br.off.resize(nq + 1); br.off[0] = 0;
    std::vector<uint32_t> lv(nq);
    for (size_t i = 0; i < nq; ++i) { lv[i] = levels[slots[i]]; br.off[i + 1] = br.off[i] + lv[i] + 1; }
    const size_t rows = br.off[nq];
    br.nb.assign(rows * m, 0); br.cnt.assign(rows, 0);
    std::lock_guard<std::mutex> lk(mu);
    use_device();
    sync_device(lazy_upper ? UPPER_SLACK_BUILD : 0);
    // one upload (slots, levels, offsets) and one download (counts, neighbours); a single insert stages both in
    // page-locked memory so that neither copy blocks on its own
    const size_t in_words = 3 * nq + 1, out_words = rows * (m + 1);
    s_bin.reserve(in_words, stream, false); s_bout.reserve(out_words, stream, false);
    s_n.reserve(nq, stream, false); s_stats.reserve(nq, stream, false);
    std::vector<uint32_t> pageable;
    uint32_t* h;
    if ((in_words + out_words) * 4 <= (size_t(256) << 10)) h = static_cast<uint32_t*>(pinned((in_words + out_words) * 4));
    else { pageable.resize(in_words + out_words); h = pageable.data(); }
    std::memcpy(h, slots.data(), nq * 4); std::memcpy(h + nq, lv.data(), nq * 4); std::memcpy(h + 2 * nq, br.off.data(), (nq + 1) * 4);
    VDB_HIP(hipMemcpyAsync(s_bin.p, h, in_words * 4, hipMemcpyHostToDevice, stream));
    VDB_HIP(hipMemsetAsync(s_bout.p, 0, rows * 4, stream));
    SearchArgs a{};
    a.g = view(); a.mode = MODE_BUILD; a.arith = opts.arith; a.query_slots = s_bin.p; a.nq = (uint32_t)nq;
    a.entry = entry_slot; a.max_level = (uint32_t)current_max_level;
    auto fl = first_live_slot(); a.fallback_live = fl ? *fl : NONE32;
    a.ef = (uint32_t)ef; a.top_k = 0; a.out = nullptr; a.out_n = s_n.p; a.stats = s_stats.p;
    // a handful of points cannot fill the GPU: their walks take several parents per step, which shortens the
    // chain of dependent steps (the explored set only grows); STRICT keeps the reference's one parent per step
    a.beam_width = (opts.arith == VDB_ARITH_FAST && nq <= 256) ? beam_width() : 1;
    a.q_levels = s_bin.p + nq; a.nb_off = s_bin.p + 2 * nq; a.m = (uint32_t)m; a.nb_cnt = s_bout.p; a.nb_out = s_bout.p + rows;
    run_search(a, nq, stream, true, nullptr);
    VDB_HIP(hipMemcpyAsync(h + in_words, s_bout.p, out_words * 4, hipMemcpyDeviceToHost, stream));
    VDB_HIP(hipStreamSynchronize(stream));
    std::memcpy(br.cnt.data(), h + in_words, rows * 4); std::memcpy(br.nb.data(), h + in_words + rows, rows * m * 4);
    return br;
}

// hnsw.rs:175-205 for one point, given its per-layer candidate lists.
void vdb_index::link_point(uint32_t s, const BuildResult& br, size_t qi) {
    const size_t level = levels[s];
    for (size_t l = level + 1; l-- > 0;) {
        const size_t row = br.off[qi] + l;
        std::vector<uint32_t> neighbors(br.nb.begin() + row * m, br.nb.begin() + row * m + br.cnt[row]);
        std::vector<uint32_t> linked = neighbors;
        if (std::find(linked.begin(), linked.end(), s) == linked.end()) linked.push_back(s);
        list_or_default(l, s) = linked;
        for (uint32_t nb : neighbors) {
            // layer 0, device block of nb current and not full: the new neighbour is an append at a known
            // position of the packed list (s is not in the list and is not nb itself), sent as one entry
            if (l == 0 && !need_full && nb != s && nb < h_deg0.size() && nb < dirty0_mark.size() && !dirty0_mark[nb] && h_deg0[nb] < ADJ_INLINE &&
                nb < layers[0].idx.size() && layers[0].idx[nb] >= 0) {
                std::vector<uint32_t>& e = layers[0].lists[layers[0].idx[nb]];
                if (std::find(e.begin(), e.end(), s) == e.end()) {
                    e.push_back(s);
                    delta_slot.push_back(nb); delta_pos.push_back(h_deg0[nb]++); delta_val.push_back(s);
                    ++l0_edges; graph_dirty = true;
                }
                continue;
            }
            std::vector<uint32_t>& e = list_or_default(l, nb);
            if (std::find(e.begin(), e.end(), s) == e.end()) e.push_back(s);
        }
    }
    if (level > current_max_level) { entry_point = ids[s]; current_max_level = level; }
}

void vdb_index::insert(uint64_t id, const float* v, size_t len) {
    if (slot_of.count(id)) return;                                                   // hnsw.rs:122-125
    if (len != dim) throw Failure(err_length_mismatch(dim, len));                    // hnsw.rs:127-133
    const uint32_t s = append_rows(&id, v, 1);
    if (!entry_point) { entry_point = id; current_max_level = levels[s]; return; }  // hnsw.rs:150-155
    uint32_t ep = slot_of.at(*entry_point);
    if (deleted[ep]) { auto f = first_live_slot(); if (f) ep = *f; }                 // hnsw.rs:158-164
    BuildResult br = build_search({s}, ep);
    link_point(s, br, 0);
}

// Extension: rounds of at most max(1, n_before / 4) points (capped) search the graph as it stood
// at the start of their round, then link in order.
void vdb_index::insert_batch(const uint64_t* pids, const float* v, size_t n, size_t len) {
    if (len != dim) throw Failure(err_length_mismatch(dim, len));
    size_t i = 0;
    while (i < n) {
        if (slot_of.count(pids[i])) { ++i; continue; }
        if (!entry_point) { insert(pids[i], v + i * dim, len); ++i; continue; }
        const size_t round = std::min<size_t>({n - i, std::max<size_t>(1, ids.size() / 4), 32768});
        std::vector<uint64_t> rid; std::vector<float> rv; rid.reserve(round); rv.reserve(round * dim);
        std::set<uint64_t> seen;
        size_t j = i;
        for (; j < n && rid.size() < round; ++j) {
            if (slot_of.count(pids[j]) || !seen.insert(pids[j]).second) continue;
            rid.push_back(pids[j]); rv.insert(rv.end(), v + j * dim, v + (j + 1) * dim);
        }
        i = j;
        if (rid.empty()) continue;
        uint32_t ep = slot_of.at(*entry_point);
        if (deleted[ep]) { auto f = first_live_slot(); if (f) ep = *f; }
        const uint32_t first = append_rows(rid.data(), rv.data(), rid.size());
        std::vector<uint32_t> slots(rid.size());
        for (size_t k = 0; k < rid.size(); ++k) slots[k] = first + (uint32_t)k;
        BuildResult br = build_search(slots, ep, true);
        // linking touches ~17 lists per point at random places of the host mirror: the list headers of the
        // point 16 ahead and the list contents of the point 8 ahead are requested before they are needed
        auto ahead = [&](size_t k, bool contents) {
            const size_t level = levels[slots[k]];
            for (size_t l = 0; l <= level && l < layers.size(); ++l) {
                const HostLayer& L = layers[l];
                const size_t row = br.off[k] + l;
                for (uint32_t j = 0; j < br.cnt[row]; ++j) {
                    const uint32_t nb = br.nb[row * m + j];
                    if (nb >= L.idx.size() || L.idx[nb] < 0) continue;
                    const std::vector<uint32_t>& v = L.lists[L.idx[nb]];
                    if (!contents) __builtin_prefetch(&v);
                    else if (!v.empty()) { __builtin_prefetch(v.data()); __builtin_prefetch(v.data() + v.size() - 1, 1); }
                }
            }
        };
        for (size_t k = 0; k < slots.size(); ++k) {
            if (k + 16 < slots.size()) ahead(k + 16, false);
            if (k + 8 < slots.size()) ahead(k + 8, true);
            link_point(slots[k], br, k);
        }
    }
}

void vdb_index::mark_deleted(uint64_t id) {                                          // hnsw.rs:111-117
    auto it = slot_of.find(id);
    if (it == slot_of.end()) pending_deleted.insert(id);
    if (it != slot_of.end() && !deleted[it->second]) { deleted[it->second] = 1; ++n_deleted; live.erase(id); deleted_dirty = true; }
    if (entry_point && *entry_point == id) {
        if (live.empty()) entry_point = std::nullopt; else entry_point = *live.begin();
    }
}

void vdb_index::add_bidirectional_edge(size_t level, uint32_t a, uint32_t b) {       // hnsw.rs:328-331
    list_or_default(level, a).push_back(b);
    list_or_default(level, b).push_back(a);
}

void vdb_index::import_graph(size_t n, const uint64_t* pids, const uint32_t* lv, const float* vecs, const uint8_t* del,
                             size_t n_layers, const vdb_layer_csr* csr, bool has_entry, uint64_t entry, size_t cur_max) {
    // ---- validate everything and build the new host state aside: a bad graph leaves the index as it was ----
    if (n >= 0x7FFFFFFFull) throw Failure(err_arg("more than 2^31 points in one index"));
    std::unordered_map<uint64_t, uint32_t> nslot; nslot.reserve(n * 2);
    for (size_t i = 0; i < n; ++i)
        if (!nslot.emplace(pids[i], (uint32_t)i).second) throw Failure(err_arg("id " + std::to_string(pids[i]) + " occurs twice"));
    if (has_entry && !nslot.count(entry)) throw Failure(err_arg("entry id unknown"));
    std::vector<HostLayer> nlayers(n_layers);
    for (size_t l = 0; l < n_layers; ++l) {
        const vdb_layer_csr& c = csr[l];
        const std::string where = "layer " + std::to_string(l);
        if (c.n_lists && (!c.owners || !c.offsets)) throw Failure(err_arg(where + ": owners / offsets are NULL"));
        if (c.n_edges && !c.neighbors) throw Failure(err_arg(where + ": neighbors is NULL"));
        if (c.n_lists == 0) { if (c.n_edges) throw Failure(err_arg(where + ": neighbours without lists")); continue; }
        if (c.offsets[0] != 0 || c.offsets[c.n_lists] != c.n_edges) throw Failure(err_arg(where + ": offsets do not cover its neighbours"));
        for (size_t k = 0; k < c.n_lists; ++k)
            if (c.offsets[k] > c.offsets[k + 1] || c.offsets[k + 1] > c.n_edges) throw Failure(err_arg(where + ": offsets of list " + std::to_string(k) + " are not ascending"));
        for (size_t e = 0; e < c.n_edges; ++e) if (c.neighbors[e] >= n) throw Failure(err_arg(where + ": neighbour slot out of range"));
        HostLayer& L = nlayers[l];
        L.idx.assign(n, -1); L.lists.resize(c.n_lists); L.owner.resize(c.n_lists);
        for (size_t k = 0; k < c.n_lists; ++k) {
            const uint32_t o = c.owners[k];
            if (o >= n) throw Failure(err_arg(where + ": list owner out of range"));
            if (L.idx[o] >= 0) throw Failure(err_arg(where + ": slot " + std::to_string(o) + " owns two lists"));
            L.idx[o] = (int32_t)k; L.owner[k] = o;
            L.lists[k].assign(c.neighbors + c.offsets[k], c.neighbors + c.offsets[k + 1]);
        }
    }
    // ---- take over ----
    use_device();
    ids.assign(pids, pids + n); slot_of.swap(nslot); hvecs.assign(vecs, vecs + n * dim); layers.swap(nlayers);
    levels.resize(n); deleted.assign(n, 0); live.clear(); pending_deleted.clear();
    dirty0.clear(); dirty0_mark.clear(); need_full = true;
    dirty_upper.clear(); dirty_upper_lost = false; upper_dirty = true;
    delta_slot.clear(); delta_pos.clear(); delta_val.clear();
    n_deleted = 0; rows_on_device = 0;
    for (size_t i = 0; i < n; ++i) {
        levels[i] = lv ? lv[i] : 0;
        if (del && del[i]) { deleted[i] = 1; ++n_deleted; } else live.insert(pids[i]);
    }
    if (has_entry) entry_point = entry; else entry_point = std::nullopt;
    current_max_level = cur_max;
    // rows go to the device as given (already in stored form)
    d_vecs.reserve(std::max<size_t>(n, 1) * stride, stream, false); d_ids.reserve(std::max<size_t>(n, 1), stream, false);
    if (n) {
        if (stride == dim) VDB_HIP(hipMemcpyAsync(d_vecs.p, vecs, n * dim * 4, hipMemcpyHostToDevice, stream));
        else {
            std::vector<float> padded(n * stride, 0.f);
            for (size_t i = 0; i < n; ++i) std::memcpy(&padded[i * stride], vecs + i * dim, dim * 4);
            VDB_HIP(hipMemcpyAsync(d_vecs.p, padded.data(), padded.size() * 4, hipMemcpyHostToDevice, stream));
            VDB_HIP(hipStreamSynchronize(stream));
        }
        VDB_HIP(hipMemcpyAsync(d_ids.p, pids, n * 8, hipMemcpyHostToDevice, stream));
        VDB_HIP(hipStreamSynchronize(stream));
    }
    rows_on_device = n; graph_dirty = true; deleted_dirty = true;
}
