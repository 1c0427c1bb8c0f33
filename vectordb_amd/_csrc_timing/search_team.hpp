// HNSW search with a team of TW wavefronts per query (one workgroup of TW x 64 lanes): the same walk as the
// register-resident beam of search_kernel.hpp (hnsw.rs:333-368 descent, hnsw.rs:420-511 ef-beam), for launches that
// cannot fill the chip with one wavefront per query (single queries, small batches, batches of long rows).
//
// Wavefront 0 owns the ascending result list in its registers, picks the parents of a step and merges the candidates.
// Every wavefront of the team takes a share of the step's parents (parent i goes to wavefront i mod TW): it fetches
// their adjacency blocks, tests their neighbours against the visited table the team shares in LDS, scores the unvisited
// rows' slots into a pool in LDS; every wavefront then scores an equal share of the pool and wavefront 0 merges it.
// Four workgroup barriers per step: parents / counts / pool / scores published (one more per round of a hub's overflow chunks).  A step expands up to 4 x TW nodes; expanding w nodes per step in total is the same walk as the
// one-wavefront beam at width w (the candidates of a step reach the list in wavefront order instead of parent order,
// which only orders exactly equal keys differently).
// The upper-layer descent makes the same moves as greedy_fused(): every wavefront reads the current node's block, the
// team splits the scoring of its neighbours and the first improving neighbour in list order is taken.
#pragma once
#include "search_kernel.hpp"

namespace vdb {
namespace {

struct TeamCtl {
    uint32_t par[32];            // parents of the step, best first
    uint32_t np;                 // their number (0: the walk has ended)
    uint32_t forgot;             // the visited table replaced an id at some point (sticky)
    uint32_t cnt[8];             // candidates each wavefront left in its segment
    uint32_t hub[8];             // overflow chunks of high-degree parents each wavefront still has to visit
    uint32_t gj[2][8];           // descent: first improving list position found by each wavefront (two alternating sets)
    float gk[2][8];              //          and its key
    uint32_t stats[3];           // rows scored, nodes expanded, adjacency entries read, summed over the team
    uint32_t pad[3];
};

template <int G, int CH, int PASSES, int METRIC, int NC, int TW>
struct TeamSearcher : Searcher<MODE_SEARCH, G, CH, PASSES, false, METRIC, NC> {
    using B = Searcher<MODE_SEARCH, G, CH, PASSES, false, METRIC, NC>;
    using B::a; using B::gv; using B::cand; using B::craw; using B::perm; using B::tags; using B::R; using B::qr; using B::qlds;
    using B::rows_scored; using B::expanded; using B::adj_entries;
#ifdef VDB_TIMING
    using B::tacc; using B::tlast;
#endif
    static constexpr int GG = B::GG, CC = B::CC, PP = B::PP, NCC = B::NCC;
    static constexpr int SEG = 256;              // candidates a wavefront can leave per round (4 parents x 62 inline neighbours)
    TeamCtl* ctl = nullptr;
    uint32_t* segs = nullptr;                    // TW segments of SEG slots followed by SEG scores
    int wv = 0;

    __device__ __forceinline__ TeamSearcher(const SearchArgs& args) : B(args) {}

    // hnsw.rs:333-368 on the fused upper-layer blocks, the moves of greedy_fused(): all wavefronts follow the same
    // node, each scores its share of the chunk's live neighbours, the first improving one in list order is taken.
    __device__ __forceinline__ uint32_t greedy_team(uint32_t cur, uint32_t& ref, uint32_t level, float& kcur, uint32_t& parity) {
        const int lane = lane_id();
        bool changed = true; int steps = 0;
        while (changed && steps < 1000) {
            ++steps; changed = false;
            if (level > (ref >> UREF_BITS)) break;                               // no list at this level
            const size_t blk = ((size_t)(ref & ((1u << UREF_BITS) - 1u)) + level - 1) * ADJ_BLOCK;
            const uint32_t w = gv.ublk[blk + lane], wr = gv.ublk_ref[blk + lane];
            const uint32_t deg = readlane_u(w, 0);
            if (deg == 0) break;
            if (wv == 0) { ++expanded; adj_entries += deg; }
            const uint32_t nchunks = deg <= ADJ_INLINE ? 1u : 1u + (deg - ADJ_INLINE + 63u) / 64u;
            const uint32_t* rest = gv.pool + readlane_u(w, 63);
            for (uint32_t c = 0; c < nchunks && !changed; ++c) {
                uint32_t nb, nr; bool in;
                if (c == 0) { nb = w; nr = wr; in = (uint32_t)(lane - 1) < (deg < ADJ_INLINE ? deg : ADJ_INLINE); }
                else {
                    const uint32_t i = (c - 1) * 64u + lane;
                    in = ADJ_INLINE + i < deg;
                    nb = in ? rest[i] : 0u;
                    nr = in ? make_uref(gv.upper_base[nb], gv.upper_cnt[nb]) : 0u;
                }
                const bool ok = in && !B::is_deleted(nb);
                const uint64_t mk = ballot(ok);
                const int m = popc64(mk);
                if (m == 0) continue;                                            // the same on every wavefront
                const uint32_t rnk = rank_below(mk);
                if (ok) cand[rnk] = nb;
                wave_sync();
                const int per = (m + TW - 1) / TW, lo = wv * per;
                const int cnt = lo < m ? (m - lo < per ? m - lo : per) : 0;
                if (cnt > 0) {
                    score_cands_fast<GG, CC, PP, METRIC>(gv, qr, cand + lo, cnt, craw + lo);
                    wave_sync();
                    rows_scored += cnt;
                }
                const bool mine = lane >= lo && lane < lo + cnt;
                const float kl = mine ? B::key_of(craw[lane]) : 0.f;
                const uint64_t bm = ballot(mine && kl < kcur);
                uint32_t j = NONE32; float kj = 0.f;
                if (bm) { j = (uint32_t)(__ffsll((unsigned long long)bm) - 1); kj = readlane_f(kl, (int)j); }
                if (lane == 0) { ctl->gj[parity][wv] = j; ctl->gk[parity][wv] = kj; }
                __syncthreads();
                uint32_t bj = NONE32; float bk = 0.f;
#pragma unroll
                for (int t = TW - 1; t >= 0; --t) { const uint32_t jj = ctl->gj[parity][t]; if (jj != NONE32) { bj = jj; bk = ctl->gk[parity][t]; } }
                parity ^= 1u;
                if (bj != NONE32) {
                    const uint64_t src = ballot(ok && rnk == bj);                // the lane that holds candidate bj
                    const int sl = __ffsll((unsigned long long)src) - 1;
                    cur = readlane_u(nb, sl); ref = readlane_u(nr, sl); kcur = bk; changed = true;
                }
                wave_sync();
            }
        }
        return cur;
    }

    // One block of up to 64 candidates (lane j holds candidate j: key kj, slot sj) merged into the register list of
    // wavefront 0: hnsw.rs:486-499 for each candidate in lane order.
    __device__ __forceinline__ void merge_block(bool have, float kj, uint32_t sj, float (&rk)[NCC], uint32_t (&rv)[NCC],
                                                uint32_t& n, float& worst, const uint32_t cap, const bool forgot) {
        const int lane = lane_id();
        const float INF = __int_as_float(0x7f800000);
        bool pass = have && (n < cap || kj < worst);                      // hnsw.rs:486
        uint64_t pm = ballot(pass);
        if (pm == 0) return;
        if (forgot) {
            // a candidate the visited table had forgotten may already be in the list (same key, same slot) or sit twice
            // among the candidates
            uint64_t dup = 0, it = pm;
            while (it) {
                const int l = __ffsll((unsigned long long)it) - 1; it &= it - 1;
                const float k = readlane_f(kj, l);
                const uint64_t before = pm & ((1ull << l) - 1ull);
                uint64_t eq = ballot(kj == k) & before;
#pragma unroll
                for (int c = 0; c < NCC; ++c) eq |= ballot(rk[c] == k);
                if (eq) {
                    const uint32_t s = readlane_u(sj, l);
                    uint64_t hit = ballot(kj == k && sj == s) & before;
#pragma unroll
                    for (int c = 0; c < NCC; ++c) hit |= ballot(rk[c] == k && (rv[c] & ~FLAG_EXPANDED) == s);
                    if (hit) dup |= 1ull << l;
                }
            }
            pm &= ~dup;
            pass = pass && !((dup >> lane) & 1ull);
            if (pm == 0) return;
        }
        if (!pass) kj = INF;
        const uint32_t mp = popc64(pm);
        uint32_t sh[NCC], mypos = 0;
#pragma unroll
        for (int c = 0; c < NCC; ++c) sh[c] = 0;
        uint64_t it = pm;
        while (it) {
            const int l = __ffsll((unsigned long long)it) - 1; it &= it - 1;
            const float k = readlane_f(kj, l);
            uint32_t pos = 0;
#pragma unroll
            for (int c = 0; c < NCC; ++c) {
                const bool gt = rk[c] > k;                                // existing entries that move back by one
                pos += WAVE - popc64(ballot(gt));
                sh[c] += gt ? 1u : 0u;
            }
            const uint64_t lt = ballot(kj < k), eq = ballot(kj == k) & pm;
            pos += popc64(lt | (eq & ((1ull << l) - 1ull)));              // candidates among themselves: lane order
            if (lane == l) mypos = pos;
        }
#pragma unroll
        for (int c = 0; c < NCC; ++c) {
            const uint32_t dst = c * WAVE + lane + sh[c];
            if (dst < NCC * WAVE) perm[dst] = make_uint2(__float_as_uint(rk[c]), rv[c]);
        }
        if (pass && mypos < NCC * WAVE) perm[mypos] = make_uint2(__float_as_uint(kj), sj);
        n = (n + mp < cap) ? n + mp : cap;
        wave_sync();
#pragma unroll
        for (int c = 0; c < NCC; ++c) {
            const uint2 e = perm[c * WAVE + lane];
            rk[c] = __uint_as_float(e.x); rv[c] = e.y;
        }
        // the list ends at `cap`, wherever in the NCC chunks that is
        float wk = INF;
#pragma unroll
        for (int c = 0; c < NCC; ++c) {
            if ((uint32_t)(c * WAVE + lane) >= cap) { rk[c] = INF; rv[c] = FLAG_EXPANDED; }
            if ((uint32_t)c == ((cap - 1) >> 6)) wk = rk[c];
        }
        wave_sync();
        worst = (n >= cap) ? readlane_f(wk, (int)((cap - 1) & 63u)) : INF;
    }

    // hnsw.rs:420-511 on layer 0.  `width` nodes are expanded per step in total (at most 4 per wavefront).
    __device__ __forceinline__ void beam_team(uint32_t entry, uint32_t ef, uint32_t width) {
        constexpr int EMAX = 4;
        const int lane = lane_id();
        const float INF = __int_as_float(0x7f800000);
        const bool full = gv.stride == (uint32_t)(GG * CC * 4);
        const TagSet ts = tags;
        const uint32_t* const adj0 = gv.adj0;
        float rk[NCC]; uint32_t rv[NCC];
#pragma unroll
        for (int c = 0; c < NCC; ++c) { rk[c] = INF; rv[c] = FLAG_EXPANDED; }
        const uint32_t cap = ef;
        uint32_t n = 0;
        bool forgot = false;
        float worst = INF;
        if (width > (uint32_t)(EMAX * TW)) width = EMAX * TW;
        if (width == 0) width = 1;
        // the beam's candidates live in one pool shared by the team: TW * SEG slots followed by as many scores
        __syncthreads();                                                  // the descent used one segment per wavefront: nobody is still reading its own
        cand = segs; craw = reinterpret_cast<float*>(segs + (size_t)TW * SEG);
        if (wv == 0) {
            uint32_t start = entry;
            if (B::is_deleted(entry) && a.fallback_live != NONE32) start = a.fallback_live;      // hnsw.rs:440-444
            if (lane == 0) { cand[0] = start; ts.insert(start); }
            wave_sync();
            if (full) B::template score_rows<true>(1); else B::template score_rows<false>(1);
            rows_scored += 1;
            const float k0 = B::key_of(craw[0]);
            if (lane == 0) { rk[0] = k0; rv[0] = start; }
            n = 1;
            if (cap == 1) worst = k0;
            wave_sync();
        }
#ifdef VDB_TIMING
        B::tlast = clock64();
        B::tacc[6] = 0; B::tacc[7] = 0; B::tacc[3] = 0;
        B::tacc[9] = wall_clock64();
#endif
        for (;;) {
            if (wv == 0) {
                // ---- the `width` best unexpanded entries, in list order ----
                uint32_t seen = 0;
#pragma unroll
                for (int c = 0; c < NCC; ++c) {
                    if (seen < width && (uint32_t)(c * WAVE) < n) {
                        const bool un = (int32_t)rv[c] >= 0;
                        const uint64_t um = ballot(un);
                        const uint32_t rnk = seen + rank_below(um);
                        const bool isp = un && rnk < width;
                        if (isp) { ctl->par[rnk] = rv[c]; rv[c] |= FLAG_EXPANDED; }
                        seen += popc64(um);
                    }
                }
                if (lane == 0) ctl->np = seen < width ? seen : width;
            }
            __syncthreads();                                              // parents published
            VDB_T_MARK(0);
            const uint32_t np = ctl->np;
            if (np == 0) break;
            const uint32_t my_np = np > (uint32_t)wv ? (np - (uint32_t)wv + TW - 1) / TW : 0u;
            uint32_t w[EMAX];
#pragma unroll
            for (int e = 0; e < EMAX; ++e) {
                const uint32_t p = ctl->par[(uint32_t)e < my_np ? e * TW + wv : 0];
                w[e] = adj0[(size_t)p * ADJ_BLOCK + lane];
            }
            uint32_t hub_e = 0, hub_ch = 1, hub_left = 0;
            int m = 0;
            uint32_t nb[EMAX];
            bool fresh[EMAX];
            {   // ---- the inline neighbours (block words 1..62) of this wavefront's parents at once ----
                uint32_t w0[EMAX], tag[EMAX], widx[EMAX], expect[EMAX], desired[EMAX], old[EMAX], half[EMAX];
                uint2 bk[EMAX];
                bool tr[EMAX], slow[EMAX], full_b[EMAX];
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    nb[e] = w[e];
                    ts.locate(nb[e], w0[e], tag[e]);
                    bk[e] = *reinterpret_cast<const uint2*>(ts.tab + w0[e]);
                }
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    uint32_t deg = (uint32_t)e < my_np ? readlane_u(w[e], 0) : 0u;
                    if (deg) { ++expanded; adj_entries += deg; }
                    hub_left += deg > ADJ_INLINE ? (deg - ADJ_INLINE + 63u) / 64u : 0u;
                    deg = deg < ADJ_INLINE ? deg : ADJ_INLINE;
                    bool want = (uint32_t)(lane - 1) < deg;                           // words 1..deg hold neighbours
                    if (gv.deleted) want = want && !bit_test(gv.deleted, nb[e]);      // hnsw.rs:475
                    const bool pres = TagSet::present(bk[e], tag[e]);
                    const uint32_t u = TagSet::used(bk[e]);
                    tr[e] = want & !pres;
                    full_b[e] = tr[e] & (u == TagSet::WAYS);                          // full bucket: one of its ids is forgotten
                    const uint32_t en = TagSet::entry_for(u, tag[e]);
                    TagSet::slot_of(bk[e], en, tag[e], widx[e], expect[e], desired[e]);
                    half[e] = (en & 1u) * 16u;
                }
#pragma unroll
                for (int e = 0; e < EMAX; ++e) { old[e] = 0; if (tr[e]) old[e] = atomicCAS(&ts.tab[w0[e] + widx[e]], expect[e], desired[e]); }
                bool any_slow = false;
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    fresh[e] = tr[e] & (old[e] == expect[e]);
                    // the word changed under us: if our entry now holds our tag, another lane (of any wavefront of the
                    // team) carried the same id and it counts as visited
                    const bool same = ((old[e] >> half[e]) & 0xFFFFu) == tag[e];
                    slow[e] = tr[e] & (old[e] != expect[e]) & !same;
                    any_slow = any_slow | slow[e] | full_b[e];
                }
                if (ballot(any_slow)) {
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) {
                        int r = 0;
                        if (ballot(slow[e])) {
                            if (slow[e]) r = ts.insert(nb[e]);
                            if (slow[e]) fresh[e] = r != 0;
                        }
                        forgot = forgot | (ballot(full_b[e] | (r == 2)) != 0);
                    }
                }
#pragma unroll
                for (int e = 0; e < EMAX; ++e) m += popc64(ballot(fresh[e]));
            }
            VDB_T_MARK(1);
            // ---- rounds of the step: the team's unvisited neighbours go into one pool (wavefront order, then parent
            // order, then list order), every wavefront scores an equal share of it, wavefront 0 merges it ----
            for (;;) {
                if (lane == 0) { ctl->cnt[wv] = (uint32_t)m; ctl->hub[wv] = hub_left; if (forgot) ctl->forgot = 1u; }
                __syncthreads();                                          // counts published
                uint32_t total_hub = 0, off = 0, total = 0;
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    const uint32_t ct = ctl->cnt[t];
                    total_hub += ctl->hub[t];
                    off += t < wv ? ct : 0u;
                    total += ct;
                }
                if (total) {
                    uint32_t pos = off;
#pragma unroll
                    for (int e = 0; e < EMAX; ++e) {
                        const uint64_t mk = ballot(fresh[e]);
                        if (fresh[e]) cand[pos + rank_below(mk)] = nb[e];
                        pos += popc64(mk);
                    }
                    __syncthreads();                                      // the pool is complete
                    const int per = (int)((total + TW - 1) / TW), lo = wv * per;
                    const int cnt = lo < (int)total ? ((int)total - lo < per ? (int)total - lo : per) : 0;
                    if (cnt > 0) {
                        rows_scored += cnt;
                        cand += lo; craw += lo;
                        if (full) B::template score_rows<true>(cnt); else B::template score_rows<false>(cnt);
                        cand -= lo; craw -= lo;
                    }
                    VDB_T_MARK(3);
                    __syncthreads();                                      // scores published
                    VDB_T_MARK(2);
                    if (wv == 0) {
                        forgot = forgot | (ctl->forgot != 0u);
                        for (int base = 0; base < (int)total; base += WAVE) {
                            const int j = base + lane;
                            const bool have = j < (int)total;
                            const float kj = have ? B::key_of(craw[j]) : INF;
                            const uint32_t sj = have ? cand[j] : 0u;
                            merge_block(have, kj, sj, rk, rv, n, worst, cap, forgot);
                        }
                    }
                    VDB_T_MARK(4);
                }
                if (total_hub == 0) break;
                __syncthreads();                                          // the merge has read the pool, every wavefront the counts: both are written again below
                m = 0;
#pragma unroll
                for (int e = 0; e < EMAX; ++e) fresh[e] = false;
                if (hub_left) {                                           // one overflow chunk of a high-degree parent
                    --hub_left;
                    Adj A;
                    for (;;) {
                        uint32_t we = w[0];
#pragma unroll
                        for (int t = 1; t < EMAX; ++t) if (hub_e == (uint32_t)t) we = w[t];
                        A = open_adj0(gv, we);
                        if (hub_ch < A.chunks()) break;
                        ++hub_e; hub_ch = 1;
                    }
                    uint32_t nbh;
                    const bool in = A.get(hub_ch, nbh);
                    ++hub_ch;
                    bool fr = in && !B::is_deleted(nbh);
                    int r = 0;
                    if (fr) r = ts.insert(nbh);
                    fr = r != 0;
                    forgot = forgot | (ballot(r == 2) != 0);
                    nb[0] = nbh; fresh[0] = fr;
                    m = popc64(ballot(fr));
                }
            }
        }
        if (wv == 0) {                                                    // hand the list over in LDS (Euclidean: root of the squared keys)
#pragma unroll
            for (int c = 0; c < NCC; ++c) {
                const uint32_t i = c * WAVE + lane;
                if (i < n) { R.key[i] = (B::metric() == VDB_EUCLIDEAN) ? __builtin_sqrtf(rk[c]) : rk[c]; R.val[i] = rv[c]; }
            }
            R.n = n; R.cap = cap;
            wave_sync();
        }
    }

    __device__ __forceinline__ void run_team(uint32_t q, char* smem) {
        const int lane = lane_id();
        const uint32_t tid = threadIdx.x, nthr = WAVE * TW;
        wv = (int)(tid >> 6);
        // ---- LDS: [list staging / query / final list] [control] [TW candidate segments] [visited table] ----
        size_t area = (size_t)NCC * WAVE * 8;
        if ((size_t)gv.stride * 4 > area) area = (size_t)gv.stride * 4;
        area = (area + 15) & ~(size_t)15;
        perm = reinterpret_cast<uint2*>(smem);
        qlds = reinterpret_cast<float*>(smem);
        R.key = reinterpret_cast<float*>(smem); R.val = reinterpret_cast<uint32_t*>(smem) + NCC * WAVE;
        R.key2 = nullptr; R.val2 = nullptr;
        ctl = reinterpret_cast<TeamCtl*>(smem + area);
        segs = reinterpret_cast<uint32_t*>(smem + area + sizeof(TeamCtl));
        cand = segs + (size_t)wv * SEG * 2;
        craw = reinterpret_cast<float*>(cand + SEG);
        tags.tab = segs + (size_t)TW * SEG * 2;
        tags.idmask = (1u << (a.tag_log2 + 13)) - 1;

        if (tid == 0) a.out_n[q] = 0;
        if (a.entry == NONE32) return;                                    // hnsw.rs:515-518
#ifdef VDB_TIMING
        const unsigned long long t_begin = wall_clock64();
#endif
        // ---- query -> LDS (normalised for Cosine: hnsw.rs:526-536, 752-764), control words and visited table cleared ----
        const float* qsrc = a.queries + (size_t)q * gv.dim;
        for (uint32_t i = tid; i < gv.stride; i += nthr) qlds[i] = (i < gv.dim) ? qsrc[i] : 0.f;
        if (tid < sizeof(TeamCtl) / 4) reinterpret_cast<uint32_t*>(ctl)[tid] = 0u;
        {
            const uint32_t words = (tags.idmask >> 15) * 2 + 2;
            for (uint32_t i = tid * 4; i < words; i += nthr * 4) *reinterpret_cast<uint4*>(tags.tab + i) = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        if (B::metric() == VDB_COSINE) {
            float s = -0.0f;                                              // every lane runs the same sequential sum
            for (uint32_t i = 0; i < gv.dim; ++i) { float p = qlds[i] * qlds[i]; s = s + p; }
            const float norm = __builtin_sqrtf(s);
            __syncthreads();
            if (norm != 0.0f) for (uint32_t i = tid; i < gv.dim; i += nthr) qlds[i] = qlds[i] / norm;
            __syncthreads();
        }
        qr.load(qlds, gv.stride, B::metric());
        __syncthreads();                                                  // the query's area is reused from here on

        uint32_t cur = a.entry;
        if (gv.ublk) {
            uint32_t ref = make_uref(gv.upper_base[cur], gv.upper_cnt[cur]);
            if (a.max_level >= 1) {
                if (lane == 0) cand[0] = cur;
                wave_sync();
                score_cands_fast<GG, CC, PP, METRIC>(gv, qr, cand, 1, craw);
                wave_sync();
                if (wv == 0) rows_scored += 1;
                float kc = B::key_of(craw[0]);
                uint32_t parity = 0;
                for (uint32_t l = a.max_level; l >= 1; --l) cur = greedy_team(cur, ref, l, kc, parity);
            }
        } else {
            // no fused upper blocks on this index: wavefront 0 descends alone, the team waits for the node it reaches
            if (wv == 0) {
                for (uint32_t l = a.max_level; l >= 1; --l) cur = B::greedy(cur, l);           // hnsw.rs:538-541
                if (lane == 0) ctl->par[0] = cur;
            }
            __syncthreads();
            cur = ctl->par[0];
            __syncthreads();
        }
        beam_team(cur, a.ef, a.team_width);
        if (wv == 0) B::emit(q, R, false);
#ifdef VDB_TIMING
        // diagnostic build only: wavefront 0's phase clocks (0 select + wait for the team, 1 adjacency + visited test, 3 scoring,
        // 2 wait for the team's candidates, 4 merge) overwrite this query's result records; lifetime and start tick replace two counters
        __syncthreads();
        if (tid == 0 && a.out && a.top_k >= 5) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.out + (size_t)q * a.top_k);
            B::tacc[9] -= t_begin; B::tacc[5] = wall_clock64() - t_begin;
            for (int i = 0; i < 10; ++i) dbg[i] = B::tacc[i];
        }
        if (a.stats) { __syncthreads(); if (tid == 0) { a.stats[q] = {(uint32_t)t_begin, expanded, (uint32_t)(wall_clock64() - t_begin), 0u}; } return; }
#endif
        if (a.stats) {
            if (lane == 0) { atomicAdd(&ctl->stats[0], rows_scored); atomicAdd(&ctl->stats[1], expanded); atomicAdd(&ctl->stats[2], adj_entries); }
            __syncthreads();
            if (tid == 0) a.stats[q] = {ctl->stats[0], ctl->stats[1], ctl->stats[2], 0u};
        }
    }
};

template <int G, int CH, int PASSES, int METRIC, int NC, int TW>
__global__ void __launch_bounds__(64 * TW) search_team_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TeamSearcher<G, CH, PASSES, METRIC, NC, TW> s(a);
    s.run_team(blockIdx.x, smem);
}

// 64-entry chunks of a team's register list: 2, 4 or 8 (ef up to 128 / 256 / 512)
inline uint32_t team_list_chunks(uint32_t ef) { return ef <= 128 ? 2u : ef <= 256 ? 4u : 8u; }

inline size_t team_lds_bytes(const SearchArgs& a) {
    size_t area = std::max<size_t>((size_t)team_list_chunks(a.ef) * 512, (size_t)a.g.stride * 4);
    area = (area + 15) & ~(size_t)15;
    return area + sizeof(TeamCtl) + (size_t)a.team * 256 * 8 + ((size_t)2 << a.tag_log2);
}

template <int METRIC, int NC, int TW>
void launch_team_shape(const SearchArgs& a, size_t lds, hipStream_t st) {
    const uint32_t chunks = a.g.stride / 4;
    if (chunks <= 32) hipLaunchKernelGGL((search_team_kernel<8, 4, 2, METRIC, NC, TW>), dim3(a.nq), dim3(64 * TW), lds, st, a);
    else if (chunks <= 256) hipLaunchKernelGGL((search_team_kernel<64, 4, 2, METRIC, NC, TW>), dim3(a.nq), dim3(64 * TW), lds, st, a);
    else if (chunks <= 512) hipLaunchKernelGGL((search_team_kernel<64, 8, 2, METRIC, NC, TW>), dim3(a.nq), dim3(64 * TW), lds, st, a);
    else throw Failure(err_arg("dimension above 2048 is not supported"));
}

template <int METRIC>
void launch_team_metric(const SearchArgs& a, hipStream_t st) {
    const size_t lds = team_lds_bytes(a);
    const uint32_t nc = team_list_chunks(a.ef);
    if (a.team == 2) {
        if (nc == 2) launch_team_shape<METRIC, 2, 2>(a, lds, st); else if (nc == 4) launch_team_shape<METRIC, 4, 2>(a, lds, st); else launch_team_shape<METRIC, 8, 2>(a, lds, st);
    } else {
        if (nc == 2) launch_team_shape<METRIC, 2, 4>(a, lds, st); else if (nc == 4) launch_team_shape<METRIC, 4, 4>(a, lds, st); else launch_team_shape<METRIC, 8, 4>(a, lds, st);
    }
}

}  // namespace
}  // namespace vdb
