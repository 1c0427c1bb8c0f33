// Scale-out behind the C ABI: one shard (an index or a segment with its own graph) per rank, one RCCL communicator per
// group.  A sharded search is the rank's own search, ONE all-gather of the per-shard top-k records and counts on the
// caller's stream, and the k-way merge by (sort_key, id) on every rank.  The reference holds a single in-memory segment
// (src/segment/segment.rs:11-18); sharding is the scaling axis this library adds.
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "index.hpp"

using namespace vdb;

struct vdb_shard_group {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    std::mutex mu;                                     // one exchange at a time per group (the staging buffers are shared)
    DevBuf<vdb_scored_point> local, gathered;          // nq x k, world x nq x k
    DevBuf<uint32_t> local_n, gathered_n;              // nq, world x nq
};

namespace {

template <class F> vdb_status guarded(F&& f) {
    try { f(); return VDB_OK; }
    catch (const Failure& e) { set_last_error(e.e.msg); return e.e.code; }
    catch (const std::bad_alloc&) { set_last_error("out of host memory"); return VDB_ERR_DEVICE; }
    catch (const std::exception& e) { set_last_error(std::string("internal error: ") + e.what()); return VDB_ERR_PANIC; }
}
void want(const void* p, const char* what) { if (!p) throw Failure(err_arg(std::string(what) + " is NULL")); }
void rccl(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw Failure(err_device(std::string(what) + ": " + ncclGetErrorString(r)));
}

// d_lists / d_counts: this rank's nq x k rows and nq counts (device).  Gathers them from every rank and merges.
void exchange(vdb_shard_group& g, const vdb_scored_point* d_lists, const uint32_t* d_counts, size_t nq, size_t k,
              vdb_scored_point* d_out, uint32_t* d_out_n, hipStream_t s) {
    if (nq == 0 || k == 0) return;
    VDB_HIP(hipSetDevice(g.device));
    const size_t w = (size_t)g.world;
    g.gathered.reserve(w * nq * k, s, false); g.gathered_n.reserve(w * nq, s, false);
    rccl(ncclAllGather(d_lists, g.gathered.p, nq * k * sizeof(vdb_scored_point), ncclChar, g.comm, s), "ncclAllGather (records)");
    rccl(ncclAllGather(d_counts, g.gathered_n.p, nq, ncclUint32, g.comm, s), "ncclAllGather (counts)");
    launch_topk_merge(g.gathered.p, g.gathered_n.p, (uint32_t)w, (uint32_t)nq, (uint32_t)k, d_out, d_out_n, s);
}

}  // namespace

extern "C" {

vdb_status vdb_shard_group_unique_id(void* out, size_t cap) {
    return guarded([&] {
        want(out, "out");
        if (cap < sizeof(ncclUniqueId)) throw Failure(err_arg("the unique id takes " + std::to_string(sizeof(ncclUniqueId)) + " bytes"));
        ncclUniqueId id;
        rccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(out, &id, sizeof(id));
    });
}

vdb_status vdb_shard_group_new(const void* unique_id, size_t id_bytes, int rank, int world, int device, vdb_shard_group** out) {
    return guarded([&] {
        want(unique_id, "unique_id"); want(out, "out");
        if (id_bytes != sizeof(ncclUniqueId)) throw Failure(err_arg("the unique id takes " + std::to_string(sizeof(ncclUniqueId)) + " bytes"));
        if (world < 1 || rank < 0 || rank >= world) throw Failure(err_arg("rank outside the world"));
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) throw Failure(err_device("no such HIP device; libvdbmi has no CPU path"));
        VDB_HIP(hipSetDevice(device));
        auto g = std::make_unique<vdb_shard_group>();
        g->rank = rank; g->world = world; g->device = device;
        ncclUniqueId id; std::memcpy(&id, unique_id, sizeof(id));
        rccl(ncclCommInitRank(&g->comm, world, id, rank), "ncclCommInitRank");
        *out = g.release();
    });
}

void vdb_shard_group_free(vdb_shard_group* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();
    if (g->comm) (void)ncclCommDestroy(g->comm);
    delete g;
}

int vdb_shard_group_size(const vdb_shard_group* g) {
    int n = 0;
    if (!g || ncclCommCount(g->comm, &n) != ncclSuccess) return 0;
    return n;
}
int vdb_shard_group_rank(const vdb_shard_group* g) {
    int r = -1;
    if (!g || ncclCommUserRank(g->comm, &r) != ncclSuccess) return -1;
    return r;
}

vdb_status vdb_shard_exchange_dev(vdb_shard_group* g, const vdb_scored_point* d_lists, const uint32_t* d_counts, size_t nq, size_t k,
                                  vdb_scored_point* d_out, uint32_t* d_out_n, void* stream) {
    return guarded([&] {
        want(g, "group"); want(d_lists, "d_lists"); want(d_counts, "d_counts"); want(d_out, "d_out"); want(d_out_n, "d_out_n");
        std::lock_guard<std::mutex> lk(g->mu);
        exchange(*g, d_lists, d_counts, nq, k, d_out, d_out_n, (hipStream_t)stream);
    });
}

vdb_status vdb_shard_search_batch_dev(vdb_shard_group* g, const vdb_index* x, const float* d_q, size_t nq, size_t top_k, size_t ef_search,
                                      vdb_scored_point* d_out, uint32_t* d_out_n, void* stream) {
    return guarded([&] {
        want(g, "group"); want(x, "index"); want(d_q, "d_queries"); want(d_out, "d_out"); want(d_out_n, "d_out_n");
        if (nq == 0 || top_k == 0) return;
        std::lock_guard<std::mutex> lk(g->mu);
        hipStream_t s = stream ? (hipStream_t)stream : x->stream;      // NULL: the index's own stream, for the search and the exchange alike
        VDB_HIP(hipSetDevice(g->device));
        g->local.reserve(nq * top_k, s, false); g->local_n.reserve(nq, s, false);
        const_cast<vdb_index*>(x)->search_batch_dev(d_q, nq, top_k, ef_search, g->local.p, g->local_n.p, nullptr, s);
        exchange(*g, g->local.p, g->local_n.p, nq, top_k, d_out, d_out_n, s);
    });
}

vdb_status vdb_shard_exact_knn_batch_dev(vdb_shard_group* g, const vdb_index* x, const float* d_q, size_t nq, size_t top_k,
                                         vdb_scored_point* d_out, uint32_t* d_out_n, void* stream) {
    return guarded([&] {
        want(g, "group"); want(x, "index"); want(d_q, "d_queries"); want(d_out, "d_out"); want(d_out_n, "d_out_n");
        if (nq == 0 || top_k == 0) return;
        std::lock_guard<std::mutex> lk(g->mu);
        hipStream_t s = stream ? (hipStream_t)stream : x->stream;
        VDB_HIP(hipSetDevice(g->device));
        g->local.reserve(nq * top_k, s, false); g->local_n.reserve(nq, s, false);
        const_cast<vdb_index*>(x)->exact_knn(d_q, nq, top_k, true, g->local.p, g->local_n.p, s);
        exchange(*g, g->local.p, g->local_n.p, nq, top_k, d_out, d_out_n, s);
    });
}

vdb_status vdb_shard_segment_search_batch_dev(vdb_shard_group* g, const vdb_segment* seg, const float* d_q, size_t nq, size_t top_k, size_t ef_search,
                                              const vdb_filter* f, vdb_scored_point* d_out, uint32_t* d_out_n, void* stream) {
    return guarded([&] {
        want(g, "group"); want(seg, "segment"); want(d_q, "d_queries"); want(d_out, "d_out"); want(d_out_n, "d_out_n");
        if (nq == 0 || top_k == 0) return;
        std::lock_guard<std::mutex> lk(g->mu);
        hipStream_t s = stream ? (hipStream_t)stream : seg->hnsw->stream;
        VDB_HIP(hipSetDevice(g->device));
        g->local.reserve(nq * top_k, s, false); g->local_n.reserve(nq, s, false);
        // a shard whose points are all deleted (segment.rs:104-107 is an error for ONE segment) simply contributes nothing
        try { seg->search_batch_dev(d_q, nq, top_k, ef_search, f, g->local.p, g->local_n.p, nullptr, s); }
        catch (const Failure& e) {
            if (e.e.code != VDB_ERR_SEARCH) throw;
            VDB_HIP(hipMemsetAsync(g->local_n.p, 0, nq * sizeof(uint32_t), s));
        }
        exchange(*g, g->local.p, g->local_n.p, nq, top_k, d_out, d_out_n, s);
    });
}

}  // extern "C"
