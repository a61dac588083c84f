// group.hip -- multi-device groups: the ilm_group_* entry points of include/illuminant_hip.h.
//
// The reference is single-GPU (one GraphicsDevice; LightingRenderer.RenderLighting returns ONE lightmap per frame,
// Illuminant/Lighting/LightingRenderer.cs:917-923,1004-1010; ParticleSystem chunks never interact,
// Illuminant/Particles/ParticleSystem.cs:743-745).  This file is what lets a host keep those call sites while the frame's row
// strips and the particle chunks live on the 8 GPUs of a node: a group owns one context per member, cuts a frame into equal
// slots of whole 16-row tile bands, renders each strip with the single-device entry point on its member's stream and exchanges
// the strips in place -- hipMemcpyPeerAsync fan-out over the xGMI mesh inside one process, ncclAllGather (RCCL) inside or across
// processes.  Everything here is host code on top of the C ABI of api.hip; no kernel of its own.
//
// RCCL is bound at run time (dlopen of librccl.so.1, which resolves to an already loaded copy -- PyTorch brings its own), so
// the library loads and every single-device path works on a machine without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
// The handful of RCCL types the bindings below need, declared here instead of including <rccl/rccl.h>: the library is bound at run time,
// so its headers are no build dependency either.  ABI facts of nccl.h 2.x / rccl.h: an opaque communicator pointer, a
// 128-byte unique id passed by value, int-sized enums with ncclSuccess == 0 and ncclInt8 == 0.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0 } ncclDataType_t;

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <algorithm>
#include "internal.hpp"

namespace {

using namespace ilm;

#define HIP_TRY(expr)                                                                                                          \
    do {                                                                                                                       \
        hipError_t _e = (expr);                                                                                                \
        if (_e != hipSuccess)                                                                                                  \
            return api_fail((int32_t)_e, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);           \
    } while (0)

constexpr int kTileRows = 16;        // the sphere-light kernel works on 16 x 16 pixel tiles (lighting.hip)
constexpr int kMaxMembers = 64;

// ---- RCCL, bound lazily ------------------------------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "";
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // ILM_RCCL_LIB: another library with the same eleven entry points -- tests/fake_rccl.cpp, the shared-memory stand-in that lets
        // several rank processes share the build box's ONE GPU (RCCL itself refuses that); never set in production
        const char* override_path = getenv("ILM_RCCL_LIB");
        const char* names[] = { override_path ? override_path : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib || override_path) break;
        }
        if (!r.lib) { snprintf(r.why, sizeof(r.why), "librccl.so not found: %s", dlerror()); return; }
#define ILM_BIND(field, symbol)                                                                      \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, symbol));                         \
        if (!r.field) { snprintf(r.why, sizeof(r.why), "librccl.so lacks %s", symbol); return; }
        ILM_BIND(GetUniqueId, "ncclGetUniqueId") ILM_BIND(CommInitRank, "ncclCommInitRank") ILM_BIND(CommInitAll, "ncclCommInitAll")
        ILM_BIND(CommDestroy, "ncclCommDestroy") ILM_BIND(CommCount, "ncclCommCount") ILM_BIND(AllGather, "ncclAllGather")
        ILM_BIND(GroupStart, "ncclGroupStart") ILM_BIND(GroupEnd, "ncclGroupEnd") ILM_BIND(Send, "ncclSend") ILM_BIND(Recv, "ncclRecv") ILM_BIND(GetErrorString, "ncclGetErrorString")
#undef ILM_BIND
        r.ok = true;
    });
    return r;
}

#define NCCL_TRY(expr)                                                                                                         \
    do {                                                                                                                       \
        ncclResult_t _r = (expr);                                                                                              \
        if (_r != ncclSuccess)                                                                                                 \
            return api_fail(ILM_ERR_STATE, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__);     \
    } while (0)

struct Group {
    uint32_t magic = kMagicGroup;
    int n_local = 0, world = 0, first_rank = 0;
    bool rank_mode = false;                 // one process per GPU: the group spans processes, exactly one local member
    int children = 0;                       // live group lightmaps
    std::vector<int> devices;
    std::vector<IlmHandle> ctx;
    // a member's context stream, joined with the context's second stepping stream (api.hip, Ctx::main) at every use
    hipStream_t stream(size_t i) const { return ctx_stream_joined(ctx[i]); }
    std::vector<hipEvent_t> events;         // one per local member: "my pushes have been queued / finished" (peer fan-out)
    std::vector<ncclComm_t> comms;          // one per local member once a communicator exists
    bool duplicates = false;                // two members share a device (RCCL refuses that)
    bool peers_ok = true;                   // every pair of distinct local devices can address each other's memory (store mode needs it)
    void* d_counts = nullptr; size_t counts_bytes = 0;     // scratch of ilm_group_live_counts (rank mode), on member 0's device
    // IPC mappings of other ranks' buffers whose group lightmap is gone: unmapped when the GROUP goes, not before (set_store_mode)
    std::vector<void*> ipc_retired;
    std::vector<void*> ipc_decoys;          // (ILM_EXP_IPC_BOGUS_MAPPING, the proof's negative control)
    uint32_t ipc_proofs = 0;                // serial number of prove_ipc_mappings (the same on every rank: a collective)
    // ILM_GATHER_ASYNC: a second stream per local member that carries the exchanges, so that the member's context stream can go on with
    // the next frame's strip (created on first use); its own scratch events for the peer fan-out
    std::vector<hipStream_t> xstream;
    std::vector<hipEvent_t> xfork, xevents;
    hipStream_t exchange_stream(size_t i, bool async) const { return async ? xstream[i] : stream(i); }
};

struct GroupLightmap {
    uint32_t magic = kMagicGroupLightmap;
    Group* group = nullptr;
    int width = 0, height = 0, format = 0, slot_rows = 0;
    size_t row_bytes = 0;
    // rows [begin[r], end[r]) rank r renders.  Default: the equal padded slots (one in-place all-gather).  ilm_group_lightmap_set_strips
    // installs other contiguous strips of whole tile bands (cost-balanced ones): they are exchanged range by range (exchange_rows).
    std::vector<int> begin, end;
    bool equal_slots = true;
    std::vector<void*> buffers;             // per local member: world * slot_rows rows
    std::vector<IlmHandle> lightmaps;       // per local member: lightmap object aliasing the buffer
    bool store_mode = false;                // ILM_GATHER_STORE armed: every member's light passes also store into the other members' buffers
    std::vector<void*> ipc_peers;           // store mode of a group that spans processes: the other ranks' buffers, IPC-mapped (hipIpcOpenMemHandle)
    std::vector<hipEvent_t> xdone;          // ILM_GATHER_ASYNC: per local member, "the exchange queued last has finished" (ilm_group_lightmap_wait)
    std::vector<bool> xpending;
};

Group* group_from(IlmHandle h) { return handle_is_live(h, kMagicGroup) ? reinterpret_cast<Group*>(static_cast<uintptr_t>(h)) : nullptr; }
GroupLightmap* glm_from(IlmHandle h) {
    return handle_is_live(h, kMagicGroupLightmap) ? reinterpret_cast<GroupLightmap*>(static_cast<uintptr_t>(h)) : nullptr;
}

size_t texel_bytes(int format) { return format == ILM_LIGHTMAP_FLOAT4 ? 16 : (format == ILM_LIGHTMAP_HALF4 ? 8 : 4); }

// padded equal slots: R rows each, a multiple of the tile height, world * R >= height
int slot_rows_for(int height, int world) {
    const int per = (height + world - 1) / world;
    return (per + kTileRows - 1) / kTileRows * kTileRows;
}

int32_t make_members(Group* g) {
    for (int i = 0; i < g->n_local; i++) {
        IlmHandle c = 0;
        const int32_t rc = ilm_ctx_create(g->devices[(size_t)i], &c);
        if (rc != ILM_OK) return rc;
        g->ctx.push_back(c);
        HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g->events.push_back(e);
    }
    for (int i = 0; i < g->n_local; i++)
        for (int j = 0; j < i; j++)
            if (g->devices[(size_t)i] == g->devices[(size_t)j]) g->duplicates = true;
    // direct xGMI transfers between distinct local devices (already enabled / unsupported pairs are not errors: the copy then stages)
    for (int i = 0; i < g->n_local; i++)
        for (int j = 0; j < g->n_local; j++) {
            const int a = g->devices[(size_t)i], b = g->devices[(size_t)j];
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
                (void)hipSetDevice(a);
                const hipError_t pe = hipDeviceEnablePeerAccess(b, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) g->peers_ok = false;
            } else {
                g->peers_ok = false;
            }
            (void)hipGetLastError();
        }
    return ILM_OK;
}

void release_group(Group* g) {
    if (!g->ipc_retired.empty()) {
        (void)hipSetDevice(g->devices[0]);
        for (void* p : g->ipc_retired) if (p) (void)hipIpcCloseMemHandle(p);
        g->ipc_retired.clear();
    }
    for (void* p : g->ipc_decoys) { (void)hipSetDevice(g->devices[0]); (void)hipFree(p); }
    for (size_t i = 0; i < g->comms.size(); i++)
        if (g->comms[i] && rccl().ok) {
            (void)hipSetDevice(g->devices[i]);
            (void)rccl().CommDestroy(g->comms[i]);
        }
    for (size_t i = 0; i < g->xstream.size(); i++) {
        (void)hipSetDevice(g->devices[i]);
        if (g->xstream[i]) { (void)hipStreamSynchronize(g->xstream[i]); (void)hipStreamDestroy(g->xstream[i]); }
        if (i < g->xfork.size() && g->xfork[i]) (void)hipEventDestroy(g->xfork[i]);
        if (i < g->xevents.size() && g->xevents[i]) (void)hipEventDestroy(g->xevents[i]);
    }
    if (g->d_counts) { (void)hipSetDevice(g->devices[0]); (void)hipFree(g->d_counts); }
    for (size_t i = 0; i < g->events.size(); i++) {
        (void)hipSetDevice(g->devices[i]);
        if (g->events[i]) (void)hipEventDestroy(g->events[i]);
    }
    for (IlmHandle c : g->ctx)
        if (c) (void)ilm_ctx_destroy(c);
    handle_retire(g);
    delete g;
}

// the in-process communicator is created on first use (a group that only uses the peer fan-out never needs RCCL)
int32_t ensure_comms(Group* g) {
    if (!g->comms.empty()) return ILM_OK;
    Rccl& r = rccl();
    if (!r.ok) return api_fail(ILM_ERR_STATE, "RCCL is not available: %s", r.why);
    if (g->rank_mode) return api_fail(ILM_ERR_STATE, "this group has no communicator");     // created by ilm_group_create_rank
    if (g->duplicates)
        return api_fail(ILM_ERR_INVALID_ARGUMENT, "RCCL needs one device per member; this group has several members on one device (use ILM_GATHER_PEER)");
    std::vector<ncclComm_t> comms((size_t)g->n_local, nullptr);
    NCCL_TRY(r.CommInitAll(comms.data(), g->n_local, g->devices.data()));
    g->comms = comms;
    return ILM_OK;
}

// In-place all-gather; see the header.  All work is stream-ordered on the members' context streams.
int32_t all_gather(Group* g, void* const* buffers, size_t bytes, int gather, bool async = false) {
    if (gather == ILM_GATHER_NONE || g->world == 1 || bytes == 0) return ILM_OK;
    std::vector<hipEvent_t>& events = async ? g->xevents : g->events;
    auto S = [&](size_t i) { return g->exchange_stream(i, async); };
    if (gather == ILM_GATHER_PEER) {
        if (g->rank_mode)
            return api_fail(ILM_ERR_INVALID_ARGUMENT, "ILM_GATHER_PEER needs every member in this process; a group that spans processes gathers with RCCL");
        const int n = g->n_local;
        // 1. nobody may overwrite a slot that its owner's earlier work (the previous frame's consumers) still reads, and nobody's push may
        //    start before the destination has finished the work queued before this call: every stream waits for every other's "here" event
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            HIP_TRY(hipEventRecord(events[(size_t)i], S((size_t)i)));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            for (int j = 0; j < n; j++)
                if (j != i) HIP_TRY(hipStreamWaitEvent(S((size_t)i), events[(size_t)j], 0));
        }
        // 2. member i pushes its slot to the n - 1 others on its own stream: on a full xGMI mesh that is one transfer per link
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            const size_t off = (size_t)(g->first_rank + i) * bytes;
            for (int k = 1; k < n; k++) {
                const int j = (i + k) % n;            // staggered destinations: no two members start on the same target
                HIP_TRY(hipMemcpyPeerAsync(static_cast<char*>(buffers[j]) + off, g->devices[(size_t)j],
                                           static_cast<const char*>(buffers[i]) + off, g->devices[(size_t)i], bytes, S((size_t)i)));
            }
            HIP_TRY(hipEventRecord(events[(size_t)i], S((size_t)i)));
        }
        // 3. a member's later work sees the whole frame: its stream waits for every push
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            for (int j = 0; j < n; j++)
                if (j != i) HIP_TRY(hipStreamWaitEvent(S((size_t)i), events[(size_t)j], 0));
        }
        return ILM_OK;
    }
    if (gather != ILM_GATHER_RCCL) return api_fail(ILM_ERR_INVALID_ARGUMENT, "unknown gather mode %d", gather);
    const int32_t rc = ensure_comms(g);
    if (rc != ILM_OK) return rc;
    Rccl& r = rccl();
    NCCL_TRY(r.GroupStart());
    for (int i = 0; i < g->n_local; i++) {
        HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
        char* buf = static_cast<char*>(buffers[i]);
        // in place: the send buffer is this rank's slot of the receive buffer
        NCCL_TRY(r.AllGather(buf + (size_t)(g->first_rank + i) * bytes, buf, bytes, ncclInt8, g->comms[(size_t)i], S((size_t)i)));
    }
    NCCL_TRY(r.GroupEnd());
    return ILM_OK;
}

// The exchange for strips of unequal size: rank r owns `bytes[r]` at `offset[r]` of every member's buffer and every other member needs
// them.  Peer mode: as all_gather, every local member pushes its range to the others (one transfer per xGMI link).  RCCL: one group of
// point-to-point transfers -- rank r sends its range to each of the world - 1 others and receives theirs in place; on the fully
// connected xGMI mesh of one node that is again one transfer per link and direction, with no ring hop in between.
int32_t exchange_ranges(Group* g, void* const* buffers, const std::vector<size_t>& offset, const std::vector<size_t>& bytes, int32_t gather, bool async = false) {
    if (gather == ILM_GATHER_NONE || g->world == 1) return ILM_OK;
    std::vector<hipEvent_t>& events = async ? g->xevents : g->events;
    auto S = [&](size_t i) { return g->exchange_stream(i, async); };
    if (gather == ILM_GATHER_PEER) {
        if (g->rank_mode) return api_fail(ILM_ERR_STATE, "ILM_GATHER_PEER needs every member in this process: use ILM_GATHER_RCCL");
        const int n = g->n_local;
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            HIP_TRY(hipEventRecord(events[(size_t)i], S((size_t)i)));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            for (int j = 0; j < n; j++)
                if (j != i) HIP_TRY(hipStreamWaitEvent(S((size_t)i), events[(size_t)j], 0));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            const size_t off = offset[(size_t)(g->first_rank + i)], len = bytes[(size_t)(g->first_rank + i)];
            for (int k = 1; k < n && len > 0; k++) {
                const int j = (i + k) % n;
                HIP_TRY(hipMemcpyPeerAsync(static_cast<char*>(buffers[j]) + off, g->devices[(size_t)j],
                                           static_cast<const char*>(buffers[i]) + off, g->devices[(size_t)i], len, S((size_t)i)));
            }
            HIP_TRY(hipEventRecord(events[(size_t)i], S((size_t)i)));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            for (int j = 0; j < n; j++)
                if (j != i) HIP_TRY(hipStreamWaitEvent(S((size_t)i), events[(size_t)j], 0));
        }
        return ILM_OK;
    }
    if (gather != ILM_GATHER_RCCL) return api_fail(ILM_ERR_INVALID_ARGUMENT, "unknown gather mode %d", gather);
    const int32_t rc = ensure_comms(g);
    if (rc != ILM_OK) return rc;
    Rccl& r = rccl();
    NCCL_TRY(r.GroupStart());
    for (int i = 0; i < g->n_local; i++) {
        HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
        char* buf = static_cast<char*>(buffers[i]);
        const int me = g->first_rank + i;
        for (int k = 1; k < g->world; k++) {
            const int to = (me + k) % g->world, from = (me - k + g->world) % g->world;      // staggered: no two ranks start on the same peer
            if (bytes[(size_t)me] > 0) NCCL_TRY(r.Send(buf + offset[(size_t)me], bytes[(size_t)me], ncclInt8, to, g->comms[(size_t)i], S((size_t)i)));
            if (bytes[(size_t)from] > 0) NCCL_TRY(r.Recv(buf + offset[(size_t)from], bytes[(size_t)from], ncclInt8, from, g->comms[(size_t)i], S((size_t)i)));
        }
    }
    NCCL_TRY(r.GroupEnd());
    return ILM_OK;
}

int32_t host_all_gather(Group* g, const void* local, void* out, size_t bytes);
int32_t ensure_counts_scratch(Group* g, size_t total);

// Every local member's stream waits for everything queued on every other member's stream so far (events; nothing blocks the host).
// A group that spans processes has no common events: there the fence is an 8-byte in-place all-gather on the context streams -- a rank's
// collective cannot complete before every rank has started its own, which its stream does only after the work queued in front of it.
int32_t fence_members(Group* g) {
    if (g->rank_mode) {
        if (g->world < 2) return ILM_OK;
        const int32_t rc = ensure_counts_scratch(g, 8u * (size_t)g->world);
        if (rc != ILM_OK) return rc;
        void* bufs[1] = { g->d_counts };
        return all_gather(g, bufs, 8, ILM_GATHER_RCCL);
    }
    const int n = g->n_local;
    if (n < 2) return ILM_OK;
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
        HIP_TRY(hipEventRecord(g->events[(size_t)i], g->stream((size_t)i)));
    }
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
        for (int j = 0; j < n; j++)
            if (j != i) HIP_TRY(hipStreamWaitEvent(g->stream((size_t)i), g->events[(size_t)j], 0));
    }
    return ILM_OK;
}

// Proof that every mapping made by hipIpcOpenMemHandle addresses the buffer it was exported for (a collective, once per group lightmap,
// after every rank has mapped every buffer): rank r stores (a one-lane kernel: the access the mirror stores will use) a 16-byte stamp
// { serial, r } into slot r of the first and of the last 16 * world bytes of EVERY other rank's buffer through its mapping, and every rank then finds all the stamps in its own buffer -- or
// nobody arms.  A mapping that resolves elsewhere (seen once this round, set_store_mode) would otherwise show as a frame with holes, or
// as a memory fault in the middle of a frame.  The bytes under the stamps are saved and put back.
int32_t prove_ipc_mappings(GroupLightmap* m, const std::vector<void*>& peers) {
    Group* g = m->group;
    const int world = g->world, me = g->first_rank;
    const size_t bytes = m->row_bytes * (size_t)m->slot_rows * (size_t)world, span = 16u * (size_t)world;
    if (bytes < 2 * span) return ILM_OK;                                   // (a frame that small has no room for the proof; it has no holes to hide either)
    char* own = static_cast<char*>(m->buffers[0]);
    const size_t at[2] = { 0, (bytes - span) & ~(size_t)15 };             // (16-byte stores: aligned)
    std::vector<unsigned char> saved(2 * span), seen(2 * span);
    uint64_t token = 0; std::vector<uint64_t> all((size_t)world, 0);
    // From here to the last collective a LOCAL failure (a copy, a memset, the stamping kernel) never returns: it folds into this rank's
    // verdict word and the rank keeps walking the same three all-gathers as the others -- a rank that left early would pair its next
    // collective with the wrong one of its peers and hang them.
    uint64_t ok = 1;
    auto soft = [&ok](hipError_t e) { if (e != hipSuccess) { (void)hipGetLastError(); ok = 0; } };
    soft(hipStreamSynchronize(g->stream(0)));
    for (int k = 0; k < 2; k++) soft(hipMemcpy(saved.data() + (size_t)k * span, own + at[k], span, hipMemcpyDeviceToHost));
    const bool have_saved = ok != 0;
    for (int k = 0; k < 2 && ok; k++) soft(hipMemset(own + at[k], 0, span));
    { const int32_t rc = host_all_gather(g, &token, all.data(), sizeof(uint64_t)); if (rc != ILM_OK) return rc; }      // everybody's slots are blank
    const uint32_t serial = ++g->ipc_proofs;
    const uint32_t stamp[4] = { 0x494c4d00u, serial, (uint32_t)me, ~serial ^ (uint32_t)me };
    for (void* p : peers)
        for (int k = 0; k < 2 && ok; k++)
            soft(launch_stamp16(static_cast<char*>(p) + at[k] + 16u * (size_t)me, stamp, g->stream(0)));
    soft(hipStreamSynchronize(g->stream(0)));
    { const int32_t rc = host_all_gather(g, &token, all.data(), sizeof(uint64_t)); if (rc != ILM_OK) return rc; }      // everybody has stamped
    for (int k = 0; k < 2 && ok; k++) soft(hipMemcpy(seen.data() + (size_t)k * span, own + at[k], span, hipMemcpyDeviceToHost));
    int missing = -1;
    for (int k = 0; k < 2 && ok; k++)
        for (int r = 0; r < world; r++) {
            if (r == me) continue;
            const uint32_t want[4] = { 0x494c4d00u, serial, (uint32_t)r, ~serial ^ (uint32_t)r };
            if (memcmp(seen.data() + (size_t)k * span + 16u * (size_t)r, want, 16) != 0 && missing < 0) missing = r;
        }
    if (missing >= 0) ok = 0;
    std::vector<uint64_t> verdicts((size_t)world, 0);
    verdicts[(size_t)me] = ok;
    { const int32_t rc = host_all_gather(g, &verdicts[(size_t)me], verdicts.data(), sizeof(uint64_t)); if (rc != ILM_OK) return rc; }
    if (have_saved)
        for (int k = 0; k < 2; k++)
            if (hipMemcpy(own + at[k], saved.data() + (size_t)k * span, span, hipMemcpyHostToDevice) != hipSuccess) (void)hipGetLastError();
    for (int r = 0; r < world; r++)
        if (!verdicts[(size_t)r])
            return api_fail(ILM_ERR_STATE, "ILM_GATHER_STORE: the IPC mappings do not address the ranks' lightmaps (rank %d did not find every other rank's stamp in its buffer): no rank arms the mode", r);
    return ILM_OK;
}

// ILM_GATHER_STORE (r05): the exchange disappears into the light kernel's epilogue.  Every member's lightmap object is handed the
// buffers of the OTHER members (lightmap_set_mirrors): a light pass stores each texel of its strip at the same offset of every copy of
// the frame -- its own and, through the peer mapping (hipDeviceEnablePeerAccess, make_members), the others' over xGMI: n stores of 8 B
// per pixel and member instead of a copy phase behind the strip.  What is left of the "gather" is ordering: a member's later readers of
// the frame wait for the other members' passes (fence_members).  A group that spans processes maps the other ranks' buffers through IPC
// handles instead (below) and fences with an 8-byte collective.
int32_t set_store_mode(GroupLightmap* m, bool enable) {
    Group* g = m->group;
    if (enable == m->store_mode) return ILM_OK;
    if (g->rank_mode) {
        // One process per GPU: the other ranks' buffers are mapped into this process through IPC handles (hipIpcGetMemHandle of the own
        // buffer, the 64-byte handles all-gathered, hipIpcOpenMemHandle of the others': the mapping RCCL's own peer transport uses).  A
        // COLLECTIVE, like set_strips: every rank arms or nobody does.
        const int world = g->world, me = g->first_rank;
        HIP_TRY(hipSetDevice(g->devices[0]));
        if (!enable) {
            int32_t rc = lightmap_set_mirrors(m->lightmaps[0], nullptr, 0);          // (drains this rank's stream: its stores into the others are done)
            m->store_mode = false;                                                   // (the mappings stay: see below)
            // nobody frees a buffer another rank may still be storing into: all ranks have drained when this returns
            uint64_t token = 0; std::vector<uint64_t> all((size_t)world, 0);
            const int32_t rb = host_all_gather(g, &token, all.data(), sizeof(uint64_t));
            return rc != ILM_OK ? rc : rb;
        }
        // The mappings are made ONCE per group lightmap (the buffers never move) and live until the GROUP is destroyed: re-arming reuses
        // them, and a destroyed lightmap hands them to Group::ipc_retired instead of unmapping.  Measured reason (r05, ROCm 7.2, four rank
        // processes): after hipIpcCloseMemHandle, the next large hipMalloc of the process lands in the address range the mapping had, and
        // a handle exported for THAT allocation resolved to another buffer in the importing ranks (their stores went elsewhere, then
        // faulted) -- deterministic with bench.py's second lit frame, gone with no unmap in between.
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
        const bool mapped = world > 1 && m->ipc_peers.size() == (size_t)(world - 1);
        std::vector<hipIpcMemHandle_t> handles((size_t)world);
        hipIpcMemHandle_t mine;
        uint64_t ok = 1;
        std::vector<void*> peers = m->ipc_peers;
        if (world > 1 && !mapped) {
            peers.clear();
            if (hipIpcGetMemHandle(&mine, m->buffers[0]) != hipSuccess) { (void)hipGetLastError(); ok = 0; memset(&mine, 0, sizeof(mine)); }
            handles[(size_t)me] = mine;
            const int32_t rc = host_all_gather(g, &handles[(size_t)me], handles.data(), sizeof(hipIpcMemHandle_t));
            if (rc != ILM_OK) return rc;
            for (int r = 0; r < world && ok; r++) {
                if (r == me) continue;
                void* p = nullptr;
                if (hipIpcOpenMemHandle(&p, handles[(size_t)r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = 0; break; }
                peers.push_back(p);
            }
        }
        // every rank learns whether every rank could map every buffer
        std::vector<uint64_t> verdicts((size_t)world, 0);
        verdicts[(size_t)me] = ok;
        if (world > 1) {
            const int32_t rc = host_all_gather(g, &verdicts[(size_t)me], verdicts.data(), sizeof(uint64_t));
            if (rc != ILM_OK) { if (!mapped) for (void* p : peers) g->ipc_retired.push_back(p); return rc; }
        }
        int failed = -1;
        for (int r = 0; r < world; r++) if (!verdicts[(size_t)r] && failed < 0) failed = r;
        if (failed >= 0) {
            if (!mapped) for (void* p : peers) g->ipc_retired.push_back(p);
            return api_fail(ILM_ERR_STATE, "ILM_GATHER_STORE: rank %d could not map the other ranks' lightmaps through IPC handles: no rank arms the mode", failed);
        }
        if (!mapped && world > 1) {
            // NEGATIVE CONTROL of the proof below (tests/test_two_ranks_one_gpu.py): rank 0's first mapping addresses memory of its own
            if (ok && me == 0 && !peers.empty() && getenv("ILM_EXP_IPC_BOGUS_MAPPING")) {
                void* elsewhere = nullptr;
                if (hipMalloc(&elsewhere, m->row_bytes * (size_t)m->slot_rows * (size_t)world) == hipSuccess) {
                    g->ipc_retired.push_back(peers[0]); g->ipc_decoys.push_back(elsewhere); peers[0] = elsewhere;
                }
            }
            const int32_t rc = prove_ipc_mappings(m, peers);
            if (rc != ILM_OK) {
                for (void* p : peers)
                    if (std::find(g->ipc_decoys.begin(), g->ipc_decoys.end(), p) == g->ipc_decoys.end()) g->ipc_retired.push_back(p);
                return rc;
            }
        }
        const int32_t rc = lightmap_set_mirrors(m->lightmaps[0], peers.data(), (int)peers.size());
        if (rc != ILM_OK) { if (!mapped) for (void* p : peers) g->ipc_retired.push_back(p); return rc; }
        m->ipc_peers = peers;
        m->store_mode = true;
        return ILM_OK;
    }
    if (enable && !g->peers_ok) return api_fail(ILM_ERR_STATE, "ILM_GATHER_STORE needs peer access between every pair of the group's devices");
    for (int i = 0; i < g->n_local; i++) {
        std::vector<void*> others;
        for (int j = 0; j < g->n_local && enable; j++)
            if (j != i) others.push_back(m->buffers[(size_t)j]);
        const int32_t rc = lightmap_set_mirrors(m->lightmaps[(size_t)i], others.data(), (int)others.size());
        if (rc != ILM_OK) {
            // (never half-armed: a member that stores into the others while they do not would leave frames that differ by member)
            for (int j = 0; j <= i && enable; j++) (void)lightmap_set_mirrors(m->lightmaps[(size_t)j], nullptr, 0);
            return rc;
        }
    }
    m->store_mode = enable;
    return ILM_OK;
}

// the exchange of a group lightmap's strips: one in-place all-gather for the equal slots, range by range otherwise
// the second stream (and its events) of every local member, for ILM_GATHER_ASYNC
int32_t ensure_exchange_streams(Group* g) {
    if (g->xstream.size() == (size_t)g->n_local) return ILM_OK;
    // built aside and installed whole: after a partial failure the group has none, not a table shorter than n_local
    std::vector<hipStream_t> streams; std::vector<hipEvent_t> forks, events;
    hipError_t e = hipSuccess;
    for (int i = 0; i < g->n_local && e == hipSuccess; i++) {
        hipStream_t st = nullptr; hipEvent_t a = nullptr, b = nullptr;
        e = hipSetDevice(g->devices[(size_t)i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
        streams.push_back(st); forks.push_back(a); events.push_back(b);
    }
    if (e != hipSuccess) {
        for (size_t i = 0; i < streams.size(); i++) {
            (void)hipSetDevice(g->devices[i]);
            if (streams[i]) (void)hipStreamDestroy(streams[i]);
            if (forks[i]) (void)hipEventDestroy(forks[i]);
            if (events[i]) (void)hipEventDestroy(events[i]);
        }
        (void)hipGetLastError();
        return api_fail((int32_t)e, "exchange streams of the group: %s", hipGetErrorString(e));
    }
    g->xstream.swap(streams); g->xfork.swap(forks); g->xevents.swap(events);
    return ILM_OK;
}

int32_t gather_lightmap(GroupLightmap* m, int32_t gather) {
    Group* g = m->group;
    const bool async = (gather & ILM_GATHER_ASYNC) != 0;
    gather &= ~ILM_GATHER_ASYNC;
    if (gather == ILM_GATHER_STORE) {
        if (async) return api_fail(ILM_ERR_INVALID_ARGUMENT, "ILM_GATHER_STORE has no exchange to move to another stream");
        if (!m->store_mode) return api_fail(ILM_ERR_STATE, "ILM_GATHER_STORE: arm the group lightmap first (ilm_group_lightmap_store_mode)");
        return fence_members(g);
    }
    if (m->store_mode && gather != ILM_GATHER_NONE)
        return api_fail(ILM_ERR_STATE, "the group lightmap is in store mode: its members already hold every strip (gather with ILM_GATHER_STORE, or switch the mode off)");
    if (async) {
        // ILM_GATHER_ASYNC (r05): the exchange of THIS frame runs on the members' exchange streams, behind what their context streams have
        // queued so far (the strip), and the context streams go on -- with the next frame's strip into ANOTHER group lightmap.  Nobody may
        // touch this lightmap again before ilm_group_lightmap_wait.
        if (gather != ILM_GATHER_PEER && gather != ILM_GATHER_RCCL) return api_fail(ILM_ERR_INVALID_ARGUMENT, "ILM_GATHER_ASYNC goes with ILM_GATHER_PEER or ILM_GATHER_RCCL");
        const int32_t rc = ensure_exchange_streams(g);
        if (rc != ILM_OK) return rc;
        if (m->xdone.size() != (size_t)g->n_local) {
            // (built aside and installed whole, like the exchange streams: never a table shorter than n_local)
            std::vector<hipEvent_t> done;
            hipError_t err = hipSuccess;
            for (int i = 0; i < g->n_local && err == hipSuccess; i++) {
                hipEvent_t e = nullptr;
                err = hipSetDevice(g->devices[(size_t)i]);
                if (err == hipSuccess) err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
                done.push_back(e);
            }
            if (err != hipSuccess) {
                for (size_t i = 0; i < done.size(); i++)
                    if (done[i]) { (void)hipSetDevice(g->devices[i]); (void)hipEventDestroy(done[i]); }
                (void)hipGetLastError();
                return api_fail((int32_t)err, "exchange events of the group lightmap: %s", hipGetErrorString(err));
            }
            m->xdone.swap(done);
            m->xpending.assign((size_t)g->n_local, false);
        }
        for (int i = 0; i < g->n_local; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            HIP_TRY(hipEventRecord(g->xfork[(size_t)i], g->stream((size_t)i)));
            HIP_TRY(hipStreamWaitEvent(g->xstream[(size_t)i], g->xfork[(size_t)i], 0));
        }
    }
    int32_t rc;
    if (m->equal_slots) {
        rc = all_gather(g, m->buffers.data(), m->row_bytes * (size_t)m->slot_rows, gather, async);
    } else {
        std::vector<size_t> offset((size_t)g->world), bytes((size_t)g->world);
        for (int r = 0; r < g->world; r++) {
            offset[(size_t)r] = m->row_bytes * (size_t)m->begin[(size_t)r];
            bytes[(size_t)r] = m->row_bytes * (size_t)(m->end[(size_t)r] - m->begin[(size_t)r]);
        }
        rc = exchange_ranges(g, m->buffers.data(), offset, bytes, gather, async);
    }
    if (rc != ILM_OK) return rc;
    if (async)
        for (int i = 0; i < g->n_local; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            HIP_TRY(hipEventRecord(m->xdone[(size_t)i], g->xstream[(size_t)i]));
            m->xpending[(size_t)i] = true;
        }
    return ILM_OK;
}

// every member's context stream waits for the exchange queued last on this lightmap (nothing blocks the host)
int32_t wait_lightmap_exchange(GroupLightmap* m) {
    Group* g = m->group;
    for (size_t i = 0; i < m->xpending.size(); i++)
        if (m->xpending[i]) {
            HIP_TRY(hipSetDevice(g->devices[i]));
            HIP_TRY(hipStreamWaitEvent(g->stream(i), m->xdone[i], 0));
            m->xpending[i] = false;
        }
    return ILM_OK;
}

// device scratch of the small collectives (member 0's device)
int32_t ensure_counts_scratch(Group* g, size_t total) {
    HIP_TRY(hipSetDevice(g->devices[0]));
    if (total > g->counts_bytes) {
        if (g->d_counts) { HIP_TRY(hipStreamSynchronize(g->stream((size_t)0))); HIP_TRY(hipFree(g->d_counts)); g->d_counts = nullptr; g->counts_bytes = 0; }
        const size_t cap = total < 4096 ? 4096 : total;
        HIP_TRY(hipMalloc(&g->d_counts, cap));
        g->counts_bytes = cap;
    }
    return ILM_OK;
}

// Small host payloads (liveness counters, timings): rank r's `bytes` land at out + r * bytes on every process.  Members of this
// process are copied; a group that spans processes stages its slot in device memory and all-gathers it with RCCL.  Synchronises.
int32_t host_all_gather(Group* g, const void* local, void* out, size_t bytes) {
    if (bytes == 0) return ILM_OK;
    char* o = static_cast<char*>(out);
    const size_t mine = (size_t)g->first_rank * bytes;
    if (o + mine != local) memmove(o + mine, local, bytes * (size_t)g->n_local);
    if (!(g->rank_mode && g->world > 1)) return ILM_OK;
    const size_t total = bytes * (size_t)g->world;
    { const int32_t rc = ensure_counts_scratch(g, total); if (rc != ILM_OK) return rc; }
    char* d = static_cast<char*>(g->d_counts);
    HIP_TRY(hipMemcpyAsync(d + mine, o + mine, bytes, hipMemcpyHostToDevice, g->stream((size_t)0)));
    void* bufs[1] = { g->d_counts };
    const int32_t rc = all_gather(g, bufs, bytes, ILM_GATHER_RCCL);
    if (rc != ILM_OK) return rc;
    HIP_TRY(hipMemcpyAsync(o, d, total, hipMemcpyDeviceToHost, g->stream((size_t)0)));
    HIP_TRY(hipStreamSynchronize(g->stream((size_t)0)));
    return ILM_OK;
}


// ilm_group_gather_chunks: the sharded particle state made whole.  Chunk c of the table lives on rank c % world as chunk c / world of
// that member's source system; afterwards chunk c of every member's `gathered` system holds `count` consecutive component planes of it.
// The planes of a chunk are contiguous (count * stride floats), so a chunk travels as ONE range straight from the owner's planes into
// the destination's planes -- no pack, no unpack.  Peer mode: the owner pushes to the n - 1 others (one transfer per xGMI link and
// chunk).  RCCL: one group of ncclSend / ncclRecv, rank r sends each of its chunks to every other rank and receives theirs in place,
// peers staggered as in exchange_ranges; several transfers between one pair match in issue order (chunk order on both sides).
struct ChunkRange { char* base; size_t bytes; };
int32_t gather_chunks(Group* g, const IlmHandle* sources, const IlmHandle* gathered, int total_chunks, int first, int count, int32_t gather) {
    const int world = g->world, n = g->n_local;
    if (total_chunks == 0) return ILM_OK;
    // every local member's view of its own chunks and of the whole table, validated before anything is queued
    std::vector<std::vector<ChunkRange>> src((size_t)n), dst((size_t)n);
    int64_t stride0 = 0; int32_t size0 = 0;
    for (int i = 0; i < n; i++) {
        const int rank = g->first_rank + i;
        const int owned = (total_chunks - rank + world - 1) / world;
        int32_t have = 0;
        int32_t rc = ilm_system_chunk_count(sources[i], &have);
        if (rc != ILM_OK) return rc;
        if (have != owned)
            return api_fail(ILM_ERR_STATE, "rank %d holds %d chunks but owns %d of a table of %d (chunk c lives on rank c %% %d)", rank, have, owned, total_chunks, world);
        rc = ilm_system_chunk_count(gathered[i], &have);
        if (rc != ILM_OK) return rc;
        if (have != total_chunks)
            return api_fail(ILM_ERR_STATE, "rank %d's gathered system has %d chunks, the table has %d", rank, have, total_chunks);
        for (int pass = 0; pass < 2; pass++) {
            const IlmHandle sys = pass ? gathered[i] : sources[i];
            const int chunks = pass ? total_chunks : owned;
            for (int k = 0; k < chunks; k++) {
                float* base = nullptr; int64_t stride = 0; int32_t size = 0; IlmHandle ctx = 0;
                rc = system_chunk_view(sys, k, pass != 0, &base, &stride, &size, &ctx);
                if (rc != ILM_OK) return rc;
                if (ctx != g->ctx[(size_t)i])
                    return api_fail(ILM_ERR_INVALID_ARGUMENT, "the %s system of member %d lives on another context than the member's", pass ? "gathered" : "source", i);
                if (stride0 == 0) { stride0 = stride; size0 = size; }
                if (stride != stride0 || size != size0)
                    return api_fail(ILM_ERR_INVALID_ARGUMENT, "the systems' engines differ in chunk size (%d / %d): one chunk geometry per table", size, size0);
                ChunkRange r = { reinterpret_cast<char*>(base + (int64_t)first * stride), sizeof(float) * (size_t)count * (size_t)stride };
                (pass ? dst : src)[(size_t)i].push_back(r);
            }
        }
    }
    auto S = [&](size_t i) { return g->stream(i); };
    // a member's own chunks: device-local copies into its gathered system
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
        const int rank = g->first_rank + i;
        for (size_t k = 0; k < src[(size_t)i].size(); k++) {
            const ChunkRange& from = src[(size_t)i][k]; const ChunkRange& to = dst[(size_t)i][k * (size_t)world + (size_t)rank];
            if (to.base != from.base) HIP_TRY(hipMemcpyAsync(to.base, from.base, from.bytes, hipMemcpyDeviceToDevice, S((size_t)i)));
        }
    }
    if (gather == ILM_GATHER_NONE || world == 1) return ILM_OK;
    if (gather == ILM_GATHER_PEER) {
        if (g->rank_mode) return api_fail(ILM_ERR_STATE, "ILM_GATHER_PEER needs every member in this process: use ILM_GATHER_RCCL");
        // as all_gather: nobody pushes into a member before the work that member queued earlier (readers of the old table) is done
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            HIP_TRY(hipEventRecord(g->events[(size_t)i], S((size_t)i)));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            for (int j = 0; j < n; j++)
                if (j != i) HIP_TRY(hipStreamWaitEvent(S((size_t)i), g->events[(size_t)j], 0));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            for (size_t k = 0; k < src[(size_t)i].size(); k++)
                for (int step = 1; step < n; step++) {
                    const int j = (i + step) % n;
                    const ChunkRange& from = src[(size_t)i][k]; const ChunkRange& to = dst[(size_t)j][k * (size_t)world + (size_t)(g->first_rank + i)];
                    HIP_TRY(hipMemcpyPeerAsync(to.base, g->devices[(size_t)j], from.base, g->devices[(size_t)i], from.bytes, S((size_t)i)));
                }
            HIP_TRY(hipEventRecord(g->events[(size_t)i], S((size_t)i)));
        }
        for (int i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
            for (int j = 0; j < n; j++)
                if (j != i) HIP_TRY(hipStreamWaitEvent(S((size_t)i), g->events[(size_t)j], 0));
        }
        return ILM_OK;
    }
    if (gather != ILM_GATHER_RCCL) return api_fail(ILM_ERR_INVALID_ARGUMENT, "ilm_group_gather_chunks exchanges with ILM_GATHER_PEER or ILM_GATHER_RCCL (mode %d)", gather);
    const int32_t rc = ensure_comms(g);
    if (rc != ILM_OK) return rc;
    Rccl& r = rccl();
    NCCL_TRY(r.GroupStart());
    for (int i = 0; i < n; i++) {
        HIP_TRY(hipSetDevice(g->devices[(size_t)i]));
        const int me = g->first_rank + i;
        for (int step = 1; step < world; step++) {
            const int to = (me + step) % world, from = (me - step + world) % world;
            for (const ChunkRange& c : src[(size_t)i]) NCCL_TRY(r.Send(c.base, c.bytes, ncclInt8, to, g->comms[(size_t)i], S((size_t)i)));
            const int theirs = (total_chunks - from + world - 1) / world;
            for (int k = 0; k < theirs; k++) {
                const ChunkRange& c = dst[(size_t)i][(size_t)k * (size_t)world + (size_t)from];
                NCCL_TRY(r.Recv(c.base, c.bytes, ncclInt8, from, g->comms[(size_t)i], S((size_t)i)));
            }
        }
    }
    NCCL_TRY(r.GroupEnd());
    return ILM_OK;
}

}  // namespace

extern "C" {

int32_t ilm_group_create(const int32_t* device_ids, int32_t n, IlmHandle* out_group) {
    if (!out_group) return api_fail(ILM_ERR_INVALID_ARGUMENT, "out_group is NULL");
    *out_group = 0;
    if (!device_ids || n < 1 || n > kMaxMembers) return api_fail(ILM_ERR_INVALID_ARGUMENT, "a group has 1 .. %d members", kMaxMembers);
    const int visible = ilm_device_count();
    if (visible <= 0) return api_fail(ILM_ERR_NO_DEVICE, "no HIP device visible: libilluminant_hip has no CPU fallback");
    for (int i = 0; i < n; i++)
        if (device_ids[i] < 0 || device_ids[i] >= visible)
            return api_fail(ILM_ERR_OUT_OF_RANGE, "device %d outside [0, %d)", device_ids[i], visible);
    Group* g = new (std::nothrow) Group();
    if (!g) return api_fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    g->n_local = n; g->world = n; g->first_rank = 0;
    g->devices.assign(device_ids, device_ids + n);
    handle_register(g, kMagicGroup);
    const int32_t rc = make_members(g);
    if (rc != ILM_OK) { release_group(g); return rc; }
    *out_group = static_cast<IlmHandle>(reinterpret_cast<uintptr_t>(g));
    return ILM_OK;
}

int32_t ilm_group_unique_id(void* out_id128) {
    if (!out_id128) return api_fail(ILM_ERR_INVALID_ARGUMENT, "out_id128 is NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    Rccl& r = rccl();
    if (!r.ok) return api_fail(ILM_ERR_STATE, "RCCL is not available: %s", r.why);
    ncclUniqueId id;
    NCCL_TRY(r.GetUniqueId(&id));
    memcpy(out_id128, &id, sizeof(id));
    return ILM_OK;
}

int32_t ilm_group_create_rank(int32_t device_id, int32_t rank, int32_t world, const void* id128, IlmHandle* out_group) {
    if (!out_group) return api_fail(ILM_ERR_INVALID_ARGUMENT, "out_group is NULL");
    *out_group = 0;
    if (!id128) return api_fail(ILM_ERR_INVALID_ARGUMENT, "id128 is NULL");
    if (world < 1 || rank < 0 || rank >= world) return api_fail(ILM_ERR_OUT_OF_RANGE, "rank %d outside [0, %d)", rank, world);
    const int visible = ilm_device_count();
    if (visible <= 0) return api_fail(ILM_ERR_NO_DEVICE, "no HIP device visible: libilluminant_hip has no CPU fallback");
    if (device_id < 0 || device_id >= visible) return api_fail(ILM_ERR_OUT_OF_RANGE, "device %d outside [0, %d)", device_id, visible);
    Rccl& r = rccl();
    if (!r.ok) return api_fail(ILM_ERR_STATE, "RCCL is not available: %s", r.why);
    Group* g = new (std::nothrow) Group();
    if (!g) return api_fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    g->n_local = 1; g->world = world; g->first_rank = rank; g->rank_mode = true;
    g->devices.assign(1, device_id);
    handle_register(g, kMagicGroup);
    int32_t rc = make_members(g);
    if (rc != ILM_OK) { release_group(g); return rc; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    if (hipSetDevice(device_id) != hipSuccess) { release_group(g); return api_fail(ILM_ERR_STATE, "hipSetDevice(%d) failed", device_id); }
    const ncclResult_t nr = r.CommInitRank(&comm, world, id, rank);      // collective over all `world` processes
    if (nr != ncclSuccess) {
        rc = api_fail(ILM_ERR_STATE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, r.GetErrorString(nr));
        release_group(g);
        return rc;
    }
    g->comms.assign(1, comm);
    *out_group = static_cast<IlmHandle>(reinterpret_cast<uintptr_t>(g));
    return ILM_OK;
}

int32_t ilm_group_destroy(IlmHandle h) {
    Group* g = group_from(h);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    if (g->children > 0) return api_fail(ILM_ERR_STATE, "%d group lightmap(s) are still alive", g->children);
    // member contexts refuse to go while objects of theirs live: find that out before anything is torn down
    for (int i = 0; i < g->n_local; i++) {
        const int live = ctx_child_count(g->ctx[(size_t)i]);
        if (live > 0)
            return api_fail(ILM_ERR_STATE, "%d object(s) of member %d's context are still alive: destroy engines, fields, G-buffers and lightmaps first", live, i);
    }
    release_group(g);
    return ILM_OK;
}

int32_t ilm_group_info(IlmHandle h, int32_t* out_local, int32_t* out_world, int32_t* out_first_rank, int32_t* out_comm_ranks) {
    Group* g = group_from(h);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    if (out_local) *out_local = g->n_local;
    if (out_world) *out_world = g->world;
    if (out_first_rank) *out_first_rank = g->first_rank;
    if (out_comm_ranks) {
        *out_comm_ranks = 0;
        if (!g->comms.empty()) {
            int count = 0;
            NCCL_TRY(rccl().CommCount(g->comms[0], &count));
            *out_comm_ranks = count;
        }
    }
    return ILM_OK;
}

int32_t ilm_group_ctx(IlmHandle h, int32_t local_index, IlmHandle* out_ctx) {
    Group* g = group_from(h);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    if (!out_ctx) return api_fail(ILM_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
    if (local_index < 0 || local_index >= g->n_local) return api_fail(ILM_ERR_OUT_OF_RANGE, "member %d outside [0, %d)", local_index, g->n_local);
    *out_ctx = g->ctx[(size_t)local_index];
    return ILM_OK;
}

int32_t ilm_group_sync(IlmHandle h) {
    Group* g = group_from(h);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    for (int i = 0; i < g->n_local; i++) {
        const int32_t rc = ilm_ctx_sync(g->ctx[(size_t)i]);
        if (rc != ILM_OK) return rc;
        if ((size_t)i < g->xstream.size()) { HIP_TRY(hipSetDevice(g->devices[(size_t)i])); HIP_TRY(hipStreamSynchronize(g->xstream[(size_t)i])); }
    }
    return ILM_OK;
}

int32_t ilm_group_all_gather(IlmHandle h, void* const* buffers, uint64_t bytes_per_rank, int32_t gather) {
    ILM_TRACE_RANGE("ilm_group_all_gather");
    Group* g = group_from(h);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    if (!buffers) return api_fail(ILM_ERR_INVALID_ARGUMENT, "buffers is NULL");
    for (int i = 0; i < g->n_local; i++)
        if (!buffers[i]) return api_fail(ILM_ERR_INVALID_ARGUMENT, "buffers[%d] is NULL", i);
    return all_gather(g, buffers, (size_t)bytes_per_rank, gather);
}

int32_t ilm_group_host_all_gather(IlmHandle h, const void* local, void* out_all, uint32_t bytes_per_rank) {
    ILM_TRACE_RANGE("ilm_group_host_all_gather");
    Group* g = group_from(h);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    if (!local || !out_all) return api_fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (bytes_per_rank > (1u << 20)) return api_fail(ILM_ERR_OUT_OF_RANGE, "host payloads are small (<= 1 MiB per rank): use ilm_group_all_gather for device buffers");
    // every member's queued work is finished first: the call doubles as the barrier between timed regions (the exchange streams of
    // ILM_GATHER_ASYNC included: one communicator never has two operations in flight on two streams)
    for (int i = 0; i < g->n_local; i++) {
        const int32_t rc = ilm_ctx_sync(g->ctx[(size_t)i]);
        if (rc != ILM_OK) return rc;
        if ((size_t)i < g->xstream.size()) { HIP_TRY(hipSetDevice(g->devices[(size_t)i])); HIP_TRY(hipStreamSynchronize(g->xstream[(size_t)i])); }
    }
    return host_all_gather(g, local, out_all, (size_t)bytes_per_rank);
}

int32_t ilm_group_lightmap_create(IlmHandle h, int32_t width, int32_t height, int32_t format, IlmHandle* out) {
    Group* g = group_from(h);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    if (!out) return api_fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = 0;
    if (width <= 0 || height <= 0) return api_fail(ILM_ERR_OUT_OF_RANGE, "bad lightmap size %dx%d", width, height);
    if (format < ILM_LIGHTMAP_FLOAT4 || format > ILM_LIGHTMAP_RGBA8) return api_fail(ILM_ERR_INVALID_ARGUMENT, "unknown lightmap format %d", format);
    GroupLightmap* m = new (std::nothrow) GroupLightmap();
    if (!m) return api_fail(ILM_ERR_INVALID_ARGUMENT, "out of host memory");
    m->group = g; m->width = width; m->height = height; m->format = format;
    m->slot_rows = slot_rows_for(height, g->world);
    m->row_bytes = texel_bytes(format) * (size_t)width;
    for (int r = 0; r < g->world; r++) {
        m->begin.push_back(std::min(r * m->slot_rows, height));
        m->end.push_back(std::min((r + 1) * m->slot_rows, height));
    }
    handle_register(m, kMagicGroupLightmap);
    g->children++;
    const IlmHandle self = static_cast<IlmHandle>(reinterpret_cast<uintptr_t>(m));
    const size_t bytes = m->row_bytes * (size_t)m->slot_rows * (size_t)g->world;
    for (int i = 0; i < g->n_local; i++) {
        void* p = nullptr;
        hipError_t e = hipSetDevice(g->devices[(size_t)i]);
        if (e == hipSuccess) e = hipMalloc(&p, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(p, 0, bytes, g->stream((size_t)i));
        if (e != hipSuccess) {
            if (p) (void)hipFree(p);
            const int32_t rc = api_fail((int32_t)e, "group lightmap of %zu bytes on device %d: %s", bytes, g->devices[(size_t)i], hipGetErrorString(e));
            (void)ilm_group_lightmap_destroy(self);
            return rc;
        }
        m->buffers.push_back(p);
        IlmHandle lm = 0;
        const int32_t rc = ilm_lightmap_create(g->ctx[(size_t)i], width, m->slot_rows * g->world, format, p, &lm);
        if (rc != ILM_OK) { (void)ilm_group_lightmap_destroy(self); return rc; }
        m->lightmaps.push_back(lm);
    }
    *out = self;
    return ILM_OK;
}

int32_t ilm_group_lightmap_destroy(IlmHandle h) {
    GroupLightmap* m = glm_from(h);
    if (!m) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group lightmap handle");
    Group* g = m->group;
    if (m->store_mode && m->lightmaps.size() == (size_t)g->n_local) (void)set_store_mode(m, false);
    for (void* p : m->ipc_peers) g->ipc_retired.push_back(p);
    m->ipc_peers.clear();
    for (size_t i = 0; i < g->xstream.size(); i++) { (void)hipSetDevice(g->devices[i]); (void)hipStreamSynchronize(g->xstream[i]); }
    for (hipEvent_t e : m->xdone) if (e) (void)hipEventDestroy(e);
    for (IlmHandle lm : m->lightmaps) (void)ilm_lightmap_destroy(lm);        // synchronises the member's stream
    for (size_t i = 0; i < m->buffers.size(); i++) {
        (void)hipSetDevice(g->devices[i]);
        (void)hipStreamSynchronize(g->stream((size_t)i));
        (void)hipFree(m->buffers[i]);
    }
    g->children--;
    handle_retire(m);
    delete m;
    return ILM_OK;
}

int32_t ilm_group_lightmap_member(IlmHandle h, int32_t local_index, IlmHandle* out_lightmap) {
    GroupLightmap* m = glm_from(h);
    if (!m) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group lightmap handle");
    if (!out_lightmap) return api_fail(ILM_ERR_INVALID_ARGUMENT, "out_lightmap is NULL");
    if (local_index < 0 || local_index >= m->group->n_local)
        return api_fail(ILM_ERR_OUT_OF_RANGE, "member %d outside [0, %d)", local_index, m->group->n_local);
    *out_lightmap = m->lightmaps[(size_t)local_index];
    return ILM_OK;
}

int32_t ilm_group_lightmap_strip(IlmHandle h, int32_t rank, int32_t* out_row_begin, int32_t* out_row_end, int32_t* out_slot_rows) {
    GroupLightmap* m = glm_from(h);
    if (!m) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group lightmap handle");
    if (rank < 0 || rank >= m->group->world) return api_fail(ILM_ERR_OUT_OF_RANGE, "rank %d outside [0, %d)", rank, m->group->world);
    if (out_row_begin) *out_row_begin = m->begin[(size_t)rank];
    if (out_row_end) *out_row_end = m->end[(size_t)rank];
    if (out_slot_rows) *out_slot_rows = m->slot_rows;
    return ILM_OK;
}

int32_t ilm_group_lightmap_set_strips(IlmHandle h, const int32_t* row_begins, const int32_t* row_ends) {
    ILM_TRACE_RANGE("ilm_group_lightmap_set_strips");
    GroupLightmap* m = glm_from(h);
    if (!m) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group lightmap handle");
    Group* g = m->group;
    const int world = g->world;
    // 1. validate into a LOCAL verdict (nothing is installed yet, and nobody returns before the ranks have compared notes)
    enum : uint64_t { kTableEqualSlots = 0x45515541u, kTableInvalid = 0x42414421u };        // sentinels of the cross-rank check below
    const bool reset = !row_begins || !row_ends;           // NULL / NULL: back to the equal slots -- a collective like any other table
    char why[192] = "";
    bool valid = true;
    if (!reset) {
        // contiguous, in rank order, whole tile bands (the last one may be ragged), covering the frame exactly
        int at = 0;
        for (int r = 0; r < world && valid; r++) {
            if (row_begins[r] != at || row_ends[r] < row_begins[r] || row_ends[r] > m->height) {
                snprintf(why, sizeof(why), "strip %d = [%d, %d) does not continue at row %d of a %d-row frame", r, row_begins[r], row_ends[r], at, m->height);
                valid = false;
            } else if ((row_ends[r] % kTileRows) != 0 && row_ends[r] != m->height) {
                snprintf(why, sizeof(why), "strip %d ends at row %d: strips are whole %d-row tile bands", r, row_ends[r], kTileRows);
                valid = false;
            }
            at = row_ends[r];
        }
        if (valid && at != m->height) {
            snprintf(why, sizeof(why), "the strips end at row %d of a %d-row frame", at, m->height);
            valid = false;
        }
    }
    // 2. One process per GPU: every rank sizes its ncclSend / ncclRecv from its own copy of the table, so two ranks with different tables
    //    would exchange mismatched byte counts (a hang, or rows in the wrong place).  The call is ALWAYS a collective there:
    //    every rank -- also one whose table is malformed, and one that resets to the equal slots -- enters the same 8-byte host all-gather
    //    with a hash of what it was handed (FNV-1a over (begin, end) in rank order; a sentinel for "invalid" and one for "equal slots"),
    //    and all ranks succeed or fail together.  (A bad handle cannot take part: it names no group.)
    uint64_t hash = 1469598103934665603ull;
    if (!valid) hash = kTableInvalid;
    else if (reset) hash = kTableEqualSlots;
    else
        for (int r = 0; r < world; r++)
            for (const int v : { row_begins[r], row_ends[r] })
                for (int b = 0; b < 4; b++) { hash ^= (uint64_t)((v >> (8 * b)) & 0xFF); hash *= 1099511628211ull; }
    int disagree = -1, invalid_rank = valid ? -1 : g->first_rank;
    if (g->rank_mode && world > 1) {
        std::vector<uint64_t> all((size_t)world, 0);
        all[(size_t)g->first_rank] = hash;
        const int32_t rc = host_all_gather(g, &all[(size_t)g->first_rank], all.data(), sizeof(uint64_t));
        if (rc != ILM_OK) return rc;
        for (int r = 0; r < world; r++) {
            if (all[(size_t)r] == kTableInvalid && invalid_rank < 0) invalid_rank = r;
            if (all[(size_t)r] != hash && disagree < 0) disagree = r;
        }
    }
    // 3. all ranks reach the same verdict from the same gathered words; a failed installation leaves every rank's table as it was
    if (!valid) return api_fail(ILM_ERR_INVALID_ARGUMENT, "%s", why);
    if (invalid_rank >= 0) return api_fail(ILM_ERR_STATE, "rank %d was handed a malformed strip table: no rank installs one", invalid_rank);
    if (disagree >= 0)
        return api_fail(ILM_ERR_STATE, "rank %d installed a different strip table than rank %d: every rank must pass the same strips", disagree, g->first_rank);
    bool equal = true;
    for (int r = 0; r < world; r++) {
        const int eb = std::min(r * m->slot_rows, m->height), ee = std::min((r + 1) * m->slot_rows, m->height);
        m->begin[(size_t)r] = reset ? eb : row_begins[r];
        m->end[(size_t)r] = reset ? ee : row_ends[r];
        equal = equal && m->begin[(size_t)r] == eb && m->end[(size_t)r] == ee;
    }
    m->equal_slots = equal;
    return ILM_OK;
}

int32_t ilm_group_lightmap_gather(IlmHandle h, int32_t gather) {
    ILM_TRACE_RANGE("ilm_group_lightmap_gather");
    GroupLightmap* m = glm_from(h);
    if (!m) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group lightmap handle");
    return gather_lightmap(m, gather);
}

int32_t ilm_group_lightmap_wait(IlmHandle h) {
    ILM_TRACE_RANGE("ilm_group_lightmap_wait");
    GroupLightmap* m = glm_from(h);
    if (!m) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group lightmap handle");
    return wait_lightmap_exchange(m);
}

int32_t ilm_group_lightmap_store_mode(IlmHandle h, int32_t enable) {
    GroupLightmap* m = glm_from(h);
    if (!m) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group lightmap handle");
    return set_store_mode(m, enable != 0);
}

int32_t ilm_group_render_sphere_lights(IlmHandle hgroup, const IlmLightVertex* lights, int32_t light_count, const IlmEnvironment* env,
                                       const IlmDistanceFieldUniforms* df, const IlmHandle* gbuffers, const IlmHandle* sdfs,
                                       const float ambient[4], IlmHandle hlightmap, int32_t gather, IlmRenderStats* stats) {
    ILM_TRACE_RANGE("ilm_group_render_sphere_lights");
    Group* g = group_from(hgroup);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    GroupLightmap* m = glm_from(hlightmap);
    if (!m) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group lightmap handle");
    if (m->group != g) return api_fail(ILM_ERR_INVALID_ARGUMENT, "the lightmap belongs to another group");
    if ((gather & ~ILM_GATHER_ASYNC) < ILM_GATHER_NONE || (gather & ~ILM_GATHER_ASYNC) > ILM_GATHER_STORE) return api_fail(ILM_ERR_INVALID_ARGUMENT, "unknown gather mode %d", gather);
    if (stats) { stats->SdfSamples = 0; stats->PixelLightPairs = 0; stats->TracedPairs = 0; }
    // (a lightmap whose last exchange was asynchronous: the strips about to be rendered into it wait for that exchange)
    { const int32_t rc = wait_lightmap_exchange(m); if (rc != ILM_OK) return rc; }
    // store mode: armed for this call when the host has not armed the lightmap itself; nobody's pass may write into a member's frame
    // before that member's earlier readers of it are done (the fence in front), and the fence behind is the "gather"
    const bool arm_here = (gather == ILM_GATHER_STORE) && !m->store_mode;
    if (gather != ILM_GATHER_STORE && gather != ILM_GATHER_NONE && m->store_mode)
        return api_fail(ILM_ERR_STATE, "the group lightmap is in store mode: gather with ILM_GATHER_STORE (or switch it off)");
    if (arm_here) { const int32_t rc = set_store_mode(m, true); if (rc != ILM_OK) return rc; }
    if (gather == ILM_GATHER_STORE) { const int32_t rc = fence_members(g); if (rc != ILM_OK) return rc; }
    // every member's strip is queued before anything is waited for: the launches are asynchronous, the devices run concurrently
    // (the instrumented variant synchronises per member; it is a diagnostic)
    int32_t strip_rc = ILM_OK;
    for (int i = 0; i < g->n_local && strip_rc == ILM_OK; i++) {
        int32_t b = 0, e = 0;
        (void)ilm_group_lightmap_strip(hlightmap, g->first_rank + i, &b, &e, nullptr);
        IlmRenderStats part = { 0, 0, 0 };
        strip_rc = ilm_render_sphere_lights(g->ctx[(size_t)i], lights, light_count, env, df, gbuffers ? gbuffers[i] : 0, sdfs ? sdfs[i] : 0,
                                            ambient, m->lightmaps[(size_t)i], b, e, stats ? &part : nullptr);
        if (strip_rc == ILM_OK && stats) { stats->SdfSamples += part.SdfSamples; stats->PixelLightPairs += part.PixelLightPairs; stats->TracedPairs += part.TracedPairs; }
    }
    if (strip_rc != ILM_OK && !(g->rank_mode && g->world > 1)) {
        if (arm_here) (void)set_store_mode(m, false);
        return strip_rc;
    }
    // One process per GPU: the other ranks cannot know that this rank's strip failed -- they are on their way into the exchange (or the
    // fence of the store mode) and, when the call armed the mode itself, into the disarming collective behind it.  A failing rank walks
    // the SAME sequence (its rows of the frame are whatever the buffer held) and reports its error afterwards; leaving early would pair
    // its next collective with the wrong one of its peers and hang them.
    char strip_why[512];
    if (strip_rc != ILM_OK) snprintf(strip_why, sizeof(strip_why), "%s", ilm_last_error());
    const int32_t rc = gather_lightmap(m, gather);
    int32_t rc2 = ILM_OK;
    if (arm_here) rc2 = set_store_mode(m, false);
    if (strip_rc != ILM_OK) return api_fail(strip_rc, "%s", strip_why);
    return rc != ILM_OK ? rc : rc2;
}

int32_t ilm_group_live_counts(IlmHandle hgroup, const IlmHandle* systems, int32_t total_chunks, uint32_t* out_counts, int32_t capacity,
                              int32_t saturate16) {
    ILM_TRACE_RANGE("ilm_group_live_counts");
    Group* g = group_from(hgroup);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    if (!systems || (!out_counts && total_chunks > 0)) return api_fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (total_chunks < 0 || capacity < total_chunks) return api_fail(ILM_ERR_OUT_OF_RANGE, "capacity %d < chunk count %d", capacity, total_chunks);
    const int world = g->world;
    const int per_rank = (total_chunks + world - 1) / world;
    if (per_rank == 0) return ILM_OK;
    // this process's slots of the table, rank-major: slot (r, k) = chunk k * world + r
    std::vector<uint32_t> table((size_t)world * (size_t)per_rank, 0u);
    std::vector<uint32_t> local((size_t)per_rank, 0u);
    for (int i = 0; i < g->n_local; i++) {
        const int rank = g->first_rank + i;
        const int owned = (total_chunks - rank + world - 1) / world;          // chunks c < total with c % world == rank
        int32_t have = 0;
        int32_t rc = ilm_system_chunk_count(systems[i], &have);
        if (rc != ILM_OK) return rc;
        if (have != owned)
            return api_fail(ILM_ERR_STATE, "rank %d holds %d chunks but owns %d of a table of %d (chunk c lives on rank c %% %d)", rank, have, owned,
                            total_chunks, world);
        if (owned > 0) {
            rc = ilm_system_step_counts(systems[i], local.data(), per_rank, saturate16);
            if (rc != ILM_OK) return rc;
            memcpy(&table[(size_t)rank * (size_t)per_rank], local.data(), sizeof(uint32_t) * (size_t)owned);
        }
    }
    if (g->rank_mode && world > 1) {
        // the one exchange of the particle path: world * per_rank counters through RCCL
        const int32_t rc = host_all_gather(g, &table[(size_t)g->first_rank * (size_t)per_rank], table.data(), sizeof(uint32_t) * (size_t)per_rank);
        if (rc != ILM_OK) return rc;
    }
    for (int c = 0; c < total_chunks; c++)
        out_counts[c] = table[(size_t)(c % world) * (size_t)per_rank + (size_t)(c / world)];
    return ILM_OK;
}

int32_t ilm_group_gather_chunks(IlmHandle hgroup, const IlmHandle* sources, const IlmHandle* gathered, int32_t total_chunks, int32_t first_component,
                                int32_t component_count, int32_t gather) {
    ILM_TRACE_RANGE("ilm_group_gather_chunks");
    Group* g = group_from(hgroup);
    if (!g) return api_fail(ILM_ERR_INVALID_HANDLE, "not a group handle");
    if (!sources || !gathered) return api_fail(ILM_ERR_INVALID_ARGUMENT, "NULL argument");
    if (total_chunks < 0) return api_fail(ILM_ERR_OUT_OF_RANGE, "total_chunks %d", total_chunks);
    if (first_component < 0 || component_count < 1 || first_component + component_count > kComponents)
        return api_fail(ILM_ERR_OUT_OF_RANGE, "components [%d, %d) outside [0, %d)", first_component, first_component + component_count, kComponents);
    return gather_chunks(g, sources, gathered, total_chunks, first_component, component_count, gather);
}


}  // extern "C"
