// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY (tests/test_two_ranks_one_gpu.py builds and loads it through ILM_RCCL_LIB).
//
// RCCL refuses a communicator with two ranks on one device, and the build's GPU box has ONE GPU: the one-process-per-GPU shape of
// group.hip (ilm_group_create_rank: ncclCommInitRank, the ncclSend / ncclRecv range exchange, the in-place ncclAllGather of the small
// host payloads) and the whole N > 1 branch of bench.py could therefore only run at world size 1.  This file is a stand-in for
// librccl.so.1 with the eleven entry points group.hip binds, moving the bytes through POSIX shared memory between RANK PROCESSES THAT
// SHARE ONE GPU: device -> shared host memory -> device.  Semantics are STRONGER than RCCL's (every operation of a group has completed
// when ncclGroupEnd returns: the stream is drained first), never weaker, so a protocol that is correct over this is not thereby proven
// correct over RCCL's asynchrony -- but every offset, byte count, rank pairing, collective ordering and Python-level path is exercised
// for real at world size > 1.  Nothing in the product loads it.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kBox = (size_t)4 << 20;           // bytes per mailbox chunk

struct Shm {
    std::atomic<uint32_t> ready;                    // rank 0 has initialised the region
    std::atomic<uint32_t> barrier_count, barrier_generation;
    std::atomic<uint64_t> sent[kMaxRanks][kMaxRanks], taken[kMaxRanks][kMaxRanks];     // chunks written into / read from box (src, dst)
    // followed by world * world boxes of kBox bytes
};

struct Comm {
    int rank = 0, world = 0;
    Shm* shm = nullptr; size_t bytes = 0; char name[64] = "";
    char* box(int src, int dst) const { return reinterpret_cast<char*>(shm) + 4096 + ((size_t)src * (size_t)world + (size_t)dst) * kBox; }
};

struct Op { bool send; Comm* comm; char* device; size_t bytes, done; int peer; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

void nap() { struct timespec t = { 0, 20000 }; nanosleep(&t, nullptr); }

void barrier(Comm* c) {
    const uint32_t generation = c->shm->barrier_generation.load();
    if (c->shm->barrier_count.fetch_add(1) + 1 == (uint32_t)c->world) {
        c->shm->barrier_count.store(0);
        c->shm->barrier_generation.fetch_add(1);
    } else {
        while (c->shm->barrier_generation.load() == generation) nap();
    }
}

// every queued operation, advanced chunk by chunk without ever blocking on one of them (a rank that only sent first and received later
// would deadlock against a peer doing the same as soon as a message is larger than its box)
int run_ops() {
    for (Op& o : g_ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;       // what the stream has queued so far precedes the operation
    bool busy = true;
    while (busy) {
        busy = false;
        bool progressed = false;
        for (size_t index = 0; index < g_ops.size(); index++) {
            Op& o = g_ops[index];
            if (o.done >= o.bytes) continue;
            busy = true;
            // (operations of one direction with one peer share a box: strictly one after the other)
            bool behind = false;
            for (size_t e = 0; e < index && !behind; e++)
                behind = g_ops[e].done < g_ops[e].bytes && g_ops[e].send == o.send && g_ops[e].peer == o.peer && g_ops[e].comm == o.comm;
            if (behind) continue;
            Comm* c = o.comm;
            const size_t n = (o.bytes - o.done < kBox) ? o.bytes - o.done : kBox;
            if (o.send) {
                std::atomic<uint64_t>& sent = c->shm->sent[c->rank][o.peer];
                if (sent.load() != c->shm->taken[c->rank][o.peer].load()) continue;          // the box is still full
                if (hipMemcpy(c->box(c->rank, o.peer), o.device + o.done, n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
                sent.fetch_add(1);
            } else {
                std::atomic<uint64_t>& taken = c->shm->taken[o.peer][c->rank];
                if (c->shm->sent[o.peer][c->rank].load() == taken.load()) continue;          // nothing there yet
                if (hipMemcpy(o.device + o.done, c->box(o.peer, c->rank), n, hipMemcpyHostToDevice) != hipSuccess) return 1;
                taken.fetch_add(1);
            }
            o.done += n;
            progressed = true;
        }
        if (busy && !progressed) nap();
    }
    g_ops.clear();
    return 0;
}

int queue(Op o) {
    if (o.bytes == 0) return 0;
    g_ops.push_back(o);
    return g_depth > 0 ? 0 : run_ops();
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 16) != 16) return 1;
    close(fd);
    return 0;
}

int ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return 4;
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    c->bytes = 4096 + (size_t)world * (size_t)world * kBox;
    static_assert(sizeof(Shm) <= 4096, "header fits its page");
    snprintf(c->name, sizeof(c->name), "/ilm_fake_rccl_%02x%02x%02x%02x%02x%02x%02x%02x", (unsigned char)id.internal[0], (unsigned char)id.internal[1],
             (unsigned char)id.internal[2], (unsigned char)id.internal[3], (unsigned char)id.internal[4], (unsigned char)id.internal[5],
             (unsigned char)id.internal[6], (unsigned char)id.internal[7]);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 2;
    } else {
        for (int tries = 0; tries < 30000 && fd < 0; tries++) {             // ~10 minutes at most
            fd = shm_open(c->name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->bytes)) { close(fd); fd = -1; }
            if (fd < 0) nap();
        }
        if (fd < 0) return 2;
    }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->shm = static_cast<Shm*>(p);
    if (rank == 0) {
        memset(p, 0, 4096);
        c->shm->ready.store(1);
    } else {
        while (c->shm->ready.load() != 1) nap();
    }
    barrier(c);
    *out = c;
    return 0;
}

int ncclCommInitAll(void**, int, const int*) { return 5; }      // several devices in one process: not what this stand-in is for

int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    barrier(c);
    munmap(c->shm, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

int ncclCommCount(const void* comm, int* count) { *count = static_cast<const Comm*>(comm)->world; return 0; }

int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() { return (--g_depth == 0) ? run_ops() : 0; }

int ncclSend(const void* buf, size_t count, int /* ncclInt8 */, int peer, void* comm, hipStream_t stream) {
    return queue(Op{ true, static_cast<Comm*>(comm), static_cast<char*>(const_cast<void*>(buf)), count, 0, peer, stream });
}
int ncclRecv(void* buf, size_t count, int, int peer, void* comm, hipStream_t stream) {
    return queue(Op{ false, static_cast<Comm*>(comm), static_cast<char*>(buf), count, 0, peer, stream });
}

// in place or not: rank r's `count` bytes end at recv + r * count on every rank
int ncclAllGather(const void* send, void* recv, size_t count, int, void* comm, hipStream_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    char* mine = static_cast<char*>(recv) + (size_t)c->rank * count;
    if (send != mine && count > 0) {
        if (hipStreamSynchronize(stream) != hipSuccess) return 1;
        if (hipMemcpy(mine, send, count, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    }
    g_depth++;
    for (int k = 1; k < c->world; k++) {
        const int to = (c->rank + k) % c->world, from = (c->rank - k + c->world) % c->world;
        queue(Op{ true, c, mine, count, 0, to, stream });
        queue(Op{ false, c, static_cast<char*>(recv) + (size_t)from * count, count, 0, from, stream });
    }
    return (--g_depth == 0) ? run_ops() : 0;
}

const char* ncclGetErrorString(int code) {
    switch (code) { case 0: return "success"; case 1: return "HIP error inside the stand-in"; case 2: return "shared memory"; case 4: return "invalid argument";
                    case 5: return "not supported by the stand-in"; default: return "error"; }
}

}  // extern "C"
