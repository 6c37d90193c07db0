// comm.hip -- the one collective of the evaluation path: start-up broadcast of the encoded (NTT-form) weights and the
// evaluation keys over RCCL (xGMI inside a node), plus the small helpers a multi-GPU driver needs to verify it.
//
// images are independent, so a batch shards over the GPUs with no data-path collective.  Rank `root` encodes the
// model once (lift + NTT of ~0.5 M plaintexts, 35-200 GiB), everybody else receives it with ncclBroadcast.  xGMI is
// point-to-point (7 links x ~153 GB/s per GPU) and RCCL's broadcast is a ring/tree of per-link transfers, so a few large
// messages are what we want: pieces of 1 GiB keep RCCL's staging small and its pipelining busy.  The reference has no
// analogue (its only parallelism is the std::thread fan-out of convolutionalLayer.cpp:177-191).
#include "kernels.h"
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/random.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

// Rehearsal transport (CRC_COMM_TRANSPORT=shm).  RCCL refuses two ranks on one GPU, and the boxes this engine is developed on have one: with this switch the
// SAME entry points (crc_comm_unique_id / crc_comm_create / crc_broadcast_weights / crc_comm_allgather_u64) move their bytes through a POSIX shared-memory
// segment staged by hipMemcpy, so the multi-rank host code above them -- Network::broadcastParameters, bench_host's rendezvous, barrier and max-over-ranks
// timing -- runs end to end with two processes on one device (tests/test_gpu_comm.py).  It is a transport for tests, never chosen by default and never faster
// than RCCL.
struct ShmSeg {
    std::atomic<uint64_t> arrived, generation;    // sense-reversing barrier of `world` processes
    uint64_t gather[64 * 64];                      // all-gather slots: [rank][kScratchWords]
    unsigned char stage[1];                        // broadcast staging (kShmStage bytes)
};
static const size_t kShmStage = (size_t)64 << 20;
static const char kShmMagic[8] = {'C', 'R', 'C', 'S', 'H', 'M', ':', 0};

struct crc_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    u64 *d_scratch = nullptr;                     // [kScratchWords * (world + 1)]: all-gather send + receive staging
    ShmSeg *shm = nullptr;                        // rehearsal transport (see above): the mapped segment, else null
    char shm_name[64] = {0};
};

static const size_t kPieceWords = (size_t)1 << 27;     // 1 GiB
static const size_t kScratchWords = 64;                // per-rank payload limit of crc_comm_allgather_u64

static thread_local int g_last_nccl = 0;
extern "C" int crc_last_comm_error(void) { return g_last_nccl; }
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { g_last_nccl = (int)r_; return CRC_ERR_COMM; } } while (0)

static_assert(sizeof(ncclUniqueId) == CRC_COMM_ID_BYTES, "rendezvous id size");

// makes `device` current for the calling thread and puts the caller's device back on scope exit: in the one-process, many-GPU flow (crc_comm_create_all /
// crc_broadcast_weights_all) a comm call must not leave the thread on the last communicator's GPU, or the next kernel call for another context runs there
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int device) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != device) ok = hipSetDevice(device) == hipSuccess; }
    ~DeviceGuard() { int cur = -1; if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev); }
};
#define GUARD(dev) DeviceGuard guard_(dev); if (!guard_.ok) return crc_set_hip_error(hipErrorInvalidDevice)

static bool shm_wanted() { const char *e = std::getenv("CRC_COMM_TRANSPORT"); return e && !std::strcmp(e, "shm"); }
// all `world` processes arrive, the last one opens the next generation; bounded: a rank that never arrives is an error after 10 minutes, not a hang
static int shm_barrier(crc_comm *cm)
{
    ShmSeg *s = cm->shm;
    const uint64_t gen = s->generation.load(std::memory_order_acquire);
    if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint64_t)cm->world) {
        s->arrived.store(0, std::memory_order_relaxed);
        s->generation.store(gen + 1, std::memory_order_release);
        return CRC_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load(std::memory_order_acquire) == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::minutes(10)) return CRC_ERR_COMM;
    }
    return CRC_OK;
}

extern "C" int crc_comm_unique_id(uint8_t *h_id)
{
    if (!h_id) return CRC_ERR_INVALID_ARGUMENT;
    if (shm_wanted()) {                           // the "id" names a fresh shared-memory segment
        std::memset(h_id, 0, CRC_COMM_ID_BYTES);
        std::memcpy(h_id, kShmMagic, sizeof kShmMagic);
        // an unguessable name, and the segment itself made HERE, exclusively (O_EXCL: a name somebody else pre-created is an error, never adopted) and sized; the
        // ranks -- this one included -- open it without O_CREAT in crc_comm_create.  A fresh segment reads as zeros, which is the barrier's initial state
        unsigned long long rnd[2] = {0, 0};
        if (getrandom(rnd, sizeof rnd, 0) != (ssize_t)sizeof rnd) return CRC_ERR_COMM;
        std::snprintf((char *)h_id + 8, 48, "/crc_comm_%016llx%016llx", rnd[0], rnd[1]);
        const int fd = shm_open((const char *)h_id + 8, O_CREAT | O_EXCL | O_RDWR, 0600);
        const bool ok = fd >= 0 && ftruncate(fd, (off_t)(sizeof(ShmSeg) + kShmStage)) == 0;
        if (fd >= 0) close(fd);
        if (!ok) { if (fd >= 0) shm_unlink((const char *)h_id + 8); return CRC_ERR_COMM; }
        return CRC_OK;
    }
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    std::memcpy(h_id, &id, sizeof id);
    return CRC_OK;
}

static int comm_finish(crc_comm *cm)
{
    GUARD(cm->device);
    HIPCHK(hipMalloc((void **)&cm->d_scratch, kScratchWords * (size_t)(cm->world + 1) * 8));
    return CRC_OK;
}

extern "C" int crc_comm_create(crc_ctx *c, int world, int rank, const uint8_t *h_id, crc_comm **out)
{
    if (!c || c->device < 0 || !h_id || !out || world < 1 || rank < 0 || rank >= world) return CRC_ERR_INVALID_ARGUMENT;
    GUARD(c->device);
    if (!std::memcmp(h_id, kShmMagic, sizeof kShmMagic)) {
        if (world > 64) return CRC_ERR_INVALID_ARGUMENT;
        crc_comm *cm = new crc_comm(); cm->world = world; cm->rank = rank; cm->device = c->device;
        std::snprintf(cm->shm_name, sizeof cm->shm_name, "%s", (const char *)h_id + 8);
        const size_t bytes = sizeof(ShmSeg) + kShmStage;
        // the segment exists (crc_comm_unique_id made and sized it): no rank creates it here
        const int fd = shm_open(cm->shm_name, O_RDWR, 0);
        void *p = fd >= 0 ? mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
        if (fd >= 0) close(fd);
        if (p == MAP_FAILED) { delete cm; return CRC_ERR_COMM; }
        cm->shm = (ShmSeg *)p;
        const int rc = comm_finish(cm);
        if (rc || (rc == 0 && shm_barrier(cm))) { crc_comm_destroy(cm); return rc ? rc : CRC_ERR_COMM; }
        *out = cm;
        return CRC_OK;
    }
    ncclUniqueId id; std::memcpy(&id, h_id, sizeof id);
    crc_comm *cm = new crc_comm(); cm->world = world; cm->rank = rank; cm->device = c->device;
    ncclResult_t r = ncclCommInitRank(&cm->comm, world, id, rank);
    if (r != ncclSuccess) { g_last_nccl = (int)r; delete cm; return CRC_ERR_COMM; }
    const int rc = comm_finish(cm);
    if (rc) { crc_comm_destroy(cm); return rc; }
    *out = cm;
    return CRC_OK;
}

extern "C" int crc_comm_create_all(crc_ctx *const *ctxs, int ndev, crc_comm **out)
{
    if (!ctxs || !out || ndev < 1) return CRC_ERR_INVALID_ARGUMENT;
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) {
        if (!ctxs[i] || ctxs[i]->device < 0) return CRC_ERR_INVALID_ARGUMENT;
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++) if (devs[j] == devs[i]) return CRC_ERR_INVALID_ARGUMENT;      // one rank per GPU
    }
    std::vector<ncclComm_t> comms(ndev);
    GUARD(devs[0]);                                      // ncclCommInitAll walks the devices: the guard puts the caller's back
    NCCLCHK(ncclCommInitAll(comms.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; i++) {
        crc_comm *cm = new crc_comm(); cm->comm = comms[i]; cm->world = ndev; cm->rank = i; cm->device = devs[i];
        out[i] = cm;
        const int rc = comm_finish(cm);
        if (rc) { for (int j = 0; j <= i; j++) { crc_comm_destroy(out[j]); out[j] = nullptr; } for (int j = i + 1; j < ndev; j++) ncclCommDestroy(comms[j]);
            return rc; }
    }
    return CRC_OK;
}

extern "C" void crc_comm_destroy(crc_comm *cm)
{
    if (!cm) return;
    DeviceGuard guard_(cm->device);
    if (cm->d_scratch) (void)hipFree(cm->d_scratch);
    if (cm->shm) { munmap(cm->shm, sizeof(ShmSeg) + kShmStage); if (cm->rank == 0) shm_unlink(cm->shm_name); }
    if (cm->comm) ncclCommDestroy(cm->comm);
    delete cm;
}
extern "C" int crc_comm_rank(const crc_comm *cm) { return cm ? cm->rank : CRC_ERR_INVALID_ARGUMENT; }
extern "C" int crc_comm_world(const crc_comm *cm) { return cm ? cm->world : CRC_ERR_INVALID_ARGUMENT; }

extern "C" int crc_broadcast_weights(crc_comm *cm, uint64_t *d_w, size_t words, int root, void *stream)
{
    if (!cm || (!d_w && words) || root < 0 || root >= cm->world) return CRC_ERR_INVALID_ARGUMENT;
    GUARD(cm->device);
    if (cm->shm) {                                       // rehearsal transport: root -> staging -> everybody else, a staging buffer at a time
        hipStream_t st = (hipStream_t)stream;
        for (size_t o = 0; o < words; o += kShmStage / 8) {
            const size_t cnt = words - o < kShmStage / 8 ? words - o : kShmStage / 8;
            if (cm->rank == root) { HIPCHK(hipMemcpyAsync(cm->shm->stage, d_w + o, cnt * 8, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); }
            if (shm_barrier(cm)) return CRC_ERR_COMM;
            if (cm->rank != root) { HIPCHK(hipMemcpyAsync(d_w + o, cm->shm->stage, cnt * 8, hipMemcpyHostToDevice, st)); HIPCHK(hipStreamSynchronize(st)); }
            if (shm_barrier(cm)) return CRC_ERR_COMM;
        }
        return CRC_OK;
    }
    for (size_t o = 0; o < words; o += kPieceWords) {
        const size_t cnt = words - o < kPieceWords ? words - o : kPieceWords;
        NCCLCHK(ncclBroadcast(d_w + o, d_w + o, cnt, ncclUint64, root, cm->comm, (hipStream_t)stream));
    }
    return CRC_OK;
}

extern "C" int crc_broadcast_weights_all(crc_comm *const *comms, int ndev, uint64_t *const *d_w, size_t words, int root, void *const *streams)
{
    if (!comms || !d_w || ndev < 1 || root < 0 || root >= ndev) return CRC_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < ndev; i++) if (!comms[i] || comms[i]->world != ndev || (!d_w[i] && words)) return CRC_ERR_INVALID_ARGUMENT;
    GUARD(comms[0]->device);                             // RCCL switches devices inside the group: the caller's comes back on exit
    for (size_t o = 0; o < words; o += kPieceWords) {
        const size_t cnt = words - o < kPieceWords ? words - o : kPieceWords;
        NCCLCHK(ncclGroupStart());                 // one thread drives every rank: the calls of a piece must be grouped
        for (int i = 0; i < ndev; i++) {
            ncclResult_t r = ncclBroadcast(d_w[i] + o, d_w[i] + o, cnt, ncclUint64, root, comms[i]->comm, streams ? (hipStream_t)streams[i] : nullptr);
            if (r != ncclSuccess) { g_last_nccl = (int)r; ncclGroupEnd(); return CRC_ERR_COMM; }
        }
        NCCLCHK(ncclGroupEnd());
    }
    return CRC_OK;
}

extern "C" int crc_comm_allgather_u64(crc_comm *cm, const uint64_t *h_in, size_t words, uint64_t *h_out, void *stream)
{
    if (!cm || !h_in || !h_out || words == 0 || words > kScratchWords) return CRC_ERR_INVALID_ARGUMENT;
    GUARD(cm->device);
    hipStream_t st = (hipStream_t)stream;
    if (cm->shm) {
        HIPCHK(hipStreamSynchronize(st));                // (the RCCL form synchronises the stream too: callers use this as their barrier)
        std::memcpy(cm->shm->gather + (size_t)cm->rank * kScratchWords, h_in, words * 8);
        if (shm_barrier(cm)) return CRC_ERR_COMM;
        for (int r = 0; r < cm->world; r++) std::memcpy(h_out + (size_t)r * words, cm->shm->gather + (size_t)r * kScratchWords, words * 8);
        return shm_barrier(cm) ? CRC_ERR_COMM : CRC_OK;
    }
    u64 *send = cm->d_scratch, *recv = cm->d_scratch + kScratchWords;
    HIPCHK(hipMemcpyAsync(send, h_in, words * 8, hipMemcpyHostToDevice, st));
    NCCLCHK(ncclAllGather(send, recv, words, ncclUint64, cm->comm, st));
    HIPCHK(hipMemcpyAsync(h_out, recv, words * 8 * (size_t)cm->world, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return CRC_OK;
}

// ---- checksum of a device buffer (what a receiving rank compares with the root's) -------------------------------------
__global__ void __launch_bounds__(256) checksum_kernel(const u64 *w, size_t words, u64 *out)
{
    u64 x = 0, s = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) { const u64 v = w[i]; x ^= v; s += v * (2 * (u64)i + 1); }
    for (int off = 32; off > 0; off >>= 1) { x ^= __shfl_down(x, off, 64); s += __shfl_down(s, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicXor((unsigned long long *)&out[0], (unsigned long long)x);
        atomicAdd((unsigned long long *)&out[1], (unsigned long long)s); }
}

extern "C" int crc_checksum64(crc_ctx *c, const uint64_t *d_words, size_t words, uint64_t *h_out, void *stream)
{
    if (!c || c->device < 0 || !h_out || (!d_words && words)) return CRC_ERR_INVALID_ARGUMENT;
    // the kernel and the copies below belong to THIS context's GPU, whatever the thread's current device is
    GUARD(c->device);
    hipStream_t st = (hipStream_t)stream;
    // the two accumulator words live in the context's scratch: 256 slots handed out round robin, so that checksums issued from several host threads (each on
    // its own stream) neither share accumulators nor wait for one another; a slot comes round again after 255 other calls, each of which has synchronised its
    // stream
    const unsigned slot = c->scratch_next.fetch_add(1, std::memory_order_relaxed) & 255u;
    u64 *acc = c->d_scratch + 2 * slot;
    HIPCHK(hipMemsetAsync(acc, 0, 16, st));
    if (words) {
        size_t blocks = (words + 255) / 256; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(checksum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_words, words, acc);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(h_out, acc, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return CRC_OK;
}
