// The transport interface of comm.hip (one per communicator) -- shared with the TEST transport plugin
// (test_transport.cpp -> libathena_mp_testcomm.so), which is built from this header by the same compiler and is NOT part of
// libathena_mp.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace amp_comm {

struct Transport {
    int rank = 0, world = 1;
    // what this rank has put on the wire, for athena_mp_comm_stats: bytes to every peer (grouped sends and its blocks of
    // all-gathers), all-reduce payload, number of transfers started
    std::vector<int64_t> sent;
    int64_t allreduce_bytes = 0, transfers = 0;
    void count_sent(int p, size_t bytes)
    {
        if ((int)sent.size() != world) sent.assign((size_t)world, 0);
        if (p >= 0 && p < world) sent[(size_t)p] += (int64_t)bytes;
    }
    virtual ~Transport() {}
    virtual const char *name() const = 0;
    virtual int ranks_seen() const { return world; }   // how many ranks the transport's own communicator reports
    virtual int version() const { return 0; }          // RCCL: ncclGetVersion
    // device buffers; bytes may be 0 (skipped); entries for p == rank are ignored.  Enqueued on / ordered after `s`.
    virtual int exchange(const void *const *sendp, const size_t *sendb, void *const *recvp, const size_t *recvb,
                         hipStream_t s) = 0;
    virtual int allreduce_f32(float *buf, size_t count, hipStream_t s) = 0;
    // every rank contributes `bytes` from send; recv holds world * bytes, block p at p * bytes (send may be recv + rank * bytes)
    virtual int allgather(const void *send, void *recv, size_t bytes, hipStream_t s) = 0;
    // what every caller in this file uses: the same three, counted
    int counted_exchange(const void *const *sendp, const size_t *sendb, void *const *recvp, const size_t *recvb, hipStream_t s)
    {
        for (int p = 0; p < world; ++p)
            if (p != rank) count_sent(p, sendb[p]);
        ++transfers;
        return exchange(sendp, sendb, recvp, recvb, s);
    }
    int counted_allreduce_f32(float *buf, size_t count, hipStream_t s)
    {
        allreduce_bytes += (int64_t)(4 * count);
        ++transfers;
        return allreduce_f32(buf, count, s);
    }
    int counted_allgather(const void *send, void *recv, size_t bytes, hipStream_t s)
    {
        for (int p = 0; p < world; ++p)
            if (p != rank) count_sent(p, bytes);
        ++transfers;
        return allgather(send, recv, bytes, s);
    }
};

// what the test plugin exports (extern "C"); comm.hip resolves them with dlsym when ATHENA_MP_COMM_TRANSPORT=shm
typedef void (*error_fn)(const char *message);
typedef Transport *(*test_create_fn)(const void *id128, error_fn on_error);
typedef int (*test_unique_id_fn)(void *id128);

} // namespace amp_comm
