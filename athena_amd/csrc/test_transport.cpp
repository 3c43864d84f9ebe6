// TEST transport of comm.hip as a plugin of its own: libathena_mp_testcomm.so (built by build.sh next to libathena_mp.so, loaded
// by comm.hip only when ATHENA_MP_COMM_TRANSPORT=shm).  The same calls as the RCCL transport with the bytes staged through files
// under /dev/shm -- TEST INFRASTRUCTURE for boxes with ONE GPU (RCCL refuses two ranks on one device: "Duplicate GPU detected") and
// for bench.py's one-device dry run; it exercises everything in comm.hip except the nccl* calls themselves.  Never part of the
// product library, never selected by default, and bench.py refuses a line that moved its halos through it outside the dry run.
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "transport.h"

using amp_comm::Transport;

namespace {
amp_comm::error_fn g_on_error = nullptr;
void set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g_on_error) g_on_error(buf);
}
#define TT_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

// test-only: bytes through /dev/shm files, one file per (sequence number, source, destination)
struct ShmTransport : Transport {
    std::string dir;
    uint64_t seq = 0;
    std::vector<char> host;
    const char *name() const override { return "shm (test transport: host-staged files, not RCCL)"; }
    ~ShmTransport() override { (void)rmdir(dir.c_str()); }   // succeeds for the last rank out (directory empty)
    std::string path(uint64_t q, int src, int dst) const
    {
        char b[96];
        snprintf(b, sizeof(b), "/m_%llu_%d_%d", (unsigned long long)q, src, dst);
        return dir + b;
    }
    int put(const std::string &file, const void *data, size_t bytes)
    {
        const std::string tmp = file + ".tmp";
        int fd = open(tmp.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0600);
        if (fd < 0) {
            set_error("comm(shm): cannot create %s: %s", tmp.c_str(), strerror(errno));
            return 1;
        }
        const char *p = (const char *)data;
        size_t left = bytes;
        while (left) {
            ssize_t w = write(fd, p, left);
            if (w <= 0) {
                close(fd);
                set_error("comm(shm): write failed: %s", strerror(errno));
                return 1;
            }
            p += w;
            left -= (size_t)w;
        }
        close(fd);
        if (rename(tmp.c_str(), file.c_str())) {
            set_error("comm(shm): rename failed: %s", strerror(errno));
            return 1;
        }
        return 0;
    }
    int get(const std::string &file, void *data, size_t bytes)
    {
        const double t0 = now();
        int fd = -1;
        while ((fd = open(file.c_str(), O_RDONLY)) < 0) {
            if (now() - t0 > 300.0) {
                set_error("comm(shm): timed out waiting for %s", file.c_str());
                return 1;
            }
            usleep(200);
        }
        char *p = (char *)data;
        size_t left = bytes;
        while (left) {
            ssize_t r = read(fd, p, left);
            if (r <= 0) {
                close(fd);
                set_error("comm(shm): short read of %s", file.c_str());
                return 1;
            }
            p += r;
            left -= (size_t)r;
        }
        close(fd);
        unlink(file.c_str());
        return 0;
    }
    static double now()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec + 1e-9 * ts.tv_nsec;
    }
    int exchange(const void *const *sendp, const size_t *sendb, void *const *recvp, const size_t *recvb,
                 hipStream_t s) override
    {
        const uint64_t q = seq++;
        TT_HIP(hipStreamSynchronize(s));
        for (int p = 0; p < world; ++p) {
            if (p == rank || !sendb[p]) continue;
            host.resize(sendb[p]);
            TT_HIP(hipMemcpy(host.data(), sendp[p], sendb[p], hipMemcpyDeviceToHost));
            if (put(path(q, rank, p), host.data(), sendb[p])) return 1;
        }
        for (int p = 0; p < world; ++p) {
            if (p == rank || !recvb[p]) continue;
            host.resize(recvb[p]);
            if (get(path(q, p, rank), host.data(), recvb[p])) return 1;
            TT_HIP(hipMemcpy(recvp[p], host.data(), recvb[p], hipMemcpyHostToDevice));
        }
        return 0;
    }
    int allreduce_f32(float *buf, size_t count, hipStream_t s) override
    {
        if (!count) return 0;
        const uint64_t q = seq++;
        TT_HIP(hipStreamSynchronize(s));
        std::vector<float> mine(count), other(count), sum(count, 0.0f);
        TT_HIP(hipMemcpy(mine.data(), buf, 4 * count, hipMemcpyDeviceToHost));
        for (int p = 0; p < world; ++p)
            if (p != rank && put(path(q, rank, p), mine.data(), 4 * count)) return 1;
        for (int p = 0; p < world; ++p) {   // rank order: every rank forms the same sum
            const float *src = mine.data();
            if (p != rank) {
                if (get(path(q, p, rank), other.data(), 4 * count)) return 1;
                src = other.data();
            }
            for (size_t i = 0; i < count; ++i) sum[i] += src[i];
        }
        TT_HIP(hipMemcpy(buf, sum.data(), 4 * count, hipMemcpyHostToDevice));
        return 0;
    }
    int allgather(const void *send, void *recv, size_t bytes, hipStream_t s) override
    {
        if (!bytes) return 0;
        const uint64_t q = seq++;
        TT_HIP(hipStreamSynchronize(s));
        host.resize(bytes);
        TT_HIP(hipMemcpy(host.data(), send, bytes, hipMemcpyDeviceToHost));
        for (int p = 0; p < world; ++p)
            if (p != rank && put(path(q, rank, p), host.data(), bytes)) return 1;
        if ((const char *)send != (char *)recv + (size_t)rank * bytes)
            TT_HIP(hipMemcpy((char *)recv + (size_t)rank * bytes, send, bytes, hipMemcpyDeviceToDevice));
        for (int p = 0; p < world; ++p) {
            if (p == rank) continue;
            if (get(path(q, p, rank), host.data(), bytes)) return 1;
            TT_HIP(hipMemcpy((char *)recv + (size_t)p * bytes, host.data(), bytes, hipMemcpyHostToDevice));
        }
        return 0;
    }
};

} // namespace

extern "C" {

int athena_mp_test_transport_unique_id(void *id128)
{
    memset(id128, 0, 128);
    int fd = open("/dev/urandom", O_RDONLY);
    const bool ok = fd >= 0 && read(fd, id128, 16) == 16;
    if (fd >= 0) close(fd);
    return ok ? 0 : 1;
}

Transport *athena_mp_test_transport_create(const void *id128, amp_comm::error_fn on_error)
{
    g_on_error = on_error;
    ShmTransport *t = new ShmTransport();
    char hex[40];
    const unsigned char *b = (const unsigned char *)id128;
    for (int i = 0; i < 16; ++i) snprintf(hex + 2 * i, 3, "%02x", b[i]);
    t->dir = std::string("/dev/shm/athena_mp_") + hex;
    if (mkdir(t->dir.c_str(), 0700) && errno != EEXIST) {
        set_error("comm_create(shm): cannot create %s: %s", t->dir.c_str(), strerror(errno));
        delete t;
        return nullptr;
    }
    return t;
}

} // extern "C"
