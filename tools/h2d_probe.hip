// Host-to-device copy rates of page-locked memory, as the capture loop uses it (vslam_host_alloc + vslam_upload_async):
// which hipHostMalloc flags, how many streams, and a copy kernel reading the pinned buffer over the link instead of the
// DMA engine.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_probe tools/h2d_probe.hip && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

__global__ void pull_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t bytes = (size_t)128 * 1280 * 720 * 3;   // two of the loop's 64-frame buffers
    uint8_t *dev = nullptr;
    CK(hipMalloc(&dev, bytes));
    hipStream_t st[4];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Kind {
        const char *name;
        unsigned flags;
    } kinds[] = {{"default", hipHostMallocDefault},
                 {"non-coherent", hipHostMallocNonCoherent},
                 {"write-combined", hipHostMallocWriteCombined},
                 {"numa-user", hipHostMallocNumaUser},
                 {"portable|mapped", hipHostMallocPortable | hipHostMallocMapped}};
    for (const Kind &k : kinds) {
        uint8_t *h = nullptr;
        if (hipHostMalloc((void **)&h, bytes, k.flags) != hipSuccess) {
            std::printf("%-16s allocation refused\n", k.name);
            (void)hipGetLastError();
            continue;
        }
        std::memset(h, 1, bytes);
        for (int streams : {1, 2, 4}) {
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                const size_t part = bytes / streams;
                for (int s = 0; s < streams; s++)
                    CK(hipMemcpyAsync(dev + part * s, h + part * s, part, hipMemcpyHostToDevice, st[s]));
                CK(hipDeviceSynchronize());
                best = std::min(best, now() - t0);
            }
            std::printf("%-16s memcpy, %d stream(s): %6.1f GB/s\n", k.name, streams, bytes / best / 1e9);
        }
        for (int blocks : {64, 256, 1024}) {
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                pull_kernel<<<blocks, 256, 0, st[0]>>>((const uint4 *)h, (uint4 *)dev, bytes / 16);
                CK(hipDeviceSynchronize());
                best = std::min(best, now() - t0);
            }
            std::printf("%-16s kernel pull, %4d workgroups: %6.1f GB/s\n", k.name, blocks, bytes / best / 1e9);
        }
        CK(hipHostFree(h));
    }
    {   // ordinary memory registered afterwards
        uint8_t *h = (uint8_t *)std::aligned_alloc(4096, bytes);
        std::memset(h, 1, bytes);
        CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
        double best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, st[0]));
            CK(hipDeviceSynchronize());
            best = std::min(best, now() - t0);
        }
        std::printf("%-16s memcpy, 1 stream(s): %6.1f GB/s\n", "registered", bytes / best / 1e9);
        CK(hipHostUnregister(h));
        std::free(h);
    }
    {   // host side alone: how fast do T threads fill a page-locked buffer from the page cache (tmpfs), frame by frame as
        // the capture loop's readers do, with nothing else going on and with an upload of another buffer beside them?
        const size_t frame = (size_t)1280 * 720 * 3, frames = bytes / frame;
        uint8_t *h = nullptr, *h2 = nullptr;
        CK(hipHostMalloc((void **)&h, bytes, hipHostMallocDefault));
        CK(hipHostMalloc((void **)&h2, bytes, hipHostMallocDefault));
        std::memset(h, 3, bytes);
        std::memset(h2, 4, bytes);
        const char *path = "/dev/shm/h2d_probe.bin";
        int fd = ::open(path, O_CREAT | O_TRUNC | O_RDWR, 0600);
        if (fd < 0 || ::write(fd, h, bytes) != (ssize_t)bytes) {
            std::fprintf(stderr, "cannot write %s\n", path);
            return 1;
        }
        for (int beside : {0, 1})
            for (int threads : {1, 2, 4, 8, 12, 16, 24, 32}) {
                double best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipDeviceSynchronize());
                    const double t0 = now();
                    if (beside) CK(hipMemcpyAsync(dev, h2, bytes, hipMemcpyHostToDevice, st[0]));
                    std::vector<std::thread> pool;
                    for (int t = 0; t < threads; t++)
                        pool.emplace_back([&, t] {
                            for (size_t f = (size_t)t; f < frames; f += (size_t)threads) {
                                size_t got = 0;
                                while (got < frame) {
                                    const ssize_t r = ::pread(fd, h + f * frame + got, frame - got, (off_t)(f * frame + got));
                                    if (r <= 0) std::exit(2);
                                    got += (size_t)r;
                                }
                            }
                        });
                    for (auto &th : pool) th.join();
                    const double t_read = now() - t0;
                    CK(hipDeviceSynchronize());
                    best = std::min(best, beside ? now() - t0 : t_read);
                }
                std::printf("pread into page-locked memory, %2d threads%s: %6.1f GB/s\n", threads,
                            beside ? " + an upload of the same size beside them (both done)" : "", bytes / best / 1e9);
            }
        ::close(fd);
        ::unlink(path);
        CK(hipHostFree(h));
        CK(hipHostFree(h2));
    }
    CK(hipFree(dev));
    return 0;
}
