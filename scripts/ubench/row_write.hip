// Micro-benchmark: storing [N][R] fp32 rows, R = 64 (256 B: two whole lines) against R = 72 (288 B: rows straddle 128 B
// lines), rows visited in natural order or in the order of a tile permutation (16-row tiles of a random vertex order, as
// the bucket-sorted Duvenaud kernels visit them).  Persistent waves (2048 x 4), 16 lanes x 16 B per row.
// Build: hipcc --offload-arch=gfx950 -O3 row_write.hip -o row_write
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));

// mode 0: write rows; mode 1: read rows (sum); R in floats, chunks = R / 4; ids [N] row of each slot
template <int MODE>
__global__ __launch_bounds__(256) void rows_k(const int *__restrict__ ids, float *__restrict__ y, int N, int R, float *out)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int chunks = R / 4, tiles = N / 16;
    v4f acc = {0, 0, 0, 0};
    for (int t = wave; t < tiles; t += nw) {
        for (int i = 0; 64 * i < 16 * chunks; ++i) {
            const int sl = 64 * i + lane;
            if (sl < 16 * chunks) {
                const int row = ids[t * 16 + sl / chunks], col = 4 * (sl % chunks);
                v4f *p = reinterpret_cast<v4f *>(y + (size_t)row * R + col);
                if (MODE == 0) *p = v4f{1.0f, 2.0f, 3.0f, (float)row};
                else acc = acc + *p;
            }
        }
    }
    if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}
int main()
{
    const int N = 2340496;   // configs[2] vertices, a multiple of 16
    std::vector<int> nat(N), perm(N), win(N);
    for (int i = 0; i < N; ++i) nat[i] = perm[i] = win[i] = i;
    srand(1);
    for (int i = N - 1; i > 0; --i) std::swap(perm[i], perm[rand() % (i + 1)]);          // fully random
    for (int b = 0; b + 32768 <= N; b += 32768)                                            // random inside 32 k-row windows
        for (int i = 32767; i > 0; --i) std::swap(win[b + i], win[b + rand() % (i + 1)]);
    // "xcd": workgroup b runs on XCD b % 8; every XCD gets its own contiguous eighth of the rows and visits it in an order that
    // is random inside 4096-row windows -- adjacent rows are then written by the SAME XCD (one L2) within a short time
    std::vector<int> xcd(N);
    {
        const int nw = 2048 * 4, per = N / 8;
        std::vector<std::vector<int>> reg(8);
        for (int x = 0; x < 8; ++x) {
            reg[x].resize(per);
            for (int i = 0; i < per; ++i) reg[x][i] = x * per + i;
            for (int b = 0; b + 4096 <= per; b += 4096)
                for (int i = 4095; i > 0; --i) std::swap(reg[x][b + i], reg[x][b + rand() % (i + 1)]);
        }
        std::vector<int> pos(8, 0);
        for (int t = 0; t < N / 16; ++t) {
            const int wave = t % nw, x = (wave / 4) % 8;
            for (int k = 0; k < 16; ++k) xcd[t * 16 + k] = pos[x] < per ? reg[x][pos[x]++] : reg[x][per - 1];
        }
    }
    int *d_ids; float *y, *o;
    (void)hipMalloc(&d_ids, 4 * (size_t)N); (void)hipMalloc(&y, 4 * (size_t)N * 96); (void)hipMalloc(&o, 4);
    (void)hipMemset(y, 0, 4 * (size_t)N * 96);
    const char *names[] = {"natural", "random ", "windowed", "xcd-local"};
    const std::vector<int> *orders[] = {&nat, &perm, &win, &xcd};
    for (int R : {64, 72, 96})
        for (int k = 0; k < 4; ++k) {
            (void)hipMemcpy(d_ids, orders[k]->data(), 4 * (size_t)N, hipMemcpyHostToDevice);
            float ms = timeit([&] { hipLaunchKernelGGL((rows_k<0>), dim3(2048), dim3(256), 0, 0, d_ids, y, N, R, o); });
            printf("R %2d %s write: %.3f ms  %.2f TB/s\n", R, names[k], ms, 4.0 * N * R / ms / 1e9);
            ms = timeit([&] { hipLaunchKernelGGL((rows_k<1>), dim3(2048), dim3(256), 0, 0, d_ids, y, N, R, o); });
            printf("R %2d %s read : %.3f ms  %.2f TB/s\n", R, names[k], ms, 4.0 * N * R / ms / 1e9);
        }
    return 0;
}
