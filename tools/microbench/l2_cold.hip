// l2_cold.hip -- does a table stay in the XCD's L2 from one kernel launch to the next?  A dependent-load chase over a
// 2 MB table (one 128-byte line per hop), one wave per workgroup, 8 workgroups (one per XCD as a rule):
//   launch 1: pass A (lines never touched before) and pass B (the same lines again, same launch);
//   launch 2, same stream, right behind: pass A again (what the previous launch left in L2 -- or not), pass B.
// Prints ns per hop for each.  If launch 2's pass A costs what launch 1's pass A cost, every launch starts with a cold L2
// for data other launches read (the per-XCD L2s are invalidated at the kernel boundary) and a table read through a chain
// of dependent loads pays the fabric / Infinity-Cache latency on first touch in EVERY launch.
//     hipcc --offload-arch=gfx950 -O3 tools/microbench/l2_cold.hip -o tools/microbench/l2_cold
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int LINE_INTS = 32;       // 128 bytes
constexpr int HOPS = 2048;

__global__ void chase(const int *table, int n_lines, long long *out, int which) {
    if (threadIdx.x != 0) return;
    const int idx = (blockIdx.x * 977) % n_lines;
    for (int pass = 0; pass < 2; ++pass) {
        int cur = idx;
        const long long t0 = wall_clock64();
        for (int h = 0; h < HOPS; ++h) cur = table[(size_t)cur * LINE_INTS];
        const long long t1 = wall_clock64();
        out[(which * 2 + pass) * gridDim.x + blockIdx.x] = (t1 - t0) + (cur == -1);
    }
}

int main() {
    const int n_lines = 16384;                      // 2 MB
    std::vector<int> h((size_t)n_lines * LINE_INTS, 0);
    // one cycle through all lines in a scrambled order (a fixed odd stride modulo a power of two)
    for (int i = 0; i < n_lines; ++i) h[(size_t)i * LINE_INTS] = (int)(((long long)i * 6151 + 3571) % n_lines);
    int *d;
    long long *out;
    const int blocks = 8, launches = 4;
    CHECK(hipMalloc(&d, h.size() * sizeof(int)));
    CHECK(hipMalloc(&out, sizeof(long long) * 2 * launches * blocks));
    CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, n_lines, out, l);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> r(2 * launches * blocks);
    CHECK(hipMemcpy(r.data(), out, r.size() * sizeof(long long), hipMemcpyDeviceToHost));
    printf("ns per dependent hop (128-byte lines of a 2 MB table, %d hops, one wave per workgroup), by workgroup:\n", HOPS);
    for (int l = 0; l < launches; ++l)
        for (int p = 0; p < 2; ++p) {
            printf("launch %d pass %c:", l + 1, 'A' + p);
            double sum = 0;
            for (int b = 0; b < blocks; ++b) {
                const double ns = 10.0 * r[(l * 2 + p) * blocks + b] / HOPS;      // wall_clock64: 100 MHz
                printf(" %6.0f", ns);
                sum += ns;
            }
            printf("   mean %.0f\n", sum / blocks);
        }
    return 0;
}
