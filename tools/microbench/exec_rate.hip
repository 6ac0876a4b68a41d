// exec_rate.hip -- does a vector instruction cost less when only part of the wave is active?  The step kernel's sub-shots are
// wave-uniform float64 arithmetic (one env's values computed in all 64 lanes): if the SIMD skipped the inactive quarters of a
// wave, running that arithmetic with EXEC = lane 0 would be cheaper.  Streams of independent v_fma_f64 / v_add_f32 /
// v_cmp+v_cndmask with 64, 32, 16 and 1 lanes active, four waves a SIMD.
//     hipcc --offload-arch=gfx950 -O3 tools/microbench/exec_rate.hip -o tools/microbench/exec_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                      \
    do {                                                              \
        hipError_t e_ = (x);                                          \
        if (e_ != hipSuccess) {                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));   \
            exit(1);                                                  \
        }                                                             \
    } while (0)

constexpr int UNROLL = 8, REPEAT = 16, ITERS = 2048;

template <int OP>
__global__ __launch_bounds__(256) void stream_kernel(double *sink, double seed, int active, int first_lane) {
    double d[UNROLL];
    float f[UNROLL];
    unsigned u[UNROLL];
    for (int k = 0; k < UNROLL; ++k) {
        d[k] = 1.0 + seed * (k + threadIdx.x);
        f[k] = (float)d[k];
        u[k] = threadIdx.x + k;
    }
    const double b = 1.0 + seed;
    const float bf = (float)b;
    const int lane = threadIdx.x & 63;
    if (lane >= first_lane && lane < first_lane + active) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int r = 0; r < REPEAT; ++r) {
#pragma unroll
                for (int k = 0; k < UNROLL; ++k) {
                    if constexpr (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[k]) : "v"(b));
                    else if constexpr (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(bf));
                    else asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[k]) : "v"(d[k]), "v"(b), "v"(bf) : "vcc");
                }
            }
        }
    }
    double acc = 0;
    for (int k = 0; k < UNROLL; ++k) acc += d[k] + f[k] + u[k];
    if (acc == 123.456) sink[0] = acc;
}

template <int OP>
void run(const char *name, int insts, double *d_sink, FILE *out) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 4;           // four waves a SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int cases[][2] = {{64, 0}, {32, 0}, {32, 32}, {16, 0}, {16, 48}, {1, 0}, {1, 63}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(stream_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_sink, 1e-9, c[0], c[1]);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_sink, 1e-9, c[0], c[1]);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double n_inst = (double)UNROLL * REPEAT * ITERS * insts;
        const double ns = (double)ms * 1e6 / (n_inst * 4);
        fprintf(out, "%-28s lanes %2d from %2d: %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, c[0], c[1], ns, ns * 2.4);
    }
}

int main(int argc, char **argv) {
    double *d_sink;
    CHECK(hipMalloc(&d_sink, 64));
    FILE *out = argc > 1 ? fopen(argv[1], "w") : stdout;
    run<0>("v_fma_f64", 1, d_sink, out);
    run<1>("v_add_f32", 1, d_sink, out);
    run<2>("v_cmp_lt_f64+v_cndmask_b32", 2, d_sink, out);
    if (out != stdout) fclose(out);
    return 0;
}
