// valu_rate.hip -- issue cost of the vector instruction classes the step kernel is made of, on one SIMD of gfx950,
// at 1 / 2 / 4 waves per SIMD (the step kernel runs at 4).  Prints cycles per wave-instruction as seen by one wave
// (s_memtime around an unrolled stream of independent instructions) and the SIMD's throughput (wall clock).
// Feeds bench.py's `second_bound` model (profiles/r03_valu_rate.json).
//     hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate   (tools/collect_profiles.sh runs it)
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

constexpr int UNROLL = 8, REPEAT = 16, ITERS = 2048;      // 8 independent chains x 16 per loop trip x 256 trips

enum Op { ADD_F32, FMA_F32, ADD_F64, MUL_F64, FMA_F64, RCP_F64, SQRT_F64, RSQ_F64, CMP_CND_F64, READLANE, MOV_DPP, MIN_U32_DPP, ADD_U32,
          LSHL_B64, BALLOT_F64, CVT_F32_F64, POPC_B64, OP_COUNT };
const char *op_name[OP_COUNT] = {"v_add_f32", "v_fma_f32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64",
                                 "v_cmp_lt_f64+v_cndmask_b32", "v_readlane_b32", "v_mov_b32_dpp", "v_min_u32_dpp", "v_add_u32",
                                 "v_lshlrev_b64", "v_cmp_lt_f64 (ballot to sgpr pair)", "v_cvt_f32_f64", "v_bcnt_u32 x2"};
const int op_insts[OP_COUNT] = {1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 2};

template <int OP>
__device__ __forceinline__ void one(double &d, float &f, unsigned &u, double b, float bf) {
    if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(bf));
    else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(bf));
    else if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(b));
    else if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(b));
    else if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(b));
    else if constexpr (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d));
    else if constexpr (OP == SQRT_F64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d));
    else if constexpr (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d));
    else if constexpr (OP == CMP_CND_F64) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u) : "v"(d), "v"(b), "v"(bf) : "vcc");
    else if constexpr (OP == READLANE) { unsigned s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(u)); asm volatile("" ::"s"(s)); }
    else if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u));
    else if constexpr (OP == MIN_U32_DPP) asm volatile("v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u));
    else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(bf));
    else if constexpr (OP == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d));
    else if constexpr (OP == BALLOT_F64) { unsigned long long s; asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(s) : "v"(d), "v"(b)); asm volatile("" ::"s"(s)); }
    else if constexpr (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(d));
    else if constexpr (OP == POPC_B64) asm volatile("v_bcnt_u32_b32 %0, %1, 0\n\tv_bcnt_u32_b32 %0, %2, %0" : "+v"(u) : "v"(f), "v"(bf));
}

template <int OP>
__global__ __launch_bounds__(256) void stream_kernel(unsigned long long *cycles, double *sink, double seed) {
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
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REPEAT; ++r) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) one<OP>(d[k], f[k], u[k], b, bf);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double acc = 0;
    for (int k = 0; k < UNROLL; ++k) acc += d[k] + f[k] + u[k];
    if (acc == 123.456) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd, unsigned long long *d_cycles, double *d_sink, FILE *json, bool &first) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * waves_per_simd;                  // 256 threads = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink, 1e-9);      // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink, 1e-9);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cycles, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= h.size();
    const double n_inst = (double)UNROLL * REPEAT * ITERS * op_insts[OP];
    // s_memtime ticks at a fixed 100 MHz on gfx950: convert with the wall clock of the launch instead, and report both
    const double per_wave_ticks = mean / n_inst;
    const double ns_per_inst_simd = (double)ms * 1e6 / (n_inst * waves_per_simd);        // one SIMD's time per wave-instruction
    printf("%-36s waves/SIMD %d: %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz); memtime ticks per instruction of one wave %.4f\n",
           op_name[OP], waves_per_simd, ns_per_inst_simd, ns_per_inst_simd * 2.4, per_wave_ticks);
    fprintf(json, "%s\n  {\"op\": \"%s\", \"waves_per_simd\": %d, \"ns_per_wave_instruction_per_simd\": %.4f, \"cycles_at_2p4ghz\": %.3f, \"launch_ms\": %.4f}",
            first ? "" : ",", op_name[OP], waves_per_simd, ns_per_inst_simd, ns_per_inst_simd * 2.4, ms);
    first = false;
}

template <int OP>
void run_all(unsigned long long *c, double *s, FILE *json, bool &first) {
    for (int w : {1, 2, 4}) run<OP>(w, c, s, json, first);
    if constexpr (OP + 1 < OP_COUNT) run_all<OP + 1>(c, s, json, first);
}

int main(int argc, char **argv) {
    unsigned long long *d_cycles;
    double *d_sink;
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * 4 * 4096));
    CHECK(hipMalloc(&d_sink, 64));
    FILE *json = fopen(argc > 1 ? argv[1] : "valu_rate.json", "w");
    fprintf(json, "{\"note\": \"independent instruction streams (8 chains), 256-thread blocks = one wave per SIMD, W blocks per CU; launch time includes ~10 us of launch overhead over %d instructions per wave\", \"rows\": [", UNROLL * REPEAT * ITERS);
    bool first = true;
    run_all<0>(d_cycles, d_sink, json, first);
    fprintf(json, "\n]}\n");
    fclose(json);
    return 0;
}
