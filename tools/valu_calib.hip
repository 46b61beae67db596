// valu_calib.hip - what do the SQ counters read when the gfx950 vector pipe is REALLY saturated?
//
// k_icp_pipe's roofline block said "86 % of the VALU issue slots" from
//   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x 4 waves per SIMD,
// while instruction counts give  SQ_INSTS_VALU x 2 cycles / (SIMDs x kernel cycles) = 32-36 %.
// This program runs streams whose instruction count is known exactly (inline asm, nothing for the compiler to
// fold) with the SAME launch shape as k_icp_pipe (one 1024-thread workgroup per CU = 4 waves per SIMD) and with
// 1 and 2 waves per SIMD, times them with s_memtime (shader cycles) and HIP events, and is run a second time under
// `rocprofv3 --pmc` (tools/valu_calib.sh) so that both formulas can be evaluated at true saturation.
//
// Streams:
//   fma_indep   8 independent v_fma_f32 accumulators            (throughput of the f32 pipe)
//   fma_dep     one dependent v_fma_f32 chain                   (latency of a dependent vector instruction)
//   add_indep   8 independent v_add_f32
//   pk_indep    8 independent v_pk_fma_f32 (2 lanes of f32 per VGPR pair)
//   key_min     the 64-bit key minimum of the ICP searches: v_cmp_lt_u64 + 2 v_cndmask per candidate, 4 chains
//   lds_chase   a dependent ds_read_b32 pointer chase with 8 dependent v_fma_f32 between two reads
//               (the shape of a search step: LDS round trip, then arithmetic on what came back)
//   mix_search  per trip: 2 ds_read_b128 (independent addresses), 16 f32 ops, 2 key minimum updates - the grid
//               walk's point loop
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_calib tools/valu_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 64;   // instruction groups per loop trip (the loop overhead - 3 scalar instructions - is < 1 %)

struct Out { unsigned long long cycles; float sink; };

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }   // s_memtime: shader cycles

template <int MODE>
__global__ void __launch_bounds__(1024) k_calib(int trips, Out* out, float seed) {
    __shared__ unsigned s_chain[4096];
    __shared__ float4 s_pts[2048];
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float x = 1.0000001f, y = 1.0e-7f;
    unsigned idx = threadIdx.x;
    unsigned long long k0 = threadIdx.x * 0x9e3779b97f4a7c15ull, k1 = ~k0, k2 = k0 * 3, k3 = k0 * 5, best = ~0ull;
    unsigned long long b1 = ~0ull, b2 = ~0ull, b3 = ~0ull;
    if (MODE == 5 || MODE == 6) {
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_chain[i] = (i * 1237u + 17u) & 4095u;
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_pts[i] = make_float4(i * 0.001f, i * 0.002f, __uint_as_float(i), i * 0.003f);
        __syncthreads();
    }
    const unsigned long long t0 = now();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (MODE == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
            } else if (MODE == 1) {
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(x), "v"(y));
            } else if (MODE == 2) {
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
            } else if (MODE == 3) {
                // 8 packed instructions on four 64-bit register pairs (two chains per pair would alias: use 4 pairs twice)
                float2 p0 = make_float2(a0, a1), p1 = make_float2(a2, a3), p2 = make_float2(a4, a5), p3 = make_float2(a6, a7);
                const float2 xx = make_float2(x, x), yy = make_float2(y, y);
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(xx), "v"(yy));
                a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
            } else if (MODE == 4) {
                // four independent (compare, select low, select high) chains: 12 vector instructions (checked in the ISA;
                // the empty asm keeps the compiler from folding the repeated minimum of an unchanged key)
                asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
                best = k0 < best ? k0 : best;
                b1 = k1 < b1 ? k1 : b1;
                b2 = k2 < b2 ? k2 : b2;
                b3 = k3 < b3 ? k3 : b3;
            } else if (MODE == 5) {
                // dependent LDS round trip, then 8 dependent fma on the result
                unsigned nxt = s_chain[idx & 4095u];
                asm volatile("" : "+v"(nxt));
                float f = __uint_as_float(nxt | 0x3f800000u);
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(f) : "v"(x), "v"(y));
                a0 += f;
                idx = nxt + (__float_as_uint(f) & 1u);
            } else if (MODE == 6) {
                // the grid walk's point loop: two points per trip, canonical dist2 (no fma), two 64-bit key minimum updates
                const unsigned i0 = (idx + u) & 2046u;
                const float4 p = s_pts[i0], q = s_pts[i0 + 1];
                const float dx = __fsub_rn(a0, p.x), dy = __fsub_rn(a1, p.y), dz = __fsub_rn(a2, p.w);
                const float ex = __fsub_rn(a0, q.x), ey = __fsub_rn(a1, q.y), ez = __fsub_rn(a2, q.w);
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const float e = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
                unsigned long long kd = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(p.z);
                unsigned long long ke = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(q.z);
                asm volatile("" : "+v"(kd));
                asm volatile("" : "+v"(ke));
                best = kd < best ? kd : best;
                best = ke < best ? ke : best;
            }
        }
        if (MODE == 6) idx += 2 * UNROLL + (unsigned)(best >> 63);
    }
    const unsigned long long t1 = now();
    float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)) + (float)(best & 1) + (float)(b1 & 1) + (float)(b2 & 1) + (float)(b3 & 1) + (float)idx;
    if ((threadIdx.x & 63) == 0) {
        Out* o = out + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        o->cycles = t1 - t0;
        o->sink = s;
    }
}

struct Mode { const char* name; int valu_per_group; };
static const Mode MODES[] = {
    {"fma_indep", 8}, {"fma_dep", 8}, {"add_indep", 8}, {"pk_indep", 8}, {"key_min", 12}, {"lds_chase", -1}, {"mix_search", -1},
};

template <int MODE>
static void run(int n_wg, int threads, int trips, Out* d_out, std::vector<Out>& h, hipEvent_t e0, hipEvent_t e1, double clock_mhz) {
    const int waves = n_wg * threads / 64;
    hipLaunchKernelGGL(k_calib<MODE>, dim3(n_wg), dim3(threads), 0, 0, 16, d_out, 1.0f);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_calib<MODE>, dim3(n_wg), dim3(threads), 0, 0, trips, d_out, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h.data(), d_out, sizeof(Out) * waves, hipMemcpyDeviceToHost));
    double cyc = 0.0;
    for (int i = 0; i < waves; ++i) cyc += (double)h[i].cycles;
    cyc /= waves;
    const Mode& m = MODES[MODE];
    const double groups = (double)trips * UNROLL;
    const int wps = threads / 256;   // waves per SIMD (one workgroup per CU)
    printf("{\"stream\": \"%s\", \"waves_per_simd\": %d, \"workgroups\": %d, \"trips\": %d, \"mean_wave_cycles\": %.0f, \"event_ms\": %.4f, "
           "\"event_cycles_at_%.0fMHz\": %.0f",
           m.name, wps, n_wg, trips, cyc, ms, clock_mhz, ms * 1e-3 * clock_mhz * 1e6);
    if (m.valu_per_group > 0) {
        const double instr_per_wave = groups * m.valu_per_group;
        printf(", \"valu_per_wave\": %.0f, \"cycles_per_valu_per_wave\": %.3f, \"simd_cycles_per_valu\": %.3f", instr_per_wave, cyc / instr_per_wave,
               cyc / (instr_per_wave * wps));
    } else {
        printf(", \"cycles_per_group\": %.2f, \"simd_cycles_per_group\": %.2f", cyc / groups, cyc / (groups * wps));
    }
    printf("}\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    int trips = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f}\n", prop.gcnArchName, n_cu, mhz);
    Out* d_out;
    CHECK(hipMalloc(&d_out, sizeof(Out) * n_cu * 16));
    std::vector<Out> h(n_cu * 16);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int threads : {256, 512, 1024}) {
        run<0>(n_cu, threads, trips, d_out, h, e0, e1, mhz);
        run<1>(n_cu, threads, trips, d_out, h, e0, e1, mhz);
        run<2>(n_cu, threads, trips, d_out, h, e0, e1, mhz);
        run<3>(n_cu, threads, trips, d_out, h, e0, e1, mhz);
        run<4>(n_cu, threads, trips, d_out, h, e0, e1, mhz);
        run<5>(n_cu, threads, trips / 4, d_out, h, e0, e1, mhz);
        run<6>(n_cu, threads, trips / 4, d_out, h, e0, e1, mhz);
    }
    return 0;
}
