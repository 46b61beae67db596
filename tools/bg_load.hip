// bg_load.hip - background kernels for interference experiments (tools/probe_interference.py): what does a resident kernel
// of k_icp_lat's shape take away from the front end - wave slots / registers (kind 0: parked waves), vector issue (kind 1: a
// dependent FMA chain per wave), or memory-pipe slots (kind 2: a dependent chain of L2 loads and stores)?
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/libbg_load.so tools/bg_load.hip
#include <hip/hip_runtime.h>

template <int VGPRS>
__global__ void __launch_bounds__(256) k_bg(int kind, long long ticks, float* buf, int n, float* sink) {
    float keep[VGPRS];   // hold registers: the values stay live across the loop
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) keep[i] = (float)(threadIdx.x + i);
    const long long t0 = wall_clock64();
    float a = (float)threadIdx.x, b = 1.0001f;
    size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) % (size_t)n;
    while (wall_clock64() - t0 < ticks) {
        if (kind == 0) { __builtin_amdgcn_s_sleep(64); }
        else if (kind == 1) {
#pragma unroll
            for (int i = 0; i < 64; ++i) a = a * b + 0.5f;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float v = buf[idx]; buf[idx] = v + 1.f; idx = (idx + 4097) % (size_t)n; a += v; }
        }
    }
    float s = a;
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) s += keep[i];
    if (s == 12345.678f) *sink = s;
}

extern "C" int bg_launch(void* stream, int kind, int n_wg, int vgprs, double ms, float* buf, int n, float* sink) {
    const long long ticks = (long long)(ms * 1e5);   // 100 MHz
    if (vgprs >= 96) hipLaunchKernelGGL(k_bg<100>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, kind, ticks, buf, n, sink);
    else if (vgprs >= 48) hipLaunchKernelGGL(k_bg<40>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, kind, ticks, buf, n, sink);
    else hipLaunchKernelGGL(k_bg<1>, dim3(n_wg), dim3(256), 0, (hipStream_t)stream, kind, ticks, buf, n, sink);
    return (int)hipGetLastError();
}
