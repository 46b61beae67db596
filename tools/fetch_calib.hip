// fetch_calib.hip - what does rocprofv3's FETCH_SIZE read for the access patterns of this library?
// MI355X_MICROARCH.md: "On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
// Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern."  tools/pmc_traffic.py
// doubles FETCH_SIZE for every kernel, which is right for the crop (16 B/lane streaming) - this program checks the others:
//   stream16   every lane 16 B, consecutive lanes consecutive: 1 KiB per wave instruction (the crop, the ICP source points)
//   stream4    every lane 4 B, consecutive (the sort's keys and payloads): 256 B per wave instruction
//   quad64     every QUAD of lanes reads 64 contiguous bytes (4 x 16 B) at a pseudo-random 64-byte-aligned place, every
//              segment of the buffer exactly once (the centroid kernel's run reads: 1-4 points of a run per quad)
//   quad64u    the same at 16-byte alignment (runs start anywhere): segments may straddle a 64-byte boundary
// Every mode reads the whole 1 GiB buffer exactly once (larger than the 256 MiB Infinity Cache).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/fetch_calib tools/fetch_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_stream16(const float4* __restrict__ in, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = in[i]; acc += v.x + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream4(const float* __restrict__ in, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc == 123.456f) out[0] = acc;
}
// segment s (64 bytes) of the buffer is read by quad q = perm(s): an odd multiplier modulo a power of two is a permutation
__global__ void __launch_bounds__(256) k_quad64(const float4* __restrict__ in, size_t nseg, int shift16, float* out) {
    float acc = 0.f;
    const size_t quads = (size_t)gridDim.x * 64;
    for (size_t q = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2); q < nseg; q += quads) {
        const size_t s = (q * 2654435761ull) & (nseg - 1);
        size_t e = s * 4 + (threadIdx.x & 3) + (size_t)shift16;      // float4 index; shift16 = 1..3 moves every segment off the 64-byte grid
        if (e >= nseg * 4) e -= nseg * 4;
        const float4 v = in[e];
        acc += v.x + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const size_t bytes = 1ull << 30, n16 = bytes / 16, n4 = bytes / 4, nseg = bytes / 64;
    void* buf; float* out;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc((void**)&out, 64));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_stream16, dim3(256 * 8), dim3(256), 0, 0, (const float4*)buf, n16, out);
    hipLaunchKernelGGL(k_stream4, dim3(256 * 8), dim3(256), 0, 0, (const float*)buf, n4, out);
    hipLaunchKernelGGL(k_quad64, dim3(256 * 8), dim3(256), 0, 0, (const float4*)buf, nseg, 0, out);
    hipLaunchKernelGGL(k_quad64, dim3(256 * 8), dim3(256), 0, 0, (const float4*)buf, nseg, 1, out);
    CHECK(hipDeviceSynchronize());
    printf("each kernel read %zu bytes exactly once: k_stream16, k_stream4, k_quad64 (aligned), k_quad64 (16 bytes off)\n", bytes);
    return 0;
}
