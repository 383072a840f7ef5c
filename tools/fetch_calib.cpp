// GPU box, under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and a second pass with WRITE_SIZE): what the L2's memory-side counters report
// for a KNOWN number of bytes in the access patterns of the leaf kernel - the guide calibrates FETCH_SIZE (x 2) for 16 B / lane streaming
// reads only.  Each kernel reads (or writes) BYTES bytes exactly once, far more than the 256 MB of Infinity Cache:
//   calib_read16   16 B per lane, consecutive lanes consecutive (the guide's pattern)
//   calib_read8     8 B per lane: a wave's load = 512 consecutive bytes (a leaf's tape words, lane = op)
//   calib_read4     4 B per lane at an 8-byte stride: a wave's load touches 512 consecutive bytes, half of them wanted (the depth half of
//                  the 64-bit z-buffer words of a footprint row)
//   calib_entry16  16 B per lane, lanes 4 KB apart (a column of leaf-table entries, lane = layer): 64 lines per wave-load
//   calib_atomic8  64-bit atomic max per lane, consecutive (the z-buffer update)
//   calib_write16  16 B per lane streaming writes (the image)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/_bin/fetch_calib tools/fetch_calib.cpp ; run: tools/fetch_calib (prints the byte counts)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
static const size_t BYTES = (size_t)2 << 30;
__global__ void calib_read16(const uint4* p, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_read8(const uint2* p, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint2 v = p[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_read4(const uint32_t* p, size_t n, uint32_t* sink) {       // n = number of 8-byte words; reads the upper half of each
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[2 * i + 1];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_entry16(const uint4* p, size_t n, uint32_t* sink) {       // element (lane, k) at lane * 256 + k: lanes 4 KB apart
    uint32_t acc = 0;
    const size_t lane = threadIdx.x & 63, wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t c = wave; c * 64 * 256 < n; c += waves)           // chunk c: 64 lanes x 256 entries
        for (size_t k = 0; k < 256; k++) { uint4 v = p[c * 64 * 256 + lane * 256 + k]; acc ^= v.x ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_atomic8(unsigned long long* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicMax(&p[i], (unsigned long long)i);
}
__global__ void calib_write16(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
int main() {
    void* buf; uint32_t* sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("no memory\n"); return 1; }
    hipMemset(buf, 1, BYTES);
    hipDeviceSynchronize();
    const dim3 g(256 * 16), b(256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(calib_read16, g, b, 0, 0, (const uint4*)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(calib_read8, g, b, 0, 0, (const uint2*)buf, BYTES / 8, sink);
        hipLaunchKernelGGL(calib_read4, g, b, 0, 0, (const uint32_t*)buf, BYTES / 8, sink);
        hipLaunchKernelGGL(calib_entry16, g, b, 0, 0, (const uint4*)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(calib_atomic8, g, b, 0, 0, (unsigned long long*)buf, BYTES / 8);
        hipLaunchKernelGGL(calib_write16, g, b, 0, 0, (uint4*)buf, BYTES / 16);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel launch: read16 %zu read8 %zu read4 %zu wanted of %zu touched entry16 %zu atomic8 %zu write16 %zu\n", BYTES, BYTES, BYTES / 2, BYTES, BYTES, BYTES, BYTES);
    return 0;
}
