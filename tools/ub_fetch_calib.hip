// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library uses
// (MI355X_MICROARCH.md: FETCH_SIZE is known to report 1/2 of the bytes of a 16 B/lane streaming read; other widths
// and WRITE_SIZE are "uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel moves exactly NBYTES (1 GiB, four times the Infinity Cache) once; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// and divide:  tools/fetch_calib_summary.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t NBYTES = 1ull << 30;

template <typename T>
__global__ void read_k(const T* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = p[i];
        const unsigned* w = reinterpret_cast<const unsigned*>(&v);
        for (unsigned j = 0; j < (sizeof(T) + 3) / 4; ++j) acc ^= sizeof(T) >= 4 ? w[j] : (unsigned)*reinterpret_cast<const uint16_t*>(&v);
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <typename T>
__global__ void write_k(T* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v;
        unsigned char* b = reinterpret_cast<unsigned char*>(&v);
        for (unsigned j = 0; j < sizeof(T); ++j) b[j] = (unsigned char)(i + j);
        p[i] = v;
    }
}
// the stem's pattern: a 16x16-output tile reads a 35-row x 29-dword patch of a uint8 HWC image (halo re-reads included)
__global__ void read_stem_patch(const uint32_t* __restrict__ img, int H, int W, unsigned* sink) {
    const int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
    const int d = threadIdx.x % 29, rg = threadIdx.x / 29;
    unsigned acc = 0;
    if (rg < 8)
        for (int r = rg; r < 35; r += 8) {
            const int y = min(max(ty * 32 - 2 + r, 0), H - 1);
            const long long boff = (long long)tx * 96 - 6 - 2 + 4 * d;
            const long long cb = min(max(boff, 0ll), (long long)W * 3 - 4);
            acc ^= img[(((size_t)b * H + y) * W * 3 + cb) / 4];
        }
    if (acc == 0x12345678u) *sink = acc;
}
struct alignas(16) V16 { unsigned x[4]; };
struct alignas(8) V8 { unsigned x[2]; };

int main() {
    void* buf; unsigned* sink;
    hipMalloc(&buf, NBYTES); hipMalloc(&sink, 4);
    hipMemset(buf, 1, NBYTES);
    hipDeviceSynchronize();
    const dim3 g(256 * 16), b(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_k<V16>, g, b, 0, 0, (const V16*)buf, NBYTES / 16, sink);
        hipLaunchKernelGGL(read_k<V8>, g, b, 0, 0, (const V8*)buf, NBYTES / 8, sink);
        hipLaunchKernelGGL(read_k<unsigned>, g, b, 0, 0, (const unsigned*)buf, NBYTES / 4, sink);
        hipLaunchKernelGGL(read_k<uint16_t>, g, b, 0, 0, (const uint16_t*)buf, NBYTES / 2, sink);
        hipLaunchKernelGGL(write_k<V16>, g, b, 0, 0, (V16*)buf, NBYTES / 16);
        hipLaunchKernelGGL(write_k<V8>, g, b, 0, 0, (V8*)buf, NBYTES / 8);
        hipLaunchKernelGGL(write_k<unsigned>, g, b, 0, 0, (unsigned*)buf, NBYTES / 4);
        hipLaunchKernelGGL(write_k<uint16_t>, g, b, 0, 0, (uint16_t*)buf, NBYTES / 2);
        // 64 images of 640x640x3 bytes = 78.6 MB unique; the tiles' patches total 64*20*20*35*116 B = 103.9 MB
        hipLaunchKernelGGL(read_stem_patch, dim3(20, 20, 64), dim3(256), 0, 0, (const uint32_t*)buf, 640, 640, sink);
        hipDeviceSynchronize();
    }
    printf("moved %zu bytes per kernel; stem patch: unique 78643200 B, requested 103936000 B\n", NBYTES);
    return 0;
}
