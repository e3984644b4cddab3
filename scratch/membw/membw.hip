// Memory-system yardsticks for the roofline of the fused kernels: what a write-only, a read-only and a copy
// stream of the SAME footprint (320 MB = one H matrix of the bench batch) reach on this part.
// hipcc --offload-arch=gfx950 -O3 membw.hip -o membw ; ./membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

__global__ void fill_k(v4 *dst, size_t n, float x) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    const v4 v = {x, x + 1.f, x + 2.f, x + 3.f};
    for (; i < n; i += st) dst[i] = v;
}
__global__ void fill_nt_k(v4 *dst, size_t n, float x) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    const v4 v = {x, x + 1.f, x + 2.f, x + 3.f};
    for (; i < n; i += st) __builtin_nontemporal_store(v, dst + i);
}
__global__ void read_k(const v4 *src, size_t n, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    v4 a = {0, 0, 0, 0};
    for (; i < n; i += st) a += src[i];
    if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;   // never true: keeps the loads
}
__global__ void copy_k(const v4 *src, v4 *dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) dst[i] = src[i];
}
// contiguous 64 KB pieces per workgroup step (the tile shape of the fused kernels: one (graph, slice) tile)
__global__ void fill_tiles_k(v4 *dst, size_t n_tiles, float x) {
    const v4 v = {x, x + 1.f, x + 2.f, x + 3.f};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        v4 *p = dst + t * 4096;   // 4096 x 16 B = 64 KB
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) p[i] = v;
    }
}

int main() {
    const size_t bytes = 320u << 20, n = bytes / 16;
    v4 *a, *b; float *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { int grid, block; };
    const Cfg cfgs[] = {{256, 1024}, {512, 1024}, {1024, 1024}, {2048, 512}, {8192, 256}, {65536, 256}};
    for (const Cfg &c : cfgs) {
        for (int which = 0; which < 5; ++which) {
            std::vector<float> ms;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0));
                switch (which) {
                case 0: fill_k<<<c.grid, c.block>>>(a, n, (float)rep); break;
                case 1: fill_nt_k<<<c.grid, c.block>>>(a, n, (float)rep); break;
                case 2: read_k<<<c.grid, c.block>>>(a, n, o); break;
                case 3: copy_k<<<c.grid, c.block>>>(a, b, n); break;
                case 4: fill_tiles_k<<<c.grid, c.block>>>(a, n / 4096, (float)rep); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            const double med = ms[ms.size() / 2], moved = (which == 3 ? 2.0 : 1.0) * bytes;
            const char *names[] = {"fill", "fill_nt", "read", "copy", "fill_tiles64K"};
            printf("%-14s grid %6d x %4d : %7.1f us  %6.2f TB/s\n", names[which], c.grid, c.block, med * 1e3, moved / med / 1e9);
        }
    }
    return 0;
}
