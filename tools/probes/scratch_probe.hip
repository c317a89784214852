// Does per-lane scratch (private segment) memory keep its contents when every wave slot of the chip is in use?
// Each lane writes a signature into a small private array that the compiler must keep in scratch memory
// (volatile), waits a little, and reads it back.  Blocks of 256 threads with 20 KB of LDS, eight per CU,
// __launch_bounds__(256, 8): the shape of the library's traversal kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/scratch_probe.hip -o gpurun_out/scratch_probe && gpurun_out/scratch_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(2);                                                                \
        }                                                                           \
    } while (0)

template <int N>
__global__ __launch_bounds__(256, 8) void probe(unsigned long long* errs, int iters, int hold) {
    __shared__ uint32_t pad[5120];  // 20 KB
    volatile uint32_t a[N];
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    pad[threadIdx.x] = g;
    unsigned long long bad = 0;
    for (int it = 0; it < iters; it++) {
        for (int k = 0; k < N; k++) a[k] = g * 1664525u + (uint32_t)k * 1013904223u + (uint32_t)it;
        for (int h = 0; h < hold; h++) pad[(threadIdx.x + h) & 5119] += g;  // (time passes, LDS traffic as in the real kernels)
        for (int k = 0; k < N; k++)
            if (a[k] != g * 1664525u + (uint32_t)k * 1013904223u + (uint32_t)it) bad++;
    }
    if (bad) atomicAdd(errs, bad);
    if (pad[threadIdx.x] == 0xdeadbeefu) atomicAdd(errs + 1, 1ull);
}

template <int N>
void run(int blocks_per_cu, int cus, unsigned long long* d_errs) {
    CHECK(hipMemset(d_errs, 0, 16));
    hipLaunchKernelGGL(probe<N>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, d_errs, 200, 50);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[2];
    CHECK(hipMemcpy(h, d_errs, 16, hipMemcpyDeviceToHost));
    printf("private array of %2d dwords (%3d B per lane), %d blocks per CU: %llu mismatches\n", N, 4 * N, blocks_per_cu, h[0]);
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.gcnArchName, cus);
    unsigned long long* d_errs;
    CHECK(hipMalloc(&d_errs, 16));
    for (int b : {1, 4, 8, 16}) {
        run<5>(b, cus, d_errs);
        run<6>(b, cus, d_errs);
        run<7>(b, cus, d_errs);
        run<14>(b, cus, d_errs);
        run<64>(b, cus, d_errs);
    }
    return 0;
}
