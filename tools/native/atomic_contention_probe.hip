// Probe (GPU box): what do the sort's range reservations cost?  Every block of the sampler (256 blocks of 256 threads) reserves its range in
// each of the 18 bins with one device-scope atomicAdd per (block, bin) on 18 CONSECUTIVE words - one 128-byte line, one L2 channel; the
// fused-sampling k_step does the same from 1 041 one-wave workgroups.  Here: B blocks, each block's threads 0..17 add to counter[k * stride]
// and wait for the result (as sort_append does); stride 1 (one line) against stride 32 (a line per bin).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/native/atomic_contention_probe tools/native/atomic_contention_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_res(unsigned* ctr, int stride, int nb, unsigned* out) {
    __shared__ unsigned base[32];
    if (threadIdx.x < nb) base[threadIdx.x] = atomicAdd(&ctr[threadIdx.x * stride], 3u);
    __syncthreads();
    if (threadIdx.x == 0 && base[threadIdx.x % nb] == 0xFFFFFFFFu) out[0] = 1;
}
__global__ void k_none(unsigned* out) { if (threadIdx.x == 1024) out[0] = 1; }
static int run(const char* name, int blocks, int threads, int stride, int nb, unsigned* ctr, unsigned* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 12; r++) {
        CK(hipMemsetAsync(ctr, 0, 4096 * 4, 0));
        hipLaunchKernelGGL(k_none, dim3(256), dim3(256), 0, 0, out);
        CK(hipEventRecord(e0, 0));
        if (nb > 0) hipLaunchKernelGGL(k_res, dim3(blocks), dim3(threads), 0, 0, ctr, stride, nb, out);
        else hipLaunchKernelGGL(k_none, dim3(blocks), dim3(threads), 0, 0, out);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-52s %5d blocks x %4d threads, %2d bins, stride %2d words: mean %6.2f us  best %6.2f us\n", name, blocks, threads, nb, stride, sum / 10 * 1e3, best * 1e3);
    return 0;
}
int main() {
    unsigned *ctr, *out;
    CK(hipMalloc((void**)&ctr, 4096 * 4)); CK(hipMalloc((void**)&out, 64));
    run("empty kernel", 256, 256, 1, 0, ctr, out);
    run("sampler-shaped, bins in one line", 256, 256, 1, 18, ctr, out);
    run("sampler-shaped, a line per bin", 256, 256, 32, 18, ctr, out);
    run("sampler-shaped, 1 024-thread blocks, one line", 64, 1024, 1, 18, ctr, out);
    run("k_step-shaped (fused sampling), one line", 1041, 64, 1, 18, ctr, out);
    run("k_step-shaped (fused sampling), a line per bin", 1041, 64, 32, 18, ctr, out);
    run("k_step-shaped, ONE bin", 1041, 64, 32, 1, ctr, out);
    return 0;
}
