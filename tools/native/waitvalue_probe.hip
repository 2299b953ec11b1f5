// Probe (GPU box): can a stream wait (hipStreamWaitValue32) on a word that a KERNEL on another stream writes, and what does the hand-over
// cost compared with an event record + wait?  Build: hipcc --offload-arch=gfx950 -O2 -o tools/native/waitvalue_probe tools/native/waitvalue_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_work(unsigned* p, int n) { unsigned v = 0; for (int i = 0; i < n; i++) v = v * 1664525u + 1013904223u + threadIdx.x; if (v == 12345u) p[1] = v; }
__global__ void k_signal(unsigned* sig, unsigned val, unsigned* ctr, int n) {
    unsigned v = 0; for (int i = 0; i < n; i++) v = v * 1664525u + 1013904223u + threadIdx.x; if (v == 12345u) ctr[1] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ctr, 1u) == gridDim.x - 1) { *ctr = 0; __threadfence(); __hip_atomic_store(sig, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}
__global__ void k_after(unsigned* out, unsigned val) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = val; }
__global__ void k_flag(unsigned* sig, unsigned val) { if (threadIdx.x == 0) __hip_atomic_store(sig, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// the scheme: the kernel's first workgroup announces "everything before me on this stream is done" (sig_step = i), every workgroup then
// waits (bounded) until the side stream has finished iteration i - depth (sig_done >= i - depth)
__global__ void k_main(unsigned* sig_step, unsigned* sig_done, unsigned i, unsigned need, unsigned* err, int n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sig_step, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(sig_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
            if (wall_clock64() - t0 > 2000000) { atomicAdd(err, 1u); break; }       // 20 ms at 100 MHz: give up, never hang
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    unsigned v = 0; for (int k = 0; k < n; k++) v = v * 1664525u + 1013904223u + threadIdx.x; if (v == 12345u) err[1] = v;
}
__global__ void k_side(unsigned* out, unsigned val, int n) { unsigned v = 0; for (int k = 0; k < n; k++) v = v * 1664525u + 1013904223u + threadIdx.x; if (v == 12345u) out[3] = v; if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = val; }
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned* sig = nullptr; unsigned* buf = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    CK(hipMalloc((void**)&buf, 64));
    CK(hipMemset(buf, 0, 64));
    CK(hipMemset(sig, 0, 8));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int iters = 2000, work = 2000;
    // (1) baseline: main stream runs k_work back to back; (2) + event record, side stream waits; (3) kernel signals, side stream waits on the value
    unsigned* sig2 = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig2, 8, hipMallocSignalMemory)); CK(hipMemset(sig2, 0, 8));
    for (int mode = 0; mode < 8; mode++) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= iters; i++) {
            if (mode == 7) {      // the full scheme, depth 2: no packet on stream a; the side kernel is ~2/3 of the main kernel
                hipLaunchKernelGGL(k_main, dim3(1024), dim3(64), 0, a, sig, sig2, (unsigned)(i + 700000), (unsigned)(i >= 3 ? i - 2 + 700000 : 0), buf + 8, work);
                if (i >= 2) {
                    CK(hipStreamWaitValue32(b, sig, (unsigned)(i + 700000), hipStreamWaitValueGte, 0xFFFFFFFFu));      // kernel i started => kernel i - 1 is complete
                    hipLaunchKernelGGL(k_side, dim3(2048), dim3(64), 0, b, buf, (unsigned)i, work * 2 / 3);
                    CK(hipStreamWriteValue32(b, sig2, (unsigned)(i - 1 + 700000), 0));
                }
                continue;
            }
            if (mode == 6) {      // the same work with events (record on a, b waits; b records, a waits two iterations later)
                static hipEvent_t evd[3]; static bool init = false;
                if (!init) { for (int k = 0; k < 3; k++) CK(hipEventCreateWithFlags(&evd[k], hipEventDisableTiming)); init = true; }
                if (i >= 3) CK(hipStreamWaitEvent(a, evd[(i - 2) % 3], 0));
                hipLaunchKernelGGL(k_work, dim3(1024), dim3(64), 0, a, buf + 4, work);
                CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0));
                hipLaunchKernelGGL(k_side, dim3(2048), dim3(64), 0, b, buf, (unsigned)i, work * 2 / 3);
                CK(hipEventRecord(evd[i % 3], b));
                continue;
            }
            if (mode == 4) { hipLaunchKernelGGL(k_signal, dim3(1024), dim3(64), 0, a, sig, (unsigned)(i + 400000), buf + 2, work); continue; }
            if (mode == 5) {
                hipLaunchKernelGGL(k_work, dim3(1024), dim3(64), 0, a, buf + 4, work);
                hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, a, sig, (unsigned)(i + 500000));
                CK(hipStreamWaitValue32(b, sig, (unsigned)(i + 500000), hipStreamWaitValueGte, 0xFFFFFFFFu)); hipLaunchKernelGGL(k_after, dim3(1), dim3(64), 0, b, buf, (unsigned)i);
                continue;
            }
            if (mode == 3) hipLaunchKernelGGL(k_signal, dim3(1024), dim3(64), 0, a, sig, (unsigned)(i + mode * 100000), buf + 2, work);
            else hipLaunchKernelGGL(k_work, dim3(1024), dim3(64), 0, a, buf + 4, work);
            if (mode == 1 || mode == 2) CK(hipEventRecord(ev, a));
            if (mode == 2) { CK(hipStreamWaitEvent(b, ev, 0)); hipLaunchKernelGGL(k_after, dim3(1), dim3(64), 0, b, buf, (unsigned)i); }
            if (mode == 3) { CK(hipStreamWaitValue32(b, sig, (unsigned)(i + mode * 100000), hipStreamWaitValueGte, 0xFFFFFFFFu)); hipLaunchKernelGGL(k_after, dim3(1), dim3(64), 0, b, buf, (unsigned)i); }
        }
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        unsigned out = 0; CK(hipMemcpy(&out, buf, 4, hipMemcpyDeviceToHost));
        const char* names[8] = {"kernels back to back", "+ event record per kernel", "+ event record, side stream waits + kernel", "kernel-written value (last-workgroup counter), side stream hipStreamWaitValue32 + kernel",
                                "the last-workgroup counter alone", "+ a one-wave flag kernel, side stream hipStreamWaitValue32 + kernel",
                                "side work per iteration, hand-overs by events (record + wait on the main stream)", "side work per iteration, hand-overs in memory (no packet on the main stream)"};
        unsigned errs = 0; CK(hipMemcpy(&errs, buf + 8, 4, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %.2f us per iteration, side result %u, spin time-outs %u\n", mode, names[mode], us, out, errs);
    }
    return 0;
}
