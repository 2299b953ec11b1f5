// Probe (GPU box): how long does the hardware take to START the waves of a k_step-shaped launch, and what does it depend on?
// k_step is 1 041 one-wave workgroups with 29 KB of LDS and 172 VGPRs each; inside the deferred loop its waves start over ~9 us
// (profiles/r05_k_step_timeline.txt) and the launch lasts that ramp + its last waves.  Here: a kernel whose waves note their start
// (100 MHz wall clock), where they run (HW_ID, XCC_ID) and then busy-wait `hold` us, launched as G workgroups of W waves with L bytes
// of LDS and (variants) many registers; printed: the spread of the start times, the number of distinct SIMDs, waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/native/dispatch_ramp_probe tools/native/dispatch_ramp_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

template <int REGS>
__global__ void k_probe(unsigned long long* out, int hold_ticks) {
    extern __shared__ unsigned lds[];
    if constexpr (REGS == 176) asm volatile("" ::: "v175");
    if constexpr (REGS == 272) asm volatile("" ::: "v175", "a95");
    const long long t0 = wall_clock64();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = (unsigned long long)t0;
        out[2 * w + 1] = (unsigned long long)((hw & 0x0FFFFFFFu) | ((xcc & 15u) << 28));
    }
    lds[threadIdx.x] = hw;
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(1);
    if (lds[threadIdx.x ^ 1] == 0xFFFFFFFFu) out[0] = 0;
}
__global__ void k_pre(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1; }

template <int REGS>
static int run(const char* name, int G, int W, int ldsb, int hold_us, unsigned long long* d, unsigned* d2) {
    const int waves = G * W;
    std::vector<unsigned long long> h(2 * waves);
    double p1 = 0, p50 = 0, p90 = 0, p99 = 0, last = 0, simds = 0, shared = 0, mx = 0, ev_ms = 0;
    const int reps = 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < reps + 1; r++) {
        hipLaunchKernelGGL(k_pre, dim3(256), dim3(256), 0, 0, d2);                 // (a kernel in front, as the sampler is in the loop)
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe<REGS>, dim3(G), dim3(64 * W), ldsb, 0, d, hold_us * 100);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        if (r == 0) continue;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ev_ms += ms;
        CK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost));
        std::vector<long long> st(waves);
        std::map<unsigned long long, int> per;
        for (int w = 0; w < waves; w++) { st[w] = (long long)h[2 * w]; per[((h[2 * w + 1] >> 28) << 32) | ((h[2 * w + 1] >> 4) & 0xFFFFFF)]++; }
        const long long t0 = *std::min_element(st.begin(), st.end());
        std::sort(st.begin(), st.end());
        auto at = [&](double q) { return (st[(size_t)(q * (waves - 1))] - t0) / 100.0; };
        p1 += at(0.01); p50 += at(0.5); p90 += at(0.9); p99 += at(0.99); last += at(1.0);
        int sh = 0, m = 0;
        for (auto& kv : per) { if (kv.second > 1) sh++; m = std::max(m, kv.second); }
        simds += per.size(); shared += sh; mx = std::max(mx, (double)m);
    }
    printf("%-34s G %5d x %d waves, LDS %6d B, hold %2d us: starts p1 %5.2f p50 %5.2f p90 %5.2f p99 %5.2f last %5.2f us | SIMDs %6.1f, with >1 wave %6.1f, max %d | events %6.2f us\n",
           name, G, W, ldsb, hold_us, p1 / reps, p50 / reps, p90 / reps, p99 / reps, last / reps, simds / reps, shared / reps, (int)mx, ev_ms / reps * 1e3);
    return 0;
}

int main() {
    unsigned long long* d = nullptr; unsigned* d2 = nullptr;
    CK(hipMalloc((void**)&d, sizeof(unsigned long long) * 2 * 65536));
    CK(hipMalloc((void**)&d2, 64));
    for (int hold : { 0, 15 }) {
        run<0>("few regs, no LDS", 1041, 1, 256, hold, d, d2);
        run<0>("few regs, 29 KB", 1041, 1, 29280, hold, d, d2);
        run<176>("176 regs, 29 KB (k_step)", 1041, 1, 29280, hold, d, d2);
        run<272>("272 regs, 29 KB", 1041, 1, 29280, hold, d, d2);
        run<176>("176 regs, 29 KB, 1024 WGs", 1024, 1, 29280, hold, d, d2);
        run<176>("176 regs, 58 KB, 2 waves", 521, 2, 58560, hold, d, d2);
        run<176>("176 regs, 117 KB, 4 waves", 261, 4, 117120, hold, d, d2);
        run<176>("176 regs, 4 waves, 29 KB / WG", 261, 4, 29280, hold, d, d2);
        run<176>("176 regs, 8 waves, 29 KB / WG", 131, 8, 29280, hold, d, d2);
        run<0>("few regs, 256 x 4 waves (sampler)", 256, 4, 256, hold, d, d2);
    }
    return 0;
}
